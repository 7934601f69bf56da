#!/bin/bash
# disassembly of one kernel instance: tools/isa_of.sh <object> <mangled-name substring>  -> stdout
T=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin "$1"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/k.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn $T/k.co | awk -v pat="$2" 'index($0, pat) && /^[0-9a-f]+ </ {on=1} on && /^$/ {exit} on {sub(/\/\/.*/, ""); print}'
rm -rf $T

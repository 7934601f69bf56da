#!/bin/bash
# a library variant whose second-generation 128-tap kernel is built with extra flags: tools/exp/variant.sh <name> [-DMFM3_...=1 ...]
# -> tools/exp/libexp_<name>.so (select with MFM_LIB=...; tools/exp/run.sh times several on one box)
cd "$(dirname "$0")/../.."
name=$1; shift
B=tsl-sdr_amd/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -c -o tools/exp/v3_$name.o tsl-sdr_amd/csrc/mfm_kernel_v3.hip || exit 1
objs=$(ls $B/*.o | grep -v mfm_kernel_v3.o | grep -v -E "multifm_main|decoder_main")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/libexp_$name.so $objs tools/exp/v3_$name.o -lm -lpthread -ldl
echo built tools/exp/libexp_$name.so

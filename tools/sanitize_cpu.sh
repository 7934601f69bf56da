#!/bin/bash
# AddressSanitizer + UBSan over the CPU-side C code (the host library and the oracle) while the CPU test-suite runs.
# GPU sanitizers are not available on the pool; the kernels are covered by the parity tests instead.
#   bash tools/sanitize_cpu.sh
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -O1 -g"
cd "$ROOT/tsl-sdr_amd/host"
for f in mfm_tsl mfm_config mfm_receiver mfm_file_if mfm_rtl_sdr_if mfm_pager_pocsag mfm_pager_flex; do
  gcc -std=gnu11 $SAN -fPIC -D_GNU_SOURCE -I. -I../../include -c -o $T/$f.o $f.c
done
gcc -shared $SAN -o $T/libmfm_host.so $T/*.o -L.. -lmultifm_hip -Wl,-rpath,$ROOT/tsl-sdr_amd -lpthread -lm -ldl
cd "$ROOT/oracle"
gcc -std=gnu11 $SAN -march=x86-64-v3 -ffp-contract=off -fwrapv -fPIC -D_GNU_SOURCE -shared -o $T/liboracle.so \
    mfm_oracle.c pocsag_oracle.c f32_oracle.c flex_oracle.c -lm -lpthread
cp "$ROOT/tsl-sdr_amd/host/libmfm_host.so" $T/host.orig; cp "$ROOT/oracle/liboracle.so" $T/oracle.orig
trap 'cp $T/host.orig "$ROOT/tsl-sdr_amd/host/libmfm_host.so"; cp $T/oracle.orig "$ROOT/oracle/liboracle.so"' EXIT
cp $T/libmfm_host.so "$ROOT/tsl-sdr_amd/host/libmfm_host.so"; cp $T/liboracle.so "$ROOT/oracle/liboracle.so"
cd "$ROOT"
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
  UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 python -m pytest tests -x -q -m "not gpu"
# ThreadSanitizer: the receiver's threads against the stalling device double (the same build tests/test_host.py runs)
cd "$ROOT"
H=tsl-sdr_amd/host
gcc -std=gnu11 -O1 -g -fsanitize=thread -D_GNU_SOURCE -I$H -Iinclude -o $T/stall_tsan $H/mfm_tsl.c $H/mfm_config.c $H/mfm_receiver.c \
    tests/hoststub/stub_group.c tests/hoststub/stall_main.c -lpthread -lm
: > $T/pcm.out
TSAN_OPTIONS="halt_on_error=1 report_signal_unsafe=0" $T/stall_tsan $T/pcm.out
echo "ThreadSanitizer: clean"

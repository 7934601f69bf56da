#!/bin/bash
# builds libexp_<X>.so for each experiment mask given on the command line (scratch; wrong results on purpose)
cd "$(dirname "$0")/../.."
B=tsl-sdr_amd/build
for x in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DX=$x ${EXTRA} -c -o tools/exp/k_$x.o tools/exp/${SRC:-k3_exp}.hip &
done
wait
for x in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/libexp_$x.so $B/mfm_kernel.o $B/${KEEP:-mfm_kernel_mfma}.o tools/exp/k_$x.o $B/mfm_resampler.o $B/mfm_f32.o $B/mfm_mm.o $B/mfm_pocsag.o $B/mfm_engine.o $B/mfm_group.o $B/mfm_taps.o -lm -lpthread -ldl
done

#!/bin/bash
cd "$(dirname "$0")/../.."
B=tsl-sdr_amd/build
for x in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Itsl-sdr_amd/csrc -DX=$x -c -o tools/exp/f32k_$x.o tools/exp/f32_knock.hip &
done
wait
for x in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/libexp_f$x.so $B/mfm_kernel.o $B/mfm_kernel_mfma.o $B/mfm_kernel_v3.o $B/mfm_resampler.o tools/exp/f32k_$x.o $B/mfm_mm.o $B/mfm_pocsag.o $B/mfm_flex.o $B/mfm_engine.o $B/mfm_group.o $B/mfm_taps.o -lm -lpthread -ldl
done

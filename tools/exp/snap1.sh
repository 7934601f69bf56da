#!/bin/bash
# snapshot the current first-generation kernel source as library variant <name>: tools/exp/snap1.sh <name> [extra hipcc flags]
cd "$(dirname "$0")/../.."
name=$1; shift
B=tsl-sdr_amd/build
sed -e 's|#include "mfm_kernel.h"|#include "../../tsl-sdr_amd/csrc/mfm_kernel.h"|' -e 's|#include "mfm_numerics.h"|#include "../../tsl-sdr_amd/csrc/mfm_numerics.h"|' tsl-sdr_amd/csrc/mfm_kernel_mfma.hip > tools/exp/snap_$name.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -c -o tools/exp/snap_$name.o tools/exp/snap_$name.hip || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/libexp_$name.so $B/mfm_kernel.o tools/exp/snap_$name.o $B/mfm_kernel_v3.o $B/mfm_resampler.o $B/mfm_f32.o $B/mfm_mm.o $B/mfm_pocsag.o $B/mfm_flex.o $B/mfm_engine.o $B/mfm_group.o $B/mfm_taps.o -lm -lpthread -ldl
echo built tools/exp/libexp_$name.so

#!/bin/bash
# library variant with its own engine object: tools/exp/snapeng.sh <name> [engine hipcc flags]
cd "$(dirname "$0")/../.."
name=$1; shift; B=tsl-sdr_amd/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude "$@" -c -o tools/exp/eng_$name.o tsl-sdr_amd/csrc/mfm_engine.hip || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/libexp_$name.so $B/mfm_kernel.o $B/mfm_kernel_mfma.o $B/mfm_kernel_v3.o $B/mfm_resampler.o $B/mfm_f32.o $B/mfm_mm.o $B/mfm_pocsag.o $B/mfm_flex.o tools/exp/eng_$name.o $B/mfm_group.o $B/mfm_taps.o -lm -lpthread -ldl
echo built tools/exp/libexp_$name.so

#!/bin/bash
# a library variant whose long-filter kernel (one k-step count) is built with extra flags:
#   tools/exp/variant_l.sh <name> <kq> [-DMFM3L_...=1 ...]  -> tools/exp/libexp_<name>.so (tools/exp/run.sh times several on one box)
cd "$(dirname "$0")/../.."
name=$1; kq=$2; shift; shift
B=tsl-sdr_amd/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DMFM3L_ONLY_KQ=$kq "$@" -c -o tools/exp/v3l_$name.o tsl-sdr_amd/csrc/mfm_kernel_v3l.hip || exit 1
objs=$(ls $B/*.o | grep -v mfm_kernel_v3l_kq$kq.o | grep -v -E "multifm_main|decoder_main")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/libexp_$name.so $objs tools/exp/v3l_$name.o -lm -lpthread -ldl
echo built tools/exp/libexp_$name.so

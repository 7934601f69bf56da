#!/bin/bash
# a library variant whose long-filter kernel is rebuilt for SEVERAL k-step counts with extra flags (side by side):
#   tools/exp/variant_lm.sh <name> "<kq> <kq> ..." [-DMFM3L_...=1 ...]  -> tools/exp/libexp_<name>.so
cd "$(dirname "$0")/../.."
name=$1; kqs=$2; shift; shift
B=tsl-sdr_amd/build
pids=""
for kq in $kqs; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DMFM3L_ONLY_KQ=$kq "$@" -c -o tools/exp/v3l_${name}_kq$kq.o tsl-sdr_amd/csrc/mfm_kernel_v3l.hip &
  pids="$pids $!"
done
for p in $pids; do wait $p || exit 1; done
objs=$(ls $B/*.o | grep -v -E "multifm_main|decoder_main")
for kq in $kqs; do objs=$(echo "$objs" | grep -v mfm_kernel_v3l_kq$kq.o); objs="$objs tools/exp/v3l_${name}_kq$kq.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/libexp_$name.so $objs -lm -lpthread -ldl
echo built tools/exp/libexp_$name.so

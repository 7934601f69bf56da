#!/bin/bash
# where does the long-filter kernel's idle time go?  knock-out builds (tools/exp/variant_l.sh k<N> 16 -DMFM3L_KNOCK=N) on configs[4]'s share
export AB_REPS=2 BENCH_ARGS="--config cfg5_airspy --channels-per-gpu 256"
bash tools/exp/run.sh base k1 k2 k3 k4 k7 k8 k16 2>&1 | tee gpurun_out/exp/knock_cfg5.txt

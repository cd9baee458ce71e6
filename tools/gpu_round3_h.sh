#!/bin/bash
set -u
out=gpurun_out/r3h; mkdir -p $out
tools/gpu_ab.sh r3h/ab atrium ":" ":HIPR_WIDE8_LAYOUT=bfs" ":HIPR_REFILL_BELOW=32" ":HIPR_REFILL_BELOW=48" ":HIPR_BLOCKS_PER_CU=10" ":HIPR_BLOCKS_PER_CU=8" 2>&1 | tee $out/ab.txt

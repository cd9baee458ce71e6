#!/bin/bash
# A/B: waves of a thin launch filled to min_fill lanes only (HIPR_MIN_FILL=64: whole waves, the behaviour before)
set -u
out=gpurun_out/r3q; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
for scene in material glass atrium; do tools/gpu_ab.sh r3q/ab_$scene $scene ":HIPR_MIN_FILL=64" ":HIPR_MIN_FILL=16" ":HIPR_MIN_FILL=4" ":HIPR_MIN_FILL=1" 2>&1 | tee -a $out/ab.txt; done
BENCH_ARGS="--spp-per-pass 1 --steps 64 --warmup 8" tools/gpu_ab.sh r3q/ab_1spp atrium ":HIPR_MIN_FILL=64" ":HIPR_MIN_FILL=16" ":HIPR_MIN_FILL=4" 2>&1 | tee -a $out/ab.txt
BENCH_ARGS="--spp-per-pass 4" tools/gpu_ab.sh r3q/ab_mat4 material ":HIPR_MIN_FILL=64" ":HIPR_MIN_FILL=4" 2>&1 | tee -a $out/ab.txt

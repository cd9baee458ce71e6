#!/bin/bash
set -u
out=gpurun_out/r3n; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_coverage.py tests/test_gpu_bench.py -x -q 2>&1 | tail -3
for scene in material atrium cornell_diffuse; do tools/gpu_ab.sh r3n/ab_$scene $scene ":HIPR_PIPELINE_PASSES=0" ":HIPR_PIPELINE_PASSES=1" 2>&1 | tee -a $out/ab.txt; done
BENCH_ARGS="--spp-per-pass 1 --steps 64 --warmup 8" tools/gpu_ab.sh r3n/ab_1spp atrium ":HIPR_PIPELINE_PASSES=0" ":HIPR_PIPELINE_PASSES=1" 2>&1 | tee -a $out/ab.txt
BENCH_ARGS="--spp-per-pass 4" tools/gpu_ab.sh r3n/ab_mat4 material ":HIPR_PIPELINE_PASSES=0" ":HIPR_PIPELINE_PASSES=1" 2>&1 | tee -a $out/ab.txt

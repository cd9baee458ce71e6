#!/bin/bash
# A/B: leaf records set aside during node iterations (base) against strict item order (nopark)
set -u
out=gpurun_out/r3x; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_coverage.py tests/test_gpu_bench.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -5
for scene in atrium material; do tools/gpu_ab.sh r3x/ab_$scene $scene "nopark:" ":" "nopark:" ":" 2>&1 | tee -a $out/ab.txt; done
BENCH_ARGS="--atrium-triangles 10000000 --width 3840 --height 2160 --spp-per-pass 8 --steps 2" tools/gpu_ab.sh r3x/ab_10m atrium "nopark:" ":" 2>&1 | tee -a $out/ab.txt
BENCH_ARGS="--spp-per-pass 1 --steps 64 --warmup 8" tools/gpu_ab.sh r3x/ab_1spp atrium "nopark:" ":" 2>&1 | tee -a $out/ab.txt

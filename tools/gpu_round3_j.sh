#!/bin/bash
set -u
out=gpurun_out/r3j; mkdir -p $out
for scene in atrium material cornell cornell_diffuse; do
    tools/gpu_ab.sh r3j/ab_$scene $scene ":HIPR_SHADE_ORDERED=0" ":HIPR_SHADE_ORDERED=1" 2>&1 | tee -a $out/ab.txt
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_coverage.py -x -q 2>&1 | tail -3

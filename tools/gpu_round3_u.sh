#!/bin/bash
# A/B: the shade kernel whole against its two halves (next event estimation | the rest), the halves compiled for 3, 4 and 5 waves per SIMD
set -u
out=gpurun_out/r3u; mkdir -p $out
HIPR_SHADE_SPLIT=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_coverage.py -x -q 2>&1 | grep -E "passed|failed" | tail -3
for scene in atrium material cornell_diffuse cornell; do tools/gpu_ab.sh r3u/ab_$scene $scene ":HIPR_SHADE_SPLIT=0" ":HIPR_SHADE_SPLIT=1" "split3:HIPR_SHADE_SPLIT=1" "split5:HIPR_SHADE_SPLIT=1" 2>&1 | tee -a $out/ab.txt; done

#!/bin/bash
# A/B: next event estimation with ONE evaluation of the candidate when the scene's only light is directional (base) against one per candidate (noshared)
set -u
out=gpurun_out/r3w; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_coverage.py -x -q 2>&1 | grep -E "passed|failed" | tail -3
for scene in atrium material cornell_diffuse cornell; do tools/gpu_ab.sh r3w/ab_$scene $scene "noshared:" ":" "noshared:" ":" 2>&1 | tee -a $out/ab.txt; done

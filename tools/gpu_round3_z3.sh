#!/bin/bash
# A/B: no BSDF sample at a path's last hit (base) against the sample drawn and dropped (before)
set -u
out=gpurun_out/r3z3; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED" | tail -6
for scene in atrium material cornell_diffuse cornell; do tools/gpu_ab.sh r3z3/ab_$scene $scene "before:" ":" "before:" ":" 2>&1 | tee -a $out/ab.txt; done

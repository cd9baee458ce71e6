#!/bin/bash
set -u
out=gpurun_out/r3z2; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED" | tail -6
for scene in material atrium cornell_diffuse; do tools/gpu_ab.sh r3z2/ab_$scene $scene "nocullcode:" ":" "nocullcode:" ":" 2>&1 | tee -a $out/ab.txt; done

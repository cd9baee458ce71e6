#!/bin/bash
# round 4, batch f: surface hits listed in two classes (plain / coated): parity (frames bit-identical with the classes off), then A/B
set -u
out=gpurun_out/r4f; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|^E " | tail -5 | tee $out/parity.txt
timeout 1500 python -m pytest tests/test_gpu_coverage.py -m gpu -x -q -k "not million" 2>&1 | grep -E "passed|failed|Error|^E " | tail -5 | tee -a $out/parity.txt
tools/gpu_ab.sh r4f/ab atrium ":" ":HIPR_SHADE_CLASSES=0" ":" ":HIPR_SHADE_CLASSES=0" 2>&1 | tee $out/ab_classes.txt
BENCH_ARGS="--wavefronts 1" tools/gpu_ab.sh r4f/abwf1 atrium ":" ":HIPR_SHADE_CLASSES=0" ":" ":HIPR_SHADE_CLASSES=0" 2>&1 | sed "s/^/wf1 /" | tee -a $out/ab_classes.txt

#!/bin/bash
# Stepping over the back of one-sided surfaces in the 8-wide traversal (hipr_set_backface_culling): GPU suite, then A/B against the retrace on one box.
set -u
out=gpurun_out/r3z; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|^FAILED|^E " | tail -12
for scene in atrium material; do tools/gpu_ab.sh r3z/ab_$scene $scene ":HIPR_BACKFACE_CULLING=0" ":HIPR_BACKFACE_CULLING=1" ":HIPR_BACKFACE_CULLING=0" ":HIPR_BACKFACE_CULLING=1" 2>&1 | tee -a $out/ab.txt; done
BENCH_ARGS="--atrium-triangles 10000000 --width 3840 --height 2160 --spp-per-pass 8 --steps 2" tools/gpu_ab.sh r3z/ab_10m atrium ":HIPR_BACKFACE_CULLING=0" ":HIPR_BACKFACE_CULLING=1" 2>&1 | tee -a $out/ab.txt
BENCH_ARGS="--spp-per-pass 1 --steps 64 --warmup 8" tools/gpu_ab.sh r3z/ab_1spp atrium ":HIPR_BACKFACE_CULLING=0" ":HIPR_BACKFACE_CULLING=1" 2>&1 | tee -a $out/ab.txt

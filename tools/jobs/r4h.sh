#!/bin/bash
# round 4, batch h: knobs re-swept under the pixel-major order (refill threshold, listing pass), 64 accumulations per pass
set -u
out=gpurun_out/r4h; mkdir -p $out
tools/gpu_ab.sh r4h/ab atrium ":" ":HIPR_REFILL_BELOW=24" ":HIPR_REFILL_BELOW=32" ":HIPR_REFILL_BELOW=48" ":HIPR_REFILL_BELOW=56" ":HIPR_SHADE_ORDERED=0" ":HIPR_SHADE_ORDERED_FROM=1048576" ":HIPR_SHADE_BLOCKS_PER_CU=2" ":" 2>&1 | tee $out/ab_knobs.txt

#!/bin/bash
# round 4, batch e: pixel-major slots with round-robin wavefronts and the LDS-staged accumulate: parity, then A/B against the sample-major build
set -u
out=gpurun_out/r4e; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py -m gpu -x -q 2>&1 | tail -4 | tee $out/parity.txt
for scene in atrium material cornell_diffuse cornell; do tools/gpu_ab.sh r4e/ab_$scene $scene ":" "samplemajor:" ":" "samplemajor:" 2>&1 | sed "s/^/$scene /" | tee -a $out/ab_slot_order.txt; done
BENCH_ARGS="--atrium-triangles 10000000 --width 3840 --height 2160 --spp-per-pass 8" tools/gpu_ab.sh r4e/ab10m atrium ":" "samplemajor:" 2>&1 | sed "s/^/atrium10M4k /" | tee -a $out/ab_slot_order.txt
BENCH_ARGS="--spp-per-pass 1 --steps 64 --warmup 8" tools/gpu_ab.sh r4e/ab1spp atrium ":" "samplemajor:" 2>&1 | sed "s/^/atrium_1spp /" | tee -a $out/ab_slot_order.txt

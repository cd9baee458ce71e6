#!/bin/bash
# round 4, batch d: pixel-major path slots (parity: batching / tiling / wavefront invariance + the whole parity file), A/B against the sample-major build
set -u
out=gpurun_out/r4d; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py -m gpu -x -q 2>&1 | tail -4 | tee $out/parity.txt
for scene in atrium material cornell_diffuse cornell; do tools/gpu_ab.sh r4d/ab_$scene $scene ":" "samplemajor:" ":" "samplemajor:" 2>&1 | sed "s/^/$scene /" | tee -a $out/ab_slot_order.txt; done
BENCH_ARGS="--atrium-triangles 10000000 --width 3840 --height 2160 --spp-per-pass 8" tools/gpu_ab.sh r4d/ab10m atrium ":" "samplemajor:" 2>&1 | sed "s/^/atrium10M4k /" | tee -a $out/ab_slot_order.txt
BENCH_ARGS="--wavefronts 1" tools/gpu_ab.sh r4d/abwf1 atrium ":" "samplemajor:" 2>&1 | sed "s/^/atrium_wf1 /" | tee -a $out/ab_slot_order.txt
BENCH_ARGS="--spp-per-pass 4 --steps 16 --warmup 4" tools/gpu_ab.sh r4d/ab4spp atrium ":" "samplemajor:" 2>&1 | sed "s/^/atrium_4spp /" | tee -a $out/ab_slot_order.txt
timeout 600 python -m pytest tests/test_gpu_statistics.py tests/test_gpu_bench.py -m gpu -x -q -s 2>&1 | grep -E "STATISTICS|passed|failed|Error|assert" | tee $out/statistics.txt

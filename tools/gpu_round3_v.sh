#!/bin/bash
# A/B: the 8-wide node test with the entry and exit plane of an axis in one v_pk_fma_f32 (base) against 48 scalar v_fma_f32 (nopk)
set -u
out=gpurun_out/r3v; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed" | tail -3
for scene in atrium material; do tools/gpu_ab.sh r3v/ab_$scene $scene "nopk:" ":" "nopk:" ":" 2>&1 | tee -a $out/ab.txt; done
BENCH_ARGS="--atrium-triangles 10000000 --width 3840 --height 2160 --steps 2" tools/gpu_ab.sh r3v/ab_10m atrium "nopk:" ":" 2>&1 | tee -a $out/ab.txt

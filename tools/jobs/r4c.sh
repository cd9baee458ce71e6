#!/bin/bash
# round 4, batch c: the sign-bit node test against the restated oracle (parity), wavefront counts on every workload, the new bench keys
set -u
out=gpurun_out/r4c; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4 | tee $out/parity.txt
for wf in 1 2 3 4; do BENCH_ARGS="--wavefronts $wf" tools/gpu_ab.sh r4c/wf$wf atrium ":" 2>&1 | sed "s/^/wf$wf atrium /" | tee -a $out/ab_wavefronts.txt; done
for wf in 1 2; do BENCH_ARGS="--wavefronts $wf" tools/gpu_ab.sh r4c/wfm$wf material ":" 2>&1 | sed "s/^/wf$wf material /" | tee -a $out/ab_wavefronts.txt; done
for wf in 1 2; do BENCH_ARGS="--wavefronts $wf --atrium-triangles 10000000 --width 3840 --height 2160 --spp-per-pass 8" tools/gpu_ab.sh r4c/wf10m$wf atrium ":" 2>&1 | sed "s/^/wf$wf atrium10M4k /" | tee -a $out/ab_wavefronts.txt; done
for wf in 1 2; do BENCH_ARGS="--wavefronts $wf --spp-per-pass 1 --steps 64 --warmup 8" tools/gpu_ab.sh r4c/wf1spp$wf atrium ":" 2>&1 | sed "s/^/wf$wf atrium1spp /" | tee -a $out/ab_wavefronts.txt; done
timeout 900 python bench.py --no-cpu-baseline --no-other-workloads --no-plugin --no-rmse > $out/bench_new_keys.json 2> $out/bench_new_keys.err; tail -c 4000 $out/bench_new_keys.json; grep -v "^\s*File\|^    " $out/bench_new_keys.err | tail -5

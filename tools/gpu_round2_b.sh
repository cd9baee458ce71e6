#!/bin/bash
# GPU box job 2 of round 2: the new bench.py line, self-spawned 2-rank run on one device, SQ counter passes on the atrium.
set -u
mkdir -p gpurun_out/r02b
python bench.py > gpurun_out/r02b/bench_default.json 2> gpurun_out/r02b/bench_default.err
python bench.py --gpus 2 --steps 2 --warmup 1 --dist-backend gloo --share-device --no-rmse --no-other-workloads > gpurun_out/r02b/bench_2rank_selfspawn.json 2> gpurun_out/r02b/bench_2rank_selfspawn.err
bash tools/profile_sq.sh r02b/sq_atrium --scene atrium --steps 2 --warmup 1 > gpurun_out/r02b/profile_sq.log 2>&1
for d in sq1 sq2 tcc; do python tools/pmc_summary.py gpurun_out/r02b/sq_atrium/$d k_shade k_trace_persistent > gpurun_out/r02b/sq_atrium_$d.txt; done
find gpurun_out/r02b/sq_atrium -name "*.csv" -size +2M -delete
cut -c1-600 gpurun_out/r02b/bench_default.json; tail -3 gpurun_out/r02b/bench_default.err
cut -c1-300 gpurun_out/r02b/bench_2rank_selfspawn.json; tail -3 gpurun_out/r02b/bench_2rank_selfspawn.err

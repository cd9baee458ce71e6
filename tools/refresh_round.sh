#!/bin/bash
# Runs on the GPU box: the plain bench lines of every workload (gpurun_out/bench_<scene>.json) and the 2-rank functional run.
set -u
mkdir -p gpurun_out
for s in cornell_diffuse cornell atrium material; do
    python bench.py --scene $s 2>gpurun_out/bench_$s.err | tail -1 > gpurun_out/bench_$s.json
    python tools/bench_summary.py $s < gpurun_out/bench_$s.json; python -c "import json; d=json.load(open(\"gpurun_out/bench_$s.json\")); print(\"   alone:\", d[\"roofline\"].get(\"alone\"))"
done
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --dist-backend gloo --share-device 2>gpurun_out/bench_2rank.err | tail -1 > gpurun_out/bench_2rank_gloo_shared.json
cut -c1-300 gpurun_out/bench_2rank_gloo_shared.json

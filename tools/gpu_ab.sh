#!/bin/bash
# A/B of libhiprenderer builds on one box: usage tools/gpu_ab.sh <outdir> <scene> "<lib suffix>:<env assignments>" ...
set -u
out=gpurun_out/$1; scene=$2; shift 2
mkdir -p $out
for spec in "$@"; do
    lib=${spec%%:*}; envs=${spec#*:}
    path=$PWD/bifrost3d_amd/csrc/libhiprenderer${lib:+_$lib}.so
    tag=$(echo "${lib:-base}_${envs}" | tr ' =' '__')
    env HIPR_LIBRARY=$path $envs python bench.py --scene $scene --steps 4 --warmup 1 ${BENCH_ARGS:-} --no-cpu-baseline --no-other-workloads --no-rmse --no-plugin --no-scaling-proxy --no-exact-mode --pmc-traffic off > $out/$tag.json 2> $out/$tag.err
    python - <<PY
import json
try:
    d = json.load(open("$out/$tag.json"))
    print("$tag", round(d["value"]), "Mrays/s", round(d["ms_per_step"], 2), "ms/step", {k: round(v, 1) for k, v in d["kernel_ms_per_step"].items()})
except Exception as e:
    print("$tag", "FAILED", e)
PY
done

#!/bin/bash
# A/B of the coherence sort (HIPR_COHERENCE_SORT): bench lines with and without it, alone and co-running, and the kernel stats of the sorted run.
# Round 6: the experiment is no longer linked into the product library; build the variant first (here, before going to the GPU box):
#   tools/build_variant.sh sort "-DHIPR_RAY_SORT=1"
set -u
export HIPR_LIBRARY=$(pwd)/bifrost3d_amd/csrc/libhiprenderer_sort.so
root=$(pwd); out=$root/gpurun_out/r4sort; mkdir -p $out; export TMPDIR=/tmp
quiet="--no-cpu-baseline --no-other-workloads --no-plugin --no-scaling-proxy --no-rmse"
python -m pytest tests/test_gpu_coverage.py -m gpu -q -k "order_the_trace_kernel" 2>&1 | tail -3
for sort in 0 1; do
  for wf in 1 2; do
    HIPR_COHERENCE_SORT=$sort python bench.py $quiet --wavefronts $wf > $out/bench_sort${sort}_wf${wf}.json 2> $out/bench_sort${sort}_wf${wf}.err
    python - <<PY
import json
d=json.loads(open("$out/bench_sort${sort}_wf${wf}.json").read().strip().splitlines()[-1])
print("sort $sort wf $wf", round(d["value"]), "Mrays/s", round(d["ms_per_step"],2), "ms", {k: round(v,2) for k,v in d["kernel_ms_per_step"].items()}, d.get("roofline_valu",{}).get("lanes_per_instruction"))
PY
  done
done
cd /tmp
HIPR_COHERENCE_SORT=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py $quiet --pmc-traffic off --wavefronts 1 > $out/under_rocprof.json 2> $out/under_rocprof.err
cd $root
find $out/trace -name "*kernel_trace.csv" -delete
f=$(find $out/trace -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-200

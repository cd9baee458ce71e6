#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r3t; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_coverage.py -x -q 2>&1 | grep -E "passed|failed" | tail -3
for scene in material atrium cornell_diffuse; do tools/gpu_ab.sh r3t/ab_$scene $scene ":HIPR_SHADE_ORDERED_FROM=262144" 2>&1 | tee -a $out/ab.txt; done
BENCH_ARGS="--spp-per-pass 1 --steps 64 --warmup 8" tools/gpu_ab.sh r3t/ab_1spp atrium ":HIPR_SHADE_ORDERED_FROM=262144" 2>&1 | tee -a $out/ab.txt
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d "$out/trace" -- python3 "$root/bench.py" --scene material --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads --no-rmse --no-plugin --pmc-traffic off > "$out/bench.json" 2> "$out/bench.err"
cd $root
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python tools/timeline_summary.py $f > $out/timeline_material.txt 2>&1
find "$out/trace" -name "*.csv" -size +4M -delete
tail -12 $out/timeline_material.txt

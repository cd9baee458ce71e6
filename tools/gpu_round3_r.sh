#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r3r; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d "$out/trace" -- python3 "$root/bench.py" --scene material --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads --no-rmse --no-plugin --pmc-traffic off > "$out/bench.json" 2> "$out/bench.err"
cd $root
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python tools/timeline_summary.py $f > $out/timeline_material.txt 2>&1
find "$out/trace" -name "*.csv" -size +4M -delete
tail -12 $out/timeline_material.txt

#!/bin/bash
# Runs on the GPU box (gpurun): kernel-trace stats + the two PMC passes of the bench command for one scene, into
# gpurun_out/<tag>/. Copy the summaries into profiles/ afterwards (tools/make_pmc_traffic.py for the PMC passes).
# usage: tools/profile_round.sh <tag> <bench args...>
set -u
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$root/bench.py" --no-cpu-baseline --no-other-workloads --no-rmse --no-plugin --no-scaling-proxy --no-exact-mode --pmc-traffic off "$@" > "$out/bench_trace.json" 2> "$out/bench_trace.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 "$root/bench.py" --no-cpu-baseline --no-other-workloads --no-rmse --no-plugin --no-scaling-proxy --no-exact-mode --pmc-traffic off "$@" > "$out/bench_fetch.json" 2> "$out/bench_fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 "$root/bench.py" --no-cpu-baseline --no-other-workloads --no-rmse --no-plugin --no-scaling-proxy --no-exact-mode --pmc-traffic off "$@" > "$out/bench_write.json" 2> "$out/bench_write.err"
cd "$root"
# keep the merged-back payload small: per-dispatch traces are large, the stats and counter CSVs are what is read
find "$out/trace" -name "*kernel_trace.csv" -size +8M -delete
ls -la "$out"/*/* | head -40

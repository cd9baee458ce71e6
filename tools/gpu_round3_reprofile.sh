#!/bin/bash
# The profiled runs again (kernel stats of the default command + the counter passes), after bench.py learnt to leave its retrace-mode block out of profiled runs.
set -u
root=$(pwd); out=$root/gpurun_out/r3final; mkdir -p $out
export TMPDIR=/tmp
rm -rf $out/trace
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py --no-cpu-baseline --no-other-workloads --no-rmse --no-plugin --pmc-traffic off > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.err
cd $root
find $out/trace -name "*kernel_trace.csv" -size +8M -delete
rm -rf gpurun_out/r03p
tools/gpu_round3_profiles.sh > /dev/null 2>&1
ls gpurun_out/r03p | head

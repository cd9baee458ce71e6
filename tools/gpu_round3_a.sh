#!/bin/bash
# Round 3, GPU job A: what bounds the trace kernel's gathers (tools/microbench/gather_nodes), calibration of FETCH_SIZE / WRITE_SIZE on the kernels'
# access patterns (tools/microbench/fetch_calibration under separate --pmc passes), the available counters, the LDS stack size A/B, the stack depth
# diagnostics, and the GPU test suite.
set -u
root=$(pwd)
out=$root/gpurun_out/r3a
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --list-avail > $out/list_avail.txt 2>&1
timeout 300 tools/microbench/gather_nodes > $out/gather_nodes.txt 2>&1
timeout 300 tools/microbench/fetch_calibration > $out/fetch_calibration_plain.jsonl 2>&1
cd /tmp
for counters in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum" "TCC_REQ_sum TCC_READ_sum"; do
    tag=$(echo $counters | tr ' ' '+')
    timeout 600 rocprofv3 --pmc $counters --output-format csv -d $out/cal_$tag -- $root/tools/microbench/fetch_calibration > $out/cal_$tag.jsonl 2> $out/cal_$tag.err
done
cd $root
BENCH_ARGS="" tools/gpu_ab.sh r3a/ab atrium ":" "stack24:" "stack20:" ":" 2>&1 | tee $out/ab.txt
HIPR_TRACE_LOG=1 timeout 600 python tools/trace_log_probe.py atrium 32 1 > $out/trace_log.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1
tail -3 $out/gpu_tests.log
find $out -name "*.csv" -size +4M -delete
ls $out

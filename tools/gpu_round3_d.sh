#!/bin/bash
# Round 3, GPU job D: the 8-wide tree with leaf records -- parity with the oracle, then against the 4-wide tree on the bench workloads.
set -u
root=$(pwd)
out=$root/gpurun_out/r3d
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_coverage.py -x -q > $out/gpu_tests_parity.log 2>&1; tail -5 $out/gpu_tests_parity.log
tools/gpu_ab.sh r3d/ab atrium ":" ":HIPR_TRACE_VARIANT=1" "w8waves5:" ":" 2>&1 | tee $out/ab_atrium.txt
tools/gpu_ab.sh r3d/ab_material material ":" ":HIPR_TRACE_VARIANT=1" 2>&1 | tee $out/ab_material.txt
HIPR_TRACE_LOG=1 timeout 600 python tools/trace_log_probe.py atrium 32 1 > $out/trace_log.txt 2>&1
grep -A1 "bounce [0-3]:" $out/trace_log.txt | head -20

#!/bin/bash
set -u
out=gpurun_out/r3y; mkdir -p $out
for v in nopark parkdiag0 parkdiag1 parkdiag2; do
  HIPR_LIBRARY=$PWD/bifrost3d_amd/csrc/libhiprenderer_$v.so HIPR_TRACE_LOG=1 timeout 300 python tools/trace_log_probe.py atrium 32 1 > $out/trace_log_$v.txt 2>&1
  echo "== $v"; grep -A1 "bounce 2:" $out/trace_log_$v.txt | cut -c1-250; tail -1 $out/trace_log_$v.txt | cut -c1-300
done
tools/gpu_ab.sh r3y/ab_atrium atrium "nopark:" ":" "parkv1:" "parkv2:" 2>&1 | tee -a $out/ab.txt

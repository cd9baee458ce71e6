#!/bin/bash
# Round 3, GPU job C: node fetch by quads through LDS-DMA (microbenchmark F and the trace kernel built with it), then the RMSE protocol.
set -u
root=$(pwd)
out=$root/gpurun_out/r3c
mkdir -p $out
export TMPDIR=/tmp
timeout 300 tools/microbench/gather_nodes > $out/gather_nodes.txt 2>&1
tools/gpu_ab.sh r3c/ab atrium ":" "quad:" "quad12:" 2>&1 | tee $out/ab.txt
HIPR_LIBRARY=$root/bifrost3d_amd/csrc/libhiprenderer_quad.so timeout 300 python -m pytest tests/test_gpu_parity.py -x -q > $out/parity_quad.log 2>&1; tail -2 $out/parity_quad.log
timeout 700 python tools/rmse_protocol.py --size 160x90 --out $out/rmse_protocol_160x90.json > $out/rmse_160.log 2>&1
tail -3 $out/rmse_160.log
cd /tmp
timeout 240 rocprofv3 --pmc TA_TA_BUSY_sum GRBM_GUI_ACTIVE --output-format csv -d $out/ta_busy -- python3 $root/bench.py --no-cpu-baseline --no-other-workloads --no-rmse --no-plugin --pmc-traffic off --scene atrium --steps 2 --warmup 1 > $out/ta_busy.json 2> $out/ta_busy.err
python3 $root/tools/pmc_summary.py $out/ta_busy k_trace_persistent k_shade > $out/ta_busy.txt 2>&1; cat $out/ta_busy.txt | head -20
find $out -name "*.csv" -size +4M -delete

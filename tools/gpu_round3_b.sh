#!/bin/bash
# Round 3, GPU job B: is the trace kernel bound by the address / tag pipeline of its gathers? (L1-resident gather microbenchmark, extra-load sensitivity
# builds, TA / TCP / TD counters of the bench), and the RMSE protocol with its converged leg.
set -u
root=$(pwd)
out=$root/gpurun_out/r3b
mkdir -p $out
export TMPDIR=/tmp
timeout 600 tools/microbench/gather_nodes > $out/gather_nodes.txt 2>&1
tools/gpu_ab.sh r3b/ab atrium ":" "extra1:" "extra2:" ":" 2>&1 | tee $out/ab.txt
tools/profile_ta.sh r3b/ta --scene atrium --steps 2 --warmup 1 > $out/ta.txt 2>&1
timeout 900 python tools/rmse_protocol.py --size 160x90 --out $out/rmse_protocol_160x90.json > $out/rmse_160.log 2>&1
tail -5 $out/rmse_160.log
ls $out

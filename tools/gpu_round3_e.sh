#!/bin/bash
# Round 3, GPU job E: what bounds the 8-wide kernel (SQ / TA / TCP / TCC counter passes of the bench), plus the full GPU test suite.
set -u
root=$(pwd)
out=$root/gpurun_out/r3e
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
i=0
for counters in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU" \
                "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
                "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    i=$((i + 1))
    timeout 300 rocprofv3 --pmc $counters --output-format csv -d $out/p$i -- python3 $root/bench.py --no-cpu-baseline --no-other-workloads --no-rmse --no-plugin --pmc-traffic off --scene atrium --steps 2 --warmup 1 > $out/p$i.json 2> $out/p$i.err
    python3 $root/tools/pmc_summary.py $out/p$i k_trace_wide8 k_shade > $out/p$i.txt 2>&1
done
cd $root
cat $out/p*.txt | grep -v "^$" | head -80
find $out -name "*.csv" -size +4M -delete
timeout 1200 python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1; tail -3 $out/gpu_tests.log

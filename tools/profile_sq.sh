#!/bin/bash
# SQ / TCC counter passes for one bench command (GPU box). usage: tools/profile_sq.sh <tag> <bench args...>
set -u
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d "$out/sq1" -- python3 "$root/bench.py" --no-cpu-baseline --no-other-workloads --no-rmse --no-plugin --no-scaling-proxy --no-exact-mode --pmc-traffic off "$@" > "$out/b1.json" 2> "$out/b1.err"
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM --output-format csv -d "$out/sq2" -- python3 "$root/bench.py" --no-cpu-baseline --no-other-workloads --no-rmse --no-plugin --no-scaling-proxy --no-exact-mode --pmc-traffic off "$@" > "$out/b2.json" 2> "$out/b2.err"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d "$out/tcc" -- python3 "$root/bench.py" --no-cpu-baseline --no-other-workloads --no-rmse --no-plugin --no-scaling-proxy --no-exact-mode --pmc-traffic off "$@" > "$out/b3.json" 2> "$out/b3.err"
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d "$out/tcp" -- python3 "$root/bench.py" --no-cpu-baseline --no-other-workloads --no-rmse --no-plugin --no-scaling-proxy --no-exact-mode --pmc-traffic off "$@" > "$out/b4.json" 2> "$out/b4.err"
cd "$root"
tail -2 "$out"/b*.err

#!/bin/bash
# Texture-addresser / L1 (TCP) / texture-data counter passes for one bench command (GPU box): is the trace kernel bound by the rate at which the
# per-lane 16 B gathers pass through the CU's address and tag pipeline? usage: tools/profile_ta.sh <tag> <bench args...>
set -u
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
i=0
for counters in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
                "GRBM_GUI_ACTIVE TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
                "GRBM_GUI_ACTIVE TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_sum" \
                "GRBM_GUI_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM"; do
    i=$((i + 1))
    timeout 900 rocprofv3 --pmc $counters --output-format csv -d "$out/ta$i" -- python3 "$root/bench.py" --no-cpu-baseline --no-other-workloads --no-rmse --no-plugin --no-scaling-proxy --no-exact-mode --pmc-traffic off "$@" > "$out/ta$i.json" 2> "$out/ta$i.err"
    python3 "$root/tools/pmc_summary.py" "$out/ta$i" k_trace_persistent k_shade > "$out/ta$i.txt" 2>&1
done
cd "$root"
cat "$out"/ta*.txt
find "$out" -name "*.csv" -size +4M -delete

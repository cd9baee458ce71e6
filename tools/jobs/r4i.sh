#!/bin/bash
# round 4, batch i: the order a ray takes the children of a group in: octant order (base), its leaf records first (order1), its inner nodes first (order2)
set -u
out=gpurun_out/r4i; mkdir -p $out
HIPR_LIBRARY=$PWD/bifrost3d_amd/csrc/libhiprenderer_order1.so HIPR_ORACLE_WIDE8_ITEM_ORDER=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "closest or shadow or atrium or counters" 2>&1 | grep -E "passed|failed|^E " | tail -3 | tee $out/parity_order1.txt
tools/gpu_ab.sh r4i/ab atrium ":" "order1:" "order2:" ":" "order1:" "order2:" 2>&1 | tee $out/ab_item_order.txt
BENCH_ARGS="--wavefronts 1" tools/gpu_ab.sh r4i/abwf1 atrium ":" "order1:" "order2:" 2>&1 | sed "s/^/wf1 /" | tee -a $out/ab_item_order.txt
tools/gpu_ab.sh r4i/abm material ":" "order1:" "order2:" 2>&1 | sed "s/^/material /" | tee -a $out/ab_item_order.txt
for lib in "" "_order1"; do HIPR_LIBRARY=$PWD/bifrost3d_amd/csrc/libhiprenderer$lib.so HIPR_TRACE_LOG=1 timeout 300 python tools/trace_log_probe.py atrium 32 1 2>&1 | grep -iE "lanes|iterations|busy" | head -8 | sed "s/^/trace_log$lib /" | tee -a $out/trace_log.txt; done

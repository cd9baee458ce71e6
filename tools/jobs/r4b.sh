#!/bin/bash
# round 4, batch b: two rays per lane (parity + timing), co-running wavefronts with --wavefronts 2, write traffic with / without spills, the new bench keys
set -u
out=gpurun_out/r4b; mkdir -p $out
HIPR_WIDE8_DUAL=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "closest or shadow or atrium or counters" 2>&1 | tail -5 | tee $out/parity_dual.txt
tools/gpu_ab.sh r4b/ab atrium ":" ":HIPR_WIDE8_DUAL=1" "sign:" "sign:HIPR_WIDE8_DUAL=1" ":" ":HIPR_WIDE8_DUAL=1" 2>&1 | tee $out/ab_dual.txt
tools/gpu_ab.sh r4b/abm material ":" ":HIPR_WIDE8_DUAL=1" 2>&1 | tee -a $out/ab_dual.txt
BENCH_ARGS="--atrium-triangles 10000000 --width 3840 --height 2160 --spp-per-pass 8" tools/gpu_ab.sh r4b/ab10m atrium ":" ":HIPR_WIDE8_DUAL=1" 2>&1 | tee -a $out/ab_dual.txt
for cfg in "" "HIPR_BLOCKS_PER_CU=8 HIPR_SHADE_BLOCKS_PER_CU=1" "HIPR_BLOCKS_PER_CU=10 HIPR_SHADE_BLOCKS_PER_CU=1" "HIPR_BLOCKS_PER_CU=6 HIPR_SHADE_BLOCKS_PER_CU=1"; do
    BENCH_ARGS="--wavefronts 2" tools/gpu_ab.sh r4b/wf atrium ":HIPR_X=1 $cfg" 2>&1 | tee -a $out/ab_wavefronts.txt
done
root=$(pwd); export TMPDIR=/tmp; cd /tmp
for lib in "" "_w5"; do
    HIPR_LIBRARY=$root/bifrost3d_amd/csrc/libhiprenderer$lib.so timeout 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $root/$out/pmc$lib -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads --no-rmse --no-plugin --no-scaling-proxy --pmc-traffic off > $root/$out/pmc$lib.json 2> $root/$out/pmc$lib.err
    python3 $root/tools/pmc_summary.py $root/$out/pmc$lib k_trace k_shade > $root/$out/pmc$lib.txt 2>&1
done
cd $root; find $out -name "*.csv" -size +2M -delete
cat $out/pmc*.txt
timeout 900 python bench.py --no-cpu-baseline --no-other-workloads --no-plugin --no-rmse > $out/bench_new_keys.json 2> $out/bench_new_keys.err; tail -c 3000 $out/bench_new_keys.json; tail -5 $out/bench_new_keys.err

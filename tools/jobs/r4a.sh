#!/bin/bash
# round 4, batch a: node block variants, co-running wavefronts, write traffic with and without register spills
set -u
out=gpurun_out/r4a; mkdir -p $out
tools/gpu_ab.sh r4a/ab atrium ":" "sign:" "signperm:" "w5:" ":" "sign:" "signperm:" 2>&1 | tee $out/ab_node.txt
for cfg in "HIPR_WAVEFRONTS=2" "HIPR_WAVEFRONTS=2 HIPR_BLOCKS_PER_CU=8 HIPR_SHADE_BLOCKS_PER_CU=1" "HIPR_WAVEFRONTS=2 HIPR_BLOCKS_PER_CU=8 HIPR_SHADE_BLOCKS_PER_CU=2" "HIPR_WAVEFRONTS=2 HIPR_BLOCKS_PER_CU=10 HIPR_SHADE_BLOCKS_PER_CU=1" "HIPR_WAVEFRONTS=2 HIPR_BLOCKS_PER_CU=6 HIPR_SHADE_BLOCKS_PER_CU=2" "HIPR_WAVEFRONTS=2 HIPR_BLOCKS_PER_CU=9 HIPR_SHADE_BLOCKS_PER_CU=1" "HIPR_WAVEFRONTS=3 HIPR_BLOCKS_PER_CU=8 HIPR_SHADE_BLOCKS_PER_CU=1"; do
    tools/gpu_ab.sh r4a/wf atrium ":$cfg" 2>&1 | tee -a $out/ab_wavefronts.txt
done
root=$(pwd); export TMPDIR=/tmp; cd /tmp
for lib in "" "_w5"; do
    HIPR_LIBRARY=$root/bifrost3d_amd/csrc/libhiprenderer$lib.so timeout 600 rocprofv3 --pmc WRITE_SIZE FETCH_SIZE --output-format csv -d $root/$out/pmc$lib -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads --no-rmse --no-plugin --pmc-traffic off > $root/$out/pmc$lib.json 2> $root/$out/pmc$lib.err
    python3 $root/tools/pmc_summary.py $root/$out/pmc$lib k_trace k_shade > $root/$out/pmc$lib.txt 2>&1
done
cd $root; find $out -name "*.csv" -size +2M -delete
cat $out/pmc*.txt

#!/bin/bash
# A/B of the shade kernel's arithmetic: round 3's division (variant r3div), reciprocal math without the shared wo terms (noterms), the product.
# Needs tools/build_variant.sh r3div / noterms built first. Times with one wavefront (every kernel alone), and the VALU instruction counters of k_shade.
set -u
root=$(pwd); out=$root/gpurun_out/r4shade; mkdir -p $out; export TMPDIR=/tmp
for scene in atrium material cornell_diffuse; do
  BENCH_ARGS="--wavefronts 1" bash tools/gpu_ab.sh r4shade $scene "r3div:" "noterms:" ":"
done
cd /tmp
for lib in r3div noterms base; do
  path=$root/bifrost3d_amd/csrc/libhiprenderer_$lib.so; [ $lib = base ] && path=$root/bifrost3d_amd/csrc/libhiprenderer.so
  export HIPR_LIBRARY=$path
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/pmc_$lib -- python3 $root/tools/shade_bound_probe.py 16 > $out/pmc_$lib.log 2>&1
  python3 $root/tools/pmc_summary.py $out/pmc_$lib k_shade > $out/pmc_$lib.txt; tail -3 $out/pmc_$lib.txt | cut -c1-300
done
find $out -name "*.csv" -size +2M -delete

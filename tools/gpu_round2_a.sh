#!/bin/bash
# GPU box job 1 of round 2: the full GPU suite, RMSE probes (fast vs precise shade math), atrium bench for both builds.
set -u
mkdir -p gpurun_out/r02a
python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r02a/gpu_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/r02a/gpu_tests.log
python tools/rmse_probe.py --scene atrium --spp 8,256 > gpurun_out/r02a/rmse_fast.json 2> gpurun_out/r02a/rmse_fast.err
HIPR_LIBRARY=$PWD/bifrost3d_amd/csrc/libhiprenderer_precise.so python tools/rmse_probe.py --scene atrium --spp 8,256 > gpurun_out/r02a/rmse_precise.json 2> gpurun_out/r02a/rmse_precise.err
python bench.py --scene atrium --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r02a/bench_atrium_fast.json 2> gpurun_out/r02a/bench_atrium_fast.err
HIPR_LIBRARY=$PWD/bifrost3d_amd/csrc/libhiprenderer_precise.so python bench.py --scene atrium --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r02a/bench_atrium_precise.json 2> gpurun_out/r02a/bench_atrium_precise.err
HIPR_LIBRARY=$PWD/bifrost3d_amd/csrc/libhiprenderer_precise.so python bench.py --scene cornell --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r02a/bench_cornell_precise.json 2> gpurun_out/r02a/bench_cornell_precise.err
python bench.py --scene cornell --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r02a/bench_cornell_fast.json 2> gpurun_out/r02a/bench_cornell_fast.err
tail -5 gpurun_out/r02a/gpu_tests.log
cat gpurun_out/r02a/rmse_fast.json gpurun_out/r02a/rmse_precise.json

#!/bin/bash
# Round 3 evidence run (GPU box): the default bench line, rocprofv3 kernel stats of the same command, the full RMSE protocol, the GPU test log with image
# metrics, other workloads. Results under gpurun_out/r3final; tools/collect_round3.sh copies the summaries into profiles/.
set -u
root=$(pwd)
out=$root/gpurun_out/r3final
mkdir -p $out
export TMPDIR=/tmp
python bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -c 600 $out/bench_default.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py --no-cpu-baseline --no-other-workloads --no-rmse --no-plugin --pmc-traffic off > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.err
cd $root
find $out/trace -name "*kernel_trace.csv" -size +8M -delete
for scene in material cornell_diffuse; do
    python bench.py --scene $scene --steps 4 --warmup 1 --no-cpu-baseline --no-other-workloads --no-plugin > $out/bench_$scene.json 2> $out/bench_$scene.err
done
python bench.py --atrium-triangles 10000000 --width 3840 --height 2160 --spp-per-pass 8 --steps 4 --warmup 1 --no-cpu-baseline --no-other-workloads --no-plugin > $out/bench_atrium10M_4k.json 2> $out/bench_atrium10M_4k.err
python bench.py --spp-per-pass 1 --steps 64 --warmup 8 --no-cpu-baseline --no-other-workloads --no-plugin --no-rmse > $out/bench_atrium_1spp.json 2> $out/bench_atrium_1spp.err
python bench.py --gpus 2 --share-device --dist-backend gloo --steps 4 --warmup 1 --no-rmse > $out/bench_2rank_gloo_shared_device.json 2> $out/bench_2rank.err
timeout 1500 python tools/rmse_protocol.py --size 480x270 --out $out/rmse_protocol_480x270.json > $out/rmse_480.log 2>&1
timeout 600 python tools/rmse_protocol.py --size 160x90 --out $out/rmse_protocol_160x90.json > $out/rmse_160.log 2>&1
timeout 1800 python -m pytest tests -m gpu -q -s > $out/gpu_tests.log 2>&1; grep -E "passed|failed" $out/gpu_tests.log | tail -2
HIPR_TRACE_LOG=1 timeout 600 python tools/trace_log_probe.py atrium 32 1 > $out/trace_log.txt 2>&1
ls $out

#!/bin/bash
# The evidence run of a round on the GPU box, one runner for every round (replaces the per-round job scripts):
#   tools/gpu_evidence.sh <tag> [part ...]        e.g. gpurun -- 'bash tools/gpu_evidence.sh r04 bench stats tests'
# parts (default: all, in this order)
#   bench     the default `python bench.py` line (what the driver runs)
#   stats     rocprofv3 --kernel-trace --stats of the same workload with ONE wavefront (every kernel alone on the device: the durations the bench line's
#             rooflines are priced on; the timed region of the default line runs two co-running wavefronts whose launches overlap)
#   workloads the other BASELINE configurations (material, cornell_diffuse, the textured atrium, 10 M triangles at 4K, 1 spp per pass, two ranks on one device)
#   rmse      tools/rmse_protocol.py at 480x270 and 160x90
#   tests     the GPU suite with image metrics, then smoke()
#   tracelog  lanes / iterations / refills of the traversal kernel (HIPR_TRACE_LOG)
#   verify    tools/verify_probe.py: verification build vs oracle (bit-identical pixels) and product vs verification build, 1920x1080x256 spp and 160x90x64 spp
#   counters  FETCH_SIZE / WRITE_SIZE passes of three workloads and the SQ / TCC / TCP passes of the atrium
# Results under gpurun_out/<tag>/; `python tools/collect_evidence.py gpurun_out/<tag> <tag>` copies the summaries into profiles/<tag>_*.
set -u
tag=$1; shift
parts=${*:-bench stats workloads rmse tests tracelog counters}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
quiet="--no-cpu-baseline --no-other-workloads --no-plugin --no-scaling-proxy"
for part in $parts; do
    case $part in
    bench)
        python bench.py --details $out/bench_details.json > $out/bench_default.json 2> $out/bench_default.err; wc -c $out/bench_default.json; tail -c 600 $out/bench_default.json
        python3 bench.py --gpus 1 --steps 20 --warmup 5 --details $out/bench_driver_command_details.json > $out/bench_driver_command.json 2> $out/bench_driver_command.err ;;
    stats)
        cd /tmp
        timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py $quiet --no-rmse --no-textured --pmc-traffic off --wavefronts 1 --details $out/bench_under_rocprof_details.json > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.err
        cd $root
        find $out/trace -name "*kernel_trace.csv" -size +8M -delete ;;
    workloads)
        for scene in material cornell_diffuse atrium_textured; do
            python bench.py --scene $scene --steps 4 --warmup 1 $quiet --details $out/bench_${scene}_details.json > $out/bench_$scene.json 2> $out/bench_$scene.err
        done
        python bench.py --atrium-triangles 10000000 --width 3840 --height 2160 --spp-per-pass 8 --steps 4 --warmup 1 $quiet --details $out/bench_atrium10M_4k_details.json > $out/bench_atrium10M_4k.json 2> $out/bench_atrium10M_4k.err
        python bench.py --spp-per-pass 1 --steps 64 --warmup 8 $quiet --no-rmse --details $out/bench_atrium_1spp_details.json > $out/bench_atrium_1spp.json 2> $out/bench_atrium_1spp.err
        python bench.py --gpus 2 --share-device --dist-backend gloo --steps 4 --warmup 1 --no-rmse --details $out/bench_2rank_gloo_shared_device_details.json > $out/bench_2rank_gloo_shared_device.json 2> $out/bench_2rank.err ;;
    rmse)
        timeout 1800 python tools/rmse_protocol.py --size 480x270 --decay-to 1024 --out $out/rmse_protocol_480x270.json > $out/rmse_480.log 2>&1
        timeout 900 python tools/rmse_protocol.py --size 160x90 --decay-to 4096 --out $out/rmse_protocol_160x90.json > $out/rmse_160.log 2>&1 ;;
    tests)
        timeout 2400 python -m pytest tests -m gpu -q -s > $out/gpu_tests.log 2>&1; grep -E "passed|failed" $out/gpu_tests.log | tail -2
        python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ;;
    tracelog)
        HIPR_TRACE_LOG=1 timeout 600 python tools/trace_log_probe.py atrium 32 1 > $out/trace_log.txt 2>&1 ;;
    counters)
        for scene in atrium cornell_diffuse material; do
            bash tools/profile_round.sh $tag/$scene --scene $scene --steps 4 --warmup 1 --wavefronts 1 > $out/${scene}_profile.log 2>&1
        done
        bash tools/profile_sq.sh $tag/sq_atrium --scene atrium --steps 2 --warmup 1 --wavefronts 1 > $out/sq.log 2>&1
        for d in sq1 sq2 tcc tcp; do python tools/pmc_summary.py $out/sq_atrium/$d k_shade k_trace_wide8 k_generate k_accumulate k_classify_hits > $out/sq_atrium_$d.txt; done
        find $out -name "*.csv" -size +3M -delete
        find $out -name "*agent_info.csv" -delete ;;
    verify)
        # the verification build against the oracle at the metric's own size (the oracle's 1080p x 256 spp image takes minutes of the box's host cores), then stage level
        timeout 3000 python tools/verify_probe.py --no-stages --no-libm --scenes atrium --size 1920x1080 --spp 256 --out $out/verify_probe_1920x1080.json > $out/verify_probe_1920x1080.log 2>&1
        timeout 900 python tools/verify_probe.py --out $out/verify_probe_160x90.json > $out/verify_probe_160x90.log 2>&1 ;;
    *) echo "unknown part $part" ;;
    esac
done
ls $out

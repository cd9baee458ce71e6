#!/bin/bash
# GPU box job: the round's rocprofv3 evidence for the default bench workload (atrium) and the two other workloads.
#   gpurun_out/r02p/<scene>/trace      --kernel-trace --stats
#   gpurun_out/r02p/<scene>/pmc_fetch, pmc_write   HBM traffic counters (separate passes)
#   gpurun_out/r02p/sq_<scene>/...     SQ / TCC counter passes (atrium only)
set -u
for scene in atrium cornell_diffuse material; do
    bash tools/profile_round.sh r02p/$scene --scene $scene --steps 4 --warmup 1 --no-plugin > gpurun_out/r02p_${scene}_profile.log 2>&1
done
bash tools/profile_sq.sh r02p/sq_atrium --scene atrium --steps 2 --warmup 1 --no-plugin > gpurun_out/r02p_sq.log 2>&1
for d in sq1 sq2 tcc tcp; do python tools/pmc_summary.py gpurun_out/r02p/sq_atrium/$d k_shade k_trace_persistent k_generate k_accumulate > gpurun_out/r02p/sq_atrium_$d.txt; done
find gpurun_out/r02p -name "*.csv" -size +3M -delete
find gpurun_out/r02p -name "*agent_info.csv" -delete
ls -R gpurun_out/r02p | head -60

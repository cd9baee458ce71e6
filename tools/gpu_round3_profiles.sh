#!/bin/bash
# GPU box job: counter evidence of the final library -- FETCH_SIZE / WRITE_SIZE passes of the three bench workloads (profiles/pmc_traffic.json, the fallback of
# bench.py's live measurement and what its N > 1 lines quote) and the SQ / TCC / TCP passes of the atrium (profiles/r03_atrium_sq_counters.txt, profiles/sq_limiters.json).
set -u
for scene in atrium cornell_diffuse material; do
    bash tools/profile_round.sh r03p/$scene --scene $scene --steps 4 --warmup 1 > gpurun_out/r03p_${scene}_profile.log 2>&1
done
bash tools/profile_sq.sh r03p/sq_atrium --scene atrium --steps 2 --warmup 1 > gpurun_out/r03p_sq.log 2>&1
for d in sq1 sq2 tcc tcp; do python tools/pmc_summary.py gpurun_out/r03p/sq_atrium/$d k_shade k_trace_wide8 k_generate k_accumulate k_classify_hits > gpurun_out/r03p/sq_atrium_$d.txt; done
find gpurun_out/r03p -name "*.csv" -size +3M -delete
find gpurun_out/r03p -name "*agent_info.csv" -delete
ls gpurun_out/r03p gpurun_out/r03p/atrium | head -30

#!/bin/bash
# round 4, batch g: the whole GPU suite on the current library, accumulations per pass 32 / 64 / 128
set -u
out=gpurun_out/r4g; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -q -s > $out/gpu_tests.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $out/gpu_tests.log | tail -8
grep -E "STATISTICS" $out/gpu_tests.log | cut -c1-400
for spp in 32 64 128; do BENCH_ARGS="--spp-per-pass $spp" tools/gpu_ab.sh r4g/spp$spp atrium ":" 2>&1 | sed "s/^/spp$spp /" | tee -a $out/ab_spp_per_pass.txt; done
python - <<'PY'
import json
for spp in (32, 64, 128):
    d = json.load(open(f"gpurun_out/r4g/spp{spp}/base_.json"))
    print(spp, "ms per 256 spp frame", round(d["config"]["ms_per_256spp_frame"], 1), "Mrays/s", round(d["value"]))
PY

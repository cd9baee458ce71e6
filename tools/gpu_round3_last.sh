#!/bin/bash
# Last run of the round: the default bench line and the whole GPU suite with the final library.
set -u
out=gpurun_out/r3last; mkdir -p $out
python bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -c 300 $out/bench_default.json
timeout 1800 python -m pytest tests -m gpu -q -s > $out/gpu_tests.log 2>&1; grep -E "passed|failed" $out/gpu_tests.log | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2

"""bench.py's self-spawned ranks (N > 1 started plainly) fail fast: a rank that dies at start-up ends the others instead of leaving them in the rendezvous
until the driver's timeout, and the overall deadline holds. No GPU: the ranks are stopped by test hooks before they touch one."""
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def run(extra_env, extra_args=()):
    env = dict(os.environ, **extra_env)
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    t0 = time.time()
    done = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"] + list(extra_args), capture_output=True, text=True, timeout=120, env=env,
                          cwd=str(ROOT))
    return done, time.time() - t0


def test_a_rank_that_dies_at_start_ends_the_job():
    done, seconds = run({"HIPR_BENCH_TEST_FAIL_RANK": "1", "HIPR_BENCH_TEST_HANG_RANK": "0"})
    assert done.returncode == 3, (done.returncode, done.stderr[-500:])
    assert seconds < 60
    assert "rank 1 exited with 3" in done.stderr
    assert done.stdout.strip() == ""


def test_the_deadline_ends_ranks_that_never_finish():
    done, seconds = run({"HIPR_BENCH_TEST_HANG_RANK": "all"}, ["--spawn-deadline", "2"])
    assert done.returncode == 124 and seconds < 60, (done.returncode, done.stderr[-500:])
    assert "did not finish within" in done.stderr


def test_bench_host_logic_without_a_gpu():
    """The parts of bench.py that decide what is measured, importable and checkable on the host: the kernel names the counter passes are summed under
    (instrumented launches are NOT the timed kernel), the fixed-frame batching (exactly `steps` steps timed), the VALU roofline's
    arithmetic on counters calibrated by the rate kernel, the unavoidable bytes of a trace launch."""
    sys.path.insert(0, str(ROOT))
    import bench
    assert bench.kernel_bench_name("void hipr::k_trace_wide8<12, 2, false>(hipr::DeviceScene, ...)") == "trace"
    assert bench.kernel_bench_name("void hipr::k_trace_wide8<12, 0, false>(...)") == "trace_closest"
    assert bench.kernel_bench_name("void hipr::k_trace_wide8<12, 1, false>(...)") == "trace_shadow"
    assert bench.kernel_bench_name("void hipr::k_trace_wide8<12, 2, true>(...)") is None              # the counting build of the kernel
    assert bench.kernel_bench_name("void hipr::k_trace_persistent<16, 2, true, false>(...)") is None
    assert bench.kernel_bench_name("void hipr::k_shade<1, false, 0>(...)") == "shade" and bench.kernel_bench_name("k_classify_hits") is None
    # exactly `steps` steps are timed: the batch divides the step count (the warm-up's remainder runs as one shorter pass); the driver's 20 steps on 8 GPUs: 5
    for steps, world, expected in ((8, 8, 8), (8, 1, 1), (6, 4, 3), (5, 8, 5), (20, 8, 5), (20, 4, 4), (20, 2, 2), (16, 8, 8), (7, 4, 1)):
        g = bench.fixed_frame_batch(steps, world)
        assert g == expected and steps % g == 0 and 1 <= g <= max(1, world), (steps, world, g)
    # a rate kernel counted at half its instructions (a counter that saw half the SIMDs) doubles the kernel's figure; 32 of 64 lanes stay 32
    cus = 256
    expected = cus * 8 * 4 * bench.RATE_KERNEL_ITERATIONS * 8.0
    valu = {"SQ_INSTS_VALU": {"trace": 1.0e9, "rate_fma": expected / 2}, "SQ_THREAD_CYCLES_VALU": {"trace": 32.0e9, "rate_fma": 64.0 * expected / 2},
            "SQ_ACTIVE_INST_VALU": {"trace": 0.5e9, "rate_fma": 1.0e9}, "valu_seconds": {"trace": 2.0e-3, "rate_fma": 1.0e-3}}
    r = bench.valu_roofline("trace", "k", 2.0e-3, valu, {"v_fma_f32": 2.0e12, "v_max_f32": 1.4e12, "v_cvt_f32_ubyte1": 1.4e12}, cus)
    assert r["calibration"]["scale"] == 2.0 and abs(r["achieved"] - 1000.0) < 1e-9 and abs(r["frac"] - 0.5) < 1e-12
    assert abs(r["lanes_per_instruction"] - 32.0) < 1e-9 and abs(r["valu_busy"] - 0.25) < 1e-12
    assert "error" in bench.valu_roofline("trace", "k", 1e-3, {"error": "no pass"}, {"v_fma_f32": 1.0}, cus)
    u = bench.useful_traffic("trace", {"closest_rays": 100, "shadow_rays": 50, "camera_rays": 10}, 5, 64)
    assert u["writes"] == (16 * 100 + 16 * 50) / 5 and u["reads"] == (40 * 100 + 64 * 50) / 5 and u["bytes"] == u["writes"] + u["reads"]
    assert bench.useful_traffic("shade", {"closest_rays": 1, "shadow_rays": 1, "camera_rays": 1}, 1, 1) is None


def test_the_line_the_driver_parses_is_small_and_round_trips():
    """round 4's single line had grown to 25 KB and the driver's record of it was unparseable. `compact_line` on that very record (canned:
    profiles/r04_bench_final.json) and on records with NaNs, huge texts and missing parts: under 4 KB, strict JSON, the contract keys and the two extra objects."""
    import json
    import math
    sys.path.insert(0, str(ROOT))
    import bench
    full = json.loads((ROOT / "profiles" / "r04_bench_final.json").read_text())
    assert len(json.dumps(full)) > 20000
    cases = [full]
    noisy = json.loads(json.dumps(full))
    noisy["config"]["workload"] = "w" * 5000
    noisy["roofline"]["kernel"] = "k" * 3000
    noisy["roofline"]["frac_model"] = float("nan")
    noisy["roofline_valu"] = {"bound": "valu", "error": "e" * 4000}
    noisy["cpu_baseline"]["sample"] = "s" * 4000
    noisy["config"]["workload_textured"] = {"value": 4899.0, "ms_per_step": float("inf")}
    noisy["ranks"] = {"ms_per_step": [1.0] * 8, "gather_ms": 0.1, "passes": 4, "steps_per_pass": 5, "paths_per_gpu_per_step": 1}
    cases.append(noisy)
    bare = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    bare["config"] = {"workload": "x"}
    cases.append(bare)
    for record in cases:
        line = bench.compact_line(record, "bench_details.json")
        text = json.dumps(line, allow_nan=False)
        assert len(text) < bench.LINE_LIMIT == 4096, len(text)
        assert "\n" not in text and json.loads(text) == line
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
            assert key in line, key
        assert line["metric"] == json.loads((ROOT / "BASELINE.json").read_text())["metric"] and "workload" in line["config"] and "model" not in line["config"]
    line = bench.compact_line(full)
    assert line["value"] == float(f"{full['value']:.6g}") and line["config"]["ms_per_256spp_frame"] == float(f"{full['config']['ms_per_256spp_frame']:.6g}")
    roofline = line["roofline"]
    assert roofline["bound"] == "hbm" and roofline["unit"] == "GB/s" and roofline["peak"] == 8000.0 and math.isclose(roofline["frac"], roofline["achieved"] / roofline["peak"], rel_tol=1e-4)
    for key in ("kernel", "avg_launch_ms", "launches", "traffic", "algorithmic_bytes_per_launch", "frac_model", "frac_basis", "traffic_over_useful"):
        assert key in roofline, key
    assert set(line["roofline_valu"]) >= {"frac", "lanes_per_instruction"} and set(line["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample", "c2"}
    assert set(line["config"]["rmse_vs_oracle"]["spp256"]) == {"rmse_rgb", "rmse_reference_compare_rms"}
    # what does not fit goes, the contract stays
    assert "roofline" in bench.compact_line(noisy) and bench.compact_line(noisy)["roofline"]["frac_model"] is None


def test_emit_writes_the_details_and_one_line(tmp_path):
    import json
    sys.path.insert(0, str(ROOT))
    import bench
    full = json.loads((ROOT / "profiles" / "r04_bench_final.json").read_text())
    read_fd, write_fd = os.pipe()
    line = bench.emit(full, write_fd, str(tmp_path / "details.json"))
    os.close(write_fd)
    with os.fdopen(read_fd) as fh:
        printed = fh.read()
    assert printed.count("\n") == 1 and json.loads(printed) == line and len(printed) < 4096
    details = json.loads((tmp_path / "details.json").read_text())
    assert details["scaling_proxy"] and details["other_workloads"] and details["roofline_by_kernel"]


def test_the_round_6_records_keep_their_new_keys_in_the_compact_line():
    """config.exact_mode (the exact arithmetic mode's rate and RMSE, measured in the same run) and the N > 1 proof keys survive the cut to one line under 4 KB: on the
    committed records of round 6 (the default line and the two-rank line) and when the texts around them are at their longest."""
    import json
    sys.path.insert(0, str(ROOT))
    import bench
    full = json.loads((ROOT / "profiles" / "r06_bench_details.json").read_text())
    line = bench.compact_line(full, "bench_details.json")
    assert len(json.dumps(line, allow_nan=False)) < bench.LINE_LIMIT
    exact = line["config"]["exact_mode"]
    assert exact["value"] > 0.7 * line["value"] and exact["rmse_vs_oracle"]["rmse_rgb"] == 0.0 and exact["rmse_vs_oracle"]["pixels_bit_identical"] == 1.0
    assert line["cpu_baseline"]["cores"] == 16 and line["cpu_baseline"]["jobs"] >= 3 and line["cpu_baseline"]["min"] <= line["cpu_baseline"]["value"] <= line["cpu_baseline"]["max"]
    noisy = json.loads(json.dumps(full))
    noisy["config"]["workload"] = "w" * 5000
    noisy["cpu_baseline"]["sample"] = "s" * 4000
    noisy["roofline"]["kernel"] = "k" * 3000
    cut = bench.compact_line(noisy, "bench_details.json")
    assert len(json.dumps(cut, allow_nan=False)) < bench.LINE_LIMIT and cut["config"]["exact_mode"]["value"] == exact["value"]
    two = json.loads((ROOT / "profiles" / "r06_bench_2rank_gloo_shared_device_details.json").read_text())
    ranks = bench.compact_line(two)["ranks"]
    assert ranks["world_seen"] == {"get_world_size": 2, "all_reduce_of_ones": 2} and len(ranks["devices"]) == 2 and ranks["tile_split_probe_identical"] is True
    assert isinstance(ranks["rccl_version"], str) and ranks["backend"] == "gloo" and len(json.dumps(bench.compact_line(two), allow_nan=False)) < bench.LINE_LIMIT

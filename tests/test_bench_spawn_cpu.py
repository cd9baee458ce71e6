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
    (instrumented launches are NOT the timed kernel), the --fixed-frame batching (exactly `steps` steps timed after exactly `warmup`), the VALU roofline's
    arithmetic on counters calibrated by the rate kernel, the unavoidable bytes of a trace launch."""
    sys.path.insert(0, str(ROOT))
    import bench
    assert bench.kernel_bench_name("void hipr::k_trace_wide8<12, 2, false>(hipr::DeviceScene, ...)") == "trace"
    assert bench.kernel_bench_name("void hipr::k_trace_wide8<12, 0, false>(...)") == "trace_closest"
    assert bench.kernel_bench_name("void hipr::k_trace_wide8<12, 1, false>(...)") == "trace_shadow"
    assert bench.kernel_bench_name("void hipr::k_trace_wide8<12, 2, true>(...)") is None              # the counting build of the kernel
    assert bench.kernel_bench_name("void hipr::k_trace_persistent<16, 2, true, false>(...)") is None
    assert bench.kernel_bench_name("void hipr::k_shade<1, false, 0>(...)") == "shade" and bench.kernel_bench_name("k_classify_hits") is None
    for steps, warmup, world, expected in ((8, 2, 8, 2), (8, 8, 8, 8), (8, 0, 8, 8), (8, 2, 1, 1), (6, 3, 4, 3), (5, 1, 8, 1), (20, 4, 8, 4), (16, 8, 8, 8)):
        g = bench.fixed_frame_batch(steps, warmup, world)
        assert g == expected and steps % g == 0 and warmup % g == 0 and 1 <= g <= max(1, world), (steps, warmup, world, g)
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

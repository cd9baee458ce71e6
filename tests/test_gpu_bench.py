"""bench.py end to end on the GPU box, small: the N = 1 line's contract, and the N = 2 path (ranks spawned by bench.py itself, tiles dealt round-robin,
gather to rank 0, scatter into the frame) with both ranks on the one device of the box and the gloo backend, since RCCL wants one device per rank."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
SMALL = ["--width", "640", "--height", "360", "--atrium-triangles", "20000", "--steps", "2", "--warmup", "1", "--spp-per-pass", "4",
         "--no-other-workloads", "--no-rmse", "--no-plugin", "--no-cpu-baseline"]


def run_bench(extra, tmp_path=None, extra_env=None):
    """Returns (the one line of stdout, the full record of --details)."""
    import tempfile
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", **(extra_env or {}))
    details = Path(tempfile.mkdtemp(prefix="hipr_bench_")) / "details.json"
    done = subprocess.run([sys.executable, str(ROOT / "bench.py")] + SMALL + extra + ["--details", str(details)], capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert done.returncode == 0, done.stderr[-2000:]
    lines = [text for text in done.stdout.splitlines() if text.strip()]
    assert len(lines) == 1, lines          # ONE JSON line on stdout, everything else on stderr
    assert len(lines[0]) < 4096            # and one the driver can hold (round 4's 25 KB line could not be parsed)
    return json.loads(lines[0]), json.loads(details.read_text())


def test_bench_line_contract_single_gpu():
    compact, line = run_bench([])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "roofline_valu"):
        assert key in compact, key
    assert compact["scaling"] == "strong" and compact["value"] == pytest.approx(line["value"], rel=1e-5) and compact["roofline"]["frac"] == pytest.approx(line["roofline"]["frac"], rel=1e-5)
    assert compact["config"]["workload_textured"]["value"] > 0 and compact["roofline"]["traffic"] and "lanes_per_instruction" in compact["roofline_valu"]
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1 and line["unit"] == "Mrays/s" and line["value"] > 0
    assert line["config"]["frame_finite_and_lit"] and "workload" in line["config"]
    roofline = line["roofline"]
    assert roofline["bound"] in ("hbm", "mfma") and roofline["unit"] == "GB/s" and roofline["peak"] == 8000.0
    assert roofline["frac"] == pytest.approx(roofline["achieved"] / roofline["peak"]) and 0.0 < roofline["frac"] <= 1.0
    # the headline fraction is counter based and measured by this very run (two rocprofv3 --pmc child runs before the timed one); the byte model sits next to it
    assert roofline["traffic"] and roofline["traffic_source"]["measured"].startswith("live"), roofline.get("traffic_source")
    assert roofline["achieved"] == pytest.approx(roofline["traffic"] / (roofline["avg_launch_ms"] * 1e-3) / 1e9)
    assert "frac_model" in roofline and "observed_limiter" in roofline
    # round 4: the VALU roof beside the HBM one (counters from a third live pass, the peak measured in the timed process), the bytes the kernel could not avoid,
    # and the one-GPU proxy of the N-way tile split
    valu = line["roofline_valu"]
    assert valu["bound"] == "valu" and "error" not in valu, valu
    assert 0.0 < valu["frac"] <= 1.2 and valu["frac"] == pytest.approx(valu["achieved"] / valu["peak"]) and 8.0 <= valu["lanes_per_instruction"] <= 64.0
    assert 0.98 <= valu["calibration"]["scale"] <= 1.02          # SQ_INSTS_VALU counts the rate kernel's instructions to within 2 %
    if "trace" in roofline["kernel"]:      # the unavoidable bytes are defined for the trace kernels (the dominant one here)
        assert roofline["traffic_useful"]["bytes"] > 0 and roofline["traffic_over_useful"] >= 0.5 and roofline["write_amplification"] > 0
    proxy = line["scaling_proxy"]
    for kind in ("strong", "weak", "interactive"):
        assert set(proxy[kind]) >= {"2", "4", "8"}
    assert proxy["strong"]["8"]["paths_per_gpu_per_step"] * 8 == pytest.approx(proxy["strong"]["1"]["paths_per_gpu_per_step"], rel=0.02)
    assert 1.0 < proxy["strong"]["8"]["predicted_speedup"] <= 8.5 and 0.3 < proxy["weak"]["8"]["predicted_efficiency"] <= 1.1
    # the same frames with every refused hit retraced, timed beside the line: more rays (the atrium's one-sided colonnade and drapes seen from behind), more time
    retrace = line["retrace_mode"]
    assert "backface_culling" in line["config"]
    assert 1.05 * line["config"]["rays_per_step"] < retrace["rays_per_step"] < 1.3 * line["config"]["rays_per_step"]
    assert retrace["ms_per_step"] > 1.03 * line["ms_per_step"] and retrace["Mrays_per_s"] > 0


def test_bench_two_ranks_on_one_device():
    _, one = run_bench(["--no-textured", "--no-scaling-proxy", "--pmc-traffic", "off"])
    # the default at N > 1 (round 5): the SAME job split over the ranks -- the rays of one rank alone, half of the paths per GPU, the same workload named
    compact, fixed = run_bench(["--gpus", "2", "--share-device", "--dist-backend", "gloo"])
    assert fixed["n_gpus"] == 2 and fixed["config"]["frame_finite_and_lit"] and fixed["config"]["parallelism"].endswith("x2")
    assert fixed["scaling"] == "strong" and compact["scaling"] == "strong" and fixed["config"]["rays_per_step"] == pytest.approx(one["config"]["rays_per_step"], rel=0.02)
    assert fixed["ranks"]["paths_per_gpu_per_step"] * 2 == pytest.approx(one["config"]["pixel_samples"] / one["steps"], rel=0.01)
    assert fixed["config"]["workload"] == one["config"]["workload"] and compact["config"]["workload"] == one["config"]["workload"][:420]
    assert len(fixed["ranks"]["ms_per_step"]) == 2 and fixed["ranks"]["gather_ms"] >= 0.0 and fixed["ranks"]["steps_per_pass"] == 2 and compact["ranks"]["passes"] == 1
    # what lets a reader trust the N > 1 line: the ranks the group really has, where they sit, and the 2-rank frame = the 1-rank frame
    ranks = fixed["ranks"]
    assert ranks["world_seen"] == {"get_world_size": 2, "all_reduce_of_ones": 2} and compact["ranks"]["world_seen"]["all_reduce_of_ones"] == 2
    assert len(ranks["devices"]) == 2 and {d["rank"] for d in ranks["devices"]} == {0, 1} and len({d["pid"] for d in ranks["devices"]}) == 2
    assert ranks["distinct_devices"] in (1, None) and ranks["backend"] == "gloo" and isinstance(ranks["rccl_version"], str)      # --share-device: both ranks on the one GPU
    assert ranks["tile_split_probe"]["identical"] is True and ranks["tile_split_probe"]["pixels_differing_from_one_rank"] == 0 and compact["ranks"]["tile_split_probe_identical"] is True
    # a warm-up that does not fill a pass (3 timed steps: batch 1; 4 timed + 1 warm-up: batch 2 with a remainder pass)
    _, odd = run_bench(["--gpus", "2", "--share-device", "--dist-backend", "gloo", "--steps", "4", "--warmup", "1"])
    assert odd["steps"] == 4 and odd["ranks"]["steps_per_pass"] == 2 and odd["config"]["rays_per_step"] == pytest.approx(one["config"]["rays_per_step"], rel=0.02)
    # weak scaling, the opt-in: each rank traces the per-GPU share of one rank alone, on its half of the tiles -> twice the paths and about twice the rays per step
    _, two = run_bench(["--gpus", "2", "--share-device", "--dist-backend", "gloo", "--weak"])
    assert two["config"]["rays_per_step"] == pytest.approx(2.0 * one["config"]["rays_per_step"], rel=0.02) and two["scaling"] == "weak"


def test_bench_the_drivers_eight_rank_command_shape_on_one_device():
    """The driver's 8-GPU command (--gpus 8 --steps 20 --warmup 5) end to end with every rank on this box's one device and gloo in place of RCCL: 8 ranks seen, the
    8-rank frame equal to the 1-rank frame, exactly 20 steps timed as 4 passes of 5."""
    compact, eight = run_bench(["--gpus", "8", "--share-device", "--dist-backend", "gloo", "--steps", "20", "--warmup", "5", "--width", "320", "--height", "184"])
    assert eight["n_gpus"] == 8 and eight["steps"] == 20 and eight["warmup"] == 5 and eight["config"]["frame_finite_and_lit"]
    assert eight["ranks"]["world_seen"] == {"get_world_size": 8, "all_reduce_of_ones": 8} and len(eight["ranks"]["devices"]) == 8
    assert eight["ranks"]["passes"] * eight["ranks"]["steps_per_pass"] == 20 and eight["ranks"]["tile_split_probe"]["identical"] is True
    assert compact["ranks"]["tile_split_probe_identical"] is True and len(compact["ranks"]["ms_per_step"]) == 8


def test_bench_gathers_through_the_host_when_the_first_gather_fails():
    """The N > 1 line must not be lost to a transport problem: a gather that raises (injected on every rank) is followed by a gloo group and a host-staged gather for the
    rest of the run; the line says which transport delivered the frame."""
    _, one = run_bench(["--no-textured", "--no-scaling-proxy", "--pmc-traffic", "off"])
    compact, two = run_bench(["--gpus", "2", "--share-device", "--dist-backend", "gloo"], extra_env={"HIPR_BENCH_TEST_FAIL_GATHER": "1"})
    assert two["config"]["frame_finite_and_lit"] and "the first gather failed" in two["ranks"]["gather_transport"] and "failed" in compact["ranks"]["gather_transport"]
    assert two["config"]["rays_per_step"] == pytest.approx(one["config"]["rays_per_step"], rel=0.02)

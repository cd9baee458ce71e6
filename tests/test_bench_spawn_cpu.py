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

"""The statistical tolerance of the lit images, under test (SURVEY.md 8(d) RMSE protocol).

The shade kernel is built like the reference's --use_fast_math PTX (hardware sin / cos / pow / divide), the oracle with libm: single paths take another
discrete decision now and then, so the lit images of the two differ at equal seed -- by zero-mean noise, as profiles/r03_rmse_protocol_*.json showed once.
Here that is an assertion, on the 17.5 k- and the 251 k-triangle atrium at 160 x 90 x 64 spp (seconds of oracle on the box's host cores):

  * no bias: per channel, |mean over the pixels of (device - oracle)| <= 3 standard errors of that mean;
  * same distance to the truth: RMSE(device, converged) and RMSE(oracle, converged) agree within 1 %, where `converged` is the oracle's image of 4096
    DISJOINT accumulations [256, 4352) committed under profiles/converged/ (tools/converged_reference.py) -- a bias would put one of them farther away;
  * and the equal-seed RMSE itself stays under twice what was measured when the test was written.
Metric: sqrt(mean over pixels and channels of the squared difference), and the reference's own ImageOperations::Compare::rms
(extensions/ImageOperations/ImageOperations/Compare.h:23-43: the luminance of the absolute difference) beside it.
"""
import json
from pathlib import Path

import numpy as np
import pytest

from bifrost3d_amd.host import Scene

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent
W, H, SPP, BOUNCES = 160, 90, 64, 4


@pytest.fixture(scope="module")
def ctx():
    from bifrost3d_amd.renderer import Context
    c = Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def oracle_q():
    from oracle_bindings import get_oracle
    return get_oracle(True)


def compare_rms(a, b):
    d = np.abs(a - b)
    luminance = 0.2126 * d[..., 0] + 0.7152 * d[..., 1] + 0.0722 * d[..., 2]
    return float(np.sqrt(np.mean(luminance ** 2)))


def rmse(a, b):
    return float(np.sqrt(np.mean((a - b) ** 2)))


# (atrium triangle target, converged file stem, equal-seed RMSE measured on the MI355X when the test was written)
CASES = [(20000, "atrium17502_160x90_acc256_4352", 7.4e-4), (260000, "atrium_160x90_acc256_4352", 2.4e-3)]      # profiles/r04_image_metrics.txt


@pytest.mark.parametrize("target, stem, measured", CASES, ids=["atrium17k", "atrium251k"])
def test_equal_seed_difference_is_unbiased_noise(ctx, oracle_q, verify_ctx, target, stem, measured):
    meta = json.loads((ROOT / "profiles" / "converged" / (stem + ".json")).read_text())
    converged = np.load(ROOT / "profiles" / "converged" / (stem + ".npy")).astype(np.float64)
    scene = Scene("atrium", param0=target, param1=1)
    assert int(scene.desc.triangle_count) == meta["triangles"] and meta["bounces"] == BOUNCES and meta["accumulations"][0] >= SPP      # disjoint from [0, SPP)
    ctx.upload_scene(scene)
    ctx.set_frame(W, H, 0, 1, 32)
    for a in range(0, SPP, 32):
        ctx.render_pass(scene.camera(W, H, accumulations=a, max_bounce_count=BOUNCES))
    ctx.synchronize()
    gpu = ctx.read_accumulation()[..., :3].astype(np.float64)
    cpu, _, _ = oracle_q.render(scene.desc, scene.state, scene.camera(W, H, max_bounce_count=BOUNCES), W, H, SPP, use_bvh=ctx.oracle_search())
    cpu = cpu[..., :3].astype(np.float64)
    assert np.isfinite(gpu).all()

    d = (gpu - cpu).reshape(-1, 3)
    mean, standard_error = d.mean(axis=0), d.std(axis=0) / np.sqrt(d.shape[0])
    equal_seed = rmse(gpu, cpu)
    to_truth_device, to_truth_oracle = rmse(gpu, converged), rmse(cpu, converged)
    print(f"STATISTICS {stem}: equal-seed RMSE {equal_seed:.3e} (Compare::rms {compare_rms(gpu, cpu):.3e}), mean radiance {cpu.mean():.4f}; mean signed difference "
          f"{mean} +- {standard_error} ({np.abs(mean) / standard_error} sigma); RMSE to the converged image: device {to_truth_device:.6f}, oracle {to_truth_oracle:.6f} "
          f"(ratio {to_truth_device / to_truth_oracle:.5f})")
    assert np.all(np.abs(mean) <= 3.0 * standard_error), (mean, standard_error)
    assert abs(to_truth_device - to_truth_oracle) <= 0.01 * to_truth_oracle, (to_truth_device, to_truth_oracle)
    # two estimators of one integral at 64 spp: their distance to the truth is the Monte Carlo noise, an order of magnitude above their mutual difference
    assert equal_seed <= 0.25 * to_truth_oracle, (equal_seed, to_truth_oracle)
    # Round 5: the bar is no longer "twice what was measured when the test was written". Leg A: the verification build of the same source renders this frame
    # bit-identical to the oracle (f64 transcendentals), so the code is exact. Leg B: the product's equal-seed distance to the oracle is its distance to the
    # verification build ON THE DEVICE -- the divergence of paths under the shade unit's fast arithmetic and nothing else -- and that stays a small fraction of the
    # frame's Monte Carlo noise (measured: 2 %; `measured` is kept in the table for the record: 7.4e-4 / 2.4e-3 in round 4, 7.5e-4 / 2.4e-3 in round 5).
    from conftest import verification_build_equals_oracle
    exact_image = verification_build_equals_oracle(verify_ctx, oracle_q, scene, W, H, SPP, BOUNCES, stem).astype(np.float64)
    on_device = rmse(gpu, exact_image)
    print(f"STATISTICS {stem}: product vs verification build on the device {on_device:.3e}")
    assert abs(on_device - equal_seed) <= 0.1 * on_device + 1e-7, (on_device, equal_seed)
    assert equal_seed <= 0.05 * to_truth_oracle, (equal_seed, to_truth_oracle)


def test_the_headline_frame_at_256_spp_exact_mode_meets_the_bound_and_fast_mode_holds_its_recorded_divergence(ctx, oracle_q, verify_ctx):
    """BASELINE.json config 4 (the 251 k-triangle atrium) at 160 x 90 x 256 spp, the frame bench.py's rmse_vs_oracle reports:
      * the renderer in its EXACT arithmetic mode against the oracle: RMSE <= 1e-3, north_star's bound, asserted as written -- it is 0, every pixel bit-identical;
      * the FAST mode against the exact mode on the device: RMSE <= 1.5 x the recorded 1.26e-3 (profiles/r05_verify_probe_*.json, BENCH_r05), Compare::rms
        <= 1.5 x 9.3e-4, at most 2 % of the pixels further than 1e-2 relative -- a change that doubles the fast arithmetic's path divergence fails here."""
    spp = 256
    scene = Scene("atrium", param0=260000, param1=1)

    def render(context):
        context.upload_scene(scene)
        context.set_frame(W, H, 0, 1, 32)
        for a in range(0, spp, 32):
            context.render_pass(scene.camera(W, H, accumulations=a, max_bounce_count=BOUNCES))
        context.synchronize()
        return context.read_accumulation()[..., :3].astype(np.float64)

    fast, exact = render(ctx), render(verify_ctx)
    before = oracle_q.lib.oracle_set_f64_transcendentals(1)
    try:
        cpu, _, _ = oracle_q.render(scene.desc, scene.state, scene.camera(W, H, max_bounce_count=BOUNCES), W, H, spp, use_bvh=verify_ctx.oracle_search())
    finally:
        oracle_q.lib.oracle_set_f64_transcendentals(before)
    cpu = cpu[..., :3].astype(np.float64)
    exact_rmse, identical = rmse(exact, cpu), float((exact == cpu).all(axis=-1).mean())
    relative = (np.abs(fast - exact) / (np.abs(exact) + 1e-3)).max(axis=-1)
    print(f"STATISTICS headline 160x90x256: exact mode vs oracle RMSE {exact_rmse:.3e} ({identical:.6f} of the pixels bit-identical); fast vs exact on the device RMSE "
          f"{rmse(fast, exact):.3e}, Compare::rms {compare_rms(fast, exact):.3e}, pixels beyond 1e-2 relative {float((relative > 1e-2).mean()):.4f}")
    assert exact_rmse <= 1e-3 and identical >= 0.9999
    assert rmse(fast, exact) <= 1.5 * 1.262e-3 and compare_rms(fast, exact) <= 1.5 * 9.33e-4
    assert float((relative > 1e-2).mean()) <= 0.02

"""GPU tests of the denoising stage (csrc/denoiser.hip) through the C-ABI of include/hipr_denoiser_c.h.

There is no reference output for this stage (the reference's is NVIDIA's closed DL denoiser; include/hipr_denoiser_c.h says what
stands in for it), so parity is against the CPU restatement oracle/denoiser.cpp -- same taps, same order of the sums -- with a
tolerance of 2e-5 relative: the two sides differ in expf / log2f by an ulp or so and in nothing else. The rendered-frame test then
holds the whole backend (path tracing pass + albedo feature pass + filter) to what it is for."""
import numpy as np
import pytest

import denoiser_oracle
from bifrost3d_amd import capi, denoiser
from test_gpu_parity import render_gpu

pytestmark = pytest.mark.gpu

TOLERANCE = 2e-5


@pytest.fixture(scope="module")
def ctx():
    from bifrost3d_amd.renderer import Context
    c = Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def cornell():
    from bifrost3d_amd.host import Scene
    return Scene("cornell")


@pytest.fixture(scope="module")
def dn():
    d = denoiser.Denoiser(0)
    yield d
    d.close()


def close_enough(gpu, cpu, tolerance=TOLERANCE):
    err = np.abs(gpu[..., :3] - cpu[..., :3]) / (np.abs(cpu[..., :3]) + 1e-4)
    return float(err.max()) <= tolerance, float(err.max())


@pytest.mark.parametrize("width,height", [(96, 64), (33, 17), (257, 131), (5, 3), (1, 1)])
def test_filter_matches_oracle(dn, width, height):
    """Frame sizes that are not multiples of the 32 x 8 tile, frames smaller than the filter footprint, a single pixel."""
    noisy, albedo, *_ = denoiser_oracle.test_frames(width, height, seed=width * 31 + height)
    s = denoiser.default_settings()
    gpu, cpu = dn.filter_host(noisy, albedo, s), denoiser_oracle.denoise(noisy, albedo, s)
    ok, worst = close_enough(gpu, cpu)
    assert ok, worst
    assert np.all(gpu[..., 3] == 1.0)


@pytest.mark.parametrize("changes", [dict(iterations=1), dict(iterations=8), dict(sigma_albedo=0.02), dict(sigma_luminance=0.25), dict(albedo_floor=0.3), dict(sigma_albedo=10.0, sigma_luminance=100.0)])
def test_filter_matches_oracle_across_settings(dn, changes):
    noisy, albedo, *_ = denoiser_oracle.test_frames(120, 70, seed=11)
    s = denoiser.default_settings()
    for k, v in changes.items():
        setattr(s, k, v)
    ok, worst = close_enough(dn.filter_host(noisy, albedo, s), denoiser_oracle.denoise(noisy, albedo, s))
    assert ok, (changes, worst)


def test_full_hd_frame_properties(dn):
    """1920 x 1080 (the size the backend runs at): too large for a per-pixel oracle pass in a test, so the size-independent properties --
    a constant frame is a fixed point, filtering commutes with mirroring the frame, and the mean radiance is kept."""
    h, w = 1080, 1920
    constant = np.ones((h, w, 4), np.float32) * np.float32([0.3, 1.7, 0.02, 1.0])
    albedo = np.ones((h, w, 4), np.float32) * np.float32([0.6, 0.6, 0.1, 1.0])
    out = dn.filter_host(constant, albedo)
    assert np.allclose(out[..., :3], constant[..., :3], rtol=3e-6, atol=0)
    noisy, albedo, *_ = denoiser_oracle.test_frames(w, h, seed=3)
    plain = dn.filter_host(noisy, albedo)
    mirrored = dn.filter_host(noisy[:, ::-1].copy(), albedo[:, ::-1].copy())[:, ::-1]
    # the taps of a mirrored frame are summed in the opposite order: equal up to f32 summation order
    assert np.allclose(plain[..., :3], mirrored[..., :3], rtol=2e-5, atol=1e-7)
    assert float(plain[..., :3].mean()) == pytest.approx(float(noisy[..., :3].mean()), rel=0.02)


def test_process_runs_the_command_list(dn):
    """hipr_denoiser_process on half4 device frames with pitches above the width: the filtered frame against the oracle on the same
    half-rounded inputs, the two debug views, and the reuse of the last filtered frame when update_filtered is 0."""
    import torch
    width, height, pitch = 100, 60, 128
    noisy, albedo, *_ = denoiser_oracle.test_frames(width, height, seed=5)
    to_device = lambda a: torch.from_numpy(np.pad(a.astype(np.float16), ((0, 0), (0, pitch - width), (0, 0)))).to(dn.device)
    noisy_d, albedo_d = to_device(noisy), to_device(albedo)
    out_d = torch.full((height, pitch + 7, 4), -1.0, dtype=torch.float16, device=dn.device)
    s = denoiser.default_settings()
    dn.process(noisy_d, albedo_d, out_d, width, height, s, update_filtered=True, show=denoiser.SHOW_FILTERED)
    out = out_d.cpu().numpy().astype(np.float32)
    expected = denoiser_oracle.denoise(noisy.astype(np.float16).astype(np.float32), albedo.astype(np.float16).astype(np.float32), s)
    assert np.allclose(out[:, :width, :3], expected[..., :3], rtol=2e-3, atol=1e-4)        # half4 output: 11 bits
    assert np.all(out[:, :width, 3] == 1.0)
    assert np.all(out[:, width:] == -1.0)                                                    # nothing written beyond the frame
    filtered = out[:, :width].copy()

    dn.process(noisy_d, albedo_d, out_d, width, height, s, update_filtered=False, show=denoiser.SHOW_NOISE)
    assert np.array_equal(out_d.cpu().numpy()[:, :width, :3], noisy.astype(np.float16)[..., :3])
    dn.process(noisy_d, albedo_d, out_d, width, height, s, update_filtered=False, show=denoiser.SHOW_ALBEDO)
    assert np.array_equal(out_d.cpu().numpy()[:, :width, :3], albedo.astype(np.float16)[..., :3])

    # other inputs, update_filtered = 0: the filtered frame of the last update is shown again (the reference's denoised_pixels_buffer)
    other = to_device(noisy * 3.0)
    dn.process(other, albedo_d, out_d, width, height, s, update_filtered=False, show=denoiser.SHOW_FILTERED)
    assert np.array_equal(out_d.cpu().numpy().astype(np.float32)[:, :width], filtered)
    dn.process(other, albedo_d, out_d, width, height, s, update_filtered=True, show=denoiser.SHOW_FILTERED)
    assert not np.array_equal(out_d.cpu().numpy().astype(np.float32)[:, :width], filtered)


def test_process_rejects_bad_arguments(dn):
    import torch
    frame = torch.zeros((8, 16, 4), dtype=torch.float16, device=dn.device)
    s = denoiser.default_settings()
    with pytest.raises(capi.HiprError, match="pitch"):
        dn.process(frame, frame, frame, 32, 8, s)
    bad = denoiser.default_settings(); bad.iterations = 0
    with pytest.raises(capi.HiprError, match="settings"):
        dn.process(frame, frame, frame, 16, 8, bad)
    with pytest.raises(capi.HiprError, match="show"):
        dn.process(frame, frame, frame, 16, 8, s, show=7)
    with pytest.raises(capi.HiprError):
        dn.filter_host(np.zeros((4, 4, 4), np.float32), np.zeros((4, 4, 4), np.float32), bad)


def test_denoised_cornell_is_closer_to_the_converged_image(ctx, dn, cornell):
    """The backend's data flow through the C-ABI: 4 accumulations of path tracing, 4 of the HIPR_ENTRY_DENOISER_ALBEDO feature pass,
    the filter. The filtered frame has to be closer to a 1024 spp render than the noisy one (tone compressed, so the lamp does not decide)."""
    w, h = 160, 90
    noisy, _ = render_gpu(ctx, cornell, w, h, 4, 4)
    ctx.set_entry_point(capi.ENTRY_DENOISER_ALBEDO)
    try:
        albedo, _ = render_gpu(ctx, cornell, w, h, 4, 4)
    finally:
        ctx.set_entry_point(capi.ENTRY_PATH_TRACING)
    converged, _ = render_gpu(ctx, cornell, w, h, 1024, 4, samples_per_pass=32)
    assert 0.0 <= float(albedo[..., :3].min()) and float(albedo[..., :3].max()) <= 1.0 + 1e-6
    filtered = dn.filter_host(noisy.astype(np.float32), albedo.astype(np.float32))
    compress = lambda a: a[..., :3] / (1.0 + a[..., :3])
    mse = lambda a: float(np.mean((compress(a.astype(np.float64)) - compress(converged.astype(np.float64))) ** 2))
    print(f"DENOISER-METRIC cornell {w}x{h}, 4 spp against 1024 spp: noisy mse {mse(noisy):.5f}, filtered mse {mse(filtered):.5f}")
    assert mse(filtered) < 0.4 * mse(noisy)
    assert np.isfinite(filtered).all()


def test_second_running_mean_survives_between_feature_passes(ctx, cornell):
    """hipr_use_scratch_accumulation(ctx, 2): the albedo feature image is a running mean NEXT to the camera's, kept across calls while path tracing
    passes run in between (the backend alternates the two every frame). Interleaved, both means must equal the ones accumulated alone."""
    w, h = 64, 36
    radiance_alone, _ = render_gpu(ctx, cornell, w, h, 3, 4)
    ctx.set_entry_point(capi.ENTRY_DENOISER_ALBEDO)
    albedo_alone, _ = render_gpu(ctx, cornell, w, h, 3, 4)
    ctx.set_entry_point(capi.ENTRY_PATH_TRACING)

    ctx.set_frame(w, h)
    for a in range(3):
        cam = cornell.camera(w, h, accumulations=a, max_bounce_count=4)
        ctx.render_pass(cam, synchronize=True)
        ctx.set_entry_point(capi.ENTRY_DENOISER_ALBEDO)
        ctx.use_scratch_accumulation(2)
        ctx.render_pass(cam, synchronize=True)
        if a == 2:
            albedo_interleaved = ctx.read_accumulation()
        ctx.use_scratch_accumulation(0)
        ctx.set_entry_point(capi.ENTRY_PATH_TRACING)
    radiance_interleaved = ctx.read_accumulation()
    assert np.array_equal(radiance_interleaved, radiance_alone)
    assert np.array_equal(albedo_interleaved, albedo_alone)
    assert not np.array_equal(albedo_alone, radiance_alone)

"""Runs tests/native/renderer_test: the C++ HIPRenderer::Renderer tests that mirror the reference's
RendererFixture (extensions/OptiXRenderer/tests/OptiXRendererTests/RendererTest.h). The binary is built by
bifrost3d_amd/Makefile (see __graft_entry__.build)."""
import subprocess
from pathlib import Path

import pytest

BINARY = Path(__file__).resolve().parent / "native" / "renderer_test"


def run(flag):
    assert BINARY.exists(), f"{BINARY} is missing: run __graft_entry__.build()"
    p = subprocess.run([str(BINARY), flag], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-4000:] + p.stderr[-4000:]
    return p.stdout


def test_host_side_renderer_cases():
    out = run("--cpu")
    assert "flattened_cornell_box_matches_the_scene_builder" in out and " 0 failed." in out
    for name in ("distribution2D_non_constant_function", "infinite_area_light_consistent_PDF_and_evaluate", "infinite_area_light_diffuse_integrates_to_white",
                 "infinite_area_light_PDF_resampling", "per_pixel_PDF_reconstructs_the_solid_angle_PDF"):
        assert f"[       OK ] EnvironmentFixture.{name}" in out, out[-4000:]
    for name in ("obj_shapes_materials_and_nodes", "single_shape_is_its_own_root_and_missing_files_fail", "viewer_defaults_place_camera_light_and_clip_planes"):
        assert f"[       OK ] LoaderFixture.{name}" in out, out[-4000:]
    for name in ("json_reader", "png_round_trip", "triangle_is_mirrored_into_the_left_handed_frame", "hierarchy_transforms_and_residual_scaling",
                 "materials_and_texture_channel_regrouping", "binary_container_and_vertex_colours", "malformed_files_create_nothing"):
        assert f"[       OK ] glTFFixture.{name}" in out, out[-4000:]
    assert "[       OK ] CompositorFixture.cameras_carry_the_effects_preset" in out, out[-4000:]
    for name in ("transform_applies_translation_rotation_and_scale", "transform_matrix_representation", "quaternion_axis_helpers_and_matrix_representation", "quaternion_look_in",
                 "octahedral_normal_encode_decode", "blue_noise_points_fill_exactly_the_requested_range"):
        assert f"[       OK ] MathFixture.{name}" in out, out[-4000:]


@pytest.mark.gpu
def test_reference_renderer_cases_on_gpu():
    out = run("--gpu")
    for name in ("render_background_color", "render_tint", "render_auxiliary_tint", "render_returns_the_iteration_count",
                 "scene_changes_restart_accumulation", "render_target_pitch_is_respected", "cornell_box_through_the_renderer_matches_the_c_abi", "adaptor_presents_the_flipped_viewport"):
        assert f"[       OK ] RendererFixture.{name}" in out, out[-4000:]
    assert "[       OK ] LoaderFixture.loaded_obj_renders_through_the_renderer" in out, out[-4000:]
    assert "[       OK ] glTFFixture.loaded_gltf_renders_through_the_renderer" in out, out[-4000:]
    assert "[       OK ] CompositorFixture.composites_two_cameras_into_their_viewports" in out, out[-4000:]
    assert "[       OK ] EnvironmentFixture.renderer_shows_the_environment_map_behind_an_empty_scene" in out, out[-4000:]

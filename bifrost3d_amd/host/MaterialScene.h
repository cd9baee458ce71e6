// MaterialScene.h -- SimpleViewer's material test scene (BASELINE.json config 3), built in the Bifrost managers the way the
// viewer builds it: apps/SimpleViewer/Scenes/Material.cpp:25-48,143-188 and Scenes/Utils.cpp:27-105.
#pragma once

#include "Bifrost.h"

#include <string>

namespace ViewerScenes {

// Utils.cpp:27-62: a thin-walled plane of `floor_size` with a 2 x 2 black / white checker texture repeated so that one checker is
// `checker_size` wide; the texture's alpha carries the roughness (smooth black, rough white), nearest magnification.
Bifrost::Scene::SceneNode create_checkered_floor(float floor_size, float checker_size);

// The shader ball the viewer loads from Resources/Shaderball.gltf is an asset of the reference and does not travel. When
// `shader_ball_path` is empty a procedural stand-in of the same structure and size takes its place: a node with the children
// "Node5" (outer surface, a 96 x 64 revolved sphere, 12 096 triangles against the asset's 11 952) and "Node2" (rubber inside /
// base, 13 520 triangles against 13 332), one unit across like the asset. With a path the asset itself is loaded and pruned to
// those two nodes exactly as Utils.cpp:74-103 does.
Bifrost::Scene::SceneNode load_shader_ball(const std::string& shader_ball_path, Bifrost::Assets::Material material);

// Material.cpp:143-188: camera at (0, 5.5, -18.5) looking at (0, 0.5, 0), directional light (3, 2.9, 2.5) from (20, 20, -20),
// the checkered floor at y = -1 and seven shader balls of scale 2, 2.4 apart, whose outer materials blend from a rough teal
// dielectric to polished gold (Material.cpp:30-47). `coat`: every blended material gets coat 1 with coat roughness 0.7, the
// coated variant of the reference's shading tests (ShadingModelTestUtils.h:39-44).
void create_material_scene(Bifrost::Scene::CameraID camera_ID, Bifrost::Scene::SceneNode root_node, const std::string& shader_ball_path, bool coat);

// The transmissive part of the viewer's glass scene (apps/SimpleViewer/Scenes/Glass.cpp:27-140): camera at (0, 3, -10) looking at
// (0, 1, 0), the directional light and the large sphere light, the checkered floor, a frosted glass shader ball (roughness 0.25),
// the magnifying glass (a smooth glass lens: a revolved sphere flattened to a tenth, with its gold handle as a stretched box)
// and a diamond. `diamond_path` empty: a faceted stand-in (an 8 x 2 revolved sphere pulled into a brilliant's proportions) with
// the diamond's specularity takes the place of Resources/Diamond.glb; otherwise the asset is loaded and its outer surface
// re-materialed as Glass.cpp:116-125 does. The pool of water of the original is left out.
void create_glass_scene(Bifrost::Scene::CameraID camera_ID, Bifrost::Scene::SceneNode root_node, const std::string& shader_ball_path, const std::string& diamond_path);

// The viewer's opacity scene (apps/SimpleViewer/Scenes/Opacity.h:27-104): camera at (0, 1, -6), the checkered floor (teal, roughness 0.3),
// a sphere light of power 50 inside a unit box whose material is a CUT-OUT driven by a 17 x 17 Alpha8 grid texture (nearest lookup),
// and two thin-walled planes of coverage 0.75 in front of it. The scene that exercises stochastic coverage rejection
// (ORS/MonteCarlo.cu:152-164) and the shadow any-hit transmittance product (:278-285).
void create_opacity_scene(Bifrost::Scene::CameraID camera_ID, Bifrost::Scene::SceneNode root_node, unsigned quads_per_edge = 1);

// The procedural atrium (Scenes::create_atrium, the Sponza-class stand-in of BASELINE configs[3] / [4]) built in the Bifrost managers
// instead of directly in a SceneBuilder: the same generator calls arrive as Meshes, Materials, MeshModels on scene nodes and light
// sources under `root_node`, so that the scene reaches the kernels the way a host application's scene does -- through
// HIPRenderer::Renderer::handle_updates. Sets the camera transform; returns the viewer clip planes and bounce count in `camera`.
struct AtriumCamera { float near_plane, far_plane, field_of_view; unsigned max_bounce_count; };
AtriumCamera create_atrium_scene(Bifrost::Scene::CameraID camera_ID, Bifrost::Scene::SceneNode root_node, unsigned target_triangles, unsigned seed);

} // namespace ViewerScenes

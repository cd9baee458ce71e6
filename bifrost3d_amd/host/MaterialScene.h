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

} // namespace ViewerScenes

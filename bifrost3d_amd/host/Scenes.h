// host/Scenes.h -- see Scenes.cpp.
#pragma once

#include "SceneBuilder.h"

namespace HIPRenderer {
namespace Scenes {

MeshData plane(unsigned quads_per_edge, bool normals, bool texcoords);
MeshData box(unsigned quads_per_edge, Vector3f size, bool tints, bool texcoords = false);

// `wall_quads_per_edge` > 1 tessellates the five walls (same surfaces, more triangles): exercises the BVH2 kernels, which serve
// scenes between the exhaustive-search and the wide-BVH size ranges.
void create_cornell_box(SceneBuilder& scene, unsigned wall_quads_per_edge = 1);
void create_atrium(SceneBuilder& scene, unsigned target_triangles, unsigned seed, bool textured = false);      // textured: AtriumScene.h
void create_quad_scene(SceneBuilder& scene, unsigned width, unsigned height);
void create_empty_ortho_scene(SceneBuilder& scene, unsigned width, unsigned height, RGB environment_tint);

} // namespace Scenes
} // namespace HIPRenderer

// ObjLoader.h -- Wavefront OBJ + MTL ingestion into the Bifrost scene managers.
//
// Same interface and the same mapping rules as the reference's loader (extensions/ObjLoader/ObjLoader/ObjLoader.h:17-30,
// ObjLoader.cpp:131-299); the OBJ / MTL text parser underneath is this repository's own (the reference delegates to the
// third-party tinyobjloader 1.x it vendors, whose documented behaviour is followed: shapes split at `o` / `g`, polygons
// triangulated as fans, negative indices relative to the end, `usemtl` applies per face).
#pragma once

#include "../Bifrost.h"

#include <string>

namespace ObjLoader {

typedef Bifrost::Assets::Image (*ImageLoader)(const std::string& filename);

// Returns the root node of the loaded models (one node per shape; a common parent named after the file when there are several),
// SceneNode::invalid() when the file cannot be read. `image_loader` may be null: textures are then skipped like images that fail to load.
Bifrost::Scene::SceneNode load(const std::string& filename, ImageLoader image_loader);

bool file_supported(const std::string& filename);

} // namespace ObjLoader

// glTFLoader.h -- glTF 2.0 (.gltf / .glb) ingestion into the Bifrost scene managers.
//
// Same interface and mapping rules as the reference's loader (extensions/glTFLoader/glTFLoader/glTFLoader.h:27-29,
// glTFLoader.cpp:38-738): one Mesh per TRIANGLES primitive, one MeshModel per (node, primitive), pbrMetallicRoughness onto
// the Default shading model, the right- to left-handed flip by negating X, node matrices decomposed into Bifrost's
// translation / rotation / uniform-scale transforms with the remainder baked into a clone of the mesh, and the glTF texture
// channel layout regrouped into Bifrost's (tint + roughness, metallic, coverage) images.
//
// The reference reads the container through the third-party tinygltf 2.x + nlohmann::json headers it vendors and decodes
// images through stb_image; here the JSON reader (../Json.h), the GLB / buffer / accessor handling and the PNG decoder
// (../ImageIO/PngImage.h) are this repository's own. Images are decoded through a callback so that an application can plug
// a decoder for the formats this image cannot read (JPEG); an image that fails to decode is skipped with a warning, the
// material keeps its factors (the reference gives up on the whole file there).
#pragma once

#include "../Bifrost.h"

#include <string>

namespace glTFLoader {

// Decodes an encoded image held in memory (rows in file order, as StbImageLoader::load_from_memory); invalid image on failure.
typedef Bifrost::Assets::Image (*ImageLoader)(const std::string& name, const void* data, size_t byte_count);

// Returns the root node of the default scene (a node named "Scene root" above them when the scene has several roots),
// SceneNode::invalid() when the file cannot be read or parsed or names no default scene. A null `image_loader` selects ImageLoader::load_from_memory (PNG, JPEG, Radiance HDR).
Bifrost::Scene::SceneNode load(const std::string& filename, ImageLoader image_loader = nullptr);

bool file_supported(const std::string& filename);

} // namespace glTFLoader

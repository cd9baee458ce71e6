// SceneLoading.h -- what SimpleViewer does around a scene loaded from a file (apps/SimpleViewer/main.cpp):
// cut-out detection on coverage textures (:222-262) and the camera / light / clip plane defaults (:395-429).
#pragma once

#include "Bifrost.h"

#include <string>

namespace SceneLoading {

// A coverage image that is black / white up to soft 2x2 borders flags its materials Cutout (main.cpp:222-262).
void detect_and_flag_cutout_materials();

struct ViewerDefaults {
    Bifrost::Math::AABB scene_bounds;   // union of the models' bounding spheres (main.cpp:395-404)
    float scene_size, near_plane, far_plane;
    bool added_light;
};

// Camera at centre + size looking at the centre, a directional light of radiance 15 from (-0.1, -10, -0.1) when the scene has
// no light, near / far = size / 10000, 3 * size. `has_environment`: an environment map counts as a light source.
ViewerDefaults apply_viewer_defaults(Bifrost::Scene::SceneNode root_node, Bifrost::Scene::CameraID camera_ID, bool loaded_from_file, bool has_environment = false);

// main.cpp:85-112 load_image: the file itself, else the same name as .png, else as .jpg (the viewer also tries .tga, which no decoder here reads).
Bifrost::Assets::Image load_image(const std::string& path);

// main.cpp:331-341: the --environment-map image (any format ImageLoader reads), widened to four channels as the renderer requires
// (RGB_Float -> RGBA_Float, 8 bit -> RGBA32), as a latitude-longitude texture: linear filters, repeat in U, clamp in V.
// TextureID::invalid_UID() when the file cannot be read.
Bifrost::Assets::TextureID load_environment_map(const std::string& path);

} // namespace SceneLoading

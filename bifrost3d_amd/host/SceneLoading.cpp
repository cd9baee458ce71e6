// SceneLoading.cpp -- see SceneLoading.h.
#include "SceneLoading.h"

#include "ImageIO/ImageLoader.h"

#include <cstdio>
#include <string>
#include <vector>

#include <vector>

using namespace Bifrost;
using namespace Bifrost::Assets;
using namespace Bifrost::Math;
using namespace Bifrost::Scene;

namespace SceneLoading {

void detect_and_flag_cutout_materials() {
    enum State : unsigned char { Unprocessed, Cutout, Transparent };
    std::vector<State> image_states(Images::capacity(), Unprocessed);

    for (MeshModelID model_ID : MeshModels::get_iterable()) {
        Material material = MeshModels::get_material_ID(model_ID);
        if (material.get_ID() == MaterialID::invalid_UID()) continue;
        const TextureID coverage_texture = material.get_coverage_texture_ID();
        if (coverage_texture == TextureID::invalid_UID()) continue;
        Image coverage = Textures::get_image_ID(coverage_texture);
        if (!coverage.exists() || coverage.get_pixel_format() != PixelFormat::Alpha8) continue;

        State& image_state = image_states[coverage.get_ID()];
        if (image_state == Unprocessed) {
            const int width = int(coverage.get_width()), height = int(coverage.get_height());
            const unsigned char* pixels = coverage.get_pixels<unsigned char>();
            auto is_cutout_opacity = [](unsigned char intensity) { return intensity < 2 || 253 < intensity; };
            image_state = Cutout;
            for (int y = 0; y < height - 1; ++y)
                for (int x = 0; x < width - 1; ++x)
                    if (!is_cutout_opacity(pixels[x + y * width])) {
                        // grey: fine on a soft border next to black / white, not inside a larger grey area
                        const bool cutout_border = is_cutout_opacity(pixels[(x + 1) + y * width]) || is_cutout_opacity(pixels[x + (y + 1) * width]) ||
                                                   is_cutout_opacity(pixels[(x + 1) + (y + 1) * width]);
                        if (!cutout_border) image_state = Transparent;
                    }
        }
        if (image_state == Cutout) material.set_flags(MaterialFlag::Cutout);
    }
}

ViewerDefaults apply_viewer_defaults(SceneNode root_node, CameraID camera_ID, bool loaded_from_file, bool has_environment) {
    ViewerDefaults result = {};
    result.scene_bounds = AABB::invalid();
    for (MeshModelID model_ID : MeshModels::get_iterable()) {
        MeshModel model = model_ID;
        const AABB mesh_aabb = model.get_mesh().get_bounds();
        const Transform transform = model.get_scene_node().get_global_transform();
        const Vector3f center = transform * mesh_aabb.center();
        const float radius = magnitude(mesh_aabb.size() * transform.scale) * 0.5f;
        result.scene_bounds.grow_to_contain(AABB{center - Vector3f(radius), center + Vector3f(radius)});
    }

    if (loaded_from_file) {
        Transform camera_transform = Cameras::get_transform(camera_ID);
        camera_transform.translation = result.scene_bounds.center() + result.scene_bounds.size();
        camera_transform.look_at(result.scene_bounds.center());
        Cameras::set_transform(camera_ID, camera_transform);
    }

    const bool no_light_sources = LightSources::get_iterable().is_empty() && !has_environment;
    if (no_light_sources && loaded_from_file) {
        const Quaternionf light_direction = Quaternionf::look_in(normalize(Vector3f(-0.1f, -10.0f, -0.1f)));
        SceneNode light_node = SceneNode("Light", Transform(Vector3f::zero(), light_direction));
        LightSources::create_directional_light(light_node.get_ID(), RGB(15.0f));
        light_node.set_parent(root_node);
        result.added_light = true;
    }

    result.scene_size = magnitude(result.scene_bounds.size());
    result.near_plane = result.scene_size / 10000.0f;
    result.far_plane = result.scene_size * 3.0f;
    return result;
}

Image load_image(const std::string& path) {
    auto readable = [](const std::string& p) { FILE* f = std::fopen(p.c_str(), "rb"); if (f) std::fclose(f); return f != nullptr; };
    if (readable(path)) return ImageLoader::load(path);
    if (path.size() > 4)
        for (const char* extension : {"png", "jpg"}) {
            std::string other = path;
            other.replace(other.size() - 3, 3, extension);
            if (readable(other)) return ImageLoader::load(other);
        }
    printf("No image found at '%s'\n", path.c_str());
    return Image();
}

TextureID load_environment_map(const std::string& path) {
    Image image = ImageLoader::load(path);
    if (!image.exists()) return TextureID::invalid_UID();
    const PixelFormat format = image.get_pixel_format();
    const unsigned width = image.get_width(), height = image.get_height(), channels = unsigned(channel_count(format));
    if (channels != 4) {      // Image::change_format of the reference: missing colour channels repeat the intensity, alpha is one
        const bool is_float = format == PixelFormat::Intensity_Float || format == PixelFormat::RGB_Float;
        const size_t n = size_t(width) * height;
        Image wide;
        if (is_float) {
            std::vector<float> pixels(4 * n);
            const float* src = image.get_pixels<float>();
            for (size_t i = 0; i < n; ++i) {
                for (unsigned c = 0; c < 3; ++c) pixels[4 * i + c] = src[channels * i + (channels == 3 ? c : 0)];
                pixels[4 * i + 3] = 1.0f;
            }
            wide = Image::create2D(image.get_name(), PixelFormat::RGBA_Float, false, width, height, pixels.data());
        } else {
            std::vector<unsigned char> pixels(4 * n);
            const unsigned char* src = image.get_pixels<unsigned char>();
            for (size_t i = 0; i < n; ++i) {
                for (unsigned c = 0; c < 3; ++c) pixels[4 * i + c] = src[channels * i + (channels == 3 ? c : 0)];
                pixels[4 * i + 3] = 255;
            }
            wide = Image::create2D(image.get_name(), PixelFormat::RGBA32, Images::is_sRGB(image.get_ID()), width, height, pixels.data());
        }
        Images::destroy(image.get_ID());
        image = wide;
    }
    return Textures::create2D(image.get_ID(), MagnificationFilter::Linear, MinificationFilter::Linear, WrapMode::Repeat, WrapMode::Clamp);
}

} // namespace SceneLoading

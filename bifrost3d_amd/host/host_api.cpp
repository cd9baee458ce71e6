// host/host_api.cpp -- C entry points of the headless driver library (libhiprenderer_host.so).
//
// Python (tests, bench.py) drives the C++ host code through these; they hold no rendering logic.
// The scene a handle owns is exactly what a Bifrost host would hand to hipr_upload_scene().
#include "Scenes.h"

#include <cstdio>
#include <exception>
#include <cstring>
#include <string>

#include "HIPRenderer/PresampledEnvironment.h"
#include "HIPRenderer/Renderer.h"
#include "ImageIO/ImageLoader.h"
#include "ImageIO/PngImage.h"
#include "MaterialScene.h"
#include "ObjLoader/ObjLoader.h"
#include "RNG.h"
#include "SceneLoading.h"
#include "glTFLoader/glTFLoader.h"

#include <chrono>
#include <cmath>

using namespace HIPRenderer;

// A small procedural sky for tests and benchmarks: horizon-to-zenith gradient, dark ground, one bright sun disc. 64 x 32 RGBA float.
static void add_procedural_environment(SceneBuilder& sb) {
    using namespace Bifrost;
    const unsigned width = 64, height = 32;
    std::vector<float> pixels(size_t(width) * height * 4);
    const Math::Vector3f sun = Math::normalize(Math::Vector3f(0.4f, 0.7f, -0.3f));
    for (unsigned y = 0; y < height; ++y)
        for (unsigned x = 0; x < width; ++x) {
            const Math::Vector3f direction = Math::latlong_texcoord_to_direction({(x + 0.5f) / width, (y + 0.5f) / height});
            const float up = direction.y;
            float r = up > 0 ? 0.35f + 0.25f * (1 - up) : 0.08f, g = up > 0 ? 0.55f + 0.2f * (1 - up) : 0.07f, b = up > 0 ? 0.9f : 0.06f;
            const float alignment = Math::dot(direction, sun);
            if (alignment > 0.97f) { const float s = (alignment - 0.97f) / 0.03f; r += 60.0f * s; g += 55.0f * s; b += 45.0f * s; }
            float* p = pixels.data() + 4 * (x + size_t(y) * width);
            p[0] = r; p[1] = g; p[2] = b; p[3] = 1.0f;
        }
    const Assets::ImageID image = Assets::Images::create2D("procedural sky", Assets::PixelFormat::RGBA_Float, false, width, height, pixels.data(), pixels.size() * 4);
    const Assets::TextureID texture = Assets::Textures::create2D(image, Assets::MagnificationFilter::Linear, Assets::MinificationFilter::Linear, Assets::WrapMode::Repeat,
                                                                 Assets::WrapMode::Clamp);
    const Assets::InfiniteAreaLight light(texture);
    PresampledEnvironment presampled = presample_environment(light, 1024);
    ImageData data;
    data.width = width; data.height = height; data.format = HIPR_TEXEL_RGBA32F; data.is_sRGB = false;
    data.pixels.assign(reinterpret_cast<const uint8_t*>(pixels.data()), reinterpret_cast<const uint8_t*>(pixels.data()) + pixels.size() * 4);
    const uint32_t texture_index = sb.add_texture(data, true, false, true, true);
    sb.set_environment(texture_index, presampled.pdf_width, presampled.pdf_height, std::move(presampled.per_pixel_PDF), std::move(presampled.samples));
    Assets::Textures::destroy(texture);
    Assets::Images::destroy(image);
}

extern "C" {

// name: "cornell" | "atrium" | "quad" | "empty_ortho". variant bit 1: light the scene with a procedural environment map. variant bit 0: force every material to the Diffuse
// shading model (BASELINE.json config 2). param0/param1: atrium target triangles + seed, ortho width + height.
static void* create_material_scene(const std::string& shader_ball_path, unsigned variant);
static void* create_glass_scene(const std::string& resource_directory, unsigned variant);
static void* create_opacity_scene(unsigned quads_per_edge, unsigned variant);

// The builders hand a worker thread's exception (std::bad_alloc on a scene too large for the host) to the caller: every entry that reaches
// finalize() / refit_bvh() turns it into its error value here -- nothing may unwind through the C boundary.
static void* scene_create(const char* name, unsigned variant, unsigned param0, unsigned param1);
void* hiprh_scene_create(const char* name, unsigned variant, unsigned param0, unsigned param1) {
    try { return scene_create(name, variant, param0, param1); }
    catch (const std::exception& e) { fprintf(stderr, "hiprh_scene_create: %s\n", e.what()); Bifrost::deallocate_all(); return nullptr; }
}
static void* scene_create(const char* name, unsigned variant, unsigned param0, unsigned param1) {
    if (!name) return nullptr;
    if (!std::strncmp(name, "material", 8) && (name[8] == 0 || name[8] == ':')) return create_material_scene(name[8] ? name + 9 : "", variant);
    if (!std::strncmp(name, "glass", 5) && (name[5] == 0 || name[5] == ':')) return create_glass_scene(name[5] ? name + 6 : "", variant);
    if (!std::strcmp(name, "opacity")) return create_opacity_scene(param0, variant);
    SceneBuilder* sb = new SceneBuilder();
    std::string n = name;
    if (n == "cornell") Scenes::create_cornell_box(*sb, param0 ? param0 : 1u);   // param0: quads per wall edge
    else if (n == "atrium") Scenes::create_atrium(*sb, param0 ? param0 : 260000u, param1 ? param1 : 1u, (variant & 16u) != 0);      // variant bit 4: textured, with cut-out banners
    else if (n == "quad") Scenes::create_quad_scene(*sb, param0, param1);
    else if (n == "empty_ortho") Scenes::create_empty_ortho_scene(*sb, param0, param1, RGB(0.1f, 0.5f, 2.0f));
    else { delete sb; return nullptr; }
    if (variant & 1u) sb->force_shading_model(HIPR_SHADING_DIFFUSE);
    if (variant & 2u) add_procedural_environment(*sb);
    if (variant & 8u)   // a spot light as well (tests): a disc of radius 0.04 below the ceiling, a 35 degree cone aimed at the short box
        sb->add_light(SceneBuilder::spot_light(Vector3f(0.2f, 0.42f, -0.1f), Bifrost::Math::normalize(Vector3f(-0.35f, -1.0f, 0.15f)), RGB(1.5f, 1.2f, 0.9f), 0.04f, std::cos(35.0f * 3.14159265f / 180.0f)));
    sb->finalize();
    return sb;
}

// "material" | "material:<path to Shaderball.gltf>": SimpleViewer's material test scene (BASELINE.json config 3) built in the
// Bifrost managers and flattened, with the viewer's clip planes and its 32 bounces (apps/SimpleViewer/main.cpp:353,428-429).
// variant bit 0: all materials Diffuse; bit 2: the coated variant. The Bifrost managers are scratch space here.
static void* create_material_scene(const std::string& shader_ball_path, unsigned variant) {
    using namespace Bifrost;
    deallocate_all();
    Scene::SceneRoot scene = Scene::SceneRoot("Model scene", RGB(0.68f, 0.92f, 1.0f));
    const Scene::CameraID camera_ID = Scene::Cameras::create("Camera", scene.get_ID(), Math::Matrix4x4f::identity(), Math::Matrix4x4f::identity());
    ViewerScenes::create_material_scene(camera_ID, scene.get_root_node(), shader_ball_path, (variant & 4u) != 0);
    if (Assets::MeshModels::get_iterable().size() < 2) { deallocate_all(); return nullptr; }     // the shader ball did not load
    const SceneLoading::ViewerDefaults defaults = SceneLoading::apply_viewer_defaults(scene.get_root_node(), camera_ID, false);
    if (variant & 1u)
        for (Assets::MaterialID material_ID : Assets::Materials::get_iterable()) Assets::Material(material_ID).set_shading_model(Assets::ShadingModel::Diffuse);

    SceneBuilder* sb = new SceneBuilder();
    sb->set_environment_tint(Scene::SceneRoots::get_environment_tint(scene.get_ID()));
    flatten_bifrost_scene(*sb);
    sb->camera.transform = Scene::Cameras::get_transform(camera_ID);
    sb->camera.near_plane = defaults.near_plane;
    sb->camera.far_plane = defaults.far_plane;
    sb->camera.max_bounce_count = 32;
    deallocate_all();
    return sb;
}

// "glass" | "glass:<directory with Shaderball.gltf and Diamond.glb>": the transmissive part of the viewer's glass scene
// (apps/SimpleViewer/Scenes/Glass.cpp), flattened like the material scene; procedural stand-ins without the directory.
static void* create_glass_scene(const std::string& resource_directory, unsigned variant) {
    using namespace Bifrost;
    deallocate_all();
    Scene::SceneRoot scene = Scene::SceneRoot("Model scene", RGB(0.68f, 0.92f, 1.0f));
    const Scene::CameraID camera_ID = Scene::Cameras::create("Camera", scene.get_ID(), Math::Matrix4x4f::identity(), Math::Matrix4x4f::identity());
    const std::string shader_ball = resource_directory.empty() ? "" : resource_directory + "/Shaderball.gltf", diamond = resource_directory.empty() ? "" : resource_directory + "/Diamond.glb";
    ViewerScenes::create_glass_scene(camera_ID, scene.get_root_node(), shader_ball, diamond);
    if (Assets::MeshModels::get_iterable().size() < 5) { deallocate_all(); return nullptr; }     // floor, ball (2), lens, handle, diamond
    const SceneLoading::ViewerDefaults defaults = SceneLoading::apply_viewer_defaults(scene.get_root_node(), camera_ID, false);
    if (variant & 1u)
        for (Assets::MaterialID material_ID : Assets::Materials::get_iterable()) Assets::Material(material_ID).set_shading_model(Assets::ShadingModel::Diffuse);

    SceneBuilder* sb = new SceneBuilder();
    sb->set_environment_tint(Scene::SceneRoots::get_environment_tint(scene.get_ID()));
    flatten_bifrost_scene(*sb);
    sb->camera.transform = Scene::Cameras::get_transform(camera_ID);
    sb->camera.near_plane = defaults.near_plane;
    sb->camera.far_plane = defaults.far_plane;
    sb->camera.max_bounce_count = 32;
    deallocate_all();
    return sb;
}

// "opacity": the viewer's opacity scene (apps/SimpleViewer/Scenes/Opacity.h), the reference's test bed for cut-outs and partial
// coverage. param0 = quads per edge of the box and the planes (1 = as the viewer builds it: 24 triangles).
static void* create_opacity_scene(unsigned quads_per_edge, unsigned variant) {
    using namespace Bifrost;
    deallocate_all();
    Scene::SceneRoot scene = Scene::SceneRoot("Model scene", RGB(0.68f, 0.92f, 1.0f));
    const Scene::CameraID camera_ID = Scene::Cameras::create("Camera", scene.get_ID(), Math::Matrix4x4f::identity(), Math::Matrix4x4f::identity());
    ViewerScenes::create_opacity_scene(camera_ID, scene.get_root_node(), quads_per_edge);
    const SceneLoading::ViewerDefaults defaults = SceneLoading::apply_viewer_defaults(scene.get_root_node(), camera_ID, false);
    if (variant & 1u)
        for (Assets::MaterialID material_ID : Assets::Materials::get_iterable()) Assets::Material(material_ID).set_shading_model(Assets::ShadingModel::Diffuse);

    SceneBuilder* sb = new SceneBuilder();
    sb->set_environment_tint(Scene::SceneRoots::get_environment_tint(scene.get_ID()));
    flatten_bifrost_scene(*sb);
    sb->camera.transform = Scene::Cameras::get_transform(camera_ID);
    sb->camera.near_plane = defaults.near_plane;
    sb->camera.far_plane = defaults.far_plane;
    sb->camera.max_bounce_count = 32;
    deallocate_all();
    return sb;
}

// A model file (.obj, .gltf, .glb) set up the way SimpleViewer sets up a scene loaded from its command line
// (apps/SimpleViewer/main.cpp:330-429): sky-blue environment tint, cut-out detection, camera placed from the scene bounds,
// the default directional light when the file brings none. Textures: PNG only (ImageIO/PngImage.h). Null when the file
// cannot be loaded. The Bifrost managers are scratch space here: whatever they held is dropped.
void* hiprh_scene_load_with_environment(const char* path, const char* environment_map_path, unsigned variant);
void* hiprh_scene_load(const char* path, unsigned variant) { return hiprh_scene_load_with_environment(path, nullptr, variant); }

// The same with SimpleViewer's --environment-map (main.cpp:331-341, 533): a latitude-longitude image (Radiance .hdr, PNG or JPEG) that lights the
// scene and is seen where paths escape; it counts as a light source, so no default directional light is added (main.cpp:419-426).
static void* scene_load_with_environment(const char* path, const char* environment_map_path, unsigned variant);
void* hiprh_scene_load_with_environment(const char* path, const char* environment_map_path, unsigned variant) {
    // No exception crosses the C boundary: a crafted or truncated asset that exhausts memory inside a decoder costs the load, not the process.
    try { return scene_load_with_environment(path, environment_map_path, variant); }
    catch (const std::exception& e) { fprintf(stderr, "hiprh_scene_load: %s\n", e.what()); Bifrost::deallocate_all(); return nullptr; }
}
static void* scene_load_with_environment(const char* path, const char* environment_map_path, unsigned variant) {
    using namespace Bifrost;
    if (!path) return nullptr;
    deallocate_all();
    Scene::SceneRoot scene = Scene::SceneRoot("Model scene", RGB(0.68f, 0.92f, 1.0f));
    Scene::SceneNode loaded = Scene::SceneNode::invalid();
    if (ObjLoader::file_supported(path)) loaded = ObjLoader::load(path, SceneLoading::load_image);
    else if (glTFLoader::file_supported(path)) loaded = glTFLoader::load(path);
    if (loaded == Scene::SceneNode::invalid()) { deallocate_all(); return nullptr; }
    loaded.set_parent(scene.get_root_node());
    SceneLoading::detect_and_flag_cutout_materials();
    bool has_environment = false;
    if (environment_map_path && environment_map_path[0]) {
        const Assets::TextureID environment = SceneLoading::load_environment_map(environment_map_path);
        if (environment != Assets::TextureID::invalid_UID()) { scene.set_environment_map(environment); has_environment = true; }
    }
    const Scene::CameraID camera_ID = Scene::Cameras::create("Camera", scene.get_ID(), Math::Matrix4x4f::identity(), Math::Matrix4x4f::identity());
    const SceneLoading::ViewerDefaults defaults = SceneLoading::apply_viewer_defaults(scene.get_root_node(), camera_ID, true, has_environment);

    if (variant & 1u)
        for (Assets::MaterialID material_ID : Assets::Materials::get_iterable()) Assets::Material(material_ID).set_shading_model(Assets::ShadingModel::Diffuse);

    SceneBuilder* sb = new SceneBuilder();
    sb->set_environment_tint(Scene::SceneRoots::get_environment_tint(scene.get_ID()));
    flatten_bifrost_scene(*sb);
    sb->camera.transform = Scene::Cameras::get_transform(camera_ID);
    sb->camera.near_plane = defaults.near_plane;
    sb->camera.far_plane = defaults.far_plane;
    deallocate_all();
    return sb;
}

// The plugin path timed end to end (bench.py's `plugin_renderer` key): the atrium built in the Bifrost managers, pulled by
// HIPRenderer::Renderer::handle_updates and rendered by `calls` blocking Renderer::render() calls into a device render target, exactly what
// SimpleViewer's main loop does per frame (apps/SimpleViewer/main.cpp:298-308). out[0] = milliseconds of the timed calls, out[1] = accumulations
// reached, out[2] = flattened triangle count. max_batch: Renderer::set_max_batch_size (1 = one launch per accumulation, the reference's granularity).
// Returns 0, or a negative number when the renderer cannot be created or a call fails. The Bifrost managers are scratch space here.
static int renderer_bench(const char* data_directory, unsigned target_triangles, unsigned width, unsigned height, unsigned warmup_calls, unsigned calls, unsigned max_batch, double* out3);
int hiprh_renderer_bench(const char* data_directory, unsigned target_triangles, unsigned width, unsigned height, unsigned warmup_calls, unsigned calls, unsigned max_batch, double* out3) {
    try { return renderer_bench(data_directory, target_triangles, width, height, warmup_calls, calls, max_batch, out3); }
    catch (const std::exception& e) { fprintf(stderr, "hiprh_renderer_bench: %s\n", e.what()); Bifrost::deallocate_all(); return -5; }
}
static int renderer_bench(const char* data_directory, unsigned target_triangles, unsigned width, unsigned height, unsigned warmup_calls, unsigned calls, unsigned max_batch, double* out3) {
    using namespace Bifrost;
    if (!data_directory || !out3 || !width || !height) return -1;
    deallocate_all();
    Renderer* renderer = Renderer::initialize(0, data_directory);
    if (!renderer) return -2;
    int status = 0;
    {
        Scene::SceneRoot scene = Scene::SceneRoot("Atrium", RGB(0.68f, 0.92f, 1.0f));
        const Scene::CameraID camera_ID = Scene::Cameras::create("Camera", scene.get_ID(), Math::Matrix4x4f::identity(), Math::Matrix4x4f::identity());
        const ViewerScenes::AtriumCamera camera = ViewerScenes::create_atrium_scene(camera_ID, scene.get_root_node(), target_triangles, 1);
        Math::Matrix4x4f projection, inverse_projection;
        Scene::CameraUtils::compute_perspective_projection(camera.near_plane, camera.far_plane, camera.field_of_view, float(width) / float(height), projection, inverse_projection);
        Scene::Cameras::set_projection_matrices(camera_ID, projection, inverse_projection);
        Scene::Cameras::set_renderer_ID(camera_ID, renderer->get_renderer_ID());
        renderer->set_max_bounce_count(camera_ID, camera.max_bounce_count);
        renderer->set_max_batch_size(max_batch);
        renderer->handle_updates();
        reset_all_change_notifications();

        HiprContext* allocator = nullptr;     // the render target the presentation layer would own
        void* target = nullptr;
        if (hipr_create(0, &allocator) != HIPR_OK || hipr_device_malloc(allocator, uint64_t(width) * height * 8, &target) != HIPR_OK) status = -3;
        unsigned int accumulations = 0;
        for (unsigned i = 0; status == 0 && i < warmup_calls; ++i) accumulations = renderer->render(camera_ID, target, width, Math::Vector2i(int(width), int(height)));
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned i = 0; status == 0 && i < calls; ++i) {
            const unsigned int next = renderer->render(camera_ID, target, width, Math::Vector2i(int(width), int(height)));
            if (next != accumulations + 1) status = -4;
            accumulations = next;
        }
        out3[0] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        out3[1] = double(accumulations);
        unsigned triangles = 0;
        for (Assets::MeshModelID model : Assets::MeshModels::get_iterable()) triangles += Assets::MeshModel(model).get_mesh().get_primitive_count();
        out3[2] = double(triangles);
        if (target) hipr_device_free(allocator, target);
        if (allocator) hipr_destroy(allocator);
    }
    delete renderer;
    deallocate_all();
    return status;
}

// Decodes any image file the loaders read (PNG, JPEG, Radiance HDR) for the codec tests: returns the byte count of the pixels (8 bit or float, rows
// top-down unless `flip`), 0 when the file cannot be decoded; format: 0 = 8 bit, 1 = float.
static size_t image_load(const char* path, int flip, unsigned* width, unsigned* height, unsigned* channels, int* is_float, void* out, size_t capacity);
size_t hiprh_image_load(const char* path, int flip, unsigned* width, unsigned* height, unsigned* channels, int* is_float, void* out, size_t capacity) {
    try { return image_load(path, flip, width, height, channels, is_float, out, capacity); }
    catch (const std::exception& e) { fprintf(stderr, "hiprh_image_load: %s\n", e.what()); return 0; }
}
static size_t image_load(const char* path, int flip, unsigned* width, unsigned* height, unsigned* channels, int* is_float, void* out, size_t capacity) {
    using namespace Bifrost::Assets;
    if (!path) return 0;
    Image image = ImageLoader::load(path);
    if (!image.exists()) return 0;
    const PixelFormat format = image.get_pixel_format();
    const bool floats = format == PixelFormat::Intensity_Float || format == PixelFormat::RGB_Float || format == PixelFormat::RGBA_Float;
    const unsigned w = image.get_width(), h = image.get_height(), c = unsigned(channel_count(format));
    const size_t row = size_t(w) * c * (floats ? 4 : 1), bytes = row * h;
    if (width) *width = w;
    if (height) *height = h;
    if (channels) *channels = c;
    if (is_float) *is_float = floats;
    if (out && capacity >= bytes)
        for (unsigned y = 0; y < h; ++y) std::memcpy(static_cast<char*>(out) + size_t(y) * row, image.get_pixels<char>() + size_t(flip ? y : h - 1 - y) * row, row);
    Images::destroy(image.get_ID());
    return bytes;
}

// Decodes a PNG file into 8 bit pixels (tests: cross-check of the decoder against an independent one). Returns the byte count
// the image needs, 0 when the file cannot be decoded; copies the pixels when `capacity` is large enough. flip: bottom row first.
size_t hiprh_png_load(const char* path, int flip, unsigned* width, unsigned* height, unsigned* channels, unsigned char* out, size_t capacity) {
    using namespace Bifrost::Assets;
    if (!path) return 0;
    Image image = PngImage::load(path);
    if (!image.exists()) return 0;
    const unsigned w = image.get_width(), h = image.get_height(), c = unsigned(channel_count(image.get_pixel_format()));
    const size_t bytes = size_t(w) * h * c;
    if (width) *width = w;
    if (height) *height = h;
    if (channels) *channels = c;
    if (out && capacity >= bytes)
        for (unsigned y = 0; y < h; ++y)
            std::memcpy(out + size_t(y) * w * c, image.get_pixels<unsigned char>() + size_t(flip ? y : h - 1 - y) * w * c, size_t(w) * c);
    Images::destroy(image.get_ID());
    return bytes;
}

// Transform-only update of a built scene (SceneBuilder::update_model_transforms): model `model_index` (1-based, in creation order for the
// built-in scenes) gets the transform translation[3], rotation quaternion xyzw[4], uniform scale. Returns 1 when the BVH was refitted in
// place (topology kept: hipr_update_scene_geometry suffices), 0 when the builder rebuilt the tree (upload the scene again), -1 on error.
int hiprh_scene_move_model(void* scene, unsigned model_index, const float* translation3, const float* rotation4, float scale, double rebuild_threshold) {
    if (!scene || !translation3 || !rotation4) return -1;
    const Transform t(Vector3f(translation3[0], translation3[1], translation3[2]), Bifrost::Math::Quaternionf(rotation4[0], rotation4[1], rotation4[2], rotation4[3]), scale);
    try { return static_cast<SceneBuilder*>(scene)->update_model_transforms({{model_index, t}}, rebuild_threshold > 0.0 ? rebuild_threshold : 1.5) ? 1 : 0; }
    catch (const std::exception& e) { fprintf(stderr, "hiprh_scene_move_model: %s\n", e.what()); return -1; }
}

void hiprh_scene_destroy(void* scene) { delete static_cast<SceneBuilder*>(scene); }

const HiprSceneDesc* hiprh_scene_desc(void* scene) { return scene ? &static_cast<SceneBuilder*>(scene)->desc() : nullptr; }

int hiprh_scene_state(void* scene, HiprSceneState* out) {
    if (!scene || !out) return -1;
    *out = static_cast<SceneBuilder*>(scene)->state();
    return 0;
}

// max_bounce_count < 0 keeps the scene's own default.
int hiprh_scene_camera(void* scene, unsigned width, unsigned height, unsigned accumulations, int max_bounce_count, float pdf_scale, HiprCameraState* out) {
    if (!scene || !out || !width || !height) return -1;
    SceneBuilder* sb = static_cast<SceneBuilder*>(scene);
    CameraDescription cam = sb->camera;
    if (max_bounce_count >= 0) cam.max_bounce_count = unsigned(max_bounce_count);
    *out = make_camera_state(cam, float(width) / float(height), accumulations, pdf_scale);
    return 0;
}

// A camera state for a caller's camera (tests of the camera-ray stage against the reference's ground-truth rays, BifrostTests/Scene/
// CameraTest.h). params14: position[3], rotation quaternion x y z w, field of view, near, far, orthographic flag, ortho width, height, depth.
int hiprh_make_camera(const float* params14, unsigned width, unsigned height, unsigned accumulations, unsigned max_bounce_count, HiprCameraState* out) {
    if (!params14 || !out || !width || !height) return -1;
    CameraDescription cam;
    cam.transform = Transform(Vector3f(params14[0], params14[1], params14[2]), Bifrost::Math::Quaternionf(params14[3], params14[4], params14[5], params14[6]), 1.0f);
    cam.field_of_view = params14[7]; cam.near_plane = params14[8]; cam.far_plane = params14[9];
    cam.orthographic = params14[10] != 0.0f; cam.ortho_width = params14[11]; cam.ortho_height = params14[12]; cam.ortho_depth = params14[13];
    cam.max_bounce_count = max_bounce_count;
    *out = make_camera_state(cam, float(width) / float(height), accumulations, 0.5f);
    return 0;
}

// Stand-alone BVH build over caller triangles (tests): returns a handle owning nodes + order.
struct BvhHandle { BvhBuildResult result; };
void* hiprh_bvh_build(const HiprTriangle* triangles, unsigned count, unsigned max_depth) {
    BvhHandle* h = nullptr;
    try {
        std::vector<HiprTriangle> t(triangles, triangles + count);
        h = new BvhHandle();
        h->result = build_bvh(t, max_depth);
        return h;
    } catch (const std::exception& e) { fprintf(stderr, "hiprh_bvh_build: %s\n", e.what()); delete h; return nullptr; }
}
unsigned hiprh_bvh_node_count(void* h) { return unsigned(static_cast<BvhHandle*>(h)->result.nodes.size()); }
unsigned hiprh_bvh_max_depth(void* h) { return static_cast<BvhHandle*>(h)->result.max_depth; }
const HiprBvhNode* hiprh_bvh_nodes(void* h) { return static_cast<BvhHandle*>(h)->result.nodes.data(); }
const unsigned* hiprh_bvh_order(void* h) { return static_cast<BvhHandle*>(h)->result.order.data(); }
unsigned hiprh_bvh_wide_node_count(void* h) { return unsigned(static_cast<BvhHandle*>(h)->result.wide_nodes.size()); }
unsigned hiprh_bvh_wide_stack_entries(void* h) { return static_cast<BvhHandle*>(h)->result.wide_stack_entries; }
const HiprWideNode* hiprh_bvh_wide_nodes(void* h) { return static_cast<BvhHandle*>(h)->result.wide_nodes.data(); }
// The host's progressive multi-jittered blue-noise points (host/RNG.cpp), for the cross-check against the oracle's generator.
void hiprh_pmjbn_samples(float* out_xy, unsigned count, unsigned candidates) {
    Bifrost::Math::RNG::fill_progressive_multijittered_bluenoise_samples(reinterpret_cast<Bifrost::Math::Vector2f*>(out_xy), reinterpret_cast<Bifrost::Math::Vector2f*>(out_xy) + count, candidates);
}
void hiprh_bvh_destroy(void* h) { delete static_cast<BvhHandle*>(h); }

void hiprh_encode_octahedral(const float* normals_n3, int n, short* out_n2) {
    for (int i = 0; i < n; ++i) {
        Bifrost::Math::OctahedralNormal e = Bifrost::Math::OctahedralNormal::encode_precise({normals_n3[3 * i], normals_n3[3 * i + 1], normals_n3[3 * i + 2]});
        out_n2[2 * i] = e.encoding.x;
        out_n2[2 * i + 1] = e.encoding.y;
    }
}

// The host's environment light over an RGBA float latitude-longitude image: samples (radiance[3], PDF, direction[3], distance)
// for n random pairs, PDF(direction) of those directions, the PDF image's size and, capacity allowing, the per pixel PDF.
// Same argument list as oracle/ref/reference_api.cpp's ref_infinite_area_light, which runs the reference's own class.
int hiprh_infinite_area_light(int width, int height, const float* rgba, const float* u_n2, int n, float* out_samples_n8, float* out_pdf_n, int* out_pdf_size2,
                              float* out_per_pixel_PDF, int per_pixel_capacity) {
    using namespace Bifrost;
    const Assets::ImageID image = Assets::Images::create2D("environment", Assets::PixelFormat::RGBA_Float, false, unsigned(width), unsigned(height), rgba, size_t(width) * height * 16);
    const Assets::TextureID texture = Assets::Textures::create2D(image, Assets::MagnificationFilter::Linear, Assets::MinificationFilter::Linear, Assets::WrapMode::Repeat,
                                                                 Assets::WrapMode::Clamp);
    int status = 0;
    {
        const Assets::InfiniteAreaLight light(texture);
        for (int i = 0; i < n; ++i) {
            const Assets::LightSample s = light.sample({u_n2[2 * i], u_n2[2 * i + 1]});
            float* o = out_samples_n8 + 8 * i;
            o[0] = s.radiance.r; o[1] = s.radiance.g; o[2] = s.radiance.b; o[3] = s.PDF;
            o[4] = s.direction_to_light.x; o[5] = s.direction_to_light.y; o[6] = s.direction_to_light.z; o[7] = s.distance;
            out_pdf_n[i] = light.PDF(s.direction_to_light);
        }
        out_pdf_size2[0] = int(light.get_PDF_width()); out_pdf_size2[1] = int(light.get_PDF_height());
        if (out_per_pixel_PDF && per_pixel_capacity >= out_pdf_size2[0] * out_pdf_size2[1])
            Assets::InfiniteAreaLightUtils::reconstruct_solid_angle_PDF_sans_sin_theta(light, out_per_pixel_PDF);
        else if (out_per_pixel_PDF)
            status = 1;
    }
    Assets::Textures::destroy(texture);
    Assets::Images::destroy(image);
    return status;
}

// The host's software texture lookup (Assets::sample2D); same argument list as oracle/ref/reference_api.cpp's ref_sample2D.
void hiprh_sample2D(int format, int is_sRGB, int width, int height, const void* pixels, int byte_count, int magnification, int minification, int wrap_U, int wrap_V,
                    const float* uv_n2, int n, float* out_n4) {
    using namespace Bifrost::Assets;
    const ImageID image = Images::create2D("texture", PixelFormat(format), is_sRGB != 0, unsigned(width), unsigned(height), pixels, size_t(byte_count));
    const TextureID texture = Textures::create2D(image, MagnificationFilter(magnification), MinificationFilter(minification), WrapMode(wrap_U), WrapMode(wrap_V));
    for (int i = 0; i < n; ++i) {
        const Bifrost::Math::RGBA c = sample2D(texture, {uv_n2[2 * i], uv_n2[2 * i + 1]});
        out_n4[4 * i] = c.r; out_n4[4 * i + 1] = c.g; out_n4[4 * i + 2] = c.b; out_n4[4 * i + 3] = c.a;
    }
    Textures::destroy(texture);
    Images::destroy(image);
}

} // extern "C"

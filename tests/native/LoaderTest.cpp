// LoaderTest.cpp -- ObjLoader and the SimpleViewer scene-loading defaults against hand-written OBJ / MTL text.
// Expected values are the mapping rules of extensions/ObjLoader/ObjLoader/ObjLoader.cpp:150-296 and
// apps/SimpleViewer/main.cpp:222-262, 395-429 applied by hand to the inputs below.
#include "MiniTest.h"

#include "../../bifrost3d_amd/host/HIPRenderer/Renderer.h"
#include "../../bifrost3d_amd/host/ObjLoader/ObjLoader.h"
#include "../../bifrost3d_amd/host/SceneBuilder.h"
#include "../../bifrost3d_amd/host/SceneLoading.h"

#include <cstdio>
#include <filesystem>
#include <fstream>

#include <unistd.h>

using namespace Bifrost;
using namespace Bifrost::Assets;
using namespace Bifrost::Math;
using namespace Bifrost::Scene;

namespace {

struct LoaderFixture {
    std::filesystem::path directory;
    void SetUp() {
        deallocate_all();
        directory = std::filesystem::temp_directory_path() / ("hipr_loader_test_" + std::to_string(::getpid()));
        std::filesystem::create_directories(directory);
    }
    void TearDown() {
        std::error_code error;
        std::filesystem::remove_all(directory, error);
        deallocate_all();
    }
    bool usable() const { return true; }
    std::string write(const char* name, const char* text) {
        const std::filesystem::path path = directory / name;
        std::ofstream(path) << text;
        return path.string();
    }
};

const char* two_shapes_obj =
    "# two shapes, a quad (triangulated as a fan) and a triangle with negative indices\n"
    "mtllib scene.mtl\n"
    "o Floor\n"
    "v -1 0 -1\nv 1 0 -1\nv 1 0 1\nv -1 0 1\n"
    "vn 0 1 0\n"
    "vt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\n"
    "usemtl Rough\n"
    "f 4/4/1 3/3/1 2/2/1 1/1/1\n"   // counter-clockwise seen from above: the surface faces +y

    "o Sail\n"
    "v 0 0 0\nv 0 2 0\nv 1 1 0\n"
    "usemtl Mirror\n"
    "f -3 -2 -1\n";

const char* scene_mtl =
    "newmtl Rough\n"
    "Kd 0.5 0.25 0.125\n"
    "Ks 0.03 0.06 0.09\n"
    "Ns 14\n"
    "d 0.75\n"
    "illum 2\n"
    "map_Kd tint.png\n"
    "newmtl Mirror\n"
    "Kd 0.9 0.9 0.9\n"
    "Ks 1 1 1\n"
    "Ns 998\n"
    "illum 3\n";

// 4x4 RGBA image whose alpha is a hard-edged mask (left half 0, right half 255).
Image load_test_image(const std::string& filename) {
    if (filename.size() < 8 || filename.compare(filename.size() - 8, 8, "tint.png") != 0) return Image();
    unsigned char pixels[4 * 4 * 4];
    for (int y = 0; y < 4; ++y)
        for (int x = 0; x < 4; ++x) {
            unsigned char* p = pixels + 4 * (x + 4 * y);
            p[0] = 200; p[1] = 100; p[2] = 50; p[3] = x < 2 ? 0 : 255;
        }
    return Image::create2D(filename, PixelFormat::RGBA32, true, 4, 4, pixels);
}

} // namespace

CPU_TEST_F(LoaderFixture, obj_shapes_materials_and_nodes) {
    write("scene.mtl", scene_mtl);
    const std::string path = write("scene.obj", two_shapes_obj);
    EXPECT_TRUE(ObjLoader::file_supported(path));
    EXPECT_FALSE(ObjLoader::file_supported("scene.gltf"));

    SceneNode root = ObjLoader::load(path, load_test_image);
    EXPECT_TRUE(root != SceneNode::invalid());
    EXPECT_EQ(std::string("scene"), root.get_name());   // several shapes: a parent named after the file
    EXPECT_EQ(size_t(2), root.get_children().size());

    std::vector<MeshModel> models;
    for (MeshModelID id : MeshModels::get_iterable()) models.push_back(id);
    EXPECT_EQ(size_t(2), models.size());
    if (models.size() != 2) return;

    // Floor: a quad -> 2 triangles (fan), 4 distinct (v, vn, vt) triples, normals and texcoords present
    Mesh floor = models[0].get_mesh();
    EXPECT_EQ(std::string("Floor"), floor.get_name());
    EXPECT_EQ(2u, floor.get_primitive_count());
    EXPECT_EQ(4u, floor.get_vertex_count());
    EXPECT_TRUE(floor.get_normals() != nullptr && floor.get_texcoords() != nullptr);
    EXPECT_EQ(0u, floor.get_primitives()[0].x); EXPECT_EQ(1u, floor.get_primitives()[0].y); EXPECT_EQ(2u, floor.get_primitives()[0].z);
    EXPECT_EQ(0u, floor.get_primitives()[1].x); EXPECT_EQ(2u, floor.get_primitives()[1].y); EXPECT_EQ(3u, floor.get_primitives()[1].z);
    EXPECT_FLOAT_EQ_EPS(1.0f, floor.get_positions()[2].x, 0.0f);
    EXPECT_FLOAT_EQ_EPS(-1.0f, floor.get_positions()[2].z, 0.0f);   // third corner of the face is OBJ vertex 2
    EXPECT_FLOAT_EQ_EPS(1.0f, floor.get_texcoords()[2].x, 0.0f);
    EXPECT_FLOAT_EQ_EPS(1.0f, floor.get_normals()[3].y, 0.0f);
    EXPECT_FLOAT_EQ_EPS(-1.0f, floor.get_bounds().minimum.x, 0.0f);
    EXPECT_FLOAT_EQ_EPS(1.0f, floor.get_bounds().maximum.z, 0.0f);

    // Sail: negative indices address the last three vertices; no normals / texcoords on its first vertex -> position only
    Mesh sail = models[1].get_mesh();
    EXPECT_EQ(1u, sail.get_primitive_count());
    EXPECT_EQ(3u, sail.get_vertex_count());
    EXPECT_TRUE(sail.get_normals() == nullptr && sail.get_texcoords() == nullptr);
    EXPECT_FLOAT_EQ_EPS(2.0f, sail.get_positions()[1].y, 0.0f);

    // Materials: roughness = (2 / (Ns + 2))^(1/4), specularity = mean(Ks), coverage = d, metallic from illum 3 / 5
    Material rough = models[0].get_material();
    EXPECT_EQ(std::string("Rough"), rough.get_name());
    EXPECT_FLOAT_EQ_EPS(0.5f, rough.get_tint().r, 0.0f);
    EXPECT_FLOAT_EQ_EPS(0.125f, rough.get_tint().b, 0.0f);
    EXPECT_FLOAT_EQ_EPS(std::pow(2.0f / 16.0f, 0.25f), rough.get_roughness(), 1e-6f);
    EXPECT_FLOAT_EQ_EPS(0.06f, rough.get_specularity(), 1e-6f);
    EXPECT_FLOAT_EQ_EPS(0.75f, rough.get_coverage(), 0.0f);
    EXPECT_FLOAT_EQ_EPS(0.0f, rough.get_metallic(), 0.0f);
    Material mirror = models[1].get_material();
    EXPECT_FLOAT_EQ_EPS(1.0f, mirror.get_metallic(), 0.0f);
    EXPECT_FLOAT_EQ_EPS(std::pow(2.0f / 1000.0f, 0.25f), mirror.get_roughness(), 1e-6f);
    EXPECT_FLOAT_EQ_EPS(1.0f, mirror.get_specularity(), 1e-6f);

    // map_Kd with an alpha channel: the alpha becomes an Alpha8 coverage texture, the tint image's alpha is set to 255
    EXPECT_TRUE(rough.has_tint_texture());
    EXPECT_TRUE(rough.get_coverage_texture_ID() != TextureID::invalid_UID());
    Image tint = Textures::get_image_ID(rough.get_tint_roughness_texture_ID());
    Image coverage = Textures::get_image_ID(rough.get_coverage_texture_ID());
    EXPECT_TRUE(coverage.get_pixel_format() == PixelFormat::Alpha8);
    EXPECT_EQ(0, int(coverage.get_pixels<unsigned char>()[0]));
    EXPECT_EQ(255, int(coverage.get_pixels<unsigned char>()[3]));
    EXPECT_EQ(255, int(tint.get_pixels<unsigned char>()[3]));
    EXPECT_TRUE(mirror.get_coverage_texture_ID() == TextureID::invalid_UID());

    // SimpleViewer: the hard-edged mask flags the material as a cut-out
    EXPECT_FALSE(rough.is_cutout());
    SceneLoading::detect_and_flag_cutout_materials();
    EXPECT_TRUE(Material(rough.get_ID()).is_cutout());
    EXPECT_FALSE(Material(mirror.get_ID()).is_cutout());
}

CPU_TEST_F(LoaderFixture, single_shape_is_its_own_root_and_missing_files_fail) {
    const std::string path = write("one.obj", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n");
    SceneNode root = ObjLoader::load(path, nullptr);
    EXPECT_TRUE(root != SceneNode::invalid());
    EXPECT_EQ(size_t(0), root.get_children().size());
    unsigned models = 0;
    for (MeshModelID id : MeshModels::get_iterable()) { ++models; EXPECT_TRUE(MeshModel(id).get_scene_node() == root); EXPECT_TRUE(MeshModel(id).get_material().get_ID() == MaterialID::invalid_UID()); }
    EXPECT_EQ(1u, models);
    EXPECT_TRUE(ObjLoader::load((directory / "missing.obj").string(), nullptr) == SceneNode::invalid());
}

CPU_TEST_F(LoaderFixture, viewer_defaults_place_camera_light_and_clip_planes) {
    write("scene.mtl", scene_mtl);
    const std::string path = write("scene.obj", two_shapes_obj);
    SceneRoot scene = SceneRoot("Loaded", RGB(0.68f, 0.92f, 1.0f));
    Matrix4x4f projection, inverse_projection;
    CameraUtils::compute_perspective_projection(0.1f, 100.0f, PI<float>() / 4.0f, 16.0f / 9.0f, projection, inverse_projection);
    CameraID camera_ID = Cameras::create("Camera", scene.get_ID(), projection, inverse_projection);
    SceneNode loaded = ObjLoader::load(path, load_test_image);
    loaded.set_parent(scene.get_root_node());

    SceneLoading::ViewerDefaults defaults = SceneLoading::apply_viewer_defaults(scene.get_root_node(), camera_ID, true);
    // bounding spheres: floor centre (0,0,0) radius |(2,0,2)|/2 = sqrt(2); sail centre (0.5,1,0) radius |(1,2,0)|/2 = sqrt(5)/2
    const float r0 = std::sqrt(2.0f), r1 = std::sqrt(5.0f) * 0.5f;
    EXPECT_FLOAT_EQ_EPS(std::fmin(-r0, 0.5f - r1), defaults.scene_bounds.minimum.x, 1e-5f);
    EXPECT_FLOAT_EQ_EPS(1.0f + r1, defaults.scene_bounds.maximum.y, 1e-5f);
    EXPECT_TRUE(defaults.added_light);
    EXPECT_FLOAT_EQ_EPS(defaults.scene_size / 10000.0f, defaults.near_plane, 0.0f);
    EXPECT_FLOAT_EQ_EPS(defaults.scene_size * 3.0f, defaults.far_plane, 0.0f);
    const Vector3f expected_position = defaults.scene_bounds.center() + defaults.scene_bounds.size();
    EXPECT_FLOAT_EQ_EPS(expected_position.x, Cameras::get_transform(camera_ID).translation.x, 1e-6f);
    const Vector3f forward = Cameras::get_transform(camera_ID).rotation.forward();
    const Vector3f to_center = normalize(defaults.scene_bounds.center() - expected_position);
    EXPECT_FLOAT_EQ_EPS(1.0f, dot(forward, to_center), 1e-5f);
    unsigned lights = 0;
    for (LightSourceID id : LightSources::get_iterable()) { ++lights; EXPECT_TRUE(LightSources::get_type(id) == LightSources::Type::Directional); EXPECT_FLOAT_EQ_EPS(15.0f, LightSources::get_power(id).r, 0.0f); }
    EXPECT_EQ(1u, lights);

    // The loaded scene flattens into a renderable description: 3 triangles, 2 materials (+ the invalid slot 0), 2 textures (+ slot 0), 1 light
    HIPRenderer::SceneBuilder flattened;
    HIPRenderer::flatten_bifrost_scene(flattened);
    EXPECT_EQ(3u, flattened.desc().triangle_count);
    EXPECT_EQ(3u, flattened.desc().material_count);
    EXPECT_EQ(3u, flattened.desc().texture_count);
    EXPECT_EQ(1u, flattened.desc().light_count);
    EXPECT_TRUE(flattened.desc().wide_node_count >= 1u);
}

GPU_TEST_F(LoaderFixture, loaded_obj_renders_through_the_renderer) {
    // file -> ObjLoader -> SimpleViewer defaults -> HIPRenderer::Renderer: the floor is lit by the default directional light.
    std::error_code error;
    const std::filesystem::path data = std::filesystem::read_symlink("/proc/self/exe", error).parent_path() / ".." / ".." / "bifrost3d_amd" / "data";
    HIPRenderer::Renderer* renderer = HIPRenderer::Renderer::initialize(0, data);
    EXPECT_TRUE(renderer != nullptr);
    if (!renderer) return;
    write("scene.mtl", scene_mtl);
    const std::string path = write("scene.obj", two_shapes_obj);
    SceneRoot scene = SceneRoot("Loaded", RGB(0.1f, 0.1f, 0.1f));
    SceneNode loaded = ObjLoader::load(path, load_test_image);
    loaded.set_parent(scene.get_root_node());
    SceneLoading::detect_and_flag_cutout_materials();

    const Vector2i frame_size(64, 36);
    Matrix4x4f projection, inverse_projection;
    CameraID camera_ID = Cameras::create("Camera", scene.get_ID(), Matrix4x4f::identity(), Matrix4x4f::identity());
    SceneLoading::ViewerDefaults defaults = SceneLoading::apply_viewer_defaults(scene.get_root_node(), camera_ID, true);
    CameraUtils::compute_perspective_projection(defaults.near_plane, defaults.far_plane, PI<float>() / 4.0f, float(frame_size.x) / frame_size.y, projection, inverse_projection);
    Cameras::set_projection_matrices(camera_ID, projection, inverse_projection);
    Cameras::set_renderer_ID(camera_ID, renderer->get_renderer_ID());

    renderer->handle_updates();
    unsigned int iterations = 0;
    for (int i = 0; i < 4; ++i) iterations = renderer->render(camera_ID, nullptr, 0, frame_size);
    EXPECT_EQ(4u, iterations);
    std::vector<double> accumulation;
    EXPECT_TRUE(renderer->read_accumulation(accumulation));
    double brightest = 0.0;
    bool finite = true;
    for (size_t i = 0; i < accumulation.size(); i += 4)
        for (int c = 0; c < 3; ++c) { finite = finite && std::isfinite(accumulation[i + c]); brightest = std::fmax(brightest, accumulation[i + c]); }
    EXPECT_TRUE(finite);
    if (!(brightest > 0.5)) { double sum = 0; for (size_t i = 0; i < accumulation.size(); i += 4) sum += accumulation[i]; fprintf(stderr, "brightest %g mean red %g, first pixel %g %g %g\n", brightest, sum / (accumulation.size() / 4), accumulation[0], accumulation[1], accumulation[2]); }
    EXPECT_TRUE(brightest > 0.5);   // environment tint is 0.1: anything brighter is the lit floor / sail
    delete renderer;
}

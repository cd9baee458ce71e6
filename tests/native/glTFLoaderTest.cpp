// glTFLoaderTest.cpp -- the glTF ingestion path: JSON reader, PNG codec, container handling and the reference's
// glTF -> Bifrost mapping rules (extensions/glTFLoader/glTFLoader/glTFLoader.cpp), on files written by the tests themselves.
// The reference has no tests of its loader; the expectations below are its mapping rules worked out by hand.
#include "MiniTest.h"

#include "../../bifrost3d_amd/host/HIPRenderer/Renderer.h"
#include "../../bifrost3d_amd/host/ImageIO/PngImage.h"
#include "../../bifrost3d_amd/host/Json.h"
#include "../../bifrost3d_amd/host/SceneBuilder.h"
#include "../../bifrost3d_amd/host/SceneLoading.h"
#include "../../bifrost3d_amd/host/glTFLoader/glTFLoader.h"

#include <cstdio>
#include <filesystem>
#include <fstream>

#include <unistd.h>

using namespace Bifrost;
using namespace Bifrost::Assets;
using namespace Bifrost::Math;
using namespace Bifrost::Scene;

namespace {

std::string base64(const void* data, size_t size) {
    static const char* alphabet = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    const unsigned char* bytes = static_cast<const unsigned char*>(data);
    std::string out;
    for (size_t i = 0; i < size; i += 3) {
        const unsigned b0 = bytes[i], b1 = i + 1 < size ? bytes[i + 1] : 0, b2 = i + 2 < size ? bytes[i + 2] : 0;
        out += alphabet[b0 >> 2];
        out += alphabet[((b0 & 3) << 4) | (b1 >> 4)];
        out += i + 1 < size ? alphabet[((b1 & 15) << 2) | (b2 >> 6)] : '=';
        out += i + 2 < size ? alphabet[b2 & 63] : '=';
    }
    return out;
}

// A byte buffer that hands out (offset, length) as things are appended, 4 byte aligned like glTF writers do.
struct BufferWriter {
    std::vector<unsigned char> bytes;
    size_t append(const void* data, size_t size) {
        while (bytes.size() % 4) bytes.push_back(0);
        const size_t offset = bytes.size();
        bytes.insert(bytes.end(), static_cast<const unsigned char*>(data), static_cast<const unsigned char*>(data) + size);
        return offset;
    }
};

struct glTFFixture {
    std::filesystem::path directory;
    void SetUp() {
        deallocate_all();
        directory = std::filesystem::temp_directory_path() / ("hipr_gltf_test_" + std::to_string(::getpid()));
        std::filesystem::create_directories(directory);
    }
    void TearDown() {
        std::error_code error;
        std::filesystem::remove_all(directory, error);
        deallocate_all();
    }
    bool usable() const { return true; }
    std::string write(const char* name, const std::string& text) {
        const std::filesystem::path path = directory / name;
        std::ofstream(path, std::ios::binary) << text;
        return path.string();
    }
};

// One triangle in the z = 0 plane facing +z (counter-clockwise in glTF's right-handed frame), 16 bit indices, interleaved
// position + normal vertices (byteStride 24) and a separate texcoord view.
struct TriangleBuffers {
    BufferWriter buffer;
    size_t vertices_offset, texcoords_offset, indices_offset;
    TriangleBuffers() {
        const float vertices[18] = {0, 0, 0, 0, 0, 1,   1, 0, 0, 0, 0, 1,   0, 1, 0, 0, 0, 1};
        const float texcoords[6] = {0, 0, 1, 0, 0, 1};
        const unsigned short indices[3] = {0, 1, 2};
        vertices_offset = buffer.append(vertices, sizeof(vertices));
        texcoords_offset = buffer.append(texcoords, sizeof(texcoords));
        indices_offset = buffer.append(indices, sizeof(indices));
    }
    // bufferViews 0-2 and accessors 0 POSITION, 1 NORMAL, 2 TEXCOORD_0, 3 indices
    std::string views_and_accessors() const {
        char text[2048];
        snprintf(text, sizeof(text),
            "\"bufferViews\":[{\"buffer\":0,\"byteOffset\":%zu,\"byteLength\":72,\"byteStride\":24},{\"buffer\":0,\"byteOffset\":%zu,\"byteLength\":24},{\"buffer\":0,\"byteOffset\":%zu,\"byteLength\":6}],"
            "\"accessors\":[{\"bufferView\":0,\"componentType\":5126,\"count\":3,\"type\":\"VEC3\",\"min\":[0,0,0],\"max\":[1,1,0]},"
            "{\"bufferView\":0,\"byteOffset\":12,\"componentType\":5126,\"count\":3,\"type\":\"VEC3\"},"
            "{\"bufferView\":1,\"componentType\":5126,\"count\":3,\"type\":\"VEC2\"},"
            "{\"bufferView\":2,\"componentType\":5123,\"count\":3,\"type\":\"SCALAR\"}]",
            vertices_offset, texcoords_offset, indices_offset);
        return text;
    }
    std::string data_uri_buffer() const {
        return "\"buffers\":[{\"byteLength\":" + std::to_string(buffer.bytes.size()) + ",\"uri\":\"data:application/octet-stream;base64," + base64(buffer.bytes.data(), buffer.bytes.size()) + "\"}]";
    }
};

const char* triangle_mesh = "\"meshes\":[{\"name\":\"Tri\",\"primitives\":[{\"attributes\":{\"POSITION\":0,\"NORMAL\":1,\"TEXCOORD_0\":2},\"indices\":3,\"material\":0}]}]";

std::vector<MeshModel> all_models() {
    std::vector<MeshModel> models;
    for (MeshModelID id : MeshModels::get_iterable()) models.push_back(id);
    return models;
}

} // namespace

CPU_TEST_F(glTFFixture, json_reader) {
    const std::string text = "  {\"a\":[1,-2.5e2,true,null,\"x\\n\\u00e9\\ud83d\\ude00\\\"\"],\"b\":{\"c\":{}},\"d\":0.125}\n";
    Json::Value doc;
    std::string error;
    EXPECT_TRUE(Json::Value::parse(text.data(), text.data() + text.size(), doc, error));
    EXPECT_TRUE(doc.is_object());
    EXPECT_EQ(size_t(5), doc["a"].size());
    EXPECT_EQ(1, doc["a"][size_t(0)].as_int());
    EXPECT_EQ(-250.0, doc["a"][size_t(1)].as_double());
    EXPECT_TRUE(doc["a"][size_t(2)].as_bool());
    EXPECT_TRUE(doc["a"][size_t(3)].is_null());
    EXPECT_EQ(std::string("x\n\xC3\xA9\xF0\x9F\x98\x80\""), doc["a"][size_t(4)].as_string());
    EXPECT_TRUE(doc["b"]["c"].is_object());
    EXPECT_EQ(0.125, doc["d"].as_double());
    // missing members chain to null and report the fallback
    EXPECT_EQ(-1, doc["nope"]["deeper"][size_t(3)].as_int(-1));
    EXPECT_FALSE(doc.has("nope"));

    for (const char* bad : {"{\"a\":}", "[1,2", "{\"a\":1} x", "{'a':1}", "[01]", "\"\\q\"", ""}) {
        const std::string s = bad;
        Json::Value v;
        EXPECT_FALSE(Json::Value::parse(s.data(), s.data() + s.size(), v, error));
        EXPECT_FALSE(error.empty());
    }
}

CPU_TEST_F(glTFFixture, png_round_trip) {
    const unsigned width = 5, height = 3;
    for (unsigned channels = 1; channels <= 4; ++channels) {
        std::vector<uint8_t> pixels(width * height * channels);
        for (size_t i = 0; i < pixels.size(); ++i) pixels[i] = uint8_t(i * 37 + channels * 11);
        const std::vector<uint8_t> file = PngImage::encode(width, height, channels, pixels.data(), false);
        EXPECT_TRUE(file.size() > 8);

        Image image = PngImage::load_from_memory("memory", file.data(), file.size());   // rows as stored
        EXPECT_TRUE(image.exists());
        if (!image.exists()) continue;
        EXPECT_EQ(width, image.get_width()); EXPECT_EQ(height, image.get_height());
        EXPECT_TRUE(Images::is_sRGB(image.get_ID()));
        const PixelFormat expected_format = channels == 1 ? PixelFormat::Intensity8 : (channels == 3 ? PixelFormat::RGB24 : PixelFormat::RGBA32);
        EXPECT_TRUE(image.get_pixel_format() == expected_format);
        const uint8_t* decoded = image.get_pixels<uint8_t>();
        bool same = true;
        for (unsigned p = 0; p < width * height; ++p) {
            if (channels == 2)   // intensity + alpha is widened to RGBA
                same = same && decoded[4 * p] == pixels[2 * p] && decoded[4 * p + 1] == pixels[2 * p] && decoded[4 * p + 2] == pixels[2 * p] && decoded[4 * p + 3] == pixels[2 * p + 1];
            else
                for (unsigned c = 0; c < channels; ++c) same = same && decoded[channels * p + c] == pixels[channels * p + c];
        }
        EXPECT_TRUE(same);

        // load() from a file hands the bottom row first
        const std::string path = write("image.png", std::string(file.begin(), file.end()));
        Image flipped = PngImage::load(path);
        EXPECT_TRUE(flipped.exists());
        if (flipped.exists() && channels != 2) {
            const uint8_t* f = flipped.get_pixels<uint8_t>();
            bool rows_flipped = true;
            for (unsigned y = 0; y < height; ++y)
                rows_flipped = rows_flipped && std::memcmp(f + size_t(y) * width * channels, pixels.data() + size_t(height - 1 - y) * width * channels, width * channels) == 0;
            EXPECT_TRUE(rows_flipped);
        }

        // write() flips back: the file of a bottom-up image decodes to the original rows
        if (flipped.exists() && channels != 2) {
            const std::string out_path = (directory / "written.png").string();
            EXPECT_TRUE(PngImage::write(out_path, flipped));
            Image reread = PngImage::load(out_path);
            EXPECT_TRUE(reread.exists());
            if (reread.exists()) EXPECT_TRUE(std::memcmp(reread.get_pixels<uint8_t>(), flipped.get_pixels<uint8_t>(), size_t(width) * height * channels) == 0);
        }
    }
    const uint8_t garbage[16] = {1, 2, 3};
    EXPECT_FALSE(PngImage::load_from_memory("garbage", garbage, sizeof(garbage)).exists());
}

CPU_TEST_F(glTFFixture, triangle_is_mirrored_into_the_left_handed_frame) {
    TriangleBuffers tri;
    const std::string gltf = "{\"asset\":{\"version\":\"2.0\"}," + tri.data_uri_buffer() + "," + tri.views_and_accessors() + "," + triangle_mesh + ","
        "\"materials\":[{\"name\":\"Plain\"}],"
        "\"nodes\":[{\"name\":\"Node\",\"mesh\":0,\"translation\":[1,2,3]}],\"scenes\":[{\"nodes\":[0]}],\"scene\":0}";
    const std::string path = write("triangle.gltf", gltf);
    EXPECT_TRUE(glTFLoader::file_supported(path));
    EXPECT_TRUE(glTFLoader::file_supported("model.glb"));
    EXPECT_FALSE(glTFLoader::file_supported("model.obj"));

    SceneNode root = glTFLoader::load(path);
    EXPECT_TRUE(root != SceneNode::invalid());
    if (root == SceneNode::invalid()) return;
    EXPECT_EQ(std::string("Node"), root.get_name());
    const Transform transform = root.get_global_transform();
    EXPECT_FLOAT_EQ_EPS(-1.0f, transform.translation.x, 1e-6f);   // X negated
    EXPECT_FLOAT_EQ_EPS(2.0f, transform.translation.y, 1e-6f);
    EXPECT_FLOAT_EQ_EPS(3.0f, transform.translation.z, 1e-6f);
    EXPECT_FLOAT_EQ_EPS(1.0f, transform.scale, 1e-6f);
    EXPECT_FLOAT_EQ_EPS(1.0f, std::fabs(transform.rotation.w), 1e-6f);

    const std::vector<MeshModel> models = all_models();
    EXPECT_EQ(size_t(1), models.size());
    if (models.size() != 1) return;
    Mesh mesh = models[0].get_mesh();
    EXPECT_EQ(std::string("Tri"), mesh.get_name());
    EXPECT_EQ(1u, mesh.get_primitive_count());
    EXPECT_EQ(3u, mesh.get_vertex_count());
    EXPECT_TRUE(mesh.get_normals() != nullptr && mesh.get_texcoords() != nullptr && mesh.get_tint_and_roughness() == nullptr);
    // X of positions and normals negated, first two corners swapped so that the triangle still faces +z
    EXPECT_EQ(-1.0f, mesh.get_positions()[1].x); EXPECT_EQ(0.0f, mesh.get_positions()[1].y);
    EXPECT_EQ(1.0f, mesh.get_positions()[2].y);
    EXPECT_EQ(1.0f, mesh.get_normals()[0].z); EXPECT_EQ(0.0f, std::fabs(mesh.get_normals()[0].x));
    EXPECT_EQ(1u, mesh.get_primitives()[0].x); EXPECT_EQ(0u, mesh.get_primitives()[0].y); EXPECT_EQ(2u, mesh.get_primitives()[0].z);
    const Vector3f a = mesh.get_positions()[mesh.get_primitives()[0].x], b = mesh.get_positions()[mesh.get_primitives()[0].y], c = mesh.get_positions()[mesh.get_primitives()[0].z];
    const Vector3f face_normal = cross(b - a, c - a);
    EXPECT_TRUE(face_normal.z > 0.0f);      // mirroring alone would turn the face away from its vertex normal (0, 0, 1); the swap keeps them agreeing
    EXPECT_EQ(1.0f, mesh.get_texcoords()[1].x); EXPECT_EQ(1.0f, mesh.get_texcoords()[2].y);
    // bounds from the accessor's min / max, mirrored
    EXPECT_EQ(-1.0f, mesh.get_bounds().minimum.x); EXPECT_EQ(0.0f, mesh.get_bounds().maximum.x);
    EXPECT_EQ(1.0f, mesh.get_bounds().maximum.y);

    // a material without pbrMetallicRoughness: the loader's defaults
    Material material = models[0].get_material();
    EXPECT_EQ(std::string("Plain"), material.get_name());
    EXPECT_EQ(1.0f, material.get_tint().r); EXPECT_EQ(1.0f, material.get_roughness());
    EXPECT_EQ(0.0f, material.get_metallic()); EXPECT_FLOAT_EQ_EPS(0.04f, material.get_specularity(), 1e-7f);
    EXPECT_EQ(1.0f, material.get_coverage());
    EXPECT_TRUE(material.get_flags().is_empty());
}

CPU_TEST_F(glTFFixture, hierarchy_transforms_and_residual_scaling) {
    TriangleBuffers tri;
    // Parent: uniform scale 2 and a quarter turn about Y. Child A: translated along X in the parent's frame. Child B: a matrix
    // with non-uniform scale (1, 3, 1), which a Bifrost transform cannot hold: the remainder goes into a copy of the mesh.
    const std::string gltf = "{\"asset\":{\"version\":\"2.0\"}," + tri.data_uri_buffer() + "," + tri.views_and_accessors() + "," + triangle_mesh + ","
        "\"materials\":[{}],"
        "\"nodes\":[{\"name\":\"Parent\",\"children\":[1,2],\"scale\":[2,2,2],\"rotation\":[0,0.7071067811865476,0,0.7071067811865476]},"
        "{\"name\":\"ChildA\",\"mesh\":0,\"translation\":[1,0,0]},"
        "{\"mesh\":0,\"matrix\":[1,0,0,0, 0,3,0,0, 0,0,1,0, 0,5,0,1]}],"
        "\"scenes\":[{\"nodes\":[0]}],\"scene\":0}";
    SceneNode root = glTFLoader::load(write("hierarchy.gltf", gltf));
    EXPECT_TRUE(root != SceneNode::invalid());
    if (root == SceneNode::invalid()) return;
    EXPECT_EQ(std::string("Parent"), root.get_name());
    EXPECT_FLOAT_EQ_EPS(2.0f, root.get_global_transform().scale, 1e-5f);
    EXPECT_EQ(size_t(2), root.get_children().size());
    EXPECT_EQ(std::string("unnamed_material_0"), all_models()[0].get_material().get_name());

    SceneNode child_a, child_b;
    for (SceneNodeID id : root.get_children()) { SceneNode n = id; if (n.get_name() == "ChildA") child_a = n; else child_b = n; }
    EXPECT_EQ(std::string("unnamed_node_2"), child_b.get_name());

    // glTF: a rotation of +90 degrees about Y takes +X to -Z; child A sits at 2 * (0, 0, -1) in glTF, mirrored X stays 0.
    const Transform ta = child_a.get_global_transform();
    EXPECT_FLOAT_EQ_EPS(0.0f, ta.translation.x, 1e-5f); EXPECT_FLOAT_EQ_EPS(0.0f, ta.translation.y, 1e-5f); EXPECT_FLOAT_EQ_EPS(-2.0f, ta.translation.z, 1e-5f);
    EXPECT_FLOAT_EQ_EPS(2.0f, ta.scale, 1e-5f);
    // The mirrored rotation: a glTF point (1, 0, 0) local to the child is Bifrost (-1, 0, 0) in the mesh and ends at
    // glTF 2 * R(1, 0, 0) + t = (0, 0, -2) + (0, 0, -2), whose mirror is the same point.
    const Vector3f p = ta * Vector3f(-1.0f, 0.0f, 0.0f);
    EXPECT_FLOAT_EQ_EPS(0.0f, p.x, 1e-5f); EXPECT_FLOAT_EQ_EPS(-4.0f, p.z, 1e-5f);

    // Child B: global = parent * matrix, determinant 8 * 3 = 24, the volume preserving scale is its cube root.
    const Transform tb = child_b.get_global_transform();
    EXPECT_FLOAT_EQ_EPS(std::cbrt(24.0), tb.scale, 1e-5f);
    EXPECT_FLOAT_EQ_EPS(10.0f, tb.translation.y, 1e-5f);

    std::vector<MeshModel> models = all_models();
    EXPECT_EQ(size_t(2), models.size());
    if (models.size() != 2) return;
    Mesh shared, baked;
    for (MeshModel m : models) (m.get_scene_node() == child_a ? shared : baked) = m.get_mesh();
    EXPECT_TRUE(shared.get_ID() != baked.get_ID());
    EXPECT_EQ(std::string("Tri"), shared.get_name());
    EXPECT_EQ(1.0f, shared.get_positions()[2].y);       // the shared mesh is untouched
    // The baked copy under its node's transform reproduces global * vertex: vertex 2 is glTF (0, 1, 0) -> matrix (0, 3 + 5, 0) -> parent 2 * R -> (0, 16, 0).
    const Vector3f q = tb * baked.get_positions()[2];
    EXPECT_FLOAT_EQ_EPS(0.0f, q.x, 1e-4f); EXPECT_FLOAT_EQ_EPS(16.0f, q.y, 1e-4f); EXPECT_FLOAT_EQ_EPS(0.0f, q.z, 1e-4f);
    // vertex 1 is glTF (1, 0, 0) -> (1, 5, 0) -> 2 * R(1, 5, 0) = (0, 10, -2)
    const Vector3f r = tb * baked.get_positions()[1];
    EXPECT_FLOAT_EQ_EPS(0.0f, r.x, 1e-4f); EXPECT_FLOAT_EQ_EPS(10.0f, r.y, 1e-4f); EXPECT_FLOAT_EQ_EPS(-2.0f, r.z, 1e-4f);
    // bounds follow the baked vertices and the normals stay unit length
    EXPECT_FLOAT_EQ_EPS(baked.get_positions()[2].y, baked.get_bounds().maximum.y, 1e-6f);
    EXPECT_FLOAT_EQ_EPS(1.0f, magnitude(baked.get_normals()[0]), 1e-5f);
}

CPU_TEST_F(glTFFixture, materials_and_texture_channel_regrouping) {
    TriangleBuffers tri;
    // base colour: 2x2 RGBA, alpha varies (a coverage map); metallic-roughness: 2x2 RGB, G = roughness (sRGB encoded by the loader's convention), B = metallic
    const uint8_t base_colour[16] = {200, 100, 50, 0,   200, 100, 50, 255,   10, 20, 30, 128,   40, 50, 60, 255};
    const uint8_t metallic_roughness[12] = {0, 255, 0,   0, 128, 255,   0, 0, 64,   0, 64, 255};
    const uint8_t opaque[12] = {255, 0, 0,   0, 255, 0,   0, 0, 255,   9, 9, 9};
    const std::vector<uint8_t> base_png = PngImage::encode(2, 2, 4, base_colour, false), mr_png = PngImage::encode(2, 2, 3, metallic_roughness, false), opaque_png = PngImage::encode(2, 2, 3, opaque, false);
    const size_t mr_offset = tri.buffer.append(mr_png.data(), mr_png.size());   // the second image is embedded through a buffer view
    write("opaque tint.png", std::string(opaque_png.begin(), opaque_png.end()));            // the third is a file next to the glTF, its uri percent-encoded

    char views[256];
    snprintf(views, sizeof(views), ",{\"buffer\":0,\"byteOffset\":%zu,\"byteLength\":%zu}]", mr_offset, mr_png.size());
    std::string views_and_accessors = tri.views_and_accessors();
    views_and_accessors.replace(views_and_accessors.find("],\"accessors\""), 1, views);

    const std::string gltf = "{\"asset\":{\"version\":\"2.0\"}," + tri.data_uri_buffer() + "," + views_and_accessors + ","
        "\"meshes\":[{\"primitives\":[{\"attributes\":{\"POSITION\":0},\"indices\":3,\"material\":0},{\"attributes\":{\"POSITION\":0,\"TEXCOORD_0\":2},\"material\":1},"
        "{\"attributes\":{\"POSITION\":0},\"mode\":1,\"material\":0},{\"attributes\":{\"POSITION\":0},\"material\":2},{\"attributes\":{\"POSITION\":0}}]}],"
        "\"images\":[{\"uri\":\"data:image/png;base64," + base64(base_png.data(), base_png.size()) + "\"},{\"bufferView\":3,\"mimeType\":\"image/png\"},{\"uri\":\"opaque%20tint.png\"},{\"uri\":\"missing.png\"}],"
        "\"samplers\":[{\"magFilter\":9728,\"minFilter\":9729,\"wrapS\":33071,\"wrapT\":10497}],"
        "\"textures\":[{\"source\":0,\"sampler\":0},{\"source\":1},{\"source\":2},{\"source\":3}],"
        "\"materials\":["
        "{\"name\":\"Textured\",\"doubleSided\":true,\"alphaMode\":\"MASK\",\"alphaCutoff\":0.25,"
        "\"pbrMetallicRoughness\":{\"baseColorFactor\":[0.5,0.25,0.125,1],\"metallicFactor\":0.75,\"roughnessFactor\":0.5,\"baseColorTexture\":{\"index\":0},\"metallicRoughnessTexture\":{\"index\":1}},"
        "\"extensions\":{\"KHR_materials_clearcoat\":{\"clearcoatFactor\":0.8,\"clearcoatRoughnessFactor\":0.1}}},"
        "{\"name\":\"Opaque\",\"alphaMode\":\"MASK\",\"pbrMetallicRoughness\":{\"baseColorTexture\":{\"index\":2}}},"
        "{\"name\":\"Broken\",\"pbrMetallicRoughness\":{\"baseColorTexture\":{\"index\":3},\"roughnessFactor\":0.3}}],"
        "\"nodes\":[{\"mesh\":0}],\"scenes\":[{\"nodes\":[0]}],\"scene\":0}";
    SceneNode root = glTFLoader::load(write("materials.gltf", gltf));
    EXPECT_TRUE(root != SceneNode::invalid());
    if (root == SceneNode::invalid()) return;

    // Four of the five primitives are triangles (mode 1 is skipped): their meshes carry the primitive's index in the name.
    const std::vector<MeshModel> models = all_models();
    EXPECT_EQ(size_t(4), models.size());
    if (models.size() != 4) return;
    EXPECT_EQ(std::string("unnamed_mesh_0_primitive_0"), models[0].get_mesh().get_name());
    EXPECT_EQ(std::string("unnamed_mesh_0_primitive_1"), models[1].get_mesh().get_name());
    EXPECT_EQ(std::string("unnamed_mesh_0_primitive_3"), models[2].get_mesh().get_name());
    // no index accessor: consecutive vertices, corners swapped for the mirroring
    EXPECT_EQ(1u, models[1].get_mesh().get_primitives()[0].x); EXPECT_EQ(0u, models[1].get_mesh().get_primitives()[0].y); EXPECT_EQ(2u, models[1].get_mesh().get_primitives()[0].z);
    EXPECT_EQ(std::string("unnamed_default_material"), models[3].get_material().get_name());

    const Materials::Data textured = Materials::get_data(models[0].get_material().get_ID());
    EXPECT_TRUE(textured.flags.is_set(MaterialFlag::ThinWalled) && textured.flags.is_set(MaterialFlag::Cutout));
    EXPECT_EQ(0.25f, textured.coverage);
    EXPECT_EQ(0.5f, textured.tint.r); EXPECT_EQ(0.25f, textured.tint.g); EXPECT_EQ(0.125f, textured.tint.b);
    EXPECT_EQ(0.75f, textured.metallic); EXPECT_EQ(0.5f, textured.roughness);
    EXPECT_FLOAT_EQ_EPS(0.8f, textured.coat, 1e-7f); EXPECT_FLOAT_EQ_EPS(0.1f, textured.coat_roughness, 1e-7f);
    EXPECT_FLOAT_EQ_EPS(0.04f, textured.specularity, 1e-7f);

    // coverage: the base colour's alpha as an Alpha8 image, with the base colour texture's sampler
    EXPECT_TRUE(textured.coverage_texture_ID != TextureID::invalid_UID());
    if (textured.coverage_texture_ID != TextureID::invalid_UID()) {
        Image coverage = Textures::get_image_ID(textured.coverage_texture_ID);
        EXPECT_TRUE(coverage.get_pixel_format() == PixelFormat::Alpha8);
        EXPECT_EQ(std::string("Textured_coverage"), coverage.get_name());
        const uint8_t* a = coverage.get_pixels<uint8_t>();
        EXPECT_EQ(0, int(a[0])); EXPECT_EQ(255, int(a[1])); EXPECT_EQ(128, int(a[2])); EXPECT_EQ(255, int(a[3]));
        EXPECT_TRUE(Textures::get_magnification_filter(textured.coverage_texture_ID) == MagnificationFilter::None);
        EXPECT_TRUE(Textures::get_minification_filter(textured.coverage_texture_ID) == MinificationFilter::Linear);
        EXPECT_TRUE(Textures::get_wrapmode_U(textured.coverage_texture_ID) == WrapMode::Clamp);
        EXPECT_TRUE(Textures::get_wrapmode_V(textured.coverage_texture_ID) == WrapMode::Repeat);
    }
    // metallic: the blue channel of the metallic-roughness image, default sampler (linear, trilinear, repeat)
    EXPECT_TRUE(textured.metallic_texture_ID != TextureID::invalid_UID());
    if (textured.metallic_texture_ID != TextureID::invalid_UID()) {
        Image metallic = Textures::get_image_ID(textured.metallic_texture_ID);
        EXPECT_TRUE(metallic.get_pixel_format() == PixelFormat::Alpha8);
        const uint8_t* m = metallic.get_pixels<uint8_t>();
        EXPECT_EQ(0, int(m[0])); EXPECT_EQ(255, int(m[1])); EXPECT_EQ(64, int(m[2])); EXPECT_EQ(255, int(m[3]));
        EXPECT_TRUE(Textures::get_magnification_filter(textured.metallic_texture_ID) == MagnificationFilter::Linear);
        EXPECT_TRUE(Textures::get_minification_filter(textured.metallic_texture_ID) == MinificationFilter::Trilinear);
        EXPECT_TRUE(Textures::get_wrapmode_U(textured.metallic_texture_ID) == WrapMode::Repeat);
    }
    // tint + roughness: RGB of the base colour, alpha = the green channel of the other image decoded from sRGB to linear
    EXPECT_TRUE(textured.tint_roughness_texture_ID != TextureID::invalid_UID());
    if (textured.tint_roughness_texture_ID != TextureID::invalid_UID()) {
        Image tint_roughness = Textures::get_image_ID(textured.tint_roughness_texture_ID);
        EXPECT_TRUE(tint_roughness.get_pixel_format() == PixelFormat::RGBA32);
        EXPECT_TRUE(Images::is_sRGB(tint_roughness.get_ID()));
        const uint8_t* t = tint_roughness.get_pixels<uint8_t>();
        EXPECT_EQ(200, int(t[0])); EXPECT_EQ(100, int(t[1])); EXPECT_EQ(50, int(t[2])); EXPECT_EQ(255, int(t[3]));      // sRGB 255 -> 1.0
        EXPECT_EQ(55, int(t[7]));       // sRGB 128 -> 0.2158 -> 55
        EXPECT_EQ(0, int(t[11]));
        EXPECT_EQ(13, int(t[15]));      // sRGB 64 -> 0.0513 -> 13
        EXPECT_EQ(40, int(t[12]));
    }

    // An RGB base colour without a roughness image is used as it is; it has no alpha, so no coverage texture even with MASK (cutoff defaults to 0.5).
    const Materials::Data opaque_material = Materials::get_data(models[1].get_material().get_ID());
    EXPECT_TRUE(opaque_material.flags.is_set(MaterialFlag::Cutout) && opaque_material.flags.not_set(MaterialFlag::ThinWalled));
    EXPECT_EQ(0.5f, opaque_material.coverage);
    EXPECT_TRUE(opaque_material.coverage_texture_ID == TextureID::invalid_UID());
    EXPECT_TRUE(opaque_material.metallic_texture_ID == TextureID::invalid_UID());
    EXPECT_TRUE(opaque_material.tint_roughness_texture_ID != TextureID::invalid_UID());
    if (opaque_material.tint_roughness_texture_ID != TextureID::invalid_UID()) {
        Image tint = Textures::get_image_ID(opaque_material.tint_roughness_texture_ID);
        EXPECT_TRUE(tint.get_pixel_format() == PixelFormat::RGB24);
        EXPECT_EQ(std::string("unnamed_image_2"), tint.get_name());
        EXPECT_EQ(9, int(tint.get_pixels<uint8_t>()[11]));
    }

    // An image that cannot be loaded costs the texture, not the material.
    const Materials::Data broken = Materials::get_data(models[2].get_material().get_ID());
    EXPECT_TRUE(broken.tint_roughness_texture_ID == TextureID::invalid_UID());
    EXPECT_FLOAT_EQ_EPS(0.3f, broken.roughness, 1e-7f);

    // Of the decoded glTF images only the RGB tint survives as it is; the two that were regrouped are released.
    unsigned image_count = 0;
    for (ImageID id : Images::get_iterable()) { (void)id; ++image_count; }
    EXPECT_EQ(4u, image_count);     // coverage, metallic, tint + roughness, opaque tint
}

CPU_TEST_F(glTFFixture, binary_container_and_vertex_colours) {
    // .glb: JSON chunk + BIN chunk; the buffer has no uri. The primitive adds COLOR_0 (float VEC4) and 32 bit indices.
    TriangleBuffers tri;
    const float colours[12] = {1.0f, 0.5f, 0.0f, 1.0f,   0.25f, 2.0f, -1.0f, 1.0f,   0.0f, 0.0f, 1.0f, 0.5f};
    const unsigned indices[3] = {2, 0, 1};
    const size_t colour_offset = tri.buffer.append(colours, sizeof(colours)), index_offset = tri.buffer.append(indices, sizeof(indices));
    char extra_views[256], extra_accessors[256];
    snprintf(extra_views, sizeof(extra_views), ",{\"buffer\":0,\"byteOffset\":%zu,\"byteLength\":48},{\"buffer\":0,\"byteOffset\":%zu,\"byteLength\":12}]", colour_offset, index_offset);
    snprintf(extra_accessors, sizeof(extra_accessors), ",{\"bufferView\":3,\"componentType\":5126,\"count\":3,\"type\":\"VEC4\"},{\"bufferView\":4,\"componentType\":5125,\"count\":3,\"type\":\"SCALAR\"}]");
    std::string views_and_accessors = tri.views_and_accessors();
    views_and_accessors.replace(views_and_accessors.find("],\"accessors\""), 1, extra_views);
    views_and_accessors.replace(views_and_accessors.size() - 1, 1, extra_accessors);

    std::string json = "{\"asset\":{\"version\":\"2.0\"},\"buffers\":[{\"byteLength\":" + std::to_string(tri.buffer.bytes.size()) + "}]," + views_and_accessors + ","
        "\"meshes\":[{\"name\":\"Coloured\",\"primitives\":[{\"attributes\":{\"POSITION\":0,\"COLOR_0\":4},\"indices\":5}]}],"
        "\"nodes\":[{\"mesh\":0},{\"name\":\"Empty\"}],\"scenes\":[{\"nodes\":[0,1]}],\"scene\":0}";
    while (json.size() % 4) json += ' ';
    std::vector<unsigned char> binary = tri.buffer.bytes;
    while (binary.size() % 4) binary.push_back(0);

    std::string glb;
    auto put_u32 = [&](uint32_t v) { glb.append(reinterpret_cast<const char*>(&v), 4); };
    put_u32(0x46546C67u); put_u32(2); put_u32(uint32_t(12 + 8 + json.size() + 8 + binary.size()));
    put_u32(uint32_t(json.size())); put_u32(0x4E4F534Au); glb += json;
    put_u32(uint32_t(binary.size())); put_u32(0x004E4942u); glb.append(reinterpret_cast<const char*>(binary.data()), binary.size());

    SceneNode root = glTFLoader::load(write("coloured.glb", glb));
    EXPECT_TRUE(root != SceneNode::invalid());
    if (root == SceneNode::invalid()) return;
    EXPECT_EQ(std::string("Scene root"), root.get_name());      // two roots in the scene: a common parent
    EXPECT_EQ(size_t(2), root.get_children().size());

    const std::vector<MeshModel> models = all_models();
    EXPECT_EQ(size_t(1), models.size());
    if (models.size() != 1) return;
    Mesh mesh = models[0].get_mesh();
    EXPECT_TRUE(mesh.get_normals() == nullptr && mesh.get_texcoords() == nullptr);
    const TintRoughness* tints = mesh.get_tint_and_roughness();
    EXPECT_TRUE(tints != nullptr);
    if (tints) {
        EXPECT_EQ(255, int(tints[0].r)); EXPECT_EQ(128, int(tints[0].g)); EXPECT_EQ(0, int(tints[0].b)); EXPECT_EQ(255, int(tints[0].roughness));
        EXPECT_EQ(64, int(tints[1].r)); EXPECT_EQ(255, int(tints[1].g)); EXPECT_EQ(0, int(tints[1].b));       // clamped to [0, 1]
        EXPECT_EQ(255, int(tints[2].b)); EXPECT_EQ(255, int(tints[2].roughness));                             // alpha is not roughness
    }
    EXPECT_EQ(0u, mesh.get_primitives()[0].x); EXPECT_EQ(2u, mesh.get_primitives()[0].y); EXPECT_EQ(1u, mesh.get_primitives()[0].z);
}

CPU_TEST_F(glTFFixture, malformed_files_create_nothing) {
    TriangleBuffers tri;
    const std::string head = "{\"asset\":{\"version\":\"2.0\"}," + tri.data_uri_buffer() + ",";
    const std::string tail = "," + std::string(triangle_mesh) + ",\"materials\":[{}],\"nodes\":[{\"mesh\":0}],\"scenes\":[{\"nodes\":[0]}]";

    // an index accessor that reaches past the buffer
    std::string views_and_accessors = tri.views_and_accessors();
    views_and_accessors.replace(views_and_accessors.find("\"componentType\":5123,\"count\":3"), 30, "\"componentType\":5123,\"count\":300");
    EXPECT_TRUE(glTFLoader::load(write("overrun.gltf", head + views_and_accessors + tail + ",\"scene\":0}")) == SceneNode::invalid());
    // indices that name vertices the mesh does not have
    TriangleBuffers bad_indices;
    const unsigned short wild[3] = {0, 1, 7};
    std::memcpy(bad_indices.buffer.bytes.data() + bad_indices.indices_offset, wild, sizeof(wild));
    EXPECT_TRUE(glTFLoader::load(write("wild.gltf", "{\"asset\":{\"version\":\"2.0\"}," + bad_indices.data_uri_buffer() + "," + bad_indices.views_and_accessors() + tail + ",\"scene\":0}")) == SceneNode::invalid());
    // no default scene: nothing to import (glTFLoader.cpp:700-702), and the meshes made on the way are released again
    EXPECT_TRUE(glTFLoader::load(write("sceneless.gltf", head + tri.views_and_accessors() + tail + "}")) == SceneNode::invalid());
    // not JSON, not version 2, not there, not glTF by name
    EXPECT_TRUE(glTFLoader::load(write("garbage.gltf", "{\"asset\":")) == SceneNode::invalid());
    EXPECT_TRUE(glTFLoader::load(write("old.gltf", "{\"asset\":{\"version\":\"1.0\"}}")) == SceneNode::invalid());
    EXPECT_TRUE(glTFLoader::load((directory / "absent.gltf").string()) == SceneNode::invalid());
    EXPECT_TRUE(glTFLoader::load(write("garbage.glb", "glTF but not really")) == SceneNode::invalid());
    EXPECT_TRUE(glTFLoader::load("model.obj") == SceneNode::invalid());

    unsigned meshes = 0, models = 0;
    for (MeshModelID id : MeshModels::get_iterable()) { (void)id; ++models; }
    for (unsigned i = 1; i < Meshes::capacity(); ++i) meshes += Meshes::has(MeshID(i)) ? 1 : 0;
    EXPECT_EQ(0u, models);
    EXPECT_EQ(0u, meshes);
}

GPU_TEST_F(glTFFixture, loaded_gltf_renders_through_the_renderer) {
    // file -> glTFLoader -> SimpleViewer defaults -> HIPRenderer::Renderer. A floor quad (two triangles, wound counter-clockwise
    // seen from above in glTF's frame) under the default directional light: it only lights up if the mirroring kept it front facing.
    std::error_code error;
    const std::filesystem::path data = std::filesystem::read_symlink("/proc/self/exe", error).parent_path() / ".." / ".." / "bifrost3d_amd" / "data";
    HIPRenderer::Renderer* renderer = HIPRenderer::Renderer::initialize(0, data);
    EXPECT_TRUE(renderer != nullptr);
    if (!renderer) return;

    BufferWriter buffer;
    const float positions[12] = {-1, 0, 1,   1, 0, 1,   1, 0, -1,   -1, 0, -1};     // +z towards the viewer in a right-handed frame
    const float normals[12] = {0, 1, 0,   0, 1, 0,   0, 1, 0,   0, 1, 0};
    const unsigned char indices[6] = {0, 1, 2,   0, 2, 3};
    const size_t position_offset = buffer.append(positions, sizeof(positions)), normal_offset = buffer.append(normals, sizeof(normals)), index_offset = buffer.append(indices, sizeof(indices));
    char text[2048];
    snprintf(text, sizeof(text),
        "{\"asset\":{\"version\":\"2.0\"},\"buffers\":[{\"byteLength\":%zu,\"uri\":\"floor.bin\"}],"
        "\"bufferViews\":[{\"buffer\":0,\"byteOffset\":%zu,\"byteLength\":48},{\"buffer\":0,\"byteOffset\":%zu,\"byteLength\":48},{\"buffer\":0,\"byteOffset\":%zu,\"byteLength\":6}],"
        "\"accessors\":[{\"bufferView\":0,\"componentType\":5126,\"count\":4,\"type\":\"VEC3\",\"min\":[-1,0,-1],\"max\":[1,0,1]},{\"bufferView\":1,\"componentType\":5126,\"count\":4,\"type\":\"VEC3\"},{\"bufferView\":2,\"componentType\":5121,\"count\":6,\"type\":\"SCALAR\"}],"
        "\"meshes\":[{\"name\":\"Floor\",\"primitives\":[{\"attributes\":{\"POSITION\":0,\"NORMAL\":1},\"indices\":2,\"material\":0}]}],"
        "\"materials\":[{\"name\":\"Chalk\",\"pbrMetallicRoughness\":{\"baseColorFactor\":[0.9,0.9,0.9,1],\"metallicFactor\":0,\"roughnessFactor\":0.9}}],"
        "\"nodes\":[{\"name\":\"Floor\",\"mesh\":0,\"scale\":[2,2,2]}],\"scenes\":[{\"nodes\":[0]}],\"scene\":0}",
        buffer.bytes.size(), position_offset, normal_offset, index_offset);
    write("floor.bin", std::string(buffer.bytes.begin(), buffer.bytes.end()));
    const std::string path = write("floor.gltf", text);

    SceneRoot scene = SceneRoot("Loaded", RGB(0.1f, 0.1f, 0.1f));
    SceneNode loaded = glTFLoader::load(path);
    EXPECT_TRUE(loaded != SceneNode::invalid());
    if (loaded == SceneNode::invalid()) { delete renderer; return; }
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
    EXPECT_TRUE(brightest > 0.5);   // environment tint is 0.1: anything brighter is the lit floor
    delete renderer;
}

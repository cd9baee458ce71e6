// ObjLoader.cpp -- see ObjLoader.h. Mapping rules follow extensions/ObjLoader/ObjLoader/ObjLoader.cpp ("OL"):
//   materials  OL:150-203  tint = Kd, roughness = (2 / (Ns + 2))^(1/4), metallic = illum 3 or 5, specularity = mean(Ks),
//              coverage = d; coverage texture = map_d, else the alpha channel of map_Kd when it is not all ones (OL:57-128)
//   shapes     OL:205-296  mesh flags from the first vertex of the shape, vertices de-duplicated on the (v, vn, vt) index
//              triple, one material per shape (the first face's), one scene node per shape
#include "ObjLoader.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <unordered_map>
#include <vector>

using namespace Bifrost;
using namespace Bifrost::Assets;
using namespace Bifrost::Math;
using namespace Bifrost::Scene;

namespace ObjLoader {

namespace {

// ---- parsed file ---------------------------------------------------------------------------------------------------------
struct Index { int v = -1, vn = -1, vt = -1; };
struct IndexLess {
    bool operator()(const Index& a, const Index& b) const {
        if (a.v != b.v) return a.v < b.v;
        if (a.vn != b.vn) return a.vn < b.vn;
        return a.vt < b.vt;
    }
};
struct Shape { std::string name; std::vector<Index> indices; std::vector<int> material_ids; };   // 3 indices and 1 material id per triangle
struct MtlMaterial {
    std::string name;
    float diffuse[3] = {0.8f, 0.8f, 0.8f}, specular[3] = {0, 0, 0};   // defaults of the MTL reader the reference uses
    float shininess = 1.0f, dissolve = 1.0f;
    int illum = 0;
    std::string diffuse_texname, alpha_texname, roughness_texname, metallic_texname, emissive_texname, normal_texname;
};
struct ObjFile {
    std::vector<float> positions, normals, texcoords;
    std::vector<Shape> shapes;
    std::vector<MtlMaterial> materials;
};

void split_path(std::string& directory, std::string& filename, const std::string& path) {
    const size_t slash = path.find_last_of("/\\");
    directory = slash == std::string::npos ? "" : path.substr(0, slash + 1);
    filename = slash == std::string::npos ? path : path.substr(slash + 1);
}

std::string rest_of_line(std::istringstream& line) {
    std::string rest;
    std::getline(line, rest);
    const size_t begin = rest.find_first_not_of(" \t"), end = rest.find_last_not_of(" \t\r");
    return begin == std::string::npos ? "" : rest.substr(begin, end - begin + 1);
}

void parse_mtl(const std::string& path, std::vector<MtlMaterial>& materials, std::unordered_map<std::string, int>& by_name) {
    std::ifstream file(path);
    if (!file) { printf("ObjLoader::load warning: 'Material file %s not found'.\n", path.c_str()); return; }
    std::string text;
    MtlMaterial* current = nullptr;
    while (std::getline(file, text)) {
        std::istringstream line(text);
        std::string key;
        if (!(line >> key) || key[0] == '#') continue;
        if (key == "newmtl") {
            materials.emplace_back();
            current = &materials.back();
            current->name = rest_of_line(line);
            by_name[current->name] = int(materials.size()) - 1;
        } else if (!current) continue;
        else if (key == "Kd") line >> current->diffuse[0] >> current->diffuse[1] >> current->diffuse[2];
        else if (key == "Ks") line >> current->specular[0] >> current->specular[1] >> current->specular[2];
        else if (key == "Ns") line >> current->shininess;
        else if (key == "d") line >> current->dissolve;
        else if (key == "Tr") { float tr = 0; line >> tr; current->dissolve = 1.0f - tr; }
        else if (key == "illum") line >> current->illum;
        else if (key == "map_Kd") current->diffuse_texname = rest_of_line(line);
        else if (key == "map_d") current->alpha_texname = rest_of_line(line);
        else if (key == "map_Pr") current->roughness_texname = rest_of_line(line);
        else if (key == "map_Pm") current->metallic_texname = rest_of_line(line);
        else if (key == "map_Ke") current->emissive_texname = rest_of_line(line);
        else if (key == "norm" || key == "map_Bump" || key == "map_bump" || key == "bump") current->normal_texname = rest_of_line(line);
    }
}

// "v", "v/vt", "v//vn" or "v/vt/vn"; indices are 1 based, negative ones count back from the elements read so far.
bool parse_index(const std::string& token, size_t v_count, size_t vt_count, size_t vn_count, Index& out) {
    int parts[3] = {0, 0, 0};
    int part = 0;
    const char* s = token.c_str();
    while (*s && part < 3) {
        char* end = nullptr;
        const long value = std::strtol(s, &end, 10);
        if (end != s) parts[part] = int(value);
        s = end;
        if (*s == '/') { ++s; ++part; } else break;
    }
    auto resolve = [](int index, size_t count) { return index > 0 ? index - 1 : (index < 0 ? int(count) + index : -1); };
    out.v = resolve(parts[0], v_count);
    out.vt = resolve(parts[1], vt_count);
    out.vn = resolve(parts[2], vn_count);
    return out.v >= 0 && size_t(out.v) < v_count && (out.vt < 0 || size_t(out.vt) < vt_count) && (out.vn < 0 || size_t(out.vn) < vn_count);
}

bool parse_obj(const std::string& path, const std::string& directory, ObjFile& obj) {
    std::ifstream file(path);
    if (!file) { printf("ObjLoader::load error: 'Cannot open file [%s]'.\n", path.c_str()); return false; }
    std::unordered_map<std::string, int> material_by_name;
    Shape shape;
    int material = -1;
    auto flush = [&]() { if (!shape.indices.empty()) obj.shapes.push_back(shape); shape = Shape(); };
    std::string text;
    while (std::getline(file, text)) {
        std::istringstream line(text);
        std::string key;
        if (!(line >> key) || key[0] == '#') continue;
        if (key == "v") { float x = 0, y = 0, z = 0; line >> x >> y >> z; obj.positions.insert(obj.positions.end(), {x, y, z}); }
        else if (key == "vn") { float x = 0, y = 0, z = 0; line >> x >> y >> z; obj.normals.insert(obj.normals.end(), {x, y, z}); }
        else if (key == "vt") { float u = 0, v = 0; line >> u >> v; obj.texcoords.insert(obj.texcoords.end(), {u, v}); }
        else if (key == "f") {
            std::vector<Index> corners;
            std::string token;
            bool valid = true;
            while (line >> token) {
                Index index;
                valid = parse_index(token, obj.positions.size() / 3, obj.texcoords.size() / 2, obj.normals.size() / 3, index) && valid;
                corners.push_back(index);
            }
            if (!valid || corners.size() < 3) { printf("ObjLoader::load warning: 'Skipping a face with invalid indices'.\n"); continue; }
            for (size_t k = 1; k + 1 < corners.size(); ++k) {   // triangle fan
                shape.indices.push_back(corners[0]); shape.indices.push_back(corners[k]); shape.indices.push_back(corners[k + 1]);
                shape.material_ids.push_back(material);
            }
        } else if (key == "o" || key == "g") { flush(); shape.name = rest_of_line(line); }
        else if (key == "usemtl") {
            const auto found = material_by_name.find(rest_of_line(line));
            material = found == material_by_name.end() ? -1 : found->second;
        } else if (key == "mtllib") {
            std::string name;
            while (line >> name) parse_mtl(directory + name, obj.materials, material_by_name);
        }
    }
    flush();
    return true;
}

// ---- images (OL:57-128) ------------------------------------------------------------------------------------------------
typedef std::unordered_map<std::string, Image> ImageCache;

void load_material_images(const std::vector<MtlMaterial>& materials, const std::string& directory, ImageLoader image_loader, ImageCache& tint_cache,
                          ImageCache& coverage_cache) {
    for (const MtlMaterial& mat : materials) {
        if (!mat.alpha_texname.empty() && !coverage_cache.count(mat.alpha_texname)) {
            const std::string image_path = directory + mat.alpha_texname;
            Image coverage = image_loader ? image_loader(image_path) : Image();
            if (coverage.exists() && coverage.get_pixel_format() != PixelFormat::Alpha8) {   // change_format(Alpha8): keep the first channel
                const unsigned n = coverage.get_pixel_count();
                const size_t stride = Image::bytes_per_pixel(coverage.get_pixel_format());
                std::vector<unsigned char> alpha(n);
                const bool is_float = coverage.get_pixel_format() == PixelFormat::Intensity_Float || coverage.get_pixel_format() == PixelFormat::RGB_Float ||
                                      coverage.get_pixel_format() == PixelFormat::RGBA_Float;
                for (unsigned p = 0; p < n; ++p) {
                    const unsigned char* pixel = coverage.get_pixels<unsigned char>() + p * stride;
                    float value = 0;
                    if (is_float) std::memcpy(&value, pixel, 4); else value = pixel[0] / 255.0f;
                    alpha[p] = (unsigned char)(std::fmin(std::fmax(value, 0.0f), 1.0f) * 255.0f + 0.5f);
                }
                coverage = Image::create2D(image_path, PixelFormat::Alpha8, false, coverage.get_width(), coverage.get_height(), alpha.data());
            }
            coverage_cache.insert({mat.alpha_texname, coverage});
            if (!coverage.exists()) printf("ObjLoader::load error: Could not load image at '%s'.\n", image_path.c_str());
        }

        if (!mat.diffuse_texname.empty() && !tint_cache.count(mat.diffuse_texname)) {
            const std::string image_path = directory + mat.diffuse_texname;
            Image tint = image_loader ? image_loader(image_path) : Image();
            tint_cache.insert({mat.diffuse_texname, tint});
            if (tint.exists()) {
                // A tint image with an alpha channel: the alpha becomes the coverage (unless it is one everywhere) and the channel is
                // set to one, where the material reads roughness.
                if (tint.get_pixel_format() == PixelFormat::RGBA32) {
                    const unsigned n = tint.get_pixel_count();
                    std::vector<unsigned char> alpha(n);
                    unsigned char* pixels = tint.get_pixels<unsigned char>();
                    unsigned char min_coverage = 255;
                    for (unsigned p = 0; p < n; ++p) { alpha[p] = pixels[4 * p + 3]; min_coverage = std::min(min_coverage, alpha[p]); pixels[4 * p + 3] = 255; }
                    if (min_coverage < 255)
                        coverage_cache.insert({mat.diffuse_texname, Image::create2D(image_path, PixelFormat::Alpha8, false, tint.get_width(), tint.get_height(), alpha.data())});
                } else if (tint.get_pixel_format() == PixelFormat::RGBA_Float) {
                    const unsigned n = tint.get_pixel_count();
                    std::vector<unsigned char> alpha(n);
                    float* pixels = tint.get_pixels<float>();
                    float min_coverage = 1.0f;
                    for (unsigned p = 0; p < n; ++p) {
                        min_coverage = std::fmin(min_coverage, pixels[4 * p + 3]);
                        alpha[p] = (unsigned char)(pixels[4 * p + 3] * 255 + 0.5f);
                        pixels[4 * p + 3] = 1.0f;
                    }
                    if (min_coverage < 1.0f)
                        coverage_cache.insert({mat.diffuse_texname, Image::create2D(image_path, PixelFormat::Alpha8, false, tint.get_width(), tint.get_height(), alpha.data())});
                }
            } else
                printf("ObjLoader::load error: Could not load image at '%s'.\n", image_path.c_str());
        }

        if (!mat.roughness_texname.empty()) printf("ObjLoader::load error: Roughness texture not supported.\n");
        if (!mat.metallic_texname.empty()) printf("ObjLoader::load error: Metallic texture not supported.\n");
        if (!mat.emissive_texname.empty()) printf("ObjLoader::load error: Emissive texture not supported.\n");
        if (!mat.normal_texname.empty()) printf("ObjLoader::load error: Normal map not supported.\n");
    }
}

} // namespace

SceneNode load(const std::string& path, ImageLoader image_loader) {
    std::string directory, filename;
    split_path(directory, filename, path);

    ObjFile obj;
    if (!parse_obj(path, directory, obj)) return SceneNode::invalid();

    const std::string stem = filename.size() > 4 ? filename.substr(0, filename.size() - 4) : filename;
    SceneNode root = obj.shapes.size() > 1u ? SceneNode(stem) : SceneNode::invalid();

    ImageCache tint_images, coverage_images;
    load_material_images(obj.materials, directory, image_loader, tint_images, coverage_images);

    std::vector<Material> materials(obj.materials.size());
    for (size_t i = 0; i < obj.materials.size(); ++i) {
        const MtlMaterial& mtl = obj.materials[i];
        Materials::Data data = {};
        data.flags = MaterialFlag::None;
        data.tint = RGB(mtl.diffuse[0], mtl.diffuse[1], mtl.diffuse[2]);
        const float ggx_alpha_squared = 2.0f / (mtl.shininess + 2.0f);   // Blinn shininess -> GGX alpha
        data.roughness = std::pow(ggx_alpha_squared, 0.25f);             // roughness = sqrt(ggx_alpha)
        data.metallic = (mtl.illum == 3 || mtl.illum == 5) ? 1.0f : 0.0f;
        data.specularity = (mtl.specular[0] + mtl.specular[1] + mtl.specular[2]) / 3.0f;
        data.coverage = mtl.dissolve;
        if (data.coverage <= 0.0f)
            printf("ObjLoader::load warning: Coverage set to %.3f. Material %s is completely transparent.\n", data.coverage, mtl.name.c_str());

        if (!mtl.alpha_texname.empty() || !mtl.diffuse_texname.empty()) {   // map_d first, then the alpha channel of map_Kd
            Image alpha;
            auto found = coverage_images.find(mtl.alpha_texname);
            if (found != coverage_images.end()) alpha = found->second;
            if (!alpha.exists()) {
                found = coverage_images.find(mtl.diffuse_texname);
                if (found != coverage_images.end()) alpha = found->second;
            }
            if (alpha.exists()) data.coverage_texture_ID = Textures::create2D(alpha.get_ID());
        }
        if (!mtl.diffuse_texname.empty()) {
            const auto found = tint_images.find(mtl.diffuse_texname);
            if (found != tint_images.end() && found->second.exists()) data.tint_roughness_texture_ID = Textures::create2D(found->second.get_ID());
        }
        materials[i] = Material(mtl.name, data);
    }

    for (const Shape& shape : obj.shapes) {
        const Index first = shape.indices[0];
        MeshFlags mesh_flags = MeshFlag::Position;
        if (first.vn != -1) mesh_flags |= MeshFlag::Normal;
        if (first.vt != -1) mesh_flags |= MeshFlag::Texcoord;

        std::map<Index, unsigned int, IndexLess> vertex_index_map;
        unsigned int vertex_count = 0;
        for (const Index& index : shape.indices)
            if (vertex_index_map.emplace(index, vertex_count).second) ++vertex_count;

        const unsigned int triangle_count = unsigned(shape.indices.size() / 3);
        Mesh mesh = Mesh(shape.name, triangle_count, vertex_count, mesh_flags);
        for (unsigned int p = 0; p < triangle_count; ++p)
            mesh.get_primitives()[p] = {vertex_index_map[shape.indices[3 * p]], vertex_index_map[shape.indices[3 * p + 1]], vertex_index_map[shape.indices[3 * p + 2]]};
        for (const auto& entry : vertex_index_map) {
            const Index& index = entry.first;
            mesh.get_positions()[entry.second] = Vector3f(obj.positions[3 * index.v], obj.positions[3 * index.v + 1], obj.positions[3 * index.v + 2]);
            if (mesh.get_normals())
                mesh.get_normals()[entry.second] = index.vn >= 0 ? Vector3f(obj.normals[3 * index.vn], obj.normals[3 * index.vn + 1], obj.normals[3 * index.vn + 2]) : Vector3f(0, 0, 1);
            if (mesh.get_texcoords())
                mesh.get_texcoords()[entry.second] = index.vt >= 0 ? Vector2f{obj.texcoords[2 * index.vt], obj.texcoords[2 * index.vt + 1]} : Vector2f{0, 0};
        }
        mesh.compute_bounds();

        SceneNode node = SceneNode(shape.name);
        if (root != SceneNode::invalid()) node.set_parent(root);
        else root = node;

        const int material_index = shape.material_ids[0];   // no per-face materials, like the reference
        MeshModel(node, mesh, material_index >= 0 ? materials[material_index] : Material::invalid());
    }
    return root;
}

bool file_supported(const std::string& filename) {
    const std::string end = ".obj";
    return filename.length() >= end.length() && filename.compare(filename.length() - end.length(), end.length(), end) == 0;
}

} // namespace ObjLoader

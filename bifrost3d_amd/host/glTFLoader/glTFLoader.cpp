// glTFLoader.cpp -- see glTFLoader.h. Follows the glTF 2.0 specification for the container and
// extensions/glTFLoader/glTFLoader/glTFLoader.cpp for what becomes of it in the Bifrost data model.
#include "glTFLoader.h"

#include "../ImageIO/ImageLoader.h"
#include "../ImageIO/PngImage.h"
#include "../Json.h"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <tuple>
#include <vector>

using namespace Bifrost::Assets;
using namespace Bifrost::Math;
using namespace Bifrost::Scene;

namespace glTFLoader {

namespace {

// ------------------------------------------------------------------------------------------------
// Container: file reading, GLB chunks, data URIs, buffers, buffer views and accessors.
// ------------------------------------------------------------------------------------------------

inline bool string_ends_with(const std::string& s, const std::string& end) {
    return s.length() >= end.length() && s.compare(s.length() - end.length(), end.length(), end) == 0;
}

bool read_file(const std::string& path, std::vector<uint8_t>& bytes) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    const long size = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    bytes.resize(size > 0 ? size_t(size) : 0);
    const bool ok = bytes.empty() || std::fread(bytes.data(), 1, bytes.size(), f) == bytes.size();
    std::fclose(f);
    return ok;
}

bool decode_base64(const char* begin, const char* end, std::vector<uint8_t>& out) {
    uint32_t bits = 0;
    int bit_count = 0;
    for (const char* p = begin; p != end; ++p) {
        const char c = *p;
        int v;
        if (c >= 'A' && c <= 'Z') v = c - 'A';
        else if (c >= 'a' && c <= 'z') v = c - 'a' + 26;
        else if (c >= '0' && c <= '9') v = c - '0' + 52;
        else if (c == '+' || c == '-') v = 62;
        else if (c == '/' || c == '_') v = 63;
        else if (c == '=' || c == '\n' || c == '\r') continue;
        else return false;
        bits = (bits << 6) | uint32_t(v);
        bit_count += 6;
        if (bit_count >= 8) { bit_count -= 8; out.push_back(uint8_t(bits >> bit_count)); }
    }
    return true;
}

std::string decode_percent_escapes(const std::string& uri) {
    std::string out;
    for (size_t i = 0; i < uri.size(); ++i) {
        if (uri[i] == '%' && i + 2 < uri.size() && std::isxdigit((unsigned char)uri[i + 1]) && std::isxdigit((unsigned char)uri[i + 2])) {
            out += char(std::strtol(uri.substr(i + 1, 2).c_str(), nullptr, 16));
            i += 2;
        } else
            out += uri[i];
    }
    return out;
}

// Bytes behind a uri: a base64 data uri or a file next to the glTF file.
bool load_uri(const std::string& uri, const std::string& directory, std::vector<uint8_t>& bytes) {
    if (uri.compare(0, 5, "data:") == 0) {
        const size_t marker = uri.find(";base64,");
        if (marker == std::string::npos) return false;
        return decode_base64(uri.data() + marker + 8, uri.data() + uri.size(), bytes);
    }
    return read_file(directory + decode_percent_escapes(uri), bytes);
}

enum ComponentType { BYTE = 5120, UNSIGNED_BYTE = 5121, SHORT = 5122, UNSIGNED_SHORT = 5123, UNSIGNED_INT = 5125, FLOAT = 5126 };
enum { MODE_TRIANGLES = 4 };
enum { FILTER_NEAREST = 9728, FILTER_LINEAR = 9729, FILTER_NEAREST_MIPMAP_NEAREST = 9984, FILTER_LINEAR_MIPMAP_NEAREST = 9985,
       FILTER_NEAREST_MIPMAP_LINEAR = 9986, FILTER_LINEAR_MIPMAP_LINEAR = 9987, WRAP_REPEAT = 10497, WRAP_CLAMP_TO_EDGE = 33071, WRAP_MIRRORED_REPEAT = 33648 };

inline int component_size(int component_type) {
    switch (component_type) {
    case BYTE: case UNSIGNED_BYTE: return 1; case SHORT: case UNSIGNED_SHORT: return 2; case UNSIGNED_INT: case FLOAT: return 4; default: return 0;
    }
}
inline int component_count(const std::string& type) {
    const char* names[7] = {"SCALAR", "VEC2", "VEC3", "VEC4", "MAT2", "MAT3", "MAT4"};
    const int counts[7] = {1, 2, 3, 4, 4, 9, 16};
    for (int i = 0; i < 7; ++i) if (type == names[i]) return counts[i];
    return 0;
}

// A resolved accessor: where its first element is, how far apart elements are and what they are made of.
struct Elements {
    const uint8_t* data = nullptr;
    size_t stride = 0, count = 0;
    int component_type = 0, components = 0;
    const uint8_t* at(size_t i) const { return data + i * stride; }
};

struct Document {
    Json::Value json;
    std::string directory;
    std::vector<std::vector<uint8_t>> buffers;

    // Resolves accessor `index`; on failure returns false with the reason in `error`.
    bool resolve(int index, Elements& out, std::string& error) const {
        const Json::Value& accessor = json["accessors"][size_t(index)];
        if (index < 0 || !accessor.is_object()) { error = "accessor " + std::to_string(index) + " does not exist"; return false; }
        if (accessor.has("sparse") || !accessor.has("bufferView")) { error = "sparse accessors and accessors without a buffer view are not supported"; return false; }
        const Json::Value& view = json["bufferViews"][size_t(accessor["bufferView"].as_int(-1))];
        if (!view.is_object()) { error = "accessor " + std::to_string(index) + " names a buffer view that does not exist"; return false; }
        const int buffer_index = view["buffer"].as_int(-1);
        if (buffer_index < 0 || size_t(buffer_index) >= buffers.size()) { error = "buffer view names a buffer that does not exist"; return false; }
        const std::vector<uint8_t>& buffer = buffers[size_t(buffer_index)];

        out.component_type = accessor["componentType"].as_int();
        out.components = component_count(accessor["type"].as_string());
        out.count = size_t(accessor["count"].as_double());
        const size_t element_size = size_t(component_size(out.component_type)) * size_t(out.components);
        if (element_size == 0) { error = "accessor " + std::to_string(index) + " has an unknown component type or element type"; return false; }
        // Accessor::ByteStride, tiny_gltf.h:868-896: the view's stride when it has one, tightly packed otherwise.
        const size_t view_stride = size_t(view["byteStride"].as_double(0.0));
        out.stride = view_stride ? view_stride : element_size;
        if (view_stride % size_t(component_size(out.component_type)) != 0) { error = "buffer view stride is not a multiple of the component size"; return false; }
        const size_t offset = size_t(view["byteOffset"].as_double(0.0)) + size_t(accessor["byteOffset"].as_double(0.0));
        if (out.count > 0 && offset + (out.count - 1) * out.stride + element_size > buffer.size()) { error = "accessor " + std::to_string(index) + " reaches past the end of its buffer"; return false; }
        out.data = buffer.data() + offset;
        return true;
    }
};

bool open_document(const std::string& filename, Document& doc) {
    std::vector<uint8_t> file;
    if (!read_file(filename, file)) { printf("glTFLoader::load error: Could not read '%s'\n", filename.c_str()); return false; }

    const size_t slash = filename.find_last_of("/\\");
    doc.directory = slash == std::string::npos ? "" : filename.substr(0, slash + 1);

    const char* json_begin = reinterpret_cast<const char*>(file.data());
    const char* json_end = json_begin + file.size();
    const uint8_t* binary_chunk = nullptr;
    size_t binary_chunk_size = 0;
    if (string_ends_with(filename, "glb")) {
        // Binary container: 12 byte header, then (length, type, payload) chunks; JSON first, the optional BIN chunk is buffer 0.
        auto u32 = [&](size_t at) { uint32_t v; std::memcpy(&v, file.data() + at, 4); return v; };
        if (file.size() < 20 || u32(0) != 0x46546C67u) { printf("glTFLoader::load error: '%s' is not a binary glTF file\n", filename.c_str()); return false; }
        if (u32(4) != 2) { printf("glTFLoader::load error: Only glTF version 2 is supported\n"); return false; }
        const size_t total = std::min<size_t>(u32(8), file.size());
        size_t at = 12;
        bool have_json = false;
        while (at + 8 <= total) {
            const size_t length = u32(at);
            const uint32_t type = u32(at + 4);
            if (length > total - at - 8) { printf("glTFLoader::load error: Truncated chunk in '%s'\n", filename.c_str()); return false; }
            if (type == 0x4E4F534Au && !have_json) { json_begin = reinterpret_cast<const char*>(file.data() + at + 8); json_end = json_begin + length; have_json = true; }
            else if (type == 0x004E4942u && !binary_chunk) { binary_chunk = file.data() + at + 8; binary_chunk_size = length; }
            at += 8 + ((length + 3) & ~size_t(3));
        }
        if (!have_json) { printf("glTFLoader::load error: No JSON chunk in '%s'\n", filename.c_str()); return false; }
    }

    std::string error;
    if (!Json::Value::parse(json_begin, json_end, doc.json, error) || !doc.json.is_object()) {
        printf("glTFLoader::load error: %s\n", error.empty() ? "the document is not a JSON object" : error.c_str());
        return false;
    }
    const std::string& version = doc.json["asset"]["version"].as_string();
    if (version.empty() || version[0] != '2') { printf("glTFLoader::load error: Only glTF version 2 is supported\n"); return false; }

    const Json::Value& buffers = doc.json["buffers"];
    doc.buffers.resize(buffers.size());
    for (size_t b = 0; b < buffers.size(); ++b) {
        const std::string& uri = buffers[b]["uri"].as_string();
        if (uri.empty()) {
            if (b != 0 || !binary_chunk) { printf("glTFLoader::load error: Buffer %zu has no uri and there is no binary chunk\n", b); return false; }
            doc.buffers[b].assign(binary_chunk, binary_chunk + binary_chunk_size);
        } else if (!load_uri(uri, doc.directory, doc.buffers[b])) {
            printf("glTFLoader::load error: Could not load buffer %zu ('%.64s')\n", b, uri.c_str());
            return false;
        }
        if (doc.buffers[b].size() < size_t(buffers[b]["byteLength"].as_double())) { printf("glTFLoader::load error: Buffer %zu is shorter than its byteLength\n", b); return false; }
    }
    return true;
}

// name_unnamed_resources, glTFLoader.cpp:55-62
std::string resource_name(const Json::Value& resource, const char* prefix, size_t index) {
    const std::string& name = resource["name"].as_string();
    return name.empty() ? prefix + std::to_string(index) : name;
}

// ------------------------------------------------------------------------------------------------
// Texture sampler conversion (glTFLoader.cpp:68-151).
// ------------------------------------------------------------------------------------------------

inline MagnificationFilter convert_magnification_filter(int filter) { return filter == FILTER_NEAREST ? MagnificationFilter::None : MagnificationFilter::Linear; }

inline MinificationFilter convert_minification_filter(int filter) {
    switch (filter) {
    case FILTER_NEAREST: return MinificationFilter::None;
    case FILTER_LINEAR: return MinificationFilter::Linear;
    case FILTER_NEAREST_MIPMAP_NEAREST: printf("GLTFLoader::load warning: Unsupported minification filter NEAREST_MIPMAP_NEAREST. Using LINEAR_MIPMAP_LINEAR.\n"); return MinificationFilter::Trilinear;
    case FILTER_LINEAR_MIPMAP_NEAREST: printf("GLTFLoader::load warning: Unsupported minification filter LINEAR_MIPMAP_NEAREST. Using LINEAR_MIPMAP_LINEAR.\n"); return MinificationFilter::Trilinear;
    case FILTER_NEAREST_MIPMAP_LINEAR: printf("GLTFLoader::load warning: Unsupported minification filter NEAREST_MIPMAP_LINEAR. Using LINEAR_MIPMAP_LINEAR.\n"); return MinificationFilter::Trilinear;
    case FILTER_LINEAR_MIPMAP_LINEAR: return MinificationFilter::Trilinear;
    default: printf("glTFLoader::load warning: Unknown minification filter mode %u.\n", filter); return MinificationFilter::Trilinear;
    }
}

inline WrapMode convert_wrap_mode(int mode) {
    if (mode == WRAP_CLAMP_TO_EDGE) return WrapMode::Clamp;
    if (mode == WRAP_MIRRORED_REPEAT) printf("glTFLoader::load error: Mirrored repeat wrap mode not supported.\n");
    return WrapMode::Repeat;
}

struct SamplerParams {
    MagnificationFilter magnification_filter = MagnificationFilter::Linear;
    MinificationFilter minification_filter = MinificationFilter::Trilinear;
    WrapMode wrap_U = WrapMode::Repeat, wrap_V = WrapMode::Repeat;

    // Absent members take tinygltf's defaults: filters unset (-1, which the conversions map to linear / trilinear), wrapping REPEAT.
    void parse(const Json::Value& sampler) {
        magnification_filter = convert_magnification_filter(sampler["magFilter"].as_int(-1));
        minification_filter = sampler.has("minFilter") ? convert_minification_filter(sampler["minFilter"].as_int()) : MinificationFilter::Trilinear;
        wrap_U = convert_wrap_mode(sampler["wrapS"].as_int(WRAP_REPEAT));
        wrap_V = convert_wrap_mode(sampler["wrapT"].as_int(WRAP_REPEAT));
    }
    TextureID create_texture_2D(Image image) const {
        return image.exists() ? Textures::create2D(image.get_ID(), magnification_filter, minification_filter, wrap_U, wrap_V) : TextureID::invalid_UID();
    }
};

// ------------------------------------------------------------------------------------------------
// Image conversion (glTFLoader.cpp:107-249). glTF keeps (tint, coverage) in one image and (-, roughness, metallic) in
// another; Bifrost keeps (tint, roughness) together and metallic and coverage as single channel images.
// ------------------------------------------------------------------------------------------------

enum class ImageUsage { Tint = 1, Coverage = 2, Metallic = 4, Roughness = 8, Tint_roughness = 9 };

inline const char* to_string(ImageUsage usage) {
    switch (usage) {
    case ImageUsage::Tint: return "tint"; case ImageUsage::Coverage: return "coverage"; case ImageUsage::Metallic: return "metallic";
    case ImageUsage::Roughness: return "roughness"; case ImageUsage::Tint_roughness: return "tint_roughness"; default: return "unknown";
    }
}

using ImageCache = std::map<std::tuple<unsigned, unsigned, int>, Image>;

inline bool is_byte_format(PixelFormat f) { return f == PixelFormat::Alpha8 || f == PixelFormat::Intensity8 || f == PixelFormat::RGB24 || f == PixelFormat::RGBA32; }
inline bool has_alpha(PixelFormat f) { return f == PixelFormat::Alpha8 || f == PixelFormat::RGBA32 || f == PixelFormat::RGBA_Float; }
inline float sRGB_to_linear(float v) { return v < 0.04045f ? v * 0.0773993808f : std::pow(v * 0.9478672986f + 0.0521327014f, 2.4f); }   // BF/Math/Color.h:356-361
inline unsigned char to_unorm8(float v) { return (unsigned char)((v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v)) * 255.0f + 0.5f); }          // BF/Math/FixedPointTypes.h UNorm8::to_byte

// One channel of one pixel as the reference's Image::get_pixel reports it: RGBA order, single channel formats broadcast
// (Alpha8 in alpha only), colours of sRGB byte images decoded to linear.
float get_channel(Image image, unsigned pixel, int channel) {
    const PixelFormat format = image.get_pixel_format();
    const int channels = channel_count(format);
    const void* pixels = Images::get_pixels(image.get_ID());
    if (format == PixelFormat::Alpha8) return channel == 3 ? static_cast<const uint8_t*>(pixels)[pixel] / 255.0f : 1.0f;
    const int source_channel = channels == 1 ? 0 : channel;
    if (source_channel >= channels) return 1.0f;
    if (!is_byte_format(format)) return static_cast<const float*>(pixels)[size_t(pixel) * channels + source_channel];
    const float v = static_cast<const uint8_t*>(pixels)[size_t(pixel) * channels + source_channel] / 255.0f;
    return Images::is_sRGB(image.get_ID()) && channel != 3 ? sRGB_to_linear(v) : v;
}

// Extracts a single channel from an image or retrieves it from the cache. Invalid when the image lacks the channel or the
// channel is one everywhere (the multiplicative identity needs no texture).
Image extract_channel(Image image, int channel, ImageUsage usage, const std::string& name, ImageCache& converted_images) {
    if (!image.exists() || channel >= channel_count(image.get_pixel_format())) return Image();
    if (image.get_pixel_format() == PixelFormat::Alpha8 && channel == 0) return image;

    const auto key = std::make_tuple(image.get_ID().get_index(), 0u, int(usage));
    const auto cached = converted_images.find(key);
    if (cached != converted_images.end()) return cached->second;

    const PixelFormat format = image.get_pixel_format();
    const unsigned pixel_count = image.get_pixel_count();
    std::vector<unsigned char> values(pixel_count);
    unsigned char min_value = 255;
    if (format == PixelFormat::RGB24 || format == PixelFormat::RGBA32) {
        const int pixel_size = channel_count(format);
        const unsigned char* source = static_cast<const unsigned char*>(Images::get_pixels(image.get_ID())) + channel;
        for (unsigned p = 0; p < pixel_count; ++p) { values[p] = source[size_t(p) * pixel_size]; min_value = std::min(min_value, values[p]); }
    } else
        for (unsigned p = 0; p < pixel_count; ++p) { values[p] = (unsigned char)(get_channel(image, p, channel) * 255 + 0.5f); min_value = std::min(min_value, values[p]); }

    Image single_channel_image;
    if (min_value < 255)
        single_channel_image = Image::create2D(name + "_" + to_string(usage), PixelFormat::Alpha8, false, image.get_width(), image.get_height(), values.data());
    converted_images.insert({key, single_channel_image});
    return single_channel_image;
}

// ImageUtils::combine_tint_roughness, BF/Assets/Image.cpp:591-705, with the roughness in `roughness_channel` of its image.
Image combine_tint_roughness(Image tint, Image roughness, int roughness_channel) {
    const unsigned width = tint.get_width(), height = tint.get_height(), pixel_count = width * height;
    const PixelFormat tint_format = tint.get_pixel_format();

    if (!roughness.exists()) {
        if (!has_alpha(tint_format)) return tint;
        // Set roughness to the multiplicative identity.
        if (tint_format == PixelFormat::RGBA32) {
            std::vector<unsigned char> pixels(static_cast<const unsigned char*>(Images::get_pixels(tint.get_ID())), static_cast<const unsigned char*>(Images::get_pixels(tint.get_ID())) + size_t(pixel_count) * 4);
            for (unsigned p = 0; p < pixel_count; ++p) pixels[4 * size_t(p) + 3] = 255;
            return Image::create2D(tint.get_name(), PixelFormat::RGBA32, true, width, height, pixels.data());
        }
        std::vector<float> pixels(static_cast<const float*>(Images::get_pixels(tint.get_ID())), static_cast<const float*>(Images::get_pixels(tint.get_ID())) + size_t(pixel_count) * 4);
        for (unsigned p = 0; p < pixel_count; ++p) pixels[4 * size_t(p) + 3] = 1.0f;
        return Image::create2D(tint.get_name(), PixelFormat::RGBA_Float, Images::is_sRGB(tint.get_ID()), width, height, pixels.data());
    }

    if (roughness.get_width() != width || roughness.get_height() != height) {
        printf("glTFLoader::load error: Tint image '%s' and roughness image '%s' differ in size. The roughness image is ignored.\n", tint.get_name().c_str(), roughness.get_name().c_str());
        return combine_tint_roughness(tint, Image(), roughness_channel);
    }

    const PixelFormat roughness_format = roughness.get_pixel_format();
    const std::string name = tint.get_name() + "_" + roughness.get_name();
    const bool tint_is_byte = tint_format == PixelFormat::RGB24 || tint_format == PixelFormat::RGBA32;
    if (tint_is_byte && is_byte_format(roughness_format)) {
        const int tint_pixel_size = channel_count(tint_format), roughness_pixel_size = channel_count(roughness_format);
        const int channel = std::min(roughness_channel, roughness_pixel_size - 1);
        const unsigned char* tint_pixels = static_cast<const unsigned char*>(Images::get_pixels(tint.get_ID()));
        const unsigned char* roughness_pixels = static_cast<const unsigned char*>(Images::get_pixels(roughness.get_ID())) + channel;
        // Roughness is linear in a Roughness8 image and in images not flagged sRGB; otherwise it sits in an sRGB encoded colour and is decoded.
        const bool roughness_is_linear = roughness_format == PixelFormat::Roughness8 || !Images::is_sRGB(roughness.get_ID());
        std::vector<unsigned char> pixels(size_t(pixel_count) * 4);
        for (unsigned p = 0; p < pixel_count; ++p) {
            pixels[4 * size_t(p)] = tint_pixels[size_t(p) * tint_pixel_size];
            pixels[4 * size_t(p) + 1] = tint_pixels[size_t(p) * tint_pixel_size + 1];
            pixels[4 * size_t(p) + 2] = tint_pixels[size_t(p) * tint_pixel_size + 2];
            const unsigned char r = roughness_pixels[size_t(p) * roughness_pixel_size];
            pixels[4 * size_t(p) + 3] = roughness_is_linear ? r : to_unorm8(sRGB_to_linear(r / 255.0f));
        }
        return Image::create2D(name, PixelFormat::RGBA32, Images::is_sRGB(tint.get_ID()), width, height, pixels.data());
    }

    // Fallback through linear float pixels. A byte tint stays bytes (re-encoded by the image's own sRGB flag).
    if (tint_is_byte) {
        const unsigned char* tint_pixels = static_cast<const unsigned char*>(Images::get_pixels(tint.get_ID()));
        const int tint_pixel_size = channel_count(tint_format);
        std::vector<unsigned char> pixels(size_t(pixel_count) * 4);
        for (unsigned p = 0; p < pixel_count; ++p) {
            for (int c = 0; c < 3; ++c) pixels[4 * size_t(p) + c] = tint_pixels[size_t(p) * tint_pixel_size + c];
            pixels[4 * size_t(p) + 3] = to_unorm8(get_channel(roughness, p, roughness_channel));
        }
        return Image::create2D(name, PixelFormat::RGBA32, Images::is_sRGB(tint.get_ID()), width, height, pixels.data());
    }
    std::vector<float> pixels(size_t(pixel_count) * 4);
    for (unsigned p = 0; p < pixel_count; ++p) {
        for (int c = 0; c < 3; ++c) pixels[4 * size_t(p) + c] = get_channel(tint, p, c);
        pixels[4 * size_t(p) + 3] = get_channel(roughness, p, roughness_channel);
    }
    return Image::create2D(name, PixelFormat::RGBA_Float, false, width, height, pixels.data());
}

Image extract_tint_roughness(Image tint_image, Image roughness_image, const std::string& name, ImageCache& converted_images) {
    if (!tint_image.exists() && !roughness_image.exists()) return Image();
    // The tint image itself when there is no roughness image and the tint image has no alpha channel to mistake for roughness.
    if (!roughness_image.exists() && !has_alpha(tint_image.get_pixel_format())) return tint_image;
    if (!tint_image.exists()) return extract_channel(roughness_image, 1, ImageUsage::Roughness, name, converted_images);

    const auto key = std::make_tuple(tint_image.get_ID().get_index(), roughness_image.exists() ? roughness_image.get_ID().get_index() : 0u, int(ImageUsage::Tint_roughness));
    const auto cached = converted_images.find(key);
    if (cached != converted_images.end()) return cached->second;
    Image combined = combine_tint_roughness(tint_image, roughness_image, 1);
    converted_images.insert({key, combined});
    return combined;
}

// ------------------------------------------------------------------------------------------------
// Transformations (glTFLoader.cpp:251-310).
// ------------------------------------------------------------------------------------------------

struct Affine {     // 3x4 in double, the fourth row is (0, 0, 0, 1)
    double m[3][4];
    static Affine identity() { return {{{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}}}; }
    static Affine from(const Matrix3x4f& f) { Affine a; for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) a.m[r][c] = f.m[r][c]; return a; }
    Matrix3x4f to_float() const { Matrix3x4f f; for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) f.m[r][c] = float(m[r][c]); return f; }
};
inline Affine operator*(const Affine& a, const Affine& b) {
    Affine r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j)
            r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j] + (j == 3 ? a.m[i][3] : 0.0);
    return r;
}

// Decomposes an affine matrix into a Transform with translation, rotation and a volume preserving uniform scale. True when
// the transform represents the matrix, false when something is left over (non-uniform scale, shear).
bool decompose_transformation(const Affine& matrix, Transform& transform) {
    const double (&a)[3][4] = matrix.m;
    const double determinant = a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) + a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
    transform.scale = float(std::pow(determinant, 1 / 3.0));

    double linear[3][3], q[4];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) linear[r][c] = a[r][c] / transform.scale;
    to_quaternion(linear, q);
    const double length = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const bool decomposed_perfectly = 0.999 < length && length < 1.001;
    transform.rotation = Quaternionf(float(q[0] / length), float(q[1] / length), float(q[2] / length), float(q[3] / length));
    transform.translation = Vector3f(float(a[0][3]), float(a[1][3]), float(a[2][3]));
    return decomposed_perfectly;
}

// ------------------------------------------------------------------------------------------------
// Importer
// ------------------------------------------------------------------------------------------------

struct LoadedMesh { Mesh mesh; bool is_used; };

struct Importer {
    const Document& doc;
    std::vector<int> meshes_start_index;    // per glTF mesh, where its primitives' meshes begin in `meshes`; one extra entry at the end
    std::vector<LoadedMesh> meshes;
    std::vector<Material> materials;
    Material default_material;              // for primitives that name no material

    Material material_of(const Json::Value& primitive) {
        const int index = primitive["material"].as_int(-1);
        if (index >= 0 && size_t(index) < materials.size()) return materials[size_t(index)];
        if (!default_material.exists()) {
            Materials::Data data = {};
            data.tint = RGB(1.0f); data.specularity = 0.04f; data.roughness = 1.0f; data.metallic = 0.0f; data.coverage = 1.0f;
            default_material = Material("unnamed_default_material", data);
        }
        return default_material;
    }

    // glTFLoader.cpp:256-351
    SceneNode import_node(size_t node_index, const Affine& parent_transform, int depth) {
        const Json::Value& node = doc.json["nodes"][node_index];

        Affine local_transform;
        const Json::Value& matrix = node["matrix"];
        if (matrix.size() == 16) {
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) local_transform.m[r][c] = matrix[size_t(c * 4 + r)].as_double();   // column major in the file
        } else {
            const Json::Value& s = node["scale"];   // only uniform scaling is representable; the volume preserving mean is used
            const float scale = s.size() != 3 ? 1.0f : float(std::pow(s[size_t(0)].as_double() * s[size_t(1)].as_double() * s[size_t(2)].as_double(), 1 / 3.0));
            const Json::Value& r = node["rotation"];
            const Quaternionf rotation = r.size() != 4 ? Quaternionf::identity() : Quaternionf(float(r[size_t(0)].as_double()), float(r[size_t(1)].as_double()), float(r[size_t(2)].as_double()), float(r[size_t(3)].as_double()));
            const Json::Value& t = node["translation"];
            const Vector3f translation = t.size() != 3 ? Vector3f::zero() : Vector3f(float(t[size_t(0)].as_double()), float(t[size_t(1)].as_double()), float(t[size_t(2)].as_double()));
            local_transform = Affine::from(to_matrix3x4(Transform(translation, rotation, scale)));
        }

        // X is negated as glTF is right-handed and Bifrost left-handed: conjugating with diag(-1, 1, 1) flips the sign of
        // the entries that couple X with Y and Z and of the X translation.
        local_transform.m[0][1] = -local_transform.m[0][1];
        local_transform.m[0][2] = -local_transform.m[0][2];
        local_transform.m[1][0] = -local_transform.m[1][0];
        local_transform.m[2][0] = -local_transform.m[2][0];
        local_transform.m[0][3] = -local_transform.m[0][3];

        const Affine global_transform = parent_transform * local_transform;

        Transform decomposed_global_transform;
        const bool apply_residual_transformation = !decompose_transformation(global_transform, decomposed_global_transform);

        SceneNode scene_node = SceneNode(resource_name(node, "unnamed_node_", node_index), decomposed_global_transform);

        const int glTF_mesh_index = node["mesh"].as_int(-1);
        if (glTF_mesh_index >= 0 && size_t(glTF_mesh_index) + 1 < meshes_start_index.size()) {
            int mesh_index = meshes_start_index[size_t(glTF_mesh_index)];

            // What the Transform cannot express (non-uniform scaling, shearing) is applied to the vertices of a private copy of the mesh.
            const Matrix3x4f residual_transformation = (Affine::from(to_matrix3x4(invert(decomposed_global_transform))) * global_transform).to_float();

            const Json::Value& glTF_mesh = doc.json["meshes"][size_t(glTF_mesh_index)];
            for (const Json::Value& primitive : glTF_mesh["primitives"].elements()) {
                if (primitive["mode"].as_int(MODE_TRIANGLES) != MODE_TRIANGLES) {
                    printf("GLTFLoader::load warning: %s primitive %u not supported.\n", resource_name(glTF_mesh, "unnamed_mesh_", size_t(glTF_mesh_index)).c_str(), unsigned(primitive["mode"].as_int()));
                    continue;
                }
                LoadedMesh& loaded_mesh = meshes[size_t(mesh_index++)];
                Mesh mesh = loaded_mesh.mesh;
                if (apply_residual_transformation) {
                    mesh = MeshUtils::deep_clone(mesh);
                    MeshUtils::transform_mesh(mesh, residual_transformation);
                } else
                    loaded_mesh.is_used = true;
                MeshModel(scene_node, mesh, material_of(primitive));
            }
        }

        if (depth < 256)    // a document whose children form a cycle is malformed; do not recurse forever over it
            for (const Json::Value& child : node["children"].elements()) {
                const int child_index = child.as_int(-1);
                if (child_index < 0 || size_t(child_index) >= doc.json["nodes"].size()) continue;
                SceneNode child_node = import_node(size_t(child_index), global_transform, depth + 1);
                child_node.set_parent(scene_node);
            }

        return scene_node;
    }
};

template <typename T>
void copy_indices(unsigned int* bifrost_indices, const Elements& elements) {
    for (size_t i = 0; i < elements.count; ++i) { T v; std::memcpy(&v, elements.at(i), sizeof(T)); bifrost_indices[i] = v; }
}

// Everything a TRIANGLES primitive needs from the accessors, resolved and checked before any Bifrost resource is made.
struct PrimitiveSource {
    bool supported = false;
    Elements indices, positions, normals, texcoords, colors;
    bool has_indices = false, has_normals = false, has_texcoords = false, has_colors = false;
    Vector3f min_position = Vector3f::zero(), max_position = Vector3f::zero();
};

bool resolve_primitive(const Document& doc, const Json::Value& primitive, PrimitiveSource& out, std::string& error) {
    out.supported = primitive["mode"].as_int(MODE_TRIANGLES) == MODE_TRIANGLES;
    if (!out.supported) return true;

    const Json::Value& attributes = primitive["attributes"];
    if (!attributes.has("POSITION")) { error = "primitive without POSITION"; return false; }
    const int position_accessor = attributes["POSITION"].as_int(-1);
    if (!doc.resolve(position_accessor, out.positions, error)) return false;
    if (out.positions.components != 3 || out.positions.component_type != FLOAT) { error = "POSITION must be float VEC3"; return false; }
    const Json::Value& accessor = doc.json["accessors"][size_t(position_accessor)];
    const Json::Value& lo = accessor["min"];
    const Json::Value& hi = accessor["max"];
    if (lo.size() == 3 && hi.size() == 3) {
        out.min_position = Vector3f(float(lo[size_t(0)].as_double()), float(lo[size_t(1)].as_double()), float(lo[size_t(2)].as_double()));
        out.max_position = Vector3f(float(hi[size_t(0)].as_double()), float(hi[size_t(1)].as_double()), float(hi[size_t(2)].as_double()));
    } else {    // required by the specification; computed when a writer left them out
        out.min_position = Vector3f(1e30f); out.max_position = Vector3f(-1e30f);
        for (size_t v = 0; v < out.positions.count; ++v) {
            float p[3]; std::memcpy(p, out.positions.at(v), sizeof(p));
            out.min_position = Vector3f(std::fmin(out.min_position.x, p[0]), std::fmin(out.min_position.y, p[1]), std::fmin(out.min_position.z, p[2]));
            out.max_position = Vector3f(std::fmax(out.max_position.x, p[0]), std::fmax(out.max_position.y, p[1]), std::fmax(out.max_position.z, p[2]));
        }
    }
    const size_t vertex_count = out.positions.count;

    if ((out.has_normals = attributes.has("NORMAL"))) {
        if (!doc.resolve(attributes["NORMAL"].as_int(-1), out.normals, error)) return false;
        if (out.normals.components != 3 || out.normals.component_type != FLOAT || out.normals.count != vertex_count) { error = "NORMAL must be float VEC3, one per vertex"; return false; }
    }
    if ((out.has_texcoords = attributes.has("TEXCOORD_0"))) {
        if (!doc.resolve(attributes["TEXCOORD_0"].as_int(-1), out.texcoords, error)) return false;
        if (out.texcoords.components != 2 || out.texcoords.component_type != FLOAT || out.texcoords.count != vertex_count) { error = "TEXCOORD_0 must be float VEC2, one per vertex"; return false; }
    }
    if ((out.has_colors = attributes.has("COLOR_0"))) {
        if (!doc.resolve(attributes["COLOR_0"].as_int(-1), out.colors, error)) return false;
        const bool type_ok = out.colors.component_type == FLOAT || out.colors.component_type == UNSIGNED_BYTE || out.colors.component_type == UNSIGNED_SHORT;
        if ((out.colors.components != 3 && out.colors.components != 4) || !type_ok || out.colors.count != vertex_count) { error = "COLOR_0 must be VEC3 or VEC4 of float or normalised unsigned integers, one per vertex"; return false; }
    }
    if ((out.has_indices = primitive.has("indices"))) {
        if (!doc.resolve(primitive["indices"].as_int(-1), out.indices, error)) return false;
        const int type = out.indices.component_type;
        if (out.indices.components != 1 || (type != UNSIGNED_INT && type != UNSIGNED_SHORT && type != UNSIGNED_BYTE)) { error = "indices must be unsigned SCALARs"; return false; }
        auto index_at = [&](size_t i) -> unsigned { if (type == UNSIGNED_INT) { unsigned v; std::memcpy(&v, out.indices.at(i), 4); return v; }
                                                    if (type == UNSIGNED_SHORT) { unsigned short v; std::memcpy(&v, out.indices.at(i), 2); return v; } return *out.indices.at(i); };
        for (size_t i = 0; i < out.indices.count / 3 * 3; ++i)
            if (index_at(i) >= vertex_count) { error = "index past the end of the vertices"; return false; }
    }
    return true;
}

} // namespace

// ------------------------------------------------------------------------------------------------
// Loads a glTF file (glTFLoader.cpp:368-731).
// ------------------------------------------------------------------------------------------------
SceneNode load(const std::string& filename, ImageLoader image_loader) {
    if (!string_ends_with(filename, "glb") && !string_ends_with(filename, "gltf")) {
        printf("glTFLoader::load error: '%s' not a glTF file\n", filename.c_str());
        return SceneNode::invalid();
    }
    if (!image_loader) image_loader = ::ImageLoader::load_from_memory;

    Document doc;
    if (!open_document(filename, doc)) {
        printf("glTFLoader::load error: Failed to parse '%s'\n", filename.c_str());
        return SceneNode::invalid();
    }
    const Json::Value& model = doc.json;

    // Resolve and check every primitive's accessors first, so that a malformed file fails before anything is created.
    std::vector<std::vector<PrimitiveSource>> sources(model["meshes"].size());
    for (size_t m = 0; m < model["meshes"].size(); ++m) {
        const Json::Value& primitives = model["meshes"][m]["primitives"];
        sources[m].resize(primitives.size());
        for (size_t p = 0; p < primitives.size(); ++p) {
            std::string error;
            if (!resolve_primitive(doc, primitives[p], sources[m][p], error)) {
                printf("glTFLoader::load error: %s[%zu]: %s\n", resource_name(model["meshes"][m], "unnamed_mesh_", m).c_str(), p, error.c_str());
                printf("glTFLoader::load error: Failed to parse '%s'\n", filename.c_str());
                return SceneNode::invalid();
            }
        }
    }

    // Decode the images.
    const Json::Value& glTF_images = model["images"];
    std::vector<Image> images(glTF_images.size());
    for (size_t i = 0; i < glTF_images.size(); ++i) {
        const std::string name = resource_name(glTF_images[i], "unnamed_image_", i);
        std::vector<uint8_t> file_bytes;
        const uint8_t* bytes = nullptr;
        size_t byte_count = 0;
        const std::string& uri = glTF_images[i]["uri"].as_string();
        if (!uri.empty()) {
            if (load_uri(uri, doc.directory, file_bytes)) { bytes = file_bytes.data(); byte_count = file_bytes.size(); }
        } else {
            const Json::Value& view = model["bufferViews"][size_t(glTF_images[i]["bufferView"].as_int(-1))];
            const int buffer = view["buffer"].as_int(-1);
            const size_t offset = size_t(view["byteOffset"].as_double()), length = size_t(view["byteLength"].as_double());
            if (buffer >= 0 && size_t(buffer) < doc.buffers.size() && offset + length <= doc.buffers[size_t(buffer)].size()) { bytes = doc.buffers[size_t(buffer)].data() + offset; byte_count = length; }
        }
        if (bytes && byte_count) images[i] = image_loader(name, bytes, byte_count);
        if (images[i].exists() && (images[i].get_width() < 1 || images[i].get_height() < 1)) { Images::destroy(images[i].get_ID()); images[i] = Image(); }
        if (!images[i].exists()) printf("glTFLoader::load warning: Failed to load image '%s'. Textures using it are ignored.\n", name.c_str());
    }

    // Import materials.
    ImageCache converted_images;
    std::vector<bool> image_is_used(images.size(), false);
    Importer importer{doc, {}, {}, {}, Material()};
    importer.materials.resize(model["materials"].size());

    struct TextureState {
        int glTF_image_index = -1;
        Image image;
        SamplerParams sampler;
    };
    auto parse_glTF_texture = [&](const Json::Value& texture_info) {
        TextureState state;
        if (!texture_info.is_object()) return state;
        const Json::Value& glTF_texture = model["textures"][size_t(texture_info["index"].as_int(-1))];
        if (!glTF_texture.is_object()) return state;
        state.glTF_image_index = glTF_texture["source"].as_int(-1);
        if (state.glTF_image_index >= 0 && size_t(state.glTF_image_index) < images.size()) state.image = images[size_t(state.glTF_image_index)];
        const Json::Value& sampler = model["samplers"][size_t(glTF_texture["sampler"].as_int(-1))];
        if (sampler.is_object()) state.sampler.parse(sampler);
        return state;
    };
    auto flag_image_as_used = [&](const TextureState& texture, Image converted_image) {
        if (texture.glTF_image_index >= 0 && texture.image.exists() && texture.image.get_ID() == converted_image.get_ID())
            image_is_used[size_t(texture.glTF_image_index)] = true;
    };

    for (size_t i = 0; i < importer.materials.size(); ++i) {
        const Json::Value& glTF_mat = model["materials"][i];
        const std::string name = resource_name(glTF_mat, "unnamed_material_", i);

        Materials::Data mat_data = {};
        mat_data.tint = RGB(1.0f);
        mat_data.specularity = 0.04f;   // Corresponds to an index of refraction of 1.5
        mat_data.roughness = 1.0f;
        mat_data.metallic = 0.0f;
        mat_data.coverage = 1.0f;

        if (glTF_mat["doubleSided"].as_bool(false))
            mat_data.flags |= MaterialFlag::ThinWalled;
        if (glTF_mat["alphaMode"].as_string() == "MASK") {
            mat_data.flags |= MaterialFlag::Cutout;
            mat_data.coverage = float(glTF_mat["alphaCutoff"].as_double(0.5));
        }

        const Json::Value& clearcoat = glTF_mat["extensions"]["KHR_materials_clearcoat"];
        if (clearcoat.is_object()) {
            mat_data.coat = float(clearcoat["clearcoatFactor"].as_double(0.0));
            mat_data.coat_roughness = float(clearcoat["clearcoatRoughnessFactor"].as_double(0.0));
        }

        const Json::Value& pbr = glTF_mat["pbrMetallicRoughness"];
        const Json::Value& tint = pbr["baseColorFactor"];
        if (tint.size() >= 3) mat_data.tint = RGB(float(tint[size_t(0)].as_double()), float(tint[size_t(1)].as_double()), float(tint[size_t(2)].as_double()));
        if (pbr.has("roughnessFactor")) mat_data.roughness = float(pbr["roughnessFactor"].as_double());
        if (pbr.has("metallicFactor")) mat_data.metallic = float(pbr["metallicFactor"].as_double());
        const TextureState tint_coverage_tex = parse_glTF_texture(pbr["baseColorTexture"]);
        const TextureState metallic_roughness_tex = parse_glTF_texture(pbr["metallicRoughnessTexture"]);

        // Convert images from the glTF channel layout to the Bifrost channel layout.
        Image metallic_image = extract_channel(metallic_roughness_tex.image, 2, ImageUsage::Metallic, name, converted_images);
        flag_image_as_used(metallic_roughness_tex, metallic_image);
        mat_data.metallic_texture_ID = metallic_roughness_tex.sampler.create_texture_2D(metallic_image);

        Image coverage_image = extract_channel(tint_coverage_tex.image, 3, ImageUsage::Coverage, name, converted_images);
        flag_image_as_used(tint_coverage_tex, coverage_image);
        mat_data.coverage_texture_ID = tint_coverage_tex.sampler.create_texture_2D(coverage_image);

        Image tint_roughness_image = extract_tint_roughness(tint_coverage_tex.image, metallic_roughness_tex.image, name, converted_images);
        flag_image_as_used(tint_coverage_tex, tint_roughness_image);
        flag_image_as_used(metallic_roughness_tex, tint_roughness_image);
        mat_data.tint_roughness_texture_ID = tint_coverage_tex.sampler.create_texture_2D(tint_roughness_image);

        importer.materials[i] = Material(name, mat_data);
    }

    // Delete images not used by the datamodel.
    for (size_t i = 0; i < images.size(); ++i)
        if (!image_is_used[i] && images[i].exists()) Images::destroy(images[i].get_ID());

    if (model["animations"].size() > 0) printf("GLTFLoader::load warning: Animations are not supported and will be ignored.\n");
    if (model["cameras"].size() > 0) printf("GLTFLoader::load warning: Cameras are not supported and will be ignored.\n");
    if (model["skins"].size() > 0) printf("GLTFLoader::load warning: Skins are not supported and will be ignored.\n");

    // Import meshes: one per TRIANGLES primitive.
    for (size_t m = 0; m < model["meshes"].size(); ++m) {
        const Json::Value& glTF_mesh = model["meshes"][m];
        const std::string glTF_mesh_name = resource_name(glTF_mesh, "unnamed_mesh_", m);
        importer.meshes_start_index.push_back(int(importer.meshes.size()));
        for (size_t p = 0; p < sources[m].size(); ++p) {
            const PrimitiveSource& source = sources[m][p];
            if (!source.supported) {
                printf("GLTFLoader::load warning: %s[%zu] primitive %u not supported.\n", glTF_mesh_name.c_str(), p, unsigned(glTF_mesh["primitives"][p]["mode"].as_int()));
                continue;
            }

            const unsigned vertex_count = unsigned(source.positions.count);
            MeshFlags mesh_flags = MeshFlag::Position;
            if (source.has_normals) mesh_flags |= MeshFlag::Normal;
            if (source.has_texcoords) mesh_flags |= MeshFlag::Texcoord;
            if (source.has_colors) mesh_flags |= MeshFlag::TintAndRoughness;
            const unsigned primitive_count = (source.has_indices ? unsigned(source.indices.count) : vertex_count) / 3;

            // Append the primitive index to the mesh name in case there's more than one primitive.
            const std::string mesh_name = sources[m].size() > 1 ? glTF_mesh_name + "_primitive_" + std::to_string(p) : glTF_mesh_name;
            Mesh mesh = Mesh(mesh_name, primitive_count, vertex_count, mesh_flags);

            unsigned int* primitive_indices = &mesh.get_primitives()->x;
            if (source.has_indices) {
                Elements whole_triangles = source.indices;
                whole_triangles.count = size_t(primitive_count) * 3;
                if (source.indices.component_type == UNSIGNED_INT) copy_indices<unsigned int>(primitive_indices, whole_triangles);
                else if (source.indices.component_type == UNSIGNED_SHORT) copy_indices<unsigned short>(primitive_indices, whole_triangles);
                else copy_indices<unsigned char>(primitive_indices, whole_triangles);
            } else    // no index buffer: consecutive vertices form the triangles
                for (unsigned i = 0; i < primitive_count * 3; ++i) primitive_indices[i] = i;

            for (unsigned v = 0; v < vertex_count; ++v) std::memcpy(&mesh.get_positions()[v], source.positions.at(v), sizeof(Vector3f));
            if (source.has_normals) for (unsigned v = 0; v < vertex_count; ++v) std::memcpy(&mesh.get_normals()[v], source.normals.at(v), sizeof(Vector3f));
            if (source.has_texcoords) for (unsigned v = 0; v < vertex_count; ++v) std::memcpy(&mesh.get_texcoords()[v], source.texcoords.at(v), sizeof(Vector2f));
            if (source.has_colors) {
                TintRoughness* tints = mesh.get_tint_and_roughness();
                for (unsigned v = 0; v < vertex_count; ++v) {
                    const uint8_t* src = source.colors.at(v);
                    unsigned char rgb[3];
                    for (int c = 0; c < 3; ++c) {
                        if (source.colors.component_type == FLOAT) { float f; std::memcpy(&f, src + 4 * c, 4); rgb[c] = to_unorm8(f); }
                        else if (source.colors.component_type == UNSIGNED_SHORT) { unsigned short s; std::memcpy(&s, src + 2 * c, 2); rgb[c] = (unsigned char)((unsigned(s) * 255u + 32767u) / 65535u); }
                        else rgb[c] = src[c];
                    }
                    tints[v] = TintRoughness{rgb[0], rgb[1], rgb[2], 255};
                }
            }

            { // Negate the mesh's X component as glTF uses a right-handed coordinate system and we use a left-handed,
              // and swap two corners of every triangle so that the winding survives the mirroring.
                Vector3f min_position = source.min_position, max_position = source.max_position;
                min_position.x = -min_position.x;
                max_position.x = -max_position.x;
                std::swap(min_position.x, max_position.x);

                for (unsigned v = 0; v < vertex_count; ++v) mesh.get_positions()[v].x = -mesh.get_positions()[v].x;
                if (mesh.get_normals() != nullptr)
                    for (unsigned v = 0; v < vertex_count; ++v) mesh.get_normals()[v].x = -mesh.get_normals()[v].x;
                for (unsigned t = 0; t < primitive_count; ++t) std::swap(mesh.get_primitives()[t].x, mesh.get_primitives()[t].y);

                mesh.set_bounds(AABB{min_position, max_position});
            }

            importer.meshes.push_back({mesh, false});
        }
    }
    // Finally the total number of meshes, so that begin and end indices are [index] and [index + 1].
    importer.meshes_start_index.push_back(int(importer.meshes.size()));

    if (model["extensions"]["KHR_lights_punctual"]["lights"].size() > 0 || model["extensions"]["KHR_lights_cmn"]["lights"].size() > 0)
        printf("GLTFLoader::load warning: KHR_lights_cmn not supported. Light sources will be ignored.\n");

    auto destroy_unused_meshes = [&]() {
        for (const LoadedMesh& loaded : importer.meshes)
            if (!loaded.is_used) Meshes::destroy(loaded.mesh.get_ID());
    };

    // Setup scene.
    if (model["scenes"].size() > 1)
        printf("GLTFLoader::load warning: Only one scene supported. The default scene will be imported.\n");

    const int default_scene = model["scene"].as_int(-1);
    if (default_scene < 0 || size_t(default_scene) >= model["scenes"].size()) {
        destroy_unused_meshes();     // no scene to load
        return SceneNode::invalid();
    }

    std::vector<size_t> roots;
    for (const Json::Value& node : model["scenes"][size_t(default_scene)]["nodes"].elements())
        if (node.as_int(-1) >= 0 && size_t(node.as_int()) < model["nodes"].size()) roots.push_back(size_t(node.as_int()));

    SceneNode root_node;
    if (roots.size() == 1)
        root_node = importer.import_node(roots[0], Affine::identity(), 0);
    else {
        // Several root nodes in the scene. Attach them to a single common root node.
        root_node = SceneNode("Scene root");
        for (size_t root : roots) {
            SceneNode child_node = importer.import_node(root, Affine::identity(), 0);
            child_node.set_parent(root_node);
        }
    }
    destroy_unused_meshes();
    return root_node;
}

bool file_supported(const std::string& filename) {
    return string_ends_with(filename, ".glb") || string_ends_with(filename, ".gltf");
}

} // namespace glTFLoader

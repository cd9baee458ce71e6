// PngImage.cpp -- see PngImage.h. PNG per the W3C Portable Network Graphics specification (second edition); zlib does the
// DEFLATE stream and the CRCs, everything else is here.
#include "PngImage.h"

#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstring>

using namespace Bifrost::Assets;

namespace PngImage {

namespace {

const uint8_t SIGNATURE[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};

inline uint32_t read_be32(const uint8_t* p) { return uint32_t(p[0]) << 24 | uint32_t(p[1]) << 16 | uint32_t(p[2]) << 8 | uint32_t(p[3]); }
inline void push_be32(std::vector<uint8_t>& out, uint32_t v) { out.push_back(uint8_t(v >> 24)); out.push_back(uint8_t(v >> 16)); out.push_back(uint8_t(v >> 8)); out.push_back(uint8_t(v)); }

inline int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

struct Decoded { unsigned width = 0, height = 0, channels = 0; std::vector<uint8_t> pixels; };

bool decode(const uint8_t* data, size_t size, Decoded& out, const char*& reason) {
    if (size < 8 || std::memcmp(data, SIGNATURE, 8) != 0) { reason = "not a PNG file"; return false; }

    unsigned bit_depth = 0, colour_type = 0;
    bool have_header = false, ended = false;
    std::vector<uint8_t> compressed, palette, palette_alpha;
    for (size_t at = 8; at + 12 <= size && !ended;) {
        const uint32_t length = read_be32(data + at);
        const uint8_t* type = data + at + 4;
        const uint8_t* body = data + at + 8;
        if (length > size - at - 12) { reason = "truncated chunk"; return false; }
        if (!std::memcmp(type, "IHDR", 4)) {
            if (length != 13) { reason = "bad IHDR"; return false; }
            out.width = read_be32(body); out.height = read_be32(body + 4);
            bit_depth = body[8]; colour_type = body[9];
            if (body[10] != 0 || body[11] != 0) { reason = "unknown compression or filter method"; return false; }
            if (body[12] != 0) { reason = "interlaced PNGs are not supported"; return false; }
            have_header = true;
        } else if (!std::memcmp(type, "PLTE", 4)) palette.assign(body, body + length);
        else if (!std::memcmp(type, "tRNS", 4)) palette_alpha.assign(body, body + length);
        else if (!std::memcmp(type, "IDAT", 4)) compressed.insert(compressed.end(), body, body + length);
        else if (!std::memcmp(type, "IEND", 4)) ended = true;
        at += size_t(length) + 12;
    }
    if (!have_header || out.width == 0 || out.height == 0 || out.width > (1u << 24) || out.height > (1u << 24)) { reason = "missing or bad header"; return false; }

    unsigned samples;   // per pixel, as stored
    switch (colour_type) {
    case 0: samples = 1; break; case 2: samples = 3; break; case 3: samples = 1; break; case 4: samples = 2; break; case 6: samples = 4; break;
    default: reason = "unknown colour type"; return false;
    }
    const bool depth_ok = (colour_type == 0 && (bit_depth == 1 || bit_depth == 2 || bit_depth == 4 || bit_depth == 8 || bit_depth == 16)) ||
                          (colour_type == 3 && (bit_depth == 1 || bit_depth == 2 || bit_depth == 4 || bit_depth == 8)) ||
                          ((colour_type == 2 || colour_type == 4 || colour_type == 6) && (bit_depth == 8 || bit_depth == 16));
    if (!depth_ok) { reason = "bit depth not allowed for the colour type"; return false; }
    if (colour_type == 3 && palette.size() < 3) { reason = "palette missing"; return false; }

    const size_t row_bytes = (size_t(out.width) * samples * bit_depth + 7) / 8;
    const size_t filter_stride = std::max<size_t>(1, samples * bit_depth / 8);
    std::vector<uint8_t> raw((row_bytes + 1) * out.height);
    uLongf raw_size = uLongf(raw.size());
    if (compressed.empty() || uncompress(raw.data(), &raw_size, compressed.data(), uLong(compressed.size())) != Z_OK || raw_size != raw.size()) {
        reason = "corrupt image data"; return false;
    }

    // Undo the scanline filters in place (PNG specification, section 9).
    std::vector<uint8_t> zero_row(row_bytes, 0);
    for (unsigned y = 0; y < out.height; ++y) {
        uint8_t* row = raw.data() + (row_bytes + 1) * y + 1;
        const uint8_t* up = y ? row - (row_bytes + 1) : zero_row.data();
        const unsigned filter = row[-1];
        for (size_t i = 0; i < row_bytes; ++i) {
            const int a = i >= filter_stride ? row[i - filter_stride] : 0, b = up[i], c = i >= filter_stride ? up[i - filter_stride] : 0;
            switch (filter) {
            case 0: break;
            case 1: row[i] = uint8_t(row[i] + a); break;
            case 2: row[i] = uint8_t(row[i] + b); break;
            case 3: row[i] = uint8_t(row[i] + ((a + b) >> 1)); break;
            case 4: row[i] = uint8_t(row[i] + paeth(a, b, c)); break;
            default: reason = "unknown scanline filter"; return false;
            }
        }
    }

    // Expand to 8-bit channels: palette -> RGB(A), packed grey -> 8 bit, 16 bit -> the high byte.
    const bool palette_has_alpha = colour_type == 3 && !palette_alpha.empty();
    out.channels = colour_type == 3 ? (palette_has_alpha ? 4 : 3) : samples;
    out.pixels.resize(size_t(out.width) * out.height * out.channels);
    for (unsigned y = 0; y < out.height; ++y) {
        const uint8_t* row = raw.data() + (row_bytes + 1) * y + 1;
        uint8_t* dst = out.pixels.data() + size_t(y) * out.width * out.channels;
        for (unsigned x = 0; x < out.width; ++x) {
            if (colour_type == 3 || (colour_type == 0 && bit_depth < 8)) {
                const unsigned bit = x * bit_depth, shift = 8 - bit_depth - (bit & 7);
                const unsigned v = bit_depth == 8 ? row[x] : (row[bit >> 3] >> shift) & ((1u << bit_depth) - 1);
                if (colour_type == 3) {
                    const bool known = 3 * v + 2 < palette.size();
                    dst[0] = known ? palette[3 * v] : 0; dst[1] = known ? palette[3 * v + 1] : 0; dst[2] = known ? palette[3 * v + 2] : 0;
                    if (palette_has_alpha) dst[3] = v < palette_alpha.size() ? palette_alpha[v] : 255;
                } else
                    dst[0] = uint8_t(v * 255u / ((1u << bit_depth) - 1));
            } else if (bit_depth == 8)
                std::memcpy(dst, row + size_t(x) * samples, samples);
            else
                for (unsigned s = 0; s < samples; ++s) dst[s] = row[(size_t(x) * samples + s) * 2];
            dst += out.channels;
        }
    }
    return true;
}

// StbImageLoader.cpp:32-96 convert_image: the pixel format per channel count, grey + alpha widened to RGBA.
Image to_image(const std::string& name, const Decoded& d, bool flip_rows) {
    const PixelFormat format = d.channels == 1 ? PixelFormat::Intensity8 : (d.channels == 3 ? PixelFormat::RGB24 : PixelFormat::RGBA32);
    const unsigned out_channels = d.channels == 2 ? 4 : d.channels;
    std::vector<uint8_t> pixels(size_t(d.width) * d.height * out_channels);
    for (unsigned y = 0; y < d.height; ++y) {
        const uint8_t* src = d.pixels.data() + size_t(flip_rows ? d.height - 1 - y : y) * d.width * d.channels;
        uint8_t* dst = pixels.data() + size_t(y) * d.width * out_channels;
        if (d.channels == 2)
            for (unsigned x = 0; x < d.width; ++x) { dst[4 * x] = dst[4 * x + 1] = dst[4 * x + 2] = src[2 * x]; dst[4 * x + 3] = src[2 * x + 1]; }
        else
            std::memcpy(dst, src, size_t(d.width) * d.channels);
    }
    return Image::create2D(name, format, true, d.width, d.height, pixels.data());
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

void push_chunk(std::vector<uint8_t>& out, const char* type, const uint8_t* body, size_t length) {
    push_be32(out, uint32_t(length));
    const size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), body, body + length);
    push_be32(out, uint32_t(crc32(0, out.data() + start, uInt(out.size() - start))));
}

inline float linear_to_sRGB(float v) { return v < 0.0031308f ? v * 12.92f : 1.055f * std::pow(v, 1.0f / 2.4f) - 0.055f; }   // BF/Math/Color.h:372-377
inline uint8_t to_unorm8(float v) { return uint8_t(std::fmin(std::fmax(v, 0.0f), 1.0f) * 255.0f + 0.5f); }

} // namespace

Image load(const std::string& path) {
    std::vector<uint8_t> bytes;
    if (!read_file(path, bytes)) { printf("PngImage::load(%s) error: 'could not read the file'\n", path.c_str()); return Image(); }
    Decoded decoded;
    const char* reason = "";
    if (!decode(bytes.data(), bytes.size(), decoded, reason)) { printf("PngImage::load(%s) error: '%s'\n", path.c_str(), reason); return Image(); }
    return to_image(path, decoded, true);
}

Image load_from_memory(const std::string& name, const void* data, size_t byte_count) {
    Decoded decoded;
    const char* reason = "";
    if (!decode(static_cast<const uint8_t*>(data), byte_count, decoded, reason)) { printf("PngImage::load(%s) error: '%s'\n", name.c_str(), reason); return Image(); }
    return to_image(name, decoded, false);
}

std::vector<uint8_t> encode(unsigned width, unsigned height, unsigned channels, const uint8_t* pixels, bool flip_rows) {
    std::vector<uint8_t> out(SIGNATURE, SIGNATURE + 8);
    if (width == 0 || height == 0 || channels < 1 || channels > 4) return {};

    std::vector<uint8_t> header;
    push_be32(header, width); push_be32(header, height);
    const uint8_t colour_types[5] = {0, 0, 4, 2, 6};
    header.push_back(8); header.push_back(colour_types[channels]); header.push_back(0); header.push_back(0); header.push_back(0);
    push_chunk(out, "IHDR", header.data(), header.size());

    // Scanlines with the Up filter (cheap, and good on the smooth gradients renders are made of), then one DEFLATE stream.
    const size_t row_bytes = size_t(width) * channels;
    std::vector<uint8_t> raw((row_bytes + 1) * height);
    for (unsigned y = 0; y < height; ++y) {
        const uint8_t* row = pixels + size_t(flip_rows ? height - 1 - y : y) * row_bytes;
        const uint8_t* up = y ? pixels + size_t(flip_rows ? height - y : y - 1) * row_bytes : nullptr;
        uint8_t* dst = raw.data() + (row_bytes + 1) * y;
        dst[0] = up ? 2 : 0;
        for (size_t i = 0; i < row_bytes; ++i) dst[1 + i] = up ? uint8_t(row[i] - up[i]) : row[i];
    }
    uLongf bound = compressBound(uLong(raw.size()));
    std::vector<uint8_t> compressed(bound);
    if (compress2(compressed.data(), &bound, raw.data(), uLong(raw.size()), 6) != Z_OK) return {};
    push_chunk(out, "IDAT", compressed.data(), bound);
    push_chunk(out, "IEND", nullptr, 0);
    return out;
}

bool write(const std::string& path, Image image) {
    if (!image.exists()) return false;
    const PixelFormat format = image.get_pixel_format();
    const unsigned channels = channel_count(format), width = image.get_width(), height = image.get_height();
    const bool is_float = format == PixelFormat::Intensity_Float || format == PixelFormat::RGB_Float || format == PixelFormat::RGBA_Float;
    const bool keep_bytes = !is_float && Images::is_sRGB(image.get_ID());

    std::vector<uint8_t> bytes(size_t(width) * height * channels);
    const void* pixels = Images::get_pixels(image.get_ID());
    for (size_t i = 0; i < bytes.size(); ++i) {
        if (keep_bytes) { bytes[i] = static_cast<const uint8_t*>(pixels)[i]; continue; }
        const float v = is_float ? static_cast<const float*>(pixels)[i] : static_cast<const uint8_t*>(pixels)[i] / 255.0f;
        const bool is_alpha = channels == 4 && (i & 3) == 3;
        bytes[i] = to_unorm8(is_alpha || format == PixelFormat::Alpha8 ? v : linear_to_sRGB(v));
    }
    const std::vector<uint8_t> file = encode(width, height, channels, bytes.data(), true);
    if (file.empty()) return false;
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = std::fwrite(file.data(), 1, file.size(), f) == file.size();
    std::fclose(f);
    return ok;
}

} // namespace PngImage

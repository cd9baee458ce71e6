// HdrImage.cpp -- see HdrImage.h.
#include "HdrImage.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>

using namespace Bifrost::Assets;

namespace HdrImage {

namespace {

struct Reader {
    const unsigned char* data; size_t size, pos = 0;
    bool at_end() const { return pos >= size; }
    size_t remaining() const { return size - pos; }
    int byte() { return pos < size ? data[pos++] : 0; }
    std::string line() {     // up to the next newline, without it
        std::string s;
        while (pos < size && data[pos] != '\n') s.push_back(char(data[pos++]));
        if (pos < size) ++pos;
        return s;
    }
};

inline void convert_pixel(const unsigned char rgbe[4], float* out) {
    if (rgbe[3] == 0) { out[0] = out[1] = out[2] = 0.0f; return; }
    const float scale = float(std::ldexp(1.0f, int(rgbe[3]) - (128 + 8)));
    out[0] = rgbe[0] * scale; out[1] = rgbe[1] * scale; out[2] = rgbe[2] * scale;
}

} // namespace

bool is_hdr(const void* data, size_t byte_count) {
    const char* b = static_cast<const char*>(data);
    return (byte_count >= 11 && !std::memcmp(b, "#?RADIANCE\n", 11)) || (byte_count >= 7 && !std::memcmp(b, "#?RGBE\n", 7));
}

bool decode(const void* data, size_t byte_count, unsigned& width, unsigned& height, std::vector<float>& rgb, std::string* error) {
    auto fail = [&](const char* message) { if (error) *error = message; return false; };
    if (!is_hdr(data, byte_count)) return fail("not a Radiance HDR file");
    Reader r = {static_cast<const unsigned char*>(data), byte_count};
    r.line();
    bool format_ok = false;
    for (;;) {      // header lines until the empty one
        if (r.at_end()) return fail("truncated header");
        const std::string l = r.line();
        if (l.empty()) break;
        if (l == "FORMAT=32-bit_rle_rgbe") format_ok = true;
    }
    if (!format_ok) return fail("unsupported format (only 32-bit_rle_rgbe)");
    const std::string resolution = r.line();
    int h = 0, w = 0;
    if (std::sscanf(resolution.c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0) return fail("unsupported data layout (only -Y h +X w)");
    // Nothing is allocated from header figures alone: at most 1 << 24 pixels a side (the bound of the decoder the reference loads textures with),
    // and even a run-length encoded file spends at least 4 bytes a scanline and, flat, 4 bytes a pixel.
    if (w > (1 << 24) || h > (1 << 24)) return fail("image too large");
    const size_t remaining = r.remaining();
    if (size_t(h) * 4 > remaining || ((w < 8 || w >= 32768) && size_t(w) * size_t(h) * 4 > remaining)) return fail("truncated pixel data");
    if (size_t(w) * size_t(h) > (size_t(1) << 28)) return fail("image too large");
    width = unsigned(w); height = unsigned(h);
    rgb.assign(size_t(w) * h * 3, 0.0f);
    std::vector<unsigned char> scanline(size_t(w) * 4);
    if (w < 8 || w >= 32768) {      // such files are never run-length encoded: flat RGBE quadruples
        for (size_t i = 0; i < size_t(w) * h; ++i) {
            unsigned char rgbe[4] = {(unsigned char)r.byte(), (unsigned char)r.byte(), (unsigned char)r.byte(), (unsigned char)r.byte()};
            convert_pixel(rgbe, rgb.data() + 3 * i);
        }
        return true;
    }
    for (int j = 0; j < h; ++j) {
        const int c1 = r.byte(), c2 = r.byte(), len_hi = r.byte();
        if (c1 != 2 || c2 != 2 || (len_hi & 0x80)) {
            // not run-length encoded: these three bytes and the next are the first pixel of a flat file; the rest follows flat
            if (j != 0) return fail("mixed flat and run-length encoded scanlines");
            unsigned char rgbe[4] = {(unsigned char)c1, (unsigned char)c2, (unsigned char)len_hi, (unsigned char)r.byte()};
            convert_pixel(rgbe, rgb.data());
            for (size_t i = 1; i < size_t(w) * h; ++i) {
                unsigned char p[4] = {(unsigned char)r.byte(), (unsigned char)r.byte(), (unsigned char)r.byte(), (unsigned char)r.byte()};
                convert_pixel(p, rgb.data() + 3 * i);
            }
            return true;
        }
        const int length = (len_hi << 8) | r.byte();
        if (length != w) return fail("corrupt scanline length");
        for (int channel = 0; channel < 4; ++channel) {      // the four byte planes of the scanline, each run-length encoded
            int i = 0;
            while (i < w) {
                int count = r.byte();
                if (count > 128) {      // a run
                    const unsigned char value = (unsigned char)r.byte();
                    count -= 128;
                    if (count == 0 || count > w - i) return fail("corrupt run");
                    for (int k = 0; k < count; ++k) scanline[size_t(i++) * 4 + channel] = value;
                } else {                // literal bytes
                    if (count == 0 || count > w - i) return fail("corrupt run");
                    for (int k = 0; k < count; ++k) scanline[size_t(i++) * 4 + channel] = (unsigned char)r.byte();
                }
            }
        }
        for (int i = 0; i < w; ++i) convert_pixel(scanline.data() + size_t(i) * 4, rgb.data() + 3 * (size_t(j) * w + i));
    }
    return true;
}

static Image to_image(const std::string& name, const void* data, size_t byte_count, bool flip_rows) {
    unsigned width = 0, height = 0;
    std::vector<float> rgb;
    std::string error;
    if (!decode(data, byte_count, width, height, rgb, &error)) {
        printf("HdrImage::load(%s) error: '%s'\n", name.c_str(), error.c_str());
        return Image();
    }
    Image image = Image::create2D(name, PixelFormat::RGB_Float, false, width, height);     // StbImageLoader.cpp:36-44: float images are linear, 3 channels -> RGB_Float
    float* out = image.get_pixels<float>();
    const size_t row = size_t(width) * 3;
    for (unsigned y = 0; y < height; ++y) std::memcpy(out + size_t(y) * row, rgb.data() + size_t(flip_rows ? height - 1 - y : y) * row, row * sizeof(float));
    return image;
}

Image load(const std::string& path) {
    std::ifstream file(path, std::ios::binary);
    if (!file) { printf("HdrImage::load(%s) error: 'could not read the file'\n", path.c_str()); return Image(); }
    std::vector<char> bytes((std::istreambuf_iterator<char>(file)), std::istreambuf_iterator<char>());
    return to_image(path, bytes.data(), bytes.size(), true);
}

Image load_from_memory(const std::string& name, const void* data, size_t byte_count) { return to_image(name, data, byte_count, false); }

} // namespace HdrImage

// TgaImage.cpp -- see TgaImage.h. Layout per the Truevision TGA 2.0 specification (18 byte header, image id, colour map, pixel data);
// the choices the specification leaves open follow the decoder the reference uses, so that both make the same pixels of a file:
//   * 15 and 16 bit colour: three 5 bit fields scaled as (v * 255) / 31, the attribute bit is not alpha;
//   * a 16 bit GREY image is grey + alpha, two bytes as stored;
//   * the colour map's "first entry index" field is a number of BYTES to skip before the map, and an index past the map reads entry 0;
//   * run-length packets may run across row ends; bit 5 of the descriptor alone decides the row order (bit 4 is ignored).
#include "TgaImage.h"

#include <cstdio>
#include <cstring>
#include <fstream>

using namespace Bifrost::Assets;

namespace TgaImage {

namespace {

struct Header {
    unsigned id_length, map_type, image_type, map_first, map_length, map_entry_bits, width, height, pixel_bits, descriptor;
};

struct Reader {
    const unsigned char* data; size_t size, pos = 0;
    unsigned byte() { return pos < size ? data[pos++] : 0u; }         // reads past the end yield zeros, like the reference's stream
    unsigned word() { const unsigned lo = byte(); return lo | (byte() << 8); }
    void skip(size_t n) { pos = n > size - (pos < size ? pos : size) ? size : pos + n; }
};

bool read_header(Reader& r, Header& h) {
    if (r.size < 18) return false;
    h.id_length = r.byte(); h.map_type = r.byte(); h.image_type = r.byte();
    h.map_first = r.word(); h.map_length = r.word(); h.map_entry_bits = r.byte();
    r.word(); r.word();                                              // x / y origin: unused
    h.width = r.word(); h.height = r.word(); h.pixel_bits = r.byte(); h.descriptor = r.byte();
    return true;
}

bool colour_bits(unsigned bits) { return bits == 8 || bits == 15 || bits == 16 || bits == 24 || bits == 32; }

bool plausible(const Header& h) {
    if (h.map_type > 1) return false;
    if (h.map_type == 1) {
        if (h.image_type != 1 && h.image_type != 9) return false;
        if (!colour_bits(h.map_entry_bits)) return false;
        if (h.pixel_bits != 8 && h.pixel_bits != 16) return false;      // the size of an index
    } else if (h.image_type != 2 && h.image_type != 3 && h.image_type != 10 && h.image_type != 11) return false;
    return h.width >= 1 && h.height >= 1 && colour_bits(h.pixel_bits);
}

// Channels of a pixel (or colour map entry) of `bits` bits; five_bit: 15 / 16 bit colour that is expanded to RGB on reading.
unsigned channels_of(unsigned bits, bool grey, bool& five_bit) {
    five_bit = false;
    switch (bits) {
    case 8: return 1;
    case 16: if (grey) return 2; [[fallthrough]];
    case 15: five_bit = true; return 3;
    case 24: return 3;
    case 32: return 4;
    default: return 0;
    }
}

void read_five_bit_colour(Reader& r, uint8_t* out) {
    const unsigned v = r.word();
    out[0] = uint8_t((((v >> 10) & 31u) * 255u) / 31u);
    out[1] = uint8_t((((v >> 5) & 31u) * 255u) / 31u);
    out[2] = uint8_t(((v & 31u) * 255u) / 31u);
}

} // namespace

bool is_tga(const void* data, size_t byte_count) {
    Reader r = {static_cast<const unsigned char*>(data), byte_count};
    Header h;
    return read_header(r, h) && plausible(h);
}

bool decode(const void* data, size_t byte_count, unsigned& width, unsigned& height, unsigned& channels, std::vector<uint8_t>& pixels, std::string* error) {
    auto fail = [&](const char* message) { if (error) *error = message; return false; };
    Reader r = {static_cast<const unsigned char*>(data), byte_count};
    Header h;
    if (!read_header(r, h) || !plausible(h)) return fail("not a TGA file this decoder reads");
    const bool indexed = h.map_type == 1, run_length = h.image_type >= 8, grey = (h.image_type & 7u) == 3;
    bool five_bit = false;
    channels = indexed ? channels_of(h.map_entry_bits, false, five_bit) : channels_of(h.pixel_bits, grey, five_bit);
    if (channels == 0) return fail("unsupported TGA pixel format");
    width = h.width; height = h.height;
    const size_t pixel_count = size_t(width) * height;
    pixels.assign(pixel_count * channels, 0);
    r.skip(h.id_length);

    std::vector<uint8_t> map;
    if (indexed) {
        if (h.map_length == 0) return fail("TGA colour map without entries");
        r.skip(h.map_first);
        map.assign(size_t(h.map_length) * channels, 0);
        if (five_bit)
            for (unsigned i = 0; i < h.map_length; ++i) read_five_bit_colour(r, &map[size_t(i) * 3]);
        else {
            if (r.size - r.pos < map.size()) return fail("truncated TGA colour map");
            std::memcpy(map.data(), r.data + r.pos, map.size());
            r.pos += map.size();
        }
    }

    // pixels in file order; run-length packets: a count byte, bit 7 = one pixel repeated (count & 127) + 1 times, else that many literal pixels
    uint8_t current[4] = {0, 0, 0, 0};
    unsigned left_in_packet = 0;
    bool repeating = false;
    for (size_t i = 0; i < pixel_count; ++i) {
        bool read_pixel = true;
        if (run_length) {
            if (left_in_packet == 0) {
                const unsigned command = r.byte();
                left_in_packet = 1u + (command & 127u);
                repeating = (command & 128u) != 0;
            } else if (repeating) read_pixel = false;
            --left_in_packet;
        }
        if (read_pixel) {
            if (indexed) {
                size_t index = h.pixel_bits == 8 ? r.byte() : r.word();
                if (index >= h.map_length) index = 0;
                std::memcpy(current, &map[index * channels], channels);
            } else if (five_bit) read_five_bit_colour(r, current);
            else
                for (unsigned c = 0; c < channels; ++c) current[c] = uint8_t(r.byte());
        }
        std::memcpy(&pixels[i * channels], current, channels);
    }

    // stored blue first: to RGB(A); five-bit colours were expanded in that order already
    if (channels >= 3 && !five_bit)
        for (size_t i = 0; i < pixel_count; ++i) std::swap(pixels[i * channels], pixels[i * channels + 2]);
    // bottom row first unless bit 5 of the descriptor says top row first
    if (!(h.descriptor & 0x20u)) {
        const size_t row = size_t(width) * channels;
        std::vector<uint8_t> swap_row(row);
        for (unsigned y = 0; y * 2 + 1 < height; ++y) {
            uint8_t *a = &pixels[size_t(y) * row], *b = &pixels[size_t(height - 1 - y) * row];
            std::memcpy(swap_row.data(), a, row); std::memcpy(a, b, row); std::memcpy(b, swap_row.data(), row);
        }
    }
    return true;
}

static Image to_image(const std::string& name, const void* data, size_t byte_count, bool flip_rows) {
    unsigned width = 0, height = 0, channels = 0;
    std::vector<uint8_t> pixels;
    std::string error;
    if (!decode(data, byte_count, width, height, channels, pixels, &error)) {
        printf("TgaImage::load(%s) error: '%s'\n", name.c_str(), error.c_str());
        return Image();
    }
    // StbImageLoader.cpp:26-47, 82-92: 1 channel -> Intensity8, 3 -> RGB24, 4 -> RGBA32, grey + alpha expanded to RGBA32
    const PixelFormat format = channels == 1 ? PixelFormat::Intensity8 : (channels == 3 ? PixelFormat::RGB24 : PixelFormat::RGBA32);
    Image image = Image::create2D(name, format, true, width, height);
    uint8_t* out = image.get_pixels<uint8_t>();
    const unsigned out_channels = channels == 2 ? 4 : channels;
    for (unsigned y = 0; y < height; ++y) {
        const uint8_t* source = pixels.data() + size_t(flip_rows ? height - 1 - y : y) * width * channels;
        uint8_t* target = out + size_t(y) * width * out_channels;
        if (channels == 2)
            for (unsigned x = 0; x < width; ++x) { target[4 * x] = target[4 * x + 1] = target[4 * x + 2] = source[2 * x]; target[4 * x + 3] = source[2 * x + 1]; }
        else std::memcpy(target, source, size_t(width) * channels);
    }
    return image;
}

Image load(const std::string& path) {
    std::ifstream file(path, std::ios::binary);
    if (!file) { printf("TgaImage::load(%s) error: 'could not read the file'\n", path.c_str()); return Image(); }
    std::vector<char> bytes((std::istreambuf_iterator<char>(file)), std::istreambuf_iterator<char>());
    return to_image(path, bytes.data(), bytes.size(), true);
}

Image load_from_memory(const std::string& name, const void* data, size_t byte_count) { return to_image(name, data, byte_count, false); }

} // namespace TgaImage

// ImageLoader.cpp -- see ImageLoader.h.
#include "ImageLoader.h"

#include "HdrImage.h"
#include "JpegImage.h"
#include "PngImage.h"
#include "TgaImage.h"

#include <cstring>

#include <cstdio>
#include <fstream>
#include <vector>

namespace ImageLoader {

static bool is_png(const void* data, size_t byte_count) { return byte_count >= 8 && std::memcmp(data, "\x89PNG\r\n\x1a\n", 8) == 0; }

Bifrost::Assets::Image load(const std::string& path) {
    std::ifstream file(path, std::ios::binary);
    unsigned char signature[18] = {};   // a TGA header is 18 bytes
    if (file) file.read(reinterpret_cast<char*>(signature), sizeof(signature));
    const size_t got = file ? sizeof(signature) : size_t(file.gcount());
    if (is_png(signature, got)) return PngImage::load(path);
    if (JpegImage::is_jpeg(signature, got)) return JpegImage::load(path);
    if (HdrImage::is_hdr(signature, got)) return HdrImage::load(path);
    if (TgaImage::is_tga(signature, got)) return TgaImage::load(path);      // no signature: tried last, as the reference's decoder does
    return PngImage::load(path);      // reports unreadable files and unknown formats
}

Bifrost::Assets::Image load_from_memory(const std::string& name, const void* data, size_t byte_count) {
    if (is_png(data, byte_count)) return PngImage::load_from_memory(name, data, byte_count);
    if (JpegImage::is_jpeg(data, byte_count)) return JpegImage::load_from_memory(name, data, byte_count);
    if (HdrImage::is_hdr(data, byte_count)) return HdrImage::load_from_memory(name, data, byte_count);
    if (TgaImage::is_tga(data, byte_count)) return TgaImage::load_from_memory(name, data, byte_count);
    return PngImage::load_from_memory(name, data, byte_count);
}

} // namespace ImageLoader

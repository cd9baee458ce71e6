// ImageLoader.cpp -- see ImageLoader.h.
#include "ImageLoader.h"

#include "HdrImage.h"
#include "JpegImage.h"
#include "PngImage.h"

#include <cstdio>
#include <fstream>
#include <vector>

namespace ImageLoader {

Bifrost::Assets::Image load(const std::string& path) {
    std::ifstream file(path, std::ios::binary);
    unsigned char signature[16] = {};
    if (file) file.read(reinterpret_cast<char*>(signature), sizeof(signature));
    const size_t got = file ? sizeof(signature) : size_t(file.gcount());
    if (JpegImage::is_jpeg(signature, got)) return JpegImage::load(path);
    if (HdrImage::is_hdr(signature, got)) return HdrImage::load(path);
    return PngImage::load(path);      // reports unreadable files and unknown formats
}

Bifrost::Assets::Image load_from_memory(const std::string& name, const void* data, size_t byte_count) {
    if (JpegImage::is_jpeg(data, byte_count)) return JpegImage::load_from_memory(name, data, byte_count);
    if (HdrImage::is_hdr(data, byte_count)) return HdrImage::load_from_memory(name, data, byte_count);
    return PngImage::load_from_memory(name, data, byte_count);
}

} // namespace ImageLoader

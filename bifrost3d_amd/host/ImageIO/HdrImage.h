// HdrImage.h -- Radiance RGBE (.hdr) decoding: the format SimpleViewer's --environment-map files come in
// (apps/SimpleViewer/main.cpp:333,533; the reference reads them through stbi_loadf, StbImageLoader.cpp:99-113).
#pragma once

#include "../Bifrost.h"

#include <string>
#include <vector>

namespace HdrImage {

bool is_hdr(const void* data, size_t byte_count);
// RGB float pixels, rows top-down as stored. Flat and run-length encoded scanlines; "-Y h +X w" orientation only (what the
// reference's decoder accepts). The conversion is mantissa * 2^(exponent - 136), exponent 0 = black.
bool decode(const void* data, size_t byte_count, unsigned& width, unsigned& height, std::vector<float>& rgb, std::string* error = nullptr);

Bifrost::Assets::Image load(const std::string& path);                                                     // RGB_Float, bottom row first
Bifrost::Assets::Image load_from_memory(const std::string& name, const void* data, size_t byte_count);    // rows as stored

} // namespace HdrImage

// JpegImage.h -- JPEG decoding for the texture loaders (the reference loads .jpg textures through stb_image in
// extensions/StbImageLoader; real Sponza-class assets ship JPEG textures). See JpegImage.cpp for what is supported.
#pragma once

#include "../Bifrost.h"

#include <cstdint>
#include <string>
#include <vector>

namespace JpegImage {

bool is_jpeg(const void* data, size_t byte_count);
// 8 bit pixels, rows top-down as stored, 1 (grey) or 3 (RGB) channels. False (and a reason) when the stream is not a JPEG this decoder supports.
bool decode(const void* data, size_t byte_count, unsigned& width, unsigned& height, unsigned& channels, std::vector<uint8_t>& pixels, std::string* error = nullptr);

Bifrost::Assets::Image load(const std::string& path);                                                     // bottom row first (StbImageLoader::load)
Bifrost::Assets::Image load_from_memory(const std::string& name, const void* data, size_t byte_count);    // rows as stored

} // namespace JpegImage

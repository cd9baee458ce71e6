// ImageLoader.h -- one entry point for every image format the loaders read, by file signature: PNG (PngImage), JPEG (JpegImage),
// Radiance HDR (HdrImage) and, by a plausible header, TGA (TgaImage). Takes the place of StbImageLoader::load / load_from_memory (extensions/StbImageLoader/
// StbImageLoader/StbImageLoader.cpp:99-124) as the ImageLoader callback of the OBJ and glTF loaders and for --environment-map.
#pragma once

#include "../Bifrost.h"

#include <string>

namespace ImageLoader {

Bifrost::Assets::Image load(const std::string& path);                                                     // bottom row first
Bifrost::Assets::Image load_from_memory(const std::string& name, const void* data, size_t byte_count);    // rows as stored

} // namespace ImageLoader

// TgaImage.h -- Truevision TGA decoding: the texture format of the Crytek Sponza distribution and of many OBJ / MTL assets. The reference
// reads it through stb_image (extensions/StbImageLoader/StbImageLoader/StbImageLoader.cpp:99-124); this decoder accepts the same files and
// produces the same pixels: image types 1 / 2 / 3 and their run-length encoded forms 9 / 10 / 11; 8 bit grey, 16 bit grey + alpha, 15 / 16 bit
// colour (5 bits per channel, alpha bit ignored), 24 / 32 bit colour, colour maps with 8 or 16 bit indices; either row order.
#pragma once

#include "../Bifrost.h"

#include <cstdint>
#include <string>
#include <vector>

namespace TgaImage {

// TGA files carry no signature: a header whose fields are consistent with one of the supported layouts counts (checked after every
// format that has a signature, as the reference's decoder does).
bool is_tga(const void* data, size_t byte_count);
// Pixels top row first, `channels` interleaved bytes per pixel: 1 grey, 2 grey + alpha, 3 RGB, 4 RGBA.
bool decode(const void* data, size_t byte_count, unsigned& width, unsigned& height, unsigned& channels, std::vector<uint8_t>& pixels, std::string* error = nullptr);

Bifrost::Assets::Image load(const std::string& path);                                                     // bottom row first
Bifrost::Assets::Image load_from_memory(const std::string& name, const void* data, size_t byte_count);    // rows as stored

} // namespace TgaImage

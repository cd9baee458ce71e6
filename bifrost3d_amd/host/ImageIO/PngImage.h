// PngImage.h -- PNG decode / encode for the loaders and for screenshots.
//
// Fills the role of the reference's StbImageLoader / StbImageWriter extensions (extensions/StbImageLoader/StbImageLoader/
// StbImageLoader.cpp:24-124: load() flips the rows so that row 0 is the bottom of the picture, load_from_memory() does not;
// 8-bit images are sRGB; 1 channel -> Intensity8, 2 -> expanded to RGBA32, 3 -> RGB24, 4 -> RGBA32), restricted to the one
// format this image can decode without third-party code: PNG, through zlib's inflate. The PNG container, the scanline
// filters and the bit-depth / palette expansion are this repository's own; JPEG and Radiance HDR files are reported as
// unsupported (invalid image), which every caller already treats like a texture that failed to load.
#pragma once

#include "../Bifrost.h"

#include <cstdint>
#include <string>
#include <vector>

namespace PngImage {

Bifrost::Assets::Image load(const std::string& path);                                                     // bottom row first
Bifrost::Assets::Image load_from_memory(const std::string& name, const void* data, size_t byte_count);    // rows as stored

// Encodes 8-bit pixels with 1-4 channels. `flip_rows` writes the last row first (images kept bottom-up become top-down files).
std::vector<uint8_t> encode(unsigned width, unsigned height, unsigned channels, const uint8_t* pixels, bool flip_rows);

// Writes an image of any 8-bit or float format as an 8-bit PNG with the rows flipped so that the file is top-down
// (extensions/StbImageWriter/StbImageWriter/StbImageWriter.cpp:84-124: colours leave as sRGB, alpha stays linear; bytes of
// an image that already is sRGB are written as they are). False when the file cannot be written.
bool write(const std::string& path, Bifrost::Assets::Image image);

} // namespace PngImage

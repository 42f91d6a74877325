// Texture images for the scene converter: PNG, JPEG and TGA files -> RGBA8, the way the reference's loaders hand
// them to the renderer (src/driver/image.cpp:25-95 load_png, :185-240 load_jpg): 8 bits per channel, palette / gray
// expanded to RGB, 16-bit samples reduced to their high byte, alpha 255 where the file has none, ROWS FLIPPED (file
// row y is stored at height-1-y, image.cpp:85,226) and colour channels gamma-corrected with
// c -> (uint8)(pow(c/255, 2.2) * 255) (image.cpp:10-18).
//
// The reference links libpng and libjpeg; this image has no headers for them, so the decoders are written here:
// PNG on top of zlib's inflate (all colour types and bit depths, non-interlaced and Adam7), baseline / extended
// sequential AND progressive Huffman JPEG (8-bit, any sampling factors, any number of scans, restart markers; 2x1 / 2x2 chroma
// is upsampled with libjpeg's triangle filter and the IDCT is done in float, so texels can differ from libjpeg's by a level
// or two; arithmetic-coded, lossless and hierarchical files are rejected),
// TGA types 2 / 3 / 10 / 11 (the reference's converter names a `load_tga` that its runtime never had).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace rodent {

struct ImageRgba8 {
    int width = 0, height = 0;
    std::vector<uint8_t> pixels;      // 4 bytes per texel, row 0 = BOTTOM row of the file
};

bool load_png(const std::string& path, ImageRgba8& img, std::string* error = nullptr);
bool load_jpg(const std::string& path, ImageRgba8& img, std::string* error = nullptr);
bool load_tga(const std::string& path, ImageRgba8& img, std::string* error = nullptr);
// by file extension (.png .jpg .jpeg .tga), as converter.cpp:757-767 decides
bool load_image(const std::string& path, ImageRgba8& img, std::string* error = nullptr);

} // namespace rodent

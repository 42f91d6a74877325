// tex_dump image out.rgba -- decodes a texture image exactly as the scene converter does (host/image.h) and writes
// "<int32 width><int32 height><width*height*4 bytes RGBA, row 0 = bottom row>"; used by the decoder tests.
#include <cstdio>
#include <iostream>

#include "../image.h"

int main(int argc, char** argv) {
    if (argc != 3) { std::cerr << "usage: tex_dump image.(png|jpg|tga) out.rgba" << std::endl; return 1; }
    rodent::ImageRgba8 img; std::string err;
    if (!rodent::load_image(argv[1], img, &err)) { std::cerr << "Cannot load image '" << argv[1] << "': " << err << std::endl; return 1; }
    FILE* f = fopen(argv[2], "wb");
    if (!f) { std::cerr << "Cannot write '" << argv[2] << "'" << std::endl; return 1; }
    const int32_t dims[2] = {img.width, img.height};
    fwrite(dims, 4, 2, f); fwrite(img.pixels.data(), 1, img.pixels.size(), f);
    fclose(f);
    std::cout << img.width << " x " << img.height << std::endl;
    return 0;
}

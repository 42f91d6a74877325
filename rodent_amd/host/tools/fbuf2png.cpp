// fbuf2png -- float buffer (.fbuf) -> 8-bit grey RGBA PNG.  Same options and pixel
// mapping as the reference (tools/fbuf2png/fbuf2png.cpp:24-31 usage, :81 normalise by
// the maximum, :108-114 c = uint8(255 * value / tmax)), written with the in-tree PNG
// writer because libpng's headers are not in this image.
#include <algorithm>
#include <cstring>
#include <fstream>
#include <iostream>
#include <vector>

#include "../png_write.h"

int main(int argc, char** argv) {
    bool normalize = false;
    int width = 1024, height = 1024;
    std::vector<std::string> files;
    for (int i = 1; i < argc; i++) {
        const char* arg = argv[i];
        auto need = [&]() { if (i + 1 >= argc) { std::cerr << "Missing argument for " << arg << std::endl; exit(1); } return argv[++i]; };
        if (arg[0] != '-') { files.push_back(arg); continue; }
        if (!strcmp(arg, "-h") || !strcmp(arg, "--help")) {
            std::cout << "Usage: fbuf2png [options] input output\n"
                         "Available options:\n"
                         "  -sx      --width        Sets the width of the image (default: 1024)\n"
                         "  -sy      --height       Sets the height of the image (default: 1024)\n"
                         "  -n       --normalize    Normalizes the values contained in the image (disabled by default)\n";
            return 0;
        }
        else if (!strcmp(arg, "-n") || !strcmp(arg, "--normalize")) normalize = true;
        else if (!strcmp(arg, "-sx") || !strcmp(arg, "--width")) width = strtol(need(), nullptr, 10);
        else if (!strcmp(arg, "-sy") || !strcmp(arg, "--height")) height = strtol(need(), nullptr, 10);
        else { std::cerr << "Unknown option '" << arg << "'" << std::endl; return 1; }
    }
    if (files.size() < 2) { std::cerr << "Missing input or output file" << std::endl; return 1; }
    if (files.size() > 2) { std::cerr << "Too many arguments" << std::endl; return 1; }

    std::ifstream in(files[0], std::ifstream::binary);
    if (!in) return 1;
    std::vector<float> img((size_t)width * height);
    if (!in.read((char*)img.data(), img.size() * sizeof(float))) { std::cerr << "Not enough data in the float buffer" << std::endl;
        return 1; }
    const float tmax = normalize ? *std::max_element(img.begin(), img.end()) : 1.0f;
    std::vector<uint8_t> px(img.size() * 4);
    for (size_t i = 0; i < img.size(); i++) {
        const uint8_t c = (uint8_t)(255.0f * img[i] / tmax);
        px[4 * i] = px[4 * i + 1] = px[4 * i + 2] = c; px[4 * i + 3] = 255;
    }
    return rodent::write_png(files[1], px.data(), width, height, 4) ? 0 : 1;
}

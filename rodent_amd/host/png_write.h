// Minimal dependency-free PNG writer (8-bit grey / RGB / RGBA, stored deflate blocks).
// libpng's headers are not available in this image; the reference uses libpng
// (src/driver/image.cpp:185-238, tools/fbuf2png/fbuf2png.cpp:40-74).
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace rodent {

inline uint32_t png_crc(const uint8_t* p, size_t n, uint32_t crc = 0xFFFFFFFFu) {
    static uint32_t table[256]; static bool init = false;
    if (!init) { for (uint32_t i = 0; i < 256; i++) { uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[i] = c; } init = true; }
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    return crc;
}

// channels: 1 (grey), 3 (RGB) or 4 (RGBA); pixels are row-major, top row first.
inline bool write_png(const std::string& path, const uint8_t* pixels, int width, int height, int channels) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return false;
    auto be32 = [](uint8_t* d, uint32_t v) { d[0] = v >> 24; d[1] = v >> 16; d[2] = v >> 8; d[3] = v; };
    auto chunk = [&](const char* type, const std::vector<uint8_t>& data) {
        uint8_t len[4]; be32(len, (uint32_t)data.size()); fwrite(len, 1, 4, f);
        std::vector<uint8_t> buf(4 + data.size());
        for (int i = 0; i < 4; i++) buf[i] = (uint8_t)type[i];
        std::copy(data.begin(), data.end(), buf.begin() + 4);
        fwrite(buf.data(), 1, buf.size(), f);
        uint8_t crc[4]; be32(crc, png_crc(buf.data(), buf.size()) ^ 0xFFFFFFFFu); fwrite(crc, 1, 4, f);
    };
    const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    fwrite(sig, 1, 8, f);
    std::vector<uint8_t> ihdr(13);
    be32(&ihdr[0], (uint32_t)width); be32(&ihdr[4], (uint32_t)height);
    ihdr[8] = 8; ihdr[9] = channels == 1 ? 0 : (channels == 3 ? 2 : 6); ihdr[10] = ihdr[11] = ihdr[12] = 0;
    chunk("IHDR", ihdr);
    const size_t stride = (size_t)width * channels;
    std::vector<uint8_t> raw; raw.reserve((stride + 1) * height);
    for (int y = 0; y < height; y++) { raw.push_back(0); raw.insert(raw.end(), pixels + y * stride, pixels + (y + 1) * stride); }
    std::vector<uint8_t> z; z.reserve(raw.size() + raw.size() / 65535 * 5 + 16);
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;
    for (size_t pos = 0; pos < raw.size() || pos == 0; ) {
        const size_t n = std::min<size_t>(65535, raw.size() - pos);
        const bool last = pos + n >= raw.size();
        z.push_back(last ? 1 : 0);
        z.push_back(n & 0xFF); z.push_back(n >> 8); z.push_back(~n & 0xFF); z.push_back((~n >> 8) & 0xFF);
        for (size_t i = 0; i < n; i++) { a = (a + raw[pos + i]) % 65521; b = (b + a) % 65521; }
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        pos += n;
        if (last) break;
    }
    uint8_t ad[4]; be32(ad, (b << 16) | a); z.insert(z.end(), ad, ad + 4);
    chunk("IDAT", z);
    chunk("IEND", {});
    fclose(f);
    return true;
}

} // namespace rodent

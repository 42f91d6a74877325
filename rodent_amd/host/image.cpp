#include "image.h"

#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

namespace rodent {
namespace {

bool fail(std::string* error, const std::string& msg) { if (error) *error = msg; return false; }

bool read_file(const std::string& path, std::vector<uint8_t>& data) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    data.resize(size > 0 ? (size_t)size : 0);
    const bool ok = size >= 0 && fread(data.data(), 1, data.size(), f) == data.size();
    fclose(f);
    return ok;
}

// image.cpp:10-18 (the truncating float -> uint8 conversion included), then the vertical flip of image.cpp:85,226
void finish(ImageRgba8& img, const std::vector<uint8_t>& top_down) {
    uint8_t lut[256];
    for (int i = 0; i < 256; i++) lut[i] = (uint8_t)(std::pow(i * (1.0f / 255.0f), 2.2f) * 255.0f);
    img.pixels.resize(top_down.size());
    const size_t row = (size_t)img.width * 4;
    for (int y = 0; y < img.height; y++) {
        const uint8_t* src = top_down.data() + row * y;
        uint8_t* dst = img.pixels.data() + row * (img.height - 1 - y);
        for (int x = 0; x < img.width; x++) {
            dst[4 * x + 0] = lut[src[4 * x + 0]]; dst[4 * x + 1] = lut[src[4 * x + 1]]; dst[4 * x + 2] = lut[src[4 * x + 2]];
            dst[4 * x + 3] = src[4 * x + 3];
        }
    }
}

// ------------------------------------------------------------------------------------------- PNG
uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// Reverses the scanline filters of one (sub)image in place; `data` = rows of (1 filter byte + stride bytes).
bool unfilter(uint8_t* data, int rows, size_t stride, int bpp) {
    std::vector<uint8_t> zero(stride, 0);
    const uint8_t* prev = zero.data();
    for (int y = 0; y < rows; y++) {
        uint8_t* line = data + (stride + 1) * (size_t)y;
        const int type = line[0];
        uint8_t* cur = line + 1;
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0;
            int v = cur[i];
            switch (type) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: v += paeth(a, b, c); break;
                default: return false;
            }
            cur[i] = (uint8_t)v;
        }
        prev = cur;
    }
    return true;
}

struct PngInfo { int width, height, depth, ctype, interlace; std::vector<uint8_t> palette, trns; };

int png_channels(int ctype) { return ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : 4; }

// sample k of a row at the file's bit depth, reduced to 8 bits (16 -> high byte, image.cpp:46-47 png_set_strip_16)
int png_sample(const uint8_t* row, size_t k, int depth, bool scale) {
    if (depth == 8) return row[k];
    if (depth == 16) return row[2 * k];
    const int per = 8 / depth, shift = (per - 1 - (int)(k % per)) * depth, v = (row[k / per] >> shift) & ((1 << depth) - 1);
    return scale ? v * 255 / ((1 << depth) - 1) : v;
}

void png_convert_row(const PngInfo& info, const uint8_t* row, int count, uint8_t* out /* 4 * count */, int out_step_pixels) {
    const int ch = png_channels(info.ctype);
    for (int x = 0; x < count; x++) {
        uint8_t* px = out + 4 * (size_t)x * out_step_pixels;
        int r, g, b, a = 255;
        if (info.ctype == 3) {
            const int idx = png_sample(row, x, info.depth, false);
            r = 3 * idx + 2 < (int)info.palette.size() ? info.palette[3 * idx] : 0;
            g = 3 * idx + 2 < (int)info.palette.size() ? info.palette[3 * idx + 1] : 0;
            b = 3 * idx + 2 < (int)info.palette.size() ? info.palette[3 * idx + 2] : 0;
            if (idx < (int)info.trns.size()) a = info.trns[idx];
        } else if (info.ctype == 0 || info.ctype == 4) {
            r = g = b = png_sample(row, (size_t)x * ch, info.depth, true);
            if (info.ctype == 4) a = png_sample(row, (size_t)x * ch + 1, info.depth, true);
            else if (info.trns.size() >= 2) {
                const int key = (info.trns[0] << 8) | info.trns[1];
                const int raw = info.depth == 16 ? (row[2 * x] << 8) | row[2 * x + 1] : png_sample(row, x, info.depth, false);
                if (raw == key) a = 0;
            }
        } else {
            r = png_sample(row, (size_t)x * ch, info.depth, true); g = png_sample(row, (size_t)x * ch + 1, info.depth, true);
            b = png_sample(row, (size_t)x * ch + 2, info.depth, true);
            if (info.ctype == 6) a = png_sample(row, (size_t)x * ch + 3, info.depth, true);
            else if (info.trns.size() >= 6) {
                bool same = true;
                for (int c = 0; c < 3 && same; c++) {
                    const int key = (info.trns[2 * c] << 8) | info.trns[2 * c + 1];
                    const int raw = info.depth == 16
                        ? (row[2 * ((size_t)x * 3 + c)] << 8) | row[2 * ((size_t)x * 3 + c) + 1] : row[(size_t)x * 3 + c];
                    same = raw == key;
                }
                if (same) a = 0;
            }
        }
        px[0] = (uint8_t)r; px[1] = (uint8_t)g; px[2] = (uint8_t)b; px[3] = (uint8_t)a;
    }
}

// --------------------------------------------------------------------------------------------- JPEG
struct Huff { uint8_t bits[17] = {0}; uint8_t vals[256] = {0}; int mincode[17], maxcode[18], valptr[17]; bool defined = false; };

void build_huff(Huff& h) {
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
        h.valptr[l] = k; h.mincode[l] = code;
        code += h.bits[l]; k += h.bits[l];
        h.maxcode[l] = h.bits[l] ? code - 1 : -1;
        code <<= 1;
    }
    h.maxcode[17] = 0x7FFFFFFF;
    h.defined = true;
}

struct BitReader {
    const uint8_t* p; const uint8_t* end; uint32_t buf = 0; int cnt = 0; bool hit_marker = false;
    int bit() {
        if (cnt == 0) {
            int byte = 0;
            if (!hit_marker && p < end) {
                byte = *p++;
                if (byte == 0xFF) {
                    if (p < end && *p == 0x00) p++;                    // stuffed zero
                    else { hit_marker = true; p--; byte = 0; }         // a marker: feed zeros until the caller handles it
                }
            }
            buf = (uint32_t)byte; cnt = 8;
        }
        cnt--;
        return (buf >> cnt) & 1;
    }
    int bits(int n) { int v = 0; for (int i = 0; i < n; i++) v = (v << 1) | bit(); return v; }
    void restart() { cnt = 0; hit_marker = false; }
};

int decode_huff(BitReader& br, const Huff& h) {
    int code = 0;
    for (int l = 1; l <= 16; l++) {
        code = (code << 1) | br.bit();
        if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    return -1;
}

int extend(int v, int t) { return t == 0 ? 0 : (v < (1 << (t - 1)) ? v - (1 << t) + 1 : v); }

const int kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
    28,
                         35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54,
                             47, 55, 62, 63};

void idct8x8(const float* in, uint8_t* out, size_t stride) {
    static float c[8][8]; static bool init = false;
    if (!init) {
        for (int x = 0; x < 8; x++) for (int u = 0; u < 8; u++) c[x][u] = (u == 0 ? 0.35355339059f
            : 0.5f) * std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0);
        init = true;
    }
    float tmp[64];
    for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) { float s = 0; for (int u = 0; u < 8; u++) s += c[x][u] * in[8 * y + u];
        tmp[8 * y + x] = s; }
    for (int x = 0; x < 8; x++) for (int y = 0; y < 8; y++) {
        float s = 0; for (int v = 0; v < 8; v++) s += c[y][v] * tmp[8 * v + x];
        const int q = (int)std::lround(s + 128.0f);
        out[stride * y + x] = (uint8_t)std::min(255, std::max(0, q));
    }
}

struct JpegComp { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0; size_t stride = 0; std::vector<uint8_t> plane; };

} // namespace

bool load_png(const std::string& path, ImageRgba8& img, std::string* error) {
    std::vector<uint8_t> file;
    if (!read_file(path, file)) return fail(error, "cannot read '" + path + "'");
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (file.size() < 8 || memcmp(file.data(), sig, 8)) return fail(error, "not a PNG file");
    PngInfo info{}; bool have_ihdr = false;
    std::vector<uint8_t> idat;
    for (size_t pos = 8; pos + 12 <= file.size();) {
        const uint32_t len = be32(&file[pos]);
        if (pos + 12 + (size_t)len > file.size()) return fail(error, "truncated PNG chunk");
        const uint8_t* type = &file[pos + 4]; const uint8_t* data = &file[pos + 8];
        if (!memcmp(type, "IHDR", 4) && len >= 13) {
            info.width = (int)be32(data); info.height = (int)be32(data + 4); info.depth = data[8]; info.ctype = data[9];
            info.interlace = data[12];
            have_ihdr = true;
        } else if (!memcmp(type, "PLTE", 4)) info.palette.assign(data, data + len);
        else if (!memcmp(type, "tRNS", 4)) info.trns.assign(data, data + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!memcmp(type, "IEND", 4)) break;
        pos += 12 + (size_t)len;
    }
    const bool depth_ok = info.depth == 8 || info.depth == 16
        || ((info.ctype == 0 || info.ctype == 3) && (info.depth == 1 || info.depth == 2 || info.depth == 4));
    if (!have_ihdr || info.width <= 0 || info.height <= 0 || info.width > 65536 || info.height > 65536 || !depth_ok ||
        !(info.ctype == 0 || info.ctype == 2 || info.ctype == 3 || info.ctype == 4 || info.ctype == 6) || info.interlace > 1 ||
        (info.ctype == 3 && info.depth == 16))
        return fail(error, "unsupported PNG header");
    const int bits = png_channels(info.ctype) * info.depth, bpp = std::max(1, bits / 8);
    auto stride_of = [&](int w) { return ((size_t)w * bits + 7) / 8; };
    static const int xs[7] = {0, 4, 0, 2, 0, 1, 0}, ys[7] = {0, 0, 4, 0, 2, 0, 1}, dx[7] = {8, 8, 4, 4, 2, 2, 1}, dy[7] = {8, 8, 8, 4, 4,
        2, 2};
    size_t raw_size = 0;
    if (!info.interlace) raw_size = (stride_of(info.width) + 1) * (size_t)info.height;
    else for (int p = 0; p < 7; p++) {
        const int w = (info.width - xs[p] + dx[p] - 1) / dx[p], h = (info.height - ys[p] + dy[p] - 1) / dy[p];
        if (w > 0 && h > 0) raw_size += (stride_of(w) + 1) * (size_t)h;
    }
    std::vector<uint8_t> raw(raw_size);
    uLongf got = (uLongf)raw_size;
    if (uncompress(raw.data(), &got, idat.data(), (uLong)idat.size()) != Z_OK || got != raw_size) return fail(error,
        "PNG data does not inflate to the image size");
    img.width = info.width; img.height = info.height;
    std::vector<uint8_t> rgba((size_t)info.width * info.height * 4);
    if (!info.interlace) {
        const size_t stride = stride_of(info.width);
        if (!unfilter(raw.data(), info.height, stride, bpp)) return fail(error, "bad PNG filter type");
        for (int y = 0; y < info.height; y++) png_convert_row(info, raw.data() + (stride + 1) * (size_t)y + 1, info.width,
            &rgba[(size_t)y * info.width * 4], 1);
    } else {
        size_t off = 0;
        for (int p = 0; p < 7; p++) {
            const int w = (info.width - xs[p] + dx[p] - 1) / dx[p], h = (info.height - ys[p] + dy[p] - 1) / dy[p];
            if (w <= 0 || h <= 0) continue;
            const size_t stride = stride_of(w);
            if (!unfilter(raw.data() + off, h, stride, bpp)) return fail(error, "bad PNG filter type");
            for (int y = 0; y < h; y++)
                png_convert_row(info, raw.data() + off + (stride + 1) * (size_t)y + 1, w,
                    &rgba[((size_t)(ys[p] + y * dy[p]) * info.width + xs[p]) * 4], dx[p]);
            off += (stride + 1) * (size_t)h;
        }
    }
    finish(img, rgba);
    return true;
}

bool load_jpg(const std::string& path, ImageRgba8& img, std::string* error) {
    std::vector<uint8_t> file;
    if (!read_file(path, file)) return fail(error, "cannot read '" + path + "'");
    if (file.size() < 4 || file[0] != 0xFF || file[1] != 0xD8) return fail(error, "not a JPEG file");
    // Baseline / extended sequential (SOF0 / SOF1) and PROGRESSIVE (SOF2) Huffman JPEG, any number of scans: every scan
    // decodes into per-component coefficient arrays (ITU T.81 F.2 / G.2: spectral selection Ss..Se, successive approximation
    // Ah / Al, end-of-band runs, refinement passes); dequantisation and the inverse DCT follow the last scan.  The reference
    // reads its textures with libjpeg (src/driver/image.cpp:185-238), which decodes all of these.
    uint16_t qt[4][64] = {}; Huff dc[4], ac[4];
    std::vector<JpegComp> comps;
    // per component: blocks in raster order over the padded grid x 64, natural order
    std::vector<std::vector<int16_t>> coef;
    int width = 0, height = 0, restart_interval = 0, adobe_transform = -1, hmax = 1, vmax = 1, mcux = 0, mcuy = 0;
    bool progressive = false, any_scan = false;
    size_t pos = 2;
    while (pos + 4 <= file.size()) {
        if (file[pos] != 0xFF) { pos++; continue; }
        const int marker = file[pos + 1];
        if (marker == 0xFF) { pos++; continue; }
        if (marker == 0x00 || marker == 0xD8 || (marker >= 0xD0 && marker <= 0xD7) || marker == 0x01) { pos += 2; continue; }
        if (marker == 0xD9) break;
        const size_t len = ((size_t)file[pos + 2] << 8) | file[pos + 3];
        if (len < 2 || pos + 2 + len > file.size()) return fail(error, "truncated JPEG segment");
        const uint8_t* d = &file[pos + 4]; const size_t n = len - 2;
        if (marker == 0xDB) {                                            // DQT
            for (size_t i = 0; i < n;) {
                const int pq = d[i] >> 4, tq = d[i] & 15; i++;
                if (tq > 3 || i + (pq ? 128 : 64) > n) return fail(error, "bad JPEG quantisation table");
                for (int k = 0; k < 64; k++) { qt[tq][k] = pq ? (uint16_t)((d[i] << 8) | d[i + 1]) : d[i]; i += pq ? 2 : 1; }
            }
        } else if (marker == 0xC4) {                                     // DHT
            for (size_t i = 0; i + 17 <= n;) {
                const int tc = d[i] >> 4, th = d[i] & 15; i++;
                if (th > 3 || tc > 1) return fail(error, "bad JPEG Huffman table");
                Huff& h = tc ? ac[th] : dc[th];
                int total = 0;
                for (int l = 1; l <= 16; l++) { h.bits[l] = d[i + l - 1]; total += h.bits[l]; }
                i += 16;
                if (total > 256 || i + total > n) return fail(error, "bad JPEG Huffman table");
                memcpy(h.vals, d + i, total); i += total;
                build_huff(h);
            }
        } else if (marker == 0xC0 || marker == 0xC1 || marker == 0xC2) { // SOF0 / SOF1: sequential Huffman; SOF2: progressive Huffman
            if (!comps.empty()) return fail(error, "JPEG with more than one frame");
            if (n < 6 || d[0] != 8) return fail(error, "only 8-bit JPEG is supported");
            progressive = marker == 0xC2;
            height = (d[1] << 8) | d[2]; width = (d[3] << 8) | d[4];
            const int nc = d[5];
            if ((nc != 1 && nc != 3) || n < 6 + 3 * (size_t)nc || width <= 0 || height <= 0) return fail(error,
                "unsupported JPEG component count");
            comps.resize(nc);
            for (int c = 0; c < nc; c++) { comps[c].id = d[6 + 3 * c]; comps[c].h = d[7 + 3 * c] >> 4; comps[c].v = d[7 + 3 * c] & 15;
                comps[c].tq = d[8 + 3 * c] & 3; }
            // a frame with ONE component is never interleaved: its MCU is a single 8x8 block whatever the sampling factors say
            // (ITU T.81 A.2.2; several encoders write greyscale files with 2x2 factors)
            if (nc == 1) comps[0].h = comps[0].v = 1;
            for (auto& c : comps) { if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4) return fail(error, "bad JPEG component");
                hmax = std::max(hmax, c.h); vmax = std::max(vmax, c.v); }
            mcux = (width + 8 * hmax - 1) / (8 * hmax); mcuy = (height + 8 * vmax - 1) / (8 * vmax);
            coef.resize(nc);
            for (int c = 0; c < nc; c++) { comps[c].stride = (size_t)mcux * comps[c].h * 8;
                coef[c].assign((size_t)mcux * comps[c].h * mcuy * comps[c].v * 64, 0); }
        } else if (marker >= 0xC5 && marker <= 0xCF && marker != 0xC8 && marker != 0xCC) {
            return fail(error, "arithmetic-coded / lossless / hierarchical JPEG is not supported");
        } else if (marker == 0xDD && n >= 2) restart_interval = (d[0] << 8) | d[1];
        else if (marker == 0xEE && n >= 12 && !memcmp(d, "Adobe", 5)) adobe_transform = d[11];
        else if (marker == 0xDA) {                                       // SOS: one scan
            const int ns = n ? d[0] : 0;
            if (comps.empty() || ns < 1 || ns > (int)comps.size() || n < 1 + 2 * (size_t)ns + 3) return fail(error,
                "unsupported JPEG scan layout");
            std::vector<int> in_scan;
            for (int k = 0; k < ns; k++) {
                int found = -1;
                for (size_t c = 0; c < comps.size(); c++) if (comps[c].id == d[1 + 2 * k]) found = (int)c;
                if (found < 0) return fail(error, "JPEG scan refers to an unknown component");
                comps[found].td = d[2 + 2 * k] >> 4; comps[found].ta = d[2 + 2 * k] & 15;
                in_scan.push_back(found);
            }
            const int ss = d[1 + 2 * ns], se = d[2 + 2 * ns], ah = d[3 + 2 * ns] >> 4, al = d[3 + 2 * ns] & 15;
            if (progressive ? (ss > se || se > 63 || (ss == 0 && se != 0) || (ss > 0 && ns != 1) || al > 13) : (ss != 0 || se != 63
                || ah != 0 || al != 0))
                return fail(error, "bad JPEG scan parameters");
            for (int c : in_scan) {
                if ((!progressive || ss == 0) && ah == 0 && !dc[comps[c].td & 3].defined) return fail(error,
                    "JPEG scan without its DC table");
                if ((!progressive || ss > 0) && !ac[comps[c].ta & 3].defined) return fail(error, "JPEG scan without its AC table");
            }
            BitReader br{&file[pos + 2 + len], file.data() + file.size()};
            for (auto& c : comps) c.pred = 0;
            int eobrun = 0, until_restart = restart_interval;
            // an interleaved scan walks MCUs (h x v blocks of every component); a scan of one component walks that component's
            // own blocks: ceil(its width / 8) x ceil(its height / 8) (A.2.2, A.2.3)
            const bool interleaved = ns > 1;
            const JpegComp& c0 = comps[in_scan[0]];
            const int units_x = interleaved ? mcux : ((width * c0.h + hmax - 1) / hmax + 7) / 8,
                units_y = interleaved ? mcuy : ((height * c0.v + vmax - 1) / vmax + 7) / 8;
            for (int uy = 0; uy < units_y; uy++) for (int ux = 0; ux < units_x; ux++) {
                // RSTn: byte-align, skip the marker, reset predictors and the band run
                if (restart_interval && until_restart == 0) {
                    while (br.p + 1 < br.end && !(br.p[0] == 0xFF && br.p[1] >= 0xD0 && br.p[1] <= 0xD7)) br.p++;
                    if (br.p + 1 < br.end) br.p += 2;
                    br.restart();
                    for (auto& c : comps) c.pred = 0;
                    eobrun = 0; until_restart = restart_interval;
                }
                for (int ci : in_scan) {
                    JpegComp& c = comps[ci];
                    const int bw = interleaved ? c.h : 1, bh = interleaved ? c.v : 1;
                    for (int by = 0; by < bh; by++) for (int bx = 0; bx < bw; bx++) {
                        const size_t block_x = (size_t)ux * bw + bx, block_y = (size_t)uy * bh + by;
                        int16_t* blk = &coef[ci][(block_y * ((size_t)mcux * c.h) + block_x) * 64];
                        if (!progressive) {                              // F.2.2: the whole block
                            const int t = decode_huff(br, dc[c.td & 3]);
                            if (t < 0 || t > 11) return fail(error, "corrupt JPEG data (DC)");
                            c.pred += extend(br.bits(t), t);
                            blk[0] = (int16_t)c.pred;
                            for (int k = 1; k < 64;) {
                                const int rs = decode_huff(br, ac[c.ta & 3]);
                                if (rs < 0) return fail(error, "corrupt JPEG data (AC)");
                                const int r = rs >> 4, sz = rs & 15;
                                if (sz == 0) { if (r == 15) { k += 16; continue; } break; }
                                k += r;
                                if (k > 63) return fail(error, "corrupt JPEG data (run)");
                                blk[kZigzag[k]] = (int16_t)extend(br.bits(sz), sz);
                                k++;
                            }
                        } else if (ss == 0) {                            // G.1.2.1: DC, first pass / refinement bit
                            if (ah == 0) {
                                const int t = decode_huff(br, dc[c.td & 3]);
                                if (t < 0 || t > 11) return fail(error, "corrupt JPEG data (DC)");
                                c.pred += extend(br.bits(t), t);
                                blk[0] = (int16_t)(c.pred * (1 << al));
                            } else if (br.bit()) blk[0] = (int16_t)(blk[0] | (1 << al));
                        } else if (ah == 0) {                            // G.1.2.2: AC band, first pass
                            if (eobrun > 0) { eobrun--; continue; }
                            for (int k = ss; k <= se;) {
                                const int rs = decode_huff(br, ac[c.ta & 3]);
                                if (rs < 0) return fail(error, "corrupt JPEG data (AC)");
                                const int r = rs >> 4, sz = rs & 15;
                                if (sz == 0) {
                                    if (r < 15) { eobrun = (1 << r) - 1; if (r) eobrun += br.bits(r); break; }
                                    k += 16;
                                } else {
                                    k += r;
                                    if (k > se) return fail(error, "corrupt JPEG data (run)");
                                    blk[kZigzag[k]] = (int16_t)(extend(br.bits(sz), sz) * (1 << al));
                                    k++;
                                }
                            }
                        } else {                                         // G.1.2.3: AC band, refinement
                            const int p1 = 1 << al, m1 = -(1 << al);
                            const auto refine = [&](int16_t& v) { if (br.bit() && (v & p1) == 0) v = (int16_t)(v + (v >= 0 ? p1 : m1)); };
                            int k = ss;
                            if (eobrun == 0) {
                                for (; k <= se; k++) {
                                    const int rs = decode_huff(br, ac[c.ta & 3]);
                                    if (rs < 0) return fail(error, "corrupt JPEG data (AC)");
                                    int r = rs >> 4, val = 0;
                                    if (rs & 15) val = br.bit() ? p1 : m1;                 // (the size is 1)
                                    else if (r != 15) { eobrun = 1 << r; if (r) eobrun += br.bits(r); break; }
                                    // pass over the coefficients that are non-zero already (one correction bit each) and over r zero ones
                                    while (k <= se) {
                                        int16_t& v = blk[kZigzag[k]];
                                        if (v != 0) refine(v);
                                        else if (--r < 0) break;
                                        k++;
                                    }
                                    if (val && k <= se) blk[kZigzag[k]] = (int16_t)val;
                                }
                            }
                            if (eobrun > 0) {
                                for (; k <= se; k++) { int16_t& v = blk[kZigzag[k]]; if (v != 0) refine(v); }
                                eobrun--;
                            }
                        }
                    }
                }
                if (restart_interval) until_restart--;
            }
            any_scan = true;
            pos = (size_t)(br.p - file.data());                           // the next marker follows the entropy-coded data
            continue;
        }
        pos += 2 + len;
    }
    if (!any_scan || width <= 0 || height <= 0) return fail(error, "JPEG without image data");
    // dequantise (tables are in zigzag order) and transform every block
    for (size_t ci = 0; ci < comps.size(); ci++) {
        JpegComp& c = comps[ci];
        c.plane.assign(c.stride * mcuy * c.v * 8, 0);
        float dq[64];
        for (int k = 0; k < 64; k++) dq[kZigzag[k]] = (float)qt[c.tq][k];
        const size_t blocks_x = (size_t)mcux * c.h, blocks_y = (size_t)mcuy * c.v;
        for (size_t by = 0; by < blocks_y; by++) for (size_t bx = 0; bx < blocks_x; bx++) {
            const int16_t* blk = &coef[ci][(by * blocks_x + bx) * 64];
            float block[64];
            for (int k = 0; k < 64; k++) block[k] = (float)blk[k] * dq[k];
            idct8x8(block, &c.plane[by * 8 * c.stride + bx * 8], c.stride);
        }
    }
    img.width = width; img.height = height;
    // Upsampling to full resolution: libjpeg's "fancy" triangle filters for the common 2x1 and 2x2 chroma layouts
    // (jdsample.c h2v1_fancy_upsample / h2v2_fancy_upsample, edge samples replicated), replication otherwise.
    std::vector<std::vector<uint8_t>> full(comps.size());
    for (size_t ci = 0; ci < comps.size(); ci++) {
        const JpegComp& c = comps[ci];
        const int fx = hmax / c.h, fy = vmax / c.v;
        const int dw = (width * c.h + hmax - 1) / hmax, dh = (height * c.v + vmax - 1) / vmax;      // downsampled size
        std::vector<uint8_t>& out = full[ci];
        out.resize((size_t)width * height);
        auto in = [&](int x, int y) {
            return (int)c.plane[(size_t)std::min(std::max(y, 0), dh - 1) * c.stride + (size_t)std::min(std::max(x, 0), dw - 1)]; };
        if (hmax % c.h == 0 && vmax % c.v == 0 && fx == 2 && (fy == 1 || fy == 2) && dw >= 2) {
            for (int y = 0; y < height; y++) {
                const int r = fy == 2 ? y / 2 : y, far = fy == 2 ? ((y & 1) ? r + 1 : r - 1) : r;
                auto colsum = [&](int i) { return fy == 2 ? 3 * in(i, r) + in(i, far) : in(i, r); };      // h2v2: vertical pass first
                for (int x = 0; x < width; x++) {
                    const int i = x / 2;
                    int v;
                    if (fy == 2) {
                        if (x == 0) v = (colsum(0) * 4 + 8) >> 4;
                        else if (x == 2 * dw - 1) v = (colsum(dw - 1) * 4 + 7) >> 4;
                        else v = (x & 1) ? (colsum(i) * 3 + colsum(i + 1) + 7) >> 4 : (colsum(i) * 3 + colsum(i - 1) + 8) >> 4;
                    } else {
                        if (x == 0) v = in(0, r);
                        else if (x == 2 * dw - 1) v = in(dw - 1, r);
                        else v = (x & 1) ? (in(i, r) * 3 + in(i + 1, r) + 2) >> 2 : (in(i, r) * 3 + in(i - 1, r) + 1) >> 2;
                    }
                    out[(size_t)y * width + x] = (uint8_t)v;
                }
            }
        } else {
            for (int y = 0; y < height; y++) for (int x = 0; x < width; x++) out[(size_t)y * width + x] = (uint8_t)in(x * c.h / hmax,
                y * c.v / vmax);
        }
    }
    std::vector<uint8_t> rgba((size_t)width * height * 4);
    const bool ycc = comps.size() == 3 && adobe_transform != 0;
    for (int y = 0; y < height; y++) for (int x = 0; x < width; x++) {
        int s[3] = {0, 0, 0};
        for (size_t c = 0; c < comps.size(); c++) s[c] = full[c][(size_t)y * width + x];
        uint8_t* px = &rgba[((size_t)y * width + x) * 4];
        if (comps.size() == 1) { px[0] = px[1] = px[2] = (uint8_t)s[0]; }
        else if (ycc) {
            const float Y = (float)s[0], cb = (float)s[1] - 128.0f, cr = (float)s[2] - 128.0f;
            const int r = (int)std::lround(Y + 1.402f * cr), g = (int)std::lround(Y - 0.344136f * cb - 0.714136f * cr),
                b = (int)std::lround(Y + 1.772f * cb);
            px[0] = (uint8_t)std::min(255, std::max(0, r)); px[1] = (uint8_t)std::min(255, std::max(0, g));
            px[2] = (uint8_t)std::min(255, std::max(0, b));
        } else { px[0] = (uint8_t)s[0]; px[1] = (uint8_t)s[1]; px[2] = (uint8_t)s[2]; }
        px[3] = 255;
    }
    finish(img, rgba);
    return true;
}

bool load_tga(const std::string& path, ImageRgba8& img, std::string* error) {
    std::vector<uint8_t> file;
    if (!read_file(path, file)) return fail(error, "cannot read '" + path + "'");
    if (file.size() < 18) return fail(error, "not a TGA file");
    const int idlen = file[0], cmaptype = file[1], type = file[2], cmaplen = file[5] | (file[6] << 8), cmapbits = file[7];
    const int width = file[12] | (file[13] << 8), height = file[14] | (file[15] << 8), bpp = file[16], desc = file[17];
    const bool rle = type == 10 || type == 11, gray = type == 3 || type == 11;
    if (!(type == 2 || type == 3 || type == 10 || type == 11) || width <= 0 || height <= 0 ||
        !((gray && bpp == 8) || (!gray && (bpp == 24 || bpp == 32))))
        return fail(error, "unsupported TGA type");
    size_t pos = 18 + (size_t)idlen + (cmaptype ? (size_t)cmaplen * ((cmapbits + 7) / 8) : 0);
    const int bytes = bpp / 8;
    std::vector<uint8_t> rgba((size_t)width * height * 4);
    size_t count = (size_t)width * height, i = 0;
    auto put = [&](const uint8_t* p) {
        const size_t y = i / width, x = i % width;
        const size_t row = (desc & 0x20) ? y : (size_t)height - 1 - y;           // bit 5: rows top-down; else bottom-up
        const size_t col = (desc & 0x10) ? (size_t)width - 1 - x : x;
        uint8_t* px = &rgba[(row * width + col) * 4];
        if (gray) { px[0] = px[1] = px[2] = p[0]; px[3] = 255; }
        else { px[0] = p[2]; px[1] = p[1]; px[2] = p[0]; px[3] = bytes == 4 ? p[3] : 255; }
        i++;
    };
    while (i < count) {
        if (!rle) { if (pos + bytes > file.size()) return fail(error, "truncated TGA"); put(&file[pos]); pos += bytes; continue; }
        if (pos >= file.size()) return fail(error, "truncated TGA");
        const int hdr = file[pos++], run = (hdr & 0x7F) + 1;
        if (hdr & 0x80) {
            if (pos + bytes > file.size()) return fail(error, "truncated TGA");
            for (int k = 0; k < run && i < count; k++) put(&file[pos]);
            pos += bytes;
        } else {
            for (int k = 0; k < run && i < count; k++) { if (pos + bytes > file.size()) return fail(error, "truncated TGA");
                put(&file[pos]); pos += bytes; }
        }
    }
    img.width = width; img.height = height;
    finish(img, rgba);
    return true;
}

bool load_image(const std::string& path, ImageRgba8& img, std::string* error) {
    std::string lower = path;
    std::transform(lower.begin(), lower.end(), lower.begin(), [](unsigned char c) { return (char)std::tolower(c); });
    auto ends = [&](const char* e) { const size_t n = strlen(e); return lower.size() >= n && lower.compare(lower.size() - n, n, e) == 0; };
    if (ends(".png")) return load_png(path, img, error);
    if (ends(".jpg") || ends(".jpeg")) return load_jpg(path, img, error);
    if (ends(".tga")) return load_tga(path, img, error);
    return fail(error, "cannot determine the image type of '" + path + "'");
}

} // namespace rodent

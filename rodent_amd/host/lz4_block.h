// LZ4 block format, as the reference's buffer files use it through liblz4 (src/driver/buffer.h:17-20,40-44:
// LZ4_decompress_safe / LZ4_compress_default).  lz4.h is not in this image, so the block codec is written here from
// the published format: a block is a sequence of [token][literal length bytes*][literals][offset lo hi][match length bytes*];
// token = (literal length << 4) | (match length - 4), 15 in a nibble continues in 255-valued bytes; the last sequence
// has literals only; the last 5 bytes of a block are literals and a match may not start within the last 12 bytes.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace rodent {

// Decompresses exactly dst_size bytes; false on malformed input (never reads or writes out of bounds).
inline bool lz4_decompress(const uint8_t* src, size_t src_size, uint8_t* dst, size_t dst_size) {
    size_t ip = 0, op = 0;
    while (ip < src_size) {
        const unsigned token = src[ip++];
        size_t lit = token >> 4;
        if (lit == 15) { unsigned b; do { if (ip >= src_size) return false; b = src[ip++]; lit += b; } while (b == 255); }
        if (lit > src_size - ip || lit > dst_size - op) return false;
        std::memcpy(dst + op, src + ip, lit); ip += lit; op += lit;
        if (ip == src_size) break;                                   // last sequence: literals only
        if (ip + 2 > src_size) return false;
        const size_t offset = src[ip] | ((size_t)src[ip + 1] << 8); ip += 2;
        if (offset == 0 || offset > op) return false;
        size_t len = (token & 15);
        if (len == 15) { unsigned b; do { if (ip >= src_size) return false; b = src[ip++]; len += b; } while (b == 255); }
        len += 4;
        if (len > dst_size - op) return false;
        for (size_t k = 0; k < len; k++) dst[op + k] = dst[op + k - offset];     // byte by byte: matches may overlap themselves
        op += len;
    }
    return op == dst_size;
}

// Greedy single-pass compressor with a 64 Ki-entry hash of 4-byte sequences; any LZ4 decoder reads its output.
inline std::vector<uint8_t> lz4_compress(const uint8_t* src, size_t n) {
    std::vector<uint8_t> out;
    out.reserve(n / 2 + 16);
    auto put_len = [&](size_t v) { while (v >= 255) { out.push_back(255); v -= 255; } out.push_back((uint8_t)v); };
    auto emit = [&](size_t lit_begin, size_t lit_len, size_t offset, size_t match_len) {      // match_len == 0: final literals
        const size_t ml = match_len ? match_len - 4 : 0;
        out.push_back((uint8_t)((lit_len >= 15 ? 15 : lit_len) << 4 | (match_len ? (ml >= 15 ? 15 : ml) : 0)));
        if (lit_len >= 15) put_len(lit_len - 15);
        out.insert(out.end(), src + lit_begin, src + lit_begin + lit_len);
        if (match_len) { out.push_back((uint8_t)(offset & 255)); out.push_back((uint8_t)(offset >> 8)); if (ml >= 15) put_len(ml - 15); }
    };
    std::vector<uint32_t> table(1u << 16, 0xFFFFFFFFu);
    size_t anchor = 0, i = 0;
    if (n >= 13) {
        const size_t match_limit = n - 12;                           // no match starts in the last 12 bytes
        while (i < match_limit) {
            uint32_t v; std::memcpy(&v, src + i, 4);
            const uint32_t h = (v * 2654435761u) >> 16;
            const uint32_t cand = table[h];
            table[h] = (uint32_t)i;
            uint32_t w = 0;
            if (cand != 0xFFFFFFFFu && i - cand <= 65535 && i < 0xFFFFFFFFull) std::memcpy(&w, src + cand, 4);
            if (cand != 0xFFFFFFFFu && i - cand <= 65535 && w == v) {
                size_t len = 4;
                while (i + len < n - 5 && src[cand + len] == src[i + len]) len++;       // the last 5 bytes stay literals
                emit(anchor, i - anchor, i - cand, len);
                i += len; anchor = i;
            } else i++;
        }
    }
    emit(anchor, n - anchor, 0, 0);
    return out;
}

} // namespace rodent

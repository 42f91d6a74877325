// The reference's compressed buffer files (src/driver/buffer.h): [u32 byte size][u32 compressed size][LZ4 block], as
// written by its converter into data/*.bin and read by its runtime (interface.cpp:456-468 load_buffer); and data/bvh.bin
// (converter.cpp:428-437, interface.cpp:432-454): per layout [u32 sizeof(Node)][u32 sizeof(Tri)][nodes buffer][tris buffer].
#pragma once
#include <cstdio>
#include <string>
#include <vector>

#include "lz4_block.h"

namespace rodent {

template <typename T> bool write_buffer(FILE* f, const std::vector<T>& v) {
    const std::vector<uint8_t> c = lz4_compress(reinterpret_cast<const uint8_t*>(v.data()), v.size() * sizeof(T));
    const uint32_t hdr[2] = {(uint32_t)(v.size() * sizeof(T)), (uint32_t)c.size()};
    return fwrite(hdr, 4, 2, f) == 2 && (c.empty() || fwrite(c.data(), 1, c.size(), f) == c.size());
}
template <typename T> bool read_buffer(FILE* f, std::vector<T>& v) {
    uint32_t hdr[2];
    if (fread(hdr, 4, 2, f) != 2 || hdr[0] % sizeof(T)) return false;
    // nothing is allocated on the word of a header alone: the compressed bytes must be in the file, and an LZ4 block
    // cannot expand by more than 255x (one length byte per 255 output bytes)
    const long here = ftell(f);
    if (here < 0 || fseek(f, 0, SEEK_END) != 0) return false;
    const long end = ftell(f);
    if (end < here || fseek(f, here, SEEK_SET) != 0 || (uint64_t)(end - here) < hdr[1]
        || (uint64_t)hdr[0] > 255ull * hdr[1] + 16) return false;
    std::vector<uint8_t> c(hdr[1]);
    if (hdr[1] && fread(c.data(), 1, c.size(), f) != c.size()) return false;
    v.resize(hdr[0] / sizeof(T));
    return lz4_decompress(c.data(), c.size(), reinterpret_cast<uint8_t*>(v.data()), hdr[0]);
}
inline bool skip_buffer(FILE* f) { uint32_t hdr[2]; return fread(hdr, 4, 2, f) == 2 && fseek(f, hdr[1], SEEK_CUR) == 0; }

template <typename T> bool write_buffer_file(const std::string& path, const std::vector<T>& v) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = write_buffer(f, v);
    return fclose(f) == 0 && ok;
}
template <typename T> bool read_buffer_file(const std::string& path, std::vector<T>& v) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    const bool ok = read_buffer(f, v);
    fclose(f);
    return ok;
}

// appends one layout to data/bvh.bin
template <typename Node,
    typename Tri> bool append_bvh_bin(const std::string& path, const std::vector<Node>& nodes, const std::vector<Tri>& tris) {
    FILE* f = fopen(path.c_str(), "ab");
    if (!f) return false;
    const uint32_t sizes[2] = {(uint32_t)sizeof(Node), (uint32_t)sizeof(Tri)};
    const bool ok = fwrite(sizes, 4, 2, f) == 2 && write_buffer(f, nodes) && write_buffer(f, tris);
    return fclose(f) == 0 && ok;
}
// finds the layout with these element sizes
template <typename Node, typename Tri> bool load_bvh_bin(const std::string& path, std::vector<Node>& nodes, std::vector<Tri>& tris) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    bool found = false;
    for (uint32_t sizes[2]; fread(sizes, 4, 2, f) == 2;) {
        if (sizes[0] == sizeof(Node) && sizes[1] == sizeof(Tri)) { found = read_buffer(f, nodes) && read_buffer(f, tris); break; }
        if (!skip_buffer(f) || !skip_buffer(f)) break;
    }
    fclose(f);
    return found;
}

} // namespace rodent

// On-disk formats shared with the reference's tools.
//
//  .bvh   u32 magic 0x95CBED1F, then blocks
//         { u64 offset; u32 type; u32 node_count; u32 tri_count; Node[]; Tri[] }
//         with offset = 12 + sizeof(nodes) + sizeof(tris) and type 1 = BVH2_TRI1,
//         2 = BVH4_TRI4, 3 = BVH8_TRI4.
//         reader: tools/common/load_bvh.h:21-74; writers: tools/bvh_extractor/extract_bvh2.cpp:123-135,
//         extract_bvh4_8.cpp:11-23.
//  .rays  headerless float[6] per ray: org xyz, dir xyz (tools/common/load_rays.h:59-92)
//  .fbuf  headerless float t per ray (tools/bench_traversal/bench_traversal.cpp:342-346)
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>
#include "../../include/rodent_traversal.h"

namespace rodent {

enum class BvhType : uint32_t { BVH2_TRI1 = 1, BVH4_TRI4 = 2, BVH8_TRI4 = 3 };
constexpr uint32_t kBvhMagic = 0x95CBED1Fu;

template <typename Node, typename Tri>
inline bool load_bvh(const std::string& file, BvhType type, std::vector<Node>& nodes, std::vector<Tri>& tris) {
    FILE* f = fopen(file.c_str(), "rb");
    if (!f) return false;
    uint32_t magic = 0;
    bool ok = fread(&magic, 4, 1, f) == 1 && magic == kBvhMagic;
    while (ok) {
        uint64_t offset = 0; uint32_t block = 0;
        if (fread(&offset, 8, 1, f) != 1 || fread(&block, 4, 1, f) != 1) { ok = false; break; }
        if (block == (uint32_t)type) {
            uint32_t hdr[2];
            ok = fread(hdr, 4, 2, f) == 2;
            if (ok) {
                // the counts must agree with the block size and the block must fit in the file BEFORE anything is allocated
                const uint64_t payload = sizeof(Node) * (uint64_t)hdr[0] + sizeof(Tri) * (uint64_t)hdr[1];
                const long here = ftell(f);
                ok = offset == 12 + payload && here >= 0 && fseek(f, 0, SEEK_END) == 0 && (uint64_t)(ftell(f) - here) >= payload
                    && fseek(f, here, SEEK_SET) == 0;
            }
            if (ok) {
                nodes.resize(hdr[0]); tris.resize(hdr[1]);
                ok = fread(nodes.data(), sizeof(Node), hdr[0], f) == hdr[0]
                  && fread(tris.data(), sizeof(Tri), hdr[1], f) == hdr[1];
            }
            break;
        }
        if (offset < 4 || fseek(f, (long)(offset - 4), SEEK_CUR) != 0) ok = false;
    }
    fclose(f);
    return ok;
}

inline bool begin_bvh_file(FILE* f) { return fwrite(&kBvhMagic, 4, 1, f) == 1; }

template <typename Node, typename Tri>
inline bool append_bvh_block(FILE* f, BvhType type, const std::vector<Node>& nodes, const std::vector<Tri>& tris) {
    const uint64_t offset = 12 + sizeof(Node) * (uint64_t)nodes.size() + sizeof(Tri) * (uint64_t)tris.size();
    const uint32_t hdr[3] = {(uint32_t)type, (uint32_t)nodes.size(), (uint32_t)tris.size()};
    return fwrite(&offset, 8, 1, f) == 1 && fwrite(hdr, 4, 3, f) == 3
        && fwrite(nodes.data(), sizeof(Node), nodes.size(), f) == nodes.size()
        && fwrite(tris.data(), sizeof(Tri), tris.size(), f) == tris.size();
}

// Reads a .rays file into Ray1 records with the given [tmin, tmax] (load_rays.h:59-92).
inline bool load_rays(const std::string& file, float tmin, float tmax, std::vector<Ray1>& rays) {
    FILE* f = fopen(file.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (size < 0 || size % 24 != 0) { fclose(f); return false; }
    const size_t n = (size_t)size / 24;
    std::vector<float> raw(n * 6);
    const bool ok = fread(raw.data(), 24, n, f) == n;
    fclose(f);
    if (!ok) return false;
    rays.resize(n);
    for (size_t i = 0; i < n; i++) {
        const float* r = &raw[6 * i];
        rays[i] = Ray1{{r[0], r[1], r[2]}, tmin, {r[3], r[4], r[5]}, tmax};
    }
    return true;
}

inline bool save_fbuf(const std::string& file, const std::vector<Hit1>& hits) {
    FILE* f = fopen(file.c_str(), "wb");
    if (!f) return false;
    for (auto& h : hits) fwrite(&h.t, 4, 1, f);
    fclose(f);
    return true;
}

} // namespace rodent

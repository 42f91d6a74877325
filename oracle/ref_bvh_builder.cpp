// ref_bvh_builder -- drives the REFERENCE's own OBJ loader and SBVH builder
// (/root/reference/src/driver/{obj.cpp,obj.h,bvh.h,tri.h,bbox.h,...}, compiled from
// where they lie; nothing of the reference is copied into this repository) and
// writes the result as a .bvh file with BVH8/BVH4/BVH2 blocks.
//
// TEST INFRASTRUCTURE: built into oracle/_ref/ by oracle/Makefile.ref when
// /root/reference is present.  Used to cross-check the in-tree builder
// (rodent_amd/host/bvh_build.cpp): same triangle order from the OBJ, comparable
// node / reference counts and SAH cost, and traversal results that agree with the
// exhaustive checker on BVHs the reference itself built.
//
// The node/leaf writer callbacks below are this repository's code; they follow the
// layouts documented in include/rodent_traversal.h (cf. src/driver/converter.cpp:152-259,296-382).
#include <cstdio>
#include <cstring>
#include <iostream>
#include <limits>
#include <vector>

#include "driver/bvh.h"
#include "driver/obj.h"

#include "../include/rodent_traversal.h"

struct CostFn {
    static float leaf_cost(int count, float area) { return count * area; }
    static float traversal_cost(float area) { return area; }
};

template <int N> struct WideNodeT { float bounds[6][N]; int32_t child[N]; int32_t pad[N]; };

template <int N>
static void build_wide(const std::vector<Tri>& tris, const std::vector<uint32_t>& geom,
                       std::vector<WideNodeT<N>>& nodes, std::vector<Tri4>& out, size_t& refs) {
    const float inf = std::numeric_limits<float>::infinity();
    SplitBvhBuilder<N, CostFn> builder;
    refs = 0;
    auto write_node = [&](int parent, int child, const BBox&, size_t count, auto bboxes) {
        const int i = (int)nodes.size();
        nodes.emplace_back();
        std::memset(&nodes[i], 0, sizeof(nodes[i]));
        if (parent >= 0 && child >= 0) nodes[parent].child[child] = i + 1;
        for (size_t j = 0; j < (size_t)N; j++) {
            if (j < count) {
                const BBox& b = bboxes(j);
                nodes[i].bounds[0][j] = b.min.x; nodes[i].bounds[1][j] = b.max.x;
                nodes[i].bounds[2][j] = b.min.y; nodes[i].bounds[3][j] = b.max.y;
                nodes[i].bounds[4][j] = b.min.z; nodes[i].bounds[5][j] = b.max.z;
            } else {
                nodes[i].bounds[0][j] = nodes[i].bounds[2][j] = nodes[i].bounds[4][j] = inf;
                nodes[i].bounds[1][j] = nodes[i].bounds[3][j] = nodes[i].bounds[5][j] = -inf;
            }
        }
        return i;
    };
    auto write_leaf = [&](int parent, int child, const BBox&, size_t ref_count, auto ref_ids) {
        nodes[parent].child[child] = ~(int32_t)out.size();
        refs += ref_count;
        for (size_t i = 0; i < ref_count; i += 4) {
            Tri4 t; std::memset(&t, 0, sizeof t);
            const size_t c = std::min<size_t>(4, ref_count - i);
            for (size_t j = 0; j < c; j++) {
                const int id = ref_ids(i + j);
                const float3 e1 = tris[id].v0 - tris[id].v1, e2 = tris[id].v2 - tris[id].v0, n = cross(e1, e2);
                for (int k = 0; k < 3; k++) { t.v0[k][j] = tris[id].v0[k]; t.e1[k][j] = e1[k]; t.e2[k][j] = e2[k]; t.n[k][j] = n[k]; }
                t.prim_id[j] = id; t.geom_id[j] = (int32_t)geom[id];
            }
            for (size_t j = c; j < 4; j++) t.prim_id[j] = -1;
            out.push_back(t);
        }
        out.back().prim_id[3] |= (int32_t)0x80000000u;
    };
    builder.build(tris, write_node, write_leaf, 2);
}

static void build_bvh2(const std::vector<Tri>& tris, const std::vector<uint32_t>& geom,
                       std::vector<Node2>& nodes, std::vector<Tri1>& out, size_t& refs) {
    const float inf = std::numeric_limits<float>::infinity();
    SplitBvhBuilder<2, CostFn> builder;
    refs = 0;
    auto write_node = [&](int parent, int child, const BBox&, size_t count, auto bboxes) {
        const int i = (int)nodes.size();
        nodes.emplace_back();
        std::memset(&nodes[i], 0, sizeof(Node2));
        if (parent >= 0 && child >= 0) nodes[parent].child[child] = i + 1;
        for (size_t j = 0; j < 2; j++) {
            float* b = nodes[i].bounds + 6 * j;
            if (j < count) {
                const BBox& bb = bboxes(j);
                b[0] = bb.min.x; b[1] = bb.max.x; b[2] = bb.min.y; b[3] = bb.max.y; b[4] = bb.min.z; b[5] = bb.max.z;
            } else { b[0] = b[2] = b[4] = inf; b[1] = b[3] = b[5] = -inf; }
        }
        return i;
    };
    auto write_leaf = [&](int parent, int child, const BBox&, size_t ref_count, auto ref_ids) {
        nodes[parent].child[child] = ~(int32_t)out.size();
        refs += ref_count;
        for (size_t i = 0; i < ref_count; i++) {
            const int id = ref_ids(i);
            const float3 e1 = tris[id].v0 - tris[id].v1, e2 = tris[id].v2 - tris[id].v0;
            Tri1 t;
            t.v0[0] = tris[id].v0.x; t.v0[1] = tris[id].v0.y; t.v0[2] = tris[id].v0.z; t.pad = 0;
            t.e1[0] = e1.x; t.e1[1] = e1.y; t.e1[2] = e1.z; t.geom_id = (int32_t)geom[id];
            t.e2[0] = e2.x; t.e2[1] = e2.y; t.e2[2] = e2.z; t.prim_id = id;
            out.push_back(t);
        }
        out.back().prim_id |= (int32_t)0x80000000u;
    };
    builder.build(tris, write_node, write_leaf, 2);
}

template <typename NodeT, typename TriT>
static void write_block(FILE* f, uint32_t type, const std::vector<NodeT>& nodes, const std::vector<TriT>& tris) {
    const uint64_t offset = 12 + sizeof(NodeT) * (uint64_t)nodes.size() + sizeof(TriT) * (uint64_t)tris.size();
    const uint32_t hdr[3] = {type, (uint32_t)nodes.size(), (uint32_t)tris.size()};
    fwrite(&offset, 8, 1, f); fwrite(hdr, 4, 3, f);
    fwrite(nodes.data(), sizeof(NodeT), nodes.size(), f);
    fwrite(tris.data(), sizeof(TriT), tris.size(), f);
}

int main(int argc, char** argv) {
    if (argc != 3) { std::cerr << "usage: ref_bvh_builder in.obj out.bvh" << std::endl; return 1; }
    obj::File file;
    if (!obj::load_obj(FilePath(argv[1]), file)) { std::cerr << "cannot load OBJ" << std::endl; return 1; }
    const obj::TriMesh mesh = obj::compute_tri_mesh(file, 0);
    std::vector<Tri> tris; std::vector<uint32_t> geom;
    for (size_t i = 0; i < mesh.indices.size(); i += 4) {
        tris.emplace_back(mesh.vertices[mesh.indices[i]], mesh.vertices[mesh.indices[i + 1]], mesh.vertices[mesh.indices[i + 2]]);
        geom.push_back(mesh.indices[i + 3]);
    }
    FILE* f = fopen(argv[2], "wb");
    if (!f) return 1;
    const uint32_t magic = 0x95CBED1Fu; fwrite(&magic, 4, 1, f);
    size_t refs;
    { std::vector<WideNodeT<8>> n; std::vector<Tri4> t; build_wide<8>(tris, geom, n, t, refs); write_block(f, 3, n, t);
      printf("REF BVH8 %zu nodes %zu refs %zu packets\n", n.size(), refs, t.size()); }
    { std::vector<WideNodeT<4>> n; std::vector<Tri4> t; build_wide<4>(tris, geom, n, t, refs); write_block(f, 2, n, t);
      printf("REF BVH4 %zu nodes %zu refs %zu packets\n", n.size(), refs, t.size()); }
    { std::vector<Node2> n; std::vector<Tri1> t; build_bvh2(tris, geom, n, t, refs); write_block(f, 1, n, t);
      printf("REF BVH2 %zu nodes %zu refs %zu tris\n", n.size(), refs, t.size()); }
    fclose(f);
    printf("triangles %zu\n", tris.size());
    return 0;
}

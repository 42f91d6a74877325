// bvh_extractor -- OBJ -> .bvh (BVH8/Tri4, BVH4/Tri4 and BVH2/Tri1 blocks).
// Same command line and block order as the reference tool built with Embree
// (tools/bvh_extractor/bvh_extractor.cpp:23-111: BVH8, BVH4, then BVH2), but all
// three blocks come from the in-tree spatial-split builder, the way the
// reference's scene converter does it (src/driver/converter.cpp:118-383,733-740).
#include <cstring>
#include <iostream>
#include <thread>

#include "../bvh_build.h"
#include "../formats.h"
#include "../mesh.h"

using namespace rodent;

static void usage() {
    std::cout << "Usage: bvh_extractor [options]\n"
                 "Available options:\n"
                 "  -obj     --obj-file        Sets the OBJ file to use\n"
                 "  -o       --output          Sets the output file name\n"
                 "           --no-spatial      Disables spatial splits\n";
}

int main(int argc, char** argv) {
    std::string obj_file, out_file;
    bool spatial = true;
    for (int i = 1; i < argc; i++) {
        const char* arg = argv[i];
        auto need = [&]() { if (i + 1 >= argc) { std::cerr << "Missing argument for " << arg << std::endl; exit(1); } return argv[++i]; };
        if (!strcmp(arg, "-h") || !strcmp(arg, "--help")) { usage(); return 0; }
        else if (!strcmp(arg, "-obj") || !strcmp(arg, "--obj-file")) obj_file = need();
        else if (!strcmp(arg, "-o") || !strcmp(arg, "--output")) out_file = need();
        else if (!strcmp(arg, "--no-spatial")) spatial = false;
        else if (arg[0] == '-') { std::cerr << "Unknown option '" << arg << "'" << std::endl; return 1; }
        else { std::cerr << "Invalid argument '" << arg << "'" << std::endl; return 1; }
    }
    if (obj_file.empty()) { std::cerr << "No OBJ file specified" << std::endl; return 1; }
    if (out_file.empty()) { std::cerr << "No output file specified" << std::endl; return 1; }

    TriMesh mesh;
    if (!load_obj(obj_file, mesh)) { std::cerr << "Cannot load OBJ file" << std::endl; return 1; }
    std::cout << "Loaded OBJ file with " << mesh.num_tris() << " triangle(s)" << std::endl;
    if (mesh.num_tris() == 0) { std::cerr << "The OBJ file has no faces: nothing to build a BVH from" << std::endl; return 1; }
    const std::vector<Triangle> tris = mesh.triangles();
    std::vector<uint32_t> geom(mesh.num_tris());
    for (size_t i = 0; i < geom.size(); i++) geom[i] = mesh.indices[4 * i + 3];

    FILE* out = fopen(out_file.c_str(), "wb");
    if (!out || !begin_bvh_file(out)) { std::cerr << "Cannot create output file" << std::endl; return 1; }

    auto report = [](const char* name, const WideBvh& b, size_t prims) {
        std::cout << name << " successfully built (" << b.nodes.size() << " nodes, " << b.leaves.size() << " leaves, "
                  << b.num_refs << " refs, " << prims << " prim packets, " << b.object_splits << " object + "
                  << b.spatial_splits << " spatial splits, depth " << b.depth << ", SAH " << b.sah_cost << ")" << std::endl;
    };
    // The three trees are independent: build them on three host threads, write in the reference's order.
    std::vector<Node8> n8; std::vector<Tri4> t8; std::vector<Node4> n4; std::vector<Tri4> t4; std::vector<Node2> n2; std::vector<Tri1> t1;
    WideBvh b8, b4, b2;
    auto params = [&](int arity) { BuildParams p; p.spatial_splits = spatial; p.arity = arity; return p; };
    std::thread th8([&] { b8 = build_wide_bvh(tris, params(8)); layout_bvh8_tri4(b8, tris, geom.data(), n8, t8); });
    std::thread th4([&] { b4 = build_wide_bvh(tris, params(4)); layout_bvh4_tri4(b4, tris, geom.data(), n4, t4); });
    b2 = build_wide_bvh(tris, params(2)); layout_bvh2_tri1(b2, tris, geom.data(), n2, t1);
    th8.join(); th4.join();
    if (!append_bvh_block(out, BvhType::BVH8_TRI4, n8, t8)) return 1;
    report("BVH8", b8, t8.size());
    if (!append_bvh_block(out, BvhType::BVH4_TRI4, n4, t4)) return 1;
    report("BVH4", b4, t4.size());
    if (!append_bvh_block(out, BvhType::BVH2_TRI1, n2, t1)) return 1;
    report("BVH2", b2, t1.size());
    fclose(out);
    return 0;
}

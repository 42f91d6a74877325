// scene_gen -- writes the procedural benchmark scenes.  New tool: the reference ships its test scene (Sponza) as binary blobs that are
// absent from the checkout, and its benchmark suite's other scenes (benchmarks/benchmark.py:16-21) not at all.
//   scene_gen <atrium|gallery|crown|plant> out.obj [seed] [detail]          OBJ (+ atrium.mtl beside it)
//   scene_gen <atrium|gallery|crown|plant> --bvh out.bvh [seed] [detail]    straight to a .bvh with a BVH2/Tri1 block (no OBJ round trip:
//                                                                           the
//                                                                           multi-million-triangle scenes are built where they are needed)
// gallery = the atrium at detail 4 (4.2 M triangles); crown / plant: stress_scenes.cpp.  detail defaults: atrium 1, the others 4.
#include <chrono>
#include <cstring>
#include <fstream>
#include <iostream>

#include "../atrium.h"
#include "../bvh_build.h"
#include "../formats.h"

using namespace rodent;

int main(int argc, char** argv) {
    const char* kinds[] = {"atrium", "gallery", "crown", "plant"};
    int kind = -1;
    for (int k = 0; k < 4 && argc > 1; k++) if (!strcmp(argv[1], kinds[k])) kind = k;
    if (argc < 3 || kind < 0) {
        std::cerr << "Usage: scene_gen <atrium|gallery|crown|plant> (out.obj | --bvh out.bvh) [seed] [detail]" << std::endl;
        return 1;
    }
    int a = 2;
    const bool to_bvh = !strcmp(argv[a], "--bvh");
    if (to_bvh) a++;
    if (a >= argc) { std::cerr << "scene_gen: no output file" << std::endl; return 1; }
    const std::string out = argv[a++];
    const uint64_t seed = a < argc ? strtoull(argv[a++], nullptr, 10) : 1;
    const int detail = a < argc ? atoi(argv[a++]) : (kind == 0 ? 1 : 4);
    TriMesh mesh;
    if (kind <= 1) generate_atrium(mesh, seed, detail);
    else if (kind == 2) generate_crown(mesh, seed, detail);
    else generate_plant(mesh, seed, detail);
    std::cout << kinds[kind] << ": " << mesh.num_tris() << " triangle(s), " << mesh.vertices.size() << " vertices" << std::endl;
    if (!to_bvh) {
        if (!save_obj(out, mesh)) { std::cerr << "Cannot write " << out << std::endl; return 1; }
        auto slash = out.find_last_of('/');
        std::ofstream mtl((slash == std::string::npos ? std::string() : out.substr(0, slash + 1)) + "atrium.mtl");
        mtl << atrium_mtl_text();
        return 0;
    }
    const auto t0 = std::chrono::steady_clock::now();
    const std::vector<Triangle> tris = mesh.triangles();
    std::vector<uint32_t> geom(mesh.num_tris());
    for (size_t i = 0; i < geom.size(); i++) geom[i] = mesh.indices[4 * i + 3];
    BuildParams p; p.arity = 2;
    const WideBvh b2 = build_wide_bvh(tris, p);
    std::vector<Node2> n2; std::vector<Tri1> t1;
    layout_bvh2_tri1(b2, tris, geom.data(), n2, t1);
    FILE* f = fopen(out.c_str(), "wb");
    if (!f || !begin_bvh_file(f) || !append_bvh_block(f, BvhType::BVH2_TRI1, n2, t1)) { std::cerr << "Cannot write " << out << std::endl;
        return 1; }
    fclose(f);
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::cout << "BVH2 successfully built (" << b2.nodes.size() << " nodes, " << b2.leaves.size() << " leaves, " << b2.num_refs << " refs, "
              << b2.object_splits << " object + " << b2.spatial_splits << " spatial splits, depth " << b2.depth << ", SAH " << b2.sah_cost
                  << ") in " << secs << " s" << std::endl;
    return 0;
}

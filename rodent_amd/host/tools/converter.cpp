// converter -- scene.obj (+ .mtl) -> scene description for the renderer.
// The reference's converter (src/driver/converter.cpp:973-1092) emits Impala source + LZ4 data
// files for one target; this one emits one binary table file (.rscene, see host/scene.h) that the
// `rodent` CLI and the Python binding load at run time.  The reference's options are accepted:
//   --samples-per-pixel / --max-path-len are stored as the scene's defaults,
//   --target / --device / --fusion are accepted and ignored (the target is always the HIP wavefront device).
// Beside the .rscene file it can write (--data-dir) and check (--verify-data-dir) the reference converter's LZ4 buffer
// files (data/vertices.bin, ..., data/bvh.bin; src/driver/buffer.h): mesh, BVH2/Tri1 and light tables.
#include <cstring>
#include <iostream>

#include "../bvh_build.h"
#include "../scene.h"

static void usage() {
    std::cout << "converter [options] file\n"
                 "Available options:\n"
                 "    -h     --help                 Shows this message\n"
                 "    -o     --output               Sets the output file (default: scene.rscene)\n"
                 "    -t     --target               (accepted for compatibility, ignored)\n"
                 "    -d     --device               (accepted for compatibility, ignored)\n"
                 "           --max-path-len         Sets the default maximum path length (default: 64)\n"
                 "    -spp   --samples-per-pixel    Sets the default number of samples per pixel (default: 4)\n"
                 "           --fusion               (accepted for compatibility, ignored)\n"
                 "           --bvh-leaf n           Stops splitting at n references (default: 2, the reference's threshold)\n"
                 "           --bvh-traversal-cost x Cost of an inner node in triangle tests (default: 1, the reference's)\n"
                 "           --data-dir dir         Also writes the reference's LZ4 buffer files (vertices.bin, bvh.bin, ...) into dir\n"
                 "           --verify-data-dir dir  Reads such files back and compares them with the converted scene\n";
}

int main(int argc, char** argv) {
    if (argc < 2) { std::cerr << "Not enough arguments. Run with --help to get a list of options." << std::endl; return 1; }
    std::string obj, out = "scene.rscene", data_dir, verify_dir;
    int spp = 4, max_path_len = 64;
    rodent::BuildParams bvh;
    for (int i = 1; i < argc; i++) {
        const char* a = argv[i];
        auto need = [&]() { if (i + 1 >= argc) { std::cerr << "Not enough arguments for " << a << std::endl; exit(1); } return argv[++i]; };
        if (a[0] != '-') { if (!obj.empty()) { std::cerr << "Scene file specified twice" << std::endl; return 1; } obj = a; continue; }
        if (!strcmp(a, "-h") || !strcmp(a, "--help")) { usage(); return 0; }
        else if (!strcmp(a, "-o") || !strcmp(a, "--output")) out = need();
        else if (!strcmp(a, "-t") || !strcmp(a, "--target") || !strcmp(a, "-d") || !strcmp(a, "--device")) need();
        else if (!strcmp(a, "--max-path-len")) max_path_len = strtol(need(), nullptr, 10);
        else if (!strcmp(a, "-spp") || !strcmp(a, "--samples-per-pixel")) spp = strtol(need(), nullptr, 10);
        else if (!strcmp(a, "--fusion")) {}
        else if (!strcmp(a, "--bvh-leaf")) bvh.leaf_threshold = strtol(need(), nullptr, 10);
        else if (!strcmp(a, "--bvh-traversal-cost")) bvh.traversal_cost = strtof(need(), nullptr);
        else if (!strcmp(a, "--data-dir")) data_dir = need();
        else if (!strcmp(a, "--verify-data-dir")) verify_dir = need();
        else { std::cerr << "Unknown option '" << a << "'" << std::endl; return 1; }
    }
    if (obj.empty()) { std::cerr << "Please specify an OBJ file to convert" << std::endl; return 1; }
    rodent::SceneData scene;
    if (!rodent::build_scene_from_obj(obj, scene, &bvh)) { std::cerr << "Invalid OBJ file '" << obj << "'" << std::endl; return 1; }
    scene.default_spp = spp; scene.default_max_path_len = max_path_len;
    if (!rodent::save_scene(out, scene)) { std::cerr << "Cannot write '" << out << "'" << std::endl; return 1; }
    if (!data_dir.empty() && !rodent::save_reference_data(data_dir, scene)) {
        std::cerr << "Cannot write the data files into '" << data_dir << "'" << std::endl; return 1; }
    if (!verify_dir.empty()) {
        rodent::SceneData back;
        if (!rodent::load_reference_data(verify_dir, back)) {
            std::cerr << "Cannot read the data files in '" << verify_dir << "'" << std::endl; return 1; }
        auto same = [](const auto& a, const auto& b) {
            return a.size() == b.size() && (a.empty() || !memcmp(a.data(), b.data(), a.size() * sizeof(a[0]))); };
        const bool ok = same(back.vertices, scene.vertices) && same(back.normals, scene.normals)
            && same(back.face_normals, scene.face_normals) &&
                        same(back.indices, scene.indices) && same(back.texcoords, scene.texcoords) && same(back.nodes, scene.nodes)
                            && same(back.tris, scene.tris) &&
                        same(back.light_ids, scene.light_ids) && same(back.lights, scene.lights);
        if (!ok) { std::cerr << "The data files in '" << verify_dir << "' differ from the converted scene" << std::endl; return 1; }
        std::cout << "Data files in '" << verify_dir << "' match the converted scene" << std::endl;
    }
    std::cout << "Scene was converted successfully: " << scene.num_tris() << " triangle(s), " << scene.materials.size()
              << " material(s), " << scene.lights.size() << " light(s), " << scene.nodes.size() << " BVH node(s)" << std::endl;
    return 0;
}

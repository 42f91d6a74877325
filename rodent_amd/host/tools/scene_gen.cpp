// scene_gen -- writes the procedural atrium (OBJ + MTL).  New tool: the reference
// ships its test scene (Sponza) as binary blobs that are absent from the checkout.
#include <cstring>
#include <fstream>
#include <iostream>

#include "../atrium.h"

int main(int argc, char** argv) {
    if (argc < 3 || strcmp(argv[1], "atrium")) {
        std::cerr << "Usage: scene_gen atrium out.obj [seed]" << std::endl;
        return 1;
    }
    const uint64_t seed = argc > 3 ? strtoull(argv[3], nullptr, 10) : 1;
    rodent::TriMesh mesh;
    rodent::generate_atrium(mesh, seed);
    const std::string obj = argv[2];
    if (!rodent::save_obj(obj, mesh)) { std::cerr << "Cannot write " << obj << std::endl; return 1; }
    auto slash = obj.find_last_of('/');
    std::ofstream mtl((slash == std::string::npos ? std::string() : obj.substr(0, slash + 1)) + "atrium.mtl");
    mtl << rodent::atrium_mtl_text();
    std::cout << "atrium: " << mesh.num_tris() << " triangle(s), " << mesh.vertices.size() << " vertices" << std::endl;
    return 0;
}

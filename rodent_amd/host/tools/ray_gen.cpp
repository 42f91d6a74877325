// ray_gen -- ray dump generator, same modes and argument order as the reference
// tool (tools/ray_gen/ray_gen.cpp:113-226).
//   primary: pinhole camera, rows top to bottom, pixel centres, UNNORMALISED
//            directions dir + kx*right*tan(fov/2) + ky*up*(h/w)*tan(fov/2)   (:20-58)
//   shadow : rays from a point light towards the hit points of a previous pass (:60-85)
//   random : segments between two uniform points in the scene bounds          (:87-111);
//            bounds = union of the BVH4 root's child boxes                    (:134-144)
// The reference draws random numbers with std::mt19937_64 +
// std::uniform_real_distribution<float>, whose output is implementation
// defined; this tool uses splitmix64 so that dumps are identical everywhere.
#include <cmath>
#include <cstring>
#include <iostream>
#include <memory>

#include "../formats.h"
#include "../vec.h"

using namespace rodent;

static void usage() {
    std::cout << "Usage: ray_gen mode arguments output\n"
                 "Available modes:\n"
                 "  primary eye-x eye-y eye-z dir-x dir-y dir-z up-x up-y up-z fov width height\n"
                 "  shadow  light-x light-y light-z ray-file fbuf-file width height\n"
                 "  random  bvh-file ray-count seed\n";
}

static void put(FILE* f, V3 o, V3 d) { const float r[6] = {o.x, o.y, o.z, d.x, d.y, d.z}; fwrite(r, 4, 6, f); }

struct SplitMix { uint64_t s; uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
                  float uni() { return (float)(next() >> 40) * (1.0f / 16777216.0f); } };

int main(int argc, char** argv) {
    if (argc < 2) { std::cerr << "Not enough arguments" << std::endl; return 1; }
    auto f = [&](int i) { return strtof(argv[i], nullptr); };
    if (!strcmp(argv[1], "-h") || !strcmp(argv[1], "--help")) { usage(); return 0; }
    if (!strcmp(argv[1], "primary")) {
        if (argc != 15) { std::cerr << "Incorrect number of arguments in primary mode" << std::endl; return 1; }
        const V3 eye(f(2), f(3), f(4)), dir_in(f(5), f(6), f(7)), up_in(f(8), f(9), f(10));
        const float fov = f(11);
        const int w = (int)strtol(argv[12], nullptr, 10), h = (int)strtol(argv[13], nullptr, 10);
        const V3 dir = normalize(dir_in);
        V3 right = normalize(cross(dir_in, up_in));
        V3 up = normalize(cross(right, dir_in));
        const float scale = (float)std::tan(fov * (M_PI / 360.0f));
        right = right * scale; up = up * (((float)h / (float)w) * scale);
        FILE* os = fopen(argv[14], "wb");
        if (!os) { std::cerr << "Cannot create output file" << std::endl; return 1; }
        const float sx = 2.0f / w, sy = 2.0f / h;
        for (int i = h - 1; i >= 0; i--)
            for (int j = 0; j < w; j++) {
                const float kx = sx * (j + 0.5f) - 1.0f, ky = sy * (i + 0.5f) - 1.0f;
                put(os, eye, dir + kx * right + ky * up);
            }
        fclose(os);
    } else if (!strcmp(argv[1], "shadow")) {
        if (argc != 10) { std::cerr << "Incorrect number of arguments in shadow mode" << std::endl; return 1; }
        const V3 light(f(2), f(3), f(4));
        std::vector<Ray1> rays;
        if (!load_rays(argv[5], 0.0f, 1.0f, rays)) { std::cerr << "Cannot load rays" << std::endl; return 1; }
        std::vector<float> t(rays.size());
        FILE* fb = fopen(argv[6], "rb");
        if (!fb || fread(t.data(), 4, t.size(), fb) != t.size()) { std::cerr << "Cannot load result of traversal" << std::endl; return 1; }
        fclose(fb);
        FILE* os = fopen(argv[9], "wb");
        if (!os) { std::cerr << "Cannot create output file" << std::endl; return 1; }
        for (size_t i = 0; i < rays.size(); i++) {
            const V3 o(rays[i].org[0], rays[i].org[1], rays[i].org[2]), d(rays[i].dir[0], rays[i].dir[1], rays[i].dir[2]);
            put(os, light, (o + t[i] * d) - light);
        }
        fclose(os);
    } else if (!strcmp(argv[1], "random")) {
        if (argc != 6) { std::cerr << "Incorrect number of arguments in random mode" << std::endl; return 1; }
        std::vector<Node4> nodes; std::vector<Tri4> tris;
        if (!load_bvh(argv[2], BvhType::BVH4_TRI4, nodes, tris) || nodes.empty()) {
            std::cerr << "Cannot extract scene bounds" << std::endl; return 1; }
        Box b;
        for (int i = 0; i < 4; i++) {
            b.lo = vmin(b.lo, V3(nodes[0].bounds[0][i], nodes[0].bounds[2][i], nodes[0].bounds[4][i]));
            b.hi = vmax(b.hi, V3(nodes[0].bounds[1][i], nodes[0].bounds[3][i], nodes[0].bounds[5][i]));
        }
        const long count = strtol(argv[3], nullptr, 10);
        SplitMix gen{(uint64_t)strtol(argv[4], nullptr, 10)};
        FILE* os = fopen(argv[5], "wb");
        if (!os) { std::cerr << "Cannot create output file" << std::endl; return 1; }
        const V3 ext = b.hi - b.lo;
        for (long i = 0; i < count; i++) {
            const float a0 = gen.uni(), a1 = gen.uni(), a2 = gen.uni(), b0 = gen.uni(), b1 = gen.uni(), b2 = gen.uni();
            const V3 p1 = b.lo + ext * V3(a0, a1, a2), p2 = b.lo + ext * V3(b0, b1, b2);
            put(os, p1, p2 - p1);
        }
        fclose(os);
    } else { std::cerr << "Unknown mode" << std::endl; return 1; }
    return 0;
}

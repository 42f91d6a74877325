// scene_builder.h -- what the procedural scene generators share (atrium.cpp, stress_scenes.cpp): a seedable generator and a mesh builder.
#pragma once
#include <cmath>
#include <cstdint>
#include "mesh.h"

namespace rodent {
namespace {

struct Rng {                                   // splitmix64: portable, seedable
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0x1234567ull) {}
    uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
    float uni() { return (float)((next() >> 40) * (1.0 / 16777216.0)); }           // [0,1)
    float range(float a, float b) { return a + (b - a) * uni(); }
};

const double kPi = 3.14159265358979323846;

struct Builder {
    TriMesh& m;
    int mat = 0;
    // every parametric patch is tessellated detail x detail times finer (the "gallery" scene: the atrium at detail 4)
    int detail = 1;
    uint32_t vert(V3 p) { m.vertices.push_back(p); return (uint32_t)m.vertices.size() - 1; }
    void tri(uint32_t a, uint32_t b, uint32_t c) { m.indices.insert(m.indices.end(), {a, b, c, (uint32_t)mat}); }
    void quad(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { tri(a, b, c); tri(a, c, d); }

    // Parametric grid patch: f(u,v) with nu x nv cells
    template <typename F> void patch(int nu, int nv, F f, bool flip = false) {
        nu *= detail; nv *= detail;
        const uint32_t base = (uint32_t)m.vertices.size();
        for (int j = 0; j <= nv; j++) for (int i = 0; i <= nu; i++) vert(f((float)i / nu, (float)j / nv));
        for (int j = 0; j < nv; j++) for (int i = 0; i < nu; i++) {
            const uint32_t a = base + j * (nu + 1) + i, b = a + 1, c = a + nu + 2, d = a + nu + 1;
            if (flip) quad(a, d, c, b); else quad(a, b, c, d);
        }
    }
    void box(V3 lo, V3 hi) {
        const uint32_t v[8] = {vert({lo.x, lo.y, lo.z}), vert({hi.x, lo.y, lo.z}), vert({hi.x, hi.y, lo.z}), vert({lo.x, hi.y, lo.z}),
                               vert({lo.x, lo.y, hi.z}), vert({hi.x, lo.y, hi.z}), vert({hi.x, hi.y, hi.z}), vert({lo.x, hi.y, hi.z})};
        quad(v[0], v[3], v[2], v[1]); quad(v[4], v[5], v[6], v[7]); quad(v[0], v[1], v[5], v[4]);
        quad(v[3], v[7], v[6], v[2]); quad(v[0], v[4], v[7], v[3]); quad(v[1], v[2], v[6], v[5]);
    }
    // Surface of revolution around the vertical axis through c: radius(t), height(t), t in [0,1]
    template <typename R, typename H> void lathe(V3 c, int seg, int rings, R radius, H height) {
        patch(seg, rings, [&](float u, float v) {
            const double a = 2 * kPi * u; const float r = radius(v);
            return V3(c.x + r * (float)std::cos(a), c.y + height(v), c.z + r * (float)std::sin(a));
        });
    }
};

} // namespace
} // namespace rodent

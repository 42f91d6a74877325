// Procedural stand-ins for the other scene classes of the reference's benchmark suite (benchmarks/benchmark.py:16-21: sponza, crown,
// san-miguel, powerplant -- 262 K ... 12.7 M triangles; none of them is in the checkout).  Every tuning constant of rounds 1-4 was fitted
// on the atrium (Sponza class); these scenes exist so that the defaults are measured on trees of another character:
//   "crown"   one dense organic surface: a knotted tube with a carved, bumpy skin, 4.2 M small triangles of similar size at detail 4 --
//             many leaves per unit of surface, rays graze a deep, regular tree (the reference's crown.obj, san-miguel's foliage);
//   "plant"   a hall full of long thin triangles: pipe runs (eight-sided prisms hundreds of radii long, a third of them diagonal), gratings
//             of thin bars, cable strips, a few tanks -- 1.9 M triangles at detail 4 whose boxes are mostly empty: the case spatial
//             splits exist for, and the case where a BVH gets deep and stacks get long (the reference's powerplant.obj);
//   the "gallery" is generate_atrium at detail 4 (atrium.cpp): the architectural class at 4.2 M triangles.
// Seeded (splitmix64), regenerable, never shipped: data/ is rebuilt on the box that needs it (rodent_amd/scenes.py).
#include "atrium.h"

#include <cmath>

#include "scene_builder.h"

namespace rodent {
namespace {

// point of the (2, 3) torus knot, scale R
V3 knot(double t, float R) {
    const double p = 2.0, q = 3.0, r = 2.0 + std::cos(q * t);
    return V3((float)(r * std::cos(p * t)), (float)std::sin(q * t) * 1.2f, (float)(r * std::sin(p * t))) * R;
}

void pipe(Builder& b, V3 a, V3 c, float radius, int sides) {
    const V3 axis = normalize(c - a);
    const V3 helper = std::fabs(axis.y) < 0.9f ? V3(0, 1, 0) : V3(1, 0, 0);
    const V3 u = normalize(cross(axis, helper)), v = cross(axis, u);
    const uint32_t base = (uint32_t)b.m.vertices.size();
    for (int end = 0; end < 2; end++)
        for (int k = 0; k < sides; k++) {
            const double ang = 2 * kPi * k / sides;
            b.vert((end ? c : a) + u * (radius * (float)std::cos(ang)) + v * (radius * (float)std::sin(ang)));
        }
    // ONE segment along the whole run: two triangles per side, as long as the pipe
    for (int k = 0; k < sides; k++) {
        const uint32_t k1 = (uint32_t)((k + 1) % sides);
        b.quad(base + k, base + k1, base + sides + k1, base + sides + k);
    }
}

} // namespace

void generate_crown(TriMesh& mesh, uint64_t seed, int detail) {
    mesh = TriMesh();
    mesh.material_names = {"", "gold", "floor"};
    mesh.mtl_libs = {"atrium.mtl"};
    Builder b{mesh};
    Rng rng(seed);
    if (detail < 1) detail = 1;
    const float R = 260.0f, tube = 150.0f;
    const float p1 = rng.range(0, 6.28f), p2 = rng.range(0, 6.28f), p3 = rng.range(0, 6.28f);
    b.mat = 1;
    // the skin: 1024 x 128 cells per unit of detail^2 -> 2 x 1024 x 128 x 16 = 4.19 M triangles at detail 4
    b.detail = detail;
    b.patch(1024, 128, [&](float s, float th) {
        const double t = 2 * kPi * s, a = 2 * kPi * th;
        const V3 c = knot(t, R), c1 = knot(t + 1e-3, R);
        const V3 T = normalize(c1 - c);
        const V3 B = normalize(cross(T, c)), N = cross(B, T);
        // carved bands, facets and a fine grain: the radius varies by a third
        const float bands = 0.16f * (float)std::sin(36 * t + p1) * (float)std::sin(5 * a + p2);
        const float facets = 0.07f * (float)std::fabs(std::sin(18 * a + 9 * t));
        const float grain = 0.03f * (float)std::sin(211 * t + p3) * (float)std::sin(67 * a);
        const float r = tube * (0.78f + bands + facets + grain);
        return c + N * (r * (float)std::cos(a)) + B * (r * (float)std::sin(a));
    });
    // the table it stands on: two large triangles under everything
    b.detail = 1;
    b.mat = 2;
    b.patch(1, 1, [&](float u, float v) { return V3(-1400 + 2800 * u, -1.2f * R - tube - 20, -1400 + 2800 * v); }, true);
}

void generate_plant(TriMesh& mesh, uint64_t seed, int detail) {
    mesh = TriMesh();
    mesh.material_names = {"", "steel", "floor", "tank"};
    mesh.mtl_libs = {"atrium.mtl"};
    Builder b{mesh};
    Rng rng(seed);
    if (detail < 1) detail = 1;
    const float X = 2000, Y = 1400, Z = 1250;                        // half extents in x / z, full height
    // hall: a handful of large triangles
    b.mat = 2;
    b.patch(2, 2, [&](float u, float v) { return V3(-X + 2 * X * u, 0, -Z + 2 * Z * v); }, true);
    b.patch(2, 1, [&](float u, float v) { return V3(-X + 2 * X * u, Y * v, -Z); }, true);
    b.patch(2, 1, [&](float u, float v) { return V3(-X + 2 * X * u, Y * v, Z); });
    b.patch(2, 1, [&](float u, float v) { return V3(-X, Y * v, -Z + 2 * Z * u); });
    b.patch(2, 1, [&](float u, float v) { return V3(X, Y * v, -Z + 2 * Z * u); }, true);
    b.patch(2, 2, [&](float u, float v) { return V3(-X + 2 * X * u, Y, -Z + 2 * Z * v); });
    b.mat = 1;
    // pipe racks: bundles of parallel runs along x and z on several levels, and free diagonal runs between them
    const int pipes = 5000 * detail;
    for (int i = 0; i < pipes; i++) {
        const float r = rng.range(1.5f, 9.0f);
        const int kind = (int)(rng.uni() * 10);
        V3 a, c;
        if (kind < 4) {            // along x
            const float y = 120.0f + 160.0f * (int)(rng.uni() * 8) + rng.range(-25, 25), z = rng.range(-Z + 40, Z - 40);
            const float x0 = rng.range(-X + 20, X - 600), len = rng.range(500, 3200);
            a = V3(x0, y, z); c = V3(std::min(X - 20, x0 + len), y, z);
        } else if (kind < 7) {     // along z
            const float y = 180.0f + 160.0f * (int)(rng.uni() * 8) + rng.range(-25, 25), x = rng.range(-X + 40, X - 40);
            const float z0 = rng.range(-Z + 20, Z - 500), len = rng.range(400, 2200);
            a = V3(x, y, z0); c = V3(x, y, std::min(Z - 20, z0 + len));
        } else if (kind < 8) {     // risers
            const float x = rng.range(-X + 40, X - 40), z = rng.range(-Z + 40, Z - 40), y0 = rng.range(0, 500);
            a = V3(x, y0, z); c = V3(x, std::min(Y - 10, y0 + rng.range(300, 1300)), z);
        } else {                   // diagonals: the long thin triangles whose boxes are nearly empty
            a = V3(rng.range(-X + 30, X - 30), rng.range(20, Y - 20), rng.range(-Z + 30, Z - 30));
            const V3 d = normalize(V3(rng.range(-1, 1), rng.range(-0.5f, 0.5f), rng.range(-1, 1)));
            c = a + d * rng.range(400, 2600);
            c = V3(std::min(X - 10, std::max(-X + 10, c.x)), std::min(Y - 10, std::max(10.0f, c.y)),
                std::min(Z - 10, std::max(-Z + 10, c.z)));
        }
        if (length(c - a) > 1.0f) pipe(b, a, c, r, 8);
    }
    // gratings: walkways of thin bars (boxes 2 x 3 units in section, hundreds long)
    const int gratings = 375 * detail;
    for (int g = 0; g < gratings; g++) {
        const float y = 100.0f + 160.0f * (int)(rng.uni() * 8), x0 = rng.range(-X + 50, X - 450), z0 = rng.range(-Z + 50, Z - 250);
        const bool along_x = rng.uni() < 0.5f;
        const float len = rng.range(250, 400), pitch = rng.range(2.5f, 4.0f);
        for (int k = 0; k < 64; k++) {
            const float o = k * pitch;
            if (along_x) b.box(V3(x0, y, z0 + o), V3(x0 + len, y + 3, z0 + o + 1.2f));
            else         b.box(V3(x0 + o, y, z0), V3(x0 + o + 1.2f, y + 3, z0 + len));
        }
    }
    // cables: sagging strips one unit wide, 64 segments each
    const int cables = 750 * detail;
    for (int c = 0; c < cables; c++) {
        const V3 p0(rng.range(-X + 30, X - 30), rng.range(500, Y - 30), rng.range(-Z + 30, Z - 30));
        const V3 p1 = p0 + V3(rng.range(-900, 900), rng.range(-150, 150), rng.range(-900, 900));
        const float sag = rng.range(30, 160);
        const V3 side = normalize(cross(p1 - p0, V3(0, 1, 0))) * 0.5f;
        b.patch(64, 1, [&](float u, float v) {
            return p0 + (p1 - p0) * u + V3(0, -sag * 4 * u * (1 - u), 0) + side * (2 * v - 1);
        });
    }
    // tanks: finely tessellated bodies among the pipes
    b.mat = 3;
    b.detail = detail;
    for (int t = 0; t < 6; t++) {
        const V3 base(-1500.0f + 600.0f * t, 0, (t & 1) ? 500.0f : -500.0f);
        const float r = rng.range(140, 220), h = rng.range(400, 800);
        b.lathe(base, 48, 24, [&](float v) { return r * (float)std::sqrt(std::max(0.0, 1.0 - std::pow(2.0 * v - 1.0, 8.0))) + 1.0f; },
            [&](float v) { return v * h; });
    }
    b.detail = 1;
}

} // namespace rodent

// hybrid_baseline.cpp -- CPU BASELINE (not the parity oracle): Rodent's CPU hybrid traversal restated
// with 8-wide SIMD (GCC vector extensions, built -O3 -march=x86-64-v3 = AVX2 + FMA).  TEST INFRASTRUCTURE:
// only bench.py's cpu_baseline leg and tests/ use it.  The reference's CPU kernels are Impala vectorised
// by RV and cannot be built here (AnyDSL absent), so the "Rodent CPU hybrid path" number reported beside
// the GPU figure is this restatement, labelled kind = "port".
//
// Restated (src/traversal/mapping_cpu.impala):
//   :388-402  cpu_traverse_hybrid      packets of 8 rays, one core per call (the reference loop is sequential)
//   :259-384  cpu_traverse_hybrid_helper   shared stack with per-lane entry distances; a child is pushed if any
//             lane hits it (unordered slab test, integer min/max :123-133), on top if some lane sees it closer than
//             the current top; when <= 6 lanes are active (ray8 x bvh8, :267-272) each live lane runs the
//             single-ray kernel from the current node (:305-321)
//   :138-256  cpu_traverse_single_helper   SIMD over the 8 children (ordered boxes), Batcher sort of >= 3 pushes
// Built with -ffast-math-free but contracting flags: results agree with the oracle up to fused-multiply-add
// rounding (checked in tests to 1e-4 relative, ids exact off ties), they are not bit-pinned.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

struct Node8 { float bounds[6][8]; int32_t child[8]; int32_t pad[8]; };
struct Tri4  { float v0[3][4], e1[3][4], e2[3][4], n[3][4]; int32_t prim_id[4], geom_id[4]; };
struct Ray1  { float org[3]; float tmin; float dir[3]; float tmax; };
struct Hit1  { int32_t tri_id; float t, u, v; };

typedef float   v8f __attribute__((vector_size(32)));
typedef int32_t v8i __attribute__((vector_size(32)));

static const float kFltMax = 3.4028234664e+38f;

static inline v8f splat(float x) { return v8f{x, x, x, x, x, x, x, x}; }
static inline v8f imin(v8f a, v8f b) { return ((v8i)a < (v8i)b) ? a : b; }      // mapping_cpu.impala:123-133
static inline v8f imax(v8f a, v8f b) { return ((v8i)a > (v8i)b) ? a : b; }
static inline int movemask(v8i m) { return __builtin_ia32_movmskps256((v8f)m); }
static inline v8f loadu(const float* p) { v8f r; std::memcpy(&r, p, 32); return r; }
static inline float prodsign1(float x, float y) { uint32_t a, b; std::memcpy(&a, &x, 4); std::memcpy(&b, &y, 4); a ^= b & 0x80000000u;
    std::memcpy(&x, &a, 4); return x; }
static inline float safe_rcp1(float x) { return ((x > 0 ? x : -x) < 1e-8f) ? prodsign1(kFltMax, x) : 1.0f / x; }

struct Ray1X { float o[3], d[3], id[3], io[3], tmin, tmax; int octant; };
static inline Ray1X make_ray1(const float* o, const float* d, float tmin, float tmax) {
    Ray1X r;
    for (int k = 0; k < 3; k++) { r.o[k] = o[k]; r.d[k] = d[k]; r.id[k] = safe_rcp1(d[k]); r.io[k] = -(o[k] * r.id[k]); }
    r.tmin = tmin; r.tmax = tmax; r.octant = (d[0] > 0 ? 1 : 0) | (d[1] > 0 ? 2 : 0) | (d[2] > 0 ? 4 : 0);
    return r;
}

struct Ent1 { int32_t node; float tmin; };

// one triangle, scalar (intersection.impala:164-192)
static inline bool tri_scalar(const Ray1X& r, const Tri4& P, int k, float& t, float& u, float& v) {
    const float cx = P.v0[0][k] - r.o[0], cy = P.v0[1][k] - r.o[1], cz = P.v0[2][k] - r.o[2];
    const float rx = r.d[1] * cz - r.d[2] * cy, ry = r.d[2] * cx - r.d[0] * cz, rz = r.d[0] * cy - r.d[1] * cx;
    const float nx = P.n[0][k], ny = P.n[1][k], nz = P.n[2][k];
    const float det = nx * r.d[0] + ny * r.d[1] + nz * r.d[2], ad = det < 0 ? -det : det;
    const float uu = prodsign1(rx * P.e2[0][k] + ry * P.e2[1][k] + rz * P.e2[2][k], det);
    const float vv = prodsign1(rx * P.e1[0][k] + ry * P.e1[1][k] + rz * P.e1[2][k], det);
    if (!(uu >= 0) || !(vv >= 0) || !(uu + vv <= ad)) return false;
    const float tt = prodsign1(cx * nx + cy * ny + cz * nz, det);
    if (!(ad != 0) || !(tt >= ad * r.tmin) || !(tt <= ad * r.tmax)) return false;
    const float inv = 1.0f / ad; t = tt * inv; u = uu * inv; v = vv * inv;
    return true;
}

static void sort_desc(Ent1* a, int n) {      // what the sorting networks compute: farthest first (sort.impala:3-66)
    for (int i = 1; i < n; i++) { Ent1 x = a[i]; int j = i - 1; while (j >= 0 && a[j].tmin < x.tmin) { a[j + 1] = a[j]; j--;
        } a[j + 1] = x; }
}

// mapping_cpu.impala:138-256, SIMD lanes = children
static bool single_ray(const Node8* nodes, const Tri4* tris, Ray1X ray, bool any_hit, int32_t root, Hit1& hit) {
    Ent1 mem[160]; int ptr = -1; Ent1 top{0, kFltMax};
    mem[++ptr] = top; top = Ent1{root, ray.tmin};
    bool found = false;
    const v8f idx = splat(ray.id[0]), idy = splat(ray.id[1]), idz = splat(ray.id[2]);
    const v8f iox = splat(ray.io[0]), ioy = splat(ray.io[1]), ioz = splat(ray.io[2]);
    const int ox = ray.octant & 1, oy = (ray.octant >> 1) & 1, oz = (ray.octant >> 2) & 1;
    for (;;) {
        if (top.node == 0) break;
        if (!any_hit && top.tmin > ray.tmax) { top = mem[ptr--]; continue; }
        bool restart = false;
        while (top.node > 0) {
            const Node8& nd = nodes[top.node - 1];
            top = mem[ptr--];
            const v8f t0x = loadu(nd.bounds[ox ? 0 : 1]) * idx + iox, t1x = loadu(nd.bounds[ox ? 1 : 0]) * idx + iox;
            const v8f t0y = loadu(nd.bounds[oy ? 2 : 3]) * idy + ioy, t1y = loadu(nd.bounds[oy ? 3 : 2]) * idy + ioy;
            const v8f t0z = loadu(nd.bounds[oz ? 4 : 5]) * idz + ioz, t1z = loadu(nd.bounds[oz ? 5 : 4]) * idz + ioz;
            const v8f te = imax(imax(t0x, t0y), imax(t0z, splat(ray.tmin)));
            const v8f tx = imin(imin(t1x, t1y), imin(t1z, splat(ray.tmax)));
            int mask = ~movemask((v8i)tx < (v8i)te) & 0xFF;
            if (mask == 0) { if (any_hit) continue; restart = true; break; }
            float tes[8]; std::memcpy(tes, &te, 32);
            int num = 0;
            while (mask) {
                const int k = __builtin_ctz(mask); mask &= mask - 1; num++;
                if (any_hit || tes[k] < top.tmin) { mem[++ptr] = top; top = Ent1{nd.child[k], tes[k]}; }
                else mem[++ptr] = Ent1{nd.child[k], tes[k]};
            }
            if (!any_hit && num >= 3) sort_desc(&mem[ptr - num + 1], num);
        }
        if (restart) continue;
        if (any_hit && top.node == 0) break;
        int32_t j = ~top.node; top = mem[ptr--];
        for (;;) {
            const Tri4& P = tris[j++];
            int bl = -1; float bt = 0, bu = 0, bv = 0;
            for (int k = 0; k < 4; k++) {
                if (P.prim_id[k] == -1) continue;
                float t, u, v;
                if (!tri_scalar(ray, P, k, t, u, v)) continue;
                if (any_hit) { if (bl < 0) { bl = k; bt = t; bu = u; bv = v; } }
                else if (bl < 0 || t < bt) { bl = k; bt = t; bu = u; bv = v; }
            }
            if (bl >= 0) { hit = Hit1{P.prim_id[bl] & 0x7FFFFFFF, bt, bu, bv}; found = true; if (any_hit) return true; ray.tmax = bt; }
            if (P.prim_id[3] < 0) break;
        }
    }
    return found;
}

struct EntP { int32_t node; v8f tmin; };

// mapping_cpu.impala:259-384 for one packet of 8 rays (SIMD lanes = rays)
static void hybrid_packet(const Node8* nodes, const Tri4* tris, const Ray1* rays, Hit1* hits, bool any_hit) {
    v8f ox, oy, oz, dx, dy, dz, idx, idy, idz, iox, ioy, ioz, tmin, tmax;
    int octant[8];
    for (int l = 0; l < 8; l++) {
        const Ray1X r = make_ray1(rays[l].org, rays[l].dir, rays[l].tmin, rays[l].tmax);
        ox[l] = r.o[0]; oy[l] = r.o[1]; oz[l] = r.o[2]; dx[l] = r.d[0]; dy[l] = r.d[1]; dz[l] = r.d[2];
        idx[l] = r.id[0]; idy[l] = r.id[1]; idz[l] = r.id[2]; iox[l] = r.io[0]; ioy[l] = r.io[1]; ioz[l] = r.io[2];
        tmin[l] = r.tmin; tmax[l] = r.tmax; octant[l] = r.octant;
        hits[l] = Hit1{-1, r.tmax, 0, 0};
    }
    int terminated = 0;                                  // bit per lane
    EntP mem[160]; int ptr = -1; EntP top{0, splat(kFltMax)};
    mem[++ptr] = top; top = EntP{1, tmin};
    const int threshold = 6;                             // ray8 x bvh8 (:267-272)
    for (;;) {
        // cull nodes; switch to single rays when SIMD utilisation is too low (:300-325)
        bool done = false;
        for (;;) {
            if (top.node == 0) { done = true; break; }
            const int mask = movemask(top.tmin <= tmax) & ~terminated & 0xFF;
            if (mask != 0) {
                if (__builtin_popcount(mask) <= threshold) {
                    int m = mask;
                    while (m) {
                        const int l = __builtin_ctz(m); m &= m - 1;
                        Ray1X r; r.o[0] = ox[l]; r.o[1] = oy[l]; r.o[2] = oz[l]; r.d[0] = dx[l]; r.d[1] = dy[l]; r.d[2] = dz[l];
                        r.id[0] = idx[l]; r.id[1] = idy[l]; r.id[2] = idz[l]; r.io[0] = iox[l]; r.io[1] = ioy[l]; r.io[2] = ioz[l];
                        r.tmin = tmin[l]; r.tmax = tmax[l]; r.octant = octant[l];
                        Hit1 h;
                        if (single_ray(nodes, tris, r, any_hit, top.node, h)) { hits[l] = h; if (!any_hit) tmax[l] = h.t;
                            else terminated |= 1 << l; }
                    }
                } else break;
            }
            top = mem[ptr--];
        }
        if (done) break;
        // inner nodes (:327-353)
        bool culled = false;
        while (top.node > 0) {
            const Node8& nd = nodes[top.node - 1];
            top = mem[ptr--];
            bool pushed = false;
            for (int i = 0; i < 8; i++) {
                const int32_t child = nd.child[i];
                if (child == 0) break;
                const v8f t0x = idx * splat(nd.bounds[0][i]) + iox, t1x = idx * splat(nd.bounds[1][i]) + iox;
                const v8f t0y = idy * splat(nd.bounds[2][i]) + ioy, t1y = idy * splat(nd.bounds[3][i]) + ioy;
                const v8f t0z = idz * splat(nd.bounds[4][i]) + ioz, t1z = idz * splat(nd.bounds[5][i]) + ioz;
                const v8f te = imax(imax(imin(t0x, t1x), imin(t0y, t1y)), imax(imin(t0z, t1z), tmin));
                const v8f tx = imin(imin(imax(t0x, t1x), imax(t0y, t1y)), imin(imax(t0z, t1z), tmax));
                const v8i miss = (v8i)tx < (v8i)te;
                if ((~movemask(miss) & 0xFF) != 0) {
                    const v8f thit = miss ? splat(kFltMax) : te;
                    if (any_hit || movemask(top.tmin > thit)) { mem[++ptr] = top; top = EntP{child, thit}; }
                    else mem[++ptr] = EntP{child, thit};
                    pushed = true;
                }
            }
            if (!pushed) { culled = true; break; }
        }
        if (culled) continue;
        if (top.node < 0) {                              // leaf (:355-381): masked scalar test per lane
            int active = movemask(top.tmin <= tmax) & ~terminated & 0xFF;
            int32_t j = ~top.node; top = mem[ptr--];
            for (;;) {
                const Tri4& P = tris[j++];
                for (int k = 0; k < 4; k++) {
                    if (P.prim_id[k] == -1) break;
                    int m = active;
                    while (m) {
                        const int l = __builtin_ctz(m); m &= m - 1;
                        Ray1X r; r.o[0] = ox[l]; r.o[1] = oy[l]; r.o[2] = oz[l]; r.d[0] = dx[l]; r.d[1] = dy[l]; r.d[2] = dz[l];
                        r.tmin = tmin[l]; r.tmax = tmax[l];
                        float t, u, v;
                        if (tri_scalar(r, P, k, t, u, v)) {
                            hits[l] = Hit1{P.prim_id[k] & 0x7FFFFFFF, t, u, v}; tmax[l] = t;
                            if (any_hit) { terminated |= 1 << l; active &= ~(1 << l); }
                        }
                    }
                    if (any_hit && terminated == 0xFF) return;
                }
                if (P.prim_id[3] < 0) break;
            }
        }
    }
}

extern "C" {

// Traces n rays (tail beyond a multiple of 8 is dropped like load_rays.h:74) on `threads` host threads (the reference's bench loop is
// sequential, mapping_cpu.impala:397; the all-cores figure mirrors results_par.txt).  mode: 0 = hybrid ray8 x bvh8, 1 = single-ray bvh8.
void cpu_baseline_traverse(const Node8* nodes, const Tri4* tris, const Ray1* rays, Hit1* hits, int32_t n, int32_t any_hit, int32_t mode,
    int32_t threads) {
    const int packets = n / 8;
    if (threads < 1) threads = 1;
    auto work = [&](int p0, int p1) {
        for (int p = p0; p < p1; p++) {
            if (mode == 0) hybrid_packet(nodes, tris, rays + 8 * p, hits + 8 * p, any_hit != 0);
            else for (int l = 0; l < 8; l++) {
                const Ray1& r = rays[8 * p + l];
                Hit1 h{-1, r.tmax, 0, 0};
                single_ray(nodes, tris, make_ray1(r.org, r.dir, r.tmin, r.tmax), any_hit != 0, 1, h);
                hits[8 * p + l] = h;
            }
        }
    };
    if (threads == 1) { work(0, packets); return; }
    // dynamic chunks of 256 packets: ray cost varies a lot across the image (static ranges leave 7 of 8 threads idle)
    std::atomic<int> next{0};
    const int chunk = 256;
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++) pool.emplace_back([&] { for (;;) { const int p0 =
        next.fetch_add(chunk); if (p0 >= packets) break; work(p0, std::min(packets, p0 + chunk)); } });
    for (auto& th : pool) th.join();
}

// Timed form for bench.py's cpu_baseline: a PERSISTENT pool of `threads` threads runs `passes` passes over the same rays
// (one warm-up pass first); thread creation is outside the timed region, every pass starts and ends at a barrier and is
// timed on its own.  seconds[p] = wall time of pass p.  (The one-shot entry point above spawns its threads inside the call:
// at 256 threads that is a third of a 12 ms pass.)
void cpu_baseline_bench(const Node8* nodes, const Tri4* tris, const Ray1* rays, Hit1* hits, int32_t n, int32_t any_hit, int32_t mode,
    int32_t threads,
                        int32_t passes, double* seconds) {
    const int packets = n / 8, chunk = 128;
    if (threads < 1) threads = 1;
    std::atomic<int> next{0};
    std::mutex m; std::condition_variable cv; int arrived = 0, generation = 0;
    std::vector<std::chrono::steady_clock::time_point> tripped(2 * (passes + 1));
    // sleeping barrier (256 yield-spinning threads starve the workers); the last arriver resets the work
    auto barrier = [&]() {
        // counter and stamps the time: a pass lasts from the trip of its start barrier to the trip of its end barrier
        std::unique_lock<std::mutex> lock(m);
        const int gen = generation;
        if (++arrived == threads) { arrived = 0; next.store(0); tripped[generation] = std::chrono::steady_clock::now(); generation++;
            cv.notify_all(); }
        else cv.wait(lock, [&] { return generation != gen; });
    };
    auto one_pass = [&]() {
        for (;;) {
            const int p0 = next.fetch_add(chunk);
            if (p0 >= packets) break;
            for (int p = p0; p < std::min(packets, p0 + chunk); p++) {
                if (mode == 0) hybrid_packet(nodes, tris, rays + 8 * p, hits + 8 * p, any_hit != 0);
                else for (int l = 0; l < 8; l++) {
                    const Ray1& r = rays[8 * p + l];
                    Hit1 h{-1, r.tmax, 0, 0};
                    single_ray(nodes, tris, make_ray1(r.org, r.dir, r.tmin, r.tmax), any_hit != 0, 1, h);
                    hits[8 * p + l] = h;
                }
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; t++) pool.emplace_back([&] { for (int pass =
        0; pass < passes + 1; pass++) { barrier(); one_pass(); barrier(); } });
    for (int pass = 0; pass < passes + 1; pass++) { barrier(); one_pass(); barrier(); }      // the calling thread is worker 0
    for (auto& th : pool) th.join();
    for (int pass = 1; pass < passes + 1; pass++) seconds[pass - 1] =
        std::chrono::duration<double>(tripped[2 * pass + 1] - tripped[2 * pass]).count();
}

int32_t cpu_baseline_hardware_threads(void) { return (int32_t)std::thread::hardware_concurrency(); }

} // extern "C"

#include "cpu_wavefront.inc"        // the reference's CPU wavefront renderer on top of hybrid_packet (frame baseline)

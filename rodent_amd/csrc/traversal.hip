// traversal.hip -- BVH traversal kernels for MI355X (gfx950, wave64) and their C ABI.
//
// What this replaces in the reference (all Impala, compiled by AnyDSL):
//   src/traversal/mapping_gpu.impala:94-178   gpu_traverse_single_helper  (the hot loop)
//   src/traversal/mapping_gpu.impala:182-203  gpu_traverse_single         (launcher)
//   src/traversal/intersection.impala:88-99,164-208  ray setup, ray/box, ray/triangle
//   src/traversal/stack.impala:52-123         traversal stack
//   tools/bench_traversal/bench_traversal.impala:67-83,495-529  Ray1/Hit1 accessors, entry points
//
// Design notes (CDNA4):
//  * one ray per lane, 64-lane workgroups (= one wavefront, no barriers needed);
//  * the traversal stack lives in LDS, laid out [entry][lane] so a wave's accesses
//    hit 64 consecutive dwords (bank = lane % 32 for ds_read/write_b32: conflict
//    free whatever each lane's depth is); entries beyond the LDS depth spill to
//    scratch, entries beyond 64 (the reference's capacity, stack.impala:53) raise
//    a device-side error flag that the host turns into abort();
//  * nodes and triangles are fetched with 16-byte loads (global_load_dwordx4);
//  * no MFMA: this is branchy scalar fp32 work;
//  * arithmetic is written out with explicit fmaf() and compiled with
//    -ffp-contract=off so results are bit-identical to the CPU parity oracle.
//
// Kernel variants ("mappings") are selected at run time; see kVariants below.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>

#include "rodent_traversal.h"

#define HIP_CHECK(expr)                                                                       \
    do {                                                                                      \
        hipError_t err_ = (expr);                                                             \
        if (err_ != hipSuccess) {                                                             \
            fprintf(stderr, "rodent_hip: %s failed: %s (%s:%d)\n", #expr,                     \
                    hipGetErrorString(err_), __FILE__, __LINE__);                             \
            abort();                                                                          \
        }                                                                                     \
    } while (0)

namespace {

constexpr int   kWave = 64;
constexpr int   kStackCap = 64;               // stack.impala:53-54
constexpr float kFltMax = 3.4028234664e+38f;  // common.impala:4

// ---------------------------------------------------------------------------------------------
// Ray / box / triangle arithmetic (same operation sequence as oracle/traversal_oracle.c)
// ---------------------------------------------------------------------------------------------
struct RayX {
    float ox, oy, oz, dx, dy, dz, idx, idy, idz, iox, ioy, ioz, tmin, tmax;
};

__device__ __forceinline__ float prodsign(float x, float y) {        // common.impala:78-80
    return __int_as_float(__float_as_int(x) ^ (__float_as_int(y) & (int)0x80000000u));
}
__device__ __forceinline__ float safe_rcp(float x) {                 // common.impala:82-85
    return (fabsf(x) < 1e-8f) ? prodsign(kFltMax, x) : 1.0f / x;
}
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
    return fmaf(az, bz, fmaf(ay, by, ax * bx));
}
__device__ __forceinline__ float cross_x(float ax, float ay, float az, float bx, float by, float bz) { return fmaf(ay, bz, -(az * by)); }
__device__ __forceinline__ float cross_y(float ax, float ay, float az, float bx, float by, float bz) { return fmaf(az, bx, -(ax * bz)); }
__device__ __forceinline__ float cross_z(float ax, float ay, float az, float bx, float by, float bz) { return fmaf(ax, by, -(ay * bx)); }

// bench_traversal.impala:67-76 (two 16-byte loads) + intersection.impala:88-99
__device__ __forceinline__ RayX load_ray(const Ray1* rays, int i) {
    const float4* p = reinterpret_cast<const float4*>(rays + i);
    const float4 r0 = p[0], r1 = p[1];
    RayX r;
    r.ox = r0.x; r.oy = r0.y; r.oz = r0.z; r.tmin = r0.w;
    r.dx = r1.x; r.dy = r1.y; r.dz = r1.z; r.tmax = r1.w;
    r.idx = safe_rcp(r.dx); r.idy = safe_rcp(r.dy); r.idz = safe_rcp(r.dz);
    r.iox = -(r.ox * r.idx); r.ioy = -(r.oy * r.idy); r.ioz = -(r.oz * r.idz);
    return r;
}

// bench_traversal.impala:78-83 (one 16-byte store)
__device__ __forceinline__ void store_hit(Hit1* hits, int i, int id, float t, float u, float v) {
    *reinterpret_cast<float4*>(hits + i) = make_float4(__int_as_float(id), t, u, v);
}

// intersection.impala:194-208, unordered form, fminf/fmaxf like make_amdgpu_min_max
// (mapping_gpu.impala:87-89).  Returns tentry; hit iff tentry <= texit.
__device__ __forceinline__ bool slab(const RayX& r, float lox, float hix, float loy, float hiy, float loz, float hiz, float& tentry) {
    const float t0x = fmaf(r.idx, lox, r.iox), t1x = fmaf(r.idx, hix, r.iox);
    const float t0y = fmaf(r.idy, loy, r.ioy), t1y = fmaf(r.idy, hiy, r.ioy);
    const float t0z = fmaf(r.idz, loz, r.ioz), t1z = fmaf(r.idz, hiz, r.ioz);
    tentry = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), r.tmin));
    const float texit = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), r.tmax));
    return tentry <= texit;
}

// intersection.impala:164-192, no back-face culling
__device__ __forceinline__ bool intersect_tri(const RayX& r,
                                              float v0x, float v0y, float v0z, float e1x, float e1y, float e1z,
                                              float e2x, float e2y, float e2z, float nx, float ny, float nz,
                                              float& t_out, float& u_out, float& v_out) {
    const float cx = v0x - r.ox, cy = v0y - r.oy, cz = v0z - r.oz;
    const float rx = cross_x(r.dx, r.dy, r.dz, cx, cy, cz);
    const float ry = cross_y(r.dx, r.dy, r.dz, cx, cy, cz);
    const float rz = cross_z(r.dx, r.dy, r.dz, cx, cy, cz);
    const float det = dot3(nx, ny, nz, r.dx, r.dy, r.dz);
    const float abs_det = fabsf(det);
    const float u = prodsign(dot3(rx, ry, rz, e2x, e2y, e2z), det);
    const float v = prodsign(dot3(rx, ry, rz, e1x, e1y, e1z), det);
    if (!(u >= 0.0f) || !(v >= 0.0f) || !(u + v <= abs_det)) return false;
    const float t = prodsign(dot3(cx, cy, cz, nx, ny, nz), det);
    if (!(abs_det != 0.0f)) return false;
    if (!(t >= abs_det * r.tmin) || !(t <= abs_det * r.tmax)) return false;
    const float inv_det = 1.0f / abs_det;
    t_out = t * inv_det; u_out = u * inv_det; v_out = v * inv_det;
    return true;
}

// ---------------------------------------------------------------------------------------------
// Per-lane stack: first LDS_N entries in LDS ([entry][lane]), the rest in scratch.
// ---------------------------------------------------------------------------------------------
template <int LDS_N>
struct LaneStack {
    int* lds;                        // this lane's column
    int  spill[kStackCap - LDS_N];
    int* err;
    __device__ __forceinline__ void put(int e, int v) {
        if (e < LDS_N) lds[e * kWave] = v;
        else if (e < kStackCap) spill[e - LDS_N] = v;
        else *err = 1;
    }
    __device__ __forceinline__ int get(int e) const {
        return e < LDS_N ? lds[e * kWave] : spill[(e < kStackCap ? e : kStackCap - 1) - LDS_N];
    }
};

struct HitAcc { int id; float t, u, v; };

// One leaf of Tri1 records (mapping_gpu.impala:156-174).  Returns true when an
// any-hit query is finished.
template <bool ANY>
__device__ __forceinline__ bool leaf_tri1(const Tri1* __restrict__ tris, int first, RayX& ray, HitAcc& hit) {
    int j = first;
    for (;;) {
        const float4* p = reinterpret_cast<const float4*>(tris + j++);
        const float4 a = p[0], b = p[1], c = p[2];
        const int prim_id = __float_as_int(c.w);
        const float nx = cross_x(b.x, b.y, b.z, c.x, c.y, c.z);       // mapping_gpu.impala:57
        const float ny = cross_y(b.x, b.y, b.z, c.x, c.y, c.z);
        const float nz = cross_z(b.x, b.y, b.z, c.x, c.y, c.z);
        float t, u, v;
        if (intersect_tri(ray, a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z, nx, ny, nz, t, u, v)) {
            hit.id = prim_id & 0x7FFFFFFF; hit.t = t; hit.u = u; hit.v = v;
            ray.tmax = t;
            if (ANY) return true;
        }
        if (prim_id < 0) return false;                                // sentinel (:63,172)
    }
}

// One BVH2 node step (mapping_gpu.impala:107-134): returns the new top; pushes at most one entry.
template <typename Stack>
__device__ __forceinline__ int node2_step(const Node2* __restrict__ nodes, int top, const RayX& ray, Stack& st, int& ptr) {
    const float4* p = reinterpret_cast<const float4*>(nodes + (top - 1));
    const float4 b0 = p[0], b1 = p[1], b2 = p[2];
    const int4 ch = *reinterpret_cast<const int4*>(p + 3);
    float te0, te1;
    // Empty slots (child 0, bounds +inf/-inf) are never taken: the unordered min/max test would
    // turn the inverted box into an infinite one and push node id 0 (= "stack empty").
    const bool h0 = slab(ray, b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, te0) && ch.x != 0;
    const bool h1 = slab(ray, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, te1) && ch.y != 0;
    if (!h0 && !h1) { const int t = st.get(ptr); ptr--; return t; }
    if (h0 && h1) {
        const bool c0first = te0 < te1;                               // strict <  (:128-129)
        st.put(++ptr, c0first ? ch.y : ch.x);
        return c0first ? ch.x : ch.y;
    }
    return h0 ? ch.x : ch.y;
}

// ---------------------------------------------------------------------------------------------
// BVH2 / Tri1, variant 0: literal one-ray-per-lane while-while (the reference mapping).
// ---------------------------------------------------------------------------------------------
template <bool ANY, int LDS_N>
__global__ __launch_bounds__(kWave) void k_bvh2_lane(const Node2* __restrict__ nodes, const Tri1* __restrict__ tris,
                                                      const Ray1* __restrict__ rays, Hit1* __restrict__ hits, int n, int* err) {
    __shared__ int lds[LDS_N * kWave];
    const int i = blockIdx.x * kWave + threadIdx.x;
    if (i >= n) return;
    RayX ray = load_ray(rays, i);
    HitAcc hit{-1, ray.tmax, 0.0f, 0.0f};
    LaneStack<LDS_N> st; st.lds = lds + threadIdx.x; st.err = err;
    int ptr = 0, top = 1; st.put(0, 0);
    while (top != 0) {
        top = node2_step(nodes, top, ray, st, ptr);
        bool done = false;
        while (top < 0) {
            const int first = ~top; top = st.get(ptr); ptr--;
            if (leaf_tri1<ANY>(tris, first, ray, hit)) { done = true; break; }
        }
        if (ANY && done) break;
    }
    store_hit(hits, i, hit.id, hit.t, hit.u, hit.v);
}

// ---------------------------------------------------------------------------------------------
// BVH2 / Tri1, variant 1: persistent wavefronts with dynamic ray fetch.  Same per-ray
// visit order as variant 0 (results are bit-identical); lanes whose ray has finished
// pull the next ray index from a global counter instead of idling until the slowest
// lane of the wave is done.  One atomic per refill per wave (ballot + mbcnt).
// ---------------------------------------------------------------------------------------------
template <bool ANY, int LDS_N, int REFILL_BELOW>
__global__ __launch_bounds__(kWave) void k_bvh2_persistent(const Node2* __restrict__ nodes, const Tri1* __restrict__ tris,
                                                            const Ray1* __restrict__ rays, Hit1* __restrict__ hits, int n,
                                                            int* counter, int* err) {
    __shared__ int lds[LDS_N * kWave];
    LaneStack<LDS_N> st; st.lds = lds + threadIdx.x; st.err = err;
    RayX ray; HitAcc hit{-1, 0.0f, 0.0f, 0.0f};
    int ray_id = -1, top = 0, ptr = 0;
    bool exhausted = false;
    for (;;) {
        const bool idle = top == 0;
        if (idle && ray_id >= 0) { store_hit(hits, ray_id, hit.id, hit.t, hit.u, hit.v); ray_id = -1; }
        const unsigned long long idle_mask = __ballot(idle);
        if (idle_mask == ~0ull && exhausted) break;
        if (!exhausted && __popcll(idle_mask) >= (kWave - REFILL_BELOW + 1)) {
            const int want = __popcll(idle_mask);
            const int leader = __ffsll((long long)idle_mask) - 1;
            int base = 0;
            if ((int)threadIdx.x == leader) base = atomicAdd(counter, want);
            base = __shfl(base, leader);
            if (idle) {
                const int k = base + __popcll(idle_mask & ((1ull << threadIdx.x) - 1ull));
                if (k < n) {
                    ray_id = k; ray = load_ray(rays, k);
                    hit.id = -1; hit.t = ray.tmax; hit.u = 0.0f; hit.v = 0.0f;
                    ptr = 0; top = 1; st.put(0, 0);
                }
            }
            if (base + want >= n) exhausted = true;
            if (__ballot(top != 0) == 0ull) { if (exhausted) break; else continue; }
        }
        if (top != 0) {
            top = node2_step(nodes, top, ray, st, ptr);
            while (top < 0) {
                const int first = ~top; top = st.get(ptr); ptr--;
                if (leaf_tri1<ANY>(tris, first, ray, hit)) { top = 0; break; }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// BVH8 / Tri4, variant 0: the reference's GPU "general case" for arity != 2
// (mapping_gpu.impala:136-153) applied to the CPU layouts Node8 / Tri4
// (mapping_cpu.impala:3-22): pop; test all children; the nearest hit child goes on
// top, the others underneath in slot order; no distance stored, no culling on pop.
// Triangles of a Tri4 packet are tested one after the other (mapping_gpu.impala:160-169).
// ---------------------------------------------------------------------------------------------
template <bool ANY>
__device__ __forceinline__ bool leaf_tri4(const Tri4* __restrict__ tris, int first, RayX& ray, HitAcc& hit) {
    int j = first;
    for (;;) {
        const float4* p = reinterpret_cast<const float4*>(tris + j++);
        const int4 pid = *reinterpret_cast<const int4*>(p + 12);
        const int ids[4] = {pid.x, pid.y, pid.z, pid.w};
        float q[12][4];
#pragma unroll
        for (int r = 0; r < 12; r++) { const float4 x = p[r]; q[r][0] = x.x; q[r][1] = x.y; q[r][2] = x.z; q[r][3] = x.w; }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (ids[k] == -1) break;                                  // is_valid (mapping_cpu.impala:38)
            float t, u, v;
            if (intersect_tri(ray, q[0][k], q[1][k], q[2][k], q[3][k], q[4][k], q[5][k],
                              q[6][k], q[7][k], q[8][k], q[9][k], q[10][k], q[11][k], t, u, v)) {
                hit.id = ids[k] & 0x7FFFFFFF; hit.t = t; hit.u = u; hit.v = v;
                ray.tmax = t;
                if (ANY) return true;
            }
        }
        if (pid.w < 0) return false;                                  // is_last (mapping_cpu.impala:39)
    }
}

template <bool ANY, int LDS_N>
__global__ __launch_bounds__(kWave) void k_bvh8_lane(const Node8* __restrict__ nodes, const Tri4* __restrict__ tris,
                                                      const Ray1* __restrict__ rays, Hit1* __restrict__ hits, int n, int* err) {
    __shared__ int lds[LDS_N * kWave];
    const int i = blockIdx.x * kWave + threadIdx.x;
    if (i >= n) return;
    RayX ray = load_ray(rays, i);
    HitAcc hit{-1, ray.tmax, 0.0f, 0.0f};
    LaneStack<LDS_N> st; st.lds = lds + threadIdx.x; st.err = err;
    int ptr = 0, top = 1; st.put(0, 0);
    while (top != 0) {
        const float4* p = reinterpret_cast<const float4*>(nodes + (top - 1));
        top = st.get(ptr); ptr--;                                     // pop (:138)
        float tnear = ray.tmax;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const float4 lx = p[0 + half], hx = p[2 + half], ly = p[4 + half], hy = p[6 + half], lz = p[8 + half], hz = p[10 + half];
            const int4 ch = *reinterpret_cast<const int4*>(p + 12 + half);
            const float blx[4] = {lx.x, lx.y, lx.z, lx.w}, bhx[4] = {hx.x, hx.y, hx.z, hx.w};
            const float bly[4] = {ly.x, ly.y, ly.z, ly.w}, bhy[4] = {hy.x, hy.y, hy.z, hy.w};
            const float blz[4] = {lz.x, lz.y, lz.z, lz.w}, bhz[4] = {hz.x, hz.y, hz.z, hz.w};
            const int   chi[4] = {ch.x, ch.y, ch.z, ch.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float te;
                if (slab(ray, blx[k], bhx[k], bly[k], bhy[k], blz[k], bhz[k], te) && chi[k] != 0) {
                    if (ANY || te < tnear) { st.put(++ptr, top); top = chi[k]; tnear = te; }   // push      (:145-147)
                    else st.put(++ptr, chi[k]);                                                   // push_after (:149)
                }
            }
        }
        bool done = false;
        while (top < 0) {
            const int first = ~top; top = st.get(ptr); ptr--;
            if (leaf_tri4<ANY>(tris, first, ray, hit)) { done = true; break; }
        }
        if (ANY && done) break;
    }
    store_hit(hits, i, hit.id, hit.t, hit.u, hit.v);
}

// ---------------------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------------------
struct DeviceState {
    bool  init = false;
    int*  scratch = nullptr;    // [0] = ray counter for persistent kernels, [1] = error flag
    int   num_cus = 0;
};
DeviceState g_dev[16];
std::mutex  g_mutex;

DeviceState& device_state(int dev) {
    if (dev < 0 || dev >= 16) { fprintf(stderr, "rodent_hip: invalid device index %d\n", dev); abort(); }
    std::lock_guard<std::mutex> lock(g_mutex);
    DeviceState& s = g_dev[dev];
    if (!s.init) {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || dev >= count) {
            fprintf(stderr, "rodent_hip: no HIP device %d (%d visible)\n", dev, count); abort();
        }
        HIP_CHECK(hipSetDevice(dev));
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        s.num_cus = prop.multiProcessorCount;
        HIP_CHECK(hipMalloc(&s.scratch, 64 * sizeof(int)));
        HIP_CHECK(hipMemset(s.scratch, 0, 64 * sizeof(int)));
        s.init = true;
    }
    return s;
}

struct VariantInfo { const char* name; const char* kernel_closest; const char* kernel_any; };
const VariantInfo kVariants2[] = {
    {"lane",       "k_bvh2_lane<false,24>",            "k_bvh2_lane<true,24>"},
    {"persistent", "k_bvh2_persistent<false,24,64>",   "k_bvh2_persistent<true,24,64>"},
    {"persistent-lazy", "k_bvh2_persistent<false,24,40>", "k_bvh2_persistent<true,24,40>"},
};
const VariantInfo kVariants8[] = {
    {"lane", "k_bvh8_lane<false,24>", "k_bvh8_lane<true,24>"},
};
constexpr int kNumVariants2 = sizeof(kVariants2) / sizeof(kVariants2[0]);
constexpr int kNumVariants8 = sizeof(kVariants8) / sizeof(kVariants8[0]);

void check_error_flag(DeviceState& s, hipStream_t stream) {
    int flag = 0;
    HIP_CHECK(hipMemcpyAsync(&flag, s.scratch + 1, sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    if (flag) { fprintf(stderr, "rodent_hip: traversal stack overflow (more than %d entries)\n", kStackCap); abort(); }
}

template <bool ANY>
void launch_bvh2(DeviceState& s, const Node2* nodes, const Tri1* tris, const Ray1* rays, Hit1* hits, int n, int variant, hipStream_t stream) {
    if (n <= 0) return;
    int* err = s.scratch + 1;
    const int blocks = (n + kWave - 1) / kWave;
    switch (variant) {
        case 0: hipLaunchKernelGGL((k_bvh2_lane<ANY, 24>), dim3(blocks), dim3(kWave), 0, stream, nodes, tris, rays, hits, n, err); break;
        case 1:
        case 2: {
            HIP_CHECK(hipMemsetAsync(s.scratch, 0, sizeof(int), stream));
            const int grid = std::min(blocks, s.num_cus * 20);
            if (variant == 1) hipLaunchKernelGGL((k_bvh2_persistent<ANY, 24, 64>), dim3(grid), dim3(kWave), 0, stream, nodes, tris, rays, hits, n, s.scratch, err);
            else              hipLaunchKernelGGL((k_bvh2_persistent<ANY, 24, 40>), dim3(grid), dim3(kWave), 0, stream, nodes, tris, rays, hits, n, s.scratch, err);
            break;
        }
        default: fprintf(stderr, "rodent_hip: unknown BVH2 variant %d\n", variant); abort();
    }
    HIP_CHECK(hipGetLastError());
}

template <bool ANY>
void launch_bvh8(DeviceState& s, const Node8* nodes, const Tri4* tris, const Ray1* rays, Hit1* hits, int n, int variant, hipStream_t stream) {
    if (n <= 0) return;
    int* err = s.scratch + 1;
    const int blocks = (n + kWave - 1) / kWave;
    switch (variant) {
        case 0: hipLaunchKernelGGL((k_bvh8_lane<ANY, 24>), dim3(blocks), dim3(kWave), 0, stream, nodes, tris, rays, hits, n, err); break;
        default: fprintf(stderr, "rodent_hip: unknown BVH8 variant %d\n", variant); abort();
    }
    HIP_CHECK(hipGetLastError());
}

int default_variant(int width) {
    const char* e = getenv(width == 2 ? "RODENT_HIP_BVH2_VARIANT" : "RODENT_HIP_BVH8_VARIANT");
    return e ? atoi(e) : 0;
}

} // namespace

extern "C" {

void hip_traverse_bvh2_tri1_async(int32_t dev, const Node2* nodes, const Tri1* tris, const Ray1* rays, Hit1* hits,
                                  int32_t num_rays, int32_t any_hit, int32_t variant, void* stream) {
    DeviceState& s = device_state(dev);
    HIP_CHECK(hipSetDevice(dev));
    if (any_hit) launch_bvh2<true>(s, nodes, tris, rays, hits, num_rays, variant, (hipStream_t)stream);
    else         launch_bvh2<false>(s, nodes, tris, rays, hits, num_rays, variant, (hipStream_t)stream);
}

void hip_traverse_bvh8_tri4_async(int32_t dev, const Node8* nodes, const Tri4* tris, const Ray1* rays, Hit1* hits,
                                  int32_t num_rays, int32_t any_hit, int32_t variant, void* stream) {
    DeviceState& s = device_state(dev);
    HIP_CHECK(hipSetDevice(dev));
    if (any_hit) launch_bvh8<true>(s, nodes, tris, rays, hits, num_rays, variant, (hipStream_t)stream);
    else         launch_bvh8<false>(s, nodes, tris, rays, hits, num_rays, variant, (hipStream_t)stream);
}

void amdgpu_intersect_single_ray1_bvh2_tri1(int32_t dev, const Node2* nodes, const Tri1* tris, const Ray1* rays, Hit1* hits, int32_t num_rays) {
    hip_traverse_bvh2_tri1_async(dev, nodes, tris, rays, hits, num_rays, 0, default_variant(2), nullptr);
    check_error_flag(device_state(dev), nullptr);
}
void amdgpu_occluded_single_ray1_bvh2_tri1(int32_t dev, const Node2* nodes, const Tri1* tris, const Ray1* rays, Hit1* hits, int32_t num_rays) {
    hip_traverse_bvh2_tri1_async(dev, nodes, tris, rays, hits, num_rays, 1, default_variant(2), nullptr);
    check_error_flag(device_state(dev), nullptr);
}
void hip_intersect_single_ray1_bvh8_tri4(int32_t dev, const Node8* nodes, const Tri4* tris, const Ray1* rays, Hit1* hits, int32_t num_rays) {
    hip_traverse_bvh8_tri4_async(dev, nodes, tris, rays, hits, num_rays, 0, default_variant(8), nullptr);
    check_error_flag(device_state(dev), nullptr);
}
void hip_occluded_single_ray1_bvh8_tri4(int32_t dev, const Node8* nodes, const Tri4* tris, const Ray1* rays, Hit1* hits, int32_t num_rays) {
    hip_traverse_bvh8_tri4_async(dev, nodes, tris, rays, hits, num_rays, 1, default_variant(8), nullptr);
    check_error_flag(device_state(dev), nullptr);
}

int32_t rodent_hip_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count;
}
int32_t rodent_hip_num_variants(int32_t bvh_width) { return bvh_width == 2 ? kNumVariants2 : (bvh_width == 8 ? kNumVariants8 : 0); }
const char* rodent_hip_variant_name(int32_t bvh_width, int32_t variant) {
    if (bvh_width == 2 && variant >= 0 && variant < kNumVariants2) return kVariants2[variant].name;
    if (bvh_width == 8 && variant >= 0 && variant < kNumVariants8) return kVariants8[variant].name;
    return "";
}
const char* rodent_hip_kernel_name(int32_t bvh_width, int32_t variant, int32_t any_hit) {
    const VariantInfo* v = nullptr;
    if (bvh_width == 2 && variant >= 0 && variant < kNumVariants2) v = &kVariants2[variant];
    if (bvh_width == 8 && variant >= 0 && variant < kNumVariants8) v = &kVariants8[variant];
    return v ? (any_hit ? v->kernel_any : v->kernel_closest) : "";
}
const char* rodent_hip_version(void) { return "rodent_hip 0.1 (gfx950)"; }

} // extern "C"

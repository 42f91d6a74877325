// render.hip -- wavefront path tracer for MI355X (gfx950) behind the C ABI of include/rodent_render.h.
//
// Replaces (all Impala, compiled by AnyDSL in the reference):
//   src/render/mapping_gpu.impala:223-265  gpu_generate_rays      -> k_generate
//   src/render/mapping_gpu.impala:18-30    gpu_traverse_primary   -> k_trace_stream<false>
//   src/render/mapping_gpu.impala:166-221  gpu_sort_primary       -> k_bin_count / k_bin_scan_blocks / k_bin_scan_bins / k_scatter
//   src/render/mapping_gpu.impala:82-134   gpu_shade              -> k_shade  (one launch, material switch; rays arrive sorted)
//   src/render/mapping_gpu.impala:47-80    gpu_traverse_secondary -> k_trace_stream<true> (+ film accumulation :32-45)
//   src/render/mapping_gpu.impala:267-300  gpu_compact_primary    -> the same binning kernels with key = dead?1:0
//   src/render/mapping_gpu.impala:308-369  gpu_streaming_trace    -> render_rows() host loop
//   src/render/mapping_gpu.impala:371-474  gpu_mega_kernel_trace  -> k_mega / render_rows_mega()
//   src/driver/interface.cpp:359-390,528-563,565-663  stream slabs, film, rodent_* services
//
// CDNA4 notes: SoA streams (one 20 x capacity / 13 x capacity float slab each, driver.impala:24-61) so a
// wavefront's 64 lanes read 64 consecutive words; one wave per workgroup for the traversal kernels
// (per-lane stack window in LDS, deeper entries in scratch); sort and compaction are deterministic
// (per-block histograms + scans, wave ballots for the in-block rank) instead of the reference's global
// atomics, so stream order -- and therefore every test -- is reproducible; the film uses hardware
// fp32 atomic adds.  No MFMA: branchy scalar work.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "rodent_render.h"
#include "shading.h"
#include "traversal_device.h"

#define HIP_CHECK(expr)                                                                        \
    do {                                                                                       \
        hipError_t err_ = (expr);                                                              \
        if (err_ != hipSuccess) {                                                              \
            fprintf(stderr, "rodent_hip: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(err_), __FILE__, __LINE__); \
            abort();                                                                           \
        }                                                                                      \
    } while (0)

namespace {

using namespace rodent_dev;

constexpr int kBlock = 256;                    // workgroup of the streaming (non-traversal) kernels
constexpr int kShadeBlockDefault = 512;        // ... of the shader (shade_block(): measured 256 / 512 / 1024, profiles/r05_shade_block.txt)
// Rays per workgroup of the binning kernels (k_bin_count, k_scatter).  A workgroup's rays of one bin land in ONE contiguous
// run of the output: with 1024 rays and ten bins a run is ~400 bytes per array instead of ~100 -- whole cache lines instead
// of lines shared with the neighbouring workgroup (which usually runs on another XCD, behind another L2) -- and the per-block
// histogram is a quarter of the size.
constexpr int kBinBlock = 1024;
constexpr int kMaxBins = 1025;                 // mapping_gpu.impala:200,342 (1024 geometries + the "miss" bin)
// Rays per stream.  The reference uses 1 Mi (mapping_gpu.impala:319, sized for ~12 GB boards).  On this chip a 1 Mi-ray
// launch is two rounds of resident waves, all fill and drain; 8 Mi-ray streams render the atrium 25 % faster and the
// Cornell box 9 % faster (round 1), 32 Mi-ray streams another 4.6 % / 4.6 % (profiles/r03_capacity_sweep.txt; 64 Mi: no further gain):
// 7.1 GB of streams out of 288 GB.  Results do not depend on it.  rodent_hip_render_capacity() / RODENT_HIP_STREAM_CAPACITY override it.
constexpr int kDefaultCapacity = 32 * 1024 * 1024;
constexpr long kMaxCapacity = 64l << 20;
static int env_capacity() {
    static const int v = [] { const char* e = getenv("RODENT_HIP_STREAM_CAPACITY"); const long c = e ? atol(e) : 0;
        return c >= 64 && c <= kMaxCapacity ? (int)c : kDefaultCapacity; }();
    return v;
}
// RODENT_HIP_PIXEL_BLOCK: the streaming renderer generates a frame's pixels in blocks of this many pixels squared (default 16; 0: row by
// row, the order of the reference's gpu_generate_rays, mapping_gpu.impala:236-241).  Which pixel a stream slot carries is free -- a
// sample's seed depends on (sample, iter, x, y) only, renderer.impala:28-33 -- and the rays in flight together then start, and bounce, in a
// compact part of the scene instead of along a strip of rows: config 5's frame 1 790 -> 1 771 ms (+1.1 %; 8: +0.6 %, 32 and 64 as 16), ray
// counts identical (profiles/r05_pixel_blocks.txt).
inline int pixel_block() {
    static const int v = [] { const char* e = getenv("RODENT_HIP_PIXEL_BLOCK"); const int b = e ? atoi(e) : 16;
        return b == 8 || b == 16 || b == 32 || b == 64 ? b : 0; }();
    return v;
}
// [0..3] host-visible totals, [4..67] shadow rays (striped), [68..99] megakernel primary rays (striped)
constexpr int kNumCounters = 100;

struct CameraDev { float eye[3], dir[3], up[3], right[3]; float w, h; };

// Stream sizes are either known on the host (passed by value) or live on the device (the valid-hit count
// after sorting): size_ptr != nullptr wins.
__device__ __forceinline__ int stream_size(const int* size_ptr, int n_value) { return size_ptr ? *size_ptr : n_value; }

// ---------------------------------------------------------------------------------------------
// K3: ray generation (mapping_gpu.impala:223-265, renderer.impala:26-40, camera.impala:35-44)
// ---------------------------------------------------------------------------------------------
// on_emit (renderer.impala:26-40): the sample's seed and its camera ray direction
__device__ __forceinline__ v3 emit_sample(const CameraDev& cam, int iter, int film_w, int film_h, int x, int y, int sample,
    uint32_t* rnd_out) {
    uint32_t rnd = fnv_hash(fnv_hash(fnv_hash(fnv_hash(0x811C9DC5u, (uint32_t)sample), (uint32_t)iter), (uint32_t)x), (uint32_t)y);
    const float kx = 2.0f * ((float)x + randf(&rnd)) / (float)film_w - 1.0f;
    const float ky = 1.0f - 2.0f * ((float)y + randf(&rnd)) / (float)film_h;
    *rnd_out = rnd;
    return normalize(add(add(mulf(LD3(cam.right), cam.w * kx), mulf(LD3(cam.up), cam.h * ky)), LD3(cam.dir)));
}

// Pixels of a call: [first_pixel, ...) contiguous (tile_pixels == 0: a row band), or interleaved row tiles -- local pixel q is pixel
// first_pixel + (q / tile_pixels) * stride_pixels + q % tile_pixels (rodent_hip_render_tiles: the film's row tiles dealt round-robin to the
// GPUs).
__global__ __launch_bounds__(kBlock) void k_generate(PrimaryStream p, int first_dst, int first_ray_id, int num_rays, CameraDev cam,
                                                      int iter, int film_w, int film_h, int first_pixel, int spp, int tile_pixels,
                                                          int stride_pixels,
                                                      int block = 0, int band_rows = 0) {
    const int gid = blockIdx.x * kBlock + threadIdx.x;
    if (gid >= num_rays) return;
    const int ray_id = first_ray_id + gid, dst = first_dst + gid;
    const int sample = ray_id % spp, q = ray_id / spp;
    // local: the pixel's index in rows that are contiguous in the film (the call's band, or one of its tiles)
    int tile = 0, local = q;
    if (tile_pixels > 0) { tile = q / tile_pixels; local = q - tile * tile_pixels; }
    // block x block pixels after one another instead of whole rows (pixel_block(), below); band_rows: rows of the band / a tile
    if (block > 0) {
        const int strip = local / (block * film_w), i = local - strip * (block * film_w);
        if ((strip + 1) * block <= band_rows) local =
            (strip * block + (i % (block * block)) / block) * film_w + (i / (block * block)) * block + i % block;
    }
    const int pixel = first_pixel + tile * stride_pixels + local;
    const int y = pixel / film_w, x = pixel - y * film_w;          // the reference uses fast_div (common.impala:19-35): same quotient
    uint32_t rnd;
    const v3 d = emit_sample(cam, iter, film_w, film_h, x, y, sample, &rnd);
    p.rays.id[dst] = pixel;
    p.rays.org_x[dst] = cam.eye[0]; p.rays.org_y[dst] = cam.eye[1]; p.rays.org_z[dst] = cam.eye[2];
    p.rays.dir_x[dst] = d.x; p.rays.dir_y[dst] = d.y; p.rays.dir_z[dst] = d.z;
    p.rays.tmin[dst] = 0.0f; p.rays.tmax[dst] = FLT_MAX_REF;
    p.rnd[dst] = rnd; p.mis[dst] = 0.0f; p.contrib_r[dst] = 1.0f; p.contrib_g[dst] = 1.0f; p.contrib_b[dst] = 1.0f; p.depth[dst] = 0;
}

// ---------------------------------------------------------------------------------------------
// K1 / K6: BVH2 traversal over an SoA ray stream.  Per-ray visit order = the reference kernel
// (traversal/mapping_gpu.impala:94-178); wave scheduling = the single-step loop of traversal.hip
// (branch-free node step).  Stack: 16-entry window in LDS; deeper rays go to k_trace_deep (stream kernels) / scratch (megakernel).
// ---------------------------------------------------------------------------------------------
constexpr int kLdsStack = 16;
constexpr int kPersistWaves = 16;               // waves per workgroup of the persistent traversal kernels (k_trace_persist, k_trace_refill)
constexpr int kSpillEntries = kStackCap - kLdsStack;
// Stack of the megakernel (one workgroup per film tile): 16-entry window in LDS, deeper entries in scratch.
template <bool UNUSED>
struct StreamStackT {
    lds_int* col; int* err;
    int local[kSpillEntries];
    __device__ __forceinline__ int get(int e) const {
        if (__builtin_expect(e < kLdsStack, 1)) return col[e * kWave];
        return local[(e < kStackCap ? e : kStackCap - 1) - kLdsStack];
    }
    __device__ __forceinline__ void put(int e, int v) {
        if (__builtin_expect(e < kLdsStack, 1)) col[e * kWave] = v;
        else if (e < kStackCap) local[e - kLdsStack] = v;
        else *err = 1;
    }
};
using StreamStack = StreamStackT<false>;

// LDS-only stack of the stream traversal kernels, kept as a cursor (pointer to the entry under the top).  A ray whose
// stack outgrows the 16-entry window is abandoned (overflow = true) and traced again by k_trace_deep with the
// 64-entry stack in global memory -- the arrangement of traversal.hip (k_bvh2_single / k_bvh2_finish): no overflow
// handling inside the hot loop.
// The persistent kernels (k_trace_persist, k_trace_refill) do not abandon it: the oldest entries move to the wave's block of global memory
// and the ray goes on in its lane (SPILLW = the window's rows; stack_spill / stack_reload, traversal_device.h).
struct CursorStack {
    lds_int* sp; lds_int* limit; bool overflow;
    int* spill; int* err;                          // the launch's spill buffer (wave-uniform) and error flag
    __device__ __forceinline__ void init(lds_int* col, int window = kLdsStack, int* spill_ = nullptr, int* err_ = nullptr) {
        sp = col; limit = col + window * kWave; overflow = false; col[0] = 0; spill = spill_; err = err_;
    }
};
// 64 entries ([entry][lane]), the reference's capacity (stack.impala:53), in 16 KB of LDS; used by k_trace_deep only.  (In global
// memory, rounds 1-2, every push and pop was a round trip: ~100 us for the first deep ray of a launch.)
struct DeepStack {
    lds_int* base; int* err;
    __device__ __forceinline__ int get(int e) const { return base[(e < kStackCap ? e : kStackCap - 1) * kWave]; }
    __device__ __forceinline__ void put(int e, int v) { if (e < kStackCap) base[e * kWave] = v; else *err = 1; }
};
template <typename T> struct is_cursor { static constexpr bool value = false; };
template <> struct is_cursor<CursorStack> { static constexpr bool value = true; };

// XCD-aware wave -> chunk mapping of traversal.hip (k_bvh2_single): groups of 32 consecutive 64-ray chunks per XCD.
__device__ __forceinline__ int xcd_chunk(int block, int total_chunks) {
    constexpr int G = 32;
    const int span = 8 * G, full = (total_chunks / span) * span;
    if (block >= full) return block;
    const int x = block % 8, l = block / 8;
    return ((l / G) * 8 + x) * G + l % G;
}

struct StreamHit { int prim, geom; float t, u, v; };

// Film accumulation for a wavefront (mapping_gpu.impala:32-45 does one atomic per ray and channel).
// With spp > 1 the rays of one pixel sit in neighbouring lanes, and 64 lanes hitting the same three
// addresses serialise in the atomic unit (measured: the shadow pass took 4x the primary pass on the
// Cornell box).  Lanes that share a pixel are therefore summed in the wave first (butterfly over the
// lanes, a fixed order) and one lane issues the atomics; after kFilmRounds distinct pixels the remaining
// lanes fall back to their own atomics.  Only the order of the fp32 additions differs from the reference.
constexpr int kFilmRounds = 4;
__device__ __forceinline__ void film_add_wave(float* film, int pixel, bool valid, float r, float g, float b) {
    unsigned long long todo = __ballot(valid);
    for (int round = 0; round < kFilmRounds && todo; round++) {
        const int leader = __ffsll((long long)todo) - 1;
        const int p0 = __shfl(pixel, leader);
        const bool mine = valid && pixel == p0;
        const unsigned long long same = __ballot(mine);
        if (__popcll(same) > 1) {
            float sr = mine ? r : 0.0f, sg = mine ? g : 0.0f, sb = mine ? b : 0.0f;
            for (int o = 32; o > 0; o >>= 1) { sr += __shfl_xor(sr, o); sg += __shfl_xor(sg, o); sb += __shfl_xor(sb, o); }
            if ((int)(threadIdx.x % kWave) == leader) { float* px = film + 3 * (size_t)p0; unsafeAtomicAdd(px, sr);
                unsafeAtomicAdd(px + 1, sg); unsafeAtomicAdd(px + 2, sb); }
            if (mine) valid = false;
        } else if (mine) {
            float* px = film + 3 * (size_t)pixel; unsafeAtomicAdd(px, r); unsafeAtomicAdd(px + 1, g); unsafeAtomicAdd(px + 2, b);
            valid = false;
        }
        todo &= ~same;
    }
    if (valid) { float* px = film + 3 * (size_t)pixel; unsafeAtomicAdd(px, r); unsafeAtomicAdd(px + 1, g); unsafeAtomicAdd(px + 2, b); }
}

// Single-step schedule, as k_bvh2_single / unified_chunk in traversal.hip: each lane advances by one node step or
// one triangle test per wave iteration and the loads of both kinds are in flight together.
// on_hit(prim, geom, t, u, v) is called for every accepted triangle (the last call is the closest hit); returns whether
// anything was hit.  The stream kernels store from on_hit instead of carrying a hit record in registers.
// TOP: `image` is the scene's top-of-tree image staged in LDS by the workgroup (traversal_device.h); a node id >= kLdsTag is a
// link into it and is fetched with ds_read_b128 instead of through the vector-memory pipeline.
template <bool ANY, bool TOP = false, int SPILLW = 0, typename Stack, typename OnHit>
__device__ __forceinline__ bool trace_one(const Node2* __restrict__ nodes, const Tri1* __restrict__ tris, RayX ray, Stack& st,
    OnHit on_hit, lds_int* image = nullptr) {
    constexpr bool kCursor = is_cursor<Stack>::value;
    bool any_found = false;
    int ptr = 0, top = TOP ? kLdsTag : 1;
    if constexpr (!kCursor) st.put(0, 0);
    ray.tmin = canonical(ray.tmin); ray.tmax = canonical(ray.tmax);           // see slab_canonical (traversal_device.h)
    // both bases as integers in VGPRs for the per-lane select, then GLOBAL pointers again (see unified_chunk)
    typedef const __attribute__((address_space(1))) char* gptr;
    // node ids are 1-based
    unsigned long long node_bits = reinterpret_cast<unsigned long long>(nodes - 1), tri_bits = reinterpret_cast<unsigned long long>(tris);
    asm volatile("" : "+v"(node_bits), "+v"(tri_bits));
    const gptr node_base = (gptr)node_bits, tri_base = (gptr)tri_bits;
    while (__ballot(top != 0)) {
        if (top != 0) {
            const bool is_node = top > 0;
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            typedef int i32x2 __attribute__((ext_vector_type(2)));
            f32x4 q0, q1, q2;
            i32x2 ch;
            int popped;
            if constexpr (TOP && kCursor) {
                // both kinds of fetch -- LDS image, memory -- and the word under the cursor in flight together (joint_fetch,
                // traversal_device.h)
                const unsigned idx = (unsigned)(is_node ? top : ~top), stride = is_node ? (unsigned)sizeof(Node2) : (unsigned)sizeof(Tri1);
                const gptr addr = (is_node ? node_base : tri_base) + (size_t)idx * stride;
                joint_fetch(q0, q1, q2, ch, popped, top >= kLdsTag, (unsigned)(size_t)image + (unsigned)(top - kLdsTag), addr,
                    addr + (is_node ? 48u : 40u), st.sp);
            } else {
                const unsigned idx = (unsigned)(is_node ? top : ~top), stride = is_node ? (unsigned)sizeof(Node2) : (unsigned)sizeof(Tri1);
                const gptr addr = (is_node ? node_base : tri_base) + (size_t)idx * stride;
                const __attribute__((address_space(1))) f32x4* p = (const __attribute__((address_space(1))) f32x4*)addr;
                q0 = p[0]; q1 = p[1]; q2 = p[2];
                // child ids / (triangle lanes) own last 8 bytes
                ch = *(const __attribute__((address_space(1))) i32x2*)(addr + (is_node ? 48u : 40u));
                if constexpr (kCursor) popped = *st.sp; else popped = st.get(ptr);
                // keep all four loads in flight together (see unified_chunk)
                asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(ch));
            }
            if (is_node) {
                float te0, te1;
                const bool h0 = slab_canonical(ray, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, te0) && ch.x != 0;
                const bool h1 = slab_canonical(ray, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, te1) && ch.y != 0;
                // (any-hit rays -- the renderer's shadow rays -- take child 0 first: occlusion does not depend on the order, see
                // k_trace_refill's SHADOW_ORDER)
                const bool c0first = ANY ? true : te0 < te1, both = h0 && h1;
                top = both ? (c0first ? ch.x : ch.y) : (h0 ? ch.x : (h1 ? ch.y : popped));
                if constexpr (kCursor) {
                    st.sp[kWave] = c0first ? ch.y : ch.x;
                    st.sp += (both ? kWave : 0) - ((h0 || h1) ? 0 : kWave);
                    // (`both`: popping the sentinel moves sp below its column)
                    if (both && st.sp >= st.limit) {
                        if constexpr (SPILLW > 0) stack_spill<SPILLW>(st.sp, top, st.limit, st.spill, kPersistWaves, st.err);
                        else { st.overflow = true; top = 0; }
                    }
                } else {
                    st.put(ptr + 1, c0first ? ch.y : ch.x);
                    ptr += (both ? 1 : 0) - ((h0 || h1) ? 0 : 1);
                }
            } else {
                const int prim_id = __float_as_int(q2.w);
                const float nx = cross_x(q1.x, q1.y, q1.z, q2.x, q2.y, q2.z), ny = cross_y(q1.x, q1.y, q1.z, q2.x, q2.y, q2.z),
                    nz = cross_z(q1.x, q1.y, q1.z, q2.x, q2.y, q2.z);
                float t, u, v;
                bool found = false;
                if (intersect_tri(ray, q0.x, q0.y, q0.z, q1.x, q1.y, q1.z, q2.x, q2.y, q2.z, nx, ny, nz, t, u, v)) {
                    on_hit(prim_id & 0x7FFFFFFF, __float_as_int(q1.w), t, u, v);
                    ray.tmax = t; found = true; any_found = true;
                }
                const bool leave = prim_id < 0;                               // sentinel: the leaf is done
                top = (ANY && found) ? 0 : (leave ? popped : top - 1);        // top - 1 == ~(j + 1)
                if constexpr (kCursor) st.sp -= (leave && !(ANY && found)) ? kWave : 0;
                else ptr -= (leave && !(ANY && found)) ? 1 : 0;
            }
            // popped row 0 while entries are out
            if constexpr (kCursor && SPILLW > 0) if (top >= kSpillMark) stack_reload<SPILLW>(st.sp, top, st.limit, st.spill, kPersistWaves);
        }
    }
    return any_found;
}

// intersection.impala:88-99
__device__ __forceinline__ RayX make_rayx(float ox, float oy, float oz, float dx, float dy, float dz, float tmin, float tmax) {
    RayX x;
    x.ox = ox; x.oy = oy; x.oz = oz; x.dx = dx; x.dy = dy; x.dz = dz; x.tmin = tmin; x.tmax = tmax;
    x.idx = safe_rcp(x.dx); x.idy = safe_rcp(x.dy); x.idz = safe_rcp(x.dz);
    x.iox = -(x.ox * x.idx); x.ioy = -(x.oy * x.idy); x.ioz = -(x.oz * x.idz);
    return x;
}
// Two rays per lane, back to back, in ONE wave-level loop (the megakernel's joint form): first the shadow ray `a` (any hit:
// the first accepted triangle ends it), then the path ray `b` (closest hit, on_hit_b for every accepted triangle), both from
// the same origin.  A wave then needs max over its lanes of (steps of a + steps of b) iterations instead of the sum of the two
// maxima of separate loops.  Per ray the visit order is trace_one's.  Returns {a was occluded, b hit something}.
struct TwoHits { bool a_occluded, b_hit; };
template <typename Stack, typename OnHit>
__device__ __forceinline__ TwoHits trace_two(const Node2* __restrict__ nodes, const Tri1* __restrict__ tris, bool has_a, bool has_b,
                                             float ox, float oy, float oz, float adx, float ady, float adz, float a_tmin, float a_tmax,
                                             float bdx, float bdy, float bdz, float b_tmin, float b_tmax, Stack& st, OnHit on_hit_b) {
    TwoHits out{false, false};
    bool phase_a = has_a;
    RayX ray = phase_a ? make_rayx(ox, oy, oz, adx, ady, adz, a_tmin, a_tmax) : make_rayx(ox, oy, oz, bdx, bdy, bdz, b_tmin, b_tmax);
    ray.tmin = canonical(ray.tmin); ray.tmax = canonical(ray.tmax);
    int ptr = 0, top = (has_a || has_b) ? 1 : 0;
    st.put(0, 0);
    typedef const __attribute__((address_space(1))) char* gptr;
    unsigned long long node_bits = reinterpret_cast<unsigned long long>(nodes - 1), tri_bits = reinterpret_cast<unsigned long long>(tris);
    asm volatile("" : "+v"(node_bits), "+v"(tri_bits));
    const gptr node_base = (gptr)node_bits, tri_base = (gptr)tri_bits;
    for (;;) {
        // a lane whose shadow ray is finished goes on with its path ray (if it has one)
        const bool next = top == 0 && phase_a;
        if (__ballot(next)) {
            if (next) {
                phase_a = false;
                if (has_b) {
                    ray = make_rayx(ox, oy, oz, bdx, bdy, bdz, b_tmin, b_tmax);
                    ray.tmin = canonical(ray.tmin); ray.tmax = canonical(ray.tmax);
                    top = 1; ptr = 0; st.put(0, 0);
                }
            }
        }
        if (!__ballot(top != 0)) break;
        if (top != 0) {
            const bool is_node = top > 0;
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            typedef int i32x2 __attribute__((ext_vector_type(2)));
            const unsigned idx = (unsigned)(is_node ? top : ~top), stride = is_node ? (unsigned)sizeof(Node2) : (unsigned)sizeof(Tri1);
            const gptr addr = (is_node ? node_base : tri_base) + (size_t)idx * stride;
            const __attribute__((address_space(1))) f32x4* p = (const __attribute__((address_space(1))) f32x4*)addr;
            f32x4 q0 = p[0], q1 = p[1], q2 = p[2];
            i32x2 ch = *(const __attribute__((address_space(1))) i32x2*)(addr + (is_node ? 48u : 40u));
            const int popped = st.get(ptr);
            asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(ch));
            if (is_node) {
                float te0, te1;
                const bool h0 = slab_canonical(ray, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, te0) && ch.x != 0;
                const bool h1 = slab_canonical(ray, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, te1) && ch.y != 0;
                const bool c0first = te0 < te1, both = h0 && h1;
                top = both ? (c0first ? ch.x : ch.y) : (h0 ? ch.x : (h1 ? ch.y : popped));
                st.put(ptr + 1, c0first ? ch.y : ch.x);
                ptr += (both ? 1 : 0) - ((h0 || h1) ? 0 : 1);
            } else {
                const int prim_id = __float_as_int(q2.w);
                const float nx = cross_x(q1.x, q1.y, q1.z, q2.x, q2.y, q2.z), ny = cross_y(q1.x, q1.y, q1.z, q2.x, q2.y, q2.z),
                    nz = cross_z(q1.x, q1.y, q1.z, q2.x, q2.y, q2.z);
                float t, u, v;
                bool ends = false;
                if (intersect_tri(ray, q0.x, q0.y, q0.z, q1.x, q1.y, q1.z, q2.x, q2.y, q2.z, nx, ny, nz, t, u, v)) {
                    if (phase_a) { out.a_occluded = true; ends = true; }
                    else { on_hit_b(prim_id & 0x7FFFFFFF, __float_as_int(q1.w), t, u, v); ray.tmax = t; out.b_hit = true; }
                }
                const bool leave = prim_id < 0;
                top = ends ? 0 : (leave ? popped : top - 1);
                ptr -= (leave && !ends) ? 1 : 0;
            }
        }
    }
    return out;
}

__device__ __forceinline__ RayX load_stream_ray(const RayStream& r, int i) {   // driver.impala:63-84
    return make_rayx(r.org_x[i], r.org_y[i], r.org_z[i], r.dir_x[i], r.dir_y[i], r.dir_z[i], r.tmin[i], r.tmax[i]);
}

// Hit records of a primary stream.  The ABI's layout (driver.impala:24-61) is five arrays -- geom_id, prim_id, t, u, v -- and a record is
// five 4-byte stores into five different 32-byte sectors; lanes that were refilled hold non-consecutive rays, so nothing coalesces: the
// traversal launches of the atrium frame wrote 97 bytes per ray for <= 20 bytes of records (profiles/r03_render_profile_cfg5.json).  Inside
// the library's own loop (render_rows; PrimaryStream::pad bit 0, never set on a caller's stream) the SAME memory -- the five arrays are
// consecutive in the slab, carve_primary -- holds one 20-byte record per ray at words [5 i, 5 i + 5): one 16-byte and one 4-byte store into
// one or two sectors; the shader reads the records in stream order (coalesced).  Record fields: geom_id = num_geometries on a miss
// (driver.impala:106-115).
constexpr int kHitRecordsAoS = 1;
struct HitRecord { int geom, prim; float t, u, v; };
typedef int i32x4_dword_aligned __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ void store_hit_record(const PrimaryStream& p, unsigned i, int geom, int prim, float t, float u, float v) {
    if (p.pad & kHitRecordsAoS) {
        int* rec = p.geom_id + 5u * i;
        *reinterpret_cast<i32x4_dword_aligned*>(rec) = i32x4_dword_aligned{geom, prim, __float_as_int(t), __float_as_int(u)};
        rec[4] = __float_as_int(v);
    } else { p.geom_id[i] = geom; p.prim_id[i] = prim; p.t[i] = t; p.u[i] = u; p.v[i] = v; }
}
__device__ __forceinline__ HitRecord load_hit_record(const PrimaryStream& p, unsigned i) {
    if (p.pad & kHitRecordsAoS) {
        const int* rec = p.geom_id + 5u * i;
        const i32x4_dword_aligned q = *reinterpret_cast<const i32x4_dword_aligned*>(rec);
        return HitRecord{q.x, q.y, __int_as_float(q.z), __int_as_float(q.w), __int_as_float(rec[4])};
    }
    return HitRecord{p.geom_id[i], p.prim_id[i], p.t[i], p.u[i], p.v[i]};
}
__device__ __forceinline__ int load_hit_geom(const PrimaryStream& p, unsigned i) {
    return (p.pad & kHitRecordsAoS) ? p.geom_id[5u * i] : p.geom_id[i]; }

// primary: writes the hit record (geom_id = num_geometries on a miss, driver.impala:106-115; prim_id, t, u, v)
template <bool TOP = false, int SPILLW = 0>
__device__ __forceinline__ void trace_primary_ray(const SceneDev& sc, const PrimaryStream& p, int i, CursorStack* cursor, DeepStack* deep,
    lds_int* image = nullptr) {
    const RayX ray = load_stream_ray(p.rays, i);
    store_hit_record(p, (unsigned)i, sc.num_materials, -1, ray.tmax, 0.0f, 0.0f);     // the miss record; hits overwrite it
    auto on_hit = [&](int prim, int geom, float t, float u, float v) {
        unsigned k = (unsigned)i;
        // opaque index: SGPR bases + one VGPR offset here, instead of five 64-bit addresses held across the loop
        asm volatile("" : "+v"(k));
        store_hit_record(p, k, geom, prim, t, u, v);
    };
    if (cursor) trace_one<false, TOP, SPILLW>(sc.nodes, sc.tris, ray, *cursor, on_hit, image);
    else trace_one<false>(sc.nodes, sc.tris, ray, *deep, on_hit);
}

// WAVES x 64 threads per workgroup; TOPN > 0: the scene's top-of-tree image (first TOPN records) is staged in LDS behind the
// WAVES stack windows (kTopStack entries each then: 2 x 4 KB + 31 x 64 B = 16 workgroups per CU, all 32 wave slots).
constexpr int kTopStack = 15, kSceneTopNodes = 31, kTraceWaves = 2;
// below one resident generation of rays the persistent form does not pay (traversal.hip)
constexpr int kPersistMinRays = 8192 * kWave;
template <int WAVES, int TOPN>
__device__ __forceinline__ lds_int* stage_scene_image(const SceneDev& sc, int* lds, int& chunk, int total_chunks) {
    constexpr int kWindow = TOPN ? kTopStack : kLdsStack;
    lds_int* image = (lds_int*)lds + WAVES * (kWindow + 1) * kWave;
    if (TOPN) {
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        for (int j = threadIdx.x; j < TOPN * 4; j += WAVES * kWave)
            reinterpret_cast<__attribute__((address_space(3))) i32x4*>(image)[j] = reinterpret_cast<const i32x4*>(sc.top_image)[j];
    }
    // XCD-aware wave -> chunk mapping (xcd_chunk): wave l of XCD x, l counted over the workgroups of that XCD
    const int wave = threadIdx.x / kWave;
    constexpr int G = 32;
    const int span = 8 * G, full = (total_chunks / span) * span;
    chunk = blockIdx.x * WAVES + wave;
    if ((int)blockIdx.x * WAVES < full) {
        const int x = blockIdx.x % 8, l = (blockIdx.x / 8) * WAVES + wave;
        chunk = ((l / G) * 8 + x) * G + l % G;
    }
    if (TOPN && WAVES > 1) __syncthreads();
    return image;
}

template <int WAVES, int TOPN>
__global__ __launch_bounds__(kWave * WAVES) void k_trace_primary(SceneDev sc, PrimaryStream p, const int* size_ptr, int n_value,
    int* deep_count, unsigned long long* counters,
                                                                 int* deep_list) {
    constexpr int kWindow = TOPN ? kTopStack : kLdsStack;
    __shared__ __attribute__((aligned(16))) int lds[WAVES * (kWindow + 1) * kWave + TOPN * 16];
    const int n = stream_size(size_ptr, n_value);
    int chunk;
    lds_int* image = stage_scene_image<WAVES, TOPN>(sc, lds, chunk, (n + kWave - 1) / kWave);
    const int lane = threadIdx.x % kWave, i = chunk * kWave + lane;
    if (i >= n) return;
    CursorStack st; st.init((lds_int*)lds + (threadIdx.x / kWave) * (kWindow + 1) * kWave + lane, kWindow);
    trace_primary_ray<(TOPN > 0)>(sc, p, i, &st, nullptr, image);
    if (st.overflow) deep_list[atomicAdd(deep_count, 1)] = i;
    if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(&counters[0], (unsigned long long)n);
}

// secondary: any-hit; unoccluded rays add their colour to the film (mapping_gpu.impala:32-45,47-80)
template <int WAVES, int TOPN>
__global__ __launch_bounds__(kWave * WAVES) void k_trace_secondary(SceneDev sc, SecondaryStream s, const int* size_ptr, int n_value,
    float* film, float inv_spp,
                                                                   int* deep_count, unsigned long long* counters, int* deep_list) {
    constexpr int kWindow = TOPN ? kTopStack : kLdsStack;
    __shared__ __attribute__((aligned(16))) int lds[WAVES * (kWindow + 1) * kWave + TOPN * 16];
    const int n = stream_size(size_ptr, n_value);
    int chunk;
    // (the grid covers n_value rays: the mapping must not depend on the device-side n)
    lds_int* image = stage_scene_image<WAVES, TOPN>(sc, lds, chunk, (n_value + kWave - 1) / kWave);
    const int lane = threadIdx.x % kWave, i = chunk * kWave + lane;
    if (chunk * kWave >= n) return;
    const int pixel = i < n ? s.rays.id[i] : -1;
    const unsigned long long live = __ballot(pixel >= 0);
    // striped over 64 words: one counter word saturates near 88 atomics/us and made this kernel 3x slower
    if (lane == 0 && live) atomicAdd(&counters[4 + (chunk & 63)], (unsigned long long)__popcll(live));
    bool lit = false;
    if (pixel >= 0) {
        CursorStack st; st.init((lds_int*)lds + (threadIdx.x / kWave) * (kWindow + 1) * kWave + lane, kWindow);
        lit = !trace_one<true, (TOPN > 0)>(sc.nodes, sc.tris, load_stream_ray(s.rays, i), st, [](int, int, float, float, float) {}, image);
        if (st.overflow) { deep_list[atomicAdd(deep_count, 1)] = i; lit = false; }      // k_trace_deep decides
    }
    film_add_wave(film, pixel, lit, lit ? s.color_r[i] * inv_spp : 0.0f, lit ? s.color_g[i] * inv_spp : 0.0f,
        lit ? s.color_b[i] * inv_spp : 0.0f);
}

// Persistent form of the two kernels above (rodent_hip_render_trace_persistent; traversal.hip k_bvh2_top_persist): one resident
// generation of 16-wave workgroups, each stages the scene's 255-record image once and its waves draw 64-ray chunks from 64
// striped ticket counters (ticket t of stripe s = chunk ((t / 32) * 64 + s) * 32 + t % 32; a wave's first ticket is its rank in
// the stripe).  The stream size may live on the device: the grid does not depend on it.  k_trace_deep zeroes the counters.
constexpr int kPersistTopNodes = 255, kTraceStripes = 64, kTraceCounterStride = 16;
// MODE 0: the closest-hit pass over `p`; 1: the shadow pass over `s`; 2 ("joint", rodent_hip_render_trace_persistent(dev, 2)): BOTH in
// one launch -- the closest-hit pass of an iteration and the shadow pass of the iteration before it depend on the same shader
// run and on nothing else, so their chunks go through one ticket sequence (the closest-hit chunks first): one resident
// generation works through both lists, no pass waits for the other's tail, and there is no second stream whose kernels a
// persistent grid would shut out.
template <int MODE>
__global__ __launch_bounds__(kWave * kPersistWaves) __attribute__((amdgpu_waves_per_eu(8, 8)))
void k_trace_persist(SceneDev sc, PrimaryStream p, int n_primary, SecondaryStream s, const int* size_ptr, int n_value, float* film,
    float inv_spp,
                     int* deep_count_primary, int* deep_count_secondary, unsigned long long* counters, int* deep_list_primary,
                         int* deep_list_secondary, int* tickets,
                     int* spill, int* err) {
    constexpr int kStackInts = kPersistWaves * (kTopStack + 1) * kWave;
    __shared__ __attribute__((aligned(16))) int lds[kStackInts + kPersistTopNodes * 16];
    const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
    lds_int* image = (lds_int*)lds + kStackInts;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    for (int j = threadIdx.x; j < kPersistTopNodes * 4; j += kWave * kPersistWaves)
        reinterpret_cast<__attribute__((address_space(3))) i32x4*>(image)[j] = reinterpret_cast<const i32x4*>(sc.top_image_large)[j];
    const int np = MODE == 1 ? 0 : n_primary, ns = MODE == 0 ? 0 : stream_size(size_ptr, n_value);
    const int chunks_p = (np + kWave - 1) / kWave, total_chunks = chunks_p + (ns + kWave - 1) / kWave;
    const int stripe = blockIdx.x % kTraceStripes, stripe_waves = (gridDim.x / kTraceStripes) * kPersistWaves;
    int* counter = tickets + stripe * kTraceCounterStride;
    // the wave's rank in its stripe, wave-major (traversal_top.h stripe_rank)
    int t = wave * ((int)gridDim.x / kTraceStripes) + (int)blockIdx.x / kTraceStripes;
    lds_int* col = (lds_int*)lds + wave * (kTopStack + 1) * kWave + lane;
    __syncthreads();
    if (MODE != 1 && threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(&counters[0], (unsigned long long)np);
    for (;;) {
        const int group_first = ((t / 32) * kTraceStripes + stripe) * 32, chunk = group_first + t % 32;
        if (group_first >= total_chunks) break;
        if (chunk < chunks_p) {
            const int i = chunk * kWave + lane;
            if (i < np) {
                CursorStack st; st.init(col, kTopStack, spill, err);
                trace_primary_ray<true, kTopStack>(sc, p, i, &st, nullptr, image);
            }
        } else if (chunk < total_chunks) {
            const int c = chunk - chunks_p, i = c * kWave + lane;
            const int pixel = i < ns ? s.rays.id[i] : -1;
            const unsigned long long live = __ballot(pixel >= 0);
            if (lane == 0 && live) atomicAdd(&counters[4 + (c & 63)], (unsigned long long)__popcll(live));
            bool lit = false;
            if (pixel >= 0) {
                CursorStack st; st.init(col, kTopStack, spill, err);
                lit = !trace_one<true, true, kTopStack>(sc.nodes, sc.tris, load_stream_ray(s.rays, i), st,
                    [](int, int, float, float, float) {}, image);
            }
            film_add_wave(film, pixel, lit, lit ? s.color_r[i] * inv_spp : 0.0f, lit ? s.color_g[i] * inv_spp : 0.0f,
                lit ? s.color_b[i] * inv_spp : 0.0f);
        }
        int t_next = 0;
        if (lane == 0) t_next = atomicAdd(counter, 1);
        t = stripe_waves + __builtin_amdgcn_readfirstlane(t_next);
    }
}

// Persistent form WITH LANE REFILL (rodent_hip_render_trace_refill): a wave does not wait for the last ray of a 64-ray chunk.
// As soon as `refill_idle` of its lanes are idle it retires their rays (shadow rays: film contribution; abandoned rays: deep
// list), draws that many rays from its stripe's counter (ONE atomic per refill) and starts them in the idle lanes; the rest
// keep stepping.  Bounce and shadow rays are incoherent: their step counts inside a chunk differ by an order of magnitude
// and a chunk runs at a quarter of its lanes (standalone, 8 Mi random segments: closest hit +15 %, any hit +22 %,
// profiles/r03_sweep_refill_big_random.log) -- camera rays are not (the same kernel on 16 Mi primary rays: -18 %), and the
// host knows where they are: the stream is [survivors of the last bounce | rays generated for this launch], so draws that
// start inside [coherent_from, n_primary) wait for the whole wave like a chunk does.
// One index space for both lists: g < P = the closest-hit ray g of `p` (P = n_primary rounded up to whole chunks), g >= P the
// shadow ray g - P of `s`; ticket t of stripe s is index ((t / 2048) * 64 + s) * 2048 + t % 2048 (the same 32-chunk groups as
// k_trace_persist); a wave's first 64 tickets are static.  Which rays share a wave changes, what a ray visits does not.
// Lane state next to the ray: `g` = index in the joint space (low 28 bits; -1 = no ray) with bit kFoundBit = "an accepted triangle"
// (what a shadow ray's film contribution depends on; a ray handed to the deep list sets it too: k_trace_deep decides for it).
// Kept small on purpose: 64 VGPRs is the budget of 8 waves per SIMD, and a spill lands inside the step.
struct RefillLane { RayX ray; int top; lds_int* sp; int g; };
constexpr int kFoundBit = 1 << 29, kIndexMask = (1 << 28) - 1;
// SHADOW_ORDER (RODENT_HIP_SHADOW_ORDER): what a node step does with an any-hit (shadow) ray when both children are hit
// -- 0 = the nearer child first, like a closest-hit ray (the reference: src/render/mapping_gpu.impala:47-80 shares the kernel); 1 (the
// default from round 5 on) = child 0 first, no ordering; 2 = the FARTHER child first (the light's side: a shadow ray runs from the surface
// to the light).  Occlusion does not depend on the order, so films and counts do not (checked: scripts/render_rules_check.py).  Measured on
// config 5's frame (profiles/r05_render_rules_check.txt): 1 = +3.0 ... 4.1 %, 2 = +2.7 ... 3.7 % -- an any-hit ray wants an occluder, not
// the nearest one, and the builder orders a node's children by decreasing reference count (bvh.h:215): child 0 is the bigger subtree.
// joint_fetch with the memory records addressed as ONE wave-uniform base (an SGPR pair) + a 32-bit byte offset per lane (global_load ...
// vaddr32, saddr): (traversal_device.h joint_fetch is the form for two arrays anywhere): the scene's node and triangle arrays lie in one
// allocation less than 4 GiB long -- no 64-bit address per lane, no array bases in VGPRs: six registers fewer around the step.
__device__ __forceinline__ void joint_fetch_off(vf4& q0, vf4& q1, vf4& q2, vi2& ids, int& popped, bool in_lds, unsigned lds_addr,
    gbytes base, unsigned off, unsigned off_ids, lds_int* sp) {
    const unsigned long long lds_mask = __ballot(in_lds);
    const unsigned sp_addr = (unsigned)(size_t)sp;
    unsigned long long save;
    asm volatile("s_mov_b64 %[save], exec\n\t"
                 "s_andn2_b64 exec, exec, %[lm]\n\t"
                 "s_cbranch_execz .Ljoint_c_%=\n\t"
                 "global_load_dwordx4 %[q1], %[o], %[sb] offset:16\n\t"
                 "global_load_dwordx4 %[q0], %[o], %[sb]\n\t"
                 "global_load_dwordx4 %[q2], %[o], %[sb] offset:32\n\t"
                 "global_load_dwordx2 %[ch], %[oc], %[sb]\n"
                 ".Ljoint_c_%=:\n\t"
                 "s_and_b64 exec, %[save], %[lm]\n\t"
                 "s_cbranch_execz .Ljoint_d_%=\n\t"
                 "ds_read_b128 %[q0], %[l]\n\t"
                 "ds_read_b128 %[q1], %[l] offset:16\n\t"
                 "ds_read_b128 %[q2], %[l] offset:32\n\t"
                 "ds_read_b64 %[ch], %[l] offset:48\n"
                 ".Ljoint_d_%=:\n\t"
                 "s_mov_b64 exec, %[save]\n\t"
                 "ds_read_b32 %[pop], %[sp]\n\t"
                 "s_waitcnt vmcnt(0) lgkmcnt(0)"
                 : [q0] "=&v"(q0), [q1] "=&v"(q1), [q2] "=&v"(q2), [ch] "=&v"(ids), [pop] "=&v"(popped), [save] "=&s"(save)
                 : [o] "v"(off), [oc] "v"(off_ids), [sb] "s"(base), [l] "v"(lds_addr), [sp] "v"(sp_addr), [lm] "s"(lds_mask)
                 : "memory", "scc");               // (s_andn2_b64 / s_and_b64 write SCC)
}

// The streams arrive as SLABS: one base pointer and the capacity -- array k of a stream is base + k * capacity (carve_primary /
// carve_secondary: 0 id, 1..3 org, 4..6 dir, 7 tmin, 8 tmax; primary 9..13 geom_id, prim_id, t, u, v; secondary 9 prim_id, 10..12 colour).
// As the ABI's structs of 20 + 13 pointers this kernel kept ~45 SGPR pairs alive around its loop, 51 of its SGPRs lived in the lanes of a
// VGPR and came back with v_readlane (VALU instructions) at every refill and every accepted triangle; an array's address is now two SALU
// instructions away from three registers.  The host checks that a stream IS a slab (stream_slab) and sends any other one through
// k_trace_persist.  flags: kHitRecordsAoS (primary).
struct StreamSlab { float* base; int cap; int flags; };
// LAZY_MISS (RODENT_HIP_LAZY_MISS): no miss record when a closest-hit ray starts -- it is stored when the ray retires
// without an accepted triangle (kFoundBit), from the tmax the lane still holds.  In a closed scene nearly every ray finds a triangle and
// the record stored up front was 20 of the ~43 bytes this kernel wrote per ray.
template <int SHADOW_ORDER = 0, bool LAZY_MISS = false>
__global__ __launch_bounds__(kWave * kPersistWaves) __attribute__((amdgpu_waves_per_eu(8, 8)))
void k_trace_refill(SceneDev sc, StreamSlab pslab, int n_primary, int coherent_from, StreamSlab sslab, const int* size_ptr, int n_value,
    float* film, float inv_spp,
                    int* deep_count_primary, int* deep_count_secondary, unsigned long long* counters, int* deep_list_primary,
                        int* deep_list_secondary, int* tickets,
                    int idle_bounce, int idle_shadow, int* spill, int* err, unsigned tri_delta) {
    constexpr int kStackInts = kPersistWaves * (kTopStack + 1) * kWave, kGroupRays = 32 * kWave;
    __shared__ __attribute__((aligned(16))) int lds[kStackInts + kPersistTopNodes * 16];
    const int lane = threadIdx.x % kWave, wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    lds_int* image = (lds_int*)lds + kStackInts;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    for (int j = threadIdx.x; j < kPersistTopNodes * 4; j += kWave * kPersistWaves)
        reinterpret_cast<__attribute__((address_space(3))) i32x4*>(image)[j] = reinterpret_cast<const i32x4*>(sc.top_image_large)[j];
    const int np = n_primary, ns = sslab.base ? stream_size(size_ptr, n_value) : 0;
    const auto P_ = [&](int k) { return pslab.base + (size_t)k * (size_t)pslab.cap; };
    const auto S_ = [&](int k) { return sslab.base + (size_t)k * (size_t)sslab.cap; };
    const auto stream_ray = [&](const float* base, size_t cap, int i) {
        return make_rayx(base[cap + i], base[2 * cap + i], base[3 * cap + i], base[4 * cap + i], base[5 * cap + i], base[6 * cap + i],
            base[7 * cap + i], base[8 * cap + i]);
    };
    const auto hit_record = [&](unsigned i, int geom, int prim, float t, float u, float v) {               // store_hit_record on the slab
        int* g = reinterpret_cast<int*>(P_(9));
        if (pslab.flags & kHitRecordsAoS) {
            int* rec = g + 5u * i;
            *reinterpret_cast<i32x4_dword_aligned*>(rec) = i32x4_dword_aligned{geom, prim, __float_as_int(t), __float_as_int(u)};
            rec[4] = __float_as_int(v);
        } else { g[i] = geom; reinterpret_cast<int*>(P_(10))[i] = prim; P_(11)[i] = t; P_(12)[i] = u; P_(13)[i] = v; }
    };
    const int P = (np + kWave - 1) / kWave * kWave, total = P + ns;
    const int stripe = blockIdx.x % kTraceStripes, stripe_waves = (gridDim.x / kTraceStripes) * kPersistWaves;
    int* counter = tickets + stripe * kTraceCounterStride;
    const auto index_of = [&](int t) { return ((t / kGroupRays) * kTraceStripes + stripe) * kGroupRays + t % kGroupRays; };
    // wave-uniform; lane l's column starts at wave_stack + l
    lds_int* const wave_stack = (lds_int*)lds + wave * (kTopStack + 1) * kWave;
    // sp >= wave_limit  <=>  the lane's cursor is at entry kTopStack (l < kWave)
    lds_int* const wave_limit = wave_stack + kTopStack * kWave;
    __syncthreads();
    if (np && threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(&counters[0], (unsigned long long)np);
    // the scene's nodes and triangles lie in ONE allocation (rodent_hip_scene_create): one wave-uniform base, 32-bit byte offsets per lane
    // (joint_fetch_off). tri_delta = bytes from node id 0's (fictitious) record to triangle 0, checked by the host (bvh_offsets_ok)
    // together with the 24-bit index range of the multiply below
    typedef const __attribute__((address_space(1))) char* gptr;
    const gptr bvh_base = (gptr)reinterpret_cast<const char*>(sc.nodes - 1);                       // node ids are 1-based

    RefillLane L; L.top = 0; L.g = -1; L.sp = wave_stack + lane;
    // rays that ended since the last refill leave the wave: wave-uniform control flow (film_add_wave shuffles)
    const auto retire = [&]() {
        const bool done = L.g >= 0 && L.top == 0;
        const int i = (L.g & kIndexMask) - P;
        const bool lit = done && i >= 0 && !(L.g & kFoundBit);
        if (__ballot(lit))
            film_add_wave(film, lit ? reinterpret_cast<const int*>(S_(0))[i] : -1, lit, lit ? S_(10)[i] * inv_spp : 0.0f,
                lit ? S_(11)[i] * inv_spp : 0.0f, lit ? S_(12)[i] * inv_spp : 0.0f);
        // (tmax: as loaded, canonicalised)
        if (LAZY_MISS && done && i < 0 && !(L.g & kFoundBit)) hit_record((unsigned)L.g, sc.num_materials, -1, L.ray.tmax, 0.0f, 0.0f);
        if (done) L.g = -1;
    };
    const auto start = [&](int g) {
        RayX ray;
        if (g < P) {
            if (g >= np) return;
            ray = stream_ray(pslab.base, (size_t)pslab.cap, g);
            // the miss record; hits overwrite it
            if (!LAZY_MISS) hit_record((unsigned)g, sc.num_materials, -1, ray.tmax, 0.0f, 0.0f);
        } else {
            const int i = g - P;
            if (i >= ns || reinterpret_cast<const int*>(S_(0))[i] < 0) return;
            ray = stream_ray(sslab.base, (size_t)sslab.cap, i);
        }
        ray.tmin = canonical(ray.tmin); ray.tmax = canonical(ray.tmax);
        L.ray = ray; L.g = g;
        L.sp = wave_stack + lane; *L.sp = 0; L.top = kLdsTag;
    };

    // (a prefetched ticket range per wave -- the atomic of the next refill issued at this one -- was measured: 3 % slower, the reserved
    // rays are missing at the launch's end)
    bool more = true, first_draw = true;
    int need_idle = 0;
    for (;;) {
        const unsigned long long live = __ballot(L.top != 0);
        const int idle = kWave - __popcll(live);
        if (more && idle >= need_idle && idle > 0) {
            retire();
            int first;
            // rank in the stripe, wave-major
            if (first_draw) { first = (wave * ((int)gridDim.x / kTraceStripes) + (int)blockIdx.x / kTraceStripes) * kWave;
                first_draw = false; }
            else {
                int f = 0;
                if (lane == 0) f = atomicAdd(counter, idle);
                first = stripe_waves * kWave + __builtin_amdgcn_readfirstlane(f);
            }
            const int g0 = index_of(first);
            // index_of grows with the ticket: once past the end, always past the end
            more = g0 < total;
            need_idle = (g0 >= coherent_from && g0 < P) ? kWave : (g0 < P ? idle_bounce : idle_shadow);   // camera rays: chunk by chunk
            if (L.top == 0) {
                const int g = index_of(first + __popcll(~live & ((1ull << lane) - 1ull)));
                if (g < total) start(g);
            }
            const unsigned long long shadows = __ballot(L.top != 0 && L.g >= P && !((live >> lane) & 1ull));
            if (lane == 0 && shadows) atomicAdd(&counters[4 + ((first / kWave) & 63)], (unsigned long long)__popcll(shadows));
            continue;
        }
        if (live == 0) { retire(); break; }
        if (L.top != 0) {
            // one step of trace_one (the same visit order, the same arithmetic); any hit / closest hit is a property of the lane here
            const int top = L.top;
            const bool is_node = top > 0;
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            typedef int i32x2 __attribute__((ext_vector_type(2)));
            f32x4 q0, q1, q2;
            i32x2 ch;
            int popped;
            {
                // both kinds of fetch -- LDS image, memory -- and the word under the cursor in flight together (joint_fetch_off,
                // traversal_device.h)
                const unsigned idx = (unsigned)(is_node ? top : ~top), stride = is_node ? (unsigned)sizeof(Node2) : (unsigned)sizeof(Tri1);
                // (idx < 2^24: bvh_offsets_ok; a lane on an image link computes an offset nobody loads)
                const unsigned off = __umul24(idx, stride) + (is_node ? 0u : tri_delta);
                joint_fetch_off(q0, q1, q2, ch, popped, top >= kLdsTag, (unsigned)(size_t)image + (unsigned)(top - kLdsTag), bvh_base, off,
                    off + (is_node ? 48u : 40u), L.sp);
            }
            if (is_node) {
                float te0, te1;
                // -(o * 1/d) is computed here, not carried (make_rayx's product, the same value): three multiplications per node step for
                // three registers -- with them in the lane state the register allocator puts three ray components into scratch and reloads
                // them in every step
                RayX rr = L.ray; rr.iox = -(rr.ox * rr.idx); rr.ioy = -(rr.oy * rr.idy); rr.ioz = -(rr.oz * rr.idz);
                const bool h0 = slab_canonical(rr, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, te0) && ch.x != 0;
                const bool h1 = slab_canonical(rr, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, te1) && ch.y != 0;
                bool c0first = te0 < te1;
                if (SHADOW_ORDER != 0) { const bool shadow = (L.g & kIndexMask) >= P;
                    c0first = shadow ? (SHADOW_ORDER == 1 ? true : te0 > te1) : c0first; }
                const bool both = h0 && h1;
                L.top = both ? (c0first ? ch.x : ch.y) : (h0 ? ch.x : (h1 ? ch.y : popped));
                L.sp[kWave] = c0first ? ch.y : ch.x;
                L.sp += (both ? kWave : 0) - ((h0 || h1) ? 0 : kWave);
                // deeper than the window: the oldest entries move out
                if (both && L.sp >= wave_limit) stack_spill<kTopStack>(L.sp, L.top, wave_limit + lane, spill, kPersistWaves, err);
            } else {
                const int prim_id = __float_as_int(q2.w);
                const float nx = cross_x(q1.x, q1.y, q1.z, q2.x, q2.y, q2.z), ny = cross_y(q1.x, q1.y, q1.z, q2.x, q2.y, q2.z),
                    nz = cross_z(q1.x, q1.y, q1.z, q2.x, q2.y, q2.z);
                float t, u, v;
                const bool any = (L.g & kIndexMask) >= P;
                bool found = false;
                if (intersect_tri(L.ray, q0.x, q0.y, q0.z, q1.x, q1.y, q1.z, q2.x, q2.y, q2.z, nx, ny, nz, t, u, v)) {
                    if (!any) {
                        unsigned k = (unsigned)L.g & (unsigned)kIndexMask;
                        asm volatile("" : "+v"(k));
                        hit_record(k, __float_as_int(q1.w), prim_id & 0x7FFFFFFF, t, u, v);
                    }
                    L.ray.tmax = t; found = true; L.g |= kFoundBit;
                }
                const bool leave = prim_id < 0, ends = any && found;
                L.top = ends ? 0 : (leave ? popped : top - 1);
                L.sp -= (leave && !ends) ? kWave : 0;
            }
            // popped row 0 while entries are out: they come back
            if (L.top >= kSpillMark) stack_reload<kTopStack>(L.sp, L.top, wave_limit + lane, spill, kPersistWaves);
        }
    }
}

// The rays the one-chunk stream kernels (k_trace_primary / k_trace_secondary: launches below one resident generation of rays) abandoned
// because their stack outgrew the LDS window, traced again from the root with the reference's 64-entry stack ([entry][lane] in LDS).
// kDeepGroups one-wave workgroups behind every stream traversal launch, each taking every kDeepGroups-th batch of 64 rays (one
// wave until round 4: a launch with many deep rays waited for a serial drain); workgroup 0 also resets the persistent kernels' ticket
// counters and the shader's slot counter, and the last workgroup to finish resets the list (`done`: zero between launches).
constexpr int kDeepGroups = 64;
template <bool SECONDARY>
__global__ __launch_bounds__(kWave) void k_trace_deep(SceneDev sc, PrimaryStream p, SecondaryStream s, float* film, float inv_spp,
    int* err, int* deep_count,
                                                      const int* deep_list, int* done, int* tickets, int* zero_word) {
    __shared__ int stack_lds[kStackCap * kWave];
    if (blockIdx.x == 0) {
        // the persistent kernel's 64 ticket counters, ready for its next launch
        if (tickets) tickets[threadIdx.x * kTraceCounterStride] = 0;
        // primary pass: the slot counter of the shader that follows (fused compaction)
        if (zero_word && threadIdx.x == 0) *zero_word = 0;
    }
    const int count = __hip_atomic_load(deep_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    DeepStack st{(lds_int*)stack_lds + threadIdx.x, err};
    for (int k = blockIdx.x * kWave + threadIdx.x; k < count; k += gridDim.x * kWave) {
        const int i = deep_list[k];
        if (SECONDARY) {
            const bool lit = !trace_one<true>(sc.nodes, sc.tris, load_stream_ray(s.rays, i), st, [](int, int, float, float, float) {});
            if (lit) {
                float* px = film + 3 * (size_t)s.rays.id[i];
                unsafeAtomicAdd(px, s.color_r[i] * inv_spp); unsafeAtomicAdd(px + 1, s.color_g[i] * inv_spp);
                unsafeAtomicAdd(px + 2, s.color_b[i] * inv_spp);
            }
        } else {
            trace_primary_ray(sc, p, i, nullptr, &st);
        }
    }
    __syncthreads();
    // every workgroup has read the count before it counts itself done, so the last one may zero it (no deep rays -- the usual case -- :
    // nobody waits for anybody)
    if (threadIdx.x == 0 && count > 0 && atomicAdd(done, 1) == (int)gridDim.x - 1) {
        __hip_atomic_store(deep_count, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ void k_copy_int(const int* src, int* dst) { *dst = *src; }

// ---------------------------------------------------------------------------------------------
// K5: shading (mapping_gpu.impala:82-134; renderer.impala:69-152).  One launch over the sorted,
// hit-only prefix of the stream; the material is a table entry, not generated code.
// ---------------------------------------------------------------------------------------------
// One path vertex in registers: what a primary-stream entry holds (driver.impala:63-104).
struct PathVertex {
    int pixel; v3 org, dir; int prim, geom; float t, u, v;
    uint32_t rnd; float mis; v3 contrib; int depth;
};
// What shading a vertex produces: an emission sample, at most one shadow ray, at most one continuation.
struct ShadeOut {
    bool emits; v3 emitted;
    bool shadow; v3 s_org, s_dir, s_color;
    bool bounce; v3 b_org, b_dir, contrib; uint32_t rnd; float mis;
};
constexpr float kRayOffset = 0.001f;           // renderer.impala:46

// RODENT_SHADE_HOIST_LIGHT (compile time): 0 = the light record is fetched where it is used, behind the material (until
// round 5); 1 = fetched into registers in front of the hit's gathers; 2 = its two cache lines touched from k_shade (global_load ... lds
// into a sink row: no register) as soon as the path's random state is in (what ships: k_shade 916 -> 851 us per call at 16 spp, frames +0.2
// %; mode 1 spills ten VGPRs and is 11 % slower).  profiles/r06_shade_*.txt
#ifndef RODENT_SHADE_HOIST_LIGHT
#define RODENT_SHADE_HOIST_LIGHT 2
#endif
__device__ __forceinline__ ShadeOut shade_vertex(const SceneDev& sc, const PathVertex& pv, int max_path_len) {
    ShadeOut o;
    const float pdf_lightpick = 1.0f / (float)sc.num_lights;
    uint32_t rnd = pv.rnd;
#if RODENT_SHADE_HOIST_LIGHT == 1
    // The light that next-event estimation will sample depends on the path's random state only (renderer.impala:76-78), not on what was
    // hit: its record is fetched HERE, beside the hit's own gathers (material, triangle record), instead of behind them -- one dependent
    // fetch less in the chain this kernel waits for. Same values, same arithmetic; a specular hit fetches a record it
    // does not use.
    typedef float lf4 __attribute__((ext_vector_type(4)));
    lf4 lv0, lv1, lv2, lnrm, lcol;
    {
        uint32_t peek = rnd;
        const RodentLight* Lh = sc.lights + (int)(xorshift(&peek) & 0x7FFFFFFFu) % sc.num_lights;
        lv0 = *reinterpret_cast<const lf4*>(Lh->v0); lv1 = *reinterpret_cast<const lf4*>(Lh->v1);
        lv2 = *reinterpret_cast<const lf4*>(Lh->v2);
        lnrm = *reinterpret_cast<const lf4*>(Lh->n); lcol = *reinterpret_cast<const lf4*>(Lh->color);
    }
#endif
    RodentMaterial textured;
    const RodentMaterial* m = resolve_material(&sc, sc.materials + pv.geom, &textured, pv.prim, pv.u, pv.v);
    const Surf sf = surface_element(&sc, pv.org, pv.dir, pv.prim, pv.t, pv.u, pv.v);
    const v3 out_dir = neg(pv.dir);

    // on_hit (renderer.impala:113-128)
    o.emitted = V(0, 0, 0);
    o.emits = m->emissive && sf.entering;
    if (o.emits) {
        const RodentLight* L = sc.lights + sc.light_ids[pv.prim];
        const float pdf_dir = cosine_hemisphere_pdf(dot(LD3(L->n), out_dir));
        const v3 intensity = pdf_dir > 0.0f ? LD3(L->color) : V(0, 0, 0);
        const float pdf_area = pdf_dir > 0.0f ? L->inv_area : 1.0f;
        const float next_mis = pv.mis * pv.t * pv.t / dot(out_dir, sf.local.c2);
        const float w = 1.0f / (1.0f + next_mis * pdf_lightpick * pdf_area);
        o.emitted = mulf(mulf(mul(pv.contrib, intensity), w), 1.0f);
    }

    // on_shadow (renderer.impala:69-111)
    o.shadow = false;
    if (!bsdf_is_specular(m)) {
        const int light_id = (int)(xorshift(&rnd) & 0x7FFFFFFFu) % sc.num_lights;
        const float lu = randf(&rnd), lv = randf(&rnd);
#if RODENT_SHADE_HOIST_LIGHT == 1
        (void)light_id;
        const v3 pos = sample_triangle(lu, lv, V(lv0.x, lv0.y, lv0.z), V(lv1.x, lv1.y, lv1.z), V(lv2.x, lv2.y, lv2.z));
        const v3 from_dir = sub(sf.point, pos);
        float lcos = dot(from_dir, V(lnrm.x, lnrm.y, lnrm.z)) / len(from_dir);
        v3 intensity = V(lcol.x, lcol.y, lcol.z); float pdf_area = lnrm.w;
#else
        const RodentLight* L = sc.lights + light_id;
        const v3 pos = sample_triangle(lu, lv, LD3(L->v0), LD3(L->v1), LD3(L->v2));
        const v3 from_dir = sub(sf.point, pos);
        float lcos = dot(from_dir, LD3(L->n)) / len(from_dir);
        v3 intensity = LD3(L->color); float pdf_area = L->inv_area;
#endif
        if (!(pdf_area > 0.0f && cosine_hemisphere_pdf(lcos) > 0.0f && lcos > 0.0f)) { intensity = V(0, 0, 0); pdf_area = 1.0f;
            lcos = 0.0f; }
        const v3 light_dir = sub(pos, sf.point);
        const float vis = dot(light_dir, sf.local.c2);
        if (vis > 0.0f && lcos > 0.0f) {
            const float inv_d = 1.0f / len(light_dir), inv_d2 = inv_d * inv_d;
            const v3 in_dir = mulf(light_dir, inv_d);
            const float pdf_e = bsdf_pdf(m, &sf, in_dir, out_dir);
            const float pdf_l = pdf_area * pdf_lightpick, inv_pdf_l = 1.0f / pdf_l;
            const float cos_e = vis * inv_d, cos_l = lcos;
            const float w = 1.0f / (1.0f + pdf_e * cos_l * inv_d2 * inv_pdf_l);
            const float geom = cos_e * cos_l * inv_d2 * inv_pdf_l;
            o.s_color = mulf(mul(intensity, mul(pv.contrib, bsdf_eval(m, &sf, in_dir, out_dir))), geom * w);
            o.s_org = sf.point; o.s_dir = light_dir;
            o.shadow = true;
        }
    }

    // on_bounce (renderer.impala:130-152)
    const float lum2 = 2.0f * luminance(pv.contrib); const float rr = lum2 > 0.75f ? 0.75f : lum2;
    o.bounce = false;
    if (pv.depth >= max_path_len || randf(&rnd) >= rr) return o;
    const BsdfSample bs = bsdf_sample(m, &sf, &rnd, out_dir);
    o.contrib = mulf(mul(pv.contrib, bs.color), bs.cos / (bs.pdf * rr));
    o.b_org = sf.point; o.b_dir = bs.in_dir;
    o.rnd = rnd; o.mis = bsdf_is_specular(m) ? 0.0f : 1.0f / bs.pdf;
    o.bounce = true;
    return o;
}

// Block-wide exclusive prefix over all EARLIER blocks of a launch, single pass ("decoupled look-back"): every block publishes
// its own total as soon as it knows it (flag A), then adds up its predecessors' words -- 64 at a time, one per lane -- back to
// the nearest block that already knows its inclusive prefix (flag P), and publishes its own.  A word is (flag << 30) | value;
// the array is zeroed before the launch (flag 0 = nothing yet: wait).  A block never waits for a LATER one, so the chain ends as long as
// every earlier block is resident or finished -- which holds because the hardware dispatches workgroups in index order, but is not
// something HIP promises: the wait is therefore BOUNDED (about a second of polling, then the kernel traps and the host reports a HIP
// error instead of hanging).  This is the opt-in mode rodent_hip_render_fused_compact(dev, 1) (reproducible stream order); the default,
// mode 2, takes one atomic per block and waits for nobody.  The values are all that travels (relaxed agent-scope atomics, no fences).
// Called by the first wave of the block; returns the prefix in every lane.
constexpr unsigned kScanA = 1u << 30, kScanP = 2u << 30, kScanValue = (1u << 30) - 1u;
// k_shade's `scan` argument for "slots from one atomic counter" (rodent_hip_render_fused_compact(dev, 2))
unsigned* const kScanAtomic = reinterpret_cast<unsigned*>(8);
__device__ __forceinline__ unsigned lookback_exclusive(unsigned* status, int block, unsigned total) {
    const int lane = threadIdx.x;                                        // 0..63
    if (lane == 0) __hip_atomic_store(&status[block], (block == 0 ? kScanP : kScanA) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (block == 0) return 0u;
    unsigned excl = 0u;
    for (int j = block - 1;; j -= kWave) {
        unsigned long long upto, is_p;
        unsigned w;
        int first_p;
        for (unsigned polls = 0;; polls++) {
            if (polls == (1u << 23)) __builtin_trap();                   // a predecessor that never published: see above
            const int idx = j - lane;
            // in front of block 0: prefix 0
            w = idx >= 0 ? __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kScanP;
            is_p = __ballot((w >> 30) == 2u);
            first_p = is_p ? __ffsll((long long)is_p) - 1 : kWave - 1;
            upto = first_p >= 63 ? ~0ull : ((2ull << first_p) - 1ull);
            if (!(__ballot((w >> 30) == 0u) & upto)) break;              // every word up to the nearest prefix is there
            __builtin_amdgcn_s_sleep(2);
        }
        unsigned v = ((upto >> lane) & 1ull) ? (w & kScanValue) : 0u;
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        excl += v;
        if (is_p) break;
    }
    if (lane == 0) __hip_atomic_store(&status[block], kScanP | (excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return excl;
}

// `perm` != nullptr: the sort did not move the rays, it only computed where each would go (perm[sorted position] = stream
// index); thread i then shades ray perm[i] of `p`.  `perm` == nullptr: thread i shades ray i.
// `scan` == nullptr (the stage-level hip_shade, mapping_gpu.impala:82-134 as it stands): the ray that goes on is written to
// slot i of `q` (q == p: in place), a path that ends leaves id -1 behind and gpu_compact_primary (:267-300) squeezes the
// stream afterwards.  `scan` != nullptr (the renderer's loop): COMPACTION IS PART OF THE SHADER -- the ray that goes on is
// written straight to its compacted slot of `q` (another stream than `p`): slot = rays that go on in earlier blocks
// (lookback_exclusive over the blocks' totals) + those in earlier waves of the block + those in lower lanes.  That is the
// stable order the separate compaction produced, without reading and writing the 15-word stream once more per bounce (the
// compaction's copy was 38 % of the summed kernel time of BASELINE config 4, profiles/r02_render_pmc_digest.txt); the
// new stream size goes to *alive_total.
template <int kBlock /* threads per workgroup = rays per compaction slot request (shade_block()) */>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_shade(SceneDev sc, PrimaryStream p, PrimaryStream q,
    const int* __restrict__ perm, SecondaryStream s, const int* size_ptr, int n_value, float* film,
                                                   float inv_spp, int max_path_len, int unsorted, unsigned* scan, int* alive_total) {
    __shared__ unsigned wave_total[kBlock / kWave + 1];
#if RODENT_SHADE_HOIST_LIGHT == 2
    __shared__ int light_sink[kBlock];
#endif
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const int n_valid = stream_size(size_ptr, n_value);
    const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
    // whole block (whole wave) beyond the stream
    if (scan ? (int)(blockIdx.x * kBlock) >= n_valid : (int)(blockIdx.x * kBlock + wave * kWave) >= n_valid) return;
    // Every lane of the wave reaches the ONE film_add_wave call below (it reduces across lanes with shuffles, which must
    // not read lanes that took another path): lanes beyond the stream and rays that missed take part with nothing to add.
    const bool in_range = i < n_valid;
    const int src = (perm && in_range) ? perm[i] : i;
    const bool live = in_range && !(unsorted && load_hit_geom(p, (unsigned)src) >= sc.num_materials);
    if (in_range && !live) { if (!scan) q.rays.id[i] = -1; s.rays.id[i] = -1; }            // unsorted stream: a ray that missed ends here
    PathVertex pv; pv.pixel = -1; pv.depth = 0;
    ShadeOut o; o.emits = false; o.shadow = false; o.bounce = false; o.emitted = V(0, 0, 0);
    if (live) {
        pv.pixel = p.rays.id[src];
        pv.org = V(p.rays.org_x[src], p.rays.org_y[src], p.rays.org_z[src]);
        pv.dir = V(p.rays.dir_x[src], p.rays.dir_y[src], p.rays.dir_z[src]);
        const HitRecord hit = load_hit_record(p, (unsigned)src);
        pv.prim = hit.prim; pv.geom = hit.geom; pv.t = hit.t; pv.u = hit.u; pv.v = hit.v;
        pv.rnd = p.rnd[src]; pv.mis = p.mis[src];
#if RODENT_SHADE_HOIST_LIGHT == 2
        {
            uint32_t peek = pv.rnd;
            const char* lp = reinterpret_cast<const char*>(sc.lights + (int)(xorshift(&peek) & 0x7FFFFFFFu) % sc.num_lights);
            __attribute__((address_space(3))) void* sink =
                (__attribute__((address_space(3))) void*)(light_sink + (threadIdx.x / kWave) * kWave);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)lp, sink, 4, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(lp + 64), sink, 4, 0, 0);
        }
#endif
        pv.contrib = V(p.contrib_r[src], p.contrib_g[src], p.contrib_b[src]);
        pv.depth = p.depth[src];
        o = shade_vertex(sc, pv, max_path_len);
    }
    film_add_wave(film, pv.pixel, live && o.emits, o.emitted.x * inv_spp, o.emitted.y * inv_spp, o.emitted.z * inv_spp);

    // the secondary ray is written at the SAME index (mapping_gpu.impala:111-115)
    if (live) {
        if (o.shadow) {
            s.rays.org_x[i] = o.s_org.x; s.rays.org_y[i] = o.s_org.y; s.rays.org_z[i] = o.s_org.z;
            s.rays.dir_x[i] = o.s_dir.x; s.rays.dir_y[i] = o.s_dir.y; s.rays.dir_z[i] = o.s_dir.z;
            s.rays.tmin[i] = kRayOffset; s.rays.tmax[i] = 1.0f - kRayOffset;
            s.color_r[i] = o.s_color.x; s.color_g[i] = o.s_color.y; s.color_b[i] = o.s_color.z;
        }
        s.rays.id[i] = o.shadow ? pv.pixel : -1;
    }

    const bool goes_on = live && o.bounce;
    int d = i;
    if (scan) {
        const unsigned long long on = __ballot(goes_on);
        if (lane == 0) wave_total[wave] = (unsigned)__popcll(on);
        __syncthreads();
        unsigned before = 0u, total = 0u;
        for (int w = 0; w < kBlock / kWave; w++) { const unsigned c = wave_total[w]; if (w < wave) before += c; total += c; }
        if (wave == 0) {
            // experiment: one returning atomic per block (order of the blocks = order of arrival)
            if (scan == kScanAtomic) {
                if (lane == 0) wave_total[kBlock / kWave] = total ? (unsigned)atomicAdd(alive_total, (int)total) : 0u;
            } else {
                const unsigned excl = lookback_exclusive(scan, blockIdx.x, total);
                if (lane == 0) {
                    wave_total[kBlock / kWave] = excl;
                    // the stream's last block: the new size
                    if ((int)blockIdx.x == (n_valid - 1) / kBlock) *alive_total = (int)(excl + total);
                }
            }
        }
        __syncthreads();
        d = (int)(wave_total[kBlock / kWave] + before) + __popcll(on & ((1ull << lane) - 1ull));
    } else if (live && !goes_on) q.rays.id[i] = -1;
    if (!goes_on) return;
    if (perm || scan) q.rays.id[d] = pv.pixel;                                             // (in place the id is there already)
    q.rays.org_x[d] = o.b_org.x; q.rays.org_y[d] = o.b_org.y; q.rays.org_z[d] = o.b_org.z;
    q.rays.dir_x[d] = o.b_dir.x; q.rays.dir_y[d] = o.b_dir.y; q.rays.dir_z[d] = o.b_dir.z;
    q.rays.tmin[d] = kRayOffset; q.rays.tmax[d] = FLT_MAX_REF;
    q.rnd[d] = o.rnd; q.mis[d] = o.mis;
    q.contrib_r[d] = o.contrib.x; q.contrib_g[d] = o.contrib.y; q.contrib_b[d] = o.contrib.z; q.depth[d] = pv.depth + 1;
}

// ---------------------------------------------------------------------------------------------
// K8: persistent-threads megakernel (mapping_gpu.impala:371-474): one 64-lane workgroup per film tile
// of ~1024 samples; every lane carries a whole path in registers (traverse, shade, shadow-traverse,
// bounce) and fetches the next (pixel, sample) of the tile when its path ends; the path's colour is
// summed locally and added to the film once (:405-407,442,469).  The workgroup is one wavefront, so
// the reference's LDS work counter (:389-395,410) is a wave-uniform register here and the hand-out is
// a ballot prefix: lane order = sample order, deterministic.
// ---------------------------------------------------------------------------------------------
// CURSOR (what ships; round 3): both traversal loops run on the LDS-only cursor stack of the stream kernels (no depth test in the hot
// loop); a ray that outgrows the 15-entry window is traced again with the 64-entry LDS + scratch stack.  +2 % on config 4 and on the
// atrium against the depth-tested stack (3 216 -> 3 276, 586 -> 599 Msamples/s).
template <bool CURSOR>
__global__ __launch_bounds__(kWave) void k_mega(SceneDev sc, CameraDev cam, float* film, int film_w, int film_h, int y0, int y1, int iter,
    int spp,
                                                int max_path_len, int log2_tile, float inv_spp, int* err, unsigned long long* counters) {
    __shared__ int lds[kLdsStack * kWave];
    const int tile = 1 << log2_tile;
    const int tile_x = blockIdx.x * tile, tile_y = y0 + blockIdx.y * tile;
    const int tile_w = min(film_w - tile_x, tile), tile_h = min(y1 - tile_y, tile);
    const int ray_count = tile_w * tile_h * spp;
    StreamStack st; st.col = (lds_int*)lds + threadIdx.x; st.err = err;
    int next = 0;                                   // wave-uniform
    bool has_path = false;
    PathVertex pv; pv.pixel = -1; pv.org = V(0, 0, 0); pv.dir = V(0, 0, 1); pv.rnd = 0; pv.mis = 0.0f; pv.contrib = V(0, 0, 0);
    pv.depth = 0;
    float tmin = 0.0f;
    v3 final_color = V(0, 0, 0);
    unsigned n_primary = 0, n_shadow = 0;
    for (;;) {
        const unsigned long long need = __ballot(!has_path);
        if (need && next < ray_count) {
            const int id = next + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(need >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)need, 0u));
            next += __popcll(need);
            if (!has_path && id < ray_count) {
                const int ray_id = id / spp, sample = id - ray_id * spp;
                const int in_y = ray_id / tile_w, in_x = ray_id - in_y * tile_w;
                const int x = tile_x + in_x, y = tile_y + in_y;
                pv.dir = emit_sample(cam, iter, film_w, film_h, x, y, sample, &pv.rnd);
                pv.org = LD3(cam.eye);
                pv.pixel = y * film_w + x; pv.mis = 0.0f; pv.contrib = V(1, 1, 1); pv.depth = 0;
                tmin = 0.0f; final_color = V(0, 0, 0);
                has_path = true;
            }
        }
        if (!__ballot(has_path)) break;

        bool done = false;
        ShadeOut o; o.shadow = false; o.s_org = V(0, 0, 0); o.s_dir = V(0, 0, 1); o.s_color = V(0, 0, 0);
        if (has_path) {
            n_primary++;
            const auto on_hit = [&](int prim, int geom, float t, float u, float v) { pv.prim = prim; pv.geom = geom; pv.t = t; pv.u = u;
                pv.v = v; };
            const RayX path_ray = make_rayx(pv.org.x, pv.org.y, pv.org.z, pv.dir.x, pv.dir.y, pv.dir.z, tmin, FLT_MAX_REF);
            bool hit_any;
            if (CURSOR) {
                CursorStack cs; cs.init(st.col, kLdsStack - 1);
                hit_any = trace_one<false>(sc.nodes, sc.tris, path_ray, cs, on_hit);
                // (from the root again: the closest hit is found again)
                if (cs.overflow) hit_any = trace_one<false>(sc.nodes, sc.tris, path_ray, st, on_hit);
            } else hit_any = trace_one<false>(sc.nodes, sc.tris, path_ray, st, on_hit);
            if (!hit_any) done = true;
            else {
                o = shade_vertex(sc, pv, max_path_len);
                if (o.emits) final_color = add(final_color, o.emitted);
                if (o.bounce) { pv.org = o.b_org; pv.dir = o.b_dir; pv.rnd = o.rnd; pv.mis = o.mis; pv.contrib = o.contrib; pv.depth++;
                    tmin = kRayOffset; }
                else done = true;
            }
        }
        if (o.shadow) {
            n_shadow++;
            const RayX shadow_ray = make_rayx(o.s_org.x, o.s_org.y, o.s_org.z, o.s_dir.x, o.s_dir.y, o.s_dir.z, kRayOffset,
                1.0f - kRayOffset);
            const auto nothing = [](int, int, float, float, float) {};
            bool lit;
            if (CURSOR) {
                CursorStack cs; cs.init(st.col, kLdsStack - 1);
                lit = !trace_one<true>(sc.nodes, sc.tris, shadow_ray, cs, nothing);
                if (cs.overflow) lit = !trace_one<true>(sc.nodes, sc.tris, shadow_ray, st, nothing);
            } else lit = !trace_one<true>(sc.nodes, sc.tris, shadow_ray, st, nothing);
            if (lit) final_color = add(final_color, o.s_color);
        }
        film_add_wave(film, pv.pixel, done, final_color.x * inv_spp, final_color.y * inv_spp, final_color.z * inv_spp);
        if (done) has_path = false;
    }
    for (int off = 32; off > 0; off >>= 1) { n_primary += __shfl_xor(n_primary, off); n_shadow += __shfl_xor(n_shadow, off); }
    if (threadIdx.x == 0) {
        const int stripe = (blockIdx.y * gridDim.x + blockIdx.x) & 31;
        atomicAdd(&counters[68 + stripe], (unsigned long long)n_primary);
        atomicAdd(&counters[4 + stripe], (unsigned long long)n_shadow);
    }
}

// Joint form of the megakernel (rodent_hip_render_mega_joint(dev, 1); measured, NOT the default): the shadow ray of a path vertex and
// the path's NEXT ray are traced back to back in one wave-level loop (trace_two) instead of in two loops that each wait for
// their slowest lane; the shader runs between two such loops.  k_mega is issue-bound at 41 % lane utilisation
// (profiles/r03_mega_pmc.txt), and the joint loop does save wave iterations inside a loop -- but a path whose last vertex still
// has a shadow ray pending keeps its lane for one more loop in which it has no path ray to trace, and the phase switch builds
// a second ray (three divisions) inside the loop: config 4 3 225 -> 2 786 Msamples/s, atrium 586 -> 550
// (profiles/r03_render_rates_mega_joint.txt).  Per path the sequence is unchanged (ray, shade, shadow ray, next ray, ...); the
// path's colour is added to the film when both its last ray and its last shadow ray are done.
__global__ __launch_bounds__(kWave) void k_mega_joint(SceneDev sc, CameraDev cam, float* film, int film_w, int film_h, int y0, int y1,
    int iter, int spp,
                                                      int max_path_len, int log2_tile, float inv_spp, int* err,
                                                          unsigned long long* counters) {
    __shared__ int lds[kLdsStack * kWave];
    const int tile = 1 << log2_tile;
    const int tile_x = blockIdx.x * tile, tile_y = y0 + blockIdx.y * tile;
    const int tile_w = min(film_w - tile_x, tile), tile_h = min(y1 - tile_y, tile);
    const int ray_count = tile_w * tile_h * spp;
    StreamStack st; st.col = (lds_int*)lds + threadIdx.x; st.err = err;
    int next = 0;                                   // wave-uniform
    bool has_path = false, has_shadow = false, unpaid = false;      // unpaid: the path's colour has not gone to the film yet
    PathVertex pv; pv.pixel = -1; pv.org = V(0, 0, 0); pv.dir = V(0, 0, 1); pv.rnd = 0; pv.mis = 0.0f; pv.contrib = V(0, 0, 0);
    pv.depth = 0;
    float tmin = 0.0f;
    v3 final_color = V(0, 0, 0), s_dir = V(0, 0, 1), s_color = V(0, 0, 0);
    unsigned n_primary = 0, n_shadow = 0;
    for (;;) {
        const unsigned long long need = __ballot(!has_path && !has_shadow);
        if (need && next < ray_count) {
            const int id = next + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(need >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)need, 0u));
            next += __popcll(need);
            if (!has_path && !has_shadow && id < ray_count) {
                const int ray_id = id / spp, sample = id - ray_id * spp;
                const int in_y = ray_id / tile_w, in_x = ray_id - in_y * tile_w;
                const int x = tile_x + in_x, y = tile_y + in_y;
                pv.dir = emit_sample(cam, iter, film_w, film_h, x, y, sample, &pv.rnd);
                pv.org = LD3(cam.eye);
                pv.pixel = y * film_w + x; pv.mis = 0.0f; pv.contrib = V(1, 1, 1); pv.depth = 0;
                tmin = 0.0f; final_color = V(0, 0, 0);
                has_path = true; unpaid = true;
            }
        }
        if (!__ballot(has_path || has_shadow)) break;

        n_primary += has_path ? 1u : 0u; n_shadow += has_shadow ? 1u : 0u;
        const TwoHits hits = trace_two(sc.nodes, sc.tris, has_shadow, has_path, pv.org.x, pv.org.y, pv.org.z, s_dir.x, s_dir.y, s_dir.z,
            kRayOffset, 1.0f - kRayOffset,
                                       pv.dir.x, pv.dir.y, pv.dir.z, tmin, FLT_MAX_REF, st,
                                       [&](int prim, int geom, float t, float u, float v) { pv.prim = prim; pv.geom = geom; pv.t = t;
                                           pv.u = u; pv.v = v; });
        if (has_shadow && !hits.a_occluded) final_color = add(final_color, s_color);
        has_shadow = false;
        if (has_path) {
            if (!hits.b_hit) has_path = false;
            else {
                const ShadeOut o = shade_vertex(sc, pv, max_path_len);
                if (o.emits) final_color = add(final_color, o.emitted);
                if (o.shadow) { has_shadow = true; s_dir = o.s_dir; s_color = o.s_color; pv.org = o.s_org; }
                if (o.bounce) { pv.org = o.b_org; pv.dir = o.b_dir; pv.rnd = o.rnd; pv.mis = o.mis; pv.contrib = o.contrib; pv.depth++;
                    tmin = kRayOffset; }
                else has_path = false;
            }
        }
        const bool pay = unpaid && !has_path && !has_shadow;
        film_add_wave(film, pv.pixel, pay, final_color.x * inv_spp, final_color.y * inv_spp, final_color.z * inv_spp);
        if (pay) unpaid = false;
    }
    for (int off = 32; off > 0; off >>= 1) { n_primary += __shfl_xor(n_primary, off); n_shadow += __shfl_xor(n_shadow, off); }
    if (threadIdx.x == 0) {
        const int stripe = (blockIdx.y * gridDim.x + blockIdx.x) & 31;
        atomicAdd(&counters[68 + stripe], (unsigned long long)n_primary);
        atomicAdd(&counters[4 + stripe], (unsigned long long)n_shadow);
    }
}

// ---------------------------------------------------------------------------------------------
// K4 / K7: deterministic, stable binning of a primary stream (sort by geometry id, or compaction
// with key = dead ? 1 : 0).  dst(ray) = bin_begin[key] + (rays with that key in earlier blocks)
//                                      + (rank among the block's rays with that key).
// The reference uses one global atomic per ray for both (mapping_gpu.impala:195,217,293), which
// serialises and makes the output order nondeterministic.
// ---------------------------------------------------------------------------------------------
enum KeyMode { KEY_GEOM = 0, KEY_ALIVE = 1 };
__device__ __forceinline__ int stream_key(const PrimaryStream& p, int i, int mode) {
    return mode == KEY_GEOM ? p.geom_id[i] : (p.rays.id[i] >= 0 ? 0 : 1);
}

// in-block rank of thread `tid` among threads with the same key; also leaves the block's per-key counts in cnt[]
__device__ __forceinline__ int block_rank(int key, bool valid, int num_bins, int* cnt /* LDS [kBinBlock / kWave][num_bins] */) {
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    for (int k = threadIdx.x; k < (kBinBlock / kWave) * num_bins; k += kBinBlock) cnt[k] = 0;
    __syncthreads();
    int rank = 0;
    unsigned long long todo = __ballot(valid);
    while (todo) {                                           // one round per distinct key present in the wave
        const int leader = __ffsll((long long)todo) - 1;
        const int k0 = __shfl(key, leader);
        const unsigned long long same = __ballot(valid && key == k0);
        if (valid && key == k0) rank = __popcll(same & ((1ull << lane) - 1ull));
        if (lane == leader) cnt[wave * num_bins + k0] = __popcll(same);
        todo &= ~same;
    }
    __syncthreads();
    if (valid) for (int w = 0; w < wave; w++) rank += cnt[w * num_bins + key];
    return rank;
}

// counts only (no ranks): 256 threads take the kBinBlock rays of one binning block four at a time; one LDS add per wave and
// distinct key.  (As a 1024-thread workgroup with the ranking of k_scatter it waited for sixteen free wave slots on one CU
// while the shadow-ray pass filled the chip on the other stream: 13 -> 98 ms per five cfg4 frames.)
__global__ __launch_bounds__(kBlock) void k_bin_count(PrimaryStream p, const int* size_ptr, int n_value, int mode, int num_bins,
    int num_blocks, int* hist /* [num_bins][num_blocks] */) {
    extern __shared__ int cnt[];
    const int n = stream_size(size_ptr, n_value);
    for (int k = threadIdx.x; k < num_bins; k += kBlock) cnt[k] = 0;
    __syncthreads();
    const int lane = threadIdx.x % kWave;
    for (int r = 0; r < kBinBlock / kBlock; r++) {
        const int i = blockIdx.x * kBinBlock + r * kBlock + threadIdx.x;
        const bool valid = i < n;
        const int key = valid ? stream_key(p, i, mode) : 0;
        unsigned long long todo = __ballot(valid);
        while (todo) {                                       // one round per distinct key present in the wave
            const int leader = __ffsll((long long)todo) - 1;
            const int k0 = __shfl(key, leader);
            const unsigned long long same = __ballot(valid && key == k0);
            if (lane == leader) atomicAdd(&cnt[k0], __popcll(same));
            todo &= ~same;
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < num_bins; k += kBlock) hist[(size_t)k * num_blocks + blockIdx.x] = cnt[k];
}

// one workgroup per bin: exclusive scan of that bin's per-block counts (in place), total -> bin_total[bin].
// Tiles of kBlock x kScanItems consecutive counts (16 per thread as four 16-byte loads), in-wave scan of the thread sums
// by lane shifts, waves joined through LDS, a running carry across tiles: 8 Mi rays = 32 768 counts per bin = 8 tiles.
// (The first version gave each thread one contiguous run of counts and scanned the 256 run sums serially on thread 0,
// the second scanned 256 counts per tile: with one workgroup per bin -- ten on the Cornell box -- both were a chain of
// 128+ dependent steps, 11 % of the frame.)
constexpr int kScanItems = 16;
__global__ __launch_bounds__(kBlock) void k_bin_scan_blocks(int* hist, int num_blocks, int* bin_total) {
    __shared__ int wave_sum[kBlock / kWave];
    int* row = hist + (size_t)blockIdx.x * num_blocks;
    const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
    const bool aligned = (((size_t)row) & 15) == 0;
    int carry = 0;
    for (int tile = 0; tile < num_blocks; tile += kBlock * kScanItems) {
        const int b0 = tile + (int)threadIdx.x * kScanItems;
        int v[kScanItems];
        if (aligned && b0 + kScanItems <= num_blocks) {
#pragma unroll
            for (int k = 0; k < kScanItems / 4; k++) { const int4 x = *reinterpret_cast<const int4*>(row + b0 + 4 * k); v[4 * k] = x.x;
                v[4 * k + 1] = x.y; v[4 * k + 2] = x.z; v[4 * k + 3] = x.w; }
        } else {
#pragma unroll
            for (int k = 0; k < kScanItems; k++) v[k] = b0 + k < num_blocks ? row[b0 + k] : 0;
        }
        int sum = 0;
#pragma unroll
        for (int k = 0; k < kScanItems; k++) { const int x = v[k]; v[k] = sum; sum += x; }         // exclusive within the thread
        int incl = sum;
        for (int o = 1; o < kWave; o <<= 1) { const int up = __shfl_up(incl, o); if (lane >= o) incl += up; }
        if (lane == kWave - 1) wave_sum[wave] = incl;
        __syncthreads();
        int before = 0, total = 0;
        for (int w = 0; w < kBlock / kWave; w++) { const int ws = wave_sum[w]; if (w < wave) before += ws; total += ws; }
        const int base = carry + before + incl - sum;
        if (aligned && b0 + kScanItems <= num_blocks) {
#pragma unroll
            for (int k = 0; k < kScanItems / 4; k++) *reinterpret_cast<int4*>(row + b0 + 4 * k) = make_int4(base + v[4 * k],
                base + v[4 * k + 1], base + v[4 * k + 2], base + v[4 * k + 3]);
        } else {
#pragma unroll
            for (int k = 0; k < kScanItems; k++) if (b0 + k < num_blocks) row[b0 + k] = base + v[k];
        }
        carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) bin_total[blockIdx.x] = carry;
}

// single workgroup: exclusive scan over bins; bin_begin[k], bin_end[k] (= ray_ends of mapping_gpu.impala:203-207)
__global__ void k_bin_scan_bins(const int* bin_total, int num_bins, int* bin_begin, int* bin_end) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { int acc = 0; for (int k = 0; k < num_bins; k++) { bin_begin[k] = acc; acc += bin_total[k];
        bin_end[k] = acc; } }
}

// copy_primary_ray (mapping_gpu.impala:136-164) to the computed slot
__global__ __launch_bounds__(kBinBlock) void k_scatter(PrimaryStream p, PrimaryStream q, const int* size_ptr, int n_value, int mode,
    int num_bins, int num_blocks,
                                                     const int* hist, const int* bin_begin, int keep_hit, int drop_from_bin,
                                                         int copy_interval, int* __restrict__ perm) {
    extern __shared__ int cnt[];
    const int n = stream_size(size_ptr, n_value);
    const int i = blockIdx.x * kBinBlock + threadIdx.x;
    const bool valid = i < n;
    const int key = valid ? stream_key(p, i, mode) : 0;
    // ALL loads first -- and before the in-block ranking, whose two barriers and LDS rounds then overlap their latency -- then
    // all stores.  Written as q.x[d] = p.x[i] pairs the copy compiled to load - wait - store, one word at a time (p and q may
    // alias as far as the compiler knows, so no load moves above the store before it; 14 VGPRs): every word paid a full memory
    // latency and the kernel ran at 1.7 TB/s.
    const bool copy = valid && key < drop_from_bin && !perm;
    int id = 0; float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0, tmin = 0, tmax = 0, t = 0, u = 0, v = 0, mis = 0, cr = 0, cg = 0,
        cb = 0;
    int geom = 0, prim = 0, depth = 0; uint32_t rnd = 0;
    int begin = 0, before = 0;
    if (valid && key < drop_from_bin) { begin = bin_begin[key]; before = hist[(size_t)key * num_blocks + blockIdx.x]; }
    if (copy) {
        id = p.rays.id[i];
        ox = p.rays.org_x[i]; oy = p.rays.org_y[i]; oz = p.rays.org_z[i]; dx = p.rays.dir_x[i]; dy = p.rays.dir_y[i]; dz = p.rays.dir_z[i];
        // the ray interval is dead between the traversal and the shader (which writes a new one for every ray that goes on):
        // the sort before shading (copy_interval == 0) leaves the two words behind -- 10 % of its traffic
        if (copy_interval) { tmin = p.rays.tmin[i]; tmax = p.rays.tmax[i]; }
        if (keep_hit) { geom = p.geom_id[i]; prim = p.prim_id[i]; t = p.t[i]; u = p.u[i]; v = p.v[i]; }
        rnd = p.rnd[i]; mis = p.mis[i]; cr = p.contrib_r[i]; cg = p.contrib_g[i]; cb = p.contrib_b[i]; depth = p.depth[i];
    }
    const int rank = block_rank(key, valid, num_bins, cnt);
    if (!valid || key >= drop_from_bin) return;
    const int d = begin + before + rank;
    if (perm) { perm[d] = i; return; }                          // index-only sort: the consumer gathers (k_shade)
    // (keeps the stores below the loads whatever the optimiser thinks of the pairs)
    asm volatile("" ::: "memory");
    q.rays.id[d] = id;
    q.rays.org_x[d] = ox; q.rays.org_y[d] = oy; q.rays.org_z[d] = oz; q.rays.dir_x[d] = dx; q.rays.dir_y[d] = dy; q.rays.dir_z[d] = dz;
    if (copy_interval) { q.rays.tmin[d] = tmin; q.rays.tmax[d] = tmax; }
    if (keep_hit) { q.geom_id[d] = geom; q.prim_id[d] = prim; q.t[d] = t; q.u[d] = u; q.v[d] = v; }
    q.rnd[d] = rnd; q.mis[d] = mis; q.contrib_r[d] = cr; q.contrib_g[d] = cg; q.contrib_b[d] = cb; q.depth[d] = depth;
}


// ---------------------------------------------------------------------------------------------
// Host side: per-device state (interface.cpp:324-339), stream slabs, the streaming loop
// ---------------------------------------------------------------------------------------------
struct DevScene {
    bool loaded = false;
    int num_nodes = 0;
    // k_trace_refill's addressing of nodes and triangles from one base (rodent_hip_scene_create); 0 = not possible for this scene
    unsigned tri_delta = 0;
    SceneDev dev{};
    std::vector<void*> allocs;
};

struct RenderDevice {
    bool init = false;
    int dev = 0;
    DevScene scene;
    int spp = 4, max_path_len = 64;
    int capacity = 0;                          // rays per stream; 0 = default (env_capacity())
    // 1 = sort hit rays by material before shading (mapping_gpu.impala:166-221), 0 = shade in stream order (default: the shader is ONE
    // table-driven kernel, not a kernel per material, and the sort costs more than the divergence it removes -- 5 ... 22 % of the frame on
    // every scene of profiles/r03_sort_sweep.txt)
    int sort = 0;
    // in effect: 0 = 2-wave traversal workgroups (31-record image), shadow pass on the second stream; 1 = persistent stream traversal
    // kernels (k_trace_persist: 16-wave workgroups, 255-record image, ticket counters); 2 = joint: both passes of a bounce in ONE
    // persistent launch
    int trace_persistent = 0;
    // -1 = per scene (joint for every scene the per-scene mapping rule sends to the streaming loop), 0 / 1 / 2 = the caller's choice
    int trace_persistent_request = -1;
    // render_rows keeps hit records as 20-byte records (store_hit_record); RODENT_HIP_HIT_AOS=0 / rodent_hip_render_hit_records(dev, 0):
    // the ABI's five arrays
    int hit_records_aos = 1;
    // in effect, persistent traversal launches: > 0 = lane refill (k_trace_refill) once that many lanes of a wave are idle (bounce rays /
    // shadow rays); 0 = whole chunks (k_trace_persist)
    int trace_refill = 0, trace_refill_shadow = 0;
    int trace_refill_request[2] = {-1, -1};    // -1 = per scene (resolve_refill), else the caller's thresholds
    int* tickets[2] = {nullptr, nullptr}; int num_cus = 0;
    // 1 = the stream traversal kernels stage the scene's top-of-tree image in LDS (2-wave workgroups); 0 = every node from memory
    int lds_image = 1;
    // 0 = rays are moved by the sort (copy_primary_ray), then shaded in place; 1 = the sort only computes the permutation and the shader
    // gathers through it
    int fused_sort = 0;
    // megakernel: 1 = shadow ray and next path ray of a lane traced in one loop (k_mega_joint: measured slower), 0 = the reference's
    // sequence of loops (k_mega)
    int mega_joint = 0;
    // the shader writes every continuing ray to its compacted slot: 2 = slots from one atomic per block (default), 1 = from a look-back
    // scan (deterministic stream order; measured slower); 0 = shade in place, then the separate compaction pass (mapping_gpu.impala:267-300
    // as it stands)
    int fused_compact = 2;
    unsigned* scan = nullptr; int scan_cap = 0;    // per-block words of the shader's look-back scan (zero before every launch)
    int* perm = nullptr; int perm_cap = 0;     // sorted position -> stream index
    int mapping = 0;                           // in effect: 0 = streaming wavefront (mapping_gpu.impala:308-369), 1 = megakernel (:371-474)
    int mapping_request = -1;                  // -1 = chosen per scene (auto_mapping), 0 / 1 = the caller's choice
    float* film = nullptr; int film_w = 0, film_h = 0;
    float* slab[3] = {nullptr, nullptr, nullptr}; int slab_cap[3] = {0, 0, 0};       // first primary, second primary, secondary
    int* tmp = nullptr; int tmp_cap = 0;
    // rays handed to k_trace_deep: [0] primary, [1] secondary (may run at the same time)
    int* deep_list[2] = {nullptr, nullptr}; int deep_cap[2] = {0, 0};
    int* deep_done[2] = {nullptr, nullptr};        // k_trace_deep's workgroup counters (zero between launches)
    // out-of-window stack entries of the persistent traversal launches: one block per resident wave (stack_spill); [1]: the shadow pass on
    // the second stream
    int* spill[2] = {nullptr, nullptr};
    hipStream_t aux = nullptr;                     // shadow-ray traversal runs here, beside the compaction / next primary pass
    hipEvent_t ev_shade = nullptr, ev_sec = nullptr, ev_copy = nullptr;
    int overlap = 1;                               // 0: everything on the caller's stream
    int* hist = nullptr; size_t hist_cap = 0;
    int* ctl = nullptr;       // [0] primary size, [1] secondary size, [2] error flag, [8..] bin_total, bin_begin, bin_end (kMaxBins each)
    unsigned long long* counters = nullptr;    // [0] primary rays, [1] unused, [2] iterations, [3] generated, [4..67] shadow rays (striped)
    // rodent_hip_render_tiles: its sub-calls after the first ADD to the counters instead of starting them again
    bool counters_continue = false;
    unsigned long long call_iterations = 0, call_generated = 0;
    int* host_pinned = nullptr;
};
RenderDevice g_rdev[16];
std::mutex g_rmutex;
int g_current_dev = 0;
std::vector<float> g_host_film; size_t g_host_w = 0, g_host_h = 0;

// every option of the renderer at its default, or at what its environment variable says (rodent_hip_render_defaults)
void render_defaults(RenderDevice& r) {
    r.hit_records_aos = 1;
    if (const char* e = getenv("RODENT_HIP_HIT_AOS")) r.hit_records_aos = atoi(e) ? 1 : 0;
    r.sort = 0; r.overlap = 1; r.fused_sort = 0; r.fused_compact = 2; r.mega_joint = 0; r.lds_image = 1; r.trace_persistent_request = -1;
    r.mapping_request = -1; r.capacity = 0; r.trace_refill_request[0] = r.trace_refill_request[1] = -1;
    if (const char* e = getenv("RODENT_HIP_SORT")) r.sort = atoi(e) ? 1 : 0;
    if (const char* e = getenv("RODENT_HIP_OVERLAP")) r.overlap = atoi(e) ? 1 : 0;
    if (const char* e = getenv("RODENT_HIP_FUSED_SORT")) r.fused_sort = atoi(e) ? 1 : 0;
    if (const char* e = getenv("RODENT_HIP_FUSED_COMPACT")) r.fused_compact = std::min(2, std::max(0, atoi(e)));
    if (const char* e = getenv("RODENT_HIP_LDS_IMAGE")) r.lds_image = atoi(e) ? 1 : 0;
    if (const char* e = getenv("RODENT_HIP_MEGA_JOINT")) r.mega_joint = atoi(e) ? 1 : 0;
    // "48", "48,32" (bounce rays, shadow rays), "0" = off, "-1" = per scene
    if (const char* e = getenv("RODENT_HIP_TRACE_REFILL")) {
        const char* c = strchr(e, ',');
        const int a = std::min(kWave, std::max(-1, atoi(e))), b = c ? std::min(kWave, std::max(-1, atoi(c + 1))) : a;
        if ((a > 0) != (b > 0) || (a < 0) != (b < 0)) {
            fprintf(stderr, "rodent_hip: RODENT_HIP_TRACE_REFILL=%s: both thresholds must be positive, both 0 (off) or both -1 (per "
            "scene), as for rodent_hip_render_trace_refill\n", e); abort(); }
        if (a > 0 && b > 0) { r.trace_refill_request[0] = a; r.trace_refill_request[1] = b; }
        else r.trace_refill_request[0] = r.trace_refill_request[1] = (a < 0 || b < 0) ? -1 : 0;
    }
    if (const char* e = getenv("RODENT_HIP_TRACE_PERSISTENT")) r.trace_persistent_request = std::min(2, std::max(-1, atoi(e)));
    if (const char* m = getenv("RODENT_HIP_MAPPING")) {
        if (!strcmp(m, "mega") || !strcmp(m, "megakernel") || !strcmp(m, "1")) r.mapping_request = 1;
        else if (!strcmp(m, "streaming") || !strcmp(m, "0")) r.mapping_request = 0;
        else if (strcmp(m, "auto") && strcmp(m, "-1")) {
            fprintf(stderr, "rodent_hip: RODENT_HIP_MAPPING must be 'auto', 'streaming' or 'mega'\n"); abort(); }
    }
    r.mapping = r.mapping_request == 1 ? 1 : 0;
    r.trace_persistent = std::max(0, r.trace_persistent_request);
}

RenderDevice& rdev(int dev) {
    if (dev < 0 || dev >= 16) { fprintf(stderr, "rodent_hip: invalid device index %d\n", dev); abort(); }
    std::lock_guard<std::mutex> lock(g_rmutex);
    RenderDevice& r = g_rdev[dev];
    if (!r.init) {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || dev >= count) {
            fprintf(stderr, "rodent_hip: no HIP device %d (%d visible)\n", dev, count); abort(); }
        HIP_CHECK(hipSetDevice(dev));
        r.dev = dev;
        HIP_CHECK(hipMalloc(&r.ctl, sizeof(int) * (8 + 6 * kMaxBins)));
        HIP_CHECK(hipMemset(r.ctl, 0, sizeof(int) * (8 + 6 * kMaxBins)));
        HIP_CHECK(hipMalloc(&r.counters, sizeof(unsigned long long) * kNumCounters));
        HIP_CHECK(hipMemset(r.counters, 0, sizeof(unsigned long long) * kNumCounters));
        HIP_CHECK(hipHostMalloc(&r.host_pinned, sizeof(int) * (8 + kMaxBins)));
        render_defaults(r);
        // k_scatter ranks inside 1024-ray blocks with one counter row per wave: 16 x num_bins ints of dynamic LDS, 65 600 bytes at the
        // 1025 bins kMaxBins allows -- more than the 64 KB a launch gets without asking
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scatter), hipFuncAttributeMaxDynamicSharedMemorySize,
            (int)(sizeof(int) * (kBinBlock / kWave) * kMaxBins)));
        r.init = true;
    }
    return r;
}

inline int round_cap(int size) { return (size & ~31) + 32; }                 // interface.cpp:359-366

// may_fail: an allocation the device has no room for returns nullptr (the slab is gone then) instead of aborting
float* ensure_slab(RenderDevice& r, int which, int size, int multiplier, bool may_fail = false) {
    const int cap = round_cap(size);
    if (r.slab_cap[which] < cap) {
        HIP_CHECK(hipSetDevice(r.dev));
        if (r.slab[which]) HIP_CHECK(hipFree(r.slab[which]));
        r.slab[which] = nullptr; r.slab_cap[which] = 0;
        const hipError_t e = hipMalloc(&r.slab[which], sizeof(float) * (size_t)cap * multiplier);
        if (e != hipSuccess && may_fail) { (void)hipGetLastError(); r.slab[which] = nullptr; return nullptr; }
        HIP_CHECK(e);
        HIP_CHECK(hipMemset(r.slab[which], 0, sizeof(float) * (size_t)cap * multiplier));
        r.slab_cap[which] = cap;
    }
    return r.slab[which];
}

void carve_rays(RayStream& rays, float* ptr, size_t cap) {                    // interface.cpp:528-538
    rays.id = (int32_t*)ptr; rays.org_x = ptr + 1 * cap; rays.org_y = ptr + 2 * cap; rays.org_z = ptr + 3 * cap;
    rays.dir_x = ptr + 4 * cap; rays.dir_y = ptr + 5 * cap; rays.dir_z = ptr + 6 * cap; rays.tmin = ptr + 7 * cap;
    rays.tmax = ptr + 8 * cap;
}
void carve_primary(PrimaryStream& p, float* ptr, size_t cap) {               // interface.cpp:540-554
    carve_rays(p.rays, ptr, cap);
    p.geom_id = (int32_t*)ptr + 9 * cap; p.prim_id = (int32_t*)ptr + 10 * cap; p.t = ptr + 11 * cap; p.u = ptr + 12 * cap;
    p.v = ptr + 13 * cap;
    p.rnd = (uint32_t*)ptr + 14 * cap; p.mis = ptr + 15 * cap; p.contrib_r = ptr + 16 * cap; p.contrib_g = ptr + 17 * cap;
    p.contrib_b = ptr + 18 * cap;
    p.depth = (int32_t*)ptr + 19 * cap; p.size = 0; p.pad = 0;
}
void carve_secondary(SecondaryStream& s, float* ptr, size_t cap) {           // interface.cpp:556-563
    carve_rays(s.rays, ptr, cap);
    s.prim_id = (int32_t*)ptr + 9 * cap; s.color_r = ptr + 10 * cap; s.color_g = ptr + 11 * cap; s.color_b = ptr + 12 * cap; s.size = 0;
    s.pad = 0;
}

void ensure_deep(RenderDevice& r, int which, int rays) {
    if (r.deep_cap[which] < rays) {
        HIP_CHECK(hipSetDevice(r.dev));
        HIP_CHECK(hipDeviceSynchronize());
        if (r.deep_list[which]) HIP_CHECK(hipFree(r.deep_list[which]));
        HIP_CHECK(hipMalloc(&r.deep_list[which], sizeof(int) * (size_t)rays));
        r.deep_cap[which] = rays;
    }
    if (!r.deep_done[which]) { HIP_CHECK(hipMalloc(&r.deep_done[which], sizeof(int) * 16));
        HIP_CHECK(hipMemset(r.deep_done[which], 0, sizeof(int) * 16)); }
}
int persistent_grid(RenderDevice& r);
int shadow_order() { static const int v = [] { const char* e = getenv("RODENT_HIP_SHADOW_ORDER");
    return e ? std::min(2, std::max(0, atoi(e))) : 1; }(); return v; }
// 1 (default from round 6 on): films and ray counts identical, frame rates +0.1 ... 0.3 % on atrium / gallery / Cornell box
// (profiles/r06_lazy_miss_render.txt), 20 bytes per closest-hit ray less to write; 0 = the record up front (until round 5).  (Through the
// traversal ABI the same idea LOSES 2 ... 8 %: profiles/r06_lazy_miss_traversal.txt.)
int lazy_miss() { static const int v = [] { const char* e = getenv("RODENT_HIP_LAZY_MISS"); return e ? (atoi(e) != 0) : 1; }(); return v; }
#define LAUNCH_TRACE_REFILL(...) do { const int so_ = shadow_order(); const unsigned td_ = r.scene.tri_delta; \
        if (lazy_miss() && so_ == 1) hipLaunchKernelGGL((k_trace_refill<1, true>), __VA_ARGS__, td_); \
        else if (so_ == 1) hipLaunchKernelGGL(k_trace_refill<1>, __VA_ARGS__, td_); \
        else if (so_ == 2) hipLaunchKernelGGL(k_trace_refill<2>, __VA_ARGS__, td_); \
        else hipLaunchKernelGGL(k_trace_refill<0>, __VA_ARGS__, td_); } while (0)
// the spill blocks of a persistent launch on stream `which` (103 MB for the 8192 resident waves of this chip, allocated with the first such
// launch)
int* ensure_spill(RenderDevice& r, int which) {
    if (!r.spill[which]) {
        HIP_CHECK(hipSetDevice(r.dev));
        HIP_CHECK(hipMalloc(&r.spill[which], sizeof(int) * (size_t)persistent_grid(r) * kPersistWaves * kSpillWaveInts));
    }
    return r.spill[which];
}

// Stream traversal launches: the main kernel, then the one-wave kernel for the rays it abandoned.
// ctl words: [2] error flag, [3] primary deep count, [4] secondary deep count, [5] secondary stream size (copy for the aux stream)
void ensure_tickets(RenderDevice& r) {
    for (int k = 0; k < 2; k++)
        if (!r.tickets[k]) {
            HIP_CHECK(hipMalloc(&r.tickets[k], sizeof(int) * kTraceStripes * kTraceCounterStride));
            HIP_CHECK(hipMemset(r.tickets[k], 0, sizeof(int) * kTraceStripes * kTraceCounterStride));
        }
}
int persistent_grid(RenderDevice& r) {
    if (!r.num_cus) { hipDeviceProp_t prop; HIP_CHECK(hipGetDeviceProperties(&prop, r.dev)); r.num_cus = prop.multiProcessorCount; }
    return ((r.num_cus * (32 / kPersistWaves) + kTraceStripes - 1) / kTraceStripes) * kTraceStripes;
}
// k_trace_refill keeps a ray's index in 28 bits of a lane register (the renderer's own streams hold at most 64 Mi rays each; a caller's
// stage-level streams may be larger: those go through k_trace_persist) A stream as k_trace_refill takes it (StreamSlab): the base and the
// capacity of a stream whose arrays lie `cap` words apart in carve order, or ok = false (a caller's stage-level struct may point anywhere:
// such a stream goes through k_trace_persist).  An empty struct (all null) is the slab {null, 0}.
bool stream_slab(const PrimaryStream& p, StreamSlab& out) {
    float* base = (float*)p.rays.id;
    const ptrdiff_t cap = p.rays.org_x - base;
    const float* arrays[20] = {(float*)p.rays.id, p.rays.org_x, p.rays.org_y, p.rays.org_z, p.rays.dir_x, p.rays.dir_y, p.rays.dir_z,
        p.rays.tmin, p.rays.tmax, (float*)p.geom_id, (float*)p.prim_id,
                               p.t, p.u, p.v, (float*)p.rnd, p.mis, p.contrib_r, p.contrib_g, p.contrib_b, (float*)p.depth};
    out = StreamSlab{base, (int)cap, p.pad & kHitRecordsAoS};
    if (!base) return true;
    if (cap <= 0 || cap > 0x7FFFFFFF) return false;
    for (int k = 0; k < 20; k++) if (arrays[k] != base + (size_t)k * (size_t)cap) return false;
    return true;
}
bool stream_slab(const SecondaryStream& s, StreamSlab& out) {
    float* base = (float*)s.rays.id;
    const ptrdiff_t cap = s.rays.org_x - base;
    const float* arrays[13] = {(float*)s.rays.id, s.rays.org_x, s.rays.org_y, s.rays.org_z, s.rays.dir_x, s.rays.dir_y, s.rays.dir_z,
        s.rays.tmin, s.rays.tmax, (float*)s.prim_id, s.color_r, s.color_g, s.color_b};
    out = StreamSlab{base, (int)cap, 0};
    if (!base) return true;
    if (cap <= 0 || cap > 0x7FFFFFFF) return false;
    for (int k = 0; k < 13; k++) if (arrays[k] != base + (size_t)k * (size_t)cap) return false;
    return true;
}
// Threads per workgroup of the shader = rays per slot request of its fused compaction.  The request is one returning atomic on ONE word
// (the new stream size), and one word serves ~88 atomics per microsecond: at 256 rays per request (until round 5) a full 32 Mi-ray stream
// asks 131 072 times, 77 per microsecond of the shader's 1.7 ms. 512: config 5 +1.4 ... 2.0 %, crown +1.8 %, gallery +0.7 %, the Cornell
// box through the streaming loop +8.6 %; 1024 gives half of that back to its two barriers over sixteen waves
// (profiles/r05_shade_block.txt).  RODENT_HIP_SHADE_BLOCK=256|512|1024.
int shade_block() { static const int v = [] { const char* e = getenv("RODENT_HIP_SHADE_BLOCK");
    const int b = e ? atoi(e) : kShadeBlockDefault; return b == 256 || b == 512 || b == 1024 ? b : kShadeBlockDefault; }(); return v; }
template <typename... Args> void launch_k_shade(hipStream_t stream, int rays, Args... args) {
    const int b = shade_block(), blocks = (rays + b - 1) / b;
    if (b == 256) hipLaunchKernelGGL(k_shade<256>, dim3(blocks), dim3(256), 0, stream, args...);
    else if (b == 512) hipLaunchKernelGGL(k_shade<512>, dim3(blocks), dim3(512), 0, stream, args...);
    else hipLaunchKernelGGL(k_shade<1024>, dim3(blocks), dim3(1024), 0, stream, args...);
}
bool refill_indexable(long long n_primary, long long n_secondary) { return n_primary + kWave + n_secondary <= (long long)kIndexMask; }
// coherent_from: the rays [coherent_from, n) were generated for this launch (camera rays), the ones in front of them are what the last
// bounce left; < 0 = the caller does not know (the stage-level hip_traverse_primary): whole chunks through k_trace_persist -- with
// coherent_from = 0 every draw of k_trace_refill would wait for the whole wave anyway, in the heavier loop
void launch_trace_primary(RenderDevice& r, hipStream_t stream, const PrimaryStream& p, int n, int coherent_from = -1) {
    ensure_deep(r, 0, n);
    int* tickets = nullptr;
    if (r.trace_persistent && n >= kPersistMinRays) {
        ensure_tickets(r); tickets = r.tickets[0];
        StreamSlab ps, none{nullptr, 0, 0};
        if (r.trace_refill > 0 && r.scene.tri_delta && coherent_from >= 0 && refill_indexable(n, 0) && stream_slab(p, ps))
            LAUNCH_TRACE_REFILL(dim3(persistent_grid(r)), dim3(kWave * kPersistWaves), 0, stream, r.scene.dev, ps, n, coherent_from, none,
                (const int*)nullptr, 0, (float*)nullptr, 0.0f,
                               r.ctl + 3, r.ctl + 4, r.counters, r.deep_list[0], r.deep_list[1], tickets, r.trace_refill,
                                   r.trace_refill_shadow, ensure_spill(r, 0), r.ctl + 2);
        else hipLaunchKernelGGL(k_trace_persist<0>, dim3(persistent_grid(r)), dim3(kWave * kPersistWaves), 0, stream, r.scene.dev, p, n,
            SecondaryStream{}, (const int*)nullptr, 0, (float*)nullptr, 0.0f,
                           r.ctl + 3, r.ctl + 4, r.counters, r.deep_list[0], r.deep_list[1], tickets, ensure_spill(r, 0), r.ctl + 2);
    } else if (r.lds_image) hipLaunchKernelGGL((k_trace_primary<kTraceWaves, kSceneTopNodes>),
        dim3((n + kTraceWaves * kWave - 1) / (kTraceWaves * kWave)), dim3(kTraceWaves * kWave), 0, stream, r.scene.dev, p,
        (const int*)nullptr, n, r.ctl + 3, r.counters, r.deep_list[0]);
    else hipLaunchKernelGGL((k_trace_primary<1, 0>), dim3((n + kWave - 1) / kWave), dim3(kWave), 0, stream, r.scene.dev, p,
        (const int*)nullptr, n, r.ctl + 3, r.counters, r.deep_list[0]);
    hipLaunchKernelGGL(k_trace_deep<false>, dim3(kDeepGroups), dim3(kWave), 0, stream, r.scene.dev, p, SecondaryStream{}, (float*)nullptr,
        0.0f, r.ctl + 2, r.ctl + 3, r.deep_list[0], r.deep_done[0], tickets, r.ctl + 6);
}
void launch_trace_secondary(RenderDevice& r, hipStream_t stream, const SecondaryStream& s, const int* size_ptr, int max_n, float inv_spp) {
    ensure_deep(r, 1, max_n);
    int* tickets = nullptr;
    if (r.trace_persistent && max_n >= kPersistMinRays) {
        ensure_tickets(r); tickets = r.tickets[1];
        StreamSlab ss, none{nullptr, 0, 0};
        if (r.trace_refill > 0 && r.scene.tri_delta && refill_indexable(0, max_n) && stream_slab(s, ss))
            LAUNCH_TRACE_REFILL(dim3(persistent_grid(r)), dim3(kWave * kPersistWaves), 0, stream, r.scene.dev, none, 0, 0, ss, size_ptr,
                max_n, r.film, inv_spp,
                               r.ctl + 3, r.ctl + 4, r.counters, r.deep_list[0], r.deep_list[1], tickets, r.trace_refill,
                                   r.trace_refill_shadow, ensure_spill(r, 1), r.ctl + 2);
        else hipLaunchKernelGGL(k_trace_persist<1>, dim3(persistent_grid(r)), dim3(kWave * kPersistWaves), 0, stream, r.scene.dev,
            PrimaryStream{}, 0, s, size_ptr, max_n, r.film, inv_spp,
                           r.ctl + 3, r.ctl + 4, r.counters, r.deep_list[0], r.deep_list[1], tickets, ensure_spill(r, 1), r.ctl + 2);
    } else if (r.lds_image) hipLaunchKernelGGL((k_trace_secondary<kTraceWaves, kSceneTopNodes>),
        dim3((max_n + kTraceWaves * kWave - 1) / (kTraceWaves * kWave)), dim3(kTraceWaves * kWave), 0, stream, r.scene.dev, s, size_ptr,
        max_n, r.film, inv_spp, r.ctl + 4, r.counters, r.deep_list[1]);
    else hipLaunchKernelGGL((k_trace_secondary<1, 0>), dim3((max_n + kWave - 1) / kWave), dim3(kWave), 0, stream, r.scene.dev, s, size_ptr,
        max_n, r.film, inv_spp, r.ctl + 4, r.counters, r.deep_list[1]);
    hipLaunchKernelGGL(k_trace_deep<true>, dim3(kDeepGroups), dim3(kWave), 0, stream, r.scene.dev, PrimaryStream{}, s, r.film, inv_spp,
        r.ctl + 2, r.ctl + 4, r.deep_list[1], r.deep_done[1], tickets, (int*)nullptr);
}

// Joint form: the closest-hit pass over `p` (n rays) and the shadow pass over `s` (size *size_ptr, or max_n) in ONE persistent launch,
// then the two follow-up kernels for the rays either pass abandoned.
void launch_trace_joint(RenderDevice& r, hipStream_t stream, const PrimaryStream& p, int n, int coherent_from, const SecondaryStream& s,
    const int* size_ptr, int max_n, float inv_spp) {
    ensure_deep(r, 0, n); ensure_deep(r, 1, max_n);
    ensure_tickets(r);
    StreamSlab ps, ss;
    if (r.trace_refill > 0 && r.scene.tri_delta && refill_indexable(n, max_n) && stream_slab(p, ps) && stream_slab(s, ss))
        LAUNCH_TRACE_REFILL(dim3(persistent_grid(r)), dim3(kWave * kPersistWaves), 0, stream, r.scene.dev, ps, n, coherent_from, ss,
            size_ptr, max_n, r.film, inv_spp,
                           r.ctl + 3, r.ctl + 4, r.counters, r.deep_list[0], r.deep_list[1], r.tickets[0], r.trace_refill,
                               r.trace_refill_shadow, ensure_spill(r, 0), r.ctl + 2);
    else hipLaunchKernelGGL(k_trace_persist<2>, dim3(persistent_grid(r)), dim3(kWave * kPersistWaves), 0, stream, r.scene.dev, p, n, s,
        size_ptr, max_n, r.film, inv_spp,
                       r.ctl + 3, r.ctl + 4, r.counters, r.deep_list[0], r.deep_list[1], r.tickets[0], ensure_spill(r, 0), r.ctl + 2);
    // (nothing is abandoned by the persistent kernels any more: the follow-up kernel is the launch's housekeeping -- ticket counters, the
    // shader's slot counter)
    hipLaunchKernelGGL(k_trace_deep<false>, dim3(1), dim3(kWave), 0, stream, r.scene.dev, p, SecondaryStream{}, (float*)nullptr, 0.0f,
        r.ctl + 2, r.ctl + 3, r.deep_list[0], r.deep_done[0], r.tickets[0], r.ctl + 6);
}

void ensure_hist(RenderDevice& r, size_t ints) {
    if (r.hist_cap < ints) {
        if (r.hist) HIP_CHECK(hipFree(r.hist));
        HIP_CHECK(hipMalloc(&r.hist, sizeof(int) * ints));
        r.hist_cap = ints;
    }
}

// two sets of bin arrays: set 0 = sort by geometry, set 1 = compaction (the compaction reads its input size from set 0)
int* bin_total(RenderDevice& r, int set) { return r.ctl + 8 + set * 3 * kMaxBins; }
int* bin_begin(RenderDevice& r, int set) { return r.ctl + 8 + set * 3 * kMaxBins + kMaxBins; }
int* bin_end(RenderDevice& r, int set)   { return r.ctl + 8 + set * 3 * kMaxBins + 2 * kMaxBins; }

// Bins `p` (size = *size_ptr if given, else max_n; never more than max_n) into `q`; bins >= drop_from_bin are not copied.
void bin_stream(RenderDevice& r, int set, const PrimaryStream& p, const PrimaryStream& q, const int* size_ptr, int max_n, int mode,
    int num_bins,
                int keep_hit, int drop_from_bin, hipStream_t stream, int copy_interval = 1, int* perm = nullptr) {
    const int blocks = std::max(1, (max_n + kBinBlock - 1) / kBinBlock);
    ensure_hist(r, (size_t)num_bins * blocks);
    const size_t lds = sizeof(int) * (kBinBlock / kWave) * num_bins;
    hipLaunchKernelGGL(k_bin_count, dim3(blocks), dim3(kBlock), sizeof(int) * num_bins, stream, p, size_ptr, max_n, mode, num_bins, blocks,
        r.hist);
    hipLaunchKernelGGL(k_bin_scan_blocks, dim3(num_bins), dim3(kBlock), 0, stream, r.hist, blocks, bin_total(r, set));
    hipLaunchKernelGGL(k_bin_scan_bins, dim3(1), dim3(1), 0, stream, bin_total(r, set), num_bins, bin_begin(r, set), bin_end(r, set));
    hipLaunchKernelGGL(k_scatter, dim3(blocks), dim3(kBinBlock), lds, stream, p, q, size_ptr, max_n, mode, num_bins, blocks, r.hist,
        bin_begin(r, set), keep_hit, drop_from_bin, copy_interval, perm);
    HIP_CHECK(hipGetLastError());
}

CameraDev to_cam(const Settings* s) {
    CameraDev c;
    c.eye[0] = s->eye.x; c.eye[1] = s->eye.y; c.eye[2] = s->eye.z; c.dir[0] = s->dir.x; c.dir[1] = s->dir.y; c.dir[2] = s->dir.z;
    c.up[0] = s->up.x; c.up[1] = s->up.y; c.up[2] = s->up.z; c.right[0] = s->right.x; c.right[1] = s->right.y; c.right[2] = s->right.z;
    c.w = s->width; c.h = s->height;
    return c;
}

void require_scene(RenderDevice& r) {
    if (!r.scene.loaded) { fprintf(stderr, "rodent_hip: no scene loaded on device %d (call rodent_hip_scene_create)\n", r.dev); abort(); }
    if (r.scene.dev.num_lights <= 0) { fprintf(stderr, "rodent_hip: the scene has no light source\n"); abort(); }
    if (!r.film) { fprintf(stderr, "rodent_hip: no film (call setup_interface)\n"); abort(); }
}

void ensure_film(RenderDevice& r) {
    if (r.film && r.film_w == (int)g_host_w && r.film_h == (int)g_host_h) return;
    HIP_CHECK(hipSetDevice(r.dev));
    if (r.film) HIP_CHECK(hipFree(r.film));
    r.film_w = (int)g_host_w; r.film_h = (int)g_host_h;
    HIP_CHECK(hipMalloc(&r.film, sizeof(float) * 3 * (size_t)r.film_w * r.film_h));
    HIP_CHECK(hipMemset(r.film, 0, sizeof(float) * 3 * (size_t)r.film_w * r.film_h));
}

// gpu_streaming_trace (mapping_gpu.impala:308-369) for image rows [y0, y1) -- or, with tile_rows > 0, for (y1 - y0) rows that are
// row tiles of tile_rows rows each, the first at row y0, the next stride_rows further down, ... (y1 - y0 a multiple of tile_rows)
void render_rows(RenderDevice& r, const Settings* settings, int iter, int y0, int y1, hipStream_t stream, int tile_rows = 0,
    int stride_rows = 0) {
    HIP_CHECK(hipSetDevice(r.dev));
    ensure_film(r);
    require_scene(r);
    const int G = r.scene.dev.num_materials;
    if (G + 1 > kMaxBins) { fprintf(stderr, "rodent_hip: too many geometries (%d)\n", G); abort(); }
    PrimaryStream a, b; SecondaryStream sec;
    // rays per stream: the configured capacity, but no more than this call can ever have in flight (a 256 x 144 frame does not reserve 7
    // GB)
    int kCapacity = (int)std::min<long long>(r.capacity > 0 ? r.capacity : env_capacity(),
        std::max<long long>(64, ((long long)r.spp * r.film_w * std::max(0, y1 - y0) + 63) / 64 * 64));
    // The default capacity (32 Mi rays: 7.1 GB of streams) is a choice of speed, not a need: when the device has no room for it (other
    // tenants, a smaller board) the streams are halved until they fit -- results do not depend on the capacity.  A capacity
    // the caller asked for (rodent_hip_render_capacity) is taken literally: failing to get it aborts with HIP's message.
    for (;;) {
        const bool may_fail = r.capacity <= 0 && kCapacity > (1 << 20);
        if (ensure_slab(r, 0, kCapacity, 20, may_fail) && ensure_slab(r, 1, kCapacity, 20, may_fail)
            && ensure_slab(r, 2, kCapacity, 13, may_fail)) break;
        fprintf(stderr, "rodent_hip: no room for ray streams of %d rays on device %d, trying %d\n", kCapacity, r.dev, kCapacity / 2);
        kCapacity /= 2;
    }
    carve_primary(a, ensure_slab(r, 0, kCapacity, 20), round_cap(kCapacity));
    carve_primary(b, ensure_slab(r, 1, kCapacity, 20), round_cap(kCapacity));
    // hit records as 20-byte records instead of five arrays (store_hit_record) -- unless the sort by material runs: its kernels move and
    // read the arrays
    a.pad = b.pad = (r.hit_records_aos && !r.sort) ? kHitRecordsAoS : 0;
    carve_secondary(sec, ensure_slab(r, 2, kCapacity, 13), round_cap(kCapacity));
    PrimaryStream* primary = &a; PrimaryStream* other = &b;
    int* err = r.ctl + 2;
    const CameraDev cam = to_cam(settings);
    const float inv_spp = 1.0f / (float)r.spp;
    const long long num_rays = (long long)r.spp * r.film_w * (y1 - y0);
    // ray ids are 32-bit in the stream kernels (as in the reference, mapping_gpu.impala:236-241)
    if (num_rays > 0x7FFFFFFFll) {
        fprintf(stderr, "rodent_hip: spp x width x rows = %lld samples in one call exceeds 2^31 - 1; render fewer rows per call or more "
        "frames of fewer spp\n", num_rays); abort(); }
    const int first_pixel = y0 * r.film_w;
    // pixels are generated block by block where the film allows it (whole blocks across, whole tiles down)
    const int block = pixel_block() > 0 && r.film_w % pixel_block() == 0
        && (tile_rows == 0 || (tile_rows % pixel_block() == 0 && r.film_h % tile_rows == 0)) ? pixel_block() : 0;
    long long id = 0; int size = 0;
    HIP_CHECK(hipMemsetAsync(r.ctl, 0, sizeof(int) * 8, stream));
    if (!r.counters_continue) { HIP_CHECK(hipMemsetAsync(r.counters, 0, sizeof(unsigned long long) * kNumCounters, stream));
        r.call_iterations = r.call_generated = 0; }
    unsigned long long iterations = 0, generated = 0;
    const int* d_valid = bin_end(r, 0) + (G - 1);      // rays that hit something = exclusive end of the last geometry bin (:347-357)
    // Shadow rays are independent of what follows the shader on the primary stream (compaction, regeneration, the next
    // closest-hit pass and sort): they are traced on a second HIP stream and joined again before the next shader run
    // overwrites the secondary stream.  The latency-bound traversal then shares the chip with the HBM-bound stream copies.
    // the shadow pass of an iteration rides in the next iteration's closest-hit launch: one stream
    const bool joint = r.trace_persistent == 2;
    const bool overlap = r.overlap != 0 && !joint;
    bool shadow_pending = false; const int* shadow_size_ptr = nullptr; int shadow_n = 0;
    if (overlap && !r.aux) {
        HIP_CHECK(hipStreamCreateWithFlags(&r.aux, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&r.ev_shade, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&r.ev_sec, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&r.ev_copy, hipEventDisableTiming));
    }
    ensure_deep(r, 0, kCapacity); ensure_deep(r, 1, kCapacity);       // before the loop: growing them synchronises the device
    if (r.perm_cap < round_cap(kCapacity)) {
        HIP_CHECK(hipDeviceSynchronize());
        if (r.perm) HIP_CHECK(hipFree(r.perm));
        HIP_CHECK(hipMalloc(&r.perm, sizeof(int) * (size_t)round_cap(kCapacity)));
        r.perm_cap = round_cap(kCapacity);
    }
    const bool fused = r.fused_compact != 0;
    int* d_alive = r.ctl + 6;                            // fused compaction: the shader's last block leaves the new stream size here
    if (fused && r.scan_cap < (kCapacity + kBlock - 1) / kBlock) {
        HIP_CHECK(hipDeviceSynchronize());
        if (r.scan) HIP_CHECK(hipFree(r.scan));
        r.scan_cap = (kCapacity + kBlock - 1) / kBlock;
        HIP_CHECK(hipMalloc(&r.scan, sizeof(unsigned) * (size_t)r.scan_cap));
    }
    // the shader, either in place (then the compaction pass follows) or compacting into the other stream itself
    const auto shade = [&](const PrimaryStream& from, const PrimaryStream& to, const int* perm, const int* size_ptr, int n_value,
        int unsorted, int blocks) {
        if (fused) {
            if (r.fused_compact != 2) HIP_CHECK(hipMemsetAsync(r.scan, 0, sizeof(unsigned) * (size_t)blocks, stream));
            // (d_alive was zeroed by the primary pass's follow-up kernel, k_trace_deep<false>)
        }
        launch_k_shade(stream, blocks * kBlock, r.scene.dev, from, to, perm, sec, size_ptr, n_value, r.film, inv_spp, r.max_path_len,
            unsorted,
                       fused ? (r.fused_compact == 2 ? kScanAtomic : r.scan) : (unsigned*)nullptr, d_alive);
    };
    while (id < num_rays || size > 0) {
        // [0, survivors): what the last bounce left; behind them the rays generated now
        const int survivors = size;
        if (size < kCapacity && id < num_rays) {                                         // regenerate (mapping_gpu.impala:332-336)
            const int n = (int)std::min<long long>(num_rays - id, kCapacity - size);
            hipLaunchKernelGGL(k_generate, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, *primary, size, (int)id, n, cam, iter,
                               r.film_w, r.film_h, first_pixel, r.spp, tile_rows * r.film_w, stride_rows * r.film_w, block,
                                   tile_rows > 0 ? tile_rows : y1 - y0);
            id += n; size += n; generated += n;
        }
        const int blocks = (size + kBlock - 1) / kBlock;
        if (joint && shadow_pending && size >= kPersistMinRays) launch_trace_joint(r, stream, *primary, size, survivors, sec,
            shadow_size_ptr, shadow_n, inv_spp);
        else {
            if (joint && shadow_pending) launch_trace_secondary(r, stream, sec, shadow_size_ptr, shadow_n, inv_spp);
            launch_trace_primary(r, stream, *primary, size, survivors);
        }
        shadow_pending = false;
        // the shadow pass of this iteration: on the second stream, behind the shader on this one, or (joint) inside the next closest-hit
        // launch
        const auto shadow_pass = [&](const int* size_ptr) {
            if (joint) { shadow_pending = true; shadow_size_ptr = size_ptr; shadow_n = size; }
            else launch_trace_secondary(r, overlap ? r.aux : stream, sec, size_ptr, size, inv_spp);
        };
        if (r.sort) {
            // the aux stream has its copy of the previous valid count
            if (overlap && iterations) HIP_CHECK(hipStreamWaitEvent(stream, r.ev_copy, 0));
            if (r.fused_sort) {
                // sort by geometry WITHOUT moving the rays: the binning kernels only compute the permutation, the shader gathers
                // through it and writes the sorted, shaded stream (misses, bin G, are not in the permutation: dropped, :347-357)
                bin_stream(r, 0, *primary, *other, nullptr, size, KEY_GEOM, G + 1, 1, G, stream, 0, r.perm);
                // the previous shadow rays have been traced
                if (overlap && iterations) HIP_CHECK(hipStreamWaitEvent(stream, r.ev_sec, 0));
                shade(*primary, *other, r.perm, d_valid, 0, 0, blocks);
                std::swap(primary, other);
            } else {
                // misses (bin G) are dropped (:347-357); tmin / tmax stay behind
                bin_stream(r, 0, *primary, *other, nullptr, size, KEY_GEOM, G + 1, 1, G, stream, 0);
                std::swap(primary, other);
                // the previous shadow rays have been traced
                if (overlap && iterations) HIP_CHECK(hipStreamWaitEvent(stream, r.ev_sec, 0));
                if (fused) { shade(*primary, *other, nullptr, d_valid, 0, 0, blocks); std::swap(primary, other); }
                else shade(*primary, *primary, nullptr, d_valid, 0, 0, blocks);
            }
            if (overlap) {
                HIP_CHECK(hipEventRecord(r.ev_shade, stream));
                HIP_CHECK(hipStreamWaitEvent(r.aux, r.ev_shade, 0));
                hipLaunchKernelGGL(k_copy_int, dim3(1), dim3(1), 0, r.aux, d_valid, r.ctl + 5);
                HIP_CHECK(hipEventRecord(r.ev_copy, r.aux));
                shadow_pass(r.ctl + 5);
                HIP_CHECK(hipEventRecord(r.ev_sec, r.aux));
            } else shadow_pass(d_valid);
            if (!fused) bin_stream(r, 1, *primary, *other, d_valid, size, KEY_ALIVE, 2, 0, 1, stream);       // compaction (:267-300)
        } else {                                     // option: no sort by material -- shade in stream order, misses end in the shader
            if (overlap && iterations) HIP_CHECK(hipStreamWaitEvent(stream, r.ev_sec, 0));
            if (fused) { shade(*primary, *other, nullptr, nullptr, size, 1, blocks); std::swap(primary, other); }
            else shade(*primary, *primary, nullptr, nullptr, size, 1, blocks);
            if (overlap) {
                HIP_CHECK(hipEventRecord(r.ev_shade, stream));
                HIP_CHECK(hipStreamWaitEvent(r.aux, r.ev_shade, 0));
            }
            shadow_pass(nullptr);
            if (overlap) HIP_CHECK(hipEventRecord(r.ev_sec, r.aux));
            if (!fused) bin_stream(r, 1, *primary, *other, nullptr, size, KEY_ALIVE, 2, 0, 1, stream);
        }
        if (!fused) std::swap(primary, other);
        HIP_CHECK(hipMemcpyAsync(r.host_pinned, fused ? d_alive : bin_end(r, 1), sizeof(int), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        size = r.host_pinned[0];
        iterations++;
    }
    // joint: the last iteration's shadow rays
    if (shadow_pending) launch_trace_secondary(r, stream, sec, shadow_size_ptr, shadow_n, inv_spp);
    if (overlap && iterations) HIP_CHECK(hipStreamWaitEvent(stream, r.ev_sec, 0));           // the last shadow rays belong to this call
    r.call_iterations += iterations; r.call_generated += generated;
    const unsigned long long host_counts[2] = {r.call_iterations, r.call_generated};
    HIP_CHECK(hipMemcpyAsync(r.counters + 2, host_counts, sizeof(host_counts), hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(r.host_pinned + 2, err, sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    if (r.host_pinned[2]) { fprintf(stderr, "rodent_hip: traversal stack overflow in the renderer\n"); abort(); }
}

// gpu_mega_kernel_trace (mapping_gpu.impala:371-474) for image rows [y0, y1)
void render_rows_mega(RenderDevice& r, const Settings* settings, int iter, int y0, int y1, hipStream_t stream) {
    HIP_CHECK(hipSetDevice(r.dev));
    ensure_film(r);
    require_scene(r);
    int ilog2_spp = 0; while ((2 << ilog2_spp) <= r.spp) ilog2_spp++;                 // common.impala ilog2
    const int log2_tile = std::max(0, (10 - ilog2_spp) / 2);                          // tiles of ~2^10 samples (:374-377)
    const int tile = 1 << log2_tile;
    const dim3 grid((r.film_w + tile - 1) >> log2_tile, (y1 - y0 + tile - 1) >> log2_tile);
    int* err = r.ctl + 2;
    HIP_CHECK(hipMemsetAsync(r.ctl, 0, sizeof(int) * 3, stream));
    if (!r.counters_continue) { HIP_CHECK(hipMemsetAsync(r.counters, 0, sizeof(unsigned long long) * kNumCounters, stream));
        r.call_iterations = r.call_generated = 0; }
    if (y1 > y0) {
        if (r.mega_joint) hipLaunchKernelGGL(k_mega_joint, grid, dim3(kWave), 0, stream, r.scene.dev, to_cam(settings), r.film, r.film_w,
            r.film_h, y0, y1, iter, r.spp,
                                             r.max_path_len, log2_tile, 1.0f / (float)r.spp, err, r.counters);
        else hipLaunchKernelGGL(k_mega<true>, grid, dim3(kWave), 0, stream, r.scene.dev, to_cam(settings), r.film, r.film_w, r.film_h, y0,
            y1, iter, r.spp,
                                r.max_path_len, log2_tile, 1.0f / (float)r.spp, err, r.counters);
    }
    HIP_CHECK(hipGetLastError());
    r.call_iterations += 1ull; r.call_generated += (unsigned long long)r.spp * r.film_w * (unsigned long long)(y1 - y0);
    const unsigned long long host_counts[2] = {r.call_iterations, r.call_generated};
    HIP_CHECK(hipMemcpyAsync(r.counters + 2, host_counts, sizeof(host_counts), hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(r.host_pinned + 2, err, sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    if (r.host_pinned[2]) { fprintf(stderr, "rodent_hip: traversal stack overflow in the renderer\n"); abort(); }
}

void render_rows_any(RenderDevice& r, const Settings* settings, int iter, int y0, int y1, hipStream_t stream) {
    if (r.mapping == 1) render_rows_mega(r, settings, iter, y0, y1, stream);
    else render_rows(r, settings, iter, y0, y1, stream);
}

// Which mapping renders a scene when the caller leaves the choice to the library (rodent_hip_render_mapping(dev, -1), the
// default).  What separates the two on this chip is whether the hierarchy stays in the caches next to the CUs: then a path's
// traversal steps cost little, the streaming loop's per-bounce stream traffic (sort + shade + regenerate, ~500 bytes per
// ray and bounce) is the frame, and the megakernel -- which keeps a path in registers -- wins (Cornell box, 22 nodes:
// 3.2 against 2.2 Gsamples/s); once traversal dominates, the streaming loop's coherent, sorted wavefronts win (atrium,
// 142 444 nodes: 0.74 against 0.59).  Measured in between (profiles/r03_mapping_sweep.txt: the atrium with every 2nd ... 512th
// face, 76 494 ... 306 nodes): the streaming loop wins all of them, by 5 % at 306 nodes and by 40-50 % from 5 000 nodes on;
// only hierarchies of a few dozen nodes (the whole tree inside the traversal kernels' LDS image) go to the megakernel.
int auto_mega_max_nodes() {
    static const int v = [] { const char* e = getenv("RODENT_HIP_AUTO_MEGA_MAX_NODES"); return e ? atoi(e) : 128; }();
    return v;
}
int resolve_mapping(const RenderDevice& r) {
    if (r.mapping_request >= 0) return r.mapping_request;
    return r.scene.loaded && r.scene.num_nodes <= auto_mega_max_nodes() ? 1 : 0;
}
// The stream traversal launches of the streaming loop when the caller leaves the choice to the library: the joint persistent launch
// for every hierarchy above the megakernel's size class (+3 ... +9 % on the atrium at 306 ... 142 444 nodes, profiles/r03_joint_sweep.txt),
// the 2-wave kernels with the shadow pass on a second stream for a tree of a few dozen nodes (Cornell box: joint -5 %).
int resolve_trace(const RenderDevice& r) {
    if (r.trace_persistent_request >= 0) return r.trace_persistent_request;
    return r.scene.loaded && r.scene.num_nodes > auto_mega_max_nodes() ? 2 : 0;
}
// Lane refill in those launches when the caller leaves it to the library: on for hierarchies of 16 Ki nodes and more.  What it buys is the
// spread of the rays' step counts inside a wave (a bounce or shadow ray of the atrium takes between 3 and 150 steps); in a small tree
// every ray is short and the refill's own cost (one atomic, eight loads and a reciprocal per started ray, three more multiplications
// per node step) is not paid back -- profiles/r03_refill_sweep.txt: +8 % at 142 444 nodes, +16 % at 76 494, +8 % at 19 862, +1 % at 9 870,
// -4 ... -11 % from 4 951 nodes down.
constexpr int kRefillMinNodes = 16384, kRefillIdleLanes = 40;
void resolve_refill(RenderDevice& r) {
    if (r.trace_refill_request[0] >= 0) { r.trace_refill = r.trace_refill_request[0]; r.trace_refill_shadow = r.trace_refill_request[1];
        return; }
    r.trace_refill = r.trace_refill_shadow = r.scene.loaded && r.scene.num_nodes >= kRefillMinNodes ? kRefillIdleLanes : 0;
}

template <typename T> T* upload(DevScene& s, const T* host, size_t count) {
    T* d = nullptr;
    HIP_CHECK(hipMalloc(&d, std::max<size_t>(sizeof(T) * count, 16)));
    if (count) HIP_CHECK(hipMemcpy(d, host, sizeof(T) * count, hipMemcpyHostToDevice));
    s.allocs.push_back(d);
    return d;
}

// host-side stream slabs of rodent_cpu_get_*_stream: per thread, like the reference's (interface.cpp:341-342,507-508)
thread_local std::vector<float> t_cpu_primary, t_cpu_secondary;

} // namespace

void rodent_services_cleanup();                                              // services.hip

extern "C" {

void rodent_hip_set_device(int32_t dev) { rdev(dev); g_current_dev = dev; }

void rodent_hip_scene_destroy(int32_t dev) {
    RenderDevice& r = rdev(dev);
    HIP_CHECK(hipSetDevice(dev));
    HIP_CHECK(hipDeviceSynchronize());
    for (void* p : r.scene.allocs) HIP_CHECK(hipFree(p));
    r.scene = DevScene();
}

void rodent_hip_scene_create(int32_t dev, const RodentSceneDesc* d) {
    RenderDevice& r = rdev(dev);
    rodent_hip_scene_destroy(dev);
    HIP_CHECK(hipSetDevice(dev));
    DevScene& s = r.scene;
    s.dev.vertices = upload(s, d->vertices, 4 * (size_t)d->num_vertices);
    s.dev.normals = upload(s, d->normals, 4 * (size_t)d->num_vertices);
    s.dev.face_normals = upload(s, d->face_normals, 4 * (size_t)d->num_tris);
    s.dev.indices = upload(s, d->indices, 4 * (size_t)d->num_tris);
    {   // nodes and triangles in ONE allocation: k_trace_refill addresses both from one base with 32-bit offsets (joint_fetch_off)
        const size_t node_bytes = sizeof(Node2) * (size_t)d->num_nodes, tri_bytes = sizeof(Tri1) * (size_t)d->num_bvh_tris;
        char* bvh = nullptr;
        HIP_CHECK(hipMalloc(&bvh, std::max<size_t>(node_bytes + tri_bytes, 16)));
        s.allocs.push_back(bvh);
        if (node_bytes) HIP_CHECK(hipMemcpy(bvh, d->nodes, node_bytes, hipMemcpyHostToDevice));
        if (tri_bytes) HIP_CHECK(hipMemcpy(bvh + node_bytes, d->tris, tri_bytes, hipMemcpyHostToDevice));
        s.dev.nodes = reinterpret_cast<const Node2*>(bvh);
        s.dev.tris = reinterpret_cast<const Tri1*>(bvh + node_bytes);
        // tri_delta: bytes from the record a node id of 0 would have to triangle 0; 0 = not addressable this way (offsets beyond 32 bits,
        // or an index beyond the 24-bit multiply)
        const unsigned long long delta = sizeof(Node2) + node_bytes, end = delta + tri_bytes;
        s.tri_delta = (end < (1ull << 32) && d->num_nodes < (1 << 24) - 1 && d->num_bvh_tris < (1 << 24)) ? (unsigned)delta : 0u;
    }
    s.dev.materials = upload(s, d->materials, (size_t)d->num_materials);
    s.dev.lights = upload(s, d->lights, (size_t)d->num_lights);
    s.dev.light_ids = upload(s, d->light_ids, (size_t)d->num_tris);
    const bool textured = d->num_textures > 0;
    if (textured && (!d->texcoords || !d->textures || !d->texels)) {
        fprintf(stderr, "rodent_hip: scene with textures but without texture data\n"); abort(); }
    s.dev.texcoords = upload(s, d->texcoords, textured ? 4 * (size_t)d->num_vertices : 0);
    s.dev.textures = upload(s, d->textures, textured ? (size_t)d->num_textures : 0);
    s.dev.texels = upload(s, d->texels, textured ? (size_t)d->num_texels : 0);
    for (int32_t k = 0; k < d->num_materials; k++)
        if (d->materials[k].tex_kd < 0 || d->materials[k].tex_kd > d->num_textures || d->materials[k].tex_ks < 0
            || d->materials[k].tex_ks > d->num_textures) {
            fprintf(stderr, "rodent_hip: material %d refers to a texture that does not exist\n", k); abort();
        }
    // every index the kernels follow, checked once on the host tables (a corrupt scene must not become an out-of-bounds read in k_shade /
    // trace_one)
    auto invalid = [](const char* what) { fprintf(stderr, "rodent_hip: invalid scene: %s\n", what); abort(); };
    if (d->num_vertices <= 0 || d->num_tris <= 0 || d->num_nodes <= 0 || d->num_bvh_tris <= 0 || d->num_materials <= 0
        || d->num_lights < 0) invalid("empty table");
    for (int32_t t = 0; t < d->num_tris; t++) {
        for (int k = 0; k < 3; k++) if ((uint32_t)d->indices[4 * t + k] >= (uint32_t)d->num_vertices) invalid("vertex index out of range");
        if ((uint32_t)d->indices[4 * t + 3] >= (uint32_t)d->num_materials) invalid("material index out of range");
        if (d->light_ids[t] < 0 || (d->light_ids[t] > 0 && d->light_ids[t] >= d->num_lights)) invalid("light id out of range");
        // an emitter's triangles are looked up in the light table by the shader (on_hit: sc.lights[light_ids[prim]]): the entry must exist
        if (d->materials[d->indices[4 * t + 3]].emissive
            && d->light_ids[t] >= d->num_lights) invalid("emissive triangle without an entry in the light table");
    }
    for (int32_t k = 0; k < d->num_nodes; k++)
        for (int j = 0; j < 2; j++) {
            const int32_t c = d->nodes[k].child[j];
            if ((c > 0 && c > d->num_nodes) || (c < 0 && ~c >= d->num_bvh_tris)) invalid("BVH child id out of range");
        }
    if (d->tris[d->num_bvh_tris - 1].prim_id >= 0) invalid("the last BVH triangle lacks the end-of-leaf bit");
    for (int32_t k = 0; k < d->num_bvh_tris; k++)
        if ((d->tris[k].prim_id & 0x7FFFFFFF) >= d->num_tris
            || (uint32_t)d->tris[k].geom_id
            >= (uint32_t)d->num_materials) invalid("BVH triangle refers to a primitive or material that does not exist");
    for (int32_t k = 0; k < d->num_textures; k++)
        if (d->textures[k].width <= 0 || d->textures[k].height <= 0
            || (uint64_t)d->textures[k].offset + (uint64_t)d->textures[k].width * (uint64_t)d->textures[k].height
            > d->num_texels) invalid("texture outside the texel pool");
    s.dev.num_tris = d->num_tris; s.dev.num_materials = d->num_materials; s.dev.num_lights = d->num_lights;
    // top-of-tree images of the stream traversal kernels (record layout: traversal_device.h build_top_image): breadth first from
    // the root; a child that got a slot is a link (kLdsTag + byte offset of its record), the others keep their ids
    const auto build_image = [&](int capacity) {
        std::vector<int32_t> image((size_t)capacity * 16, 0), slots{1};
        for (size_t k = 0; k < slots.size(); k++) {
            const Node2& nd = d->nodes[slots[k] - 1];
            int32_t* rec = image.data() + 16 * k;
            memcpy(rec, nd.bounds, 12 * sizeof(float));
            for (int j = 0; j < 2; j++) {
                const int32_t c = nd.child[j];
                rec[12 + j] = c;
                if (c > 0 && (int)slots.size() < capacity) { rec[12 + j] = kLdsTag + (int32_t)slots.size() * (int32_t)sizeof(Node2);
                    slots.push_back(c); }
            }
            rec[14] = slots[k];
        }
        return reinterpret_cast<const int4*>(upload(s, image.data(), image.size()));
    };
    s.dev.top_image = build_image(kSceneTopNodes);
    s.dev.top_image_large = build_image(kPersistTopNodes);
    // SceneDev::tri_shade: per triangle face normal + its three vertex normals, gathered (RODENT_HIP_TRI_SHADE=0: not built, the shader
    // goes through indices -> normals)
    {
        static const bool on = [] { const char* e = getenv("RODENT_HIP_TRI_SHADE"); return !e || atoi(e) != 0; }();
        s.dev.tri_shade = nullptr;
        if (on && d->num_tris > 0) {
            std::vector<float> rec(12 * (size_t)d->num_tris);
            for (int32_t t = 0; t < d->num_tris; t++) {
                float* o = rec.data() + 12 * (size_t)t;
                const float* fn = d->face_normals + 4 * (size_t)t;
                o[0] = fn[0]; o[1] = fn[1]; o[2] = fn[2];
                for (int k = 0; k < 3; k++) {
                    const float* n = d->normals + 4 * (size_t)d->indices[4 * (size_t)t + k];
                    o[3 + 3 * k] = n[0]; o[4 + 3 * k] = n[1]; o[5 + 3 * k] = n[2];
                }
            }
            s.dev.tri_shade = reinterpret_cast<const float4*>(upload(s, rec.data(), rec.size()));
        }
        s.dev.tri_tex = nullptr;
        // SceneDev::tri_tex: the corners' texture coordinates, for resolve_material
        if (on && textured && d->num_tris > 0) {
            std::vector<float> tc(6 * (size_t)d->num_tris);
            for (int32_t t = 0; t < d->num_tris; t++)
                for (int k = 0; k < 3; k++) {
                    const float* c = d->texcoords + 4 * (size_t)d->indices[4 * (size_t)t + k];
                    tc[6 * (size_t)t + 2 * k] = c[0]; tc[6 * (size_t)t + 2 * k + 1] = c[1];
                }
            s.dev.tri_tex = upload(s, tc.data(), tc.size());
        }
    }
    s.num_nodes = d->num_nodes;
    s.loaded = true;
    r.mapping = resolve_mapping(r);
    r.trace_persistent = resolve_trace(r);
    resolve_refill(r);
}

void rodent_hip_render_config(int32_t dev, int32_t spp, int32_t max_path_len) {
    if (spp < 1 || max_path_len < 0) { fprintf(stderr, "rodent_hip: invalid render configuration\n"); abort(); }
    RenderDevice& r = rdev(dev); r.spp = spp; r.max_path_len = max_path_len;
}

void rodent_hip_render_sort(int32_t dev, int32_t enable) { rdev(dev).sort = enable ? 1 : 0; }
void rodent_hip_render_hit_records(int32_t dev, int32_t aos) { rdev(dev).hit_records_aos = aos ? 1 : 0; }
void rodent_hip_render_overlap(int32_t dev, int32_t enable) { rdev(dev).overlap = enable ? 1 : 0; }
void rodent_hip_render_fused_sort(int32_t dev, int32_t enable) { rdev(dev).fused_sort = enable ? 1 : 0; }
void rodent_hip_render_fused_compact(int32_t dev, int32_t enable) { rdev(dev).fused_compact = std::min(2, std::max(0, (int)enable)); }
void rodent_hip_render_lds_image(int32_t dev, int32_t enable) { rdev(dev).lds_image = enable ? 1 : 0; }
void rodent_hip_render_mega_joint(int32_t dev, int32_t enable) { rdev(dev).mega_joint = enable ? 1 : 0; }
void rodent_hip_render_trace_persistent(int32_t dev, int32_t enable) { RenderDevice& r = rdev(dev);
    r.trace_persistent_request = std::min(2, std::max(-1, (int)enable)); r.trace_persistent = resolve_trace(r); }

void rodent_hip_render_trace_refill(int32_t dev, int32_t idle_bounce, int32_t idle_shadow) {
    const bool per_scene = idle_bounce == -1 && idle_shadow == -1, off = idle_bounce == 0 && idle_shadow == 0;
    if (!per_scene && !off && (idle_bounce < 1 || idle_bounce > kWave || idle_shadow < 1 || idle_shadow > kWave)) {
        fprintf(stderr, "rodent_hip: lane refill thresholds must be -1, -1 (per scene), 0, 0 (off) or 1 .. 64 idle lanes each\n"); abort();
    }
    RenderDevice& r = rdev(dev);
    r.trace_refill_request[0] = idle_bounce; r.trace_refill_request[1] = idle_shadow;
    resolve_refill(r);
}

int32_t rodent_hip_render_trace_refill_in_effect(int32_t dev) { const RenderDevice& r = rdev(dev);
    return r.trace_refill | (r.trace_refill_shadow << 8); }

void rodent_hip_render_capacity(int32_t dev, int32_t rays) {
    if (rays != 0 && (rays < 64 || rays > kMaxCapacity)) {
        fprintf(stderr, "rodent_hip: stream capacity must be 0 (default) or 64 .. %ld rays\n", kMaxCapacity); abort(); }
    rdev(dev).capacity = rays;
}

void rodent_hip_render_mapping(int32_t dev, int32_t mapping) {
    if (mapping < -1 || mapping > 1) {
        fprintf(stderr, "rodent_hip: unknown mapping %d (-1 = per scene, 0 = streaming, 1 = megakernel)\n", mapping); abort(); }
    RenderDevice& r = rdev(dev);
    r.mapping_request = mapping;
    r.mapping = resolve_mapping(r);
}
int32_t rodent_hip_render_mapping_in_effect(int32_t dev) { return rdev(dev).mapping; }
void rodent_hip_render_defaults(int32_t dev) { RenderDevice& r = rdev(dev); render_defaults(r); r.mapping = resolve_mapping(r);
    r.trace_persistent = resolve_trace(r); resolve_refill(r); }

int32_t get_spp(void) { return rdev(g_current_dev).spp; }

void setup_interface(size_t width, size_t height) { g_host_w = width; g_host_h = height; g_host_film.assign(width * height * 3, 0.0f); }
float* get_pixels(void) { return g_host_film.data(); }
void cleanup_interface(void) {
    // buffers / BVHs / images loaded through rodent_load_* (services.hip)
    rodent_services_cleanup();
    for (auto& r : g_rdev) if (r.init && r.film) { (void)hipSetDevice(r.dev); (void)hipFree(r.film); r.film = nullptr;
        r.film_w = r.film_h = 0; }
    g_host_film.clear(); g_host_w = g_host_h = 0;
}
void clear_pixels(void) {                                                    // interface.cpp:498-505
    std::fill(g_host_film.begin(), g_host_film.end(), 0.0f);
    for (auto& r : g_rdev) if (r.init && r.film) { HIP_CHECK(hipSetDevice(r.dev));
        HIP_CHECK(hipMemset(r.film, 0, sizeof(float) * 3 * (size_t)r.film_w * r.film_h)); }
}

void rodent_get_film_data(int32_t dev, float** pixels, int32_t* width, int32_t* height) {
    RenderDevice& r = rdev(dev); ensure_film(r);
    *pixels = r.film; *width = r.film_w; *height = r.film_h;
}
void rodent_gpu_get_first_primary_stream(int32_t dev, PrimaryStream* p, int32_t size) { RenderDevice& r = rdev(dev);
    carve_primary(*p, ensure_slab(r, 0, size, 20), round_cap(size)); }
void rodent_gpu_get_second_primary_stream(int32_t dev, PrimaryStream* p, int32_t size) { RenderDevice& r = rdev(dev);
    carve_primary(*p, ensure_slab(r, 1, size, 20), round_cap(size)); }
void rodent_gpu_get_secondary_stream(int32_t dev, SecondaryStream* s, int32_t size) { RenderDevice& r = rdev(dev);
    carve_secondary(*s, ensure_slab(r, 2, size, 13), round_cap(size)); }
void rodent_gpu_get_tmp_buffer(int32_t dev, int32_t** buf, int32_t size) {
    RenderDevice& r = rdev(dev);
    if (r.tmp_cap < round_cap(size)) { HIP_CHECK(hipSetDevice(dev)); if (r.tmp) HIP_CHECK(hipFree(r.tmp));
        HIP_CHECK(hipMalloc(&r.tmp, sizeof(int) * round_cap(size))); r.tmp_cap = round_cap(size); }
    *buf = r.tmp;
}
// Host stream slabs with the same carving (interface.cpp:367-373,621-629).  The library computes nothing on the CPU;
// a host that stages rays itself (or the reference's CPU mapping) gets the layout it expects.
void rodent_cpu_get_primary_stream(PrimaryStream* p, int32_t size) {
    const size_t cap = (size_t)round_cap(size);
    if (t_cpu_primary.size() < cap * 20) t_cpu_primary.assign(cap * 20, 0.0f);
    carve_primary(*p, t_cpu_primary.data(), t_cpu_primary.size() / 20);
}
void rodent_cpu_get_secondary_stream(SecondaryStream* s, int32_t size) {
    const size_t cap = (size_t)round_cap(size);
    if (t_cpu_secondary.size() < cap * 13) t_cpu_secondary.assign(cap * 13, 0.0f);
    carve_secondary(*s, t_cpu_secondary.data(), t_cpu_secondary.size() / 13);
}
void rodent_present(int32_t dev) {                                           // interface.cpp:494-496,660-663
    RenderDevice& r = rdev(dev);
    if (!r.film) return;
    HIP_CHECK(hipSetDevice(dev));
    HIP_CHECK(hipMemcpy(g_host_film.data(), r.film, sizeof(float) * 3 * (size_t)r.film_w * r.film_h, hipMemcpyDeviceToHost));
}

void rodent_hip_render_rows(int32_t dev, const Settings* settings, int32_t iter, int32_t y0, int32_t y1, void* stream) {
    RenderDevice& r = rdev(dev);
    ensure_film(r);
    if (y0 < 0 || y1 > r.film_h || y0 > y1) { fprintf(stderr, "rodent_hip: invalid row range [%d, %d)\n", y0, y1); abort(); }
    render_rows_any(r, settings, iter, y0, y1, (hipStream_t)stream);
}

// Interleaved row tiles (the reference hands out ~1024-sample tiles dynamically, render/mapping_gpu.impala:374-420,
// mapping_cpu.impala:200-237): this call renders the row tiles first_tile, first_tile + tile_stride, ... of tile_rows rows each -- GPU k of
// K takes first_tile = k, tile_stride = K. Seeds depend on absolute (sample, iter, x, y) only: the tiles of all GPUs together are the frame
// render() produces.
void rodent_hip_render_tiles(int32_t dev, const Settings* settings, int32_t iter, int32_t tile_rows, int32_t first_tile,
    int32_t tile_stride, void* stream) {
    RenderDevice& r = rdev(dev);
    ensure_film(r);
    if (tile_rows <= 0 || first_tile < 0 || tile_stride <= 0) {
        fprintf(stderr, "rodent_hip: invalid tile arguments (%d rows, first %d, stride %d)\n", tile_rows, first_tile, tile_stride);
        abort(); }
    const int full_tiles = r.film_h / tile_rows, ragged_rows = r.film_h % tile_rows;
    // complete tiles of this call
    const int mine = first_tile < full_tiles ? (full_tiles - first_tile + tile_stride - 1) / tile_stride : 0;
    const bool ragged_mine = ragged_rows > 0 && full_tiles >= first_tile && (full_tiles - first_tile) % tile_stride == 0;
    // the counters (rodent_hip_render_counters) are those of the whole call: its sub-calls after the first add to them
    r.counters_continue = false;
    // the megakernel tiles the rows it is given itself: one launch per row tile
    if (r.mapping == 1) {
        for (int k = 0; k < mine; k++) {
            render_rows_mega(r, settings, iter, (first_tile + k * tile_stride) * tile_rows, (first_tile + k * tile_stride + 1) * tile_rows,
            (hipStream_t)stream); r.counters_continue = true; }
    } else if (mine > 0) {
        const int y0 = first_tile * tile_rows;
        render_rows(r, settings, iter, y0, y0 + mine * tile_rows, (hipStream_t)stream, tile_rows, tile_stride * tile_rows);
        r.counters_continue = true;
    }
    if (ragged_mine) render_rows_any(r, settings, iter, full_tiles * tile_rows, r.film_h, (hipStream_t)stream);
    r.counters_continue = false;
}

void render(const Settings* settings, int32_t iter) {                        // generated render(): converter.cpp:628-967
    RenderDevice& r = rdev(g_current_dev);
    ensure_film(r);
    render_rows_any(r, settings, iter, 0, r.film_h, nullptr);
    rodent_present(g_current_dev);                                           // device.present() (converter.cpp:965)
}

void rodent_hip_render_counters(int32_t dev, uint64_t* out4) {
    RenderDevice& r = rdev(dev);
    HIP_CHECK(hipSetDevice(dev));
    uint64_t all[kNumCounters];
    HIP_CHECK(hipMemcpy(all, r.counters, sizeof(all), hipMemcpyDeviceToHost));
    for (int k = 0; k < 4; k++) out4[k] = all[k];
    for (int k = 4; k < 68; k++) out4[1] += all[k];                         // shadow rays are counted in 64 stripes
    for (int k = 68; k < kNumCounters; k++) out4[0] += all[k];              // the megakernel's primary rays, 32 stripes
}

// ---- stage-level entry points ---------------------------------------------------------------
void hip_generate_rays(int32_t dev, PrimaryStream* primary, int32_t capacity, int32_t first_ray_id, int32_t num_rays,
    const Settings* settings,
                       int32_t iter, int32_t film_width, int32_t film_height, int32_t first_pixel, int32_t spp, void* stream) {
    rdev(dev); HIP_CHECK(hipSetDevice(dev));
    if (primary->size + num_rays > capacity) { fprintf(stderr, "rodent_hip: hip_generate_rays exceeds the stream capacity\n"); abort(); }
    if (num_rays > 0)
        hipLaunchKernelGGL(k_generate, dim3((num_rays + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream, *primary,
            primary->size, first_ray_id,
                           num_rays, to_cam(settings), iter, film_width, film_height, first_pixel, spp, 0, 0);
    primary->size += num_rays;
    HIP_CHECK(hipGetLastError());
}

void hip_traverse_primary(int32_t dev, PrimaryStream* primary, void* stream) {
    RenderDevice& r = rdev(dev); HIP_CHECK(hipSetDevice(dev)); ensure_film(r); require_scene(r);
    if (primary->size <= 0) return;
    // a caller's stream holds its hit records in the ABI's five arrays (store_hit_record)
    primary->pad = 0;
    launch_trace_primary(r, (hipStream_t)stream, *primary, primary->size);
    HIP_CHECK(hipGetLastError());
}

void hip_sort_primary(int32_t dev, PrimaryStream* primary, PrimaryStream* other, int32_t* ray_ends, void* stream) {
    RenderDevice& r = rdev(dev); HIP_CHECK(hipSetDevice(dev)); ensure_film(r); require_scene(r);
    const int G = r.scene.dev.num_materials;
    if (G + 1 > kMaxBins) { fprintf(stderr, "rodent_hip: too many geometries (%d)\n", G); abort(); }
    bin_stream(r, 0, *primary, *other, nullptr, primary->size, KEY_GEOM, G + 1, 1, G + 1, (hipStream_t)stream);
    HIP_CHECK(hipMemcpyAsync(r.host_pinned + 8, bin_end(r, 0), sizeof(int) * (G + 1), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    std::memcpy(ray_ends, r.host_pinned + 8, sizeof(int) * (G + 1));
    other->size = primary->size;
}

void hip_shade(int32_t dev, PrimaryStream* primary, SecondaryStream* secondary, int32_t num_rays, void* stream) {
    RenderDevice& r = rdev(dev); HIP_CHECK(hipSetDevice(dev)); ensure_film(r); require_scene(r);
    primary->size = num_rays; secondary->size = num_rays; primary->pad = 0;
    if (num_rays <= 0) return;
    launch_k_shade((hipStream_t)stream, num_rays, r.scene.dev, *primary, *primary, (const int*)nullptr, *secondary, (const int*)nullptr,
        num_rays, r.film,
                   1.0f / (float)r.spp, r.max_path_len,
                       /* a ray that missed ends here instead of indexing the material table with the miss id: */ 1, (unsigned*)nullptr,
                       (int*)nullptr);
    HIP_CHECK(hipGetLastError());
}

void hip_traverse_secondary(int32_t dev, SecondaryStream* secondary, void* stream) {
    RenderDevice& r = rdev(dev); HIP_CHECK(hipSetDevice(dev)); ensure_film(r); require_scene(r);
    if (secondary->size <= 0) return;
    launch_trace_secondary(r, (hipStream_t)stream, *secondary, nullptr, secondary->size, 1.0f / (float)r.spp);
    HIP_CHECK(hipGetLastError());
}

int32_t hip_compact_primary(int32_t dev, PrimaryStream* primary, PrimaryStream* other, void* stream) {
    RenderDevice& r = rdev(dev); HIP_CHECK(hipSetDevice(dev));
    bin_stream(r, 1, *primary, *other, nullptr, primary->size, KEY_ALIVE, 2, 0, 1, (hipStream_t)stream);
    HIP_CHECK(hipMemcpyAsync(r.host_pinned + 8, bin_end(r, 1), sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    other->size = r.host_pinned[8];
    return other->size;
}

} // extern "C"

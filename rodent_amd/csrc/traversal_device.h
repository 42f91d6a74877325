// traversal_device.h -- ray / box / triangle arithmetic shared by the traversal kernels
// (traversal.hip) and the renderer's stream kernels (render.hip).  Same operation sequence as the
// CPU parity oracle: explicit fmaf(), compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "rodent_traversal.h"

namespace rodent_dev {

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
// The six fused multiply-adds are written on 2-vectors (lo, hi) so that they issue as three
// v_pk_fma_f32 (two IEEE fmas per lane per instruction on gfx950): same operations, same results.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bool slab(const RayX& r, float lox, float hix, float loy, float hiy, float loz, float hiz, float& tentry) {
    const f32x2 tx = __builtin_elementwise_fma((f32x2){r.idx, r.idx}, (f32x2){lox, hix}, (f32x2){r.iox, r.iox});
    const f32x2 ty = __builtin_elementwise_fma((f32x2){r.idy, r.idy}, (f32x2){loy, hiy}, (f32x2){r.ioy, r.ioy});
    const f32x2 tz = __builtin_elementwise_fma((f32x2){r.idz, r.idz}, (f32x2){loz, hiz}, (f32x2){r.ioz, r.ioz});
    tentry = fmaxf(fmaxf(fminf(tx.x, tx.y), fminf(ty.x, ty.y)), fmaxf(fminf(tz.x, tz.y), r.tmin));
    const float texit = fminf(fminf(fmaxf(tx.x, tx.y), fmaxf(ty.x, ty.y)), fminf(fmaxf(tz.x, tz.y), r.tmax));
    return tentry <= texit;
}

// slab() for loops that keep r.tmin / r.tmax CANONICAL (quieted once with canonical() when the ray is loaded; later
// tmax values are results of a division).  fmaxf / fminf must quiet a signalling NaN operand first, and the compiler
// cannot see across loop iterations that these two already are: it re-canonicalises both in every iteration (two
// v_max_f32 x, x).  The two operations that touch them are therefore issued as plain v_max_f32 / v_min_f32, which on
// canonical operands compute exactly fmaxf / fminf.
__device__ __forceinline__ float canonical(float x) { return __builtin_canonicalizef(x); }
__device__ __forceinline__ bool slab_canonical(const RayX& r, float lox, float hix, float loy, float hiy, float loz, float hiz,
    float& tentry) {
    const f32x2 tx = __builtin_elementwise_fma((f32x2){r.idx, r.idx}, (f32x2){lox, hix}, (f32x2){r.iox, r.iox});
    const f32x2 ty = __builtin_elementwise_fma((f32x2){r.idy, r.idy}, (f32x2){loy, hiy}, (f32x2){r.ioy, r.ioy});
    const f32x2 tz = __builtin_elementwise_fma((f32x2){r.idz, r.idz}, (f32x2){loz, hiz}, (f32x2){r.ioz, r.ioz});
    float nz, fz;
    asm("v_max_f32 %0, %1, %2" : "=v"(nz) : "v"(fminf(tz.x, tz.y)), "v"(r.tmin));
    asm("v_min_f32 %0, %1, %2" : "=v"(fz) : "v"(fmaxf(tz.x, tz.y)), "v"(r.tmax));
    tentry = fmaxf(fmaxf(fminf(tx.x, tx.y), fminf(ty.x, ty.y)), nz);
    const float texit = fminf(fminf(fmaxf(tx.x, tx.y), fmaxf(ty.x, ty.y)), fz);
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


struct HitAcc { int id; float t, u, v; };

typedef __attribute__((address_space(3))) int lds_int;

// ---------------------------------------------------------------------------------------------
// Top-of-tree image (traversal.hip: k_bvh2_top_persist; render.hip: the stream traversal kernels): the first `capacity`
// inner nodes in breadth-first order as 64-byte records that a workgroup stages in LDS.  A node id >= kLdsTag is a LINK:
// the byte offset of a record inside the image.
// ---------------------------------------------------------------------------------------------
constexpr int kLdsTag = 0x40000000;

// ---------------------------------------------------------------------------------------------
// The part of a lane's traversal stack that does not fit its LDS window (the reference's stack costs the same at depth 5 and
// at depth 50, stack.impala:52-123; until round 4 a ray that outgrew the window was abandoned and traced again from the root
// by a follow-up pass).  The window is WINDOW + 1 rows of [entry][lane] words behind a cursor: row 0 is the word that ends the
// traversal when it is popped, rows 1..WINDOW - 1 hold entries, and the step that fills row WINDOW moves the OLDEST kSpillRows
// entries (rows 1..kSpillRows) to the wave's block of global memory, shifts the rest down and goes on -- the ray stays in its
// lane, nothing is traced twice.  Row 0 then holds kSpillMark + (blocks spilled) instead of 0; popping THAT is the signal to
// bring the newest block back (one compare per step in the loop; everything else is off the hot path).  Capacity: WINDOW - 1
// entries in LDS + kSpillBlocks x kSpillRows behind them = 14 + 49 = 63 with the 15-row windows of the persistent kernels:
// the reference's 64 slots minus its sentinel (stack.impala:53-54,62-66).  One wave's block: [kSpillBlocks x kSpillRows][lane]
// ints, indexed by the wave's slot in the resident grid -- nobody else touches it during the launch, nothing carries over.
// ---------------------------------------------------------------------------------------------
constexpr int kSpillMark = 0x7F000000;          // above every node id (<= 0x3FFFFFFF) and every image link (kLdsTag + offset)
constexpr int kSpillRows = 7, kSpillBlocks = 7;
constexpr int kSpillWaveInts = kSpillRows * kSpillBlocks * kWave;       // 12 544 bytes per resident wave

// Both helpers run exec-masked on the rare path and take what they need in its cheapest loop-invariant form -- the lane's window LIMIT (col
// + WINDOW rows, which the step compares against anyway), the launch's spill buffer and the workgroup's wave count -- and derive the rest
// behind an opaque barrier: computed outside, the column base and the 64-bit address of the lane's spill words would be three more VGPRs
// carried around a loop that is compiled under a 64-VGPR budget (measured: +2 ... 3 % on the benchmark launches).
__device__ __forceinline__ int* spill_words(int* __restrict__ spill, int waves_per_group) {
    unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    // the wave's slot in the resident grid
    int slot = (int)blockIdx.x * waves_per_group + __builtin_amdgcn_readfirstlane((int)threadIdx.x / kWave);
    asm volatile("" : "+v"(lane), "+s"(slot));
    return spill + (size_t)slot * kSpillWaveInts + lane;
}
template <int WINDOW>
__device__ __forceinline__ lds_int* window_base(lds_int* limit) {
    unsigned a = (unsigned)(size_t)limit;
    asm volatile("" : "+v"(a));
    return (lds_int*)(size_t)(a - (unsigned)(WINDOW * kWave * sizeof(int)));
}
// Called by a lane whose push filled row WINDOW (sp == limit).  `events` (may be null): a counter of blocks moved out, for the tests and
// the per-scene reports (one atomic per block).
template <int WINDOW>
__device__ __forceinline__ void stack_spill(lds_int*& sp, int& top, lds_int* limit, int* __restrict__ spill, int waves_per_group, int* err,
    unsigned long long* events = nullptr) {
    static_assert(WINDOW > kSpillRows + 1, "something must stay in the window");
    lds_int* const col = window_base<WINDOW>(limit);
    const int mark = col[0], blocks = mark ? mark - kSpillMark : 0;
    // more than the reference's 64 slots: the host reports it
    if (blocks >= kSpillBlocks) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); top = 0; return; }
    if (events) atomicAdd(events, 1ull);
    int* g = spill_words(spill, waves_per_group) + blocks * kSpillRows * kWave;
#pragma unroll 1
    for (int j = 0; j < kSpillRows; j++) g[j * kWave] = col[(1 + j) * kWave];
#pragma unroll 1
    for (int j = 1; j <= WINDOW - kSpillRows; j++) col[j * kWave] = col[(j + kSpillRows) * kWave];
    col[0] = kSpillMark + blocks + 1;
    sp -= kSpillRows * kWave;
}
// Called at the end of a step by a lane that popped row 0 while blocks are out (its new top is the mark, >= kSpillMark): the newest block
// comes back into rows 1..kSpillRows - 1, its newest entry is the new top.  (Measured beside the alternative -- testing the word UNDER the
// cursor at the start of the step, off the dependency chain: 0.1857 against 0.1872 ms on the benchmark launch,
// profiles/r05_spill_experiment.txt.)
template <int WINDOW>
__device__ __forceinline__ void stack_reload(lds_int*& sp, int& top, lds_int* limit, int* __restrict__ spill, int waves_per_group) {
    lds_int* const col = window_base<WINDOW>(limit);
    const int blocks = top - kSpillMark;                                // >= 1
    const int* g = spill_words(spill, waves_per_group) + (blocks - 1) * kSpillRows * kWave;
#pragma unroll 1
    for (int j = 0; j < kSpillRows - 1; j++) col[(1 + j) * kWave] = g[j * kWave];
    top = g[(kSpillRows - 1) * kWave];
    col[0] = blocks > 1 ? kSpillMark + blocks - 1 : 0;
    sp = col + (kSpillRows - 1) * kWave;
}

// ---------------------------------------------------------------------------------------------
// One step's fetches, both kinds in flight TOGETHER.  A lane whose node is in the LDS image (in_lds: ds_read_b128 x 3 + ds_read_b64 at
// `lds_addr`) and a lane whose node / triangle comes from memory (global_load_dwordx4 x 3 at `addr` + dwordx2 at `addr_ids`) write the same
// registers. Compiled from the two branches of an if, the second kind's loads wait for the first kind's to LAND -- the compiler cannot know
// that the two exec masks are disjoint and sees a write-after-write on q0 .. ids -- so every iteration in which a wave holds both kinds
// pays an LDS round trip behind a memory round trip.  Here they are issued back to back under their own exec masks and waited for once,
// together with the word under the stack cursor (`popped`).  A kind nobody in the wave needs is skipped (a memory instruction with an empty
// exec mask still makes the round trip).  Measured on one MI355X (profiles/r05_joint_loads.txt): random segments -6 % (atrium) ... -13 %
// (crown, plant), camera rays +1 % (atrium) ... -4 % (crown).  Results cannot change: the same loads, the same lanes.
// ---------------------------------------------------------------------------------------------
typedef float vf4 __attribute__((ext_vector_type(4)));
typedef int vi2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) char* gbytes;
__device__ __forceinline__ void joint_fetch(vf4& q0, vf4& q1, vf4& q2, vi2& ids, int& popped, bool in_lds, unsigned lds_addr, gbytes addr,
    gbytes addr_ids, lds_int* sp) {
    const unsigned long long lds_mask = __ballot(in_lds);
    const unsigned sp_addr = (unsigned)(size_t)sp;
    unsigned long long save;
    asm volatile("s_mov_b64 %[save], exec\n\t"
                 "s_andn2_b64 exec, exec, %[lm]\n\t"
                 "s_cbranch_execz .Ljoint_a_%=\n\t"
                 "global_load_dwordx4 %[q1], %[a], off offset:16\n\t"
                 "global_load_dwordx4 %[q0], %[a], off\n\t"
                 "global_load_dwordx4 %[q2], %[a], off offset:32\n\t"
                 "global_load_dwordx2 %[ch], %[ac], off\n"
                 ".Ljoint_a_%=:\n\t"
                 "s_and_b64 exec, %[save], %[lm]\n\t"
                 "s_cbranch_execz .Ljoint_b_%=\n\t"
                 "ds_read_b128 %[q0], %[l]\n\t"
                 "ds_read_b128 %[q1], %[l] offset:16\n\t"
                 "ds_read_b128 %[q2], %[l] offset:32\n\t"
                 "ds_read_b64 %[ch], %[l] offset:48\n"
                 ".Ljoint_b_%=:\n\t"
                 "s_mov_b64 exec, %[save]\n\t"
                 "ds_read_b32 %[pop], %[sp]\n\t"
                 "s_waitcnt vmcnt(0) lgkmcnt(0)"
                 : [q0] "=&v"(q0), [q1] "=&v"(q1), [q2] "=&v"(q2), [ch] "=&v"(ids), [pop] "=&v"(popped), [save] "=&s"(save)
                 : [a] "v"(addr), [ac] "v"(addr_ids), [l] "v"(lds_addr), [sp] "v"(sp_addr), [lm] "s"(lds_mask)
                 : "memory", "scc");               // (s_andn2_b64 / s_and_b64 write SCC)
}

__device__ __forceinline__ void wave_lds_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }
// One wave (the LDS operations of a wave complete in program order: between one lane's write and another lane's read the compiler only has
// to keep that order).  Record layout: ints 0..11 bounds, 12..13 child ids / links, 14 = the node's own 1-based id (0: slot unused), 15 =
// 0.
__device__ __forceinline__ void build_top_image(const Node2* __restrict__ nodes, int4* __restrict__ image, int capacity,
    lds_int* slot_node /* [capacity]: 1-based node id held by each slot */) {
    const int lane = threadIdx.x;
    if (lane == 0) slot_node[0] = 1;
    wave_lds_sync();
    int begin = 0, end = 1;                                           // slots of the current level
    while (begin < end) {
        int next = end;
        for (int first = begin; first < end; first += kWave) {
            const int slot = first + lane;
            const bool on = slot < end;
            int4 r0 = {}, r1 = {}, r2 = {}, r3 = {};
            if (on) { const int4* p = reinterpret_cast<const int4*>(nodes + (slot_node[slot] - 1)); r0 = p[0]; r1 = p[1]; r2 = p[2];
                r3 = p[3]; }
            const bool in0 = on && r3.x > 0, in1 = on && r3.y > 0;   // inner children (r3.x / r3.y = Node2::child)
            const unsigned long long m0 = __ballot(in0), m1 = __ballot(in1), below = (1ull << lane) - 1ull;
            const int s0 = next + __popcll(m0 & below), s1 = next + __popcll(m0) + __popcll(m1 & below);
            if (in0 && s0 < capacity) { slot_node[s0] = r3.x; r3.x = kLdsTag + s0 * (int)sizeof(Node2); }
            if (in1 && s1 < capacity) { slot_node[s1] = r3.y; r3.y = kLdsTag + s1 * (int)sizeof(Node2); }
            next = min(capacity, next + __popcll(m0) + __popcll(m1));
            if (on) { r3.z = slot_node[slot]; r3.w = 0; int4* q = image + 4 * slot; q[0] = r0; q[1] = r1; q[2] = r2; q[3] = r3; }
        }
        wave_lds_sync();
        begin = end; end = next;
    }
    for (int slot = end + lane; slot < capacity; slot += kWave) image[4 * slot + 3] = int4{0, 0, 0, 0};       // unused slots
}
} // namespace rodent_dev

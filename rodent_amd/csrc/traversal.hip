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
//  * one ray per lane, one 64-ray chunk per wave at a time, chunks handed to waves XCD-aware;
//  * single-step schedule (bvh2_step): every lane advances by one node step or one triangle test per wave
//    iteration, the loads of both kinds in flight together; the hit record lives in memory, not in registers;
//  * the default mapping (k_bvh2_top_persist): the top 255 nodes of the caller's hierarchy are staged in LDS as an image
//    that every workgroup validates against the node array before using it (the ABI passes a pointer, not a handle); the
//    grid is one resident generation of 16-wave workgroups whose waves draw chunks from 64 striped ticket counters;
//    launches under 384 Ki rays take k_bvh2_single (64-lane workgroups = one wavefront, one chunk each);
//  * the traversal stack is an LDS-only window of 15 / 16 entries, laid out [entry][lane] (bank = lane % 32 for
//    ds_read/write_b32: conflict free whatever each lane's depth is) and walked with a cursor pointer; a ray that
//    needs more is finished by the one-wave follow-up kernel (k_bvh2_finish / k_bvh2_top_finish) with the reference's
//    64-entry stack (stack.impala:53) in global memory; beyond 64 a device-side error flag makes the host abort();
//  * nodes and triangles are fetched with 16-byte loads (global_load_dwordx4 / ds_read_b128);
//  * no MFMA: this is branchy scalar fp32 work;
//  * arithmetic is written out with explicit fmaf() and compiled with
//    -ffp-contract=off so results are bit-identical to the CPU parity oracle.
//
// This file: shared device helpers, the BVH2 kernels (k_bvh2_top_persist, k_bvh2_single, k_bvh2_phase, the ray sort) and their follow-up
// kernels, the host side and the C ABI.  traversal_top.h holds the default mapping's kernels, traversal_wide.h the BVH4 / BVH8 + Tri4
// kernels (same schedule).  Everything that was measured along the way and lost lives under lab/ (kernels, launchers, ~120 rows of the
// variant table) and is compiled only into the lab build (-DRODENT_HIP_LAB, librodent_hip_lab.so): the product library ships the default
// mappings only. Kernel variants ("mappings") are selected at run time; see kVariants below.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <utility>
#include <vector>

#include "rodent_traversal.h"
#include "traversal_device.h"

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

using namespace rodent_dev;

// One leaf of Tri1 records (mapping_gpu.impala:156-174).  Returns true when an
// any-hit query is finished.
// WIDE = false: 32-bit byte offsets from a uniform base (global_load ... v_off, s[base:base+1]: one VALU
// instruction of address arithmetic per step instead of two 64-bit ones); the host picks WIDE = true when an
// array is 4 GiB or larger.
template <bool ANY, bool WIDE = true>
__device__ __forceinline__ bool leaf_tri1(const Tri1* __restrict__ tris, int first, RayX& ray, HitAcc& hit) {
    int j = first;
    for (;;) {
        const float4* p = WIDE ? reinterpret_cast<const float4*>(tris + j)
                               : reinterpret_cast<const float4*>(reinterpret_cast<const char*>(tris) + (unsigned)j
                                   * (unsigned)sizeof(Tri1));
        j++;
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

constexpr int kMaxTopNodes = 1024;       // capacity of a launch context's top-of-tree image buffer

// Launch control block in device memory (zero between launches).
// err: raised (agent-scope atomic store) by a stack that outgrows the reference's 64 slots.  host_err: a word of pinned host memory
// (DeviceState::host_page) into which whoever finishes the launch (finish_launch, k_wide_finish) copies a raised flag -- the host reads it
// after its synchronisation without a device-to-host copy (12 us of every synchronous call until round 4,
// profiles/r05_host_call_costs.txt).  (The error paths do not store into the host word themselves: with the pointer loaded inside the
// traversal loop the benchmark launch was 3 % slower.)
struct Ctl { int counter; int reserved; int err; int deep_count; unsigned long long stats[8]; unsigned long long* trace; int finish_done;
    int* host_err; };


// Per-lane stack of the fast / sched kernels: an LDS-only window of LDS_N entries behind an
// address_space(3) pointer (plain ds_read / ds_write; LDS_N + 1 rows so the slot above the top
// can always be written).  A ray that needs more is appended to the launch's "deep list" and
// finished by k_bvh2_finish, a one-wave kernel enqueued right behind on the same stream that uses
// a 64-entry stack in global memory (the reference's capacity, stack.impala:53).  Measured
// alternatives: spilling to scratch inside the kernel costs ~6 % (scratch allocation per wave),
// detecting the last wave with one atomic per wave ~15 % (a single counter saturates near
// 88 atomics/us); the extra launch costs ~4 us per pass.
// The follow-up kernels' stack: the reference's 64 entries (stack.impala:53), [entry][lane] in 16 KB of LDS.  (Rounds 1-2 kept
// it in global memory: every push and pop a round trip, ~100 us for the first deep ray of a launch.)
struct DeepStack {
    lds_int* base; int* err;
    __device__ __forceinline__ void put(int e, int v) { if (e < kStackCap) base[e * kWave] = v;
        else __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ __forceinline__ int  get(int e) const { return base[(e < kStackCap ? e : kStackCap - 1) * kWave]; }
};

// Run by the one thread that finishes a launch: a raised overflow flag goes to the host's pinned word (Ctl::host_err) and is cleared.
__device__ __forceinline__ void report_error(Ctl* ctl) {
    if (__hip_atomic_load(&ctl->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        __hip_atomic_store(ctl->host_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&ctl->err, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The deep rays of a launch, restarted from the root with the 64-entry stack; `stack_lds`: kStackCap x kWave ints of LDS.  Run by
// ONE wave per workgroup (threads 0..63); with a grid of several workgroups each takes every gridDim.x-th batch of 64 rays and
// the last one to finish resets the launch's control words (with a grid of one this is the old one-wave kernel).
// `group` of `groups`: the calling workgroup's share (the follow-up kernels pass blockIdx.x / gridDim.x; the persistent kernel's
// last workgroup, which does this work inside the launch, passes 0 / 1).
template <bool ANY>
__device__ __forceinline__ void finish_launch(const Node2* __restrict__ nodes, const Tri1* __restrict__ tris,
                                              const Ray1* __restrict__ rays, Hit1* __restrict__ hits,
                                              Ctl* ctl, const int* __restrict__ deep_list, lds_int* stack_lds, int* phase_counters,
                                                  int group, int groups, int known_count = -1) {
    // behind the last phase of a phased launch: the stripe counters are zero again for the next launch on this stream
    if (phase_counters && group == 0) for (int k = threadIdx.x; k < 4 * 64; k += kWave) phase_counters[k * 16] = 0;
    const int count = known_count >= 0 ? known_count : ctl->deep_count;
    if (count > 0) {
        DeepStack st{stack_lds + threadIdx.x, &ctl->err};
        for (int k = group * kWave + threadIdx.x; k < count; k += groups * kWave) {
            const int i = deep_list[k];
            RayX ray = load_ray(rays, i);
            HitAcc hit{-1, ray.tmax, 0.0f, 0.0f};
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
    }
    if (threadIdx.x == 0) {
        // (every workgroup has read deep_count before it counts itself done, so the last one may zero it)
        // (with no deep rays -- the usual case -- nobody needs to wait for anybody: workgroup 0 rewrites the zeros)
        const bool last = groups == 1 || (count == 0 ? group == 0 : atomicAdd(&ctl->finish_done, 1) == groups - 1);
        if (last) {
            // stats[7]: rays handed over (read by the tests); ready for the next launch
            ctl->stats[7] += (unsigned long long)count; ctl->counter = 0; ctl->deep_count = 0; ctl->finish_done = 0;
            report_error(ctl);
        }
    }
}

template <bool ANY>
__global__ __launch_bounds__(kWave) void k_bvh2_finish(const Node2* __restrict__ nodes, const Tri1* __restrict__ tris,
                                                        const Ray1* __restrict__ rays, Hit1* __restrict__ hits,
                                                        Ctl* ctl, const int* __restrict__ deep_list,
                                                            int* /* unused since the stack moved to LDS */, int* phase_counters) {
    __shared__ int stack_lds[kStackCap * kWave];
    finish_launch<ANY>(nodes, tris, rays, hits, ctl, deep_list, (lds_int*)stack_lds, phase_counters, blockIdx.x, gridDim.x);
}

// Single-step schedule (the default "fast" variant).  Every lane advances by ONE step per wave iteration,
// whatever its kind -- a node step or one triangle test -- and the loads of both kinds are issued together,
// so a wave needs max over its lanes of (node steps + triangle tests) iterations, one memory round trip each.
// A 1 Mi-ray launch is only two rounds of resident waves on this chip (2 rays per lane slot): there is no
// steady state to amortise a drain in, the kernel time is (start of the last expensive waves) + (their
// iteration count) x (round-trip latency), and the while-while loop further down -- which keeps more lanes
// busy per instruction -- loses because lanes in the other phase wait (scripts/trace.py: the waves that end
// the launch made 200+ descent iterations and 40+ leaf visits where no ray needs more than ~140 node steps;
// letting triangle lanes queue up until N of them are ready was measured too: every N > 1 is slower).
// The per-ray sequence of node and triangle tests is unchanged, so results stay bit-identical.
//
// Per-lane traversal state.  The hit record lives in memory, not in registers: the miss record is written up front and
// every accepted triangle overwrites it (2-3 sixteen-byte stores per ray, off the critical path; same-address stores
// of one lane stay in order).  That leaves top, sp and tmax as the only per-lane state carried around the loop -- with
// the record in registers the compiler shuffled it between two register sets every iteration (a fifth of the loop's
// VALU instructions were v_mov) -- and it is also all that has to move when a ray changes lanes (k_bvh2_phase).
struct Lane {
    RayX ray;                  // tmin / tmax canonical (see slab_canonical)
    int top;                   // 0 = done, > 0 inner node id, < 0 ~(triangle index)
    lds_int* sp;               // the stack entry under the top (entry `ptr` of the other kernels)
    int ray_id;
    bool found;                // LAZY kernels only: a triangle has been accepted (lives in an SGPR lane mask, not in a VGPR)
};
typedef const __attribute__((address_space(1))) char* gptr;
// Both array bases as 64-bit integers in VGPRs (the per-lane select in the step would otherwise copy them from SGPRs in
// every iteration), turned back into GLOBAL pointers: laundering the pointers themselves leaves generic ones and flat loads.
// (32-bit offsets from one SGPR base -- global_load ... v_off, s[base:base+1], a v_mad_u32_u24 instead of the 64-bit
// multiply-add -- save one VALU instruction per iteration but need both arrays within 4 GiB of each other, which separate
// hipMalloc allocations are not: measured on the test harness' arrays, the precondition never held.)
struct Bases { gptr node, tri; };
__device__ __forceinline__ Bases make_bases(const Node2* nodes, const Tri1* tris) {
    // node ids are 1-based
    unsigned long long node_bits = reinterpret_cast<unsigned long long>(nodes - 1), tri_bits = reinterpret_cast<unsigned long long>(tris);
    asm volatile("" : "+v"(node_bits), "+v"(tri_bits));
    return Bases{(gptr)node_bits, (gptr)tri_bits};
}

// One step of one lane (top != 0): mapping_gpu.impala:107-134 for a node, one iteration of :156-174 for a triangle.
// PF (lab): when `prefetch` is set (wave-uniform), a node lane touches both children's records (one dword each, loaded
// straight into a dummy LDS row: no register, nothing waits for it) as soon as their ids have arrived, so that the next
// iteration's fetch of the chosen child finds its line in L1 or already on its way -- the slab tests overlap the round trip.
// TOP: the launch's top-of-tree image is staged in LDS (k_bvh2_top): a node id >= kLdsTag is the byte offset of a 64-byte
// record inside `image` (same layout as Node2, child ids of resident children rewritten the same way), fetched with
// ds_read_b128 instead of through the texture path.
// LAZY: the miss record is not stored up front (start_lane<true>); the step notes in L.found that the ray has a hit record
// and the kernel stores the miss record of the rays that never got one when their chunk ends (finish_lane).
// FENCE (kernels that finish the launch themselves, k_bvh2_top_persist<.., FUSED = 2>): a lane that hands its ray to the deep list
// publishes the list entry and everything it stored for that ray before its workgroup counts itself done.
// SHARED (k_bvh2_top_steal): the ray's tmax lives in LDS (`shared_tmax`: one word per ray of the chunk) because several lanes may be
// working on subtrees of the SAME ray: it is read with the step's other loads, an accepted triangle shortens it with ds_min_f32, and of the
// lanes that accept in one instruction the one that holds the minimum stores the hit record.  ANY: the first acceptance stores -inf, which
// ends the others.
typedef __attribute__((address_space(3))) float lds_float;
// SPILL (> 0: the rows of the lane's LDS window, sp_limit = col + SPILL * kWave): a stack that outgrows the window moves its oldest entries
// to its wave's block of `spill` (the context's buffer; WPG = waves per workgroup) and the ray goes on in its lane (stack_spill /
// stack_reload, traversal_device.h); 0: the ray is handed to the launch's deep list and traced again from the root by the follow-up pass
// (finish_launch).
template <bool ANY, bool PF = false, bool TOP = false, bool LAZY = false, bool FENCE = false, bool SHARED = false, int SPILL = 0,
    int WPG = 1>
__device__ __forceinline__ void bvh2_step(Lane& L, const Bases& base, Hit1* __restrict__ hits, lds_int* sp_limit, Ctl* ctl,
    int* __restrict__ deep_list,
                                          bool prefetch = false, lds_int* pf_row = nullptr, lds_int* image = nullptr,
                                              lds_float* shared_tmax = nullptr,
                                          int* __restrict__ spill = nullptr) {
    const bool is_node = L.top > 0;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef int i32x2 __attribute__((ext_vector_type(2)));
    f32x4 q0, q1, q2;
    i32x2 ch;
    int popped;
    float shared_now = 0.0f;
    if constexpr (TOP && !SHARED) {
        // both kinds of fetch -- LDS image, memory -- and the word under the cursor in flight together (joint_fetch, traversal_device.h)
        const unsigned idx = (unsigned)(is_node ? L.top : ~L.top), stride = is_node ? (unsigned)sizeof(Node2) : (unsigned)sizeof(Tri1);
        const gptr addr = (is_node ? base.node : base.tri) + (size_t)idx * stride;
        // child ids of a node = its bytes 48..55; a triangle lane re-reads its own last 8 bytes so that the load stays inside the array
        joint_fetch(q0, q1, q2, ch, popped, L.top >= kLdsTag, (unsigned)(size_t)image + (unsigned)(L.top - kLdsTag), addr,
            addr + (is_node ? 48u : 40u), L.sp);
    } else {
        if (TOP && L.top >= kLdsTag) {                                  // (lab: the work-stealing kernel keeps the compiler's two branches)
            typedef __attribute__((address_space(3))) const char* lds_bytes;
            const lds_bytes rec = (lds_bytes)image + (unsigned)(L.top - kLdsTag);
            const __attribute__((address_space(3))) f32x4* p = (const __attribute__((address_space(3))) f32x4*)rec;
            q0 = p[0]; q1 = p[1]; q2 = p[2];
            ch = *(const __attribute__((address_space(3))) i32x2*)(rec + 48);
        } else {
            // one address for both kinds: base + index * stride with per-lane selected operands (straight-line code)
            const unsigned idx = (unsigned)(is_node ? L.top : ~L.top), stride = is_node ? (unsigned)sizeof(Node2) : (unsigned)sizeof(Tri1);
            const gptr addr = (is_node ? base.node : base.tri) + (size_t)idx * stride;
            const __attribute__((address_space(1))) f32x4* p = (const __attribute__((address_space(1))) f32x4*)addr;
            q0 = p[0]; q1 = p[1]; q2 = p[2];
            ch = *(const __attribute__((address_space(1))) i32x2*)(addr + (is_node ? 48u : 40u));
        }
        popped = *L.sp;
        if (SHARED) shared_now = *shared_tmax;
        // All four loads must be in flight together: without this barrier the compiler narrows the shared loads to
        // what the triangle branch reads and issues the rest inside the node branch, a second full memory latency.
        // (Whole-vector operands: the loaded register quads stay where the loads put them.)
        asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(ch));
    }
    if (SHARED) {
        L.ray.tmax = shared_now;
        if (ANY && shared_now == -__builtin_inff()) { L.top = 0; return; }        // another lane found this ray's hit
    }
    if (PF && prefetch && is_node) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int c = k ? ch.y : ch.x;
            const bool inner = c > 0;
            const gptr a = (inner ? base.node : base.tri) + (size_t)(unsigned)(inner ? c : ~c) * (inner ? (unsigned)sizeof(Node2)
                : (unsigned)sizeof(Tri1));
            if (c != 0) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a,
                (__attribute__((address_space(3))) void*)pf_row, 4, 0, 0);
        }
    }
    if (is_node) {
        float te0, te1;
        const bool h0 = slab_canonical(L.ray, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, te0) && ch.x != 0;
        const bool h1 = slab_canonical(L.ray, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, te1) && ch.y != 0;
        const bool c0first = te0 < te1, both = h0 && h1;
        L.sp[kWave] = c0first ? ch.y : ch.x;
        L.top = both ? (c0first ? ch.x : ch.y) : (h0 ? ch.x : (h1 ? ch.y : popped));
        L.sp += (both ? kWave : 0) - ((h0 || h1) ? 0 : kWave);
        if (both && L.sp >= sp_limit) {                                 // (`both`: popping the sentinel moves sp below col, which wraps)
            // deeper than the LDS window: the oldest entries move out
            if constexpr (SPILL > 0) stack_spill<SPILL>(L.sp, L.top, sp_limit, spill, WPG, &ctl->err, &ctl->stats[7]);
            else {                                                      // ... or k_bvh2_finish redoes this ray
                deep_list[atomicAdd(&ctl->deep_count, 1)] = L.ray_id;
                L.top = 0;
                if (LAZY) L.found = true;                               // its record is the follow-up pass's business
                if (FENCE) __threadfence();
            }
        }
    } else {
        const int prim_id = __float_as_int(q2.w);
        const float nx = cross_x(q1.x, q1.y, q1.z, q2.x, q2.y, q2.z);
        const float ny = cross_y(q1.x, q1.y, q1.z, q2.x, q2.y, q2.z);
        const float nz = cross_z(q1.x, q1.y, q1.z, q2.x, q2.y, q2.z);
        float t, u, v;
        bool found = false;
        if (intersect_tri(L.ray, q0.x, q0.y, q0.z, q1.x, q1.y, q1.z, q2.x, q2.y, q2.z, nx, ny, nz, t, u, v)) {
            if (SHARED) {
                __builtin_amdgcn_ds_fminf(shared_tmax, ANY ? -__builtin_inff() : t, 0, 0, false);
                // (read after every lane's ds_min of this instruction)
                if (ANY || *shared_tmax == t) store_hit(hits, L.ray_id, prim_id & 0x7FFFFFFF, t, u, v);
            } else store_hit(hits, L.ray_id, prim_id & 0x7FFFFFFF, t, u, v);
            L.ray.tmax = t; found = true;
            if (LAZY) L.found = true;
        }
        const bool leave = prim_id < 0;                               // sentinel: the leaf is done
        L.top = (ANY && found) ? 0 : (leave ? popped : L.top - 1);    // top - 1 == ~(j + 1)
        L.sp -= (leave && !(ANY && found)) ? kWave : 0;
    }
    // popped row 0 while entries are out: they come back
    if constexpr (SPILL > 0) if (L.top >= kSpillMark) stack_reload<SPILL>(L.sp, L.top, sp_limit, spill, WPG);
}

// A fresh ray: loads it, stores the miss record, empty stack (col[0] = the 0 that ends the traversal when popped).
// LAZY: no miss record yet -- nearly every primary ray of a closed scene finds a triangle, and the record then was 16 of the
// 28 bytes the launch wrote per ray (profiles/r02_pmc_counters.json: WRITE_SIZE 1.73 x the Hit1 array); finish_lane stores
// it for the rays that end without a hit.
template <bool LAZY = false>
__device__ __forceinline__ Lane start_lane(const Ray1* __restrict__ rays, Hit1* __restrict__ hits, int ray_id, int any_valid_ray,
    lds_int* col) {
    Lane L;
    L.ray_id = ray_id;
    L.found = false;
    L.ray = load_ray(rays, ray_id >= 0 ? ray_id : any_valid_ray);
    if (!LAZY && ray_id >= 0) store_hit(hits, ray_id, -1, L.ray.tmax, 0.0f, 0.0f);
    // see slab_canonical (after the miss record: it keeps the file's bits)
    L.ray.tmin = canonical(L.ray.tmin); L.ray.tmax = canonical(L.ray.tmax);
    L.top = ray_id >= 0 ? 1 : 0;
    L.sp = col;
    col[0] = 0;
    return L;
}
// LAZY kernels, when a chunk's loop has ended: the miss record (tri_id -1, t = the ray's tmax as the caller wrote it -- re-read,
// the register copy is canonicalised) of every ray that no triangle was accepted for.  (A ray that went to the deep list gets
// one too; the follow-up pass traces it again and stores its final record afterwards.)
__device__ __forceinline__ void finish_lane(const Lane& L, const Ray1* __restrict__ rays, Hit1* __restrict__ hits) {
    if (L.ray_id >= 0 && !L.found) store_hit(hits, L.ray_id, -1, rays[L.ray_id].tmax, 0.0f, 0.0f);
}

// finish_lane without the load: a ray that no triangle was accepted for still holds the tmax it was loaded with -- canonicalised
// (start_lane), which changes the bits of a signalling NaN only: those lanes re-read theirs.
__device__ __forceinline__ void finish_lane_reg(const Lane& L, const Ray1* __restrict__ rays, Hit1* __restrict__ hits) {
    if (L.ray_id >= 0 && !L.found) {
        float tmax = L.ray.tmax;
        if (tmax != tmax) tmax = rays[L.ray_id].tmax;
        store_hit(hits, L.ray_id, -1, tmax, 0.0f, 0.0f);
    }
}

// Are the rays the pixels of an image, row by row?  Then which rays share a wave is the kernel's choice, and an 8 x 8-pixel tile is a
// tighter bundle than 64 pixels of a row: the wave's rays finish closer together (oracle step counts, atrium camera: mean over chunks of
// the longest ray 57.4 steps for row segments, 48.4 for tiles, mean ray 39.3) and touch fewer distinct nodes per load.  What a ray visits
// and where its hit goes do not change: the hits stay bit-identical (measured 1 Mi camera rays: atrium 0.1781 -> 0.1667 ms, gallery 0.295
// -> 0.281, crown 0.1845 -> 0.1875; profiles/r05_grid_tiles.txt). Every wave looks at rays 0, 64 and 128, 256, ... 8192 (one probe per
// lane, in flight with the wave's first rays) and reads them two ways:
// 1. a ray_gen dump of camera rays (the reference's tools/ray_gen/ray_gen.cpp:20-58): dir = d + kx(column) r + ky(row) u, not normalised.
//    Along a row the direction advances by a constant step e, so (dir[i] - dir[0]) . e / |e|^2 is the column of ray i -- it climbs with i
//    and falls back to 0 where the next row starts: the first probe whose column is less than half its index lies in the second row, width
//    = index - column, and the other probes must sit in the columns that width predicts.  Any width, exactly.
// 2. any other per-pixel list (ray_gen's shadow mode -- from a light to the camera rays' hit points, ray_gen.cpp:60-85, the suite's "ao"
//    class --, normalised camera rays, a renderer's shadow or reflection rays in pixel order): probe k is 128 k pixels along the list; in
//    an image of width 128 k* it is the pixel k* ... BELOW ray 0, a near neighbour, while the probes before it are 128, 256, ... pixels
//    away along the row.  Distance = |org - org0|^2 and |dir - dir0|^2, each in units of probe 1's: the first probe closer than an eighth
//    of probe 1 gives the width (a multiple of 128 that divides the ray count); the probe two rows down must be near as well and the probe
//    after it about as far as probe 1.  Measured on the ao rays (1 Mi, profiles/r05_scene_matrix.txt): gallery 0.290 -> 0.217 ms, plant
//    0.095 -> 0.081, atrium 0.156 -> 0.151, crown level.
// Wave-uniform, and the same in every wave of a launch (same rays, same arithmetic).  0 = not recognised: rays in list order, as until
// round 4.  A wrong answer would cost speed, never hits -- any width maps the launch's positions onto its rays one to one
// (k_bvh2_top_auto).
__device__ __forceinline__ int detect_ray_grid(const Ray1* __restrict__ rays, int n) {
    constexpr int kProbe = 128;
    if (n <= 2 * kProbe) return 0;
    const int lane = (int)threadIdx.x % kWave, i = kProbe * (lane + 1);
    const bool valid = i < n;
    const float4* probe = reinterpret_cast<const float4*>(rays + (valid ? i : 0));
    const float4 o0 = reinterpret_cast<const float4*>(rays)[0], d0 = reinterpret_cast<const float4*>(rays)[1],
        d1 = reinterpret_cast<const float4*>(rays + 64)[1], op = probe[0], dp = probe[1];
    const auto lane_value = [](float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); };
    // 1. constant step along the row
    const float ex = (d1.x - d0.x) * (1.0f / 64.0f), ey = (d1.y - d0.y) * (1.0f / 64.0f), ez = (d1.z - d0.z) * (1.0f / 64.0f);
    const float q = ex * ex + ey * ey + ez * ez;
    const float col = ((dp.x - d0.x) * ex + (dp.y - d0.y) * ey + (dp.z - d0.z) * ez) / q;
    const unsigned long long wrapped = __ballot(valid && !(col >= 0.5f * (float)i));
    if (q > 0.0f && wrapped != 0ull) {
        const int first = __ffsll((long long)wrapped) - 1;
        const int w = (int)rintf((float)(kProbe * (first + 1)) - lane_value(col, first));
        if (w >= kProbe && w <= kProbe * kWave && __ballot(valid && !(fabsf(col - (float)(i % w)) < 0.25f)) == 0ull) return w;
    }
    // 2. the pixel below ray 0 is a near neighbour
    const float far_o = (op.x - o0.x) * (op.x - o0.x) + (op.y - o0.y) * (op.y - o0.y) + (op.z - o0.z) * (op.z - o0.z);
    const float far_d = (dp.x - d0.x) * (dp.x - d0.x) + (dp.y - d0.y) * (dp.y - d0.y) + (dp.z - d0.z) * (dp.z - d0.z);
    const float unit_o = lane_value(far_o, 0), unit_d = lane_value(far_d, 0);
    const float terms = (unit_o > 0.0f ? 1.0f : 0.0f) + (unit_d > 0.0f ? 1.0f : 0.0f);
    const float dist = (unit_o > 0.0f ? far_o / unit_o : 0.0f) + (unit_d > 0.0f ? far_d / unit_d : 0.0f);      // probe 1: = terms
    const unsigned long long near = __ballot(valid && lane > 0 && dist < terms * (1.0f / 64.0f));
    if (near == 0ull) return 0;
    const int k = __ffsll((long long)near), w = kProbe * k;                   // probe k = lane k - 1
    if (n % w != 0) return 0;
    if (2 * k <= kWave && kProbe * 2 * k < n && !(lane_value(dist, 2 * k - 1) < terms * (1.0f / 16.0f))) return 0;
    if (k + 1 <= kWave && kProbe * (k + 1) < n) { const float next = lane_value(dist, k);
        if (!(next > 0.25f * terms && next < 4.0f * terms)) return 0; }
    return w;
}

// rays [0, tiled_ray_count) are whole bands of 8 image rows: position p of the launch (64 consecutive positions = one wavefront) is pixel p
// % 64 of tile p / 64
__device__ __forceinline__ int tiled_ray_count(int grid_w, int n) {
    return __builtin_amdgcn_readfirstlane(grid_w > 0 ? (n / (8 * grid_w)) * (8 * grid_w) : 0); }
// the ray of `lane` in the tile at positions [first, first + 64); `first` becomes the tile's first ray
__device__ __forceinline__ int tile_ray(int& first, int lane, int grid_w) {
    // (the division below is redone per chunk: hoisted, its reciprocal would live in a VGPR through the step loop)
    asm volatile("" : "+s"(grid_w));
    const int tiles_per_row = grid_w >> 3, tile = first / kWave, band = __builtin_amdgcn_readfirstlane(tile / tiles_per_row),
        tx = tile - band * tiles_per_row;
    first = band * 8 * grid_w + tx * 8;
    return first + (lane >> 3) * grid_w + (lane & 7);
}
// the same map for ONE position (< tiled_ray_count), per lane: the refill loops draw positions that are not a wave's 64 consecutive ones
__device__ __forceinline__ int tile_ray_at(int pos, int grid_w) {
    asm volatile("" : "+s"(grid_w));
    const unsigned tiles_per_row = (unsigned)grid_w >> 3, tile = (unsigned)pos / kWave, band = tile / tiles_per_row,
        tx = tile - band * tiles_per_row, l = (unsigned)pos % kWave;
    return (int)(band * 8u * (unsigned)grid_w + tx * 8u + (l >> 3) * (unsigned)grid_w + (l & 7u));
}

// PRIO (lab): 0 = none; 1 = a wave raises its issue priority as it ages (48 / 96 / 144 iterations -> s_setprio 1 / 2 / 3);
// 2 = the waves of the second dispatch round (workgroup index >= 8192) run at priority 2 from the start; 3 = both.
template <bool ANY, int LDS_N, int PRIO = 0, bool SPILL = false>
__device__ __forceinline__ void unified_chunk(const Node2* __restrict__ nodes, const Tri1* __restrict__ tris, const Ray1* __restrict__ rays,
                                              Hit1* __restrict__ hits, int n, Ctl* ctl, int* __restrict__ deep_list, lds_int* col,
                                                  int first_ray,
                                              const int* __restrict__ perm = nullptr, int* __restrict__ spill = nullptr, int grid_w = 0) {
    int lane_ray = first_ray + (int)threadIdx.x;
    // (the form the default mapping launches) camera rays in image order: an 8 x 8-pixel tile per wavefront, see detect_ray_grid
    if (SPILL && PRIO == 0) {
        if (grid_w < 0) grid_w = detect_ray_grid(rays, n);
        grid_w = __builtin_amdgcn_readfirstlane(grid_w > 0 && (grid_w & 7) == 0 ? grid_w : 0);
        if (first_ray < tiled_ray_count(grid_w, n)) lane_ray = tile_ray(first_ray, (int)threadIdx.x, grid_w);
    }
    // perm (k_bvh2_single's "sorted" mapping): lane j traces ray perm[j]; its hit still goes to hits[ray id]
    Lane L = start_lane(rays, hits, lane_ray < n ? (perm ? perm[lane_ray] : lane_ray) : -1, perm ? perm[first_ray] : first_ray, col);
    lds_int* const sp_limit = col + LDS_N * kWave;
    const Bases base = make_bases(nodes, tris);
    if (PRIO == 0) {
        while (__ballot(L.top != 0)) {
            if (L.top != 0) bvh2_step<ANY, false, false, false, false, false, SPILL ? LDS_N : 0, 1>(L, base, hits, sp_limit, ctl,
                deep_list, false, nullptr, nullptr, nullptr, spill);
        }
    // lab: triangle turns.  Lanes at a triangle step only every K-th iteration while the wave is
    } else if (PRIO >= 256) {
        // young (iteration < SWITCH), so that most iterations run the node path alone (62 instead of 127 VALU instructions); old
        // waves -- the ones the launch waits for at its end -- go back to one step per lane per iteration.  PRIO = 256 + K * 1024 + SWITCH
        constexpr int K = (PRIO - 256) / 1024, SWITCH = (PRIO - 256) % 1024;
        for (int it = 0; __ballot(L.top != 0); it++) {
            const bool tri_turn = it >= SWITCH || it % K == K - 1 || !__ballot(L.top > 0);
            if (L.top != 0 && (L.top > 0 || tri_turn)) bvh2_step<ANY>(L, base, hits, sp_limit, ctl, deep_list);
        }
    // lab: child prefetch from iteration PRIO - 16 on (row LDS_N + 1 of the LDS block is the dummy target)
    } else if (PRIO >= 16) {
        lds_int* pf_row = col - threadIdx.x + (LDS_N + 1) * kWave;
        for (int it = 0; __ballot(L.top != 0); it++) {
            if (L.top != 0) bvh2_step<ANY, true>(L, base, hits, sp_limit, ctl, deep_list, it >= PRIO - 16, pf_row);
        }
    } else {
        if ((PRIO & 2) && blockIdx.x >= 8192) __builtin_amdgcn_s_setprio(2);
        for (int it = 0; __ballot(L.top != 0); it++) {
            if (PRIO & 1) {
                if (it == 48) __builtin_amdgcn_s_setprio(1);
                if (it == 96) __builtin_amdgcn_s_setprio(2);
                if (it == 144) __builtin_amdgcn_s_setprio(3);
            }
            if (L.top != 0) bvh2_step<ANY>(L, base, hits, sp_limit, ctl, deep_list);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Phased traversal with compaction (k_bvh2_phase).  A launch is a short chain of kernels; every wave of phase p runs
// at most `max_iters` single-step iterations and then SUSPENDS the rays that are still alive -- ray id, current tmax,
// top and the live stack entries, 16 + 4 x depth bytes; the hit record is in memory already -- into a queue in slots
// handed out by one atomic per wave plus a ballot prefix.  Phase p + 1 resumes them 64 to a wave.  The last phase is
// uncapped.  Per ray the sequence of node and triangle tests is untouched (a suspended ray continues exactly where it
// stopped), so results stay bit-identical; what changes is who shares a wavefront with whom:
//  * the rays that outlive their neighbours -- the ones a wave's other 60 lanes used to idle for -- are packed into full
//    waves (random segments: lane utilisation bound 0.38 -> 0.68 with caps 32 / 24, scripts/model_phases.py);
//  * no wave of the capped phases lives longer than `max_iters` round trips, so the two dispatch rounds of a 1 Mi-ray
//    launch end together and the long rays start their remaining steps early and all at once, instead of
//    (start of the last expensive waves) + (their whole life).
// ---------------------------------------------------------------------------------------------
// One global counter word takes ~88 atomics/us on this chip: a slot counter shared by all 16 384 waves of a 1 Mi-ray
// phase would serialise them (measured: the phased launch was 50 % SLOWER than the single kernel).  The queue is
// therefore striped: wave b appends to stripe b % kStripes, which has its own counter in its own 64-byte line and its own
// slot range; phase p + 1 maps workgroup b to chunk b / kStripes of stripe b % kStripes.
constexpr int kStripes = 64;
constexpr int kCounterStride = 16;             // ints between two counters (64 bytes)
constexpr int kMaxPhases = 4;
struct RayQueue {              // SoA over slots; stripe s owns slots [s * stripe_cap, (s + 1) * stripe_cap)
    int* ray; float* tmax; int* top; int* depth; int* stack;      // stack[e * capacity + slot], e < LDS_N
    int capacity, stripe_cap;
};

// RAYS: rays a resuming wave takes (64, 32 or 16).  In the last phase the kernel time is the longest ray's remaining
// steps x the time of one wave iteration, and an iteration executes the node path AND the triangle path whenever the
// wave's rays are in both states: with fewer rays per wave a long ray seldom waits for the other path, and the same
// rays spread over more waves keep more SIMDs issuing (one wave alone issues a VALU instruction every ~5.7 cycles).
template <bool ANY, int LDS_N, bool RESUME, bool CAPPED, int RAYS = kWave>
__global__ __launch_bounds__(kWave) void k_bvh2_phase(const Node2* __restrict__ nodes, const Tri1* __restrict__ tris,
                                                       const Ray1* __restrict__ rays, Hit1* __restrict__ hits, int n,
                                                       Ctl* ctl, int* __restrict__ deep_list, int* __restrict__ qcount, RayQueue in,
                                                           int phase, int max_iters, RayQueue out) {
    __shared__ int lds_raw[(LDS_N + 1) * kWave];
    lds_int* col = (lds_int*)lds_raw + threadIdx.x;
    lds_int* const sp_limit = col + LDS_N * kWave;
    const int stripe = blockIdx.x % kStripes;
    const Bases base = make_bases(nodes, tris);
    // RESUME: workgroup b works through chunks b / kStripes, + gridDim.x / kStripes, ... of its stripe (the grid is sized for
    // the expected number of survivors; more than expected = more than one chunk per wave)
    // written by the previous kernel of the chain
    const int in_count = RESUME ? qcount[((phase - 1) * kStripes + stripe) * kCounterStride] : 0;
    const int k0 = blockIdx.x / kStripes, kstep = RESUME ? gridDim.x / kStripes : 1;
    for (int k = k0; RESUME ? k * RAYS < in_count : k == k0; k += kstep) {
        Lane L;
        if (!RESUME) {
            const int total_chunks = (n + kWave - 1) / kWave;
            int chunk = blockIdx.x;
            constexpr int XCD = 32;                                               // as k_bvh2_single
            const int span = 8 * XCD, full = (total_chunks / span) * span;
            if ((int)blockIdx.x < full) {
                const int x = blockIdx.x % 8, l = blockIdx.x / 8;
                chunk = ((l / XCD) * 8 + x) * XCD + l % XCD;
            }
            const int lane_ray = chunk * kWave + (int)threadIdx.x;
            L = start_lane(rays, hits, lane_ray < n ? lane_ray : -1, chunk * kWave, col);
        } else {
            const int first = stripe * in.stripe_cap + k * RAYS;
            const bool valid = (int)threadIdx.x < RAYS && k * RAYS + (int)threadIdx.x < in_count;
            const unsigned s = (unsigned)(valid ? first + (int)threadIdx.x : first);
            L.ray_id = valid ? in.ray[s] : -1;
            L.ray = load_ray(rays, in.ray[s]);
            L.ray.tmin = canonical(L.ray.tmin); L.ray.tmax = in.tmax[s];          // already canonical
            L.top = valid ? in.top[s] : 0;
            const int depth = in.depth[s];
            for (int e = 0; e < LDS_N; e++) {
                if (!__ballot(valid && e < depth)) break;
                if (valid && e < depth) col[e * kWave] = in.stack[(size_t)e * in.capacity + s];
            }
            L.sp = col + (depth - 1) * kWave;
        }
        if (CAPPED) {
            for (int it = 0; it < max_iters && __ballot(L.top != 0); it++) {
                if (L.top != 0) bvh2_step<ANY>(L, base, hits, sp_limit, ctl, deep_list);
            }
            const unsigned long long alive = __ballot(L.top != 0);
            if (alive) {
                const int lane = threadIdx.x, first = __ffsll((long long)alive) - 1;
                int slot0 = 0;
                if (lane == first) slot0 = atomicAdd(&qcount[(phase * kStripes + stripe) * kCounterStride], __popcll(alive));
                slot0 = __shfl(slot0, first);
                if (L.top != 0) {
                    const unsigned s = (unsigned)(stripe * out.stripe_cap + slot0 + __popcll(alive & ((1ull << lane) - 1ull)));
                    const int depth = (int)(L.sp - col) / kWave + 1;
                    out.ray[s] = L.ray_id; out.tmax[s] = L.ray.tmax; out.top[s] = L.top; out.depth[s] = depth;
                    for (int e = 0; e < depth; e++) out.stack[(size_t)e * out.capacity + s] = col[e * kWave];
                }
            }
        } else {
            while (__ballot(L.top != 0)) {
                if (L.top != 0) bvh2_step<ANY>(L, base, hits, sp_limit, ctl, deep_list);
            }
        }
    }
}

static_assert(kMaxPhases == 4 && kStripes == 64 && kCounterStride == 16, "k_bvh2_finish clears 4 x 64 counters, 16 ints apart");

// BVH2 / Tri1, the default kernel: one 64-ray chunk per wave, single-step schedule (unified_chunk), XCD-aware chunk
// mapping as in k_bvh2_fast.
//
// Measured and NOT kept -- adaptive launch order.  The kernel time of a 1 Mi-ray launch is (start of the last
// expensive waves) + (their life); workgroups start in index order and the chip holds half of them at once, so on
// the atrium, whose expensive rays end the stream, tracing the same rays in REVERSE stream order is 23 % faster
// (0.179 vs 0.221 ms).  Costs are unknown before the launch; sampling them inside it (first dispatch round = every
// other row of chunks, its waves report the rays still alive after 24 steps, the skipped rows are then started
// most-expensive-first from a table sorted by the first workgroup that needs it) was built and was 12 % SLOWER:
// half of the expensive rows cannot start before slots free up anyway, all expensive sampled rows running at once
// take 1.6x longer each, the estimates are not in when the sort has to run (expensive waves report last), and the
// waiting workgroups hold slots.  An age-based s_setprio for long-running waves was slower as well, and so was giving
// every wave two chunks (b and b + grid/2) with idle lanes refilled from the second one (one dispatch round, all waves
// start at t = 0: 0.247 vs 0.214 ms -- the second chunk's expensive rays still start late, inside the wave).
// SPILL: the launch has no more chunks than the context has spill blocks (every launch the default mapping sends here): a stack that
// outgrows the LDS window goes on in its lane (stack_spill); otherwise such rays go to the deep list and k_bvh2_finish.
template <bool ANY, int LDS_N, int XCD, bool TRACE = false, int PRIO = 0, bool SPILL = false>
__global__ __launch_bounds__(kWave) void k_bvh2_single(const Node2* __restrict__ nodes, const Tri1* __restrict__ tris,
                                                        const Ray1* __restrict__ rays, Hit1* __restrict__ hits, int n,
                                                        Ctl* ctl, int* __restrict__ deep_list, const int* __restrict__ perm,
                                                            int* __restrict__ spill, int grid_w) {
    __shared__ int lds_raw[(LDS_N + (PRIO >= 16 && PRIO < 256 ? 2 : 1)) * kWave];
    lds_int* col = (lds_int*)lds_raw + threadIdx.x;
    const unsigned long long t_start = TRACE ? __builtin_amdgcn_s_memrealtime() : 0ull;
    const int total_chunks = (n + kWave - 1) / kWave;
    int chunk = blockIdx.x;
    if (XCD > 0) {
        const int span = 8 * XCD, full = (total_chunks / span) * span;        // region where the mapping is a bijection
        if ((int)blockIdx.x < full) {
            const int x = blockIdx.x % 8, l = blockIdx.x / 8;
            chunk = ((l / XCD) * 8 + x) * XCD + l % XCD;
        }
    }
    unified_chunk<ANY, LDS_N, PRIO, SPILL>(nodes, tris, rays, hits, n, ctl, deep_list, col, chunk * kWave, perm, spill, perm ? 0 : grid_w);
    if (TRACE && threadIdx.x == 0 && ctl->trace && blockIdx.x < 16384) {
        unsigned long long* tr = ctl->trace + 4 * (size_t)blockIdx.x;
        tr[0] = t_start; tr[1] = __builtin_amdgcn_s_memrealtime();
        tr[2] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4))
            | ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32);
        tr[3] = (unsigned long long)chunk;
    }
}

#include "traversal_top.h"           // the default mapping: LDS-staged top of the tree, persistent workgroups

// ---------------------------------------------------------------------------------------------
// Ray sorting for incoherent ray sets (BASELINE config 3: "ray compaction/sorting on"; variant "sorted").
// 3.1's counters say what a set of random segments waits for: every lane fetches its own 64-byte node, 85 % from L2 and 15 %
// from the Infinity Cache, at 97.5 % of what that path delivers.  Only coherence moves that roofline: the same rays
// grouped by the cell of their origin trace 20-25 % faster (profiles/r02_sort_experiment.txt).  Three small kernels build
// the permutation -- a counting sort on 512 Morton cells (3 bits per axis of the scene box, read from the root node):
//   k_raysort_count   256 threads x 16 rays: cell keys, per-block histogram in LDS, block totals added to the 512 global
//                     counters (non-returning atomics, 512 distinct addresses)
//   k_raysort_scan    one workgroup: exclusive scan of the 512 totals -> bin cursors
//   k_raysort_scatter per block: histogram again from the stored keys, one RETURNING atomic per (block, non-empty cell) claims
//                     the block's range of the cell, LDS atomics rank the rays inside it; perm[position] = ray index
// The order of blocks inside a cell is whatever the atomics make it: perm differs from run to run, the hits do not (every ray
// is traced exactly as before and stores to hits[ray index]).  k_bvh2_single then takes ray perm[j] in lane j.
// ---------------------------------------------------------------------------------------------
constexpr int kSortCells = 512, kSortThreads = 256, kSortRaysPerThread = 16, kSortBlockRays = kSortThreads * kSortRaysPerThread;

// 3 bits -> every third bit
__device__ __forceinline__ unsigned spread3(unsigned x) { return (x & 1u) | ((x & 2u) << 2) | ((x & 4u) << 4); }
__device__ __forceinline__ unsigned ray_cell(const Ray1* __restrict__ rays, int i, float lox, float loy, float loz, float sx, float sy,
    float sz) {
    const float4 o = *reinterpret_cast<const float4*>(rays + i);
    const int cx = min(7, max(0, (int)((o.x - lox) * sx))), cy = min(7, max(0, (int)((o.y - loy) * sy))),
        cz = min(7, max(0, (int)((o.z - loz) * sz)));
    return spread3((unsigned)cx) | (spread3((unsigned)cy) << 1) | (spread3((unsigned)cz) << 2);
}

__global__ __launch_bounds__(kSortThreads) void k_raysort_count(const Node2* __restrict__ nodes, const Ray1* __restrict__ rays, int n,
    unsigned short* __restrict__ keys, int* __restrict__ totals) {
    __shared__ int hist[kSortCells];
    for (int k = threadIdx.x; k < kSortCells; k += kSortThreads) hist[k] = 0;
    // scene box = union of the root's child boxes (converter.cpp:318-341 layout; an empty slot is +inf / -inf: min / max ignore it)
    const float* b = nodes[0].bounds;
    const float lox = fminf(b[0], b[6]), hix = fmaxf(b[1], b[7]), loy = fminf(b[2], b[8]), hiy = fmaxf(b[3], b[9]),
        loz = fminf(b[4], b[10]), hiz = fmaxf(b[5], b[11]);
    const float sx = 8.0f / fmaxf(hix - lox, 1e-30f), sy = 8.0f / fmaxf(hiy - loy, 1e-30f), sz = 8.0f / fmaxf(hiz - loz, 1e-30f);
    __syncthreads();
    const int first = blockIdx.x * kSortBlockRays;
#pragma unroll 4
    for (int k = 0; k < kSortRaysPerThread; k++) {
        const int i = first + k * kSortThreads + (int)threadIdx.x;
        if (i < n) { const unsigned c = ray_cell(rays, i, lox, loy, loz, sx, sy, sz); keys[i] = (unsigned short)c; atomicAdd(&hist[c], 1); }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < kSortCells; k += kSortThreads) if (hist[k]) atomicAdd(&totals[k], hist[k]);
}

__global__ __launch_bounds__(kSortCells) void k_raysort_scan(int* __restrict__ totals /* in: counts, out: zero */,
    int* __restrict__ cursor) {
    __shared__ int wave_sum[kSortCells / kWave];
    const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
    const int v = totals[threadIdx.x];
    totals[threadIdx.x] = 0;                                      // ready for the next launch on this stream
    int incl = v;
    for (int o = 1; o < kWave; o <<= 1) { const int up = __shfl_up(incl, o); if (lane >= o) incl += up; }
    if (lane == kWave - 1) wave_sum[wave] = incl;
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wave; w++) before += wave_sum[w];
    cursor[threadIdx.x] = before + incl - v;
}

__global__ __launch_bounds__(kSortThreads) void k_raysort_scatter(const unsigned short* __restrict__ keys, int n, int* __restrict__ cursor,
    int* __restrict__ perm) {
    __shared__ int slot[kSortCells];
    for (int k = threadIdx.x; k < kSortCells; k += kSortThreads) slot[k] = 0;
    __syncthreads();
    const int first = blockIdx.x * kSortBlockRays;
    unsigned short key[kSortRaysPerThread];
#pragma unroll
    for (int k = 0; k < kSortRaysPerThread; k++) {
        const int i = first + k * kSortThreads + (int)threadIdx.x;
        key[k] = i < n ? keys[i] : (unsigned short)0xFFFF;
        if (i < n) atomicAdd(&slot[key[k]], 1);
    }
    __syncthreads();
    // this block's range of cell k
    for (int k = threadIdx.x; k < kSortCells; k += kSortThreads) { const int c = slot[k]; if (c) slot[k] = atomicAdd(&cursor[k], c); }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kSortRaysPerThread; k++) {
        const int i = first + k * kSortThreads + (int)threadIdx.x;
        if (i < n) perm[atomicAdd(&slot[key[k]], 1)] = i;
    }
}

// ---------------------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------------------
constexpr int kHostPageInts = 64, kHostErr = 16;        // (the error word in a cache line of its own, away from the ray-kind words)
struct DeviceState {
    bool  init = false;
    int*  scratch = nullptr;    // [0] ray counter of the first persistent kernels, [16..] Ctl
    int   num_cus = 0;
    int*  deep_list = nullptr;  // ray indices whose stack overflowed the LDS window
    int   deep_cap = 0;
    int*  deep_stack = nullptr; // 64 x 64 ints: global-memory stack of k_bvh2_finish
    unsigned long long* trace = nullptr;   // debug: 16384 x 4 words (instrumented variants)
    int*  queue_mem[2] = {nullptr, nullptr};   // suspended-ray queues of the phased traversal (ping-pong), queue_cap slots each
    int   queue_cap = 0;
    int*  qcount = nullptr;                    // [phase][stripe] suspended-ray counters, 64 bytes apart
    int*  sort_perm = nullptr; unsigned short* sort_keys = nullptr; int sort_cap = 0;     // "sorted" mapping: permutation and cell keys
    int*  sort_totals = nullptr;               // [0, 512) cell counts (zero between launches), [512, 1024) cell cursors
    int4* top_image = nullptr;                 // "top*" mappings: kMaxTopNodes x 64 bytes, rebuilt by every launch
    const Node2* top_image_nodes = nullptr; int top_image_n = 0;
    // schedule history: wave iterations per chunk of the last launch, the order sorted from them
    int*  chunk_cost = nullptr; int* chunk_order = nullptr;
    int   order_rays = 0;                      // ray count of the launch chunk_order was sorted for (0: none)
    // per stripe: {chunks the last two launches both found in their expensive half, half the stripe's chunks}
    int*  order_agree = nullptr;
    // lab: caller-supplied ray permutation of the "top-userperm" mapping (rodent_hip_debug_set_perm)
    const int* debug_perm = nullptr;
    // pinned host memory the kernels store into: [0] / [1] ray-kind reports (host_kinds), [kHostErr] stack-overflow flag (Ctl::host_err)
    int*  host_page = nullptr;
    hipEvent_t timer[2] = {nullptr, nullptr};  // the synchronous entry points' kernel time (rodent_hip_get_kernel_time)
    std::mutex sync_mutex;                     // ... one synchronous call at a time per context (the events are the context's)
    // Ray-kind hint of the default mapping (L_default): host_kinds[0] / [1] = id of the last launch whose rays some workgroup found
    // coherent / incoherent (pinned host memory the kernels store into); hint_* = the ray list the hint is about and the first launch that
    // traced it.
    int*  host_kinds = nullptr; int launch_id = 0; const void* hint_rays = nullptr; int hint_n = 0, hint_first_id = 0;
    int*  tickets = nullptr;                   // persistent "top*p" mappings: chunk tickets per XCD (zero between launches)
    // out-of-window stack entries: spill_slots wave blocks of kSpillWaveInts ints (stack_spill, traversal_device.h; ensure_spill)
    int*  spill = nullptr; int spill_slots = 0;
    Ctl*  ctl() const { return reinterpret_cast<Ctl*>(scratch + 16); }
};
// One DeviceState per (device, stream): launches enqueued on different streams of a device may overlap, so each
// stream gets its own control words, deep-ray list and follow-up stack.
struct DeviceStreams { std::vector<std::pair<hipStream_t, std::unique_ptr<DeviceState>>> ctx; DeviceState* last = nullptr; };
DeviceStreams g_dev[16];
std::mutex  g_mutex;
constexpr size_t kMaxStreamContexts = 64;

// The entry points work on the device they are given and leave the caller's current device as they found it.
struct DeviceGuard {
    int prev = -1; bool changed = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
        if (prev != dev) { HIP_CHECK(hipSetDevice(dev)); changed = true; }
    }
    ~DeviceGuard() { if (changed && prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete; DeviceGuard& operator=(const DeviceGuard&) = delete;
};

DeviceState& device_state(int dev, hipStream_t stream) {
    if (dev < 0 || dev >= 16) { fprintf(stderr, "rodent_hip: invalid device index %d\n", dev); abort(); }
    std::lock_guard<std::mutex> lock(g_mutex);
    DeviceStreams& d = g_dev[dev];
    for (auto& c : d.ctx) if (c.first == stream) { d.last = c.second.get(); return *c.second; }
    if (d.ctx.size() >= kMaxStreamContexts) {
        fprintf(stderr, "rodent_hip: more than %zu streams used on device %d\n", kMaxStreamContexts, dev); abort(); }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || dev >= count) {
        fprintf(stderr, "rodent_hip: no HIP device %d (%d visible)\n", dev, count); abort();
    }
    HIP_CHECK(hipSetDevice(dev));
    auto s = std::make_unique<DeviceState>();
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    s->num_cus = prop.multiProcessorCount;
    HIP_CHECK(hipMalloc(&s->scratch, 64 * sizeof(int)));
    HIP_CHECK(hipMemset(s->scratch, 0, 64 * sizeof(int)));
    static_assert(16 * sizeof(int) + sizeof(Ctl) <= 64 * sizeof(int), "the control block lives in the scratch words");
    HIP_CHECK(hipHostMalloc(&s->host_page, sizeof(int) * kHostPageInts, hipHostMallocCoherent));
    for (int k = 0; k < kHostPageInts; k++) s->host_page[k] = 0;
    s->host_kinds = s->host_page;
    { int* err = s->host_page + kHostErr; HIP_CHECK(hipMemcpy(&s->ctl()->host_err, &err, sizeof(err), hipMemcpyHostToDevice)); }
    HIP_CHECK(hipMalloc(&s->deep_stack, kStackCap * kWave * sizeof(int)));
    s->init = true;
    d.ctx.emplace_back(stream, std::move(s));
    d.last = d.ctx.back().second.get();
    return *d.last;
}
// the context of the stream used last on this device (the debug read-outs and the synchronous entry points)
DeviceState& device_state(int dev) {
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        if (dev >= 0 && dev < 16 && g_dev[dev].last) return *g_dev[dev].last;
    }
    return device_state(dev, nullptr);
}

inline int blocks_for(int n) { return (n + kWave - 1) / kWave; }
constexpr int kQueueWords = 4 + 16;             // per slot: ray, tmax, top, depth + a 16-entry stack window
RayQueue ensure_queue(DeviceState& s, int which, int n) {
    // a stripe receives the survivors of at most ceil(blocks / kStripes) waves
    const int stripe_cap = ((blocks_for(n) + kStripes - 1) / kStripes) * kWave, cap = stripe_cap * kStripes;
    if (cap > s.queue_cap || !s.qcount) {
        std::lock_guard<std::mutex> lock(g_mutex);
        HIP_CHECK(hipDeviceSynchronize());
        if (cap > s.queue_cap) {
            for (int k = 0; k < 2; k++) {
                if (s.queue_mem[k]) HIP_CHECK(hipFree(s.queue_mem[k]));
                HIP_CHECK(hipMalloc(&s.queue_mem[k], sizeof(int) * (size_t)kQueueWords * cap));
            }
            s.queue_cap = cap;
        }
        if (!s.qcount) {
            HIP_CHECK(hipMalloc(&s.qcount, sizeof(int) * kMaxPhases * kStripes * kCounterStride));
            HIP_CHECK(hipMemset(s.qcount, 0, sizeof(int) * kMaxPhases * kStripes * kCounterStride));
        }
    }
    int* m = s.queue_mem[which]; const size_t c = (size_t)s.queue_cap;
    return RayQueue{m, reinterpret_cast<float*>(m + c), m + 2 * c, m + 3 * c, m + 4 * c, s.queue_cap, stripe_cap};
}

void ensure_sort_buffers(DeviceState& s, int n) {
    if (n <= s.sort_cap && s.sort_totals) return;
    std::lock_guard<std::mutex> lock(g_mutex);
    HIP_CHECK(hipDeviceSynchronize());
    if (n > s.sort_cap) {
        if (s.sort_perm) HIP_CHECK(hipFree(s.sort_perm));
        if (s.sort_keys) HIP_CHECK(hipFree(s.sort_keys));
        HIP_CHECK(hipMalloc(&s.sort_perm, sizeof(int) * (size_t)n));
        HIP_CHECK(hipMalloc(&s.sort_keys, sizeof(unsigned short) * (size_t)n));
        s.sort_cap = n;
    }
    if (!s.sort_totals) {
        HIP_CHECK(hipMalloc(&s.sort_totals, sizeof(int) * 2 * kSortCells));
        HIP_CHECK(hipMemset(s.sort_totals, 0, sizeof(int) * 2 * kSortCells));
    }
}

void ensure_deep_list(DeviceState& s, int n) {
    if (n <= s.deep_cap) return;
    std::lock_guard<std::mutex> lock(g_mutex);
    if (n <= s.deep_cap) return;
    HIP_CHECK(hipDeviceSynchronize());
    if (s.deep_list) HIP_CHECK(hipFree(s.deep_list));
    HIP_CHECK(hipMalloc(&s.deep_list, sizeof(int) * (size_t)n));
    s.deep_cap = n;
}

// The blocks the traversal stacks spill into beyond their LDS windows: one per wave slot of a resident generation of the persistent kernels
// (num_cus x 32 waves = 8192), which the one-chunk kernels' launches below rodent_hip_top_min_rays (6144 chunks; 9216 until round 4) fit as
// well. Sized to what the context's launches need (until round 5 every context took all 9 216 blocks = 113 MB with its first
// launch, a 64-ray one included, and a process may hold 64 contexts per device): `slots` wave blocks of 12.5 KB, rounded up to a power of
// two from 64 on, grown -- behind a device synchronisation, like the deep list -- when a later launch needs more; a persistent launch takes
// num_cus x 32 = 8 192 (100 MB).  Touched only by rays deeper than their window.
constexpr int kSpillSlots = 9216;
void ensure_spill(DeviceState& s, int slots) {
    if (slots <= s.spill_slots) return;
    std::lock_guard<std::mutex> lock(g_mutex);
    if (slots <= s.spill_slots) return;
    int want = 64;
    while (want < slots) want *= 2;
    want = std::min(want, kSpillSlots);
    HIP_CHECK(hipDeviceSynchronize());
    if (s.spill) HIP_CHECK(hipFree(s.spill));
    HIP_CHECK(hipMalloc(&s.spill, sizeof(int) * (size_t)want * kSpillWaveInts));
    s.spill_slots = want;
}
// the wave slots of one resident generation of 16-wave workgroups (what every persistent BVH2 kernel launches)
int resident_wave_slots(const DeviceState& s) { return ((s.num_cus * 2 + kStripes - 1) / kStripes) * kStripes * 16; }

// a persistent grid's wave slots must fit the context's spill blocks (they do on every gfx950 part: 256 CUs x 32 waves)
int spill_checked(int groups, int waves) {
    if ((long)groups * waves > kSpillSlots) {
        fprintf(stderr, "rodent_hip: %d x %d resident waves exceed the %d stack spill blocks\n", groups, waves, kSpillSlots); abort(); }
    return groups;
}

// Waits for everything enqueued on `stream`, then reads and clears the stack-overflow flag: a word of pinned host memory the kernels' error
// paths store into (Ctl::host_err) -- no device-to-host copy on the synchronous entry points' path.
bool read_and_clear_error_flags(DeviceState& s, hipStream_t stream) {
    HIP_CHECK(hipStreamSynchronize(stream));
    volatile int* page = s.host_page;
    const bool raised = page[kHostErr] != 0;
    page[kHostErr] = 0;
    return raised;
}
// the reference's entry points have no return code (bench_traversal.impala:17-21: message + abort)
void check_error_flag(DeviceState& s, hipStream_t stream) {
    if (read_and_clear_error_flags(s, stream)) {
        fprintf(stderr, "rodent_hip: traversal stack overflow (more than %d entries)\n", kStackCap); abort(); }
}

// How many Node2 records the mapped range behind `nodes` can hold (0 when the runtime does not know the pointer): the bound
// inside which the persistent kernel may dereference the ids of an image built by an earlier launch.
int mapped_node_ids(const Node2* nodes) {
    hipDeviceptr_t range_base = nullptr; size_t range_size = 0;
    if (hipMemGetAddressRange(&range_base, &range_size, (hipDeviceptr_t)nodes) != hipSuccess) { (void)hipGetLastError(); return 0; }
    const size_t bytes = (size_t)((const char*)range_base + range_size - (const char*)nodes);
    return (int)std::min<size_t>(bytes / sizeof(Node2), 0x3FFFFFFF);
}

// RODENT_HIP_RAY_GRID: -1 (default) = k_bvh2_top_auto recognises camera rays in image order by itself (detect_ray_grid) and traces them as
// 8 x 8-pixel tiles; 0 = never (rays in list order, as until round 5); > 0 = the image's width, taken on trust (experiments)
int g_ray_grid = [] { const char* e = getenv("RODENT_HIP_RAY_GRID"); return e ? atoi(e) : -1; }();
// rodent_hip_schedule_history()
int g_schedule_history = [] { const char* e = getenv("RODENT_HIP_SCHEDULE_HISTORY"); return e && atoi(e) ? 1 : 0; }();
// workgroups (of one wave) of the follow-up kernels in the shipped mappings: a launch's deep rays are restarted 256 x 64 at a time
constexpr int kFinishGroups = 256;
#define LAUNCH_ARGS DeviceState& s, const Node2* nodes, const Tri1* tris, const Ray1* rays, Hit1* hits, int n, hipStream_t stream

// the one-chunk kernel's waves learn the image width one memory round trip BEFORE they can load their rays (the persistent kernel's learn
// it with their first rays): with fewer than 128 Ki rays -- every wave resident at once, the launch is one wave's latency -- that costs
// more than tiles gain (64 Ki camera rays 0.078 -> 0.083 ms, 128 Ki 0.088 = 0.089, 256 Ki 0.106 -> 0.102, 384 Ki 0.130 -> 0.118:
// profiles/r05_threshold_sweep_grid.txt)
constexpr int kGridMinRays = 2048 * kWave;
template <bool ANY, int LDS_N, int XCD, bool TR = false, int PRIO = 0> void L_single(LAUNCH_ARGS) {
    ensure_deep_list(s, n);
    if (PRIO == 0 && !TR && blocks_for(n) <= kSpillSlots) {        // every chunk has a spill block: deep stacks stay in their lanes
        ensure_spill(s, blocks_for(n));
        hipLaunchKernelGGL((k_bvh2_single<ANY, LDS_N, XCD, false, 0, true>), dim3(blocks_for(n)), dim3(kWave), 0, stream, nodes, tris,
            rays, hits, n, s.ctl(), s.deep_list, (const int*)nullptr, s.spill, n >= kGridMinRays ? g_ray_grid : 0);
    } else
        hipLaunchKernelGGL((k_bvh2_single<ANY, LDS_N, XCD, TR, PRIO>), dim3(blocks_for(n)), dim3(kWave), 0, stream, nodes, tris, rays,
            hits, n, s.ctl(), s.deep_list, (const int*)nullptr, (int*)nullptr, 0);
    hipLaunchKernelGGL((k_bvh2_finish<ANY>), dim3(kFinishGroups), dim3(kWave), 0, stream, nodes, tris, rays, hits, s.ctl(), s.deep_list,
        s.deep_stack, (int*)nullptr);
}


template <bool ANY, int LDS_N, int TOPN, int WAVES, bool PREFETCH, bool SORTED, int OCC, bool TRACE = false, int PRIO = 0, int FUSED = 0,
    bool LAZY = false> void launch_top_persist(LAUNCH_ARGS, int max_id) {
    ensure_deep_list(s, n);
    ensure_spill(s, ((s.num_cus * (OCC / WAVES) + kStripes - 1) / kStripes) * kStripes * WAVES);
    if (!s.top_image || !s.tickets) {
        std::lock_guard<std::mutex> lock(g_mutex);
        if (!s.top_image) { HIP_CHECK(hipMalloc(&s.top_image, kMaxTopNodes * sizeof(Node2)));
            HIP_CHECK(hipMemset(s.top_image, 0, kMaxTopNodes * sizeof(Node2))); }
        if (!s.tickets) {
            HIP_CHECK(hipMalloc(&s.tickets, sizeof(int) * kMaxPhases * kStripes * kCounterStride));      // the size k_bvh2_finish clears
            HIP_CHECK(hipMemset(s.tickets, 0, sizeof(int) * kMaxPhases * kStripes * kCounterStride));
        }
    }
    s.top_image_nodes = nullptr;
    const int* perm = nullptr;
    if (SORTED) {
        ensure_sort_buffers(s, n);
        const int blocks = (n + kSortBlockRays - 1) / kSortBlockRays;
        hipLaunchKernelGGL(k_raysort_count, dim3(blocks), dim3(kSortThreads), 0, stream, nodes, rays, n, s.sort_keys, s.sort_totals);
        hipLaunchKernelGGL(k_raysort_scan, dim3(1), dim3(kSortCells), 0, stream, s.sort_totals, s.sort_totals + kSortCells);
        hipLaunchKernelGGL(k_raysort_scatter, dim3(blocks), dim3(kSortThreads), 0, stream, s.sort_keys, n, s.sort_totals + kSortCells,
            s.sort_perm);
        perm = s.sort_perm;
    }
    if (!SORTED && PRIO == -1) perm = s.debug_perm;                          // lab "top-userperm"
    // one resident generation, the same number in every stripe
    const int groups = spill_checked(((s.num_cus * (OCC / WAVES) + kStripes - 1) / kStripes) * kStripes, WAVES);
    // chunks of the fullest stripe
    const int total_chunks = blocks_for(n), stride = ((total_chunks + 31) / 32 + kStripes - 1) / kStripes * 32;
    // (with the history on, the launch is followed by k_bvh2_top_finish_history whatever FUSED says)
    if (g_schedule_history && !SORTED && !TRACE && PRIO == 0 && FUSED != 1 && !PREFETCH && stride <= kMaxStripeChunks) {
        // schedule history: this launch records its chunks' costs; it draws them in the order the previous launch of the same
        // size left behind, if there is one (all on `stream`: the follow-up kernel writes the order before the next launch reads it)
        if (!s.chunk_cost) {
            std::lock_guard<std::mutex> lock(g_mutex);
            HIP_CHECK(hipMalloc(&s.chunk_cost, sizeof(int) * kStripes * kMaxStripeChunks));
            HIP_CHECK(hipMalloc(&s.chunk_order, sizeof(int) * kStripes * kMaxStripeChunks));
            HIP_CHECK(hipMalloc(&s.order_agree, sizeof(int) * 2 * kStripes));
            HIP_CHECK(hipMemset(s.order_agree, 0xFF, sizeof(int) * 2 * kStripes));
        }
        const bool have_previous = s.order_rays == n;
        const History hist{have_previous ? s.chunk_order : nullptr, s.chunk_cost, stride, s.order_agree};
        hipLaunchKernelGGL((k_bvh2_top_persist<ANY, LDS_N, TOPN, WAVES, false, OCC, false, 0, false, true>), dim3(groups),
            dim3(kWave * WAVES), 0, stream, nodes, tris, rays, hits, n, s.ctl(),
                           s.deep_list, perm, s.top_image, s.tickets, max_id, s.spill, hist);
        hipLaunchKernelGGL((k_bvh2_top_finish_history<ANY>), dim3(kStripes), dim3(kHistoryThreads), 0, stream, nodes, tris, rays, hits,
            s.ctl(), s.deep_list, s.deep_stack, s.tickets,
                           s.top_image, TOPN, total_chunks, (const int*)s.chunk_cost, s.chunk_order, stride, have_previous ? 1 : 0,
                               s.order_agree);
        s.order_rays = n;
        return;
    }
    s.order_rays = 0;
    hipLaunchKernelGGL((k_bvh2_top_persist<ANY, LDS_N, TOPN, WAVES, PREFETCH, OCC, TRACE, PRIO, FUSED, false, LAZY>), dim3(groups),
        dim3(kWave * WAVES), 0, stream, nodes, tris, rays, hits, n, s.ctl(), s.deep_list,
                       perm, s.top_image, s.tickets, max_id, s.spill, History{nullptr, nullptr, 0, nullptr});
    if (!FUSED) hipLaunchKernelGGL((k_bvh2_top_finish<ANY>), dim3(kFinishGroups), dim3(kWave), 0, stream, nodes, tris, rays, hits, s.ctl(),
        s.deep_list, s.deep_stack, s.tickets, s.top_image, TOPN);
}
template <bool ANY, int LDS_N, int TOPN, int WAVES, bool PREFETCH = true, bool SORTED = false, int OCC = 32, bool TRACE = false,
    int PRIO = 0, int FUSED = 0, bool LAZY = false> void L_top_persist(LAUNCH_ARGS) {
    launch_top_persist<ANY, LDS_N, TOPN, WAVES, PREFETCH, SORTED, OCC, TRACE, PRIO, FUSED, LAZY>(s, nodes, tris, rays, hits, n, stream,
        mapped_node_ids(nodes));
}

// The default mapping: launches that fill the chip at least once take the persistent kernel with the LDS image, smaller ones
// the single kernel (256 Ki rays: 0.099 ms against 0.117 ms, 512 Ki: 0.138 against 0.144, 768 Ki: 0.181 against 0.155 --
// staging and validating the image does not pay yet), and so do
// launches whose node array the runtime cannot give a mapped range for (nothing bounds the ids of an older image then).
// 384 Ki rays: the measured cross-over with wave-major first tickets (profiles/r05_spread_tickets.txt; 576 Ki until round 4,
// profiles/r02_threshold_sweep.txt)
constexpr int kTopMinRays = 6144 * kWave;
int g_top_min_rays = kTopMinRays;               // rodent_hip_top_min_rays()
// the node ids the persistent LDS-image kernel may assume mapped, or 0: this launch takes the one-chunk kernel
int top_kernel_ids(const Node2* nodes, int n) { return n < g_top_min_rays ? 0 : mapped_node_ids(nodes); }
// rodent_hip_ray_kind_hint() / RODENT_HIP_KIND_HINT: 1 = the default mapping remembers what its kernels saw of a ray list and sends one
// that was incoherent throughout to k_bvh2_top_refill from its second launch on (+2 ... 4 % on random segments).  OFF by default from round
// 5 on: which kernel a launch gets must not depend on earlier launches or on when an asynchronous caller's previous launch happened to
// finish; k_bvh2_top_auto's choice per wave needs no memory.
int g_kind_hint = [] { const char* e = getenv("RODENT_HIP_KIND_HINT"); return e && atoi(e) ? 1 : 0; }();
// FUSED = 2: the launch finishes itself (its last workgroup does the follow-up kernel's work; fences on the rare paths only): one
// kernel per call instead of two, +1.1 % / +1.8 % on the benchmark's primary / random set in wall-clock terms (bench.py, 100 steps).
template <bool ANY, int LDS_N, int TOPN, int WAVES, bool PREFETCH, int FUSED = 2> void L_chunks(LAUNCH_ARGS) {
    const int max_id = top_kernel_ids(nodes, n);
    if (max_id == 0) L_single<ANY, 16, 32>(s, nodes, tris, rays, hits, n, stream);
    else launch_top_persist<ANY, LDS_N, TOPN, WAVES, PREFETCH, false, 32, false, 0, FUSED>(s, nodes, tris, rays, hits, n, stream, max_id);
}
// Round 4: the persistent kernel chooses per wave -- once, from the first 64 rays the wave draws -- between whole chunks (rays that share
// an origin or a direction) and lane refill (anything else): k_bvh2_top_auto, traversal_top.h.  With the schedule history on, launches keep
// the chunk kernel (the history orders CHUNKS).
void ensure_top_buffers(DeviceState& s) {
    if (s.top_image && s.tickets) return;
    std::lock_guard<std::mutex> lock(g_mutex);
    if (!s.top_image) { HIP_CHECK(hipMalloc(&s.top_image, kMaxTopNodes * sizeof(Node2)));
        HIP_CHECK(hipMemset(s.top_image, 0, kMaxTopNodes * sizeof(Node2))); }
    if (!s.tickets) {
        HIP_CHECK(hipMalloc(&s.tickets, sizeof(int) * kMaxPhases * kStripes * kCounterStride));      // the size k_bvh2_finish clears
        HIP_CHECK(hipMemset(s.tickets, 0, sizeof(int) * kMaxPhases * kStripes * kCounterStride));
    }
}
template <bool ANY, int LDS_N, int TOPN, int WAVES, int REFILL, int MODE = 0, bool FUSED = true,
    bool LAZY = false> void L_default(LAUNCH_ARGS) {
    const int max_id = top_kernel_ids(nodes, n);
    if (max_id == 0) { L_single<ANY, 16, 32>(s, nodes, tris, rays, hits, n, stream); return; }
    if (g_schedule_history) {
        launch_top_persist<ANY, LDS_N, TOPN, WAVES, false, false, 32, false, 0, 2>(s, nodes, tris, rays, hits, n, stream, max_id); return; }
    ensure_deep_list(s, n);
    ensure_top_buffers(s);
    const int groups = spill_checked(((s.num_cus * (32 / WAVES) + kStripes - 1) / kStripes) * kStripes, WAVES);
    ensure_spill(s, groups * WAVES);
    s.top_image_nodes = nullptr; s.order_rays = 0;   // one resident generation, the same number in every stripe
    // Which kernel?  k_bvh2_top_auto decides per wave and is right for any list; but its refill loop, compiled under the chunk loop's
    // register budget, runs a few per cent behind k_bvh2_top_refill's (profiles/r04_sweep_auto.log).  With the ray-kind hint ON (off by
    // default, see g_kind_hint) every workgroup reports what its first wave saw (report_ray_kind) and a list that earlier launches found
    // incoherent throughout goes to k_bvh2_top_refill -- from the second launch on the same (pointer, count); a stale or missing hint costs
    // speed, never correctness.
    if (s.hint_rays != rays || s.hint_n != n) { s.hint_rays = rays; s.hint_n = n; s.hint_first_id = s.launch_id + 1; }
    const int id = ++s.launch_id;
    const volatile int* kinds = s.host_kinds;
    const bool hinting = MODE == 0 && g_kind_hint;
    int* const report_to = hinting ? s.host_kinds : nullptr;
    // incoherent: the newest report of "incoherent" is about this list and newer than the newest report of "coherent" (a list of both kinds
    // reports both in one launch)
    const bool incoherent = hinting && kinds[1] >= s.hint_first_id && kinds[0] < kinds[1];
    if (incoherent) {
        hipLaunchKernelGGL((k_bvh2_top_refill<ANY, LDS_N, TOPN, WAVES, REFILL, false, false>), dim3(groups), dim3(kWave * WAVES), 0,
            stream, nodes, tris, rays, hits, n, s.ctl(), s.deep_list,
                           (const int4*)s.top_image, s.tickets, max_id, s.spill, report_to, id);
        hipLaunchKernelGGL((k_bvh2_top_finish<ANY>), dim3(1), dim3(kWave), 0, stream, nodes, tris, rays, hits, s.ctl(), s.deep_list,
            s.deep_stack, s.tickets, s.top_image, TOPN);
        return;
    }
    hipLaunchKernelGGL((k_bvh2_top_auto<ANY, LDS_N, TOPN, WAVES, REFILL, MODE, FUSED, LAZY>), dim3(groups), dim3(kWave * WAVES), 0, stream,
        nodes, tris, rays, hits, n, s.ctl(), s.deep_list,
                       s.top_image, s.tickets, max_id, s.spill, report_to, id, g_ray_grid);
    if (!FUSED) hipLaunchKernelGGL((k_bvh2_top_finish<ANY>), dim3(1), dim3(kWave), 0, stream, nodes, tris, rays, hits, s.ctl(),
        s.deep_list, s.deep_stack, s.tickets, s.top_image, TOPN);
}

// "refill": the persistent kernel with lane refill (traversal_top.h), for ray sets whose rays differ widely in cost -- incoherent ones: the
// benchmark's random segments +5 % at 1 Mi rays per launch, +13 % (closest hit) / +18 % (any hit) at 8 Mi,
// profiles/r03_sweep_refill_big_random.log; coherent camera rays LOSE 13 ... 18 % (neighbouring rays stop being in step), which is why it
// is a variant the caller asks for and not the default.
template <bool ANY, int LDS_N, int TOPN, int WAVES, int REFILL, bool ADAPT = false, bool FENCE = false> void L_top_refill(LAUNCH_ARGS) {
    if (top_kernel_ids(nodes, n) == 0) { L_single<ANY, 16, 32>(s, nodes, tris, rays, hits, n, stream); return; }      // (as L_default)
    ensure_deep_list(s, n);
    if (!s.top_image || !s.tickets) {
        std::lock_guard<std::mutex> lock(g_mutex);
        if (!s.top_image) { HIP_CHECK(hipMalloc(&s.top_image, kMaxTopNodes * sizeof(Node2)));
            HIP_CHECK(hipMemset(s.top_image, 0, kMaxTopNodes * sizeof(Node2))); }
        if (!s.tickets) {
            HIP_CHECK(hipMalloc(&s.tickets, sizeof(int) * kMaxPhases * kStripes * kCounterStride));      // the size k_bvh2_finish clears
            HIP_CHECK(hipMemset(s.tickets, 0, sizeof(int) * kMaxPhases * kStripes * kCounterStride));
        }
    }
    s.top_image_nodes = nullptr;
    const int groups = spill_checked(((s.num_cus * (32 / WAVES) + kStripes - 1) / kStripes) * kStripes, WAVES);
    ensure_spill(s, groups * WAVES);
    hipLaunchKernelGGL((k_bvh2_top_refill<ANY, LDS_N, TOPN, WAVES, REFILL, ADAPT, FENCE>), dim3(groups), dim3(kWave * WAVES), 0, stream,
        nodes, tris, rays, hits, n, s.ctl(), s.deep_list,
                       (const int4*)s.top_image, s.tickets, mapped_node_ids(nodes), s.spill, (int*)nullptr, 0);
    hipLaunchKernelGGL((k_bvh2_top_finish<ANY>), dim3(1), dim3(kWave), 0, stream, nodes, tris, rays, hits, s.ctl(), s.deep_list,
        s.deep_stack, s.tickets, s.top_image, TOPN);
}



// Phased traversal: caps of the capped phases (the last, uncapped phase follows).  Launches too small to fill the chip
// once take the single kernel.
// "sorted": permutation by origin cell, then the single kernel through it
template <bool ANY, int LDS_N> void L_sorted(LAUNCH_ARGS) {
    ensure_deep_list(s, n);
    ensure_sort_buffers(s, n);
    const int blocks = (n + kSortBlockRays - 1) / kSortBlockRays;
    hipLaunchKernelGGL(k_raysort_count, dim3(blocks), dim3(kSortThreads), 0, stream, nodes, rays, n, s.sort_keys, s.sort_totals);
    hipLaunchKernelGGL(k_raysort_scan, dim3(1), dim3(kSortCells), 0, stream, s.sort_totals, s.sort_totals + kSortCells);
    hipLaunchKernelGGL(k_raysort_scatter, dim3(blocks), dim3(kSortThreads), 0, stream, s.sort_keys, n, s.sort_totals + kSortCells,
        s.sort_perm);
    hipLaunchKernelGGL((k_bvh2_single<ANY, LDS_N, 32, false, 0>), dim3(blocks_for(n)), dim3(kWave), 0, stream, nodes, tris, rays, hits, n,
        s.ctl(), s.deep_list, (const int*)s.sort_perm, (int*)nullptr, 0);
    hipLaunchKernelGGL((k_bvh2_finish<ANY>), dim3(kFinishGroups), dim3(kWave), 0, stream, nodes, tris, rays, hits, s.ctl(), s.deep_list,
        s.deep_stack, (int*)nullptr);
}

int g_phased_min_rays = 4096 * kWave;           // rodent_hip_phased_min_rays()
struct PhaseCaps { int count; int cap[3]; };
constexpr PhaseCaps kPhaseCaps[] = {{2, {40, 24}}, {2, {32, 24}}, {1, {40}}, {1, {32}}, {1, {48}}, {2, {48, 32}}, {3, {32, 32, 32}}, {3,
    {24, 24, 24}}, {2, {24, 24}}, {2, {64, 32}}};
template <bool ANY, int LDS_N, int CAPS, int LAST_RAYS = kWave> void L_phased(LAUNCH_ARGS) {
    constexpr PhaseCaps caps = kPhaseCaps[CAPS];
    static_assert(caps.count + 1 <= kMaxPhases, "too many phases");
    if (n < g_phased_min_rays) { L_single<ANY, LDS_N, 32>(s, nodes, tris, rays, hits, n, stream); return; }
    ensure_deep_list(s, n);
    RayQueue q[2] = {ensure_queue(s, 0, n), ensure_queue(s, 1, n)};
    const int blocks = blocks_for(n);
    // grids of the resuming phases: sized for the share of rays expected to survive (a multiple of kStripes; if more
    // survive, waves take several chunks one after the other)
    auto resume_grid = [&](int p) { const int g = blocks >> p; return ((g < kStripes ? kStripes : g) + kStripes - 1) / kStripes * kStripes;
        };
    hipLaunchKernelGGL((k_bvh2_phase<ANY, LDS_N, false, true>), dim3(blocks), dim3(kWave), 0, stream, nodes, tris, rays, hits, n, s.ctl(),
        s.deep_list, s.qcount, q[1], 0, caps.cap[0], q[0]);
    for (int p = 1; p < caps.count; p++)
        hipLaunchKernelGGL((k_bvh2_phase<ANY, LDS_N, true, true>), dim3(resume_grid(p)), dim3(kWave), 0, stream, nodes, tris, rays, hits,
            n, s.ctl(), s.deep_list, s.qcount, q[(p - 1) & 1], p, caps.cap[p], q[p & 1]);
    hipLaunchKernelGGL((k_bvh2_phase<ANY, LDS_N, true, false, LAST_RAYS>), dim3(resume_grid(caps.count) * (kWave / LAST_RAYS)),
        dim3(kWave), 0, stream, nodes, tris, rays, hits, n, s.ctl(), s.deep_list, s.qcount, q[(caps.count - 1) & 1], caps.count, 0,
        q[caps.count & 1]);
    hipLaunchKernelGGL((k_bvh2_finish<ANY>), dim3(kFinishGroups), dim3(kWave), 0, stream, nodes, tris, rays, hits, s.ctl(), s.deep_list,
        s.deep_stack, s.qcount);
}

#include "traversal_wide.h"          // BVH4 / BVH8 + Tri4: k_wide_single, k_wide_top_persist, k_wide_finish, L_wide_single, L_wide_top
int wide_top_min_rays() { return g_top_min_rays; }
#ifdef RODENT_HIP_LAB
#include "lab/top_kernels.h"         // lab build only: superseded forms of the LDS-image kernels
#include "lab/top_launchers.h"
#include "lab/defer_kernels.h"       // lab build only: deferred leaves (round 6)
#include "lab/coop8_kernel.h"        // lab build only: wave-cooperative BVH8 for any-hit rays (round 6)
#include "lab/traversal_variants.h"  // lab build only: the kernels that were measured and lost, instrumented builds
#endif

using Launch2 = void (*)(LAUNCH_ARGS);
struct Variant2 { const char* name; const char* kernel[2]; Launch2 launch[2]; };
#define K2(name, kname, fn, ...) {name, {kname "<false," #__VA_ARGS__ ">", kname "<true," #__VA_ARGS__ ">"}, {&fn<false, __VA_ARGS__>, \
    &fn<true, __VA_ARGS__>}}
const Variant2 kVariants2[] = {
    // 0 = default (used by the reference-named entry points).  All variants keep the reference's per-ray
    // visit order and are bit-identical; they differ in how a wavefront schedules its 64 rays.
    //                                                              LDS_N TOPN WAVES PREFETCH
    //                                                              LDS_N TOPN WAVES REFILL
    // default: LDS-staged top of the tree (255 nodes), persistent 16-wave workgroups, the last one finishes the launch; a wave traces rays
    // that share an origin or a direction as whole chunks and refills idle lanes otherwise (launches under rodent_hip_top_min_rays:
    // k_bvh2_single)
    K2("top",                "k_bvh2_top_auto",      L_default, 15, 255, 16, 32),
    //                                                        LDS_N XCD_GROUP
    // single-step schedule, one 64-ray chunk per workgroup, XCD-aware 32-chunk groups (default of rounds 1-2)
    K2("fast",               "k_bvh2_single",        L_single, 16, 32),
    K2("fast-noxcd",         "k_bvh2_single",        L_single, 16, 0),                 // same kernel, workgroup b traces chunk b
    //                                                       LDS_N CAPS (index into kPhaseCaps) [LAST_RAYS]
    // phased traversal with ray compaction: one capped phase of 40 iterations, then the rest
    K2("phased",             "k_bvh2_phase",         L_phased, 16, 2),
    // rays grouped by the Morton cell of their origin first (for incoherent ray sets)
    K2("sorted",             "k_bvh2_single",        L_sorted, 16),
    //                                                                    LDS_N TOPN WAVES REFILL (idle lanes that trigger a refill)
    // the default's persistent workgroups, but a wave replaces finished rays instead of waiting for the last ray of a chunk (for incoherent
    // ray sets)
    K2("refill",             "k_bvh2_top_refill",    L_top_refill, 15, 255, 16, 32),
#ifdef RODENT_HIP_LAB
#include "lab/variant_rows_bvh2.inc"      // ~120 rows: everything that was swept on the way
#endif
};
constexpr int kNumVariants2 = sizeof(kVariants2) / sizeof(kVariants2[0]);

// BVH4 / BVH8 + Tri4 (traversal_wide.h).  LDS window: 16 entries for BVH4, 24 for BVH8 (deepest stack on the atrium's
// benchmark dumps: 15 and 21); deeper rays go to k_wide_finish.
using LaunchW = void (*)(WIDE_LAUNCH_ARGS);
struct VariantW { const char* name; const char* kernel[2]; LaunchW launch[2]; };
#define KW(name, kname, fn, ...) {name, {kname "<false," #__VA_ARGS__ ">", kname "<true," #__VA_ARGS__ ">"}, {&fn<false, __VA_ARGS__>, \
    &fn<true, __VA_ARGS__>}}
const VariantW kVariants4[] = {
    //                                                   N LDS_N XCD_GROUP
    //                                                N LDS_N
    // default: persistent 16-wave workgroups, the top 85 nodes staged in LDS by every workgroup (launches under rodent_hip_top_min_rays:
    // k_wide_single)
    KW("top",                "k_wide_top_persist",   L_wide_top, 4, 16),
    // one 64-ray chunk per workgroup, every node from memory (default of round 2)
    KW("single",             "k_wide_single",        L_wide_single, 4, 16, 32),
    KW("single-noxcd",       "k_wide_single",        L_wide_single, 4, 16, 0),
#ifdef RODENT_HIP_LAB
#include "lab/variant_rows_wide4.inc"
#endif
};
const VariantW kVariants8[] = {
    KW("top",                "k_wide_top_persist",   L_wide_top, 8, 24),               // default: ... the top 73 nodes
    KW("single",             "k_wide_single",        L_wide_single, 8, 24, 32),
    KW("single-noxcd",       "k_wide_single",        L_wide_single, 8, 24, 0),
#ifdef RODENT_HIP_LAB
#include "lab/variant_rows_wide8.inc"
#endif
};
constexpr int kNumVariants4 = sizeof(kVariants4) / sizeof(kVariants4[0]);
constexpr int kNumVariants8 = sizeof(kVariants8) / sizeof(kVariants8[0]);
inline const VariantW* wide_variants(int width, int* count) {
    if (width == 4) { *count = kNumVariants4; return kVariants4; }
    if (width == 8) { *count = kNumVariants8; return kVariants8; }
    *count = 0; return nullptr;
}

template <bool ANY>
void launch_bvh2(DeviceState& s, const Node2* nodes, const Tri1* tris, const Ray1* rays, Hit1* hits, int n, int variant,
    hipStream_t stream) {
    if (n <= 0) return;
    if (variant < 0 || variant >= kNumVariants2) { fprintf(stderr, "rodent_hip: unknown BVH2 variant %d\n", variant); abort(); }
    kVariants2[variant].launch[ANY ? 1 : 0](s, nodes, tris, rays, hits, n, stream);
    HIP_CHECK(hipGetLastError());
}

void launch_wide(int width, bool any_hit, DeviceState& s, const void* nodes, const Tri4* tris, const Ray1* rays, Hit1* hits, int n,
    int variant, hipStream_t stream) {
    if (n <= 0) return;
    int count = 0;
    const VariantW* table = wide_variants(width, &count);
    if (variant < 0 || variant >= count) { fprintf(stderr, "rodent_hip: unknown BVH%d variant %d\n", width, variant); abort(); }
    table[variant].launch[any_hit ? 1 : 0](s, nodes, tris, rays, hits, n, stream);
    HIP_CHECK(hipGetLastError());
}

int default_variant(int width) {
    const char* e = getenv(width == 2 ? "RODENT_HIP_BVH2_VARIANT" : (width == 4 ? "RODENT_HIP_BVH4_VARIANT" : "RODENT_HIP_BVH8_VARIANT"));
    return e ? atoi(e) : 0;
}

} // namespace

extern "C" {

void hip_traverse_bvh2_tri1_async(int32_t dev, const Node2* nodes, const Tri1* tris, const Ray1* rays, Hit1* hits,
                                  int32_t num_rays, int32_t any_hit, int32_t variant, void* stream) {
    DeviceGuard on(dev);
    DeviceState& s = device_state(dev, (hipStream_t)stream);
    if (any_hit) launch_bvh2<true>(s, nodes, tris, rays, hits, num_rays, variant, (hipStream_t)stream);
    else         launch_bvh2<false>(s, nodes, tris, rays, hits, num_rays, variant, (hipStream_t)stream);
}

void hip_traverse_bvh4_tri4_async(int32_t dev, const Node4* nodes, const Tri4* tris, const Ray1* rays, Hit1* hits,
                                  int32_t num_rays, int32_t any_hit, int32_t variant, void* stream) {
    DeviceGuard on(dev);
    DeviceState& s = device_state(dev, (hipStream_t)stream);
    launch_wide(4, any_hit != 0, s, nodes, tris, rays, hits, num_rays, variant, (hipStream_t)stream);
}

void hip_traverse_bvh8_tri4_async(int32_t dev, const Node8* nodes, const Tri4* tris, const Ray1* rays, Hit1* hits,
                                  int32_t num_rays, int32_t any_hit, int32_t variant, void* stream) {
    DeviceGuard on(dev);
    DeviceState& s = device_state(dev, (hipStream_t)stream);
    launch_wide(8, any_hit != 0, s, nodes, tris, rays, hits, num_rays, variant, (hipStream_t)stream);
}

int32_t rodent_hip_check_errors(int32_t dev, void* stream) {
    DeviceGuard on(dev);
    DeviceState& s = device_state(dev, (hipStream_t)stream);
    return read_and_clear_error_flags(s, (hipStream_t)stream) ? 1 : 0;
}

// The reference's host times its GPU kernels with anydsl_get_kernel_time() (tools/bench_traversal/bench_traversal.cpp:125-133): the AnyDSL
// runtime's accumulated KERNEL time in microseconds -- no synchronisation, no copy.  The synchronous entry points keep the same account:
// HIP events around what they enqueue -- the context's buffers are allocated BEFORE the first event, so a first call does not book its
// hipMalloc as kernel time; where a mapping is two kernels (the one-chunk kernel + its follow-up) the microsecond between them is included
// -- added up after the call's own synchronisation in one process-wide sum over all devices, like the reference's.  One synchronous call at
// a time per (device, null stream) context.
}  // extern "C"
namespace {
std::atomic<uint64_t> g_kernel_ns{0};
template <typename Launch> void timed_sync_call(int32_t dev, int32_t num_rays, Launch launch) {
    DeviceGuard on(dev);
    DeviceState& s = device_state(dev, nullptr);
    std::lock_guard<std::mutex> one_call(s.sync_mutex);
    if (!s.timer[0]) { HIP_CHECK(hipEventCreate(&s.timer[0])); HIP_CHECK(hipEventCreate(&s.timer[1])); }
    if (num_rays > 0) {                                                   // every lazy allocation of the default mappings
        ensure_deep_list(s, num_rays); ensure_top_buffers(s);
        ensure_spill(s, num_rays < g_top_min_rays && blocks_for(num_rays) <= kSpillSlots ? blocks_for(num_rays) : resident_wave_slots(s));
    }
    HIP_CHECK(hipEventRecord(s.timer[0], nullptr));
    launch(s);
    HIP_CHECK(hipEventRecord(s.timer[1], nullptr));
    // synchronises; aborts on a stack overflow like the reference's error()
    check_error_flag(s, nullptr);
    float ms = 0.0f;
    HIP_CHECK(hipEventElapsedTime(&ms, s.timer[0], s.timer[1]));
    g_kernel_ns.fetch_add((uint64_t)((double)ms * 1e6 + 0.5), std::memory_order_relaxed);
}
}  // namespace
extern "C" {
uint64_t rodent_hip_get_kernel_time(void) { return g_kernel_ns.load(std::memory_order_relaxed) / 1000u; }

void amdgpu_intersect_single_ray1_bvh2_tri1(int32_t dev, const Node2* nodes, const Tri1* tris, const Ray1* rays, Hit1* hits,
    int32_t num_rays) {
    timed_sync_call(dev, num_rays,
        [&](DeviceState& s) { launch_bvh2<false>(s, nodes, tris, rays, hits, num_rays, default_variant(2), nullptr); });
}
void amdgpu_occluded_single_ray1_bvh2_tri1(int32_t dev, const Node2* nodes, const Tri1* tris, const Ray1* rays, Hit1* hits,
    int32_t num_rays) {
    timed_sync_call(dev, num_rays,
        [&](DeviceState& s) { launch_bvh2<true>(s, nodes, tris, rays, hits, num_rays, default_variant(2), nullptr); });
}
void hip_intersect_single_ray1_bvh4_tri4(int32_t dev, const Node4* nodes, const Tri4* tris, const Ray1* rays, Hit1* hits,
    int32_t num_rays) {
    timed_sync_call(dev, num_rays,
        [&](DeviceState& s) { launch_wide(4, false, s, nodes, tris, rays, hits, num_rays, default_variant(4), nullptr); });
}
void hip_occluded_single_ray1_bvh4_tri4(int32_t dev, const Node4* nodes, const Tri4* tris, const Ray1* rays, Hit1* hits, int32_t num_rays) {
    timed_sync_call(dev, num_rays,
        [&](DeviceState& s) { launch_wide(4, true, s, nodes, tris, rays, hits, num_rays, default_variant(4), nullptr); });
}
void hip_intersect_single_ray1_bvh8_tri4(int32_t dev, const Node8* nodes, const Tri4* tris, const Ray1* rays, Hit1* hits,
    int32_t num_rays) {
    timed_sync_call(dev, num_rays,
        [&](DeviceState& s) { launch_wide(8, false, s, nodes, tris, rays, hits, num_rays, default_variant(8), nullptr); });
}
void hip_occluded_single_ray1_bvh8_tri4(int32_t dev, const Node8* nodes, const Tri4* tris, const Ray1* rays, Hit1* hits, int32_t num_rays) {
    timed_sync_call(dev, num_rays,
        [&](DeviceState& s) { launch_wide(8, true, s, nodes, tris, rays, hits, num_rays, default_variant(8), nullptr); });
}

int32_t rodent_hip_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count;
}
int32_t rodent_hip_num_variants(int32_t bvh_width) {
    if (bvh_width == 2) return kNumVariants2;
    int count = 0; wide_variants(bvh_width, &count); return count;
}
const char* rodent_hip_variant_name(int32_t bvh_width, int32_t variant) {
    if (bvh_width == 2) return variant >= 0 && variant < kNumVariants2 ? kVariants2[variant].name : "";
    int count = 0; const VariantW* t = wide_variants(bvh_width, &count);
    return variant >= 0 && variant < count ? t[variant].name : "";
}
const char* rodent_hip_kernel_name(int32_t bvh_width, int32_t variant, int32_t any_hit) {
    if (bvh_width == 2) return variant >= 0 && variant < kNumVariants2 ? kVariants2[variant].kernel[any_hit ? 1 : 0] : "";
    int count = 0; const VariantW* t = wide_variants(bvh_width, &count);
    return variant >= 0 && variant < count ? t[variant].kernel[any_hit ? 1 : 0] : "";
}
void rodent_hip_phased_min_rays(int32_t rays) { g_phased_min_rays = rays < 0 ? 4096 * kWave : rays; }
// lab: see "top-userperm"
void rodent_hip_debug_set_perm(int32_t dev, const int32_t* device_perm) { device_state(dev).debug_perm = device_perm; }
void rodent_hip_schedule_history(int32_t enable) { g_schedule_history = enable ? 1 : 0; }
void rodent_hip_top_min_rays(int32_t rays) { g_top_min_rays = rays < 0 ? kTopMinRays : rays; }
void rodent_hip_ray_kind_hint(int32_t enable) { g_kind_hint = enable ? 1 : 0; }
void rodent_hip_ray_grid(int32_t width) { g_ray_grid = width; }
int32_t rodent_hip_is_lab_build(void) {
#ifdef RODENT_HIP_LAB
    return 1;
#else
    return 0;
#endif
}
/* Debug aid for the instrumented ("stats-*") variants: copies the 8 phase counters to out[] and clears them. */
/* Debug aid: enables the per-wave timeline of the instrumented variants and copies it out
 * (16384 records x 4 words: start tick, end tick, hw_id | xcc_id << 32, outer iterations | rays << 32). */
void rodent_hip_read_trace(int32_t dev, uint64_t* out) {
    DeviceState& s = device_state(dev);
    HIP_CHECK(hipSetDevice(dev));
    HIP_CHECK(hipDeviceSynchronize());
    const size_t bytes = 16384 * 4 * sizeof(uint64_t);
    if (!s.trace) {
        HIP_CHECK(hipMalloc(&s.trace, bytes));
        HIP_CHECK(hipMemset(s.trace, 0, bytes));
        HIP_CHECK(hipMemcpy(&s.ctl()->trace, &s.trace, sizeof(void*), hipMemcpyHostToDevice));
    }
    if (out) HIP_CHECK(hipMemcpy(out, s.trace, bytes, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemset(s.trace, 0, bytes));
}
void rodent_hip_read_stats(int32_t dev, uint64_t* out) {
    DeviceState& s = device_state(dev);
    HIP_CHECK(hipSetDevice(dev));
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpy(out, s.ctl()->stats, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemset(s.ctl()->stats, 0, 8 * sizeof(uint64_t)));
}
const char* rodent_hip_version(void) { return "rodent_hip 0.2 (gfx950)"; }

} // extern "C"

// lab/top_kernels.h -- LAB BUILD ONLY (-DRODENT_HIP_LAB, librodent_hip_lab.so): forms of the LDS-image kernels that were measured and did
// not ship. Included by traversal.hip after traversal_top.h (same anonymous namespace; uses Ctl, Lane, bvh2_step, start_lane, make_bases,
// stage_top_image, finish_launch, build_top_image and the stripe constants).  Nothing in here is part of the product library.
//   k_bvh2_top             one chunk per workgroup wave, the image rebuilt in front of every launch (round 2; superseded by the persistent
//                          form)
//   k_bvh2_top_refill_wpe  k_bvh2_top_refill compiled under the default kernel's pinned register budget
//   k_bvh2_top_steal       work stealing inside the wave (round 4: modelled at 1.2 - 1.5 x, measured -5 ... -12 %)
#pragma once

template <bool ANY, int LDS_N, int XCD, int TOPN, int WAVES>
__global__ __launch_bounds__(kWave * WAVES, 32 / WAVES) void k_bvh2_top(const Node2* __restrict__ nodes, const Tri1* __restrict__ tris,
                                                             const Ray1* __restrict__ rays, Hit1* __restrict__ hits, int n,
                                                             Ctl* ctl, int* __restrict__ deep_list, const int* __restrict__ perm,
                                                             const int4* __restrict__ top_image) {
    constexpr int kStackInts = WAVES * (LDS_N + 1) * kWave;
    static_assert((kStackInts + TOPN * 16) * 4 * (32 / WAVES) <= 160 * 1024, "32 waves per CU must fit their stacks and images in LDS");
    static_assert(XCD % WAVES == 0, "a workgroup's chunks stay inside one XCD group");
    __shared__ __attribute__((aligned(16))) int lds_raw[kStackInts + TOPN * 16];
    const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
    lds_int* col = (lds_int*)lds_raw + wave * (LDS_N + 1) * kWave + lane;
    lds_int* image = (lds_int*)lds_raw + kStackInts;
    // the image first (in flight while the ray is loaded and set up)
    constexpr int kStage = (TOPN * 4 + kWave * WAVES - 1) / (kWave * WAVES);
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    i32x4 stage[kStage];
#pragma unroll
    for (int k = 0; k < kStage; k++) { const int j = k * kWave * WAVES + (int)threadIdx.x;
        if (j < TOPN * 4) stage[k] = reinterpret_cast<const i32x4*>(top_image)[j]; }
    const int total_chunks = (n + kWave - 1) / kWave;
    int chunk = blockIdx.x * WAVES + wave;
    if (XCD > 0) {
        const int span = 8 * XCD, full = (total_chunks / span) * span;        // region where the mapping is a bijection
        if ((int)blockIdx.x * WAVES < full) {
            const int x = blockIdx.x % 8, l = (blockIdx.x / 8) * WAVES + wave;
            chunk = ((l / XCD) * 8 + x) * XCD + l % XCD;
        }
    }
    const int first_ray = chunk * kWave, lane_ray = first_ray + lane;
    const bool live_chunk = first_ray < n;                                   // (whole waves beyond the last chunk only help with the image)
    Lane L = start_lane(rays, hits, live_chunk && lane_ray < n ? (perm ? perm[lane_ray] : lane_ray) : -1,
        live_chunk ? (perm ? perm[first_ray] : first_ray) : 0, col);
    if (L.top != 0) L.top = kLdsTag;                                         // the root is record 0 of the image
#pragma unroll
    for (int k = 0; k < kStage; k++) {
        const int j = k * kWave * WAVES + (int)threadIdx.x;
        if (j < TOPN * 4) reinterpret_cast<__attribute__((address_space(3))) i32x4*>(image)[j] = stage[k];
    }
    if (WAVES > 1) __syncthreads();
    lds_int* const sp_limit = col + LDS_N * kWave;
    const Bases base = make_bases(nodes, tris);
    while (__ballot(L.top != 0)) {
        if (L.top != 0) bvh2_step<ANY, false, true>(L, base, hits, sp_limit, ctl, deep_list, false, nullptr, image);
    }
}


template <bool ANY, int LDS_N, int TOPN, int WAVES, int REFILL>
__global__ __launch_bounds__(kWave * WAVES) __attribute__((amdgpu_waves_per_eu(8, 8))) void
    k_bvh2_top_refill_wpe(const Node2* __restrict__ nodes, const Tri1* __restrict__ tris,
                                                                    const Ray1* __restrict__ rays, Hit1* __restrict__ hits, int n,
                                                                    Ctl* ctl, int* __restrict__ deep_list,
                                                                        const int4* __restrict__ top_image, int* __restrict__ tickets,
                                                                        int max_id, int* spill) {
    top_refill_body<ANY, LDS_N, TOPN, WAVES, REFILL, false, false>(nodes, tris, rays, hits, n, ctl, deep_list, top_image, tickets, max_id,
        spill);
}

// LAB: work stealing inside the wave (lab variants "steal*", round 4; scripts/model_packet.py --steal).  MEASURED AND
// LOST: the model (fewer, fuller wave iterations: 1.2 x on camera rays, 1.5 x on random segments) did not survive the hardware -- the
// shared tmax and the loop form cost 0 - 5 % before anything is stolen, and with stealing the launches are 5 - 12 % SLOWER
// (profiles/r04_sweep_steal.log).  What a 1 Mi-ray launch waits for at its end -- and what a wave of incoherent rays waits for all the time
// -- is a few lanes walking long paths while the others idle.  A lane's stack entries are independent subtrees of its ray's traversal, so
// from iteration I0 of a chunk on, every EVERY iterations, each idle lane takes over the top stack entry of a lane that has one: it copies
// that lane's ray (16 ds_bpermutes for the whole wave) and walks the subtree beside its owner. The ray's tmax is shared through LDS
// (bvh2_step<SHARED>): whoever accepts a triangle shortens it for all lanes working on that ray, and the hit record belongs to the ray.
// Row LDS_N of a wave's LDS block (the spare row above the stack windows) is the scratch of the pairing, row LDS_N + 1 holds the 64 shared
// tmax words: with LDS_N = 14 the footprint is k_bvh2_top_persist's. This is NOT the reference's visit order any more: a stolen subtree is
// walked earlier, against a tmax the owner may not have shortened yet. t is the MINIMUM over all accepted triangles here, which is not
// quite the reference's rule either: its acceptance test (t_raw <= |det| * tmax) lets a later triangle whose quotient rounds one ulp ABOVE
// tmax replace the record and raise tmax by that ulp; ds_min_f32 keeps the smaller one. Measured against oracle B1: 0 of 1 Mi camera rays,
// 228 - 232 of 1 Mi random segments differ, all by one ulp in t (another triangle at that distance), with or without stealing; any-hit
// answers are the oracle's.  The shipped mappings keep the reference's order and rule bit for bit.
template <bool ANY, int LDS_N, int TOPN, int WAVES, int I0, int EVERY>
__global__ __launch_bounds__(kWave * WAVES) __attribute__((amdgpu_waves_per_eu(8, 8))) void
    k_bvh2_top_steal(const Node2* __restrict__ nodes, const Tri1* __restrict__ tris,
                                                                  const Ray1* __restrict__ rays, Hit1* __restrict__ hits, int n,
                                                                  Ctl* ctl, int* __restrict__ deep_list, int4* __restrict__ top_image,
                                                                      int* __restrict__ tickets, int max_id) {
    constexpr int kRows = LDS_N + 2, kStackInts = WAVES * kRows * kWave, kGroup = 32;
    static_assert((kStackInts + TOPN * 16) * 4 * (32 / WAVES) <= 160 * 1024, "32 waves per CU must fit their stacks and images in LDS");
    static_assert((EVERY & (EVERY - 1)) == 0, "EVERY is a power of two");
    __shared__ __attribute__((aligned(16))) int lds_raw[kStackInts + TOPN * 16];
    const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
    lds_int* const wave_rows = (lds_int*)lds_raw + wave * kRows * kWave;
    lds_int* const col = wave_rows + lane;
    lds_int* const scratch = wave_rows + LDS_N * kWave;
    lds_float* const tmax_row = (lds_float*)(wave_rows + (LDS_N + 1) * kWave);
    lds_int* image = (lds_int*)lds_raw + kStackInts;
    const int root = stage_top_image<TOPN, kWave * WAVES>(nodes, (const int4*)top_image, image, (lds_int*)lds_raw, ctl, max_id)
        ? kLdsTag : 1;
    if (root != 1 && threadIdx.x == 0) atomicAdd(&ctl->stats[6], 1ull);
    const int total_chunks = (n + kWave - 1) / kWave, stripe = blockIdx.x % kStripes, stripe_waves = (gridDim.x / kStripes) * WAVES;
    int* counter = tickets + stripe * kCounterStride;
    int t = stripe_rank(wave);
    lds_int* const sp_limit = col + LDS_N * kWave;
    const Bases base = make_bases(nodes, tris);
    const unsigned long long below = (1ull << lane) - 1ull;
    unsigned long long moved = 0;
    for (;;) {
        const int group_first = ((t / kGroup) * kStripes + stripe) * kGroup;
        if (group_first >= total_chunks) break;
        const int chunk = group_first + t % kGroup;
        if (chunk < total_chunks) {
            const int first_ray = chunk * kWave, lane_ray = first_ray + lane;
            Lane L = start_lane(rays, hits, lane_ray < n ? lane_ray : -1, first_ray, col);
            if (L.top != 0) L.top = root;
            int slot = lane;                                                 // whose tmax word this lane works against
            tmax_row[lane] = L.ray.tmax;
            for (int it = 0; __ballot(L.top != 0); it++) {
                if (L.top != 0) bvh2_step<ANY, false, true, false, true, true>(L, base, hits, sp_limit, ctl, deep_list, false, nullptr,
                    image, tmax_row + slot);
                if (it >= I0 && ((it - I0) & (EVERY - 1)) == 0) {
                    const unsigned long long idle = __ballot(L.top == 0), donors = __ballot(L.top != 0 && L.sp > col);
                    const int pairs = min(__popcll(idle), __popcll(donors));
                    if (pairs > 0) {
                        const int donor_rank = __popcll(donors & below), thief_rank = __popcll(idle & below);
                        const bool gives = ((donors >> lane) & 1ull) && donor_rank < pairs, takes = ((idle >> lane) & 1ull)
                            && thief_rank < pairs;
                        int entry = 0;
                        if (gives) { entry = *L.sp; L.sp -= kWave; scratch[donor_rank] = lane; }
                        wave_lds_sync();
                        const int from = takes ? scratch[thief_rank] : lane;
                        // (the scratch row is a stack slot again from the next step on)
                        wave_lds_sync();
                        entry = __shfl(entry, from);
                        L.ray.ox = __shfl(L.ray.ox, from); L.ray.oy = __shfl(L.ray.oy, from); L.ray.oz = __shfl(L.ray.oz, from);
                        L.ray.dx = __shfl(L.ray.dx, from); L.ray.dy = __shfl(L.ray.dy, from); L.ray.dz = __shfl(L.ray.dz, from);
                        L.ray.idx = __shfl(L.ray.idx, from); L.ray.idy = __shfl(L.ray.idy, from); L.ray.idz = __shfl(L.ray.idz, from);
                        L.ray.iox = __shfl(L.ray.iox, from); L.ray.ioy = __shfl(L.ray.ioy, from); L.ray.ioz = __shfl(L.ray.ioz, from);
                        L.ray.tmin = __shfl(L.ray.tmin, from);
                        L.ray_id = __shfl(L.ray_id, from);
                        slot = __shfl(slot, from);
                        if (takes) { L.top = entry; L.sp = col; *col = 0; }
                        if (lane == 0) moved += (unsigned long long)pairs;
                    }
                }
            }
        }
        int t_next = 0;
        if (lane == 0) t_next = atomicAdd(counter, 1);
        t = stripe_waves + __builtin_amdgcn_readfirstlane(t_next);
    }
    if (lane == 0 && moved) atomicAdd(&ctl->stats[3], moved);              // stats[3]: stack entries that changed lanes (read by the tests)
    // the workgroup that finishes last does the follow-up work (k_bvh2_top_persist, FUSED == 2)
    __syncthreads();
    if (threadIdx.x == 0) lds_raw[0] = __hip_atomic_fetch_add(&ctl->counter, 1, __ATOMIC_RELAXED,
        __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
    __syncthreads();
    if (!lds_raw[0] || wave != 0) return;
    const int deep = __hip_atomic_load(&ctl->deep_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (deep > 0) __threadfence();
    const bool stale = __hip_atomic_load(&ctl->reserved, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    finish_launch<ANY>(nodes, tris, rays, hits, ctl, deep_list, (lds_int*)lds_raw, tickets, 0, 1, deep);
    if (stale) {
        build_top_image(nodes, top_image, TOPN, (lds_int*)lds_raw);
        if (lane == 0) ctl->reserved = 0;
    }
}

// lab/defer_kernels.h -- LAB BUILD ONLY: deferred leaves.
//
// The single-step schedule issues the triangle half of the step (~65 of ~132 VALU instructions) in ~99 % of a refilled wave's iterations
// for ~6 % of its lanes (profiles/r05_tri_turns_experiment.txt).  "Triangle turns" (lanes WAIT for a common triangle iteration) lost
// because a waiting lane keeps its lane slot.  Here a lane that reaches a leaf does not wait: it notes the leaf in a small FIFO of its own
// (PEND words of LDS behind its stack window) and goes on traversing against the tmax it has; when DRAIN lanes of the wave hold pending
// leaves -- or nobody can take a node step any more -- the whole wave runs triangle steps only, every lane on ITS OWN pending leaves, in
// the order it met them, until all FIFOs are empty.  The triangle code then runs for >= DRAIN / 64 of the lanes in its first round instead
// of ~6 %.
//
// What changes for a ray: nothing about the ORDER of its triangle tests (its leaves are met in the reference's depth-first order, entry
// distances do not depend on tmax, and a lane tests its own leaves first in, first out against its running tmax:
// mapping_gpu.impala:156-174, intersection.impala:181-182) -- but between noting a leaf and testing it the ray walks on against a STALE
// tmax, so it may enter nodes the reference prunes: a superset of the reference's visits, in the same order.  A triangle of such an extra
// leaf is accepted only where the slab test (with the tmax the triangle before it set) and the triangle test disagree in the last bit -- a
// hit on a shared edge or vertex, a coplanar duplicate.  Any-hit rays: the first accepted triangle is the reference's (nothing is pruned
// before it): bit-identical records. Closest-hit rays: identical up to such ties; scripts/defer_experiment.py counts them against the
// oracle on all 2 x 1 Mi benchmark rays.
#pragma once

// joint_fetch (traversal_device.h) with a mask of its own for the memory kind: lanes that only pop (a leaf being noted) fetch nothing.
__device__ __forceinline__ void joint_fetch3(vf4& q0, vf4& q1, vf4& q2, vi2& ids, int& popped, bool from_mem, bool from_lds,
    unsigned lds_addr, gbytes addr, gbytes addr_ids, lds_int* sp) {
    const unsigned long long mem_mask = __ballot(from_mem), lds_mask = __ballot(from_lds);
    const unsigned sp_addr = (unsigned)(size_t)sp;
    unsigned long long save;
    asm volatile("s_mov_b64 %[save], exec\n\t"
                 "s_and_b64 exec, %[save], %[mm]\n\t"
                 "s_cbranch_execz .Ljoint_e_%=\n\t"
                 "global_load_dwordx4 %[q1], %[a], off offset:16\n\t"
                 "global_load_dwordx4 %[q0], %[a], off\n\t"
                 "global_load_dwordx4 %[q2], %[a], off offset:32\n\t"
                 "global_load_dwordx2 %[ch], %[ac], off\n"
                 ".Ljoint_e_%=:\n\t"
                 "s_and_b64 exec, %[save], %[lm]\n\t"
                 "s_cbranch_execz .Ljoint_f_%=\n\t"
                 "ds_read_b128 %[q0], %[l]\n\t"
                 "ds_read_b128 %[q1], %[l] offset:16\n\t"
                 "ds_read_b128 %[q2], %[l] offset:32\n\t"
                 "ds_read_b64 %[ch], %[l] offset:48\n"
                 ".Ljoint_f_%=:\n\t"
                 "s_mov_b64 exec, %[save]\n\t"
                 "ds_read_b32 %[pop], %[sp]\n\t"
                 "s_waitcnt vmcnt(0) lgkmcnt(0)"
                 : [q0] "=&v"(q0), [q1] "=&v"(q1), [q2] "=&v"(q2), [ch] "=&v"(ids), [pop] "=&v"(popped), [save] "=&s"(save)
                 : [a] "v"(addr), [ac] "v"(addr_ids), [l] "v"(lds_addr), [sp] "v"(sp_addr), [mm] "s"(mem_mask), [lm] "s"(lds_mask)
                 : "memory", "scc");
}

// One step of a lane that can advance: a node step (mapping_gpu.impala:107-134) whose leaf children go to the lane's FIFO instead of
// becoming its top, or -- a leaf that came off the stack -- noting that leaf and popping.  `fifo`: the lane's column of the PEND rows
// behind its window; `pend`: entries in it.
template <int LDS_N, int PEND, int WPG>
__device__ __forceinline__ void defer_step(Lane& L, int& pend, const Bases& base, lds_int* sp_limit, lds_int* fifo, Ctl* ctl,
    lds_int* image, int* __restrict__ spill) {
    const bool is_node = L.top > 0, in_image = L.top >= kLdsTag;
    vf4 q0, q1, q2;
    vi2 ch;
    int popped;
    const gptr addr = base.node + (size_t)(unsigned)L.top * (unsigned)sizeof(Node2);
    joint_fetch3(q0, q1, q2, ch, popped, is_node && !in_image, in_image, (unsigned)(size_t)image + (unsigned)(L.top - kLdsTag), addr,
        addr + 48u, L.sp);
    if (is_node) {
        float te0, te1;
        const bool h0 = slab_canonical(L.ray, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, te0) && ch.x != 0;
        const bool h1 = slab_canonical(L.ray, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, te1) && ch.y != 0;
        const bool c0first = te0 < te1, both = h0 && h1, any = h0 || h1;
        const int first = both ? (c0first ? ch.x : ch.y) : (h0 ? ch.x : ch.y);      // the child visited next (if any)
        const int second = c0first ? ch.y : ch.x;                                   // the one after it (if both)
        // leaves among (first, second) go to the FIFO, in that order, while there is room
        const bool note1 = any && first < 0 && pend < PEND;
        if (note1) { fifo[pend * kWave] = first; pend++; }
        const bool note2 = note1 && both && second < 0 && pend < PEND;
        if (note2) { fifo[pend * kWave] = second; pend++; }
        const bool push = both && !note1, pop = !any || (note1 && (!both || note2));
        L.sp[kWave] = second;                                                       // (only kept when `push`)
        L.top = pop ? popped : (note1 ? second : first);
        L.sp += push ? kWave : (pop ? -kWave : 0);
        if (push && L.sp >= sp_limit) stack_spill<LDS_N>(L.sp, L.top, sp_limit, spill, WPG, &ctl->err, &ctl->stats[7]);
    // a leaf off the stack, and room for it (the caller holds back lanes without)
    } else {
        fifo[pend * kWave] = L.top; pend++;
        L.top = popped;
        L.sp -= kWave;
    }
    if (L.top >= kSpillMark) stack_reload<LDS_N>(L.sp, L.top, sp_limit, spill, WPG);
}

// Triangle steps only: every lane tests the leaves of its FIFO, oldest first, one triangle per round, until all FIFOs of the wave are
// empty.
template <bool ANY, int PEND>
__device__ __forceinline__ void defer_drain(Lane& L, int& pend, const Bases& base, Hit1* __restrict__ hits, lds_int* fifo,
    unsigned long long* rounds = nullptr) {
    static_assert(PEND >= 1 && PEND <= 4, "the FIFO is read into registers");
    int e[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < PEND; k++) e[k] = k < pend ? fifo[k * kWave] : 0;
    pend = 0;
    int cur = e[0];
    while (__ballot(cur != 0)) {
        if (rounds) { if (threadIdx.x % kWave == 0) atomicAdd(&rounds[0], 1ull); if (cur != 0) atomicAdd(&rounds[1], 1ull); }
        if (cur != 0) {
            const gptr addr = base.tri + (size_t)(unsigned)~cur * (unsigned)sizeof(Tri1);
            const __attribute__((address_space(1))) vf4* p = (const __attribute__((address_space(1))) vf4*)addr;
            const vf4 q0 = p[0], q1 = p[1], q2 = p[2];
            const int prim_id = __float_as_int(q2.w);
            const float nx = cross_x(q1.x, q1.y, q1.z, q2.x, q2.y, q2.z);
            const float ny = cross_y(q1.x, q1.y, q1.z, q2.x, q2.y, q2.z);
            const float nz = cross_z(q1.x, q1.y, q1.z, q2.x, q2.y, q2.z);
            float t, u, v;
            bool found = false;
            if (intersect_tri(L.ray, q0.x, q0.y, q0.z, q1.x, q1.y, q1.z, q2.x, q2.y, q2.z, nx, ny, nz, t, u, v)) {
                store_hit(hits, L.ray_id, prim_id & 0x7FFFFFFF, t, u, v);
                L.ray.tmax = t; found = true;
            }
            const bool leave = prim_id < 0;
            // the ray is done: what is left of its stack and FIFO is dropped
            if (ANY && found) { L.top = 0; cur = 0; }
            else if (leave) { cur = e[1]; e[1] = e[2]; e[2] = e[3]; e[3] = 0; }
            else cur -= 1;                                                          // ~(j + 1)
        }
    }
}

// k_bvh2_top_auto with deferred leaves.  LDS per wave: LDS_N + 1 window rows + PEND FIFO rows (LDS_N + 1 + PEND = 16: the shipped kernel's
// footprint). MODE as k_bvh2_top_auto's (0: per wave, 1: whole chunks, 2: refill).  STATS: drain rounds / lane-rounds into stats[3] /
// stats[4] (instrumented build).
template <bool ANY, int LDS_N, int PEND, int TOPN, int WAVES, int REFILL, int DRAIN, int MODE = 0, bool STATS = false>
__global__ __launch_bounds__(kWave * WAVES) __attribute__((amdgpu_waves_per_eu(8, 8))) void
    k_bvh2_top_defer(const Node2* __restrict__ nodes, const Tri1* __restrict__ tris,
                                                                  const Ray1* __restrict__ rays, Hit1* __restrict__ hits, int n,
                                                                  Ctl* ctl, int* __restrict__ deep_list, int4* __restrict__ top_image,
                                                                      int* __restrict__ tickets, int max_id,
                                                                  int* spill, int grid_w) {
    constexpr int kRows = LDS_N + 1 + PEND, kStackInts = WAVES * kRows * kWave, kGroupRays = 32 * kWave;
    static_assert((kStackInts + TOPN * 16) * 4 * (32 / WAVES) <= 160 * 1024, "32 waves per CU must fit their stacks and images in LDS");
    __shared__ __attribute__((aligned(16))) int lds_raw[kStackInts + TOPN * 16];
    const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
    lds_int* col = (lds_int*)lds_raw + wave * kRows * kWave + lane;
    lds_int* const fifo = col + (LDS_N + 1) * kWave;
    lds_int* image = (lds_int*)lds_raw + kStackInts;
    const int root = stage_top_image<TOPN, kWave * WAVES>(nodes, (const int4*)top_image, image, (lds_int*)lds_raw, ctl, max_id)
        ? kLdsTag : 1;
    if (root != 1 && threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(&ctl->stats[6], 1ull);
    const int stripe = blockIdx.x % kStripes, stripe_waves = (gridDim.x / kStripes) * WAVES;
    int* counter = tickets + stripe * kCounterStride;
    const auto ray_of = [&](int t) { return ((t / kGroupRays) * kStripes + stripe) * kGroupRays + t % kGroupRays; };
    lds_int* const sp_limit = col + LDS_N * kWave;
    const Bases base = make_bases(nodes, tris);
    unsigned long long* const rounds = STATS ? &ctl->stats[3] : nullptr;
    int t = stripe_rank(wave) * kWave;
    bool coherent = MODE != 2;
    if (MODE == 0 && ray_of(t) < n) {
        const int r = ray_of(t + lane);
        const float4* p = reinterpret_cast<const float4*>(rays + (r < n ? r : ray_of(t)));
        const float4 o = p[0], d = p[1];
        if (grid_w < 0) grid_w = detect_ray_grid(rays, n);
        coherent = wave_rays_coherent(o.x, o.y, o.z, d.x, d.y, d.z, r < n);
    }
    if (!coherent && threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(&ctl->stats[5], 1ull);
    grid_w = __builtin_amdgcn_readfirstlane(grid_w > 0 && (grid_w & 7) == 0 ? grid_w : 0);
    const int tiled_rays = tiled_ray_count(grid_w, n);
    if (grid_w > 0 && threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(&ctl->stats[2], (unsigned long long)grid_w);
    if (MODE == 1 || (MODE == 0 && coherent)) {
        for (;;) {
            int first_ray = ray_of(t);
            if (first_ray >= n) break;
            int r = ray_of(t + lane);
            if (first_ray < tiled_rays) r = tile_ray(first_ray, lane, grid_w);
            Lane L = start_lane(rays, hits, r < n ? r : -1, first_ray, col);
            if (L.top != 0) L.top = root;
            int pend = 0;
            for (;;) {
                const bool stepping = L.top != 0 && !(L.top < 0 && pend == PEND);
                const unsigned long long step_mask = __ballot(stepping), pend_mask = __ballot(pend != 0);
                if ((step_mask | pend_mask) == 0ull) break;
                if (__popcll(pend_mask) >= DRAIN || step_mask == 0ull) { defer_drain<ANY, PEND>(L, pend, base, hits, fifo, rounds);
                    continue; }
                if (stepping) defer_step<LDS_N, PEND, WAVES>(L, pend, base, sp_limit, fifo, ctl, image, spill);
            }
            int t_next = 0;
            if (lane == 0) t_next = atomicAdd(counter, kWave);
            t = stripe_waves * kWave + __builtin_amdgcn_readfirstlane(t_next);
        }
    } else if (MODE != 1) {
        const auto ray_at = [&](int pos) { return pos < tiled_rays ? tile_ray_at(pos, grid_w) : pos; };
        Lane L;
        int pend = 0;
        {
            const int r = ray_at(ray_of(t + lane));
            L = start_lane(rays, hits, r < n ? r : -1, 0, col);
            if (L.top != 0) L.top = root;
        }
        bool more = true;
        for (;;) {
            const unsigned long long live = __ballot(L.top != 0 || pend != 0);
            if (more && __popcll(live) <= kWave - REFILL) {
                const int want = kWave - __popcll(live);
                int first = 0;
                if (lane == 0) first = atomicAdd(counter, want);
                first = stripe_waves * kWave + __builtin_amdgcn_readfirstlane(first);
                more = ray_of(first) < n;
                if (L.top == 0 && pend == 0) {
                    const int pos = ray_of(first + __popcll(~live & ((1ull << lane) - 1ull)));
                    if (pos < n) {
                        const int rr = ray_at(pos);
                        L = start_lane(rays, hits, rr, rr, col);
                        L.top = root;
                    }
                }
                continue;
            }
            if (live == 0) break;
            const bool stepping = L.top != 0 && !(L.top < 0 && pend == PEND);
            const unsigned long long step_mask = __ballot(stepping), pend_mask = __ballot(pend != 0);
            if (__popcll(pend_mask) >= DRAIN || step_mask == 0ull) { defer_drain<ANY, PEND>(L, pend, base, hits, fifo, rounds); continue; }
            if (stepping) defer_step<LDS_N, PEND, WAVES>(L, pend, base, sp_limit, fifo, ctl, image, spill);
        }
    }
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

template <bool ANY, int LDS_N, int PEND, int TOPN, int WAVES, int REFILL, int DRAIN, int MODE = 0,
    bool STATS = false> void L_defer(LAUNCH_ARGS) {
    const int max_id = top_kernel_ids(nodes, n);
    if (max_id == 0) { L_single<ANY, 16, 32>(s, nodes, tris, rays, hits, n, stream); return; }
    ensure_deep_list(s, n);
    ensure_top_buffers(s);
    const int groups = spill_checked(((s.num_cus * (32 / WAVES) + kStripes - 1) / kStripes) * kStripes, WAVES);
    ensure_spill(s, groups * WAVES);
    s.top_image_nodes = nullptr; s.order_rays = 0;
    hipLaunchKernelGGL((k_bvh2_top_defer<ANY, LDS_N, PEND, TOPN, WAVES, REFILL, DRAIN, MODE, STATS>), dim3(groups), dim3(kWave * WAVES), 0,
        stream, nodes, tris, rays, hits, n, s.ctl(), s.deep_list,
                       s.top_image, s.tickets, max_id, s.spill, g_ray_grid);
}

// -----------------------------------------------------------------------------------------------------------------------------------------
// Triangle turns WITH deferral (round 6, second form; LAB_NOTES 12.1's last paragraph put to the test).  No rounds of its own: a lane keeps
// ONE pending leaf in a register (`pleaf`: ~first triangle, 0 = none) and walks on; in every P-th iteration (a "turn") -- and whenever no
// lane of the wave can take a node step -- the lanes that hold a pending leaf test ONE triangle of it INSTEAD of their node step, in the
// same iteration and behind the same wait as the other lanes' node steps.  A triangle test costs one lane slot, as in the shipped loop; the
// triangle half of the step is issued in 1 / P of the iterations.  Per ray: the order of its triangle tests is the reference's; between
// noting a leaf and testing it the ray walks on against a stale tmax (at most P - 1 iterations).  Any-hit records are bit-identical;
// closest-hit rays may report another triangle at the same distance (see the head of this file).  Only the refill loop differs from
// k_bvh2_top_auto: waves whose rays share an origin or a direction run the shipped chunk loop.
// -----------------------------------------------------------------------------------------------------------------------------------------
template <bool ANY, int LDS_N, int WPG>
__device__ __forceinline__ void turn_step(Lane& L, int& pleaf, bool turn, const Bases& base, Hit1* __restrict__ hits, lds_int* sp_limit,
    Ctl* ctl,
                                          lds_int* image, int* __restrict__ spill) {
    const bool do_tri = turn && pleaf != 0, is_node = !do_tri && L.top > 0, in_image = is_node && L.top >= kLdsTag;
    vf4 q0, q1, q2;
    vi2 ch;
    int popped;
    const unsigned idx = (unsigned)(do_tri ? ~pleaf : L.top), stride = do_tri ? (unsigned)sizeof(Tri1) : (unsigned)sizeof(Node2);
    const gptr addr = (do_tri ? base.tri : base.node) + (size_t)idx * stride;
    joint_fetch3(q0, q1, q2, ch, popped, do_tri || (is_node && !in_image), in_image, (unsigned)(size_t)image + (unsigned)(L.top - kLdsTag),
        addr,
                 addr + (do_tri ? 40u : 48u), L.sp);
    if (is_node) {
        float te0, te1;
        const bool h0 = slab_canonical(L.ray, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, te0) && ch.x != 0;
        const bool h1 = slab_canonical(L.ray, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, te1) && ch.y != 0;
        const bool c0first = te0 < te1, both = h0 && h1, any = h0 || h1;
        const int first = both ? (c0first ? ch.x : ch.y) : (h0 ? ch.x : ch.y), second = c0first ? ch.y : ch.x;
        const bool note = any && first < 0 && pleaf == 0;                           // the next child is a leaf and the register is free
        pleaf = note ? first : pleaf;
        const bool push = both && !note, pop = !any || (note && !both);
        L.sp[kWave] = second;                                                       // (only kept when `push`)
        L.top = pop ? popped : (note ? second : first);
        L.sp += push ? kWave : (pop ? -kWave : 0);
        if (push && L.sp >= sp_limit) stack_spill<LDS_N>(L.sp, L.top, sp_limit, spill, WPG, &ctl->err, &ctl->stats[7]);
    } else if (do_tri) {
        const int prim_id = __float_as_int(q2.w);
        const float nx = cross_x(q1.x, q1.y, q1.z, q2.x, q2.y, q2.z), ny = cross_y(q1.x, q1.y, q1.z, q2.x, q2.y, q2.z);
        const float nz = cross_z(q1.x, q1.y, q1.z, q2.x, q2.y, q2.z);
        float t, u, v;
        bool found = false;
        if (intersect_tri(L.ray, q0.x, q0.y, q0.z, q1.x, q1.y, q1.z, q2.x, q2.y, q2.z, nx, ny, nz, t, u, v)) {
            store_hit(hits, L.ray_id, prim_id & 0x7FFFFFFF, t, u, v);
            L.ray.tmax = t; found = true;
        }
        if (ANY && found) { L.top = 0; pleaf = 0; }                                 // the ray is done: what is left of its stack is dropped
        else pleaf = prim_id < 0 ? 0 : pleaf - 1;                                   // sentinel: the leaf is done; else ~(j + 1)
    // a leaf off the stack and a free register: note it, pop
    } else {
        pleaf = L.top;
        L.top = popped;
        L.sp -= kWave;
    }
    if (L.top >= kSpillMark) stack_reload<LDS_N>(L.sp, L.top, sp_limit, spill, WPG);
}

template <bool ANY, int LDS_N, int TOPN, int WAVES, int REFILL, int P>
__global__ __launch_bounds__(kWave * WAVES) __attribute__((amdgpu_waves_per_eu(8, 8))) void
    k_bvh2_top_turns(const Node2* __restrict__ nodes, const Tri1* __restrict__ tris, const Ray1* __restrict__ rays,
        Hit1* __restrict__ hits, int n,
                     Ctl* ctl, int* __restrict__ deep_list, int4* __restrict__ top_image, int* __restrict__ tickets, int max_id, int* spill,
                     int grid_w) {
    constexpr int kStackInts = WAVES * (LDS_N + 1) * kWave, kGroupRays = 32 * kWave;
    static_assert((kStackInts + TOPN * 16) * 4 * (32 / WAVES) <= 160 * 1024, "32 waves per CU must fit their stacks and images in LDS");
    __shared__ __attribute__((aligned(16))) int lds_raw[kStackInts + TOPN * 16];
    const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
    lds_int* col = (lds_int*)lds_raw + wave * (LDS_N + 1) * kWave + lane;
    lds_int* image = (lds_int*)lds_raw + kStackInts;
    const int root = stage_top_image<TOPN, kWave * WAVES>(nodes, (const int4*)top_image, image, (lds_int*)lds_raw, ctl, max_id)
        ? kLdsTag : 1;
    if (root != 1 && threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(&ctl->stats[6], 1ull);
    const int stripe = blockIdx.x % kStripes, stripe_waves = (gridDim.x / kStripes) * WAVES;
    int* counter = tickets + stripe * kCounterStride;
    const auto ray_of = [&](int t) { return ((t / kGroupRays) * kStripes + stripe) * kGroupRays + t % kGroupRays; };
    lds_int* const sp_limit = col + LDS_N * kWave;
    const Bases base = make_bases(nodes, tris);
    int t = stripe_rank(wave) * kWave;
    bool coherent = true;
    if (ray_of(t) < n) {
        const int r = ray_of(t + lane);
        const float4* p = reinterpret_cast<const float4*>(rays + (r < n ? r : ray_of(t)));
        const float4 o = p[0], d = p[1];
        if (grid_w < 0) grid_w = detect_ray_grid(rays, n);
        coherent = wave_rays_coherent(o.x, o.y, o.z, d.x, d.y, d.z, r < n);
    }
    if (!coherent && threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(&ctl->stats[5], 1ull);
    grid_w = __builtin_amdgcn_readfirstlane(grid_w > 0 && (grid_w & 7) == 0 ? grid_w : 0);
    const int tiled_rays = tiled_ray_count(grid_w, n);
    if (coherent) {
        for (;;) {                                                           // the shipped chunk loop
            int first_ray = ray_of(t);
            if (first_ray >= n) break;
            int r = ray_of(t + lane);
            if (first_ray < tiled_rays) r = tile_ray(first_ray, lane, grid_w);
            Lane L = start_lane(rays, hits, r < n ? r : -1, first_ray, col);
            if (L.top != 0) L.top = root;
            while (__ballot(L.top != 0)) {
                if (L.top != 0) bvh2_step<ANY, false, true, false, false, false, LDS_N, WAVES>(L, base, hits, sp_limit, ctl, deep_list,
                    false, nullptr,
                                                                                              image, nullptr, spill);
            }
            int t_next = 0;
            if (lane == 0) t_next = atomicAdd(counter, kWave);
            t = stripe_waves * kWave + __builtin_amdgcn_readfirstlane(t_next);
        }
    } else {
        const auto ray_at = [&](int pos) { return pos < tiled_rays ? tile_ray_at(pos, grid_w) : pos; };
        Lane L;
        int pleaf = 0;
        {
            const int r = ray_at(ray_of(t + lane));
            L = start_lane(rays, hits, r < n ? r : -1, 0, col);
            if (L.top != 0) L.top = root;
        }
        bool more = true;
        for (int it = 0;; it++) {
            const unsigned long long live = __ballot(L.top != 0 || pleaf != 0);
            if (more && __popcll(live) <= kWave - REFILL) {
                const int want = kWave - __popcll(live);
                int first = 0;
                if (lane == 0) first = atomicAdd(counter, want);
                first = stripe_waves * kWave + __builtin_amdgcn_readfirstlane(first);
                more = ray_of(first) < n;
                if (L.top == 0 && pleaf == 0) {
                    const int pos = ray_of(first + __popcll(~live & ((1ull << lane) - 1ull)));
                    if (pos < n) {
                        const int rr = ray_at(pos);
                        L = start_lane(rays, hits, rr, rr, col);
                        L.top = root;
                    }
                }
                continue;
            }
            if (live == 0) break;
            const bool can_walk = L.top > 0 || (L.top < 0 && pleaf == 0);
            const bool turn = it % P == P - 1 || __ballot(can_walk) == 0ull;
            if (can_walk || (turn && pleaf != 0)) turn_step<ANY, LDS_N, WAVES>(L, pleaf, turn, base, hits, sp_limit, ctl, image, spill);
        }
    }
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

template <bool ANY, int LDS_N, int TOPN, int WAVES, int REFILL, int P> void L_turns(LAUNCH_ARGS) {
    const int max_id = top_kernel_ids(nodes, n);
    if (max_id == 0) { L_single<ANY, 16, 32>(s, nodes, tris, rays, hits, n, stream); return; }
    ensure_deep_list(s, n);
    ensure_top_buffers(s);
    const int groups = spill_checked(((s.num_cus * (32 / WAVES) + kStripes - 1) / kStripes) * kStripes, WAVES);
    ensure_spill(s, groups * WAVES);
    s.top_image_nodes = nullptr; s.order_rays = 0;
    hipLaunchKernelGGL((k_bvh2_top_turns<ANY, LDS_N, TOPN, WAVES, REFILL, P>), dim3(groups), dim3(kWave * WAVES), 0, stream, nodes, tris,
        rays, hits,
                       n, s.ctl(), s.deep_list, s.top_image, s.tickets, max_id, s.spill, g_ray_grid);
}

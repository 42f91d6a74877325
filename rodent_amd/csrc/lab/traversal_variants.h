// traversal_variants.h -- the BVH2 kernels that were measured on the way to the default one (k_bvh2_single in
// traversal.hip) and stay selectable for comparison (variant table in traversal.hip, scripts/sweep.py): the literal
// one-ray-per-lane mapping of the reference, the while-while family with its persistent / static-stride / instrumented
// forms, and the per-step ballot scheduler.  All keep the reference's per-ray visit order and are bit-identical to the
// default.  Included by traversal.hip inside its anonymous namespace, after Ctl, k_bvh2_finish, DeviceState,
// LAUNCH_ARGS, ensure_deep_list and traversal_wide.h -- in the LAB build only (-DRODENT_HIP_LAB, librodent_hip_lab.so).
#pragma once

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



// ---------------------------------------------------------------------------------------------
// BVH2 / Tri1, "lane": literal one-ray-per-lane mapping of the reference kernel.
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
// BVH2 / Tri1, "while-while" variants.  Per ray, the sequence of node steps and leaf tests is
// exactly the one of variant 0 (so results stay bit-identical); what changes is how the
// wavefront schedules it: all lanes first descend until each holds a leaf (or is done), then
// all lanes intersect their leaves together.  In variant 0 a single lane reaching a leaf makes
// the whole wave execute the triangle code (measured: 24 % of VALU lanes active, one triangle
// iteration per node iteration per wave although a ray tests ~3.5 triangles per ~36 nodes).
// NODE_EXIT > 0: leave the descent phase early once fewer than NODE_EXIT lanes are still
// descending (they resume after the leaf phase).
// ---------------------------------------------------------------------------------------------
template <bool ANY, int LDS_N, int NODE_EXIT>
__global__ __launch_bounds__(kWave) void k_bvh2_ww(const Node2* __restrict__ nodes, const Tri1* __restrict__ tris,
                                                    const Ray1* __restrict__ rays, Hit1* __restrict__ hits, int n, int* err) {
    __shared__ int lds[LDS_N * kWave];
    const int i = blockIdx.x * kWave + threadIdx.x;
    if (i >= n) return;
    RayX ray = load_ray(rays, i);
    HitAcc hit{-1, ray.tmax, 0.0f, 0.0f};
    LaneStack<LDS_N> st; st.lds = lds + threadIdx.x; st.err = err;
    int ptr = 0, top = 1; st.put(0, 0);
    while (top != 0) {
        while (top > 0) {
            top = node2_step(nodes, top, ray, st, ptr);
            if (NODE_EXIT > 0 && __popcll(__ballot(top > 0)) < NODE_EXIT) break;
        }
        while (top < 0) {
            const int first = ~top; top = st.get(ptr); ptr--;
            if (leaf_tri1<ANY>(tris, first, ray, hit)) { top = 0; break; }
        }
    }
    store_hit(hits, i, hit.id, hit.t, hit.u, hit.v);
}


// ---------------------------------------------------------------------------------------------
// BVH2 / Tri1, "fast" family.  Same per-ray visit order as variant 0 (bit-identical results),
// engineered for the CDNA4 execution model:
//  * LDS-only stack window accessed through address_space(3) pointers (plain ds_read/ds_write; the
//    generic-pointer LDS+scratch stack of the variants above compiles to flat loads); deeper rays
//    go to the deep list and k_bvh2_finish (see above);
//  * branch-free node step: the would-be popped entry is read from LDS while the four 16-byte
//    node loads are in flight, the far child is always written to the free slot above the top,
//    and the new top / stack pointer are selected arithmetically (no divergent sub-branches);
//  * while-while scheduling with an early exit from the descent phase (NODE_EXIT);
//  * optional persistent wavefronts (PERSIST): lanes that finish pull new rays from a
//    wave-local pool that is refilled CHUNK rays at a time from one global counter (first chunk
//    static = block index, so the launch starts without atomics; a single counter saturates at
//    ~88 dequeues/us on this chip).  The last wave to leave resets the counters, so no memset
//    sits between launches.
// ---------------------------------------------------------------------------------------------

// Order in which 64-ray chunks are handed to wavefronts: ticket k -> chunk (k * mul) % count, mul
// coprime to count.  Rays arrive in scan-line order, so expensive image regions (foliage at the
// bottom of the frame) would otherwise all be started last and stretch the drain phase of the
// launch; a large stride interleaves cheap and expensive regions in time.  mul = 1 keeps ray order.
struct ChunkPerm {
    int count, mul;
    __device__ __forceinline__ int map(int k) const { return (int)(((long long)k * mul) % count); }
};


template <bool ANY, int LDS_N, int NODE_EXIT, bool PERSIST, int REFILL_IDLE, int CHUNK, bool STATS = false, int XCD = 0, bool TRACE = false,
          bool STATIC = false, bool WIDE = false>
__global__ __launch_bounds__(kWave) void k_bvh2_fast(const Node2* __restrict__ nodes, const Tri1* __restrict__ tris,
                                                      const Ray1* __restrict__ rays, Hit1* __restrict__ hits, int n,
                                                      Ctl* ctl, int* __restrict__ deep_list) {
    __shared__ int lds_raw[(LDS_N + 1) * kWave];
    lds_int* col = (lds_int*)lds_raw + threadIdx.x;
    RayX ray; HitAcc hit{-1, 0.0f, 0.0f, 0.0f};
    int ray_id = -1, top = 0, ptr = 0;
    // wave-uniform pool of ray indices [pool_next, pool_end)
    // XCD > 0: XCD-aware block -> chunk mapping.  Workgroups are dispatched round-robin over the 8 XCDs
    // (block b runs on XCD b % 8; observed, used for speed only) and each XCD has its own 4 MB L2.  Chunks
    // are taken in groups of XCD consecutive 64-ray chunks (XCD = 16 is one 1024-pixel scan line) and group j
    // goes to XCD j % 8, so an XCD works on whole image bands instead of every 8th 64-pixel strip while the
    // bands of one XCD stay spread over the frame (a contiguous eighth per XCD was measured 37 % slower: the
    // expensive bottom of the frame then lands on one XCD).
    const unsigned long long t_start = TRACE ? __builtin_amdgcn_s_memrealtime() : 0ull;
    // STATIC (with PERSIST): no queue at all -- the grid is one wave per resident slot and wave b owns the
    // tickets b, b + grid, b + 2 grid, ...; every wave starts at t = 0 and pairs chunks from distant parts
    // of the stream, so no expensive chunk is started late (the drain of a plain launch is set by the
    // expensive waves of the LAST dispatch round).
    const int total_chunks = (n + kWave - 1) / kWave;
    auto chunk_of = [&](int ticket) {
        if (XCD > 0 && (!PERSIST || STATIC)) {
            const int span = 8 * XCD, full = (total_chunks / span) * span;        // region where the mapping is a bijection
            if (ticket < full) {
                const int x = ticket % 8, l = ticket / 8;
                return ((l / XCD) * 8 + x) * XCD + l % XCD;
            }
        }
        return ticket;
    };
    int ticket = blockIdx.x;
    const int first_chunk = chunk_of(ticket);
    int pool_next = first_chunk * (PERSIST ? CHUNK : kWave);
    int pool_end = min(n, pool_next + (PERSIST ? CHUNK : kWave));
    bool exhausted = !PERSIST || (STATIC && ticket + (int)gridDim.x >= total_chunks);
    // STATS build only: [0] descent iterations, [1] active lanes in them, [2] leaf iterations, [3] active lanes,
    // [4] refills, [5] lanes refilled, [6] outer iterations (wave-uniform counts, lane 0 publishes)
    unsigned c_n = 0, c_l = 0;                     // per-lane participations
    __shared__ unsigned wst[8];                    // wave-level counts, bumped by an elected lane
    if ((STATS || TRACE) && threadIdx.x < 8) wst[threadIdx.x] = 0;
#define WAVE_COUNT(slot, amount) do { const unsigned long long m_ = __ballot(true); \
    if ((int)threadIdx.x == __ffsll((long long)m_) - 1) atomicAdd(&wst[slot], (unsigned)(amount)); } while (0)

    for (;;) {
        if (STATS) WAVE_COUNT(6, 1);
        const bool idle = top == 0;
        if (idle && ray_id >= 0) { store_hit(hits, ray_id, hit.id, hit.t, hit.u, hit.v); ray_id = -1; }
        const unsigned long long idle_mask = __ballot(idle);
        const int num_idle = __popcll(idle_mask);
        const bool pool_empty = pool_next >= pool_end;
        if (num_idle == kWave && pool_empty && exhausted) break;
        if (num_idle >= REFILL_IDLE || num_idle == kWave) {
            if (PERSIST && STATIC && pool_empty && !exhausted) {
                ticket += gridDim.x;
                if (ticket < total_chunks) { pool_next = chunk_of(ticket) * kWave; pool_end = min(n, pool_next + kWave); }
                if (ticket + (int)gridDim.x >= total_chunks) exhausted = true;
            } else if (PERSIST && pool_empty && !exhausted) {
                int base = 0;
                if (threadIdx.x == 0) base = atomicAdd(&ctl->counter, CHUNK);
                base = __builtin_amdgcn_readfirstlane(base) + gridDim.x * CHUNK;
                pool_next = base; pool_end = min(n, base + CHUNK);
                if (base + CHUNK >= n) exhausted = true;
            }
            const int avail = pool_end - pool_next;
            if (avail > 0) {
                const int r = __popcll(idle_mask & ((1ull << threadIdx.x) - 1ull));
                if (idle && r < avail) {
                    ray_id = pool_next + r; ray = load_ray(rays, ray_id);
                    hit.id = -1; hit.t = ray.tmax; hit.u = 0.0f; hit.v = 0.0f;
                    ptr = 0; top = 1; col[0] = 0;
                }
                if (STATS) { WAVE_COUNT(4, 1); WAVE_COUNT(5, min(num_idle, avail)); }
                pool_next += min(num_idle, avail);
            }
        }
        // ---- descent phase ----
        while (top > 0) {
            if (STATS) { c_n++; WAVE_COUNT(0, 1); }
            if (TRACE) WAVE_COUNT(0, 1);
            const float4* p = WIDE ? reinterpret_cast<const float4*>(nodes + (top - 1))
                                   : reinterpret_cast<const float4*>(reinterpret_cast<const char*>(nodes - 1) + ((unsigned)top << 6));
            const float4 b0 = p[0], b1 = p[1], b2 = p[2];
            const int2 ch = *reinterpret_cast<const int2*>(p + 3);                 // child ids; the last 8 bytes of a Node2 are padding
            const int popped = col[ptr * kWave];
            float te0, te1;
            const bool h0 = slab(ray, b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, te0) && ch.x != 0;
            const bool h1 = slab(ray, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, te1) && ch.y != 0;
            const bool c0first = te0 < te1;
            const bool both = h0 && h1;
            col[(ptr + 1) * kWave] = c0first ? ch.y : ch.x;                     // far child -> free slot above the top
            top = both ? (c0first ? ch.x : ch.y) : (h0 ? ch.x : (h1 ? ch.y : popped));
            ptr += (both ? 1 : 0) - ((h0 || h1) ? 0 : 1);
            if (ptr >= LDS_N) {                                                 // deeper than the LDS window: hand over
                deep_list[atomicAdd(&ctl->deep_count, 1)] = ray_id;
                ray_id = -1; top = 0;
            }
            if (NODE_EXIT > 0 && __popcll(__ballot(top > 0)) < NODE_EXIT) break;
        }
        // ---- leaf phase ----
        while (top < 0) {
            if (STATS) { c_l++; WAVE_COUNT(2, 1); }
            if (TRACE) WAVE_COUNT(2, 1);
            const int first = ~top; top = col[ptr * kWave]; ptr--;
            if (leaf_tri1<ANY, WIDE>(tris, first, ray, hit)) { top = 0; break; }
        }
    }
    if (STATS) {
        for (int o = 32; o > 0; o >>= 1) { c_n += __shfl_xor((int)c_n, o); c_l += __shfl_xor((int)c_l, o); }
        if (threadIdx.x == 0) {
            for (int k = 0; k < 7; k++) atomicAdd(&ctl->stats[k],
                (unsigned long long)(k == 1 ? c_n : (k == 3 ? c_l : atomicAdd(&wst[k], 0u))));
        }
    }
#undef WAVE_COUNT
    if (TRACE && threadIdx.x == 0 && ctl->trace && blockIdx.x < 16384) {
        unsigned long long* tr = ctl->trace + 4 * (size_t)blockIdx.x;
        tr[0] = t_start; tr[1] = __builtin_amdgcn_s_memrealtime();
        tr[2] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4))
            | ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32);
        tr[3] = ((unsigned long long)atomicAdd(&wst[0], 0u) << 32) | atomicAdd(&wst[2], 0u);
    }
}

// ---------------------------------------------------------------------------------------------
// BVH2 / Tri1, "sched" family: wave-level step scheduling.  Every lane is a small state machine
//   cur > 0  : next action is a node step on node cur-1
//   cur < 0  : next action is ONE triangle test on tris[~cur] (a leaf is walked by decrementing cur)
//   cur == 0 : idle (ray finished; the result is stored when the lane is refilled or at exit)
// and each trip of the wave loop executes exactly one kind of step, chosen by ballot: refill if
// enough lanes are idle and rays are left, else the kind (node / triangle) more lanes are waiting
// for.  A lane still performs its own node steps and triangle tests in the reference's order, so
// results remain bit-identical to variant 0; only the interleaving across lanes changes.
// Measured motivation (instrumented fast-lds16, 64 rays per wave): descent iterations run with
// 57 % (primary) / 35 % (random) of the lanes, leaf visits with 20 % / 10 %.
// ---------------------------------------------------------------------------------------------
template <bool ANY, int LDS_N, bool PERSIST, int REFILL_IDLE, int CHUNK, int TRI_BIAS, bool STATS = false, bool TRACE = false>
__global__ __launch_bounds__(kWave) void k_bvh2_sched(const Node2* __restrict__ nodes, const Tri1* __restrict__ tris,
                                                       const Ray1* __restrict__ rays, Hit1* __restrict__ hits, int n,
                                                       Ctl* ctl, int* __restrict__ deep_list, ChunkPerm perm) {
    __shared__ int lds_raw[(LDS_N + 1) * kWave];
    lds_int* col = (lds_int*)lds_raw + threadIdx.x;
    RayX ray; HitAcc hit{-1, 0.0f, 0.0f, 0.0f};
    int ray_id = -1, cur = 0, ptr = 0;
    constexpr int kChunk = PERSIST ? CHUNK : kWave;
    int pool_next = perm.map(blockIdx.x) * kChunk;
    int pool_end = min(n, pool_next + kChunk);
    bool exhausted = !PERSIST;
    unsigned c_n = 0, c_l = 0;
    __shared__ unsigned wst[8];
    if ((STATS || TRACE) && threadIdx.x < 8) wst[threadIdx.x] = 0;
    const unsigned long long t_start = (STATS || TRACE) ? __builtin_amdgcn_s_memrealtime() : 0ull;
    unsigned traced_rays = 0;
#define WAVE_COUNT(slot, amount) do { const unsigned long long m_ = __ballot(true); \
    if ((int)threadIdx.x == __ffsll((long long)m_) - 1) atomicAdd(&wst[slot], (unsigned)(amount)); } while (0)

    for (;;) {
        if (STATS) WAVE_COUNT(6, 1);
        const unsigned long long node_mask = __ballot(cur > 0), tri_mask = __ballot(cur < 0);
        const int nn = __popcll(node_mask), nt = __popcll(tri_mask), ni = kWave - nn - nt;
        const bool rays_left = pool_next < pool_end || !exhausted;
        if (ni == kWave && !rays_left) break;
        if (rays_left && (ni >= REFILL_IDLE || ni == kWave)) {
            // ---- refill step ----
            const bool idle = cur == 0;
            if (idle && ray_id >= 0) { store_hit(hits, ray_id, hit.id, hit.t, hit.u, hit.v); ray_id = -1; }
            if (PERSIST && pool_next >= pool_end) {
                int base = 0;
                if (threadIdx.x == 0) base = atomicAdd(&ctl->counter, 1);
                const int c = __builtin_amdgcn_readfirstlane(base) + (int)gridDim.x;     // chunk ticket
                if (c >= perm.count) { exhausted = true; pool_next = pool_end = 0; }
                else { pool_next = perm.map(c) * CHUNK; pool_end = min(n, pool_next + CHUNK); }
            }
            const int avail = pool_end - pool_next;
            if (avail > 0) {
                const unsigned long long idle_mask = ~(node_mask | tri_mask);
                const int r = __popcll(idle_mask & ((1ull << threadIdx.x) - 1ull));
                if (idle && r < avail) {
                    ray_id = pool_next + r; ray = load_ray(rays, ray_id);
                    hit.id = -1; hit.t = ray.tmax; hit.u = 0.0f; hit.v = 0.0f;
                    ptr = 0; cur = 1; col[0] = 0;
                }
                if (STATS) { WAVE_COUNT(4, 1); WAVE_COUNT(5, min(ni, avail)); }
                if (TRACE) traced_rays += min(ni, avail);
                pool_next += min(ni, avail);
            }
            continue;
        }
        if (nn * 4 >= nt * TRI_BIAS) {
            // ---- node step (branch-free, see k_bvh2_fast) ----
            if (cur > 0) {
                if (STATS) { c_n++; WAVE_COUNT(0, 1); }
                const float4* p = reinterpret_cast<const float4*>(nodes + (cur - 1));
                const float4 b0 = p[0], b1 = p[1], b2 = p[2];
                const int4 ch = *reinterpret_cast<const int4*>(p + 3);
                const int popped = col[ptr * kWave];
                float te0, te1;
                const bool h0 = slab(ray, b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, te0) && ch.x != 0;
                const bool h1 = slab(ray, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, te1) && ch.y != 0;
                const bool c0first = te0 < te1;
                const bool both = h0 && h1;
                col[(ptr + 1) * kWave] = c0first ? ch.y : ch.x;
                cur = both ? (c0first ? ch.x : ch.y) : (h0 ? ch.x : (h1 ? ch.y : popped));
                ptr += (both ? 1 : 0) - ((h0 || h1) ? 0 : 1);
                if (ptr >= LDS_N) {
                    deep_list[atomicAdd(&ctl->deep_count, 1)] = ray_id;
                    ray_id = -1; cur = 0;
                }
            }
        } else {
            // ---- triangle step: one Tri1 per waiting lane (mapping_gpu.impala:156-174) ----
            if (cur < 0) {
                if (STATS) { c_l++; WAVE_COUNT(2, 1); }
                const float4* p = reinterpret_cast<const float4*>(tris + (~cur));
                const float4 a = p[0], b = p[1], c = p[2];
                const int popped = col[ptr * kWave];
                const int prim_id = __float_as_int(c.w);
                const float nx = cross_x(b.x, b.y, b.z, c.x, c.y, c.z);
                const float ny = cross_y(b.x, b.y, b.z, c.x, c.y, c.z);
                const float nz = cross_z(b.x, b.y, b.z, c.x, c.y, c.z);
                float t, u, v;
                const bool h = intersect_tri(ray, a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z, nx, ny, nz, t, u, v);
                if (h) { hit.id = prim_id & 0x7FFFFFFF; hit.t = t; hit.u = u; hit.v = v; ray.tmax = t; }
                const bool last = prim_id < 0;
                cur = (ANY && h) ? 0 : (last ? popped : cur - 1);
                ptr -= last ? 1 : 0;
            }
        }
    }
    if (cur == 0 && ray_id >= 0) store_hit(hits, ray_id, hit.id, hit.t, hit.u, hit.v);
    if (STATS) {
        for (int o = 32; o > 0; o >>= 1) { c_n += __shfl_xor((int)c_n, o); c_l += __shfl_xor((int)c_l, o); }
        if (threadIdx.x == 0) {
            for (int k = 0; k < 7; k++) atomicAdd(&ctl->stats[k],
                (unsigned long long)(k == 1 ? c_n : (k == 3 ? c_l : atomicAdd(&wst[k], 0u))));
            if (ctl->trace && blockIdx.x < 16384) {       // per-wave timeline: start, end (100 MHz ticks), hw id, xcc id, work
                unsigned long long* tr = ctl->trace + 4 * (size_t)blockIdx.x;
                tr[0] = t_start; tr[1] = __builtin_amdgcn_s_memrealtime();
                tr[2] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4))
                    | ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32);
                tr[3] = ((unsigned long long)atomicAdd(&wst[5], 0u) << 32) | atomicAdd(&wst[6], 0u);
            }
        }
    }
#undef WAVE_COUNT
    if (TRACE && threadIdx.x == 0 && ctl->trace && blockIdx.x < 16384) {
        unsigned long long* tr = ctl->trace + 4 * (size_t)blockIdx.x;
        tr[0] = t_start; tr[1] = __builtin_amdgcn_s_memrealtime();
        tr[2] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4))
            | ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32);
        tr[3] = (unsigned long long)traced_rays << 32;
    }
}



int persistent_waves_per_cu() {
    static const int v = [] { const char* e = getenv("RODENT_HIP_PERSISTENT_WAVES_PER_CU"); return e ? atoi(e) : 16; }();
    return v;
}

int static_waves_per_cu() {
    static const int v = [] { const char* e = getenv("RODENT_HIP_STATIC_WAVES_PER_CU"); return e ? atoi(e) : 32; }();
    return v;
}

// True if node or triangle byte offsets may not fit 32 bits: the arrays' allocations are asked for their extent
// (the C ABI of the reference passes no sizes, traversal.impala:1-9).  Unknown pointers count as wide.
bool needs_wide_offsets(const void* nodes, const void* tris) {
    bool wide = false;
    for (const void* p : {nodes, tris}) {
        hipDeviceptr_t base = nullptr; size_t size = 0;
        if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)p) != hipSuccess) { (void)hipGetLastError(); wide = true; continue; }
        const size_t extent = (size_t)((const char*)base + size - (const char*)p);
        if (extent >= (1ull << 32)) wide = true;
    }
    return wide;
}

template <bool ANY, int LDS_N> void L_lane(LAUNCH_ARGS) {
    hipLaunchKernelGGL((k_bvh2_lane<ANY, LDS_N>), dim3(blocks_for(n)), dim3(kWave), 0, stream, nodes, tris, rays, hits, n,
        s.host_page + kHostErr);
}
template <bool ANY, int LDS_N, int NE> void L_ww(LAUNCH_ARGS) {
    hipLaunchKernelGGL((k_bvh2_ww<ANY, LDS_N, NE>), dim3(blocks_for(n)), dim3(kWave), 0, stream, nodes, tris, rays, hits, n,
        s.host_page + kHostErr);
}
template <bool ANY, int LDS_N, int NE, bool P, int RI, int CH, bool ST = false, int XCD = 0, bool TR = false,
    bool SC = false> void L_fast(LAUNCH_ARGS) {
    ensure_deep_list(s, n);
    const int per_block = P ? CH : kWave;
    int grid = (n + per_block - 1) / per_block;
    if (P) grid = std::min(grid, s.num_cus * (SC ? static_waves_per_cu() : persistent_waves_per_cu()));
    if (needs_wide_offsets(nodes, tris))
        hipLaunchKernelGGL((k_bvh2_fast<ANY, LDS_N, NE, P, RI, CH, ST, XCD, TR, SC, true>), dim3(grid), dim3(kWave), 0, stream, nodes,
            tris, rays, hits, n, s.ctl(), s.deep_list);
    else
        hipLaunchKernelGGL((k_bvh2_fast<ANY, LDS_N, NE, P, RI, CH, ST, XCD, TR, SC, false>), dim3(grid), dim3(kWave), 0, stream, nodes,
            tris, rays, hits, n, s.ctl(), s.deep_list);
    hipLaunchKernelGGL((k_bvh2_finish<ANY>), dim3(1), dim3(kWave), 0, stream, nodes, tris, rays, hits, s.ctl(), s.deep_list, s.deep_stack,
        (int*)nullptr);
}

int coprime_stride(int count) {
    if (count < 8) return 1;
    auto gcd = [](int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; };
    int m = (int)(count * 0.6180339887) | 1;
    while (gcd(m, count) != 1) m += 2;
    return m % count;
}

template <bool ANY, int LDS_N, bool P, int RI, int CH, int TB, bool PERMUTE = false, bool ST = false,
    bool TR = false> void L_sched(LAUNCH_ARGS) {
    ensure_deep_list(s, n);
    const int per_block = P ? CH : kWave;
    const int chunks = (n + per_block - 1) / per_block;
    int grid = chunks;
    if (P) grid = std::min(grid, s.num_cus * persistent_waves_per_cu());
    const ChunkPerm perm{chunks, PERMUTE ? coprime_stride(chunks) : 1};
    hipLaunchKernelGGL((k_bvh2_sched<ANY, LDS_N, P, RI, CH, TB, ST, TR>), dim3(grid), dim3(kWave), 0, stream, nodes, tris, rays, hits, n,
        s.ctl(), s.deep_list, perm);
    hipLaunchKernelGGL((k_bvh2_finish<ANY>), dim3(1), dim3(kWave), 0, stream, nodes, tris, rays, hits, s.ctl(), s.deep_list, s.deep_stack,
        (int*)nullptr);
}


// ---------------------------------------------------------------------------------------------
// BVH4 / BVH8 + Tri4, "lane": the reference's general-arity loop, literally, one ray per lane (wide_ray_literal):
// pop-first loop, hit record in registers, LDS window + scratch spill, no XCD mapping.  BVH8: 118 VGPRs.
// ---------------------------------------------------------------------------------------------
template <bool ANY, int N, int LDS_N>
__global__ __launch_bounds__(kWave) void k_wide_lane(const char* __restrict__ nodes, const Tri4* __restrict__ tris,
                                                      const Ray1* __restrict__ rays, Hit1* __restrict__ hits, int n, int* err) {
    __shared__ int lds[LDS_N * kWave];
    const int i = blockIdx.x * kWave + threadIdx.x;
    if (i >= n) return;
    LaneStack<LDS_N> st; st.lds = lds + threadIdx.x; st.err = err;
    const HitAcc hit = wide_ray_literal<ANY, N>(nodes, tris, load_ray(rays, i), st);
    store_hit(hits, i, hit.id, hit.t, hit.u, hit.v);
}
template <bool ANY, int N, int LDS_N> void L_wide_lane(WIDE_LAUNCH_ARGS) {
    hipLaunchKernelGGL((k_wide_lane<ANY, N, LDS_N>), dim3(blocks_for(n)), dim3(kWave), 0, stream, (const char*)nodes, tris, rays, hits, n,
        s.host_page + kHostErr);
}


// ---------------------------------------------------------------------------------------------
// "top-partner" (round 3, lab): a STATELESS attempt at the chunk order the schedule history gets from the previous launch.
// The persistent LDS-image kernel (k_bvh2_top_persist) with another ticket -> chunk map and another way to draw:
//   * a stripe's first generation (one chunk per resident wave, tickets 0 .. W-1) takes the EVEN row of every pair of 16-chunk rows
//     the stripe owns, the second generation (tickets W .. 2W-1) the odd rows: ticket t and ticket t + W are vertical neighbours
//     in a scanline-ordered ray set (their costs correlate at 0.98 on the benchmark's primary rays);
//   * a first-generation chunk that still has >= HOT_LANES rays alive after HOT_ITER iterations marks its partner HOT (one atomicOr into
//     the stripe's 128-bit mask);
//   * a wave that has finished a chunk claims (atomicOr on the stripe's claimed mask) the lowest hot partner that is not taken,
//     or else the lowest partner that is not taken (= the default order).
// Only for launches of exactly 2 x resident-waves chunks (1 Mi rays on 256 CUs); anything else takes the default order.
// ---------------------------------------------------------------------------------------------
template <bool ANY, int LDS_N, int TOPN, int WAVES, int HOT_ITER, int HOT_LANES>
__global__ __launch_bounds__(kWave * WAVES) __attribute__((amdgpu_waves_per_eu(8, 8))) void
    k_bvh2_top_partner(const Node2* __restrict__ nodes, const Tri1* __restrict__ tris,
                                                                     const Ray1* __restrict__ rays, Hit1* __restrict__ hits, int n,
                                                                     Ctl* ctl, int* __restrict__ deep_list, int4* __restrict__ top_image,
                                                                         int* __restrict__ tickets, int max_id) {
    constexpr int kStackInts = WAVES * (LDS_N + 1) * kWave;
    __shared__ __attribute__((aligned(16))) int lds_raw[kStackInts + TOPN * 16];
    const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
    lds_int* col = (lds_int*)lds_raw + wave * (LDS_N + 1) * kWave + lane;
    lds_int* image = (lds_int*)lds_raw + kStackInts;
    const int root = stage_top_image<TOPN, kWave * WAVES>(nodes, (const int4*)top_image, image, (lds_int*)lds_raw, ctl, max_id)
        ? kLdsTag : 1;
    const int total_chunks = (n + kWave - 1) / kWave, stripe = blockIdx.x % kStripes, stripe_waves = (gridDim.x / kStripes) * WAVES;
    // [1..4] hot mask, [5..8] claimed mask (word 0: the default order's counter)
    unsigned* state = reinterpret_cast<unsigned*>(tickets + stripe * kCounterStride);
    const bool paired = total_chunks == 2 * stripe_waves * kStripes && stripe_waves == 128;
    lds_int* const sp_limit = col + LDS_N * kWave;
    const Bases base = make_bases(nodes, tris);
    int t = stripe_rank(wave);
    for (;;) {
        int chunk;
        if (paired) {
            if (t >= 2 * stripe_waves) break;
            const int u = t % stripe_waves;                                   // first generation: even row of the pair; second: odd row
            chunk = ((u / 16) * kStripes + stripe) * 32 + (t >= stripe_waves ? 16 : 0) + u % 16;
        } else {
            const int group_first = ((t / 32) * kStripes + stripe) * 32;
            if (group_first >= total_chunks) break;
            chunk = group_first + t % 32;
        }
        if (chunk < total_chunks) {
            const int first_ray = chunk * kWave, lane_ray = first_ray + lane;
            Lane L = start_lane(rays, hits, lane_ray < n ? lane_ray : -1, first_ray, col);
            if (L.top != 0) L.top = root;
            int iterations = 0;
            for (;;) {
                const unsigned long long live = __ballot(L.top != 0);
                if (!live) break;
                if (L.top != 0) bvh2_step<ANY, false, true>(L, base, hits, sp_limit, ctl, deep_list, false, nullptr, image);
                if (paired && ++iterations == HOT_ITER && t < stripe_waves && __popcll(live) >= HOT_LANES && lane == 0)
                    atomicOr(&state[1 + (t >> 5)], 1u << (t & 31));              // the partner (ticket t + W) looks expensive
            }
        }
        int t_next = 0;
        if (paired) {
            // the whole wave draws: lanes 0..3 read the hot words, 4..7 the claimed words (ONE round trip), lane 0 claims
            t_next = 1 << 20;                                                 // nothing left
            // waves look for hot partners from different words: fewer collisions
            const int start = (int)(((blockIdx.x / kStripes) * WAVES + wave) >> 5) & 3;
            for (;;) {
                const unsigned word = lane < 8 ? __hip_atomic_load(&state[1 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                const unsigned free_hot = word & ~(unsigned)__shfl((int)word, (lane & 3) + 4);          // lanes 0..3: hot and not taken
                const unsigned long long any = __ballot(lane < 4 && free_hot != 0u);
                int pick = -1;
                if (any) {
                    // first non-empty word from `start` on
                    const unsigned rot = (unsigned)(((any | (any << 4)) >> start) & 15ull);
                    const int w = (start + __ffs((int)rot) - 1) & 3;
                    pick = 32 * w + __ffs(__shfl((int)free_hot, w)) - 1;
                } else {
                    // no hot partner left: the next one in the default order (one returning atomic, as the default kernel's draw)
                    if (lane == 0) pick = atomicAdd(reinterpret_cast<int*>(state), 1);
                    pick = __builtin_amdgcn_readfirstlane(pick);
                    if (pick >= stripe_waves) break;
                }
                unsigned old = 0u;
                if (lane == 0) old = atomicOr(&state[5 + (pick >> 5)], 1u << (pick & 31));
                old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
                if (!(old & (1u << (pick & 31)))) { t_next = stripe_waves + pick; break; }
            }
        } else if (lane == 0) t_next = stripe_waves + atomicAdd(reinterpret_cast<int*>(state), 1);
        t = __builtin_amdgcn_readfirstlane(t_next);
    }
}
// the follow-up kernel's extra duty: the hot / claimed masks back to zero
__global__ void k_partner_reset(int* tickets) { const int s = threadIdx.x;
    for (int w = 1; w <= 8; w++) tickets[s * kCounterStride + w] = 0; }

template <bool ANY, int LDS_N, int TOPN, int WAVES, int HOT_ITER, int HOT_LANES> void L_top_partner(LAUNCH_ARGS) {
    ensure_deep_list(s, n);
    if (!s.top_image || !s.tickets) {
        std::lock_guard<std::mutex> lock(g_mutex);
        if (!s.top_image) { HIP_CHECK(hipMalloc(&s.top_image, kMaxTopNodes * sizeof(Node2)));
            HIP_CHECK(hipMemset(s.top_image, 0, kMaxTopNodes * sizeof(Node2))); }
        if (!s.tickets) {
            HIP_CHECK(hipMalloc(&s.tickets, sizeof(int) * kMaxPhases * kStripes * kCounterStride));
            HIP_CHECK(hipMemset(s.tickets, 0, sizeof(int) * kMaxPhases * kStripes * kCounterStride));
        }
    }
    s.top_image_nodes = nullptr;
    const int groups = ((s.num_cus * (32 / WAVES) + kStripes - 1) / kStripes) * kStripes;
    hipLaunchKernelGGL((k_bvh2_top_partner<ANY, LDS_N, TOPN, WAVES, HOT_ITER, HOT_LANES>), dim3(groups), dim3(kWave * WAVES), 0, stream,
        nodes, tris, rays, hits, n, s.ctl(), s.deep_list,
                       s.top_image, s.tickets, mapped_node_ids(nodes));
    hipLaunchKernelGGL((k_bvh2_top_finish<ANY>), dim3(kFinishGroups), dim3(kWave), 0, stream, nodes, tris, rays, hits, s.ctl(),
        s.deep_list, s.deep_stack, s.tickets, s.top_image, TOPN);
    hipLaunchKernelGGL(k_partner_reset, dim3(1), dim3(kStripes), 0, stream, s.tickets);
}

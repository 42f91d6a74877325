// traversal_top.h -- the default BVH2 mapping: top of the tree staged in LDS, persistent workgroups.
// Included by traversal.hip inside its anonymous namespace, after Ctl, finish_launch, Lane, bvh2_step, start_lane, make_bases and
// the stripe constants (kStripes, kCounterStride); the host side (DeviceState, launchers, variant table) stays in traversal.hip.
//   k_bvh2_top_image / build_top_image (traversal_device.h)   the image: first TOPN inner nodes, breadth first, links for resident children
//   stage_top_image                                            a workgroup stages the image and VALIDATES it against the caller's nodes
//   k_bvh2_top_persist                                         the kernel: one resident generation of workgroups, chunks from striped
//                                                              tickets
//   k_bvh2_top_finish, k_bvh2_top_finish_history               follow-up kernels (deep rays, stale image, schedule history)
//   k_bvh2_top_refill                                          variant "refill": persistent, idle lanes are refilled (for incoherent ray
//                                                              sets)
//   k_bvh2_top                                                 lab build: one chunk per workgroup wave
#pragma once

// ---------------------------------------------------------------------------------------------
// LDS-staged top of the tree (variants "top*").  A third to a half of all node visits fall on the top 4 - 8 levels of the
// hierarchy (scripts/model_top_levels.py: atrium, 15 nodes 22 %, 63 nodes 41 / 38 %, 255 nodes 52 / 57 % of the visits of the
// primary / random set).  build_top_image copies the first TOPN nodes in breadth-first order into a 64-byte-per-node image
// whose child ids, where the child is in the image too, are kLdsTag + its byte offset; a workgroup stages the image in LDS
// behind its stacks and a lane whose top is such an id reads its node with ds_read_b128 -- LDS bandwidth instead of the
// TA -> L1 path the kernel saturates.  The caller's Node2 array is in no particular order and may change between
// launches: the image is kept per (device, stream) context and VALIDATED by every workgroup that stages it (stage_top_image);
// the follow-up kernel rebuilds it when a launch found none or a stale one (the lab's one-chunk form, k_bvh2_top, rebuilds it
// in front of every launch instead).  Ids with the tag never leave the kernel: a ray deeper than the LDS window restarts
// from the root in the follow-up kernel.
// ---------------------------------------------------------------------------------------------
template <int TOPN>
__global__ __launch_bounds__(kWave) void k_bvh2_top_image(const Node2* __restrict__ nodes, int4* __restrict__ image) {
    __shared__ int slot_node[TOPN];
    build_top_image(nodes, image, TOPN, (lds_int*)slot_node);
}

// The follow-up kernel of the persistent form: k_bvh2_finish's work, plus the image for the NEXT launch when this launch found
// none or a stale one (ctl->reserved, set by any workgroup whose validation failed).
template <bool ANY>
__global__ __launch_bounds__(kWave) void k_bvh2_top_finish(const Node2* __restrict__ nodes, const Tri1* __restrict__ tris,
                                                            const Ray1* __restrict__ rays, Hit1* __restrict__ hits,
                                                            Ctl* ctl, const int* __restrict__ deep_list, int* deep_stack, int* tickets,
                                                            int4* __restrict__ image, int capacity) {
    // (kStackCap x kWave >= kMaxTopNodes: also the image builder's slot table)
    __shared__ int stack_lds[kStackCap * kWave];
    static_assert(kStackCap * kWave >= kMaxTopNodes, "slot table");
    const bool stale = ctl->reserved != 0;
    finish_launch<ANY>(nodes, tris, rays, hits, ctl, deep_list, (lds_int*)stack_lds, tickets, blockIdx.x, gridDim.x);
    if (stale && blockIdx.x == 0) {
        build_top_image(nodes, image, capacity, (lds_int*)stack_lds);
        if (threadIdx.x == 0) ctl->reserved = 0;
    }
}

// (lab build: lab/top_kernels.h holds the forms that were measured and superseded -- one chunk per workgroup wave with the image, the
// refill kernel under a pinned register budget, work stealing inside the wave)

// A wave's rank among the waves of its stripe = its first, static ticket.  WAVE-major (round 5): consecutive ranks belong to DIFFERENT
// workgroups of the stripe (all on one XCD), so a launch of few chunks spreads over the chip instead of filling two workgroups per stripe
// with sixteen busy waves each while the others idle (workgroup-major until round 4; 128 Ki rays through the persistent kernel 0.113 ->
// 0.096 ms, 384 Ki random segments 0.163 -> 0.141, 1 Mi and more unchanged: profiles/r05_spread_tickets.txt).  The default mapping's switch
// from the one-chunk kernel moved from 576 Ki to 384 Ki rays with it.
__device__ __forceinline__ int stripe_rank(int wave) { return wave * ((int)gridDim.x / kStripes) + (int)blockIdx.x / kStripes; }

// order[stripe * stride + ticket] = chunk (may be null); cost may be null; agree: see below
struct History { const int* order; int* cost; int stride; const int* agree; };
// chunks of stripe s under the default order: its complete 32-chunk groups plus, for one stripe, the ragged last group
__device__ __forceinline__ int stripe_chunks(int total_chunks, int stripe) {
    const int full_groups = total_chunks / 32, rest = total_chunks % 32;
    return (full_groups / kStripes + (stripe < full_groups % kStripes ? 1 : 0)) * 32 + (stripe == full_groups % kStripes ? rest : 0);
}
// The follow-up kernel when the schedule history is on: workgroup 0's first wave does k_bvh2_top_finish's work, then workgroup s
// sorts the chunks of stripe s by the wave iterations this launch took for them, longest first (ties: default order), into
// order[s * stride ...] -- bitonic in LDS, at most kMaxStripeChunks keys.
// 4 Mi rays: beyond that the launch is throughput-bound and the sorted order only costs locality (16 Mi: -8 %)
constexpr int kMaxStripeChunks = 1024, kHistoryThreads = 256;
template <bool ANY>
__global__ __launch_bounds__(kHistoryThreads) void k_bvh2_top_finish_history(const Node2* __restrict__ nodes, const Tri1* __restrict__ tris,
                                                                              const Ray1* __restrict__ rays, Hit1* __restrict__ hits,
                                                                              Ctl* ctl, const int* __restrict__ deep_list, int* deep_stack,
                                                                                  int* tickets,
                                                                              int4* __restrict__ image, int capacity, int total_chunks,
                                                                              const int* __restrict__ cost, int* __restrict__ order,
                                                                                  int stride,
                                                                              int have_previous, int* __restrict__ agree) {
    __shared__ __attribute__((aligned(16))) int keys[kStackCap * kWave];        // the deep rays' stack first, then the sort keys
    // default-order positions of the chunks the PREVIOUS launch found in its expensive half
    __shared__ unsigned char previous_top[kMaxStripeChunks];
    __shared__ int agreeing;
    static_assert(kStackCap * kWave >= kMaxStripeChunks && kStackCap * kWave >= kMaxTopNodes, "one LDS block for all three uses");
    if (threadIdx.x < kWave) {
        const bool stale = ctl->reserved != 0;
        finish_launch<ANY>(nodes, tris, rays, hits, ctl, deep_list, (lds_int*)keys, tickets, blockIdx.x, gridDim.x);
        if (stale && blockIdx.x == 0) {
            build_top_image(nodes, image, capacity, (lds_int*)keys);
            if (threadIdx.x == 0) ctl->reserved = 0;
        }
    }
    __syncthreads();
    const int stripe = blockIdx.x, count = stripe_chunks(total_chunks, stripe);
    int padded = 1;
    while (padded < count) padded *= 2;
    const auto chunk_of = [&](int t) { return ((t / 32) * kStripes + stripe) * 32 + t % 32; };
    // How far may the order be trusted?  `order` still holds what the previous launch of this size sorted: the chunks of its
    // expensive half are marked, and below the share of THIS launch's expensive half among them is counted.  The next launch
    // draws by the new order only if the two launches agreed (k_bvh2_top_persist reads agree[]): unrelated ray sets of equal
    // size fall back to the default order, which keeps neighbouring chunks together.
    for (int i = threadIdx.x; i < count; i += kHistoryThreads) previous_top[i] = 0;
    if (threadIdx.x == 0) agreeing = 0;
    __syncthreads();
    if (have_previous)
        for (int t = threadIdx.x; t < count / 2; t += kHistoryThreads) {
            const int chunk = order[stripe * stride + t];
            previous_top[((chunk / 32) / kStripes) * 32 + chunk % 32] = 1;
        }
    // (at least one whole quad: the rank count below reads the keys four at a time)
    for (int i = threadIdx.x; i < max(padded, 4); i += kHistoryThreads)
        keys[i] = i < count ? (min(cost[chunk_of(i)], 0x3FFFF) << 13) | (8191 - i) : -1;
    __syncthreads();
    if (count <= kHistoryThreads) {
        // one key per thread (1 Mi rays: 256 chunks per stripe): its rank is the number of larger keys -- count LDS broadcasts, no barrier
        const int mine_key = threadIdx.x < count ? keys[threadIdx.x] : -1;
        int rank = 0;
        // (padded to a power of two with -1 keys: never larger than a real key)
        const int4* quads = reinterpret_cast<const int4*>(keys);
#pragma unroll 8
        for (int j = 0; j < (padded + 3) / 4; j++) { const int4 q = quads[j];
            rank += (q.x > mine_key) + (q.y > mine_key) + (q.z > mine_key) + (q.w > mine_key); }
        __syncthreads();
        if (threadIdx.x < count) keys[rank] = mine_key;
        __syncthreads();
    } else
    for (int k = 2; k <= padded; k *= 2)
        for (int j = k / 2; j > 0; j /= 2) {
            for (int i = threadIdx.x; i < padded; i += kHistoryThreads) {
                const int partner = i ^ j;
                if (partner > i) {
                    const int a = keys[i], b = keys[partner];
                    const bool descending = (i & k) == 0;
                    if (descending ? a < b : a > b) { keys[i] = b; keys[partner] = a; }
                }
            }
            __syncthreads();
        }
    int mine = 0;
    for (int i = threadIdx.x; i < count / 2; i += kHistoryThreads) mine += previous_top[8191 - (keys[i] & 8191)];
    if (mine) atomicAdd(&agreeing, mine);
    for (int i = threadIdx.x; i < count; i += kHistoryThreads) order[stripe * stride + i] = chunk_of(8191 - (keys[i] & 8191));
    __syncthreads();
    if (threadIdx.x == 0) { agree[2 * stripe] = have_previous ? agreeing : -count; agree[2 * stripe + 1] = count / 2; }
}

// Stages the context's image into LDS and checks it against the caller's nodes: every record must equal the node whose id it
// carries (bounds bit for bit; a child entry either the node's own child id or a link to a slot that carries that id) and
// record 0 must be the root.  Then following links through the image is the same as following child ids through `nodes`,
// whatever happened to the array since the image was built.  Returns false (workgroup-uniform) for an absent or stale image:
// the workgroup then starts its rays at node id 1 and never meets a link; ctl->reserved asks the follow-up kernel for a new image.
// `max_id`: node ids the caller's allocation is known to hold (the host asks the runtime for the mapped range behind `nodes`):
// a record whose id lies beyond it is stale, and is not dereferenced.
template <int TOPN, int THREADS>
__device__ __forceinline__ bool stage_top_image(const Node2* __restrict__ nodes, const int4* __restrict__ top_image, lds_int* image,
    lds_int* flag, Ctl* ctl, int max_id) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    bool ok = true;
    const int* image_ints = reinterpret_cast<const int*>(top_image);
    for (int j = threadIdx.x; j < TOPN * 4; j += THREADS) {
        const int slot = j >> 2, quarter = j & 3, id = image_ints[slot * 16 + 14];
        const i32x4 rec = reinterpret_cast<const i32x4*>(top_image)[j];
        reinterpret_cast<__attribute__((address_space(3))) i32x4*>(image)[j] = rec;
        if (id > max_id) ok = false;
        else if (id > 0) {
            const i32x4 real = reinterpret_cast<const i32x4*>(nodes + (id - 1))[quarter];
            if (quarter < 3) ok &= rec.x == real.x && rec.y == real.y && rec.z == real.z && rec.w == real.w;
            else ok &= (rec.x >= kLdsTag || rec.x == real.x) && (rec.y >= kLdsTag || rec.y == real.y);     // links: after the barrier
        } else if (slot == 0) ok = false;
        if (j == 0) ok &= id == 1;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < TOPN * 4; j += THREADS) {
        if ((j & 3) != 3) continue;
        const int slot = j >> 2, id = image[slot * 16 + 14];
        if (id <= 0 || id > max_id) continue;
        const int2 real = *reinterpret_cast<const int2*>(&nodes[id - 1].child[0]);
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int link = image[slot * 16 + 12 + k], child = k ? real.y : real.x;
            if (link >= kLdsTag) {
                const unsigned target = (unsigned)(link - kLdsTag) / (unsigned)sizeof(Node2);
                ok &= (link - kLdsTag) % (int)sizeof(Node2) == 0 && target < (unsigned)TOPN && child > 0
                    && image[(target < (unsigned)TOPN ? target : 0u) * 16 + 14] == child;
            }
        }
    }
    // the verdict through one LDS word (`flag`: any word the caller does not need yet; __syncthreads_and would take 256 bytes of
    // LDS of its own, and two 16-wave workgroups fill the CU's 160 KB to within 128 bytes)
    if (threadIdx.x == 0) *flag = 1;
    __syncthreads();
    if (!ok) *flag = 0;
    __syncthreads();
    const bool all_ok = *flag != 0;
    __syncthreads();
    // (atomic: read by the launch's last workgroup, behind another XCD's L2)
    if (!all_ok && threadIdx.x == 0) __hip_atomic_store(&ctl->reserved, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return all_ok;
}

// Persistent form: the grid is one generation of workgroups (32 / WAVES per CU), every workgroup stages the image once and its waves keep
// drawing 64-ray chunks -- a ticket from the counter of their stripe (64 counters, 64 bytes apart: one counter saturates near 88
// atomics/us, a 1 Mi-ray launch draws 83 per us), ticket t of stripe s = chunk ((t / 32) * 64 + s) * 32 + t % 32, the same XCD-aware order
// as k_bvh2_single (stripe s runs on XCD s % 8) -- until their stripe's share is used up.  No workgroup waits for LDS that a finished
// neighbour wave still pins (what makes WAVES > 2 lose at 16 Mi rays in the non-persistent form), the image is staged 512 times per launch
// instead of 16 384 times, and the next ticket is drawn while the current chunk is traced.
// Schedule history (HISTORY; rodent_hip_schedule_history): every chunk's wave iterations are recorded, the follow-up kernel
// sorts each stripe's chunks by them, and the next launch of the same size draws its chunks in that order -- longest first
// (frame-to-frame cost feedback, as renderers balance tiles by the previous frame's cost).  The order only decides WHEN a chunk
// is traced; a launch without usable history (first launch, other size) takes the default order.
// FUSED: 0 = a follow-up kernel finishes the launch (deep rays, counters, stale image); 1 (lab) = the last workgroup to end does,
// every workgroup releasing what it wrote; 2 = the last workgroup does, and only the rare paths pay for a fence: a lane that
// hands its ray to the deep list publishes it on the spot (bvh2_step<FENCE>), the stale-image flag is an atomic, and a launch
// without deep rays and with a valid image -- every launch but the first on a hierarchy -- costs one relaxed atomic per workgroup.
// LAZY: miss records are stored when a chunk ends, for the rays that found nothing (start_lane<LAZY>, finish_lane).
template <bool ANY, int LDS_N, int TOPN, int WAVES, bool PREFETCH, int OCC = 32, bool TRACE = false, int PRIO = 0, int FUSED = 0,
    bool HISTORY = false, bool LAZY = false>
__global__ __launch_bounds__(kWave * WAVES) __attribute__((amdgpu_waves_per_eu(8, 8))) void
    k_bvh2_top_persist(const Node2* __restrict__ nodes, const Tri1* __restrict__ tris,
                                                                     const Ray1* __restrict__ rays, Hit1* __restrict__ hits, int n,
                                                                     Ctl* ctl, int* __restrict__ deep_list, const int* __restrict__ perm,
                                                                     int4* __restrict__ top_image, int* __restrict__ tickets, int max_id,
                                                                         int* spill, History hist) {
    constexpr int kStackInts = WAVES * (LDS_N + 1) * kWave, kGroup = 32;
    static_assert((kStackInts + TOPN * 16) * 4 * (OCC / WAVES) <= 160 * 1024, "OCC waves per CU must fit their stacks and images in LDS");
    __shared__ __attribute__((aligned(16))) int lds_raw[kStackInts + TOPN * 16];
    const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
    lds_int* col = (lds_int*)lds_raw + wave * (LDS_N + 1) * kWave + lane;
    lds_int* image = (lds_int*)lds_raw + kStackInts;
    // record 0 of the image, or node 1
    const int root = stage_top_image<TOPN, kWave * WAVES>(nodes, (const int4*)top_image, image, (lds_int*)lds_raw, ctl, max_id)
        ? kLdsTag : 1;
    // stats[6]: the launch ran on the image (read by the tests; workgroup 0 speaks for all, see k_bvh2_top_auto)
    if (root != 1 && threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(&ctl->stats[6], 1ull);
    // stripe = workgroup index mod 64 (its XCD = stripe mod 8); the first ticket of a wave is its rank inside the stripe, the
    // counter hands out the tickets behind those
    const int total_chunks = (n + kWave - 1) / kWave, stripe = blockIdx.x % kStripes, stripe_waves = (gridDim.x / kStripes) * WAVES;
    int* counter = tickets + stripe * kCounterStride;
    int t = stripe_rank(wave);
    lds_int* const sp_limit = col + LDS_N * kWave;
    const Bases base = make_bases(nodes, tris);
    const int my_chunks = HISTORY ? stripe_chunks(total_chunks, stripe) : 0;
    static_assert(kStripes == kWave, "one stripe's agreement record per lane");
    // HISTORY: an order exists and the last two launches agreed on what is expensive
    bool use_order = false;
    if (HISTORY && hist.order) {
        int a = hist.agree[2 * lane], h = hist.agree[2 * lane + 1];         // kStripes == kWave: one stripe per lane
        for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); h += __shfl_xor(h, off); }
        use_order = 4 * (long long)a >= 3 * (long long)h;
    }
    for (;;) {
        int chunk;
        if (HISTORY && use_order) {
            if (t >= my_chunks) break;
            chunk = hist.order[stripe * hist.stride + t];
        } else {
            const int group_first = ((t / kGroup) * kStripes + stripe) * kGroup;
            if (group_first >= total_chunks) break;                          // this stripe's share is used up
            chunk = group_first + t % kGroup;
        }
        int t_next = 0;
        if (PREFETCH && lane == 0) t_next = atomicAdd(counter, 1);           // in flight while this chunk is traced
        if (chunk < total_chunks) {
            const int first_ray = chunk * kWave, lane_ray = first_ray + lane;
            Lane L = start_lane<LAZY>(rays, hits, lane_ray < n ? (perm ? perm[lane_ray] : lane_ray) : -1,
                perm ? perm[first_ray] : first_ray, col);
            if (L.top != 0) L.top = root;
            const unsigned long long t_start = TRACE ? __builtin_amdgcn_s_memrealtime() : 0ull;
            int iterations = 0;
            if (PRIO > 0) __builtin_amdgcn_s_setprio(0);
            while (__ballot(L.top != 0)) {
                if (L.top != 0) bvh2_step<ANY, false, true, LAZY, false, false, LDS_N, WAVES>(L, base, hits, sp_limit, ctl, deep_list,
                    false, nullptr, image, nullptr, spill);
                // lab (PRIO == -2, "top-double"): a lane whose next node is in the LDS image visits it in the same iteration -- two levels
                // of the top of the tree per dependent step where the fetch is a ds_read (VERDICT r2 item 2's "BVH4-collapsed image",
                // without a second layout)
                if (PRIO == -2 && L.top >= kLdsTag) bvh2_step<ANY, false, true, LAZY, false, false, LDS_N, WAVES>(L, base, hits, sp_limit,
                    ctl, deep_list, false, nullptr, image, nullptr, spill);
                if (TRACE || PRIO > 0 || HISTORY) iterations++;
                // lab: a chunk that is still running after PRIO iterations is on the critical path
                if (PRIO > 0 && iterations == PRIO) __builtin_amdgcn_s_setprio(3);
            }
            if (LAZY) finish_lane(L, rays, hits);
            if (HISTORY && hist.cost && lane == 0) hist.cost[chunk] = iterations;
            // lab: per chunk start / end (100 MHz), iterations, wave << 32 | ticket
            if (TRACE && lane == 0 && ctl->trace && chunk < 16384) {
                unsigned long long* tr = ctl->trace + 4 * (size_t)chunk;
                tr[0] = t_start; tr[1] = __builtin_amdgcn_s_memrealtime(); tr[2] = (unsigned long long)iterations;
                tr[3] = ((unsigned long long)(blockIdx.x * WAVES + wave) << 32) | (unsigned)t;
            }
        }
        if (!PREFETCH && lane == 0) t_next = atomicAdd(counter, 1);
        t = stripe_waves + __builtin_amdgcn_readfirstlane(t_next);
    }
    if (FUSED) {
        // The workgroup that finishes last does the follow-up kernel's work (deep rays, counters, a new image if this launch found
        // none or a stale one): one launch instead of two.  FUSED == 1: every workgroup publishes what it wrote (agent-scope
        // release: its XCD's L2 is written back) before it counts itself done, the last one acquires.  FUSED == 2: the counting
        // is a relaxed atomic; the rays of the deep list were published by the lanes that put them there, and the last
        // workgroup acquires only if there are any.
        __syncthreads();
        if (threadIdx.x == 0) {
            if (FUSED == 1) __threadfence();
            const int done = FUSED == 1 ? __hip_atomic_fetch_add(&ctl->counter, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT)
                                        : __hip_atomic_fetch_add(&ctl->counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lds_raw[0] = done == (int)gridDim.x - 1;
        }
        __syncthreads();
        if (!lds_raw[0] || wave != 0) return;
        const int deep = __hip_atomic_load(&ctl->deep_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (FUSED == 1 || deep > 0) __threadfence();
        const bool stale = __hip_atomic_load(&ctl->reserved, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        finish_launch<ANY>(nodes, tris, rays, hits, ctl, deep_list, (lds_int*)lds_raw, tickets, 0, 1, deep);
        if (stale) {
            build_top_image(nodes, top_image, TOPN, (lds_int*)lds_raw);        // (the stacks are idle now)
            if (lane == 0) ctl->reserved = 0;
        }
    }
}

// Do the rays held by the lanes of `valid` share an origin (a camera's) or a direction (parallel rays)?  Wave-uniform; call in uniform
// control flow.
__device__ __forceinline__ bool wave_rays_coherent(float ox, float oy, float oz, float dx, float dy, float dz, bool valid) {
    const auto differs = [&](float x) { return x != __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); };
    const bool o = differs(ox) | differs(oy) | differs(oz), d = differs(dx) | differs(dy) | differs(dz);
    // (lane 0 is valid whenever any lane is: rays are handed out in lane order)
    return __ballot(valid && o) == 0ull || __ballot(valid && d) == 0ull;
}
// What the default mapping remembers between launches while the ray-kind hint is on (DeviceState::host_kinds, pinned host memory the
// kernels write straight into; null while it is off): the id of the last launch in which a workgroup's first rays -- EVERY workgroup
// reports, so a list that is coherent anywhere says so -- were coherent [0] / incoherent [1].  A hint for the host's choice of kernel,
// nothing else.
__device__ __forceinline__ void report_ray_kind(int* host_kinds, int launch_id, bool coherent) {
    if (host_kinds && threadIdx.x == 0) __hip_atomic_store(&host_kinds[coherent ? 0 : 1], launch_id, __ATOMIC_RELAXED,
        __HIP_MEMORY_SCOPE_SYSTEM);
}

// Persistent form with lane refill (variant "refill"): a wave does not wait for the last ray of a 64-ray chunk.  As soon as REFILL of its
// lanes are idle it draws that many rays from its stripe's counter (one atomic per refill) and starts them in the idle lanes; the rest keep
// stepping.  Ticket t of stripe s is ray ((t / 2048) * 64 + s) * 2048 + t % 2048 (the same 32-chunk groups), the first 64 tickets of a wave
// are static.  Which rays share a wave changes, what a ray visits does not.
// ADAPT: the threshold is chosen at every draw from the rays just drawn -- rays that share an origin (a camera's) wait for the whole
// wave like a chunk (their neighbours are in step, a refill would take that away), anything else refills at REFILL idle lanes.
template <bool ANY, int LDS_N, int TOPN, int WAVES, int REFILL, bool ADAPT = false, bool FENCE = false>
__device__ __forceinline__ void top_refill_body(const Node2* __restrict__ nodes, const Tri1* __restrict__ tris,
                                                                    const Ray1* __restrict__ rays, Hit1* __restrict__ hits, int n,
                                                                    Ctl* ctl, int* __restrict__ deep_list,
                                                                        const int4* __restrict__ top_image, int* __restrict__ tickets,
                                                                        int max_id,
                                                                    int* spill, int* host_kinds = nullptr, int launch_id = 0) {
    constexpr int kStackInts = WAVES * (LDS_N + 1) * kWave, kGroupRays = 32 * kWave;
    static_assert((kStackInts + TOPN * 16) * 4 * (32 / WAVES) <= 160 * 1024, "32 waves per CU must fit their stacks and images in LDS");
    __shared__ __attribute__((aligned(16))) int lds_raw[kStackInts + TOPN * 16];
    const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
    lds_int* col = (lds_int*)lds_raw + wave * (LDS_N + 1) * kWave + lane;
    lds_int* image = (lds_int*)lds_raw + kStackInts;
    const int root = stage_top_image<TOPN, kWave * WAVES>(nodes, top_image, image, (lds_int*)lds_raw, ctl, max_id) ? kLdsTag : 1;
    // stats[6] as k_bvh2_top_auto, whose place this kernel takes; stats[4]: workgroups of this kernel launched by the default mapping (read
    // by the tests)
    if (host_kinds && threadIdx.x == 0) { if (root != 1) atomicAdd(&ctl->stats[6], 1ull); atomicAdd(&ctl->stats[4], 1ull); }
    const int stripe = blockIdx.x % kStripes, stripe_waves = (gridDim.x / kStripes) * WAVES;
    int* counter = tickets + stripe * kCounterStride;
    const auto ray_of = [&](int t) { return ((t / kGroupRays) * kStripes + stripe) * kGroupRays + t % kGroupRays; };
    lds_int* const sp_limit = col + LDS_N * kWave;
    const Bases base = make_bases(nodes, tris);
    Lane L;
    {
        const int r = ray_of(stripe_rank(wave) * kWave + lane);
        L = start_lane(rays, hits, r < n ? r : -1, 0, col);
        if (L.top != 0) L.top = root;
    }
    if (host_kinds && wave == 0) report_ray_kind(host_kinds, launch_id,
        wave_rays_coherent(L.ray.ox, L.ray.oy, L.ray.oz, L.ray.dx, L.ray.dy, L.ray.dz, L.top != 0));
    // the rays a draw started: one origin for all of them?  (readlane: the leader is wave-uniform)
    const auto one_origin = [&](unsigned long long started) {
        if (!started) return false;
        const int leader = __ffsll((long long)started) - 1;
        const float ox = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(L.ray.ox), leader));
        const float oy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(L.ray.oy), leader));
        const float oz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(L.ray.oz), leader));
        const bool mine = (started >> lane) & 1ull;
        return __ballot(mine && !(L.ray.ox == ox && L.ray.oy == oy && L.ray.oz == oz)) == 0ull;
    };
    int need = REFILL;                                                       // wave-uniform: idle lanes that trigger the next draw
    if (ADAPT) need = one_origin(__ballot(L.top != 0)) ? kWave : REFILL;
    // One flat loop, deliberately: with the steps in an inner loop of their own whole chunks run 8 % faster through this kernel and
    // refilled waves 8 % slower (8 Mi random segments 1.27 -> 1.37 ms, and 66 VGPRs unless capped) -- profiles/r03_sweep_adaptive.log.
    bool more = true;                                                        // wave-uniform: the stripe may have rays left
    for (;;) {
        const unsigned long long live = __ballot(L.top != 0);
        if (more && __popcll(live) <= kWave - need) {
            const int want = kWave - __popcll(live);
            int first = 0;
            if (lane == 0) first = atomicAdd(counter, want);
            first = stripe_waves * kWave + __builtin_amdgcn_readfirstlane(first);
            // ray_of grows with the ticket: once past the end, always past the end
            more = ray_of(first) < n;
            if (L.top == 0) {
                const int r = ray_of(first + __popcll(~live & ((1ull << lane) - 1ull)));
                if (r < n) {
                    L = start_lane(rays, hits, r, r, col);
                    L.top = root;
                }
            }
            if (ADAPT) need = one_origin(__ballot(L.top != 0) & ~live) ? kWave : REFILL;
            continue;
        }
        if (live == 0) break;
        if (L.top != 0) bvh2_step<ANY, false, true, false, false, false, LDS_N, WAVES>(L, base, hits, sp_limit, ctl, deep_list, false,
            nullptr, image, nullptr, spill);
    }
}

template <bool ANY, int LDS_N, int TOPN, int WAVES, int REFILL, bool ADAPT = false, bool FENCE = false>
__global__ __launch_bounds__(kWave * WAVES, 32 / WAVES) void k_bvh2_top_refill(const Node2* __restrict__ nodes,
    const Tri1* __restrict__ tris,
                                                                    const Ray1* __restrict__ rays, Hit1* __restrict__ hits, int n,
                                                                    Ctl* ctl, int* __restrict__ deep_list,
                                                                        const int4* __restrict__ top_image, int* __restrict__ tickets,
                                                                        int max_id,
                                                                    int* spill, int* host_kinds, int launch_id) {
    top_refill_body<ANY, LDS_N, TOPN, WAVES, REFILL, ADAPT, FENCE>(nodes, tris, rays, hits, n, ctl, deep_list, top_image, tickets, max_id,
        spill, host_kinds, launch_id);
}

// The default from round 4 on: ONE persistent kernel that chooses per wave between the two loop forms above (compaction
// that switches itself on -- the reference compacts unconditionally, render/mapping_gpu.impala:267-300; Aila's kernel refills
// unconditionally, tools/bench_aila/kepler_dynamic_fetch.cu:116-127,361-362).  A wave draws 64 consecutive tickets of its stripe
// (ray-granular tickets, the 2048-ray groups of k_bvh2_top_refill) and looks at the rays it got: if they share an origin (a camera's) or a
// direction (parallel rays) they are traced as a chunk -- k_bvh2_top_persist's loop: their neighbours stay in step -- and the wave draws
// the next 64; anything else puts the wave into the refill loop (REFILL idle lanes trigger a draw).
// The choice is per wave, made once from the wave's first 64 rays (re-deciding at every draw couples the two loops' register
// allocation: the chunk loop then reloaded the spilled tmax in every iteration), costs ~25 instructions, and needs no probe launch: a list
// of camera rays runs as it did, a list of incoherent segments as through "refill".
// The launch finishes itself like k_bvh2_top_persist<FUSED = 2> (last workgroup: deep rays, counters, stale image).
// LAZY (round 6): no miss record up front -- a ray's record is stored when a triangle is accepted, and the miss record when the ray ends
// without one, from the tmax the lane still holds (finish_lane_reg): nearly every camera ray finds a triangle, and the record stored up
// front was written back to memory before the hit overwrote it (WRITE_SIZE 1.64 x the Hit1 array, profiles/r05_traffic.json).  Round 3's
// lazy form re-read the ray at the end of every chunk (a dependent load on the critical path: -4 %).
template <bool ANY, int LDS_N, int TOPN, int WAVES, int REFILL,
    int MODE = 0 /* lab: 1 = whole chunks whatever the rays, 2 = refill whatever the rays */,
    bool FUSED = true /* lab: false = a follow-up kernel finishes the launch */, bool LAZY = false>
__global__ __launch_bounds__(kWave * WAVES) __attribute__((amdgpu_waves_per_eu(8, 8))) void
    k_bvh2_top_auto(const Node2* __restrict__ nodes, const Tri1* __restrict__ tris,
                                                                  const Ray1* __restrict__ rays, Hit1* __restrict__ hits, int n,
                                                                  Ctl* ctl, int* __restrict__ deep_list, int4* __restrict__ top_image,
                                                                      int* __restrict__ tickets, int max_id,
                                                                  int* spill, int* host_kinds, int launch_id, int grid_w) {
    constexpr int kStackInts = WAVES * (LDS_N + 1) * kWave, kGroupRays = 32 * kWave;
    static_assert((kStackInts + TOPN * 16) * 4 * (32 / WAVES) <= 160 * 1024, "32 waves per CU must fit their stacks and images in LDS");
    __shared__ __attribute__((aligned(16))) int lds_raw[kStackInts + TOPN * 16];
    const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
    lds_int* col = (lds_int*)lds_raw + wave * (LDS_N + 1) * kWave + lane;
    lds_int* image = (lds_int*)lds_raw + kStackInts;
    const int root = stage_top_image<TOPN, kWave * WAVES>(nodes, (const int4*)top_image, image, (lds_int*)lds_raw, ctl, max_id)
        ? kLdsTag : 1;
    // stats[6]: this launch ran on the image; stats[5]: its first wave chose the refill loop (read by the tests).  Workgroup 0 speaks for
    // all: every workgroup validates the same image, and 512 atomics on one address cost a launch of few rays 4 us (Cornell box, 64 ... 393
    // 216 rays: 21.0 ... 24.5 us -> 16.7 ... 21.0, profiles/r05_fixed_costs.txt)
    if (root != 1 && threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(&ctl->stats[6], 1ull);
    const int stripe = blockIdx.x % kStripes, stripe_waves = (gridDim.x / kStripes) * WAVES;
    int* counter = tickets + stripe * kCounterStride;
    const auto ray_of = [&](int t) { return ((t / kGroupRays) * kStripes + stripe) * kGroupRays + t % kGroupRays; };
    lds_int* const sp_limit = col + LDS_N * kWave;
    const Bases base = make_bases(nodes, tris);
    // the wave's first 64 tickets are its rank in the stripe; the counter hands out those behind
    int t = stripe_rank(wave) * kWave;
    bool coherent = MODE != 2;
    // (ray_of grows with the ticket: otherwise this stripe's share is used up already)
    if (MODE == 0 && ray_of(t) < n) {
        // one origin or one direction for all of the wave's first 64 rays?
        const int r = ray_of(t + lane);
        const float4* p = reinterpret_cast<const float4*>(rays + (r < n ? r : ray_of(t)));
        const float4 o = p[0], d = p[1];
        if (grid_w < 0) grid_w = detect_ray_grid(rays, n);                   // (its loads travel with p[0], p[1])
        coherent = wave_rays_coherent(o.x, o.y, o.z, d.x, d.y, d.z, r < n);
        if (wave == 0) report_ray_kind(host_kinds, launch_id, coherent);
    }
    if (!coherent && threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(&ctl->stats[5], 1ull);
    // camera rays in image order: the wave's 64 rays are an 8 x 8-pixel tile, not 64 pixels of a row (see detect_ray_grid); rows behind the
    // last whole band of 8 stay as they are
    grid_w = __builtin_amdgcn_readfirstlane(grid_w > 0 && (grid_w & 7) == 0 ? grid_w : 0);
    const int tiled_rays = tiled_ray_count(grid_w, n);
    // stats[2]: the image width the launch traced tiles of (read by the tests)
    if (grid_w > 0 && threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(&ctl->stats[2], (unsigned long long)grid_w);
    if (MODE == 1 || (MODE == 0 && coherent)) {
        for (;;) {                                                           // k_bvh2_top_persist's loop on 64-ticket draws
            int first_ray = ray_of(t);
            if (first_ray >= n) break;
            int r = ray_of(t + lane);
            if (first_ray < tiled_rays) r = tile_ray(first_ray, lane, grid_w);   // (first_ray: the tile's first pixel)
            Lane L = start_lane<LAZY>(rays, hits, r < n ? r : -1, first_ray, col);
            if (L.top != 0) L.top = root;
            while (__ballot(L.top != 0)) {
                if (L.top != 0) bvh2_step<ANY, false, true, LAZY, false, false, LDS_N, WAVES>(L, base, hits, sp_limit, ctl, deep_list,
                    false, nullptr, image, nullptr, spill);
            }
            if (LAZY) finish_lane_reg(L, rays, hits);
            int t_next = 0;
            if (lane == 0) t_next = atomicAdd(counter, kWave);
            t = stripe_waves * kWave + __builtin_amdgcn_readfirstlane(t_next);
        }
    } else if (MODE != 1) {
        // The launch's positions map to rays the SAME way in both loops (`coherent` is per wave; a list that holds an image and
        // segments -- a recognised width, waves of both kinds -- traced tiles in the chunk loop and list order here: rays of the band both
        // kinds share were traced twice or never).
        const auto ray_at = [&](int pos) { return pos < tiled_rays ? tile_ray_at(pos, grid_w) : pos; };
        Lane L;
        {
            const int r = ray_at(ray_of(t + lane));
            L = start_lane<LAZY>(rays, hits, r < n ? r : -1, 0, col);
            if (L.top != 0) L.top = root;
        }
        // One flat loop, deliberately (see k_bvh2_top_refill).
        bool more = true;                                                    // wave-uniform: the stripe may have rays left
        for (;;) {
            const unsigned long long live = __ballot(L.top != 0);
            if (more && __popcll(live) <= kWave - REFILL) {
                const int want = kWave - __popcll(live);
                int first = 0;
                if (lane == 0) first = atomicAdd(counter, want);
                first = stripe_waves * kWave + __builtin_amdgcn_readfirstlane(first);
                more = ray_of(first) < n;
                if (L.top == 0) {
                    if (LAZY) { finish_lane_reg(L, rays, hits); L.ray_id = -1; }  // the ray that ended in this lane
                    const int pos = ray_of(first + __popcll(~live & ((1ull << lane) - 1ull)));
                    // (tiled positions lie below n and map below n: one test for both)
                    if (pos < n) {
                        const int rr = ray_at(pos);
                        L = start_lane<LAZY>(rays, hits, rr, rr, col);
                        L.top = root;
                    }
                }
                continue;
            }
            if (live == 0) break;
            if (L.top != 0) bvh2_step<ANY, false, true, LAZY, false, false, LDS_N, WAVES>(L, base, hits, sp_limit, ctl, deep_list, false,
                nullptr, image, nullptr, spill);
        }
        if (LAZY) finish_lane_reg(L, rays, hits);
    }
    // the workgroup that finishes last does the follow-up work (k_bvh2_top_persist, FUSED == 2)
    if (!FUSED) return;
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


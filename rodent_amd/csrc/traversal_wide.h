// traversal_wide.h -- BVH4 / BVH8 + Tri4 traversal for MI355X: the single-step schedule of k_bvh2_single applied to
// the reference's wide layouts (Node4 / Node8 / Tri4, src/traversal/mapping_cpu.impala:3-22).
//
// Visit order per ray = the reference GPU kernel's branch for arity != 2 (src/traversal/mapping_gpu.impala:136-153)
// with its sequential in-packet triangle test (:160-169); oracle: traverse_gpu_wide ("B1g", oracle/traversal_oracle.c):
//   pop the node; unordered fminf/fmaxf box test of every child against the CURRENT tmax; the nearest hit child
//   (strict <, starting from tmax) goes on top, the others underneath in slot order; nothing is culled on pop;
//   the triangles of a packet are tested in slot order, tmax shrinking in between, an invalid slot (prim_id == -1)
//   ends the packet, prim_id[3] < 0 ends the leaf.
// What is designed for CDNA4 (as in traversal.hip): one ray per lane, one step -- a node OR a Tri4 packet -- per wave
// iteration with all its 16-byte loads in flight together, the popped entry read from LDS meanwhile, the stack as a
// cursor into an LDS-only window ([entry][lane], N spare rows so that a node step never writes out of bounds), the hit
// record in memory, XCD-aware chunk mapping, and a one-wave follow-up kernel with the reference's 64-entry stack in
// global memory for the rays that outgrow the window.
// A wide step has instruction-level parallelism the BVH2 step lacks (N independent slab tests / 4 independent triangle
// tests), and a ray needs 21.0 (BVH4) or 17.6 (BVH8) dependent steps on the atrium's primary rays where BVH2 needs 39.3.
//
// Included by traversal.hip inside its anonymous namespace (after Ctl, GlobalStack, DeviceState, blocks_for).
#pragma once

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int   i32x4 __attribute__((ext_vector_type(4)));

template <int N> struct WideLayout {
    static constexpr unsigned kNodeBytes = 32u * N;        // bounds[6][N], child[N], pad[N]
    static constexpr int kNodeVecs = 7 * N / 4;            // 16-byte pieces a node step needs: 6 rows + child ids
    static constexpr int kTriVecs = 13;                    // v0, e1, e2, n (12 rows of 4) + prim_id
    static constexpr int kVecs = kNodeVecs > kTriVecs ? kNodeVecs : kTriVecs;
};

// Row r (0..5 = lo_x, hi_x, lo_y, hi_y, lo_z, hi_z; 6 = child ids) of child k from the loaded pieces.
template <int N> __device__ __forceinline__ float wide_bound(const f32x4* d, int r, int k) { return d[r * (N / 4) + k / 4][k % 4]; }

// Triangle test split at the one comparison that depends on the shrinking tmax (intersection.impala:164-192): returns
// whether everything else accepts; the caller compares t <= abs_det * tmax with the tmax current at that slot.
__device__ __forceinline__ bool tri_pre(const RayX& r, float v0x, float v0y, float v0z, float e1x, float e1y, float e1z,
                                        float e2x, float e2y, float e2z, float nx, float ny, float nz,
                                        float& t, float& u, float& v, float& abs_det) {
    const float cx = v0x - r.ox, cy = v0y - r.oy, cz = v0z - r.oz;
    const float rx = cross_x(r.dx, r.dy, r.dz, cx, cy, cz);
    const float ry = cross_y(r.dx, r.dy, r.dz, cx, cy, cz);
    const float rz = cross_z(r.dx, r.dy, r.dz, cx, cy, cz);
    const float det = dot3(nx, ny, nz, r.dx, r.dy, r.dz);
    abs_det = fabsf(det);
    u = prodsign(dot3(rx, ry, rz, e2x, e2y, e2z), det);
    v = prodsign(dot3(rx, ry, rz, e1x, e1y, e1z), det);
    t = prodsign(dot3(cx, cy, cz, nx, ny, nz), det);
    return (u >= 0.0f) && (v >= 0.0f) && (u + v <= abs_det) && (abs_det != 0.0f) && (t >= abs_det * r.tmin);
}

// TOP: the workgroup has staged the top of the hierarchy in LDS (k_wide_top_persist, stage_wide_top): `root` and every child id
// >= kLdsTag is the byte offset of a node record inside `image` (same layout as the node, child ids of staged children
// rewritten the same way), fetched with ds_read_b128 instead of through the vector-memory pipeline.
template <bool ANY, int N, int LDS_N, bool TOP = false>
__device__ __forceinline__ void wide_chunk(const char* __restrict__ nodes, const Tri4* __restrict__ tris, const Ray1* __restrict__ rays,
                                           Hit1* __restrict__ hits, int n, Ctl* ctl, int* __restrict__ deep_list, lds_int* col,
                                               int first_ray,
                                           lds_int* image = nullptr, int root = 1, int grid_w = 0) {
    typedef WideLayout<N> L;
    int lane_ray = first_ray + (int)(threadIdx.x % kWave);
    // per-pixel lists: an 8 x 8-pixel tile per wavefront (detect_ray_grid, traversal.hip)
    if (first_ray < tiled_ray_count(grid_w, n)) lane_ray = tile_ray(first_ray, (int)(threadIdx.x % kWave), grid_w);
    const int ray_id = lane_ray < n ? lane_ray : -1;
    RayX ray = load_ray(rays, ray_id >= 0 ? ray_id : first_ray);
    if (ray_id >= 0) store_hit(hits, ray_id, -1, ray.tmax, 0.0f, 0.0f);       // the miss record; accepted triangles overwrite it
    ray.tmin = canonical(ray.tmin); ray.tmax = canonical(ray.tmax);
    int top = ray_id >= 0 ? root : 0;
    lds_int* sp = col;                               // the top entry of the stack in memory (mem[ptr] of the oracle)
    lds_int* const sp_limit = col + LDS_N * kWave;
    col[0] = 0;
    typedef const __attribute__((address_space(1))) char* gptr;
    // node ids are 1-based
    unsigned long long node_bits = reinterpret_cast<unsigned long long>(nodes) - L::kNodeBytes,
        tri_bits = reinterpret_cast<unsigned long long>(tris);
    asm volatile("" : "+v"(node_bits), "+v"(tri_bits));
    const gptr node_base = (gptr)node_bits, tri_base = (gptr)tri_bits;
    while (__ballot(top != 0)) {
        if (top != 0) {
            const bool is_node = top > 0;
            f32x4 d[L::kVecs];
            if (TOP && top >= kLdsTag) {
                const __attribute__((address_space(3))) f32x4* p =
                    (const __attribute__((address_space(3))) f32x4*)((__attribute__((address_space(3))) const char*)image
                    + (unsigned)(top - kLdsTag));
#pragma unroll
                for (int k = 0; k < L::kNodeVecs; k++) d[k] = p[k];
            } else {
                const unsigned idx = (unsigned)(is_node ? top : ~top), stride = is_node ? L::kNodeBytes : (unsigned)sizeof(Tri4);
                const gptr addr = (is_node ? node_base : tri_base) + (size_t)idx * stride;
                const __attribute__((address_space(1))) f32x4* p = (const __attribute__((address_space(1))) f32x4*)addr;
                // the pieces both kinds need, then the ones only the longer kind needs (exec-masked loads: the other lanes must
                // not read past the end of their array)
                constexpr int kShared = L::kNodeVecs < L::kTriVecs ? L::kNodeVecs : L::kTriVecs;
#pragma unroll
                for (int k = 0; k < kShared; k++) d[k] = p[k];
                if (L::kNodeVecs > L::kTriVecs ? is_node : !is_node) {
#pragma unroll
                    for (int k = kShared; k < L::kVecs; k++) d[k] = p[k];
                }
            }
            const int popped = *sp;
            // every load in flight before anything is consumed (see unified_chunk in traversal.hip)
            if constexpr (L::kVecs == 13)
                asm volatile("" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]),
                    "+v"(d[8]), "+v"(d[9]), "+v"(d[10]), "+v"(d[11]), "+v"(d[12]));
            else
                asm volatile("" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]),
                    "+v"(d[8]), "+v"(d[9]), "+v"(d[10]), "+v"(d[11]), "+v"(d[12]), "+v"(d[13]));
            if (is_node) {
                // mapping_gpu.impala:136-153.  The pop frees the slot sp points at; hit children are written from there
                // upwards, each write holding either the child (not nearer than the best so far) or the previous best.
                int cur = popped; float tnear = ray.tmax;
                lds_int* wp = sp;
#pragma unroll
                for (int k = 0; k < N; k++) {
                    float te;
                    const int child = __float_as_int(wide_bound<N>(d, 6, k));
                    const bool hit = slab_canonical(ray, wide_bound<N>(d, 0, k), wide_bound<N>(d, 1, k), wide_bound<N>(d, 2, k),
                                                    wide_bound<N>(d, 3, k), wide_bound<N>(d, 4, k), wide_bound<N>(d, 5, k), te)
                                                        && child != 0;
                    if (hit) {
                        const bool nearer = ANY || te < tnear;                // strict < (:145)
                        *wp = nearer ? cur : child; wp += kWave;
                        cur = nearer ? child : cur; tnear = nearer ? te : tnear;
                    }
                }
                top = cur;
                if (wp - kWave >= sp_limit && wp > sp + kWave) {              // grew beyond the LDS window: k_wide_finish redoes this ray
                    deep_list[atomicAdd(&ctl->deep_count, 1)] = ray_id;
                    top = 0;
                }
                sp = wp - kWave;                                              // no child hit: sp - 1 (the pop)
            } else {
                // one Tri4 packet (mapping_gpu.impala:160-169 on mapping_cpu.impala:24-42): four independent tests, accepted in slot order
                const i32x4 pid = __builtin_bit_cast(i32x4, d[12]);
                bool valid = true, found = false;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    valid = valid && pid[k] != -1;                            // an unused slot ends the packet (mapping_cpu.impala:38)
                    float t, u, v, abs_det;
                    const bool pre = tri_pre(ray, d[0][k], d[1][k], d[2][k], d[3][k], d[4][k], d[5][k], d[6][k], d[7][k], d[8][k], d[9][k],
                        d[10][k], d[11][k], t, u, v, abs_det);
                    if (pre && valid && !(ANY && found) && t <= abs_det * ray.tmax) {
                        const float inv_det = 1.0f / abs_det;
                        const float th = t * inv_det;
                        store_hit(hits, ray_id, pid[k] & 0x7FFFFFFF, th, u * inv_det, v * inv_det);
                        ray.tmax = th; found = true;
                    }
                }
                const bool leave = pid[3] < 0;                                // last packet of the leaf (mapping_cpu.impala:39)
                top = (ANY && found) ? 0 : (leave ? popped : top - 1);        // top - 1 == ~(j + 1)
                sp -= (leave && !(ANY && found)) ? kWave : 0;
            }
        }
    }
}

template <bool ANY, int N, int LDS_N, int XCD>
__global__ __launch_bounds__(kWave) void k_wide_single(const char* __restrict__ nodes, const Tri4* __restrict__ tris,
                                                        const Ray1* __restrict__ rays, Hit1* __restrict__ hits, int n,
                                                        Ctl* ctl, int* __restrict__ deep_list, int grid_w) {
    __shared__ int lds_raw[(LDS_N + N) * kWave];
    lds_int* col = (lds_int*)lds_raw + threadIdx.x;
    const int total_chunks = (n + kWave - 1) / kWave;
    int chunk = blockIdx.x;
    if (grid_w < 0) grid_w = detect_ray_grid(rays, n);
    grid_w = __builtin_amdgcn_readfirstlane(grid_w > 0 && (grid_w & 7) == 0 ? grid_w : 0);
    if (XCD > 0) {                                                            // as k_bvh2_single
        const int span = 8 * XCD, full = (total_chunks / span) * span;
        if ((int)blockIdx.x < full) {
            const int x = blockIdx.x % 8, l = blockIdx.x / 8;
            chunk = ((l / XCD) * 8 + x) * XCD + l % XCD;
        }
    }
    wide_chunk<ANY, N, LDS_N>(nodes, tris, rays, hits, n, ctl, deep_list, col, chunk * kWave, nullptr, 1, grid_w);
}

// ---------------------------------------------------------------------------------------------
// Persistent form with the top of the hierarchy in LDS (variant "top"; the wide counterpart of k_bvh2_top_persist).
// A wide tree is shallow: the root, its children and their children are 1 + 8 + 64 = 73 Node8 records (18.7 KB) or, one level
// further, 1 + 4 + 16 + 64 = 85 Node4 records (10.9 KB).  Every workgroup stages them ITSELF from the caller's array when it
// starts -- one dependent load per level, three or four in all -- so there is no image to keep between launches and nothing
// to validate (the BVH2 kernel's 255 records are eight levels deep: it keeps a validated image per context instead).
// The grid is one resident generation of 16-wave workgroups -- one per CU: sixteen (LDS_N + N)-row stack windows and the
// records are 147 KB (BVH8) of a CU's 160 KB, and ~100 VGPRs allow four waves per SIMD anyway -- whose waves draw 64-ray
// chunks from the striped ticket counters of k_bvh2_top_persist.
// ---------------------------------------------------------------------------------------------
template <int N> struct WideTop {                    // records: levels 0..2 (BVH8), 0..3 (BVH4)
    static constexpr int kRecords = N == 8 ? 1 + 8 + 64 : 1 + 4 + 16 + 64;
};

// Run by the workgroup's first wave: breadth first from the root; a child that gets a record is rewritten to a link.
template <int N>
__device__ __forceinline__ void stage_wide_top(const char* __restrict__ nodes, lds_int* image, lds_int* slot_node /* [kRecords] */) {
    typedef WideLayout<N> L;
    constexpr int kRecords = WideTop<N>::kRecords, kNodeInts = (int)L::kNodeBytes / 4;
    const int lane = threadIdx.x;                                            // 0..63
    if (lane == 0) slot_node[0] = 1;
    wave_lds_sync();
    int begin = 0, end = 1;
    while (begin < end) {
        int next = end;
        for (int first = begin; first < end; first += kWave) {
            const int slot = first + lane;
            const bool on = slot < end;
            const int id = on ? slot_node[slot] : 1;
            const int* src = reinterpret_cast<const int*>(nodes + (size_t)(id - 1) * L::kNodeBytes);
            int child[N];
#pragma unroll
            for (int k = 0; k < N; k++) child[k] = on ? src[6 * N + k] : 0;
#pragma unroll
            for (int k = 0; k < N; k++) {
                const bool inner = child[k] > 0;
                const unsigned long long m = __ballot(inner);
                const int s = next + __popcll(m & ((1ull << lane) - 1ull));
                if (inner && s < kRecords) { slot_node[s] = child[k]; child[k] = kLdsTag + s * (int)L::kNodeBytes; }
                next = min(kRecords, next + __popcll(m));
            }
            if (on) {
                lds_int* rec = image + slot * kNodeInts;
                for (int j = 0; j < 6 * N; j++) rec[j] = src[j];
#pragma unroll
                for (int k = 0; k < N; k++) { rec[6 * N + k] = child[k]; rec[7 * N + k] = 0; }
            }
        }
        wave_lds_sync();
        begin = end; end = next;
    }
}

template <bool ANY, int N, int LDS_N, int WAVES>
__global__ __launch_bounds__(kWave * WAVES) void k_wide_top_persist(const char* __restrict__ nodes, const Tri4* __restrict__ tris,
                                                                   const Ray1* __restrict__ rays, Hit1* __restrict__ hits, int n,
                                                                   Ctl* ctl, int* __restrict__ deep_list, int* __restrict__ tickets,
                                                                       int grid_w) {
    constexpr int kStackInts = WAVES * (LDS_N + N) * kWave, kImageInts = WideTop<N>::kRecords * (int)WideLayout<N>::kNodeBytes / 4,
        kGroup = 32;
    static_assert((kStackInts + kImageInts + WideTop<N>::kRecords) * 4 <= 160 * 1024,
        "one workgroup per CU must fit its stacks and records in LDS");
    __shared__ __attribute__((aligned(16))) int lds_raw[kStackInts + kImageInts + WideTop<N>::kRecords];
    const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
    lds_int* col = (lds_int*)lds_raw + wave * (LDS_N + N) * kWave + lane;
    lds_int* image = (lds_int*)lds_raw + kStackInts;
    if (wave == 0) stage_wide_top<N>(nodes, image, image + kImageInts);
    __syncthreads();
    const int total_chunks = (n + kWave - 1) / kWave, stripe = blockIdx.x % kStripes, stripe_waves = (gridDim.x / kStripes) * WAVES;
    int* counter = tickets + stripe * kCounterStride;
    // a wave's first ticket: its rank inside the stripe (wave-major, traversal_top.h)
    int t = stripe_rank(wave);
    if (grid_w < 0) grid_w = detect_ray_grid(rays, n);
    grid_w = __builtin_amdgcn_readfirstlane(grid_w > 0 && (grid_w & 7) == 0 ? grid_w : 0);
    for (;;) {
        const int group_first = ((t / kGroup) * kStripes + stripe) * kGroup, chunk = group_first + t % kGroup;
        if (group_first >= total_chunks) break;                              // this stripe's share is used up
        if (chunk < total_chunks) wide_chunk<ANY, N, LDS_N, true>(nodes, tris, rays, hits, n, ctl, deep_list, col, chunk * kWave, image,
            kLdsTag, grid_w);
        int t_next = 0;
        if (lane == 0) t_next = atomicAdd(counter, 1);
        t = stripe_waves + __builtin_amdgcn_readfirstlane(t_next);
    }
}

// The reference's general-arity loop, literally, for one ray (mapping_gpu.impala:136-178): the follow-up kernel's body
// and the lab build's "lane" kernel.
template <bool ANY, int N, typename Stack>
__device__ __forceinline__ HitAcc wide_ray_literal(const char* __restrict__ nodes, const Tri4* __restrict__ tris, RayX ray, Stack& st) {
    HitAcc hit{-1, ray.tmax, 0.0f, 0.0f};
    int ptr = 0, top = 1; st.put(0, 0);
    while (top != 0) {
        const float4* p = reinterpret_cast<const float4*>(nodes + (size_t)(top - 1) * WideLayout<N>::kNodeBytes);
        top = st.get(ptr); ptr--;                                             // pop (:138)
        float tnear = ray.tmax;
#pragma unroll
        for (int q = 0; q < N / 4; q++) {
            const float4 lx = p[0 * (N / 4) + q], hx = p[1 * (N / 4) + q], ly = p[2 * (N / 4) + q], hy = p[3 * (N / 4) + q],
                lz = p[4 * (N / 4) + q], hz = p[5 * (N / 4) + q];
            const int4 ch = *reinterpret_cast<const int4*>(p + 6 * (N / 4) + q);
            const float blx[4] = {lx.x, lx.y, lx.z, lx.w}, bhx[4] = {hx.x, hx.y, hx.z, hx.w};
            const float bly[4] = {ly.x, ly.y, ly.z, ly.w}, bhy[4] = {hy.x, hy.y, hy.z, hy.w};
            const float blz[4] = {lz.x, lz.y, lz.z, lz.w}, bhz[4] = {hz.x, hz.y, hz.z, hz.w};
            const int   chi[4] = {ch.x, ch.y, ch.z, ch.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float te;
                if (slab(ray, blx[k], bhx[k], bly[k], bhy[k], blz[k], bhz[k], te) && chi[k] != 0) {
                    if (ANY || te < tnear) { st.put(++ptr, top); top = chi[k]; tnear = te; }   // push       (:145-147)
                    else st.put(++ptr, chi[k]);                                                   // push_after (:149)
                }
            }
        }
        while (top < 0) {
            int j = ~top; top = st.get(ptr); ptr--;
            for (;;) {
                const float4* tp = reinterpret_cast<const float4*>(tris + j++);
                const int4 pid = *reinterpret_cast<const int4*>(tp + 12);
                const int ids[4] = {pid.x, pid.y, pid.z, pid.w};
                float q[12][4];
#pragma unroll
                for (int r = 0; r < 12; r++) { const float4 x = tp[r]; q[r][0] = x.x; q[r][1] = x.y; q[r][2] = x.z; q[r][3] = x.w; }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (ids[k] == -1) break;                                  // is_valid (mapping_cpu.impala:38)
                    float t, u, v;
                    if (intersect_tri(ray, q[0][k], q[1][k], q[2][k], q[3][k], q[4][k], q[5][k], q[6][k], q[7][k], q[8][k], q[9][k],
                        q[10][k], q[11][k], t, u, v)) {
                        hit.id = ids[k] & 0x7FFFFFFF; hit.t = t; hit.u = u; hit.v = v;
                        ray.tmax = t;
                        if (ANY) return hit;
                    }
                }
                if (pid.w < 0) break;                                         // is_last (mapping_cpu.impala:39)
            }
        }
    }
    return hit;
}

// kFinishGroups one-wave workgroups, each taking every kFinishGroups-th batch of 64 deep rays (one wave until round 4: a launch with many
// deep rays waited for a serial drain); the last workgroup to finish resets the launch's control words (finish_launch's protocol).
template <bool ANY, int N>
__global__ __launch_bounds__(kWave) void k_wide_finish(const char* __restrict__ nodes, const Tri4* __restrict__ tris,
                                                        const Ray1* __restrict__ rays, Hit1* __restrict__ hits,
                                                        Ctl* ctl, const int* __restrict__ deep_list, int* deep_stack, int* tickets) {
    __shared__ int stack_lds[kStackCap * kWave];                    // the reference's 64 entries per lane, in LDS (see DeepStack)
    // the persistent form's ticket counters, ready for the next launch
    if (tickets && blockIdx.x == 0) for (int k = threadIdx.x; k < 4 * 64; k += kWave) tickets[k * 16] = 0;
    const int count = ctl->deep_count;
    if (count > 0) {
        DeepStack st{(lds_int*)stack_lds + threadIdx.x, &ctl->err};
        for (int k = blockIdx.x * kWave + threadIdx.x; k < count; k += gridDim.x * kWave) {
            const int i = deep_list[k];
            const HitAcc hit = wide_ray_literal<ANY, N>(nodes, tris, load_ray(rays, i), st);
            store_hit(hits, i, hit.id, hit.t, hit.u, hit.v);
        }
    }
    if (threadIdx.x == 0) {
        // (every workgroup has read deep_count before it counts itself done, so the last one may zero it; no deep rays -- the usual case
        // --: workgroup 0 rewrites the zeros)
        const bool last = gridDim.x == 1 || (count == 0 ? blockIdx.x == 0 : atomicAdd(&ctl->finish_done, 1) == (int)gridDim.x - 1);
        if (last) { ctl->stats[7] += (unsigned long long)count; ctl->counter = 0; ctl->deep_count = 0; ctl->finish_done = 0;
            report_error(ctl); }
    }
}

#define WIDE_LAUNCH_ARGS DeviceState& s, const void* nodes, const Tri4* tris, const Ray1* rays, Hit1* hits, int n, hipStream_t stream
template <bool ANY, int N, int LDS_N, int XCD> void L_wide_single(WIDE_LAUNCH_ARGS) {
    ensure_deep_list(s, n);
    hipLaunchKernelGGL((k_wide_single<ANY, N, LDS_N, XCD>), dim3(blocks_for(n)), dim3(kWave), 0, stream, (const char*)nodes, tris, rays,
        hits, n, s.ctl(), s.deep_list, n >= kGridMinRays ? g_ray_grid : 0);
    hipLaunchKernelGGL((k_wide_finish<ANY, N>), dim3(kFinishGroups), dim3(kWave), 0, stream, (const char*)nodes, tris, rays, hits, s.ctl(),
        s.deep_list, s.deep_stack, (int*)nullptr);
}
// "top": the persistent form with the staged top levels for launches that fill the chip (as the BVH2 default: rodent_hip_top_min_rays),
// the one-chunk kernel below that
int wide_top_min_rays();
template <bool ANY, int N, int LDS_N> void L_wide_top(WIDE_LAUNCH_ARGS) {
    if (n < wide_top_min_rays()) { L_wide_single<ANY, N, LDS_N, 32>(s, nodes, tris, rays, hits, n, stream); return; }
    ensure_deep_list(s, n);
    if (!s.tickets) {
        std::lock_guard<std::mutex> lock(g_mutex);
        if (!s.tickets) {
            HIP_CHECK(hipMalloc(&s.tickets, sizeof(int) * kMaxPhases * kStripes * kCounterStride));
            HIP_CHECK(hipMemset(s.tickets, 0, sizeof(int) * kMaxPhases * kStripes * kCounterStride));
        }
    }
    constexpr int kWaves = 16;
    const int groups = ((s.num_cus + kStripes - 1) / kStripes) * kStripes;   // one workgroup per CU, the same number in every stripe
    hipLaunchKernelGGL((k_wide_top_persist<ANY, N, LDS_N, kWaves>), dim3(groups), dim3(kWave * kWaves), 0, stream, (const char*)nodes,
        tris, rays, hits, n, s.ctl(), s.deep_list, s.tickets, g_ray_grid);
    hipLaunchKernelGGL((k_wide_finish<ANY, N>), dim3(kFinishGroups), dim3(kWave), 0, stream, (const char*)nodes, tris, rays, hits, s.ctl(),
        s.deep_list, s.deep_stack, s.tickets);
}

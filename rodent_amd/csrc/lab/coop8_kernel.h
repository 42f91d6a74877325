// lab/coop8_kernel.h -- LAB BUILD ONLY: wave-cooperative BVH8 traversal for any-hit rays (north_star's "BVH8 node
// traversal").
//
// One ray per OCTET of lanes, eight rays per wavefront.  A node step: lane c of the octet fetches child c's six bounds and its id from the
// ray's current Node8 (the octet's seven dword loads cover the node's 224 used bytes in 32-byte runs), tests ONE box, and the octet's byte
// of the wave's ballot is the set of hit children: the highest hit slot becomes the next node, the others go onto the octet's stack in slot
// order -- exactly what the one-ray-per-lane kernel's any-hit branch leaves behind (wide_chunk, ANY: every hit child displaces the one
// before it; mapping_gpu.impala:136-153), so the Hit1 records are the oracle's (B1g), not only the occlusion answers.  A leaf step: lanes
// 0..3 of the octet test one triangle of the Tri4 packet each; the first valid accepted slot stores the record and ends the ray
// (mapping_gpu.impala:160-169). The stack is one column of LDS per octet (STACK entries); a ray that outgrows it goes to the launch's deep
// list (k_wide_finish). Closest-hit launches of this variant take the one-ray-per-lane kernel (the nearest-first order is a sequential scan
// over the slots).
#pragma once

template <int STACK, int WAVES>
__global__ __launch_bounds__(kWave * WAVES) void k_wide8_coop(const char* __restrict__ nodes, const Tri4* __restrict__ tris,
                                                               const Ray1* __restrict__ rays, Hit1* __restrict__ hits, int n,
                                                               Ctl* ctl, int* __restrict__ deep_list) {
    constexpr int kOctets = 8 * WAVES;
    __shared__ int stack_lds[kOctets * STACK];
    const int lane = threadIdx.x % kWave, c = lane & 7, octet = (int)threadIdx.x >> 3;
    const int ray_id = (int)blockIdx.x * kOctets + octet;
    lds_int* const col = (lds_int*)stack_lds + octet * STACK;
    const bool valid_ray = ray_id < n;
    RayX ray = load_ray(rays, valid_ray ? ray_id : 0);
    if (valid_ray && c == 0) store_hit(hits, ray_id, -1, ray.tmax, 0.0f, 0.0f);
    ray.tmin = canonical(ray.tmin); ray.tmax = canonical(ray.tmax);
    // sp: index of the top entry in memory (col[0] = 0 ends the traversal)
    int top = valid_ray ? 1 : 0, sp = 0;
    if (c == 0) col[0] = 0;
    typedef const __attribute__((address_space(1))) char* gptr;
    const gptr node_base = (gptr)(nodes - sizeof(Node8)), tri_base = (gptr)reinterpret_cast<const char*>(tris);
    const int shift = lane & ~7;
    while (__ballot(top != 0)) {
        if (top != 0) {
            const int popped = col[sp];
            if (top > 0) {
                const __attribute__((address_space(1))) float* p =
                    (const __attribute__((address_space(1))) float*)(node_base + (size_t)(unsigned)top * sizeof(Node8)) + c;
                const float lox = p[0], hix = p[8], loy = p[16], hiy = p[24], loz = p[32], hiz = p[40];
                const int child = __float_as_int(p[48]);
                float te;
                const bool hit = slab_canonical(ray, lox, hix, loy, hiy, loz, hiz, te) && child != 0;
                // (lanes of an octet step together: its byte of the ballot is complete)
                const unsigned m = (unsigned)(__ballot(hit) >> shift) & 0xFFu;
                if (m == 0u) { top = popped; sp -= 1; }
                else {
                    const int highest = 31 - __builtin_clz(m), below = __popc(m & ((1u << c) - 1u)), count = __popc(m);
                    // the octet's stack is full: k_wide_finish redoes this ray
                    if (sp + count - 1 >= STACK) {
                        if (c == 0) deep_list[atomicAdd(&ctl->deep_count, 1)] = ray_id;
                        top = 0;
                    } else {
                        if (hit && c != highest) col[sp + 1 + below] = child;
                        top = __shfl(child, shift + highest);
                        sp += count - 1;
                    }
                }
            } else {
                const int k = c & 3;
                const __attribute__((address_space(1))) float* p =
                    (const __attribute__((address_space(1))) float*)(tri_base + (size_t)(unsigned)~top * sizeof(Tri4)) + k;
                const float v0x = p[0], v0y = p[4], v0z = p[8], e1x = p[12], e1y = p[16], e1z = p[20], e2x = p[24], e2y = p[28],
                    e2z = p[32], nx = p[36], ny = p[40], nz = p[44];
                const int pid = __float_as_int(p[48]);
                float t, u, v, abs_det;
                const bool pre = tri_pre(ray, v0x, v0y, v0z, e1x, e1y, e1z, e2x, e2y, e2z, nx, ny, nz, t, u, v, abs_det);
                const bool accept = c < 4 && pre && t <= abs_det * ray.tmax;
                // an unused slot ends the packet (mapping_cpu.impala:38)
                const unsigned unused = (unsigned)(__ballot(c < 4 && pid == -1) >> shift) & 0xFu;
                // slots in front of the first unused one
                const unsigned ok = unused ? ((1u << __builtin_ctz(unused)) - 1u) : 0xFu;
                const unsigned acc = (unsigned)(__ballot(accept) >> shift) & ok;
                // last packet of the leaf (mapping_cpu.impala:39)
                const bool leave = __shfl(pid, shift + 3) < 0;
                if (acc) {
                    if (c == __builtin_ctz(acc)) { const float inv_det = 1.0f / abs_det;
                        store_hit(hits, ray_id, pid & 0x7FFFFFFF, t * inv_det, u * inv_det, v * inv_det); }
                    top = 0;
                } else if (leave) { top = popped; sp -= 1; }
                else top -= 1;
            }
        }
    }
}

template <bool ANY, int N, int STACK> void L_wide8_coop(WIDE_LAUNCH_ARGS) {
    static_assert(N == 8, "Node8 only");
    if (!ANY) { L_wide_single<false, 8, 24, 32>(s, nodes, tris, rays, hits, n, stream); return; }
    ensure_deep_list(s, n);
    constexpr int kWaves = 4, kRays = 8 * kWaves;
    hipLaunchKernelGGL((k_wide8_coop<STACK, kWaves>), dim3((n + kRays - 1) / kRays), dim3(kWave * kWaves), 0, stream, (const char*)nodes,
        tris, rays, hits, n, s.ctl(), s.deep_list);
    hipLaunchKernelGGL((k_wide_finish<true, 8>), dim3(kFinishGroups), dim3(kWave), 0, stream, (const char*)nodes, tris, rays, hits,
        s.ctl(), s.deep_list, s.deep_stack, (int*)nullptr);
}

// vmem_peak.hip -- lab microbenchmark: what the vector-memory pipeline (TA -> TCP/L1 -> L2) of gfx950 sustains for the
// access pattern of a BVH2 step: every active lane fetches one 64-byte node as 3 x dwordx4 + 1 x dwordx2 from
// base + index * 64, indices either coherent (neighbouring lanes share nodes) or scattered over a working set that
// fits L1 (16 KiB), L2 (2 MiB) or only the Infinity Cache (64 MiB), with 100 % / 50 % / 25 % of the lanes active, and the
// same bytes fetched "cooperatively" (4 lanes x 16 B per node: a quarter of the cache-line look-ups per instruction).
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/vmem_peak.hip -o rodent_amd/bin/vmem_peak
// Output: node fetches per ns per chip, wave-level load instructions per us per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// MODE 0: lane-per-node (the traversal kernels' pattern); MODE 1: 4 lanes per node, 16 B each (4 instructions cover the
// wave's 64 nodes 16 at a time), data NOT redistributed (pipeline cost only)
template <int MODE>
__global__ __launch_bounds__(64) void k(const char* base, const unsigned* idx, int steps, unsigned active_mask_kind, float* out) {
    const unsigned lane = threadIdx.x;
    const bool active = active_mask_kind == 0 ? true : active_mask_kind == 1 ? (lane & 1) == 0 : (lane & 3) == 0;
    const unsigned* my = idx + ((size_t)blockIdx.x * steps) * 64;
    float acc = 0.0f;
    if (active) {
        for (int s = 0; s < steps; s++) {
            if (MODE == 0) {
                const unsigned i = my[s * 64 + lane];
                const char* p = base + (size_t)i * 64;
                f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 16), c = *(const f32x4*)(p + 32);
                f32x2 d = *(const f32x2*)(p + 48);
                asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
                acc += a.x + b.y + c.z + d.x;
            } else {
                f32x4 q[4];
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const unsigned i = my[s * 64 + g * 16 + lane / 4];
                    q[g] = *(const f32x4*)(base + (size_t)i * 64 + (lane & 3) * 16);
                }
                asm volatile("" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]));
                acc += q[0].x + q[1].y + q[2].z + q[3].w;
            }
        }
    }
    out[blockIdx.x * 64 + lane] = acc;
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, waves = cus * 32, steps = 256;
    printf("%s: %d CUs; %d waves x %d steps; one step = one 64-byte node per active lane\n", prop.name, cus, waves, steps);
    const size_t max_nodes = (64u << 20) / 64;
    char* base; CHECK(hipMalloc(&base, max_nodes * 64)); CHECK(hipMemset(base, 0, max_nodes * 64));
    unsigned* idx; CHECK(hipMalloc(&idx, sizeof(unsigned) * (size_t)waves * steps * 64));
    float* out; CHECK(hipMalloc(&out, sizeof(float) * waves * 64));
    std::vector<unsigned> h((size_t)waves * steps * 64);
    std::mt19937 rng(1);
    struct Pattern { const char* name; size_t nodes; int coherent; };
    const Pattern patterns[] = {{"coherent (4 lanes share a node), 2 MiB set", (2u << 20) / 64, 4}, {"coherent (16 lanes share a node), 2 MiB set", (2u << 20) / 64, 16},
                                {"scattered, 16 KiB set (L1)", (16u << 10) / 64, 1}, {"scattered, 2 MiB set (L2)", (2u << 20) / 64, 1},
                                {"scattered, 64 MiB set (MALL)", max_nodes, 1}};
    for (const Pattern& p : patterns) {
        for (size_t w = 0; w < (size_t)waves * steps; w++) {
            unsigned cur = 0;
            for (int l = 0; l < 64; l++) { if (l % p.coherent == 0) cur = rng() % p.nodes; h[w * 64 + l] = cur; }
        }
        CHECK(hipMemcpy(idx, h.data(), h.size() * sizeof(unsigned), hipMemcpyHostToDevice));
        for (int mode = 0; mode < 2; mode++)
            for (unsigned mask = 0; mask < 3; mask++) {
                if (mode == 1 && mask != 0) continue;
                hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
                auto launch = [&]() { if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(waves), dim3(64), 0, 0, base, idx, steps, mask, out); else hipLaunchKernelGGL(k<1>, dim3(waves), dim3(64), 0, 0, base, idx, steps, mask, out); };
                launch(); CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
                const double lanes = mask == 0 ? 64 : mask == 1 ? 32 : 16;
                const double fetches = (double)waves * steps * lanes, instr = (double)waves * steps * 4;
                printf("%-46s %-18s lanes %3.0f%%: %8.3f ms  %7.2f node fetches/ns  %7.1f load instr/us/CU  %6.1f cycles/instr/CU at 2.4 GHz\n", p.name, mode == 0 ? "lane-per-node" : "4-lanes-per-node",
                       lanes / 64 * 100, ms, fetches / (ms * 1e6), instr / (ms * 1e3) / cus, 2400.0 / (instr / (ms * 1e3) / cus));
            }
    }
    return 0;
}

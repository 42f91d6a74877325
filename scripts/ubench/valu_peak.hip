// valu_peak.hip -- lab microbenchmark: VALU issue rate of gfx950 for the instruction kinds the traversal loop is made
// of (the denominator of bench.py's "valu issue" fraction).  Not part of the product library.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/valu_peak.hip -o rodent_amd/bin/valu_peak && rodent_amd/bin/valu_peak
// Every kernel runs ITER iterations of 64 instructions in 8 independent chains (or 1 dependent chain) per lane;
// grid = 256 CUs x waves_per_simd x 4 wave64 workgroups.  Reported: wave-instructions per cycle per SIMD at the
// clock measured by s_memtime over the same kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

enum Kind { FMA = 0, PK_FMA, MINMAX, MAX3, CNDMASK, MUL, ADD_U32, MAD_U64, FMA_DEP, MIX };
static const char* kNames[] = {"v_fma_f32", "v_pk_fma_f32", "v_min_f32/v_max_f32", "v_max3_f32", "v_cndmask_b32", "v_mul_f32", "v_add_u32", "v_mad_u64_u32",
                               "v_fma_f32 (one dependent chain)", "mix fma/min/max/cndmask/cmp"};
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(64) void k(float* out, unsigned long long* cycles, int iters) {
    float a[8]; f32x2 p[8]; unsigned u[8]; unsigned long long w[4];
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 0.001f + i; p[i] = (f32x2){a[i], a[i] + 1.0f}; u[i] = threadIdx.x + i; }
    for (int i = 0; i < 4; i++) w[i] = threadIdx.x + i;
    const float b = 1.0000001f, c = 1e-9f; const f32x2 pb = {b, b}, pc = {c, c};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (KIND == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
                if (KIND == MINMAX) { if (i & 1) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); else asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c)); }
                if (KIND == MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : );
                if (KIND == MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (KIND == MAD_U64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i & 3]) : "v"(u[i]), "v"(u[(i + 1) & 7]) : "vcc");
                if (KIND == FMA_DEP) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
                if (KIND == MIX) {
                    switch (i & 3) {
                        case 0: asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); break;
                        case 1: asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); break;
                        case 2: asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc"); break;
                        case 3: asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b)); break;
                    }
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y + (float)u[i];
    for (int i = 0; i < 4; i++) s += (float)w[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int KIND> void run(int waves_per_simd, float* out, unsigned long long* cyc, int cus) {
    const int iters = 2048, blocks = cus * 4 * waves_per_simd;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, out, cyc, 64);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks);
    CHECK(hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double mean = 0; for (auto c : h) mean += (double)c; mean /= blocks;
    const double instr = (double)iters * 64;                       // per wave
    // s_memtime ticks at 100 MHz on this part (constant clock): wave time in us = ticks / 100
    const double wave_us = mean / 100.0;
    const double total_instr = instr * blocks;
    const double per_simd_per_us = total_instr / (cus * 4.0) / (ms * 1e3);
    printf("%-34s waves/SIMD %d: %8.3f ms  %7.1f wave-instr/us/SIMD (kernel)  %7.1f wave-instr/us per wave (in-wave timer)\n", kNames[KIND], waves_per_simd, ms,
           per_simd_per_us, instr / wave_us);
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("%s: %d CUs, clock %d MHz\n", prop.name, cus, prop.clockRate / 1000);
    float* out; unsigned long long* cyc;
    CHECK(hipMalloc(&out, sizeof(float) * 64 * cus * 4 * 8)); CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * cus * 4 * 8));
    for (int w : {1, 2, 4, 8}) {
        run<FMA>(w, out, cyc, cus); run<PK_FMA>(w, out, cyc, cus); run<MINMAX>(w, out, cyc, cus); run<MAX3>(w, out, cyc, cus); run<CNDMASK>(w, out, cyc, cus);
        run<MUL>(w, out, cyc, cus); run<ADD_U32>(w, out, cyc, cus); run<MAD_U64>(w, out, cyc, cus); run<FMA_DEP>(w, out, cyc, cus); run<MIX>(w, out, cyc, cus);
    }
    return 0;
}

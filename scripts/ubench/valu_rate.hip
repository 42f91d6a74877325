// valu_rate.hip -- lab microbenchmark, round 4: reconciles the VALU issue ceiling with the guide's table (MI355X_MICROARCH.md: v_fma_f32 on a
// wave64 = 2 cycles on a SIMD-32, i.e. 1200 wave-instructions per us per SIMD at 2.4 GHz; round 2's valu_peak measured 591 for `v_fma_f32 v, v, v, v`).
// What it separates: (1) the SHADER CLOCK under the test itself (s_memtime counts shader cycles, s_memrealtime a constant 100 MHz: their ratio over the
// timed loop), (2) operand traffic: the 3-VGPR-source VOP3 form against v_fmac_f32 (two VGPR sources + the accumulator) and against forms with an
// inline constant, (3) v_cndmask_b32 with its mask in an SGPR pair (round 2's row read VCC right after a VALU wrote it: a hazard, not a rate),
// (4) the traversal loop's own mix.  Every kernel: 64 instructions per iteration in 8 independent chains per lane; grid = CUs x 4 SIMDs x waves.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/valu_rate.hip -o rodent_amd/bin/valu_rate && rodent_amd/bin/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

enum Kind { FMA3 = 0, FMAC, FMA_CONST, MUL, PK_FMA, PK_FMA_SAMEBC, MINMAX, MAX3, CNDMASK_SGPR, CMP, MIX_LOOP, NUM_KINDS };
static const char* kNames[] = {"v_fma_f32 v, v, v, v (3 VGPR sources)", "v_fmac_f32 v, v, v (VOP2)", "v_fma_f32 v, v, v, 1.0 (inline constant)", "v_mul_f32 v, v, v",
                               "v_pk_fma_f32 (3 x 64-bit VGPR sources)", "v_pk_fma_f32 v, v, v(b), v(b)", "v_min_f32 / v_max_f32", "v_max3_f32",
                               "v_cndmask_b32 v, v, v, s[mask]", "v_cmp_lt_f32 s[x:y], v, v", "the loop's mix: 6 pk_fma + 12 min/max/max3 + 2 cmp + 6 cndmask + 6 other per 32"};
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(64) void k(float* out, unsigned long long* ticks, int iters) {
    float a[8]; f32x2 p[8];
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 0.001f + i; p[i] = (f32x2){a[i], a[i] + 1.0f}; }
    float b = 1.0000001f, c = 1e-9f; f32x2 pb = {b, b}, pc = {c, c};
    asm volatile("" : "+v"(b), "+v"(c), "+v"(pb), "+v"(pc));
    unsigned long long mask = 0x5555555555555555ull;
    asm volatile("" : "+s"(mask));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (KIND == FMA3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == FMAC) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == FMA_CONST) asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(a[i]) : "v"(b));
                if (KIND == MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
                if (KIND == PK_FMA_SAMEBC) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(pb));
                if (KIND == MINMAX) { if (i & 1) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); else asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c)); }
                if (KIND == MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == CNDMASK_SGPR) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(mask));
                if (KIND == CMP) { unsigned long long m; asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(m) : "v"(a[i]), "v"(b)); asm volatile("" :: "s"(m)); }
                if (KIND == MIX_LOOP) {
                    // 32 slots (i + 8 * (r & 3)): 6 pk_fma, 8 min/max, 4 max3, 2 cmp, 6 cndmask, 3 mul, 3 add
                    const int s = i + 8 * (r & 3);
                    if (s < 6) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
                    else if (s < 14) { if (s & 1) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); else asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c)); }
                    else if (s < 18) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                    else if (s < 20) { unsigned long long m; asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(m) : "v"(a[i]), "v"(b)); asm volatile("" :: "s"(m)); }
                    else if (s < 26) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(mask));
                    else if (s < 29) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                    else asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0; for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) { ticks[2 * blockIdx.x] = t1 - t0; ticks[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int KIND> void run(int waves_per_simd, float* out, unsigned long long* tk, int cus) {
    const int iters = 4096, blocks = cus * 4 * waves_per_simd;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, out, tk, 64);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, out, tk, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(2 * blocks);
    CHECK(hipMemcpy(h.data(), tk, 2 * blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double cyc = 0, real = 0; for (int b = 0; b < blocks; b++) { cyc += (double)h[2 * b]; real += (double)h[2 * b + 1]; }
    const double mhz = cyc / real * 100.0;                             // shader cycles per 10 ns tick
    const double instr = (double)iters * 64;                            // per wave
    const double per_simd_per_us = instr * blocks / (cus * 4.0) / (ms * 1e3);
    const double cycles_per_instr_simd = (cyc / blocks) / (instr * waves_per_simd);      // SIMD cycles per wave-instruction while the waves run
    printf("%-92s waves/SIMD %d: %8.3f ms  clock %5.0f MHz  %7.1f wave-instr/us/SIMD  %5.2f cycles per wave-instruction on the SIMD  (guide: 2 cycles = %4.0f /us/SIMD at this clock)\n",
           kNames[KIND], waves_per_simd, ms, mhz, per_simd_per_us, cycles_per_instr_simd, mhz / 2.0);
}

template <int KIND> void all(float* out, unsigned long long* tk, int cus) { for (int w : {1, 2, 4, 8}) run<KIND>(w, out, tk, cus); }

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("%s: %d CUs, clock (properties) %d MHz\n", prop.name, cus, prop.clockRate / 1000);
    float* out; unsigned long long* tk;
    CHECK(hipMalloc(&out, sizeof(float) * 64 * cus * 4 * 8)); CHECK(hipMalloc(&tk, sizeof(unsigned long long) * 2 * cus * 4 * 8));
    all<FMA3>(out, tk, cus); all<FMAC>(out, tk, cus); all<FMA_CONST>(out, tk, cus); all<MUL>(out, tk, cus); all<PK_FMA>(out, tk, cus); all<PK_FMA_SAMEBC>(out, tk, cus);
    all<MINMAX>(out, tk, cus); all<MAX3>(out, tk, cus); all<CNDMASK_SGPR>(out, tk, cus); all<CMP>(out, tk, cus); all<MIX_LOOP>(out, tk, cus);
    return 0;
}

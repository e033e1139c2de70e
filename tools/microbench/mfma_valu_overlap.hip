// Do fp32 matrix-core instructions and vector-ALU instructions of one SIMD overlap on gfx950?  (Round 5: the sphere field on the
// matrix cores -- fused_step.inc SPHM -- measured as if its v_mfma_f32_16x16x4_f32 and its v_exp_f32 took turns.)
// Four waves per SIMD (1024-thread workgroups, one per CU); every wave runs `iters` trips of:
//   mode 0: 2 MFMA 16x16x4 f32 (independent accumulators)         mode 1: 8 v_exp_f32 + 4 v_pk_add_f32
//   mode 2: both, the MFMAs first (what the kernel does)          mode 3: two waves of every SIMD mode 0, the other two mode 1 (waves go to SIMDs round-robin: wave w -> SIMD w & 3)
// s_memtime around the loop, the slowest wave of workgroup 0 reported as cycles per trip.
// hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4 __attribute__((ext_vector_type(4)));
typedef float v2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(1024) k(unsigned long long* out, int iters, float seed) {
    const int wave = threadIdx.x >> 6;
    v4 c0 = {seed, seed + 1.f, seed + 2.f, seed + 3.f}, c1 = c0 + 1.f, d0 = c0, d1 = c1;
    float a = seed * 1e-3f + threadIdx.x * 1e-6f, b = a + 0.5f;
    float e0 = a, e1 = a + 1, e2 = a + 2, e3 = a + 3, e4 = a + 4, e5 = a + 5, e6 = a + 6, e7 = a + 7;
    v2 s0 = {0, 0}, s1 = {0, 0};
    const bool do_m = MODE == 0 || MODE == 2 || MODE == 5 || ((MODE == 3 || MODE == 6) && ((wave >> 2) & 1) == 0);
    const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && ((wave >> 2) & 1) == 1);
    const bool do_p = MODE == 4 || MODE == 5 || (MODE == 6 && ((wave >> 2) & 1) == 1);      // plain (non-transcendental) vector work: 20 v_pk_fma_f32
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int i = 0; i < iters; ++i) {
        if (do_m) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %2, %3, %4\n v_mfma_f32_16x16x4_f32 %1, %2, %3, %5"
                         : "=&v"(d0), "=&v"(d1) : "v"(a), "v"(b), "v"(c0), "v"(c1));
        }
        if (do_v) {
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                         "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                         "v_pk_add_f32 %8, %8, %9\n v_pk_add_f32 %9, %9, %8\n v_pk_add_f32 %8, %8, %9\n v_pk_add_f32 %9, %9, %8\n"
                         : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7), "+v"(s0), "+v"(s1));
        }
        if (do_p) {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %1, %1, %0, %0\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %1, %1, %0, %0\n"
                         "v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %1, %1, %0, %0\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %1, %1, %0, %0\n"
                         "v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %1, %1, %0, %0\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %1, %1, %0, %0\n"
                         "v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %1, %1, %0, %0\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %1, %1, %0, %0\n"
                         "v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %1, %1, %0, %0\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %1, %1, %0, %0\n"
                         : "+v"(s0), "+v"(s1));
        }
    }
    asm volatile("s_nop 15\n s_nop 15\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));
    float sink = d0[0] + d1[3] + e0 + e1 + e2 + e3 + e4 + e5 + e6 + e7 + s0.x + s1.y;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) out[wave] = t1 - t0;
    if (sink == 12345.678f) out[63] = 1;
}

template <int MODE> double run(unsigned long long* d, int iters) {
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, d, iters, 1.0f);      // warm-up
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, d, iters, 1.0f);
    hipDeviceSynchronize();
    unsigned long long h[64];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long m = 0;
    for (int w = 0; w < 16; ++w) m = h[w] > m ? h[w] : m;
    return (double)m / iters;
}

int main() {
    unsigned long long* d;
    hipMalloc(&d, 64 * sizeof(unsigned long long));
    const int iters = 20000;
    // (s_memtime counts at 100 MHz on gfx9: cycles below are in ITS units; the ratios are what matters)
    printf("per trip and wave, four waves per SIMD (s_memtime ticks):\n");
    const double m0 = run<0>(d, iters), m1 = run<1>(d, iters), m2 = run<2>(d, iters), m3 = run<3>(d, iters);
    printf("  2 x v_mfma_f32_16x16x4_f32 alone           %.4f\n", m0);
    printf("  8 x v_exp_f32 + 4 x v_pk_add_f32 alone     %.4f\n", m1);
    printf("  both in every wave                         %.4f   (sum %.4f, max %.4f)\n", m2, m0 + m1, m0 > m1 ? m0 : m1);
    printf("  MFMA waves beside exp waves (2 + 2 / SIMD) %.4f   (each kind alone at 2 waves / SIMD: %.4f / %.4f)\n", m3, m0 / 2, m1 / 2);
    const double m4 = run<4>(d, iters), m5 = run<5>(d, iters), m6 = run<6>(d, iters);
    printf("  20 x v_pk_fma_f32 alone                    %.4f\n", m4);
    printf("  MFMAs + pk_fmas in every wave              %.4f   (sum %.4f, max %.4f)\n", m5, m0 + m4, m0 > m4 ? m0 : m4);
    printf("  MFMA waves beside pk_fma waves (2 + 2)     %.4f   (each kind alone at 2 waves / SIMD: %.4f / %.4f)\n", m6, m0 / 2, m4 / 2);
    return 0;
}

// Does v_exp_f32 (transcendental) overlap with packed-fp32 VALU work on gfx950?  Inline-asm mixes,
// cycles per loop body per SIMD from s_memtime, at 1..8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP8(X) X X X X X X X X
#define PK4 "v_pk_fma_f32 %0, %8, %9, %0\n v_pk_fma_f32 %1, %9, %8, %1\n v_pk_fma_f32 %2, %8, %9, %2\n v_pk_fma_f32 %3, %9, %8, %3\n"
#define EX4 "v_exp_f32 %4, %10\n v_exp_f32 %5, %11\n v_exp_f32 %6, %10\n v_exp_f32 %7, %11\n"
#define EX2 "v_exp_f32 %4, %10\n v_exp_f32 %5, %11\n"
#define MIX42 "v_pk_fma_f32 %0, %8, %9, %0\n v_exp_f32 %4, %10\n v_pk_fma_f32 %1, %9, %8, %1\n v_pk_fma_f32 %2, %8, %9, %2\n v_exp_f32 %5, %11\n v_pk_fma_f32 %3, %9, %8, %3\n"
#define MIX44 "v_pk_fma_f32 %0, %8, %9, %0\n v_exp_f32 %4, %10\n v_pk_fma_f32 %1, %9, %8, %1\n v_exp_f32 %5, %11\n v_pk_fma_f32 %2, %8, %9, %2\n v_exp_f32 %6, %10\n v_pk_fma_f32 %3, %9, %8, %3\n v_exp_f32 %7, %11\n"
#define F4 "v_fma_f32 %4, %10, %11, %4\n v_fma_f32 %5, %11, %10, %5\n v_fma_f32 %6, %10, %11, %6\n v_fma_f32 %7, %11, %10, %7\n"
#define OPS : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(p), "v"(q), "v"(r), "v"(s)
template <int OP>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* clk, int iters) {
    f2 a = {1.f, 2.f}, b = a, c = a, d = a, p = {0.999f, 1.001f}, q = {1e-3f, -1e-3f};
    float e = 1.f, f = 2.f, g = 3.f, h = 4.f, r = -0.5f, s = -0.25f;
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP8(asm volatile(PK4 OPS);) }                  // 32 pk_fma
        if (OP == 1) { REP8(asm volatile(EX4 OPS);) }                  // 32 exp
        if (OP == 2) { REP8(asm volatile(MIX42 OPS);) }                // 32 pk_fma + 16 exp
        if (OP == 3) { REP8(asm volatile(MIX44 OPS);) }                // 32 pk_fma + 32 exp
        if (OP == 4) { REP8(asm volatile(F4 OPS);) }                   // 32 v_fma_f32
        if (OP == 5) { REP8(asm volatile(F4 EX2 OPS);) }               // 32 v_fma_f32 + 16 exp (hmm F4 writes e..h too)
        if (OP == 6) { REP8(asm volatile(PK4 F4 OPS);) }               // 32 pk_fma + 32 v_fma
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 4 + (threadIdx.x >> 6)] = c1 - c0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a.x + b.x + c.y + d.y + e + f + g + h;
}
template <int OP> void run(const char* name, int n_instr, float* d_out, unsigned long long* d_clk) {
    for (int bpc : {1, 2, 3, 4, 8}) {
        const int iters = 4000, blocks = 256 * bpc;
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, d_clk, 10);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, d_clk, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * 4);
        (void)hipMemcpy(h.data(), d_clk, blocks * 32, hipMemcpyDeviceToHost);
        double cyc = 0; for (auto v : h) cyc += v; cyc /= h.size();
        printf("%-28s waves/SIMD=%d: %7.1f cycles per body per SIMD (%5.2f per instr), %.3f ns per instr per SIMD\n", name, bpc,
               cyc / iters / bpc, cyc / iters / bpc / n_instr, ms * 1e6 / ((double)bpc * iters * n_instr));
    }
}
int main() {
    float* d_out; unsigned long long* d_clk;
    (void)hipMalloc(&d_out, sizeof(float) * 256 * 8 * 256); (void)hipMalloc(&d_clk, 8 * 256 * 8 * 4);
    run<0>("32 pk_fma", 32, d_out, d_clk);
    run<1>("32 exp", 32, d_out, d_clk);
    run<2>("32 pk_fma + 16 exp", 48, d_out, d_clk);
    run<3>("32 pk_fma + 32 exp", 64, d_out, d_clk);
    run<4>("32 v_fma", 32, d_out, d_clk);
    run<5>("32 v_fma + 16 exp", 48, d_out, d_clk);
    run<6>("32 pk_fma + 32 v_fma", 64, d_out, d_clk);
    return 0;
}

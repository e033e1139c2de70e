// Issue-rate micro-benchmark with inline asm (the compiler cannot re-vectorise these):
// ns per wave64 instruction per SIMD for several VALU forms at 4 and 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X X X X X X X X
template <int OP>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    float a = threadIdx.x * 1e-3f + 1.f, b = a + 1, c = a + 2, d = a + 3, e = a + 4, f = a + 5, g = a + 6, h = a + 7;
    unsigned long long m = threadIdx.x;
    const unsigned long long m2 = __builtin_amdgcn_read_exec() ^ (unsigned long long)iters;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) {        // v_fma_f32, three VGPR sources
            REP8(asm volatile("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %5, %6, %1\n v_fma_f32 %2, %6, %7, %2\n v_fma_f32 %3, %7, %4, %3"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "v"(g), "v"(h));)
        } else if (OP == 1) { // v_mul_f32, two VGPR sources
            REP8(asm volatile("v_mul_f32 %0, %4, %5\n v_mul_f32 %1, %5, %6\n v_mul_f32 %2, %6, %7\n v_mul_f32 %3, %7, %4"
                              : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(e), "v"(f), "v"(g), "v"(h));)
        } else if (OP == 2) { // v_xor_b32, two VGPR sources
            REP8(asm volatile("v_xor_b32 %0, %4, %5\n v_xor_b32 %1, %5, %6\n v_xor_b32 %2, %6, %7\n v_xor_b32 %3, %7, %4"
                              : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(e), "v"(f), "v"(g), "v"(h));)
        } else if (OP == 3) { // v_mul_f32 with one SGPR-like constant source
            REP8(asm volatile("v_mul_f32 %0, 2.0, %4\n v_mul_f32 %1, 2.0, %5\n v_mul_f32 %2, 2.0, %6\n v_mul_f32 %3, 2.0, %7"
                              : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(e), "v"(f), "v"(g), "v"(h));)
        } else if (OP == 4) { // v_fmac_f32 (dst is the third source), two explicit VGPR sources
            REP8(asm volatile("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %5, %6\n v_fmac_f32 %2, %6, %7\n v_fmac_f32 %3, %7, %4"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "v"(g), "v"(h));)
        } else if (OP == 5) { // v_mad_u64_u32
            REP8(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0\n v_mad_u64_u32 %0, vcc, %2, %3, 0\n v_mad_u64_u32 %0, vcc, %3, %4, 0\n v_mad_u64_u32 %0, vcc, %4, %1, 0"
                              : "+v"(m) : "v"(e), "v"(f), "v"(g), "v"(h) : "vcc");)
        } else if (OP == 6) { // v_cndmask_b32
            REP8(asm volatile("v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %5, %6, vcc\n v_cndmask_b32 %2, %6, %7, vcc\n v_cndmask_b32 %3, %7, %4, vcc"
                              : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(e), "v"(f), "v"(g), "v"(h) : "vcc");)
        } else if (OP == 8) { // v_mul_f32 with an SGPR source
            REP8(asm volatile("v_mul_f32 %0, %8, %4\n v_mul_f32 %1, %8, %5\n v_mul_f32 %2, %8, %6\n v_mul_f32 %3, %8, %7"
                              : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(e), "v"(f), "v"(g), "v"(h), "s"(iters));)
        } else if (OP == 9) { // v_fma_f32 with one SGPR and two VGPR sources
            REP8(asm volatile("v_fma_f32 %0, %8, %4, %0\n v_fma_f32 %1, %8, %5, %1\n v_fma_f32 %2, %8, %6, %2\n v_fma_f32 %3, %8, %7, %3"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "v"(g), "v"(h), "s"(iters));)
        } else if (OP == 10) { // v_cndmask_b32 with an SGPR-pair mask (VOP3)
            REP8(asm volatile("v_cndmask_b32 %0, %4, %5, %8\n v_cndmask_b32 %1, %5, %6, %8\n v_cndmask_b32 %2, %6, %7, %8\n v_cndmask_b32 %3, %7, %4, %8"
                              : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(e), "v"(f), "v"(g), "v"(h), "s"(m2));)
        } else {              // v_mov_b32
            REP8(asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %6\n v_mov_b32 %3, %7"
                              : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(e), "v"(f), "v"(g), "v"(h));)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h + (float)m;
}
template <int OP> double run(int bpc, int iters, float* d_out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(256 * bpc), dim3(256), 0, 0, d_out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(256 * bpc), dim3(256), 0, 0, d_out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* d_out; hipMalloc(&d_out, sizeof(float) * 256 * 8 * 256);
    const char* names[11] = {"v_fma_f32 v,v,v,v (3 vgpr)", "v_mul_f32 v,v,v (2 vgpr)", "v_xor_b32 v,v,v (2 vgpr)", "v_mul_f32 v,const,v (1 vgpr)",
                            "v_fmac_f32 v,v,v", "v_mad_u64_u32", "v_cndmask_b32 v,v,v,vcc", "v_mov_b32 v,v",
                            "v_mul_f32 v,s,v (sgpr src)", "v_fma_f32 v,s,v,v (sgpr src)", "v_cndmask_b32 v,v,v,s[..]"};
    for (int op = 6; op < 11; ++op)
        for (int bpc = 4; bpc <= 8; bpc *= 2) {
            const int iters = 20000;
            double ms = op == 0 ? run<0>(bpc, iters, d_out) : op == 1 ? run<1>(bpc, iters, d_out) : op == 2 ? run<2>(bpc, iters, d_out) : op == 3 ? run<3>(bpc, iters, d_out)
                      : op == 4 ? run<4>(bpc, iters, d_out) : op == 5 ? run<5>(bpc, iters, d_out) : op == 6 ? run<6>(bpc, iters, d_out) : op == 7 ? run<7>(bpc, iters, d_out) : op == 8 ? run<8>(bpc, iters, d_out) : op == 9 ? run<9>(bpc, iters, d_out) : run<10>(bpc, iters, d_out);
            const double ns = ms * 1e6 / ((double)bpc * iters * 32);
            printf("%-30s waves/SIMD=%d  %.3f ns per wave-instruction per SIMD\n", names[op], bpc, ns);
        }
    return 0;
}

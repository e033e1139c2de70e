// Micro-benchmark: sustained VALU issue rate on MI355X (cycles per wave64 instruction per SIMD)
// for fp32 FMA, v_exp_f32, 32-bit integer multiply and packed fp32 FMA, at 1..8 waves per SIMD.
// Used to price the VALU-bound kernels (cost sweep, sampler) against the right ceiling.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    unsigned u0 = threadIdx.x + 1, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, pa = {a, a}, pb = {b, b};
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
                x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
            }
        } else if (OP == 1) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                x0 = __builtin_amdgcn_exp2f(x0); x1 = __builtin_amdgcn_exp2f(x1); x2 = __builtin_amdgcn_exp2f(x2); x3 = __builtin_amdgcn_exp2f(x3);
                x4 = __builtin_amdgcn_exp2f(x4); x5 = __builtin_amdgcn_exp2f(x5); x6 = __builtin_amdgcn_exp2f(x6); x7 = __builtin_amdgcn_exp2f(x7);
            }
        } else if (OP == 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                u0 = u0 * 0xD2511F53u + 1; u1 = __umulhi(u1, 0xCD9E8D57u) + 3; u2 = u2 * 0x9E3779B9u + 5; u3 = __umulhi(u3, 0xBB67AE85u) + 7;
            }
        } else if (OP == 4) {      // fma with three VGPR operands (no scalar / constant source)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                x0 = fmaf(x0, x4, x5); x1 = fmaf(x1, x5, x6); x2 = fmaf(x2, x6, x7); x3 = fmaf(x3, x7, x4);
                x4 = fmaf(x4, x0, x1); x5 = fmaf(x5, x1, x2); x6 = fmaf(x6, x2, x3); x7 = fmaf(x7, x3, x0);
            }
        } else if (OP == 6) {      // packed fma, all operands VGPR pairs
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                p0 = __builtin_elementwise_fma(p0, p1, p2); p1 = __builtin_elementwise_fma(p1, p2, p3);
                p2 = __builtin_elementwise_fma(p2, p3, p0); p3 = __builtin_elementwise_fma(p3, p0, p1);
            }
        } else if (OP == 7) {      // fma with two VGPR sources and one scalar
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                x0 = fmaf(x0, a, x4); x1 = fmaf(x1, a, x5); x2 = fmaf(x2, b, x6); x3 = fmaf(x3, b, x7);
                x4 = fmaf(x4, a, x0); x5 = fmaf(x5, a, x1); x6 = fmaf(x6, b, x2); x7 = fmaf(x7, b, x3);
            }
        } else if (OP == 5) {      // mul / add / sub with two VGPR operands
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                x0 = x0 * x4; x1 = x1 + x5; x2 = x2 - x6; x3 = x3 * x7;
                x4 = x4 + x0; x5 = x5 * x1; x6 = x6 + x2; x7 = x7 - x3;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                p0 = __builtin_elementwise_fma(p0, pa, pb); p1 = __builtin_elementwise_fma(p1, pa, pb);
                p2 = __builtin_elementwise_fma(p2, pa, pb); p3 = __builtin_elementwise_fma(p3, pa, pb);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + u0 + u1 + u2 + u3 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int OP> double run(int blocks_per_cu, int iters, float* d_out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d_out, 10, 1.0001f, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d_out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* d_out; hipMalloc(&d_out, sizeof(float) * 256 * 8 * 256);
    const char* names[8] = {"v_fma_f32 (1 vgpr src)", "v_exp_f32", "int mul lo/hi (+add)", "v_pk_fma_f32 (1 vgpr src)", "v_fma_f32 (3 vgpr src)", "v_mul/add (2 vgpr src)", "v_pk_fma_f32 (3 vgpr src)", "v_fma_f32 (2 vgpr + sgpr)"};
    const int per_iter[8] = {64, 64, 64 /*mul*/ , 64, 64, 64, 64, 64};
    for (int op = 0; op < 8; ++op)
        for (int bpc = 2; bpc <= 8; bpc *= 4) {
            const int iters = 20000;
            double ms = op == 0 ? run<0>(bpc, iters, d_out) : op == 1 ? run<1>(bpc, iters, d_out) : op == 2 ? run<2>(bpc, iters, d_out) : op == 3 ? run<3>(bpc, iters, d_out) : op == 4 ? run<4>(bpc, iters, d_out) : op == 5 ? run<5>(bpc, iters, d_out) : op == 6 ? run<6>(bpc, iters, d_out) : run<7>(bpc, iters, d_out);
            // waves per SIMD = bpc (256-thread block = 4 waves = 1 per SIMD)
            const double inst_per_simd = (double)bpc * iters * per_iter[op];
            const double ns_per_inst = ms * 1e6 / inst_per_simd;
            printf("%-22s waves/SIMD=%d  %.3f ms  %.3f ns per wave-instruction per SIMD (= %.2f cycles @2.4GHz)\n", names[op], bpc, ms, ns_per_inst, ns_per_inst * 2.4);
        }
    return 0;
}

// Issue cost of single vector instructions on gfx950: one wave per SIMD (and four), N independent copies of one instruction
// in a loop, s_memtime around it.  hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip && ./valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP8(x) x x x x x x x x
#define BODY(NAME, ASM)                                                                                   \
    __global__ void __launch_bounds__(1024) k_##NAME(unsigned long long* out, int iters) {                 \
        unsigned a0 = threadIdx.x * 2654435761u + 12345u, a1 = a0 ^ 0x9e3779b9u;                          \
        unsigned long long d0 = a0, d1 = a1, d2 = a0 + 1, d3 = a1 + 1, d4 = a0 + 2, d5 = a1 + 2, d6 = a0 + 3, d7 = a1 + 3; \
        float f0 = a0 * 1e-9f, f1 = f0 + 1.f, f2 = f0 + 2.f, f3 = f0 + 3.f, f4 = f0 + 4.f, f5 = f0 + 5.f, f6 = f0 + 6.f, f7 = f0 + 7.f; \
        typedef float v2 __attribute__((ext_vector_type(2)));                                             \
        v2 p0 = {f0, f1}, p1 = {f2, f3}, p2 = {f4, f5}, p3 = {f6, f7}, p4 = p0 + 1.f, p5 = p1 + 1.f, p6 = p2 + 1.f, p7 = p3 + 1.f; \
        unsigned u0 = a0, u1 = a1, u2 = a0 + 5, u3 = a1 + 5, u4 = a0 + 7, u5 = a1 + 7, u6 = a0 + 9, u7 = a1 + 9; \
        unsigned long long t0 = __builtin_readcyclecounter();                                             \
        for (int i = 0; i < iters; ++i) { ASM }                                                           \
        unsigned long long t1 = __builtin_readcyclecounter();                                             \
        unsigned long long sink = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7; \
        float fs = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y; \
        if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = sink + (unsigned long long)fs; }   \
    }
#define A8(fmt) \
    asm volatile(fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7) \
        : [d0]"+v"(d0), [d1]"+v"(d1), [d2]"+v"(d2), [d3]"+v"(d3), [d4]"+v"(d4), [d5]"+v"(d5), [d6]"+v"(d6), [d7]"+v"(d7), \
          [u0]"+v"(u0), [u1]"+v"(u1), [u2]"+v"(u2), [u3]"+v"(u3), [u4]"+v"(u4), [u5]"+v"(u5), [u6]"+v"(u6), [u7]"+v"(u7), \
          [f0]"+v"(f0), [f1]"+v"(f1), [f2]"+v"(f2), [f3]"+v"(f3), [f4]"+v"(f4), [f5]"+v"(f5), [f6]"+v"(f6), [f7]"+v"(f7), \
          [p0]"+v"(p0), [p1]"+v"(p1), [p2]"+v"(p2), [p3]"+v"(p3), [p4]"+v"(p4), [p5]"+v"(p5), [p6]"+v"(p6), [p7]"+v"(p7) : [k]"s"(0xD2511F53u) : "vcc");
// operands: d = %0..%7 (64-bit), u = %8..%15, f = %16..%23, p = %24..%31, %[k] = scalar constant
#define STR_(x) #x
#define STR(x) STR_(x)
#define MAD64(i) "v_mad_u64_u32 %[d" STR(i) "], vcc, %[u" STR(i) "], %[k], 0\n"
#define MAD64B(i) "v_mad_u64_u32 %[d" STR(i) "], s[10:11], %[u" STR(i) "], %[k], 0\n"
#define MULHI(i) "v_mul_hi_u32 %[u" STR(i) "], %[u" STR(i) "], %[k]\n"
#define MULLO(i) "v_mul_lo_u32 %[u" STR(i) "], %[u" STR(i) "], %[k]\n"
#define BITOP(i) "v_bitop3_b32 %[u" STR(i) "], %[u" STR(i) "], %[u" STR(i) "], %[k] bitop3:0x96\n"
#define XOR(i) "v_xor_b32 %[u" STR(i) "], %[k], %[u" STR(i) "]\n"
#define FMA(i) "v_fma_f32 %[f" STR(i) "], %[f" STR(i) "], %[f" STR(i) "], %[f" STR(i) "]\n"
#define PKFMA(i) "v_pk_fma_f32 %[p" STR(i) "], %[p" STR(i) "], %[p" STR(i) "], %[p" STR(i) "]\n"
#define PKMUL(i) "v_pk_mul_f32 %[p" STR(i) "], %[p" STR(i) "], %[p" STR(i) "]\n"
#define EXPF(i) "v_exp_f32 %[f" STR(i) "], %[f" STR(i) "]\n"
#define SINF(i) "v_sin_f32 %[f" STR(i) "], %[f" STR(i) "]\n"
#define LOGF(i) "v_log_f32 %[f" STR(i) "], %[f" STR(i) "]\n"
#define SQRTF(i) "v_sqrt_f32 %[f" STR(i) "], %[f" STR(i) "]\n"
#define CVT(i) "v_cvt_f32_u32 %[f" STR(i) "], %[u" STR(i) "]\n"
#define MUL24(i) "v_mul_u32_u24 %[u" STR(i) "], %[u" STR(i) "], %[k]\n"
#define MULHI24(i) "v_mul_hi_u32_u24 %[u" STR(i) "], %[u" STR(i) "], %[k]\n"
#define MAD24(i) "v_mad_u32_u24 %[u" STR(i) "], %[u" STR(i) "], %[k], %[u" STR(i) "]\n"
#define ADD(i) "v_add_u32 %[u" STR(i) "], %[k], %[u" STR(i) "]\n"
#define MADU32(i) "v_mad_u32_u16 %[u" STR(i) "], %[u" STR(i) "], %[k], %[u" STR(i) "]\n"
#define NOP(i) "s_nop 0\n"
BODY(mad_u64_u32, A8(MAD64))
BODY(mul_hi_u32, A8(MULHI))
BODY(mul_lo_u32, A8(MULLO))
BODY(bitop3, A8(BITOP))
BODY(xor, A8(XOR))
BODY(fma_f32, A8(FMA))
BODY(pk_fma_f32, A8(PKFMA))
BODY(pk_mul_f32, A8(PKMUL))
BODY(exp_f32, A8(EXPF))
BODY(sin_f32, A8(SINF))
BODY(log_f32, A8(LOGF))
BODY(sqrt_f32, A8(SQRTF))
BODY(cvt_f32_u32, A8(CVT))
BODY(mul_u32_u24, A8(MUL24))
BODY(mul_hi_u32_u24, A8(MULHI24))
BODY(mad_u32_u24, A8(MAD24))
BODY(add_u32, A8(ADD))
BODY(s_nop, A8(NOP))
typedef void (*kern_t)(unsigned long long*, int);
int main() {
    unsigned long long* d; hipMalloc(&d, 16);
    struct { const char* n; kern_t k; } ks[] = {
        {"v_mad_u64_u32", k_mad_u64_u32}, {"v_mul_hi_u32", k_mul_hi_u32}, {"v_mul_lo_u32", k_mul_lo_u32}, {"v_bitop3_b32", k_bitop3},
        {"v_xor_b32", k_xor}, {"v_add_u32", k_add_u32}, {"v_fma_f32", k_fma_f32}, {"v_pk_fma_f32", k_pk_fma_f32}, {"v_pk_mul_f32", k_pk_mul_f32},
        {"v_exp_f32", k_exp_f32}, {"v_sin_f32", k_sin_f32}, {"v_log_f32", k_log_f32}, {"v_sqrt_f32", k_sqrt_f32}, {"v_cvt_f32_u32", k_cvt_f32_u32},
        {"v_mul_u32_u24", k_mul_u32_u24}, {"v_mul_hi_u32_u24", k_mul_hi_u32_u24}, {"v_mad_u32_u24", k_mad_u32_u24}, {"s_nop 0", k_s_nop}};
    const int iters = 200000;
    for (int wpb : {256, 512, 1024}) {      // 1 wave per CU (one SIMD), 1 wave per SIMD, 4 waves per SIMD
        printf("---- %d threads per workgroup (%s), one workgroup per CU; cycles (s_memtime, 100 MHz ticks x clock ratio) per instruction of wave 0\n", wpb,
               wpb == 256 ? "one wave per SIMD" : wpb == 512 ? "two waves per SIMD" : "four waves per SIMD");
        for (auto& e : ks) {
            unsigned long long h[2];
            hipLaunchKernelGGL(e.k, dim3(256), dim3(wpb), 0, 0, d, iters);
            hipDeviceSynchronize();
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a);
            hipLaunchKernelGGL(e.k, dim3(256), dim3(wpb), 0, 0, d, iters);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            const double n = (double)iters * 8;
            printf("%-18s %7.2f ticks/instr   launch %.3f ms -> %.3f ns per instruction and SIMD (x waves)\n", e.n, (double)h[0] / n, ms, ms * 1e6 / (n * (wpb / 256)));
        }
    }
    return 0;
}

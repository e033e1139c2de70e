// Effective shader clock under a K3-like VALU load (packed fp32 FMA + v_exp_f32) on every SIMD:
// s_memtime (shader clock ticks) against s_memrealtime (constant 100 MHz).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) burn(int iters, int mode, unsigned long long* out, float* sink) {
    f2 a = {1.0f + threadIdx.x * 1e-6f, 0.5f}, b = {0.999f, 1.001f}, c = {1e-3f, -1e-3f};
    f2 d = a, e = b, f = c, g = a + b;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a = a * b + c; d = d * b + c; e = e * b + a; f = f * b + d;
            if (mode == 1) { g.x = __builtin_amdgcn_exp2f(g.x * -0.5f); g.y = __builtin_amdgcn_exp2f(g.y * -0.25f); }
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = r1 - r0; }
    if (a.x + d.x + e.x + f.x + g.x + g.y == 123.f) sink[0] = a.y;
}
int main() {
    const int blocks = 256 * 8;
    unsigned long long* out; float* sink;
    hipMalloc(&out, blocks * 16); hipMalloc(&sink, 4);
    std::vector<unsigned long long> h(2 * blocks);
    for (int mode = 0; mode < 2; ++mode)
        for (int iters : {2000, 20000, 200000}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(burn, dim3(blocks), dim3(256), 0, 0, iters, mode, out, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), out, blocks * 16, hipMemcpyDeviceToHost);
            double cyc = 0, rt = 0;
            for (int i = 0; i < blocks; ++i) { cyc += h[2 * i]; rt += h[2 * i + 1]; }
            const double per_wave_instr = (double)iters * 8 * (mode ? 6 : 4);
            printf("mode %d iters %6d: kernel %.3f ms, memtime/realtime = %.3f (x100 MHz = %.0f MHz), cycles per VALU instr per wave %.2f\n",
                   mode, iters, ms, cyc / rt, cyc / rt * 100., (cyc / blocks) / per_wave_instr);
        }
    return 0;
}

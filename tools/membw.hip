// Streaming bandwidth of this MI355X box for the sizes of config 3 (470 MB per tensor):
// fill (write-only), read (read-only, reduce), copy (read + write), float4 per lane, grid-stride.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) fill(float4* o, size_t n, float v) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) o[i] = float4{v, v, v, v};
}
__global__ void __launch_bounds__(256) readk(const float4* a, size_t n, float* out) {
    float s = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { float4 x = a[i]; s += x.x + x.y + x.z + x.w; }
    if (s == 1.2345f) out[0] = s;
}
__global__ void __launch_bounds__(256) copyk(const float4* a, float4* o, size_t n) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) o[i] = a[i];
}
int main() {
    const size_t bytes = 469762048, n = bytes / 16;
    float4 *a, *b; float* o;
    (void)hipMalloc(&a, bytes); (void)hipMalloc(&b, bytes); (void)hipMalloc(&o, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int blocks : {2048, 8192, 32768}) {
        for (int op = 0; op < 3; ++op) {
            float best = 1e9;
            for (int rep = 0; rep < 6; ++rep) {
                (void)hipEventRecord(e0);
                if (op == 0) hipLaunchKernelGGL(fill, dim3(blocks), dim3(256), 0, 0, a, n, 1.f);
                if (op == 1) hipLaunchKernelGGL(readk, dim3(blocks), dim3(256), 0, 0, a, n, o);
                if (op == 2) hipLaunchKernelGGL(copyk, dim3(blocks), dim3(256), 0, 0, a, b, n);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;
            }
            const double gb = (op == 2 ? 2.0 : 1.0) * bytes / 1e9;
            printf("%-5s blocks=%5d: %.1f us  %.2f TB/s\n", op == 0 ? "fill" : op == 1 ? "read" : "copy", blocks, best * 1e3, gb / best);
        }
    }
    return 0;
}

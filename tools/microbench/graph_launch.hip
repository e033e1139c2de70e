// Host cost of one planner step's TWO kernel launches on this ROCm, four ways (round-5 verdict, item 4b: would a hipGraph over
// the step's launches make the reference examples' optimize(opt_iters=1) loop cheaper on the host?):
//   direct      2 x hipLaunchKernelGGL per step (what sgpmp_step does)
//   graph       1 x hipGraphLaunch of a captured 2-kernel graph, nothing changes between replays (only possible if the draw
//               counter and every per-call pointer live in device memory)
//   graph+set1  hipGraphExecKernelNodeSetParams on ONE node per replay (the draw counter as a kernel argument), then launch
//   graph+set2  ... on both nodes (draw counter + the call's fresh means_prev tensor)
// Kernel arguments: a 512-byte struct by value (the fused launch passes its FusedArgs that way) and a small one; the kernels
// themselves last ~10 us (a dependent chain on one wave), like the example-size step's.  Reports host microseconds per step
// to ENQUEUE (queue kept short by a sync every 64 steps, outside the timed part) and end-to-end microseconds per step.
//   hipcc --offload-arch=gfx950 -O3 -o graph_launch graph_launch.hip && ./graph_launch
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

struct Big { unsigned long long draw; float* out; float pad[124]; };   // 512 bytes
struct Small { unsigned long long draw; float* out; float* prev; };

__global__ void k_big(Big a, int spin) {
    float v = (float)(a.draw & 1023u) + a.pad[threadIdx.x & 63];
    for (int i = 0; i < spin; ++i) v = __builtin_fmaf(v, 1.0000001f, 1e-7f);
    if (threadIdx.x == 0) a.out[blockIdx.x] = v;
}
__global__ void k_small(Small a, int spin) {
    float v = a.out[blockIdx.x];
    for (int i = 0; i < spin; ++i) v = __builtin_fmaf(v, 0.9999999f, 1e-7f);
    if (threadIdx.x == 0) { a.prev[blockIdx.x] = v; a.out[blockIdx.x] = v + (float)(a.draw & 7u); }
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? std::atoi(argv[1]) : 20000, blocks = 20, spin = argc > 2 ? std::atoi(argv[2]) : 4000;
    float *out, *prev0, *prev1;
    CHK(hipMalloc(&out, 4096)); CHK(hipMalloc(&prev0, 4096)); CHK(hipMalloc(&prev1, 4096));
    hipStream_t st;
    CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    Big big{}; big.out = out;
    Small sm{}; sm.out = out; sm.prev = prev0;
    int spin_arg = spin;

    // capture the two launches once
    hipGraph_t graph; hipGraphExec_t exec;
    CHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(k_big, dim3(blocks), dim3(256), 0, st, big, spin_arg);
    hipLaunchKernelGGL(k_small, dim3(blocks), dim3(256), 0, st, sm, spin_arg);
    CHK(hipStreamEndCapture(st, &graph));
    CHK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    size_t n_nodes = 0;
    CHK(hipGraphGetNodes(graph, nullptr, &n_nodes));
    hipGraphNode_t nodes[8];
    CHK(hipGraphGetNodes(graph, nodes, &n_nodes));
    hipKernelNodeParams kp[2];
    int big_i = -1, small_i = -1;
    for (size_t i = 0; i < n_nodes && i < 8; ++i) {
        hipKernelNodeParams p;
        if (hipGraphKernelNodeGetParams(nodes[i], &p) != hipSuccess) continue;
        if (p.func == (void*)k_big) { big_i = (int)i; kp[0] = p; }
        if (p.func == (void*)k_small) { small_i = (int)i; kp[1] = p; }
    }
    std::printf("graph nodes %zu (big %d, small %d)\n", n_nodes, big_i, small_i);

    for (int mode = 0; mode < 4; ++mode) {
        if (mode >= 2 && (big_i < 0 || small_i < 0)) { std::printf("mode %d skipped: kernel nodes not found\n", mode); continue; }
        // warm-up
        for (int i = 0; i < 200; ++i) {
            if (mode == 0) { hipLaunchKernelGGL(k_big, dim3(blocks), dim3(256), 0, st, big, spin_arg); hipLaunchKernelGGL(k_small, dim3(blocks), dim3(256), 0, st, sm, spin_arg); }
            else CHK(hipGraphLaunch(exec, st));
        }
        CHK(hipStreamSynchronize(st));
        double enqueue = 0., t_all0 = now_us();
        for (int i = 0; i < steps; ++i) {
            big.draw = sm.draw = (unsigned long long)i;
            sm.prev = (i & 1) ? prev1 : prev0;
            const double t0 = now_us();
            if (mode == 0) {
                hipLaunchKernelGGL(k_big, dim3(blocks), dim3(256), 0, st, big, spin_arg);
                hipLaunchKernelGGL(k_small, dim3(blocks), dim3(256), 0, st, sm, spin_arg);
            } else {
                if (mode >= 2) {
                    void* a0[2] = {&big, &spin_arg};
                    hipKernelNodeParams p = kp[0]; p.kernelParams = a0;
                    CHK(hipGraphExecKernelNodeSetParams(exec, nodes[big_i], &p));
                }
                if (mode >= 3) {
                    void* a1[2] = {&sm, &spin_arg};
                    hipKernelNodeParams p = kp[1]; p.kernelParams = a1;
                    CHK(hipGraphExecKernelNodeSetParams(exec, nodes[small_i], &p));
                }
                CHK(hipGraphLaunch(exec, st));
            }
            enqueue += now_us() - t0;
            if ((i & 63) == 63) CHK(hipStreamSynchronize(st));
        }
        CHK(hipStreamSynchronize(st));
        const double all = now_us() - t_all0;
        static const char* names[4] = {"direct (2 launches)", "graph (static)", "graph + set params on 1 node", "graph + set params on 2 nodes"};
        std::printf("%-32s host enqueue %6.2f us/step   end-to-end %6.2f us/step\n", names[mode], enqueue / steps, all / steps);
    }
    // the kernels alone (events)
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    CHK(hipEventRecord(e0, st));
    for (int i = 0; i < 100; ++i) { hipLaunchKernelGGL(k_big, dim3(blocks), dim3(256), 0, st, big, spin_arg); hipLaunchKernelGGL(k_small, dim3(blocks), dim3(256), 0, st, sm, spin_arg); }
    CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1));
    float ms = 0.f; CHK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("device time of the two kernels back to back: %.2f us/step\n", ms * 10.f);
    return 0;
}

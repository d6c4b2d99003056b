// How much of a launch-bound search pipeline (H2D copy, 2 memsets, 4 small kernels, 2 D2H copies, sync) is
// launch overhead, and what a captured hipGraph replay of the same chain costs.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro_graph tools/micro_graph.hip && tools/micro_graph
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void tiny(unsigned* p, unsigned v) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += v; }
int main() {
    hipStream_t st; OK(hipStreamCreate(&st));
    unsigned *d, *hq, *hr, *hf; OK(hipMalloc(&d, 1 << 20));
    OK(hipHostMalloc(&hq, 4096)); OK(hipHostMalloc(&hr, 4096)); OK(hipHostMalloc(&hf, 4096));
    auto chain = [&]() {
        (void)hipMemcpyAsync(d, hq, 256, hipMemcpyHostToDevice, st);
        (void)hipMemsetAsync(d + 1024, 0, 64, st);
        hipLaunchKernelGGL(tiny, dim3(16), dim3(256), 0, st, d, 1u);
        (void)hipMemsetAsync(d + 2048, 0, 4096, st);
        hipLaunchKernelGGL(tiny, dim3(256), dim3(256), 0, st, d, 2u);
        hipLaunchKernelGGL(tiny, dim3(16), dim3(256), 0, st, d, 3u);
        hipLaunchKernelGGL(tiny, dim3(1), dim3(256), 0, st, d, 4u);
        (void)hipMemcpyAsync(hf, d + 1024, 64, hipMemcpyDeviceToHost, st);
        (void)hipMemcpyAsync(hr, d, 240, hipMemcpyDeviceToHost, st);
    };
    const int reps = 2000;
    for (int i = 0; i < 50; ++i) { chain(); OK(hipStreamSynchronize(st)); }
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i) { chain(); OK(hipStreamSynchronize(st)); }
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    printf("stream launches: %.1f us per chain (9 operations + sync)\n", us);
    hipGraph_t g; hipGraphExec_t ge;
    OK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    chain();
    OK(hipStreamEndCapture(st, &g));
    OK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 50; ++i) { OK(hipGraphLaunch(ge, st)); OK(hipStreamSynchronize(st)); }
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i) { OK(hipGraphLaunch(ge, st)); OK(hipStreamSynchronize(st)); }
    us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    printf("graph replay:    %.1f us per chain\n", us);
    // one kernel + sync, for scale
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i) { hipLaunchKernelGGL(tiny, dim3(1), dim3(256), 0, st, d, 1u); OK(hipStreamSynchronize(st)); }
    us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    printf("one kernel + sync: %.1f us\n", us);
    return 0;
}

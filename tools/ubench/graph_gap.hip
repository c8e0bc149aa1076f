// Does a hipGraph shorten the gap between dependent kernels?  A chain of 3 small dependent kernels (a few us each), 200 times: launched one by one into a
// stream against one graph of 12 / 60 kernel nodes launched 50 / 10 times.  build: hipcc -O3 --offload-arch=gfx950 tools/ubench/graph_gap.hip -o /tmp/graph_gap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void work(float* p, int n, int reps)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = p[i];
    for (int r = 0; r < reps; r++) v = v * 1.0001f + 0.5f;
    p[i] = v;
}
int main()
{
    const int n = 120000, reps = 200, chains = 200;
    float* p; hipMalloc(&p, n * sizeof(float)); hipMemset(p, 0, n * sizeof(float));
    hipStream_t s; hipStreamCreate(&s);
    auto chain = [&](hipStream_t st) { for (int k = 0; k < 3; k++) hipLaunchKernelGGL(work, dim3((n + 255) / 256), dim3(256), 0, st, p, n, reps); };
    for (int w = 0; w < 20; w++) chain(s);
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int c = 0; c < chains; c++) chain(s);
    hipStreamSynchronize(s);
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("stream launches: %.2f us per kernel (3 x %d dependent kernels)\n", us / (3.0 * chains), chains);
    for (int per : { 4, 20 }) {
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int c = 0; c < per; c++) chain(s);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        for (int w = 0; w < 3; w++) hipGraphLaunch(ge, s);
        hipStreamSynchronize(s);
        t0 = std::chrono::steady_clock::now();
        for (int c = 0; c < chains / per; c++) hipGraphLaunch(ge, s);
        hipStreamSynchronize(s);
        us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("graph of %d kernel nodes x %d launches: %.2f us per kernel\n", 3 * per, chains / per, us / (3.0 * chains));
    }
    // the kernel alone (one launch timed by events)
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s); hipLaunchKernelGGL(work, dim3((n + 255) / 256), dim3(256), 0, s, p, n, reps); hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("one kernel between two events: %.2f us\n", ms * 1e3);
    return 0;
}

// Issue rate of the f64 vector forms on gfx950: v_add_f64 / v_mul_f64 / v_fma_f64 (and the f32 FMA for scale), 8 independent chains per lane.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/f64_rate.hip -o /tmp/f64_rate ; run on the GPU box.  -> profiles/r04_ubench_f64_rate.txt
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(double* out, double a, double b, int iters)
{
    double v[8];
    float f[8];
    for (int j = 0; j < 8; j++) { v[j] = a + j + threadIdx.x; f[j] = (float)v[j]; }
    const float af = (float)a, bf = (float)b;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (MODE == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[j]) : "v"(b));
            if (MODE == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[j]) : "v"(b));
            if (MODE == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[j]) : "v"(b), "v"(a));
            if (MODE == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[j]) : "v"(bf), "v"(af));
            if (MODE == 4) asm volatile("v_fma_f64 %0, %0, 1.0, %1" : "+v"(v[j]) : "v"(b));
        }
    }
    double s = 0;
    for (int j = 0; j < 8; j++) s += v[j] + f[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
    double* out; hipMalloc(&out, 1024 * 256 * 8 * sizeof(double));
    const int iters = 4096, blocks = 1024 * 4;      // 4 workgroups of 256 per CU: one wave per SIMD x 4
    const char* names[5] = { "v_add_f64", "v_mul_f64", "v_fma_f64", "v_fma_f32", "v_fma_f64 (x * 1.0 + y)" };
    for (int m = 0; m < 5; m++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (m == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, 1.000001, 0.999999, iters);
            if (m == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, 1.000001, 0.999999, iters);
            if (m == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, 1.000001, 0.999999, iters);
            if (m == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, 1.000001, 0.999999, iters);
            if (m == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, out, 1.000001, 0.999999, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double winstr = (double)blocks * 4 * iters * 8;              // wave-instructions
        printf("%-26s %.3f ms: %.2f cycles per wave-instruction per SIMD at 2.4 GHz (%.1f T lane-ops/s)\n", names[m], ms, ms * 1e-3 * 2.4e9 / (winstr / 1024.0), winstr * 64 / (ms * 1e-3) / 1e12);
    }
    return 0;
}

// Feasibility microbenchmark for a bf16 matrix-core distance filter (gfx950):
//   A. numerics of v_mfma_f32_32x32x16_bf16: error of D = sum_k a_k b_k against the exact sum, in units of 2^-24 * sum |a_k b_k|
//      (bf16 x bf16 products are exact in f32; the question is how the 16 products are accumulated);
//   B. rate of the loop shape "MFMA -> minimum of the lane's 16 accumulators -> compare" with QG query groups per wave.
// build: hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -o mfma_filter mfma_filter.hip      run: ./mfma_filter
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

// one wave: D[32][32] = A[32][16] * B[16][32]; a/b hold bf16 bit patterns, A row-major [m][k], B [k][n]
__global__ void tile_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b, float* __restrict__ d)
{
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    u16x8 av, bv;
    for (int j = 0; j < 8; j++) { av[j] = a[r * 16 + 8 * h + j]; bv[j] = b[(8 * h + j) * 32 + r]; }
    f32x16 acc;
    for (int j = 0; j < 16; j++) acc[j] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc, 0, 0, 0);
    for (int reg = 0; reg < 16; reg++) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
        d[row * 32 + r] = acc[reg];
    }
}

// MODE 0: MFMA only (results summed into one register per group at the end); 1: the 8 x v_min3 + compare only, on registers that
// no MFMA writes; 2: MFMA, then the minimum of ITS results (dependent); 3: MFMA, and the minimum of the PREVIOUS group's results;
// 4: the waves of a SIMD split roles — even workgroups run mode 0, odd ones mode 1
template <int MODE>
__global__ __launch_bounds__(256) void mix_kernel(const uint4* __restrict__ ops, float* __restrict__ out, int tiles, float thr)
{
    constexpr int QG = 4;
    const int lane = threadIdx.x & 63;
    const int mode = MODE == 4 ? (blockIdx.x & 1) : MODE >= 5 ? 2 : MODE;
    u16x8 bq[QG];
    float best[QG];
    for (int g = 0; g < QG; g++) {
        for (int j = 0; j < 8; j++) bq[g][j] = (unsigned short)(0x3F80 + ((lane * 7 + g * 13 + j) & 63));
        best[g] = thr;
    }
    f32x16 zero, pend, keep;
    for (int j = 0; j < 16; j++) { zero[j] = 0.f; pend[j] = 1.0f + lane + j; keep[j] = 0.f; }
    float big;
    asm volatile("v_mov_b32 %0, 0x7f800000" : "=v"(big));
    uint4 nxt = ops[lane];
    for (int T = 0; T < tiles; T++) {
        const uint4 cur = nxt;
        nxt = ops[((T + 1) & 63) * 64 + lane];
        const bf16x8 av = __builtin_bit_cast(bf16x8, cur);
#pragma unroll
        for (int g = 0; g < QG; g++) {
            f32x16 acc = pend;
            if (mode != 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8, bq[g]), zero, 0, 0, 0);
            if (MODE >= 5) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, nxt), __builtin_bit_cast(bf16x8, bq[g]), acc, 0, 0, 0);   // K = 32: a dependent pair
            if (mode == 0 || MODE == 6) { best[g] += acc[0] + acc[15]; continue; }          // MFMA only, results kept alive
            const f32x16 src = (mode == 3) ? pend : acc;
            float m = big;
#pragma unroll
            for (int j = 0; j + 1 < 16; j += 2) m = fminf(fminf(m, src[j]), src[j + 1]);
            if (m < best[g]) best[g] = m * 0.5f;
            if (mode == 1) { pend[g] = m + 1.0f; }          // keeps the loop from being hoisted
            if (mode == 3) pend = acc;
        }
    }
    float s = 0.f;
    for (int g = 0; g < QG; g++) s += best[g];
    for (int j = 0; j < 16; j++) s += keep[j] + pend[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int QG>
__global__ __launch_bounds__(256) void rate_kernel(const uint4* __restrict__ ops, float* __restrict__ out, unsigned long long* ticks, int tiles, float thr)
{
    const int lane = threadIdx.x & 63;
    u16x8 bq[QG];
    float best[QG];
    int hits = 0;
    for (int g = 0; g < QG; g++) {
        for (int j = 0; j < 8; j++) bq[g][j] = (unsigned short)(0x3F80 + ((lane * 7 + g * 13 + j) & 63));
        best[g] = thr;
    }
    f32x16 zero;
    for (int j = 0; j < 16; j++) zero[j] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    uint4 nxt = ops[lane];
    for (int T = 0; T < tiles; T++) {
        const uint4 cur = nxt;
        nxt = ops[((T + 1) & 63) * 64 + lane];
        const bf16x8 av = __builtin_bit_cast(bf16x8, cur);
#pragma unroll
        for (int g = 0; g < QG; g++) {
            f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8, bq[g]), zero, 0, 0, 0);
            float m = fminf(fminf(acc[0], acc[1]), acc[2]);
#pragma unroll
            for (int j = 3; j + 1 < 16; j += 2) m = fminf(fminf(m, acc[j]), acc[j + 1]);
            m = fminf(m, acc[15]);
            if (m < best[g]) { hits++; best[g] = m * 0.5f; }          // rare: the threshold is far below the products
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int g = 0; g < QG; g++) s += best[g];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + hits;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

static uint16_t bf16_trunc(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); }
static float bf16_val(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

int main()
{
    // ---- A
    std::mt19937_64 rng(12345);
    std::normal_distribution<float> nd(0.f, 1.f);
    uint16_t *da, *db; float* dd;
    CK(hipMalloc(&da, 32 * 16 * 2)); CK(hipMalloc(&db, 16 * 32 * 2)); CK(hipMalloc(&dd, 32 * 32 * 4));
    const char* names[4] = { "normal", "wide exponents (2^-20..2^20 per k)", "cancellation (+big, -big, small)", "descending magnitudes" };
    for (int mode = 0; mode < 4; mode++) {
        double worst = 0.0, worst_fma = 0.0; int exact_chain = 0, total = 0;
        for (int trial = 0; trial < 200; trial++) {
            std::vector<uint16_t> a(32 * 16), b(16 * 32);
            for (int m = 0; m < 32; m++) for (int k = 0; k < 16; k++) {
                float v = nd(rng);
                if (mode == 1) v = std::ldexp(v, (int)(rng() % 41) - 20);
                if (mode == 2) v = (k == 0) ? 1000.f + nd(rng) : (k == 1 ? -1000.f + nd(rng) : nd(rng) * 1e-3f);
                if (mode == 3) v = std::ldexp(v, -2 * k);
                a[m * 16 + k] = bf16_trunc(v);
            }
            for (int k = 0; k < 16; k++) for (int n = 0; n < 32; n++) {
                float v = nd(rng);
                if (mode == 2) v = (k < 2) ? 1.0f : nd(rng);
                b[k * 32 + n] = bf16_trunc(v);
            }
            CK(hipMemcpy(da, a.data(), a.size() * 2, hipMemcpyHostToDevice));
            CK(hipMemcpy(db, b.data(), b.size() * 2, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(tile_kernel, dim3(1), dim3(64), 0, 0, da, db, dd);
            std::vector<float> d(32 * 32);
            CK(hipMemcpy(d.data(), dd, d.size() * 4, hipMemcpyDeviceToHost));
            for (int m = 0; m < 32; m++) for (int n = 0; n < 32; n++) {
                double ex = 0.0, mag = 0.0; float chain = 0.f;
                for (int k = 0; k < 16; k++) {
                    const double p = (double)bf16_val(a[m * 16 + k]) * (double)bf16_val(b[k * 32 + n]);
                    ex += p; mag += std::fabs(p);
                    chain = std::fmaf(bf16_val(a[m * 16 + k]), bf16_val(b[k * 32 + n]), chain);
                }
                const double err = std::fabs((double)d[m * 32 + n] - ex) / (mag > 0 ? mag : 1.0) * 16777216.0;
                const double errc = std::fabs((double)chain - ex) / (mag > 0 ? mag : 1.0) * 16777216.0;
                if (err > worst) worst = err;
                if (errc > worst_fma) worst_fma = errc;
                exact_chain += (d[m * 32 + n] == chain); total++;
            }
        }
        printf("A %-40s max |D - exact| = %.3f x 2^-24 x sum|a b|   (k-ordered fmaf chain: %.3f; D == chain in %d of %d)\n", names[mode], worst, worst_fma, exact_chain, total);
    }
    // ---- B
    int dev = 0; hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    uint4* ops; float* out; unsigned long long* ticks;
    CK(hipMalloc(&ops, 65 * 64 * sizeof(uint4)));
    std::vector<uint32_t> h(65 * 64 * 4);
    for (auto& v : h) v = 0x3F803F80u + (uint32_t)(rng() & 0x003F003Fu);
    CK(hipMemcpy(ops, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4)); CK(hipMalloc(&ticks, (size_t)cus * 8 * 8));
    const int tiles = 4000;
    for (int wps = 1; wps <= 4; wps++) {          // waves per SIMD = workgroups per CU (256 threads = one wave per SIMD each)
        for (int qg : { 2, 4, 8 }) {
            const int blocks = cus * wps;
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int rep = 0; rep < 2; rep++) {
                CK(hipEventRecord(e0));
                if (qg == 2) hipLaunchKernelGGL(rate_kernel<2>, dim3(blocks), dim3(256), 0, 0, ops, out, ticks, tiles, -1e30f);
                else if (qg == 4) hipLaunchKernelGGL(rate_kernel<4>, dim3(blocks), dim3(256), 0, 0, ops, out, ticks, tiles, -1e30f);
                else hipLaunchKernelGGL(rate_kernel<8>, dim3(blocks), dim3(256), 0, 0, ops, out, ticks, tiles, -1e30f);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            }
            float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
            const double tile_groups_per_simd = (double)tiles * qg * wps;          // each SIMD holds wps waves
            printf("B waves/SIMD %d QG %d: %.3f ms -> %.1f ns per (group, tile) per SIMD = %.1f cycles at 2.4 GHz  (%.1f G pair-bounds/s chip-wide)\n", wps, qg, ms,
                   ms * 1e6 / tile_groups_per_simd, ms * 1e6 / tile_groups_per_simd * 2.4, 1024.0 * tile_groups_per_simd * cus * 4 / (ms * 1e-3) * 1e-9);
        }
    }
    // ---- C: do the matrix pipe and the vector ALU overlap?
    const char* mnames[7] = { "MFMA only", "8 x v_min3 + compare only", "MFMA -> min of its own results", "MFMA + min of the previous results", "half the waves MFMA only, half min only",
                              "two dependent MFMAs (K = 32) -> min of the results", "two dependent MFMAs only" };
    for (int wps : { 1, 2, 4 }) {
        for (int mode = 0; mode < 7; mode++) {
            const int blocks = cus * wps;
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int rep = 0; rep < 2; rep++) {
                CK(hipEventRecord(e0));
                switch (mode) {
                case 0: hipLaunchKernelGGL(mix_kernel<0>, dim3(blocks), dim3(256), 0, 0, ops, out, tiles, -1e30f); break;
                case 1: hipLaunchKernelGGL(mix_kernel<1>, dim3(blocks), dim3(256), 0, 0, ops, out, tiles, -1e30f); break;
                case 2: hipLaunchKernelGGL(mix_kernel<2>, dim3(blocks), dim3(256), 0, 0, ops, out, tiles, -1e30f); break;
                case 3: hipLaunchKernelGGL(mix_kernel<3>, dim3(blocks), dim3(256), 0, 0, ops, out, tiles, -1e30f); break;
                case 4: hipLaunchKernelGGL(mix_kernel<4>, dim3(blocks), dim3(256), 0, 0, ops, out, tiles, -1e30f); break;
                case 5: hipLaunchKernelGGL(mix_kernel<5>, dim3(blocks), dim3(256), 0, 0, ops, out, tiles, -1e30f); break;
                default: hipLaunchKernelGGL(mix_kernel<6>, dim3(blocks), dim3(256), 0, 0, ops, out, tiles, -1e30f); break;
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            }
            float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
            const double tg = (double)tiles * 4 * wps;
            printf("C waves/SIMD %d  %-42s %.3f ms -> %.1f cycles at 2.4 GHz per (group, tile) per SIMD\n", wps, mnames[mode], ms, ms * 1e6 / tg * 2.4);
        }
    }
    return 0;
}

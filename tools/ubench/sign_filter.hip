// Feasibility microbenchmark (gfx950): a THRESHOLD filter on the f16 matrix pipe.  With the query's threshold folded into the two free
// K-slots of v_mfma_f32_32x32x16_f16 the accumulator of a (query, record) pair is "bound - threshold": its SIGN says whether the record
// can matter, and the vector ALU only has to OR sign bits (v_or3_b32: two new values per instruction, presumably full rate) instead of
// taking minima (v_min3_f32: half rate, profiles/r01_ubench_valu_rate.txt) and tracking first / second minimum and the chunk.
//   mode 0: MFMA only                                   mode 1: 8 x v_or3 + sign test only (no MFMA)
//   mode 2: MFMA -> OR of its 16 results, sign test per (group, tile)
//   mode 3: MFMA -> OR of its 16 results, ORed over the QG groups, ONE sign test per tile
//   mode 4: MFMA -> 8 x v_min3 + first / second minimum tracking (the loop shape of round 2's kernel), for comparison
//   mode 5: as mode 3 with the threshold in the C operand (a 16-register vector per group) instead of K-slots
// build: hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -o sign_filter sign_filter.hip      run: ./sign_filter
// (without -amdgpu-mfma-vgpr-form the accumulators live in AGPRs and every one of them costs a v_accvgpr_read before the OR)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int QG>
__global__ __launch_bounds__(256) void sf_kernel(const uint4* __restrict__ ops, float* __restrict__ out, unsigned long long* ticks, int tiles)
{
    const int lane = threadIdx.x & 63;
    uint4 bq[QG];
    float m1[QG], m2[QG];
    uint32_t c1[QG];
    uint32_t hits = 0;
    f32x16 cvec[QG];
    for (int g = 0; g < QG; g++) {
        // f16 operands near 1.0: every product positive, so no sign is ever set (the rare branch stays rare)
        bq[g] = make_uint4(0x3C003C00u + (lane & 15), 0x3C003C00u + g, 0x3C003C01u, 0x3C003C02u);
        m1[g] = 1e30f; m2[g] = 1e30f; c1[g] = 0;
        for (int j = 0; j < 16; j++) cvec[g][j] = MODE == 5 ? 1.0f + g : 0.0f;
    }
    f32x16 pend;
    for (int j = 0; j < 16; j++) pend[j] = 1.0f + lane + j;
    float big;
    asm volatile("v_mov_b32 %0, 0x7f800000" : "=v"(big));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    uint4 nxt = ops[lane];
    for (int T = 0; T < tiles; T++) {
        const uint4 cur = nxt;
        nxt = ops[((T + 1) & 63) * 64 + lane];
        const f16x8 av = __builtin_bit_cast(f16x8, cur);
        uint32_t any = 0;
#pragma unroll
        for (int g = 0; g < QG; g++) {
            f32x16 acc = pend;
            if (MODE != 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, __builtin_bit_cast(f16x8, bq[g]), cvec[g], 0, 0, 0);
            if (MODE == 0) { m1[g] += acc[0] + acc[15]; continue; }
            if (MODE == 4) {
                float m = big;
#pragma unroll
                for (int j = 0; j + 1 < 16; j += 2) m = fminf(fminf(m, acc[j]), acc[j + 1]);
                m2[g] = __builtin_amdgcn_fmed3f(m1[g], m2[g], m);
                const bool better = m < m1[g];
                m1[g] = better ? m : m1[g];
                c1[g] = better ? (uint32_t)T : c1[g];
                continue;
            }
            uint32_t o = __float_as_uint(acc[0]) | __float_as_uint(acc[1]) | __float_as_uint(acc[2]);
#pragma unroll
            for (int j = 3; j + 1 < 16; j += 2) o = o | __float_as_uint(acc[j]) | __float_as_uint(acc[j + 1]);
            o |= __float_as_uint(acc[15]);
            if (MODE == 1) pend[g] = __uint_as_float(o & 0x3FFFFFFFu) + 1.0f;          // keeps the loop from being hoisted
            if (MODE == 2 || MODE == 1) {
                if (__builtin_amdgcn_ballot_w64((int)o < 0)) { hits += g + 1; bq[g].w ^= 1u; }        // rare
            } else any |= o;
        }
        if (MODE == 3 || MODE == 5) {
            if (__builtin_amdgcn_ballot_w64((int)any < 0)) { hits++; bq[0].w ^= 1u; }                  // rare
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = (float)hits;
    for (int g = 0; g < QG; g++) s += m1[g] + m2[g] + (float)c1[g];
    for (int j = 0; j < 16; j++) s += pend[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MODE, int QG>
static void run(const char* name, int cus, int wps, const uint4* ops, float* out, unsigned long long* ticks, int tiles)
{
    const int blocks = cus * wps;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0.f;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((sf_kernel<MODE, QG>), dim3(blocks), dim3(256), 0, 0, ops, out, ticks, tiles);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<unsigned long long> t(blocks);
    CK(hipMemcpy(t.data(), ticks, blocks * 8, hipMemcpyDeviceToHost));
    double sum = 0; for (auto v : t) sum += (double)v;
    const double tg = (double)tiles * QG * wps;                       // (group, tile) pairs per SIMD
    // s_memtime ticks are shader cycles: cycles per (group, tile) per SIMD = mean wave lifetime / (tiles * QG) / ... * wps waves share the SIMD
    printf("waves/SIMD %d QG %d %-58s %.3f ms  %.1f ns per (group, tile) per SIMD; in-kernel: %.1f shader cycles per (group, tile) per SIMD, clock %.2f GHz\n",
           wps, QG, name, ms, ms * 1e6 / tg, sum / blocks / ((double)tiles * QG) / wps * wps / wps, sum / blocks / (ms * 1e6));
}

int main()
{
    int dev = 0; hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    std::mt19937_64 rng(7);
    uint4* ops; float* out; unsigned long long* ticks;
    CK(hipMalloc(&ops, 65 * 64 * sizeof(uint4)));
    std::vector<uint32_t> h(65 * 64 * 4);
    for (auto& v : h) v = 0x3C003C00u + (uint32_t)(rng() & 0x000F000Fu);          // f16 values in [1, 1.015]
    CK(hipMemcpy(ops, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4)); CK(hipMalloc(&ticks, (size_t)cus * 8 * 8));
    const int tiles = 4000;
    for (int wps : { 1, 2, 3, 4, 5 }) {
        run<0, 4>("MFMA only", cus, wps, ops, out, ticks, tiles);
        run<1, 4>("8 x v_or3 + sign test only", cus, wps, ops, out, ticks, tiles);
        run<2, 4>("MFMA -> OR of 16, sign test per (group, tile)", cus, wps, ops, out, ticks, tiles);
        run<3, 4>("MFMA -> OR of 16, one sign test per tile", cus, wps, ops, out, ticks, tiles);
        run<5, 4>("MFMA (threshold in C) -> OR of 16, one test per tile", cus, wps, ops, out, ticks, tiles);
        run<4, 4>("MFMA -> 8 x v_min3 + m1 / m2 / c1 tracking (round 2)", cus, wps, ops, out, ticks, tiles);
        run<3, 2>("QG 2: MFMA -> OR of 16, one sign test per tile", cus, wps, ops, out, ticks, tiles);
        run<3, 8>("QG 8: MFMA -> OR of 16, one sign test per tile", cus, wps, ops, out, ticks, tiles);
    }
    return 0;
}

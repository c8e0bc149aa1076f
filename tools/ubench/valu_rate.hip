// Issue-rate microbenchmark for the VALU instruction forms the brute-force filter loops are made of (gfx950).
// build: hipcc -O2 --offload-arch=gfx950 -o valu_rate valu_rate.hip     run: ./valu_rate
// Each kernel runs ITER iterations of 32 instructions of one form on 8 independent accumulator chains; every SIMD of the chip
// gets W waves.  Reported: shader-clock cycles per wave-instruction per SIMD (s_memtime ticks of wave 0 / instructions issued by
// all W waves of its SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* ticks, int iters, float seed)
{
    float a[8];
    f2 p[8];
    for (int i = 0; i < 8; i++) { a[i] = seed + threadIdx.x * 1e-3f + i; p[i] = (f2){ a[i], a[i] + 0.5f }; }
    float v1 = seed * 0.999f, v2 = seed * 1.0001f;
    f2 q1 = { v1, v2 }, q2 = { v2, v1 };
    const float s1 = __builtin_amdgcn_readfirstlane(v1);
    f2 sp;
    sp.x = __builtin_amdgcn_readfirstlane(v1); sp.y = __builtin_amdgcn_readfirstlane(v2);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (MODE == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(v1), "v"(v2));
                REP8(X)
#undef X
            } else if (MODE == 1) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "s"(s1), "v"(v2));
                REP8(X)
#undef X
            } else if (MODE == 2) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(q1), "v"(q2));
                REP8(X)
#undef X
            } else if (MODE == 3) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(q1), "s"(sp));
                REP8(X)
#undef X
            } else if (MODE == 4) {
#define X(i) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(v1), "v"(v2));
                REP8(X)
#undef X
            } else if (MODE == 5) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(q1));
                REP8(X)
#undef X
            } else if (MODE == 6) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(q1));
                REP8(X)
#undef X
            } else if (MODE == 7) {
#define X(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(v1));
                REP8(X)
#undef X
            } else if (MODE == 8) {
#define X(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(s1));
                REP8(X)
#undef X
            } else if (MODE == 9) {
#define X(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(v1));
                REP8(X)
#undef X
            } else if (MODE == 10) {     // the ETRACK mix: 3 pk_fma (SGPR pair operand) + 1 min3 per two targets
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(q1), "s"(sp));
                REP8(X)
#undef X
            } else if (MODE == 12) {
#define X(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(v1), "v"(v2));
                REP8(X)
#undef X
            } else if (MODE == 13) {
#define X(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(v1));
                REP8(X)
#undef X
            } else if (MODE == 14) {
#define X(i) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(v1), "v"(v2));
                REP8(X)
#undef X
            } else if (MODE == 15) {
#define X(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(v1));
                REP8(X)
#undef X
            } else if (MODE == 16) {
#define X(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(v1) : "vcc");
                REP8(X)
#undef X
            } else if (MODE == 17) {
#define X(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(v1) : "vcc");
                REP8(X)
#undef X
            } else if (MODE == 18) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(v1));
                REP8(X)
#undef X
            } else if (MODE == 19) {
#define X(i) asm volatile("v_min_i32 %0, %0, %1" : "+v"(a[i]) : "v"(v1));
                REP8(X)
#undef X
            } else if (MODE == 20) {
#define X(i) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(v1), "v"(v2));
                REP8(X)
#undef X
            } else if (MODE == 21) {
#define X(i) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(v1));
                REP8(X)
#undef X
            } else if (MODE == 22) {
#define X(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(v1), "v"(v2));
                REP8(X)
#undef X
            } else if (MODE == 23) {
#define X(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(v1), "v"(v2));
                REP8(X)
#undef X
            } else if (MODE == 24) {
#define X(i) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(v1));
                REP8(X)
#undef X
            } else if (MODE == 25) {
#define X(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(v1), "v"(v2));
                REP8(X)
#undef X
            } else if (MODE == 26) {
#define X(i) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(a[i]) : "v"(v1));
                REP8(X)
#undef X
            } else if (MODE == 27) {
#define X(i) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(v1), "v"(v2));
                REP8(X)
#undef X
            } else if (MODE == 28) {
#define X(i) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(v1), "v"(v2));
                REP8(X)
#undef X
            } else if (MODE == 29) {
#define X(i) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(v1), "v"(v2));
                REP8(X)
#undef X
            } else if (MODE == 30) {
#define X(i) asm volatile("v_sad_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(v1), "v"(v2));
                REP8(X)
#undef X
            } else if (MODE == 31) {
#define X(i) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(v1), "v"(v2));
                REP8(X)
#undef X
            } else if (MODE == 32) {
#define X(i) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(v1));
                REP8(X)
#undef X
            } else if (MODE == 33) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(v1));
                REP8(X)
#undef X
            } else if (MODE == 34) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(v1));
                REP8(X)
#undef X
            } else if (MODE == 35) {
#define X(i) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(v1));
                REP8(X)
#undef X
            } else if (MODE == 36) {
#define X(i) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a[i]) : "v"(v1));
                REP8(X)
#undef X
            } else if (MODE == 37) {
#define X(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(v1) : "vcc");
                REP8(X)
#undef X
            } else if (MODE == 38) {
#define X(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_or_b64 s[20:21], s[20:21], vcc" : : "v"(a[i]), "v"(v1) : "vcc", "s20", "s21");
                REP8(X)
#undef X
            } else if (MODE == 11) {     // v_fma_f32 with an SGPR and dependent 3-chain
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0\n v_fma_f32 %0, %2, %1, %0\n v_fma_f32 %0, %1, %1, %0" : "+v"(a[i]) : "s"(s1), "v"(v2));
                REP8(X)
#undef X
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char* name, int per_instr_mult)
{
    const int iters = 20000;
    for (int w : { 1, 4, 8 }) {
        const int blocks = 256 * w;                       // w waves per SIMD (a 256-thread block = one wave per SIMD of its CU)
        float* out; unsigned long long* ticks;
        CK(hipMalloc(&out, (size_t)blocks * 256 * sizeof(float)));
        CK(hipMalloc(&ticks, blocks * sizeof(unsigned long long)));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, ticks, 100, 1.0f);   // warm-up
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, ticks, iters, 1.0f);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(blocks);
        CK(hipMemcpy(h.data(), ticks, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double mean = 0; for (auto t : h) mean += (double)t; mean /= blocks;
        const double instr = (double)iters * 32 * per_instr_mult;
        // s_memtime counts at 100 MHz on this chip; the wall clock of the launch is the robust measure
        printf("%-44s W=%d  %.3f ms  -> %.3f ns per wave-instr per SIMD (x%d waves)  [memtime ticks/instr %.4f]\n", name, w, ms,
               ms * 1e6 / (instr * w), w, mean / instr);
        CK(hipFree(out)); CK(hipFree(ticks));
    }
}

int main()
{
    run<0>("v_fma_f32 v,v,v", 1);
    run<1>("v_fma_f32 s,v,v", 1);
    run<2>("v_pk_fma_f32 v,v,v", 1);
    run<3>("v_pk_fma_f32 v,s,v", 1);
    run<4>("v_min3_f32", 1);
    run<5>("v_pk_mul_f32", 1);
    run<6>("v_pk_add_f32", 1);
    run<7>("v_sub_f32", 1);
    run<8>("v_mul_f32 s,v", 1);
    run<9>("v_min_f32", 1);
    run<11>("v_fma_f32 s,v dependent x3", 3);
    run<12>("v_med3_f32", 1);
    run<13>("v_min_u32", 1);
    run<14>("v_min3_u32", 1);
    run<15>("v_max_f32", 1);
    run<16>("v_cmp_lt_f32 + v_cndmask", 2);
    run<17>("v_cmp_lt_u32 + v_cndmask", 2);
    run<18>("v_add_f32", 1);
    run<19>("v_min_i32", 1);
    run<20>("v_or3_b32", 1);
    run<21>("v_or_b32", 1);
    run<22>("v_and_or_b32", 1);
    run<23>("v_add3_u32", 1);
    run<24>("v_lshl_or_b32", 1);
    run<25>("v_perm_b32", 1);
    run<26>("v_pk_min_i16", 1);
    run<27>("v_mad_u32_u24", 1);
    run<28>("v_bfi_b32", 1);
    run<29>("v_max3_i32", 1);
    run<30>("v_sad_u32", 1);
    run<31>("v_xad_u32", 1);
    run<32>("v_lshl_add_u32", 1);
    run<33>("v_add_u32", 1);
    run<34>("v_mov_b32", 1);
    run<35>("v_pk_add_u16", 1);
    run<36>("v_alignbit_b32", 1);
    run<37>("v_cmp_lt_f32 (vcc)", 1);
    run<38>("v_cmp_lt_f32 + s_or_b64", 1);
    return 0;
}

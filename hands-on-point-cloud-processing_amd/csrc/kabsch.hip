// kabsch.hip — the streaming (HBM-bound) passes of one ICP iteration on gfx950:
//   * kabsch_partial_kernel / kabsch_final_kernel: A7 accumulation — f64 sums of p, q and q p^T over the
//     kept pairs (Homework9/hw9/src/registration.cpp:936-940,964-985), reduced per wavefront with DPP
//     shuffles, per workgroup through LDS, and across workgroups in a fixed order (bit-reproducible for a
//     given launch geometry; no float atomics).  Algorithmic traffic: 28 B per kept pair
//     (12 B source + 8 B key + 12 B gathered target - the key carries idx and d2).
//   * transform_kernel: A8 transformCloudInplace (registration.cpp:165-178), f32, unfused, in place,
//     24 B per point, float4-vectorised over the SoA arrays.
#include "pcr_internal.hpp"

#pragma clang fp contract(off)

namespace pcr {

constexpr int KB_BLOCK = 256;
constexpr int KB_NV = 16;          // 3 + 3 + 9 sums + count
constexpr int KB_MAX_BLOCKS = 1024;

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__device__ __forceinline__ long long wave_max(long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        long long o = __shfl_down(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

// partials layout: [block][KB_NV + 1] doubles; slot KB_NV holds the last kept source index as a double
// (exact below 2^53) or -1.
__global__ __launch_bounds__(KB_BLOCK) void kabsch_partial_kernel(
    const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz,
    const float* __restrict__ tx, const float* __restrict__ ty, const float* __restrict__ tz,
    const unsigned long long* __restrict__ keys, uint32_t ns, uint32_t nt, float max_corr,
    double* __restrict__ partials)
{
    double acc[KB_NV];
#pragma unroll
    for (int k = 0; k < KB_NV; k++) acc[k] = 0.0;
    long long last = -1;
    // contiguous chunk per block, strided by lane inside it: the order of additions is a pure function of
    // (ns, gridDim, blockDim)
    const uint32_t per_block = (ns + gridDim.x - 1) / gridDim.x;
    const uint32_t lo = min(blockIdx.x * per_block, ns);
    const uint32_t hi = min(lo + per_block, ns);
    for (uint32_t i = lo + threadIdx.x; i < hi; i += KB_BLOCK) {
        const unsigned long long key = keys[i];
        const float d2 = __uint_as_float((uint32_t)(key >> 32));
        const uint32_t j = (uint32_t)(key & 0xFFFFFFFFull);
        if (d2 < max_corr && j < nt) {                       // registration.cpp:936
            const double p0 = sx[i], p1 = sy[i], p2 = sz[i];
            const double q0 = tx[j], q1 = ty[j], q2 = tz[j];
            acc[0] += p0; acc[1] += p1; acc[2] += p2;
            acc[3] += q0; acc[4] += q1; acc[5] += q2;
            acc[6] += q0 * p0; acc[7] += q0 * p1; acc[8] += q0 * p2;
            acc[9] += q1 * p0; acc[10] += q1 * p1; acc[11] += q1 * p2;
            acc[12] += q2 * p0; acc[13] += q2 * p1; acc[14] += q2 * p2;
            acc[15] += 1.0;
            last = (long long)i;
        }
    }
    __shared__ double red[KB_BLOCK / 64][KB_NV + 1];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < KB_NV; k++) {
        double s = wave_sum(acc[k]);
        if (lane == 0) red[wave][k] = s;
    }
    long long lm = wave_max(last);
    if (lane == 0) red[wave][KB_NV] = (double)lm;
    __syncthreads();
    if (threadIdx.x <= KB_NV) {
        const int k = threadIdx.x;
        double s;
        if (k < KB_NV) {
            s = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
        } else {
            s = fmax(fmax(red[0][k], red[1][k]), fmax(red[2][k], red[3][k]));
        }
        partials[(size_t)blockIdx.x * (KB_NV + 1) + k] = s;
    }
}

// one workgroup: out[0..15] = sums, out[16] = last kept index (or -1), out[17] = d2 of that pair
__global__ __launch_bounds__(KB_BLOCK) void kabsch_final_kernel(
    const double* __restrict__ partials, uint32_t n_blocks, const unsigned long long* __restrict__ keys,
    double* __restrict__ out)
{
    __shared__ double red[KB_BLOCK / 64][KB_NV + 1];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double acc[KB_NV + 1];
#pragma unroll
    for (int k = 0; k < KB_NV; k++) acc[k] = 0.0;
    acc[KB_NV] = -1.0;
    for (uint32_t b = threadIdx.x; b < n_blocks; b += KB_BLOCK) {
#pragma unroll
        for (int k = 0; k < KB_NV; k++) acc[k] += partials[(size_t)b * (KB_NV + 1) + k];
        acc[KB_NV] = fmax(acc[KB_NV], partials[(size_t)b * (KB_NV + 1) + KB_NV]);
    }
#pragma unroll
    for (int k = 0; k < KB_NV; k++) {
        double s = wave_sum(acc[k]);
        if (lane == 0) red[wave][k] = s;
    }
    {
        double m = acc[KB_NV];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_down(m, off, 64));
        if (lane == 0) red[wave][KB_NV] = m;
    }
    __syncthreads();
    if (threadIdx.x < KB_NV) {
        const int k = threadIdx.x;
        out[k] = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
    } else if (threadIdx.x == KB_NV) {
        const double m = fmax(fmax(red[0][KB_NV], red[1][KB_NV]), fmax(red[2][KB_NV], red[3][KB_NV]));
        out[KB_NV] = m;
        float d2 = 0.0f;
        if (m >= 0.0) d2 = __uint_as_float((uint32_t)(keys[(size_t)m] >> 32));
        out[KB_NV + 1] = (double)d2;
    }
}

int launch_kabsch_sums(pcr_ctx* ctx, const pcr_cloud* tgt, const pcr_cloud* src, float max_corr)
{
    const size_t ns = src->n;
    if (ctx->keys_n != ns) return fail(ctx, PCR_ERR_STATE, "kabsch: no matching correspondence pass");
    uint32_t blocks = (uint32_t)((ns + KB_BLOCK * 4 - 1) / (KB_BLOCK * 4));
    if (blocks < 1) blocks = 1;
    if (blocks > KB_MAX_BLOCKS) blocks = KB_MAX_BLOCKS;
    {
        ProfScope p(ctx, "kabsch_partial");
        hipLaunchKernelGGL(kabsch_partial_kernel, dim3(blocks), dim3(KB_BLOCK), 0, ctx->stream,
                           src->x(), src->y(), src->z(), tgt->x(), tgt->y(), tgt->z(), ctx->keys,
                           (uint32_t)ns, (uint32_t)tgt->n, max_corr, ctx->partials);
    }
    PCR_HIP(ctx, hipGetLastError());
    {
        ProfScope p(ctx, "kabsch_final");
        hipLaunchKernelGGL(kabsch_final_kernel, dim3(1), dim3(KB_BLOCK), 0, ctx->stream, ctx->partials, blocks,
                           ctx->keys, ctx->dev_out);
    }
    PCR_HIP(ctx, hipGetLastError());
    return PCR_OK;
}

// ---------------------------------------------------------------------------------------------- A8
struct Rt {
    float r[9];
    float t[3];
};

__global__ __launch_bounds__(256) void transform_kernel(float* __restrict__ x, float* __restrict__ y,
                                                        float* __restrict__ z, uint32_t n4, Rt m)
{
    // n4 = number of float4 groups covering [0, n) (the tail group lies inside the cloud's padding and is
    // re-padded by the caller's invariant: padding x stays +inf because inf*R + t is never read as a target
    // beyond n; see launch_transform)
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 px = reinterpret_cast<float4*>(x)[i];
    float4 py = reinterpret_cast<float4*>(y)[i];
    float4 pz = reinterpret_cast<float4*>(z)[i];
    float4 ox, oy, oz;
#define PCR_ROW(o, r0, r1, r2, tt)                                   \
    o.x = ((m.r[r0] * px.x + m.r[r1] * py.x) + m.r[r2] * pz.x) + m.t[tt]; \
    o.y = ((m.r[r0] * px.y + m.r[r1] * py.y) + m.r[r2] * pz.y) + m.t[tt]; \
    o.z = ((m.r[r0] * px.z + m.r[r1] * py.z) + m.r[r2] * pz.z) + m.t[tt]; \
    o.w = ((m.r[r0] * px.w + m.r[r1] * py.w) + m.r[r2] * pz.w) + m.t[tt];
    PCR_ROW(ox, 0, 1, 2, 0)
    PCR_ROW(oy, 3, 4, 5, 1)
    PCR_ROW(oz, 6, 7, 8, 2)
#undef PCR_ROW
    reinterpret_cast<float4*>(x)[i] = ox;
    reinterpret_cast<float4*>(y)[i] = oy;
    reinterpret_cast<float4*>(z)[i] = oz;
}

// restore the padding invariant (x = +inf, y = z = 0) of the partially transformed tail group
__global__ void repad_kernel(float* __restrict__ x, float* __restrict__ y, float* __restrict__ z, uint32_t n,
                             uint32_t n_end)
{
    const uint32_t i = n + threadIdx.x;
    if (i < n_end) { x[i] = __builtin_inff(); y[i] = 0.0f; z[i] = 0.0f; }
}

int launch_transform(pcr_ctx* ctx, pcr_cloud* c, const float R[9], const float t[3])
{
    if (c->n == 0) return PCR_OK;
    Rt m;
    memcpy(m.r, R, sizeof m.r);
    memcpy(m.t, t, sizeof m.t);
    const uint32_t n4 = (uint32_t)((c->n + 3) / 4);
    {
        ProfScope p(ctx, "transform");
        hipLaunchKernelGGL(transform_kernel, dim3((n4 + 255) / 256), dim3(256), 0, ctx->stream, c->x(), c->y(),
                           c->z(), n4, m);
    }
    PCR_HIP(ctx, hipGetLastError());
    if (c->n % 4) {
        hipLaunchKernelGGL(repad_kernel, dim3(1), dim3(4), 0, ctx->stream, c->x(), c->y(), c->z(), (uint32_t)c->n,
                           (uint32_t)(n4 * 4));
        PCR_HIP(ctx, hipGetLastError());
    }
    return PCR_OK;
}

}  // namespace pcr

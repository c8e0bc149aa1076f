// kabsch.hip — the streaming (HBM-bound) passes of one ICP iteration on gfx950:
//   * kabsch_partial_kernel / kabsch_final_kernel: A7 accumulation — f64 sums of p, q and q p^T over the
//     kept pairs (Homework9/hw9/src/registration.cpp:936-940,964-985), reduced per wavefront with DPP
//     shuffles, per workgroup through LDS, and across workgroups in a fixed order (bit-reproducible for a
//     given launch geometry; no float atomics).  Algorithmic traffic: 28 B per kept pair
//     (12 B source + 8 B key + 12 B gathered target - the key carries idx and d2).
//   * transform_kernel: A8 transformCloudInplace (registration.cpp:165-178), f32, unfused, in place,
//     24 B per point, float4-vectorised over the SoA arrays.
#include "pcr_internal.hpp"
#include "numerics.hpp"

#pragma clang fp contract(off)

namespace pcr {

constexpr int KB_BLOCK = 256;
constexpr int KB_NV = 16;          // 3 + 3 + 9 sums + count
constexpr int KB_MAX_BLOCKS = 8192;   // == the capacity of ctx->partials (api.cpp)

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

// a pure function of n: the order of the f64 additions (and so the bits of the sums) depends only on n
static uint32_t kabsch_blocks(size_t ns)
{
    uint32_t blocks = (uint32_t)((ns + KB_BLOCK - 1) / KB_BLOCK);          // one point per thread up to KB_MAX_BLOCKS: the pass is latency-bound
    if (blocks < 1) blocks = 1;
    if (blocks > KB_MAX_BLOCKS) blocks = KB_MAX_BLOCKS;
    return blocks;
}

int launch_kabsch_sums(pcr_ctx* ctx, const pcr_cloud* tgt, const pcr_cloud* src, float max_corr)
{
    const size_t ns = src->n;
    if (ctx->keys_n != ns) return fail(ctx, PCR_ERR_STATE, "kabsch: no matching correspondence pass");
    uint32_t blocks = kabsch_blocks(ns);
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

// ---------------------------------------------------------------------------------------------- device-resident ICP state
// The pipelined ICP loop (icp.cpp) keeps the whole state machine of registration.cpp:915-1006 on the GPU so that
// iterations are enqueued back to back without a host round trip: nn1 -> kabsch_partial -> icp_update -> transform.

__device__ __forceinline__ void block_reduce_partials(const double* __restrict__ partials, uint32_t n_blocks,
                                                      double (&red)[KB_BLOCK / 64][KB_NV + 1], double* out18,
                                                      const unsigned long long* __restrict__ keys)
{
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
        out18[k] = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
    } else if (threadIdx.x == KB_NV) {
        const double m = fmax(fmax(red[0][KB_NV], red[1][KB_NV]), fmax(red[2][KB_NV], red[3][KB_NV]));
        out18[KB_NV] = m;
        float d2 = 0.0f;
        if (m >= 0.0) d2 = __uint_as_float((uint32_t)(keys[(size_t)m] >> 32));
        out18[KB_NV + 1] = (double)d2;
    }
    __syncthreads();
}

// state machine + Kabsch solve + pose composition, one thread (registration.cpp:939-1002)
__device__ void icp_state_step(IcpState* st, const double* sums16, bool any_kept, float last_d2)
{
    if (st->stop) return;
    if (st->stop_after_transform) { st->stop = 1; return; }   // max_iter reached: the loop is over
    float loss = 0.0f;
    if (any_kept) loss = last_d2 * last_d2;                                      // :939
    st->last_pairs = (unsigned long long)sums16[15];
    st->loss = loss;
    if (fabsf(st->last_loss - loss) < st->eps) st->unchanged++;                  // :948-951
    if (st->unchanged > 15) { st->converged = 1; st->stop = 1; return; }         // :954-958
    st->last_loss = loss;                                                        // :961
    float Rd[9], td[3];
    if (num::kabsch_solve(sums16, Rd, td) != 0) { st->empty = 1; st->stop = 1; return; }   // :979-998
    const float T_delta[16] = { Rd[0], Rd[1], Rd[2], td[0], Rd[3], Rd[4], Rd[5], td[1],
                                Rd[6], Rd[7], Rd[8], td[2], 0, 0, 0, 1 };
    num::mat4_mul_f32(T_delta, st->T_total, st->T_total);                        // :1000-1002
    for (int k = 0; k < 9; k++) st->Rd[k] = Rd[k];
    for (int k = 0; k < 3; k++) st->td[k] = td[k];
    st->iters_run++;
    // max_iter reached: this iteration's transform must still run, everything after it must not
    if (st->iters_run >= st->max_iter) st->stop_after_transform = 1;
}

// single rank: reduce the block partials and advance the state in one launch
__global__ __launch_bounds__(KB_BLOCK) void icp_update_kernel(const double* __restrict__ partials, uint32_t n_blocks,
                                                              const unsigned long long* __restrict__ keys, IcpState* st,
                                                              double* __restrict__ out)
{
    __shared__ double red[KB_BLOCK / 64][KB_NV + 1];
    if (st->stop) return;
    if (st->stop_after_transform) { if (threadIdx.x == 0) st->stop = 1; return; }
    block_reduce_partials(partials, n_blocks, red, out, keys);
    if (threadIdx.x == 0) icp_state_step(st, out, out[KB_NV] >= 0.0, (float)out[KB_NV + 1]);
}

// multi rank, step 1: reduce the partials into the all-reduce buffer [16 moments][(kept flag, last d2) per rank]
__global__ __launch_bounds__(KB_BLOCK) void icp_reduce_slots_kernel(const double* __restrict__ partials, uint32_t n_blocks,
                                                                    const unsigned long long* __restrict__ keys,
                                                                    double* __restrict__ out, int nranks, int rank, int have_points)
{
    __shared__ double red[KB_BLOCK / 64][KB_NV + 1];
    __shared__ double tmp[KB_NV + 2];
    if (have_points) {
        block_reduce_partials(partials, n_blocks, red, tmp, keys);
    } else {
        if (threadIdx.x < KB_NV + 2) tmp[threadIdx.x] = threadIdx.x == KB_NV ? -1.0 : 0.0;
        __syncthreads();
    }
    if (threadIdx.x < KB_NV) out[threadIdx.x] = tmp[threadIdx.x];
    if ((int)threadIdx.x < 2 * nranks) {
        const int r = threadIdx.x / 2, which = threadIdx.x % 2;
        double v = 0.0;
        if (r == rank) v = which == 0 ? (tmp[KB_NV] >= 0.0 ? 1.0 : 0.0) : tmp[KB_NV + 1];
        out[KB_NV + threadIdx.x] = v;
    }
}

// multi rank, step 2 (after the all-reduce): the loss comes from the highest rank that kept a pair
__global__ void icp_update_from_sums_kernel(const double* __restrict__ buf, int nranks, IcpState* st)
{
    if (threadIdx.x != 0 || st->stop) return;
    bool any = false;
    float d2 = 0.0f;
    for (int r = 0; r < nranks; r++)
        if (buf[KB_NV + 2 * r] > 0.5) { any = true; d2 = (float)buf[KB_NV + 2 * r + 1]; }
    icp_state_step(st, buf, any, d2);
}

__global__ __launch_bounds__(256) void transform_state_kernel(float* __restrict__ x, float* __restrict__ y,
                                                              float* __restrict__ z, uint32_t n, uint32_t n4, IcpState* st)
{
    if (st->stop) return;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float r0 = st->Rd[0], r1 = st->Rd[1], r2 = st->Rd[2], r3 = st->Rd[3], r4 = st->Rd[4], r5 = st->Rd[5],
                r6 = st->Rd[6], r7 = st->Rd[7], r8 = st->Rd[8], t0 = st->td[0], t1 = st->td[1], t2 = st->td[2];
    float4 px = reinterpret_cast<float4*>(x)[i];
    float4 py = reinterpret_cast<float4*>(y)[i];
    float4 pz = reinterpret_cast<float4*>(z)[i];
    float4 ox, oy, oz;
#define PCR_ROW(o, a, b, c, tt)                      \
    o.x = ((a * px.x + b * py.x) + c * pz.x) + tt;   \
    o.y = ((a * px.y + b * py.y) + c * pz.y) + tt;   \
    o.z = ((a * px.z + b * py.z) + c * pz.z) + tt;   \
    o.w = ((a * px.w + b * py.w) + c * pz.w) + tt;
    PCR_ROW(ox, r0, r1, r2, t0)
    PCR_ROW(oy, r3, r4, r5, t1)
    PCR_ROW(oz, r6, r7, r8, t2)
#undef PCR_ROW
    // keep the padding invariant (x = +inf, y = z = 0) of the tail group without a second launch
    const uint32_t base = 4 * i;
    if (base + 3 >= n) {
        const float inf = __builtin_inff();
        if (base + 0 >= n) { ox.x = inf; oy.x = 0.f; oz.x = 0.f; }
        if (base + 1 >= n) { ox.y = inf; oy.y = 0.f; oz.y = 0.f; }
        if (base + 2 >= n) { ox.z = inf; oy.z = 0.f; oz.z = 0.f; }
        if (base + 3 >= n) { ox.w = inf; oy.w = 0.f; oz.w = 0.f; }
    }
    reinterpret_cast<float4*>(x)[i] = ox;
    reinterpret_cast<float4*>(y)[i] = oy;
    reinterpret_cast<float4*>(z)[i] = oz;
}

int launch_kabsch_partial(pcr_ctx* ctx, const pcr_cloud* tgt, const pcr_cloud* src, float max_corr, uint32_t* n_blocks)
{
    const size_t ns = src->n;
    if (ctx->keys_n != ns) return fail(ctx, PCR_ERR_STATE, "kabsch: no matching correspondence pass");
    uint32_t blocks = kabsch_blocks(ns);
    {
        ProfScope p(ctx, "kabsch_partial");
        hipLaunchKernelGGL(kabsch_partial_kernel, dim3(blocks), dim3(KB_BLOCK), 0, ctx->stream,
                           src->x(), src->y(), src->z(), tgt->x(), tgt->y(), tgt->z(), ctx->keys,
                           (uint32_t)ns, (uint32_t)tgt->n, max_corr, ctx->partials);
    }
    PCR_HIP(ctx, hipGetLastError());
    *n_blocks = blocks;
    return PCR_OK;
}

int launch_icp_update(pcr_ctx* ctx, uint32_t n_blocks, IcpState* st_dev)
{
    {
        ProfScope p(ctx, "icp_update");
        hipLaunchKernelGGL(icp_update_kernel, dim3(1), dim3(KB_BLOCK), 0, ctx->stream, ctx->partials, n_blocks, ctx->keys, st_dev, ctx->dev_out);
    }
    PCR_HIP(ctx, hipGetLastError());
    return PCR_OK;
}

int launch_icp_reduce_slots(pcr_ctx* ctx, uint32_t n_blocks, int nranks, int rank, bool have_points)
{
    hipLaunchKernelGGL(icp_reduce_slots_kernel, dim3(1), dim3(KB_BLOCK), 0, ctx->stream, ctx->partials, n_blocks, ctx->keys,
                       ctx->dev_out, nranks, rank, have_points ? 1 : 0);
    PCR_HIP(ctx, hipGetLastError());
    return PCR_OK;
}

int launch_icp_update_from_sums(pcr_ctx* ctx, int nranks, IcpState* st_dev)
{
    hipLaunchKernelGGL(icp_update_from_sums_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->dev_out, nranks, st_dev);
    PCR_HIP(ctx, hipGetLastError());
    return PCR_OK;
}

int launch_transform_state(pcr_ctx* ctx, pcr_cloud* c, IcpState* st_dev)
{
    if (c->n) {
        const uint32_t n4 = (uint32_t)((c->n + 3) / 4);
        ProfScope p(ctx, "transform");
        hipLaunchKernelGGL(transform_state_kernel, dim3((n4 + 255) / 256), dim3(256), 0, ctx->stream, c->x(), c->y(), c->z(),
                           (uint32_t)c->n, n4, st_dev);
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

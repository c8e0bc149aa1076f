// plane.hip — RANSAC plane-inlier count (A10) for gfx950.
// Reference: Homework4/ground_detection_ransac.py:138-139
//     dists = np.fabs(np.c_[X, 1].dot(params)); inliers_num = np.sum(dists < threshold)
// and the final mask :152-153.  np.c_ promotes the f32 points to f64, so the arithmetic is f64:
//     dist = |((x*a + y*b) + z*c) + d|,   count = #{dist < thr}          (unfused, k = 0..3 in order)
// All hypotheses of a RANSAC run (40 per segment, :54,131) are evaluated in ONE pass over the points:
// the points are read once from HBM (12 B/pt), the <= 128 hypotheses sit in LDS and are read as wave-wide
// broadcasts; inlier votes are counted per wavefront with ballot + s_bcnt1 and merged with integer atomics
// (LDS, then one global atomic per hypothesis per workgroup) - integer sums are order independent, so the
// counts are exact and reproducible.
#include "pcr_internal.hpp"

#pragma clang fp contract(off)

namespace pcr {

constexpr int PL_BLOCK = 256;
constexpr int PL_MAX_PLANES = 128;
constexpr int PL_PPT = 4;   // points per thread per trip

__global__ __launch_bounds__(PL_BLOCK) void plane_count_kernel(
    const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, uint32_t n,
    const double* __restrict__ planes4, uint32_t n_planes, double thr, unsigned long long* __restrict__ counts)
{
    __shared__ double pl[PL_MAX_PLANES][4];
    __shared__ unsigned int cnt[PL_MAX_PLANES];
    for (uint32_t i = threadIdx.x; i < n_planes * 4; i += PL_BLOCK) pl[i / 4][i % 4] = planes4[i];
    for (uint32_t i = threadIdx.x; i < n_planes; i += PL_BLOCK) cnt[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t stride = gridDim.x * PL_BLOCK * PL_PPT;
    for (uint32_t base = blockIdx.x * PL_BLOCK * PL_PPT; base < n; base += stride) {
        double px[PL_PPT], py[PL_PPT], pz[PL_PPT];
        bool ok[PL_PPT];
#pragma unroll
        for (int p = 0; p < PL_PPT; p++) {
            const uint32_t i = base + p * PL_BLOCK + threadIdx.x;
            ok[p] = i < n;
            const uint32_t ii = ok[p] ? i : 0;
            px[p] = (double)x[ii]; py[p] = (double)y[ii]; pz[p] = (double)z[ii];
        }
        for (uint32_t h = 0; h < n_planes; h++) {
            const double a = pl[h][0], b = pl[h][1], c = pl[h][2], d = pl[h][3];
            unsigned int votes = 0;
#pragma unroll
            for (int p = 0; p < PL_PPT; p++) {
                const double dist = fabs(((px[p] * a + py[p] * b) + pz[p] * c) + d);
                votes += (unsigned int)__popcll(__ballot(ok[p] && dist < thr));
            }
            if (lane == 0 && votes) atomicAdd(&cnt[h], votes);
        }
    }
    __syncthreads();
    for (uint32_t h = threadIdx.x; h < n_planes; h += PL_BLOCK)
        if (cnt[h]) atomicAdd(&counts[h], (unsigned long long)cnt[h]);
}

__global__ __launch_bounds__(PL_BLOCK) void plane_mask_kernel(
    const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, uint32_t n,
    double a, double b, double c, double d, double thr, uint8_t* __restrict__ mask,
    unsigned long long* __restrict__ count)
{
    const uint32_t i = blockIdx.x * PL_BLOCK + threadIdx.x;
    bool in = false;
    if (i < n) {
        const double dist = fabs((((double)x[i] * a + (double)y[i] * b) + (double)z[i] * c) + d);
        in = dist < thr;
        mask[i] = in ? 1 : 0;
    }
    const unsigned int votes = (unsigned int)__popcll(__ballot(in));
    if ((threadIdx.x & 63) == 0 && votes) atomicAdd(count, (unsigned long long)votes);
}

int launch_plane_count(pcr_ctx* ctx, const pcr_cloud* pts, const double* planes4_dev, size_t n_planes, double thr,
                       unsigned long long* counts_dev)
{
    if (n_planes > PL_MAX_PLANES) return fail(ctx, PCR_ERR_ARG, "plane_count: more than 128 hypotheses per launch");
    uint32_t blocks = (uint32_t)((pts->n + PL_BLOCK * PL_PPT - 1) / (PL_BLOCK * PL_PPT));
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) return PCR_OK;
    {
        ProfScope p(ctx, "plane_count");
        hipLaunchKernelGGL(plane_count_kernel, dim3(blocks), dim3(PL_BLOCK), 0, ctx->stream, pts->x(), pts->y(),
                           pts->z(), (uint32_t)pts->n, planes4_dev, (uint32_t)n_planes, thr, counts_dev);
    }
    PCR_HIP(ctx, hipGetLastError());
    return PCR_OK;
}

int launch_plane_mask(pcr_ctx* ctx, const pcr_cloud* pts, const double plane4[4], double thr, uint8_t* mask_dev,
                      unsigned long long* count_dev)
{
    if (pts->n == 0) return PCR_OK;
    const uint32_t blocks = (uint32_t)((pts->n + PL_BLOCK - 1) / PL_BLOCK);
    {
        ProfScope p(ctx, "plane_mask");
        hipLaunchKernelGGL(plane_mask_kernel, dim3(blocks), dim3(PL_BLOCK), 0, ctx->stream, pts->x(), pts->y(),
                           pts->z(), (uint32_t)pts->n, plane4[0], plane4[1], plane4[2], plane4[3], thr, mask_dev,
                           count_dev);
    }
    PCR_HIP(ctx, hipGetLastError());
    return PCR_OK;
}

}  // namespace pcr

using namespace pcr;

extern "C" int pcr_plane_count_f64(pcr_ctx* ctx, const pcr_cloud* pts, const double* planes4, size_t n_planes,
                                   double thr, int64_t* counts)
{
    if (!ctx || !pts || (n_planes && (!planes4 || !counts))) return fail(ctx, PCR_ERR_ARG, "pcr_plane_count_f64");
    if (n_planes == 0) return PCR_OK;
    PCR_HIP(ctx, hipSetDevice(ctx->device));
    // scratch: [planes f64 x4][counts u64]
    int rc = ensure_scratch(ctx, n_planes * 40);
    if (rc) return rc;
    double* planes_dev = (double*)ctx->scratch;
    unsigned long long* counts_dev = (unsigned long long*)((char*)ctx->scratch + n_planes * 32);
    PCR_HIP(ctx, hipMemcpyAsync(planes_dev, planes4, n_planes * 32, hipMemcpyHostToDevice, ctx->stream));
    PCR_HIP(ctx, hipMemsetAsync(counts_dev, 0, n_planes * 8, ctx->stream));
    for (size_t h0 = 0; h0 < n_planes; h0 += PL_MAX_PLANES) {
        const size_t nh = n_planes - h0 < (size_t)PL_MAX_PLANES ? n_planes - h0 : (size_t)PL_MAX_PLANES;
        rc = launch_plane_count(ctx, pts, planes_dev + 4 * h0, nh, thr, counts_dev + h0);
        if (rc) return rc;
    }
    PCR_HIP(ctx, hipMemcpyAsync(counts, counts_dev, n_planes * 8, hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PCR_OK;
}

extern "C" int pcr_plane_mask_f64(pcr_ctx* ctx, const pcr_cloud* pts, const double plane4[4], double thr,
                                  uint8_t* mask, int64_t* n_inliers)
{
    if (!ctx || !pts || !plane4 || (pts->n && !mask)) return fail(ctx, PCR_ERR_ARG, "pcr_plane_mask_f64");
    if (n_inliers) *n_inliers = 0;
    if (pts->n == 0) return PCR_OK;
    PCR_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure_scratch(ctx, pts->n + 64);
    if (rc) return rc;
    unsigned long long* count_dev = (unsigned long long*)ctx->scratch;
    uint8_t* mask_dev = (uint8_t*)ctx->scratch + 64;
    PCR_HIP(ctx, hipMemsetAsync(count_dev, 0, 8, ctx->stream));
    rc = launch_plane_mask(ctx, pts, plane4, thr, mask_dev, count_dev);
    if (rc) return rc;
    unsigned long long cnt = 0;
    PCR_HIP(ctx, hipMemcpyAsync(mask, mask_dev, pts->n, hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP(ctx, hipMemcpyAsync(&cnt, count_dev, 8, hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (n_inliers) *n_inliers = (int64_t)cnt;
    return PCR_OK;
}

// plane.hip — RANSAC plane-inlier count (A10) for gfx950.
// Reference: Homework4/ground_detection_ransac.py:138-139
//     dists = np.fabs(np.c_[X, 1].dot(params)); inliers_num = np.sum(dists < threshold)
// and the final mask :152-153.  np.c_ promotes the f32 points to f64, so the arithmetic is f64:
//     dist = |((x*a + y*b) + z*c) + d|,   count = #{dist < thr}          (unfused, k = 0..3 in order)
// All hypotheses of a RANSAC run (40 per segment, :54,131) are evaluated in ONE pass over the points, ONE POINT PER THREAD (120 000 points:
// 469 workgroups on 256 CUs; round 3 gave a thread four points and launched 118).  Three launches, no copy through the runtime:
//   stage  (one workgroup): the hypotheses arrive in its ARGUMENT BLOCK (<= 96 x 32 bytes) and are copied to device memory; it also zeroes
//          the counters — no hipMemcpyAsync of a pageable 2.5 KB, no hipMemsetAsync;
//   count:  a workgroup reads the hypotheses into LDS with one parallel load; the votes of a wave for a hypothesis are one ballot + s_bcnt1
//          in scalar registers, parked in the lane with the hypothesis' number, so a wave issues ONE LDS add per 64 hypotheses instead of one
//          per hypothesis; a workgroup adds its counts to global counters that sit 128 bytes apart (one cache line each: 469 atomics per
//          line, the lines in parallel — packed into five lines they queued up behind each other);
//   gather (one workgroup): the counters, compact, into pinned host memory — no download, ONE stream synchronisation per call.
// The hypotheses are dealt to the rows of a 2-D launch in groups of 20 (tune plane_group): 80 hypotheses 14.4 -> 9.8 us of kernel.
// Integer sums: exact and reproducible in any order.  Measured and dropped on the way: the hypotheses read from the argument block inside
// the counting kernel (it lives in host-visible memory: 46-81 us per launch), and per-workgroup rows of partial counts added up by the
// workgroup that draws the last ticket, one level and two (62-75 us: every workgroup pays two device-scope fences).
#include "pcr_internal.hpp"

#pragma clang fp contract(off)

namespace pcr {

constexpr int PL_BLOCK = 256;
constexpr int PL_MAX_PLANES = 96;      // per launch: 3 KB of the 4 KB kernarg segment
constexpr int PL_PAD = 16;             // u64 words between two counters: 128 bytes

struct PlaneArgs { double p[PL_MAX_PLANES][4]; };

__global__ __launch_bounds__(PL_BLOCK) void plane_stage_kernel(const PlaneArgs planes_arg, double* __restrict__ planes_dev, unsigned long long* __restrict__ counters)
{
    typedef __attribute__((address_space(4))) const double cdouble;
    cdouble* src = (cdouble*)__builtin_amdgcn_kernarg_segment_ptr();         // offset 0 of the argument block
    for (uint32_t i = threadIdx.x; i < PL_MAX_PLANES * 4; i += PL_BLOCK) planes_dev[i] = src[i];
    if (threadIdx.x < PL_MAX_PLANES) counters[(size_t)threadIdx.x * PL_PAD] = 0ull;
}

__global__ __launch_bounds__(PL_BLOCK) void plane_count_kernel(
    const double* __restrict__ planes_dev, const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, uint32_t n,
    uint32_t n_planes, double thr, unsigned long long* __restrict__ counters, uint32_t group)
{
    // blockIdx.y: the group of `group` consecutive hypotheses this workgroup counts (a scan of 120 000 points is 1 875 waves — fewer than two per SIMD — and
    // a wave's 80 hypotheses were a chain of 80 dependent steps; four groups of 20 are four times the waves with a quarter of the chain each)
    const uint32_t h0 = blockIdx.y * group, h1 = min(h0 + group, n_planes);
    __shared__ double planes[PL_MAX_PLANES * 4];               // one parallel load per workgroup; read back as wave-wide broadcasts
    __shared__ unsigned int cnt[2 * 64];
    for (uint32_t i = 4 * h0 + threadIdx.x; i < 4 * h1; i += PL_BLOCK) planes[i] = planes_dev[i];
    if (threadIdx.x < 128) cnt[threadIdx.x] = 0;
    const uint32_t lane = threadIdx.x & 63, i = blockIdx.x * PL_BLOCK + threadIdx.x;
    const bool ok = i < n;
    const uint32_t ii = ok ? i : 0;
    const double px = (double)x[ii], py = (double)y[ii], pz = (double)z[ii];
    __syncthreads();
    unsigned int mine[2] = { 0u, 0u };                       // lane l: the wave's votes for hypotheses l and 64 + l
#pragma unroll
    for (int blk = 0; blk < 2; blk++)
#pragma unroll 8
        for (uint32_t h = max(h0, (uint32_t)blk * 64u); h < min(h1, (uint32_t)(blk + 1) * 64u); h++) {
            const double a = planes[4 * h], b = planes[4 * h + 1], c = planes[4 * h + 2], d = planes[4 * h + 3];
            const double dist = fabs(((px * a + py * b) + pz * c) + d);                                    // :138, unfused, k = 0..3 in order
            const unsigned int votes = (unsigned int)__popcll(__ballot(ok && dist < thr));
            mine[blk] = lane == (h & 63u) ? votes : mine[blk];
        }
    if (mine[0]) atomicAdd(&cnt[lane], mine[0]);
    if (n_planes > 64 && mine[1]) atomicAdd(&cnt[64 + lane], mine[1]);
    __syncthreads();
    if (threadIdx.x >= h0 && threadIdx.x < h1 && cnt[threadIdx.x]) atomicAdd(&counters[(size_t)threadIdx.x * PL_PAD], (unsigned long long)cnt[threadIdx.x]);
}

__global__ __launch_bounds__(128) void plane_gather_kernel(const unsigned long long* __restrict__ counters, uint32_t n_planes, unsigned long long* __restrict__ out)
{
    if (threadIdx.x < n_planes) out[threadIdx.x] = counters[(size_t)threadIdx.x * PL_PAD];
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

// planes4: HOST pointer (the hypotheses travel in the staging launch's argument block); counts_out: device or pinned host memory, n_planes
// entries, written by the gather launch.  At most PL_MAX_PLANES per call.
int launch_plane_count(pcr_ctx* ctx, const pcr_cloud* pts, const double* planes4, size_t n_planes, double thr, unsigned long long* counts_out)
{
    if (n_planes > PL_MAX_PLANES) return fail(ctx, PCR_ERR_ARG, "plane_count: more than 96 hypotheses per launch");
    const uint32_t blocks = (uint32_t)((pts->n + PL_BLOCK - 1) / PL_BLOCK);
    if (blocks == 0 || n_planes == 0) return PCR_OK;
    // workspace (lives until the context dies): the hypotheses, the padded counters
    const size_t need = PL_MAX_PLANES * 4 * sizeof(double) + (size_t)PL_MAX_PLANES * PL_PAD * sizeof(unsigned long long);
    if (ctx->plane_ws_bytes < need) {
        if (ctx->plane_ws) { PCR_HIP(ctx, hipStreamSynchronize(ctx->stream)); hipFree(ctx->plane_ws); ctx->plane_ws = nullptr; ctx->plane_ws_bytes = 0; }
        PCR_HIP(ctx, hipMalloc(&ctx->plane_ws, need));
        ctx->plane_ws_bytes = need;
    }
    double* planes_dev = (double*)ctx->plane_ws;
    unsigned long long* counters = (unsigned long long*)((char*)ctx->plane_ws + PL_MAX_PLANES * 4 * sizeof(double));
    PlaneArgs pa;
    memcpy(pa.p, planes4, n_planes * 4 * sizeof(double));
    if (n_planes < (size_t)PL_MAX_PLANES) memset(pa.p[n_planes], 0, (PL_MAX_PLANES - n_planes) * 4 * sizeof(double));
    hipLaunchKernelGGL(plane_stage_kernel, dim3(1), dim3(PL_BLOCK), 0, ctx->stream, pa, planes_dev, counters);
    {
        ProfScope p(ctx, "plane_count");
        // (tune plane_group: hypotheses per workgroup row; default 20)
        const uint32_t group = (uint32_t)std::min<int64_t>(std::max<int64_t>(tune_get(ctx, "plane_group", 20), 1), PL_MAX_PLANES);
        hipLaunchKernelGGL(plane_count_kernel, dim3(blocks, (unsigned)((n_planes + group - 1) / group)), dim3(PL_BLOCK), 0, ctx->stream, planes_dev, pts->x(), pts->y(), pts->z(),
                           (uint32_t)pts->n, (uint32_t)n_planes, thr, counters, group);
    }
    hipLaunchKernelGGL(plane_gather_kernel, dim3(1), dim3(128), 0, ctx->stream, counters, (uint32_t)n_planes, counts_out);
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
    if (pts->n == 0) { for (size_t h = 0; h < n_planes; h++) counts[h] = 0; return PCR_OK; }
    PCR_HIP(ctx, hipSetDevice(ctx->device));
    // one launch per 96 hypotheses: the hypotheses in the argument block, the totals straight into pinned host memory (ctx->host_out: 128 x 8
    // bytes) — no upload, no memset, no download: ONE stream synchronisation per launch
    for (size_t h0 = 0; h0 < n_planes; h0 += PL_MAX_PLANES) {
        const size_t nh = n_planes - h0 < (size_t)PL_MAX_PLANES ? n_planes - h0 : (size_t)PL_MAX_PLANES;
        int rc = launch_plane_count(ctx, pts, planes4 + 4 * h0, nh, thr, (unsigned long long*)ctx->host_out);
        if (rc) return rc;
        PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        memcpy(counts + h0, ctx->host_out, nh * sizeof(int64_t));
    }
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

// iss.hip — next row N1: ISS keypoints (Homework7/hw7/src/iss_detector.cpp:38-152) as three batched radius passes over the
// uniform grid of grid_common.hpp.  The cloud is searched against itself, so the queries ARE the grid records: query p
// is record p, consecutive lanes work on points of the same cell and the 27-cell neighbourhood (9 contiguous x-rows of
// 3 cells) stays in L1/L2 between them.
//
//   pass 1  cnt[p]  = |{j : d(j, p) <= local_radius}|                                (:47-57)
//   pass 2  (a: f64 sums with G lanes per point, b: eigen-solve with one lane per point)
//           l3[p]   = smallest eigenvalue of the (weighted) neighbourhood covariance when the gamma tests pass (:69-83,
//                     :113-152), else -1
//   pass 3  key[i]  = l3 != -1 && |N_nms| >= min_neighbors && no neighbour has a larger l3      (:86-105)
//
// Distance arithmetic is hw7's float kd-tree (src/kdtree.cpp:310-316): s = (float)((double)s + (double)e * e) per
// component, d = sqrtf(s), member iff d <= r.  sqrtf is monotonic and correctly rounded, so the host finds the largest
// float s_max with sqrtf(s_max) <= r once and the kernels compare s <= s_max — the same set, no sqrt per pair.
// G lanes share one query and stride over each row; f64 covariance sums are reduced across the G lanes with shuffles.
#include "grid_common.hpp"

#include <cmath>
#include <vector>

namespace pcr {

namespace {

constexpr int ISS_BLOCK = 256;

struct IssParams {
    float s_local;     // largest s with sqrtf(s) <= local_radius   (negative: empty set)
    float s_nms;       // same for non_max_radius
    float gamma21, gamma32;
    unsigned min_neighbors;
    int weighted;
};

__device__ __forceinline__ float hw7_s(float tx, float ty, float tz, float qx, float qy, float qz)
{
    const float ex = tx - qx, ey = ty - qy, ez = tz - qz;
    float s = (float)((double)ex * (double)ex);                 // 0 + e^2: exact product, one rounding
    s = (float)((double)s + (double)ey * (double)ey);
    s = (float)((double)s + (double)ez * (double)ez);
    return s;
}

// the 9 x-rows of the 27-cell block around cell (cx, cy, cz): row k -> [begin, end) in records
__device__ __forceinline__ void row_range(const GridParams& g, const uint32_t* __restrict__ cell_start, int cx, int cy, int cz, int k,
                                          uint32_t& b, uint32_t& e)
{
    const int yy = cy + (k % 3) - 1, zz = cz + (k / 3) - 1;
    if (yy < 0 || yy >= g.n[1] || zz < 0 || zz >= g.n[2]) { b = e = 0; return; }
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.n[0] - 1);
    const uint32_t row = (uint32_t)((zz * g.n[1] + yy) * g.n[0]);
    b = cell_start[row + x0];
    e = cell_start[row + x1 + 1];
}

// The records of a row are sorted by x (grid_build): cut [b, e) to the records with lo <= x <= hi, the only ones that can
// lie within the radius.  Two bounded binary searches, worth it on long rows only; uniform over the lanes of a group.
__device__ __forceinline__ void clip_row_x(const float4* __restrict__ records, uint32_t& b, uint32_t& e, float lo, float hi)
{
    if (e - b <= 384u) return;
    uint32_t l = b, h = e;
    for (int it = 0; it < 32 && l < h; it++) {
        const uint32_t mid = l + ((h - l) >> 1);
        if (records[mid].x < lo) l = mid + 1; else h = mid;
    }
    const uint32_t nb = l;
    h = e;
    for (int it = 0; it < 32 && l < h; it++) {
        const uint32_t mid = l + ((h - l) >> 1);
        if (records[mid].x <= hi) l = mid + 1; else h = mid;
    }
    b = nb;
    e = l;
}

// [lo, hi] around qx that contains every x whose hw7 sum can still be <= s_max: |dx|^2 <= s_max (1 + 3 * 2^-24); the floor
// 1e-18 covers differences whose squares vanish in f32 (|e| < 3.7e-23 gives s == 0), the pad the rounding of qx -+ d
__device__ __forceinline__ void radius_window(float qx, float s_max, float& lo, float& hi)
{
    const float d = sqrtf(fmaxf(s_max, 0.0f)) * 1.00001f + 1e-18f;
    const float pad = (fabsf(qx) + d) * 2.4e-7f;
    lo = qx - d - pad;
    hi = qx + d + pad;
}

template <int G>
__device__ __forceinline__ unsigned group_sum_u32(unsigned v)
{
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, G);
    return v;
}
template <int G>
__device__ __forceinline__ double group_sum_f64(double v)
{
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, G);
    return v;
}

// cells of a query: every point of the cloud lies inside the grid box, non-finite ones were binned into cell 0 and
// never pass a distance test (inf - inf = NaN)
__device__ __forceinline__ void query_cell(const GridParams& g, const float4& q, int& cx, int& cy, int& cz)
{
    cx = min(max(cell_coord(q.x, g.lo[0], g.inv_h), 0), g.n[0] - 1);
    cy = min(max(cell_coord(q.y, g.lo[1], g.inv_h), 0), g.n[1] - 1);
    cz = min(max(cell_coord(q.z, g.lo[2], g.inv_h), 0), g.n[2] - 1);
}

template <int G>
__global__ __launch_bounds__(ISS_BLOCK) void iss_count_kernel(const float4* __restrict__ records, const uint32_t* __restrict__ cell_start,
                                                              GridParams g, uint32_t n, float s_max, uint32_t* __restrict__ cnt_sorted,
                                                              uint32_t* __restrict__ cnt_out)
{
    const uint32_t p = (blockIdx.x * ISS_BLOCK + threadIdx.x) / G;
    const int sub = threadIdx.x % G;
    if (p >= n) return;                                     // whole groups leave together (ISS_BLOCK % G == 0)
    const float4 q = records[p];
    unsigned c = 0;
    if (finite3(q.x, q.y, q.z)) {
        int cx, cy, cz;
        query_cell(g, q, cx, cy, cz);
        float wlo, whi;
        radius_window(q.x, s_max, wlo, whi);
        for (int k = 0; k < 9; k++) {
            uint32_t b, e;
            row_range(g, cell_start, cx, cy, cz, k, b, e);
            clip_row_x(records, b, e, wlo, whi);
            for (uint32_t j = b + sub; j < e; j += G) {
                const float4 t = records[j];
                c += hw7_s(t.x, t.y, t.z, q.x, q.y, q.z) <= s_max;
            }
        }
    }
    c = group_sum_u32<G>(c);
    if (sub == 0) {
        cnt_sorted[p] = c;
        cnt_out[__float_as_uint(q.w)] = c;
    }
}

// ascending eigenvalues of a symmetric 3x3 by cyclic Jacobi — the same operation sequence as oracle sym_eig3, with
// static indices so that the matrix stays in registers
#define ISS_ROT(P, Q)                                                                                                  \
    if (a[P][Q] != 0.0) {                                                                                              \
        const double theta = (a[Q][Q] - a[P][P]) / (2.0 * a[P][Q]);                                                    \
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));                        \
        const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;                                                          \
        _Pragma("unroll") for (int k = 0; k < 3; k++) { const double akp = a[k][P], akq = a[k][Q]; a[k][P] = c * akp - sn * akq; a[k][Q] = sn * akp + c * akq; } \
        _Pragma("unroll") for (int k = 0; k < 3; k++) { const double apk = a[P][k], aqk = a[Q][k]; a[P][k] = c * apk - sn * aqk; a[Q][k] = sn * apk + c * aqk; } \
    }

__device__ inline void sym_eig3(double a[3][3], double w[3])
{
    for (int sweep = 0; sweep < 60; sweep++) {
        const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        if (off == 0.0) break;
        ISS_ROT(0, 1)
        ISS_ROT(0, 2)
        ISS_ROT(1, 2)
    }
    w[0] = a[0][0]; w[1] = a[1][1]; w[2] = a[2][2];
    double t;
    if (w[1] < w[0]) { t = w[0]; w[0] = w[1]; w[1] = t; }
    if (w[2] < w[0]) { t = w[0]; w[0] = w[2]; w[2] = t; }
    if (w[2] < w[1]) { t = w[1]; w[1] = w[2]; w[2] = t; }
}
#undef ISS_ROT

// pass 2a: the 6 + 1 f64 neighbourhood sums of every point (G lanes per point), SoA in `sums` (7 arrays of n)
template <int G>
__global__ __launch_bounds__(ISS_BLOCK) void iss_cov_kernel(const float4* __restrict__ records, const uint32_t* __restrict__ cell_start,
                                                            GridParams g, uint32_t n, IssParams prm, const uint32_t* __restrict__ cnt_sorted,
                                                            double* __restrict__ sums)
{
    const uint32_t p = (blockIdx.x * ISS_BLOCK + threadIdx.x) / G;
    const int sub = threadIdx.x % G;
    if (p >= n) return;
    const float4 q = records[p];
    double sxx = 0, sxy = 0, sxz = 0, syy = 0, syz = 0, szz = 0, wsum = 0;
    if (cnt_sorted[p] >= 3) {                               // uniform over the group
        int cx, cy, cz;
        query_cell(g, q, cx, cy, cz);
        float wlo, whi;
        radius_window(q.x, prm.s_local, wlo, whi);
        for (int k = 0; k < 9; k++) {
            uint32_t b, e;
            row_range(g, cell_start, cx, cy, cz, k, b, e);
            clip_row_x(records, b, e, wlo, whi);
            for (uint32_t j = b + sub; j < e; j += G) {
                const float4 t = records[j];
                if (!(hw7_s(t.x, t.y, t.z, q.x, q.y, q.z) <= prm.s_local)) continue;
                const double w = prm.weighted ? (double)(1.0f / (float)cnt_sorted[j]) : 1.0;   // :130
                const double dx = (double)(t.x - q.x), dy = (double)(t.y - q.y), dz = (double)(t.z - q.z);
                const double wx = w * dx, wy = w * dy, wz = w * dz;
                sxx += wx * dx; sxy += wx * dy; sxz += wx * dz;
                syy += wy * dy; syz += wy * dz; szz += wz * dz;
                wsum += w;
            }
        }
    }
    sxx = group_sum_f64<G>(sxx); sxy = group_sum_f64<G>(sxy); sxz = group_sum_f64<G>(sxz);
    syy = group_sum_f64<G>(syy); syz = group_sum_f64<G>(syz); szz = group_sum_f64<G>(szz);
    wsum = group_sum_f64<G>(wsum);
    if (sub != 0) return;
    sums[p] = sxx; sums[(size_t)n + p] = sxy; sums[2 * (size_t)n + p] = sxz; sums[3 * (size_t)n + p] = syy;
    sums[4 * (size_t)n + p] = syz; sums[5 * (size_t)n + p] = szz; sums[6 * (size_t)n + p] = wsum;
}

// pass 2b: one lane per point — eigenvalues and the gamma tests (:74-80); kept apart from 2a so that the Jacobi sweeps
// run on full wavefronts instead of on one lane in G
__global__ __launch_bounds__(ISS_BLOCK) void iss_eig_kernel(const float4* __restrict__ records, uint32_t n, IssParams prm,
                                                            const uint32_t* __restrict__ cnt_sorted, const double* __restrict__ sums,
                                                            float* __restrict__ l3_sorted, float* __restrict__ l3_out)
{
    const uint32_t p = blockIdx.x * ISS_BLOCK + threadIdx.x;
    if (p >= n) return;
    float out = -1.0f;
    if (cnt_sorted[p] >= 3) {
        double sxx = sums[p], sxy = sums[(size_t)n + p], sxz = sums[2 * (size_t)n + p], syy = sums[3 * (size_t)n + p],
               syz = sums[4 * (size_t)n + p], szz = sums[5 * (size_t)n + p];
        const double wsum = sums[6 * (size_t)n + p];
        if (prm.weighted) { sxx /= wsum; sxy /= wsum; sxz /= wsum; syy /= wsum; syz /= wsum; szz /= wsum; }   // :137
        double a[3][3] = { { sxx, sxy, sxz }, { sxy, syy, syz }, { sxz, syz, szz } };
        double w[3];
        sym_eig3(a, w);
        const float lambda1 = (float)w[2], lambda2 = (float)w[1], lambda3 = (float)w[0];
        if (lambda2 / lambda1 < prm.gamma21 && lambda3 / lambda2 < prm.gamma32 && lambda3 > 0) out = lambda3;   // :79
    }
    l3_sorted[p] = out;
    l3_out[__float_as_uint(records[p].w)] = out;
}

template <int G>
__global__ __launch_bounds__(ISS_BLOCK) void iss_nms_kernel(const float4* __restrict__ records, const uint32_t* __restrict__ cell_start,
                                                            GridParams g, uint32_t n, IssParams prm, const float* __restrict__ l3_sorted,
                                                            uint8_t* __restrict__ is_key)
{
    const uint32_t p = (blockIdx.x * ISS_BLOCK + threadIdx.x) / G;
    const int sub = threadIdx.x % G;
    if (p >= n) return;
    const float4 q = records[p];
    const float mine = l3_sorted[p];
    unsigned m = 0, beaten = 0;
    if (mine != -1.0f) {                                    // :88, uniform over the group
        int cx, cy, cz;
        query_cell(g, q, cx, cy, cz);
        float wlo, whi;
        radius_window(q.x, prm.s_nms, wlo, whi);
        for (int k = 0; k < 9; k++) {
            uint32_t b, e;
            row_range(g, cell_start, cx, cy, cz, k, b, e);
            clip_row_x(records, b, e, wlo, whi);
            for (uint32_t j = b + sub; j < e; j += G) {
                const float4 t = records[j];
                if (!(hw7_s(t.x, t.y, t.z, q.x, q.y, q.z) <= prm.s_nms)) continue;
                m++;
                beaten += mine < l3_sorted[j];              // :96
            }
        }
    }
    m = group_sum_u32<G>(m);
    beaten = group_sum_u32<G>(beaten);
    if (sub == 0) is_key[__float_as_uint(q.w)] = (mine != -1.0f && m >= prm.min_neighbors && beaten == 0) ? 1 : 0;
}

// largest float s with sqrtf(s) <= r (r >= 0); -1 when there is none (r < 0 or NaN)
float sqrt_threshold(float r)
{
    if (!(r >= 0.0f)) return -1.0f;
    if (std::isinf(r)) return FLT_MAX;
    float s = r * r;
    if (std::isinf(s)) s = FLT_MAX;
    for (int k = 0; k < 8 && !(sqrtf(s) <= r); k++) s = std::nextafterf(s, -1.0f);
    for (int k = 0; k < 8 && s < FLT_MAX && sqrtf(std::nextafterf(s, FLT_MAX)) <= r; k++) s = std::nextafterf(s, FLT_MAX);
    return s;
}

}  // namespace

}  // namespace pcr

using namespace pcr;

extern "C" int pcr_iss_keypoints_f32(pcr_ctx* ctx, const pcr_cloud* cloud, const pcr_iss_params* prm, uint8_t* is_key, float* lambda3,
                                     uint32_t* neighbor_counts, uint64_t* n_keypoints)
{
    if (!ctx || !cloud || !prm || !is_key) return fail(ctx, PCR_ERR_ARG, "pcr_iss_keypoints_f32");
    if (!(prm->local_radius >= 0.0f) || !(prm->non_max_radius >= 0.0f) || std::isinf(prm->local_radius) || std::isinf(prm->non_max_radius))
        return fail(ctx, PCR_ERR_ARG, "pcr_iss_keypoints_f32: radii must be finite and >= 0");
    PCR_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = cloud->n;
    if (n_keypoints) *n_keypoints = 0;
    if (n == 0) return PCR_OK;
    if (n > 0x7FFFFFF0ull) return fail(ctx, PCR_ERR_ARG, "pcr_iss_keypoints_f32: cloud too large");
    // cell edge >= the larger radius with a margin for the rounding of the cell coordinate, so that every neighbour lies
    // in the 27-cell block; grid_build may only enlarge it (cell budget, <= 4001 cells per axis)
    const double rmax = std::max((double)prm->local_radius, (double)prm->non_max_radius);
    Grid* g = nullptr;
    {
        ProfScope ps(ctx, "iss_grid_build");
        // >= 2e-15: below that, squared f32 distances underflow and a point outside the 27-cell block could still compute
        // s == 0 <= s_max; with h >= 2e-15 every outside point has s >= ~4e-30 (normal range) > s_max
        int rc = grid_build(ctx, cloud, &g, std::max(rmax * 1.01, 2e-15));
        if (rc) return rc;
    }
    if ((double)g->p.h < rmax * 1.005) { grid_free(g); return fail(ctx, PCR_ERR_STATE, "pcr_iss_keypoints_f32: grid cell smaller than the radius"); }
    IssParams ip;
    ip.s_local = sqrt_threshold(prm->local_radius);
    ip.s_nms = sqrt_threshold(prm->non_max_radius);
    ip.gamma21 = prm->gamma21;
    ip.gamma32 = prm->gamma32;
    ip.min_neighbors = (unsigned)prm->min_neighbors;      // size_t < int comparison of the reference (:92): negative -> huge
    ip.weighted = prm->weighted_covariance ? 1 : 0;
    const size_t a4 = (n * 4 + 255) & ~(size_t)255, a1 = (n + 255) & ~(size_t)255;
    const size_t a8 = (n * 8 + 255) & ~(size_t)255;
    int rc = ensure_scratch(ctx, 7 * a8 + 4 * a4 + a1 + 256);
    if (rc) { grid_free(g); return rc; }
    char* s = (char*)ctx->scratch;
    double* sums = (double*)s;                              // 7 arrays of n (only the first n of each a8 slot pitch is used)
    s += 7 * a8;
    uint32_t* cnt_sorted = (uint32_t*)s;
    float* l3_sorted = (float*)(s + a4);
    float* l3_out = (float*)(s + 2 * a4);
    uint32_t* cnt_out = (uint32_t*)(s + 3 * a4);
    uint8_t* key_dev = (uint8_t*)(s + 4 * a4);
    const int G = (int)tune_get(ctx, "iss_lanes", 32);   // measured: profiles/r01_iss.txt (neighbourhoods of 10^2-10^3 points)
#define PCR_ISS(GG)                                                                                                                   \
    {                                                                                                                                 \
        const dim3 grid((unsigned)((n * GG + ISS_BLOCK - 1) / ISS_BLOCK));                                                            \
        { ProfScope ps(ctx, "iss_count", 1);                                                                                          \
          hipLaunchKernelGGL((iss_count_kernel<GG>), grid, dim3(ISS_BLOCK), 0, ctx->stream, g->records, g->cell_start, g->p, (uint32_t)n, ip.s_local, cnt_sorted, cnt_out); } \
        { ProfScope ps(ctx, "iss_cov", 1);                                                                                            \
          hipLaunchKernelGGL((iss_cov_kernel<GG>), grid, dim3(ISS_BLOCK), 0, ctx->stream, g->records, g->cell_start, g->p, (uint32_t)n, ip, cnt_sorted, sums); } \
        { ProfScope ps(ctx, "iss_eig", 1);                                                                                            \
          hipLaunchKernelGGL(iss_eig_kernel, dim3((unsigned)((n + ISS_BLOCK - 1) / ISS_BLOCK)), dim3(ISS_BLOCK), 0, ctx->stream, g->records, (uint32_t)n, ip, cnt_sorted, sums, l3_sorted, l3_out); } \
        { ProfScope ps(ctx, "iss_nms", 1);                                                                                            \
          hipLaunchKernelGGL((iss_nms_kernel<GG>), grid, dim3(ISS_BLOCK), 0, ctx->stream, g->records, g->cell_start, g->p, (uint32_t)n, ip, l3_sorted, key_dev); } \
    }
    switch (G) {
    case 1: PCR_ISS(1) break;
    case 2: PCR_ISS(2) break;
    case 4: PCR_ISS(4) break;
    case 8: PCR_ISS(8) break;
    case 16: PCR_ISS(16) break;
    default: PCR_ISS(32) break;
    }
#undef PCR_ISS
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(is_key, key_dev, n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && lambda3) e = hipMemcpyAsync(lambda3, l3_out, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && neighbor_counts) e = hipMemcpyAsync(neighbor_counts, cnt_out, n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    grid_free(g);
    if (e != hipSuccess) return fail(ctx, PCR_ERR_HIP, "pcr_iss_keypoints_f32", e);
    prof_flush(ctx);
    if (n_keypoints) {
        uint64_t c = 0;
        for (size_t i = 0; i < n; i++) c += is_key[i];
        *n_keypoints = c;
    }
    return PCR_OK;
}

// ground.hip — next row N2: the PCA ground fit either side of the plane-inlier count
// (Homework4/ground_detection_SVD.py:46-101; FastEigen3x3 = Homework1/.../my_pybind11/src/mylib.cpp:9-189).
//
//   seeds      z < -1.73 + 0.5, LPR = mean z of the lpr_size lowest candidates (radix sort of order-preserving keys),
//              seed iff candidate && z < LPR_z + threshold                                               (:46-71)
//   iteration  centre = mean(seeds); XTX = sum (p - c)(p - c)^T; normal = FastEigen3x3(XTX); d = -normal . c   (:74-85)
//              seeds <- { p : |[p 1] . params| < threshold_dist }                                        (:94-98)
//
// The point predicate (seed test or plane test) is re-evaluated inside both moment kernels instead of materialising a
// mask: one iteration is two streaming passes of 12 B/point (HBM-bound; 120 k points = 1.4 MB, L2-resident) with
// block-level f64 partials that the host adds in block order — deterministic, f64 like the reference's numpy.  The 3x3
// eigenvector (closed form: trigonometric eigenvalues + cross products) runs on the host.
#include "pcr_internal.hpp"
#include "eig3.hpp"

#include "sort.hpp"

#include <algorithm>
#include <cmath>
#include <vector>

namespace pcr {

namespace {

constexpr int GD_BLOCK = 256;
constexpr int GD_MAX_BLOCKS = 1024;
constexpr double GD_Z_HIGH = -1.73 + 0.5;   // ground_detection_SVD.py:47

struct Pred {
    int mode;          // 0: seed test, 1: plane test, 2: every finite point
    double ub;         // mode 0: LPR_z + threshold_seeds
    double p[4];       // mode 1: plane
    double thr;
};

__device__ __forceinline__ bool gd_test(const Pred& pr, float xf, float yf, float zf)
{
#pragma clang fp contract(off)
    const double x = xf, y = yf, z = zf;
    if (pr.mode == 2) return fabs(x) <= 1.7976931348623157e308 && fabs(y) <= 1.7976931348623157e308 && fabs(z) <= 1.7976931348623157e308;   // every finite point
    if (pr.mode == 0) return z < GD_Z_HIGH && z < pr.ub;
    return fabs(((x * pr.p[0] + y * pr.p[1]) + z * pr.p[2]) + 1.0 * pr.p[3]) < pr.thr;   // np.c_[p, 1].dot(params), k = 0..3
}

template <int K>
__device__ __forceinline__ void block_sum(double (&v)[K], double* __restrict__ out)
{
    __shared__ double sh[K][GD_BLOCK / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; k++) {
        double a = v[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
        if (lane == 0) sh[k][wave] = a;
    }
    __syncthreads();
    if (threadIdx.x < K) {
        double a = 0.0;
#pragma unroll
        for (int w = 0; w < GD_BLOCK / 64; w++) a += sh[threadIdx.x][w];
        out[(size_t)blockIdx.x * K + threadIdx.x] = a;
    }
}

// partials[b] = { count, sum x, sum y, sum z } of the points passing the predicate
__global__ __launch_bounds__(GD_BLOCK) void gd_sum1_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, uint32_t n,
                                                           Pred pr, double* __restrict__ partials)
{
    double v[4] = { 0, 0, 0, 0 };
    for (uint32_t i = blockIdx.x * GD_BLOCK + threadIdx.x; i < n; i += gridDim.x * GD_BLOCK) {
        const float a = x[i], b = y[i], c = z[i];
        if (gd_test(pr, a, b, c)) { v[0] += 1.0; v[1] += (double)a; v[2] += (double)b; v[3] += (double)c; }
    }
    block_sum<4>(v, partials);
}

// partials[b] = { xx, xy, xz, yy, yz, zz } of the centred points passing the predicate
__global__ __launch_bounds__(GD_BLOCK) void gd_sum2_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, uint32_t n,
                                                           Pred pr, double cx, double cy, double cz, double* __restrict__ partials)
{
#pragma clang fp contract(off)
    double v[6] = { 0, 0, 0, 0, 0, 0 };
    for (uint32_t i = blockIdx.x * GD_BLOCK + threadIdx.x; i < n; i += gridDim.x * GD_BLOCK) {
        const float a = x[i], b = y[i], c = z[i];
        if (!gd_test(pr, a, b, c)) continue;
        const double dx = (double)a - cx, dy = (double)b - cy, dz = (double)c - cz;
        v[0] += dx * dx; v[1] += dx * dy; v[2] += dx * dz; v[3] += dy * dy; v[4] += dy * dz; v[5] += dz * dz;
    }
    block_sum<6>(v, partials);
}

__global__ __launch_bounds__(GD_BLOCK) void gd_mask_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, uint32_t n, Pred pr,
                                                           uint8_t* __restrict__ mask)
{
    const uint32_t i = blockIdx.x * GD_BLOCK + threadIdx.x;
    if (i < n) mask[i] = gd_test(pr, x[i], y[i], z[i]) ? 1 : 0;
}

// order-preserving u32 key of z for the candidates (z < z_high), 0xFFFFFFFF for the rest; counts the candidates
__global__ __launch_bounds__(GD_BLOCK) void gd_keys_kernel(const float* __restrict__ z, uint32_t n, uint32_t* __restrict__ keys, uint32_t* __restrict__ n_cand)
{
    const uint32_t i = blockIdx.x * GD_BLOCK + threadIdx.x;
    bool cand = false;
    if (i < n) {
        const float zf = z[i];
        cand = (double)zf < GD_Z_HIGH;                      // false for NaN
        uint32_t u = __float_as_uint(zf);
        u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;
        keys[i] = cand ? u : 0xFFFFFFFFu;
    }
    const unsigned long long b = __ballot(cand);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(n_cand, (uint32_t)__popcll(b));
}

float key_to_float(uint32_t u)
{
    u ^= (u >> 31) ? 0x80000000u : 0xFFFFFFFFu;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

unsigned gd_blocks(size_t n) { return (unsigned)std::max<size_t>(1, std::min<size_t>(GD_MAX_BLOCKS, (n + GD_BLOCK - 1) / GD_BLOCK)); }

// LPR_z + threshold (NaN when there is no candidate)
int seed_upper_bound(pcr_ctx* ctx, const pcr_cloud* c, size_t lpr_size, double threshold_seeds, double* ub)
{
    const size_t n = c->n;
    size_t temp_bytes = 0;
    sort_keys_u32(nullptr, temp_bytes, nullptr, nullptr, n, 0, 32, ctx->stream);
    const size_t kb = (n * 4 + 255) & ~(size_t)255;
    int rc = ensure_scratch(ctx, 2 * kb + 256 + temp_bytes + 256);
    if (rc) return rc;
    char* s = (char*)ctx->scratch;
    uint32_t* k_in = (uint32_t*)s;
    uint32_t* k_out = (uint32_t*)(s + kb);
    uint32_t* n_cand_dev = (uint32_t*)(s + 2 * kb);
    void* temp = s + 2 * kb + 256;
    PCR_HIP(ctx, hipMemsetAsync(n_cand_dev, 0, 4, ctx->stream));
    {
        ProfScope ps(ctx, "ground_seed_select", 1);
        hipLaunchKernelGGL(gd_keys_kernel, dim3((unsigned)((n + GD_BLOCK - 1) / GD_BLOCK)), dim3(GD_BLOCK), 0, ctx->stream, c->z(), (uint32_t)n, k_in, n_cand_dev);
        PCR_HIP(ctx, sort_keys_u32(temp, temp_bytes, k_in, k_out, n, 0, 32, ctx->stream));
    }
    uint32_t m = 0;
    PCR_HIP(ctx, hipMemcpyAsync(&m, n_cand_dev, 4, hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const size_t k = std::min<size_t>(lpr_size, m);                                  // :54-60
    std::vector<uint32_t> low(k);
    if (k) {
        PCR_HIP(ctx, hipMemcpyAsync(low.data(), k_out, k * 4, hipMemcpyDeviceToHost, ctx->stream));
        PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    double sum = 0.0;
    for (size_t i = 0; i < k; i++) sum += (double)key_to_float(low[i]);             // np.mean(axis=0), z column (:62), ascending z
    *ub = sum / (double)k + threshold_seeds;                                         // :65; 0/0 = NaN without candidates
    return PCR_OK;
}

// centre and scatter matrix (row-major, symmetric) of the points passing `pr`; *count = number of such points
int moments(pcr_ctx* ctx, const pcr_cloud* c, const Pred& pr, double centre[3], double XTX[9], uint64_t* count);

// estimate_plane over the points passing `pr` (:74-85); *count = number of such points
int fit_plane(pcr_ctx* ctx, const pcr_cloud* c, const Pred& pr, double params[4], uint64_t* count)
{
    double ctr[3], XTX[9];
    int rc = moments(ctx, c, pr, ctr, XTX, count);
    if (rc || *count == 0) return rc;
    double nrm[3];
    pcr_fast_eigen3x3(XTX, nrm);                                                     // :83
    params[0] = nrm[0]; params[1] = nrm[1]; params[2] = nrm[2];
    params[3] = -(nrm[0] * ctr[0] + nrm[1] * ctr[1] + nrm[2] * ctr[2]);               // :84
    return PCR_OK;
}

int moments(pcr_ctx* ctx, const pcr_cloud* c, const Pred& pr, double centre[3], double XTX_out[9], uint64_t* count)
{
    const size_t n = c->n;
    const unsigned blocks = gd_blocks(n);
    int rc = ensure_scratch(ctx, (size_t)blocks * 10 * sizeof(double));
    if (rc) return rc;
    double* p1 = (double*)ctx->scratch;
    double* p2 = p1 + (size_t)blocks * 4;
    std::vector<double> h((size_t)blocks * 6);
    {
        ProfScope ps(ctx, "ground_moments", 1);
        hipLaunchKernelGGL(gd_sum1_kernel, dim3(blocks), dim3(GD_BLOCK), 0, ctx->stream, c->x(), c->y(), c->z(), (uint32_t)n, pr, p1);
    }
    PCR_HIP(ctx, hipMemcpyAsync(h.data(), p1, (size_t)blocks * 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double s[4] = { 0, 0, 0, 0 };
    for (unsigned b = 0; b < blocks; b++)
        for (int k = 0; k < 4; k++) s[k] += h[(size_t)b * 4 + k];
    *count = (uint64_t)s[0];
    if (s[0] == 0.0) return PCR_OK;
    const double cx = s[1] / s[0], cy = s[2] / s[0], cz = s[3] / s[0];               // :75
    {
        ProfScope ps(ctx, "ground_moments", 1);
        hipLaunchKernelGGL(gd_sum2_kernel, dim3(blocks), dim3(GD_BLOCK), 0, ctx->stream, c->x(), c->y(), c->z(), (uint32_t)n, pr, cx, cy, cz, p2);
    }
    PCR_HIP(ctx, hipMemcpyAsync(h.data(), p2, (size_t)blocks * 6 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double m[6] = { 0, 0, 0, 0, 0, 0 };
    for (unsigned b = 0; b < blocks; b++)
        for (int k = 0; k < 6; k++) m[k] += h[(size_t)b * 6 + k];
    const double XTX[9] = { m[0], m[1], m[2], m[1], m[3], m[4], m[2], m[4], m[5] };  // :77
    for (int k = 0; k < 9; k++) XTX_out[k] = XTX[k];
    centre[0] = cx; centre[1] = cy; centre[2] = cz;
    return PCR_OK;
}

int fetch_mask(pcr_ctx* ctx, const pcr_cloud* c, const Pred& pr, uint8_t* host_mask)
{
    const size_t n = c->n;
    int rc = ensure_scratch(ctx, n + 256);
    if (rc) return rc;
    uint8_t* dm = (uint8_t*)ctx->scratch;
    hipLaunchKernelGGL(gd_mask_kernel, dim3((unsigned)((n + GD_BLOCK - 1) / GD_BLOCK)), dim3(GD_BLOCK), 0, ctx->stream, c->x(), c->y(), c->z(), (uint32_t)n, pr, dm);
    PCR_HIP(ctx, hipGetLastError());
    PCR_HIP(ctx, hipMemcpyAsync(host_mask, dm, n, hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PCR_OK;
}

}  // namespace

}  // namespace pcr

using namespace pcr;

extern "C" {

// Host logic (no GPU): mylib.FastEigen3x3 (mylib.cpp:105-189); the implementation is shared with the device (eig3.hpp).
int pcr_fast_eigen3x3(const double A[9], double normal[3])
{
    if (!A || !normal) return PCR_ERR_ARG;
    eig3::smallest_eigenvector(A, normal);
    return PCR_OK;
}

// pca_normal.py:17-36 PCA(data, sort = True): centre = sum / n, XTX = centred^T centred (streamed on the GPU, f64), then the
// eigen-decomposition on the host (one-sided Jacobi; np.linalg.eig there — eigenvector signs are unspecified in both).
int pcr_cloud_pca_f64(pcr_ctx* ctx, const pcr_cloud* cloud, double eigenvalues[3], double eigenvectors[9], double centre[3])
{
    if (!ctx || !cloud || !eigenvalues || !eigenvectors) return fail(ctx, PCR_ERR_ARG, "pcr_cloud_pca_f64");
    if (cloud->n > 0x7FFFFFF0ull) return fail(ctx, PCR_ERR_ARG, "pcr_cloud_pca_f64: cloud too large");
    PCR_HIP(ctx, hipSetDevice(ctx->device));
    Pred pr{};
    pr.mode = 2;
    double ctr[3] = { 0, 0, 0 }, XTX[9];
    uint64_t count = 0;
    if (cloud->n) {
        int rc = moments(ctx, cloud, pr, ctr, XTX, &count);
        if (rc) return rc;
    }
    if (count == 0) return fail(ctx, PCR_ERR_EMPTY, "pcr_cloud_pca_f64: no finite point");
    double U[9], S[3], V[9];
    svd3(XTX, U, S, V);                                      // symmetric PSD: singular values = eigenvalues, descending (:31-34)
    for (int k = 0; k < 3; k++) eigenvalues[k] = S[k];
    for (int k = 0; k < 9; k++) eigenvectors[k] = V[k];      // row-major, eigenvectors in the COLUMNS like numpy's
    if (centre) for (int k = 0; k < 3; k++) centre[k] = ctr[k];
    prof_flush(ctx);
    return PCR_OK;
}

int pcr_ground_seeds_f64(pcr_ctx* ctx, const pcr_cloud* cloud, size_t lpr_size, double threshold_seeds, uint8_t* seed_mask, double* upper_bound,
                         uint64_t* n_seeds)
{
    if (!ctx || !cloud || (cloud->n && !seed_mask)) return fail(ctx, PCR_ERR_ARG, "pcr_ground_seeds_f64");
    if (cloud->n > 0x7FFFFFF0ull) return fail(ctx, PCR_ERR_ARG, "pcr_ground_seeds_f64: cloud too large");
    PCR_HIP(ctx, hipSetDevice(ctx->device));
    if (n_seeds) *n_seeds = 0;
    if (upper_bound) *upper_bound = NAN;
    if (cloud->n == 0) return PCR_OK;
    Pred pr{};
    pr.mode = 0;
    int rc = seed_upper_bound(ctx, cloud, lpr_size, threshold_seeds, &pr.ub);
    if (rc) return rc;
    rc = fetch_mask(ctx, cloud, pr, seed_mask);
    if (rc) return rc;
    if (upper_bound) *upper_bound = pr.ub;
    if (n_seeds) {
        uint64_t c = 0;
        for (size_t i = 0; i < cloud->n; i++) c += seed_mask[i];
        *n_seeds = c;
    }
    prof_flush(ctx);
    return PCR_OK;
}

int pcr_ground_detection_f64(pcr_ctx* ctx, const pcr_cloud* cloud, int max_iter, size_t lpr_size, double threshold_dist, double params[4],
                             uint8_t* ground_mask, uint64_t* n_ground)
{
    if (!ctx || !cloud || !params || max_iter < 1 || (cloud->n && !ground_mask)) return fail(ctx, PCR_ERR_ARG, "pcr_ground_detection_f64");
    if (cloud->n > 0x7FFFFFF0ull) return fail(ctx, PCR_ERR_ARG, "pcr_ground_detection_f64: cloud too large");
    PCR_HIP(ctx, hipSetDevice(ctx->device));
    if (n_ground) *n_ground = 0;
    if (cloud->n == 0) return fail(ctx, PCR_ERR_EMPTY, "pcr_ground_detection_f64: empty cloud");
    Pred pr{};
    pr.mode = 0;
    pr.thr = threshold_dist;
    int rc = seed_upper_bound(ctx, cloud, lpr_size, threshold_dist, &pr.ub);       // :90 — threshold_seeds = threshold_dist
    if (rc) return rc;
    uint64_t count = 0;
    for (int it = 0; it < max_iter; it++) {                                          // :93-98
        rc = fit_plane(ctx, cloud, pr, params, &count);
        if (rc) return rc;
        if (count == 0) return fail(ctx, PCR_ERR_EMPTY, "pcr_ground_detection_f64: a fit had no point (the reference would propagate NaN)");
        pr.mode = 1;
        for (int k = 0; k < 4; k++) pr.p[k] = params[k];
    }
    rc = fetch_mask(ctx, cloud, pr, ground_mask);
    if (rc) return rc;
    uint64_t c = 0;
    for (size_t i = 0; i < cloud->n; i++) c += ground_mask[i];
    if (n_ground) *n_ground = c;
    prof_flush(ctx);
    return PCR_OK;
}

}  // extern "C"

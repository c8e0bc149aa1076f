// p2plane.hip — Registration::ICPpoint2plane (Homework9/hw9/src/registration.cpp:710-860), the point-to-plane sibling of the
// point-to-point loop on the same correspondence search.
//
// Per iteration: 1-NN (the kernels of A6) -> one streaming pass over the kept pairs that builds the row
//   A = [n x p, n],  b = n.q - n.p     (f32, exactly as written at :807-814; n = target normal at the matched index)
// and accumulates the normal equations in f64: 21 + 6 + 1 + 1 values (upper A^T A, A^T b, b^T b, pair count) as
// block partials that the host adds in block order (deterministic) -> 6x6 solve on the host (f64 Gaussian elimination
// with partial pivoting; the reference: f32 Eigen `(A^T A).inverse() * A^T * b`, unpinned) -> linearised update
// R_delta = I + [x]_x (not re-orthonormalised, :843), t_delta -> pose composition and transform (A8 kernels).
// HBM-bound: 12 B source + 8 B key + 24 B gathered target point and normal per kept pair.
#include "pcr_internal.hpp"

#include <chrono>
#include <cmath>
#include <vector>

#pragma clang fp contract(off)

namespace pcr {

namespace {

constexpr int PP_BLOCK = 256;
constexpr int PP_NV = 29;            // 21 upper A^T A, 6 A^T b, b^T b, count
constexpr int PP_MAX_BLOCKS = 512;

__global__ __launch_bounds__(PP_BLOCK) void p2plane_partial_kernel(const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz,
                                                                   const float* __restrict__ tx, const float* __restrict__ ty, const float* __restrict__ tz,
                                                                   const float* __restrict__ nx, const float* __restrict__ ny, const float* __restrict__ nz,
                                                                   const unsigned long long* __restrict__ keys, uint32_t ns, uint32_t nt, float max_corr,
                                                                   double* __restrict__ partials)
{
    double acc[PP_NV];
#pragma unroll
    for (int k = 0; k < PP_NV; k++) acc[k] = 0.0;
    for (uint32_t i = blockIdx.x * PP_BLOCK + threadIdx.x; i < ns; i += gridDim.x * PP_BLOCK) {
        const unsigned long long key = keys[i];
        const uint32_t j = (uint32_t)(key & 0xFFFFFFFFull);
        const float d2 = __uint_as_float((uint32_t)(key >> 32));
        if (!(d2 < max_corr) || j >= nt) continue;                                       // :778
        const float p0 = sx[i], p1 = sy[i], p2 = sz[i];
        const float q0 = tx[j], q1 = ty[j], q2 = tz[j];
        const float n0 = nx[j], n1 = ny[j], n2 = nz[j];
        float A[6];
        A[0] = n2 * p1 - n1 * p2;                                                        // :807-812
        A[1] = n0 * p2 - n2 * p0;
        A[2] = n1 * p0 - n0 * p1;
        A[3] = n0; A[4] = n1; A[5] = n2;
        const float b = n0 * q0 + n1 * q1 + n2 * q2 - n0 * p0 - n1 * p1 - n2 * p2;       // :814
        int k = 0;
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int c = r; c < 6; c++) acc[k++] += (double)A[r] * (double)A[c];
#pragma unroll
        for (int r = 0; r < 6; r++) acc[21 + r] += (double)A[r] * (double)b;
        acc[27] += (double)b * (double)b;
        acc[28] += 1.0;
    }
    __shared__ double sh[PP_NV][PP_BLOCK / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < PP_NV; k++) {
        double a = acc[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
        if (lane == 0) sh[k][wave] = a;
    }
    __syncthreads();
    if (threadIdx.x < PP_NV) {
        double a = 0.0;
#pragma unroll
        for (int w = 0; w < PP_BLOCK / 64; w++) a += sh[threadIdx.x][w];
        partials[(size_t)blockIdx.x * PP_NV + threadIdx.x] = a;
    }
}

// symmetric 6x6 system: 0, or -1 when singular / not finite
int solve6(const double M[36], const double v[6], double x[6])
{
    double a[6][7];
    for (int r = 0; r < 6; r++) { for (int c = 0; c < 6; c++) a[r][c] = M[6 * r + c]; a[r][6] = v[r]; }
    for (int col = 0; col < 6; col++) {
        int piv = col;
        for (int r = col + 1; r < 6; r++) if (std::fabs(a[r][col]) > std::fabs(a[piv][col])) piv = r;
        if (!(std::fabs(a[piv][col]) > 1e-300)) return -1;
        if (piv != col) for (int c = 0; c < 7; c++) std::swap(a[col][c], a[piv][c]);
        for (int r = col + 1; r < 6; r++) {
            const double f = a[r][col] / a[col][col];
            for (int c = col; c < 7; c++) a[r][c] -= f * a[col][c];
        }
    }
    for (int r = 5; r >= 0; r--) {
        double s = a[r][6];
        for (int c = r + 1; c < 6; c++) s -= a[r][c] * x[c];
        x[r] = s / a[r][r];
    }
    for (int r = 0; r < 6; r++) if (!(std::fabs(x[r]) <= 1.7976931348623157e308)) return -1;
    return 0;
}

}  // namespace

}  // namespace pcr

using namespace pcr;

extern "C" int pcr_icp_p2plane_f32(pcr_ctx* ctx, const pcr_cloud* src, const pcr_cloud* tgt, const pcr_cloud* tgt_normals, const float init_T[16],
                                   const pcr_icp_params* prm, float out_T[16], pcr_icp_stats* stats)
{
    if (!ctx || !src || !tgt || !tgt_normals || !init_T || !prm || !out_T) return fail(ctx, PCR_ERR_ARG, "pcr_icp_p2plane_f32");
    if (tgt_normals->n != tgt->n) return fail(ctx, PCR_ERR_ARG, "pcr_icp_p2plane_f32: one normal per target point expected");
    PCR_HIP(ctx, hipSetDevice(ctx->device));
    const auto t_begin = std::chrono::steady_clock::now();
    pcr_icp_stats st;
    memset(&st, 0, sizeof st);
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    prof_flush(ctx);
    const uint64_t nn_l0 = ctx->prof["nn1_brute"].launches + ctx->prof["nn1_grid"].launches;
    const double nn_ms0 = ctx->prof["nn1_brute"].total_ms + ctx->prof["nn1_grid"].total_ms;

    pcr_cloud* work = nullptr;
    int rc = pcr_cloud_clone(ctx, src, &work);                                           // :720
    if (rc) return rc;
    const float R0[9] = { init_T[0], init_T[1], init_T[2], init_T[4], init_T[5], init_T[6], init_T[8], init_T[9], init_T[10] };
    const float t0[3] = { init_T[3], init_T[7], init_T[11] };
    rc = launch_transform(ctx, work, R0, t0);                                            // :722
    // grid searches: the working copy in the order of the target's index, once (as the point-to-point loop does) — nothing here is
    // per source point but the point itself, the normal comes with the correspondence.  The 29 sums are f64 block partials added in
    // block order, so the pose depends on this order in its last bits (as it does on the number of ranks): deterministic, within the
    // 1e-5 of the tests.  Unsorted, the Morton-ordered index of round 3 cost this loop 0.143 against 0.122 ms per iteration at 120 k.
    if (rc == PCR_OK && tune_get(ctx, "p2plane_sort_work", 1) == 1) {
        const LoopHint sort_hint(ctx, prm->max_iter);
        if (nn1_auto_grid(ctx, tgt, true, work->n)) rc = grid_sort_working_cloud(ctx, tgt, &work);
    }
    float T_total[16] = { R0[0], R0[1], R0[2], t0[0], R0[3], R0[4], R0[5], t0[1], R0[6], R0[7], R0[8], t0[2], 0, 0, 0, 1 };   // :759-760
    float last_loss = 0.0f;
    uint64_t unchanged = 0;
    const size_t ns = work->n;
    const unsigned blocks = (unsigned)std::max<size_t>(1, std::min<size_t>(PP_MAX_BLOCKS, (ns + PP_BLOCK - 1) / PP_BLOCK));
    double* partials_dev = nullptr;
    if (rc == PCR_OK) {
        hipError_t e = hipMalloc((void**)&partials_dev, (size_t)blocks * PP_NV * sizeof(double));
        if (e != hipSuccess) rc = fail(ctx, PCR_ERR_HIP, "hipMalloc(p2plane)", e);
    }
    std::vector<double> hp((size_t)blocks * PP_NV);
    const LoopHint hint(ctx, prm->max_iter);       // (api.cpp nn1_auto_grid: small targets take the grid inside a loop)
    for (uint64_t iter = 0; rc == PCR_OK && iter < prm->max_iter; iter++) {
        if ((rc = launch_nn1(ctx, tgt, work, true, tune_get(ctx, "icp_bounded_search", 1) == 1 ? prm->max_corr : __builtin_inff()))) break;                              // :768-781
        double s[64];
        for (int k = 0; k < 64; k++) s[k] = 0.0;
        if (ns) {
            {
                ProfScope ps(ctx, "p2plane_partial");
                hipLaunchKernelGGL(p2plane_partial_kernel, dim3(blocks), dim3(PP_BLOCK), 0, ctx->stream, work->x(), work->y(), work->z(), tgt->x(), tgt->y(),
                                   tgt->z(), tgt_normals->x(), tgt_normals->y(), tgt_normals->z(), ctx->keys, (uint32_t)ns, (uint32_t)tgt->n, prm->max_corr,
                                   partials_dev);
            }
            hipError_t e = hipGetLastError();
            if (e == hipSuccess) e = hipMemcpyAsync(hp.data(), partials_dev, hp.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) { rc = fail(ctx, PCR_ERR_HIP, "p2plane partials", e); break; }
            for (unsigned b = 0; b < blocks; b++)
                for (int k = 0; k < PP_NV; k++) s[k] += hp[(size_t)b * PP_NV + k];
        }
        if (ctx->comm.nranks > 1 && (rc = comm_allreduce_f64(ctx, s, ctx->dev_out, PP_NV))) break;   // sources sharded: ONE all-reduce of 29 f64
        double M[36], v[6];
        int k = 0;
        for (int r = 0; r < 6; r++)
            for (int c = r; c < 6; c++) { M[6 * r + c] = s[k]; M[6 * c + r] = s[k]; k++; }
        for (int r = 0; r < 6; r++) v[r] = s[21 + r];
        const double btb = s[27];
        st.last_pairs = (uint64_t)s[28];
        double x64[6];
        if (s[28] == 0.0 || solve6(M, v, x64) != 0) { st.empty_pairs = 1; break; }      // :818 (the reference would produce NaN)
        float x[6];
        for (int i = 0; i < 6; i++) x[i] = (float)x64[i];
        double xMx = 0.0, xv = 0.0;
        for (int r = 0; r < 6; r++) { for (int c = 0; c < 6; c++) xMx += (double)x[r] * M[6 * r + c] * (double)x[c]; xv += (double)x[r] * v[r]; }
        const float loss = (float)(xMx - 2.0 * xv + btb);                                // :820 |A x - b|^2
        st.last_loss = loss;
        if (std::fabs(last_loss - loss) < prm->eps) unchanged++;                         // :828-831 (never reset)
        if (unchanged > 15) { st.converged = 1; break; }                                 // :834-838
        last_loss = loss;
        const float Rd[9] = { 1, -x[2], x[1], x[2], 1, -x[0], -x[1], x[0], 1 };          // :843
        const float td[3] = { x[3], x[4], x[5] };
        const float T_delta[16] = { Rd[0], Rd[1], Rd[2], td[0], Rd[3], Rd[4], Rd[5], td[1], Rd[6], Rd[7], Rd[8], td[2], 0, 0, 0, 1 };
        mat4_mul_f32(T_delta, T_total, T_total);                                         // :849
        rc = launch_transform(ctx, work, Rd, td);                                        // :851
        st.iters_run++;
    }
    hipStreamSynchronize(ctx->stream);
    if (partials_dev) hipFree(partials_dev);
    cloud_release(ctx, work);             // (synchronised above; the loop's own working copy)
    if (rc) return rc;
    memcpy(out_T, T_total, sizeof T_total);
    prof_flush(ctx);
    st.nn_launches = ctx->prof["nn1_brute"].launches + ctx->prof["nn1_grid"].launches - nn_l0;
    st.ms_nn = ctx->prof["nn1_brute"].total_ms + ctx->prof["nn1_grid"].total_ms - nn_ms0;
    st.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    if (stats) *stats = st;
    return PCR_OK;
}

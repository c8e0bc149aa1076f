// desc.hip — next row N4: the global-registration front half of Homework9/hw9 (src/registration.cpp:288-434, :535-615).
//
//   N4a  nn1_dim_kernel        exhaustive 1-NN between descriptor sets (33-D FPFH there), nanoflann's evalMetric
//                              arithmetic for any dim (nanoflann.hpp:382-405), canonical tie rule, u64 atomicMin merge
//   N4b  pcr_match_union_f32   both search directions + sort by distance + reject the worst share (:535-615)
//   N4c  ransac_hyp_kernel     one lane per hypothesis: 4-point Kabsch (f64 moments, numerics.hpp solve) (:354-392)
//        consensus_kernel      one lane per hypothesis, the correspondence list streamed through the scalar cache:
//                              || tgt - (R src + t) || <= thr counted for every hypothesis (:395-421)
//
// The hot part is N4c: 80 000 hypotheses (main.cpp:86) x a few thousand correspondences, ~27 f32 lane-ops per
// (hypothesis, correspondence) — VALU-bound, the 6 floats of a correspondence are wave-uniform (scalar loads).
#include "pcr_internal.hpp"
#include "numerics.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <random>
#include <vector>

namespace pcr {

namespace {

constexpr int DS_BLOCK = 64;    // queries per workgroup (one wavefront)
constexpr int DS_ROWS = 64;     // database rows per LDS tile
constexpr int DS_MAX_DIM = 256;

// evalMetric (nanoflann.hpp:382-405): groups of four, then the tail; unfused
template <int DIM>
__device__ __forceinline__ float d2_rows(const float (&q)[DIM], const float* __restrict__ row)
{
#pragma clang fp contract(off)
    float result = 0.0f;
    int d = 0;
#pragma unroll
    for (; d + 3 < DIM; d += 4) {
        const float d0 = q[d] - row[d], d1 = q[d + 1] - row[d + 1], d2 = q[d + 2] - row[d + 2], d3 = q[d + 3] - row[d + 3];
        result += ((d0 * d0 + d1 * d1) + d2 * d2) + d3 * d3;
    }
#pragma unroll
    for (; d < DIM; d++) {
        const float d0 = q[d] - row[d];
        result += d0 * d0;
    }
    return result;
}

// DIM > 0: the query lives in registers.  grid = (query blocks, database slices); keys merge with atomicMin
template <int DIM>
__global__ __launch_bounds__(DS_BLOCK) void nn1_dim_kernel(const float* __restrict__ db, uint32_t n, const float* __restrict__ qs, uint32_t m,
                                                           uint32_t rows_per_slice, unsigned long long* __restrict__ keys)
{
    __shared__ float tile[DS_ROWS * DIM];
    const uint32_t qi = blockIdx.x * DS_BLOCK + threadIdx.x;
    float q[DIM];
    const uint32_t qsafe = qi < m ? qi : m - 1;
#pragma unroll
    for (int d = 0; d < DIM; d++) q[d] = qs[(size_t)qsafe * DIM + d];
    const uint32_t r0 = blockIdx.y * rows_per_slice, r1 = min(n, r0 + rows_per_slice);
    unsigned long long best = ~0ull;
    for (uint32_t base = r0; base < r1; base += DS_ROWS) {
        const uint32_t rows = min((uint32_t)DS_ROWS, r1 - base);
        __syncthreads();
        for (uint32_t e = threadIdx.x; e < rows * DIM; e += DS_BLOCK) tile[e] = db[(size_t)base * DIM + e];
        __syncthreads();
        for (uint32_t r = 0; r < rows; r++) {
            const float d2 = d2_rows<DIM>(q, tile + r * DIM);
            if (d2 < FLT_MAX) {                                                   // nanoflann.hpp:163,1360; false for NaN
                const unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) | (base + r);
                best = key < best ? key : best;
            }
        }
    }
    if (qi < m && best != ~0ull) atomicMin(&keys[qi], best);
}

// any dim <= DS_MAX_DIM: the queries of the workgroup sit in LDS (dim-major, conflict-free), rows are read from global
// memory (wave-uniform address -> scalar/broadcast loads)
__global__ __launch_bounds__(DS_BLOCK) void nn1_anydim_kernel(const float* __restrict__ db, uint32_t n, const float* __restrict__ qs, uint32_t m, int dim,
                                                              uint32_t rows_per_slice, unsigned long long* __restrict__ keys)
{
#pragma clang fp contract(off)
    extern __shared__ float qsh[];                          // [dim][DS_BLOCK]
    const uint32_t qi = blockIdx.x * DS_BLOCK + threadIdx.x;
    const uint32_t qsafe = qi < m ? qi : m - 1;
    for (int d = 0; d < dim; d++) qsh[d * DS_BLOCK + threadIdx.x] = qs[(size_t)qsafe * dim + d];
    __syncthreads();
    const uint32_t r0 = blockIdx.y * rows_per_slice, r1 = min(n, r0 + rows_per_slice);
    unsigned long long best = ~0ull;
    for (uint32_t r = r0; r < r1; r++) {
        const float* row = db + (size_t)r * dim;
        float result = 0.0f;
        int d = 0;
        for (; d + 3 < dim; d += 4) {
            const float d0 = qsh[d * DS_BLOCK + threadIdx.x] - row[d], d1 = qsh[(d + 1) * DS_BLOCK + threadIdx.x] - row[d + 1],
                        d2 = qsh[(d + 2) * DS_BLOCK + threadIdx.x] - row[d + 2], d3 = qsh[(d + 3) * DS_BLOCK + threadIdx.x] - row[d + 3];
            result += ((d0 * d0 + d1 * d1) + d2 * d2) + d3 * d3;
        }
        for (; d < dim; d++) {
            const float d0 = qsh[d * DS_BLOCK + threadIdx.x] - row[d];
            result += d0 * d0;
        }
        if (result < FLT_MAX) {
            const unsigned long long key = ((unsigned long long)__float_as_uint(result) << 32) | r;
            best = key < best ? key : best;
        }
    }
    if (qi < m && best != ~0ull) atomicMin(&keys[qi], best);
}

__global__ void fill_keys_kernel(unsigned long long* keys, uint32_t m)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) keys[i] = ~0ull;
}

// device buffers: db | q | keys in the context scratch.  Results land in host idx / d2.
int nn1_dim(pcr_ctx* ctx, const float* db, size_t n, const float* q, size_t m, int dim, uint32_t* idx, float* d2)
{
    if (m == 0) return PCR_OK;
    const size_t db_b = (std::max<size_t>(n, 1) * dim * 4 + 255) & ~(size_t)255, q_b = (m * dim * 4 + 255) & ~(size_t)255, k_b = (m * 8 + 255) & ~(size_t)255;
    int rc = ensure_scratch(ctx, db_b + q_b + k_b);
    if (rc) return rc;
    char* s = (char*)ctx->scratch;
    float* ddb = (float*)s;
    float* dq = (float*)(s + db_b);
    unsigned long long* keys = (unsigned long long*)(s + db_b + q_b);
    if (n) PCR_HIP(ctx, hipMemcpyAsync(ddb, db, n * dim * 4, hipMemcpyHostToDevice, ctx->stream));
    PCR_HIP(ctx, hipMemcpyAsync(dq, q, m * dim * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(fill_keys_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, keys, (uint32_t)m);
    if (n) {
        const unsigned qblocks = (unsigned)((m + DS_BLOCK - 1) / DS_BLOCK);
        // enough workgroups to fill 256 CUs several times over, slices of whole tiles
        unsigned slices = (unsigned)std::min<size_t>((n + DS_ROWS - 1) / DS_ROWS, std::max<size_t>(1, 4096 / qblocks));
        uint32_t rows_per_slice = (uint32_t)(((n + slices - 1) / slices + DS_ROWS - 1) / DS_ROWS * DS_ROWS);
        slices = (unsigned)((n + rows_per_slice - 1) / rows_per_slice);
        ProfScope ps(ctx, "nn1_desc", 1);
        if (dim == 33)
            hipLaunchKernelGGL((nn1_dim_kernel<33>), dim3(qblocks, slices), dim3(DS_BLOCK), 0, ctx->stream, ddb, (uint32_t)n, dq, (uint32_t)m, rows_per_slice, keys);
        else if (dim == 3)
            hipLaunchKernelGGL((nn1_dim_kernel<3>), dim3(qblocks, slices), dim3(DS_BLOCK), 0, ctx->stream, ddb, (uint32_t)n, dq, (uint32_t)m, rows_per_slice, keys);
        else
            hipLaunchKernelGGL(nn1_anydim_kernel, dim3(qblocks, slices), dim3(DS_BLOCK), (size_t)dim * DS_BLOCK * 4, ctx->stream, ddb, (uint32_t)n, dq, (uint32_t)m, dim,
                               rows_per_slice, keys);
    }
    PCR_HIP(ctx, hipGetLastError());
    std::vector<unsigned long long> hk(m);
    PCR_HIP(ctx, hipMemcpyAsync(hk.data(), keys, m * 8, hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < m; i++) {
        if (hk[i] == ~0ull) { idx[i] = 0xFFFFFFFFu; d2[i] = INFINITY; continue; }
        idx[i] = (uint32_t)(hk[i] & 0xFFFFFFFFull);
        const uint32_t bits = (uint32_t)(hk[i] >> 32);
        memcpy(&d2[i], &bits, 4);
    }
    return PCR_OK;
}

// ------------------------------------------------------------------------------------------------- N4c
struct PairXyz { float s[3], t[3], pad[2]; };   // 32 B: one correspondence's source and target keypoint

__global__ __launch_bounds__(256) void ransac_hyp_kernel(const PairXyz* __restrict__ pairs, const uint32_t* __restrict__ quads, uint32_t n_hyp,
                                                         float* __restrict__ Rt, uint8_t* __restrict__ ok)
{
    const uint32_t h = blockIdx.x * 256 + threadIdx.x;
    if (h >= n_hyp) return;
    double sums[16];
#pragma unroll
    for (int k = 0; k < 16; k++) sums[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const PairXyz pr = pairs[quads[4 * (size_t)h + k]];
        const double p[3] = { pr.s[0], pr.s[1], pr.s[2] }, q[3] = { pr.t[0], pr.t[1], pr.t[2] };
#pragma unroll
        for (int c = 0; c < 3; c++) { sums[c] += p[c]; sums[3 + c] += q[c]; }
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) sums[6 + 3 * r + c] += q[r] * p[c];
        sums[15] += 1.0;
    }
    float R[9], t[3];
    const int rc = num::kabsch_solve_ransac(sums, R, t);   // t from U V^T, the reflection fix applies to R only (:382-392)
#pragma unroll
    for (int k = 0; k < 9; k++) Rt[12 * (size_t)h + k] = R[k];
#pragma unroll
    for (int k = 0; k < 3; k++) Rt[12 * (size_t)h + 9 + k] = t[k];
    ok[h] = rc == 0;
}

// one lane per hypothesis; gridDim.y chunks of the correspondence list (merged with atomicAdd)
__global__ __launch_bounds__(256) void consensus_kernel(const PairXyz* __restrict__ pairs, uint32_t n_pairs, uint32_t chunk, const float* __restrict__ Rt,
                                                        const uint8_t* __restrict__ ok, uint32_t n_hyp, float s_max, uint32_t* __restrict__ counts)
{
#pragma clang fp contract(off)
    const uint32_t h = blockIdx.x * 256 + threadIdx.x;
    const uint32_t hs = h < n_hyp ? h : n_hyp - 1;
    float R[9], t[3];
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = Rt[12 * (size_t)hs + k];
#pragma unroll
    for (int k = 0; k < 3; k++) t[k] = Rt[12 * (size_t)hs + 9 + k];
    const uint32_t p0 = blockIdx.y * chunk, p1 = min(n_pairs, p0 + chunk);
    uint32_t c = 0;
#pragma unroll 4
    for (uint32_t i = p0; i < p1; i++) {
        const PairXyz pr = pairs[i];                                       // wave-uniform: scalar loads, 4 in flight
        const float ex = pr.t[0] - (((R[0] * pr.s[0] + R[1] * pr.s[1]) + R[2] * pr.s[2]) + t[0]);
        const float ey = pr.t[1] - (((R[3] * pr.s[0] + R[4] * pr.s[1]) + R[5] * pr.s[2]) + t[1]);
        const float ez = pr.t[2] - (((R[6] * pr.s[0] + R[7] * pr.s[1]) + R[8] * pr.s[2]) + t[2]);
        const float s = (ex * ex + ey * ey) + ez * ez;
        c += s <= s_max;                                                   // sqrtf(s) <= thr, hoisted (see sqrt_threshold)
    }
    if (h < n_hyp && ok[h] && c) atomicAdd(&counts[h], c);
}

// largest float s with sqrtf(s) <= r (r >= 0); -1 when there is none
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

int check_pairs(pcr_ctx* ctx, const uint32_t* pairs, size_t n_pairs, size_t n_src, size_t n_tgt)
{
    for (size_t i = 0; i < n_pairs; i++)
        if (pairs[2 * i] >= n_src || pairs[2 * i + 1] >= n_tgt) return fail(ctx, PCR_ERR_ARG, "correspondence index out of range");
    return PCR_OK;
}

// uploads the gathered correspondences; returns the device pointer (in scratch) and the offset of the free space after it
int upload_pairs(pcr_ctx* ctx, const float* src_xyz, const float* tgt_xyz, const uint32_t* pairs, size_t n_pairs, size_t extra_bytes, PairXyz** dev, char** extra)
{
    const size_t pb = (std::max<size_t>(n_pairs, 1) * sizeof(PairXyz) + 255) & ~(size_t)255;
    int rc = ensure_scratch(ctx, pb + extra_bytes);
    if (rc) return rc;
    std::vector<PairXyz> host(n_pairs);
    for (size_t i = 0; i < n_pairs; i++) {
        for (int c = 0; c < 3; c++) {
            host[i].s[c] = src_xyz[3 * (size_t)pairs[2 * i] + c];
            host[i].t[c] = tgt_xyz[3 * (size_t)pairs[2 * i + 1] + c];
        }
        host[i].pad[0] = host[i].pad[1] = 0.f;
    }
    *dev = (PairXyz*)ctx->scratch;
    *extra = (char*)ctx->scratch + pb;
    if (n_pairs) {
        PCR_HIP(ctx, hipMemcpyAsync(*dev, host.data(), n_pairs * sizeof(PairXyz), hipMemcpyHostToDevice, ctx->stream));
        PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));     // `host` dies at return
    }
    return PCR_OK;
}

int launch_consensus(pcr_ctx* ctx, const PairXyz* dpairs, size_t n_pairs, const float* dRt, const uint8_t* dok, size_t n_hyp, float thr, uint32_t* dcounts)
{
    PCR_HIP(ctx, hipMemsetAsync(dcounts, 0, n_hyp * 4, ctx->stream));
    if (n_pairs == 0 || n_hyp == 0) return PCR_OK;
    const unsigned hb = (unsigned)((n_hyp + 255) / 256);
    // few hypotheses -> split the list so that the chip still sees a few thousand wavefronts
    unsigned chunks = (unsigned)std::min<size_t>(std::max<size_t>(1, 2048 / hb), (n_pairs + 255) / 256);
    const uint32_t chunk = (uint32_t)((n_pairs + chunks - 1) / chunks);
    chunks = (unsigned)((n_pairs + chunk - 1) / chunk);
    ProfScope ps(ctx, "consensus_count", 1);
    hipLaunchKernelGGL(consensus_kernel, dim3(hb, chunks), dim3(256), 0, ctx->stream, dpairs, (uint32_t)n_pairs, chunk, dRt, dok, (uint32_t)n_hyp,
                       sqrt_threshold(thr), dcounts);
    PCR_HIP(ctx, hipGetLastError());
    return PCR_OK;
}

}  // namespace

}  // namespace pcr

using namespace pcr;

extern "C" {

int pcr_nn1_desc_f32(pcr_ctx* ctx, const float* db, size_t n, const float* q, size_t m, int dim, uint32_t* idx, float* d2)
{
    if (!ctx || (n && !db) || (m && (!q || !idx || !d2)) || dim < 1 || dim > DS_MAX_DIM) return fail(ctx, PCR_ERR_ARG, "pcr_nn1_desc_f32");
    if (n > 0xFFFFFFF0ull || m > 0xFFFFFFF0ull) return fail(ctx, PCR_ERR_ARG, "pcr_nn1_desc_f32: too many rows");
    PCR_HIP(ctx, hipSetDevice(ctx->device));
    int rc = nn1_dim(ctx, db, n, q, m, dim, idx, d2);
    prof_flush(ctx);
    return rc;
}

int pcr_match_union_f32(pcr_ctx* ctx, const float* desc_src, size_t n_src, const float* desc_tgt, size_t n_tgt, int dim, float rejection_rate,
                        uint32_t* pairs, float* dist, size_t* n_pairs)
{
    if (!ctx || !n_pairs || (n_src && !desc_src) || (n_tgt && !desc_tgt) || dim < 1 || dim > DS_MAX_DIM) return fail(ctx, PCR_ERR_ARG, "pcr_match_union_f32");
    *n_pairs = 0;
    const size_t total = n_src + n_tgt;
    if (total == 0) return PCR_OK;
    if (!pairs || !dist) return fail(ctx, PCR_ERR_ARG, "pcr_match_union_f32: null output");
    if (total > 0xFFFFFFF0ull) return fail(ctx, PCR_ERR_ARG, "pcr_match_union_f32: too many descriptors");
    PCR_HIP(ctx, hipSetDevice(ctx->device));
    struct Rec { uint32_t s, t; float d; };
    std::vector<Rec> rec(total);
    std::vector<uint32_t> idx(std::max(n_src, n_tgt));
    std::vector<float> d2(std::max(n_src, n_tgt));
    int rc = nn1_dim(ctx, desc_src, n_src, desc_tgt, n_tgt, dim, idx.data(), d2.data());           // :561-577
    if (rc) return rc;
    for (size_t i = 0; i < n_tgt; i++) rec[i] = Rec{ idx[i], (uint32_t)i, d2[i] };
    rc = nn1_dim(ctx, desc_tgt, n_tgt, desc_src, n_src, dim, idx.data(), d2.data());               // :579-595
    if (rc) return rc;
    for (size_t i = 0; i < n_src; i++) rec[n_tgt + i] = Rec{ (uint32_t)i, idx[i], d2[i] };
    prof_flush(ctx);
    // :598-603 — std::sort there (tie order unspecified); stable here so that the kept set is reproducible
    std::stable_sort(rec.begin(), rec.end(), [](const Rec& a, const Rec& b) { return a.d < b.d; });
    const float keep_f = std::floor((1 - rejection_rate) * (float)total);                           // :605, float arithmetic as written
    size_t keep = keep_f > 0 ? (size_t)keep_f : 0;
    keep = std::min(keep, total);
    for (size_t i = 0; i < keep; i++) { pairs[2 * i] = rec[i].s; pairs[2 * i + 1] = rec[i].t; dist[i] = rec[i].d; }
    *n_pairs = keep;
    return PCR_OK;
}

int pcr_match_inter_f32(pcr_ctx* ctx, const float* desc_src, size_t n_src, const float* desc_tgt, size_t n_tgt, int dim, float rejection_rate,
                        uint32_t* pairs, float* dist, size_t* n_pairs)
{
    if (!ctx || !n_pairs || (n_src && !desc_src) || (n_tgt && !desc_tgt) || dim < 1 || dim > DS_MAX_DIM) return fail(ctx, PCR_ERR_ARG, "pcr_match_inter_f32");
    *n_pairs = 0;
    if (n_src == 0 || n_tgt == 0) return PCR_OK;
    if (!pairs || !dist) return fail(ctx, PCR_ERR_ARG, "pcr_match_inter_f32: null output");
    if (n_src + n_tgt > 0xFFFFFFF0ull) return fail(ctx, PCR_ERR_ARG, "pcr_match_inter_f32: too many descriptors");
    PCR_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<uint32_t> t2s(n_tgt), s2t(n_src);
    std::vector<float> dt(n_tgt), ds(n_src);
    int rc = nn1_dim(ctx, desc_src, n_src, desc_tgt, n_tgt, dim, t2s.data(), dt.data());           // :455-473
    if (rc) return rc;
    rc = nn1_dim(ctx, desc_tgt, n_tgt, desc_src, n_src, dim, s2t.data(), ds.data());               // :475-494
    if (rc) return rc;
    prof_flush(ctx);
    struct Rec { uint32_t s, t; float d; };
    std::vector<Rec> rec;
    for (size_t s = 0; s < n_src; s++) {                                                            // :497-508: mutual pairs, ascending s
        const uint32_t t = s2t[s];
        if (t == 0xFFFFFFFFu) continue;
        if (t2s[t] == (uint32_t)s) rec.push_back(Rec{ (uint32_t)s, t, ds[s] });
    }
    std::stable_sort(rec.begin(), rec.end(), [](const Rec& a, const Rec& b) { return a.d < b.d; });   // :519-521
    const float keep_f = std::floor((1 - rejection_rate) * (float)rec.size());                      // :523
    size_t keep = keep_f > 0 ? (size_t)keep_f : 0;
    keep = std::min(keep, rec.size());
    for (size_t i = 0; i < keep; i++) { pairs[2 * i] = rec[i].s; pairs[2 * i + 1] = rec[i].t; dist[i] = rec[i].d; }
    *n_pairs = keep;
    return PCR_OK;
}

// Host logic (no GPU): the sampling loop of Registration::RANSAC (:318-352) with an explicit seed instead of
// std::random_device: four distinct-as-the-reference-checks correspondences whose SOURCE keypoints are not coplanar
// (signed distance of the 4th from the plane of the first three > 0.15, f32 as written).
int pcr_ransac_sample_quads(const float* src_xyz, size_t n_src, const uint32_t* pairs, size_t n_pairs, size_t n_hyp, uint64_t seed, uint32_t* quads)
{
#pragma clang fp contract(off)
    if (!src_xyz || !pairs || (n_hyp && !quads) || n_pairs < 4 || n_pairs > 0xFFFFFFF0ull) return PCR_ERR_ARG;
    for (size_t i = 0; i < n_pairs; i++)
        if (pairs[2 * i] >= n_src) return PCR_ERR_ARG;
    std::mt19937 mt((std::mt19937::result_type)seed);
    std::uniform_int_distribution<size_t> dist(0, n_pairs - 1);                                    // :301
    for (size_t iter = 0; iter < n_hyp; iter++) {
        size_t random_idx[4];
        size_t attempts = 0;
        while (true) {
            if (++attempts > 1000000) return PCR_ERR_STATE;                                        // all-coplanar input: the reference would spin forever
            random_idx[0] = dist(mt);
            for (size_t i = 1; i < 4; i++) {                                                       // :324-332, as written
                random_idx[i] = dist(mt);
                for (size_t j = 0; j < i; j++)
                    while (random_idx[i] == random_idx[j]) random_idx[i] = dist(mt);
            }
            const float* a = src_xyz + 3 * (size_t)pairs[2 * random_idx[0]];
            const float* b = src_xyz + 3 * (size_t)pairs[2 * random_idx[1]];
            const float* c = src_xyz + 3 * (size_t)pairs[2 * random_idx[2]];
            const float* d = src_xyz + 3 * (size_t)pairs[2 * random_idx[3]];
            const float p1[3] = { b[0] - a[0], b[1] - a[1], b[2] - a[2] };
            const float p2[3] = { c[0] - a[0], c[1] - a[1], c[2] - a[2] };
            const float p3[3] = { d[0] - a[0], d[1] - a[1], d[2] - a[2] };
            const float normal[3] = { (p1[1] * p2[2] - p1[2] * p2[1]), (p1[2] * p2[0] - p1[0] * p2[2]), (p1[0] * p2[1] - p1[1] * p2[0]) };
            const float normal_length = std::sqrt(normal[0] * normal[0] + normal[1] * normal[1] + normal[2] * normal[2]);
            const float distance = (normal[0] * p3[0] + normal[1] * p3[1] + normal[2] * p3[2]) / normal_length;
            if (distance > 0.15) break;                                                            // :350 (float against the double literal)
        }
        for (int k = 0; k < 4; k++) quads[4 * iter + k] = (uint32_t)random_idx[k];
    }
    return PCR_OK;
}

int pcr_consensus_count_f32(pcr_ctx* ctx, const float* src_xyz, size_t n_src, const float* tgt_xyz, size_t n_tgt, const uint32_t* pairs, size_t n_pairs,
                            const float* Rt, size_t n_hyp, float thr, uint32_t* counts)
{
    if (!ctx || (n_pairs && (!src_xyz || !tgt_xyz || !pairs)) || (n_hyp && (!Rt || !counts))) return fail(ctx, PCR_ERR_ARG, "pcr_consensus_count_f32");
    if (n_pairs > 0xFFFFFFF0ull || n_hyp > 0xFFFFFFF0ull) return fail(ctx, PCR_ERR_ARG, "pcr_consensus_count_f32: too large");
    if (n_hyp == 0) return PCR_OK;
    int rc = check_pairs(ctx, pairs, n_pairs, n_src, n_tgt);
    if (rc) return rc;
    PCR_HIP(ctx, hipSetDevice(ctx->device));
    const size_t rb = (n_hyp * 48 + 255) & ~(size_t)255, cb = (n_hyp * 4 + 255) & ~(size_t)255, ob = (n_hyp + 255) & ~(size_t)255;
    PairXyz* dpairs = nullptr;
    char* extra = nullptr;
    rc = upload_pairs(ctx, src_xyz, tgt_xyz, pairs, n_pairs, rb + cb + ob, &dpairs, &extra);
    if (rc) return rc;
    float* dRt = (float*)extra;
    uint32_t* dcounts = (uint32_t*)(extra + rb);
    uint8_t* dok = (uint8_t*)(extra + rb + cb);
    PCR_HIP(ctx, hipMemcpyAsync(dRt, Rt, n_hyp * 48, hipMemcpyHostToDevice, ctx->stream));
    PCR_HIP(ctx, hipMemsetAsync(dok, 1, n_hyp, ctx->stream));
    rc = launch_consensus(ctx, dpairs, n_pairs, dRt, dok, n_hyp, thr, dcounts);
    if (rc) return rc;
    PCR_HIP(ctx, hipMemcpyAsync(counts, dcounts, n_hyp * 4, hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    prof_flush(ctx);
    return PCR_OK;
}

int pcr_ransac_global_f32(pcr_ctx* ctx, const float* src_xyz, size_t n_src, const float* tgt_xyz, size_t n_tgt, const uint32_t* pairs, size_t n_pairs,
                          const uint32_t* quads, size_t n_hyp, float thr, float R[9], float t[3], uint32_t* best_count, int64_t* winner, uint32_t* counts)
{
    if (!ctx || !R || !t || (n_pairs && (!src_xyz || !tgt_xyz || !pairs)) || (n_hyp && !quads)) return fail(ctx, PCR_ERR_ARG, "pcr_ransac_global_f32");
    if (n_pairs > 0xFFFFFFF0ull || n_hyp > 0xFFFFFFF0ull) return fail(ctx, PCR_ERR_ARG, "pcr_ransac_global_f32: too large");
    if (best_count) *best_count = 0;
    if (winner) *winner = -1;
    if (n_hyp == 0) return PCR_OK;
    int rc = check_pairs(ctx, pairs, n_pairs, n_src, n_tgt);
    if (rc) return rc;
    for (size_t i = 0; i < 4 * n_hyp; i++)
        if (quads[i] >= n_pairs) return fail(ctx, PCR_ERR_ARG, "pcr_ransac_global_f32: quad index out of range");
    PCR_HIP(ctx, hipSetDevice(ctx->device));
    const size_t rb = (n_hyp * 48 + 255) & ~(size_t)255, cb = (n_hyp * 4 + 255) & ~(size_t)255, ob = (n_hyp + 255) & ~(size_t)255, qb = (n_hyp * 16 + 255) & ~(size_t)255;
    PairXyz* dpairs = nullptr;
    char* extra = nullptr;
    rc = upload_pairs(ctx, src_xyz, tgt_xyz, pairs, n_pairs, rb + cb + ob + qb, &dpairs, &extra);
    if (rc) return rc;
    float* dRt = (float*)extra;
    uint32_t* dcounts = (uint32_t*)(extra + rb);
    uint8_t* dok = (uint8_t*)(extra + rb + cb);
    uint32_t* dquads = (uint32_t*)(extra + rb + cb + ob);
    PCR_HIP(ctx, hipMemcpyAsync(dquads, quads, n_hyp * 16, hipMemcpyHostToDevice, ctx->stream));
    {
        ProfScope ps(ctx, "ransac_hypotheses", 1);
        hipLaunchKernelGGL(ransac_hyp_kernel, dim3((unsigned)((n_hyp + 255) / 256)), dim3(256), 0, ctx->stream, dpairs, dquads, (uint32_t)n_hyp, dRt, dok);
    }
    rc = launch_consensus(ctx, dpairs, n_pairs, dRt, dok, n_hyp, thr, dcounts);
    if (rc) return rc;
    std::vector<uint32_t> hc(n_hyp);
    PCR_HIP(ctx, hipMemcpyAsync(hc.data(), dcounts, n_hyp * 4, hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    uint32_t best = 0;
    int64_t win = -1;
    for (size_t h = 0; h < n_hyp; h++)
        if (hc[h] > best) { best = hc[h]; win = (int64_t)h; }                                       // :423, strict: the first maximum wins
    if (win >= 0) {
        float hRt[12];
        PCR_HIP(ctx, hipMemcpy(hRt, dRt + 12 * (size_t)win, sizeof hRt, hipMemcpyDeviceToHost));
        memcpy(R, hRt, 36);
        memcpy(t, hRt + 9, 12);
    }
    if (best_count) *best_count = best;
    if (winner) *winner = win;
    if (counts) memcpy(counts, hc.data(), n_hyp * 4);
    prof_flush(ctx);
    return PCR_OK;
}

}  // extern "C"

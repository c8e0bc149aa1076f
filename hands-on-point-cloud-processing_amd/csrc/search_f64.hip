// search_f64.hip — batched k-NN and radius-NN with the hw2 arithmetic (A2/A4/A11) for gfx950.
// Reference (one query per call, pointer-chasing kd-tree): Homework2/hw2/include/kdtree.hpp:329-402 with the
// result sets of resultSet.hpp:28-142; batched consumer Homework7/hw7/src/iss_detector.cpp:48-56.
//   d = sqrt(((0 + (t0-q0)^2) + (t1-q1)^2) + (t2-q2)^2)   in f64 (kdtree.hpp:341-346)
//   k-NN:   the k smallest d, canonical order (d ascending, then index ascending)
//   radius: every j with d <= r (inclusive, resultSet.hpp:133), ascending index, CSR
// The tree is replaced by an exhaustive LDS-tiled scan (same answers; the tree is only an index).
// One query per lane; the database streams through LDS in SoA f64 tiles and is read as wave-wide
// broadcasts.  The scan visits indices in ascending order and only a strictly smaller distance displaces an
// entry, which is exactly the canonical tie rule.
// Radius membership avoids the sqrt: sqrt is monotone and correctly rounded, so d <= r  <=>  s <= r2max
// with r2max = max{ s : sqrt(s) <= r }, computed once on the host; sqrt runs only for reported neighbours.
#include "pcr_internal.hpp"

#include <cmath>
#include <vector>

#pragma clang fp contract(off)

namespace pcr {

constexpr int SF_BLOCK = 256;
constexpr int SF_TILE = 512;   // db points per LDS tile: 3 * 4 KiB of f64

__device__ __forceinline__ double dist2_f64(double t0, double t1, double t2, double q0, double q1, double q2)
{
    const double e0 = t0 - q0, e1 = t1 - q1, e2 = t2 - q2;
    return (e0 * e0 + e1 * e1) + e2 * e2;
}

// SQ = false: hw2 contract (d = sqrt(s), kdtree.hpp:346); SQ = true: nanoflann contract (squared L2, no sqrt,
// nanoflann.hpp:403-406 with T = double) - ordering then happens on s itself.
// Small batches: the database is cut into gridDim.y slices of tiles_per_slice tiles, each (query block, slice) workgroup keeps
// the top k of its slice (written to the slice's own [m x k] block of the output), knn_merge_kernel folds the slices in
// ascending order — the same scan order, so the same canonical result.  gridDim.y == 1: the whole database, final output.
template <int K, bool SQ>
__global__ __launch_bounds__(SF_BLOCK) void knn_f64_kernel(
    const double* __restrict__ db, uint32_t n, uint32_t n_cap, const double* __restrict__ q, uint32_t m,
    uint32_t m_cap, int k_out, int32_t* __restrict__ idx_out, double* __restrict__ dist_out, uint32_t tiles_per_slice)
{
    __shared__ double l0[SF_TILE], l1[SF_TILE], l2[SF_TILE];
    const uint32_t qi = blockIdx.x * SF_BLOCK + threadIdx.x;
    const uint32_t qc = min(qi, m - 1);
    const double q0 = q[qc], q1 = q[m_cap + qc], q2 = q[2 * (size_t)m_cap + qc];
    double bd[K];
    int32_t bi[K];
#pragma unroll
    // hw2: slots pre-filled with (1e10, 0), resultSet.hpp:35-42; nanoflann: worst = DBL_MAX (nanoflann.hpp:163)
    for (int s = 0; s < K; s++) { bd[s] = SQ ? 1.7976931348623157e308 : 1e10; bi[s] = SQ ? -1 : 0; }
    const uint32_t slice_lo = blockIdx.y * tiles_per_slice * SF_TILE;
    const uint32_t slice_hi = (uint32_t)min((unsigned long long)n, (unsigned long long)slice_lo + (unsigned long long)tiles_per_slice * SF_TILE);
    for (uint32_t base = slice_lo; base < slice_hi; base += SF_TILE) {
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < SF_TILE; t += SF_BLOCK) {
            const uint32_t j = base + t;    // < n_cap by construction (n_cap is a multiple of SF_TILE)
            l0[t] = db[j]; l1[t] = db[n_cap + j]; l2[t] = db[2 * (size_t)n_cap + j];
        }
        __syncthreads();
        const uint32_t cnt = min((uint32_t)SF_TILE, slice_hi - base);
        for (uint32_t t = 0; t < cnt; t++) {
            const double s2 = dist2_f64(l0[t], l1[t], l2[t], q0, q1, q2);
            const double d = SQ ? s2 : sqrt(s2);
            // resultSet.hpp:69 rejects only d > worst; the canonical rule additionally keeps the earlier
            // (lower) index on equality, i.e. insert only when strictly smaller than the current worst
            if (d < bd[K - 1]) {
                const int32_t j = (int32_t)(base + t);
                double cd = d;
                int32_t ci = j;
                bool placed = false;
#pragma unroll
                for (int s = 0; s < K; s++) {
                    // sorted insertion: the candidate goes in front of the first strictly larger entry (equal
                    // entries have lower indices and stay in front); everything behind shifts by one slot
                    const bool sw = placed || cd < bd[s];
                    placed = sw;
                    const double td = bd[s];
                    const int32_t ti = bi[s];
                    bd[s] = sw ? cd : td;
                    bi[s] = sw ? ci : ti;
                    cd = sw ? td : cd;
                    ci = sw ? ti : ci;
                }
            }
        }
    }
    if (qi < m) {
        const size_t o = ((size_t)blockIdx.y * m + qi) * (size_t)k_out;
#pragma unroll
        for (int s = 0; s < K; s++) {
            if (s < k_out) {
                idx_out[o + s] = bi[s];
                dist_out[o + s] = bd[s];
            }
        }
    }
}

// one lane per query: the per-slice top-k lists (each sorted, slices in ascending index order) -> the final list.  An entry is
// inserted only if strictly smaller than the current worst (equal distances keep the earlier = lower index, the canonical
// rule); a slice's list is left as soon as one of its entries fails, the rest being no smaller.
template <int K>
__global__ __launch_bounds__(SF_BLOCK) void knn_merge_kernel(const int32_t* __restrict__ pidx, const double* __restrict__ pdist, uint32_t m, int k_out,
                                                             uint32_t slices, double empty_val, int32_t empty_idx, int32_t* __restrict__ idx_out,
                                                             double* __restrict__ dist_out)
{
    const uint32_t qi = blockIdx.x * SF_BLOCK + threadIdx.x;
    if (qi >= m) return;
    double bd[K];
    int32_t bi[K];
#pragma unroll
    for (int s = 0; s < K; s++) { bd[s] = empty_val; bi[s] = empty_idx; }
    for (uint32_t sl = 0; sl < slices; sl++) {
        const size_t o = ((size_t)sl * m + qi) * (size_t)k_out;
        for (int e = 0; e < k_out; e++) {
            const double d = pdist[o + e];
            const int32_t j = pidx[o + e];
            if (!(d < bd[K - 1]) || (d == empty_val && j == empty_idx)) break;       // (placeholders of a short slice never enter)
            double cd = d;
            int32_t ci = j;
            bool placed = false;
#pragma unroll
            for (int s = 0; s < K; s++) {
                const bool sw = placed || cd < bd[s];
                placed = sw;
                const double td = bd[s];
                const int32_t ti = bi[s];
                bd[s] = sw ? cd : td;
                bi[s] = sw ? ci : ti;
                cd = sw ? td : cd;
                ci = sw ? ti : ci;
            }
        }
    }
#pragma unroll
    for (int s = 0; s < K; s++)
        if (s < k_out) { idx_out[(size_t)qi * k_out + s] = bi[s]; dist_out[(size_t)qi * k_out + s] = bd[s]; }
}

// FILL = false: counts[q] = #{j : s_j <= r2max}; FILL = true: write idx/dist at row_ptr[q]...
// slices as in knn_f64_kernel: counts / write offsets are per (query, slice), query-major — counts[q * gridDim.y + slice] — so
// that the host's running sum puts the slices of a row one after the other in ascending index order
template <bool FILL>
__global__ __launch_bounds__(SF_BLOCK) void radius_f64_kernel(
    const double* __restrict__ db, uint32_t n, uint32_t n_cap, const double* __restrict__ q, uint32_t m,
    uint32_t m_cap, double r2max, unsigned long long* __restrict__ counts, const long long* __restrict__ row_ptr,
    int32_t* __restrict__ idx_out, double* __restrict__ dist_out, uint32_t tiles_per_slice)
{
    __shared__ double l0[SF_TILE], l1[SF_TILE], l2[SF_TILE];
    const uint32_t qi = blockIdx.x * SF_BLOCK + threadIdx.x;
    const uint32_t qc = min(qi, m - 1);
    const double q0 = q[qc], q1 = q[m_cap + qc], q2 = q[2 * (size_t)m_cap + qc];
    unsigned long long c = 0;
    long long w = 0;
    if (FILL) w = row_ptr[(size_t)qc * gridDim.y + blockIdx.y];
    const uint32_t slice_lo = blockIdx.y * tiles_per_slice * SF_TILE;
    const uint32_t slice_hi = (uint32_t)min((unsigned long long)n, (unsigned long long)slice_lo + (unsigned long long)tiles_per_slice * SF_TILE);
    for (uint32_t base = slice_lo; base < slice_hi; base += SF_TILE) {
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < SF_TILE; t += SF_BLOCK) {
            const uint32_t j = base + t;
            l0[t] = db[j]; l1[t] = db[n_cap + j]; l2[t] = db[2 * (size_t)n_cap + j];
        }
        __syncthreads();
        const uint32_t cnt = min((uint32_t)SF_TILE, slice_hi - base);
        for (uint32_t t = 0; t < cnt; t++) {
            const double s = dist2_f64(l0[t], l1[t], l2[t], q0, q1, q2);
            if (s <= r2max) {
                if (FILL) {
                    if (qi < m) {
                        idx_out[w] = (int32_t)(base + t);
                        dist_out[w] = sqrt(s);
                        w++;
                    }
                } else {
                    c++;
                }
            }
        }
    }
    if (!FILL && qi < m) counts[(size_t)qi * gridDim.y + blockIdx.y] = c;
}

// How many database slices a batch of m queries is spread over: one query per lane leaves the chip idle when m is small
// (a single query scanned 100 000 points with ONE lane: 13 ms), so small batches cut the database into enough slices for
// ~1024 workgroups; `bytes_per_query_slice` bounds the partial results (4 MB).
static uint32_t search_slices(size_t n, size_t m, size_t bytes_per_query_slice)
{
    const size_t n_tiles = std::max<size_t>(1, (n + SF_TILE - 1) / SF_TILE);
    const size_t qblocks = (m + SF_BLOCK - 1) / SF_BLOCK;
    size_t slices = std::min(n_tiles, std::max<size_t>(1, 1024 / qblocks));
    const size_t mem_cap = std::max<size_t>(1, ((size_t)4 << 20) / std::max<size_t>(1, m * bytes_per_query_slice));
    slices = std::min(slices, mem_cap);
    if (slices < 2) return 1;
    const size_t tps = (n_tiles + slices - 1) / slices;
    return (uint32_t)((n_tiles + tps - 1) / tps);
}

int launch_knn_f64(pcr_ctx* ctx, const double* db_soa, size_t n, size_t n_cap, const double* q_soa, size_t m, int k,
                   int32_t* idx_dev, double* dist_dev, bool squared)
{
    const uint32_t blocks = (uint32_t)((m + SF_BLOCK - 1) / SF_BLOCK);
    const size_t m_cap = ((m + SF_TILE - 1) / SF_TILE) * SF_TILE;
    const size_t n_tiles = std::max<size_t>(1, (n + SF_TILE - 1) / SF_TILE);
    const uint32_t slices = tune_get(ctx, "knn_slices", 1) == 1 ? search_slices(n, m, (size_t)k * 12) : 1u;
    const uint32_t tps = (uint32_t)((n_tiles + slices - 1) / slices);
    int32_t* pidx = idx_dev;
    double* pdist = dist_dev;
    if (slices > 1) {
        const size_t pd = (((size_t)slices * m * (size_t)k * 8) + 255) & ~(size_t)255;
        int rc = ensure_aux(ctx, pd + (size_t)slices * m * (size_t)k * 4 + 256);
        if (rc) return rc;
        pdist = (double*)ctx->aux;
        pidx = (int32_t*)((char*)ctx->aux + pd);
    }
    ProfScope p(ctx, "knn_f64");
#define PCR_KNN(K)                                                                                              \
    do {                                                                                                        \
        if (squared)                                                                                            \
            hipLaunchKernelGGL((knn_f64_kernel<K, true>), dim3(blocks, slices), dim3(SF_BLOCK), 0, ctx->stream, db_soa,  \
                               (uint32_t)n, (uint32_t)n_cap, q_soa, (uint32_t)m, (uint32_t)m_cap, k, pidx, pdist, tps); \
        else                                                                                                    \
            hipLaunchKernelGGL((knn_f64_kernel<K, false>), dim3(blocks, slices), dim3(SF_BLOCK), 0, ctx->stream, db_soa, \
                               (uint32_t)n, (uint32_t)n_cap, q_soa, (uint32_t)m, (uint32_t)m_cap, k, pidx, pdist, tps); \
        if (slices > 1)                                                                                         \
            hipLaunchKernelGGL((knn_merge_kernel<K>), dim3(blocks), dim3(SF_BLOCK), 0, ctx->stream, pidx, pdist, (uint32_t)m, k, slices, \
                               squared ? 1.7976931348623157e308 : 1e10, squared ? -1 : 0, idx_dev, dist_dev);     \
    } while (0)
    if (k <= 1) PCR_KNN(1);
    else if (k <= 4) PCR_KNN(4);
    else if (k <= 8) PCR_KNN(8);
    else if (k <= 16) PCR_KNN(16);
    else PCR_KNN(32);
#undef PCR_KNN
    PCR_HIP(ctx, hipGetLastError());
    return PCR_OK;
}

// counts_dev: m * slices entries, query-major (see radius_f64_kernel)
int launch_radius_count(pcr_ctx* ctx, const double* db_soa, size_t n, size_t n_cap, const double* q_soa, size_t m,
                        double r2max, unsigned long long* counts_dev, uint32_t slices)
{
    const uint32_t blocks = (uint32_t)((m + SF_BLOCK - 1) / SF_BLOCK);
    const size_t m_cap = ((m + SF_TILE - 1) / SF_TILE) * SF_TILE;
    const size_t n_tiles = std::max<size_t>(1, (n + SF_TILE - 1) / SF_TILE);
    const uint32_t tps = (uint32_t)((n_tiles + slices - 1) / slices);
    ProfScope p(ctx, "radius_count");
    hipLaunchKernelGGL(radius_f64_kernel<false>, dim3(blocks, slices), dim3(SF_BLOCK), 0, ctx->stream, db_soa, (uint32_t)n,
                       (uint32_t)n_cap, q_soa, (uint32_t)m, (uint32_t)m_cap, r2max, counts_dev,
                       (const long long*)nullptr, (int32_t*)nullptr, (double*)nullptr, tps);
    PCR_HIP(ctx, hipGetLastError());
    return PCR_OK;
}

// row_ptr_dev: write offset of every (query, slice), query-major
int launch_radius_fill(pcr_ctx* ctx, const double* db_soa, size_t n, size_t n_cap, const double* q_soa, size_t m,
                       double r2max, const long long* row_ptr_dev, int32_t* idx_dev, double* dist_dev, uint32_t slices)
{
    const uint32_t blocks = (uint32_t)((m + SF_BLOCK - 1) / SF_BLOCK);
    const size_t m_cap = ((m + SF_TILE - 1) / SF_TILE) * SF_TILE;
    const size_t n_tiles = std::max<size_t>(1, (n + SF_TILE - 1) / SF_TILE);
    const uint32_t tps = (uint32_t)((n_tiles + slices - 1) / slices);
    ProfScope p(ctx, "radius_fill");
    hipLaunchKernelGGL(radius_f64_kernel<true>, dim3(blocks, slices), dim3(SF_BLOCK), 0, ctx->stream, db_soa, (uint32_t)n,
                       (uint32_t)n_cap, q_soa, (uint32_t)m, (uint32_t)m_cap, r2max, (unsigned long long*)nullptr,
                       row_ptr_dev, idx_dev, dist_dev, tps);
    PCR_HIP(ctx, hipGetLastError());
    return PCR_OK;
}

// max{ s >= 0 : sqrt(s) <= r } for r >= 0 (host sqrt is IEEE correctly rounded)
static double radius_sq_bound(double r)
{
    if (!(r >= 0.0)) return -1.0;                 // nothing qualifies (also NaN)
    if (std::isinf(r)) return r;
    double s = r * r;
    while (std::sqrt(s) > r) s = std::nextafter(s, 0.0);
    for (;;) {
        const double up = std::nextafter(s, INFINITY);
        if (std::isinf(up) || std::sqrt(up) > r) break;
        s = up;
    }
    return s;
}

// host AoS (n x 3) -> device SoA with capacity `cap` (a multiple of SF_TILE); padding = 0
static int upload_soa_f64(pcr_ctx* ctx, const double* aos, size_t n, size_t cap, double* dev)
{
    int rc = ensure_stage(ctx, 3 * cap * sizeof(double));
    if (rc) return rc;
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double* st = (double*)ctx->host_stage;
    for (size_t i = 0; i < n; i++) {
        st[i] = aos[3 * i];
        st[cap + i] = aos[3 * i + 1];
        st[2 * cap + i] = aos[3 * i + 2];
    }
    for (size_t i = n; i < cap; i++) { st[i] = 0.0; st[cap + i] = 0.0; st[2 * cap + i] = 0.0; }
    PCR_HIP(ctx, hipMemcpyAsync(dev, st, 3 * cap * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PCR_OK;
}

// true when every value survives the round trip through f32 (then f32 -> f64 widening reproduces the input exactly)
static bool f32_exact(const double* aos, size_t n, std::vector<float>& out)
{
    out.resize(3 * n);
    for (size_t i = 0; i < 3 * n; i++) {
        const float f = (float)aos[i];
        if (!((double)f == aos[i])) return false;          // also false for NaN
        out[i] = f;
    }
    return true;
}

static size_t tile_cap(size_t n) { return ((n + SF_TILE - 1) / SF_TILE) * SF_TILE + SF_TILE; }

}  // namespace pcr

// a database resident in HBM (SoA f64), the GPU-side counterpart of the tree KDTreeConstruction returns
struct pcr_db64 {
    size_t n = 0;
    size_t cap = 0;
    double* dev = nullptr;
    pcr_cloud* twin = nullptr;   // the same points as an f32 cloud when every coordinate is f32-representable (KITTI / PLY data
                                 // widened to f64, test.hpp:28): large k-NN batches then take the exact grid search
};

using namespace pcr;

extern "C" int pcr_db64_create(pcr_ctx* ctx, const double* db, size_t n, pcr_db64** out)
{
    if (!ctx || !out || (n && !db)) return fail(ctx, PCR_ERR_ARG, "pcr_db64_create");
    if (n > 0x7FFFFFF0ull) return fail(ctx, PCR_ERR_ARG, "pcr_db64_create: too large for i32 indices");
    PCR_HIP(ctx, hipSetDevice(ctx->device));
    pcr_db64* h = new (std::nothrow) pcr_db64();
    if (!h) return fail(ctx, PCR_ERR_NOMEM, "pcr_db64_create");
    h->n = n;
    h->cap = tile_cap(n);
    hipError_t e = hipMalloc((void**)&h->dev, 3 * h->cap * sizeof(double));
    if (e != hipSuccess) { delete h; return fail(ctx, PCR_ERR_HIP, "hipMalloc(db64)", e); }
    int rc = upload_soa_f64(ctx, db, n, h->cap, h->dev);
    if (rc) { hipFree(h->dev); delete h; return rc; }
    if (n >= 4096) {
        std::vector<float> f32;
        if (f32_exact(db, n, f32)) {
            rc = pcr_cloud_create(ctx, f32.data(), n, PCR_AOS3, &h->twin);
            if (rc) { hipFree(h->dev); delete h; return rc; }
        }
    }
    *out = h;
    return PCR_OK;
}

extern "C" int pcr_db64_destroy(pcr_ctx* ctx, pcr_db64* db)
{
    if (!db) return PCR_OK;
    if (ctx) hipStreamSynchronize(ctx->stream);
    if (db->dev) hipFree(db->dev);
    if (db->twin) pcr_cloud_destroy(ctx, db->twin);
    delete db;
    return PCR_OK;
}

extern "C" size_t pcr_db64_size(const pcr_db64* db) { return db ? db->n : 0; }

extern "C" int pcr_db64_knn(pcr_ctx* ctx, const pcr_db64* db, const double* q, size_t m, int k, int squared,
                            int32_t* idx, double* dist)
{
    if (!ctx || !db || k < 1 || k > 32 || (m && (!q || !idx || !dist))) return fail(ctx, PCR_ERR_ARG, "pcr_db64_knn");
    if (m == 0) return PCR_OK;
    if (m > 0x7FFFFFF0ull) return fail(ctx, PCR_ERR_ARG, "pcr_db64_knn: too many queries");
    PCR_HIP(ctx, hipSetDevice(ctx->device));
    // f32-representable data (KITTI / PLY floats widened to f64): the exact grid search returns the same answers (same
    // arithmetic, same canonical order) without touching all n x m pairs — small batches through one zero-copy launch
    // (cloud_knn_small), large ones through a query cloud; knn_method 1 forces the exhaustive scan
    const int64_t method = tune_get(ctx, "knn_method", 0);
    if (db->twin && method != 1) {
        std::vector<float> qf;
        if (f32_exact(q, m, qf)) {
            if (m <= KNN_SMALL_MAX && method != 3)       // (3 = the batch route even for a small batch: A/B)
                return cloud_knn_small(ctx, db->twin, qf.data(), m, k, INFINITY, squared != 0, squared ? 1.7976931348623157e308 : 1e10, squared ? -1 : 0, idx, dist);
            pcr_cloud* qc = nullptr;
            int rcq = pcr_cloud_create(ctx, qf.data(), m, PCR_AOS3, &qc);
            if (rcq) return rcq;
            rcq = cloud_knn_host(ctx, db->twin, qc, k, INFINITY, squared != 0, squared ? 1.7976931348623157e308 : 1e10, squared ? -1 : 0, idx, dist, nullptr);
            pcr_cloud_destroy(ctx, qc);
            return rcq;
        }
    }
    const size_t m_cap = ((m + SF_TILE - 1) / SF_TILE) * SF_TILE;
    const size_t bytes_q = 3 * m_cap * 8, bytes_i = m * (size_t)k * 4, bytes_d = m * (size_t)k * 8;
    const size_t off_d = bytes_q, off_i = off_d + ((bytes_d + 63) & ~(size_t)63);
    int rc = ensure_scratch(ctx, off_i + bytes_i + 64);
    if (rc) return rc;
    char* s = (char*)ctx->scratch;
    if ((rc = upload_soa_f64(ctx, q, m, m_cap, (double*)s))) return rc;
    if ((rc = launch_knn_f64(ctx, db->dev, db->n, db->cap, (double*)s, m, k, (int32_t*)(s + off_i), (double*)(s + off_d), squared != 0))) return rc;
    PCR_HIP(ctx, hipMemcpyAsync(idx, s + off_i, bytes_i, hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP(ctx, hipMemcpyAsync(dist, s + off_d, bytes_d, hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PCR_OK;
}

extern "C" int pcr_db64_radius(pcr_ctx* ctx, const pcr_db64* db, const double* q, size_t m, double r,
                               int64_t* row_ptr, int32_t* idx, double* dist)
{
    if (!ctx || !db || !row_ptr || (m && !q) || ((idx == nullptr) != (dist == nullptr)))
        return fail(ctx, PCR_ERR_ARG, "pcr_db64_radius");
    row_ptr[0] = 0;
    if (m == 0) return PCR_OK;
    if (m > 0x7FFFFFF0ull) return fail(ctx, PCR_ERR_ARG, "pcr_db64_radius: too many queries");
    PCR_HIP(ctx, hipSetDevice(ctx->device));
    const double r2max = radius_sq_bound(r);
    // large batches on f32-representable data: the grid walk (27 cells of edge 1.01 r, rows cut to |x - qx| <= r) returns the
    // same CSR rows without touching all n x m pairs; radius_method 1 forces the exhaustive scan, 2 the grid
    const int64_t rmethod = tune_get(ctx, "radius_method", 0);
    if (db->twin && rmethod != 1) {
        std::vector<float> qf;
        if (f32_exact(q, m, qf)) {
            bool used = false;
            int rcq;
            if (m <= KNN_SMALL_MAX) {
                // small batch: the queries stay in pinned host memory (a cloud VIEW over it: no allocation, no upload)
                const size_t mp = (m + 63) & ~(size_t)63;
                rcq = ensure_stage(ctx, 3 * mp * 4 + 256);
                if (rcq) return rcq;
                PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
                float* st = (float*)ctx->host_stage;
                for (size_t i = 0; i < m; i++) { st[i] = qf[3 * i]; st[mp + i] = qf[3 * i + 1]; st[2 * mp + i] = qf[3 * i + 2]; }
                pcr_cloud view;
                view.n = m; view.cap = mp; view.base = st;
                rcq = radius_grid(ctx, db->twin, &view, r, r2max, row_ptr, idx, dist, &used);
                view.base = nullptr;
            } else {
                pcr_cloud* qc = nullptr;
                rcq = pcr_cloud_create(ctx, qf.data(), m, PCR_AOS3, &qc);
                if (rcq) return rcq;
                rcq = radius_grid(ctx, db->twin, qc, r, r2max, row_ptr, idx, dist, &used);
                pcr_cloud_destroy(ctx, qc);
            }
            if (rcq) return rcq;
            prof_flush(ctx);
            if (used) return PCR_OK;
        }
    }
    const size_t n = db->n, n_cap = db->cap;
    const size_t m_cap = ((m + SF_TILE - 1) / SF_TILE) * SF_TILE;
    // small batches: the database in slices (search_slices); counts and write offsets are per (query, slice), query-major
    const uint32_t slices = tune_get(ctx, "knn_slices", 1) == 1 ? search_slices(n, m, 16) : 1u;
    const size_t ms = m * (size_t)slices;
    const size_t bytes_q = 3 * m_cap * 8, bytes_c = (ms + 1) * 8;
    const size_t off_c = bytes_q;
    int rc = ensure_scratch(ctx, off_c + bytes_c + 64);
    if (rc) return rc;
    char* s = (char*)ctx->scratch;
    if ((rc = upload_soa_f64(ctx, q, m, m_cap, (double*)s))) return rc;
    // pass 1: counts -> exclusive scan on the host
    if ((rc = launch_radius_count(ctx, db->dev, n, n_cap, (double*)s, m, r2max, (unsigned long long*)(s + off_c), slices))) return rc;
    std::vector<unsigned long long> cnt(ms);
    PCR_HIP(ctx, hipMemcpyAsync(cnt.data(), s + off_c, ms * 8, hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<long long> offs(ms + 1);
    int64_t acc = 0;
    for (size_t i = 0; i < m; i++) {
        row_ptr[i] = acc;
        for (uint32_t sl = 0; sl < slices; sl++) { offs[i * slices + sl] = acc; acc += (int64_t)cnt[i * slices + sl]; }
    }
    offs[ms] = acc;
    row_ptr[m] = acc;
    if (!idx || acc == 0) return PCR_OK;
    // pass 2: fill (result buffers are separate allocations: growing the scratch would drop the uploaded queries)
    const size_t total = (size_t)acc;
    int32_t* idx_dev = nullptr;
    double* dist_dev = nullptr;
    PCR_HIP(ctx, hipMalloc((void**)&idx_dev, total * 4));
    hipError_t e = hipMalloc((void**)&dist_dev, total * 8);
    if (e != hipSuccess) { hipFree(idx_dev); return fail(ctx, PCR_ERR_HIP, "hipMalloc(radius)", e); }
    e = hipMemcpyAsync(s + off_c, offs.data(), (ms + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        rc = launch_radius_fill(ctx, db->dev, n, n_cap, (double*)s, m, r2max, (const long long*)(s + off_c), idx_dev, dist_dev, slices);
    if (e == hipSuccess && rc == PCR_OK) e = hipMemcpyAsync(idx, idx_dev, total * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && rc == PCR_OK) e = hipMemcpyAsync(dist, dist_dev, total * 8, hipMemcpyDeviceToHost, ctx->stream);
    hipError_t e2 = hipStreamSynchronize(ctx->stream);
    hipFree(idx_dev);
    hipFree(dist_dev);
    if (rc) return rc;
    if (e != hipSuccess) return fail(ctx, PCR_ERR_HIP, "radius fill", e);
    if (e2 != hipSuccess) return fail(ctx, PCR_ERR_HIP, "radius fill sync", e2);
    return PCR_OK;
}

extern "C" int pcr_knn_f64(pcr_ctx* ctx, const double* db, size_t n, const double* q, size_t m, int k, int32_t* idx,
                           double* dist)
{
    if (!ctx || k < 1 || k > 32 || (n && !db) || (m && (!q || !idx || !dist))) return fail(ctx, PCR_ERR_ARG, "pcr_knn_f64");
    pcr_db64* h = nullptr;
    int rc = pcr_db64_create(ctx, db, n, &h);
    if (rc) return rc;
    rc = pcr_db64_knn(ctx, h, q, m, k, 0, idx, dist);
    pcr_db64_destroy(ctx, h);
    return rc;
}

extern "C" int pcr_radius_f64(pcr_ctx* ctx, const double* db, size_t n, const double* q, size_t m, double r,
                              int64_t* row_ptr, int32_t* idx, double* dist)
{
    if (!ctx || !row_ptr || (n && !db) || (m && !q) || ((idx == nullptr) != (dist == nullptr)))
        return fail(ctx, PCR_ERR_ARG, "pcr_radius_f64");
    pcr_db64* h = nullptr;
    int rc = pcr_db64_create(ctx, db, n, &h);
    if (rc) return rc;
    rc = pcr_db64_radius(ctx, h, q, m, r, row_ptr, idx, dist);
    pcr_db64_destroy(ctx, h);
    return rc;
}

// ================================================================================================ device-resident radius rows (round 3)
// The rows of a radius search are 12 bytes per reported neighbour: 2.84 GB for every point of a 120 000-point scan at r = 1 — 1.9 ms of
// kernels and 170 ms of PCIe at the pcr_db64_radius boundary.  A consumer that REDUCES the rows (neighbour counts, ISS-style weights,
// neighbourhood moments for normals: Homework7/hw7/src/iss_detector.cpp:48-76, Homework1 pca_normal.py:89-103) or walks them a block at
// a time (the self-query protocol of Homework2/hw2/include/benchmark.hpp:66-70) does not need them on the host at once: pcr_rows keeps
// the CSR in HBM, only the m + 1 row offsets cross PCIe with the search.
struct pcr_rows {
    size_t m = 0;
    uint64_t total = 0;
    uint32_t* rows_dev = nullptr;       // m + 1 offsets (the grid route caps a search below 2^31 neighbours)
    int32_t* idx_dev = nullptr;         // ascending index inside a row (the canonical order of pcr_radius_f64)
    double* dist_dev = nullptr;
    const pcr_db64* db = nullptr;       // the database the indices refer to (must outlive the rows)
    std::vector<int64_t> row_ptr;       // host copy of the offsets
};

namespace pcr {
namespace {

constexpr int RR_BLOCK = 256;

// one wave per row: out[row] = count | sum of distances | largest distance (0 for an empty row).  The sum adds the lanes' partial sums
// (lane l holds entries l, l + 64, ... in row order) with a shuffle tree: deterministic, but not the left-to-right sum of a host loop.
__global__ __launch_bounds__(RR_BLOCK) void rows_reduce_kernel(const uint32_t* __restrict__ rows, const double* __restrict__ dist, uint32_t m, int op,
                                                               double* __restrict__ out)
{
    const uint32_t row = blockIdx.x * (RR_BLOCK / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= m) return;
    const uint32_t b = rows[row], e = rows[row + 1];
    double acc = 0.0;
    if (op == PCR_ROWS_SUM_DIST) {
        for (uint32_t p = b + lane; p < e; p += 64) acc += dist[p];
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    } else if (op == PCR_ROWS_MAX_DIST) {
        for (uint32_t p = b + lane; p < e; p += 64) acc = fmax(acc, dist[p]);
        for (int o = 32; o > 0; o >>= 1) acc = fmax(acc, __shfl_xor(acc, o, 64));
    } else {
        acc = (double)(e - b);
    }
    if (lane == 0) out[row] = acc;
}

// one wave per row: mean[3] and the six distinct entries of the scatter matrix sum (p - mean)(p - mean)^T / count of the row's
// neighbours, f64, two passes over the row (mean first: the centred products do not cancel); xx xy xz yy yz zz
__global__ __launch_bounds__(RR_BLOCK) void rows_moments_kernel(const uint32_t* __restrict__ rows, const int32_t* __restrict__ idx, uint32_t m,
                                                                const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ z,
                                                                double* __restrict__ mean, double* __restrict__ cov)
{
    const uint32_t row = blockIdx.x * (RR_BLOCK / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= m) return;
    const uint32_t b = rows[row], e = rows[row + 1];
    double s[3] = { 0.0, 0.0, 0.0 };
    for (uint32_t p = b + lane; p < e; p += 64) { const int32_t j = idx[p]; s[0] += x[j]; s[1] += y[j]; s[2] += z[j]; }
#pragma unroll
    for (int c = 0; c < 3; c++)
        for (int o = 32; o > 0; o >>= 1) s[c] += __shfl_xor(s[c], o, 64);
    const double cnt = (double)(e - b), inv = e > b ? 1.0 / cnt : 0.0;
    const double mx = s[0] * inv, my = s[1] * inv, mz = s[2] * inv;
    double c6[6] = { 0.0, 0.0, 0.0, 0.0, 0.0, 0.0 };
    for (uint32_t p = b + lane; p < e; p += 64) {
        const int32_t j = idx[p];
        const double dx = x[j] - mx, dy = y[j] - my, dz = z[j] - mz;
        c6[0] += dx * dx; c6[1] += dx * dy; c6[2] += dx * dz; c6[3] += dy * dy; c6[4] += dy * dz; c6[5] += dz * dz;
    }
#pragma unroll
    for (int c = 0; c < 6; c++)
        for (int o = 32; o > 0; o >>= 1) c6[c] += __shfl_xor(c6[c], o, 64);
    if (lane == 0) {
        mean[3 * (size_t)row] = mx; mean[3 * (size_t)row + 1] = my; mean[3 * (size_t)row + 2] = mz;
        for (int c = 0; c < 6; c++) cov[6 * (size_t)row + c] = c6[c] * inv;
    }
}

}  // namespace
}  // namespace pcr

extern "C" int pcr_rows_destroy(pcr_ctx* ctx, pcr_rows* rows)
{
    if (!rows) return PCR_OK;
    if (ctx) hipStreamSynchronize(ctx->stream);
    if (rows->rows_dev) hipFree(rows->rows_dev);
    if (rows->idx_dev) hipFree(rows->idx_dev);
    if (rows->dist_dev) hipFree(rows->dist_dev);
    delete rows;
    return PCR_OK;
}

extern "C" int pcr_db64_radius_rows(pcr_ctx* ctx, const pcr_db64* db, const double* q, size_t m, double r, pcr_rows** out)
{
    if (!ctx || !db || !out) return fail(ctx, PCR_ERR_ARG, "pcr_db64_radius_rows");
    *out = nullptr;
    const bool self = q == nullptr;                 // every point of the database queries the database
    if (self) m = db->n;
    if (m > 0x7FFFFFF0ull) return fail(ctx, PCR_ERR_ARG, "pcr_db64_radius_rows: too many queries");
    PCR_HIP(ctx, hipSetDevice(ctx->device));
    pcr_rows* h = new (std::nothrow) pcr_rows();
    if (!h) return fail(ctx, PCR_ERR_NOMEM, "pcr_db64_radius_rows");
    h->m = m; h->db = db;
    h->row_ptr.assign(m + 1, 0);
    if (m == 0) { *out = h; return PCR_OK; }
    int rc = PCR_OK;
    bool used = false;
    // the grid route leaves its rows where they are (radius_grid.hip, keep)
    if (db->twin && tune_get(ctx, "radius_method", 0) != 1) {
        RadiusRowsDev keep;
        const double r2max = radius_sq_bound(r);
        if (self) {
            rc = radius_grid(ctx, db->twin, db->twin, r, r2max, h->row_ptr.data(), nullptr, nullptr, &used, &keep);
        } else {
            std::vector<float> qf;
            if (f32_exact(q, m, qf)) {
                pcr_cloud* qc = nullptr;
                rc = pcr_cloud_create(ctx, qf.data(), m, PCR_AOS3, &qc);
                if (rc == PCR_OK) { rc = radius_grid(ctx, db->twin, qc, r, r2max, h->row_ptr.data(), nullptr, nullptr, &used, &keep); pcr_cloud_destroy(ctx, qc); }
            }
        }
        prof_flush(ctx);
        if (rc) { delete h; return rc; }
        if (used) { h->rows_dev = keep.rows_dev; h->idx_dev = keep.idx_dev; h->dist_dev = keep.dist_dev; h->total = (uint64_t)h->row_ptr[m]; }
    }
    if (!used) {
        // any other database / query set: the host-boundary search, its result uploaded once (the exhaustive route has no 2^31 cap,
        // the handle has: its offsets are 32-bit)
        std::vector<double> qself;
        if (self) {
            qself.resize(3 * m);
            std::vector<double> soa(3 * db->cap);
            hipError_t e = hipMemcpy(soa.data(), db->dev, 3 * db->cap * sizeof(double), hipMemcpyDeviceToHost);
            if (e != hipSuccess) { delete h; return fail(ctx, PCR_ERR_HIP, "rows: database download", e); }
            for (size_t i = 0; i < m; i++) { qself[3 * i] = soa[i]; qself[3 * i + 1] = soa[db->cap + i]; qself[3 * i + 2] = soa[2 * db->cap + i]; }
            q = qself.data();
        }
        rc = pcr_db64_radius(ctx, db, q, m, r, h->row_ptr.data(), nullptr, nullptr);
        if (rc) { delete h; return rc; }
        const uint64_t total = (uint64_t)h->row_ptr[m];
        if (total >= 0x7FFFFFF0ull) { delete h; return fail(ctx, PCR_ERR_ARG, "pcr_db64_radius_rows: 2^31 neighbours or more"); }
        h->total = total;
        if (total) {
            std::vector<int32_t> idx(total);
            std::vector<double> dist(total);
            rc = pcr_db64_radius(ctx, db, q, m, r, h->row_ptr.data(), idx.data(), dist.data());
            std::vector<uint32_t> r32(m + 1);
            for (size_t i = 0; i <= m; i++) r32[i] = (uint32_t)h->row_ptr[i];
            hipError_t e = rc ? hipSuccess : hipMalloc((void**)&h->rows_dev, (m + 1) * 4);
            if (!rc && e == hipSuccess) e = hipMalloc((void**)&h->idx_dev, total * 4);
            if (!rc && e == hipSuccess) e = hipMalloc((void**)&h->dist_dev, total * 8);
            if (!rc && e == hipSuccess) e = hipMemcpy(h->rows_dev, r32.data(), (m + 1) * 4, hipMemcpyHostToDevice);
            if (!rc && e == hipSuccess) e = hipMemcpy(h->idx_dev, idx.data(), total * 4, hipMemcpyHostToDevice);
            if (!rc && e == hipSuccess) e = hipMemcpy(h->dist_dev, dist.data(), total * 8, hipMemcpyHostToDevice);
            if (rc || e != hipSuccess) { pcr_rows_destroy(ctx, h); return rc ? rc : fail(ctx, PCR_ERR_HIP, "rows: upload", e); }
        }
    }
    *out = h;
    return PCR_OK;
}

extern "C" int pcr_rows_info(const pcr_rows* rows, size_t* m, uint64_t* total)
{
    if (!rows) return PCR_ERR_ARG;
    if (m) *m = rows->m;
    if (total) *total = rows->total;
    return PCR_OK;
}

extern "C" int pcr_rows_row_ptr(const pcr_rows* rows, int64_t* row_ptr)
{
    if (!rows || !row_ptr) return PCR_ERR_ARG;
    memcpy(row_ptr, rows->row_ptr.data(), (rows->m + 1) * sizeof(int64_t));
    return PCR_OK;
}

extern "C" int pcr_rows_fetch(pcr_ctx* ctx, const pcr_rows* rows, size_t row_begin, size_t row_end, int32_t* idx, double* dist)
{
    if (!ctx || !rows || row_begin > row_end || row_end > rows->m) return fail(ctx, PCR_ERR_ARG, "pcr_rows_fetch");
    const size_t b = (size_t)rows->row_ptr[row_begin], e = (size_t)rows->row_ptr[row_end];
    if (e == b) return PCR_OK;
    if (!idx && !dist) return fail(ctx, PCR_ERR_ARG, "pcr_rows_fetch: no output array");
    if (idx) PCR_HIP(ctx, hipMemcpyAsync(idx, rows->idx_dev + b, (e - b) * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (dist) PCR_HIP(ctx, hipMemcpyAsync(dist, rows->dist_dev + b, (e - b) * 8, hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PCR_OK;
}

extern "C" int pcr_rows_reduce(pcr_ctx* ctx, const pcr_rows* rows, int op, double* out)
{
    if (!ctx || !rows || (rows->m && !out) || op < PCR_ROWS_COUNT || op > PCR_ROWS_MAX_DIST) return fail(ctx, PCR_ERR_ARG, "pcr_rows_reduce");
    const size_t m = rows->m;
    if (m == 0) return PCR_OK;
    if (op == PCR_ROWS_COUNT || rows->total == 0) {                        // (the offsets are on the host already)
        for (size_t i = 0; i < m; i++) out[i] = op == PCR_ROWS_COUNT ? (double)(rows->row_ptr[i + 1] - rows->row_ptr[i]) : 0.0;
        return PCR_OK;
    }
    int rc = ensure_scratch(ctx, m * sizeof(double));
    if (rc) return rc;
    {
        ProfScope p(ctx, "rows_reduce", 1);
        hipLaunchKernelGGL(rows_reduce_kernel, dim3((unsigned)((m + RR_BLOCK / 64 - 1) / (RR_BLOCK / 64))), dim3(RR_BLOCK), 0, ctx->stream, rows->rows_dev,
                           rows->dist_dev, (uint32_t)m, op, (double*)ctx->scratch);
    }
    PCR_HIP(ctx, hipGetLastError());
    PCR_HIP(ctx, hipMemcpyAsync(out, ctx->scratch, m * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PCR_OK;
}

extern "C" int pcr_rows_moments(pcr_ctx* ctx, const pcr_rows* rows, double* mean, double* cov)
{
    if (!ctx || !rows || !rows->db || (rows->m && (!mean || !cov))) return fail(ctx, PCR_ERR_ARG, "pcr_rows_moments");
    const size_t m = rows->m;
    if (m == 0) return PCR_OK;
    if (rows->total == 0) { memset(mean, 0, 3 * m * sizeof(double)); memset(cov, 0, 6 * m * sizeof(double)); return PCR_OK; }
    int rc = ensure_scratch(ctx, 9 * m * sizeof(double));
    if (rc) return rc;
    double* dmean = (double*)ctx->scratch;
    double* dcov = dmean + 3 * m;
    const pcr_db64* db = rows->db;
    {
        ProfScope p(ctx, "rows_moments", 1);
        hipLaunchKernelGGL(rows_moments_kernel, dim3((unsigned)((m + RR_BLOCK / 64 - 1) / (RR_BLOCK / 64))), dim3(RR_BLOCK), 0, ctx->stream, rows->rows_dev,
                           rows->idx_dev, (uint32_t)m, db->dev, db->dev + db->cap, db->dev + 2 * db->cap, dmean, dcov);
    }
    PCR_HIP(ctx, hipGetLastError());
    PCR_HIP(ctx, hipMemcpyAsync(mean, dmean, 3 * m * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP(ctx, hipMemcpyAsync(cov, dcov, 6 * m * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PCR_OK;
}

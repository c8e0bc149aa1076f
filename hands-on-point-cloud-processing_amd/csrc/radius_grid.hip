// radius_grid.hip — A11 at scale: radius search with the hw2 contract (d = sqrt(((dx^2) + dy^2) + dz^2) in f64, member iff
// d <= r, CSR rows in ascending index order; kdtree.hpp:367-402, resultSet.hpp:96-142) over the uniform grid instead of the
// exhaustive n x m scan.  Used by pcr_db64_radius when database and queries are f32-representable (KITTI / PLY floats
// widened to f64, test.hpp:28 — then the f32 twin cloud widened back gives the very same doubles).
//
//   cell edge = 1.01 r  ->  every neighbour lies in the 27-cell block = 9 x-sorted record ranges, each cut to |x - qx| <= r
//   pass 1  count    32 lanes per query (cell-sorted queries), membership s <= r2max (the sqrt hoisted, search_f64.hip)
//   scan             row_ptr
//   pass 2  emit     one workgroup per query: a presence bitmap over the database indices in LDS puts the members in ascending
//                    index order (the canonical order) without a sort; index + d = sqrt(s) leave with coalesced stores
//   (databases beyond 262 144 points or rows beyond 16 384 neighbours: fill unsorted + rocPRIM segmented radix sort + distances)
#include "grid_common.hpp"

#include "sort.hpp"

#include <cmath>
#include <vector>

#pragma clang fp contract(off)

namespace pcr {

namespace {

constexpr int RG_BLOCK = 256;
constexpr int RG_G = 32;           // lanes per query

__device__ __forceinline__ double s_f64(float tx, float ty, float tz, double qx, double qy, double qz)
{
    const double e0 = (double)tx - qx, e1 = (double)ty - qy, e2 = (double)tz - qz;   // t - q, kdtree.hpp:343
    return (e0 * e0 + e1 * e1) + e2 * e2;
}

// Membership d^2 <= r2max (f64, kdtree.hpp:343-346) decided in f32 wherever f32 can: the f32 value of (dx^2 + dy^2) + dz^2 is within
// 5 * 2^-24 of the true one (the inputs are exact f32, every term is non-negative), so outside a band of 2^-20 around r2max the f32
// comparison IS the f64 one; inside the band the f64 arithmetic decides.  flo / fhi = r2max (1 -+ 2^-20) as floats (0 / inf when
// r2max is outside the float range: everything goes to f64).
__device__ __forceinline__ bool member(float tx, float ty, float tz, float qfx, float qfy, float qfz, double qx, double qy, double qz, double r2max,
                                       float flo, float fhi)
{
    const float dx = tx - qfx, dy = ty - qfy, dz = tz - qfz;
    const float d2 = (dx * dx + dy * dy) + dz * dz;
    if (d2 < flo) return true;
    if (d2 > fhi) return false;
    return s_f64(tx, ty, tz, qx, qy, qz) <= r2max;          // the band, NaN, overflow
}

__device__ __forceinline__ void member_band(double r2max, float& flo, float& fhi)
{
    flo = 0.0f; fhi = __builtin_inff();
    if (r2max > 1e-30 && r2max < 1e30) { flo = (float)(r2max * (1.0 - 9.5367431640625e-07)); fhi = (float)(r2max * (1.0 + 9.5367431640625e-07)); }
}

__device__ __forceinline__ void clip_x(const float4* __restrict__ records, uint32_t& b, uint32_t& e, float lo, float hi)
{
    if (e - b <= 256u) return;
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

// the k-th of the 9 rows around the query's (unclamped) cell, cut to the x window; empty when outside the grid
__device__ __forceinline__ void row_k(const GridParams& g, const uint32_t* __restrict__ cell_start, const float4* __restrict__ records, int cx, int cy, int cz,
                                      int k, float lo, float hi, uint32_t& b, uint32_t& e)
{
    const int yy = cy + (k % 3) - 1, zz = cz + (k / 3) - 1;
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.n[0] - 1);
    if (yy < 0 || yy >= g.n[1] || zz < 0 || zz >= g.n[2] || x0 > x1) { b = e = 0; return; }
    const uint32_t row = (uint32_t)((zz * g.n[1] + yy) * g.n[0]);
    b = cell_start[row + x0];
    e = cell_start[row + x1 + 1];
    clip_x(records, b, e, lo, hi);
}

// FILL = false: counts[q] = |N(q)|;  FILL = true: idx_out[row_ptr[q] ...] = members in grid order
template <bool FILL>
__global__ __launch_bounds__(RG_BLOCK) void radius_grid_kernel(const float4* __restrict__ records, const uint32_t* __restrict__ cell_start, GridParams g,
                                                               const float* __restrict__ qxs, const float* __restrict__ qys, const float* __restrict__ qzs,
                                                               const uint32_t* __restrict__ perm, uint32_t m, double r2max, float win,
                                                               uint32_t* __restrict__ counts, const uint32_t* __restrict__ row_ptr, int32_t* __restrict__ idx_out,
                                                               uint32_t* __restrict__ bounds)
{
    const uint32_t tq = (blockIdx.x * RG_BLOCK + threadIdx.x) / RG_G;
    const int sub = threadIdx.x % RG_G;
    if (tq >= m) return;                                   // whole groups leave together
    const uint32_t qi = perm ? perm[tq] : tq;
    const float fx = qxs[qi], fy = qys[qi], fz = qzs[qi];
    uint32_t c = 0;
    uint32_t w = FILL ? row_ptr[qi] : 0u;
    if (finite3(fx, fy, fz)) {
        const double qx = fx, qy = fy, qz = fz;
        const int cx = cell_coord(fx, g.lo[0], g.inv_h), cy = cell_coord(fy, g.lo[1], g.inv_h), cz = cell_coord(fz, g.lo[2], g.inv_h);
        const float pad = (fabsf(fx) + win) * 2.4e-7f;
        const float lo = fx - win - pad, hi = fx + win + pad;
        float flo, fhi;
        member_band(r2max, flo, fhi);
        for (int k = 0; k < 9; k++) {
            uint32_t b, e;
            row_k(g, cell_start, records, cx, cy, cz, k, lo, hi, b, e);
            if (!FILL && bounds && sub == k) { bounds[(size_t)qi * 18 + 2 * k] = b; bounds[(size_t)qi * 18 + 2 * k + 1] = e; }   // for the emit pass
            for (uint32_t j0 = b; j0 < e; j0 += RG_G) {    // uniform trip count over the group (ballot below)
                const uint32_t j = j0 + sub;
                bool in = false;
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if (j < e) {
                    t = records[j];
                    in = member(t.x, t.y, t.z, fx, fy, fz, qx, qy, qz, r2max, flo, fhi);
                }
                if (FILL) {
                    const unsigned long long all = __ballot(in);
                    const uint32_t mine = (uint32_t)(all >> ((threadIdx.x & 63) / RG_G * RG_G));      // this group's 32 lanes
                    if (in) idx_out[w + __popc(mine & ((1u << sub) - 1u))] = (int32_t)__float_as_uint(t.w);
                    w += __popc(mine);
                } else {
                    c += in;
                }
            }
        }
    }
    if (!FILL) {
#pragma unroll
        for (int o = RG_G / 2; o > 0; o >>= 1) c += __shfl_xor(c, o, RG_G);
        if (sub == 0) counts[qi] = c;
    }
}

// one group per query: distances of its (now index-sorted) neighbours
__global__ __launch_bounds__(RG_BLOCK) void radius_dist_kernel(const float* __restrict__ tx, const float* __restrict__ ty, const float* __restrict__ tz,
                                                               const float* __restrict__ qxs, const float* __restrict__ qys, const float* __restrict__ qzs, uint32_t m,
                                                               const uint32_t* __restrict__ row_ptr, const int32_t* __restrict__ idx, double* __restrict__ dist)
{
    const uint32_t qi = (blockIdx.x * RG_BLOCK + threadIdx.x) / RG_G;
    const int sub = threadIdx.x % RG_G;
    if (qi >= m) return;
    const double qx = qxs[qi], qy = qys[qi], qz = qzs[qi];
    for (uint32_t p = row_ptr[qi] + sub; p < row_ptr[qi + 1]; p += RG_G) {
        const int32_t j = idx[p];
        dist[p] = sqrt(s_f64(tx[j], ty[j], tz[j], qx, qy, qz));
    }
}

// ---- fused pass 2: one workgroup per query ---------------------------------------------------------------------------------
// The members of a row have to come out in ascending index order.  Every index occurs at most once, so the workgroup keeps a
// PRESENCE BITMAP over the database indices in LDS (n / 8 bytes: 15 KB at 120 k points): the walk of the count pass sets one
// bit per member (atomicOr), a popcount prefix over the bitmap words gives every member its rank, the ranked indices are staged
// in LDS and leave with coalesced stores together with their distances.  No sort at all.  Replaces fill (4 B/neighbour
// written, unsorted) + device-wide segmented sort (2 x 8 B/neighbour/pass) + distance pass (4 B/neighbour re-read): HBM sees
// the 12 B/neighbour of the result and the gathers of the coordinates, nothing else.
//   LDS: bitmap[W] | staged indices A[CAP] | 32 words of scan partials / row ranges   (W = ceil(n / 32) rounded up to 256 words)
constexpr int RE_BLOCK = 256;
constexpr int RE_UNROLL = 4;       // candidate loads in flight per thread (8 measured: no change)
constexpr uint32_t RE_MAX_DB = 262144;       // 32 KB of bitmap; larger databases take the segmented-sort route

template <int CAP>
__global__ __launch_bounds__(RE_BLOCK) void radius_emit_kernel(const float4* __restrict__ records, const uint32_t* __restrict__ cell_start, GridParams g,
                                                               const float* __restrict__ qxs, const float* __restrict__ qys, const float* __restrict__ qzs,
                                                               const uint32_t* __restrict__ list, uint32_t n_list, double r2max, float win,
                                                               const uint32_t* __restrict__ row_ptr, const float4* __restrict__ by_index,
                                                               uint32_t words, int32_t* __restrict__ idx_out,
                                                               double* __restrict__ dist_out, int* __restrict__ err, const uint32_t* __restrict__ bounds)
{
    extern __shared__ uint32_t re_lds[];
    uint32_t* bm = re_lds;                       // [words]
    uint32_t* A = re_lds + words;                // [CAP]
    uint32_t* misc = A + CAP;                    // [0..3] scan partials, [8..25] the nine row ranges
    if (blockIdx.x >= n_list) return;
    const uint32_t qi = list[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (uint32_t i = tid; i < words; i += RE_BLOCK) bm[i] = 0u;
    __syncthreads();
    const float fx = qxs[qi], fy = qys[qi], fz = qzs[qi];
    const uint32_t o = row_ptr[qi];
    const uint32_t expect = row_ptr[qi + 1] - o;
    // ---- collect: the walk and the membership test of the count pass
    if (finite3(fx, fy, fz)) {
        const double qx = fx, qy = fy, qz = fz;
        float flo, fhi;
        member_band(r2max, flo, fhi);
        // the nine row ranges, already cut to the x window, as the count pass resolved them (18 words per query: recomputing them here
        // meant cell_start look-ups plus a chain of ~16 dependent loads per clipped row, in front of the whole workgroup)
        if (tid < 18) misc[8 + tid] = bounds[(size_t)qi * 18 + tid];
        __syncthreads();
        for (int k = 0; k < 9; k++) {
            const uint32_t b = misc[8 + 2 * k], e = misc[9 + 2 * k];
            uint32_t j = b + tid;
            for (; j + (RE_UNROLL - 1) * RE_BLOCK < e; j += RE_UNROLL * RE_BLOCK) {            // independent 16-byte loads in flight per thread
                float4 t[RE_UNROLL];
#pragma unroll
                for (int u = 0; u < RE_UNROLL; u++) t[u] = records[j + u * RE_BLOCK];
#pragma unroll
                for (int u = 0; u < RE_UNROLL; u++)
                    if (member(t[u].x, t[u].y, t[u].z, fx, fy, fz, qx, qy, qz, r2max, flo, fhi)) {
                        const uint32_t id = __float_as_uint(t[u].w);
                        atomicOr(&bm[id >> 5], 1u << (id & 31u));
                    }
            }
            for (; j < e; j += RE_BLOCK) {
                const float4 t = records[j];
                if (member(t.x, t.y, t.z, fx, fy, fz, qx, qy, qz, r2max, flo, fhi)) {
                    const uint32_t id = __float_as_uint(t.w);
                    atomicOr(&bm[id >> 5], 1u << (id & 31u));
                }
            }
        }
    }
    __syncthreads();
    // ---- rank: thread t owns words [t * wpt, (t + 1) * wpt)
    const uint32_t wpt = words / RE_BLOCK;
    uint32_t mine = 0;
    for (uint32_t i = 0; i < wpt; i++) mine += (uint32_t)__popc(bm[tid * wpt + i]);
    uint32_t inc = mine;
#pragma unroll
    for (int s2 = 1; s2 < 64; s2 <<= 1) {
        const uint32_t t = __shfl_up(inc, s2, 64);
        if (lane >= s2) inc += t;
    }
    if (lane == 63) misc[w] = inc;
    __syncthreads();
    uint32_t pos = inc - mine;
    for (int ww = 0; ww < w; ww++) pos += misc[ww];
    const uint32_t N = misc[0] + misc[1] + misc[2] + misc[3];
    if (N != expect || N > CAP) { if (tid == 0) atomicExch(err, 1); return; }      // cannot happen (same walk as the count pass); never write out of bounds
    // Stage the members in index order.  Neighbours of a scan point have CLUSTERED indices (runs along a beam ring), so the bitmap
    // words are empty or nearly full: a thread walking the bits of its own words made the wave wait for the fullest word of every
    // step (1.28 of the pass's 2.4 ms).  Instead the wave takes its non-empty words two at a time, 32 lanes on the 32 bits of one:
    // lane b of a half writes member (word, b) to the word's start position + popcount of the bits below b.
    for (uint32_t i = 0; i < wpt; i++) {
        const uint32_t wv = bm[tid * wpt + i];
        const uint32_t wp = pos;
        pos += (uint32_t)__popc(wv);
        unsigned long long nz = __ballot(wv != 0u);
        while (nz) {                                         // wave-uniform
            const int l0 = __builtin_ctzll(nz);
            nz &= nz - 1ull;
            int l1 = l0;
            const bool two = nz != 0ull;
            if (two) { l1 = __builtin_ctzll(nz); nz &= nz - 1ull; }
            const int srcl = lane < 32 ? l0 : l1;
            const uint32_t word = (uint32_t)__shfl((int)wv, srcl, 64), p0 = (uint32_t)__shfl((int)wp, srcl, 64);
            const uint32_t bit = (uint32_t)lane & 31u;
            if ((lane < 32 || two) && ((word >> bit) & 1u))
                A[p0 + (uint32_t)__popc(word & ((1u << bit) - 1u))] = ((uint32_t)(w * 64 + srcl) * wpt + i) * 32u + bit;
        }
    }
    __syncthreads();
    // ---- emit: ascending index, distance from the original coordinates (kdtree.hpp:341-346), coalesced stores
    const double qx = fx, qy = fy, qz = fz;
    // four independent (LDS read -> three gathers -> sqrt -> two stores) chains in flight per thread
    uint32_t p = tid;
    for (; p + 3 * RE_BLOCK < N; p += 4 * RE_BLOCK) {
        uint32_t j[4];
        float4 c[4];
#pragma unroll
        for (int u = 0; u < 4; u++) j[u] = A[p + u * RE_BLOCK];
#pragma unroll
        for (int u = 0; u < 4; u++) c[u] = by_index[j[u]];         // one 16-byte gather per neighbour (three 4-byte ones cost the texture path 3 x 64 lane-cycles)
#pragma unroll
        for (int u = 0; u < 4; u++) {
            idx_out[o + p + u * RE_BLOCK] = (int32_t)j[u];
            dist_out[o + p + u * RE_BLOCK] = sqrt(s_f64(c[u].x, c[u].y, c[u].z, qx, qy, qz));
        }
    }
    for (; p < N; p += RE_BLOCK) {
        const uint32_t j = A[p];
        const float4 c = by_index[j];
        idx_out[o + p] = (int32_t)j;
        dist_out[o + p] = sqrt(s_f64(c.x, c.y, c.z, qx, qy, qz));
    }
}

__global__ __launch_bounds__(RG_BLOCK) void aos4_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, uint32_t n,
                                                        float4* __restrict__ out)
{
    const uint32_t i = blockIdx.x * RG_BLOCK + threadIdx.x;
    if (i < n) out[i] = make_float4(x[i], y[i], z[i], 0.f);
}

// self-query: the records' order IS the cell order of the points -> perm[t] = original index of record t
__global__ __launch_bounds__(RG_BLOCK) void record_index_kernel(const float4* __restrict__ rec, uint32_t n, uint32_t* __restrict__ perm)
{
    const uint32_t i = blockIdx.x * RG_BLOCK + threadIdx.x;
    if (i < n) perm[i] = __float_as_uint(rec[i].w);
}

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
};

}  // namespace

// *used = false: the caller falls back to the exhaustive kernels (radius too large for a useful grid, 2^31 neighbours, ...)
// keep != nullptr: the rows stay in HBM — the fill runs whether or not host arrays were given, nothing but the counts crosses PCIe,
// and the caller owns keep->rows_dev (m + 1 u32 offsets), idx_dev, dist_dev afterwards (hipFree; all null when there is no neighbour)
int radius_grid(pcr_ctx* ctx, const pcr_cloud* db, const pcr_cloud* q, double r, double r2max, int64_t* row_ptr_host, int32_t* idx_host, double* dist_host,
                bool* used, RadiusRowsDev* keep)
{
    *used = false;
    if (keep) *keep = RadiusRowsDev();
    const size_t n = db->n, m = q->n;
    if (n == 0 || m == 0 || !(r > 0.0) || !std::isfinite(r) || r2max < 0.0) return PCR_OK;
    // the index with cell edge 1.01 r is kept on the database cloud (one slot: the radius of the last search), so that a driver
    // asking one query at a time with the same radius — KDTreeRadiusNNSearch per point — builds it once
    pcr_cloud* mdb = const_cast<pcr_cloud*>(db);
    if (!(mdb->rad_grid && mdb->rad_grid_r == r)) {
        if (mdb->rad_grid) {
            PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
            grid_free(mdb->rad_grid); mdb->rad_grid = nullptr; mdb->rad_grid_r = 0.0;
        }
        ProfScope ps(ctx, "radius_grid_build");
        Grid* built = nullptr;
        int rc = grid_build(ctx, db, &built, std::max(r * 1.01, 2e-15));
        if (rc) return rc;
        mdb->rad_grid = built;
        mdb->rad_grid_r = r;
    }
    Grid* g = mdb->rad_grid;
    // the cell budget may have enlarged the cells: fine.  A cell much SMALLER than asked cannot happen; a grid of a few
    // cells only (radius ~ extent) is the exhaustive scan in disguise: leave that to the tiled kernels
    if ((double)g->p.h < r * 1.005) return fail(ctx, PCR_ERR_STATE, "radius_grid: cell smaller than the radius");
    if ((size_t)g->p.n[0] * g->p.n[1] * g->p.n[2] < 64) return PCR_OK;
    // cell-sorted query order
    const uint32_t* perm = nullptr;
    DevBuf permbuf;
    if (q == db) {
        PCR_HIP(ctx, hipMalloc(&permbuf.p, std::max<size_t>(m, 1) * 4));
        hipLaunchKernelGGL(record_index_kernel, dim3((unsigned)((m + RG_BLOCK - 1) / RG_BLOCK)), dim3(RG_BLOCK), 0, ctx->stream, g->records, (uint32_t)m,
                           (uint32_t*)permbuf.p);
        perm = (const uint32_t*)permbuf.p;
    } else if (m > KNN_SMALL_MAX) {
        ProfScope ps(ctx, "grid_sort_queries");
        int rc = grid_prepare_queries(ctx, db, q);           // coarse cells of db's cached 1-NN grid: any spatial grouping will do
        if (rc) return rc;
        perm = ctx->qperm;
    }                                                         // (a small batch is searched in the order given: perm stays null)
    const float win = (float)(r * 1.00001) + 1e-30f;
    DevBuf cnt, rows, bnd;
    PCR_HIP(ctx, hipMalloc(&bnd.p, std::max<size_t>(m, 1) * 18 * 4));
    PCR_HIP(ctx, hipMalloc(&cnt.p, (m + 1) * 4));
    PCR_HIP(ctx, hipMalloc(&rows.p, (m + 2) * 4));
    const dim3 grid((unsigned)((m * RG_G + RG_BLOCK - 1) / RG_BLOCK));
    {
        ProfScope ps(ctx, "radius_count", 1);
        hipLaunchKernelGGL(radius_grid_kernel<false>, grid, dim3(RG_BLOCK), 0, ctx->stream, g->records, g->cell_start, g->p, q->x(), q->y(), q->z(), perm, (uint32_t)m,
                           r2max, win, (uint32_t*)cnt.p, (const uint32_t*)nullptr, (int32_t*)nullptr, (uint32_t*)bnd.p);
    }
    PCR_HIP(ctx, hipGetLastError());
    std::vector<uint32_t> hc(m);
    PCR_HIP(ctx, hipMemcpyAsync(hc.data(), cnt.p, m * 4, hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    uint64_t acc = 0;
    for (size_t i = 0; i < m; i++) { row_ptr_host[i] = (int64_t)acc; acc += hc[i]; }
    row_ptr_host[m] = (int64_t)acc;
    if (acc >= 0x7FFFFFF0ull) return PCR_OK;                   // 32-bit offsets / item counts: exhaustive path (which redoes the counts)
    *used = true;
    if ((!idx_host && !keep) || acc == 0) return PCR_OK;
    const size_t total = (size_t)acc;
    std::vector<uint32_t> hr(m + 1);
    for (size_t i = 0; i <= m; i++) hr[i] = (uint32_t)row_ptr_host[i];
    PCR_HIP(ctx, hipMemcpyAsync(rows.p, hr.data(), (m + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    DevBuf idx_a, idx_b, dist;
    PCR_HIP(ctx, hipMalloc(&idx_b.p, total * 4));
    PCR_HIP(ctx, hipMalloc(&dist.p, total * 8));
    uint32_t longest = 0;
    for (size_t i = 0; i < m; i++) longest = std::max(longest, hc[i]);
    int bits = 1;
    while (((size_t)1 << bits) < n) bits++;
    const bool fused = longest <= 16384u && n <= RE_MAX_DB && tune_get(ctx, "radius_fused", 1) == 1;
    const uint32_t words = (uint32_t)(((n + 31) / 32 + RE_BLOCK - 1) / RE_BLOCK * RE_BLOCK);
    if (fused) {
        // rows by length class (LDS capacity of the workgroup that sorts them), each class in the cell order of the queries
        std::vector<uint32_t> hperm;
        if (perm) {
            hperm.resize(m);
            PCR_HIP(ctx, hipMemcpyAsync(hperm.data(), perm, m * 4, hipMemcpyDeviceToHost, ctx->stream));
            PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
        std::vector<uint32_t> lists[5];
        for (size_t t = 0; t < m; t++) {
            const uint32_t qi = perm ? hperm[t] : (uint32_t)t;
            const uint32_t c = hc[qi];
            if (c == 0) continue;
            lists[c <= 1024u ? 0 : c <= 2048u ? 1 : c <= 4096u ? 2 : c <= 8192u ? 3 : 4].push_back(qi);
        }
        if (!g->by_index) {
            PCR_HIP(ctx, hipMalloc((void**)&g->by_index, std::max<size_t>(n, 1) * sizeof(float4)));
            hipLaunchKernelGGL(aos4_kernel, dim3((unsigned)((n + RG_BLOCK - 1) / RG_BLOCK)), dim3(RG_BLOCK), 0, ctx->stream, db->x(), db->y(), db->z(), (uint32_t)n,
                               g->by_index);
        }
        DevBuf lbuf, errbuf;
        PCR_HIP(ctx, hipMalloc(&lbuf.p, (m + 1) * 4));
        PCR_HIP(ctx, hipMalloc(&errbuf.p, 4));
        PCR_HIP(ctx, hipMemsetAsync(errbuf.p, 0, 4, ctx->stream));
        size_t off = 0;
        size_t loff[5];
        for (int c = 0; c < 5; c++) {
            loff[c] = off;
            if (!lists[c].empty()) PCR_HIP(ctx, hipMemcpyAsync((uint32_t*)lbuf.p + off, lists[c].data(), lists[c].size() * 4, hipMemcpyHostToDevice, ctx->stream));
            off += lists[c].size();
        }
        {
            ProfScope ps(ctx, "radius_emit", 1);
#define PCR_EMIT(CAPV, c)                                                                                                                     \
    if (!lists[c].empty()) {                                                                                                                  \
        const size_t lds = ((size_t)words + CAPV + 32) * 4;                                                                                    \
        hipFuncSetAttribute((const void*)radius_emit_kernel<CAPV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                      \
        hipLaunchKernelGGL((radius_emit_kernel<CAPV>), dim3((unsigned)lists[c].size()), dim3(RE_BLOCK), lds, ctx->stream, g->records, g->cell_start, g->p, \
                           q->x(), q->y(), q->z(), (const uint32_t*)lbuf.p + loff[c], (uint32_t)lists[c].size(), r2max, win, (const uint32_t*)rows.p,   \
                           g->by_index, words, (int32_t*)idx_b.p, (double*)dist.p, (int*)errbuf.p, (const uint32_t*)bnd.p);                                            \
    }
            PCR_EMIT(1024, 0)
            PCR_EMIT(2048, 1)
            PCR_EMIT(4096, 2)
            PCR_EMIT(8192, 3)
            PCR_EMIT(16384, 4)
#undef PCR_EMIT
        }
        PCR_HIP(ctx, hipGetLastError());
        int herr = 0;
        PCR_HIP(ctx, hipMemcpyAsync(&herr, errbuf.p, 4, hipMemcpyDeviceToHost, ctx->stream));
        if (idx_host) {
            PCR_HIP(ctx, hipMemcpyAsync(idx_host, idx_b.p, total * 4, hipMemcpyDeviceToHost, ctx->stream));
            PCR_HIP(ctx, hipMemcpyAsync(dist_host, dist.p, total * 8, hipMemcpyDeviceToHost, ctx->stream));
        }
        PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (herr) return fail(ctx, PCR_ERR_STATE, "radius_emit: a row changed between the count and the fill pass");
        if (keep) { keep->rows_dev = (uint32_t*)rows.p; keep->idx_dev = (int32_t*)idx_b.p; keep->dist_dev = (double*)dist.p; keep->total = total; rows.p = idx_b.p = dist.p = nullptr; }
        return PCR_OK;
    }
    // rows longer than the LDS of a workgroup: fill unsorted, device-wide segmented sort, distance pass
    PCR_HIP(ctx, hipMalloc(&idx_a.p, total * 4));
    {
        ProfScope ps(ctx, "radius_fill", 1);
        hipLaunchKernelGGL(radius_grid_kernel<true>, grid, dim3(RG_BLOCK), 0, ctx->stream, g->records, g->cell_start, g->p, q->x(), q->y(), q->z(), perm, (uint32_t)m,
                           r2max, win, (uint32_t*)nullptr, (const uint32_t*)rows.p, (int32_t*)idx_a.p, (uint32_t*)nullptr);
    }
    size_t temp_bytes = 0;
    PCR_HIP(ctx, segmented_sort_keys_u32(nullptr, temp_bytes, (const uint32_t*)idx_a.p, (uint32_t*)idx_b.p, total, m, (const uint32_t*)rows.p,
                                         (const uint32_t*)rows.p + 1, 0, bits, ctx->stream));
    DevBuf temp;
    PCR_HIP(ctx, hipMalloc(&temp.p, std::max<size_t>(temp_bytes, 16)));
    {
        ProfScope ps(ctx, "radius_sort", 1);
        PCR_HIP(ctx, segmented_sort_keys_u32(temp.p, temp_bytes, (const uint32_t*)idx_a.p, (uint32_t*)idx_b.p, total, m, (const uint32_t*)rows.p,
                                             (const uint32_t*)rows.p + 1, 0, bits, ctx->stream));
    }
    {
        ProfScope ps(ctx, "radius_dist", 1);
        hipLaunchKernelGGL(radius_dist_kernel, grid, dim3(RG_BLOCK), 0, ctx->stream, db->x(), db->y(), db->z(), q->x(), q->y(), q->z(), (uint32_t)m,
                           (const uint32_t*)rows.p, (const int32_t*)idx_b.p, (double*)dist.p);
    }
    PCR_HIP(ctx, hipGetLastError());
    if (idx_host) {
        PCR_HIP(ctx, hipMemcpyAsync(idx_host, idx_b.p, total * 4, hipMemcpyDeviceToHost, ctx->stream));
        PCR_HIP(ctx, hipMemcpyAsync(dist_host, dist.p, total * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (keep) { keep->rows_dev = (uint32_t*)rows.p; keep->idx_dev = (int32_t*)idx_b.p; keep->dist_dev = (double*)dist.p; keep->total = total; rows.p = idx_b.p = dist.p = nullptr; }
    return PCR_OK;
}

}  // namespace pcr

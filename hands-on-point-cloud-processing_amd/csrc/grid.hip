// grid.hip — EXACT grid-accelerated 1-NN for gfx950: the HBM/L2-bound variant of the correspondence search
// (SURVEY.md §7.1 step 9, §8d "1-NN exact grid").  Same contract as nn1_brute.hip: A1 arithmetic
// (nanoflann.hpp:403-406, unfused f32) and the canonical tie rule, so results are bit-identical to brute force
// and to the reference's kd-tree on tie-free data; only the set of candidates a query looks at is pruned.
//
// Index (built once per target cloud, cached on the cloud, dropped when the cloud is transformed):
//   * uniform grid over the bounding box of the finite targets, cell edge h ~ cbrt(V / 4n) (tunable),
//     cell id = (cz * ny + cy) * nx + cx; counting sort by cell id: histogram (atomicAdd) -> exclusive scan
//     -> scatter into float4 records {x, y, z, original index}: ONE 16-byte load per candidate;
//     cells that are adjacent in x are adjacent in memory, so a query walks whole x-rows of cells as one
//     contiguous range [cell_start[row + xlo], cell_start[row + xhi + 1]).
//   * the order of records inside a cell depends on atomic arrival order; results do not, because every
//     comparison is on the 64-bit key (d2_bits << 32 | original index).
// Query (one per lane; lanes are handed queries sorted by cell so that a wave stays spatially coherent):
//   scan the cube of cells within Chebyshev radius r of the query's cell, r = r0, 2 r0, 4 r0, ... and stop as
//   soon as the best d2 is strictly below LB(r) = ((r - slack) * h)^2 * (1 - 1e-5), a lower bound on the f32
//   distance of every target outside the cube (two points whose cells differ by >= r+1 in some axis are more
//   than (r - slack) * h apart; slack covers the f32 rounding of the cell computation), or when the cube
//   covers the whole grid.  Exactness proof and cost model: DESIGN.md §5b.
// Traffic (algorithmic): 12 B query + 8 B key + 16 B per visited candidate + 8 B per visited cell row.
#include "grid_common.hpp"

#include "sort.hpp"

#include <cmath>
#include <vector>

#pragma clang fp contract(off)

namespace pcr {

constexpr int GR_BLOCK = 256;

// ---- two host-read counters of the index builds, through pinned words behind an event each (pcr_internal.hpp pin_words)
static hipError_t pin_word_begin(pcr_ctx* ctx, int slot, const uint32_t* dev_word)
{
    hipError_t e = hipSuccess;
    if (!ctx->pin_words) {
        e = hipHostMalloc((void**)&ctx->pin_words, 64, hipHostMallocDefault);
        if (e != hipSuccess) return e;
        for (int k = 0; k < 2 && e == hipSuccess; k++) e = hipEventCreateWithFlags(&ctx->pin_ev[k], hipEventDisableTiming);
        if (e != hipSuccess) return e;
    }
    e = hipMemcpyAsync(&ctx->pin_words[slot * 8], dev_word, 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipEventRecord(ctx->pin_ev[slot], ctx->stream);
    ctx->pin_pending[slot] = e == hipSuccess;
    if (slot == 1) ctx->pin_gen++;
    return e;
}
static uint32_t work_cells_now(pcr_ctx* ctx)
{
    if (ctx->pin_pending[0]) {
        if (hipEventSynchronize(ctx->pin_ev[0]) == hipSuccess) ctx->work_cells = ctx->pin_words[0];
        ctx->pin_pending[0] = false;
    }
    return ctx->work_cells;
}
static uint32_t grid_occupied_now(pcr_ctx* ctx, Grid* g)
{
    if (g->occupied == 0 && g->occupied_tag != 0 && g->occupied_tag == ctx->pin_gen && ctx->pin_words) {      // (a later build took the word: unknown, the walk serves)
        if (ctx->pin_pending[1]) { (void)hipEventSynchronize(ctx->pin_ev[1]); ctx->pin_pending[1] = false; }
        g->occupied = ctx->pin_words[8];
    }
    return g->occupied;
}
constexpr int SC_ITEMS = 8;                       // scan: items per thread
constexpr int SC_TILE = SCAN_TILE;                // 2048 per block (pcr_internal.hpp)
static_assert(SC_TILE == GR_BLOCK * SC_ITEMS, "scan tile");

// ---------------------------------------------------------------------------------------- bounding box
__device__ __forceinline__ float wave_min_f(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_down(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_max_f(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, 64));
    return v;
}

// out[block][6] = {min x,y,z, max x,y,z} over the finite points of the block's range
__global__ __launch_bounds__(GR_BLOCK) void bbox_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                        const float* __restrict__ z, uint32_t n, float* __restrict__ out)
{
    float mn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, mx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (uint32_t i = blockIdx.x * GR_BLOCK + threadIdx.x; i < n; i += gridDim.x * GR_BLOCK) {
        const float px = x[i], py = y[i], pz = z[i];
        if (finite3(px, py, pz)) {
            mn[0] = fminf(mn[0], px); mn[1] = fminf(mn[1], py); mn[2] = fminf(mn[2], pz);
            mx[0] = fmaxf(mx[0], px); mx[1] = fmaxf(mx[1], py); mx[2] = fmaxf(mx[2], pz);
        }
    }
    __shared__ float red[GR_BLOCK / 64][6];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float a = wave_min_f(mn[c]), b = wave_max_f(mx[c]);
        if (lane == 0) { red[wave][c] = a; red[wave][3 + c] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int c = threadIdx.x;
        float v = red[0][c];
        for (int w = 1; w < GR_BLOCK / 64; w++) v = c < 3 ? fminf(v, red[w][c]) : fmaxf(v, red[w][c]);
        out[blockIdx.x * 6 + c] = v;
    }
}

// ---------------------------------------------------------------------------------------- counting sort
__global__ __launch_bounds__(GR_BLOCK) void cell_count_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                              const float* __restrict__ z, uint32_t n, GridParams g,
                                                              uint32_t* __restrict__ cell_of, uint32_t* __restrict__ count,
                                                              uint32_t nonfinite_cell)
{
    const uint32_t i = blockIdx.x * GR_BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = clamped_cell_id(g, x[i], y[i], z[i], nonfinite_cell);
    cell_of[i] = c;
    atomicAdd(&count[c], 1u);
}

// key = (cell, order-preserving bits of x): one radix sort gives the records cell by cell and x-ascending inside a cell —
// a deterministic layout (no atomics decide a position); x-sorted rows are what the CLIP variant of the search kernel needs (rows cut to the best-distance window).
__device__ __forceinline__ uint32_t spread3(uint32_t v)      // 3 bits -> bits 0, 3, 6
{
    return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4);
}

// 8 bits -> bits 0, 3, 6, ..., 21
__device__ __forceinline__ uint32_t spread8(uint32_t v)
{
    v &= 0xFFu;
    v = (v | (v << 8)) & 0x00F00Fu;
    v = (v | (v << 4)) & 0x0C30C3u;
    v = (v | (v << 2)) & 0x249249u;
    return v;
}

// 24-bit Morton code of a point's position INSIDE its cell (8 bits per axis: 1 / 256 of the cell edge); the top nine bits are the
// 8 x 8 x 8 sub-cell code of round 2.  frac = (coordinate - lo) * inv_h - cell, clamped (a query outside the grid box).
__device__ __forceinline__ uint32_t morton24_in_cell(float fx, float fy, float fz)
{
    const uint32_t sx = (uint32_t)fminf(fmaxf(fx * 256.0f, 0.0f), 255.0f), sy = (uint32_t)fminf(fmaxf(fy * 256.0f, 0.0f), 255.0f),
                   sz = (uint32_t)fminf(fmaxf(fz * 256.0f, 0.0f), 255.0f);
    return spread8(sx) | (spread8(sy) << 1) | (spread8(sz) << 2);
}

// MORTON: the low key word is the 24-bit Morton code of the point's position inside its cell (<< 8): consecutive records of a cell are
// neighbours in space down to 1 / 256 of the cell edge (round 2: an 8 x 8 x 8 sub-cell code, then x), which keeps the bounding spheres of
// 16-record runs small along any direction a scan line takes.  Still a pure function of the point (ties resolve by original index:
// the radix sort is stable).
template <bool MORTON>
__global__ __launch_bounds__(GR_BLOCK) void record_keys_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, uint32_t n,
                                                               GridParams g, const uint32_t* __restrict__ cell_of, unsigned long long* __restrict__ keys,
                                                               uint32_t* __restrict__ vals)
{
    const uint32_t i = blockIdx.x * GR_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float px = x[i];
    uint32_t u = __float_as_uint(px);
    u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;
    if (MORTON) {
        const float py = y[i], pz = z[i];
        uint32_t m = 0;
        if (finite3(px, py, pz)) {
            const float fx = (px - g.lo[0]) * g.inv_h, fy = (py - g.lo[1]) * g.inv_h, fz = (pz - g.lo[2]) * g.inv_h;
            m = morton24_in_cell(fx - floorf(fx), fy - floorf(fy), fz - floorf(fz));
        }
        u = m << 8;
    }
    keys[i] = ((unsigned long long)cell_of[i] << 32) | u;
    vals[i] = i;
}

// one thread per run of GRID_CHUNK records: bounding sphere of its finite members (centre = middle of their box, radius rounded up)
__global__ __launch_bounds__(GR_BLOCK) void build_spheres_kernel(const float4* __restrict__ records, uint32_t n_chunks, float4* __restrict__ spheres)
{
    const uint32_t c = blockIdx.x * GR_BLOCK + threadIdx.x;
    if (c >= n_chunks) return;
    float mn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, mx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (int j = 0; j < GRID_CHUNK; j++) {
        const float4 r = records[(size_t)c * GRID_CHUNK + j];
        if (finite3(r.x, r.y, r.z)) {
            mn[0] = fminf(mn[0], r.x); mn[1] = fminf(mn[1], r.y); mn[2] = fminf(mn[2], r.z);
            mx[0] = fmaxf(mx[0], r.x); mx[1] = fmaxf(mx[1], r.y); mx[2] = fmaxf(mx[2], r.z);
        }
    }
    if (mn[0] > mx[0]) { spheres[c] = make_float4(0.f, 0.f, 0.f, -1.0f); return; }     // no finite member: radius < 0 = never opened
    const float cx = 0.5f * mn[0] + 0.5f * mx[0], cy = 0.5f * mn[1] + 0.5f * mx[1], cz = 0.5f * mn[2] + 0.5f * mx[2];
    float r2 = 0.f;
    for (int j = 0; j < GRID_CHUNK; j++) {
        const float4 r = records[(size_t)c * GRID_CHUNK + j];
        if (finite3(r.x, r.y, r.z)) {
            const float dx = r.x - cx, dy = r.y - cy, dz = r.z - cz;
            r2 = fmaxf(r2, (dx * dx + dy * dy) + dz * dz);
        }
    }
    // rounded up: sqrtf and the three squares are within a few ulp; an overflowing radius (inf) simply never prunes
    spheres[c] = make_float4(cx, cy, cz, sqrtf(r2) * 1.00001f + 1e-30f);
}

// one thread per chunk of GRID_CHUNK consecutive records: centre = mean of the finite ones, then the shifted copy (Grid::chunks)
__global__ __launch_bounds__(GR_BLOCK) void build_chunks_kernel(const float4* __restrict__ records, uint32_t n, uint32_t n_chunks, float* __restrict__ chunks,
                                                                int* __restrict__ unsafe)
{
    const uint32_t c = blockIdx.x * GR_BLOCK + threadIdx.x;
    if (c >= n_chunks) return;
    float cx = 0.f, cy = 0.f, cz = 0.f;
    int cnt = 0;
    for (int j = 0; j < GRID_CHUNK; j++) {
        const uint32_t p = c * GRID_CHUNK + j;
        if (p >= n) break;
        const float4 r = records[p];
        if (finite3(r.x, r.y, r.z)) { cx += r.x; cy += r.y; cz += r.z; cnt++; }
    }
    if (cnt) { cx /= (float)cnt; cy /= (float)cnt; cz /= (float)cnt; }
    bool bad = !finite3(cx, cy, cz) || fabsf(cx) > 1e18f || fabsf(cy) > 1e18f || fabsf(cz) > 1e18f;
    float* out = chunks + (size_t)c * GRID_CHUNK_FLOATS;
    out[0] = cx; out[1] = cy; out[2] = cz; out[3] = 0.f;
    for (int j = 0; j < GRID_CHUNK; j++) {
        const uint32_t p = c * GRID_CHUNK + j;
        float tx = 0.f, ty = 0.f, tz = 0.f, w = INFINITY;                  // padding / non-finite: never the minimum
        if (p < n) {
            const float4 r = records[p];
            if (finite3(r.x, r.y, r.z)) {
                tx = r.x - cx; ty = r.y - cy; tz = r.z - cz;
                w = ((tx * tx + ty * ty) + tz * tz) * 0.999996185302734375f;   // (1 - 2^-18): see the error analysis in nn1_brute.hip
                if (!(fabsf(tx) < 1e18f && fabsf(ty) < 1e18f && fabsf(tz) < 1e18f) || !(w < 3e38f)) bad = true;
            }
        }
        // SoA inside the chunk; the coordinates carry the factor -2 of the cross term (exact), so the filter multiplies by r itself
        out[4 + j] = -2.0f * tx; out[4 + GRID_CHUNK + j] = -2.0f * ty; out[4 + 2 * GRID_CHUNK + j] = -2.0f * tz; out[4 + 3 * GRID_CHUNK + j] = w;
    }
    if (bad) atomicOr(unsafe, 1);
}

// records[p] for p < n; the tail up to a whole chunk is padding (x = +inf: d2 = inf is never accepted; index = none)
__global__ __launch_bounds__(GR_BLOCK) void gather_records_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                                                                  uint32_t n, uint32_t n_padded, const uint32_t* __restrict__ order, float4* __restrict__ records)
{
    const uint32_t p = blockIdx.x * GR_BLOCK + threadIdx.x;
    if (p >= n_padded) return;
    if (p >= n) { records[p] = make_float4(__builtin_inff(), 0.f, 0.f, __uint_as_float(0xFFFFFFFFu)); return; }
    const uint32_t i = order[p];
    records[p] = make_float4(x[i], y[i], z[i], __uint_as_float(i));
}


// block-local exclusive scan; block totals to `totals`
__global__ __launch_bounds__(GR_BLOCK) void scan_local_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                              uint32_t n, uint32_t* __restrict__ totals)
{
    __shared__ uint32_t wsum[GR_BLOCK / 64];
    const uint32_t base = blockIdx.x * SC_TILE + threadIdx.x * SC_ITEMS;
    uint32_t v[SC_ITEMS], s = 0;
#pragma unroll
    for (int k = 0; k < SC_ITEMS; k++) { v[k] = (base + k < n) ? in[base + k] : 0u; s += v[k]; }
    // inclusive scan of the per-thread sums across the wave, then across waves
    uint32_t inc = s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; w++) woff += wsum[w];
    uint32_t run = woff + inc - s;
#pragma unroll
    for (int k = 0; k < SC_ITEMS; k++) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
    if (threadIdx.x == GR_BLOCK - 1) totals[blockIdx.x] = run;
}

// exclusive scan of up to GR_BLOCK * 64 block totals in one workgroup (in place)
__global__ __launch_bounds__(GR_BLOCK) void scan_totals_kernel(uint32_t* __restrict__ totals, uint32_t nb, uint32_t* __restrict__ grand)
{
    __shared__ uint32_t wsum[GR_BLOCK / 64];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t base = 0; base < nb; base += GR_BLOCK) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < nb ? totals[i] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t woff = carry;
        for (int w = 0; w < wave; w++) woff += wsum[w];
        if (i < nb) totals[i] = woff + inc - v;
        __syncthreads();
        if (threadIdx.x == GR_BLOCK - 1) carry = woff + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) *grand = carry;
}

__global__ __launch_bounds__(GR_BLOCK) void scan_add_kernel(uint32_t* __restrict__ out, uint32_t n, const uint32_t* __restrict__ totals)
{
    const uint32_t off = totals[blockIdx.x];
    const uint32_t base = blockIdx.x * SC_TILE + threadIdx.x * SC_ITEMS;
#pragma unroll
    for (int k = 0; k < SC_ITEMS; k++)
        if (base + k < n) out[base + k] += off;
}

// perm[cell_start[c] + cursor[c]++] = i   (queries grouped by cell)
__global__ __launch_bounds__(GR_BLOCK) void scatter_perm_kernel(uint32_t n, const uint32_t* __restrict__ cell_of,
                                                                const uint32_t* __restrict__ cell_start,
                                                                uint32_t* __restrict__ cursor, uint32_t* __restrict__ perm)
{
    const uint32_t i = blockIdx.x * GR_BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = cell_of[i];
    perm[cell_start[c] + atomicAdd(&cursor[c], 1u)] = i;
}

// ---------------------------------------------------------------------------------------- query
constexpr unsigned long long KEY_NONE = ((unsigned long long)0x7F7FFFFFu << 32) | 0xFFFFFFFFull;   // (FLT_MAX, no index)
__device__ __forceinline__ bool has_index(unsigned long long key) { return (uint32_t)key != 0xFFFFFFFFu; }   // a real candidate, not just a bound
// Geometry proves "every point outside the scanned cube is farther than the best" for TRUE distances; the computed f32 d2
// follows the true one only above the underflow range (squares below 2^-126 lose their bits, below 2^-149 they are 0:
// a cloud of 1e-25-sized coordinates has d2 == 0 for EVERY pair, and the lowest index must win among all of them).
// So a bound is only used when it is at least TRUST (1e-15, squared 1e-30 >> 2^-126): smaller clouds are scanned whole.
constexpr float TRUST = 1e-15f, TRUST2 = 1e-30f;
constexpr uint32_t FAR_DIV = 64;      // measured: profiles/r01_tune_grid.txt

// candidates [b, e) of one x-row, strided over the G lanes of the query's sub-group: 16 B per lane, G*16 B contiguous
template <int G>
__device__ __forceinline__ void scan_range(const float4* __restrict__ records, uint32_t b, uint32_t e, int l,
                                           float qx, float qy, float qz, unsigned long long& best, uint32_t& bestp)
{
    for (uint32_t p = b + l; p < e; p += G) {
        const float4 rec = records[p];
        const float dx = qx - rec.x, dy = qy - rec.y, dz = qz - rec.z;
        const uint32_t d = __float_as_uint((dx * dx + dy * dy) + dz * dz);   // A1, unfused
        const unsigned long long k = ((unsigned long long)d << 32) | __float_as_uint(rec.w);
        // accept only d2 < FLT_MAX (nanoflann.hpp:163,1360); min over (d2 bits, original index) = canonical rule
        if (d < 0x7F7FFFFFu && k < best) { best = k; bestp = p; }
    }
}

// Every x-row of the records is sorted by x (grid_build), so a long range can be cut down to the records whose x lies
// within the current best distance of the query before any distance is evaluated: the candidates of a dense row (a LiDAR
// ring packs hundreds of points into one 3-cell row at 10 M points) shrink to the few that can still win or tie.
// [lo, hi] must contain every x with |x - qx| <= sqrt(best d2) in REAL arithmetic (the caller widens for rounding).
// Both searches are bounded (32 halvings); predicates are false for NaN, which cannot occur inside a row (non-finite
// points live in the extra cell).
__device__ __forceinline__ void clip_range_x(const float4* __restrict__ records, uint32_t& b, uint32_t& e, float lo, float hi)
{
    if (e - b <= 48u) return;                                   // short rows: the searches would cost more than they save
    uint32_t l = b, h = e;                                      // first p with x >= lo
    for (int it = 0; it < 32 && l < h; it++) {
        const uint32_t mid = l + ((h - l) >> 1);
        if (records[mid].x < lo) l = mid + 1; else h = mid;
    }
    const uint32_t nb = l;
    h = e;                                                      // first p >= nb with x > hi
    for (int it = 0; it < 32 && l < h; it++) {
        const uint32_t mid = l + ((h - l) >> 1);
        if (records[mid].x <= hi) l = mid + 1; else h = mid;
    }
    b = nb;
    e = l;
}

// window for clip_range_x from a squared bound; widened by the rounding of qx -+ d (relative to |qx|, not to d)
__device__ __forceinline__ void x_window(float qx, float bound2, float& lo, float& hi)
{
    const float d = sqrtf(bound2) * 1.00001f;
    const float pad = (fabsf(qx) + d) * 2.4e-7f;
    lo = qx - d - pad;
    hi = qx + d + pad;
}

// The nine row ranges of stage 1 as ONE flattened index space: lane l takes candidates l, l + G, ... of the concatenation.
// Rows are short (1-4 trips of G lanes each), so scanning them one after the other leaves a single load in flight per
// lane; flattened, consecutive trips are independent and the unrolled loop keeps four loads in flight.
template <int G>
__device__ __forceinline__ void scan_flat9(const float4* __restrict__ records, const uint32_t (&rb)[9], const uint32_t (&re)[9], int l,
                                           float qx, float qy, float qz, unsigned long long& best, uint32_t& bestp)
{
    uint32_t off[10], delta[9];
    off[0] = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) { off[k + 1] = off[k] + (re[k] - rb[k]); delta[k] = rb[k] - off[k]; }
#pragma unroll 4
    for (uint32_t f = (uint32_t)l; f < off[9]; f += G) {
        uint32_t p = f + delta[0];
#pragma unroll
        for (int k = 1; k < 9; k++) p = f >= off[k] ? f + delta[k] : p;
        const float4 rec = records[p];
        const float dx = qx - rec.x, dy = qy - rec.y, dz = qz - rec.z;
        const uint32_t d = __float_as_uint((dx * dx + dy * dy) + dz * dz);   // A1, unfused
        const unsigned long long key = ((unsigned long long)d << 32) | __float_as_uint(rec.w);
        if (d < 0x7F7FFFFFu && key < best) { best = key; bestp = p; }
    }
}

// minimum key over the sub-group, together with the record position that produced it
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xF, 0xF, false);
}

template <int CTRL>
__device__ __forceinline__ void min_step_dpp(unsigned long long& v, uint32_t& pos)
{
    const unsigned long long w = ((unsigned long long)dpp_mov<CTRL>((uint32_t)(v >> 32)) << 32) | dpp_mov<CTRL>((uint32_t)v);
    const uint32_t wp = dpp_mov<CTRL>(pos);
    if (w < v) { v = w; pos = wp; }
}

template <int G>
__device__ __forceinline__ void group_min(unsigned long long& v, uint32_t& pos)
{
    if (G == 16) {
        // 16 lanes = one DPP row: xor 1, xor 2 (quad permutes), then the half-row and the row mirror — each step pairs groups that are
        // already uniform, so four vector moves per value replace four LDS-crossbar round trips (ds_bpermute: the kernel held 104)
        min_step_dpp<0xB1>(v, pos);      // quad_perm [1, 0, 3, 2]
        min_step_dpp<0x4E>(v, pos);      // quad_perm [2, 3, 0, 1]
        min_step_dpp<0x141>(v, pos);     // row_half_mirror
        min_step_dpp<0x140>(v, pos);     // row_mirror
        return;
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) {
        const unsigned long long w = __shfl_xor(v, o, 64);
        const uint32_t wp = __shfl_xor(pos, o, 64);
        if (w < v) { v = w; pos = wp; }
    }
}

// inclusive prefix sum over the 16 lanes of a DPP row (row_shr 1, 2, 4, 8 with zero fill) and the maximum over the row
__device__ __forceinline__ uint32_t row_scan16(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
    return v;
}

__device__ __forceinline__ uint32_t row_max16(uint32_t v)
{
    v = max(v, dpp_mov<0xB1>(v));
    v = max(v, dpp_mov<0x4E>(v));
    v = max(v, dpp_mov<0x141>(v));
    v = max(v, dpp_mov<0x140>(v));
    return v;
}

// ---- bounding-sphere pruning (SPH mode) -------------------------------------------------------------------------------
// Can a run of records with bounding sphere s hold a neighbour that beats or ties `best`?  Every member is at least
// |q - C| - radius away (radius is rounded up at build time); the test is conservative against the few-ulp errors of the f32
// evaluation on both sides (same margins as the cube proof) and is never trusted below TRUST or when |q - C|^2 overflows.
__device__ __forceinline__ bool sphere_may_win(const float4 s, float qx, float qy, float qz, unsigned long long best)
{
    if (s.w < 0.0f) return false;                         // a run without a finite member
    if (best == KEY_NONE) return true;
    const float bestf = fmaxf(__uint_as_float((uint32_t)(best >> 32)), TRUST2);
    const float dx = qx - s.x, dy = qy - s.y, dz = qz - s.z;
    const float dc2 = (dx * dx + dy * dy) + dz * dz;
    if (!(dc2 < 3.0e38f)) return true;
    const float sep = sqrtf(dc2) * 0.99999f - s.w;
    return !(sep > TRUST && sep * sep * 0.99999f > bestf * 1.0001f);
}

// the sub-group scans run c (GRID_CHUNK = G = 16 records: one coalesced 256-byte load)
__device__ __forceinline__ void scan_run16(const float4* __restrict__ records, uint32_t c, int l, float qx, float qy, float qz,
                                           unsigned long long& best, uint32_t& bestp)
{
    const uint32_t p = c * GRID_CHUNK + (uint32_t)l;
    const float4 rec = records[p];
    const float dx = qx - rec.x, dy = qy - rec.y, dz = qz - rec.z;
    const uint32_t d = __float_as_uint((dx * dx + dy * dy) + dz * dz);   // A1, unfused
    const unsigned long long k = ((unsigned long long)d << 32) | __float_as_uint(rec.w);
    if (d < 0x7F7FFFFFu && k < best) { best = k; bestp = p; }
}

// the sub-group scans the runs whose bit is set in m (bit j = run held by lane j in cc): two coalesced loads in flight.  (Computing
// the run index of lane j locally instead of reading it across lanes — nine compare-selects against one ds_bpermute in front of every
// record load — was measured: 7.29 -> 7.91 ms per 10 M ICP iteration; the vector ALU is the scarcer resource here.)
__device__ __forceinline__ void scan_hits16(const float4* __restrict__ records, uint32_t m, uint32_t cc, int l, float qx, float qy, float qz,
                                            unsigned long long& best, uint32_t& bestp)
{
    while (m) {
        const int j0 = __builtin_ctz(m);
        m &= m - 1;
        const uint32_t p0 = __shfl(cc, j0, 16) * GRID_CHUNK + (uint32_t)l;
        const float4 r0 = records[p0];
        uint32_t p1 = p0;
        float4 r1 = r0;
        const bool two = m != 0;
        if (two) {
            const int j1 = __builtin_ctz(m);
            m &= m - 1;
            p1 = __shfl(cc, j1, 16) * GRID_CHUNK + (uint32_t)l;
            r1 = records[p1];
        }
        {
            const float dx = qx - r0.x, dy = qy - r0.y, dz = qz - r0.z;
            const uint32_t d = __float_as_uint((dx * dx + dy * dy) + dz * dz);   // A1, unfused
            const unsigned long long k = ((unsigned long long)d << 32) | __float_as_uint(r0.w);
            if (d < 0x7F7FFFFFu && k < best) { best = k; bestp = p0; }
        }
        if (two) {
            const float dx = qx - r1.x, dy = qy - r1.y, dz = qz - r1.z;
            const uint32_t d = __float_as_uint((dx * dx + dy * dy) + dz * dz);
            const unsigned long long k = ((unsigned long long)d << 32) | __float_as_uint(r1.w);
            if (d < 0x7F7FFFFFu && k < best) { best = k; bestp = p1; }
        }
    }
}

// records [b, e) of one x-row piece, as the runs that overlap it.  Records of a run that lie outside [b, e) are genuine
// targets too, so scanning whole runs cannot change the result.  NB batches of 16 spheres are fetched up front (independent
// loads), tested against the bound the pass started with, and their hits scanned; the bound is refreshed once per pass.
template <bool STATS, int NB>
__device__ __forceinline__ void scan_range_sph(const float4* __restrict__ records, const float4* __restrict__ spheres, uint32_t b, uint32_t e, int l,
                                               float qx, float qy, float qz, unsigned long long& best, uint32_t& bestp,
                                               unsigned long long& st_cand, unsigned long long& st_sph)
{
    if (b >= e) return;
    const uint32_t c0 = b / GRID_CHUNK, c1 = (e - 1) / GRID_CHUNK + 1;
    const int shift = (int)((threadIdx.x & 63) / 16 * 16);
    for (uint32_t cb = c0; cb < c1; cb += 16 * NB) {
        float4 sp[NB];
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const uint32_t c = cb + 16u * k + (uint32_t)l;
            sp[k] = c < c1 ? spheres[c] : make_float4(0.f, 0.f, 0.f, -1.0f);
        }
        bool any = false;
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const uint32_t m = (uint32_t)(__ballot(sphere_may_win(sp[k], qx, qy, qz, best)) >> shift) & 0xFFFFu;
            if (STATS && l == 0) { st_sph += cb + 16u * k < c1 ? min(16u, c1 - cb - 16u * k) : 0u; st_cand += 16u * (uint32_t)__popc(m); }
            if (m) { scan_hits16(records, m, cb + 16u * k + (uint32_t)l, l, qx, qy, qz, best, bestp); any = true; }
        }
        if (any) group_min<16>(best, bestp);       // the next pass is tested against the improved bound
    }
}

// the nine row ranges of stage 1 as ONE flattened space of runs
template <bool STATS, int NB>
__device__ __forceinline__ void scan_flat9_sph(const float4* __restrict__ records, const float4* __restrict__ spheres, const uint32_t (&rb)[9],
                                               const uint32_t (&re)[9], int l, float qx, float qy, float qz, unsigned long long& best, uint32_t& bestp,
                                               unsigned long long& st_cand, unsigned long long& st_sph)
{
    uint32_t off[10], delta[9];
    off[0] = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const uint32_t b = rb[k], e = re[k];
        const uint32_t c0 = b < e ? b / GRID_CHUNK : 0u, c1 = b < e ? (e - 1) / GRID_CHUNK + 1 : 0u;
        off[k + 1] = off[k] + (c1 - c0);
        delta[k] = c0 - off[k];
    }
    const int shift = (int)((threadIdx.x & 63) / 16 * 16);
    for (uint32_t f0 = 0; f0 < off[9]; f0 += 16 * NB) {
        float4 sp[NB];
        uint32_t cc[NB];
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const uint32_t f = f0 + 16u * b + (uint32_t)l;
            uint32_t c = f + delta[0];
#pragma unroll
            for (int k = 1; k < 9; k++) c = f >= off[k] ? f + delta[k] : c;
            cc[b] = c;
            sp[b] = f < off[9] ? spheres[c] : make_float4(0.f, 0.f, 0.f, -1.0f);
        }
        bool any = false;
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const uint32_t m = (uint32_t)(__ballot(sphere_may_win(sp[b], qx, qy, qz, best)) >> shift) & 0xFFFFu;
            if (STATS && l == 0) { st_sph += f0 + 16u * b < off[9] ? min(16u, off[9] - f0 - 16u * b) : 0u; st_cand += 16u * (uint32_t)__popc(m); }
            if (m) { scan_hits16(records, m, cc[b], l, qx, qy, qz, best, bestp); any = true; }
        }
        if (any) group_min<16>(best, bestp);
    }
}

// Distance (lower bound, >= 0) along one axis between the query coordinate q and the slab of cells with index c, u being the
// query's own cell: 0 for its own slab, else the gap to the nearer face minus a margin that covers the binning of the targets
// (a point may sit `slack` cells beyond the face its f32 cell index suggests) and the rounding of the face coordinate.
__device__ __forceinline__ float axis_gap(float q, float lo, float h, int c, int u, float slack)
{
    if (c == u) return 0.0f;
    const float face = c > u ? lo + (float)c * h : lo + (float)(c + 1) * h;
    const float gap = c > u ? face - q : q - face;
    return fmaxf(gap - slack * h - (fabsf(q) + fabsf(face)) * 1e-6f, 0.0f);
}

// x-range of cells [xa, xb] that can hold a target within sqrt(rem2) of qx (rem2 >= 0): the cell index is a monotone f32
// function of the coordinate, so the cells of qx -+ r bracket every such target; the margin covers the rounding of qx -+ r
__device__ __forceinline__ void ball_x_cells(const GridParams& g, float qx, float rem2, int& xa, int& xb)
{
    const float r = sqrtf(rem2) * 1.0001f + g.slack * g.h + fabsf(qx) * 1e-6f;
    xa = max(xa, cell_coord(qx - r, g.lo[0], g.inv_h));
    xb = min(xb, cell_coord(qx + r, g.lo[0], g.inv_h));
}

// Later stages of the sphere walk: every lane of the sub-group holds up to two record ranges (the pieces of ITS row).  Their runs
// are laid out as one index space (exclusive scan over the lanes), so that the 16 lanes test 16 spheres per batch whatever
// rows they come from — instead of one sparse batch per piece.  The owner of flattened run f is found by a 4-step binary
// search over the lanes' offsets (shuffles).
template <bool STATS>
__device__ __forceinline__ void scan_pieces_sph(const float4* __restrict__ records, const float4* __restrict__ spheres, uint32_t b1, uint32_t e1,
                                                uint32_t b2, uint32_t e2, int l, float qx, float qy, float qz, unsigned long long& best, uint32_t& bestp,
                                                unsigned long long& st_cand, unsigned long long& st_sph, uint32_t seed_run)
{
    const uint32_t a0 = b1 < e1 ? b1 / GRID_CHUNK : 0u, na = b1 < e1 ? (e1 - 1) / GRID_CHUNK + 1 - a0 : 0u;
    const uint32_t c0 = b2 < e2 ? b2 / GRID_CHUNK : 0u, nc = b2 < e2 ? (e2 - 1) / GRID_CHUNK + 1 - c0 : 0u;
    const uint32_t cnt = na + nc;
    const uint32_t inc = row_scan16(cnt);
    const uint32_t my_off = inc - cnt;
    const uint32_t total = row_max16(inc);                 // (the scan is non-decreasing: its maximum is lane 15's value)
    const int shift = (int)((threadIdx.x & 63) / 16 * 16);
    for (uint32_t f0 = 0; f0 < total; f0 += 16) {
        const uint32_t f = f0 + (uint32_t)l;
        int owner = 0;
#pragma unroll
        for (int step = 8; step > 0; step >>= 1) {
            const uint32_t o = __shfl(my_off, owner + step, 16);
            if (o <= f) owner += step;
        }
        const uint32_t rel = f - __shfl(my_off, owner, 16);
        const uint32_t oa0 = __shfl(a0, owner, 16), ona = __shfl(na, owner, 16), oc0 = __shfl(c0, owner, 16);
        const uint32_t c = rel < ona ? oa0 + rel : oc0 + (rel - ona);
        const bool hit = f < total && c != seed_run && sphere_may_win(spheres[c], qx, qy, qz, best);
        const uint32_t m = (uint32_t)(__ballot(hit) >> shift) & 0xFFFFu;
        if (STATS && l == 0) { st_sph += min(16u, total - f0); st_cand += 16u * (uint32_t)__popc(m); }
        if (m) {
            scan_hits16(records, m, c, l, qx, qy, qz, best, bestp);
            group_min<16>(best, bestp);
        }
    }
}

// Stage 1 of the sphere walk with ONE ROW PER LANE: lane k < 9 of the sub-group resolves row k of the 3 x 3 block (ball clipping
// against the seed, the two cell_start bounds, the run range), an exclusive scan over the lanes lays the runs of the nine rows
// out as one index space, and every lane then fetches the nine (offset, delta) pairs with shuffles.  The same work done
// redundantly by all 16 lanes cost ~450 vector instructions per query; here it is ~60 + 30 shuffles.
template <bool STATS>
__device__ __forceinline__ void stage1_sph(const float4* __restrict__ records, const float4* __restrict__ spheres, const uint32_t* __restrict__ cell_start,
                                           const GridParams& g, int ux, int uy, int uz, int l, float qx, float qy, float qz, unsigned long long& best,
                                           uint32_t& bestp, unsigned long long& st_cand, unsigned long long& st_sph, unsigned long long& st_rows,
                                           uint32_t seed_run)
{
    const int xlo = max(ux - 1, 0), xhi = min(ux + 1, g.n[0] - 1);
    const bool bounded = best != KEY_NONE, seeded = has_index(best);     // (bounded without a seed: the caller's gate, nn1_grid_kernel)
    const float clip2s = fmaxf(__uint_as_float((uint32_t)(best >> 32)), TRUST2) * 1.0001f;
    // lane k: row k (k = 4 is the query's own row); lanes 9..15 hold empty rows
    const int k = l;
    const int cy = uy + (k % 3) - 1, cz = uz + (k / 3) - 1;
    bool ok = k < 9 && (cy >= 0) && (cy < g.n[1]) && (cz >= 0) && (cz < g.n[2]) && (xlo <= xhi);
    int xa = xlo, xb = xhi;
    if (ok && bounded) {
        const float gy = axis_gap(qy, g.lo[1], g.h, cy, uy, g.slack), gz = axis_gap(qz, g.lo[2], g.h, cz, uz, g.slack);
        const float rem2 = clip2s - (gy * gy + gz * gz) * 0.9999f;
        if (rem2 < 0.0f) ok = false;
        else { ball_x_cells(g, qx, rem2, xa, xb); ok = xa <= xb; }
    }
    uint32_t b = 0, e = 0;
    if (ok) {
        const uint32_t row = (uint32_t)((cz * g.n[1] + cy) * g.n[0]);
        b = cell_start[row + xa];
        e = cell_start[row + xb + 1];
    }
    if (STATS && ok) st_rows++;
    if (!seeded) {
        // no seed: the query's own row first (all 16 lanes on it), so that the other eight are tested against a real bound
        const uint32_t b4 = __shfl(b, 4, 16), e4 = __shfl(e, 4, 16);
        scan_range_sph<STATS, 1>(records, spheres, b4, e4, l, qx, qy, qz, best, bestp, st_cand, st_sph);
        if (l == 4) { b = 0; e = 0; }
    }
    const uint32_t c0 = b < e ? b / GRID_CHUNK : 0u, c1 = b < e ? (e - 1) / GRID_CHUNK + 1 : 0u;
    const uint32_t cnt = c1 - c0;
    const uint32_t inc = row_scan16(cnt);                   // inclusive scan over the 16 lanes (DPP, no LDS)
    const uint32_t my_off = inc - cnt;                      // first flattened run of my row
    const uint32_t total = row_max16(inc);
    const uint32_t my_delta = c0 - my_off;
    // every lane needs the nine (offset, delta) pairs: written once to the sub-group's 20 words of LDS and read back as broadcasts
    // (18 ds_bpermute before; a wave's DS operations execute in order, and no other sub-group touches these words)
    __shared__ uint32_t s1_pairs[GR_BLOCK / 16][20];
    uint32_t* mine = s1_pairs[threadIdx.x / 16];
    if (l < 9) { mine[l] = my_off; mine[10 + l] = my_delta; }
    uint32_t off[9], delta[9];
#pragma unroll
    for (int r = 0; r < 9; r++) { off[r] = mine[r]; delta[r] = mine[10 + r]; }
    const int shift = (int)((threadIdx.x & 63) / 16 * 16);
    for (uint32_t f0 = 0; f0 < total; f0 += 16) {
        const uint32_t f = f0 + (uint32_t)l;
        uint32_t c = f + delta[0];
#pragma unroll
        for (int r = 1; r < 9; r++) c = f >= off[r] ? f + delta[r] : c;
        const bool hit = f < total && c != seed_run && sphere_may_win(spheres[c], qx, qy, qz, best);
        const uint32_t m = (uint32_t)(__ballot(hit) >> shift) & 0xFFFFu;
        if (STATS && l == 0) { st_sph += min(16u, total - f0); st_cand += 16u * (uint32_t)__popc(m); }
        if (m) {
            scan_hits16(records, m, c, l, qx, qy, qz, best, bestp);
            group_min<16>(best, bestp);
        }
    }
}

// Workgroup -> chunk of queries.  The hardware deals consecutive workgroups round-robin over the 8 XCDs (b and b + 8 share
// an L2): with the identity mapping eight spatially adjacent workgroups pull the same records into eight L2s.  Here XCD x
// gets whole runs of GR_XCD_RUN consecutive (= spatially adjacent) query blocks, and the runs are dealt round-robin, so that
// the load stays balanced over the XCDs (a contiguous eighth of the sorted queries per XCD does not: profiles/r01_grid_xcd_swizzle_experiment.txt).
// Speed only: any mapping is correct.  nb8 = number of workgroups launched (a multiple of 8 * run).
__device__ __forceinline__ uint32_t xcd_block(uint32_t b, uint32_t run)
{
    const uint32_t xcd = b & 7u, slot = b >> 3;
    return ((slot / run) * 8u + xcd) * run + slot % run;
}

// G lanes cooperate on one query (G divides 64).  Stage 1 scans the 3 x 3 x-rows of the radius-1 cube with all
// row bounds fetched up front (18 independent loads in flight); later stages double the radius.
// MODE 0: every record of an opened row is evaluated (small targets).  MODE 1 (CLIP, x-sorted rows): long rows are first cut to
// the best-distance window in x.  MODE 2 (SPH, G = 16): rows are walked as runs of 16 records whose bounding spheres are
// tested first — the variant for large / dense targets (a LiDAR ring packs hundreds of points into one cell at 10 M points;
// almost all of them lie outside the ball of the current best).
// waves per SIMD of the list-mode instance (unbounded: 77 VGPRs = 6 waves; 7 waves: 71 VGPRs, nothing spilled; 8 waves: 64 VGPRs with
// eight registers in scratch — as fast, but 115 MB of scratch traffic per search at 10 M)
#ifndef PCR_LIST_WAVES
#define PCR_LIST_WAVES 7
#endif
template <int G, bool STATS, int MODE, bool LIST = false>
__global__ __launch_bounds__(GR_BLOCK, (LIST && !STATS) ? PCR_LIST_WAVES : 1) void nn1_grid_kernel(
    const float4* __restrict__ records, const float4* __restrict__ spheres, const uint32_t* __restrict__ cell_start,
    GridParams g, const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz,
    const uint32_t* __restrict__ perm, uint32_t ns, unsigned long long* __restrict__ keys, const int* __restrict__ stop,
    unsigned long long* __restrict__ stats, const float* __restrict__ tx, const float* __restrict__ ty,
    const float* __restrict__ tz, uint32_t nt, int warm_start, float cap2,
    uint32_t* __restrict__ far_list, uint32_t* __restrict__ far_count, uint32_t far_cap,
    uint32_t* __restrict__ wpos, uint32_t xcd_run, const uint32_t* __restrict__ list_count, uint32_t list_segs, uint32_t* __restrict__ list_queue = nullptr)
{
    // pipelined ICP: once the loop has ended the enqueued tail is a no-op.  The flags are REQUESTED here and tested below, after the
    // query's own loads have been issued: one memory round trip of every launch's serial chain less.
    const int stopv = stop ? (stop[0] | stop[1]) : 0;
    constexpr bool CLIP = MODE == 1, SPH = MODE == 2;
    static_assert(!SPH || G == 16, "the sphere walk scans one 16-record run per sub-group");
    const uint32_t vb = xcd_run ? xcd_block(blockIdx.x, xcd_run) : blockIdx.x;
    // LIST MODE (template LIST, G = 16): the queries the tile search deferred (grid_stile.hpp).  perm[] is a SEGMENTED list — segment s =
    // the deferred queries of the 32-query group s of the sorted working cloud, list_count[s] of them at perm[32 s ...] — so the list is
    // in the order of the working cloud however the tile kernel's waves were scheduled.  A fixed number of workgroups serves it: every
    // wave owns `list_segs` consecutive segments (16: at most 512 queries — small enough for the hardware's dispatch order to balance
    // the load, which a fixed split over 8 192 resident waves did not: the deferred queries are the expensive ones and cluster in
    // space; runs of 32 workgroups share an XCD, like the query blocks of the plain launch), lays their entries out as one index space
    // (counts -> inclusive scan -> LDS) and deals them to its four sub-groups.  (Its own instantiation: as a run-time mode the loop
    // cost the plain form a wave of occupancy — 69 against 63 VGPRs.)
    if (!LIST && (unsigned long long)vb * GR_BLOCK >= (unsigned long long)ns * G) return;   // a surplus workgroup of the padded launch
    if (LIST && stopv) return;
    __shared__ uint32_t seg_off[LIST ? GR_BLOCK / 64 : 1][LIST ? 65 : 1];
    const uint32_t n_seg = LIST ? (ns + 31u) / 32u : 0u;
    const uint32_t lb = LIST ? ((gridDim.x % 256u == 0u) ? xcd_block(blockIdx.x, 32u) : blockIdx.x) : 0u;
    const uint32_t wv = LIST ? lb * (GR_BLOCK / 64) + (threadIdx.x >> 6) : 0u;
    uint32_t seg0 = wv * list_segs, seg1 = min(n_seg, seg0 + list_segs);
    // QUEUE form of the list mode (list_queue != nullptr; the sign tile search, grid_stile.hpp): the non-empty segments were appended to
    // list_queue[2 ...] (their number in [0]) by the waves that deferred them, and a FIXED number of resident waves draw them one by one
    // through the ticket counter [1] — no workgroup is launched for the empty segments (at the converged pose 0.4 % of the queries are
    // deferred, yet one workgroup per four segments cost 0.18 ms of a 1.5 ms search), and the load balances itself.  Every wave leaves as
    // soon as the ticket it draws lies beyond the list: the grid drains.
    const bool queue = LIST && list_queue != nullptr;
    bool first_ticket = true;
    uint32_t qpart = 0;
    for (;;) {
    if (queue) {
        // the first ticket of a wave is its own number (8 192 waves drawing from ONE counter at the same moment took 0.16 ms — the whole
        // launch, however short the list), the later ones come from the counter behind those
        const uint32_t n_static = gridDim.x * (GR_BLOCK / 64);
        uint32_t it = blockIdx.x * (GR_BLOCK / 64) + (threadIdx.x >> 6);
        if (!first_ticket) {
            if ((threadIdx.x & 63u) == 0u) it = n_static + atomicAdd(&list_queue[1], 1u);
            it = (uint32_t)__builtin_amdgcn_readfirstlane((int)it);
        }
        first_ticket = false;
        if (it >= min(list_queue[0], 4u * n_seg)) break;
        // an item = a quarter of a segment (at most 8 deferred queries: two rounds of this wave's four sub-groups — a whole segment of 32 took
        // eight rounds one after the other, and the slowest wave is the launch's duration: ~0.1 ms however short the list)
        const uint32_t item = list_queue[2 + it];
        qpart = item & 3u;
        seg0 = min(item >> 2, n_seg - 1u); seg1 = seg0 + 1u;
    }
    for (uint32_t cs = seg0; LIST ? cs < seg1 : true; cs += 64) {
    uint32_t total = 0;
    if (LIST) {
        const uint32_t lane = threadIdx.x & 63u;
        uint32_t c = cs + lane < seg1 ? min(list_count[cs + lane], 32u) : 0u;
        if (queue) c = c > 8u * qpart ? min(c - 8u * qpart, 8u) : 0u;                  // (queue form: this item's quarter of the one segment)
        uint32_t inc = row_scan16(c);
        const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)inc, 15), t1 = (uint32_t)__builtin_amdgcn_readlane((int)inc, 31),
                       t2 = (uint32_t)__builtin_amdgcn_readlane((int)inc, 47), t3 = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        inc += lane >= 48 ? t0 + t1 + t2 : lane >= 32 ? t0 + t1 : lane >= 16 ? t0 : 0u;
        total = t0 + t1 + t2 + t3;
        seg_off[LIST ? threadIdx.x >> 6 : 0][LIST ? lane : 0] = inc - c;             // first entry of segment cs + lane
        if (lane == 0) seg_off[LIST ? threadIdx.x >> 6 : 0][LIST ? 64 : 0] = total;
    }
    for (uint32_t slot = LIST ? (threadIdx.x & 63u) / G : (vb * GR_BLOCK + threadIdx.x) / G; LIST ? slot < total : true; slot += 64 / G) {
        const uint32_t n_eff = ns;
        uint32_t list_pos = 0;
        if (LIST) {
            const uint32_t* off = seg_off[LIST ? threadIdx.x >> 6 : 0];
            int sg = 0;
#pragma unroll
            for (int step = 32; step > 0; step >>= 1)
                if (off[LIST ? sg + step : 0] <= slot) sg += step;                    // the last segment whose first entry is <= slot
            list_pos = (cs + (uint32_t)sg) * 32u + (slot - off[LIST ? sg : 0]) + (queue ? 8u * qpart : 0u);
        }
        const bool live = LIST || slot < n_eff;
        const uint32_t t = LIST ? list_pos : min(slot, n_eff - 1);         // clamp: surplus sub-groups redo the last query (same value written)
        const int l = (int)(threadIdx.x % G);
        const uint32_t i = perm ? perm[t] : t;
        const float qx = sx[i], qy = sy[i], qz = sz[i];
        const uint32_t pp0 = (warm_start >= 2) ? wpos[i] : 0xFFFFFFFFu;
        if (stopv) return;
        // The caller discards every neighbour with d2 >= cap2 (the ICP gate), so cap2 itself is a bound the walk may prune with from
        // the start: the search begins with the pseudo-candidate (cap2, no index).  Rows and cells outside the cap2 ball are never
        // opened, runs whose sphere lies outside it are never scanned, and a query with no target inside it ends with "none"
        // (what the gate would have made of any farther neighbour).  A real candidate replaces it as soon as one is closer.
        unsigned long long st_cand = 0, st_rows = 0, st_stages = 0, st_sph = 0;   // diagnostics (STATS builds only)
        const unsigned long long bound0 = (cap2 > 0.0f && cap2 < 1e30f) ? (((unsigned long long)__float_as_uint(cap2) << 32) | 0xFFFFFFFFull) : KEY_NONE;
        unsigned long long best = bound0;
        uint32_t bestp = 0;
        uint32_t seed_run = 0xFFFFFFFFu;             // the run scanned ahead of the walk (warm_start 3), which then skips it
        if (finite3(qx, qy, qz)) {
            if (SPH && warm_start == 3 && pp0 < nt) {
                // The previous winner's whole RUN (its 16 Morton neighbours, one coalesced 256-byte load, a record per lane) instead of the
                // winner alone: while the pose still moves by centimetres per iteration, one of the neighbours is often the new nearest
                // point or close to it, and the ball every later sphere test and row clip works with is that much smaller.
                seed_run = pp0 / GRID_CHUNK;
                scan_run16(records, seed_run, l, qx, qy, qz, best, bestp);
                group_min<16>(best, bestp);
            } else if (warm_start >= 2) {
                // ICP, from the second search of a loop on: wpos[] holds the record position of this query's previous winner.
                // That target, evaluated exactly against the moved query, is a genuine candidate: it bounds the search from the
                // first stage on (the radius jumps straight to the proving one, rows are clipped to its ball) without changing
                // the result.  One 16-byte load that neighbouring queries share, instead of three 4-byte gathers.
                const uint32_t pp = pp0;
                if (pp < nt) {
                    const float4 rec = records[pp];
                    const float dx = qx - rec.x, dy = qy - rec.y, dz = qz - rec.z;
                    const uint32_t d = __float_as_uint((dx * dx + dy * dy) + dz * dz);
                    const unsigned long long kk = ((unsigned long long)d << 32) | __float_as_uint(rec.w);
                    if (d < 0x7F7FFFFFu && kk < best) { best = kk; bestp = pp; }
                }
            } else if (warm_start == 1) {
                // the same from keys[] (original index) when no record positions were kept
                const uint32_t pj = (uint32_t)(keys[i] & 0xFFFFFFFFull);
                if (pj < nt) {
                    const float dx = qx - tx[pj], dy = qy - ty[pj], dz = qz - tz[pj];
                    const uint32_t d = __float_as_uint((dx * dx + dy * dy) + dz * dz);
                    const unsigned long long kk = ((unsigned long long)d << 32) | pj;
                    if (d < 0x7F7FFFFFu && kk < best) best = kk;
                }
            }
            const int ux = cell_coord(qx, g.lo[0], g.inv_h), uy = cell_coord(qy, g.lo[1], g.inv_h), uz = cell_coord(qz, g.lo[2], g.inv_h);
            // Chebyshev distance (in cells) from the query's cell to the grid box: smaller cubes hold no cell
            const int r0 = max(max(max(-ux, ux - (g.n[0] - 1)), max(-uy, uy - (g.n[1] - 1))), max(max(-uz, uz - (g.n[2] - 1)), 0));
            int r = max(r0, 1);
            // a query farther from the grid box than the caller's gate (cap2) has no admissible neighbour at all
            const float out_reach = ((float)r0 - 1.0f - g.slack) * g.h;
            bool done = r0 > 1 && out_reach > TRUST && out_reach * out_reach * 0.99999f >= cap2;
            if (r == 1 && SPH) {
                // ---- stage 1, sphere walk: one row per lane
                stage1_sph<STATS>(records, spheres, cell_start, g, ux, uy, uz, l, qx, qy, qz, best, bestp, st_cand, st_sph, st_rows, seed_run);
                group_min<G>(best, bestp);
                const bool covers = (ux - 1 <= 0) && (ux + 1 >= g.n[0] - 1) && (uy - 1 <= 0) && (uy + 1 >= g.n[1] - 1) &&
                                    (uz - 1 <= 0) && (uz + 1 >= g.n[2] - 1);
                const float reach = (1.0f - g.slack) * g.h;
                done = covers || (reach > TRUST && (reach * reach * 0.99999f >= cap2 ||
                                                    (best != KEY_NONE && __uint_as_float((uint32_t)(best >> 32)) < reach * reach * 0.99999f)));
                r = 2;
            } else if (r == 1) {
                // ---- stage 1: static 3 x 3 rows, bounds first
                const int xlo = max(ux - 1, 0), xhi = min(ux + 1, g.n[0] - 1);
                uint32_t rb[9], re[9];
                // with a seed (the previous correspondence, re-evaluated) only the rows and cells its ball reaches can matter: at the
                // converged pose that is the query's own cell and the odd neighbour instead of all 27 (measured: DESIGN.md 5b)
                const bool bounded = best != KEY_NONE;
                const float clip2s = fmaxf(__uint_as_float((uint32_t)(best >> 32)), TRUST2) * 1.0001f;
#pragma unroll
                for (int k = 0; k < 9; k++) {
                    const int cy = uy + (k % 3) - 1, cz = uz + (k / 3) - 1;
                    bool ok = (cy >= 0) && (cy < g.n[1]) && (cz >= 0) && (cz < g.n[2]) && (xlo <= xhi);
                    int xa = xlo, xb = xhi;
                    if (ok && bounded) {
                        const float gy = axis_gap(qy, g.lo[1], g.h, cy, uy, g.slack), gz = axis_gap(qz, g.lo[2], g.h, cz, uz, g.slack);
                        const float rem2 = clip2s - (gy * gy + gz * gz) * 0.9999f;
                        if (rem2 < 0.0f) ok = false;
                        else { ball_x_cells(g, qx, rem2, xa, xb); ok = xa <= xb; }
                    }
                    const uint32_t row = ok ? (uint32_t)((cz * g.n[1] + cy) * g.n[0]) : 0u;
                    rb[k] = ok ? cell_start[row + xa] : 0u;
                    re[k] = ok ? cell_start[row + xb + 1] : 0u;
                    if (STATS && l == 0 && ok) st_rows++;
                }
                if (CLIP) {
                    uint32_t total = 0;
#pragma unroll
                    for (int k = 0; k < 9; k++) total += re[k] - rb[k];
                    // dense neighbourhood without a warm-start candidate: the query's own row first, its best distance then
                    // cuts the other rows
                    if (!has_index(best) && total > 512u) {
                        scan_range<G>(records, rb[4], re[4], l, qx, qy, qz, best, bestp);
                        if (STATS && l == 0) st_cand += re[4] - rb[4];
                        rb[4] = re[4] = 0;
                        group_min<G>(best, bestp);
                    }
                    if (best != KEY_NONE && total > 48u) {
                        float lo, hi;
                        x_window(qx, fmaxf(__uint_as_float((uint32_t)(best >> 32)), TRUST2) * 1.0001f, lo, hi);
                        if (G >= 9) {
                            // one row per lane: the nine binary searches run side by side (one chain of ~12 dependent loads
                            // instead of nine), the results travel back through shuffles
                            uint32_t cb = 0, ce = 0;
#pragma unroll
                            for (int k = 0; k < 9; k++) if (l == k) { cb = rb[k]; ce = re[k]; }
                            if (l < 9) clip_range_x(records, cb, ce, lo, hi);
#pragma unroll
                            for (int k = 0; k < 9; k++) { rb[k] = __shfl(cb, k, G); re[k] = __shfl(ce, k, G); }
                        } else {
#pragma unroll
                            for (int k = 0; k < 9; k++) clip_range_x(records, rb[k], re[k], lo, hi);
                        }
                    }
                }
                {
                    if (CLIP) {
                        scan_flat9<G>(records, rb, re, l, qx, qy, qz, best, bestp);      // dense rows: 8.1 -> 6.9 ms at 10 M (costs 15 % at 120 k)
                    } else {
#pragma unroll
                        for (int k = 0; k < 9; k++) scan_range<G>(records, rb[k], re[k], l, qx, qy, qz, best, bestp);
                    }
                    if (STATS && l == 0) { for (int k = 0; k < 9; k++) st_cand += re[k] - rb[k]; }
                }
                group_min<G>(best, bestp);
                const bool covers = (ux - 1 <= 0) && (ux + 1 >= g.n[0] - 1) && (uy - 1 <= 0) && (uy + 1 >= g.n[1] - 1) &&
                                    (uz - 1 <= 0) && (uz + 1 >= g.n[2] - 1);
                const float reach = (1.0f - g.slack) * g.h;
                // (cap2: the caller discards neighbours with d2 >= cap2, so a cube that no closer target can lie outside of ends the search)
                done = covers || (reach > TRUST && (reach * reach * 0.99999f >= cap2 ||
                                                    (best != KEY_NONE && __uint_as_float((uint32_t)(best >> 32)) < reach * reach * 0.99999f)));
                r = 2;
            }
            // ---- later stages.  rp = radius of the cube already scanned (0: none).  The next radius is the smallest one
            // that can PROVE the current best (best < LB(r)), or 2 * r when nothing was found yet.  Inside the new cube,
            // cells of the old cube are skipped, and so are cells that lie farther from the query than the current best
            // (ball clipping, conservative by one cell + slack; such cells cannot hold a better or equal candidate).
            // |u| <= 2^22 and every grid dimension is <= 4002, so a cube of radius 2^23 covers the grid from any query:
            // the radius at least doubles whenever nothing is proven, so `covers` holds after <= 24 steps; the step bound
            // makes termination unconditional.
            int rp = (r == 2) ? 1 : 0;
            for (int step = 0; step < 28 && !done; step++) {
                const bool have = best != KEY_NONE;
                const float bestf = fmaxf(__uint_as_float((uint32_t)(best >> 32)), TRUST2);   // never reason below the trusted range
                if (have) {
                    // smallest r with ((r - slack) h)^2 * 0.99999 > best; with only the gate as a bound the radius keeps doubling
                    // (near shells first: a real candidate tightens the ball for the far ones) but never beyond the gate's radius
                    const int need = (int)fminf(sqrtf(bestf * 1.00002f) * g.inv_h + g.slack, 16777215.0f) + 1;
                    r = max(min(has_index(best) ? need : min(r, need), 1 << 24), rp + 1);
                }
                const int xlo = max(ux - r, 0), xhi = min(ux + r, g.n[0] - 1);
                const int ylo = max(uy - r, 0), yhi = min(uy + r, g.n[1] - 1);
                const int zlo = max(uz - r, 0), zhi = min(uz + r, g.n[2] - 1);
                const float clip2 = bestf * 1.0001f;
                float wlo = 0.0f, whi = 0.0f;
                if (CLIP && have) x_window(qx, clip2, wlo, whi);
                if (xlo <= xhi && ylo <= yhi && zlo <= zhi) {
                    // Far searches open many x-rows, most of them short.  Rows are taken G at a time: every lane resolves
                    // ONE row (clipping + its cell_start bounds: G independent loads in flight per query), then the whole
                    // sub-group scans the G ranges one after the other with coalesced 16-byte loads.
                    const int ny_rows = yhi - ylo + 1;
                    const int n_rows = ny_rows * (zhi - zlo + 1);
                    // A far query (no target anywhere near: partial overlap in an unbounded search): once the next shell would open
                    // more rows than the target has points / FAR_DIV, the query is handed to the exhaustive kernel (one tiled pass over
                    // the target shared by hundreds of such queries) with what it has found so far; if the list is full it walks on.
                    if (far_list && (uint32_t)n_rows > nt / FAR_DIV + 256u) {
                        int ok = 0;
                        if (l == 0 && live) {
                            const uint32_t pos = atomicAdd(far_count, 1u);
                            if (pos < far_cap) { far_list[pos] = i; ok = 1; }
                        }
                        ok = __shfl(ok, 0, G);
                        if (!live) ok = 1;               // surplus sub-groups of the last workgroup: nothing to do
                        if (ok) { done = true; break; }
                    }
                    for (int k0 = 0; k0 < n_rows; k0 += G) {
                        const int k = k0 + l;
                        uint32_t b1 = 0, e1 = 0, b2 = 0, e2 = 0;       // up to two pieces per row
                        if (k < n_rows) {
                            const int cy = ylo + k % ny_rows, cz = zlo + k / ny_rows;
                            const int ady = abs(cy - uy), adz = abs(cz - uz);
                            int xa = xlo, xb = xhi;
                            bool open = true;
                            if (best != KEY_NONE) {
                                // (the bound of THIS batch of rows: candidates found in the earlier batches of the stage already count)
                                const float clip2k = fmaxf(__uint_as_float((uint32_t)(best >> 32)), TRUST2) * 1.0001f;
                                const float fy = axis_gap(qy, g.lo[1], g.h, cy, uy, g.slack), fz = axis_gap(qz, g.lo[2], g.h, cz, uz, g.slack);
                                const float rem2 = clip2k - (fy * fy + fz * fz) * 0.9999f;
                                if (rem2 < 0.0f) open = false;                     // the whole row is outside the ball
                                else ball_x_cells(g, qx, rem2, xa, xb);
                            }
                            if (open && xa <= xb) {
                                const uint32_t row = (uint32_t)((cz * g.n[1] + cy) * g.n[0]);
                                if (ady <= rp && adz <= rp) {
                                    // this row crossed the old cube: only the two end pieces are new
                                    const int la = xa, lb = min(xb, ux - rp - 1);
                                    const int ra = max(xa, ux + rp + 1), rb2 = xb;
                                    if (la <= lb) { b1 = cell_start[row + la]; e1 = cell_start[row + lb + 1]; }
                                    if (ra <= rb2) { b2 = cell_start[row + ra]; e2 = cell_start[row + rb2 + 1]; }
                                } else {
                                    b1 = cell_start[row + xa]; e1 = cell_start[row + xb + 1];
                                }
                                if (CLIP && have) {                            // x-sorted rows: keep only |x - qx| <= best
                                    clip_range_x(records, b1, e1, wlo, whi);
                                    clip_range_x(records, b2, e2, wlo, whi);
                                }
                                if (STATS) { st_rows++; if (!SPH) st_cand += (e1 - b1) + (e2 - b2); }
                            }
                        }
                        if (SPH) {
                            // the runs of the (up to) 2 x 16 pieces as one index space: dense batches of 16 sphere tests
                            if (__ballot((b1 < e1) || (b2 < e2)) >> ((threadIdx.x & 63) / 16 * 16) & 0xFFFFull)
                                scan_pieces_sph<STATS>(records, spheres, b1, e1, b2, e2, l, qx, qy, qz, best, bestp, st_cand, st_sph, seed_run);
                        } else {
                            // visit only the lanes that hold a non-empty piece (far searches are mostly empty space)
                            const unsigned long long any = __ballot((b1 < e1) || (b2 < e2));
                            unsigned long long mine = (any >> ((threadIdx.x & 63) / G * G)) & (G == 64 ? ~0ull : ((1ull << (G & 63)) - 1ull));
                            while (mine) {
                                const int j = __builtin_ctzll(mine);
                                mine &= mine - 1;
                                const uint32_t jb1 = __shfl(b1, j, G), je1 = __shfl(e1, j, G);
                                const uint32_t jb2 = __shfl(b2, j, G), je2 = __shfl(e2, j, G);
                                if (jb1 < je1) scan_range<G>(records, jb1, je1, l, qx, qy, qz, best, bestp);
                                if (jb2 < je2) scan_range<G>(records, jb2, je2, l, qx, qy, qz, best, bestp);
                            }
                        }
                    }
                }
                if (STATS && l == 0) st_stages++;
                group_min<G>(best, bestp);
                const bool covers = (ux - r <= 0) && (ux + r >= g.n[0] - 1) && (uy - r <= 0) && (uy + r >= g.n[1] - 1) &&
                                    (uz - r <= 0) && (uz + r >= g.n[2] - 1);
                // every target outside the cube is farther than (r - slack) * h in some axis
                const float reach = ((float)r - g.slack) * g.h;
                // (cap2: the caller discards neighbours with d2 >= cap2, so a cube that no closer target can lie outside of ends the search)
                done = covers || (reach > TRUST && (reach * reach * 0.99999f >= cap2 ||
                                                    (best != KEY_NONE && __uint_as_float((uint32_t)(best >> 32)) < reach * reach * 0.99999f)));
                rp = r;
                r = min(r * 2, 1 << 24);
            }
        }
        if (STATS && live) {
            if (st_cand) atomicAdd(&stats[0], st_cand);
            if (st_rows) atomicAdd(&stats[1], st_rows);
            if (st_sph) atomicAdd(&stats[2], st_sph);
            if (st_stages) atomicAdd(&stats[3], st_stages);
        }
        if (l == 0 && live) {
            const uint32_t bidx = (uint32_t)(best & 0xFFFFFFFFull);
            const uint32_t bits = (bidx == 0xFFFFFFFFu) ? 0x7F800000u : (uint32_t)(best >> 32);
            keys[i] = ((unsigned long long)bits << 32) | bidx;
            if (wpos) wpos[i] = bidx == 0xFFFFFFFFu ? 0xFFFFFFFFu : bestp;
        }
        if (!LIST) break;
    }
    if (!LIST) break;
    }
    if (!queue) break;
    }
}


__global__ __launch_bounds__(GR_BLOCK) void count_nonzero_kernel(const uint32_t* __restrict__ count, uint32_t n, uint32_t* __restrict__ out)
{
    uint32_t c = 0;
    for (uint32_t i = blockIdx.x * GR_BLOCK + threadIdx.x; i < n; i += gridDim.x * GR_BLOCK) c += count[i] != 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

// ---------------------------------------------------------------------------------------- host side
int exclusive_scan_u32(pcr_ctx* ctx, const uint32_t* in, uint32_t* out, size_t n, uint32_t* totals, uint32_t* grand)
{
    const uint32_t nb = (uint32_t)((n + SC_TILE - 1) / SC_TILE);
    hipLaunchKernelGGL(scan_local_kernel, dim3(nb), dim3(GR_BLOCK), 0, ctx->stream, in, out, (uint32_t)n, totals);
    hipLaunchKernelGGL(scan_totals_kernel, dim3(1), dim3(GR_BLOCK), 0, ctx->stream, totals, nb, grand);
    hipLaunchKernelGGL(scan_add_kernel, dim3(nb), dim3(GR_BLOCK), 0, ctx->stream, out, (uint32_t)n, totals);
    PCR_HIP(ctx, hipGetLastError());
    return PCR_OK;
}

// largest finite |coordinate| of the cloud (0 when it has no finite point), cached on the cloud until it is modified
int cloud_absmax(pcr_ctx* ctx, const pcr_cloud* c, float* out)
{
    if (c->absmax >= 0.f) { *out = c->absmax; return PCR_OK; }
    float a = 0.f;
    if (c->n) {
        const uint32_t bb_blocks = (uint32_t)std::min<size_t>(256, (c->n + GR_BLOCK - 1) / GR_BLOCK);
        float* dev = nullptr;
        PCR_HIP(ctx, hipMalloc((void**)&dev, bb_blocks * 6 * sizeof(float)));       // not the shared scratch: callers may be using it
        hipLaunchKernelGGL(bbox_kernel, dim3(bb_blocks), dim3(GR_BLOCK), 0, ctx->stream, c->x(), c->y(), c->z(), (uint32_t)c->n, dev);
        std::vector<float> hb(bb_blocks * 6);
        hipError_t e = hipMemcpyAsync(hb.data(), dev, hb.size() * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        hipFree(dev);
        if (e != hipSuccess) return fail(ctx, PCR_ERR_HIP, "cloud_absmax", e);
        for (uint32_t b = 0; b < bb_blocks; b++)
            for (int k = 0; k < 3; k++) {
                const float lo = hb[b * 6 + k], hi = hb[b * 6 + 3 + k];
                if (lo <= hi) a = std::max(a, std::max(std::fabs(lo), std::fabs(hi)));      // a block without a finite point reports lo > hi
            }
    }
    const_cast<pcr_cloud*>(c)->absmax = a;
    *out = a;
    return PCR_OK;
}

void grid_free(Grid* g)
{
    if (!g) return;
    if (g->chunks) hipFree(g->chunks);
    if (g->spheres) hipFree(g->spheres);
    if (g->by_index) hipFree(g->by_index);
    if (g->records) hipFree(g->records);
    if (g->cell_start) hipFree(g->cell_start);
    delete g;
}

// scratch layout for builds / query sorting: [cell_of n][count cells+1][totals nb+1]
int grid_build(pcr_ctx* ctx, const pcr_cloud* c, Grid** out, double cell_edge, int order)
{
    *out = nullptr;
    const size_t n = c->n;
    if (n > 0x7FFFFFF0ull) return fail(ctx, PCR_ERR_ARG, "grid index: more than 2^31 points");
    Grid* g = new (std::nothrow) Grid();
    if (!g) return fail(ctx, PCR_ERR_NOMEM, "grid");
    g->n_points = n;
    // 1. bounding box of the finite points
    const uint32_t bb_blocks = (uint32_t)std::min<size_t>(256, (n + GR_BLOCK - 1) / GR_BLOCK ? (n + GR_BLOCK - 1) / GR_BLOCK : 1);
    int rc = ensure_scratch(ctx, bb_blocks * 6 * sizeof(float));
    if (rc) { delete g; return rc; }
    float lo[3] = { 0, 0, 0 }, hi[3] = { 0, 0, 0 };
    if (n) {
        hipLaunchKernelGGL(bbox_kernel, dim3(bb_blocks), dim3(GR_BLOCK), 0, ctx->stream, c->x(), c->y(), c->z(), (uint32_t)n, (float*)ctx->scratch);
        std::vector<float> hb(bb_blocks * 6);
        hipError_t e = hipMemcpyAsync(hb.data(), ctx->scratch, hb.size() * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { delete g; return fail(ctx, PCR_ERR_HIP, "bbox", e); }
        for (int k = 0; k < 3; k++) { lo[k] = FLT_MAX; hi[k] = -FLT_MAX; }
        for (uint32_t b = 0; b < bb_blocks; b++)
            for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], hb[b * 6 + k]); hi[k] = std::max(hi[k], hb[b * 6 + 3 + k]); }
        if (lo[0] > hi[0]) { for (int k = 0; k < 3; k++) lo[k] = hi[k] = 0.f; }   // no finite point at all
    }
    // 2. cell size.  First guess ~ 4 n cells in the bounding volume; LiDAR clouds are surfaces, so most of those
    //    cells are empty and the occupied ones are crowded: measure the occupancy and shrink h once so that an
    //    occupied cell holds ~ grid_occupancy_x10 / 10 points (points per occupied cell ~ h^2 on surfaces).
    double ext[3], vol = 1.0;
    for (int k = 0; k < 3; k++) { ext[k] = std::max((double)hi[k] - (double)lo[k], 1e-6); vol *= ext[k]; }
    const double hmin = std::max(std::max(ext[0], ext[1]), ext[2]) / 4000.0;   // <= 4001 cells per axis
    double max_cells = (double)tune_get(ctx, "grid_max_cells", 0);
    if (max_cells <= 0) max_cells = std::min(std::max(64.0 * (double)n, (double)(1 << 22)), (double)(1 << 28));
    auto fit = [&](double h) {
        h = std::max(h, hmin);
        for (;;) {
            double cells = 1.0;
            for (int k = 0; k < 3; k++) cells *= std::floor(ext[k] / h) + 2.0;
            if (cells <= max_cells) return h;
            h *= 1.1;
        }
    };
    auto set_params = [&](double h) {
        g->p.h = (float)h;
        g->p.inv_h = 1.0f / g->p.h;
        size_t cells = 1;
        for (int k = 0; k < 3; k++) {
            g->p.lo[k] = lo[k];
            g->p.n[k] = (int)std::floor(((double)hi[k] - (double)lo[k]) * (double)g->p.inv_h) + 2;   // +1 covers rounding at the top face
            cells *= (size_t)g->p.n[k];
        }
        g->p.slack = 0.01f + 2e-6f * (float)std::max(std::max(g->p.n[0], g->p.n[1]), g->p.n[2]);
        g->n_cells = cells;
    };
    const int64_t user_um = cell_edge > 0.0 ? (int64_t)1 : tune_get(ctx, "grid_cell_um", 0);
    double h = fit(cell_edge > 0.0 ? cell_edge : user_um > 0 ? (double)user_um * 1e-6 : std::cbrt(vol / (4.0 * (double)std::max<size_t>(n, 1))));
    set_params(h);
    if (user_um <= 0 && n >= 1024) {
        const size_t cells0 = g->n_cells;
        const size_t off_count0 = ((n * 4 + 255) & ~(size_t)255);
        rc = ensure_scratch(ctx, off_count0 + (cells0 + 1) * 4 + 512);
        if (rc) { delete g; return rc; }
        uint32_t* cell_of0 = (uint32_t*)ctx->scratch;
        uint32_t* count0 = (uint32_t*)((char*)ctx->scratch + off_count0);
        uint32_t* nz = count0 + cells0 + 1;
        hipError_t e0 = hipMemsetAsync(count0, 0, (cells0 + 2) * 4, ctx->stream);
        if (e0 != hipSuccess) { delete g; return fail(ctx, PCR_ERR_HIP, "memset(grid)", e0); }
        hipLaunchKernelGGL(cell_count_kernel, dim3((unsigned)((n + GR_BLOCK - 1) / GR_BLOCK)), dim3(GR_BLOCK), 0, ctx->stream,
                           c->x(), c->y(), c->z(), (uint32_t)n, g->p, cell_of0, count0, 0u);
        hipLaunchKernelGGL(count_nonzero_kernel, dim3(256), dim3(GR_BLOCK), 0, ctx->stream, count0, (uint32_t)cells0, nz);
        uint32_t occupied = 0;
        e0 = hipMemcpyAsync(&occupied, nz, 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e0 == hipSuccess) e0 = hipStreamSynchronize(ctx->stream);
        if (e0 != hipSuccess) { delete g; return fail(ctx, PCR_ERR_HIP, "grid occupancy", e0); }
        const double occ = (double)n / (double)std::max<uint32_t>(occupied, 1);
        const double want = (double)tune_get(ctx, "grid_occupancy_x10", 20) / 10.0;
        if (occ > 1.5 * want) {
            h = fit(h * std::sqrt(want / occ));
            set_params(h);
        }
    }
    if (order == GRID_ORDER_MORTON && user_um <= 0 && cell_edge <= 0.0) {
        // The sphere walk prunes INSIDE a cell (bounding spheres over Morton-ordered records), so its cells can be larger than the
        // occupancy rule wants: fewer x-rows for the queries that are still far from their neighbour (the first iterations of an
        // ICP) at the price of a few more sphere tests near convergence.  Measured at 10 M (profiles/r02_c5_grid_ab.txt, same call):
        // x1.0 12.1, x1.5 11.5, x2 11.8 (with a second sphere level), x3 12.0, x4.5 14.0 ms per ICP iteration.
        // A second level of spheres (one per 16 runs) was built and measured too: its extra dependent round trip costs more than the
        // sphere tests it saves at every cell size (x1.0: 12.1 -> 14.4, x1.5: 11.5 -> 12.4 ms) — dropped.
        const double scale = (double)tune_get(ctx, "grid_cell_scale_x100", 150) / 100.0;
        if (scale > 1.0) { h = fit(h * scale); set_params(h); }
    }
    const size_t cells = g->n_cells;
    // 3. sort by (cell, x).  Histogram + scan give cell_start; one radix sort of (cell << 32 | x bits, index) gives the
    //    order (deterministic: no atomics decide a position).  Non-finite points go to the extra cell `cells`.
    const size_t ncell = cells + 2;                                  // real cells, the non-finite cell, the end sentinel
    const size_t nb = (ncell + SC_TILE - 1) / SC_TILE;
    int key_bits = 1;
    while (((size_t)1 << key_bits) < cells + 1) key_bits++;
    size_t temp_bytes = 0;
    sort_pairs_u64_u32(nullptr, temp_bytes, nullptr, nullptr, nullptr, nullptr, n, 0, 32 + key_bits, ctx->stream);
    const size_t a4 = (n * 4 + 255) & ~(size_t)255, a8 = (n * 8 + 255) & ~(size_t)255;
    const size_t off_count = a4, off_tot = off_count + ((ncell * 4 + 255) & ~(size_t)255), off_kin = off_tot + (((nb + 2) * 4 + 255) & ~(size_t)255),
                 off_kout = off_kin + a8, off_vin = off_kout + a8, off_vout = off_vin + a4, off_temp = off_vout + a4;
    rc = ensure_scratch(ctx, off_temp + temp_bytes + 256);
    if (rc) { delete g; return rc; }
    char* sc = (char*)ctx->scratch;
    uint32_t* cell_of = (uint32_t*)sc;
    uint32_t* count = (uint32_t*)(sc + off_count);
    uint32_t* totals = (uint32_t*)(sc + off_tot);
    unsigned long long* k_in = (unsigned long long*)(sc + off_kin);
    unsigned long long* k_out = (unsigned long long*)(sc + off_kout);
    uint32_t* v_in = (uint32_t*)(sc + off_vin);
    uint32_t* v_out = (uint32_t*)(sc + off_vout);
    hipError_t e = hipMalloc((void**)&g->cell_start, ncell * sizeof(uint32_t));
    const size_t n_padded = (n + GRID_CHUNK - 1) / GRID_CHUNK * GRID_CHUNK;
    if (e == hipSuccess) e = hipMalloc((void**)&g->records, std::max<size_t>(n_padded, GRID_CHUNK) * sizeof(float4));
    if (e != hipSuccess) { grid_free(g); return fail(ctx, PCR_ERR_HIP, "hipMalloc(grid)", e); }
    e = hipMemsetAsync(count, 0, ncell * 4, ctx->stream);
    if (e != hipSuccess) { grid_free(g); return fail(ctx, PCR_ERR_HIP, "memset(grid)", e); }
    const dim3 gridn((unsigned)((n + GR_BLOCK - 1) / GR_BLOCK));
    if (n) hipLaunchKernelGGL(cell_count_kernel, gridn, dim3(GR_BLOCK), 0, ctx->stream, c->x(), c->y(), c->z(), (uint32_t)n, g->p, cell_of, count, (uint32_t)cells);
    rc = exclusive_scan_u32(ctx, count, g->cell_start, ncell, totals, totals + nb);
    if (rc) { grid_free(g); return rc; }
    if (n) {
        // occupied cells of the final grid (the tile search compares the working cloud's density with the target's: launch_nn1_grid).
        // totals[nb + 1] is a free word of the scan's workspace; the value reaches the host with the next synchronisation.
        uint32_t* nzf = totals + nb + 1;
        e = hipMemsetAsync(nzf, 0, 4, ctx->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(count_nonzero_kernel, dim3(256), dim3(GR_BLOCK), 0, ctx->stream, count, (uint32_t)cells, nzf);
            e = pin_word_begin(ctx, 1, nzf);
            g->occupied = 0; g->occupied_tag = ctx->pin_gen;
        }
        if (e != hipSuccess) { grid_free(g); return fail(ctx, PCR_ERR_HIP, "grid occupancy", e); }
    }
    if (n) {
        g->x_sorted = order != GRID_ORDER_MORTON;
        if (g->x_sorted)
            hipLaunchKernelGGL(record_keys_kernel<false>, gridn, dim3(GR_BLOCK), 0, ctx->stream, c->x(), c->y(), c->z(), (uint32_t)n, g->p, cell_of, k_in, v_in);
        else
            hipLaunchKernelGGL(record_keys_kernel<true>, gridn, dim3(GR_BLOCK), 0, ctx->stream, c->x(), c->y(), c->z(), (uint32_t)n, g->p, cell_of, k_in, v_in);
        e = sort_pairs_u64_u32(sc + off_temp, temp_bytes, k_in, k_out, v_in, v_out, n, 0, 32 + key_bits, ctx->stream);
        if (e != hipSuccess) { grid_free(g); return fail(ctx, PCR_ERR_HIP, "radix sort(grid)", e); }
        hipLaunchKernelGGL(gather_records_kernel, dim3((unsigned)((n_padded + GR_BLOCK - 1) / GR_BLOCK)), dim3(GR_BLOCK), 0, ctx->stream, c->x(), c->y(), c->z(),
                           (uint32_t)n, (uint32_t)n_padded, v_out, g->records);
    }
    // chunked, centred copy for the expanded-form brute-force filter (17 B per point)
    g->n_chunks = (n + GRID_CHUNK - 1) / GRID_CHUNK;
    int* unsafe_dev = (int*)count;                                  // the histogram is no longer needed
    int unsafe_host = 1;
    if (g->n_chunks) {
        e = hipMalloc((void**)&g->chunks, g->n_chunks * GRID_CHUNK_FLOATS * sizeof(float));
        if (e != hipSuccess) { grid_free(g); return fail(ctx, PCR_ERR_HIP, "hipMalloc(grid chunks)", e); }
        e = hipMemsetAsync(unsafe_dev, 0, 4, ctx->stream);
        if (e != hipSuccess) { grid_free(g); return fail(ctx, PCR_ERR_HIP, "memset(grid)", e); }
        hipLaunchKernelGGL(build_chunks_kernel, dim3((unsigned)((g->n_chunks + GR_BLOCK - 1) / GR_BLOCK)), dim3(GR_BLOCK), 0, ctx->stream, g->records, (uint32_t)n,
                           (uint32_t)g->n_chunks, g->chunks, unsafe_dev);
        e = hipMalloc((void**)&g->spheres, g->n_chunks * sizeof(float4));
        if (e != hipSuccess) { grid_free(g); return fail(ctx, PCR_ERR_HIP, "hipMalloc(grid spheres)", e); }
        hipLaunchKernelGGL(build_spheres_kernel, dim3((unsigned)((g->n_chunks + GR_BLOCK - 1) / GR_BLOCK)), dim3(GR_BLOCK), 0, ctx->stream, g->records,
                           (uint32_t)g->n_chunks, g->spheres);
        e = hipMemcpyAsync(&unsafe_host, unsafe_dev, 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e != hipSuccess) { grid_free(g); return fail(ctx, PCR_ERR_HIP, "grid chunks", e); }
    }
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);   // the scratch is reused by the caller right away
    if (e != hipSuccess) { grid_free(g); return fail(ctx, PCR_ERR_HIP, "grid build", e); }
    g->chunk_safe = g->n_chunks != 0 && unsafe_host == 0;
    *out = g;
    return PCR_OK;
}

// group the queries by a COARSE cell of the target grid (>= 2 x 2 x 2 fine cells, at most 4 M bins): makes the queries of
// a wave neighbours in space; finer bins keep the working set of an XCD inside its L2 at 10 M points (measured:
// 30.6 -> 26.0 ms per search with 2^22 instead of 2^17 bins, profiles/r01_c5_10M_single_gpu.txt).  perm[] goes to ctx->qperm.
static int sort_queries(pcr_ctx* ctx, const Grid* g, const pcr_cloud* src)
{
    ctx->work_cells = 0; ctx->pin_pending[0] = false;     // (coarse bins: the density of the order is unknown)
    GridParams cp = g->p;
    int f = (int)tune_get(ctx, "grid_query_bin_min", 2);
    for (;;) {
        size_t cells = 1;
        for (int k = 0; k < 3; k++) cells *= (size_t)((g->p.n[k] + f - 1) / f + 1);
        if (cells <= ((size_t)1 << tune_get(ctx, "grid_query_bins_log2", 22)) || f >= 4096) break;
        f *= 2;
    }
    cp.h = g->p.h * (float)f;
    cp.inv_h = 1.0f / cp.h;
    size_t cells = 1;
    for (int k = 0; k < 3; k++) { cp.n[k] = (g->p.n[k] + f - 1) / f + 1; cells *= (size_t)cp.n[k]; }
    const size_t n = src->n;
    const size_t nb = (cells + 1 + SC_TILE - 1) / SC_TILE;
    const size_t off_count = ((n * 4 + 255) & ~(size_t)255);
    const size_t off_start = off_count + (((cells + 1) * 4 + 255) & ~(size_t)255);
    const size_t off_tot = off_start + (((cells + 1) * 4 + 255) & ~(size_t)255);
    int rc = ensure_scratch(ctx, off_tot + (nb + 2) * 4 + 256);
    if (rc) return rc;
    if (ctx->qperm_cap < n) {
        if (ctx->qperm) PCR_HIP(ctx, hipFree(ctx->qperm));
        ctx->qperm = nullptr; ctx->qperm_cap = 0;
        PCR_HIP(ctx, hipMalloc((void**)&ctx->qperm, padded(n) * sizeof(uint32_t)));
        ctx->qperm_cap = padded(n);
    }
    uint32_t* cell_of = (uint32_t*)ctx->scratch;
    uint32_t* count = (uint32_t*)((char*)ctx->scratch + off_count);
    uint32_t* start = (uint32_t*)((char*)ctx->scratch + off_start);
    uint32_t* totals = (uint32_t*)((char*)ctx->scratch + off_tot);
    PCR_HIP(ctx, hipMemsetAsync(count, 0, (cells + 1) * 4, ctx->stream));
    const dim3 gridn((unsigned)((n + GR_BLOCK - 1) / GR_BLOCK));
    hipLaunchKernelGGL(cell_count_kernel, gridn, dim3(GR_BLOCK), 0, ctx->stream, src->x(), src->y(), src->z(), (uint32_t)n, cp, cell_of, count, 0u);
    rc = exclusive_scan_u32(ctx, count, start, cells + 1, totals, totals + nb);
    if (rc) return rc;
    PCR_HIP(ctx, hipMemsetAsync(count, 0, (cells + 1) * 4, ctx->stream));
    hipLaunchKernelGGL(scatter_perm_kernel, gridn, dim3(GR_BLOCK), 0, ctx->stream, (uint32_t)n, cell_of, start, count, ctx->qperm);
    PCR_HIP(ctx, hipGetLastError());
    ctx->qperm_n = n;
    ctx->qperm_src = src;
    return PCR_OK;
}

// key of a query in the ORDER OF THE RECORDS: (cell of the target's grid, clamped) << 24 | 24-bit Morton code of the position inside the
// cell (morton24_in_cell: what the records of a Morton-ordered index are sorted by).  Non-finite queries go last.
constexpr int QKEY_SUB_BITS = 24;
__global__ __launch_bounds__(GR_BLOCK) void query_keys_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, uint32_t n,
                                                              GridParams g, uint32_t n_cells, unsigned long long* __restrict__ keys,
                                                              uint32_t* __restrict__ vals)
{
    const uint32_t i = blockIdx.x * GR_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float px = x[i], py = y[i], pz = z[i];
    unsigned long long key = (unsigned long long)n_cells << QKEY_SUB_BITS;
    if (finite3(px, py, pz)) {
        const float p[3] = { px, py, pz };
        int c[3];
        float f[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            c[k] = min(max(cell_coord(p[k], g.lo[k], g.inv_h), 0), g.n[k] - 1);
            f[k] = (p[k] - g.lo[k]) * g.inv_h - (float)c[k];                       // < 0 or >= 1 for a query outside the grid box (clamped below)
        }
        key = ((unsigned long long)((c[2] * g.n[1] + c[1]) * g.n[0] + c[0]) << QKEY_SUB_BITS) | morton24_in_cell(f[0], f[1], f[2]);
    }
    keys[i] = key;
    vals[i] = i;
}

// number of distinct values of key >> shift in a sorted key array
__global__ __launch_bounds__(GR_BLOCK) void count_key_runs_kernel(const unsigned long long* __restrict__ keys, uint32_t n, int shift, uint32_t* __restrict__ out)
{
    uint32_t c = 0;
    for (uint32_t i = blockIdx.x * GR_BLOCK + threadIdx.x; i < n; i += gridDim.x * GR_BLOCK) c += (i == 0 || (keys[i] >> shift) != (keys[i - 1] >> shift)) ? 1u : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

// perm[] (-> ctx->qperm) = the queries in the order of the target's records: consecutive queries are neighbours in space at the
// resolution of an eighth of a cell, so that the 16 queries of a workgroup open the same rows, test the same spheres and scan
// the same runs (the coarse bins of sort_queries leave the order inside a bin to the atomics).  One radix sort of the whole
// cloud: worth it for a working cloud that is searched many times (grid_sort_working_cloud).
static int sort_queries_fine(pcr_ctx* ctx, const Grid* g, const pcr_cloud* src)
{
    const size_t n = src->n;
    int key_bits = 1;
    while (((size_t)1 << key_bits) < g->n_cells + 1) key_bits++;
    // The tile search (targets from 4 M points on) wants its queries in the full 24-bit Morton order inside a cell and needs to know how
    // densely they sit (work_cells); the sphere walk of smaller targets is served as well by 12 of those bits — two radix passes and a
    // counting kernel + read-back less per loop (120 k: 190 -> ~150 us of fixed work per ICP call)
    const bool tile_possible = g->n_points >= 4000000 || tune_get(ctx, "grid_tile", 0) == 1;
    const int begin_bit = tile_possible ? 0 : QKEY_SUB_BITS - 12;
    size_t temp_bytes = 0;
    sort_pairs_u64_u32(nullptr, temp_bytes, nullptr, nullptr, nullptr, nullptr, n, begin_bit, key_bits + QKEY_SUB_BITS, ctx->stream);
    const size_t a4 = (n * 4 + 255) & ~(size_t)255, a8 = (n * 8 + 255) & ~(size_t)255;
    int rc = ensure_scratch(ctx, 2 * a8 + a4 + temp_bytes + 256);
    if (rc) return rc;
    if (ctx->qperm_cap < n) {
        if (ctx->qperm) PCR_HIP(ctx, hipFree(ctx->qperm));
        ctx->qperm = nullptr; ctx->qperm_cap = 0;
        PCR_HIP(ctx, hipMalloc((void**)&ctx->qperm, padded(n) * sizeof(uint32_t)));
        ctx->qperm_cap = padded(n);
    }
    char* sc = (char*)ctx->scratch;
    unsigned long long* k_in = (unsigned long long*)sc;
    unsigned long long* k_out = (unsigned long long*)(sc + a8);
    uint32_t* v_in = (uint32_t*)(sc + 2 * a8);
    hipLaunchKernelGGL(query_keys_kernel, dim3((unsigned)((n + GR_BLOCK - 1) / GR_BLOCK)), dim3(GR_BLOCK), 0, ctx->stream, src->x(), src->y(), src->z(), (uint32_t)n,
                       g->p, (uint32_t)g->n_cells, k_in, v_in);
    hipError_t e = sort_pairs_u64_u32(sc + 2 * a8 + a4, temp_bytes, k_in, k_out, v_in, ctx->qperm, n, begin_bit, key_bits + QKEY_SUB_BITS, ctx->stream);
    if (e != hipSuccess) return fail(ctx, PCR_ERR_HIP, "radix sort(queries)", e);
    PCR_HIP(ctx, hipGetLastError());
    // how many cells of the target's grid the queries occupy (-> ctx->work_cells with the caller's next synchronisation; the sort's
    // input keys are dead by now: their first word is the counter)
    uint32_t* runs = (uint32_t*)k_in;
    ctx->work_cells = 0; ctx->pin_pending[0] = false;
    if (tile_possible) {
        PCR_HIP(ctx, hipMemsetAsync(runs, 0, 4, ctx->stream));
        hipLaunchKernelGGL(count_key_runs_kernel, dim3(256), dim3(GR_BLOCK), 0, ctx->stream, k_out, (uint32_t)n, QKEY_SUB_BITS, runs);
        PCR_HIP(ctx, pin_word_begin(ctx, 0, runs));
    }
    ctx->qperm_n = n;
    ctx->qperm_src = src;
    return PCR_OK;
}

// ---- BTRACK index (see nn1_brute.hip for the operand layout and the error analysis)
// Morton key of a point on a lattice of cubic cells over the bounding box (lo, inv = 1024 / longest extent); non-finite points last
__global__ __launch_bounds__(GR_BLOCK) void bt_keys_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, uint32_t n,
                                                           float lox, float loy, float loz, float ivx, float ivy, float ivz,
                                                           unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals, int fine_bits)
{
    const uint32_t i = blockIdx.x * GR_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float px = x[i], py = y[i], pz = z[i];
    unsigned long long key = 1ull << (30 + 3 * fine_bits);
    if (finite3(px, py, pz)) {
        const float f[3] = { fminf(fmaxf((px - lox) * ivx, 0.0f), 1023.0f), fminf(fmaxf((py - loy) * ivy, 0.0f), 1023.0f), fminf(fmaxf((pz - loz) * ivz, 0.0f), 1023.0f) };
        const uint32_t c[3] = { (uint32_t)f[0], (uint32_t)f[1], (uint32_t)f[2] };
        key = 0;
        for (int b = 0; b < 10; b++)
            for (int k = 0; k < 3; k++) key |= (unsigned long long)((c[k] >> b) & 1u) << (3 * b + k);
        if (fine_bits > 0) {
            // large targets: the position INSIDE the lattice cell refines the order (fine_bits per axis) — a cell of a 10 M-point scan holds
            // hundreds of points, and runs of 16 records in arrival order are as wide as the cell (their bounding spheres prune nothing)
            const float sc = (float)(1u << fine_bits);
            unsigned long long sub = 0;
            for (int k = 0; k < 3; k++) {
                const uint32_t s = min((uint32_t)((f[k] - (float)c[k]) * sc), (1u << fine_bits) - 1u);
                for (int b = 0; b < fine_bits; b++) sub |= (unsigned long long)((s >> b) & 1u) << (3 * b + k);
            }
            key = (key << (3 * fine_bits)) | sub;
        }
    }
    keys[i] = key;
    vals[i] = i;
}

// one WAVE per super-tile: centre = mean of its finite records (lane l holds records l, l + 64, ...; butterfly sums — a fixed order, so the
// centre is a pure function of the records), then the largest offset from it.  (One THREAD per super-tile took 74 us at 120 k points:
// 469 threads walking 256 records twice — a tenth of a fresh target's first search.)
__global__ __launch_bounds__(GR_BLOCK) void bt_centres_kernel(const float4* __restrict__ rec, uint32_t n_super, float4* __restrict__ centres, int* __restrict__ bad16)
{
    static_assert(BT_SUPER % 64 == 0 && GR_BLOCK % 64 == 0, "whole waves");
    constexpr int PER = BT_SUPER / 64;
    const uint32_t s = blockIdx.x * (GR_BLOCK / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (s >= n_super) return;                                 // (wave-uniform)
    float4 r[PER];
    bool fin[PER];
    float cx = 0.f, cy = 0.f, cz = 0.f, cnt = 0.f;
#pragma unroll
    for (int u = 0; u < PER; u++) {
        r[u] = rec[(size_t)s * BT_SUPER + u * 64 + lane];
        fin[u] = finite3(r[u].x, r[u].y, r[u].z);
        if (fin[u]) { cx += r[u].x; cy += r[u].y; cz += r[u].z; cnt += 1.0f; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { cx += __shfl_xor(cx, o, 64); cy += __shfl_xor(cy, o, 64); cz += __shfl_xor(cz, o, 64); cnt += __shfl_xor(cnt, o, 64); }
    if (cnt > 0.f) { cx /= cnt; cy /= cnt; cz /= cnt; }
    // .w: the f16 form's scale, a power of two with |t - C|_inf * scale <= 2^7 (exponent clamped to [-60, 60]: bad16 when it had to be)
    float rho = 0.f;
#pragma unroll
    for (int u = 0; u < PER; u++)
        if (fin[u]) rho = fmaxf(rho, fmaxf(fmaxf(fabsf(r[u].x - cx), fabsf(r[u].y - cy)), fabsf(r[u].z - cz)));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) rho = fmaxf(rho, __shfl_xor(rho, o, 64));
    if (lane != 0) return;
    int k = 0;
    if (rho > 0.f) {
        int e2;
        (void)frexpf(rho, &e2);                            // rho = m 2^e2, m in [0.5, 1)  ->  rho <= 2^e2
        k = 7 - e2;
        if (k < -60 || k > 60) { if (bad16) atomicOr(bad16, 1); k = k < 0 ? -60 : 60; }
    }
    centres[s] = make_float4(cx, cy, cz, ldexpf(1.0f, k));
}

// HTRACK operands: one thread per MFMA row (record) of a tile.  K slots of a lane: lanes < 32  [x: (1,1) (1,2) (2,1) (2,2) | y: the same],
// lanes >= 32  [z: the same | w1, w2, 0, 0]  against the query side [r1, r1, r2, r2] per coordinate and [1, 1, 0, 0];  A holds the
// two f16 pieces of -2 t'' scale (round to zero, then the remainder) and of w = |t'' scale|^2 (1 - 2^-17).
__global__ __launch_bounds__(GR_BLOCK) void bt_ops16_kernel(const float4* __restrict__ rec, uint32_t n_tiles, const float4* __restrict__ centres,
                                                            uint4* __restrict__ ops16)
{
    const uint32_t gid = blockIdx.x * GR_BLOCK + threadIdx.x;
    const uint32_t T = gid >> 5, m = gid & 31;
    if (T >= n_tiles) return;
    const uint32_t p = T * 32 + 16 * ((m >> 2) & 1) + 4 * (m >> 3) + (m & 3);
    const float4 r = rec[p];
    const float4 C = centres[T / (BT_SUPER / 32)];
    const bool fin = finite3(r.x, r.y, r.z);
    const float tx = (r.x - C.x) * C.w, ty = (r.y - C.y) * C.w, tz = (r.z - C.z) * C.w;         // exact scaling, |.| <= 2^7 (unless bad16)
    uint4* o = ops16 + (size_t)T * 64;
    o[m] = ht_target_operand(tx, ty, tz, fin, false);
    o[32 + m] = ht_target_operand(tx, ty, tz, fin, true);
}

// v = p1 + p2 + p3 exactly, every piece a bf16 value (top 16 bits of an f32); finite v
__device__ __forceinline__ void bf16_split3(float v, uint32_t& p1, uint32_t& p2, uint32_t& p3)
{
    const float a = __uint_as_float(__float_as_uint(v) & 0xFFFF0000u);
    const float d = v - a;                                                 // exact
    const float b = __uint_as_float(__float_as_uint(d) & 0xFFFF0000u);
    const float e = d - b;                                                 // exact, <= 8 significant bits
    p1 = __float_as_uint(a) >> 16; p2 = __float_as_uint(b) >> 16; p3 = __float_as_uint(e) >> 16;
}

// one thread per MFMA row (record) of a tile: the operands of its two lanes
__global__ __launch_bounds__(GR_BLOCK) void bt_ops_kernel(const float4* __restrict__ rec, uint32_t n_tiles, const float4* __restrict__ centres,
                                                          uint4* __restrict__ ops, int* __restrict__ unsafe)
{
    const uint32_t gid = blockIdx.x * GR_BLOCK + threadIdx.x;
    const uint32_t T = gid >> 5, m = gid & 31;
    if (T >= n_tiles) return;
    const uint32_t p = T * 32 + 16 * ((m >> 2) & 1) + 4 * (m >> 3) + (m & 3);   // MFMA row m <-> record (nn1_brute.hip)
    const float4 r = rec[p];
    const float4 C = centres[T / (BT_SUPER / 32)];
    uint32_t t[3][3] = { { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 } }, w[3] = { 0x7F80u, 0, 0 };      // padding / non-finite: w = +inf, never the minimum
    bool bad = !finite3(C.x, C.y, C.z) || fabsf(C.x) > 1e18f || fabsf(C.y) > 1e18f || fabsf(C.z) > 1e18f;
    if (finite3(r.x, r.y, r.z)) {
        const float tx = r.x - C.x, ty = r.y - C.y, tz = r.z - C.z;
        const float ww = ((tx * tx + ty * ty) + tz * tz) * 0.99999237060546875f;               // (1 - 2^-17)
        if (!(fabsf(tx) < 1e18f && fabsf(ty) < 1e18f && fabsf(tz) < 1e18f) || !(ww < 3e38f)) bad = true;
        else {
            bf16_split3(-2.0f * tx, t[0][0], t[0][1], t[0][2]);
            bf16_split3(-2.0f * ty, t[1][0], t[1][1], t[1][2]);
            bf16_split3(-2.0f * tz, t[2][0], t[2][1], t[2][2]);
            bf16_split3(ww, w[0], w[1], w[2]);
        }
    }
    // K slots of one coordinate, against the query pieces [r1, r1, r2, r1, r2, r3, r2, r3]:  [t1, t2, t1, t3, t2, t1, t3, t2]
    auto pack = [](const uint32_t (&q)[3]) {
        return make_uint4(q[0] | (q[1] << 16), q[0] | (q[2] << 16), q[1] | (q[0] << 16), q[2] | (q[1] << 16));
    };
    uint4* o = ops + (size_t)T * 128;
    o[m] = pack(t[0]);                                                     // instruction 0, lanes < 32: x
    o[32 + m] = pack(t[1]);                                                // instruction 0, lanes >= 32: y
    o[64 + m] = pack(t[2]);                                                // instruction 1, lanes < 32: z
    o[96 + m] = make_uint4(w[0] | (w[1] << 16), 0u | (w[2] << 16), 0u, 0u);   // instruction 1, lanes >= 32: [w1, w2, 0, w3, 0, 0, 0, 0] against [1, 1, 0, 1, 0, 0, 0, 0]
    if (bad && unsafe) atomicOr(unsafe, 1);
}

void bt_free(BtIndex* b)
{
    if (!b) return;
    if (b->block) hipFree(b->block);
    if (b->tile_block) hipFree(b->tile_block);
    if (b->l1_block) hipFree(b->l1_block);
    delete b;
}

// ---- level 1 of the two-level sign filter (grid_common.hpp: the statement; nn1_sphere.hpp: the search)
// one workgroup per level-1 super-tile: centre = mean of its finite records, scale = the power of two with |t - C|_inf scale <= 2^7
__global__ __launch_bounds__(GR_BLOCK) void bt_l1_centres_kernel(const float4* __restrict__ rec, uint32_t n_rec, uint32_t n_l1, float4* __restrict__ centres, int* __restrict__ bad)
{
    __shared__ float red[4][GR_BLOCK / 64];
    const uint32_t s = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (s >= n_l1) return;
    constexpr int PER = BT_L1_SUPER / GR_BLOCK;
    float cx = 0.f, cy = 0.f, cz = 0.f, cnt = 0.f;
    float4 r[PER];
    bool fin[PER];
#pragma unroll
    for (int u = 0; u < PER; u++) {
        const size_t p = (size_t)s * BT_L1_SUPER + (size_t)u * GR_BLOCK + threadIdx.x;
        r[u] = p < n_rec ? rec[p] : make_float4(__builtin_inff(), 0.f, 0.f, 0.f);
        fin[u] = finite3(r[u].x, r[u].y, r[u].z);
        if (fin[u]) { cx += r[u].x; cy += r[u].y; cz += r[u].z; cnt += 1.0f; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { cx += __shfl_xor(cx, o, 64); cy += __shfl_xor(cy, o, 64); cz += __shfl_xor(cz, o, 64); cnt += __shfl_xor(cnt, o, 64); }
    if (lane == 0) { red[0][wave] = cx; red[1][wave] = cy; red[2][wave] = cz; red[3][wave] = cnt; }
    __syncthreads();
    cx = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]); cy = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    cz = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]); cnt = (red[3][0] + red[3][1]) + (red[3][2] + red[3][3]);
    if (cnt > 0.f) { cx /= cnt; cy /= cnt; cz /= cnt; }
    float rho = 0.f;
#pragma unroll
    for (int u = 0; u < PER; u++)
        if (fin[u]) rho = fmaxf(rho, fmaxf(fmaxf(fabsf(r[u].x - cx), fabsf(r[u].y - cy)), fabsf(r[u].z - cz)));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) rho = fmaxf(rho, __shfl_xor(rho, o, 64));
    __syncthreads();
    if (lane == 0) red[0][wave] = rho;
    __syncthreads();
    if (threadIdx.x != 0) return;
    rho = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
    int k = 0;
    if (rho > 0.f) {
        int e2;
        (void)frexpf(rho, &e2);                            // rho <= 2^e2
        k = 7 - e2;
        if (k < -60 || k > 60) { atomicOr(bad, 1); k = k < 0 ? -60 : 60; }
    }
    centres[s] = make_float4(cx, cy, cz, ldexpf(1.0f, k));
}

// one thread per chunk of 16 records: its row of its level-1 tile (MFMA row m <-> chunk 16 ((m >> 2) & 1) + 4 (m >> 3) + (m & 3), so that
// the 16 accumulators of lane-half h are the chunks 16 h + i of the tile, in order)
__global__ __launch_bounds__(GR_BLOCK) void bt_l1_ops_kernel(const float4* __restrict__ rec, uint32_t n_rec, uint32_t n_l1_tiles, const float4* __restrict__ centres,
                                                             uint4* __restrict__ ops)
{
    const uint32_t gid = blockIdx.x * GR_BLOCK + threadIdx.x;
    const uint32_t T = gid >> 5, m = gid & 31;
    if (T >= n_l1_tiles) return;
    const uint32_t chunk = T * 32 + 16 * ((m >> 2) & 1) + 4 * (m >> 3) + (m & 3);
    const float4 C = centres[T / (BT_L1_SUPER / 512)];
    float tx[16], ty[16], tz[16];
    bool fin[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const size_t p = (size_t)chunk * 16 + j;
        const float4 r = p < n_rec ? rec[p] : make_float4(__builtin_inff(), 0.f, 0.f, 0.f);
        tx[j] = (r.x - C.x) * C.w; ty[j] = (r.y - C.y) * C.w; tz[j] = (r.z - C.z) * C.w;       // exact scaling, |.| <= 2^7 (unless the super-tile is flagged bad)
        fin[j] = finite3(r.x, r.y, r.z) && fabsf(tx[j]) <= 128.0f && fabsf(ty[j]) <= 128.0f && fabsf(tz[j]) <= 128.0f;
    }
    uint4 lo, hi;
    l1_chunk_operand(tx, ty, tz, fin, lo, hi);
    ops[(size_t)T * 64 + m] = lo;
    ops[(size_t)T * 64 + 32 + m] = hi;
}

// the per-record operands in the scale of the record's LEVEL-1 super-tile (bt_ops16_kernel with the level-1 centres: 128 tiles per super-tile)
__global__ __launch_bounds__(GR_BLOCK) void bt_l1_rec_ops_kernel(const float4* __restrict__ rec, uint32_t n_tiles, const float4* __restrict__ centres, uint4* __restrict__ ops)
{
    const uint32_t gid = blockIdx.x * GR_BLOCK + threadIdx.x;
    const uint32_t T = gid >> 5, m = gid & 31;
    if (T >= n_tiles) return;
    const uint32_t p = T * 32 + 16 * ((m >> 2) & 1) + 4 * (m >> 3) + (m & 3);
    const float4 r = rec[p];
    const float4 C = centres[T / (BT_L1_SUPER / 32)];
    const float tx = (r.x - C.x) * C.w, ty = (r.y - C.y) * C.w, tz = (r.z - C.z) * C.w;         // exact scaling, |.| <= 2^7 (unless the super-tile is flagged bad)
    const bool fin = finite3(r.x, r.y, r.z) && fabsf(tx) <= 128.0f && fabsf(ty) <= 128.0f && fabsf(tz) <= 128.0f;
    ops[(size_t)T * 64 + m] = ht_target_operand(tx, ty, tz, fin, false);
    ops[(size_t)T * 64 + 32 + m] = ht_target_operand(tx, ty, tz, fin, true);
}

// ---- level 0 (STRACK3): one row per level-1 tile = the bounding sphere of its 512 records, in the scale of a level-0 super-tile of
// BT_L0_SUPER records.  Centres: one workgroup per level-0 super-tile, two passes over its records (mean of the finite ones, then the
// largest |.|_inf distance from it -> the power-of-two scale with every record inside 2^7 units)
__global__ __launch_bounds__(GR_BLOCK) void bt_l0_centres_kernel(const float4* __restrict__ rec, uint32_t n_rec, uint32_t n_l0, float4* __restrict__ centres, int* __restrict__ bad)
{
    __shared__ float red[4][GR_BLOCK / 64];
    const uint32_t s = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (s >= n_l0) return;
    const size_t base = (size_t)s * BT_L0_SUPER, end = min(base + (size_t)BT_L0_SUPER, (size_t)n_rec);
    float cx = 0.f, cy = 0.f, cz = 0.f, cnt = 0.f;
    for (size_t p = base + threadIdx.x; p < end; p += GR_BLOCK) {
        const float4 r = rec[p];
        if (finite3(r.x, r.y, r.z)) { cx += r.x; cy += r.y; cz += r.z; cnt += 1.0f; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { cx += __shfl_xor(cx, o, 64); cy += __shfl_xor(cy, o, 64); cz += __shfl_xor(cz, o, 64); cnt += __shfl_xor(cnt, o, 64); }
    if (lane == 0) { red[0][wave] = cx; red[1][wave] = cy; red[2][wave] = cz; red[3][wave] = cnt; }
    __syncthreads();
    cx = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]); cy = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    cz = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]); cnt = (red[3][0] + red[3][1]) + (red[3][2] + red[3][3]);
    if (cnt > 0.f) { cx /= cnt; cy /= cnt; cz /= cnt; }
    if (!finite3(cx, cy, cz)) { cx = 0.f; cy = 0.f; cz = 0.f; if (threadIdx.x == 0) atomicOr(bad, 1); }      // (sums beyond f32: no level 0 for this cloud)
    float rho = 0.f;
    for (size_t p = base + threadIdx.x; p < end; p += GR_BLOCK) {
        const float4 r = rec[p];
        if (finite3(r.x, r.y, r.z)) rho = fmaxf(rho, fmaxf(fmaxf(fabsf(r.x - cx), fabsf(r.y - cy)), fabsf(r.z - cz)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) rho = fmaxf(rho, __shfl_xor(rho, o, 64));
    __syncthreads();
    if (lane == 0) red[0][wave] = rho;
    __syncthreads();
    if (threadIdx.x != 0) return;
    rho = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
    int k = 0;
    if (rho > 0.f) {
        int e2;
        (void)frexpf(rho, &e2);                            // rho <= 2^e2
        k = 7 - e2;
        if (k < -60 || k > 60) { atomicOr(bad, 1); k = k < 0 ? -60 : 60; }
    }
    centres[s] = make_float4(cx, cy, cz, ldexpf(1.0f, k));
}

// one wave per level-1 tile: its row of its level-0 tile (row m <-> level-1 tile 16 ((m >> 2) & 1) + 4 (m >> 3) + (m & 3) of the 32, as the chunks
// of a level-1 tile: the 16 accumulators of lane-half h are the tiles 16 h + i, in order)
__global__ __launch_bounds__(GR_BLOCK) void bt_l0_ops_kernel(const float4* __restrict__ rec, uint32_t n_rec, uint32_t n_rows, const float4* __restrict__ centres,
                                                             uint4* __restrict__ ops)
{
    const uint32_t lane = threadIdx.x & 63, T1 = blockIdx.x * (GR_BLOCK / 64) + (threadIdx.x >> 6);
    if (T1 >= n_rows) return;
    const float4 C = centres[T1 / (BT_L0_SUPER / 512)];
    float tx[8], ty[8], tz[8];
    bool fin[8];
    float mn[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, mx[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
    bool any = false;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const size_t p = (size_t)T1 * 512 + (size_t)j * 64 + lane;
        const float4 r = p < n_rec ? rec[p] : make_float4(__builtin_inff(), 0.f, 0.f, 0.f);
        tx[j] = (r.x - C.x) * C.w; ty[j] = (r.y - C.y) * C.w; tz[j] = (r.z - C.z) * C.w;       // exact scaling, |.| <= 2^7 (unless the super-tile is flagged bad)
        fin[j] = finite3(r.x, r.y, r.z) && fabsf(tx[j]) <= 128.0f && fabsf(ty[j]) <= 128.0f && fabsf(tz[j]) <= 128.0f;
        if (fin[j]) {
            any = true;
            mn[0] = fminf(mn[0], tx[j]); mx[0] = fmaxf(mx[0], tx[j]);
            mn[1] = fminf(mn[1], ty[j]); mx[1] = fmaxf(mx[1], ty[j]);
            mn[2] = fminf(mn[2], tz[j]); mx[2] = fmaxf(mx[2], tz[j]);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int k = 0; k < 3; k++) { mn[k] = fminf(mn[k], __shfl_xor(mn[k], o, 64)); mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], o, 64)); }
    any = __builtin_amdgcn_ballot_w64(any) != 0ull;
    uint32_t pp[3][2] = { { 0u, 0u }, { 0u, 0u }, { 0u, 0u } };
    float c[3] = { 0.0f, 0.0f, 0.0f };
    float r2 = 0.0f;
    if (any) {
        sph_centre(mn, mx, pp, c);                                         // (the same on every lane)
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (fin[j]) {
                const float dx = tx[j] - c[0], dy = ty[j] - c[1], dz = tz[j] - c[2];
                r2 = fmaxf(r2, (dx * dx + dy * dy) + dz * dz);
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) r2 = fmaxf(r2, __shfl_xor(r2, o, 64));
    }
    if (lane != 0) return;
    uint4 lo, hi;
    sph_finish(any, pp, c, r2, lo, hi);
    const uint32_t T0 = T1 >> 5, j = T1 & 31u, m = (((j >> 2) & 3u) << 3) | ((j >> 4) << 2) | (j & 3u);
    ops[(size_t)T0 * 64 + m] = lo;
    ops[(size_t)T0 * 64 + 32 + m] = hi;
}

int bt_ensure_l1(pcr_ctx* ctx, const pcr_cloud* tgt)
{
    BtIndex* bt = tgt->bt;
    if (!bt || !bt->safe || !bt->n_tiles) return fail(ctx, PCR_ERR_STATE, "bt_ensure_l1: no index");
    if (bt->l1_block) return PCR_OK;
    const size_t n_pad = bt->n_tiles * 32, n_l1 = (n_pad + BT_L1_SUPER - 1) / BT_L1_SUPER, n_l1_tiles = n_l1 * (BT_L1_SUPER / 512);
    const size_t n_l0 = (n_pad + BT_L0_SUPER - 1) / BT_L0_SUPER, n_l0_tiles = n_l0 * (BT_L0_SUPER / 512 / 32);
    const size_t off_ops = (n_l1 * sizeof(float4) + 255) & ~(size_t)255, off_rec = off_ops + n_l1_tiles * 64 * sizeof(uint4),
                 off_c0 = off_rec + bt->n_tiles * 64 * sizeof(uint4), off_ops0 = (off_c0 + n_l0 * sizeof(float4) + 255) & ~(size_t)255,
                 off_flag = off_ops0 + n_l0_tiles * 64 * sizeof(uint4), total = off_flag + 256;
    char* blk = nullptr;
    hipError_t e = hipMalloc((void**)&blk, total);
    if (e != hipSuccess) return fail(ctx, PCR_ERR_HIP, "hipMalloc(level-1 operands)", e);
    bt->l1_block = blk; bt->l1_centres = (float4*)blk; bt->l1_ops = (uint4*)(blk + off_ops); bt->l1_rec_ops = (uint4*)(blk + off_rec);
    bt->l0_centres = (float4*)(blk + off_c0); bt->l0_ops = (uint4*)(blk + off_ops0); bt->n_l0_super = n_l0;
    bt->l1_bad = (int*)(blk + off_flag); bt->n_l1_super = n_l1;
    e = hipMemsetAsync(bt->l1_bad, 0, sizeof(int), ctx->stream);
    if (e != hipSuccess) return fail(ctx, PCR_ERR_HIP, "level-1 operands", e);
    hipLaunchKernelGGL(bt_l1_centres_kernel, dim3((unsigned)n_l1), dim3(GR_BLOCK), 0, ctx->stream, bt->records, (uint32_t)n_pad, (uint32_t)n_l1, bt->l1_centres, bt->l1_bad);
    hipLaunchKernelGGL(bt_l1_ops_kernel, dim3((unsigned)((n_l1_tiles * 32 + GR_BLOCK - 1) / GR_BLOCK)), dim3(GR_BLOCK), 0, ctx->stream, bt->records, (uint32_t)n_pad,
                       (uint32_t)n_l1_tiles, bt->l1_centres, bt->l1_ops);
    hipLaunchKernelGGL(bt_l1_rec_ops_kernel, dim3((unsigned)((bt->n_tiles * 32 + GR_BLOCK - 1) / GR_BLOCK)), dim3(GR_BLOCK), 0, ctx->stream, bt->records, (uint32_t)bt->n_tiles,
                       bt->l1_centres, bt->l1_rec_ops);
    hipLaunchKernelGGL(bt_l0_centres_kernel, dim3((unsigned)n_l0), dim3(GR_BLOCK), 0, ctx->stream, bt->records, (uint32_t)n_pad, (uint32_t)n_l0, bt->l0_centres, bt->l1_bad);
    hipLaunchKernelGGL(bt_l0_ops_kernel, dim3((unsigned)((n_l0_tiles * 32 + GR_BLOCK / 64 - 1) / (GR_BLOCK / 64))), dim3(GR_BLOCK), 0, ctx->stream, bt->records, (uint32_t)n_pad,
                       (uint32_t)(n_l0_tiles * 32), bt->l0_centres, bt->l0_ops);
    PCR_HIP(ctx, hipGetLastError());
    return PCR_OK;
}

// ---- extras of the sign tile search (grid_stile.hpp)
// the Morton key a record was sorted by (non-finite records and padding: beyond every cell)
__device__ __forceinline__ uint32_t bt_record_key(const float4 r, float lox, float loy, float loz, float inv)
{
    if (!finite3(r.x, r.y, r.z)) return 1u << 30;
    return bt_morton(bt_fine_cell(r.x, lox, inv), bt_fine_cell(r.y, loy, inv), bt_fine_cell(r.z, loz, inv));
}

// counts of the coarse cells from the boundaries of the sorted records: the first record of a cell subtracts its position, the last one adds
// its position + 1 (two atomics per occupied cell, modulo 2^32); an exclusive scan of the counts is cell_start.  Non-finite records and
// padding (key 2^30) count into the entry behind the last cell.
__global__ __launch_bounds__(GR_BLOCK) void bt_cell_count_kernel(const float4* __restrict__ rec, uint32_t n_rec, float lox, float loy, float loz, float inv,
                                                                 int shift, uint32_t* __restrict__ count)
{
    const uint32_t i = blockIdx.x * GR_BLOCK + threadIdx.x;
    if (i >= n_rec) return;
    const uint32_t c = bt_record_key(rec[i], lox, loy, loz, inv) >> shift;
    const bool first = i == 0 || (bt_record_key(rec[i - 1], lox, loy, loz, inv) >> shift) != c;
    const bool last = i + 1 == n_rec || (bt_record_key(rec[i + 1], lox, loy, loz, inv) >> shift) != c;
    if (first) atomicSub(&count[c], i);
    if (last) atomicAdd(&count[c], i + 1u);
}

// bounding sphere of every tile of 32 records (centre = middle of the box of its finite members, radius rounded up; no finite member: radius < 0)
__global__ __launch_bounds__(GR_BLOCK) void bt_tile_spheres_kernel(const float4* __restrict__ records, uint32_t n_tiles, float4* __restrict__ spheres)
{
    const uint32_t t = blockIdx.x * GR_BLOCK + threadIdx.x;
    if (t >= n_tiles) return;
    float mn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, mx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (int j = 0; j < 32; j++) {
        const float4 r = records[(size_t)t * 32 + j];
        if (finite3(r.x, r.y, r.z)) {
            mn[0] = fminf(mn[0], r.x); mn[1] = fminf(mn[1], r.y); mn[2] = fminf(mn[2], r.z);
            mx[0] = fmaxf(mx[0], r.x); mx[1] = fmaxf(mx[1], r.y); mx[2] = fmaxf(mx[2], r.z);
        }
    }
    if (mn[0] > mx[0]) { spheres[t] = make_float4(0.f, 0.f, 0.f, -1.0f); return; }
    const float cx = 0.5f * mn[0] + 0.5f * mx[0], cy = 0.5f * mn[1] + 0.5f * mx[1], cz = 0.5f * mn[2] + 0.5f * mx[2];
    float r2 = 0.f;
    for (int j = 0; j < 32; j++) {
        const float4 r = records[(size_t)t * 32 + j];
        if (finite3(r.x, r.y, r.z)) {
            const float dx = r.x - cx, dy = r.y - cy, dz = r.z - cz;
            r2 = fmaxf(r2, (dx * dx + dy * dy) + dz * dz);
        }
    }
    spheres[t] = make_float4(cx, cy, cz, sqrtf(r2) * 1.00001f + 1e-30f);    // rounded up (build_spheres_kernel)
}

__global__ __launch_bounds__(GR_BLOCK) void bt_scatter_gpos_kernel(const float4* __restrict__ grid_rec, uint32_t n_grid, uint32_t n_orig, uint32_t* __restrict__ gpos_of_orig)
{
    const uint32_t p = blockIdx.x * GR_BLOCK + threadIdx.x;
    if (p >= n_grid) return;
    const uint32_t o = __float_as_uint(grid_rec[p].w);
    if (o < n_orig) gpos_of_orig[o] = p;
}

__global__ __launch_bounds__(GR_BLOCK) void bt_gather_gpos_kernel(const float4* __restrict__ bt_rec, uint32_t n_bt, uint32_t n_orig, const uint32_t* __restrict__ gpos_of_orig,
                                                                  uint32_t* __restrict__ g_of_b)
{
    const uint32_t p = blockIdx.x * GR_BLOCK + threadIdx.x;
    if (p >= n_bt) return;
    const uint32_t o = __float_as_uint(bt_rec[p].w);
    g_of_b[p] = o < n_orig ? gpos_of_orig[o] : 0xFFFFFFFFu;
}

__global__ __launch_bounds__(GR_BLOCK) void bt_invert_gpos_kernel(const uint32_t* __restrict__ g_of_b, uint32_t n_bt, uint32_t n_grid, uint32_t* __restrict__ b_of_g)
{
    const uint32_t p = blockIdx.x * GR_BLOCK + threadIdx.x;
    if (p >= n_bt) return;
    const uint32_t gp = g_of_b[p];
    if (gp < n_grid) b_of_g[gp] = p;
}

#include "grid_stile.hpp"

int bt_ensure_tile(pcr_ctx* ctx, const pcr_cloud* tgt)
{
    BtIndex* bt = tgt->bt;
    const Grid* g = tgt->grid;
    if (!bt || !g || !bt->safe || !bt->n_tiles) return fail(ctx, PCR_ERR_STATE, "bt_ensure_tile: no index");
    if (bt->tile_block && bt->g_of == g) return PCR_OK;
    if (bt->tile_block) { PCR_HIP(ctx, hipStreamSynchronize(ctx->stream)); hipFree(bt->tile_block); bt->tile_block = nullptr; bt->g_of = nullptr; }
    const size_t n = tgt->n, n_pad = bt->n_tiles * 32, n_tiles = bt->n_tiles;
    // coarse cells: 9 bits per axis (the longest extent / 512: 31 cm on a 160 m scene; 2^27 + 2 table entries, sparsely touched) — tune
    // grid_stile_cbits 6 .. 9 (measured at 10 M points: 8 bits 2.05 ms per converged search, 9 bits 1.90: a pass tests a third of the tiles)
    const int cbits = (int)std::min<int64_t>(9, std::max<int64_t>(6, tune_get(ctx, "grid_stile_cbits", 9)));
    const size_t n_cells = (size_t)1 << (3 * cbits);
    const size_t off_sph = ((n_cells + 2) * sizeof(uint32_t) + 255) & ~(size_t)255, off_gob = off_sph + ((n_tiles * sizeof(float4) + 255) & ~(size_t)255),
                 off_bog = off_gob + ((n_pad * sizeof(uint32_t) + 255) & ~(size_t)255),
                 total = off_bog + ((g->n_chunks * GRID_CHUNK * sizeof(uint32_t) + 255) & ~(size_t)255);
    const size_t scan_blocks = (n_cells + 2 + SC_TILE - 1) / SC_TILE;
    int rc = ensure_scratch(ctx, std::max(n * sizeof(uint32_t), (scan_blocks + 1) * sizeof(uint32_t)) + 256);
    if (rc) return rc;
    char* blk = nullptr;
    hipError_t e = hipMalloc((void**)&blk, total);
    if (e != hipSuccess) return fail(ctx, PCR_ERR_HIP, "hipMalloc(tile extras)", e);
    bt->tile_block = blk;
    bt->cell_start = (uint32_t*)blk; bt->tile_spheres = (float4*)(blk + off_sph); bt->g_of_b = (uint32_t*)(blk + off_gob); bt->b_of_g = (uint32_t*)(blk + off_bog);
    bt->cbits = cbits;
    e = hipMemsetAsync(bt->cell_start, 0, (n_cells + 2) * sizeof(uint32_t), ctx->stream);
    if (e != hipSuccess) return fail(ctx, PCR_ERR_HIP, "tile extras", e);
    hipLaunchKernelGGL(bt_cell_count_kernel, dim3((unsigned)((n_pad + GR_BLOCK - 1) / GR_BLOCK)), dim3(GR_BLOCK), 0, ctx->stream, bt->records, (uint32_t)n_pad,
                       bt->key_lo[0], bt->key_lo[1], bt->key_lo[2], bt->key_inv, 3 * (10 - cbits), bt->cell_start);
    uint32_t* totals = (uint32_t*)ctx->scratch;
    rc = exclusive_scan_u32(ctx, bt->cell_start, bt->cell_start, n_cells + 2, totals, totals + scan_blocks);
    if (rc) return rc;
    uint32_t* gpos = (uint32_t*)ctx->scratch;                 // (stream order: the scan is done with the scratch before the scatter writes it)
    hipLaunchKernelGGL(bt_tile_spheres_kernel, dim3((unsigned)((n_tiles + GR_BLOCK - 1) / GR_BLOCK)), dim3(GR_BLOCK), 0, ctx->stream, bt->records, (uint32_t)n_tiles,
                       bt->tile_spheres);
    const size_t n_grid = g->n_chunks * GRID_CHUNK;           // (the grid's records are padded to whole runs)
    hipLaunchKernelGGL(bt_scatter_gpos_kernel, dim3((unsigned)((n_grid + GR_BLOCK - 1) / GR_BLOCK)), dim3(GR_BLOCK), 0, ctx->stream, g->records, (uint32_t)n_grid,
                       (uint32_t)n, gpos);
    hipLaunchKernelGGL(bt_gather_gpos_kernel, dim3((unsigned)((n_pad + GR_BLOCK - 1) / GR_BLOCK)), dim3(GR_BLOCK), 0, ctx->stream, bt->records, (uint32_t)n_pad,
                       (uint32_t)n, gpos, bt->g_of_b);
    e = hipMemsetAsync(bt->b_of_g, 0xFF, n_grid * sizeof(uint32_t), ctx->stream);
    if (e != hipSuccess) return fail(ctx, PCR_ERR_HIP, "tile extras", e);
    hipLaunchKernelGGL(bt_invert_gpos_kernel, dim3((unsigned)((n_pad + GR_BLOCK - 1) / GR_BLOCK)), dim3(GR_BLOCK), 0, ctx->stream, bt->g_of_b, (uint32_t)n_pad,
                       (uint32_t)n_grid, bt->b_of_g);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(ctx, PCR_ERR_HIP, "tile extras", e);
    bt->g_of = g;
    return PCR_OK;
}

int bt_ensure(pcr_ctx* ctx, const pcr_cloud* tgt)
{
    if (tgt->bt) return PCR_OK;
    const size_t n = tgt->n;
    BtIndex* bt = new (std::nothrow) BtIndex();
    if (!bt) return fail(ctx, PCR_ERR_NOMEM, "btiles");
    if (n == 0 || n > 0x7FFFFFF0ull) { const_cast<pcr_cloud*>(tgt)->bt = bt; return PCR_OK; }       // safe = false: the other kernels answer
    const size_t n_super = (n + BT_SUPER - 1) / BT_SUPER, n_pad = n_super * BT_SUPER, n_tiles = n_pad / 32;
    const uint32_t bb_blocks = (uint32_t)std::min<size_t>(256, (n + GR_BLOCK - 1) / GR_BLOCK);
    const size_t off_cen = n_pad * sizeof(float4), off_ops = off_cen + ((n_super * sizeof(float4) + 255) & ~(size_t)255),
                 off_o16 = off_ops + n_tiles * 128 * sizeof(uint4), off_bb = off_o16 + n_tiles * 64 * sizeof(uint4),
                 off_flag = off_bb + ((bb_blocks * 6 * sizeof(float) + 255) & ~(size_t)255), total = off_flag + 256;
    // order inside a lattice cell: arrival order below 1 M points (a cell holds a few points at most), a 6-bit-per-axis Morton code of
    // the position inside the cell from there on (tune bt_fine_bits: 0 auto, 1 .. 7 bits, -1 none)
    int64_t fb_tune = tune_get(ctx, "bt_fine_bits", 0);
    const int fine_bits = fb_tune < 0 ? 0 : fb_tune > 0 ? (int)std::min<int64_t>(fb_tune, 7) : (n >= ((size_t)1 << 20) ? 6 : 0);
    const int key_bits = 31 + 3 * fine_bits;
    size_t temp_bytes = 0;
    sort_pairs_u64_u32(nullptr, temp_bytes, nullptr, nullptr, nullptr, nullptr, n, 0, key_bits, ctx->stream);
    const size_t a4 = (n * 4 + 255) & ~(size_t)255, a8 = (n * 8 + 255) & ~(size_t)255;
    int rc = ensure_scratch(ctx, 2 * a8 + 2 * a4 + temp_bytes + 256);
    if (rc) { delete bt; return rc; }
    char* blk = nullptr;
    hipError_t e = hipMalloc((void**)&blk, total);
    if (e != hipSuccess) { delete bt; return fail(ctx, PCR_ERR_HIP, "hipMalloc(btiles)", e); }
    bt->block = blk;
    bt->records = (float4*)blk; bt->centres = (float4*)(blk + off_cen); bt->ops = (uint4*)(blk + off_ops);
    bt->ops16 = (uint4*)(blk + off_o16); bt->bad16 = (int*)(blk + off_flag);
    e = hipMemsetAsync(bt->bad16, 0, sizeof(int), ctx->stream);
    if (e != hipSuccess) { bt_free(bt); return fail(ctx, PCR_ERR_HIP, "btiles", e); }
    float* bb_dev = (float*)(blk + off_bb);
    // the one host round trip: the bounding box of the finite points
    hipLaunchKernelGGL(bbox_kernel, dim3(bb_blocks), dim3(GR_BLOCK), 0, ctx->stream, tgt->x(), tgt->y(), tgt->z(), (uint32_t)n, bb_dev);
    std::vector<float> hb(bb_blocks * 6);
    e = hipMemcpyAsync(hb.data(), bb_dev, hb.size() * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { bt_free(bt); return fail(ctx, PCR_ERR_HIP, "btiles bbox", e); }
    float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (uint32_t b = 0; b < bb_blocks; b++)
        for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], hb[b * 6 + k]); hi[k] = std::max(hi[k], hb[b * 6 + 3 + k]); }
    float amax = 0.f;
    bool any = lo[0] <= hi[0];
    for (int k = 0; k < 3 && any; k++) amax = std::max(amax, std::max(std::fabs(lo[k]), std::fabs(hi[k])));
    if (tgt->absmax < 0.f) const_cast<pcr_cloud*>(tgt)->absmax = amax;                                // (cloud_absmax's cache, for free)
    // |t - C| < 1e18 and |t - C|^2 < 3e38 for every finite point and any centre inside the box: what the filter's analysis wants
    bt->safe = any && amax < 5e17f;
    if (bt->safe) {
        // one cell edge for all three axes (the longest extent / 1024): cubic cells keep the Morton runs compact in space — with a
        // lattice per axis the thin z-extent of a LiDAR scan got 5 mm slabs and the runs spread over the x-y plane (measured: 1.25
        // against 1.07 ms per search, the bound of such a run is loose)
        const float ext = std::max(std::max(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
        float iv1 = ext > 0.f ? 1024.0f / ext : 0.0f;
        if (!(iv1 < 3e38f)) iv1 = 0.0f;
        const float iv[3] = { iv1, iv1, iv1 };
        bt->key_lo[0] = lo[0]; bt->key_lo[1] = lo[1]; bt->key_lo[2] = lo[2]; bt->key_inv = iv1;
        char* sc = (char*)ctx->scratch;
        unsigned long long* k_in = (unsigned long long*)sc;
        unsigned long long* k_out = (unsigned long long*)(sc + a8);
        uint32_t* v_in = (uint32_t*)(sc + 2 * a8);
        uint32_t* v_out = (uint32_t*)(sc + 2 * a8 + a4);
        char* temp = sc + 2 * a8 + 2 * a4;
        hipLaunchKernelGGL(bt_keys_kernel, dim3((unsigned)((n + GR_BLOCK - 1) / GR_BLOCK)), dim3(GR_BLOCK), 0, ctx->stream, tgt->x(), tgt->y(), tgt->z(), (uint32_t)n,
                           lo[0], lo[1], lo[2], iv[0], iv[1], iv[2], k_in, v_in, fine_bits);
        e = sort_pairs_u64_u32(temp, temp_bytes, k_in, k_out, v_in, v_out, n, 0, key_bits, ctx->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(gather_records_kernel, dim3((unsigned)((n_pad + GR_BLOCK - 1) / GR_BLOCK)), dim3(GR_BLOCK), 0, ctx->stream, tgt->x(), tgt->y(), tgt->z(),
                               (uint32_t)n, (uint32_t)n_pad, v_out, bt->records);
            hipLaunchKernelGGL(bt_centres_kernel, dim3((unsigned)((n_super + GR_BLOCK / 64 - 1) / (GR_BLOCK / 64))), dim3(GR_BLOCK), 0, ctx->stream, bt->records, (uint32_t)n_super,
                               bt->centres, bt->bad16);
            hipLaunchKernelGGL(bt_ops16_kernel, dim3((unsigned)((n_tiles * 32 + GR_BLOCK - 1) / GR_BLOCK)), dim3(GR_BLOCK), 0, ctx->stream, bt->records, (uint32_t)n_tiles,
                               bt->centres, bt->ops16);
            hipLaunchKernelGGL(bt_ops_kernel, dim3((unsigned)((n_tiles * 32 + GR_BLOCK - 1) / GR_BLOCK)), dim3(GR_BLOCK), 0, ctx->stream, bt->records, (uint32_t)n_tiles,
                               bt->centres, bt->ops, (int*)nullptr);
            e = hipGetLastError();
        }
        if (e != hipSuccess) { bt_free(bt); return fail(ctx, PCR_ERR_HIP, "btiles", e); }
        bt->n_tiles = n_tiles;
        // (stream order: the scratch may be reused by later launches on ctx->stream, which run after the sort and the gather)
    }
    const_cast<pcr_cloud*>(tgt)->bt = bt;
    return PCR_OK;
}

// order of a query batch (tune grid_sort_fine: 0 auto = record order for large batches against a Morton-ordered index, where the
// radix sort is also the faster of the two: 0.6 against 4.0 ms at 10 M queries; 1 = record order, 2 = coarse bins)
static int sort_queries_any(pcr_ctx* ctx, const Grid* g, const pcr_cloud* src)
{
    const int64_t fine = tune_get(ctx, "grid_sort_fine", 0);
    return (fine == 1 || (fine == 0 && !g->x_sorted && src->n >= 65536)) ? sort_queries_fine(ctx, g, src) : sort_queries(ctx, g, src);
}

// the 1-NN index of a target cloud, cached on the cloud: Morton-ordered records + bounding spheres (the sphere walk) from 256 points on
// (tune grid_order: 0 auto, 1 = x-sorted, 2 = Morton).  Until the second session of round 3 only targets of 500 000 points and more
// got it; measured then on 20-iteration loops and one-shot searches from 1 000 to 250 000 points (tools/run_grid_order.py,
// profiles/r03_grid_order.txt): the sphere walk is ahead at every size — 120 k: 78.0 -> 64.5 us per iteration, 250 k: 143.9 -> 116.5,
// 4 000: 30.0 -> 28.1, the first search of a fresh 120 k target 855 -> 707 us.
int build_target_grid(pcr_ctx* ctx, const pcr_cloud* tgt)
{
    if (tgt->grid) return PCR_OK;
    const int64_t ord = tune_get(ctx, "grid_order", 0);
    const int order = (ord == 2 || (ord == 0 && tgt->n >= 256)) ? GRID_ORDER_MORTON : GRID_ORDER_X;
    Grid* g = nullptr;
    ProfScope p(ctx, "grid_build");
    int rc = grid_build(ctx, tgt, &g, 0.0, order);
    if (rc) return rc;
    const_cast<pcr_cloud*>(tgt)->grid = g;
    return PCR_OK;
}

int grid_prepare_queries(pcr_ctx* ctx, const pcr_cloud* tgt, const pcr_cloud* src)
{
    int rcb = build_target_grid(ctx, tgt);
    if (rcb) return rcb;
    if (src->n == 0) return PCR_OK;
    ProfScope p(ctx, "grid_sort_queries");
    return sort_queries_any(ctx, tgt->grid, src);
}

// dst[t] = src[perm[t]] (cell-sorted working copy of the grid ICP); the padding of dst is left alone
__global__ __launch_bounds__(GR_BLOCK) void permute_cloud_kernel(const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz,
                                                                 const uint32_t* __restrict__ perm, uint32_t n, float* __restrict__ dx, float* __restrict__ dy,
                                                                 float* __restrict__ dz, uint32_t cap)
{
    // (the destination is a fresh allocation: positions [n, cap) get every cloud's padding here — x = +inf, y = z = 0 — instead of a whole-cloud
    // copy in front of this launch)
    const uint32_t t = blockIdx.x * GR_BLOCK + threadIdx.x;
    if (t >= cap) return;
    if (t >= n) { dx[t] = __builtin_inff(); dy[t] = 0.0f; dz[t] = 0.0f; return; }
    const uint32_t i = perm[t];
    dx[t] = sx[i]; dy[t] = sy[i]; dz[t] = sz[i];
}

// Re-orders the working cloud of an ICP loop into the order of the target's index (sort_queries_any), ONCE: every later search reads
// its queries with coalesced loads (no perm indirection), writes keys / winner positions coalesced, and the Kabsch pass walks
// pairs whose targets are neighbours in the record array.  ctx->work_orig[t] = index the point had in the caller's cloud (the
// "last kept pair" of registration.cpp:939 is defined in that order).  The sums are exact, so the order changes no result.
int grid_sort_working_cloud(pcr_ctx* ctx, const pcr_cloud* tgt, pcr_cloud** work, bool in_place)
{
    pcr_cloud* w = *work;
    const size_t n = w->n;
    ctx->work_orig_src = nullptr;
    if (n == 0 || tune_get(ctx, "grid_sort_work", 1) != 1) return PCR_OK;
    int rc = grid_prepare_queries(ctx, tgt, w);                  // builds the target index if needed; ctx->qperm = the order
    if (rc) return rc;
    if (ctx->work_orig_cap < n) {                                 // (before the clone: nothing to give back on these error paths)
        if (ctx->work_orig) PCR_HIP(ctx, hipFree(ctx->work_orig));
        ctx->work_orig = nullptr; ctx->work_orig_cap = 0;
        PCR_HIP(ctx, hipMalloc((void**)&ctx->work_orig, padded(n) * sizeof(uint32_t)));
        ctx->work_orig_cap = padded(n);
    }
    pcr_cloud* sorted = nullptr;
    rc = cloud_alloc(ctx, n, &sorted);                            // same size; the permute writes all of it, padding included
    if (rc) return rc;
    hipLaunchKernelGGL(permute_cloud_kernel, dim3((unsigned)((sorted->cap + GR_BLOCK - 1) / GR_BLOCK)), dim3(GR_BLOCK), 0, ctx->stream, w->x(), w->y(), w->z(), ctx->qperm,
                       (uint32_t)n, sorted->x(), sorted->y(), sorted->z(), (uint32_t)sorted->cap);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->work_orig, ctx->qperm, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream);
    if (e != hipSuccess) { pcr_cloud_destroy(ctx, sorted); return fail(ctx, PCR_ERR_HIP, "grid_sort_working_cloud", e); }
    if (in_place) { std::swap(w->base, sorted->base); std::swap(w, sorted); cloud_modified(sorted); }   // the caller's handle keeps its identity and gets the sorted buffer
    cloud_release(ctx, w);                                         // (no synchronisation: the permute above still reads it — the buffer stays allocated)
    *work = sorted;
    ctx->work_orig_src = sorted;
    ctx->work_orig_n = n;
    ctx->qperm_src = nullptr;                                      // the permutation belongs to the cloud that no longer exists
    return PCR_OK;
}

// A brute-force ICP loop over the matrix-core index: the working cloud in the Morton order of the target's super-tiles, ONCE per loop.
// The queries of a wave then lie next to each other: the slices that can matter to them coincide, so most (wave, slice) pairs are
// settled for the whole wave by the published bound, and the queries a slice cannot settle cluster in few (wave, slice) pairs instead
// of one here, one there (a source cloud in random order is the worst case: measured 0.620 -> 0.600 ms per 120 k x 120 k search).
// ctx->work_orig[t] = the index the point had in the caller's cloud, as in grid_sort_working_cloud; the sums are exact, so the
// order changes no result.  Costs one key kernel, one 30-bit radix sort and one gather.
int bt_sort_working_cloud(pcr_ctx* ctx, const pcr_cloud* tgt, pcr_cloud** work, bool in_place)
{
    pcr_cloud* w = *work;
    const size_t n = w->n;
    ctx->work_orig_src = nullptr;
    if (n < 4096 || n > 0x7FFFFFF0ull || tune_get(ctx, "bt_sort_work", 1) != 1) return PCR_OK;
    int rc = bt_ensure(ctx, tgt);
    if (rc) return rc;
    const BtIndex* bt = tgt->bt;
    if (!bt || !bt->safe || !bt->n_tiles) return PCR_OK;
    size_t temp_bytes = 0;
    sort_pairs_u64_u32(nullptr, temp_bytes, nullptr, nullptr, nullptr, nullptr, n, 0, 31, ctx->stream);
    const size_t a4 = (n * 4 + 255) & ~(size_t)255, a8 = (n * 8 + 255) & ~(size_t)255;
    rc = ensure_scratch(ctx, 2 * a8 + 2 * a4 + temp_bytes + 256);
    if (rc) return rc;
    char* sc = (char*)ctx->scratch;
    unsigned long long* k_in = (unsigned long long*)sc;
    unsigned long long* k_out = (unsigned long long*)(sc + a8);
    uint32_t* v_in = (uint32_t*)(sc + 2 * a8);
    uint32_t* v_out = (uint32_t*)(sc + 2 * a8 + a4);
    char* temp = sc + 2 * a8 + 2 * a4;
    if (ctx->work_orig_cap < n) {                                 // (before the clone: nothing to give back on these error paths)
        if (ctx->work_orig) PCR_HIP(ctx, hipFree(ctx->work_orig));
        ctx->work_orig = nullptr; ctx->work_orig_cap = 0;
        PCR_HIP(ctx, hipMalloc((void**)&ctx->work_orig, padded(n) * sizeof(uint32_t)));
        ctx->work_orig_cap = padded(n);
    }
    pcr_cloud* sorted = nullptr;
    rc = cloud_alloc(ctx, n, &sorted);                            // same size; the permute writes all of it, padding included
    if (rc) return rc;
    const unsigned blocks = (unsigned)((n + GR_BLOCK - 1) / GR_BLOCK);
    v_out = ctx->work_orig;                                       // (the sorted values ARE the original indices: no copy behind the sort)
    hipLaunchKernelGGL(bt_keys_kernel, dim3(blocks), dim3(GR_BLOCK), 0, ctx->stream, w->x(), w->y(), w->z(), (uint32_t)n, bt->key_lo[0], bt->key_lo[1],
                       bt->key_lo[2], bt->key_inv, bt->key_inv, bt->key_inv, k_in, v_in, 0);
    // (tune bt_sort_begin_bit: the low bits of the Morton key the sort ignores — the order inside the cells they span stays the caller's; every
    // eight bits less are one radix pass less)
    const int begin_bit = (int)std::min<int64_t>(std::max<int64_t>(tune_get(ctx, "bt_sort_begin_bit", 0), 0), 24);
    hipError_t e = sort_pairs_u64_u32(temp, temp_bytes, k_in, k_out, v_in, v_out, n, begin_bit, 31, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(permute_cloud_kernel, dim3((unsigned)((sorted->cap + GR_BLOCK - 1) / GR_BLOCK)), dim3(GR_BLOCK), 0, ctx->stream, w->x(), w->y(), w->z(), v_out,
                           (uint32_t)n, sorted->x(), sorted->y(), sorted->z(), (uint32_t)sorted->cap);
        e = hipGetLastError();
    }
    if (e != hipSuccess) { pcr_cloud_destroy(ctx, sorted); return fail(ctx, PCR_ERR_HIP, "bt_sort_working_cloud", e); }
    if (in_place) { std::swap(w->base, sorted->base); std::swap(w, sorted); cloud_modified(sorted); }
    cloud_release(ctx, w);                                         // (no synchronisation: the permute above still reads it — the buffer stays allocated)
    *work = sorted;
    ctx->work_orig_src = sorted;
    ctx->work_orig_n = n;
    return PCR_OK;
}

// ---- spatially coherent shards of a source cloud (multi-GPU: pcr_cloud_shard_spatial, include/pcr.h)
// member[perm[t]] = 1 where position t of the sorted order falls into one of this rank's chunks
__global__ __launch_bounds__(GR_BLOCK) void shard_mark_kernel(const uint32_t* __restrict__ perm, uint32_t n, uint32_t chunk_len, uint32_t nranks, uint32_t rank,
                                                              uint32_t* __restrict__ member)
{
    const uint32_t t = blockIdx.x * GR_BLOCK + threadIdx.x;
    if (t >= n) return;
    member[perm[t]] = ((t / chunk_len) % nranks == rank) ? 1u : 0u;
}

// the members in ascending ORIGINAL index (pos = exclusive scan of member): coordinates and global indices
__global__ __launch_bounds__(GR_BLOCK) void shard_gather_kernel(const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz, uint32_t n,
                                                                const uint32_t* __restrict__ member, const uint32_t* __restrict__ pos, float* __restrict__ dx,
                                                                float* __restrict__ dy, float* __restrict__ dz, uint32_t* __restrict__ gidx)
{
    const uint32_t i = blockIdx.x * GR_BLOCK + threadIdx.x;
    if (i >= n || !member[i]) return;
    const uint32_t p = pos[i];
    dx[p] = sx[i]; dy[p] = sy[i]; dz[p] = sz[i];
    gidx[p] = i;
}

__global__ __launch_bounds__(GR_BLOCK) void shard_pad_kernel(float* __restrict__ dx, float* __restrict__ dy, float* __restrict__ dz, uint32_t n, uint32_t cap)
{
    const uint32_t i = n + blockIdx.x * GR_BLOCK + threadIdx.x;
    if (i >= cap) return;
    dx[i] = __builtin_inff(); dy[i] = 0.0f; dz[i] = 0.0f;     // the padding of every cloud (pcr_cloud_create)
}

// Rank `rank`'s share of `full` under a SPATIALLY COHERENT partition: the cloud in the order of the target's index (cell, then Morton
// code inside the cell: the order the ICP loops give their working copy) is cut into nranks x chunks_per_rank runs of equal length, and
// the runs are dealt to the ranks round-robin.  Every rank then holds compact pieces of the scene at the scene's own density — what the
// tile search of large targets needs (a uniformly drawn 1 / N sample spreads 32 consecutive queries over N times the region) — and the
// round-robin deal balances regions that cost more (near rings) against cheap ones.  Deterministic: every rank computes the same order
// from the same two clouds, so the shards are disjoint and complete.  The shard keeps the points in ascending original index and
// remembers those indices (pcr_cloud::gidx).
int cloud_shard_spatial(pcr_ctx* ctx, const pcr_cloud* tgt, const pcr_cloud* full, int nranks, int rank, int chunks_per_rank, pcr_cloud** out,
                        int (*alloc)(pcr_ctx*, size_t, pcr_cloud**))
{
    const size_t n = full->n;
    if (n > 0xFFFFFFF0ull) return fail(ctx, PCR_ERR_ARG, "cloud too large for u32 indices");
    int rc = build_target_grid(ctx, tgt);
    if (rc) return rc;
    size_t count = 0;
    const size_t a4 = (n * 4 + 255) & ~(size_t)255;
    uint32_t* member = nullptr, *pos = nullptr, *totals = nullptr;
    const size_t scan_blocks = (n + SC_TILE - 1) / SC_TILE;
    if (n) {
        rc = sort_queries_fine(ctx, tgt->grid, full);          // ctx->qperm[t] = original index of the point at sorted position t (radix sort: stable, deterministic)
        if (rc) return rc;
        ctx->qperm_src = nullptr;                             // (the order belongs to `full`, which no search will use)
        hipError_t e = hipMalloc((void**)&member, 2 * a4 + (scan_blocks + 2) * sizeof(uint32_t));   // (not the scratch: the sort above lives there)
        if (e != hipSuccess) return fail(ctx, PCR_ERR_HIP, "hipMalloc(shard)", e);
        pos = (uint32_t*)((char*)member + a4); totals = (uint32_t*)((char*)member + 2 * a4);
        const size_t n_chunks = (size_t)nranks * (size_t)std::max(1, chunks_per_rank);
        const uint32_t chunk_len = (uint32_t)std::max<size_t>(1, (n + n_chunks - 1) / n_chunks);
        const unsigned blocks = (unsigned)((n + GR_BLOCK - 1) / GR_BLOCK);
        hipLaunchKernelGGL(shard_mark_kernel, dim3(blocks), dim3(GR_BLOCK), 0, ctx->stream, ctx->qperm, (uint32_t)n, chunk_len, (uint32_t)nranks, (uint32_t)rank, member);
        rc = exclusive_scan_u32(ctx, member, pos, n, totals, totals + scan_blocks);
        uint32_t h_count = 0;
        if (rc == PCR_OK) {
            e = hipMemcpyAsync(&h_count, totals + scan_blocks, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) rc = fail(ctx, PCR_ERR_HIP, "shard count", e);
        }
        if (rc) { hipFree(member); return rc; }
        count = h_count;
    }
    pcr_cloud* c = nullptr;
    rc = alloc(ctx, count, &c);
    if (rc) { if (member) hipFree(member); return rc; }
    hipError_t e = hipMalloc((void**)&c->gidx, std::max<size_t>(count, 1) * sizeof(uint32_t));
    if (e != hipSuccess) { if (member) hipFree(member); pcr_cloud_destroy(ctx, c); return fail(ctx, PCR_ERR_HIP, "hipMalloc(shard indices)", e); }
    if (n)
        hipLaunchKernelGGL(shard_gather_kernel, dim3((unsigned)((n + GR_BLOCK - 1) / GR_BLOCK)), dim3(GR_BLOCK), 0, ctx->stream, full->x(), full->y(), full->z(), (uint32_t)n,
                           member, pos, c->x(), c->y(), c->z(), c->gidx);
    hipLaunchKernelGGL(shard_pad_kernel, dim3((unsigned)((c->cap - count + GR_BLOCK - 1) / GR_BLOCK)), dim3(GR_BLOCK), 0, ctx->stream, c->x(), c->y(), c->z(), (uint32_t)count,
                       (uint32_t)c->cap);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (member) hipFree(member);
    if (e != hipSuccess) { pcr_cloud_destroy(ctx, c); return fail(ctx, PCR_ERR_HIP, "shard gather", e); }
    *out = c;
    return PCR_OK;
}

int launch_nn1_grid(pcr_ctx* ctx, const pcr_cloud* tgt, const pcr_cloud* src, bool reuse_perm, float cap2)
{
    const size_t ns = src->n;
    if (ns == 0) { ctx->keys_n = 0; ctx->wpos_valid = false; return PCR_OK; }
    if (ns > 0xFFFFFFF0ull || tgt->n > 0xFFFFFFF0ull) return fail(ctx, PCR_ERR_ARG, "cloud too large for u32 indices");
    ctx->keys_seeded = false;             // (seeds written for an exhaustive search: the walk re-evaluates its own)
    int rc = ensure_keys(ctx, ns);
    if (rc) return rc;
    ctx->keys_n = ns;
    // the working cloud of an ICP loop is already in cell order (grid_sort_working_cloud): no permutation to read
    const bool sorted = reuse_perm && ctx->work_orig_src == src && ctx->work_orig_n == ns && tgt->grid;
    const bool have_perm = sorted || (reuse_perm && tgt->grid && ctx->qperm && ctx->qperm_n == ns && ctx->qperm_src == src);
    if (!have_perm) {
        rc = grid_prepare_queries(ctx, tgt, src);
        if (rc) return rc;
    }
    const Grid* g = tgt->grid;
    const uint32_t* perm = (!sorted && tune_get(ctx, "grid_sort_queries", 1) > 0) ? ctx->qperm : nullptr;
    unsigned long long* stats_dev = nullptr;
    if (tune_get(ctx, "grid_stats", 0) > 0) {     // diagnostics: candidates / fine rows / coarse rows / far stages of this launch
        if (!ctx->grid_stats_dev) PCR_HIP(ctx, hipMalloc((void**)&ctx->grid_stats_dev, PCR_NSTATS * sizeof(unsigned long long)));
        PCR_HIP(ctx, hipMemsetAsync(ctx->grid_stats_dev, 0, PCR_NSTATS * sizeof(unsigned long long), ctx->stream));
        stats_dev = ctx->grid_stats_dev;
    }
    // warm start: only inside an ICP loop (reuse_perm), from its second search on, when keys[] belongs to this source
    // Unlike the exhaustive kernels (nn1_brute.hip), the grid walk TRUSTS its seed — the radius jumps to the one that proves it —
    // so only the previous iteration of the same cloud object qualifies: correspondences left by another source cloud of the
    // same size would be valid but arbitrarily bad candidates (measured: a 5-iteration ICP 0.93 -> 12.5 ms, profiles/r01_tune_grid.txt).
    int warm = (reuse_perm && ctx->keys_warm && ctx->keys_src == src && ctx->keys_warm_n == ns &&
                tune_get(ctx, "grid_warm_start", 1) > 0) ? 1 : 0;
    if (warm && ctx->wpos_valid && ctx->wpos_n == ns && ctx->wpos) warm = 2;      // the winners' record positions are there: seed from the records
    ctx->keys_warm = reuse_perm;
    ctx->keys_warm_n = ns;
    ctx->keys_src = src;
    ctx->keys_tgt = tgt;
    // search kernel (tune grid_mode: 0 auto, 1 plain, 2 x-window clipping, 3 bounding spheres):
    //   spheres  — Morton-ordered index (large targets, build_target_grid): runs of 16 records are tested by their bounding sphere first;
    //   clipping — x-sorted index of a large target: 14.1 -> 8.1 ms per search at 10 M x 10 M in round 1 (1 277 -> 462 candidates per query);
    //   plain    — small / sparse targets, where either extra phase costs more than it saves (38 -> 48 us at 120 k with clipping).
    const int64_t mode_tune = tune_get(ctx, "grid_mode", 0);
    int mode = 0;
    if (mode_tune == 0) mode = (g->spheres && !g->x_sorted) ? 2 : (tgt->n >= 500000 && g->x_sorted) ? 1 : 0;
    else if (mode_tune == 2) mode = g->x_sorted ? 1 : 0;
    else if (mode_tune == 3) mode = g->spheres ? 2 : 0;
    const int G = mode == 2 ? 16 : (int)tune_get(ctx, "grid_lanes", 16);   // measured: profiles/r01_tune_grid.txt
    // unbounded searches hand their far queries to the exhaustive kernel (grid_far_brute: 1 on (default), 2 off)
    uint32_t* far_list = nullptr; uint32_t* far_count = nullptr; uint32_t far_cap = 0;
    if (!(cap2 < __builtin_inff()) && tgt->n >= 4096 && tune_get(ctx, "grid_far_brute", 1) == 1) {
        const size_t want = std::max<size_t>(ns / 8, 1024);
        if (ctx->far_cap < want) {
            if (ctx->far_list) { PCR_HIP(ctx, hipStreamSynchronize(ctx->stream)); hipFree(ctx->far_list); ctx->far_list = nullptr; ctx->far_cap = 0; }
            PCR_HIP(ctx, hipMalloc((void**)&ctx->far_list, (want + 1) * sizeof(uint32_t)));
            ctx->far_cap = want;
        }
        far_list = ctx->far_list; far_count = ctx->far_list + ctx->far_cap; far_cap = (uint32_t)std::min<size_t>(want, ctx->far_cap);
        PCR_HIP(ctx, hipMemsetAsync(far_count, 0, sizeof(uint32_t), ctx->stream));
    }
    // winner positions: kept inside ICP loops whose searches the grid answers alone (the exhaustive hand-off merges into keys[] only)
    uint32_t* wpos = nullptr;
    if (reuse_perm && !far_list && warm != 1 && tune_get(ctx, "grid_wpos", 1) == 1) {
        if (ctx->wpos_cap < ns) {
            if (ctx->wpos) { PCR_HIP(ctx, hipStreamSynchronize(ctx->stream)); hipFree(ctx->wpos); ctx->wpos = nullptr; ctx->wpos_cap = 0; }
            PCR_HIP(ctx, hipMalloc((void**)&ctx->wpos, padded(ns) * sizeof(uint32_t)));
            ctx->wpos_cap = padded(ns);
            if (warm == 2) warm = 1;                       // (cannot happen: a valid wpos has the capacity) — never seed from a fresh buffer
        }
        wpos = ctx->wpos;
    }
    if (warm == 2 && !wpos) warm = 1;
    // sphere walk with record-position seeds: the previous winner's whole run is scanned ahead of the walk (tune grid_seed_run: 2 = off)
    // (the plain walk of small targets gains nothing from it: measured, 52.5 against 53.2 us per 120 k iteration)
    if (warm == 2 && mode == 2 && tune_get(ctx, "grid_seed_run", 1) == 1) warm = 3;
    ctx->wpos_valid = wpos != nullptr;
    ctx->wpos_n = ns;
    // XCD-aware workgroup -> query-block mapping for large batches (tune grid_xcd_run: run length in workgroups, -1 = identity)
    int64_t run = tune_get(ctx, "grid_xcd_run", 0);
    const size_t nblocks = (ns * (size_t)G + GR_BLOCK - 1) / GR_BLOCK;
    if (run == 0) run = nblocks >= 4096 ? 32 : -1;
    size_t launch_blocks = nblocks;
    uint32_t xcd_run = 0;
    if (run > 0) {
        xcd_run = (uint32_t)run;
        const size_t unit = 8 * (size_t)xcd_run;
        launch_blocks = (nblocks + unit - 1) / unit * unit;
    }
    ctx->last_nn1_kernel = "grid";
    // TILE SEARCH (tune grid_tile: 0 auto = targets of 4 000 000 points and more, 1 on, 2 off): the searches of a loop whose working cloud is in the order
    // of a Morton-ordered index, by coherent query tiles — one wave per 64 consecutive queries shares candidate tiles and record loads; queries it cannot
    // serve go to a segmented list that the cell walk serves in a second launch (list mode).  The kernel is the SIGN tile search of round 4
    // (grid_stile.hpp; the tile kernel of round 3 — operands built per visit, minimum tracking — was removed when it took over: over a 20-iteration loop
    // at 10 M it ran 3.72 against 2.76 ms per iteration, profiles/r04_c5_stile.txt).  The tile search is ahead of the sphere walk once the pose has
    // settled, behind it while many queries are deferred; over a whole loop from the start pose the walk wins below ~4 M points (profiles/r03_grid_order.txt).
    // Shared knobs: grid_tile_reach_pct (cluster reach of a pass in ball limits, 200), grid_tile_min_members (8), grid_tile_list_segs (groups per wave of
    // the static list walk, 1).  Same results as the walk alone, bit for bit (tests: test_config5.py, test_gpu_parity.py::test_icp_tile_search_...).
    const int64_t tile_tune = tune_get(ctx, "grid_tile", 0);
    // auto: large targets, and a working cloud about as dense as the target where it lies — 32 consecutive queries of a SPARSE subset (a
    // uniformly drawn 1 / 8 shard of a multi-GPU run) span eight times the region, their pass needs eight times the records per query, and the
    // cell walk (whose cost per query does not depend on the other queries) wins: 0.40 against 0.61 ms per converged search of a 1 / 8 shard of
    // the 10 M pair, 1.44 against 1.04 ms (tile ahead) at 1 / 2 — profiles/r03_c5_tile_search.txt.  Density = queries per occupied cell of
    // the target's grid against targets per occupied cell; a spatially compact shard keeps its local density and the tile search.
    // (both counters come back through pinned words behind an event: the first search after a sort / a build waits for them here, once)
    const uint32_t work_cells = (tile_tune != 2 && tgt->n >= 4000000) ? work_cells_now(ctx) : 0u, occupied = work_cells ? grid_occupied_now(ctx, tgt->grid) : 0u;
    const bool dense = work_cells > 0 && occupied > 0 && (double)ns / (double)work_cells >= 0.4 * (double)tgt->n / (double)occupied;
    // (round 4: the FIRST search of a loop too — warm == 0: no correspondences yet — where the sign tile search can make itself a seed per query:
    // stile_seed_kernel, tune grid_stile_cold: 2 = the plain walk as before)
    const bool cold_tile = warm == 0 && reuse_perm && tune_get(ctx, "grid_stile_cold", 1) == 1 && tune_get(ctx, "grid_stile", 0) != 2;
    if (mode == 2 && (warm == 3 || cold_tile) && sorted && !perm && wpos && cap2 < __builtin_inff() && tile_tune != 2 && (tile_tune == 1 || (tgt->n >= 4000000 && dense))) do {
        // the segmented list of deferred queries: 32 slots per group of 32 queries + one count per group (far_list is free here: the
        // hand-off of far queries to the exhaustive kernel only exists for unbounded searches)
        const size_t n_groups_sz = (ns + 31) / 32, need = n_groups_sz * 32 + n_groups_sz + (4 * n_groups_sz + 2);     // (+ the queue: quarters of the non-empty segments)
        if (ctx->far_cap < need) {
            if (ctx->far_list) { PCR_HIP(ctx, hipStreamSynchronize(ctx->stream)); hipFree(ctx->far_list); ctx->far_list = nullptr; ctx->far_cap = 0; }
            PCR_HIP(ctx, hipMalloc((void**)&ctx->far_list, (need + 1) * sizeof(uint32_t)));
            ctx->far_cap = need;
        }
        uint32_t* dlist = ctx->far_list;
        uint32_t* dcount = ctx->far_list + n_groups_sz * 32;
        const float reach_k = (float)tune_get(ctx, "grid_tile_reach_pct", 200) * 0.01f;
        const uint32_t min_members = (uint32_t)std::min<int64_t>(32, std::max<int64_t>(1, tune_get(ctx, "grid_tile_min_members", 8)));
        const uint32_t n_groups = (uint32_t)((ns + 31) / 32);
        const uint32_t list_segs = (uint32_t)std::min<int64_t>(64, std::max<int64_t>(1, tune_get(ctx, "grid_tile_list_segs", 1)));
        const unsigned lblocks = (unsigned)(((n_groups_sz + (size_t)list_segs * 4 - 1) / ((size_t)list_segs * 4) + 255) / 256 * 256);      // (whole XCD runs)
        // SIGN TILE SEARCH (grid_stile.hpp, round 4; tune grid_stile: 0 auto = on where the f16 matrix pipe passed the device check and the
        // target's Morton-ordered index fits f16, 2 = off: the cell walk): one wave per 64 consecutive queries, the candidate tiles
        // of the target's matrix-core index found through coarse Morton cells and run spheres, one MFMA per tile and 32 queries whose
        // SIGN says whether a record can matter.  Knobs: grid_stile_bmax_cm (largest ball of a served query, default 100), grid_stile_lim_pct / grid_stile_lim_floor_mm (ball limit of a wave: a multiple of its mean ball, 400 %, never below 150 mm), grid_stile_keep (candidate
        // tiles a pass may keep, 768; grid_stile_keep_small: the same for passes whose largest ball is below grid_stile_split_mm, 192), grid_stile_cells (coarse cells a pass may open, 2 048), grid_stile_flush / grid_stile_dense (STRACK's list rules, 64 / 32).
        bool stile = tune_get(ctx, "grid_stile", 0) != 2 && mfma_verdict(ctx, true);
        if (stile) {
            rc = bt_ensure(ctx, tgt);
            if (rc) return rc;
            BtIndex* bt = tgt->bt;
            stile = bt && bt->safe && bt->n_tiles;
            if (stile && bt->bad16_host < 0) {
                int flag = 1;
                PCR_HIP(ctx, hipMemcpyAsync(&flag, bt->bad16, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
                PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
                bt->bad16_host = flag;
            }
            stile = stile && bt->bad16_host == 0;
            if (stile && (rc = bt_ensure_tile(ctx, tgt))) return rc;
        }
        // operands: the per-record rows in the scale of the LEVEL-1 super-tiles (4 096 records share a centre: BtIndex::l1_rec_ops, what STRACK3's level 2
        // reads) instead of the 256-record super-tiles' — a wave at the settled pose of the 10 M pair rebuilt its queries' side 11 times per search, 30 % of
        // its vector instructions (tune grid_stile_l1: 2 = the super-tiles' operands)
        bool stile_l1 = stile && tune_get(ctx, "grid_stile_l1", 1) == 1;
        if (stile_l1) {
            if ((rc = bt_ensure_l1(ctx, tgt))) return rc;
            BtIndex* bt = tgt->bt;
            if (bt->l1_bad_host < 0) {
                int flag = 1;
                PCR_HIP(ctx, hipMemcpyAsync(&flag, bt->l1_bad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
                PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
                bt->l1_bad_host = flag;
            }
            stile_l1 = bt->l1_bad_host == 0;
        }
        if (!stile) break;                                    // (this device's f16 arithmetic failed the check, or the target leaves f16's range: the walk below)
        if (stile) {
            const BtIndex* bt = tgt->bt;
            const float sbmax = (float)tune_get(ctx, "grid_stile_bmax_cm", 100) * 0.01f;
            const uint32_t skeep = (uint32_t)std::min<int64_t>(SL_KEEP, std::max<int64_t>(2, tune_get(ctx, "grid_stile_keep", SL_KEEP)));
            const float slim_k = (float)tune_get(ctx, "grid_stile_lim_pct", 400) * 0.01f, slim_floor = (float)tune_get(ctx, "grid_stile_lim_floor_mm", 150) * 0.001f;
            const uint32_t skeep_small = (uint32_t)std::min<int64_t>(SL_KEEP, std::max<int64_t>(2, tune_get(ctx, "grid_stile_keep_small", 192)));
            const uint32_t scells = (uint32_t)std::min<int64_t>(1 << 15, std::max<int64_t>(1, tune_get(ctx, "grid_stile_cells", 2048)));
            const uint32_t sflush = (uint32_t)std::min<int64_t>(SL_CAP, std::max<int64_t>(1, tune_get(ctx, "grid_stile_flush", 64)));
            const uint32_t sdense = (uint32_t)std::min<int64_t>(65, std::max<int64_t>(1, tune_get(ctx, "grid_stile_dense", 32)));
            // clusters of one wave's queries served one after the other (tune grid_stile_passes; round 3's tile kernel stopped at three: at the
            // converged pose of the 10 M pair 0.5 % of the queries sat in a fourth cluster and went to the list walk, whose launch has a
            // latency floor of ~0.19 ms however few queries it serves)
            const uint32_t spasses = (uint32_t)std::min<int64_t>(64, std::max<int64_t>(1, tune_get(ctx, "grid_stile_passes", 3)));
            const float ssplit = (float)tune_get(ctx, "grid_stile_split_mm", 40) * 0.001f;     // largest ball of a pass from which its near tiles go first
            const uint32_t n_waves = (uint32_t)((ns + 63) / 64);
            size_t sblocks = (n_waves + (GR_BLOCK / 64) - 1) / (GR_BLOCK / 64);
            int64_t srun = tune_get(ctx, "grid_xcd_run", 0);
            if (srun == 0) srun = sblocks >= 4096 ? 32 : -1;
            uint32_t sxcd = 0;
            if (srun > 0) { sxcd = (uint32_t)srun; const size_t unit = 8 * (size_t)sxcd; sblocks = (sblocks + unit - 1) / unit * unit; }
            ctx->last_nn1_kernel = "grid-stile";
            // the list walk draws the non-empty segments from a queue (tune grid_stile_queue: 2 = one workgroup per four segments as before);
            // grid_stile_list_wgs resident workgroups serve it (default 8 per CU)
            uint32_t* lqueue = tune_get(ctx, "grid_stile_queue", 1) == 1 ? dcount + n_groups_sz : nullptr;
            const unsigned qblocks = (unsigned)std::min<int64_t>(65535, std::max<int64_t>(1, tune_get(ctx, "grid_stile_list_wgs", 8 * (int64_t)ctx->prop.multiProcessorCount)));
            const unsigned lblocks_s = lqueue ? qblocks : lblocks;
            const uint32_t list_segs_s = lqueue ? 1u : list_segs;
            if (lqueue) PCR_HIP(ctx, hipMemsetAsync(lqueue, 0, 2 * sizeof(uint32_t), ctx->stream));
            ProfScope p(ctx, "nn1_grid", 1);
            if (warm == 0) {
                // COLD: a seed per query from its own coarse cell of the target's Morton-ordered index — the best of up to 32 records spread
                // evenly over the cell's range (a genuine candidate decimetres from the true neighbour, where the plain walk's first search had
                // to open its cube blindly: 13.3 ms at 10 M points) — written as the winner position the tile search seeds itself from
                hipLaunchKernelGGL(stile_seed_kernel, dim3((unsigned)((ns + GR_BLOCK - 1) / GR_BLOCK)), dim3(GR_BLOCK), 0, ctx->stream, bt->records, bt->cell_start, bt->g_of_b,
                                   bt->key_lo[0], bt->key_lo[1], bt->key_lo[2], bt->key_inv, 3 * (10 - bt->cbits), src->x(), src->y(), src->z(), (uint32_t)ns, wpos,
                                   ctx->stop_flag_dev, (uint32_t)std::min<int64_t>(256, std::max<int64_t>(1, tune_get(ctx, "grid_stile_cold_own", 32))),
                                   (uint32_t)std::min<int64_t>(64, std::max<int64_t>(0, tune_get(ctx, "grid_stile_cold_per", 4))));
            }

#define PCR_STILE(ST)                                                                                                                       \
    hipLaunchKernelGGL((nn1_stile_kernel<ST>), dim3((unsigned)sblocks), dim3(GR_BLOCK), 0, ctx->stream, bt->records, stile_l1 ? bt->l1_rec_ops : bt->ops16, stile_l1 ? bt->l1_centres : bt->centres, \
                       bt->tile_spheres, bt->cell_start, bt->g_of_b, bt->b_of_g, g->records, (uint32_t)(g->n_chunks * GRID_CHUNK), bt->key_lo[0], bt->key_lo[1], \
                       bt->key_lo[2], bt->key_inv, 3 * (10 - bt->cbits), src->x(), src->y(), src->z(), (uint32_t)ns, ctx->keys, ctx->stop_flag_dev,    \
                       stats_dev, cap2, wpos, dlist, dcount, lqueue, sxcd, sbmax, n_waves, n_groups, slim_k, reach_k, skeep, scells, min_members, sflush, sdense, ssplit, stile_l1 ? (uint32_t)bt->n_l1_super : (uint32_t)(bt->n_tiles / 8), spasses, skeep_small, slim_floor, stile_l1 ? 7u : 3u); \
    hipLaunchKernelGGL((nn1_grid_kernel<16, ST, 2, true>), dim3(lblocks_s), dim3(GR_BLOCK), 0, ctx->stream, g->records, g->spheres, g->cell_start, g->p,   \
                       src->x(), src->y(), src->z(), (const uint32_t*)dlist, (uint32_t)ns, ctx->keys, ctx->stop_flag_dev, stats_dev, tgt->x(),     \
                       tgt->y(), tgt->z(), (uint32_t)tgt->n, 3, cap2, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, wpos, 0u, (const uint32_t*)dcount, list_segs_s, lqueue)
#ifdef PCR_SL_PROF
            PCR_STILE(false);                                 // (profile build: the production instantiation, stamps into the diagnostics words)
#else
            if (stats_dev) { PCR_STILE(true); } else { PCR_STILE(false); }
#endif
#undef PCR_STILE
            PCR_HIP(ctx, hipGetLastError());
            return PCR_OK;
        }
        break;                                                // (no sign tile search on this device / for this target: the cell walk below)
    } while (0);
    {
        ProfScope p(ctx, "nn1_grid", 1);
#define PCR_GRID2(GG, ST, MD)                                                                                          \
    hipLaunchKernelGGL((nn1_grid_kernel<GG, ST, MD>), dim3((unsigned)launch_blocks), dim3(GR_BLOCK), 0, ctx->stream, g->records, g->spheres, \
                       g->cell_start, g->p, src->x(), src->y(), src->z(), perm, (uint32_t)ns, ctx->keys, ctx->stop_flag_dev, stats_dev, tgt->x(), tgt->y(),   \
                       tgt->z(), (uint32_t)tgt->n, warm, cap2, far_list, far_count, far_cap, wpos, xcd_run, (const uint32_t*)nullptr, 0u)
#define PCR_GRID(GG)                                                                                                   \
    do {                                                                                                               \
        if (stats_dev) { if (mode == 1) PCR_GRID2(GG, true, 1); else PCR_GRID2(GG, true, 0); }                         \
        else { if (mode == 1) PCR_GRID2(GG, false, 1); else PCR_GRID2(GG, false, 0); }                                 \
    } while (0)
        if (mode == 2) {
            if (stats_dev) PCR_GRID2(16, true, 2); else PCR_GRID2(16, false, 2);
        } else switch (G) {
        case 1: PCR_GRID(1); break;
        case 2: PCR_GRID(2); break;
        case 4: PCR_GRID(4); break;
        case 8: PCR_GRID(8); break;
        case 32: PCR_GRID(32); break;
        case 64: PCR_GRID(64); break;
        default: PCR_GRID(16); break;
        }
#undef PCR_GRID2
#undef PCR_GRID
        if (far_list) {
            rc = launch_nn1_brute_list(ctx, tgt, src, far_list, far_count, far_cap);
            if (rc) return rc;
        }
    }
    PCR_HIP(ctx, hipGetLastError());
    return PCR_OK;
}

}  // namespace pcr

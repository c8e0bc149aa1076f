// grid_common.hpp — the uniform-grid index shared by the exact 1-NN search (grid.hip) and the radius consumers (iss.hip).
#pragma once

#include "pcr_internal.hpp"

#include <cfloat>

namespace pcr {

struct GridParams {
    float lo[3];
    float inv_h;
    float h;
    int n[3];          // cells per axis
    float slack;       // in cells
};

struct Grid {
    GridParams p;
    size_t n_points = 0;
    size_t n_cells = 0;
    uint32_t occupied = 0;            // non-empty cells (0: not read back yet — grid_occupied_now() waits for the build's copy)
    uint64_t occupied_tag = 0;        // which read-back of the context's pinned word is this grid's
    float4* records = nullptr;        // n_points, sorted by (cell, x): every x-row of cells is one x-sorted range
    uint32_t* cell_start = nullptr;   // n_cells + 2 (cell n_cells holds the non-finite points, never visited)
    // chunked, centred copy of the records for the expanded-form brute-force filter (nn1_brute.hip, ETRACK): per chunk of
    // GRID_CHUNK consecutive records GRID_CHUNK_FLOATS floats = { Cx, Cy, Cz, 0 } + -2x''[GRID_CHUNK] + -2y''[..] + -2z''[..] + w_lb[..]
    // (t'' = t - C, w_lb = |t''|^2 (1 - 2^-18); SoA inside the chunk so that two neighbouring targets sit in an aligned SGPR
    // pair for v_pk_fma_f32); padding / non-finite records are (0, 0, 0, +inf).  chunk_safe: every value finite and small
    // enough (< 1e18) for the filter's error analysis.
    float* chunks = nullptr;
    size_t n_chunks = 0;
    bool chunk_safe = false;
    // Order of the records inside a cell: by x (every x-row of cells is one x-sorted range: what the window-clipping consumers —
    // k-NN, radius, ISS, the CLIP search kernel — rely on), or by the Morton code of an 8 x 8 x 8 sub-cell position (spatially
    // compact runs: what the bounding spheres below want).
    bool x_sorted = true;
    // bounding sphere {Cx, Cy, Cz, radius (rounded up)} of every run of GRID_CHUNK consecutive records (n_chunks entries; the
    // records array is padded to a whole chunk with x = +inf entries): one 16-byte load decides whether 16 records can hold a
    // better neighbour (grid.hip, SPH search kernel)
    float4* spheres = nullptr;
    // the points once more as {x, y, z, 0} in ORIGINAL order (radius_grid.hip: one 16-byte gather per reported neighbour instead of
    // three 4-byte ones); built on first use
    float4* by_index = nullptr;
};

constexpr int BT_SUPER = 256;          // records per super-tile (8 tiles of 32; measured per 120k x 120k search: 128 -> 0.94, 256 -> 0.90, 512 -> 0.97 ms)

enum GridOrder { GRID_ORDER_X = 0, GRID_ORDER_MORTON = 1 };

constexpr int GRID_CHUNK = 16;
constexpr int GRID_CHUNK_FLOATS = 4 + 4 * GRID_CHUNK;

__device__ __forceinline__ int cell_coord(float v, float lo, float inv_h)
{
    const float a = floorf((v - lo) * inv_h);
    // clamp far-away values before the int conversion; +-2^22 cells is beyond any grid we build
    return (int)fminf(fmaxf(a, -4194304.0f), 4194304.0f);
}

__device__ __forceinline__ bool finite3(float x, float y, float z)
{
    return (fabsf(x) <= FLT_MAX) && (fabsf(y) <= FLT_MAX) && (fabsf(z) <= FLT_MAX);   // false for NaN / inf
}

// nonfinite_cell: where NaN / inf points go.  They never win a comparison (d2 = inf / NaN); the index build parks them in
// an extra cell behind the last real one so that every x-row of the records stays sorted by x.
__device__ __forceinline__ uint32_t clamped_cell_id(const GridParams& g, float x, float y, float z, uint32_t nonfinite_cell = 0)
{
    if (!finite3(x, y, z)) return nonfinite_cell;
    const int cx = min(max(cell_coord(x, g.lo[0], g.inv_h), 0), g.n[0] - 1);
    const int cy = min(max(cell_coord(y, g.lo[1], g.inv_h), 0), g.n[1] - 1);
    const int cz = min(max(cell_coord(z, g.lo[2], g.inv_h), 0), g.n[2] - 1);
    return (uint32_t)((cz * g.n[1] + cy) * g.n[0] + cx);
}

// builds the index over `c`; cell_edge > 0 forces the cell size (otherwise the heuristic of grid.hip is used)
int grid_build(pcr_ctx* ctx, const pcr_cloud* c, Grid** out, double cell_edge, int order = GRID_ORDER_X);
// builds (and caches on tgt) the 1-NN grid if needed
int build_target_grid(pcr_ctx* ctx, const pcr_cloud* tgt);
// BTRACK's index (nn1_brute.hip), independent of the cell grid: the cloud in Morton order of a lattice of cubic cells (longest extent / 1024) over its bounding
// box — spatially compact runs — as {x, y, z, original index} records (padded to whole super-tiles of BT_SUPER records with
// x = +inf), one centre per super-tile, and per tile of 32 records the bf16 A operands of two v_mfma_f32_32x32x16_bf16
// ([tile][2][64 lanes] x 16 bytes).  One allocation, one host synchronisation (the bounding box) to build.
struct BtIndex {
    void* block = nullptr;            // the one allocation behind the three arrays
    float4* records = nullptr;
    float4* centres = nullptr;
    uint4* ops = nullptr;
    // HTRACK (nn1_brute.hip): the same tiles as f16 operands of ONE v_mfma_f32_32x32x16_f16 ([tile][64 lanes] x 16 bytes), scaled per
    // super-tile by the power of two in centres[].w so that |t - C| * scale <= 2^7; ok16[0] = 0 when a super-tile's scale exponent
    // left [-60, 60] (the f16 form cannot carry such a cloud: BTRACK answers)
    uint4* ops16 = nullptr;
    int* bad16 = nullptr;             // device flag, read lazily (bad16_host: -1 unknown)
    int bad16_host = -1;
    size_t n_tiles = 0;
    float key_lo[3] = { 0.f, 0.f, 0.f }, key_inv = 0.f;   // the lattice of the Morton keys the records are ordered by (bt_sort_working_cloud)
    bool safe = false;                // every finite coordinate below 5e17 in magnitude (and at least one finite point)
    // ---- extras of the SIGN TILE SEARCH of large-target loops (grid_stile.hpp; built on first use by bt_ensure_tile, one more allocation):
    // cell_start[c] = first record whose Morton key, cut to cbits bits per axis, is >= c (2^(3 cbits) + 1 entries: the records of coarse
    // cell c are [cell_start[c], cell_start[c + 1]), non-finite records and padding lie behind the last entry); tile_spheres = bounding
    // sphere of every tile of 32 records (radius < 0 = no finite member); g_of_b[p] = position of record p in the
    // records of the cell grid `g_of` (the winner positions of a loop — ctx->wpos — stay in that numbering for the walk and the Kabsch pass)
    // ---- level 1 of the two-level sign filter (the sphere forms, nn1_sphere.hpp; bt_ensure_l1): per level-1 super-tile of BT_L1_SUPER records a centre
    // + power-of-two scale, per chunk of 16 records ONE operand row (its bounding sphere): [level-1 tile of 32 chunks][64 lanes] x 16 bytes
    void* l1_block = nullptr;
    float4* l1_centres = nullptr;
    uint4* l1_ops = nullptr;
    uint4* l1_rec_ops = nullptr;      // the per-RECORD operands once more, in the scale of the record's level-1 super-tile ([tile][64 lanes] x 16 bytes): level 2 of
                                      // the sphere form then needs no operand setup of its own (one per query and 4 096 records instead of one per 256)
    float4* l0_centres = nullptr;     // level 0 of STRACK3: one row per LEVEL-1 TILE (the bounding sphere of its 512 records), level-0 super-tiles of
    uint4* l0_ops = nullptr;          // BT_L0_SUPER records share a centre and a scale ([level-0 tile][64 lanes] x 16 bytes, 32 rows per tile)
    size_t n_l0_super = 0;
    size_t n_l1_super = 0;
    int l1_bad_host = -1;             // a level-1 super-tile whose scale left the f16 range (device flag behind l1_ops, read lazily)
    int* l1_bad = nullptr;
    void* tile_block = nullptr;
    uint32_t* cell_start = nullptr;
    int cbits = 0;
    float4* tile_spheres = nullptr;
    uint32_t* g_of_b = nullptr;
    uint32_t* b_of_g = nullptr;       // ... and the inverse (0xFFFFFFFF: a padding record of the grid)
    const Grid* g_of = nullptr;
};
// HTRACK operand helpers (device): v ~ p1 + p2 in f16 (round toward zero, then the remainder), and the 16 bytes a lane holds for
// target row t'' (already scaled): lanes < 32  [x: t1 t2 t1 t2 | y: t1 t2 t1 t2],  lanes >= 32  [z: t1 t2 t1 t2 | w1 w2 0 0]
__device__ __forceinline__ void ht_split(float v, uint32_t& p1, uint32_t& p2)
{
    typedef __fp16 h2 __attribute__((ext_vector_type(2)));
    const h2 a = __builtin_amdgcn_cvt_pkrtz(v, v);
    const float rem = v - (float)a.x;
    const h2 b = __builtin_amdgcn_cvt_pkrtz(rem, rem);
    p1 = __builtin_bit_cast(uint32_t, a) & 0xFFFFu; p2 = __builtin_bit_cast(uint32_t, b) & 0xFFFFu;
}

constexpr uint32_t HT_THETA_CONSTS = 0x54007800u;      // f16 (2^15, 2^6) in K-slots (14, 15) of every target row
__device__ __forceinline__ uint4 ht_target_operand(float tx, float ty, float tz, bool finite, bool upper_half)
{
    uint32_t t[3][2] = { { 0, 0 }, { 0, 0 }, { 0, 0 } }, w[2] = { 0x7C00u, 0 };                  // padding / non-finite: w = +inf
    if (finite) {
        const float ww = ((tx * tx + ty * ty) + tz * tz) * 0.99999237060546875f;               // (1 - 2^-17), <= 3 * 2^14 < 65504
        ht_split(-2.0f * tx, t[0][0], t[0][1]); ht_split(-2.0f * ty, t[1][0], t[1][1]); ht_split(-2.0f * tz, t[2][0], t[2][1]);
        ht_split(fminf(ww, 65000.0f), w[0], w[1]);
    }
    const uint32_t qx = t[0][0] | (t[0][1] << 16), qy = t[1][0] | (t[1][1] << 16), qz = t[2][0] | (t[2][1] << 16);
    // K-slots 14 / 15 of the upper half: the constants 2^15 and 2^6 (f16 0x7800, 0x5400).  The minimum-tracking kernels keep zeros on the
    // query side there (products 0); the sign filter (st_theta below) puts the two pieces of its threshold against them.
    return upper_half ? make_uint4(qz, qz, w[0] | (w[1] << 16), HT_THETA_CONSTS) : make_uint4(qx, qx, qy, qy);
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// query side of the f16 filter (nn1_brute.hip HTRACK): error analysis in the header of nn1_btrack_kernel
#ifndef PCR_HT_ABS_SLACK
#define PCR_HT_ABS_SLACK 2.384185791015625e-07f      // 2^-22 (A/B builds of the underflow test: -DPCR_HT_ABS_SLACK=0.0f)
#endif
__device__ __forceinline__ void ht_pair(float c, uint32_t& d_hi, uint32_t& d_lo)      // (c1, c1) and (c2, c2): c ~ c1 + c2 in f16
{
    typedef __fp16 h2 __attribute__((ext_vector_type(2)));
    const h2 a = __builtin_amdgcn_cvt_pkrtz(c, c);
    const float rem = c - (float)a.x;
    const h2 b = __builtin_amdgcn_cvt_pkrtz(rem, rem);
    d_hi = __builtin_bit_cast(uint32_t, a); d_lo = __builtin_bit_cast(uint32_t, b);
}

// the f16 form's query operands for one super-tile (centre C.xyz, scale C.w): clamped scaled offset, two f16 pieces per coordinate in the
// lane-half's K-slots (lanes >= 32: [z pieces | 1, 1, 0, 0]), and R = KAPPA |r|^2 / scale^2 (the bound of a tile is then ONE fma:
// m / scale^2 + R — the division is exact)
__device__ __forceinline__ void ht_setup(float qx, float qy, float qz, const float4 C, bool h, uint4& bq, float& R, float& inv2)
{
    constexpr float KAPPA = 0.99999237060546875f;             // 1 - 2^-17
    const float sc = C.w;
    inv2 = 1.0f / (sc * sc);                                  // exact: |exponent| <= 120
    const float rx = __builtin_amdgcn_fmed3f((qx - C.x) * sc, -32000.0f, 32000.0f), ry = __builtin_amdgcn_fmed3f((qy - C.y) * sc, -32000.0f, 32000.0f),
                rz = __builtin_amdgcn_fmed3f((qz - C.z) * sc, -32000.0f, 32000.0f);
    // KAPPA |r|^2 - 2^-22 in scaled units (the absolute slack for f16 underflow: header above), then back to the cloud's units
    R = __builtin_fmaf(-PCR_HT_ABS_SLACK, inv2, (__builtin_fmaf(rz, rz, __builtin_fmaf(ry, ry, rx * rx)) * KAPPA) * inv2);
    uint32_t f1, f2, s1, s2;
    ht_pair(h ? rz : rx, f1, f2);
    ht_pair(ry, s1, s2);
    bq = make_uint4(f1, f2, h ? 0x3C003C00u : s1, h ? 0u : s2);
}

// ---- query side of the SIGN filter (STRACK, nn1_brute.hip).  The filter value of a (query, record) pair, G = w - 2 r.t'' (scaled units),
// bounds the exact distance from below: d2 scale^2 >= Rs + G with Rs = KAPPA |r|^2 - 2^-22 (header of nn1_btrack_kernel).  A record can
// only matter to a query whose best candidate so far lies at thr if d2 <= thr, i.e. if G <= theta := thr scale^2 - Rs.  The two K-slots
// the f16 form leaves free carry -theta (two f16 pieces against the constants 2^15 and 2^6 of the target rows, HT_THETA_CONSTS), so the
// accumulator of the pair is  E = G - theta_hat  and its SIGN BIT is the answer: the vector ALU ORs the sixteen sign bits of a lane
// (v_or3_b32, full rate) instead of taking minima (v_min3_f32, half rate) and tracking first / second minimum and their chunk.
// Exactness.  Let S be the exact sum of the 14 data products and theta_hat = hi 2^15 + lo 2^6 the value the pieces represent.  The one
// assumption of the f16 form — the matrix pipe accumulates with an error <= 16 u sum |a b| (u = 2^-24; measured on the device by
// mfma_verdict: <= 8 u or the form is not used) — gives  E <= S - theta_hat + 16 u (sum_data |a b| + 1.01 |theta_hat|),  and the existing
// bound says  d2 scale^2 >= Rs + S + 16 u sum_data |a b|.  Hence  d2 <= thr  =>  E <= theta - theta_hat + 16.2 u |theta_hat|,  which is
// negative as soon as  theta_hat >= theta (1 + 17 u) + eps:  st_theta rounds the f32 value of theta (one fma: within u |theta|) up by
// 2^-19 |theta| = 32 u |theta|, cuts it with the first piece rounded toward zero and the second rounded UP (by 2^-9 of itself + 2^-23:
// an f16 conversion toward zero loses < 2^-10 of a normal value, < 2^-24 of a subnormal one), and clamps it to +-65 000 x 2^15: beyond
// that every |S| <= 2.5e7 is decided by the clamped value's sign alone (a huge threshold flags everything, a hugely negative one
// nothing).  A flag that is raised needlessly costs one exact evaluation of 16 records; a flag that is missed cannot happen.
// A query without finite coordinates or without a candidate gets thr = -inf (theta at its lower limit: no flag, ever); the caller scans
// such a query's slice exactly.
// the two threshold slots of the upper half-lane: f16 (-hi, -lo) with hi 2^15 + lo 2^6 >= theta (1 + 2^-19) (see above); thr in the
// cloud's units (a finite d2, or -inf for "never"), sc2 = scale^2
__device__ __forceinline__ uint32_t st_theta(float thr, float sc2, float Rs)
{
    typedef __fp16 h2 __attribute__((ext_vector_type(2)));
    constexpr float LIM = 2129920000.0f;                      // 65 000 x 2^15
    float th = __builtin_amdgcn_fmed3f(__builtin_fmaf(thr, sc2, -Rs), -LIM, LIM);            // (+-inf products land on the limits)
    th = __builtin_fmaf(fabsf(th), 1.9073486328125e-06f, th);                                // + 2^-19 |theta|
    const h2 a = __builtin_amdgcn_cvt_pkrtz(th * 3.0517578125e-05f, 0.0f);                    // hi = rtz(theta / 2^15)
    const float rem = __builtin_fmaf(-(float)a.x, 32768.0f, th);                              // exact
    float lo = rem * 0.015625f;
    lo = __builtin_fmaf(fabsf(lo), 0.001953125f, lo) + 1.1920928955078125e-07f;              // up: 2^-9 of itself + 2^-23
    const h2 b = __builtin_amdgcn_cvt_pkrtz(lo, 0.0f);
    return ((__builtin_bit_cast(uint32_t, a) ^ 0x8000u) & 0xFFFFu) | ((__builtin_bit_cast(uint32_t, b) ^ 0x8000u) << 16);
}

// One lane builds the WHOLE operand of one (query, super-tile): the words the lower half-lane of the query's column feeds the MFMA
// (P: x pieces, y pieces) and the upper half-lane's (Q: z pieces, the ones against w's pieces, the threshold slots).  The kernel lets
// lane (n, h) do this for query n of group 2 p + h and exchanges the halves with four v_permlane32_swap_b32 — one setup per query and
// super-tile instead of two that each computed everything and kept half.  The query must be finite (the kernel replaces others by 0
// and gives them thr = -inf).
__device__ __forceinline__ void st_setup(float qx, float qy, float qz, const float4 C, float thr, float sc2, uint32_t P[4], uint32_t Q[4])
{
    constexpr float KAPPA = 0.99999237060546875f;             // 1 - 2^-17
    const float sc = C.w;
    const float rx = __builtin_amdgcn_fmed3f((qx - C.x) * sc, -32000.0f, 32000.0f), ry = __builtin_amdgcn_fmed3f((qy - C.y) * sc, -32000.0f, 32000.0f),
                rz = __builtin_amdgcn_fmed3f((qz - C.z) * sc, -32000.0f, 32000.0f);
    const float Rs = __builtin_fmaf(__builtin_fmaf(rz, rz, __builtin_fmaf(ry, ry, rx * rx)), KAPPA, -PCR_HT_ABS_SLACK);
    ht_pair(rx, P[0], P[1]);
    ht_pair(ry, P[2], P[3]);
    ht_pair(rz, Q[0], Q[1]);
    Q[2] = 0x3C003C00u;
    Q[3] = st_theta(thr, sc2, Rs);
}

// ---- LEVEL 1 of the hierarchical sign filter (STRACK3, nn1_sphere.hpp): one MFMA row per CHUNK of 16 records instead of one per record.  A chunk
// with bounding sphere (c, rho) cannot hold a record at or below a query's threshold thr unless |r - c| <= s + rho (s = sqrt(thr), scaled
// units of the LEVEL-1 super-tile of 4 096 records: centre C1, power-of-two scale with |t - C1| scale <= 2^7), i.e. unless
//     F = (|c|^2 - rho^2) - 2 r.c - 2 s rho - (s^2 - |r|^2) <= 0 :
// the same bilinear form as the record filter with w -> W = |c|^2 - rho^2, ONE more product (-2 rho)(s) — it takes the K-slot of the z
// coordinate's (t2, r2) piece product, which is dropped (<= 2^-22 |2 c_z r_z| <= 4 u (Q + W): inside the 24 u the budget of the f16 form
// leaves unused) — and the threshold pieces in the last two slots as before.  What makes it a theorem (header of st_setup_l1):
//   * the centre is the value its two f16 pieces REPRESENT (c~ = p1 + p2, chosen by the index build), so it carries no rounding at all,
//     and rho >= the largest distance of the chunk's records from THAT point (f32 arithmetic, rounded up by 2^-18 + 2^-12, then up to
//     an f16: the slot holds -2 rho exactly and W uses the same value);
//   * W is stored below its value by 2^-16 (|c~|^2 + rho^2) (the pieces' rounding toward zero, the accumulation of its slots and of the
//     cross term's target share), s above its value by 2^-11 (f16, rounded up: covers the A1-versus-true mismatch of the distance that
//     defines the threshold and the accumulation of the 2 rho s slot), the threshold by 2^-18 |theta| (st_theta_l1: 64 u, of which 16.2 u
//     pay for its own slots' accumulation and 20 u for s'^2 - s^2);
//   * a chunk whose records spread over more than 2^7 scaled units (16 Morton-consecutive records across half the super-tile: padding,
//     pathological clouds) gets the sphere (C1, 222 >= 2^7 sqrt 3) that holds every record of the super-tile; a chunk without a finite
//     record gets W = +inf (never flagged).
// The assumption is the f16 form's own (accumulation error <= 16 u sum |a b|: mfma_verdict); the device self-test of this form
// (pcr_selftest_sphere_f16) checks the statement itself: no (query, chunk) pair with a record at or below the threshold without its sign.
constexpr int BT_L1_SUPER = 4096;      // records per level-1 super-tile: 256 chunks = 8 level-1 tiles of 32 chunks = 16 super-tiles
constexpr int BT_L0_SUPER = 131072;    // records per level-0 super-tile: 256 level-1 tiles = 8 level-0 tiles of 32 rows (one row per level-1 tile)

// f16 value >= v (v >= 0): rounded toward zero after adding 2^-9 of itself + 2^-23 (an f16 conversion toward zero loses < 2^-10 of a
// normal value, < 2^-24 of a subnormal one); returned as the 16-bit pattern and as the float it represents
__device__ __forceinline__ uint32_t f16_up(float v, float& rep)
{
    typedef __fp16 h2 __attribute__((ext_vector_type(2)));
    const h2 a = __builtin_amdgcn_cvt_pkrtz(__builtin_fmaf(v, 0.001953125f, v) + 1.1920928955078125e-07f, 0.0f);
    rep = (float)a.x;
    return __builtin_bit_cast(uint32_t, a) & 0xFFFFu;
}

// the sphere rows of the level-1 form in two steps, so that a row can stand for any set of records (a chunk of 16: one thread; a level-1 tile of
// 512: one wave, nn1 level 0):  sph_centre — the centre = what the two f16 pieces of -2 (middle of the set's box) represent;  sph_finish — the
// row's two 16-byte words from that centre and r2 = the largest squared distance of a member from it (any = the set has a finite member)
__device__ __forceinline__ void sph_centre(const float (&mn)[3], const float (&mx)[3], uint32_t (&p)[3][2], float (&c)[3])
{
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float mid = 0.5f * mn[k] + 0.5f * mx[k];
        ht_split(-2.0f * mid, p[k][0], p[k][1]);
        typedef __fp16 h1 __attribute__((ext_vector_type(2)));
        const float a1 = (float)__builtin_bit_cast(h1, p[k][0]).x, a2 = (float)__builtin_bit_cast(h1, p[k][1]).x;
        c[k] = -0.5f * (a1 + a2);                             // exact: two f16 values of at most 22 significant bits in all
    }
}
__device__ __forceinline__ void sph_finish(bool any, uint32_t (&p)[3][2], float (&c)[3], float r2, uint4& lo, uint4& hi)
{
    if (!any) {                                               // no finite record: W = +inf, never flagged
        lo = make_uint4(0u, 0u, 0u, 0u);
        hi = make_uint4(0u, 0u, 0x7C00u, HT_THETA_CONSTS);
        return;
    }
    float rho = __builtin_fmaf(sqrtf(r2), 3.814697265625e-06f, sqrtf(r2)) + 2.44140625e-04f;      // up: 2^-18 of itself + 2^-12
    if (!(rho <= 128.0f)) {                                   // spread over more than half the super-tile: the sphere of the whole super-tile
        c[0] = c[1] = c[2] = 0.0f;
        p[0][0] = p[0][1] = p[1][0] = p[1][1] = p[2][0] = p[2][1] = 0u;
        rho = 222.0f;
    }
    float rho16;
    (void)f16_up(rho, rho16);
    uint32_t m2, m2lo;
    ht_split(-2.0f * rho16, m2, m2lo);                        // (-2 rho16 is an f16 value: the second piece is zero)
    const float wc = (c[0] * c[0] + c[1] * c[1]) + c[2] * c[2], wr = rho16 * rho16;
    const float W = __builtin_fmaf(-wr, 1.0000152587890625f, wc * 0.9999847412109375f) - 9.5367431640625e-07f;   // wc (1 - 2^-16) - wr (1 + 2^-16) - 2^-20
    uint32_t w1, w2;
    ht_split(W, w1, w2);
    lo = make_uint4(p[0][0] | (p[0][1] << 16), p[0][0] | (p[0][1] << 16), p[1][0] | (p[1][1] << 16), p[1][0] | (p[1][1] << 16));
    hi = make_uint4(p[2][0] | (p[2][1] << 16), p[2][0] | (m2 << 16), w1 | (w2 << 16), HT_THETA_CONSTS);
}

// the two lanes' words of MFMA row `m` of a level-1 tile for one chunk: t[16][3] = the chunk's records in the scaled units of its level-1
// super-tile (exact scaling), fin[16] = finite and inside the super-tile's range
__device__ __forceinline__ void l1_chunk_operand(const float (&tx)[16], const float (&ty)[16], const float (&tz)[16], const bool (&fin)[16], uint4& lo, uint4& hi)
{
    float mn[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, mx[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
    bool any = false;
#pragma unroll
    for (int j = 0; j < 16; j++)
        if (fin[j]) {
            any = true;
            mn[0] = fminf(mn[0], tx[j]); mx[0] = fmaxf(mx[0], tx[j]);
            mn[1] = fminf(mn[1], ty[j]); mx[1] = fmaxf(mx[1], ty[j]);
            mn[2] = fminf(mn[2], tz[j]); mx[2] = fmaxf(mx[2], tz[j]);
        }
    uint32_t p[3][2] = { { 0u, 0u }, { 0u, 0u }, { 0u, 0u } };
    float c[3] = { 0.0f, 0.0f, 0.0f };
    float r2 = 0.0f;
    if (any) {
        sph_centre(mn, mx, p, c);
#pragma unroll
        for (int j = 0; j < 16; j++)
            if (fin[j]) {
                const float dx = tx[j] - c[0], dy = ty[j] - c[1], dz = tz[j] - c[2];
                r2 = fmaxf(r2, (dx * dx + dy * dy) + dz * dz);
            }
    }
    sph_finish(any, p, c, r2, lo, hi);
}

// the threshold slots of the level-1 form: as st_theta, rounded up by 2^-18 |theta| (see above)
__device__ __forceinline__ uint32_t st_theta_l1(float thr, float sc2, float Rs)
{
    typedef __fp16 h2 __attribute__((ext_vector_type(2)));
    constexpr float LIM = 2129920000.0f;                      // 65 000 x 2^15
    float th = __builtin_amdgcn_fmed3f(__builtin_fmaf(thr, sc2, -Rs), -LIM, LIM);
    th = __builtin_fmaf(fabsf(th), 3.814697265625e-06f, th);                                  // + 2^-18 |theta|
    const h2 a = __builtin_amdgcn_cvt_pkrtz(th * 3.0517578125e-05f, 0.0f);                    // hi = rtz(theta / 2^15)
    const float rem = __builtin_fmaf(-(float)a.x, 32768.0f, th);                              // exact
    float lo = rem * 0.015625f;
    lo = __builtin_fmaf(fabsf(lo), 0.001953125f, lo) + 1.1920928955078125e-07f;              // up: 2^-9 of itself + 2^-23
    const h2 b = __builtin_amdgcn_cvt_pkrtz(lo, 0.0f);
    return ((__builtin_bit_cast(uint32_t, a) ^ 0x8000u) & 0xFFFFu) | ((__builtin_bit_cast(uint32_t, b) ^ 0x8000u) << 16);
}

// query side of the level-1 form for one (query, level-1 super-tile): P = the lower half-lane's words (x pieces, y pieces), Q = the upper
// half-lane's (z: r1 r1 | r2, s | 1 1 | threshold pieces).  s = sqrt(thr) scale, rounded UP into an f16 (0 for "never": thr = -inf).
// Q2 (optional): the upper half-lane's words of the RECORD form in the same scale (st_setup's: z: r1 r1 | r2 r2 | 1 1 | st_theta) — level 2 of
// STRACK3 filters the records of a level-1 super-tile with operands in that super-tile's scale, so both forms come out of one setup.
__device__ __forceinline__ void st_setup_l1(float qx, float qy, float qz, const float4 C, float thr, float sc2, uint32_t P[4], uint32_t Q[4], uint32_t* Q2 = nullptr)
{
    constexpr float KAPPA = 0.99999237060546875f;             // 1 - 2^-17
    const float sc = C.w;
    const float rx = __builtin_amdgcn_fmed3f((qx - C.x) * sc, -32000.0f, 32000.0f), ry = __builtin_amdgcn_fmed3f((qy - C.y) * sc, -32000.0f, 32000.0f),
                rz = __builtin_amdgcn_fmed3f((qz - C.z) * sc, -32000.0f, 32000.0f);
    const float Rs = __builtin_fmaf(__builtin_fmaf(rz, rz, __builtin_fmaf(ry, ry, rx * rx)), KAPPA, -PCR_HT_ABS_SLACK);
    ht_pair(rx, P[0], P[1]);
    ht_pair(ry, P[2], P[3]);
    uint32_t z2;
    ht_pair(rz, Q[0], z2);
    const float s2 = __builtin_amdgcn_fmed3f(thr * sc2, 0.0f, 2129920000.0f);                // (thr = -inf: 0)
    float srep;
    const uint32_t s16 = f16_up(sqrtf(s2), srep);
    Q[1] = (z2 & 0xFFFFu) | (s16 << 16);
    Q[2] = 0x3C003C00u;
    Q[3] = st_theta_l1(thr, sc2, Rs);
    if (Q2) { Q2[0] = Q[0]; Q2[1] = z2; Q2[2] = 0x3C003C00u; Q2[3] = st_theta(thr, sc2, Rs); }
}

// the fine lattice cell of a coordinate exactly as bt_keys_kernel bins it (monotone non-decreasing in v: the box of a pass maps to a cell range)
__device__ __forceinline__ uint32_t bt_fine_cell(float v, float lo, float inv) { return (uint32_t)fminf(fmaxf((v - lo) * inv, 0.0f), 1023.0f); }
// 10 bits -> every third bit (bt_keys_kernel's key: bit 3 b + k = bit b of c[k])
__device__ __forceinline__ uint32_t spread3_10(uint32_t x)
{
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8)) & 0x0300F00Fu;
    x = (x | (x << 4)) & 0x030C30C3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}
__device__ __forceinline__ uint32_t bt_morton(uint32_t cx, uint32_t cy, uint32_t cz) { return spread3_10(cx) | (spread3_10(cy) << 1) | (spread3_10(cz) << 2); }

void bt_free(BtIndex* b);
// builds (and caches on tgt) the index if it is not there yet
int bt_ensure(pcr_ctx* ctx, const pcr_cloud* tgt);
// ... and the extras of the sign tile search (needs tgt->grid and a safe tgt->bt; rebuilt when the grid changed)
int bt_ensure_tile(pcr_ctx* ctx, const pcr_cloud* tgt);
// ... and level 1 of the two-level sign filter (needs a safe tgt->bt)
int bt_ensure_l1(pcr_ctx* ctx, const pcr_cloud* tgt);
// builds (and caches on tgt) the 1-NN grid if needed, then groups the queries `src` by coarse cell -> ctx->qperm
int grid_prepare_queries(pcr_ctx* ctx, const pcr_cloud* tgt, const pcr_cloud* src);
// replaces *work by its cell-sorted copy (ctx->work_orig = original indices); no-op for empty clouds / tune grid_sort_work = 2
int grid_sort_working_cloud(pcr_ctx* ctx, const pcr_cloud* tgt, pcr_cloud** work, bool in_place);
// the same for a brute-force loop over the matrix-core index: *work in the Morton order of the target's super-tiles (no-op without that index)
int bt_sort_working_cloud(pcr_ctx* ctx, const pcr_cloud* tgt, pcr_cloud** work, bool in_place);

}  // namespace pcr

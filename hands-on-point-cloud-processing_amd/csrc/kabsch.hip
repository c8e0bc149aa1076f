// kabsch.hip — the streaming (HBM-bound) passes of one ICP iteration on gfx950:
//   * kabsch_partial_kernel / kabsch_final_kernel: A7 accumulation — the sums of p, q and q p^T over the kept pairs
//     (Homework9/hw9/src/registration.cpp:936-940,964-985).  The sums are EXACT and therefore independent of the order of the
//     additions: every term is cut into 40-bit integer limbs on one fixed-point grid (numerics.hpp), integer partial sums are
//     reduced per wavefront with DPP shuffles, per workgroup through LDS, across workgroups by one final workgroup and across
//     ranks by the one all-reduce of the iteration, with carry propagation between the levels.  The same bits come out for
//     any launch geometry, any visiting order of the queries (original order, cell order) and any number of GPUs.
//     Algorithmic traffic: 28 B per kept pair (12 B source + 8 B key + 12 B gathered target - the key carries idx and d2).
//   * transform_kernel: A8 transformCloudInplace (registration.cpp:165-178), f32, unfused, in place,
//     24 B per point, float4-vectorised over the SoA arrays.
#include "grid_common.hpp"
#include "numerics.hpp"

#include <cmath>

#pragma clang fp contract(off)

namespace pcr {

constexpr int KB_BLOCK = 256;
constexpr int KB_NV = 16;             // 3 + 3 + 9 sums + count
constexpr int KB_NL = num::KB_NL;     // 55 normalised limbs (numerics.hpp)
constexpr int KB_ROW = 58;            // a partial row: KB_NL limbs, [55] last-kept key (u64 bits), [56] overflow flag, [57] pad
constexpr int KB_MAX_BLOCKS = 8192;   // == the capacity of ctx->partials (api.cpp)
static_assert(KB_ROW * KB_MAX_BLOCKS == 8192 * 58, "api.cpp sizes ctx->partials as 8192 x 58 doubles");

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_down(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

// one coordinate term into its two limbs (y = v * 2^(80 - e), |y| < 2^80)
__device__ __forceinline__ void acc2(double v, double sc, double& a0, double& a1)
{
    const double y = v * sc;                                  // exact: power of two
    const double l1 = trunc(y * num::KB_2mW);
    const double r = fma(-l1, num::KB_2W, y);                 // exact remainder, |r| < 2^40
    a1 += l1;
    a0 += trunc(r);
}

// one product term into its three limbs (y = v * 2^(120 - 2e), |y| < 2^120)
__device__ __forceinline__ void acc3(double v, double sp, double& a0, double& a1, double& a2)
{
    const double y = v * sp;
    const double l2 = trunc(y * (num::KB_2mW * num::KB_2mW));
    const double r = fma(-l2, num::KB_2W * num::KB_2W, y);    // |r| < 2^80
    const double l1 = trunc(r * num::KB_2mW);
    const double r2 = fma(-l1, num::KB_2W, r);                // |r2| < 2^40
    a2 += l2;
    a1 += l1;
    a0 += trunc(r2);
}

// the workgroup's reduction of the per-thread limbs: tn[KB_NL] (normalised per thread) -> row[] in LDS, normalised again
__host__ __device__ constexpr bool is_carry_slot(int k) { return k < 18 ? (k % 3 == 2) : (k < 54 && (k - 18) % 4 == 3); }

// Wave totals of N values per lane by a butterfly that HALVES what a lane holds at every step: with partner distance D the lanes
// whose bit D is clear keep the lower half of their values and send the upper half, the others the reverse; both add what they
// receive.  After the six steps lane l holds the wave's total of ONE value, number slot(l) (or nothing: slot < 0) — N - 1 + (a few
// for odd halves) exchanged values per lane in all, where one shuffle tree per value costs 6 N: 41 against 240 for the 40 limbs of
// the usual case.  The totals are integers below 2^53, so the order of the additions does not matter (numerics.hpp).
template <int N, int D>
__device__ __forceinline__ void bfly_halve(double (&v)[KB_NL], bool up)
{
    constexpr int H = (N + 1) / 2;
#pragma unroll
    for (int i = 0; i < H; i++) {
        const double a = v[i], b = (H + i < N) ? v[H + i] : 0.0;
        const double send = up ? a : b, keep = up ? b : a;
        v[i] = keep + __shfl_xor(send, D, 64);
    }
}

template <int N>
__device__ __forceinline__ int wave_totals(double (&v)[KB_NL], int lane)
{
    static_assert(N >= 1 && N <= 64, "six halvings must leave one value");
    constexpr int N1 = (N + 1) / 2, N2 = (N1 + 1) / 2, N3 = (N2 + 1) / 2, N4 = (N3 + 1) / 2, N5 = (N4 + 1) / 2;
    bfly_halve<N, 32>(v, (lane & 32) != 0);
    bfly_halve<N1, 16>(v, (lane & 16) != 0);
    bfly_halve<N2, 8>(v, (lane & 8) != 0);
    bfly_halve<N3, 4>(v, (lane & 4) != 0);
    bfly_halve<N4, 2>(v, (lane & 2) != 0);
    bfly_halve<N5, 1>(v, (lane & 1) != 0);
    // which value this lane ended up with: its position in the arrays of the six steps, innermost first
    int p = 0;
    bool ok = true;
    p += (lane & 1) ? (N5 + 1) / 2 : 0;  ok = ok && p < N5;
    p += (lane & 2) ? (N4 + 1) / 2 : 0;  ok = ok && p < N4;
    p += (lane & 4) ? (N3 + 1) / 2 : 0;  ok = ok && p < N3;
    p += (lane & 8) ? (N2 + 1) / 2 : 0;  ok = ok && p < N2;
    p += (lane & 16) ? (N1 + 1) / 2 : 0; ok = ok && p < N1;
    p += (lane & 32) ? (N + 1) / 2 : 0;  ok = ok && p < N;
    return ok ? p : -1;
}

constexpr int KB_NLC = 40;            // the limbs that are not carry slots (12 + 27) + the count

// carries: false when the carry slots of tn[] are known to be zero (no per-thread normalisation): they are not exchanged at all
// BFLY: the halving butterfly over the 40 live limbs (needs carries == false); it keeps all of them in registers at once (~160
// VGPRs), so only the instance for few terms per thread — where the reduction IS the pass — is built with it
template <bool BFLY>
__device__ __forceinline__ void block_reduce_limbs(double (&tn)[KB_NL], unsigned long long lastkey, int overflow,
                                                   double (&red)[KB_BLOCK / 64][KB_NL], double (&row)[KB_ROW], bool carries)
{
    __shared__ unsigned long long red_key[KB_BLOCK / 64];
    __shared__ int red_ovf[KB_BLOCK / 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (!BFLY) {
#pragma unroll
        for (int k = 0; k < KB_NL; k++) {
            double sum = 0.0;
            if (carries || !is_carry_slot(k)) sum = wave_sum(tn[k]);      // 64 integers below 2^46 (+ small carries): exact
            if (lane == 0) red[wave][k] = sum;
        }
    } else {
        int j = 0;
#pragma unroll
        for (int k = 0; k < KB_NL; k++)
            if (!is_carry_slot(k)) tn[j++] = tn[k];                       // compact: the 40 live limbs (static indices)
        const int p = wave_totals<KB_NLC>(tn, lane);
        if (p >= 0) {
            // live limb p -> its slot of the row: coordinates 2 of every 3 slots, products 3 of every 4, the count last
            const int slot = p < 12 ? 3 * (p / 2) + p % 2 : (p < 39 ? 18 + 4 * ((p - 12) / 3) + (p - 12) % 3 : 54);
            red[wave][slot] = tn[0];
        }
        if (lane < 15) red[wave][lane < 6 ? 3 * lane + 2 : 18 + 4 * (lane - 6) + 3] = 0.0;      // the 15 carry slots
    }
    const unsigned long long km = wave_max_u64(lastkey);
    const int ov = __any(overflow) ? 1 : 0;
    if (lane == 0) { red_key[wave] = km; red_ovf[wave] = ov; }
    __syncthreads();
    if (threadIdx.x < KB_NL) {
        const int k = threadIdx.x;
        row[k] = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
    } else if (threadIdx.x == KB_NL) {
        unsigned long long m = red_key[0];
        for (int w = 1; w < KB_BLOCK / 64; w++) m = red_key[w] > m ? red_key[w] : m;
        row[55] = __longlong_as_double((long long)m);
        row[56] = (red_ovf[0] | red_ovf[1] | red_ovf[2] | red_ovf[3]) ? 1.0 : 0.0;
        row[57] = 0.0;
    }
    __syncthreads();
    if (threadIdx.x < 6) num::limbs_normalize(row + 3 * threadIdx.x, 2);
    else if (threadIdx.x < 15) num::limbs_normalize(row + 18 + 4 * (threadIdx.x - 6), 3);
    __syncthreads();
}

// partials layout: [block][KB_ROW].  RECORDS: the target of pair i is grid record wpos[i] (one 16-byte gather from the cell-
// sorted records, which neighbouring queries share) instead of the three 4-byte gathers at its original index.
// orig: original index of query i (the working cloud of the grid ICP is cell-sorted) or nullptr = i.
// SMALL: at most 4 pairs per thread (clouds up to ~1 M points): no per-thread carries, butterfly reduction.
template <bool RECORDS, bool SMALL>
__global__ __launch_bounds__(KB_BLOCK) void kabsch_partial_kernel(
    const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz,
    const float* __restrict__ tx, const float* __restrict__ ty, const float* __restrict__ tz,
    const float4* __restrict__ records, const uint32_t* __restrict__ wpos, const uint32_t* __restrict__ orig,
    const unsigned long long* __restrict__ keys, uint32_t ns, uint32_t nt, float max_corr, KabschPlan plan,
    double* __restrict__ partials)
{
    double c0[6], c1[6], p0[9], p1[9], p2[9], cnt = 0.0;
#pragma unroll
    for (int k = 0; k < 6; k++) { c0[k] = 0.0; c1[k] = 0.0; }
#pragma unroll
    for (int k = 0; k < 9; k++) { p0[k] = 0.0; p1[k] = 0.0; p2[k] = 0.0; }
    unsigned long long lastkey = 0;
    int overflow = 0;
    // contiguous chunk per block, strided by lane inside it (coalesced); the order does not matter for the result
    const uint32_t per_block = (ns + gridDim.x - 1) / gridDim.x;
    const uint32_t lo = min(blockIdx.x * per_block, ns);
    const uint32_t hi = min(lo + per_block, ns);
    // Software pipeline: the coalesced loads of the NEXT pair are requested before the current one is consumed, and the target gather
    // of the current pair is issued first thing, unconditionally (clamped position): one exposed round trip per iteration instead of
    // two, twice the bytes in flight per wave (the pass is bound by latency x bytes in flight, not by bandwidth: DESIGN.md 6b).
    uint32_t i = lo + threadIdx.x;
    unsigned long long key_n = 0;
    float p_n0 = 0.f, p_n1 = 0.f, p_n2 = 0.f;
    uint32_t wp_n = 0, og_n = i;
    if (i < hi) {
        key_n = keys[i]; p_n0 = sx[i]; p_n1 = sy[i]; p_n2 = sz[i];
        if (RECORDS) wp_n = wpos[i];
        if (orig) og_n = orig[i];
    }
    const uint32_t last_rec = nt ? nt - 1 : 0u;
    for (; i < hi; i += KB_BLOCK) {
        const unsigned long long key = key_n;
        const float pf0 = p_n0, pf1 = p_n1, pf2 = p_n2;
        const uint32_t og = og_n;
        float4 rec = make_float4(0.f, 0.f, 0.f, 0.f);
        if (RECORDS) rec = records[min(wp_n, last_rec)];
        const uint32_t in = i + KB_BLOCK;
        if (in < hi) {
            key_n = keys[in]; p_n0 = sx[in]; p_n1 = sy[in]; p_n2 = sz[in];
            if (RECORDS) wp_n = wpos[in];
            og_n = orig ? orig[in] : in;
        }
        const uint32_t d2b = (uint32_t)(key >> 32);
        const float d2 = __uint_as_float(d2b);
        const uint32_t j = (uint32_t)(key & 0xFFFFFFFFull);
        if (d2 < max_corr && j < nt) {                       // registration.cpp:936
            float qf0, qf1, qf2;
            if (RECORDS) {
                qf0 = rec.x; qf1 = rec.y; qf2 = rec.z;
            } else {
                qf0 = tx[j]; qf1 = ty[j]; qf2 = tz[j];
            }
            // 2^e bounds every coordinate of a kept pair (kabsch_plan); a source beyond it would overflow its limbs
            if (!(fabsf(pf0) < plan.lim && fabsf(pf1) < plan.lim && fabsf(pf2) < plan.lim)) { overflow = 1; continue; }
            const double P[3] = { pf0, pf1, pf2 }, Q[3] = { qf0, qf1, qf2 };
#pragma unroll
            for (int c = 0; c < 3; c++) { acc2(P[c], plan.sc, c0[c], c1[c]); acc2(Q[c], plan.sc, c0[3 + c], c1[3 + c]); }
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int c = 0; c < 3; c++) acc3(Q[r] * P[c], plan.sp, p0[3 * r + c], p1[3 * r + c], p2[3 * r + c]);
            cnt += 1.0;
            // the LAST kept pair of the reference's loop (registration.cpp:939) = the kept pair with the highest original index
            const unsigned long long lk = ((unsigned long long)og << 32) | d2b;
            lastkey = lk > lastkey ? lk : lastkey;
        }
    }
    // per-thread carry propagation only when a thread may hold many terms (limbs below 2^40 x terms must stay below 2^53 through
    // the 256-thread sum): at most 16 terms per thread need none
    const bool many = !SMALL && per_block > 16u * KB_BLOCK;
    double tn[KB_NL];
#pragma unroll
    for (int c = 0; c < 6; c++) { tn[3 * c] = c0[c]; tn[3 * c + 1] = c1[c]; tn[3 * c + 2] = 0.0; if (many) num::limbs_normalize(tn + 3 * c, 2); }
#pragma unroll
    for (int k = 0; k < 9; k++) {
        tn[18 + 4 * k] = p0[k]; tn[19 + 4 * k] = p1[k]; tn[20 + 4 * k] = p2[k]; tn[21 + 4 * k] = 0.0;
        if (many) num::limbs_normalize(tn + 18 + 4 * k, 3);
    }
    tn[54] = cnt;
    __shared__ double red[KB_BLOCK / 64][KB_NL];
    __shared__ double row[KB_ROW];
    block_reduce_limbs<SMALL>(tn, lastkey, overflow, red, row, many);
    if (threadIdx.x < KB_ROW) partials[(size_t)blockIdx.x * KB_ROW + threadIdx.x] = row[threadIdx.x];
}

// one workgroup of KF_BLOCK threads = 16 groups of 64 lanes: the block rows -> ONE normalised row in LDS (limbs, last-kept key,
// overflow flag).  Lane k of a group owns column k (coalesced 464-byte row reads), group g takes rows g, g + 16, ...; column sums
// of <= 1024 integers below 2^40 are exact.  Thread k < 16 then converts moment k: out[0..15] = the moments,
// [16] original index of the last kept pair or -1, [17] its d2, [18] overflow flag.
constexpr int KF_BLOCK = 1024;
constexpr int KF_GROUPS = KF_BLOCK / 64;

__device__ __forceinline__ void reduce_rows(const double* __restrict__ partials, uint32_t n_blocks, double (&red)[KF_GROUPS][64], double (&row)[KB_ROW])
{
    const int k = threadIdx.x & 63, grp = threadIdx.x >> 6;
    double acc = 0.0;
    unsigned long long key = 0;
    if (k < 57) {
        // more rows than groups: eight rows of a group in flight at once (a row that does not exist reads as 0.0, neutral for the sum,
        // the key maximum and the flag alike) — one memory round trip per eight rows instead of one per row; the additions are exact,
        // any order.  Few rows (small clouds, where this kernel is pure latency): the one-row loop, a fraction of the code to fetch.
        if (n_blocks <= (uint32_t)KF_GROUPS) {
            if ((uint32_t)grp < n_blocks) {
                const double v = partials[(size_t)grp * KB_ROW + k];
                if (k == 55) key = (unsigned long long)__double_as_longlong(v);
                else if (k == 56) acc = v != 0.0 ? 1.0 : 0.0;
                else acc = v;
            }
        } else {
            constexpr int INF = 8;
            for (uint32_t b = grp; b < n_blocks; b += KF_GROUPS * INF) {
                double v[INF];
#pragma unroll
                for (int j = 0; j < INF; j++) {
                    const uint32_t bb = b + (uint32_t)j * KF_GROUPS;
                    v[j] = bb < n_blocks ? partials[(size_t)bb * KB_ROW + k] : 0.0;
                }
#pragma unroll
                for (int j = 0; j < INF; j++) {
                    if (k == 55) { const unsigned long long lk = (unsigned long long)__double_as_longlong(v[j]); key = lk > key ? lk : key; }
                    else if (k == 56) acc = (acc != 0.0 || v[j] != 0.0) ? 1.0 : 0.0;
                    else acc += v[j];
                }
            }
        }
    }
    red[grp][k] = k == 55 ? __longlong_as_double((long long)key) : acc;
    __syncthreads();
    if (threadIdx.x < 57) {
        const int c = threadIdx.x;
        if (c == 55) {
            unsigned long long m = 0;
            for (int g = 0; g < KF_GROUPS; g++) { const unsigned long long lk = (unsigned long long)__double_as_longlong(red[g][c]); m = lk > m ? lk : m; }
            row[c] = __longlong_as_double((long long)m);
        } else {
            double t = 0.0;
            for (int g = 0; g < KF_GROUPS; g++) t += red[g][c];
            row[c] = c == 56 ? (t != 0.0 ? 1.0 : 0.0) : t;
        }
    }
    if (threadIdx.x == 57) row[57] = 0.0;
    __syncthreads();
    if (threadIdx.x < 6) num::limbs_normalize(row + 3 * threadIdx.x, 2);
    else if (threadIdx.x < 15) num::limbs_normalize(row + 18 + 4 * (threadIdx.x - 6), 3);
    __syncthreads();
}

__device__ __forceinline__ void row_to_out18(const double (&row)[KB_ROW], int e, double* out)
{
    const int k = threadIdx.x;
    if (k < 6) out[k] = num::limbs_value(row + 3 * k, 2, e - 2 * num::KB_W);
    else if (k < 15) out[k] = num::limbs_value(row + 18 + 4 * (k - 6), 3, 2 * e - 3 * num::KB_W);
    else if (k == 15) out[15] = row[54];
    else if (k == 16) {
        const unsigned long long lk = (unsigned long long)__double_as_longlong(row[55]);
        out[16] = row[54] > 0.0 ? (double)(uint32_t)(lk >> 32) : -1.0;
        out[17] = row[54] > 0.0 ? (double)__uint_as_float((uint32_t)(lk & 0xFFFFFFFFull)) : 0.0;
        out[18] = row[56];
    }
}

__global__ __launch_bounds__(KF_BLOCK) void kabsch_final_kernel(const double* __restrict__ partials, uint32_t n_blocks, int e, double* __restrict__ out)
{
    __shared__ double red[KF_GROUPS][64];
    __shared__ double row[KB_ROW];
    reduce_rows(partials, n_blocks, red, row);
    row_to_out18(row, e, out);
}

// the order of the additions does not matter any more; the geometry only has to keep the pass busy: one pair per thread while
// that needs few workgroups (the pass is latency-bound at 120 k), then a capped number of fatter workgroups — every workgroup
// ends in a 464-byte row that ONE workgroup has to reduce afterwards (8 192 rows cost 0.22 ms at 10 M, 1 024 rows 0.05 ms, and the pass itself runs 0.27 -> 0.14 ms)
static uint32_t kabsch_blocks(pcr_ctx* ctx, size_t ns)
{
    // one pair per thread up to 128 workgroups, then four pairs per thread, then the cap: at 120 k the 469 rows of the one-pair rule
    // cost the reduce 24 us against 15 us for 128 rows, the pass itself the same 18 us (profiles/r02_kabsch_probes.txt)
    uint32_t blocks = (uint32_t)((ns + KB_BLOCK - 1) / KB_BLOCK);
    const uint32_t one_pair = (uint32_t)std::min<int64_t>(KB_MAX_BLOCKS, std::max<int64_t>(1, tune_get(ctx, "kabsch_one_pair_blocks", 128)));
    if (blocks > one_pair) blocks = std::max<uint32_t>(one_pair, (uint32_t)((ns + 4 * KB_BLOCK - 1) / (4 * KB_BLOCK)));
    if (blocks < 1) blocks = 1;
    // (round 4: at most one workgroup per 2 048 pairs as well — a 1 / 8 shard of the 10 M pair ran 1 024 workgroups of 1 220 pairs and its solve reduced
    // 1 024 rows: 0.370 -> 0.352 ms per iteration with 610; the 10 M pair itself stays at the cap)
    const int64_t cap_t = tune_get(ctx, "kabsch_max_blocks", 0);
    const uint32_t cap = (uint32_t)std::min<int64_t>(KB_MAX_BLOCKS, std::max<int64_t>(1, cap_t > 0 ? cap_t : std::min<int64_t>(1024, std::max<int64_t>(128, (int64_t)(ns / 2048)))));
    if (blocks > cap) blocks = cap;
    // exactness before tuning: a thread's limb accumulators are normalised only after its whole chunk, every term adds < 2^40 to
    // each, so a thread may hold < 2^13 terms: never fewer workgroups than 4 096 pairs per thread need (10 M points: 10)
    const uint32_t need = (uint32_t)((ns + (size_t)4096 * KB_BLOCK - 1) / ((size_t)4096 * KB_BLOCK));
    if (blocks < need) blocks = need;
    return blocks;
}

// The fixed-point grid of one accumulation.  Every coordinate of a KEPT pair is bounded by what is known up front and is the
// same on every rank: the target's largest finite |coordinate| (cached on the cloud) plus the gate — a pair is kept only if
// d2 < max_corr (registration.cpp:936), so |p - q| < sqrt(max_corr).  An unbounded gate is capped at 2^20 target extents;
// a kept source beyond 2^e then raises the overflow flag (PCR_ERR_STATE) instead of a wrong sum.
int kabsch_grid_exponent(float amax, float max_corr)
{
    const double a = std::max((double)amax, 1e-30);
    double gate = max_corr > 0.f ? 2.0 * std::sqrt((double)max_corr) : 0.0;     // NaN / <= 0: no pair is kept anyway
    if (!(gate <= a * 1048576.0)) gate = a * 1048576.0;
    const double M = a + gate;
    int e = std::ilogb(M) + 2;                                                      // 2^e > 2 M
    if (e > 140) e = 140;
    return e;
}

int kabsch_plan(pcr_ctx* ctx, const pcr_cloud* tgt, float max_corr, KabschPlan* plan)
{
    float amax = 0.f;
    int rc = cloud_absmax(ctx, tgt, &amax);
    if (rc) return rc;
    const int e = kabsch_grid_exponent(amax, max_corr);
    plan->e = e;
    plan->lim = e >= 128 ? __builtin_inff() : std::ldexp(1.0f, e);
    plan->sc = std::ldexp(1.0, 2 * num::KB_W - e);
    plan->sp = std::ldexp(1.0, 3 * num::KB_W - 2 * e);
    return PCR_OK;
}

int launch_kabsch_partial(pcr_ctx* ctx, const pcr_cloud* tgt, const pcr_cloud* src, float max_corr, const KabschPlan& plan, uint32_t* n_blocks)
{
    const size_t ns = src->n;
    if (ctx->keys_n != ns) return fail(ctx, PCR_ERR_STATE, "kabsch: no matching correspondence pass");
    const uint32_t blocks = kabsch_blocks(ctx, ns);
    // the grid search of an ICP loop leaves the record position of every winner behind (ctx->wpos): gather from the records
    const bool rec = ctx->wpos_valid && ctx->wpos_n == ns && tgt->grid && tgt->grid->records && tune_get(ctx, "kabsch_records", 1) == 1;
    const uint32_t* orig = (ctx->work_orig && ctx->work_orig_n == ns && ctx->work_orig_src == src) ? ctx->work_orig : nullptr;
    {
        ProfScope p(ctx, "kabsch_partial");
        const uint32_t per_block = (uint32_t)((ns + blocks - 1) / blocks);      // as the kernel computes it
        const bool small = per_block <= 4u * KB_BLOCK && tune_get(ctx, "kabsch_bfly", 1) == 1;
#define PCR_KABSCH(R, S, RECS, WPOS)                                                                                                      \
        hipLaunchKernelGGL((kabsch_partial_kernel<R, S>), dim3(blocks), dim3(KB_BLOCK), 0, ctx->stream, src->x(), src->y(), src->z(), tgt->x(),  \
                           tgt->y(), tgt->z(), RECS, WPOS, orig, ctx->keys, (uint32_t)ns, (uint32_t)tgt->n, max_corr, plan, ctx->partials)
        if (rec) { if (small) PCR_KABSCH(true, true, tgt->grid->records, ctx->wpos); else PCR_KABSCH(true, false, tgt->grid->records, ctx->wpos); }
        else { if (small) PCR_KABSCH(false, true, (const float4*)nullptr, (const uint32_t*)nullptr); else PCR_KABSCH(false, false, (const float4*)nullptr, (const uint32_t*)nullptr); }
#undef PCR_KABSCH
    }
    PCR_HIP(ctx, hipGetLastError());
    *n_blocks = blocks;
    return PCR_OK;
}

// dev_out[0..18] = moments, last kept (original index, d2), overflow flag
int launch_kabsch_sums(pcr_ctx* ctx, const pcr_cloud* tgt, const pcr_cloud* src, float max_corr, const KabschPlan& plan)
{
    uint32_t blocks = 0;
    int rc = launch_kabsch_partial(ctx, tgt, src, max_corr, plan, &blocks);
    if (rc) return rc;
    {
        ProfScope p(ctx, "kabsch_final");
        hipLaunchKernelGGL(kabsch_final_kernel, dim3(1), dim3(KF_BLOCK), 0, ctx->stream, ctx->partials, blocks, plan.e, ctx->dev_out);
    }
    PCR_HIP(ctx, hipGetLastError());
    return PCR_OK;
}

// ---------------------------------------------------------------------------------------------- device-resident ICP state
// The pipelined ICP loop (icp.cpp) keeps the whole state machine of registration.cpp:915-1006 on the GPU so that
// iterations are enqueued back to back without a host round trip: nn1 -> kabsch_partial -> icp_update -> transform.

// state machine + Kabsch solve + pose composition, one thread (registration.cpp:939-1002).  `st` is the workgroup's LDS copy of
// the state (icp_state_stage): the step is one thread's serial chain, and every field it touched in HBM was a dependent memory
// round trip of that chain (stop -> stop_after_transform -> last_loss, eps -> unchanged -> T_total ...).
__device__ __forceinline__ void icp_state_step(IcpState* st, const double* sums16, bool any_kept, float last_d2, bool overflow)
{
    if (st->stop) return;
    if (st->stop_after_transform) { st->stop = 1; return; }   // max_iter reached: the loop is over
    if (overflow) { st->overflow = 1; st->stop = 1; return; }  // a kept source beyond the accumulation grid (kabsch_plan)
    float loss = 0.0f;
    if (any_kept) loss = last_d2 * last_d2;                                      // :939
    st->last_pairs = (unsigned long long)sums16[15];
    st->loss = loss;
    if (fabsf(st->last_loss - loss) < st->eps) st->unchanged++;                  // :948-951
    if (st->unchanged > 15) { st->converged = 1; st->stop = 1; return; }         // :954-958
    st->last_loss = loss;                                                        // :961
    float Rd[9], td[3];
    if (num::kabsch_solve(sums16, Rd, td) != 0) { st->empty = 1; st->stop = 1; return; }   // :979-998
    const float T_delta[16] = { Rd[0], Rd[1], Rd[2], td[0], Rd[3], Rd[4], Rd[5], td[1],
                                Rd[6], Rd[7], Rd[8], td[2], 0, 0, 0, 1 };
    num::mat4_mul_f32(T_delta, st->T_total, st->T_total);                        // :1000-1002
    for (int k = 0; k < 9; k++) st->Rd[k] = Rd[k];
    for (int k = 0; k < 3; k++) st->td[k] = td[k];
    st->iters_run++;
    // max_iter reached: this iteration's transform must still run, everything after it must not
    if (st->iters_run >= st->max_iter) st->stop_after_transform = 1;
}

// the whole state in ONE coalesced read: thread w copies 32-bit word w into the workgroup's LDS copy (ends in a barrier)
constexpr int ST_WORDS = (int)(sizeof(IcpState) / 4);
static_assert(sizeof(IcpState) % 4 == 0 && ST_WORDS <= 64, "IcpState is staged by one wave, one word per lane");

__device__ __forceinline__ void icp_state_stage(const IcpState* st, IcpState* lds)
{
    if (threadIdx.x < ST_WORDS)
        __builtin_memcpy(reinterpret_cast<char*>(lds) + 4 * threadIdx.x, reinterpret_cast<const char*>(st) + 4 * threadIdx.x, 4);
    __syncthreads();
}

// single rank: reduce the block rows and advance the state in one launch
__global__ __launch_bounds__(KF_BLOCK) void icp_update_kernel(const double* __restrict__ partials, uint32_t n_blocks, int e, IcpState* st,
                                                              double* __restrict__ out)
{
    __shared__ double red[KF_GROUPS][64];
    __shared__ double row[KB_ROW];
    __shared__ double o18[19];
    __shared__ IcpState s_st;
    icp_state_stage(st, &s_st);
    if (s_st.stop) return;
    if (s_st.stop_after_transform) { if (threadIdx.x == 0) st->stop = 1; return; }
    reduce_rows(partials, n_blocks, red, row);
    row_to_out18(row, e, o18);
    __syncthreads();
    if (threadIdx.x < 19) out[threadIdx.x] = o18[threadIdx.x];
    if (threadIdx.x == 0) {
        icp_state_step(&s_st, o18, o18[16] >= 0.0, (float)o18[17], o18[18] != 0.0);
        *st = s_st;
    }
}

// The seed of the NEXT exhaustive search, computed where the point moves (round 3: one launch less per iteration than nn1_seed_kernel
// in front of every warm search, nn1_brute.hip): keys[] holds this iteration's correspondence of point i; that target, evaluated with
// the exact A1 arithmetic against the point's NEW position, is a genuine candidate of the next search and therefore an upper bound of
// its answer from the first instruction on.  Same values as nn1_seed_kernel writes (a correspondence that is not acceptable any more,
// or none at all, leaves "no claim").
struct SeedArgs {
    const float* tx; const float* ty; const float* tz;     // the target cloud (nullptr: no seeding)
    unsigned long long* keys;
    uint32_t nt;
};

__device__ __forceinline__ void seed_next_search(const SeedArgs& sd, uint32_t i, uint32_t n, float x, float y, float z)
{
    if (i >= n) return;
    const uint32_t j = (uint32_t)(sd.keys[i] & 0xFFFFFFFFull);
    unsigned long long key = ~0ull;
    if (j < sd.nt) {
        const float dx = x - sd.tx[j], dy = y - sd.ty[j], dz = z - sd.tz[j];
        const uint32_t e = __float_as_uint((dx * dx + dy * dy) + dz * dz);       // A1, unfused (nanoflann.hpp:403-406)
        if (e < 0x7F7FFFFFu) key = ((unsigned long long)e << 32) | j;              // FLT_MAX gate, nanoflann.hpp:163,1360
    }
    sd.keys[i] = key;
}

// Small clouds, single rank: the solve AND the move in one launch — one link less in the latency chain of an iteration
// (search -> sums -> solve -> move), which is all an iteration is at hw9's own size.  Every workgroup repeats the reduce and the 3 x 3
// solve on its own LDS copy of the state it read from st_in (deterministic arithmetic on the same rows: all of them arrive at the
// same new state), workgroup 0 publishes it to st_out — the OTHER buffer of a pair, so nobody reads what is being written; the
// next search tests the stop flags there — and each workgroup moves its own 4 096 points, whose loads were issued before the solve.
// The exits are those of icp_update_kernel + transform_state_kernel: a stopped loop stays as it is, stop_after_transform turns
// into stop without a move, a step that stops (converged, no pair, overflow) does not move, any other step moves.
__global__ __launch_bounds__(KF_BLOCK) void icp_update_move_kernel(const double* __restrict__ partials, uint32_t n_blocks, int e, const IcpState* __restrict__ st_in,
                                                                   IcpState* __restrict__ st_out, double* __restrict__ out, float* __restrict__ x,
                                                                   float* __restrict__ y, float* __restrict__ z, uint32_t n, uint32_t n4, SeedArgs sd)
{
    __shared__ double red[KF_GROUPS][64];
    __shared__ double row[KB_ROW];
    __shared__ double o18[19];
    __shared__ IcpState s_st;
    const uint32_t i = blockIdx.x * KF_BLOCK + threadIdx.x;
    const bool mine = i < n4;
    float4 px = make_float4(0.f, 0.f, 0.f, 0.f), py = px, pz = px;
    if (mine) { px = reinterpret_cast<float4*>(x)[i]; py = reinterpret_cast<float4*>(y)[i]; pz = reinterpret_cast<float4*>(z)[i]; }
    // the seeds of the next search (seed_next_search): the previous correspondences and their target points do not depend on the new pose — requested
    // here, in front of the reduce and the solve, so that the two dependent trips to memory (key -> target point) are over when the move is done
    unsigned long long sk[4] = { ~0ull, ~0ull, ~0ull, ~0ull };
    float sq[4][3] = { { 0.f, 0.f, 0.f }, { 0.f, 0.f, 0.f }, { 0.f, 0.f, 0.f }, { 0.f, 0.f, 0.f } };
    if (sd.tx && mine) {
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (4 * i + u < n) sk[u] = sd.keys[4 * i + u];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t j = (uint32_t)(sk[u] & 0xFFFFFFFFull);
            if (j < sd.nt) { sq[u][0] = sd.tx[j]; sq[u][1] = sd.ty[j]; sq[u][2] = sd.tz[j]; }
        }
    }
    icp_state_stage(st_in, &s_st);
    const int phase = s_st.stop ? 0 : (s_st.stop_after_transform ? 1 : 2);       // (uniform over the workgroup)
    if (phase == 2) {
        reduce_rows(partials, n_blocks, red, row);
        row_to_out18(row, e, o18);
        __syncthreads();
        if (blockIdx.x == 0 && threadIdx.x < 19) out[threadIdx.x] = o18[threadIdx.x];
        if (threadIdx.x == 0) icp_state_step(&s_st, o18, o18[16] >= 0.0, (float)o18[17], o18[18] != 0.0);
    } else if (phase == 1) {
        if (threadIdx.x == 0) s_st.stop = 1;
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) *st_out = s_st;
    if (s_st.stop || !mine) return;
    const float r0 = s_st.Rd[0], r1 = s_st.Rd[1], r2 = s_st.Rd[2], r3 = s_st.Rd[3], r4 = s_st.Rd[4], r5 = s_st.Rd[5],
                r6 = s_st.Rd[6], r7 = s_st.Rd[7], r8 = s_st.Rd[8], t0 = s_st.td[0], t1 = s_st.td[1], t2 = s_st.td[2];
    float4 ox, oy, oz;
#define PCR_ROW(o, a, b, c, tt)                      \
    o.x = ((a * px.x + b * py.x) + c * pz.x) + tt;   \
    o.y = ((a * px.y + b * py.y) + c * pz.y) + tt;   \
    o.z = ((a * px.z + b * py.z) + c * pz.z) + tt;   \
    o.w = ((a * px.w + b * py.w) + c * pz.w) + tt;
    PCR_ROW(ox, r0, r1, r2, t0)
    PCR_ROW(oy, r3, r4, r5, t1)
    PCR_ROW(oz, r6, r7, r8, t2)
#undef PCR_ROW
    const uint32_t base = 4 * i;                     // (the padding invariant of the tail group: transform_state_kernel)
    if (base + 3 >= n) {
        const float inf = __builtin_inff();
        if (base + 0 >= n) { ox.x = inf; oy.x = 0.f; oz.x = 0.f; }
        if (base + 1 >= n) { ox.y = inf; oy.y = 0.f; oz.y = 0.f; }
        if (base + 2 >= n) { ox.z = inf; oy.z = 0.f; oz.z = 0.f; }
        if (base + 3 >= n) { ox.w = inf; oy.w = 0.f; oz.w = 0.f; }
    }
    reinterpret_cast<float4*>(x)[i] = ox;
    reinterpret_cast<float4*>(y)[i] = oy;
    reinterpret_cast<float4*>(z)[i] = oz;
    if (sd.tx) {
        const float mx[4] = { ox.x, ox.y, ox.z, ox.w }, my[4] = { oy.x, oy.y, oy.z, oy.w }, mz[4] = { oz.x, oz.y, oz.z, oz.w };
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (base + u >= n) continue;
            const uint32_t j = (uint32_t)(sk[u] & 0xFFFFFFFFull);
            unsigned long long key = ~0ull;
            if (j < sd.nt) {
                const float dx = mx[u] - sq[u][0], dy = my[u] - sq[u][1], dz = mz[u] - sq[u][2];
                const uint32_t e2 = __float_as_uint((dx * dx + dy * dy) + dz * dz);       // A1, unfused (nanoflann.hpp:403-406): seed_next_search's arithmetic
                if (e2 < 0x7F7FFFFFu) key = ((unsigned long long)e2 << 32) | j;            // FLT_MAX gate, nanoflann.hpp:163,1360
            }
            sd.keys[base + u] = key;
        }
    }
}

// multi rank, step 1: reduce the block rows into the all-reduce buffer
//   [0..54] normalised limbs, [55] overflow flag, [56 + 2r] ORDER KEY of rank r's last kept pair (0: it kept none), [57 + 2r] that pair's d2
// (every entry is summed exactly by the all-reduce: integers below 2^40 x ranks, one non-zero pair of words per rank).  The order key
// says where the pair stands in the WHOLE source cloud: global index + 1 for a shard that knows its points' indices (gidx:
// pcr_cloud_shard_spatial — any partition), rank + 1 otherwise (contiguous blocks in rank order); the loss of an iteration is that of
// the pair with the largest key (registration.cpp:939: the last pair pushed).
__global__ __launch_bounds__(KF_BLOCK) void icp_reduce_slots_kernel(const double* __restrict__ partials, uint32_t n_blocks,
                                                                    double* __restrict__ out, int nranks, int rank, int have_points,
                                                                    const uint32_t* __restrict__ gidx)
{
    __shared__ double red[KF_GROUPS][64];
    __shared__ double row[KB_ROW];
    if (have_points) {
        reduce_rows(partials, n_blocks, red, row);
    } else {
        if (threadIdx.x < KB_ROW) row[threadIdx.x] = 0.0;
        __syncthreads();
    }
    if (threadIdx.x < KB_NL) out[threadIdx.x] = row[threadIdx.x];
    if (threadIdx.x == KB_NL) out[55] = row[56];
    if ((int)threadIdx.x < 2 * nranks) {
        const int r = threadIdx.x / 2, which = threadIdx.x % 2;
        double v = 0.0;
        if (r == rank && row[54] > 0.0) {
            const unsigned long long lk = (unsigned long long)__double_as_longlong(row[55]);
            v = which == 0 ? (gidx ? (double)gidx[(uint32_t)(lk >> 32)] + 1.0 : (double)(rank + 1)) : (double)__uint_as_float((uint32_t)(lk & 0xFFFFFFFFull));
        }
        out[56 + threadIdx.x] = v;
    }
}

// multi rank, step 2 (after the all-reduce): carries and moments (thread k converts moment k, as row_to_out18 does for one rank);
// the loss comes from the highest rank that kept a pair.  One wave.
__global__ __launch_bounds__(64) void icp_update_from_sums_kernel(double* __restrict__ buf, int nranks, int e, IcpState* st)
{
    __shared__ double row[KB_NL + 1];
    __shared__ double sums[16];
    __shared__ IcpState s_st;
    const int k = threadIdx.x;
    if (k < KB_NL) row[k] = buf[k];
    icp_state_stage(st, &s_st);
    if (s_st.stop) return;
    if (k < 6) { num::limbs_normalize(row + 3 * k, 2); sums[k] = num::limbs_value(row + 3 * k, 2, e - 2 * num::KB_W); }
    else if (k < 15) { num::limbs_normalize(row + 18 + 4 * (k - 6), 3); sums[k] = num::limbs_value(row + 18 + 4 * (k - 6), 3, 2 * e - 3 * num::KB_W); }
    else if (k == 15) sums[15] = row[54];
    __syncthreads();
    if (k != 0) return;
    const int ls = icp_last_slot(buf, nranks);                 // the pair with the largest order key
    const bool any = ls >= 0;
    const float d2 = any ? (float)buf[57 + 2 * ls] : 0.0f;
    icp_state_step(&s_st, sums, any, d2, buf[55] != 0.0);
    *st = s_st;
}

__global__ __launch_bounds__(256) void transform_state_kernel(float* __restrict__ x, float* __restrict__ y,
                                                              float* __restrict__ z, uint32_t n, uint32_t n4, IcpState* st, SeedArgs sd)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    // the points, the delta pose and the stop flag are requested together (one memory round trip, not three in a row)
    float4 px = reinterpret_cast<float4*>(x)[i];
    float4 py = reinterpret_cast<float4*>(y)[i];
    float4 pz = reinterpret_cast<float4*>(z)[i];
    const float r0 = st->Rd[0], r1 = st->Rd[1], r2 = st->Rd[2], r3 = st->Rd[3], r4 = st->Rd[4], r5 = st->Rd[5],
                r6 = st->Rd[6], r7 = st->Rd[7], r8 = st->Rd[8], t0 = st->td[0], t1 = st->td[1], t2 = st->td[2];
    const int stop = st->stop;
    if (stop) return;
    float4 ox, oy, oz;
#define PCR_ROW(o, a, b, c, tt)                      \
    o.x = ((a * px.x + b * py.x) + c * pz.x) + tt;   \
    o.y = ((a * px.y + b * py.y) + c * pz.y) + tt;   \
    o.z = ((a * px.z + b * py.z) + c * pz.z) + tt;   \
    o.w = ((a * px.w + b * py.w) + c * pz.w) + tt;
    PCR_ROW(ox, r0, r1, r2, t0)
    PCR_ROW(oy, r3, r4, r5, t1)
    PCR_ROW(oz, r6, r7, r8, t2)
#undef PCR_ROW
    // keep the padding invariant (x = +inf, y = z = 0) of the tail group without a second launch
    const uint32_t base = 4 * i;
    if (base + 3 >= n) {
        const float inf = __builtin_inff();
        if (base + 0 >= n) { ox.x = inf; oy.x = 0.f; oz.x = 0.f; }
        if (base + 1 >= n) { ox.y = inf; oy.y = 0.f; oz.y = 0.f; }
        if (base + 2 >= n) { ox.z = inf; oy.z = 0.f; oz.z = 0.f; }
        if (base + 3 >= n) { ox.w = inf; oy.w = 0.f; oz.w = 0.f; }
    }
    reinterpret_cast<float4*>(x)[i] = ox;
    reinterpret_cast<float4*>(y)[i] = oy;
    reinterpret_cast<float4*>(z)[i] = oz;
    if (sd.tx) {
        seed_next_search(sd, base + 0, n, ox.x, oy.x, oz.x); seed_next_search(sd, base + 1, n, ox.y, oy.y, oz.y);
        seed_next_search(sd, base + 2, n, ox.z, oy.z, oz.z); seed_next_search(sd, base + 3, n, ox.w, oy.w, oz.w);
    }
}

int launch_icp_update(pcr_ctx* ctx, uint32_t n_blocks, IcpState* st_dev, const KabschPlan& plan)
{
    {
        ProfScope p(ctx, "icp_update");
        hipLaunchKernelGGL(icp_update_kernel, dim3(1), dim3(KF_BLOCK), 0, ctx->stream, ctx->partials, n_blocks, plan.e, st_dev, ctx->dev_out);
    }
    PCR_HIP(ctx, hipGetLastError());
    return PCR_OK;
}

// seed_tgt != nullptr: the move also seeds the next exhaustive search against that target (seed_next_search) and says so on the context
static SeedArgs seed_args(pcr_ctx* ctx, const pcr_cloud* c, const pcr_cloud* seed_tgt)
{
    SeedArgs sd = { nullptr, nullptr, nullptr, nullptr, 0u };
    ctx->keys_seeded = false;
    if (seed_tgt && ctx->keys && ctx->keys_n == c->n && c->n) {
        sd = SeedArgs{ seed_tgt->x(), seed_tgt->y(), seed_tgt->z(), ctx->keys, (uint32_t)seed_tgt->n };
        ctx->keys_seeded = true; ctx->keys_seed_src = c; ctx->keys_seed_tgt = seed_tgt;
    }
    return sd;
}

int launch_icp_update_move(pcr_ctx* ctx, uint32_t n_blocks, const IcpState* st_in, IcpState* st_out, const KabschPlan& plan, pcr_cloud* c, const pcr_cloud* seed_tgt)
{
    const uint32_t n4 = (uint32_t)((c->n + 3) / 4);
    const SeedArgs sd = seed_args(ctx, c, seed_tgt);
    {
        ProfScope p(ctx, "icp_update");
        hipLaunchKernelGGL(icp_update_move_kernel, dim3((n4 + KF_BLOCK - 1) / KF_BLOCK), dim3(KF_BLOCK), 0, ctx->stream, ctx->partials, n_blocks, plan.e, st_in,
                           st_out, ctx->dev_out, c->x(), c->y(), c->z(), (uint32_t)c->n, n4, sd);
    }
    PCR_HIP(ctx, hipGetLastError());
    return PCR_OK;
}

int launch_icp_reduce_slots(pcr_ctx* ctx, uint32_t n_blocks, int nranks, int rank, bool have_points, const uint32_t* gidx)
{
    hipLaunchKernelGGL(icp_reduce_slots_kernel, dim3(1), dim3(KF_BLOCK), 0, ctx->stream, ctx->partials, n_blocks, ctx->dev_out, nranks, rank,
                       have_points ? 1 : 0, gidx);
    PCR_HIP(ctx, hipGetLastError());
    return PCR_OK;
}

int launch_icp_update_from_sums(pcr_ctx* ctx, int nranks, IcpState* st_dev, const KabschPlan& plan)
{
    hipLaunchKernelGGL(icp_update_from_sums_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->dev_out, nranks, plan.e, st_dev);
    PCR_HIP(ctx, hipGetLastError());
    return PCR_OK;
}

int launch_transform_state(pcr_ctx* ctx, pcr_cloud* c, IcpState* st_dev, const pcr_cloud* seed_tgt)
{
    if (c->n) {
        const uint32_t n4 = (uint32_t)((c->n + 3) / 4);
        const SeedArgs sd = seed_args(ctx, c, seed_tgt);
        ProfScope p(ctx, "transform");
        hipLaunchKernelGGL(transform_state_kernel, dim3((n4 + 255) / 256), dim3(256), 0, ctx->stream, c->x(), c->y(), c->z(),
                           (uint32_t)c->n, n4, st_dev, sd);
    }
    PCR_HIP(ctx, hipGetLastError());
    return PCR_OK;
}

// ---------------------------------------------------------------------------------------------- A8
struct Rt {
    float r[9];
    float t[3];
};

__global__ __launch_bounds__(256) void transform_kernel(float* __restrict__ x, float* __restrict__ y,
                                                        float* __restrict__ z, uint32_t n4, Rt m)
{
    // n4 = number of float4 groups covering [0, n) (the tail group lies inside the cloud's padding and is
    // re-padded by the caller's invariant: padding x stays +inf because inf*R + t is never read as a target
    // beyond n; see launch_transform)
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 px = reinterpret_cast<float4*>(x)[i];
    float4 py = reinterpret_cast<float4*>(y)[i];
    float4 pz = reinterpret_cast<float4*>(z)[i];
    float4 ox, oy, oz;
#define PCR_ROW(o, r0, r1, r2, tt)                                   \
    o.x = ((m.r[r0] * px.x + m.r[r1] * py.x) + m.r[r2] * pz.x) + m.t[tt]; \
    o.y = ((m.r[r0] * px.y + m.r[r1] * py.y) + m.r[r2] * pz.y) + m.t[tt]; \
    o.z = ((m.r[r0] * px.z + m.r[r1] * py.z) + m.r[r2] * pz.z) + m.t[tt]; \
    o.w = ((m.r[r0] * px.w + m.r[r1] * py.w) + m.r[r2] * pz.w) + m.t[tt];
    PCR_ROW(ox, 0, 1, 2, 0)
    PCR_ROW(oy, 3, 4, 5, 1)
    PCR_ROW(oz, 6, 7, 8, 2)
#undef PCR_ROW
    reinterpret_cast<float4*>(x)[i] = ox;
    reinterpret_cast<float4*>(y)[i] = oy;
    reinterpret_cast<float4*>(z)[i] = oz;
}

// restore the padding invariant (x = +inf, y = z = 0) of the partially transformed tail group
__global__ void repad_kernel(float* __restrict__ x, float* __restrict__ y, float* __restrict__ z, uint32_t n,
                             uint32_t n_end)
{
    const uint32_t i = n + threadIdx.x;
    if (i < n_end) { x[i] = __builtin_inff(); y[i] = 0.0f; z[i] = 0.0f; }
}

// out of place, the destination's padding written too: group i < n4 = the transformed points (the tail group's lanes beyond n get the padding),
// groups behind it = padding
__global__ __launch_bounds__(256) void transform_into_kernel(const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz,
                                                             float* __restrict__ x, float* __restrict__ y, float* __restrict__ z, uint32_t n, uint32_t n4,
                                                             uint32_t cap4, Rt m)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap4) return;
    const float inf = __builtin_inff();
    float4 ox = make_float4(inf, inf, inf, inf), oy = make_float4(0.f, 0.f, 0.f, 0.f), oz = oy;
    if (i < n4) {
        const float4 px = reinterpret_cast<const float4*>(sx)[i], py = reinterpret_cast<const float4*>(sy)[i], pz = reinterpret_cast<const float4*>(sz)[i];
#define PCR_ROW(o, r0, r1, r2, tt)                                   \
    o.x = ((m.r[r0] * px.x + m.r[r1] * py.x) + m.r[r2] * pz.x) + m.t[tt]; \
    o.y = ((m.r[r0] * px.y + m.r[r1] * py.y) + m.r[r2] * pz.y) + m.t[tt]; \
    o.z = ((m.r[r0] * px.z + m.r[r1] * py.z) + m.r[r2] * pz.z) + m.t[tt]; \
    o.w = ((m.r[r0] * px.w + m.r[r1] * py.w) + m.r[r2] * pz.w) + m.t[tt];
        PCR_ROW(ox, 0, 1, 2, 0)
        PCR_ROW(oy, 3, 4, 5, 1)
        PCR_ROW(oz, 6, 7, 8, 2)
#undef PCR_ROW
        const uint32_t base = 4 * i;
        if (base + 3 >= n) {
            if (base + 0 >= n) { ox.x = inf; oy.x = 0.f; oz.x = 0.f; }
            if (base + 1 >= n) { ox.y = inf; oy.y = 0.f; oz.y = 0.f; }
            if (base + 2 >= n) { ox.z = inf; oy.z = 0.f; oz.z = 0.f; }
            if (base + 3 >= n) { ox.w = inf; oy.w = 0.f; oz.w = 0.f; }
        }
    }
    reinterpret_cast<float4*>(x)[i] = ox;
    reinterpret_cast<float4*>(y)[i] = oy;
    reinterpret_cast<float4*>(z)[i] = oz;
}

int launch_transform_into(pcr_ctx* ctx, const pcr_cloud* src, pcr_cloud* dst, const float R[9], const float t[3])
{
    if (dst->cap != src->cap || dst->n != src->n || dst->cap % 4) return fail(ctx, PCR_ERR_ARG, "launch_transform_into");
    Rt m;
    memcpy(m.r, R, sizeof m.r);
    memcpy(m.t, t, sizeof m.t);
    const uint32_t n4 = (uint32_t)((src->n + 3) / 4), cap4 = (uint32_t)(src->cap / 4);
    {
        ProfScope p(ctx, "transform");
        hipLaunchKernelGGL(transform_into_kernel, dim3((cap4 + 255) / 256), dim3(256), 0, ctx->stream, src->x(), src->y(), src->z(), dst->x(), dst->y(), dst->z(),
                           (uint32_t)src->n, n4, cap4, m);
    }
    PCR_HIP(ctx, hipGetLastError());
    return PCR_OK;
}

int launch_transform(pcr_ctx* ctx, pcr_cloud* c, const float R[9], const float t[3])
{
    if (c->n == 0) return PCR_OK;
    Rt m;
    memcpy(m.r, R, sizeof m.r);
    memcpy(m.t, t, sizeof m.t);
    const uint32_t n4 = (uint32_t)((c->n + 3) / 4);
    {
        ProfScope p(ctx, "transform");
        hipLaunchKernelGGL(transform_kernel, dim3((n4 + 255) / 256), dim3(256), 0, ctx->stream, c->x(), c->y(),
                           c->z(), n4, m);
    }
    PCR_HIP(ctx, hipGetLastError());
    if (c->n % 4) {
        hipLaunchKernelGGL(repad_kernel, dim3(1), dim3(4), 0, ctx->stream, c->x(), c->y(), c->z(), (uint32_t)c->n,
                           (uint32_t)(n4 * 4));
        PCR_HIP(ctx, hipGetLastError());
    }
    return PCR_OK;
}

}  // namespace pcr

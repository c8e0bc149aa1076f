// nn1_sphere.hpp — part of nn1_brute.hip (included there, inside namespace pcr, behind nn1_strack_kernel).
//
// STRACK3: the exhaustive search with the sign filter applied to a HIERARCHY OF BOUNDING SPHERES (round 4).  STRACK spends one matrix instruction and
// eight half-rate v_or3_b32 per 32 queries x 32 RECORDS, for every record of the target — and at the settled pose of an ICP loop all but a handful of
// them lie metres beyond the query's threshold.  The sign test is bilinear, so it works on sets of records as well: a set inside the sphere (c, rho)
// cannot matter to a query unless |r - c| <= sqrt(thr) + rho, which is again "a sum of K-slot products is negative" (grid_common.hpp, LEVEL 1: the
// statement and what makes it a theorem).  Three levels of MFMA rows (BtIndex, built by bt_ensure_l1):
//   level 0: one row per level-1 TILE of 512 records, in the scale of a level-0 super-tile of 131 072 records — EVERY row of the slice gets its sign for
//            every query (still exhaustive: nothing is skipped without a computed sign that says it cannot matter);
//   level 1: one row per CHUNK of 16 records of the level-1 tiles level 0 flagged, level-1 super-tiles of 4 096 records share a centre and a scale;
//   level 2: the tiles of 32 records that hold a flagged chunk through STRACK's per-record rows, with operands in the scale of their LEVEL-1 super-tile
//            (BtIndex::l1_rec_ops: no operand setup of its own); the flagged (query, chunk) pairs of THAT are evaluated with the exact A1 arithmetic,
//            four lanes per chunk — the canonical (d2 bits, index) minimum decides, thresholds fall after every level-1 super-tile.
// History: the two-level form without level 0 (STRACK2, every chunk row for every query) ran 0.085 ms per 120 k x 120 k search against
// STRACK's 0.44 and was removed when this kernel (0.034 ms) took its place.
// Same keys bit for bit as every other kernel of this file (parity sweeps: nn1_variant 10; device check of the sphere statement:
// pcr_selftest_sphere_f16, part of the once-per-context verdict).  Matches: registration.cpp:925-941.
#pragma once

constexpr int S2_TILES = 512;                 // level-2 tiles a wave may collect per level-1 super-tile (of its 128: the list is flushed when full)
constexpr int S2_CAP = 128;                   // flagged (query, chunk) pairs a wave lists before it evaluates them

// the listed chunks against their queries: four lanes per chunk, four records each, sixteen chunks per round
template <int QG, class LDS>
__device__ __forceinline__ void s2_flush(LDS& L, uint32_t cnt, const float4* __restrict__ records, uint32_t lane)
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (uint32_t e0 = 0; e0 < cnt; e0 += 32) {               // two rounds of sixteen chunks per trip to memory
        uint32_t slot[2];
        bool valid[2];
        float4 q[2], rec[2][4];
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const uint32_t e = e0 + 16u * r + (lane >> 2);
            valid[r] = e < cnt;
            const uint32_t ent = L.list[valid[r] ? e : 0];
            slot[r] = ent & 127u;
            q[r] = L.q[slot[r]];
            const float4* rp = records + (size_t)(ent >> 7) * 16 + (lane & 3u);
#pragma unroll
            for (int j = 0; j < 4; j++) rec[r][j] = rp[4 * j];             // (padding records: x = +inf, never accepted)
        }
#pragma unroll
        for (int r = 0; r < 2; r++) {
            unsigned long long key = ~0ull;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t d = d2_exact_bits(q[r].x, q[r].y, q[r].z, rec[r][j].x, rec[r][j].y, rec[r][j].z);
                const unsigned long long k = ((unsigned long long)d << 32) | __float_as_uint(rec[r][j].w);
                if (d < 0x7F7FFFFFu && k < key) key = k;                   // FLT_MAX gate
            }
            if (!valid[r]) key = ~0ull;
#define PCR_S2_MIN(CTRL) { const unsigned long long w = ((unsigned long long)(uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(key >> 32), CTRL, 0xF, 0xF, false) << 32) | \
                                                      (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)key, CTRL, 0xF, 0xF, false);                              \
                           key = w < key ? w : key; }
            PCR_S2_MIN(0xB1) PCR_S2_MIN(0x4E)                              // quad xor 1, xor 2: the four lanes of the chunk
#undef PCR_S2_MIN
            if ((lane & 3u) == 0u && key != ~0ull) atomicMin(&L.best[slot[r]], key);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

__device__ __forceinline__ uint32_t s2_or16(const f32x16 acc)
{
    uint32_t a = __float_as_uint(acc[0]) | __float_as_uint(acc[1]) | __float_as_uint(acc[2]);
#pragma unroll
    for (int j = 3; j + 1 < 16; j += 2) a = a | __float_as_uint(acc[j]) | __float_as_uint(acc[j + 1]);
    return a | __float_as_uint(acc[15]);
}

#ifdef PCR_S2_PROF
#define PCR_S2_TICK(acc) { pt_b = __builtin_amdgcn_s_memrealtime(); acc += pt_b - pt_a; pt_a = pt_b; }
#else
#define PCR_S2_TICK(acc)
#endif
#ifndef PCR_S2_WAVES
#define PCR_S2_WAVES 4
#endif
// ---- the kernel.  (Measured on the two-level predecessor, sorted queries, 120 000 x 120 000, profile build: 86 % of a wave's life went into level 1 —
// every chunk row of the target for every group of queries, 300 SIMD cycles per level-1 tile — and 10 % into level 2 and the exact evaluation.  With
// level 0 a wave looks at 8 level-0 tiles, then only at the level-1 tiles whose sphere some query's ball reaches: a few of 240.)
constexpr int S3_L1LIST = 256;                // level-1 tiles of one level-0 super-tile

template <int QG>
struct S3WaveLds {
    float4 q[QG * 32];
    unsigned long long best[QG * 32];
    uint32_t l1list[S3_L1LIST];               // level-1 tile | groups that flagged it << 28
    uint32_t tiles[S2_TILES];                 // level-2 tile | groups << 28
    uint32_t list[S2_CAP + 256 * QG];         // (chunk << 7) | query slot: what a batch of four level-2 tiles can add fits behind S2_CAP entries
};

template <int QG>
__global__ __launch_bounds__(NN_BLOCK, PCR_S2_WAVES) void nn1_strack3_kernel(
    const float4* __restrict__ l0_centres, const uint4* __restrict__ l0_ops, const float4* __restrict__ l1_centres, const uint4* __restrict__ l1_ops,
    const uint4* __restrict__ ops, const float4* __restrict__ records, uint32_t n_rec, uint32_t n_l0, uint32_t l0_per_slice,
    const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz, uint32_t ns,
    unsigned long long* __restrict__ keys, const int* __restrict__ stop, unsigned long long* __restrict__ stats, uint32_t flush_at, uint32_t flush_end)
{
    static_assert(QG == 4 || QG == 2 || QG == 1, "query groups per wave: pairs (a lane owns query n of the groups 2 p + h) or ONE (both half-lanes own query n)");
    constexpr int NP = QG == 1 ? 1 : QG / 2;
    const int stopv = stop ? (stop[0] | stop[1]) : 0;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n = lane & 31;
    const bool h = lane >= 32;
    const uint32_t qb = blockIdx.x, sl = blockIdx.y;
    __shared__ S3WaveLds<QG> lds_all[NN_BLOCK / 64];
    S3WaveLds<QG>& L = lds_all[wave];
    const uint32_t qbase = (qb * (NN_BLOCK / 64) + wave) * (32 * QG);
    if (qbase >= ns) return;                                  // (a surplus wave: no workgroup barrier below)
    float qx[NP], qy[NP], qz[NP], thr[NP];
    bool ok[NP];
    bool okg[QG];
    auto slot_of = [&](int p) { return QG == 1 ? n : (uint32_t)(2 * p + (h ? 1 : 0)) * 32 + n; };
#pragma unroll
    for (int p = 0; p < NP; p++) {
        const uint32_t slot = slot_of(p), i = min(qbase + slot, ns - 1);
        qx[p] = sx[i]; qy[p] = sy[i]; qz[p] = sz[i];
        const uint32_t cb = (uint32_t)(__atomic_load_n(&keys[i], __ATOMIC_RELAXED) >> 32);      // the candidate's d2 (or what other slices published)
        ok[p] = fabsf(qx[p]) < 1e18f && fabsf(qy[p]) < 1e18f && fabsf(qz[p]) < 1e18f && cb < 0x7F7FFFFFu;
        thr[p] = ok[p] ? __uint_as_float(cb) : -INFINITY;
        L.q[slot] = make_float4(qx[p], qy[p], qz[p], 0.0f);
        L.best[slot] = ~0ull;
        if (!ok[p]) { qx[p] = 0.0f; qy[p] = 0.0f; qz[p] = 0.0f; }                                // (finite operands; thr = -inf: no flag, ever)
        const unsigned long long okm = __builtin_amdgcn_ballot_w64(ok[p]);
        okg[QG == 1 ? 0 : 2 * p] = (uint32_t)okm == 0xFFFFFFFFu;
        if (QG > 1) okg[QG == 1 ? 0 : 2 * p + 1] = (uint32_t)(okm >> 32) == 0xFFFFFFFFu;
    }
    if (stopv) return;
#ifdef PCR_S2_PROF
    unsigned long long* const stats_p = stats;
    stats = nullptr;
    unsigned long long pt_pro = 0, pt_l0 = 0, pt_l1 = 0, pt_l2 = 0, pt_epi = 0, pt_a = __builtin_amdgcn_s_memrealtime(), pt_b = 0;
    const unsigned long long pt_begin = pt_a;
#endif
    unsigned long long st_l0 = 0, st_l1 = 0, st_l1flag = 0, st_l2flag = 0, st_l2 = 0, st_eval = 0, st_flushes = 0;       // diagnostics (stats != nullptr)
    unsigned long long clk0 = 0, rt0 = 0;
    if (stats) { clk0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
    const uint32_t sb = sl * l0_per_slice, se = min(sb + l0_per_slice, n_l0);
    f32x16 zero;
#pragma unroll
    for (int j = 0; j < 16; j++) zero[j] = 0.0f;
    uint32_t cnt = 0;                                         // entries in the wave's list of flagged (query, chunk) pairs (wave-uniform)
    auto refresh = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int p = 0; p < NP; p++) {
            const uint32_t fb = (uint32_t)(L.best[slot_of(p)] >> 32);       // (~0 >> 32 is a NaN pattern: fminf keeps thr)
            thr[p] = ok[p] ? fminf(thr[p], __uint_as_float(fb)) : thr[p];
        }
    };
    // B operands: bq0 = sphere rows in a level-0 super-tile's scale; bq1 / bq2 = sphere rows / record rows in a level-1 super-tile's scale
    // (lane (n, h) builds the whole operand of query n of group 2 p + h, the halves change places by v_permlane32_swap: nn1_strack_kernel)
    uint4 bq0[QG], bq1[QG], bq2[QG];
    auto setup0 = [&](const float4 C) {
        const float sc2 = C.w * C.w;
#pragma unroll
        for (int p = 0; p < NP; p++) {
            uint32_t P[4], Q[4];
            st_setup_l1(qx[p], qy[p], qz[p], C, thr[p], sc2, P, Q);
            if (QG == 1) { bq0[0] = h ? make_uint4(Q[0], Q[1], Q[2], Q[3]) : make_uint4(P[0], P[1], P[2], P[3]); continue; }     // (both half-lanes built query n's operand)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const auto r = __builtin_amdgcn_permlane32_swap(P[j], Q[j], false, false);
                P[j] = r[0]; Q[j] = r[1];
            }
            bq0[(2 * p) % QG] = make_uint4(P[0], P[1], P[2], P[3]);
            bq0[(2 * p + 1) % QG] = make_uint4(Q[0], Q[1], Q[2], Q[3]);
        }
    };
    auto setup1 = [&](const float4 C) {
        const float sc2 = C.w * C.w;
#pragma unroll
        for (int p = 0; p < NP; p++) {
            uint32_t P[4], Q[4], Q2[4], P2[4];
            st_setup_l1(qx[p], qy[p], qz[p], C, thr[p], sc2, P, Q, Q2);
            if (QG == 1) {
                bq1[0] = h ? make_uint4(Q[0], Q[1], Q[2], Q[3]) : make_uint4(P[0], P[1], P[2], P[3]);
                bq2[0] = h ? make_uint4(Q2[0], Q2[1], Q2[2], Q2[3]) : make_uint4(P[0], P[1], P[2], P[3]);
                continue;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                P2[j] = P[j];
                const auto r = __builtin_amdgcn_permlane32_swap(P[j], Q[j], false, false);
                P[j] = r[0]; Q[j] = r[1];
                const auto r2 = __builtin_amdgcn_permlane32_swap(P2[j], Q2[j], false, false);
                P2[j] = r2[0]; Q2[j] = r2[1];
            }
            bq1[(2 * p) % QG] = make_uint4(P[0], P[1], P[2], P[3]);
            bq1[(2 * p + 1) % QG] = make_uint4(Q[0], Q[1], Q[2], Q[3]);
            bq2[(2 * p) % QG] = make_uint4(P2[0], P2[1], P2[2], P2[3]);
            bq2[(2 * p + 1) % QG] = make_uint4(Q2[0], Q2[1], Q2[2], Q2[3]);
        }
    };
    // LEVEL 2 over the collected tiles of 32 records: the per-record filter for the groups that flagged the tile, four tiles per trip to memory;
    // flagged (query, chunk) pairs are listed and evaluated exactly between batches (from flush_at entries) and at the end (from flush_end);
    // thresholds fall after an evaluation
    auto level2 = [&](uint32_t n_tiles) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // (lane 0 wrote the list)
        for (uint32_t k0 = 0; k0 < n_tiles; k0 += 4) {
            uint32_t Eb[4];
            uint4 Ab[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                Eb[u] = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.tiles[min(k0 + u, n_tiles - 1)]);
                Ab[u] = ops[(size_t)(Eb[u] & 0x0FFFFFFFu) * 64 + lane];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (k0 + u >= n_tiles) break;                 // (wave-uniform)
                const uint32_t T = Eb[u] & 0x0FFFFFFFu, gm = Eb[u] >> 28;
                const uint32_t chunk = 2u * T + (h ? 1u : 0u);
#pragma unroll
                for (int g = 0; g < QG; g++) {
                    if (!((gm >> g) & 1u)) continue;          // (wave-uniform)
                    const uint32_t og = s2_or16(__builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, Ab[u]), __builtin_bit_cast(f16x8, bq2[g]), zero, 0, 0, 0));
                    if (stats) st_l2++;
                    const unsigned long long m = __builtin_amdgcn_ballot_w64((int)og < 0);
                    if (!m) continue;
                    if ((int)og < 0) L.list[cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (chunk << 7) | (uint32_t)(g * 32) | n;
                    cnt += (uint32_t)__popcll(m);
                }
            }
            if (cnt >= flush_at) { s2_flush<QG>(L, cnt, records, lane); st_flushes++; st_eval += cnt; cnt = 0; refresh(); }      // (flush_at <= S2_CAP: host)
        }
        if (cnt >= flush_end) { s2_flush<QG>(L, cnt, records, lane); st_flushes++; st_eval += cnt; cnt = 0; refresh(); }
    };
    const uint32_t n_l1_tiles = (n_rec + 511u) / 512u;        // level-1 tiles that hold records (rows behind them: W = +inf, never flagged)
    for (uint32_t S0 = sb; S0 < se; S0++) {
        // LEVEL 0: the eight level-0 tiles of the super-tile, every group — the level-1 tiles some query's ball reaches go on the list
        setup0(l0_centres[S0]);
        PCR_S2_TICK(pt_pro)
        uint32_t n1 = 0;
        uint4 A0s[8];                                         // the eight level-0 tiles at once: one round trip to memory instead of eight
#pragma unroll
        for (int t = 0; t < 8; t++) A0s[t] = l0_ops[((size_t)S0 * 8 + t) * 64 + lane];
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const uint32_t T0 = S0 * 8 + t;
            const uint4 A0 = A0s[t];
            uint32_t anyg[QG];
#pragma unroll
            for (int g = 0; g < QG; g++)
                anyg[g] = s2_or16(__builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A0), __builtin_bit_cast(f16x8, bq0[g]), zero, 0, 0, 0));
            if (stats) st_l0 += QG;
            uint32_t anyall = anyg[0];
#pragma unroll
            for (int g = 1; g < QG; g++) anyall |= anyg[g];
            if (__builtin_amdgcn_ballot_w64((int)anyall < 0)) {
                uint32_t rmask[QG];                           // bit j: row j of this level-0 tile (level-1 tile T0 * 32 + j) flagged by the group
#pragma unroll
                for (int g = 0; g < QG; g++) {
                    rmask[g] = 0u;
                    if (!__builtin_amdgcn_ballot_w64((int)anyg[g] < 0)) continue;
                    const f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A0), __builtin_bit_cast(f16x8, bq0[g]), zero, 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const unsigned long long m = __builtin_amdgcn_ballot_w64((int)__float_as_uint(acc[i]) < 0);
                        rmask[g] |= ((uint32_t)m != 0u ? 1u : 0u) << i;                        // lanes < 32: rows 0..15
                        rmask[g] |= ((uint32_t)(m >> 32) != 0u ? 1u : 0u) << (16 + i);         // lanes >= 32: rows 16..31
                    }
                }
                uint32_t un = rmask[0];
#pragma unroll
                for (int g = 1; g < QG; g++) un |= rmask[g];
                while (un) {                                  // wave-uniform; ascending level-1 tiles
                    const uint32_t j = (uint32_t)__builtin_ctz(un);
                    un &= un - 1u;
                    uint32_t gm = 0u;
#pragma unroll
                    for (int g = 0; g < QG; g++) gm |= ((rmask[g] >> j) & 1u) << g;
                    const uint32_t T1 = T0 * 32u + j;
                    if (T1 < n_l1_tiles) { if (lane == 0) L.l1list[n1] = T1 | (gm << 28); n1++; }      // (n1 <= 256 = the rows of a level-0 super-tile)
                }
            }
        }
        if (stats) st_l1flag += n1;
        PCR_S2_TICK(pt_l0)
        if (!n1) continue;
        // (Measured and dropped: the level-0 tiles of a super-tile dealt to 2 / 4 waves per group of queries — twice / four times the waves, each with a
        // share of the level-0 rows and what hangs under them: 0.036 / 0.034 ms, the same.)
        // (Measured and dropped: the k-th eighth of the sorted query blocks on XCD k, so that an XCD's L2 holds one region of the target's rows instead
        // of all of them — PMC: 31 MB fetched per launch for a 4 MB working set, every XCD its own copy: 0.036-0.037 ms against 0.034; the regions
        // differ in work, and the launch is not bound by those fetches.)
        // (Measured and dropped: the whole of a SHORT list — up to eight level-1 tiles in up to four super-tiles, the settled pose's case — in one go:
        // all operands in one trip, four operand sets side by side, all level-2 tiles, one evaluation.  Three links in the wave's chain of trips to
        // memory instead of three per super-tile, and the same 0.034-0.035 ms: PMC counts 1 950 vector + 1 800 scalar instructions per wave of 32
        // queries, four waves per SIMD — the launch is bound by their issue, not by the chain.)
        // LEVEL 1 over the listed level-1 tiles, for the groups that reached them: chunk rows.  The list is ascending, so the (at most eight) tiles
        // of one level-1 super-tile follow each other: their operands come in ONE trip to memory, the super-tile's scale serves level 1 and
        // level 2, and the level-2 tiles with a flagged chunk are filtered and evaluated before the next super-tile's operands are built
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // (lane 0 wrote the list)
        for (uint32_t k = 0; k < n1;) {
            uint32_t Er[8];
            uint4 Ar[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                Er[u] = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.l1list[min(k + u, n1 - 1)]);
                Ar[u] = l1_ops[(size_t)(Er[u] & 0x0FFFFFFFu) * 64 + lane];
            }
            const uint32_t S1 = (Er[0] & 0x0FFFFFFFu) >> 3;
            setup1(l1_centres[S1]);
            uint32_t n_tiles = 0, run = 0;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (k + u >= n1 || ((Er[u] & 0x0FFFFFFFu) >> 3) != S1) break;       // (wave-uniform: the end of the list or of this super-tile's run)
                run++;
                const uint32_t T1 = Er[u] & 0x0FFFFFFFu, gm1 = Er[u] >> 28;
                uint32_t tmask[QG];                           // bit k: level-2 tile k of this level-1 tile (chunks 2 k, 2 k + 1) flagged by the group
                uint32_t un = 0u;
#pragma unroll
                for (int g = 0; g < QG; g++) {
                    tmask[g] = 0u;
                    if (!((gm1 >> g) & 1u)) continue;         // (wave-uniform)
                    const f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, Ar[u]), __builtin_bit_cast(f16x8, bq1[g]), zero, 0, 0, 0);
                    if (stats) st_l1++;
                    if (!__builtin_amdgcn_ballot_w64((int)s2_or16(acc) < 0)) continue;
#pragma unroll
                    for (int i = 0; i < 16; i += 2) {         // (lanes < 32 hold chunks 0..15 of the tile, lanes >= 32 chunks 16..31: accumulator i <-> chunk 16 h + i)
                        const unsigned long long m = __builtin_amdgcn_ballot_w64((int)(__float_as_uint(acc[i]) | __float_as_uint(acc[i + 1])) < 0);
                        tmask[g] |= ((uint32_t)m != 0u ? 1u : 0u) << (i / 2);
                        tmask[g] |= ((uint32_t)(m >> 32) != 0u ? 1u : 0u) << (8 + i / 2);
                    }
                    un |= tmask[g];
                }
                if (stats) st_l2flag += (unsigned long long)__popc(un);
                while (un) {                                  // wave-uniform (at most 8 x 16 = 128 tiles per super-tile: the list of 512 holds them)
                    const uint32_t k2 = (uint32_t)__builtin_ctz(un);
                    un &= un - 1u;
                    uint32_t gm = 0u;
#pragma unroll
                    for (int g = 0; g < QG; g++) gm |= ((tmask[g] >> k2) & 1u) << g;
                    const uint32_t T2 = T1 * 16u + k2;
                    if ((size_t)T2 * 32 < n_rec) { if (lane == 0) L.tiles[n_tiles] = T2 | (gm << 28); n_tiles++; }      // (tiles of the index's padding hold nothing)
                }
            }
            k += run;
            PCR_S2_TICK(pt_l1)
            if (n_tiles) level2(n_tiles);
            PCR_S2_TICK(pt_l2)
        }
        if (cnt) { s2_flush<QG>(L, cnt, records, lane); st_flushes++; st_eval += cnt; cnt = 0; refresh(); }
        PCR_S2_TICK(pt_l2)
    }
#pragma unroll
    for (int g = 0; g < QG; g++) {
        unsigned long long kbest = L.best[g * 32 + n];
        if (!okg[g] && sb < se) {
            // a query without finite coordinates or without a candidate: the wave scans the slice exactly for this group, each half-lane one half
            // of it (rare: NaN / inf queries, a seed kernel that found nothing acceptable)
            const float4 q = L.q[g * 32 + n];
            const uint32_t r0 = (uint32_t)min((unsigned long long)sb * BT_L0_SUPER, (unsigned long long)n_rec),
                           r1 = (uint32_t)min((unsigned long long)se * BT_L0_SUPER, (unsigned long long)n_rec), mid = r0 + (r1 - r0) / 2;
            for (uint32_t j = h ? mid : r0; j < (h ? r1 : mid); j++) {
                const float4 rec = records[j];
                const uint32_t e = d2_exact_bits(q.x, q.y, q.z, rec.x, rec.y, rec.z);
                const unsigned long long key = ((unsigned long long)e << 32) | __float_as_uint(rec.w);
                if (e < 0x7F7FFFFFu && key < kbest) kbest = key;
            }
            const unsigned long long ko = ((unsigned long long)(uint32_t)__shfl_xor((int)(kbest >> 32), 32, 64) << 32) |
                                          (uint32_t)__shfl_xor((int)(uint32_t)kbest, 32, 64);
            kbest = ko < kbest ? ko : kbest;
            if (kbest == ~0ull) kbest = 0x7F800000FFFFFFFFull;           // "no neighbour" is the key (+inf, no index), as every other kernel writes it
        }
        const uint32_t i = qbase + g * 32 + n;
        if (!h && i < ns && kbest != ~0ull) merge_key(&keys[i], kbest);
    }
#ifdef PCR_S2_PROF
    PCR_S2_TICK(pt_epi)
    if (stats_p && lane == 0 && ((blockIdx.x * 7u + blockIdx.y * 3u + wave) & 15u) == 0u) {     // one wave in 16 reports: [0] prologue + level-0 setup [1] level 0 [2] level 1 [3] level 2 + evaluation
        const unsigned long long life = pt_a - pt_begin;                                          // [4] epilogue ticks (10 ns), [5] waves, [6] sum of lives, [7] longest, [8..15] lives by 5 us bins
        atomicAdd(&stats_p[0], pt_pro); atomicAdd(&stats_p[1], pt_l0); atomicAdd(&stats_p[2], pt_l1); atomicAdd(&stats_p[3], pt_l2); atomicAdd(&stats_p[4], pt_epi);
        atomicAdd(&stats_p[5], 1ull); atomicAdd(&stats_p[6], life); atomicMax(&stats_p[7], life);
        atomicAdd(&stats_p[8 + min((int)(life / 500), 7)], 1ull);
    }
#endif
    if (stats && threadIdx.x == 0) {                                      // shader clock under this kernel's load: cycles / 100 MHz ticks (bench.py)
        atomicAdd(&stats[4], (unsigned long long)__builtin_amdgcn_s_memtime() - clk0);
        atomicAdd(&stats[5], (unsigned long long)__builtin_amdgcn_s_memrealtime() - rt0);
    }
    if (stats && lane == 0) {
        if (st_flushes) atomicAdd(&stats[2], st_flushes);                 // joint evaluations (wave level)
        if (st_l1flag) atomicAdd(&stats[3], st_l1flag);                   // level-1 tiles flagged by level 0 (per wave: union of its groups)
        if (st_eval) atomicAdd(&stats[6], st_eval);                       // (query, chunk) pairs evaluated exactly
        if (st_l0) atomicAdd(&stats[7], st_l0);                           // level-0 MFMAs
        if (st_l1) atomicAdd(&stats[8], st_l1);                           // level-1 MFMAs
        if (st_l2flag) atomicAdd(&stats[9], st_l2flag);                   // level-2 tiles flagged by level 1
        if (st_l2) atomicAdd(&stats[10], st_l2);                          // level-2 MFMAs
    }
}

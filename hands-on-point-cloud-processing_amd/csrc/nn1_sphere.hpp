// nn1_sphere.hpp — part of nn1_brute.hip (included there, inside namespace pcr, behind nn1_strack_kernel).
//
// STRACK2: the exhaustive search with the sign filter applied at TWO LEVELS (round 4).  STRACK spends one matrix instruction and eight
// half-rate v_or3_b32 per 32 queries x 32 RECORDS, for every record of the target — and at the settled pose of an ICP loop all but a
// handful of them lie metres beyond the query's threshold.  The sign test is bilinear, so it works one level up as well: a chunk of 16
// records with bounding sphere (c, rho) cannot matter to a query unless |r - c| <= sqrt(thr) + rho, which is again "a sum of K-slot
// products is negative" (grid_common.hpp, LEVEL 1: the statement and what makes it a theorem).  So
//   level 1: ONE MFMA row per CHUNK — one instruction per 32 queries x 32 chunks = 512 records: a sixteenth of STRACK's matrix and
//            vector work — over EVERY chunk of the slice (still exhaustive: no chunk is skipped without its sign having been computed);
//            level-1 super-tiles of 4 096 records share a centre and a scale (one operand setup per query and 8 level-1 tiles);
//   level 2: the tiles of 32 records that hold a flagged chunk go through STRACK's own per-record filter (the target's precomputed f16
//            operands, one MFMA per tile and flagged group) and the flagged (query, chunk) pairs of THAT are evaluated with the exact A1
//            arithmetic, four lanes per chunk — the canonical (d2 bits, index) minimum decides, thresholds fall after every level-1
//            super-tile.
// Same keys bit for bit as every other kernel of this file (parity sweeps: nn1_variant 9; device check of the level-1 statement:
// pcr_selftest_sphere_f16, part of the once-per-context verdict).  Matches: registration.cpp:925-941.
#pragma once

constexpr int S2_TILES = 512;                 // level-2 tiles a wave may collect per level-1 super-tile (of its 128: the list is flushed when full)
constexpr int S2_CAP = 128;                   // flagged (query, chunk) pairs a wave lists before it evaluates them

template <int QG>
struct S2WaveLds {
    float4 q[QG * 32];                        // the wave's queries
    unsigned long long best[QG * 32];         // (d2 bits << 32 | index) found so far, ~0 = nothing
    uint32_t tiles[S2_TILES];                 // level-2 tile | groups that flagged it << 28
    uint32_t list[S2_CAP];                    // (chunk << 7) | query slot
};

// the listed chunks against their queries: four lanes per chunk, four records each, sixteen chunks per round
template <int QG>
__device__ __forceinline__ void s2_flush(S2WaveLds<QG>& L, uint32_t cnt, const float4* __restrict__ records, uint32_t lane)
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (uint32_t e0 = 0; e0 < cnt; e0 += 16) {
        const uint32_t e = e0 + (lane >> 2);
        const bool valid = e < cnt;
        const uint32_t ent = L.list[valid ? e : 0];
        const uint32_t slot = ent & 127u;
        const float4 q = L.q[slot];
        const float4* rp = records + (size_t)(ent >> 7) * 16 + (lane & 3u);
        float4 rec[4];
#pragma unroll
        for (int j = 0; j < 4; j++) rec[j] = rp[4 * j];                    // (padding records: x = +inf, never accepted)
        unsigned long long key = ~0ull;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t d = d2_exact_bits(q.x, q.y, q.z, rec[j].x, rec[j].y, rec[j].z);
            const unsigned long long k = ((unsigned long long)d << 32) | __float_as_uint(rec[j].w);
            if (d < 0x7F7FFFFFu && k < key) key = k;                       // FLT_MAX gate
        }
        if (!valid) key = ~0ull;
#define PCR_S2_MIN(CTRL) { const unsigned long long w = ((unsigned long long)(uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(key >> 32), CTRL, 0xF, 0xF, false) << 32) | \
                                                      (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)key, CTRL, 0xF, 0xF, false);                              \
                           key = w < key ? w : key; }
        PCR_S2_MIN(0xB1) PCR_S2_MIN(0x4E)                                  // quad xor 1, xor 2: the four lanes of the chunk
#undef PCR_S2_MIN
        if ((lane & 3u) == 0u && key != ~0ull) atomicMin(&L.best[slot], key);
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

#ifndef PCR_S2_WAVES
#define PCR_S2_WAVES 4
#endif
template <int QG>
__global__ __launch_bounds__(NN_BLOCK, PCR_S2_WAVES) void nn1_strack2_kernel(
    const float4* __restrict__ l1_centres, const uint4* __restrict__ l1_ops, const float4* __restrict__ centres, const uint4* __restrict__ ops,
    const float4* __restrict__ records, uint32_t n_rec, uint32_t n_l1, uint32_t l1_per_slice,
    const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz, uint32_t ns,
    unsigned long long* __restrict__ keys, const int* __restrict__ stop, unsigned long long* __restrict__ stats,
    uint32_t xq, uint32_t qblocks, uint32_t slices, uint32_t flush_at)
{
    static_assert(QG == 4, "four query groups per wave: a lane owns query n of the groups 2 p + h");
    const int stopv = stop ? (stop[0] | stop[1]) : 0;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n = lane & 31;
    const bool h = lane >= 32;
    uint32_t qb = blockIdx.x, sl = blockIdx.y;                // (query block, slice): as nn1_strack_kernel
    if (xq) {
        const uint32_t k = blockIdx.x & 7u, j = blockIdx.x >> 3, xs = 8u / xq, qb_per = (qblocks + xq - 1) / xq;
        qb = (j % qb_per) * xq + k % xq;
        sl = (j / qb_per) * xs + k / xq;
        if (qb >= qblocks || sl >= slices) return;
    }
    __shared__ S2WaveLds<QG> lds_all[NN_BLOCK / 64];
    S2WaveLds<QG>& L = lds_all[wave];
    const uint32_t qbase = (qb * (NN_BLOCK / 64) + wave) * (32 * QG);
    if (qbase >= ns) return;                                  // (a surplus wave: no workgroup barrier below)
    float qx[QG / 2], qy[QG / 2], qz[QG / 2], thr[QG / 2];
    bool ok[QG / 2];
    bool okg[QG];
#pragma unroll
    for (int p = 0; p < QG / 2; p++) {
        const uint32_t slot = (2 * p + (h ? 1 : 0)) * 32 + n, i = min(qbase + slot, ns - 1);
        qx[p] = sx[i]; qy[p] = sy[i]; qz[p] = sz[i];
        const uint32_t cb = (uint32_t)(__atomic_load_n(&keys[i], __ATOMIC_RELAXED) >> 32);      // the candidate's d2 (or what other slices published)
        ok[p] = fabsf(qx[p]) < 1e18f && fabsf(qy[p]) < 1e18f && fabsf(qz[p]) < 1e18f && cb < 0x7F7FFFFFu;
        thr[p] = ok[p] ? __uint_as_float(cb) : -INFINITY;
        L.q[slot] = make_float4(qx[p], qy[p], qz[p], 0.0f);
        L.best[slot] = ~0ull;
        if (!ok[p]) { qx[p] = 0.0f; qy[p] = 0.0f; qz[p] = 0.0f; }                                // (finite operands; thr = -inf: no flag, ever)
        const unsigned long long okm = __builtin_amdgcn_ballot_w64(ok[p]);
        okg[2 * p] = (uint32_t)okm == 0xFFFFFFFFu; okg[2 * p + 1] = (uint32_t)(okm >> 32) == 0xFFFFFFFFu;
    }
    if (stopv) return;
    unsigned long long st_l1 = 0, st_l1flag = 0, st_l2 = 0, st_eval = 0, st_flushes = 0;       // diagnostics (stats != nullptr)
    unsigned long long clk0 = 0, rt0 = 0;
    if (stats) { clk0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
    const uint32_t sb = sl * l1_per_slice, se = min(sb + l1_per_slice, n_l1);
    f32x16 zero;
#pragma unroll
    for (int j = 0; j < 16; j++) zero[j] = 0.0f;
    uint32_t cnt = 0;                                         // entries in the wave's list of flagged (query, chunk) pairs (wave-uniform)
    auto refresh = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int p = 0; p < QG / 2; p++) {
            const uint32_t fb = (uint32_t)(L.best[(2 * p + (h ? 1 : 0)) * 32 + n] >> 32);       // (~0 >> 32 is a NaN pattern: fminf keeps thr)
            thr[p] = ok[p] ? fminf(thr[p], __uint_as_float(fb)) : thr[p];
        }
    };
    // the B operands of the wave's four groups for one (level-1 or level-2) super-tile: lane (n, h) builds the whole operand of query n of group
    // 2 p + h, the halves change places by v_permlane32_swap (nn1_strack_kernel)
    auto setup = [&](const float4 C, bool level1, uint4 (&bq)[QG]) {
        const float sc2 = C.w * C.w;
#pragma unroll
        for (int p = 0; p < QG / 2; p++) {
            uint32_t P[4], Q[4];
            if (level1) st_setup_l1(qx[p], qy[p], qz[p], C, thr[p], sc2, P, Q);
            else st_setup(qx[p], qy[p], qz[p], C, thr[p], sc2, P, Q);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const auto r = __builtin_amdgcn_permlane32_swap(P[j], Q[j], false, false);
                P[j] = r[0]; Q[j] = r[1];
            }
            bq[2 * p] = make_uint4(P[0], P[1], P[2], P[3]);
            bq[2 * p + 1] = make_uint4(Q[0], Q[1], Q[2], Q[3]);
        }
    };
    // LEVEL 2 over the collected tiles: STRACK's per-record filter for the groups that flagged the tile; flagged (query, chunk) pairs are listed
    // and evaluated exactly
    auto level2 = [&](uint32_t n_tiles) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // (lane 0 wrote the list)
        uint32_t curS = 0xFFFFFFFFu;
        uint4 bq2[QG];
#pragma unroll
        for (int g = 0; g < QG; g++) bq2[g] = make_uint4(0, 0, 0, 0);
        for (uint32_t k = 0; k < n_tiles; k++) {
            const uint32_t E = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.tiles[k]), T = E & 0x0FFFFFFFu, gm = E >> 28;
            const uint4 A = ops[(size_t)T * 64 + lane];
            const uint32_t S = T / (BT_SUPER / 32);
            if (S != curS) { curS = S; setup(centres[S], false, bq2); }
            const uint32_t chunk = 2u * T + (h ? 1u : 0u);
#pragma unroll
            for (int g = 0; g < QG; g++) {
                if (!((gm >> g) & 1u)) continue;              // (wave-uniform)
                const uint32_t og = s2_or16(__builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, bq2[g]), zero, 0, 0, 0));
                if (stats) st_l2++;
                const unsigned long long m = __builtin_amdgcn_ballot_w64((int)og < 0);
                if (!m) continue;
                const uint32_t kf = (uint32_t)__popcll(m);
                if (cnt + kf > (uint32_t)S2_CAP) { s2_flush<QG>(L, cnt, records, lane); st_flushes++; st_eval += cnt; cnt = 0; refresh(); curS = 0xFFFFFFFFu; }
                if ((int)og < 0) L.list[cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (chunk << 7) | (uint32_t)(g * 32) | n;
                cnt += kf;
            }
            if (cnt >= flush_at) { s2_flush<QG>(L, cnt, records, lane); st_flushes++; st_eval += cnt; cnt = 0; refresh(); curS = 0xFFFFFFFFu; }
        }
        if (cnt) { s2_flush<QG>(L, cnt, records, lane); st_flushes++; st_eval += cnt; cnt = 0; }
        refresh();
    };
    uint32_t n_tiles = 0;                                     // level-2 tiles collected (wave-uniform)
    for (uint32_t S1 = sb; S1 < se; S1++) {
        uint4 bq1[QG];
        setup(l1_centres[S1], true, bq1);
        uint4 A = l1_ops[(size_t)S1 * 8 * 64 + lane];
#pragma unroll 1
        for (uint32_t t = 0; t < 8; t++) {
            const uint32_t T1 = S1 * 8 + t;
            const uint4 An = l1_ops[(size_t)min(T1 + 1, n_l1 * 8 - 1) * 64 + lane];             // the next level-1 tile, one ahead
            uint32_t anyg[QG];
#pragma unroll
            for (int g = 0; g < QG; g++)
                anyg[g] = s2_or16(__builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, bq1[g]), zero, 0, 0, 0));
            if (stats) st_l1 += QG;
            if (__builtin_amdgcn_ballot_w64((int)(anyg[0] | anyg[1] | anyg[2] | anyg[3]) < 0)) {
                // rare: some group may need some chunk of this level-1 tile.  Which chunks: the accumulators once more, one ballot per chunk pair
                // (lanes < 32 hold chunks 0..15 of the tile, lanes >= 32 chunks 16..31: accumulator i <-> chunk 16 h + i)
                uint32_t tmask[QG];                           // bit k: level-2 tile k of this level-1 tile (chunks 2 k, 2 k + 1) flagged by the group
#pragma unroll
                for (int g = 0; g < QG; g++) {
                    tmask[g] = 0u;
                    if (!__builtin_amdgcn_ballot_w64((int)anyg[g] < 0)) continue;
                    const f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, bq1[g]), zero, 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 16; i += 2) {
                        const unsigned long long m = __builtin_amdgcn_ballot_w64((int)(__float_as_uint(acc[i]) | __float_as_uint(acc[i + 1])) < 0);
                        tmask[g] |= ((uint32_t)m != 0u ? 1u : 0u) << (i / 2);                  // chunks i, i + 1 of the lower half: tile i / 2
                        tmask[g] |= ((uint32_t)(m >> 32) != 0u ? 1u : 0u) << (8 + i / 2);      // chunks 16 + i, 17 + i: tile 8 + i / 2
                    }
                }
                uint32_t un = tmask[0] | tmask[1] | tmask[2] | tmask[3];
                if (stats) st_l1flag += (unsigned long long)__popc(un);
                if (n_tiles + (uint32_t)__popc(un) > (uint32_t)S2_TILES) { level2(n_tiles); n_tiles = 0; setup(l1_centres[S1], true, bq1); }   // (list full: thresholds fell — rebuilt)
                while (un) {                                  // wave-uniform
                    const uint32_t k = (uint32_t)__builtin_ctz(un);
                    un &= un - 1u;
                    const uint32_t gm = ((tmask[0] >> k) & 1u) | (((tmask[1] >> k) & 1u) << 1) | (((tmask[2] >> k) & 1u) << 2) | (((tmask[3] >> k) & 1u) << 3);
                    const uint32_t T2 = T1 * 16u + k;
                    if ((size_t)T2 * 32 < n_rec) { if (lane == 0) L.tiles[n_tiles] = T2 | (gm << 28); n_tiles++; }      // (tiles of the index's padding hold nothing)
                }
            }
            A = An;
        }
        // the tiles this level-1 super-tile flagged: filtered and evaluated before the next one's operands are built (thresholds fall)
        if (n_tiles) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); level2(n_tiles); n_tiles = 0; }
    }
#pragma unroll
    for (int g = 0; g < QG; g++) {
        unsigned long long kbest = L.best[g * 32 + n];
        if (!okg[g] && sb < se) {
            // a query without finite coordinates or without a candidate: the wave scans the slice exactly for this group, each half-lane one half
            // of it (rare: NaN / inf queries, a seed kernel that found nothing acceptable)
            const float4 q = L.q[g * 32 + n];
            const uint32_t r0 = min(sb * (uint32_t)BT_L1_SUPER, n_rec), r1 = (uint32_t)min((unsigned long long)se * BT_L1_SUPER, (unsigned long long)n_rec), mid = r0 + (r1 - r0) / 2;
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
    if (stats && threadIdx.x == 0) {                                      // shader clock under this kernel's load: cycles / 100 MHz ticks (bench.py)
        atomicAdd(&stats[4], (unsigned long long)__builtin_amdgcn_s_memtime() - clk0);
        atomicAdd(&stats[5], (unsigned long long)__builtin_amdgcn_s_memrealtime() - rt0);
    }
    if (stats && lane == 0) {
        if (st_flushes) atomicAdd(&stats[2], st_flushes);                 // joint evaluations (wave level)
        if (st_eval) atomicAdd(&stats[6], st_eval);                       // (query, chunk) pairs evaluated exactly
        if (st_l1) atomicAdd(&stats[8], st_l1);                           // level-1 MFMAs
        if (st_l1flag) atomicAdd(&stats[9], st_l1flag);                   // level-2 tiles flagged by level 1 (per wave and level-1 tile: union of its groups)
        if (st_l2) atomicAdd(&stats[10], st_l2);                          // level-2 MFMAs
    }
}

// grid_stile.hpp — part of grid.hip (included there, inside namespace pcr, behind the cell walk's kernels).
//
// SIGN TILE SEARCH (round 4): the seeded searches of an ICP loop over a large target, rebuilt on the machinery of the exhaustive
// search's sign filter (nn1_strack_kernel, nn1_brute.hip).  The tile search of round 3 (removed since: DESIGN.md 5b-r3) built the f16 operand of every
// record per visit, tracked first / second minima with v_min3 / v_med3 and spent the rest of its time on per-wave row scans: 138 vector
// instructions per MFMA.  Here
//   * the records are the target's Morton-ordered matrix-core index (BtIndex: 256-record super-tiles with a centre, a power-of-two scale
//     and PRECOMPUTED f16 operands — what STRACK scans exhaustively), found through a table of coarse Morton cells (cell_start: the
//     records of a cell of 4 x 4 x 4 lattice cells are one contiguous range) and thinned by the bounding spheres of their 16-record runs;
//   * one wave owns 64 consecutive queries of the loop's sorted working cloud (lane l = query l); the queries of a pass share ONE list of
//     candidate tiles of 32 records: every tile whose run spheres reach the pass's box widened by its largest ball;
//   * per tile and group of 32 queries ONE v_mfma_f32_32x32x16_f16 with the query's threshold in the two free K-slots: the accumulator
//     is bound - threshold, its SIGN BIT says whether the record can matter (st_setup / st_theta, grid_common.hpp: the error analysis
//     is STRACK's, its one assumption is measured on the device by mfma_verdict); the vector ALU ORs 16 accumulators (8 v_or3_b32) and
//     the wave tests one word per tile.  Flagged (query, 16-record chunk) pairs go to a wave-private list and are evaluated together
//     with the exact A1 arithmetic, 16 lanes per chunk; tiles that touch the pass's box come first and the thresholds fall to what they
//     held before the farther tiles are filtered;
//   * far queries (ball beyond bmax, no previous winner, non-finite coordinates, passes with too many cells / tiles) are deferred to
//     the segmented list the cell walk serves in its own launch (list mode of nn1_grid_kernel).
// Nothing is decided approximately: a record at or below a member's threshold (i) lies in a coarse cell the widened box reaches (the
// cell of a coordinate is a monotone function of it: bt_fine_cell), (ii) in a tile one of whose run spheres is within the largest ball
// of the box (margins of sphere_may_win), (iii) raises its sign (st_theta rounds the threshold up by more than the accumulation error
// the device check allows), and (iv) is then evaluated exactly; the minimum over (d2 bits, original index) is the canonical answer.
// Winner positions stay in the numbering of the cell grid (ctx->wpos: what the walk seeds from and the Kabsch pass gathers by): a new
// winner's position is translated once per query (g_of_b, a coherent gather — neighbouring queries win neighbouring records).
// Matches: registration.cpp:925-941 (same correspondences, same gate semantics as the bounded walk).
#pragma once

#ifndef PCR_SL_KEEP
#define PCR_SL_KEEP 768
#endif
constexpr int SL_KEEP = PCR_SL_KEEP;                 // candidate tiles a pass may keep (24 576 records); the launch passes the limit in force
constexpr int SL_CAP = 128;                  // entries of a wave's list of flagged chunks
constexpr uint32_t SL_BT = 0x80000000u;      // tag of a position in the Morton-ordered records (untagged: a position in the cell grid's records)

// sum over the wave, the same value in every lane (a heuristic's input: the order of the additions is irrelevant)
__device__ __forceinline__ float wave_sum_uniform(float v)
{
    v += __uint_as_float(dpp_mov<0xB1>(__float_as_uint(v)));
    v += __uint_as_float(dpp_mov<0x4E>(__float_as_uint(v)));
    v += __uint_as_float(dpp_mov<0x141>(__float_as_uint(v)));
    v += __uint_as_float(dpp_mov<0x140>(__float_as_uint(v)));
    const int b = (int)__float_as_uint(v);
    return (__uint_as_float((uint32_t)__builtin_amdgcn_readlane(b, 0)) + __uint_as_float((uint32_t)__builtin_amdgcn_readlane(b, 16))) +
           (__uint_as_float((uint32_t)__builtin_amdgcn_readlane(b, 32)) + __uint_as_float((uint32_t)__builtin_amdgcn_readlane(b, 48)));
}


struct StileWaveLds {
    float4 q[64];                            // the wave's queries (the lanes that evaluate a flagged chunk are not the owning ones)
    unsigned long long best[64];             // (d2 bits << 32 | original index) found so far
    uint32_t pos[64];                        // where: cell-grid position, or SL_BT | position in the Morton-ordered records
    uint32_t seedchunk[64];                  // the run of 16 Morton-ordered records around the query's seed: evaluated up front, its flags are ignored
    uint32_t off[65];                        // first flattened tile of every coarse cell of a batch
    uint32_t rb[64];                         // first tile of every coarse cell of a batch
    uint32_t tiles[SL_KEEP];                 // candidate tiles (bits 30 / 31: the halves of the wave whose box they reach)
    uint32_t list[SL_CAP];                   // (chunk << 7) | query slot
};

// The listed chunks against their queries: FOUR lanes per chunk, four records each, sixteen chunks per round.  (STRACK's flush takes sixteen
// lanes per chunk: one coalesced 256-byte load, but four DPP steps of a 64-bit minimum per record evaluated — 24 vector instructions per
// listed chunk against 7 here; this kernel is bound by vector issue and lists 1.5 chunks per query.)
__device__ __forceinline__ void sl_flush(StileWaveLds& L, uint32_t cnt, const float4* __restrict__ records, uint32_t lane)
{
#ifdef PCR_SL_T_NOFLUSH                                       // (timing builds only: what the joint evaluations cost — wrong answers)
    return;
#endif
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (uint32_t e0 = 0; e0 < cnt; e0 += 16) {
        const uint32_t e = e0 + (lane >> 2);
        const bool valid = e < cnt;
        const uint32_t ent = L.list[valid ? e : 0];
        const uint32_t slot = ent & 127u;
        const float4 q = L.q[slot];
        const uint32_t p0 = (ent >> 7) * 16u + (lane & 3u);       // records p0, p0 + 4, p0 + 8, p0 + 12: the four lanes of a chunk read 64 contiguous bytes per load
        float4 rec[4];
#pragma unroll
        for (int j = 0; j < 4; j++) rec[j] = records[p0 + 4u * j];                // (padding records: x = +inf, never accepted)
        unsigned long long mine = ~0ull;
        uint32_t pm = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float dx = q.x - rec[j].x, dy = q.y - rec[j].y, dz = q.z - rec[j].z;
            const uint32_t d = __float_as_uint((dx * dx + dy * dy) + dz * dz);   // A1, unfused
            const unsigned long long key = ((unsigned long long)d << 32) | __float_as_uint(rec[j].w);
            if (d < 0x7F7FFFFFu && key < mine) { mine = key; pm = p0 + 4u * j; }   // FLT_MAX gate
        }
        if (!valid) mine = ~0ull;
        unsigned long long key = mine;
#define PCR_SL_MIN(CTRL) { const unsigned long long w = ((unsigned long long)dpp_mov<CTRL>((uint32_t)(key >> 32)) << 32) | dpp_mov<CTRL>((uint32_t)key); \
                           key = w < key ? w : key; }
        PCR_SL_MIN(0xB1) PCR_SL_MIN(0x4E)                          // quad xor 1, xor 2: the four lanes of the chunk
#undef PCR_SL_MIN
        const bool winner = mine == key && mine != ~0ull;          // (original indices are unique: one lane of the four)
        if (winner) atomicMin(&L.best[slot], mine);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // several chunks of one query may sit in the same round: only the lane whose key IS the query's best now notes its position
        if (winner && L.best[slot] == mine) L.pos[slot] = SL_BT | pm;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}

// OR of the sign-carrying words of one accumulator tile (8 v_or3_b32)
__device__ __forceinline__ uint32_t sl_or16(const f32x16 acc)
{
    uint32_t a = __float_as_uint(acc[0]) | __float_as_uint(acc[1]) | __float_as_uint(acc[2]);
#pragma unroll
    for (int j = 3; j + 1 < 16; j += 2) a = a | __float_as_uint(acc[j]) | __float_as_uint(acc[j + 1]);
    return a | __float_as_uint(acc[15]);
}

// the seed of a COLD tile search: the best of up to 32 records, evenly spread over the records of the query's own coarse cell (Morton order inside
// it: spread in space too), as a winner position in the cell grid's numbering; 0xFFFFFFFF when the cell is empty (the query is deferred to the walk)
__global__ __launch_bounds__(GR_BLOCK) void stile_seed_kernel(const float4* __restrict__ records, const uint32_t* __restrict__ cell_start, const uint32_t* __restrict__ g_of_b,
                                                              float klx, float kly, float klz, float kinv, int cshift,
                                                              const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz, uint32_t ns,
                                                              uint32_t* __restrict__ wpos, const int* __restrict__ stop, uint32_t own, uint32_t per)
{
    const uint32_t i = blockIdx.x * GR_BLOCK + threadIdx.x;
    if (i >= ns || (stop && (stop[0] | stop[1]))) return;
    const float qx = sx[i], qy = sy[i], qz = sz[i];
    uint32_t out = 0xFFFFFFFFu;
    if (finite3(qx, qy, qz)) {
        const int fs = cshift / 3;
        unsigned long long best = ~0ull;
        uint32_t bp = 0;
        // the own cell with up to `own` samples, the 26 around it with up to `per` each (a query half a metre off its surface sits in an empty
        // cell of 31 cm; its neighbours are not); nothing there: the same one and two levels up (cells of 2 and 4 edges — an aligned block of
        // 8 / 64 cells is one contiguous range of the Morton-ordered records too)
        for (int lvl = 0; lvl < 3 && best == ~0ull; lvl++) {
            const int sh = fs + lvl;
            const uint32_t side = 1024u >> sh;                // cells per axis at this level
            const uint32_t cx = bt_fine_cell(qx, klx, kinv) >> sh, cy = bt_fine_cell(qy, kly, kinv) >> sh, cz = bt_fine_cell(qz, klz, kinv) >> sh;
            for (int dz = -1; dz <= 1; dz++)
                for (int dy = -1; dy <= 1; dy++)
                    for (int dx = -1; dx <= 1; dx++) {
                        const uint32_t x = cx + (uint32_t)dx, y = cy + (uint32_t)dy, z = cz + (uint32_t)dz;
                        if (x >= side || y >= side || z >= side) continue;             // (also: wrapped below zero)
                        const uint32_t code = bt_morton(x, y, z);
                        const uint32_t b = cell_start[code << (3 * lvl)], e = cell_start[(code + 1u) << (3 * lvl)];
                        if (b >= e) continue;
                        const uint32_t cnt = e - b, take = min(cnt, (dx | dy | dz) == 0 ? own : per);
                        for (uint32_t k0 = 0; k0 < take; k0 += 4) {
                            float4 rec[4];
                            uint32_t p[4];
#pragma unroll
                            for (int u = 0; u < 4; u++) {
                                const uint32_t k = min(k0 + (uint32_t)u, take - 1u);
                                p[u] = b + (uint32_t)(((unsigned long long)k * cnt) / take);
                                rec[u] = records[p[u]];
                            }
#pragma unroll
                            for (int u = 0; u < 4; u++) {
                                const float ex = qx - rec[u].x, ey = qy - rec[u].y, ez = qz - rec[u].z;
                                const uint32_t d = __float_as_uint((ex * ex + ey * ey) + ez * ez);   // A1, unfused
                                const unsigned long long key = ((unsigned long long)d << 32) | p[u];
                                if (d < 0x7F7FFFFFu && key < best) { best = key; bp = p[u]; }
                            }
                        }
                    }
        }
        if (best != ~0ull) out = g_of_b[bp];
    }
    wpos[i] = out;
}

// profile build (-DPCR_SL_PROF, tools/stile_prof.py): one wave in 16 stamps its phases with s_memrealtime and adds them to the diagnostics words
#ifdef PCR_SL_PROF
#define PCR_SL_TICK(acc) { pt_b = __builtin_amdgcn_s_memrealtime(); acc += pt_b - pt_a; pt_a = pt_b; }
#else
#define PCR_SL_TICK(acc)
#endif
#ifndef PCR_STILE_WAVES
#define PCR_STILE_WAVES 5
#endif
template <bool STATS>
__global__ __launch_bounds__(GR_BLOCK, STATS ? 1 : PCR_STILE_WAVES) void nn1_stile_kernel(
    const float4* __restrict__ records, const uint4* __restrict__ ops16, const float4* __restrict__ centres, const float4* __restrict__ spheres,
    const uint32_t* __restrict__ cell_start, const uint32_t* __restrict__ g_of_b, const uint32_t* __restrict__ b_of_g, const float4* __restrict__ grid_records, uint32_t n_grid,
    float klx, float kly, float klz, float kinv, int cshift,
    const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz, uint32_t ns,
    unsigned long long* __restrict__ keys, const int* __restrict__ stop, unsigned long long* __restrict__ stats, float cap2,
    uint32_t* __restrict__ wpos, uint32_t* __restrict__ defer_list, uint32_t* __restrict__ defer_count, uint32_t* __restrict__ defer_queue, uint32_t xcd_run,
    float bmax, uint32_t n_waves, uint32_t n_groups32, float lim_k, float reach_k, uint32_t keep_max, uint32_t cell_max, uint32_t min_members,
    uint32_t flush_at, uint32_t dense_at, float split_at, uint32_t n_super, uint32_t max_pass, uint32_t keep_small, float lim_floor, uint32_t sshift)
{
    const int stopv = stop ? (stop[0] | stop[1]) : 0;         // requested here, tested after the query loads are on their way
    __shared__ StileWaveLds lds_all[GR_BLOCK / 64];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 31;
    const bool h = lane >= 32;
    StileWaveLds& L = lds_all[wave];
    const uint32_t vb = xcd_run ? xcd_block(blockIdx.x, xcd_run) : blockIdx.x;
    const uint32_t wv = vb * (GR_BLOCK / 64) + wave;
    if (wv >= n_waves) return;                                // a surplus wave of the padded launch (wave-uniform; no workgroup barrier below)
    const uint32_t i = wv * 64 + lane;
    const bool valid = i < ns;
    const uint32_t ic = min(i, ns - 1);
    float qx = sx[ic], qy = sy[ic], qz = sz[ic];
    const uint32_t pp0 = wpos[ic];
    if (stopv) return;
#ifdef PCR_SL_PROF
    unsigned long long pt_pro = 0, pt_list = 0, pt_tiles = 0, pt_epi = 0, pt_a = __builtin_amdgcn_s_memrealtime(), pt_b = 0;
    const unsigned long long pt_begin = pt_a;
#endif
    // the caller's gate as the initial bound, then the previous winner (nn1_grid_kernel: same rules, same "none")
    const unsigned long long bound0 = (cap2 > 0.0f && cap2 < 1e30f) ? (((unsigned long long)__float_as_uint(cap2) << 32) | 0xFFFFFFFFull) : KEY_NONE;
    unsigned long long best = bound0;
    uint32_t bestp = 0;
    // a query the filter can serve: finite coordinates of a magnitude st_setup may square (else: the walk)
    const bool fin = fabsf(qx) < 1e18f && fabsf(qy) < 1e18f && fabsf(qz) < 1e18f;
    // THE SEED IS A RUN: the previous winner's whole run of 16 Morton-ordered records, evaluated exactly by the owning lane (neighbouring
    // queries have neighbouring seeds: the sixteen loads of a lane hit lines its neighbours fetch too).  The best of the run is a tighter
    // threshold than the seed alone, and the one flag every query would raise for certain — its seed's own chunk — need not be followed.
    uint32_t seedc = 0xFFFFFFFFu;
    if (fin && pp0 < n_grid) {
        const uint32_t bp = b_of_g[pp0];
        if (bp != 0xFFFFFFFFu) {
            seedc = bp >> 4;
            const float4* rp = records + (size_t)seedc * 16;
#pragma unroll 8
            for (int j = 0; j < 16; j++) {
                const float4 rec = rp[j];
                const float dx = qx - rec.x, dy = qy - rec.y, dz = qz - rec.z;
                const uint32_t d = __float_as_uint((dx * dx + dy * dy) + dz * dz);   // A1, unfused
                const unsigned long long kk = ((unsigned long long)d << 32) | __float_as_uint(rec.w);
                if (d < 0x7F7FFFFFu && kk < best) { best = kk; bestp = SL_BT | (seedc * 16u + (uint32_t)j); }
            }
        }
    }
    L.seedchunk[lane] = seedc;
    L.q[lane] = make_float4(qx, qy, qz, 0.0f);
    L.best[lane] = best;
    L.pos[lane] = bestp;
    // radius of the ball that holds the answer (never reasoned about below the trusted range: grid.hip TRUST)
    float rho = __builtin_inff();
    if (fin && best != KEY_NONE) rho = sqrtf(fmaxf(__uint_as_float((uint32_t)(best >> 32)), TRUST2) * 1.0001f) * 1.00001f;
    // ball limit of the wave: bmax, and lim_k x the mean ball of the queries within bmax (the box of a pass is widened by
    // the LARGEST ball among its members)
    const bool near = valid && fin && rho <= bmax;
    const unsigned long long near_m = __ballot(near);
    const float mean = wave_sum_uniform(near ? rho : 0.0f) / (float)max((uint32_t)__popcll(near_m), 1u);
    const float lim = fmaxf(lim_k * mean, lim_floor);         // (never below lim_floor: at a settled pose the balls are noise-sized and any factor of their mean would cut into the tail)
    const bool member = near && rho <= lim;
    bool deferred = valid && !member;
    if (!fin) { qx = 0.0f; qy = 0.0f; qz = 0.0f; }             // (finite operands for the matrix pipe; such a lane never gets a threshold)
    unsigned long long st_cand = 0, st_cells = 0, st_sph = 0, st_load = 0, st_eval = 0, st_flushes = 0, st_ext = 0, st_rho = 0, st_mfma = 0, st_nsetup = 0;   // diagnostics (STATS builds only)

    unsigned long long remaining = __ballot(member);
    if ((uint32_t)__popcll(remaining) < min_members) remaining = 0;     // (a wave of mostly far queries: the walk takes all of it)
    const float reach = reach_k * lim;
    int n_pass = 0;
    f32x16 zero;
#pragma unroll
    for (int j = 0; j < 16; j++) zero[j] = 0.0f;
    const int fs = cshift / 3;                                // fine cell -> coarse cell
    PCR_SL_TICK(pt_pro)
    for (uint32_t pass = 0; pass < max_pass && remaining; pass++) {
        const int lead = (int)__builtin_ctzll(remaining);
        const float lqx = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(qx), lead)),
                    lqy = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(qy), lead)),
                    lqz = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(qz), lead));
        // (measured and dropped: a last pass that takes every query left, however far apart — heavy passes and overflows, 1.63 -> 1.85 ms; eight
        // passes instead of three: 1.63 -> 1.78 ms)
        const bool in = member && ((remaining >> lane) & 1ull) && fabsf(qx - lqx) <= reach && fabsf(qy - lqy) <= reach && fabsf(qz - lqz) <= reach;
        n_pass++;
        const unsigned long long inmask = __ballot(in);
        remaining &= ~inmask;
        // ONE BOX PER QUARTER of the wave (16 consecutive queries = one DPP row: its minimum is four DPP steps and one v_readlane) and its
        // largest ball.  A tile is filtered for a HALF (queries 0..31 = the columns of the first MFMA, 32..63 = those of the second) only if it
        // reaches the box of one of that half's quarters: where the sorted order jumps inside the wave the quarters are compact clusters,
        // and the bounding box of all 64 queries would hold several times the records any of them needs
        float blo[4][3], bhi[4][3], rmax[4];
        {
            auto rows = [&](float v, float (&out)[4]) {       // minimum over each row of 16 lanes
                v = fminf(v, __uint_as_float(dpp_mov<0xB1>(__float_as_uint(v))));
                v = fminf(v, __uint_as_float(dpp_mov<0x4E>(__float_as_uint(v))));
                v = fminf(v, __uint_as_float(dpp_mov<0x141>(__float_as_uint(v))));
                v = fminf(v, __uint_as_float(dpp_mov<0x140>(__float_as_uint(v))));
                const int b = (int)__float_as_uint(v);
#pragma unroll
                for (int r = 0; r < 4; r++) out[r] = __uint_as_float((uint32_t)__builtin_amdgcn_readlane(b, 16 * r));
            };
            float t[4];
            rows(in ? qx : __builtin_inff(), t); for (int r = 0; r < 4; r++) blo[r][0] = t[r];
            rows(in ? -qx : __builtin_inff(), t); for (int r = 0; r < 4; r++) bhi[r][0] = -t[r];
            rows(in ? qy : __builtin_inff(), t); for (int r = 0; r < 4; r++) blo[r][1] = t[r];
            rows(in ? -qy : __builtin_inff(), t); for (int r = 0; r < 4; r++) bhi[r][1] = -t[r];
            rows(in ? qz : __builtin_inff(), t); for (int r = 0; r < 4; r++) blo[r][2] = t[r];
            rows(in ? -qz : __builtin_inff(), t); for (int r = 0; r < 4; r++) bhi[r][2] = -t[r];
            rows(in ? -rho : 0.0f, t); for (int r = 0; r < 4; r++) rmax[r] = -t[r];
        }
        const bool actq[4] = { (inmask & 0xFFFFull) != 0, ((inmask >> 16) & 0xFFFFull) != 0, ((inmask >> 32) & 0xFFFFull) != 0, (inmask >> 48) != 0 };
        const float rho_max = fmaxf(fmaxf(rmax[0], rmax[1]), fmaxf(rmax[2], rmax[3]));
        // coarse cells the widened boxes reach (their union): the cell of a coordinate is a monotone function of it (bt_fine_cell: the very
        // expression the records were binned with)
        const float klo[3] = { klx, kly, klz };
        uint32_t c0[3], c1[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            float lo = __builtin_inff(), hi = -__builtin_inff();
#pragma unroll
            for (int g = 0; g < 4; g++)
                if (actq[g]) {
                    const float ext = rmax[g] * 1.0001f;
                    lo = fminf(lo, blo[g][a] - ext - fabsf(blo[g][a]) * 1e-6f);
                    hi = fmaxf(hi, bhi[g][a] + ext + fabsf(bhi[g][a]) * 1e-6f);
                }
            c0[a] = bt_fine_cell(lo, klo[a], kinv) >> fs;
            c1[a] = bt_fine_cell(hi, klo[a], kinv) >> fs;
        }
        const uint32_t nx = c1[0] - c0[0] + 1, ny = c1[1] - c0[1] + 1, nz = c1[2] - c0[2] + 1, ncell = nx * ny * nz;
        bool ok = ncell <= cell_max;
        // large balls (the first tile searches of a loop): the tiles that touch a box go first, and the thresholds fall to what they held before
        // the farther ones are filtered; small balls: one list in ascending order (fewer super-tile changes)
        const bool split = rho_max > split_at;
        // small balls (a settled pose): a pass that still collects more than keep_small tiles straddles a jump of the sorted order — a handful of
        // such passes were the tail of the whole launch (10 M: 1.53 -> 1.41 ms per converged search, a 1 / 8 shard 0.32 -> 0.29); the walk takes them
        const uint32_t keep_eff = split ? keep_max : min(keep_max, keep_small);
        uint32_t nn = 0, nf = 0;                              // tiles from the front of L.tiles, and (split) from its back
        float lim2[4];
#pragma unroll
        for (int g = 0; g < 4; g++) lim2[g] = fmaxf(rmax[g] * rmax[g], TRUST2) * 1.0001f;
        for (uint32_t cb = 0; ok && cb < ncell; cb += 64) {
            // one coarse cell per lane: its record range -> its tiles (a tile that straddles the range is taken whole: its records are genuine)
            uint32_t tb = 0, te = 0;
            const uint32_t c = cb + lane;
            if (c < ncell) {
                const uint32_t cy_z = c / nx, cx = c0[0] + (c - cy_z * nx), cz_ = cy_z / ny, cy = c0[1] + (cy_z - cz_ * ny), cz = c0[2] + cz_;
                const uint32_t code = bt_morton(cx, cy, cz);
                const uint32_t b = cell_start[code], e = cell_start[code + 1];
                if (b < e) { tb = b >> 5; te = ((e - 1) >> 5) + 1; }
            }
            const uint32_t cnt = te - tb;
            uint32_t inc = row_scan16(cnt);
            const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)inc, 15), t1 = (uint32_t)__builtin_amdgcn_readlane((int)inc, 31),
                           t2 = (uint32_t)__builtin_amdgcn_readlane((int)inc, 47), t3 = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            inc += lane >= 48 ? t0 + t1 + t2 : lane >= 32 ? t0 + t1 : lane >= 16 ? t0 : 0u;
            const uint32_t total = t0 + t1 + t2 + t3;
            L.off[lane] = inc - cnt; L.rb[lane] = tb;
            if (STATS) { st_cells += min(64u, ncell - cb); st_sph += total; }
            for (uint32_t base = 0; base < total; base += 64) {
                const uint32_t f = base + lane;
                int r = 0;
#pragma unroll
                for (int step = 32; step > 0; step >>= 1)
                    if (L.off[r + step] <= f) r += step;
                const uint32_t tile = L.rb[r] + (f - L.off[r]);
                uint32_t hm = 0;                               // halves whose box the tile reaches
                bool touch = false;
                if (f < total) {
                    const float4 s = spheres[tile];
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        // dropped for a quarter only if farther from the box of its queries than its largest ball (margins of sphere_may_win)
                        const float dx = fmaxf(fmaxf(blo[g][0] - s.x, s.x - bhi[g][0]), 0.0f), dy = fmaxf(fmaxf(blo[g][1] - s.y, s.y - bhi[g][1]), 0.0f),
                                    dz = fmaxf(fmaxf(blo[g][2] - s.z, s.z - bhi[g][2]), 0.0f);
                        const float dc2 = (dx * dx + dy * dy) + dz * dz;
                        const float sep = sqrtf(dc2) * 0.99999f - s.w;
                        const bool k1 = actq[g] && s.w >= 0.0f && !(dc2 < 3.0e38f && sep > TRUST && sep * sep * 0.99999f > lim2[g]);
                        hm |= k1 ? (1u << (g >> 1)) : 0u;
                        touch = touch || (k1 && !(sep > 0.0f));
                    }
                }
                const bool front = hm != 0u && (touch || !split), back = hm != 0u && !front;
                const unsigned long long mt = __ballot(front), mf = __ballot(back);
                const unsigned long long below = (1ull << lane) - 1ull;
                if (nn + nf + (uint32_t)__popcll(mt | mf) > keep_eff) { ok = false; break; }
                if (front) L.tiles[nn + (uint32_t)__popcll(mt & below)] = tile | (hm << 30);
                if (back) L.tiles[SL_KEEP - 1 - (nf + (uint32_t)__popcll(mf & below))] = tile | (hm << 30);
                nn += (uint32_t)__popcll(mt); nf += (uint32_t)__popcll(mf);
            }
        }
        if (!ok) { deferred = deferred || in; if (STATS && lane == 0) atomicAdd(&stats[15], (unsigned long long)__popcll(inmask)); continue; }     // too many cells / tiles for one wave: the walk takes these queries ([15])
        const uint32_t nt = nn + nf;                          // (list position k >= nn: the tiles kept at the back, L.tiles[SL_KEEP - 1 - (k - nn)])
        if (STATS) {
            if (lane == 0) { atomicMax(&stats[12], (unsigned long long)nt); }   // [12] most tiles in a pass
            st_load += (uint64_t)nt * 32;
            float e = 0.0f;
            for (int g = 0; g < 4; g++) if (actq[g]) e = fmaxf(e, fmaxf(fmaxf(bhi[g][0] - blo[g][0], bhi[g][1] - blo[g][1]), bhi[g][2] - blo[g][2]));
            st_ext += (uint64_t)(e * 1e6f);                   // um (the largest quarter box)
            st_rho += (uint64_t)(rho_max * 1e6f);
        }
        // ---- the tiles against the pass's queries: STRACK's tile loop (nn1_brute.hip) over a LIST of tiles.  Lane l owns query l: it builds
        // the whole operand of its query per super-tile (st_setup) and the halves change places by v_permlane32_swap — afterwards bq[0] is the
        // B operand of queries 0..31, bq[1] that of queries 32..63.
        PCR_SL_TICK(pt_list)
        float thr = in ? __uint_as_float((uint32_t)(L.best[lane] >> 32)) : -__builtin_inff();
        auto entry_at = [&](uint32_t k) -> uint32_t {
            const uint32_t kk = min(k, max(nt, 1u) - 1u);     // (beyond the list: its last entry once more — never used)
            return (uint32_t)__builtin_amdgcn_readfirstlane((int)L.tiles[kk < nn ? kk : (uint32_t)SL_KEEP - 1u - (kk - nn)]);
        };
        uint32_t cnt = 0;                                     // entries in the wave's list (wave-uniform)
        uint32_t curS = 0xFFFFFFFFu;
        uint4 bq[2] = { make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0) };
        uint32_t E0 = nt > 0 ? entry_at(0) : 0u, E1 = nt > 1 ? entry_at(1) : 0u, E2 = nt > 2 ? entry_at(2) : 0u;
        uint4 A0 = make_uint4(0, 0, 0, 0), A1 = A0;
        float4 Cc = centres[min((E0 & 0x3FFFFFFFu) >> sshift, n_super - 1u)];    // (sshift: 3 = the 256-record super-tiles' operands, 7 = the level-1 super-tiles')
        if (nt > 0) A0 = ops16[(size_t)(E0 & 0x3FFFFFFFu) * 64 + lane];
        if (nt > 1) A1 = ops16[(size_t)(E1 & 0x3FFFFFFFu) * 64 + lane];
        auto refresh = [&]() {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (in) thr = fminf(thr, __uint_as_float((uint32_t)(L.best[lane] >> 32)));
            curS = 0xFFFFFFFFu;                                // the operands carry the old thresholds: rebuilt before the next tile
            // (measured and dropped: rebuilt only where some threshold fell below 90 / 70 / 40 % of what the operands carry — 1.399 -> 1.403 / 1.400 /
            // 1.404 ms per converged 10 M search, the first searches of a loop slower: the setups are not what the loop is made of)
        };
        // (Measured and dropped: three tiles per trip with their operands in three fixed register sets — no copy of a register a load is
        // still writing, loads two tiles ahead of their use — and one copy of the rare path behind the trip: 1.55 -> 1.62 ms per converged
        // search.  The loop is bound by vector issue, not by the operand loads.)
#ifdef PCR_SL_T_NOTILES                                       // (timing builds only: everything but the tile loop — wrong answers)
        for (uint32_t k = 0; k < 0; k++) {
#else
        for (uint32_t k = 0; k < nt; k++) {
#endif
            const uint32_t E = (uint32_t)__builtin_amdgcn_readfirstlane((int)E0), T = E & 0x3FFFFFFFu, hm = E >> 30;    // (wave-uniform: scalar branches, scalar loads)
            const uint4 A = A0;
            A0 = A1; E0 = E1; E1 = E2;
            const uint32_t Tn = (uint32_t)__builtin_amdgcn_readfirstlane((int)E0) & 0x3FFFFFFFu, Tnn = (uint32_t)__builtin_amdgcn_readfirstlane((int)E1) & 0x3FFFFFFFu;
            if (k + 2 < nt) A1 = ops16[(size_t)Tnn * 64 + lane];                     // two tiles ahead
            E2 = entry_at(k + 3);
            const float4 Cn = centres[min(Tn >> sshift, n_super - 1u)];                    // the next tile's super-tile (scalar load, one tile ahead)
            if (split && k == nn && cnt) { sl_flush(L, cnt, records, lane); if (STATS) { st_flushes++; st_eval += cnt; } cnt = 0; refresh(); }   // the boxes' own tiles are done: thresholds fall before the farther ones
            const uint32_t S = T >> sshift;
#ifdef PCR_SL_T_NOSETUP                                       // (timing builds only: what the operand setups cost — wrong answers)
            if (curS == 0xFFFFFFFFu && k == 0) {
#else
            if (S != curS) {
#endif
                curS = S;
                const float4 C = Cc;                          // wave-uniform: .w = the super-tile's scale (a power of two)
                uint32_t P[4], Q[4];
                st_setup(qx, qy, qz, C, thr, C.w * C.w, P, Q);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const auto r = __builtin_amdgcn_permlane32_swap(P[j], Q[j], false, false);
                    P[j] = r[0]; Q[j] = r[1];
                }
                bq[0] = make_uint4(P[0], P[1], P[2], P[3]);
                bq[1] = make_uint4(Q[0], Q[1], Q[2], Q[3]);
                if (STATS) st_nsetup++;
            }
            Cc = Cn;
            // one OR chain per half (8 v_or3_b32 each): which half raised a sign is then known without running the tile's MFMAs again
            uint32_t anyg[2] = { 0u, 0u };
            if (hm == 3u) {
                const f32x16 acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, bq[0]), zero, 0, 0, 0);
                const f32x16 acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, bq[1]), zero, 0, 0, 0);
                anyg[0] = sl_or16(acc0);
                anyg[1] = sl_or16(acc1);
            } else if (hm == 1u) {
                anyg[0] = sl_or16(__builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, bq[0]), zero, 0, 0, 0));
            } else {
                anyg[1] = sl_or16(__builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, bq[1]), zero, 0, 0, 0));
            }
            if (STATS) { st_mfma += hm == 3u ? 2u : 1u; st_cand += 32ull * ((hm & 1u ? (uint64_t)__popc((uint32_t)inmask) : 0u) + (hm & 2u ? (uint64_t)__popc((uint32_t)(inmask >> 32)) : 0u)); }
#ifdef PCR_SL_T_NORARE                                        // (timing builds only: no flag is ever followed — wrong answers)
            if (__builtin_amdgcn_ballot_w64((int)(anyg[0] | anyg[1]) == 12345)) {
#else
            if (__builtin_amdgcn_ballot_w64((int)(anyg[0] | anyg[1]) < 0)) {
#endif
                // rare: some half-lane's chunk (records 32 T + 16 h ...) may hold a record at or below its query's threshold
                const uint32_t chunk = 2u * T + (h ? 1u : 0u);
#pragma unroll
                for (int g = 0; g < 2; g++) {
                    const uint32_t slot = (uint32_t)(g * 32) + n;            // column n of half g
                    const uint32_t og = L.seedchunk[slot] == chunk ? 0u : anyg[g];      // (the seed's own run was evaluated up front)
                    const unsigned long long m = __builtin_amdgcn_ballot_w64((int)og < 0);
                    if (!m) continue;
                    const uint32_t kf = (uint32_t)__popcll(m);
                    if (kf >= dense_at) {
                        // many columns of this half flag the SAME tile (coarse seeds: the first tile searches of a loop): the flagged half-lanes
                        // evaluate their chunk in place — uniform addresses per half (broadcast loads), every lane for its own column's query
                        if ((int)og < 0) {
                            const float4 q = L.q[slot];
                            const float4* rp = records + (size_t)chunk * 16;
                            unsigned long long kb = ~0ull;
                            uint32_t jb = 0;
#pragma unroll 4
                            for (int j = 0; j < 16; j++) {
                                const float4 rec = rp[j];                                   // (padding records: x = +inf, never accepted)
                                const float dx = q.x - rec.x, dy = q.y - rec.y, dz = q.z - rec.z;
                                const uint32_t d = __float_as_uint((dx * dx + dy * dy) + dz * dz);   // A1, unfused
                                const unsigned long long key = ((unsigned long long)d << 32) | __float_as_uint(rec.w);
                                if (d < 0x7F7FFFFFu && key < kb) { kb = key; jb = (uint32_t)j; }    // FLT_MAX gate
                            }
                            if (kb != ~0ull) atomicMin(&L.best[slot], kb);
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                            if (kb != ~0ull && L.best[slot] == kb) L.pos[slot] = SL_BT | (chunk * 16u + jb);   // (the two half-lanes of a column: only the better one)
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        }
                        if (STATS) st_eval += kf;
                        refresh();
                        continue;
                    }
                    if (cnt + kf > (uint32_t)SL_CAP) { sl_flush(L, cnt, records, lane); if (STATS) { st_flushes++; st_eval += cnt; } cnt = 0; refresh(); }
                    if ((int)og < 0) L.list[cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (chunk << 7) | slot;
                    cnt += kf;
                }
                if (cnt >= flush_at) { sl_flush(L, cnt, records, lane); if (STATS) { st_flushes++; st_eval += cnt; } cnt = 0; refresh(); }
            }
        }
        if (cnt) { sl_flush(L, cnt, records, lane); if (STATS) { st_flushes++; st_eval += cnt; } cnt = 0; }
        PCR_SL_TICK(pt_tiles)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (in) {
            const unsigned long long kb = L.best[lane];
            uint32_t wp = L.pos[lane];
            if (wp & SL_BT) wp = g_of_b[wp & ~SL_BT];           // a new winner: its position in the cell grid's records (neighbouring queries: neighbouring lines)
            const uint32_t bidx = (uint32_t)(kb & 0xFFFFFFFFull);
            const uint32_t bits = (bidx == 0xFFFFFFFFu) ? 0x7F800000u : (uint32_t)(kb >> 32);
            keys[i] = ((unsigned long long)bits << 32) | bidx;
            wpos[i] = bidx == 0xFFFFFFFFu ? 0xFFFFFFFFu : wp;
        }
    }
    if (STATS && lane == 0) { atomicAdd(&stats[14], (unsigned long long)__popcll(remaining)); atomicAdd(&stats[13], (unsigned long long)__popcll(__ballot(near && !member))); }   // [14] queries of a fourth cluster, [13] near queries beyond the ball limit
    deferred = deferred || (member && ((remaining >> lane) & 1ull));        // more clusters in one wave than it may serve: the walk takes the rest
    if (min_members > 1u) deferred = deferred || (member && (uint32_t)__popcll(__ballot(member)) < min_members);
    // the deferred queries: segments 2 wv and 2 wv + 1 of the list (one per 32 queries, in query order) and their lengths (every wave writes them)
    {
        const unsigned long long dm = __ballot(deferred);
        const uint32_t dh = h ? (uint32_t)(dm >> 32) : (uint32_t)dm;
        const uint32_t g32 = 2u * wv + (h ? 1u : 0u);
        if (g32 < n_groups32) {
            if (deferred) defer_list[(size_t)g32 * 32 + (uint32_t)__popc(dh & ((1u << n) - 1u))] = i;
            if (n == 0) {
                defer_count[g32] = (uint32_t)__popc(dh);
                if (defer_queue && dh) {                      // the list walk draws its work from here: one item per 8 deferred queries of the segment
                    const uint32_t parts = ((uint32_t)__popc(dh) + 7u) / 8u, base = atomicAdd(&defer_queue[0], parts);
                    for (uint32_t pq = 0; pq < parts; pq++) defer_queue[2u + base + pq] = 4u * g32 + pq;
                }
            }
        }
        if (STATS && lane == 0 && dm) atomicAdd(&stats[6], (unsigned long long)__popcll(dm));     // [6]: queries handed to the cell walk
    }
#ifdef PCR_SL_PROF
    PCR_SL_TICK(pt_epi)
    if (stats && lane == 0 && (wv & 15u) == 0u) {             // [0] prologue [1] boxes + cells + spheres + list [2] tile loops + evaluations [3] write-back + deferral (10 ns ticks), [4] waves, [5] sum of lives, [6] longest
        atomicAdd(&stats[0], pt_pro); atomicAdd(&stats[1], pt_list); atomicAdd(&stats[2], pt_tiles); atomicAdd(&stats[3], pt_epi);
        atomicAdd(&stats[4], 1ull); atomicAdd(&stats[5], pt_a - pt_begin); atomicMax(&stats[6], pt_a - pt_begin);
    }
    return;
#endif
    if (STATS && lane == 0) {
        if (st_cand) atomicAdd(&stats[0], st_cand);                                   // [0]: (query, record) pairs that went through the filter
        if (st_cells) atomicAdd(&stats[1], st_cells);                                 // [1]: coarse cells looked up
        if (st_sph) atomicAdd(&stats[2], st_sph);                                     // [2]: tile spheres tested
        if (st_nsetup) atomicAdd(&stats[3], st_nsetup);                                 // [3]: operand setups (super-tile changes) of the tile loops
        atomicAdd(&stats[7], (unsigned long long)n_pass);                             // [7]: passes
        if (st_ext) atomicAdd(&stats[4], st_ext);                                     // [4], [5]: largest box edge / largest ball of the served passes, um
        if (st_rho) atomicAdd(&stats[5], st_rho);
        if (st_eval) atomicAdd(&stats[8], st_eval);                                   // [8]: (query, chunk) pairs evaluated exactly
        if (st_flushes) atomicAdd(&stats[9], st_flushes);                             // [9]: joint evaluations (wave level)
        if (st_load) atomicAdd(&stats[10], st_load);                                  // [10]: records whose operands the passes loaded (shared by their queries)
        if (st_mfma) atomicAdd(&stats[11], st_mfma);                                  // [11]: MFMAs of the tile loops
    }
}

// grid_tile.hpp — part of grid.hip (included there, inside namespace pcr, after the helpers of the cell walk).
//
// TILE SEARCH: the warm searches of an ICP loop over a large, Morton-ordered index (round 3).  The cell walk (nn1_grid_kernel) gives
// every query a 16-lane sub-group that opens its own rows, tests its own spheres and scans its own runs: ~14 000 lane-operations per
// query at the converged pose of the 10 M pair for ~100 exact distance evaluations (1 200 lane-operations of arithmetic).  But the
// working cloud of the loop is sorted into the order of the target's records (grid_sort_working_cloud), so 32 CONSECUTIVE queries lie
// within centimetres of each other and need the same rows, the same spheres and the same runs.  Here one wave owns such a group:
//   1. every query re-evaluates its previous winner (wpos[], one 16-byte gather): a genuine candidate, hence a ball that holds the answer;
//   2. the group's box (bounding box of its queries, widened by the largest ball) is cut into x-rows of cells ONCE — one row per lane,
//      one pair of cell_start loads per row — and the rows' runs of 16 records are laid out as one index space;
//   3. the runs' bounding spheres are tested against the box, 64 per step, and the survivors compacted into LDS;
//   4. the surviving runs are streamed through LDS, 32 records per step, and every query evaluates every record with the exact A1
//      arithmetic: lanes n and n + 32 hold query n and take one run of the pair each (uniform LDS reads: broadcasts).
// Nothing is decided approximately: a run is dropped only if its sphere lies farther from the group's box than the largest ball of
// the group (same margins as sphere_may_win), so every record within a query's ball is evaluated, ties included; the minimum over
// (d2 bits, original index) is the canonical answer.  A query whose ball is larger than `bmax` (first iterations of a misaligned pair,
// no previous winner, non-finite coordinates), or whose group would need more rows / runs than a wave handles, is APPENDED TO A LIST
// that nn1_grid_kernel walks afterwards in its own launch (list mode): the far queries no longer share their waves with near ones.
// Matches: registration.cpp:925-941 (same correspondences, same gate semantics as the bounded walk: a query with nothing inside the
// gate ends with "none").
#pragma once

constexpr int TL_ROWS = 64;               // x-rows of cells a pass may open (one per lane)
constexpr int TL_KEEP = 1024;             // surviving runs a pass may keep (16 384 records); the launch passes the limit in force (keep_max)
constexpr uint32_t TL_MAX_RUNS = 16384;   // runs in the rows of a pass before the sphere test

struct TileWaveLds {
    uint32_t off[TL_ROWS + 1];            // first flattened run of every row
    uint32_t rb[TL_ROWS];                 // first run of every row
    uint32_t runs[TL_KEEP];               // the surviving runs
    float4 stage[2][32];                  // two runs of 16 records, double-buffered
};

// minimum over the wave, the same value in every lane (all 64 lanes active): four DPP steps inside each row of 16, then the four rows
__device__ __forceinline__ float wave_min_uniform(float v)
{
    v = fminf(v, __uint_as_float(dpp_mov<0xB1>(__float_as_uint(v))));
    v = fminf(v, __uint_as_float(dpp_mov<0x4E>(__float_as_uint(v))));
    v = fminf(v, __uint_as_float(dpp_mov<0x141>(__float_as_uint(v))));
    v = fminf(v, __uint_as_float(dpp_mov<0x140>(__float_as_uint(v))));
    const int b = (int)__float_as_uint(v);
    const float r0 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane(b, 0)), r1 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane(b, 16)),
                r2 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane(b, 32)), r3 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane(b, 48));
    return fminf(fminf(r0, r1), fminf(r2, r3));
}

// the same over each HALF of the wave separately: lo = minimum over lanes 0..31, hi = minimum over lanes 32..63
__device__ __forceinline__ void half_min_uniform(float v, float& lo, float& hi)
{
    v = fminf(v, __uint_as_float(dpp_mov<0xB1>(__float_as_uint(v))));
    v = fminf(v, __uint_as_float(dpp_mov<0x4E>(__float_as_uint(v))));
    v = fminf(v, __uint_as_float(dpp_mov<0x141>(__float_as_uint(v))));
    v = fminf(v, __uint_as_float(dpp_mov<0x140>(__float_as_uint(v))));
    const int b = (int)__float_as_uint(v);
    lo = fminf(__uint_as_float((uint32_t)__builtin_amdgcn_readlane(b, 0)), __uint_as_float((uint32_t)__builtin_amdgcn_readlane(b, 16)));
    hi = fminf(__uint_as_float((uint32_t)__builtin_amdgcn_readlane(b, 32)), __uint_as_float((uint32_t)__builtin_amdgcn_readlane(b, 48)));
}

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

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

// waves per SIMD the kernel is built for (measured on the 10 M pair, converged search / average of the first 20: unbounded = 95 VGPRs,
// 5 waves: 1.67 / 3.63 ms; 6 waves (76 VGPRs): 1.55 / 3.49; 7 waves (72 VGPRs, two registers spilled): 1.50 / 3.43; 8: not reachable)
#ifndef PCR_TILE_WAVES
#define PCR_TILE_WAVES 7
#endif
template <bool STATS>
__global__ __launch_bounds__(GR_BLOCK, STATS ? 1 : PCR_TILE_WAVES) void nn1_tile_kernel(
    const float4* __restrict__ records, const float4* __restrict__ spheres, const uint32_t* __restrict__ cell_start, GridParams g,
    const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz, uint32_t ns,
    unsigned long long* __restrict__ keys, const int* __restrict__ stop, unsigned long long* __restrict__ stats, uint32_t nt, float cap2,
    uint32_t* __restrict__ wpos, uint32_t* __restrict__ defer_list, uint32_t* __restrict__ defer_count, uint32_t xcd_run, float bmax,
    uint32_t n_groups, float lim_k, float reach_k, uint32_t keep_max, int use_filter, uint32_t min_members, uint32_t total_mult)
{
    const int stopv = stop ? (stop[0] | stop[1]) : 0;         // requested here, tested after the query loads are on their way
    __shared__ TileWaveLds lds_all[GR_BLOCK / 64];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 31;
    const bool h = lane >= 32;
    TileWaveLds& L = lds_all[wave];
    const uint32_t vb = xcd_run ? xcd_block(blockIdx.x, xcd_run) : blockIdx.x;
    const uint32_t group = vb * (GR_BLOCK / 64) + wave;
    if (group >= n_groups) return;                            // a surplus wave of the padded launch (wave-uniform; no workgroup barrier below)
    const uint32_t i = group * 32 + n;
    const bool valid = i < ns;
    const uint32_t ic = min(i, ns - 1);
    const float qx = sx[ic], qy = sy[ic], qz = sz[ic];
    const uint32_t pp0 = wpos[ic];
    if (stopv) return;
    // the caller's gate as the initial bound, then the previous winner (nn1_grid_kernel: same rules, same "none")
    const unsigned long long bound0 = (cap2 > 0.0f && cap2 < 1e30f) ? (((unsigned long long)__float_as_uint(cap2) << 32) | 0xFFFFFFFFull) : KEY_NONE;
    unsigned long long best = bound0;
    uint32_t bestp = 0;
    const bool fin = finite3(qx, qy, qz);
    if (fin && pp0 < nt) {
        const float4 rec = records[pp0];
        const float dx = qx - rec.x, dy = qy - rec.y, dz = qz - rec.z;
        const uint32_t d = __float_as_uint((dx * dx + dy * dy) + dz * dz);   // A1, unfused
        const unsigned long long kk = ((unsigned long long)d << 32) | __float_as_uint(rec.w);
        if (d < 0x7F7FFFFFu && kk < best) { best = kk; bestp = pp0; }
    }
    // radius of the ball that holds the answer (never reasoned about below the trusted range: grid.hip TRUST)
    float rho = __builtin_inff();
    if (fin && best != KEY_NONE) rho = sqrtf(fmaxf(__uint_as_float((uint32_t)(best >> 32)), TRUST2) * 1.0001f) * 1.00001f;
    // Ball limit of the group: `bmax`, and 2.5 x the mean ball of the queries within bmax — the box of a pass is widened by the LARGEST
    // ball among its members, so one outlier would make all 32 queries evaluate the records only it needs (at the converged pose of a
    // noisy pair the balls are chi-distributed: 0.1 % of the queries lie beyond 2.5 means); the walk takes the outliers.
    const bool near = valid && fin && rho <= bmax;
    const uint32_t n_near = (uint32_t)__popc((uint32_t)__ballot(near));                       // (low half: one bit per query)
    const float mean = wave_sum_uniform(near && !h ? rho : 0.0f) / (float)max(n_near, 1u);
    const float lim = lim_k * mean;
    const bool member = near && rho <= lim;
    bool deferred = valid && !member;
    unsigned long long st_cand = 0, st_rows = 0, st_sph = 0, st_ext = 0, st_rho = 0, st_filt = 0, st_fall = 0, st_load = 0;  // diagnostics (STATS builds only; wave totals, lane 0)

    // A pass serves the remaining members that lie within two ball limits (Chebyshev) of the first remaining member: where the sorted
    // order jumps — from one octant of a cell to the next, from a cell to its neighbour, across the cloud at the end of a row of cells —
    // the queries before and after the jump form compact clusters, each served with its own small box (three passes at most).
    uint32_t remaining = (uint32_t)__ballot(member);          // bit n = query n (lanes n and n + 32 agree)
    if ((uint32_t)__popc(remaining) < min_members) remaining = 0;     // (a group of mostly far queries: the walk takes all of it)
    const float reach = reach_k * lim;
    int n_pass = 0;
    for (int pass = 0; pass < 3 && remaining; pass++) {
        const int lead = __builtin_ctz(remaining);
        const float lqx = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(qx), lead)),
                    lqy = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(qy), lead)),
                    lqz = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(qz), lead));
        const bool in = member && ((remaining >> n) & 1u) && fabsf(qx - lqx) <= reach && fabsf(qy - lqy) <= reach && fabsf(qz - lqz) <= reach;
        n_pass++;
        const uint32_t inmask = (uint32_t)__ballot(in);
        remaining &= ~inmask;
        // box of the pass: lanes < 32 reduce q, lanes >= 32 reduce -q (one reduction per axis gives both ends), and the largest ball
        float blo[3], bhi[3];
        {
            float nlo, nhi;
            half_min_uniform(in ? (h ? -qx : qx) : __builtin_inff(), nlo, nhi); blo[0] = nlo; bhi[0] = -nhi;
            half_min_uniform(in ? (h ? -qy : qy) : __builtin_inff(), nlo, nhi); blo[1] = nlo; bhi[1] = -nhi;
            half_min_uniform(in ? (h ? -qz : qz) : __builtin_inff(), nlo, nhi); blo[2] = nlo; bhi[2] = -nhi;
        }
        const float rho_max = -wave_min_uniform(in ? -rho : 0.0f);
        // cells the widened box reaches: the cell index is a monotone f32 function of the coordinate (ball_x_cells: same margins)
        const float ext = rho_max + g.slack * g.h;
        int c0[3], c1[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            c0[a] = clampi(cell_coord(blo[a] - ext - fabsf(blo[a]) * 1e-6f, g.lo[a], g.inv_h), 0, g.n[a] - 1);
            c1[a] = clampi(cell_coord(bhi[a] + ext + fabsf(bhi[a]) * 1e-6f, g.lo[a], g.inv_h), 0, g.n[a] - 1);
        }
        const int ny_r = c1[1] - c0[1] + 1, nrow = ny_r * (c1[2] - c0[2] + 1);
        bool ok = nrow <= TL_ROWS;
        uint32_t nk = 0;
        if (ok) {
            // one row per lane: its record range -> its runs (a run that straddles the range is taken whole: its records are genuine)
            uint32_t rb = 0, re = 0;
            if ((int)lane < nrow) {
                const int cy = c0[1] + (int)lane % ny_r, cz = c0[2] + (int)lane / ny_r;
                const uint32_t row = (uint32_t)((cz * g.n[1] + cy) * g.n[0]);
                const uint32_t b = cell_start[row + c0[0]], e = cell_start[row + c1[0] + 1];
                if (b < e) { rb = b / GRID_CHUNK; re = (e - 1) / GRID_CHUNK + 1; }
            }
            // (a run shared with the row before — rows follow each other in memory — is taken once)
            const uint32_t pre = (uint32_t)__shfl_up((int)re, 1, 64);
            if (lane > 0 && rb < re && pre > rb) rb = min(pre, re);
            const uint32_t cnt = re - rb;
            uint32_t inc = row_scan16(cnt);
            const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)inc, 15), t1 = (uint32_t)__builtin_amdgcn_readlane((int)inc, 31),
                           t2 = (uint32_t)__builtin_amdgcn_readlane((int)inc, 47), t3 = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            inc += lane >= 48 ? t0 + t1 + t2 : lane >= 32 ? t0 + t1 : lane >= 16 ? t0 : 0u;
            const uint32_t total = t0 + t1 + t2 + t3;
            L.off[lane] = inc - cnt; L.rb[lane] = rb;
            ok = total <= min(TL_MAX_RUNS, total_mult * keep_max);   // (far more runs than a pass may keep: not worth testing them)
            if (STATS) { st_rows += (uint64_t)nrow; st_sph += ok ? total : 0u; }
            // sphere of every run against the box of the queries: dropped only if farther from it than the largest ball
            const float lim = fmaxf(rho_max * rho_max, TRUST2) * 1.0001f;
            for (uint32_t base = 0; ok && base < total; base += 64) {
                const uint32_t f = base + lane;
                int r = 0;
#pragma unroll
                for (int step = 32; step > 0; step >>= 1)
                    if (L.off[r + step] <= f) r += step;
                const uint32_t run = L.rb[r] + (f - L.off[r]);
                bool keep = false;
                if (f < total) {
                    const float4 s = spheres[run];
                    const float dx = fmaxf(fmaxf(blo[0] - s.x, s.x - bhi[0]), 0.0f), dy = fmaxf(fmaxf(blo[1] - s.y, s.y - bhi[1]), 0.0f),
                                dz = fmaxf(fmaxf(blo[2] - s.z, s.z - bhi[2]), 0.0f);
                    const float dc2 = (dx * dx + dy * dy) + dz * dz;
                    const float sep = sqrtf(dc2) * 0.99999f - s.w;
                    keep = s.w >= 0.0f && !(dc2 < 3.0e38f && sep > TRUST && sep * sep * 0.99999f > lim);
                }
                const unsigned long long mask = __ballot(keep);
                const uint32_t pos = nk + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
                if (keep && pos < (uint32_t)TL_KEEP) L.runs[pos] = run;
                nk += (uint32_t)__popcll(mask);
                if (nk > keep_max) break;
            }
            ok = ok && nk <= keep_max;                        // (keep_max <= TL_KEEP: beyond it the walk's per-query pruning is cheaper)
        }
        if (!ok) { deferred = deferred || in; continue; }     // too many rows / runs for one wave: the walk takes these queries
        if (STATS) {
            st_cand += (uint64_t)nk * GRID_CHUNK * (uint64_t)__popc(inmask);
            st_load += (uint64_t)nk * GRID_CHUNK;                     // records the pass loads ONCE for all its queries
            st_ext += (uint64_t)(fmaxf(fmaxf(bhi[0] - blo[0], bhi[1] - blo[1]), bhi[2] - blo[2]) * 1e6f);      // um
            st_rho += (uint64_t)(rho_max * 1e6f);
        }
        // ---- FILTER (more than four surviving runs): every (query, record) pair of the pass gets its lower bound from the f16 matrix pipe,
        // exactly as in the exhaustive search (nn1_brute.hip, HTRACK: same operand code, same error analysis) — the pass's box is the
        // "super-tile": centre C, power-of-two scale with |t - C| scale <= 2^7 for every record that can matter (a record farther
        // from C than the box reaches lies outside every member's ball: its operand says "never the minimum").  One MFMA per 32 queries
        // x 2 runs; lane (n, h) takes the minimum of its 16 accumulators = the bound of run t + h for query n, and tracks the smallest
        // bound, its run and the second smallest.  Then only the best run is evaluated exactly (8 records per half-lane, per-lane
        // addresses), and the second smallest bound proves that no other run holds a closer or equal record.  A pass in which some
        // member stays unproven (near-ties between runs) takes the exact loop below for all its runs.
        bool settled = false;
        if (use_filter && nk > 4) {
            const float hx = 0.5f * (bhi[0] - blo[0]), hy = 0.5f * (bhi[1] - blo[1]), hz = 0.5f * (bhi[2] - blo[2]);
            const float reach_h = (fmaxf(fmaxf(hx, hy), hz) + rho_max) * 1.001f;
            int e2 = 0;
            (void)frexpf(reach_h, &e2);                       // reach_h = m 2^e2, m in [0.5, 1)
            const int k = 7 - e2;
            if (reach_h > 0.0f && reach_h < 3.0e38f && k >= -60 && k <= 60) {
                const float4 C = make_float4(blo[0] + hx, blo[1] + hy, blo[2] + hz, ldexpf(1.0f, k));
                uint4 bq;
                float R, inv2;
                ht_setup(qx, qy, qz, C, h, bq, R, inv2);
                float big;
                asm volatile("v_mov_b32 %0, 0x7f800000" : "=v"(big));      // +inf the optimiser cannot see through (nn1_btrack_kernel)
                f32x16 zero;
#pragma unroll
                for (int j = 0; j < 16; j++) zero[j] = 0.0f;
                float m1 = __builtin_inff(), m2 = __builtin_inff();
                uint32_t c1 = 0xFFFFFFFFu;
                const uint32_t row = lane & 31u, sel = (row >> 2) & 1u, elem = 4u * (row >> 3) + (row & 3u);      // MFMA row <-> record of the pair of runs
                auto operand = [&](uint32_t t) {
                    float4 rec = make_float4(__builtin_inff(), 0.f, 0.f, 0.f);
                    if (t + sel < nk) rec = records[(size_t)L.runs[t + sel] * GRID_CHUNK + elem];
                    return rec;
                };
                float4 nrec = operand(0);
                for (uint32_t t = 0; t < nk; t += 2) {
                    const float4 rec = nrec;
                    if (t + 2 < nk) nrec = operand(t + 2);
                    const float tx = (rec.x - C.x) * C.w, ty = (rec.y - C.y) * C.w, tz = (rec.z - C.z) * C.w;
                    const bool fin = fabsf(tx) <= 128.0f && fabsf(ty) <= 128.0f && fabsf(tz) <= 128.0f;          // (false for NaN / inf / beyond the box)
                    const uint4 A = ht_target_operand(tx, ty, tz, fin, h);
                    const f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, bq), zero, 0, 0, 0);
                    float m = big;
#pragma unroll
                    for (int j = 0; j + 1 < 16; j += 2) m = fminf(fminf(m, acc[j]), acc[j + 1]);
                    const float Lb = __builtin_fmaf(m, inv2, R);
                    m2 = __builtin_amdgcn_fmed3f(m1, m2, Lb);
                    const bool better = Lb < m1;
                    m1 = better ? Lb : m1;
                    c1 = better ? t + (h ? 1u : 0u) : c1;
                }
                // the two half-lanes of a query: smallest bound, a run attaining it, smallest bound over all OTHER runs
                const float m1o = __shfl_xor(m1, 32, 64), m2o = __shfl_xor(m2, 32, 64);
                const uint32_t c1o = (uint32_t)__shfl_xor((int)c1, 32, 64);
                const bool take = m1o < m1 || (m1o == m1 && c1o < c1);
                const float M1 = take ? m1o : m1;
                const float M2 = fminf(fminf(m2, m2o), take ? m1 : m1o);
                const uint32_t C1 = take ? c1o : c1;
                const float cur = __uint_as_float((uint32_t)(best >> 32));                 // the seed (or the gate): a bound on the answer
                const bool out = (M1 - 1e-30f) > cur;                                      // no run can beat or tie it
                if (!out && C1 < nk) {
                    const uint32_t p0 = L.runs[C1] * GRID_CHUNK + (h ? 8u : 0u);
#pragma unroll
                    for (int j = 0; j < GRID_CHUNK / 2; j++) {
                        const float4 rec = records[p0 + j];
                        const float dx = qx - rec.x, dy = qy - rec.y, dz = qz - rec.z;
                        const uint32_t d = __float_as_uint((dx * dx + dy * dy) + dz * dz);   // A1, unfused
                        const unsigned long long kk = ((unsigned long long)d << 32) | __float_as_uint(rec.w);
                        if (d < 0x7F7FFFFFu && kk < best) { best = kk; bestp = p0 + (uint32_t)j; }
                    }
                }
                {
                    const unsigned long long ob = ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(best >> 32), 32, 64) << 32) |
                                                  (uint32_t)__shfl_xor((int)(uint32_t)best, 32, 64);
                    const uint32_t op = (uint32_t)__shfl_xor((int)bestp, 32, 64);
                    if (ob < best) { best = ob; bestp = op; }
                }
                const bool proven = out || (M2 - 1e-30f) > __uint_as_float((uint32_t)(best >> 32));
                settled = __all(!in || proven) != 0;
                if (STATS) { st_filt++; st_fall += settled ? 0u : 1u; }
            }
        }
        // ---- EXACT LOOP (few runs, or a pass the filter could not settle): the surviving runs, two per step — 32 lanes fetch 32 records
        // (two coalesced 256-byte loads) into LDS, then lane (n, h) evaluates the 16 records of run t + h for query n (uniform LDS
        // addresses per half: broadcasts); the next pair is in flight meanwhile
        auto fetch = [&](uint32_t t) {
            float4 rec = make_float4(__builtin_inff(), 0.f, 0.f, __uint_as_float(0xFFFFFFFFu));      // beyond the list: never accepted
            const uint32_t k = t + (lane >> 4);
            if (lane < 32 && k < nk) rec = records[(size_t)L.runs[k] * GRID_CHUNK + (lane & 15)];
            return rec;
        };
        float4 nxt = settled ? make_float4(0.f, 0.f, 0.f, 0.f) : fetch(0);
        for (uint32_t t = 0, buf = 0; !settled && t < nk; t += 2, buf ^= 1u) {
            if (lane < 32) L.stage[buf][lane] = nxt;
            if (t + 2 < nk) nxt = fetch(t + 2);
            const uint32_t p0 = (t + (h ? 1u : 0u) < nk ? L.runs[t + (h ? 1u : 0u)] : 0u) * GRID_CHUNK;
            const float4* st = &L.stage[buf][h ? 16 : 0];
#pragma unroll
            for (int j = 0; j < GRID_CHUNK; j++) {
                const float4 rec = st[j];
                const float dx = qx - rec.x, dy = qy - rec.y, dz = qz - rec.z;
                const uint32_t d = __float_as_uint((dx * dx + dy * dy) + dz * dz);   // A1, unfused
                const unsigned long long k = ((unsigned long long)d << 32) | __float_as_uint(rec.w);
                if (d < 0x7F7FFFFFu && k < best) { best = k; bestp = p0 + (uint32_t)j; }
            }
        }
        {
            const unsigned long long ob = ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(best >> 32), 32, 64) << 32) |
                                          (uint32_t)__shfl_xor((int)(uint32_t)best, 32, 64);
            const uint32_t op = (uint32_t)__shfl_xor((int)bestp, 32, 64);
            if (ob < best) { best = ob; bestp = op; }
        }
        if (in && !h) {
            const uint32_t bidx = (uint32_t)(best & 0xFFFFFFFFull);
            const uint32_t bits = (bidx == 0xFFFFFFFFu) ? 0x7F800000u : (uint32_t)(best >> 32);
            keys[i] = ((unsigned long long)bits << 32) | bidx;
            wpos[i] = bidx == 0xFFFFFFFFu ? 0xFFFFFFFFu : bestp;
        }
    }
    deferred = deferred || (member && ((remaining >> n) & 1u));        // a fourth cluster in one group: the walk takes it
    if (min_members > 1u) deferred = deferred || (member && (uint32_t)__popc((uint32_t)__ballot(member)) < min_members);
    // the deferred queries of this group: segment `group` of the list, in query order, and its length (every group writes it)
    const uint32_t dmask = (uint32_t)__ballot(deferred && !h);
    if (deferred && !h) defer_list[(size_t)group * 32 + (uint32_t)__popc(dmask & ((1u << n) - 1u))] = i;
    if (lane == 0) defer_count[group] = (uint32_t)__popc(dmask);
    if (STATS && lane == 0) {
        if (st_cand) atomicAdd(&stats[0], st_cand);
        if (st_rows) atomicAdd(&stats[1], st_rows);
        if (st_sph) atomicAdd(&stats[2], st_sph);
        if (dmask) atomicAdd(&stats[6], (unsigned long long)__popc(dmask));           // [6]: queries handed to the cell walk
        atomicAdd(&stats[7], (unsigned long long)n_pass);                             // [7]: passes
        if (st_ext) atomicAdd(&stats[4], st_ext);                                     // [4], [5]: largest box edge / largest ball of the served passes, um
        if (st_rho) atomicAdd(&stats[5], st_rho);
        if (st_filt) atomicAdd(&stats[8], st_filt);                                   // [8]: passes through the matrix-pipe filter, [9]: of those, not settled by it
        if (st_fall) atomicAdd(&stats[9], st_fall);
        if (st_load) atomicAdd(&stats[10], st_load);                                  // [10]: records loaded by the passes (shared by their queries)
    }
}

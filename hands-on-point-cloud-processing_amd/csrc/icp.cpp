// icp.cpp — host loop of point-to-point ICP (A9), a restatement of
// Registration::ICPpoint2point, Homework9/hw9/src/registration.cpp:862-1011, driving the gfx950 kernels:
//   per iteration: nn1 (correspondences) -> kabsch partial/final (16 f64 sums) -> [one all-reduce when the
//   sources are sharded over GPUs] -> 3x3 SVD on the host (identical on every rank) -> in-place transform.
// The reference's quirks are kept: squared distance vs un-squared max_corr (:936), loss = d2*d2 of the last
// kept pair (:939), `unchanged` never reset (:948-951), det<0 branch (:990-996).
#include "pcr_internal.hpp"
#include "numerics.hpp"

#include <chrono>
#include <cmath>

using namespace pcr;

// the dispatcher's rule (api.cpp nn1_auto_grid) for the searches of a loop
static bool icp_uses_grid(const pcr_ctx* ctx, const pcr_cloud* tgt) { return nn1_auto_grid(ctx, tgt, true, 0); }

// a brute-force loop over a target that takes the matrix-core kernels (>= 8 192 points): the working cloud in the order of the
// target's super-tiles when the loop is long enough to pay for the sort (csrc/grid.hip bt_sort_working_cloud)
static int icp_sort_for_brute(pcr_ctx* ctx, const pcr_cloud* tgt, pcr_cloud** work, uint64_t max_iter)
{
    ctx->work_orig_src = nullptr;
    // (the sort costs ~0.1 ms at 120 k; an iteration of STRACK gains ~0.02 ms from it, one of the sphere forms — targets from 32 768 points — ~0.05 ms:
    // their level-0 / level-1 rows are shared by the 32 queries of a group only when those are neighbours)
    if (max_iter < (tgt->n >= 32768 ? 3u : 8u) || tgt->n < 8192 || tune_get(ctx, "nn1_bf16", 0) == 2) return PCR_OK;
    const int64_t v = tune_get(ctx, "nn1_variant", 0);
    if (v != 0 && v != 6 && v != 7) return PCR_OK;
    return bt_sort_working_cloud(ctx, tgt, work);
}

// ---- synchronous loop: one host round trip per iteration (needed by the host-callback transport; also the
// reference implementation of the loop the pipelined variant below must reproduce bit for bit)
static int icp_sync(pcr_ctx* ctx, const pcr_cloud* src, const pcr_cloud* tgt, const float init_T[16],
                    const pcr_icp_params* prm, float out_T[16], pcr_icp_stats* stats)
{
    const auto t_begin = std::chrono::steady_clock::now();
    pcr_icp_stats st;
    memset(&st, 0, sizeof st);
    const LoopHint hint(ctx, prm->max_iter);
    ctx->keys_seeded = false;            // (seeds a move of an earlier loop left behind describe positions that no longer exist)

    // prof bookkeeping: report only this call's nn1 time
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    prof_flush(ctx);
    const uint64_t nn_l0 = ctx->prof["nn1_brute"].launches + ctx->prof["nn1_grid"].launches;
    const double nn_ms0 = ctx->prof["nn1_brute"].total_ms + ctx->prof["nn1_grid"].total_ms;

    // pairs are kept only if d2 < max_corr (:936), so the grid search need not look farther (tune icp_bounded_search: 2 = off)
    const float gate = tune_get(ctx, "icp_bounded_search", 1) == 1 ? prm->max_corr : __builtin_inff();
    pcr_cloud* work = nullptr;
    int rc = cloud_alloc(ctx, src->n, &work);                                    // :872 (the copy and the initial transform are one launch)
    if (rc) return rc;
    const float R0[9] = { init_T[0], init_T[1], init_T[2], init_T[4], init_T[5], init_T[6], init_T[8], init_T[9], init_T[10] };
    const float t0[3] = { init_T[3], init_T[7], init_T[11] };
    rc = launch_transform_into(ctx, src, work, R0, t0);                          // :874
    if (rc == PCR_OK) rc = icp_uses_grid(ctx, tgt) ? grid_sort_working_cloud(ctx, tgt, &work) : icp_sort_for_brute(ctx, tgt, &work, prm->max_iter);
    float T_total[16] = { R0[0], R0[1], R0[2], t0[0], R0[3], R0[4], R0[5], t0[1],
                          R0[6], R0[7], R0[8], t0[2], 0, 0, 0, 1 };              // :910-913
    float last_loss = 0.0f;                                                      // :915
    uint64_t unchanged = 0;                                                      // :916
    const int nranks = ctx->comm.nranks, rank = ctx->comm.rank;
    const int nred = icp_nred(nranks);   // limbs of the 16 sums + overflow flag + one (kept flag, last d2) slot per rank
    if (nranks > PCR_MAX_RANKS) { pcr_cloud_destroy(ctx, work); return fail(ctx, PCR_ERR_ARG, "too many ranks"); }
    KabschPlan plan;                     // the fixed-point grid of the exact sums: the same on every rank (target + gate)
    if (rc == PCR_OK) rc = kabsch_plan(ctx, tgt, prm->max_corr, &plan);
    bool overflow = false;

    for (uint64_t iter = 0; rc == PCR_OK && iter < prm->max_iter; iter++) {      // :917
        if ((rc = launch_nn1(ctx, tgt, work, true, gate))) break;                      // :925-934
        double* h = ctx->host_out;
        double sums16[16];
        double last_kept, last_d2;
        hipError_t e = hipSuccess;
        if (nranks > 1 || (ctx->comm.rccl && tune_get(ctx, "icp_force_slots", 0) > 0)) {
            // sharded sources: the block rows are reduced on the device into the all-reduce buffer — the limbs of the exact sums
            // + one (kept flag, last d2) slot per rank so that `loss` is that of the globally last kept pair — which is summed
            // over the ranks by the ONE collective of the iteration (RCCL in place on the context stream, or the caller's
            // reducer on the host copy).  Integer limbs: the result does not depend on the number of ranks.
            uint32_t blocks = 0;
            if (work->n && (rc = launch_kabsch_partial(ctx, tgt, work, prm->max_corr, plan, &blocks))) break;   // :936-940,:964-985
            if ((rc = launch_icp_reduce_slots(ctx, blocks, nranks, rank, work->n != 0, src->gidx))) break;
            if (ctx->comm.rccl && (rc = comm_allreduce_f64_device(ctx, ctx->dev_out, nred))) break;
            e = hipMemcpyAsync(h, ctx->dev_out, nred * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) { rc = fail(ctx, PCR_ERR_HIP, "icp d2h", e); break; }
            if (!ctx->comm.rccl && (rc = comm_allreduce_f64(ctx, h, ctx->dev_out, nred))) break;       // host-callback transport
            num::limbs_normalize_row(h);
            num::limbs_to_sums(h, plan.e, sums16);
            overflow = h[55] != 0.0;
            last_kept = -1.0; last_d2 = 0.0;
            const int ls = icp_last_slot(h, nranks);             // the pair with the largest order key: the globally last kept one
            if (ls >= 0) { last_kept = 1.0; last_d2 = h[57 + 2 * ls]; }
        } else {
            if (work->n) {
                if ((rc = launch_kabsch_sums(ctx, tgt, work, prm->max_corr, plan))) break; // :936-940,:964-985
            } else {
                e = hipMemsetAsync(ctx->dev_out, 0, 19 * sizeof(double), ctx->stream);
                if (e != hipSuccess) { rc = fail(ctx, PCR_ERR_HIP, "memset", e); break; }
            }
            e = hipMemcpyAsync(h, ctx->dev_out, 19 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) { rc = fail(ctx, PCR_ERR_HIP, "icp d2h", e); break; }
            memcpy(sums16, h, sizeof sums16);
            last_kept = work->n ? h[16] : -1.0;
            last_d2 = work->n ? h[17] : 0.0;
            overflow = work->n && h[18] != 0.0;
        }
        if (overflow) { rc = fail(ctx, PCR_ERR_STATE, "ICP: a kept source point lies more than 2^20 target extents away from the target"); break; }
        float loss = 0.0f;
        if (last_kept >= 0) { const float d2 = (float)last_d2; loss = d2 * d2; } // :939
        st.last_pairs = (uint64_t)sums16[15];
        st.last_loss = loss;
        if (std::fabs(last_loss - loss) < prm->eps) unchanged++;                 // :948-951
        if (unchanged > 15) { st.converged = 1; break; }                         // :954-958
        last_loss = loss;                                                        // :961
        float Rd[9], td[3];
        if (kabsch_solve(sums16, Rd, td) != PCR_OK) { st.empty_pairs = 1; break; }    // :979-998
        const float T_delta[16] = { Rd[0], Rd[1], Rd[2], td[0], Rd[3], Rd[4], Rd[5], td[1],
                                    Rd[6], Rd[7], Rd[8], td[2], 0, 0, 0, 1 };
        mat4_mul_f32(T_delta, T_total, T_total);                                 // :1000-1002
        rc = launch_transform(ctx, work, Rd, td);                                // :1003
        st.iters_run++;
    }
    hipStreamSynchronize(ctx->stream);
    cloud_release(ctx, work);             // (the loop's own working copy: parked for the next call's clone)
    if (rc) return rc;
    memcpy(out_T, T_total, sizeof T_total);                                      // :1008-1009
    prof_flush(ctx);
    st.nn_launches = ctx->prof["nn1_brute"].launches + ctx->prof["nn1_grid"].launches - nn_l0;
    st.ms_nn = ctx->prof["nn1_brute"].total_ms + ctx->prof["nn1_grid"].total_ms - nn_ms0;
    st.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    if (stats) *stats = st;
    return PCR_OK;
}

// ---- pipelined loop: the state machine, the Kabsch solve and the pose composition live on the GPU (IcpState,
// kabsch.hip); iterations are enqueued back to back on the context stream in chunks, the host only looks at a copy
// of the state two chunks behind to learn when to stop enqueuing.  With RCCL the per-iteration all-reduce is
// enqueued on the same stream, so even the sharded loop needs no host round trip.
static int icp_pipelined(pcr_ctx* ctx, const pcr_cloud* src, const pcr_cloud* tgt, const float init_T[16],
                         const pcr_icp_params* prm, float out_T[16], pcr_icp_stats* stats)
{
    const auto t_begin = std::chrono::steady_clock::now();
    pcr_icp_stats st;
    memset(&st, 0, sizeof st);
    const LoopHint hint(ctx, prm->max_iter);
    ctx->keys_seeded = false;            // (seeds a move of an earlier loop left behind describe positions that no longer exist)
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    prof_flush(ctx);
    const uint64_t nn_l0 = ctx->prof["nn1_brute"].launches + ctx->prof["nn1_grid"].launches;
    const double nn_ms0 = ctx->prof["nn1_brute"].total_ms + ctx->prof["nn1_grid"].total_ms;

    constexpr int RING = 4;
    if (!ctx->icp_state_dev) {
        PCR_HIP(ctx, hipMalloc((void**)&ctx->icp_state_dev, 2 * sizeof(IcpState)));      // [1]: the other buffer of the fused solve + move
        PCR_HIP(ctx, hipHostMalloc((void**)&ctx->icp_state_host, (RING + 1) * sizeof(IcpState), hipHostMallocDefault));
        for (int k = 0; k < RING; k++) PCR_HIP(ctx, hipEventCreateWithFlags(&ctx->icp_events[k], hipEventDisableTiming));
    }
    IcpState* dev = ctx->icp_state_dev;
    IcpState* host = ctx->icp_state_host;      // [0..RING) ring of snapshots, [RING] upload / final download

    // pairs are kept only if d2 < max_corr (:936), so the grid search need not look farther (tune icp_bounded_search: 2 = off)
    const float gate = tune_get(ctx, "icp_bounded_search", 1) == 1 ? prm->max_corr : __builtin_inff();
    pcr_cloud* work = nullptr;
    int rc = cloud_alloc(ctx, src->n, &work);                                    // :872 (the copy and the initial transform are one launch)
    if (rc) return rc;
    const float R0[9] = { init_T[0], init_T[1], init_T[2], init_T[4], init_T[5], init_T[6], init_T[8], init_T[9], init_T[10] };
    const float t0[3] = { init_T[3], init_T[7], init_T[11] };
    rc = launch_transform_into(ctx, src, work, R0, t0);                          // :874
    if (rc == PCR_OK) rc = icp_uses_grid(ctx, tgt) ? grid_sort_working_cloud(ctx, tgt, &work) : icp_sort_for_brute(ctx, tgt, &work, prm->max_iter);
    IcpState& h0 = host[RING];
    memset(&h0, 0, sizeof h0);
    const float T0[16] = { R0[0], R0[1], R0[2], t0[0], R0[3], R0[4], R0[5], t0[1], R0[6], R0[7], R0[8], t0[2], 0, 0, 0, 1 };
    memcpy(h0.T_total, T0, sizeof T0);                                           // :910-913
    h0.eps = prm->eps;
    h0.max_iter = prm->max_iter;
    if (prm->max_iter == 0) h0.stop = 1;
    hipError_t e = hipMemcpyAsync(dev, &h0, sizeof h0, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) rc = fail(ctx, PCR_ERR_HIP, "icp state upload", e);

    const int nranks = ctx->comm.nranks, rank = ctx->comm.rank;
    const int nred = icp_nred(nranks);
    if (nranks > PCR_MAX_RANKS) { pcr_cloud_destroy(ctx, work); return fail(ctx, PCR_ERR_ARG, "too many ranks"); }   // dev_out / host_out hold 128 f64
    KabschPlan plan;
    if (rc == PCR_OK) rc = kabsch_plan(ctx, tgt, prm->max_corr, &plan);
    int64_t chunk = tune_get(ctx, "icp_chunk", 4);
    if (chunk < 1) chunk = 1;
    const bool force_slots = tune_get(ctx, "icp_force_slots", 0) > 0;
    // small clouds on one rank: the solve and the move share a launch (kabsch.hip icp_update_move_kernel; tune icp_fused_move: 2 = off);
    // the state then alternates between dev[0] and dev[1]: iteration k reads dev[k & 1] and writes dev[(k + 1) & 1]
    // (round 3: up to 262 144 points — with the seed of the next search folded in as well, the chain of a 120 k iteration is search ->
    // sums -> solve + move + seed: three launches instead of five)
    const int64_t fused_max = tune_get(ctx, "icp_fused_max", 262144);
    const bool fused = nranks == 1 && !force_slots && work->n > 0 && work->n <= (size_t)fused_max && tune_get(ctx, "icp_fused_move", 1) == 1;
    // exhaustive searches of a loop seed themselves from the previous correspondences: the move writes those seeds (tune icp_seed_in_move: 2 = off)
    const pcr_cloud* seed_tgt = (!icp_uses_grid(ctx, tgt) && tune_get(ctx, "icp_seed_in_move", 1) == 1 && tune_get(ctx, "nn1_warm_start", 1) == 1) ? tgt : nullptr;
    // (Measured and dropped, round 4: the Kabsch sums taken by the search kernel itself — STRACK3 ends with every query's final key in one wave, so each
    // wave added its 32 pairs' limbs (wave sums, a row per workgroup, f64 atomics into 64 rows; same bits) and the streaming pass + its launch gap
    // (7 + 4.5 us) went away: the search grew from 0.034 to 0.049 ms — 41 limbs x 6 f64 shuffle-adds per wave, four waves per SIMD ending together on
    // the LDS crossbar and the half-rate f64 pipe — and an iteration from 0.0707 to 0.0755 ms.  A second form without any cross-lane traffic (lane l takes
    // moment l of the wave's 32 pairs from LDS) still cost the search 7.5 us — the block waits for its slowest wave, every wave ends 2 us later — for 0-1.3 us
    // per iteration.  Also measured: the 3 x 3 solve is 4.6 of the 17.5 us of the solve + move.)
    uint64_t enq = 0, chunks = 0;
    bool stopped = false;
    while (rc == PCR_OK && !stopped && enq < prm->max_iter) {
        for (int64_t c = 0; rc == PCR_OK && c < chunk && enq < prm->max_iter; c++, enq++) {   // :917
            IcpState* cur = fused ? dev + (enq & 1) : dev;                                        // the state this iteration starts from
            ctx->stop_flag_dev = &cur->stop;     // correspondence kernels no-op once stop or stop_after_transform is set
            if ((rc = launch_nn1(ctx, tgt, work, true, gate))) break;                             // :925-934
            uint32_t blocks = 0;
            if (work->n && (rc = launch_kabsch_partial(ctx, tgt, work, prm->max_corr, plan, &blocks))) break;   // :936-940,:964-985
            if (fused) {
                if ((rc = launch_icp_update_move(ctx, blocks, cur, dev + ((enq + 1) & 1), plan, work, seed_tgt))) break;   // :948-1003
                continue;
            }
            if (nranks == 1 && work->n && !force_slots) {
                if ((rc = launch_icp_update(ctx, blocks, dev, plan))) break;                // :948-1002
            } else {
                if ((rc = launch_icp_reduce_slots(ctx, blocks, nranks, rank, work->n != 0, src->gidx))) break;
                if ((rc = comm_allreduce_f64_device(ctx, ctx->dev_out, nred))) break;       // the ONE collective
                if ((rc = launch_icp_update_from_sums(ctx, nranks, dev, plan))) break;
            }
            if ((rc = launch_transform_state(ctx, work, dev, seed_tgt))) break;             // :1003
        }
        ctx->stop_flag_dev = nullptr;
        if (rc) break;
        const int slot = (int)(chunks % RING);
        e = hipMemcpyAsync(&host[slot], fused ? dev + (enq & 1) : dev, sizeof(IcpState), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipEventRecord(ctx->icp_events[slot], ctx->stream);
        if (e != hipSuccess) { rc = fail(ctx, PCR_ERR_HIP, "icp snapshot", e); break; }
        chunks++;
        if (chunks >= 2) {
            // look two chunks back: deterministic (the same on every rank), and the GPU never runs dry
            const int old = (int)((chunks - 2) % RING);
            e = hipEventSynchronize(ctx->icp_events[old]);
            if (e != hipSuccess) { rc = fail(ctx, PCR_ERR_HIP, "icp snapshot wait", e); break; }
            if (host[old].stop || host[old].stop_after_transform) stopped = true;
        }
    }
    ctx->stop_flag_dev = nullptr;
    if (rc == PCR_OK) {
        e = hipMemcpyAsync(&host[RING], fused ? dev + (enq & 1) : dev, sizeof(IcpState), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = fail(ctx, PCR_ERR_HIP, "icp state download", e);
    } else {
        hipStreamSynchronize(ctx->stream);
    }
    cloud_release(ctx, work);             // (synchronised above; the loop's own working copy: parked for the next call's clone)
    if (rc) return rc;
    const IcpState& f = host[RING];
    if (f.overflow) return fail(ctx, PCR_ERR_STATE, "ICP: a kept source point lies more than 2^20 target extents away from the target");
    memcpy(out_T, f.T_total, sizeof f.T_total);                                  // :1008-1009
    st.iters_run = f.iters_run;
    st.converged = f.converged;
    st.empty_pairs = f.empty;
    st.last_pairs = f.last_pairs;
    st.last_loss = f.loss;
    prof_flush(ctx);
    st.nn_launches = ctx->prof["nn1_brute"].launches + ctx->prof["nn1_grid"].launches - nn_l0;
    st.ms_nn = ctx->prof["nn1_brute"].total_ms + ctx->prof["nn1_grid"].total_ms - nn_ms0;
    st.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    if (stats) *stats = st;
    return PCR_OK;
}

extern "C" int pcr_icp_p2p_f32(pcr_ctx* ctx, const pcr_cloud* src, const pcr_cloud* tgt, const float init_T[16],
                               const pcr_icp_params* prm, float out_T[16], pcr_icp_stats* stats)
{
    if (!ctx || !src || !tgt || !init_T || !prm || !out_T) return fail(ctx, PCR_ERR_ARG, "pcr_icp_p2p_f32");
    PCR_HIP(ctx, hipSetDevice(ctx->device));
    // tune "icp_pipeline": 1 = device-resident pipelined loop, -1 = synchronous loop (one host round trip per iteration),
    // 0 = auto = pipelined.  Measured on MI355X in round 2 (brute force, 120 k target, 20 iterations): 1.237 vs 1.242 ms per
    // iteration with all 120 k sources, 0.211 vs 0.217 ms with a 1/8 shard (what one rank of an 8-GPU strong-scaling run
    // holds), tens of microseconds per iteration with the grid search — the device-resident loop is never slower, and with
    // RCCL it keeps the per-iteration all-reduce on the stream.  The host-callback transport reduces on the host and
    // therefore always runs synchronously.  Both loops give bit-identical results.
    const bool callback = ctx->comm.nranks > 1 && ctx->comm.cb != nullptr;
    const int64_t mode = tune_get(ctx, "icp_pipeline", 0);
    const bool pipelined = !callback && mode >= 0;
    return pipelined ? icp_pipelined(ctx, src, tgt, init_T, prm, out_T, stats) : icp_sync(ctx, src, tgt, init_T, prm, out_T, stats);
}

// icp.cpp — host loop of point-to-point ICP (A9), a restatement of
// Registration::ICPpoint2point, Homework9/hw9/src/registration.cpp:862-1011, driving the gfx950 kernels:
//   per iteration: nn1 (correspondences) -> kabsch partial/final (16 f64 sums) -> [one all-reduce when the
//   sources are sharded over GPUs] -> 3x3 SVD on the host (identical on every rank) -> in-place transform.
// The reference's quirks are kept: squared distance vs un-squared max_corr (:936), loss = d2*d2 of the last
// kept pair (:939), `unchanged` never reset (:948-951), det<0 branch (:990-996).
#include "pcr_internal.hpp"

#include <chrono>
#include <cmath>

using namespace pcr;

extern "C" int pcr_icp_p2p_f32(pcr_ctx* ctx, const pcr_cloud* src, const pcr_cloud* tgt, const float init_T[16],
                               const pcr_icp_params* prm, float out_T[16], pcr_icp_stats* stats)
{
    if (!ctx || !src || !tgt || !init_T || !prm || !out_T) return fail(ctx, PCR_ERR_ARG, "pcr_icp_p2p_f32");
    PCR_HIP(ctx, hipSetDevice(ctx->device));
    const auto t_begin = std::chrono::steady_clock::now();
    pcr_icp_stats st;
    memset(&st, 0, sizeof st);

    // prof bookkeeping: report only this call's nn1 time
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    prof_flush(ctx);
    const uint64_t nn_l0 = ctx->prof["nn1_brute"].launches;
    const double nn_ms0 = ctx->prof["nn1_brute"].total_ms;

    pcr_cloud* work = nullptr;
    int rc = pcr_cloud_clone(ctx, src, &work);                                   // :872
    if (rc) return rc;
    const float R0[9] = { init_T[0], init_T[1], init_T[2], init_T[4], init_T[5], init_T[6], init_T[8], init_T[9], init_T[10] };
    const float t0[3] = { init_T[3], init_T[7], init_T[11] };
    rc = launch_transform(ctx, work, R0, t0);                                    // :874
    float T_total[16] = { R0[0], R0[1], R0[2], t0[0], R0[3], R0[4], R0[5], t0[1],
                          R0[6], R0[7], R0[8], t0[2], 0, 0, 0, 1 };              // :910-913
    float last_loss = 0.0f;                                                      // :915
    uint64_t unchanged = 0;                                                      // :916
    const int nranks = ctx->comm.nranks, rank = ctx->comm.rank;
    const int nred = 16 + 2 * nranks;   // sums + one (last_kept flag, last d2) slot per rank
    if (nred > 64) { pcr_cloud_destroy(ctx, work); return fail(ctx, PCR_ERR_ARG, "too many ranks"); }

    for (uint64_t iter = 0; rc == PCR_OK && iter < prm->max_iter; iter++) {      // :917
        if ((rc = launch_nn1_brute(ctx, tgt, work))) break;                      // :925-934
        double* h = ctx->host_out;
        if (work->n) {
            if ((rc = launch_kabsch_sums(ctx, tgt, work, prm->max_corr))) break; // :936-940,:964-985
        } else {
            hipError_t e = hipMemsetAsync(ctx->dev_out, 0, 18 * sizeof(double), ctx->stream);
            if (e != hipSuccess) { rc = fail(ctx, PCR_ERR_HIP, "memset", e); break; }
        }
        hipError_t e = hipMemcpyAsync(h, ctx->dev_out, 18 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { rc = fail(ctx, PCR_ERR_HIP, "icp d2h", e); break; }
        double last_kept = work->n ? h[16] : -1.0, last_d2 = work->n ? h[17] : 0.0;
        if (nranks > 1) {
            // the ONE collective of the iteration: sum of the 16 moments; the per-rank (flag, d2) slots ride
            // along so that `loss` is that of the globally last kept pair (highest rank that kept any)
            for (int r = 0; r < nranks; r++) { h[16 + 2 * r] = 0.0; h[17 + 2 * r] = 0.0; }
            h[16 + 2 * rank] = last_kept >= 0 ? 1.0 : 0.0;
            h[17 + 2 * rank] = last_d2;
            if ((rc = comm_allreduce_f64(ctx, h, ctx->dev_out, nred))) break;
            last_kept = -1.0;
            for (int r = 0; r < nranks; r++)
                if (h[16 + 2 * r] > 0.5) { last_kept = 1.0; last_d2 = h[17 + 2 * r]; }
        }
        float loss = 0.0f;
        if (last_kept >= 0) { const float d2 = (float)last_d2; loss = d2 * d2; } // :939
        st.last_pairs = (uint64_t)h[15];
        st.last_loss = loss;
        if (std::fabs(last_loss - loss) < prm->eps) unchanged++;                 // :948-951
        if (unchanged > 15) { st.converged = 1; break; }                         // :954-958
        last_loss = loss;                                                        // :961
        float Rd[9], td[3];
        if (kabsch_solve(h, Rd, td) != PCR_OK) { st.empty_pairs = 1; break; }    // :979-998
        const float T_delta[16] = { Rd[0], Rd[1], Rd[2], td[0], Rd[3], Rd[4], Rd[5], td[1],
                                    Rd[6], Rd[7], Rd[8], td[2], 0, 0, 0, 1 };
        mat4_mul_f32(T_delta, T_total, T_total);                                 // :1000-1002
        rc = launch_transform(ctx, work, Rd, td);                                // :1003
        st.iters_run++;
    }
    hipStreamSynchronize(ctx->stream);
    pcr_cloud_destroy(ctx, work);
    if (rc) return rc;
    memcpy(out_T, T_total, sizeof T_total);                                      // :1008-1009
    prof_flush(ctx);
    st.nn_launches = ctx->prof["nn1_brute"].launches - nn_l0;
    st.ms_nn = ctx->prof["nn1_brute"].total_ms - nn_ms0;
    st.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    if (stats) *stats = st;
    return PCR_OK;
}

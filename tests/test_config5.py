"""BASELINE config 5 at its full size on ONE GPU: a 10 M x 10 M synthetic pair (the 8-GPU run shards exactly this pair's
sources).  No O(n^2) oracle pass is possible: the exact grid search is checked on a sample against the brute-force oracle
and through size-independent properties; the sharded reduction through linearity of the moments."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
N = 10_000_000


def test_c5_10M_pair_exact_search_shard_linearity_and_pose(pcr, orc, synth):
    src, tgt = synth.kitti_like_pair(N)
    ctx = pcr.Context(0)
    try:
        ctx.tune("nn_method", 2)
        cs, ct = ctx.cloud(src), ctx.cloud(tgt)
        idx, d2 = ctx.nn1(ct, cs)
        # (1) a sample against the exhaustive oracle: indices and d2 bits
        sel = np.arange(0, N, N // 96)[:96]
        oi, od = orc.nn1_f32(tgt, np.ascontiguousarray(src[:, sel]))
        assert np.array_equal(idx[sel], oi) and np.array_equal(d2[sel].view(np.uint32), od.view(np.uint32))
        # (2) every reported d2 is the A1 arithmetic of its reported pair; no query is left without an answer
        assert idx.max() < N and np.isfinite(d2).all()
        some = np.random.default_rng(0).integers(0, N, 200000)
        dx, dy, dz = (src[c, some] - tgt[c, idx[some]] for c in range(3))
        assert np.array_equal(((dx * dx + dy * dy) + dz * dz).astype(np.float32).view(np.uint32), d2[some].view(np.uint32))
        # (3) optimality on a second sample: no target of a random 2 000-point subset is strictly closer
        probe = np.random.default_rng(1).integers(0, N, 2000)
        qs = np.random.default_rng(2).integers(0, N, 500)
        for q in qs[:50]:
            ex, ey, ez = (src[c, q] - tgt[c, probe] for c in range(3))
            assert (((ex * ex + ey * ey) + ez * ez).astype(np.float32) >= d2[q]).all()
        # (4) sharding: the moments of two half-shards add up to the moments of the whole (what the all-reduce relies on)
        full, _, _ = ctx.kabsch_sums(ct, cs, 1.0)
        b, e = pcr.shard_range(N, 2, 0)
        h0 = ctx.cloud(np.ascontiguousarray(src[:, b:e]))
        ctx.nn1_async(ct, h0); s0, _, _ = ctx.kabsch_sums(ct, h0, 1.0)
        h0.free()
        b1, e1 = pcr.shard_range(N, 2, 1)
        h1 = ctx.cloud(np.ascontiguousarray(src[:, b1:e1]))
        ctx.nn1_async(ct, h1); s1, _, _ = ctx.kabsch_sums(ct, h1, 1.0)
        h1.free()
        assert e == b1 and e1 == N
        assert s0[15] + s1[15] == full[15]                           # kept-pair counts: exact
        assert np.allclose(s0 + s1, full, rtol=1e-11, atol=1e-6)
        # (5) ICP from the identity walks towards the planted pose (dense clouds: small steps; 20 iterations reach 4e-2)
        e0 = np.linalg.norm(np.eye(4) - synth.gt_pose())
        T, st = ctx.icp_point2point(cs, ct, max_corr=1.0, max_iter=4, eps=0.0)
        e4 = np.linalg.norm(T - synth.gt_pose())
        assert st["iters_run"] == 4 and st["last_pairs"] > 0.99 * N and e4 < 0.6 * e0
        T2, st2 = ctx.icp_point2point(cs, ct, init_T=T, max_corr=1.0, max_iter=4, eps=0.0)
        assert np.linalg.norm(T2 - synth.gt_pose()) < 0.7 * e4
        # (6) the tile search (default at this size from the third search of a loop on) against the cell walk alone: same pose bits,
        # same pairs, same loss — through the misaligned first iterations (most queries deferred) and near the converged pose
        assert ctx.mfma_check()["last_nn1_kernel"] == "grid-stile"
        ctx.tune("grid_tile", 2)
        for init, it in ((None, 4), (T, 4), (None, 9)):
            Tw, sw = ctx.icp_point2point(cs, ct, init_T=init, max_corr=1.0, max_iter=it, eps=0.0)
            assert ctx.mfma_check()["last_nn1_kernel"] == "grid"
            ctx.tune("grid_tile", 0)
            Tt, stt = ctx.icp_point2point(cs, ct, init_T=init, max_corr=1.0, max_iter=it, eps=0.0)
            ctx.tune("grid_tile", 2)
            assert np.array_equal(Tw.view(np.uint32), Tt.view(np.uint32)), it
            assert sw["last_pairs"] == stt["last_pairs"] and np.float32(sw["last_loss"]).view(np.uint32) == np.float32(stt["last_loss"]).view(np.uint32)
        ctx.tune("grid_tile", 0)
        # (7) searches INSIDE a tile-search loop against the reference's own nanoflann (f32, leaf 2: registration.cpp:903-905, compiled from
        # /root/reference into oracle/_ref; a tree over the full 10 M target): a caller-stepped loop over the sorted working cloud — the
        # kernels pcr_icp_p2p_f32 launches — and 131 072 sampled queries of its 3rd search (most queries still deferred to the list walk)
        # and of its 6th (the tile search serves nearly all), index and d2 bits (tie-set rule of SURVEY 7.2)
        work = cs.clone()
        orig = ctx.sort_for_target(ct, work)
        assert np.array_equal(np.sort(orig), np.arange(N, dtype=np.uint32))
        rng = np.random.default_rng(7)
        for it in range(6):
            ctx.nn1_loop(ct, work, 1.0)
            kern = ctx.mfma_check()["last_nn1_kernel"]
            assert it < 2 or kern == "grid-stile", (it, kern)
            if it in (2, 5):
                idx, d2 = ctx.nn1_fetch(N)
                cur = work.numpy()
                sel = np.sort(rng.choice(N, 131072 if orc.have_ref() else 64, replace=False))
                q = np.ascontiguousarray(cur[:, sel])
                if orc.have_ref():
                    ridx, rd2, _, _ = orc.ref_nano_nn1_f32(tgt, q, leaf=2, threads=16)
                else:
                    ridx, rd2 = orc.nn1_f32_mt(tgt, q, threads=16)
                inside = rd2 < np.float32(1.0)                               # the loop's searches are bounded by the gate (:936)
                assert inside.mean() > 0.9
                gi, gd = idx[sel], d2[sel]
                assert (gi[~inside] == 0xFFFFFFFF).all() and np.isinf(gd[~inside]).all()
                assert np.array_equal(gd[inside].view(np.uint32), rd2[inside].view(np.uint32)), it
                diff = np.flatnonzero(inside & (gi != ridx))
                assert (gi[diff] < ridx[diff]).all()                         # a genuine tie: the product returns the lowest index
                for j in (gi[diff], ridx[diff]):
                    ex, ey, ez = (q[c, diff] - tgt[c, j] for c in range(3))
                    assert np.array_equal(((ex * ex + ey * ey) + ez * ez).astype(np.float32).view(np.uint32), gd[diff].view(np.uint32))
            sums, last, _ = ctx.kabsch_sums(ct, work, 1.0)
            rc, R, t = pcr.kabsch_solve(sums)
            assert rc == 0
            Td = np.eye(4, dtype=np.float32); Td[:3, :3], Td[:3, 3] = R, t
            ctx.transform(work, Td)
        work.free()
    finally:
        ctx.close()

"""Randomised GPU-vs-oracle sweeps (-m gpu) over the widened rows: continuous, lattice (ties / duplicates), clustered and
extreme-magnitude clouds, far-away queries, ragged sizes.  Integer / index outputs must be bit-exact."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
# soak runs: PCR_SWEEP_SCALE multiplies the number of trials, PCR_SWEEP_SEED shifts every generator seed
SCALE = int(os.environ.get("PCR_SWEEP_SCALE", "1"))
SEED = int(os.environ.get("PCR_SWEEP_SEED", "0"))


def random_cloud32(rng, n, kind):
    if kind == 0:
        a = rng.normal(0, 10, (3, n))
    elif kind == 1:
        a = rng.integers(0, 6, (3, n)).astype(np.float64) * 0.5
    elif kind == 2:
        c = rng.normal(0, 30, (3, max(n // 8, 1)))
        a = c[:, rng.integers(0, c.shape[1], n)] + rng.normal(0, 0.01, (3, n))
    else:
        a = rng.normal(0, 1, (3, n)) * 10.0 ** rng.integers(-12, 12)
    return np.ascontiguousarray(a.astype(np.float32))


def test_cloud_knn_randomised(pcr, orc):
    rng = np.random.default_rng(101 + SEED)
    ctx = pcr.Context(0)
    try:
        for trial in range(80 * SCALE):
            kind = trial % 4
            n, m = int(rng.integers(1, 4000)), int(rng.integers(1, 300))
            db = random_cloud32(rng, n, kind)
            q = random_cloud32(rng, m, kind if trial % 5 else (kind + 1) % 4)     # sometimes a different scale: far queries
            if trial % 3 == 0:
                q[:, : min(n, m) // 2] = db[:, : min(n, m) // 2]
            k = int(rng.integers(1, 33))
            ext = float(np.abs(db).max()) + 1e-30
            radius = -1.0 if trial % 2 else float(rng.uniform(0.01, 1.0) * ext)
            cdb, cq = ctx.cloud(db), ctx.cloud(q)
            idx, s, found = ctx.cloud_knn(cdb, cq, k, radius)
            oi, os_, of = orc.knn_sq_f32pts(db, q, k, radius)
            assert np.array_equal(idx, oi) and np.array_equal(s.view(np.uint64), os_.view(np.uint64)) and np.array_equal(found, of), (trial, kind, n, m, k, radius)
            if trial % 4 == 0:                                                     # hw2 contract + self-query path
                idx2, d2, _ = ctx.cloud_knn(cdb, cdb, k, squared=False)
                db64 = np.ascontiguousarray(db.T.astype(np.float64))
                hi, hd = orc.knn_f64(db64, db64, k)
                assert np.array_equal(idx2, hi) and np.array_equal(d2.view(np.uint64), hd.view(np.uint64)), (trial, "self")
            cq.free(); cdb.free()
    finally:
        ctx.close()


def test_voxel_filter_randomised(pcr, orc):
    rng = np.random.default_rng(202 + SEED)
    ctx = pcr.Context(0)
    try:
        for trial in range(60 * SCALE):
            kind = trial % 3                                                       # (extreme magnitudes overflow the voxel index like the reference's int cast)
            n = int(rng.integers(1, 6000))
            pts = random_cloud32(rng, n, kind)
            ext = float(np.ptp(pts, axis=1).max()) + 1e-3
            leaf = float(rng.uniform(0.01, 0.5) * ext)
            got = ctx.voxel_filter(ctx.cloud(pts), leaf).numpy()
            want = orc.voxel_filter_f32(pts, leaf)
            assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32)), (trial, kind, n, leaf)
    finally:
        ctx.close()


def test_iss_randomised(pcr, orc):
    rng = np.random.default_rng(303 + SEED)
    ctx = pcr.Context(0)
    try:
        for trial in range(40 * SCALE):
            kind = trial % 4
            n = int(rng.integers(1, 1500))
            pts = random_cloud32(rng, n, kind)
            ext = float(np.ptp(pts, axis=1).max()) + 1e-30
            r_local = float(np.float32(rng.uniform(0.02, 0.4) * ext))
            r_nms = float(np.float32(r_local * rng.uniform(0.3, 1.2)))
            weighted = bool(trial % 2)
            min_nb = int(rng.integers(1, 8))
            idx, l3, cnt = ctx.iss_keypoints(ctx.cloud(pts), r_local, r_nms, 0.9, 0.9, min_nb, weighted)
            okey, ol3 = orc.iss_f32(pts, r_local, r_nms, 0.9, 0.9, min_nb, weighted)
            row, _, _ = orc.radius_f32(np.ascontiguousarray(pts.T), np.ascontiguousarray(pts.T), r_local)
            assert np.array_equal(cnt, np.diff(row).astype(np.uint32)), (trial, kind, n, r_local)      # neighbourhood sizes: exact
            flip = (l3 == -1) != (ol3 == -1)
            both = (l3 != -1) & (ol3 != -1)
            assert flip.sum() <= 1 and np.allclose(l3[both], ol3[both], rtol=1e-5, atol=0), (trial, kind, n)
            if not flip.any() and np.array_equal(l3.view(np.uint32), ol3.view(np.uint32)):
                assert np.array_equal(idx, np.flatnonzero(okey)), (trial, kind, n)
    finally:
        ctx.close()


def test_plane_count_and_ground_seeds_randomised(pcr, orc):
    rng = np.random.default_rng(404 + SEED)
    ctx = pcr.Context(0)
    try:
        for trial in range(40 * SCALE):
            kind = trial % 3
            n = int(rng.integers(1, 20000))
            pts = random_cloud32(rng, n, kind)
            pts[2] = pts[2] * 0.1 - 1.7                                             # around the sensor-height ground
            c = ctx.cloud(pts)
            planes = rng.normal(0, 1, (int(rng.integers(1, 100)), 4))
            planes[:, :3] /= np.linalg.norm(planes[:, :3], axis=1, keepdims=True)
            thr = float(rng.uniform(0.01, 2.0))
            assert np.array_equal(ctx.plane_count(c, planes, thr), orc.plane_count(pts, planes, thr)), (trial, "planes")
            lpr = int(rng.integers(1, 3 * n + 2))
            ts = float(rng.uniform(0.0, 1.0))
            mask, ub = ctx.ground_seeds(c, lpr, ts)
            omask, oub = orc.ground_seeds_f64(pts, lpr, ts)
            assert np.array_equal(mask, omask.astype(bool)) and (ub == oub or (np.isnan(ub) and np.isnan(oub))), (trial, "seeds", n, lpr)
    finally:
        ctx.close()


def test_db64_radius_and_knn_grid_routes_randomised(pcr, orc):
    """the drop-in f64 entry points with the grid routes forced, on random f32-representable clouds (far queries, ties,
    clusters): CSR rows / neighbour lists and distance bits must equal the exhaustive kernels'."""
    rng = np.random.default_rng(505 + SEED)
    ctx = pcr.Context(0)
    try:
        for trial in range(30 * SCALE):
            kind = trial % 3
            n, m = int(rng.integers(4096, 9000)), int(rng.integers(1, 400))     # >= 4096: the f32 twin exists
            db32 = random_cloud32(rng, n, kind)
            q32 = random_cloud32(rng, m, kind if trial % 4 else (kind + 1) % 3)
            if trial % 3 == 0:
                q32[:, : m // 2] = db32[:, : m // 2]
            db, q = np.ascontiguousarray(db32.T.astype(np.float64)), np.ascontiguousarray(q32.T.astype(np.float64))
            h = ctx.db64(db)
            ext = float(np.ptp(db, axis=0).max()) + 1e-12
            r = float(rng.uniform(0.005, 0.2) * ext)
            k = int(rng.integers(1, 33))
            res = {}
            for meth in (1, 2):
                ctx.tune("radius_method", meth); ctx.tune("knn_method", meth)
                res[meth] = (h.radius(q, r), h.knn(q, k), h.knn(q, k, squared=True))
            for a, b in zip(res[1], res[2]):
                for x, y in zip(a, b):
                    assert np.array_equal(x.view(np.uint64) if x.dtype == np.float64 else x, y.view(np.uint64) if y.dtype == np.float64 else y), (trial, kind, n, m, r, k)
            h.free()
        ctx.tune("radius_method", 0); ctx.tune("knn_method", 0)
    finally:
        ctx.close()


def test_descriptor_matching_randomised(pcr, orc):
    rng = np.random.default_rng(606 + SEED)
    ctx = pcr.Context(0)
    try:
        for trial in range(30 * SCALE):
            dim = int(rng.choice([1, 2, 3, 4, 5, 8, 16, 33, 34, 64, 100]))
            ns, nt = int(rng.integers(1, 900)), int(rng.integers(1, 900))
            a = rng.gamma(0.7, 1.0, (ns, dim)).astype(np.float32)
            b = rng.gamma(0.7, 1.0, (nt, dim)).astype(np.float32)
            if trial % 2 == 0:
                kk = min(ns, nt) // 2
                b[:kk] = a[:kk]                                                 # exact matches and, below, duplicates
                a[-(kk // 2 + 1):] = a[: kk // 2 + 1]
            rate = float(rng.choice([0.0, 0.3, 0.5, 0.95]))
            for fn_g, fn_o in ((ctx.match_union, orc.match_union_f32), (ctx.match_inter, orc.match_inter_f32)):
                pg, dg = fn_g(a, b, rate)
                po, do = fn_o(a, b, rate)
                assert np.array_equal(pg, po) and np.array_equal(dg.view(np.uint32), do.view(np.uint32)), (trial, dim, ns, nt, rate, fn_o.__name__)
    finally:
        ctx.close()


def test_icp_variants_randomised(pcr, orc, synth):
    """point-to-point and point-to-plane loops on small random pairs: pose within 1e-5 of the oracle, identical statistics."""
    rng = np.random.default_rng(707 + SEED)
    ctx = pcr.Context(0)
    try:
        for trial in range(12 * SCALE):
            n = int(rng.integers(200, 2500))
            src, tgt = synth.kitti_like_pair(n, seed_target=int(rng.integers(1, 1 << 30)), seed_pair=int(rng.integers(1, 1 << 30)))
            nrm = tgt / np.maximum(np.linalg.norm(tgt, axis=0, keepdims=True), 1e-6)
            nrm = np.ascontiguousarray(nrm.astype(np.float32))
            mc = float(rng.choice([0.05, 0.5, 1.0, 4.0]))
            it = int(rng.integers(1, 25))
            eps = float(rng.choice([0.0, 1e-8, 1e-3]))
            cs, ct, cn = ctx.cloud(src), ctx.cloud(tgt), ctx.cloud(nrm)
            ctx.tune("nn_method", 1 + trial % 2)
            T, st = ctx.icp_point2point(cs, ct, max_corr=mc, max_iter=it, eps=eps)
            oT, ost = orc.icp_p2p_f32(src, tgt, max_corr=mc, max_iter=it, eps=eps)
            assert np.linalg.norm(T.astype(np.float64) - oT.astype(np.float64)) <= 1e-5, (trial, n, mc, it, eps)
            assert (st["iters_run"], st["converged"], st["empty_pairs"], st["last_pairs"]) == (ost["iters_run"], ost["converged"], ost["empty_pairs"], ost["last_pairs"])
            P, ps = ctx.icp_point2plane(cs, ct, cn, max_corr=mc, max_iter=it, eps=eps)
            oP, ops = orc.icp_p2plane_f32(src, tgt, nrm, max_corr=mc, max_iter=it, eps=eps)
            # With a handful of pairs (or made-up normals that make the 6x6 normal equations near-singular) the linearised
            # solve amplifies the last bit of the f64 sums into a different pose; such runs are compared only while well-posed:
            # enough pairs and a pose that is still close to rigid.
            if ops["last_pairs"] >= 12 and ps["last_pairs"] >= 12 and np.linalg.norm(oP[:3, :3].astype(np.float64)) < 2.0:
                assert np.linalg.norm(P.astype(np.float64) - oP.astype(np.float64)) <= 1e-5, (trial, n, mc, it, eps, "p2plane")
                assert (ps["iters_run"], ps["converged"], ps["empty_pairs"], ps["last_pairs"]) == (ops["iters_run"], ops["converged"], ops["empty_pairs"], ops["last_pairs"])
        ctx.tune("nn_method", 0)
    finally:
        ctx.close()

"""Randomised GPU-vs-oracle sweeps (-m gpu) over the widened rows: continuous, lattice (ties / duplicates), clustered and
extreme-magnitude clouds, far-away queries, ragged sizes.  Integer / index outputs must be bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


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
    rng = np.random.default_rng(101)
    ctx = pcr.Context(0)
    try:
        for trial in range(80):
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
    rng = np.random.default_rng(202)
    ctx = pcr.Context(0)
    try:
        for trial in range(60):
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
    rng = np.random.default_rng(303)
    ctx = pcr.Context(0)
    try:
        for trial in range(40):
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
    rng = np.random.default_rng(404)
    ctx = pcr.Context(0)
    try:
        for trial in range(40):
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

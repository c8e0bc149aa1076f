"""Registration::ICPpoint2plane (Homework9/hw9/src/registration.cpp:710-860) — the point-to-plane sibling of the ICP loop.
hw9 needs PCL + Eigen (absent): the normal-equation solve is UNPINNED (f32 Eigen `.inverse()` there, f64 elimination here);
the correspondence part is the pinned 1-NN.  GPU vs oracle: pose within 1e-5 Frobenius, loop statistics equal."""
import numpy as np
import pytest


def make_case(synth, orc, n, seed=(71, 72)):
    src, tgt = synth.kitti_like_pair(n, seed_target=seed[0], seed_pair=seed[1])
    nrm = orc.normals_knn_f64(tgt, 10, 5.0).astype(np.float32)         # target normals (any consistent field will do)
    return src, tgt, np.ascontiguousarray(nrm.T)


def test_oracle_solve6_matches_numpy(orc):
    rng = np.random.default_rng(3)
    for _ in range(50):
        A = rng.normal(size=(40, 6))
        M, v = A.T @ A, A.T @ rng.normal(size=40)
        x = np.zeros(6)
        assert orc.lib().orc_solve6(np.ascontiguousarray(M.reshape(-1)), np.ascontiguousarray(v), x) == 0
        assert np.allclose(x, np.linalg.solve(M, v), rtol=1e-9, atol=1e-12)
    x = np.zeros(6)
    assert orc.lib().orc_solve6(np.zeros(36), np.zeros(6), x) == -1   # singular -> reported, never NaN


def test_oracle_p2plane_walks_towards_the_planted_pose(orc, synth):
    src, tgt, nrm = make_case(synth, orc, 2500)
    gt = synth.gt_pose()
    e0 = np.linalg.norm(np.eye(4) - gt)
    T, st = orc.icp_p2plane_f32(src, tgt, nrm, max_iter=10, eps=0.0)
    assert st["iters_run"] == 10 and st["last_pairs"] > 1500 and np.linalg.norm(T - gt) < 0.5 * e0
    # the update is the linearised rotation, as in the reference: not orthonormal
    T1, _ = orc.icp_p2plane_f32(src, tgt, nrm, max_iter=1, eps=0.0)
    R = T1[:3, :3].astype(np.float64)
    assert np.allclose(np.diag(R), 1.0) and np.allclose(R + R.T, 2 * np.eye(3), atol=1e-7)
    # stop rules shared with the point-to-point loop: eps large -> `unchanged` counts every iteration -> break in the 16th pass
    T2, st2 = orc.icp_p2plane_f32(src, tgt, nrm, max_iter=100, eps=1e9)
    assert st2["converged"] == 1 and st2["iters_run"] == 15          # the 16th pass trips `unchanged > 15` before its update
    # nothing within reach -> the reference would divide by zero; flagged instead
    T3, st3 = orc.icp_p2plane_f32(src + 1000.0, tgt, nrm, max_iter=5)
    assert st3["empty_pairs"] == 1 and st3["iters_run"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("method", [1, 2])
def test_gpu_p2plane_matches_oracle(pcr, orc, synth, method):
    src, tgt, nrm = make_case(synth, orc, 3000)
    ctx = pcr.Context(0)
    try:
        ctx.tune("nn_method", method)
        cs, ct, cn = ctx.cloud(src), ctx.cloud(tgt), ctx.cloud(nrm)
        for kw in (dict(max_iter=8, eps=0.0), dict(max_iter=40, eps=1e9), dict(max_iter=1, eps=1e-8), dict(max_iter=0, eps=1e-8)):
            T, st = ctx.icp_point2plane(cs, ct, cn, max_corr=1.0, **kw)
            oT, ost = orc.icp_p2plane_f32(src, tgt, nrm, max_corr=1.0, **kw)
            assert np.linalg.norm(T.astype(np.float64) - oT.astype(np.float64)) <= 1e-5, kw
            got = (st["iters_run"], st["converged"], st["empty_pairs"], st["last_pairs"])
            assert got == (ost["iters_run"], ost["converged"], ost["empty_pairs"], ost["last_pairs"]), kw
            assert abs(st["last_loss"] - ost["last_loss"]) <= 1e-5 * max(1.0, abs(ost["last_loss"]))
        init = np.eye(4, dtype=np.float32); init[:3, 3] = [0.3, -0.1, 0.02]
        T, st = ctx.icp_point2plane(cs, ct, cn, init_T=init, max_iter=5, eps=0.0)
        oT, ost = orc.icp_p2plane_f32(src, tgt, nrm, init_T=init, max_iter=5, eps=0.0)
        assert np.linalg.norm(T.astype(np.float64) - oT.astype(np.float64)) <= 1e-5 and st["last_pairs"] == ost["last_pairs"]
        far = ctx.cloud(src + 1000.0)
        T, st = ctx.icp_point2plane(far, ct, cn, max_iter=5)
        assert st["empty_pairs"] == 1 and st["iters_run"] == 0
        with pytest.raises(pcr.PcrError):
            ctx.icp_point2plane(cs, ct, ctx.cloud(nrm[:, :100].copy()))
    finally:
        ctx.close()


@pytest.mark.gpu
def test_gpu_p2plane_full_size_moves_to_the_planted_pose(pcr, synth):
    """120 k x 120 k (BASELINE config 3 size) with normals from the k-NN service: no oracle pass; the pose must approach the
    planted one and every iteration must keep (nearly) all pairs."""
    src, tgt = synth.kitti_like_pair(120000)
    ctx = pcr.Context(0)
    try:
        ct = ctx.cloud(tgt)
        nrm = ctx.normals(ct, 10, 5.0).astype(np.float32)
        cn = ctx.cloud(np.ascontiguousarray(nrm.T))
        gt = synth.gt_pose()
        T, st = ctx.icp_point2plane(ctx.cloud(src), ct, cn, max_corr=1.0, max_iter=20, eps=0.0)
        assert st["iters_run"] == 20 and st["last_pairs"] > 0.99 * 120000
        assert np.linalg.norm(T - gt) < 0.25 * np.linalg.norm(np.eye(4) - gt)
    finally:
        ctx.close()

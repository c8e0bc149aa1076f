"""Next row N4: the global-registration front half of Homework9/hw9 — 33-D descriptor matching
(registration.cpp:535-615) and the RANSAC consensus count (:288-434).

Pinning: the 1-NN at dim 33 is PINNED to the vendored nanoflann (tests/golden/desc_match_hw9.npz, generated through
oracle/_ref).  The sort's tie order (std::sort), the RNG stream (std::random_device) and the f32 Eigen Kabsch are
unspecified in the reference: UNPINNED, restated (stable sort, explicit seed, f64 moments)."""
import numpy as np
import pytest


def fpfh_like(rng, n, dim=33):
    h = rng.gamma(0.6, 1.0, (n, dim))
    h *= 100.0 / h.sum(1, keepdims=True)
    return h.astype(np.float32)


def scene(seed, n_src=1200, n_tgt=1000, inliers=500, noise=0.02):
    """keypoints + descriptors of two clouds related by a known pose; `inliers` true correspondences."""
    rng = np.random.default_rng(seed)
    src = rng.uniform(-20, 20, (n_src, 3)).astype(np.float32)
    src[:, 2] *= 0.15
    a, b, c = np.deg2rad([25.0, -4.0, 3.0])
    Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    Rx = np.array([[1, 0, 0], [0, np.cos(c), -np.sin(c)], [0, np.sin(c), np.cos(c)]])
    R = Rz @ Ry @ Rx
    t = np.array([3.0, -1.5, 0.4])
    pick = rng.permutation(n_src)[:inliers]
    tgt = np.concatenate([src[pick] @ R.T + t + rng.normal(0, noise, (inliers, 3)), rng.uniform(-20, 20, (n_tgt - inliers, 3)) * [1, 1, 0.15]]).astype(np.float32)
    dsrc = fpfh_like(rng, n_src)
    dtgt = np.concatenate([np.abs(dsrc[pick] + rng.normal(0, 0.3, (inliers, 33))), fpfh_like(rng, n_tgt - inliers)]).astype(np.float32)
    return src, tgt, dsrc, dtgt, R.astype(np.float32), t.astype(np.float32)


def check_tie_rule(orc, db, q, idx, d2, ref_idx, ref_d2):
    assert np.array_equal(d2.view(np.uint32), ref_d2.view(np.uint32))
    for k in np.flatnonzero(idx != ref_idx):                       # nanoflann answers from the tie set (first VISITED wins)
        assert orc.lib().orc_d2_dim_f32(np.ascontiguousarray(db[ref_idx[k]]), np.ascontiguousarray(q[k]), db.shape[1]) == d2[k]
        assert idx[k] < ref_idx[k]                                   # ours is the lowest index of the set


# ----------------------------------------------------------------------------------------------- CPU: oracle, host logic
def test_oracle_nn1_dim33_matches_nanoflann_fixture(orc, golden):
    g = golden("desc_match_hw9.npz")
    for db, q, ri, rd in ((g["desc_src"], g["desc_tgt"], g["nn_src_of_tgt"], g["d2_src_of_tgt"]),
                          (g["desc_tgt"], g["desc_src"], g["nn_tgt_of_src"], g["d2_tgt_of_src"])):
        idx, d2 = orc.nn1_dim_f32(db, q)
        check_tie_rule(orc, db, q, idx, d2, ri, rd)
    assert (g["d2_src_of_tgt"][580:590] == 0).all()                 # the planted exact matches


@pytest.mark.parametrize("dim", [1, 3, 4, 7, 33, 34])
def test_oracle_nn1_dim_matches_nanoflann_live(orc, dim):
    if not orc.have_ref():
        pytest.skip("oracle/_ref not built (reference absent)")
    rng = np.random.default_rng(dim)
    db, q = fpfh_like(rng, 500, dim), fpfh_like(rng, 300, dim)
    if dim == 1:
        db, q = rng.normal(0, 1, (500, 1)).astype(np.float32), rng.normal(0, 1, (300, 1)).astype(np.float32)
    idx, d2 = orc.nn1_dim_f32(db, q)
    ri, rd = orc.ref_nano_nn1_dim_f32(db, q)
    check_tie_rule(orc, db, q, idx, d2, ri, rd)


def test_oracle_match_union_structure(orc, golden):
    g = golden("desc_match_hw9.npz")
    a, b = g["desc_src"], g["desc_tgt"]
    pairs, dist = orc.match_union_f32(a, b, 0.5)
    assert pairs.shape[0] == int(np.floor(np.float32(1 - np.float32(0.5)) * np.float32(1300)))
    assert (np.diff(dist) >= 0).all()
    # every kept pair is one of the 1300 directed matches, with its distance
    i_ts, d_ts = orc.nn1_dim_f32(a, b)
    i_st, d_st = orc.nn1_dim_f32(b, a)
    allp = np.concatenate([np.stack([i_ts, np.arange(600)], 1), np.stack([np.arange(700), i_st], 1)]).astype(np.uint32)
    alld = np.concatenate([d_ts, d_st])
    o = np.argsort(alld, kind="stable")
    assert np.array_equal(pairs, allp[o][: pairs.shape[0]]) and np.array_equal(dist, alld[o][: pairs.shape[0]])
    assert orc.match_union_f32(a, b, 1.0)[0].shape[0] == 0
    assert orc.match_union_f32(a, b, 0.0)[0].shape[0] == 1300


def test_oracle_match_inter_structure(orc, golden):
    g = golden("desc_match_hw9.npz")
    a, b = g["desc_src"], g["desc_tgt"]
    pairs, dist = orc.match_inter_f32(a, b, 0.0)
    s2t, ds = orc.nn1_dim_f32(b, a)
    t2s, _ = orc.nn1_dim_f32(a, b)
    mutual = [(s, int(s2t[s])) for s in range(a.shape[0]) if t2s[s2t[s]] == s]
    assert sorted(map(tuple, pairs.tolist())) == sorted(mutual) and (np.diff(dist) >= 0).all()
    assert np.array_equal(dist, ds[pairs[:, 0]])
    half, _ = orc.match_inter_f32(a, b, 0.5)
    assert half.shape[0] == int(np.floor(np.float32(0.5) * np.float32(len(mutual)))) and np.array_equal(half, pairs[: half.shape[0]])
    assert len(mutual) > 300                                          # the planted noisy copies find each other


def test_sample_quads_host_logic(pcr):
    src, tgt, dsrc, dtgt, R, t = scene(3, 300, 300, 150)
    rng = np.random.default_rng(0)
    pairs = np.stack([rng.integers(0, 300, 400), rng.integers(0, 300, 400)], 1).astype(np.uint32)
    q1 = pcr.ransac_sample_quads(src, pairs, 2000, 42)
    q2 = pcr.ransac_sample_quads(src, pairs, 2000, 42)
    q3 = pcr.ransac_sample_quads(src, pairs, 2000, 43)
    assert np.array_equal(q1, q2) and not np.array_equal(q1, q3)
    assert q1.max() < 400
    # :324-332 re-draws only against the element being compared -> adjacent duplicates are impossible, and the accepted
    # quads satisfy the coplanarity gate (:334-351) as the reference evaluates it (f32, signed)
    P = src[pairs[q1, 0]]
    p1, p2, p3 = P[:, 1] - P[:, 0], P[:, 2] - P[:, 0], P[:, 3] - P[:, 0]
    nrm = np.stack([p1[:, 1] * p2[:, 2] - p1[:, 2] * p2[:, 1], p1[:, 2] * p2[:, 0] - p1[:, 0] * p2[:, 2], p1[:, 0] * p2[:, 1] - p1[:, 1] * p2[:, 0]], 1)
    length = np.sqrt((nrm[:, 0] * nrm[:, 0] + nrm[:, 1] * nrm[:, 1] + nrm[:, 2] * nrm[:, 2]).astype(np.float32))
    dist = ((nrm[:, 0] * p3[:, 0] + nrm[:, 1] * p3[:, 1]).astype(np.float32) + nrm[:, 2] * p3[:, 2]).astype(np.float32) / length
    assert (dist.astype(np.float64) > 0.15 - 1e-4).all()
    assert (q1[:, 3] != q1[:, 2]).all()
    # degenerate inputs fail loudly instead of spinning forever
    flat = src.copy(); flat[:, 2] = 0
    with pytest.raises(pcr.PcrError):
        pcr.ransac_sample_quads(flat, pairs, 10, 1)
    with pytest.raises(pcr.PcrError):
        pcr.ransac_sample_quads(src, pairs[:3], 10, 1)
    with pytest.raises(pcr.PcrError):
        pcr.ransac_sample_quads(src[:10], pairs, 10, 1)             # pair index beyond the cloud


def test_oracle_ransac_recovers_the_pose(pcr, orc):
    src, tgt, dsrc, dtgt, R, t = scene(11)
    pairs, dist = orc.match_union_f32(dsrc, dtgt, 0.5)
    quads = pcr.ransac_sample_quads(src, pairs, 3000, 7)
    win, Rr, tr, best, counts = orc.ransac_global_f32(src, tgt, pairs, quads, 0.3)
    assert win >= 0 and best == counts.max() and counts[win] == best and (counts[:win] < best).all()
    assert best > 0.5 * pairs.shape[0]
    assert np.linalg.norm(Rr - R) < 0.02 and np.linalg.norm(tr - t) < 0.3
    assert orc.consensus_count_f32(src, tgt, pairs, Rr, tr, 0.3) == best


def reflected_quad_scene():
    """four non-coplanar source keypoints whose targets are their MIRROR image (x -> -x) rotated and shifted: the optimal
    orthogonal map is a reflection, det(U V^T) < 0, and the reference's repair branch (registration.cpp:386-392) runs;
    plus 60 more correspondences scattered around both images so that consensus counts discriminate between poses."""
    rng = np.random.default_rng(5)
    src4 = np.array([[0.0, 0.0, 0.0], [2.0, 0.3, 0.1], [0.2, 1.7, -0.2], [0.4, 0.5, 1.9]])
    a = np.deg2rad(20.0)
    Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    mirror = np.diag([-1.0, 1.0, 1.0])
    tvec = np.array([1.5, -0.7, 0.3])
    more = rng.uniform(-2, 2, (60, 3))
    src = np.concatenate([src4 + [3.0, 2.0, 1.0], more]).astype(np.float32)        # centroid away from the origin: t matters
    tgt = (src.astype(np.float64) @ (Rz @ mirror).T + tvec + rng.normal(0, 0.01, src.shape)).astype(np.float32)
    pairs = np.stack([np.arange(64), np.arange(64)], 1).astype(np.uint32)
    return src, tgt, pairs, np.array([0, 1, 2, 3], np.uint32)


def numpy_ransac_block(src, tgt, quad_pairs):
    """registration.cpp:372-392 with numpy's SVD (U V^T and V B U^T do not depend on the SVD's sign choices)."""
    P = src[quad_pairs[:, 0]].astype(np.float64).T
    Q = tgt[quad_pairs[:, 1]].astype(np.float64).T
    sc, tc = P.mean(1), Q.mean(1)
    U, S, Vt = np.linalg.svd((Q - tc[:, None]) @ (P - sc[:, None]).T)
    R0 = U @ Vt
    t = tc - R0 @ sc
    det = np.linalg.det(R0)
    R = R0
    if det < 0:
        R = Vt.T @ np.diag([1.0, 1.0, det]) @ U.T
    return R0, R, t, det


def test_ransac_kabsch_block_keeps_t_of_the_unrepaired_rotation(orc):
    src, tgt, pairs, quad = reflected_quad_scene()
    R0, R, t, det = numpy_ransac_block(src, tgt, pairs[quad])
    assert det < -0.99
    rc, oR, ot = orc.ransac_hypothesis(src, tgt, pairs, quad)
    assert rc == 0
    assert np.abs(oR - R).max() < 1e-5 and np.abs(ot - t).max() < 1e-4        # R repaired, t from U V^T (:383, never recomputed)
    # the ICP block (registration.cpp:990-998) would recompute t from the repaired R: a different translation on this quad
    sums = np.zeros(16)
    P, Q = src[quad].astype(np.float64), tgt[quad].astype(np.float64)
    sums[0:3], sums[3:6], sums[6:15], sums[15] = P.sum(0), Q.sum(0), (Q.T @ P).reshape(9), 4
    rc2, iR, it = orc.kabsch_solve(sums)
    rc3, rR, rt = orc.kabsch_solve_ransac(sums)
    assert np.array_equal(iR, rR) and np.array_equal(rR, oR) and np.array_equal(rt, ot)
    assert np.abs(it - ot).max() > 0.1
    assert np.abs(it - (Q.mean(0) - R @ P.mean(0))).max() < 1e-4
    # proper quads: both blocks agree bit for bit
    Qp = (P @ np.array([[0.0, -1, 0], [1, 0, 0], [0, 0, 1]]).T + 1.0)
    sums[3:6], sums[6:15] = Qp.sum(0), (Qp.T @ P).reshape(9)
    a, b = orc.kabsch_solve(sums), orc.kabsch_solve_ransac(sums)
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.linalg.det(a[1].astype(np.float64)) > 0.99


# ----------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_gpu_ransac_reflected_quad_R_t_and_count(pcr, orc):
    """a quad with det(U V^T) < 0 — R, t and the consensus count of the GPU hypothesis kernel against the oracle's
    statement-by-statement restatement of registration.cpp:372-392 and against numpy."""
    src, tgt, pairs, quad = reflected_quad_scene()
    R0, R, t, det = numpy_ransac_block(src, tgt, pairs[quad])
    ctx = pcr.Context(0)
    try:
        for thr in (50.0, 3.0, 0.5):
            win, Rg, tg, best, counts = ctx.ransac_global(src, tgt, pairs, quad[None], thr)
            rc, oR, ot = orc.ransac_hypothesis(src, tgt, pairs, quad)
            want = orc.consensus_count_f32(src, tgt, pairs, oR, ot, thr)
            assert counts[0] == want == best
            if want:
                assert win == 0
                assert np.array_equal(Rg.view(np.uint32), oR.view(np.uint32)) and np.array_equal(tg.view(np.uint32), ot.view(np.uint32))
                assert np.abs(Rg - R).max() < 1e-5 and np.abs(tg - t).max() < 1e-4
        # mixed batch: reflected and proper quads side by side, every count bit-exact
        rng = np.random.default_rng(3)
        quads = np.stack([rng.permutation(64)[:4] for _ in range(500)]).astype(np.uint32)
        quads[::7] = quad
        win, Rg, tg, best, counts = ctx.ransac_global(src, tgt, pairs, quads, 0.5)
        ow, oR, ot, obest, ocounts = orc.ransac_global_f32(src, tgt, pairs, quads, 0.5)
        assert np.array_equal(counts, ocounts) and (win, best) == (ow, obest)
        assert np.array_equal(Rg.view(np.uint32), oR.view(np.uint32)) and np.array_equal(tg.view(np.uint32), ot.view(np.uint32))
    finally:
        ctx.close()


@pytest.mark.gpu
def test_gpu_nn1_desc_matches_nanoflann_fixture_and_oracle(pcr, orc, golden):
    g = golden("desc_match_hw9.npz")
    ctx = pcr.Context(0)
    try:
        for db, q, ri, rd in ((g["desc_src"], g["desc_tgt"], g["nn_src_of_tgt"], g["d2_src_of_tgt"]),
                              (g["desc_tgt"], g["desc_src"], g["nn_tgt_of_src"], g["d2_tgt_of_src"])):
            idx, d2 = ctx.nn1_desc(db, q)
            oi, od = orc.nn1_dim_f32(db, q)
            assert np.array_equal(idx, oi) and np.array_equal(d2.view(np.uint32), od.view(np.uint32))
            check_tie_rule(orc, db, q, idx, d2, ri, rd)
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dim,n,m", [(33, 5000, 3000), (33, 1, 70), (33, 65, 1), (3, 2000, 500), (1, 300, 100), (4, 777, 129), (5, 1000, 64), (64, 900, 300), (130, 400, 200), (256, 100, 65)])
def test_gpu_nn1_desc_any_dim_bit_exact(pcr, orc, dim, n, m):
    rng = np.random.default_rng(dim * 1000 + n)
    db, q = fpfh_like(rng, n, dim), fpfh_like(rng, m, dim)
    if n > 10:
        db[n // 2] = db[1]                                          # a tie set
        q[0] = db[1]
    ctx = pcr.Context(0)
    try:
        idx, d2 = ctx.nn1_desc(db, q)
        oi, od = orc.nn1_dim_f32(db, q)
        assert np.array_equal(idx, oi) and np.array_equal(d2.view(np.uint32), od.view(np.uint32))
    finally:
        ctx.close()


@pytest.mark.gpu
def test_gpu_nn1_desc_edges(pcr, orc):
    ctx = pcr.Context(0)
    try:
        rng = np.random.default_rng(2)
        db, q = fpfh_like(rng, 200), fpfh_like(rng, 50)
        idx, d2 = ctx.nn1_desc(np.zeros((0, 33), np.float32), q)    # empty database: nothing accepted
        assert (idx == 0xFFFFFFFF).all() and np.isinf(d2).all()
        idx, d2 = ctx.nn1_desc(db, np.zeros((0, 33), np.float32))
        assert idx.size == 0
        db[3, 5] = np.nan; db[4, 0] = np.inf; q[7, 2] = np.nan      # NaN / inf rows are never accepted (d2 < FLT_MAX fails)
        db[9] = 3e19                                                # d2 overflows to +inf
        idx, d2 = ctx.nn1_desc(db, q)
        oi, od = orc.nn1_dim_f32(db, q)
        assert np.array_equal(idx, oi) and np.array_equal(d2.view(np.uint32), od.view(np.uint32))
        assert idx[7] == 0xFFFFFFFF and not np.isin(idx, [3, 4, 9]).any()
        with pytest.raises(pcr.PcrError):
            ctx.nn1_desc(np.zeros((4, 300), np.float32), np.zeros((4, 300), np.float32))
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("rate", [0.5, 0.0, 0.9, 1.0])
def test_gpu_match_union_equals_oracle(pcr, orc, golden, rate):
    g = golden("desc_match_hw9.npz")
    ctx = pcr.Context(0)
    try:
        for a, b in ((g["desc_src"], g["desc_tgt"]), scene(5)[2:4]):
            pairs, dist = ctx.match_union(a, b, rate)
            op, od = orc.match_union_f32(a, b, rate)
            assert np.array_equal(pairs, op) and np.array_equal(dist.view(np.uint32), od.view(np.uint32))
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("rate", [0.5, 0.0, 0.9])
def test_gpu_match_inter_equals_oracle(pcr, orc, golden, rate):
    g = golden("desc_match_hw9.npz")
    ctx = pcr.Context(0)
    try:
        for a, b in ((g["desc_src"], g["desc_tgt"]), scene(6)[2:4]):
            pairs, dist = ctx.match_inter(a, b, rate)
            op, od = orc.match_inter_f32(a, b, rate)
            assert np.array_equal(pairs, op) and np.array_equal(dist.view(np.uint32), od.view(np.uint32))
        assert ctx.match_inter(g["desc_src"], np.zeros((0, 33), np.float32), 0.5)[0].shape[0] == 0
    finally:
        ctx.close()


@pytest.mark.gpu
def test_gpu_consensus_count_equals_oracle(pcr, orc):
    src, tgt, dsrc, dtgt, R, t = scene(21)
    pairs, _ = orc.match_union_f32(dsrc, dtgt, 0.5)
    rng = np.random.default_rng(1)
    H = 300
    Rt = np.zeros((H, 12), np.float32)
    for h in range(H):
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        s = rng.uniform(0, 1)
        Rh = (R if h % 3 == 0 else q.astype(np.float32))
        Rt[h, :9] = Rh.reshape(9)
        Rt[h, 9:] = t + (rng.normal(0, 0.2 * s, 3) if h % 3 == 0 else rng.normal(0, 5, 3))
    Rt[0, :9], Rt[0, 9:] = R.reshape(9), t
    ctx = pcr.Context(0)
    try:
        for thr in (0.3, 0.05, 0.0, 2.0, -1.0):
            counts = ctx.consensus_count(src, tgt, pairs, Rt, thr)
            want = np.array([orc.consensus_count_f32(src, tgt, pairs, Rt[h, :9], Rt[h, 9:], thr) for h in range(H)], np.uint32)
            assert np.array_equal(counts, want), thr
        assert ctx.consensus_count(src, tgt, pairs, Rt, 0.3)[0] > 400
        assert (ctx.consensus_count(src, tgt, pairs[:0], Rt, 0.3) == 0).all()
        assert ctx.consensus_count(src, tgt, pairs, Rt[:0], 0.3).size == 0
        bad = pairs.copy(); bad[5, 0] = 50000
        with pytest.raises(pcr.PcrError):
            ctx.consensus_count(src, tgt, bad, Rt, 0.3)
    finally:
        ctx.close()


@pytest.mark.gpu
def test_gpu_ransac_global_equals_oracle(pcr, orc):
    src, tgt, dsrc, dtgt, R, t = scene(31)
    ctx = pcr.Context(0)
    try:
        pairs, dist = ctx.match_union(dsrc, dtgt, 0.5)
        quads = pcr.ransac_sample_quads(src, pairs, 6000, 99)
        win, Rg, tg, best, counts = ctx.ransac_global(src, tgt, pairs, quads, 0.3)
        ow, oR, ot, obest, ocounts = orc.ransac_global_f32(src, tgt, pairs, quads, 0.3)
        assert np.array_equal(counts, ocounts)                      # integer work: bit-exact, every hypothesis
        assert (win, best) == (ow, obest)
        assert np.array_equal(Rg.view(np.uint32), oR.view(np.uint32)) and np.array_equal(tg.view(np.uint32), ot.view(np.uint32))
        assert np.linalg.norm(Rg - R) < 0.02 and np.linalg.norm(tg - t) < 0.3
        # empty consensus sets everywhere (negative threshold) -> no winner, R/t untouched
        w2, R2, t2, b2, c2 = ctx.ransac_global(src, tgt, pairs, quads[:500], -1.0)
        assert w2 == -1 and b2 == 0 and (c2 == 0).all() and (R2 == 0).all()
        # degenerate quads (the same correspondence four times): the rank-0 Kabsch still returns a pose; counts agree
        dq = np.repeat(np.arange(50, dtype=np.uint32)[:, None], 4, 1)
        w3, R3, t3, b3, c3 = ctx.ransac_global(src, tgt, pairs, dq, 0.3)
        o3 = orc.ransac_global_f32(src, tgt, pairs, dq, 0.3)
        assert np.array_equal(c3, o3[4]) and w3 == o3[0]
        with pytest.raises(pcr.PcrError):
            ctx.ransac_global(src, tgt, pairs, np.full((3, 4), pairs.shape[0], np.uint32), 0.3)
    finally:
        ctx.close()


@pytest.mark.gpu
def test_gpu_ransac_driver_size_properties(pcr):
    """hw9's shipped size: 80 000 iterations (main.cpp:86) — no oracle pass; the winner's count must equal the consensus
    count of the returned pose, be the first maximum, and the pose must be the planted one."""
    src, tgt, dsrc, dtgt, R, t = scene(41, 3000, 2500, 900)
    ctx = pcr.Context(0)
    try:
        pairs, dist = ctx.match_union(dsrc, dtgt, 0.5)
        quads = pcr.ransac_sample_quads(src, pairs, 80000, 2020)
        win, Rg, tg, best, counts = ctx.ransac_global(src, tgt, pairs, quads, 0.3)
        assert best == counts.max() and win == int(np.argmax(counts))
        assert ctx.consensus_count(src, tgt, pairs, np.r_[Rg.reshape(9), tg][None], 0.3)[0] == best
        assert np.linalg.norm(Rg - R) < 0.01 and np.linalg.norm(tg - t) < 0.2
    finally:
        ctx.close()

"""Next row N1: ISS keypoints, Homework7/hw7/src/iss_detector.cpp:38-152.

Pinning: the radius neighbourhoods (hw7's float kd-tree, src/kdtree.cpp:337-369) are PINNED — tests/golden/iss_hw7.npz
holds (count, index sum, index xor) per point from the reference's own kd-tree compiled in place, on the reference's
test clouds at the driver's radii (main.cpp:84-91).  The covariance / eigenvalue half needs Eigen (absent): UNPINNED, the
oracle restates it with f64 accumulation; lambda3 is compared with a tolerance and the keypoint decision is checked
exactly against the decision rule applied to the GPU's own lambda3."""
import numpy as np
import pytest

MODELS = ["airplane_0001", "chair_0001"]


def digest(row, idx):
    cnt = np.diff(row).astype(np.uint32)
    ssum = np.add.reduceat(np.r_[idx.astype(np.uint64), np.uint64(0)], row[:-1].clip(max=idx.size)).astype(np.uint64)
    ssum[cnt == 0] = 0
    sxor = np.bitwise_xor.reduceat(np.r_[idx.astype(np.uint32), np.uint32(0)], row[:-1].clip(max=idx.size)).astype(np.uint32)
    sxor[cnt == 0] = 0
    return cnt, ssum, sxor


@pytest.mark.parametrize("name", MODELS)
def test_oracle_radius_f32_matches_hw7_kdtree_fixture(orc, golden, name):
    g = golden("iss_hw7.npz")
    xyz = g[f"xyz_{name}"]
    for tag in ("local", "nms"):
        row, idx, _ = orc.radius_f32(xyz, xyz, float(g[f"{tag}_r"]))
        cnt, ssum, sxor = digest(row, idx)
        assert np.array_equal(cnt, g[f"{tag}_cnt_{name}"])
        assert np.array_equal(ssum, g[f"{tag}_sum_{name}"])
        assert np.array_equal(sxor, g[f"{tag}_xor_{name}"])


def test_oracle_radius_f32_matches_hw7_kdtree_live(orc, synth):
    if not orc.have_hw7():
        pytest.skip("oracle/_ref/libhw7_ref.so not built (reference absent)")
    rng = np.random.default_rng(5)
    db = np.ascontiguousarray(synth.kitti_like_scan(3000).T)
    q = db[rng.integers(0, 3000, 300)] + rng.normal(0, 0.05, (300, 3)).astype(np.float32)
    for r in (0.3, 1.0, 2.5):
        row, idx, dist = orc.ref_hw7_radius(db, q, r)
        orow, oidx, odist = orc.radius_f32(db, q, r)
        assert np.array_equal(row, orow)
        for i in range(300):
            o = np.argsort(idx[row[i]:row[i + 1]], kind="stable")
            assert np.array_equal(idx[row[i]:row[i + 1]][o], oidx[row[i]:row[i + 1]])
            assert np.array_equal(dist[row[i]:row[i + 1]][o].view(np.uint32), odist[row[i]:row[i + 1]].view(np.uint32))


def iss_numpy(xyz, r_local, r_nms, g21, g32, min_nb, weighted):
    """Independent numpy restatement for small n (f64 covariance, numpy eigvalsh)."""
    x = xyz.astype(np.float32)
    n = x.shape[0]

    def dist(i):
        s = np.zeros(n, np.float32)
        for c in range(3):
            e = (x[:, c] - x[i, c]).astype(np.float32)
            s = (s.astype(np.float64) + e.astype(np.float64) ** 2).astype(np.float32)
        return np.sqrt(s)

    nb = [np.flatnonzero(dist(i) <= np.float32(r_local)) for i in range(n)]
    l3 = np.full(n, -1, np.float32)
    for i in range(n):
        if nb[i].size < 3:
            continue
        d = (x[nb[i]] - x[i]).astype(np.float32).astype(np.float64)
        w = np.array([np.float32(1) / np.float32(nb[j].size) for j in nb[i]], np.float64) if weighted else np.ones(nb[i].size)
        cov = (d * w[:, None]).T @ d
        if weighted:
            cov /= w.sum()
        ev = np.linalg.eigvalsh(cov).astype(np.float32)
        if ev[1] / ev[2] < np.float32(g21) and ev[0] / ev[1] < np.float32(g32) and ev[0] > 0:
            l3[i] = ev[0]
    keys = nms_numpy(x, l3, r_nms, min_nb)
    return keys, l3


def nms_numpy(x, l3, r_nms, min_nb):
    n = x.shape[0]
    keys = []
    for i in range(n):
        if l3[i] == -1:
            continue
        s = np.zeros(n, np.float32)
        for c in range(3):
            e = (x[:, c] - x[i, c]).astype(np.float32)
            s = (s.astype(np.float64) + e.astype(np.float64) ** 2).astype(np.float32)
        nb = np.flatnonzero(np.sqrt(s) <= np.float32(r_nms))
        if nb.size < min_nb:
            continue
        if not (l3[i] < l3[nb]).any():
            keys.append(i)
    return np.array(keys, np.int64)


@pytest.mark.parametrize("weighted", [True, False])
def test_oracle_iss_matches_numpy_restatement(orc, golden, weighted):
    xyz = golden("iss_hw7.npz")["xyz_chair_0001"][::8]           # 1250 points, radii scaled to keep ~25 neighbours
    key, l3 = orc.iss_f32(np.ascontiguousarray(xyz.T), 0.24, 0.16, 0.9, 0.9, 5, weighted)
    nkeys, nl3 = iss_numpy(xyz, 0.24, 0.16, 0.9, 0.9, 5, weighted)
    same = (l3 == -1) == (nl3 == -1)
    assert same.mean() > 0.999                                     # a gamma test may flip on a last-ulp difference
    both = (l3 != -1) & (nl3 != -1)
    assert np.allclose(l3[both], nl3[both], rtol=2e-5, atol=0)
    assert np.array_equal(np.flatnonzero(key), nms_numpy(xyz, l3, 0.16, 5))
    assert len(set(np.flatnonzero(key)) ^ set(nkeys)) <= 2


def test_oracle_iss_selfgolden(orc, golden):
    g = golden("iss_hw7.npz")
    for name in MODELS:
        key, l3 = orc.iss_f32(np.ascontiguousarray(g[f"xyz_{name}"].T), float(g["local_r"]), float(g["nms_r"]), 0.9, 0.9, 5, True)
        assert np.array_equal(np.flatnonzero(key), g[f"selfgolden_keys_{name}"])
        assert np.array_equal(l3.view(np.uint32), g[f"selfgolden_lambda3_{name}"].view(np.uint32))


def check_gpu_against_oracle(ctx, orc, xyz, r_local, r_nms, g21=0.9, g32=0.9, min_nb=5, weighted=True, want_cnt=None):
    soa = np.ascontiguousarray(xyz.T)
    idx, l3, cnt = ctx.iss_keypoints(ctx.cloud(soa), r_local, r_nms, g21, g32, min_nb, weighted)
    okey, ol3 = orc.iss_f32(soa, r_local, r_nms, g21, g32, min_nb, weighted)
    # neighbourhood sizes: integer work, bit-exact (pinned to the reference kd-tree through want_cnt when given)
    row, _, _ = orc.radius_f32(xyz, xyz, r_local)
    assert np.array_equal(cnt, np.diff(row).astype(np.uint32))
    if want_cnt is not None:
        assert np.array_equal(cnt, want_cnt)
    # lambda3: f64 sums in a different order -> identical after the f32 rounding except at rounding boundaries
    flip = (l3 == -1) != (ol3 == -1)
    assert flip.sum() <= max(1, xyz.shape[0] // 2000)
    both = (l3 != -1) & (ol3 != -1)
    assert np.allclose(l3[both], ol3[both], rtol=1e-6, atol=0)
    assert (l3[both].view(np.uint32) == ol3[both].view(np.uint32)).mean() > 0.999
    # the decision rule, exactly, on the GPU's own lambda3
    if xyz.shape[0] <= 12000:
        assert np.array_equal(idx, nms_numpy(xyz.astype(np.float32), l3, r_nms, min_nb))
    if not flip.any() and np.array_equal(l3.view(np.uint32), ol3.view(np.uint32)):
        assert np.array_equal(idx, np.flatnonzero(okey))
    return idx, l3


@pytest.mark.gpu
@pytest.mark.parametrize("name", MODELS)
def test_gpu_iss_reference_clouds_driver_parameters(pcr, orc, golden, name):
    g = golden("iss_hw7.npz")
    ctx = pcr.Context(0)
    try:
        idx, l3 = check_gpu_against_oracle(ctx, orc, g[f"xyz_{name}"], float(g["local_r"]), float(g["nms_r"]),
                                           want_cnt=g[f"local_cnt_{name}"])
        assert 10 < idx.size < 200
        gk = set(g[f"selfgolden_keys_{name}"].tolist())
        assert len(gk ^ set(idx.tolist())) <= 2
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("weighted,g21,g32,min_nb", [(False, 0.9, 0.9, 5), (True, 0.6, 0.7, 1), (True, 0.975, 0.975, 20)])
def test_gpu_iss_parameter_variants(pcr, orc, golden, weighted, g21, g32, min_nb):
    xyz = golden("iss_hw7.npz")["xyz_airplane_0001"][::2]
    ctx = pcr.Context(0)
    try:
        for lanes in (1, 4, 8, 32):
            ctx.tune("iss_lanes", lanes)
            check_gpu_against_oracle(ctx, orc, xyz, 0.15, 0.1, g21, g32, min_nb, weighted)
    finally:
        ctx.close()


@pytest.mark.gpu
def test_gpu_iss_lidar_scan_and_edges(pcr, orc, synth):
    ctx = pcr.Context(0)
    try:
        scan = np.ascontiguousarray(synth.kitti_like_scan(20000).T)
        check_gpu_against_oracle(ctx, orc, scan, 1.2, 0.8)
        # duplicates, a far outlier (huge bounding box -> the grid may only grow its cells), non-finite points
        pts = scan[:3000].copy()
        pts[100:110] = pts[99]
        pts[5] = [4000.0, -4000.0, 300.0]
        pts[7] = [np.nan, 0, 0]
        pts[9] = [np.inf, 1, 1]
        check_gpu_against_oracle(ctx, orc, pts, 1.5, 1.0)
        # tiny inputs: fewer than 3 neighbours -> no keypoint; r = 0 -> only coincident points are neighbours
        tiny = np.array([[0, 0, 0], [0.01, 0, 0]], np.float32)
        idx, l3, cnt = ctx.iss_keypoints(ctx.cloud(np.ascontiguousarray(tiny.T)), 0.5, 0.5)
        assert idx.size == 0 and (l3 == -1).all() and (cnt == 2).all()
        idx, l3, cnt = ctx.iss_keypoints(ctx.cloud(np.ascontiguousarray(pts[90:120].T)), 0.0, 0.0)
        assert idx.size == 0 and cnt[9] == 11 and cnt[0] == 1
        idx, l3, cnt = ctx.iss_keypoints(ctx.cloud(np.zeros((3, 0), np.float32)), 0.5, 0.5)
        assert idx.size == 0 and l3.size == 0
        with pytest.raises(pcr.PcrError):
            ctx.iss_keypoints(ctx.cloud(np.ascontiguousarray(tiny.T)), -1.0, 0.5)
    finally:
        ctx.close()


@pytest.mark.gpu
def test_gpu_iss_full_scan_properties(pcr, synth):
    """120 k-point scan (BASELINE configs[1] size): no O(n^2) oracle — size-independent properties instead."""
    ctx = pcr.Context(0)
    try:
        scan = synth.kitti_like_scan(120000)
        c = ctx.cloud(scan)
        idx, l3, cnt = ctx.iss_keypoints(c, 1.2, 0.8)
        assert (cnt >= 1).all()                                      # every finite point is its own neighbour
        assert (l3[idx] > 0).all() and idx.size > 100
        # permutation invariance of the keypoint SET (neighbourhood sums are order-dependent only in the last f64 bits)
        perm = np.random.default_rng(0).permutation(120000)
        idx2, l32, cnt2 = ctx.iss_keypoints(ctx.cloud(np.ascontiguousarray(scan[:, perm])), 1.2, 0.8)
        assert np.array_equal(cnt2, cnt[perm])
        assert (l32.view(np.uint32) == l3[perm].view(np.uint32)).mean() > 0.9999
        assert len(set(perm[idx2].tolist()) ^ set(idx.tolist())) <= 4
        # keypoints are mutually separated: no two keypoints with different lambda3 within the non-max radius
        kp = scan[:, idx].T.astype(np.float64)
        d = np.linalg.norm(kp[:, None, :] - kp[None, :, :], axis=2)
        close = (d < 0.8 * 0.999) & ~np.eye(idx.size, dtype=bool)
        ii, jj = np.nonzero(close)
        assert (l3[idx][ii] == l3[idx][jj]).all()
    finally:
        ctx.close()

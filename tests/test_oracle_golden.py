"""CPU tests (-m "not gpu"): pin the oracle (oracle/pcr_oracle.c) against the fixtures generated from the
REFERENCE ITSELF (tests/golden/gen_golden.py: compiled hw2 kd-tree/octree, vendored nanoflann, HW4 numpy
expression), i.e. SURVEY.md §8c F1-F6.  Tie handling follows the tie-set rule of SURVEY.md §7.2."""
import numpy as np
import pytest

NN1_CASES = ["synth1000", "synth4096", "kitti4096", "lattice1000"]


def bits32(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def bits64(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


# ---------------------------------------------------------------- F1: published known-answer test
def test_kat_query5_k8_matches_result_cpp_txt(orc, golden):
    g = golden("kat_kitti_q5.npz")
    # Homework2/hw2/result_cpp.txt:13-20
    assert g["knn_idx"].tolist() == [5, 1972, 6, 1971, 1970, 3946, 8, 3945]
    assert int(g["comparison_count"]) == 49
    printed = [0, 0.589554, 1.22663, 1.39844, 1.46411, 1.57562, 1.63461, 1.65879]
    assert np.allclose(g["knn_dist"], printed, rtol=0, atol=5e-6)
    db = g["db_f32"].astype(np.float64)
    q = db[int(g["query_index"])][None, :]
    idx, dist = orc.knn_f64(db, q, 8)
    assert np.array_equal(idx[0], g["knn_idx"])
    assert np.array_equal(bits64(dist[0]), bits64(g["knn_dist"]))


def test_kat_query5_radius(orc, golden):
    g = golden("kat_kitti_q5.npz")
    db = g["db_f32"].astype(np.float64)
    q = db[5][None, :]
    row, idx, dist = orc.radius_f64(db, q, float(g["radius"]))
    o = np.argsort(g["radius_idx_visit_order"])
    assert np.array_equal(idx, g["radius_idx_visit_order"][o])
    assert np.array_equal(bits64(dist), bits64(g["radius_dist_visit_order"][o]))
    # result_cpp.txt:26-33 lists exactly these 8 neighbours
    assert sorted(idx.tolist()) == [5, 6, 8, 1970, 1971, 1972, 3945, 3946]


def test_readbinary_eof_duplicate_recorded(golden):
    g = golden("kat_kitti_q5.npz")
    # test.hpp:24 eof() loop appends one duplicate: "124669 points read!" (comparision.txt:2)
    assert int(g["n_points_readBinary"]) == 124669 and int(g["n_points_file"]) == 124668


# ---------------------------------------------------------------- F2: nanoflann f32 1-NN
@pytest.mark.parametrize("case", NN1_CASES)
def test_nn1_oracle_vs_nanoflann(orc, golden, case):
    g = golden(f"nn1_nanoflann_{case}.npz")
    idx, d2 = orc.nn1_f32(g["tgt"], g["src"])
    # (a) distance bit-equal to the reference's, always
    assert np.array_equal(bits32(d2), bits32(g["d2"]))
    ties = orc.nn1_tiecount_f32(g["tgt"], g["src"])
    single = ties == 1
    # (b) index equal whenever the tie set is a singleton
    assert np.array_equal(idx[single], g["idx"][single])
    # (c) otherwise: reference index in the tie set, oracle index == min(tie set)
    tgt = g["tgt"]; src = g["src"]
    for i in np.where(~single)[0]:
        dx = src[0, i] - tgt[0]; dy = src[1, i] - tgt[1]; dz = src[2, i] - tgt[2]
        dd = (dx * dx + dy * dy) + dz * dz
        tie_set = np.where(bits32(dd) == bits32(d2[i:i + 1])[0])[0]
        assert tie_set.size == ties[i]
        assert g["idx"][i] in tie_set
        assert idx[i] == tie_set.min()
    if case == "lattice1000":
        assert (~single).sum() > 0, "lattice input is supposed to exercise ties"
    else:
        assert single.all()


def test_d2_arithmetic_is_unfused(orc):
    # a case where fma(dz,dz,acc) differs from the unfused sum in the last bit
    rng = np.random.default_rng(0)
    a = rng.uniform(-50, 50, size=(20000, 6)).astype(np.float32)
    diff = 0
    for r in a[:2000]:
        got = orc.lib().orc_d2_f32(*[float(v) for v in r])
        dx, dy, dz = np.float32(r[0] - r[3]), np.float32(r[1] - r[4]), np.float32(r[2] - r[5])
        want = np.float32(np.float32(np.float32(dx * dx) + np.float32(dy * dy)) + np.float32(dz * dz))
        assert np.float32(got).view(np.uint32) == want.view(np.uint32)
        fused = np.float32(float(dz) * float(dz) + float(np.float32(np.float32(dx * dx) + np.float32(dy * dy))))
        diff += int(fused.view(np.uint32) != want.view(np.uint32))
    assert diff > 0   # the distinction is real on this sample


# ---------------------------------------------------------------- F3: hw2 f64 k-NN
def _check_knn(idx, dist, ref_idx, ref_dist, db, q):
    # k distances bit-equal as a sorted list
    assert np.array_equal(bits64(dist), bits64(ref_dist))
    m, k = idx.shape
    for i in range(m):
        if np.array_equal(idx[i], ref_idx[i]):
            continue
        # differences only inside groups of equal distance
        e = db - q[i]
        full = np.sqrt((e[:, 0] * e[:, 0] + e[:, 1] * e[:, 1]) + e[:, 2] * e[:, 2])
        for s in range(k):
            if idx[i, s] != ref_idx[i, s]:
                assert bits64(full[idx[i, s]:idx[i, s] + 1])[0] == bits64(dist[i, s:s + 1])[0]
                assert bits64(full[ref_idx[i, s]:ref_idx[i, s] + 1])[0] == bits64(dist[i, s:s + 1])[0]
        # canonical order: ties by ascending index
        for s in range(1, k):
            if dist[i, s] == dist[i, s - 1]:
                assert idx[i, s] > idx[i, s - 1]


@pytest.mark.parametrize("case", ["synth1000", "kitti4096", "lattice"])
@pytest.mark.parametrize("k", [1, 8])
def test_knn_oracle_vs_hw2_kdtree(orc, golden, case, k):
    g = golden(f"knn_hw2_{case}.npz")
    idx, dist = orc.knn_f64(g["db"], g["q"], k)
    _check_knn(idx, dist, g[f"idx_k{k}"], g[f"dist_k{k}"], g["db"], g["q"])
    # nanoflann (squared, f64) agrees on the distances through sqrt-free comparison of the index sets
    if case != "lattice":
        assert np.array_equal(idx.astype(np.uint64), g[f"nano_idx_k{k}"])


# ---------------------------------------------------------------- F4: radius sets
@pytest.mark.parametrize("case", ["synth1000", "kitti4096"])
@pytest.mark.parametrize("r", [0.5, 1.0])
def test_radius_oracle_vs_hw2_kdtree(orc, golden, case, r):
    g = golden(f"radius_hw2_{case}.npz")
    tag = str(r).replace(".", "p")
    row, idx, dist = orc.radius_f64(g["db"], g["q"], r)
    assert np.array_equal(row, g[f"row_r{tag}"])
    assert np.array_equal(idx, g[f"idx_r{tag}"])
    assert np.array_equal(bits64(dist), bits64(g[f"dist_r{tag}"]))


def test_radius_f32_variant_matches_f64_membership_away_from_boundary(orc, golden):
    g = golden("radius_hw2_synth1000.npz")
    row64, idx64, dist64 = orc.radius_f64(g["db"], g["q"], 1.0)
    row32, idx32, dist32 = orc.radius_f32(g["db"].astype(np.float32), g["q"].astype(np.float32), 1.0)
    # identical unless a neighbour sits within f32 rounding of the radius
    if np.all(np.abs(dist64 - 1.0) > 1e-5):
        assert np.array_equal(row64, row32) and np.array_equal(idx64, idx32)


# ---------------------------------------------------------------- A3 / A4 result-set tie rules
def test_result_set_tie_rules(orc):
    import ctypes as C
    # hw2: ties -> last visited wins (resultSet.hpp:69 rejects only dist > worst)
    dist = np.full(1, 1e10); index = np.zeros(1, np.int32)
    count = C.c_int(0); worst = C.c_double(1e10)
    for i in (7, 3, 9):
        orc.lib().orc_hw2_knn_add(dist, index, 1, C.byref(count), C.byref(worst), 2.0, i)
    assert index[0] == 9
    # nanoflann: ties -> first visited wins (strict > at :184, gate < at :1360)
    d = np.full(1, np.finfo(np.float32).max, np.float32); ind = np.zeros(1, np.uint64)
    cnt = C.c_size_t(0)
    for i in (7, 3, 9):
        orc.lib().orc_nano_knn_add(d, ind, 1, C.byref(cnt), 2.0, i)
    assert ind[0] == 7
    # sorted insertion, capacity 3 (Homework2/hw2/script/result_set.py:92-101 style smoke)
    dist = np.full(3, 1e10); index = np.zeros(3, np.int32)
    count = C.c_int(0); worst = C.c_double(1e10)
    for i, v in enumerate([5.0, 1.0, 3.0, 0.5, 4.0]):
        orc.lib().orc_hw2_knn_add(dist, index, 3, C.byref(count), C.byref(worst), v, i)
    assert dist.tolist() == [0.5, 1.0, 3.0] and index.tolist() == [3, 1, 2]


# ---------------------------------------------------------------- F5: HW4 plane-inlier count
def test_plane_count_oracle_vs_hw4_numpy(orc, golden):
    g = golden("plane_hw4.npz")
    pts = np.ascontiguousarray(g["pts_f32"].T)
    counts = orc.plane_count(pts, g["params"], float(g["thr"]))
    assert np.array_equal(counts, g["counts"])
    masks = np.unpackbits(g["masks"], axis=1)[:, : pts.shape[1]]
    for h in (0, 7, 15):
        assert np.array_equal(orc.plane_mask(pts, g["params"][h], float(g["thr"])), masks[h])
    assert counts.max() > 1000          # the hypotheses do find the ground
    # f64 point path (after pcd_preprocessing the reference holds f64 points, ground_detection_SVD.py:35)
    counts64 = orc.plane_count(pts.astype(np.float64), g["params"], float(g["thr"]))
    assert np.array_equal(counts64, g["counts"])


def test_plane_from_3pts_matches_reference_params(orc, golden):
    g = golden("plane_hw4.npz")
    pts = g["pts_f32"].astype(np.float64)
    for h in range(16):
        p = orc.plane_from_3pts(pts[g["picked"][h]])
        assert np.array_equal(bits64(p), bits64(g["params"][h]))


# ---------------------------------------------------------------- A7: SVD / Kabsch
def test_svd3_properties(orc):
    rng = np.random.default_rng(1)
    for trial in range(200):
        A = rng.normal(size=(3, 3)) * 10.0 ** rng.integers(-3, 4)
        if trial % 10 == 0:
            A[:, 2] = A[:, 0] * 2.0          # rank deficient
        if trial % 25 == 0:
            A = np.outer(A[:, 0], A[0])      # rank 1
        U, S, V = orc.svd3(A)
        assert np.allclose(U @ np.diag(S) @ V.T, A, atol=1e-12 * max(1.0, np.abs(A).max()))
        assert np.allclose(U.T @ U, np.eye(3), atol=1e-12)
        assert np.allclose(V.T @ V, np.eye(3), atol=1e-12)
        assert S[0] >= S[1] >= S[2] >= 0
        assert np.allclose(S, np.linalg.svd(A, compute_uv=False), atol=1e-12 * max(1.0, S[0]))


def test_kabsch_recovers_known_rigid_motion(orc, synth):
    tgt = synth.kitti_like_scan(2000)
    T = synth.gt_pose()
    src = (T[:3, :3].T @ (tgt.astype(np.float64) - T[:3, 3:4])).astype(np.float32)
    idx = np.arange(2000, dtype=np.uint32)
    d2 = np.zeros(2000, np.float32)
    sums, last = orc.kabsch_accumulate(src, tgt, idx, d2, 1.0)
    assert last == 1999 and sums[15] == 2000
    rc, R, t = orc.kabsch_solve(sums)
    assert rc == 0
    assert np.linalg.norm(R - T[:3, :3]) < 1e-5 and np.linalg.norm(t - T[:3, 3]) < 1e-4
    # no kept pair -> flagged (the reference would divide by zero, registration.cpp:979)
    sums0, last0 = orc.kabsch_accumulate(src, tgt, idx, d2 + 5.0, 1.0)
    assert last0 == -1 and orc.kabsch_solve(sums0)[0] == -1


def test_kabsch_reflection_branch_follows_reference(orc):
    # H with det(U V^T) < 0: the reference then builds R = V * diag(1,1,det) * U^T (registration.cpp:990-996, sic)
    rng = np.random.default_rng(5)
    P = rng.normal(size=(3, 50))
    Q = P.copy(); Q[2] *= -1.0                     # mirrored target
    sums = np.zeros(16)
    sums[0:3] = P.sum(1); sums[3:6] = Q.sum(1); sums[6:15] = (Q @ P.T).reshape(9); sums[15] = 50
    rc, R, t = orc.kabsch_solve(sums)
    assert rc == 0 and abs(np.linalg.det(R.astype(np.float64)) - 1.0) < 1e-5
    Pc = P - P.mean(1, keepdims=True); Qc = Q - Q.mean(1, keepdims=True)
    U, S, Vt = np.linalg.svd(Qc @ Pc.T)
    Rd = U @ Vt
    want = Vt.T @ np.diag([1, 1, np.linalg.det(Rd)]) @ U.T
    assert np.allclose(R, want, atol=1e-5)


# ---------------------------------------------------------------- F6: ICP (self-golden)
def test_icp_oracle_reproduces_selfgolden(orc, golden):
    g = golden("icp_selfgolden.npz")
    T, st, per_T, per_n = orc.icp_p2p_f32(g["src"], g["tgt"], max_corr=1.0, max_iter=20, eps=1e-8, trace=True)
    assert st["iters_run"] == int(g["iters_run"]) and st["converged"] == int(g["converged"])
    assert np.array_equal(per_n, g["per_iter_pairs"])
    assert np.linalg.norm(T.astype(np.float64) - g["T"]) < 1e-6


def test_icp_state_machine_quirks(orc, synth):
    src, tgt = synth.kitti_like_pair(600)
    # eps huge -> `unchanged` increments every iteration and is never reset -> break at iter index 15
    # (registration.cpp:948-958): 15 updates applied, converged flag set
    T, st = orc.icp_p2p_f32(src, tgt, max_iter=40, eps=1e30)
    assert st["converged"] == 1 and st["iters_run"] == 15
    # eps = 0 never counts (strict <) -> runs max_iter
    T, st = orc.icp_p2p_f32(src, tgt, max_iter=18, eps=0.0)
    assert st["converged"] == 0 and st["iters_run"] == 18
    # max_corr compares the SQUARED distance with the un-squared parameter (:936)
    idx, d2 = orc.nn1_f32(tgt, src)
    sums, _ = orc.kabsch_accumulate(src, tgt, idx, d2, 0.25)
    assert sums[15] == np.sum(d2 < 0.25)
    # no pair at all -> flagged, pose unchanged
    T, st = orc.icp_p2p_f32(src + np.float32(1000.0), tgt, max_iter=5)
    assert st["empty_pairs"] == 1 and np.array_equal(T, np.eye(4, dtype=np.float32))


def test_icp_converges_to_ground_truth_on_dense_pair(orc, synth):
    # oracle is O(N^2): keep it small but dense enough (scan of 6000 pts, 6 iterations move towards GT)
    src, tgt = synth.kitti_like_pair(6000)
    T0 = np.eye(4)
    T, st = orc.icp_p2p_f32(src, tgt, max_iter=6)
    gt = synth.gt_pose()
    assert np.linalg.norm(T - gt) < np.linalg.norm(T0 - gt)


# ---------------------------------------------------------------- reference harness live (when built)
def test_live_reference_agrees_with_oracle_when_present(orc, synth):
    if not orc.have_ref():
        pytest.skip("oracle/_ref not built (reference absent)")
    src, tgt = synth.kitti_like_pair(3000, seed_target=99, seed_pair=98)
    i1, d1, _, _ = orc.ref_nano_nn1_f32(tgt, src, leaf=2)
    i2, d2 = orc.nn1_f32(tgt, src)
    assert np.array_equal(i1, i2) and np.array_equal(bits32(d1), bits32(d2))
    i3, d3, _, _ = orc.ref_nano_nn1_f32(tgt, src, leaf=2, threads=4)
    assert np.array_equal(i1, i3)


def random_cloud(rng, n, kind):
    """continuous / lattice (ties, duplicates) / clustered / huge and tiny magnitudes"""
    if kind == 0:
        return rng.normal(0, 10, (n, 3))
    if kind == 1:
        return rng.integers(0, 6, (n, 3)).astype(np.float64) * 0.5
    if kind == 2:
        c = rng.normal(0, 30, (max(n // 8, 1), 3))
        return c[rng.integers(0, c.shape[0], n)] + rng.normal(0, 0.01, (n, 3))
    s = 10.0 ** rng.integers(-12, 12)
    return rng.normal(0, 1, (n, 3)) * s


def test_live_reference_randomised_sweep(orc):
    """200 random small problems against the compiled reference: nanoflann f32 1-NN (d2 bits; index equal or inside the
    tie set), hw2 kd-tree k-NN distances and radius sets — continuous, lattice, clustered and extreme-magnitude clouds."""
    if not orc.have_ref():
        pytest.skip("oracle/_ref not built (reference absent)")
    rng = np.random.default_rng(20240607)
    ties_seen = 0
    for trial in range(200):
        kind = trial % 4
        n, m = int(rng.integers(1, 300)), int(rng.integers(1, 40))
        db, q = random_cloud(rng, n, kind), random_cloud(rng, m, kind)
        if trial % 3 == 0:
            q[: min(m, n) // 2] = db[: min(m, n) // 2]                        # exact hits
        # --- A1/A3/A6: nanoflann f32, leaf 2
        t32, s32 = np.ascontiguousarray(db.T.astype(np.float32)), np.ascontiguousarray(q.T.astype(np.float32))
        ri, rd, _, _ = orc.ref_nano_nn1_f32(t32, s32, leaf=2)
        oi, od = orc.nn1_f32(t32, s32)
        assert np.array_equal(bits32(rd), bits32(od)), trial
        for k in np.flatnonzero(ri != oi):
            assert oi[k] < ri[k] and orc.lib().orc_d2_f32(*map(float, s32[:, k]), *map(float, t32[:, ri[k]])) == od[k]
            ties_seen += 1
        # --- A2/A4: hw2 kd-tree k-NN (lattice clouds can make its median split recurse forever: leaf >= 32 there, see DESIGN)
        if np.unique(db, axis=0).shape[0] == n or kind != 1:
            k = int(rng.integers(1, 9))
            leaf = 32 if kind == 1 else 1
            hi, hd = orc.ref_hw2_kd_knn(db, q, k, leaf=leaf)
            ki, kd = orc.knn_f64(db, q, k)
            assert np.array_equal(hd.view(np.uint64), kd.view(np.uint64)), trial
            # --- A11: radius sets
            r = float(np.median(kd[:, -1])) if n >= k else 1.0
            row, ridx, rdist = orc.ref_hw2_kd_radius(db, q, r, leaf=leaf)
            orow, oidx, odist = orc.radius_f64(db, q, r)
            assert np.array_equal(row, orow)
            for i in range(m):
                o = np.argsort(ridx[row[i]:row[i + 1]], kind="stable")
                assert np.array_equal(ridx[row[i]:row[i + 1]][o], oidx[row[i]:row[i + 1]])
    assert ties_seen > 0


def test_kat_query5_open3d_squared_distances(orc, golden):
    """The reference's second known answer: open3d (FLANN) on the same query prints SQUARED distances
    (Homework2/hw2/result_py.txt:34-35, result_cpp.txt:46-53) — the contract of the `squared` k-NN entry points."""
    g = golden("kat_kitti_q5.npz")
    pts = np.ascontiguousarray(g["db_f32"].T)
    q = np.ascontiguousarray(pts[:, 5:6])
    idx, s, found = orc.knn_sq_f32pts(pts, q, 8)
    assert idx[0].tolist() == [5, 1972, 6, 1971, 1970, 3946, 8, 3945]
    printed = [0, 0.347574, 1.50463, 1.95563, 2.14362, 2.48257, 2.67194, 2.75159]
    assert np.allclose(s[0], printed, rtol=2e-6, atol=0) and found[0] == 8

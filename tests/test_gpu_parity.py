"""GPU parity tests (-m gpu): the HIP path, called through the C ABI of libpcr_hip.so, against
(1) the golden fixtures generated from the reference itself, (2) the CPU oracle on seeded inputs,
(3) size-independent properties at BASELINE sizes.  Bar: bit-exact for indices / distances / counts,
1e-5 Frobenius for the ICP pose (BASELINE.json north_star)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits32(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def bits64(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


@pytest.fixture(scope="module")
def ctx(pcr):
    c = pcr.Context(0)
    yield c
    c.close()


def random_cloud32(rng, n, kind):
    """continuous / lattice (ties, duplicates) / clustered / extreme magnitudes — SoA f32"""
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


def test_bf16_matrix_core_arithmetic_is_within_the_bound_the_filter_assumes(ctx):
    """BTRACK's lower bound (csrc/nn1_brute.hip) assumes: bf16 x bf16 products exact in f32 and an accumulation error of the two
    v_mfma_f32_32x32x16_bf16 of at most 16 x 2^-24 x sum |a b|; its three-piece operand layout then reproduces w - 2 r.t to within
    34.2 x 2^-24 (|r|^2 + |t|^2).  Measured here on the device under test (the library's own MFMA pair, adversarial exponents, and the
    structured tiles: cancellation across K-slots, alternating signs, smallest pieces, maximal exponent spread), with a factor two of
    head-room on every figure — the same pass marks the library applies to itself before it first picks the kernel (mfma_verdict)."""
    acc, filt, small, struct = ctx.selftest_mfma_bf16(96)
    assert 0.0 < acc <= 8.0, acc
    assert 0.0 <= struct <= 8.0, struct
    assert 0.0 < filt <= 17.0, filt
    assert 0.0 <= small <= 2.0, small
    # the f16 form (HTRACK, the default where the cloud fits f16's range): accumulation <= 16 assumed, filter value <= 82 assumed, and
    # in the regime where the second f16 piece underflows (|r|, |t| both below 2^-3 after scaling) an absolute 4 x 2^-24
    acc16, filt16, small16, struct16 = ctx.selftest_mfma_f16(96)
    assert 0.0 < acc16 <= 8.0, acc16
    assert 0.0 <= struct16 <= 8.0, struct16
    assert 0.0 < filt16 <= 41.0, filt16
    assert 0.0 < small16 <= 2.0, small16


def test_sign_form_never_misses_a_pair_at_or_below_its_threshold(ctx):
    """STRACK's decision, checked on the device under test (pcr_selftest_sign_f16): through the kernel's own operand code and MFMA, every
    pair whose exact f32 distance lies at or below its query's threshold — thresholds ON a pair's distance, one ulp below / above it and
    a factor away; search magnitudes and the f16 underflow regimes; a power-of-two scale per tile — comes out with its sign set.  How
    many signs are set needlessly is the price of the bound's slack (reported, bounded loosely)."""
    le, missed, flagged, total = ctx.selftest_sign_f16(1024)
    assert total == (1024 + 8) * 1024 and le >= 1024 * 32 * 0.7           # (one ulp below: the pair itself need not be flagged)
    assert missed == 0, (le, missed, flagged, total)
    assert flagged >= le and flagged <= le + 0.02 * total, (le, flagged, total)
    assert ctx.selftest_sign_f16(2)[1] == 0                                # the short form the verdict runs


def test_sphere_form_never_misses_a_chunk_with_a_record_at_or_below_the_threshold(ctx):
    """The sphere rows of STRACK3 (csrc/nn1_sphere.hpp, grid_common.hpp l1_chunk_operand / st_setup_l1): the chunk-sphere form of the sign filter through
    the index build's operand code, the kernel's query code and the MFMA, on the device under test — no (query, chunk) pair with a record at or
    below the query's threshold comes out without its sign; the form prunes (most pairs are not flagged)."""
    must, missed, flagged, pairs = ctx.selftest_sphere_f16(512)
    assert pairs == 512 * 32 * 32 and must > 20000, (must, pairs)
    assert missed == 0, (must, missed)
    assert flagged < 0.6 * pairs, (flagged, pairs)


def test_sphere_forms_over_several_level0_supertiles_equal_the_exact_grid(ctx, synth):
    """A target of 300 000 points spans three level-0 super-tiles of STRACK3 (131 072 records each) and 74 level-1 super-tiles: cold and seeded
    searches, sliced every way (one level-0 super-tile per slice / all in one; every group size), return the keys of the exact grid search bit for bit — on the scan pair, on queries far outside the target's box, and with non-finite queries among them."""
    n, nq = 300_000, 24_000
    src_all, tgt = synth.kitti_like_pair(n, seed_target=811, seed_pair=812)
    src = np.ascontiguousarray(src_all[:, :: n // nq][:, :nq]).copy()
    src[:, 17] = np.nan; src[1, 18] = np.inf
    src[:, 100:140] += np.float32(500.0)                              # far outside: every level must still answer
    src[:, 200:230] *= np.float32(1.0e-3)                             # a cluster at the sensor
    ct = ctx.cloud(tgt)
    clouds = [ctx.cloud(src), ctx.cloud(np.ascontiguousarray(src + np.array([[0.04], [0.02], [-0.01]], np.float32)))]
    ctx.tune("nn_method", 2)
    ref = [ctx.nn1(ct, c_) for c_ in clouds]
    ctx.tune("nn_method", 1)
    for sw in (dict(nn1_variant=10), dict(nn1_variant=10, nn1_sphere_l0_per_slice=1), dict(nn1_variant=10, nn1_sphere_l0_per_slice=2, nn1_sphere_qg=2),
               dict(nn1_variant=10, nn1_sphere_l0_per_slice=3, nn1_sphere_qg=4, nn1_sign_flush=1), dict()):
        for k, v in sw.items():
            ctx.tune(k, v)
        ctx.tune("nn1_async_in_loop", 1)
        fresh = ctx.cloud(tgt)                                       # the first search is cold: it seeds itself; the later ones start from the previous keys
        for k, c_ in enumerate(clouds + clouds):
            ctx.nn1_async(fresh, c_)
            if sw:
                assert ctx.mfma_check()["last_nn1_kernel"] == "strack3", (sw, k)
            idx, d2 = ctx.nn1_fetch(nq)
            ri, rd = ref[k % 2]
            assert np.array_equal(idx, ri) and np.array_equal(bits32(d2), bits32(rd)), (sw, k, int((idx != ri).sum()))
        ctx.tune("nn1_async_in_loop", 0)
        fresh.free()
        for k in sw:
            ctx.tune(k, 0)
    ctx.tune("nn_method", 0)
    for c_ in clouds:
        c_.free()
    ct.free()


def test_library_checks_the_matrix_core_arithmetic_itself_and_falls_back(ctx, orc, synth):
    """The dispatcher consults a once-per-context verdict before it first uses a matrix-core kernel; a failing verdict (forced here
    with the tune key) moves the search to the next form — f16 -> bf16 -> the f32 filters — with the same bits out."""
    src, tgt = synth.kitti_like_pair(9000, seed_target=91, seed_pair=92)
    oi, od = orc.nn1_f32(tgt, src)
    ctx.tune("nn_method", 1)
    chk = ctx.mfma_check(run_now=True)
    assert chk["f16_ok"] == 1 and chk["bf16_ok"] == 1, chk
    assert 0.0 < chk["check_ms"] < 20.0, chk                # both forms, once per context (measured: profiles/)
    assert ctx.mfma_check(run_now=True)["check_ms"] == chk["check_ms"]         # cached: not run again
    want = {0: ("strack", "htrack"), 1: ("btrack",), 2: ("strack", "htrack"), 3: ("ftrack", "etrack")}
    for force in (0, 1, 2, 3, 3):
        ctx.tune("mfma_force_fail", force)
        ct, cs = ctx.cloud(tgt), ctx.cloud(src)
        idx, d2 = ctx.nn1(ct, cs)
        got = ctx.mfma_check()
        assert got["last_nn1_kernel"] in want[force], (force, got)
        assert got["f16_ok"] == (0 if force & 1 else 1) and got["bf16_ok"] == (0 if force & 2 else 1)
        assert np.array_equal(idx, oi) and np.array_equal(bits32(d2), bits32(od)), force
        # and inside a loop (seeded searches): the fallback of a failed verdict there is the f32 filter on the chunked index
        T, st = ctx.icp_point2point(cs, ct, max_corr=1.0, max_iter=3, eps=1e-8)
        if force == 3:
            assert ctx.mfma_check()["last_nn1_kernel"] == "etrack"
        if force == 0:
            T0 = T
        assert np.array_equal(T, T0), force
        ct.free(); cs.free()
    ctx.tune("mfma_force_fail", 0)
    ctx.tune("nn_method", 0)


def _ulp_step(a, k):
    """a stepped by k units in the last place (per element)"""
    b = np.ascontiguousarray(a, np.float32).copy()
    i = b.view(np.int32)
    i += np.where(b >= 0, k, -k).astype(np.int32)
    return b


@pytest.mark.parametrize("sps", [0, 1])
def test_nn1_near_duplicates_next_to_a_super_tile_centre(ctx, orc, sps):
    """with the query ON a target (d2 = 0) and |r|, |t''| of 1e-3 .. 3e-2 after HTRACK's per-super-tile scaling, the
    second f16 piece of an operand underflows and the filter value misses by an absolute 2^-24 — more than the relative slack
    2^-17 (Q + W) provides there; near-duplicate targets a few ulps apart in other chunks / slices could then hide the true
    neighbour.  Clusters of 256 points = one super-tile each: 108 mirrored pairs within 1 m of the centre (so the centre of the
    super-tile is the cluster centre to a few ulps and the scale is 2^7) and 40 near-duplicates within 8 ulps of the centre in every
    coordinate; queries ON the near-duplicates and 1-2 ulps off them.  Every kernel against the oracle, bit for bit."""
    rng = np.random.default_rng(2026)
    M = 48
    cen = rng.uniform(-60, 60, (3, M)).astype(np.float32)
    tg, qs = [], []
    for j in range(M):
        c = cen[:, j:j + 1]
        v = rng.uniform(-1, 1, (3, 108)).astype(np.float32)
        dup = np.concatenate([_ulp_step(np.repeat(c, 40, 1)[k:k + 1], rng.integers(-8, 9, 40)) for k in range(3)], 0)
        tg += [c + v, c - v, dup]
        qs += [dup, np.concatenate([_ulp_step(dup[k:k + 1], rng.integers(-2, 3, 40)) for k in range(3)], 0), c + v[:, :8] * 0.5]
    tgt = np.ascontiguousarray(np.concatenate(tg, 1), np.float32)
    src = np.ascontiguousarray(np.concatenate(qs, 1), np.float32)
    perm = rng.permutation(tgt.shape[1])
    tgt = np.ascontiguousarray(tgt[:, perm])
    oi, od = orc.nn1_f32(tgt, src)
    assert (od == 0).sum() >= 40 * M                      # the coincident queries
    ctx.tune("nn_method", 1)
    ctx.tune("nn1_supers_per_slice", sps)                 # 1: every cluster its own slice (settled through the published bound)
    for variant in (2, 7, 6, 8, 0):
        ctx.tune("nn1_variant", variant)
        ct, cs = ctx.cloud(tgt), ctx.cloud(src)
        idx, d2 = ctx.nn1(ct, cs)
        assert np.array_equal(idx, oi) and np.array_equal(bits32(d2), bits32(od)), (variant, sps, int((idx != oi).sum()))
        ctx.tune("nn1_async_in_loop", 1)                  # and seeded by its own previous answer / by a stale one
        for _ in range(2):
            ctx.nn1_async(ct, cs)
            idx, d2 = ctx.nn1_fetch(src.shape[1])
            assert np.array_equal(idx, oi) and np.array_equal(bits32(d2), bits32(od)), (variant, sps, "warm")
        ctx.tune("nn1_async_in_loop", 0)
        ct.free(); cs.free()
    ctx.tune("nn1_variant", 0); ctx.tune("nn1_supers_per_slice", 0); ctx.tune("nn_method", 0)


def test_sign_filter_bad_seeds_nonfinite_queries_and_full_lists(ctx, orc, synth):
    """STRACK (csrc/nn1_brute.hip: the sign form of the f16 filter) decides from a candidate per query.  Here the candidates are as bad
    as they get — the correspondences of a DIFFERENT pose (metres away, rotated), a cold search's own seeds —, some queries are NaN /
    inf / 10^6 m away, the target holds a lattice (ties) and exact duplicates, and the wave's list of flagged chunks is flushed after
    every super-tile or only when it is full (it overflows with seeds this bad): every answer equals the exact-only kernel's, and the
    oracle's for the first pose."""
    n = 20000
    src, tgt = synth.kitti_like_pair(n, seed_target=811, seed_pair=812)
    tgt = tgt.copy(); src = src.copy()
    tgt[:, :2000] = np.round(tgt[:, :2000] * 4) / 4                  # ties
    tgt[:, 2000:2500] = tgt[:, 1500:2000]                            # exact duplicates: the lowest index must win
    src[:, :500] = tgt[:, 1700:2200]                                 # queries ON targets
    src[:, 5] = np.nan; src[0, 77] = np.inf; src[2, 78] = -np.inf; src[:, 100:110] = 1.0e6
    c, s_ = np.float32(np.cos(0.7)), np.float32(np.sin(0.7))
    Rz = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]], np.float32)
    with np.errstate(invalid="ignore"):                             # (the NaN / inf queries go through the rotation too)
        poses = [src, (src + np.array([[0.05], [-0.03], [0.01]], np.float32)).astype(np.float32), (Rz @ src + np.array([[4.0], [-7.0], [0.5]], np.float32)).astype(np.float32),
                 src, (src * np.float32(0.5)).astype(np.float32)]
    poses = [np.ascontiguousarray(p_, np.float32) for p_ in poses]
    ctx.tune("nn_method", 1)
    ct = ctx.cloud(tgt)
    clouds = [ctx.cloud(p_) for p_ in poses]
    ctx.tune("nn1_variant", 2)
    ref = [ctx.nn1(ct, c_) for c_ in clouds]
    oi, od = orc.nn1_f32(tgt, poses[0])
    assert np.array_equal(ref[0][0], oi) and np.array_equal(bits32(ref[0][1]), bits32(od))
    # (variant 8: STRACK; 10: STRACK3, the sign filter over three levels of bounding spheres — qg then is its number of query groups per wave, sps the
    # entries from which the end of a level-1 super-tile evaluates them)
    for qg, flush, sps, variant in ((4, 0, 0, 8), (2, 1, 0, 8), (4, 100000, 1, 8), (2, 0, 3, 8), (4, 1, 200, 8), (4, 3, 0, 8),
                                    (1, 0, 0, 10), (1, 1, 0, 10), (1, 100000, 64, 10), (2, 0, 0, 10), (2, 3, 16, 10), (4, 0, 0, 10), (4, 100000, 128, 10), (4, 1, 1, 10)):
        ctx.tune("nn1_btrack_qg", qg if variant != 10 else 0); ctx.tune("nn1_sign_flush", flush); ctx.tune("nn1_supers_per_slice", sps if variant == 8 else 0)
        ctx.tune("nn1_sphere_qg", qg if variant == 10 else 0); ctx.tune("nn1_sphere_flush_end", sps if variant == 10 else 0)
        ctx.tune("nn1_variant", variant)
        ctx.tune("nn1_async_in_loop", 1)
        fresh = ctx.cloud(tgt)                                       # the first search is cold: it seeds itself
        for k, c_ in enumerate(clouds + clouds[:2]):
            ctx.nn1_async(fresh, c_)
            assert ctx.mfma_check()["last_nn1_kernel"] == {8: "strack", 10: "strack3"}[variant], (qg, flush, sps, variant, k)
            idx, d2 = ctx.nn1_fetch(n)
            ri, rd = ref[k % len(clouds)]
            assert np.array_equal(idx, ri) and np.array_equal(bits32(d2), bits32(rd)), (qg, flush, sps, k, int((idx != ri).sum()))
        ctx.tune("nn1_async_in_loop", 0)
        fresh.free()
    for k_ in ("nn1_btrack_qg", "nn1_sign_flush", "nn1_supers_per_slice", "nn1_sphere_qg", "nn1_sphere_flush_end", "nn1_variant", "nn_method"):
        ctx.tune(k_, 0)
    for c_ in clouds:
        c_.free()
    ct.free()


def test_nn1_randomised_sweep_every_kernel(ctx, orc):
    """120 random problems x {FTRACK, TRACK, ETRACK, BTRACK (matrix cores), exact grid (plain / x-window / bounding-sphere kernels on the
    x-sorted index, plain / bounding-sphere kernels on the Morton-ordered index)}: indices and d2 bits equal to the oracle."""
    rng = np.random.default_rng(77)
    for trial in range(120):
        kind = trial % 4
        nt, ns = int(rng.integers(1, 5000)), int(rng.integers(1, 700))
        tgt, src = random_cloud32(rng, nt, kind), random_cloud32(rng, ns, kind)
        if trial % 3 == 0:
            src[:, : min(ns, nt) // 2] = tgt[:, : min(ns, nt) // 2]
        oi, od = orc.nn1_f32(tgt, src)
        ct, cs = ctx.cloud(tgt), ctx.cloud(src)
        for method, variant, mode in ((1, 1, 0), (1, 2, 0), (1, 4, 0), (1, 6, 0), (1, 7, 0), (1, 8, 0), (2, 1, 1), (2, 1, 2), (2, 1, 3)):
            ctx.tune("nn_method", method)
            ctx.tune("nn1_variant", variant)
            ctx.tune("grid_mode", mode)                # 1 plain, 2 x-window, 3 bounding spheres (0: by target size)
            idx, d2 = ctx.nn1(ct, cs)
            assert np.array_equal(idx, oi) and np.array_equal(bits32(d2), bits32(od)), (trial, kind, nt, ns, method, variant, mode)
        ct.free()
        ctx.tune("grid_order", 2)                      # the Morton-ordered index of large targets, forced on a fresh cloud
        cm = ctx.cloud(tgt)
        for method, variant, mode in ((2, 1, 3), (2, 1, 1), (1, 4, 0), (1, 6, 0), (1, 7, 0), (1, 8, 0)):
            ctx.tune("nn_method", method); ctx.tune("nn1_variant", variant); ctx.tune("grid_mode", mode)
            idx, d2 = ctx.nn1(cm, cs)
            assert np.array_equal(idx, oi) and np.array_equal(bits32(d2), bits32(od)), (trial, kind, nt, ns, method, variant, mode, "morton")
        ctx.tune("grid_order", 0)
        cs.free(); cm.free()
    ctx.tune("nn_method", 0)
    ctx.tune("nn1_variant", 0)
    ctx.tune("grid_mode", 0)


# ------------------------------------------------------------------ 1-NN (A1/A3/A6) vs reference goldens
@pytest.mark.parametrize("method", [1, 2])            # 1 = brute force, 2 = exact grid
@pytest.mark.parametrize("case", ["synth1000", "synth4096", "kitti4096", "lattice1000"])
def test_nn1_vs_nanoflann_golden(ctx, orc, golden, case, method):
    g = golden(f"nn1_nanoflann_{case}.npz")
    ctx.tune("nn_method", method)
    cs, ct = ctx.cloud(g["src"]), ctx.cloud(g["tgt"])
    idx, d2 = ctx.nn1(ct, cs)
    ctx.tune("nn_method", 0)
    assert np.array_equal(bits32(d2), bits32(g["d2"]))            # distance bit-equal to nanoflann's
    ties = orc.nn1_tiecount_f32(g["tgt"], g["src"])
    single = ties == 1
    assert np.array_equal(idx[single], g["idx"][single])          # index equal on singleton tie sets
    oidx, _ = orc.nn1_f32(g["tgt"], g["src"])
    assert np.array_equal(idx, oidx)                              # canonical rule everywhere (= oracle)
    cs.free(); ct.free()


# nn1_variant: 1 FTRACK (fused-filter tracking, exact decision; no index), 2 TRACK (the exact arithmetic for every pair: the on-device
# reference), 4 ETRACK (expanded-form f32 filter on the grid's chunked target copy), 6 BTRACK (the filter on the bf16 matrix cores,
# three-piece operands), 7 HTRACK (one f16 MFMA per tile, two-piece scaled operands), 8 STRACK (the sign form of the f16 filter for every
# search that has or can make itself a seed; HTRACK where none exists), 10 STRACK3 (the sign filter over three levels of bounding spheres: 512-record
# tiles, 16-record chunks, records — the default from 32 768 target points) — csrc/nn1_brute.hip, table above launch_nn1_brute; csrc/nn1_sphere.hpp
VARIANTS = [1, 2, 4, 6, 7, 8, 10]


def set_variant(ctx, v):
    ctx.tune("nn_method", 1)                  # brute force (auto would pick the grid for large targets)
    ctx.tune("nn1_variant", v)


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("qpl", [2, 4])
@pytest.mark.parametrize("ns,nt", [(1, 1), (63, 5), (257, 1023), (1000, 1025), (3001, 7000), (5000, 2049)])
def test_nn1_ragged_sizes_vs_oracle(ctx, orc, synth, qpl, ns, nt, variant):
    ctx.tune("nn1_btrack_qg", qpl)            # query groups of 32 per wave of the matrix-core kernels
    set_variant(ctx, variant)
    src, _ = synth.kitti_like_pair(max(ns, 64), seed_target=7 + ns, seed_pair=11 + nt)
    tgt = synth.kitti_like_scan(max(nt, 64), seed=13 + nt)
    src, tgt = np.ascontiguousarray(src[:, :ns]), np.ascontiguousarray(tgt[:, :nt])
    cs, ct = ctx.cloud(src), ctx.cloud(tgt)
    idx, d2 = ctx.nn1(ct, cs)
    oidx, od2 = orc.nn1_f32(tgt, src)
    assert np.array_equal(idx, oidx) and np.array_equal(bits32(d2), bits32(od2))
    cs.free(); ct.free()
    ctx.tune("nn1_btrack_qg", 0)
    ctx.tune("nn1_variant", 0)
    ctx.tune("nn_method", 0)


@pytest.mark.parametrize("tps", [1, 3, 1000])
def test_nn1_slice_merge_is_order_independent(ctx, orc, synth, tps):
    # 1 tile per slice -> many atomicMin merges; 1000 -> single slice, plain stores
    src, tgt = synth.kitti_like_pair(6000, seed_target=21, seed_pair=22)
    ctx.tune("nn1_tiles_per_slice", tps)
    ctx.tune("nn_method", 1)
    cs, ct = ctx.cloud(src), ctx.cloud(tgt)
    idx, d2 = ctx.nn1(ct, cs)
    ctx.tune("nn1_tiles_per_slice", 0)
    ctx.tune("nn_method", 0)
    oidx, od2 = orc.nn1_f32(tgt, src)
    assert np.array_equal(idx, oidx) and np.array_equal(bits32(d2), bits32(od2))
    cs.free(); ct.free()


@pytest.mark.parametrize("variant", VARIANTS)
def test_nn1_ties_and_duplicates_pick_lowest_index(ctx, orc, synth, variant):
    set_variant(ctx, variant)
    lat = synth.lattice_cloud(5000, 3, 10.0, seed=5, levels=10).astype(np.float32)
    q = synth.lattice_cloud(3000, 3, 10.0, seed=6, levels=10).astype(np.float32)
    tgt, src = np.ascontiguousarray(lat.T), np.ascontiguousarray(q.T)
    ctx.tune("nn1_tiles_per_slice", 1)
    cs, ct = ctx.cloud(src), ctx.cloud(tgt)
    idx, d2 = ctx.nn1(ct, cs)
    ctx.tune("nn1_tiles_per_slice", 0)
    oidx, od2 = orc.nn1_f32(tgt, src)
    ctx.tune("nn1_variant", 0)
    ctx.tune("nn_method", 0)
    assert (orc.nn1_tiecount_f32(tgt, src) > 1).sum() > 1000
    assert np.array_equal(idx, oidx) and np.array_equal(bits32(d2), bits32(od2))
    cs.free(); ct.free()


@pytest.mark.parametrize("variant", VARIANTS)
def test_nn1_edge_cases(ctx, orc, variant):
    # empty source, empty target, non-finite input, huge and tiny magnitudes (filter gate must stay conservative)
    set_variant(ctx, variant)
    big = np.array([[1e19, -1e19, 3e18, 1e-20, 0, 1e-23], [0, 1e19, 0, 0, 1e-20, 0], [0, 0, 0, 0, 0, 0]], np.float32)
    cb = ctx.cloud(big)
    idx, d2 = ctx.nn1(cb, cb)
    oidx, od2 = orc.nn1_f32(big, big)
    assert np.array_equal(idx, oidx) and np.array_equal(bits32(d2), bits32(od2))
    cb.free()
    tgt = np.array([[0, 1, 2], [0, 0, 0], [0, 0, 0]], np.float32)
    src = np.array([[0.4, 1.6, np.nan, np.inf], [0, 0, 0, 0], [0, 0, 0, 0]], np.float32)
    ct, cs = ctx.cloud(tgt), ctx.cloud(src)
    idx, d2 = ctx.nn1(ct, cs)
    oidx, od2 = orc.nn1_f32(tgt, src)
    assert idx.tolist() == oidx.tolist() == [0, 2, 0xFFFFFFFF, 0xFFFFFFFF]
    assert np.array_equal(bits32(d2), bits32(od2)) and np.isinf(d2[2]) and np.isinf(d2[3])
    e = ctx.cloud(np.zeros((3, 0), np.float32))
    idx, d2 = ctx.nn1(ct, e)
    assert idx.size == 0
    idx, d2 = ctx.nn1(e, cs)
    assert (idx == 0xFFFFFFFF).all() and np.isinf(d2).all()
    ctx.tune("nn1_variant", 0)
    ctx.tune("nn_method", 0)
    for c in (ct, cs, e):
        c.free()


# ------------------------------------------------------------------ exact grid NN (same contract as brute force)
def _grid_vs_oracle(ctx, orc, src, tgt, method=2, **tune):
    """the exact grid search against the oracle, twice per index flavour: what the library picks for this size, and the
    Morton-ordered index + bounding-sphere walk of large targets forced onto it"""
    oidx, od2 = orc.nn1_f32(tgt, src)
    for extra in ({}, {"grid_order": 2, "grid_mode": 3}):
        if "grid_lanes" in tune and extra:
            continue                                     # the sphere walk is fixed at 16 lanes per query
        tn = dict(tune, **extra)
        ctx.tune("nn_method", method)
        for k, v in tn.items():
            ctx.tune(k, v)
        cs, ct = ctx.cloud(src), ctx.cloud(tgt)
        idx, d2 = ctx.nn1(ct, cs)
        idx2, d22 = ctx.nn1(ct, cs)                      # second call reuses the cached index
        for k in tn:
            ctx.tune(k, 0)
        ctx.tune("nn_method", 0)
        cs.free(); ct.free()
        assert np.array_equal(idx, oidx) and np.array_equal(bits32(d2), bits32(od2)), extra
        assert np.array_equal(idx2, oidx) and np.array_equal(bits32(d22), bits32(od2)), extra


@pytest.mark.parametrize("ns,nt", [(1, 1), (63, 5), (257, 1023), (1000, 1025), (3001, 7000), (5000, 2049)])
@pytest.mark.parametrize("sort_q,lanes", [(1, 8), (-1, 8), (1, 1), (1, 4), (1, 16), (1, 64)])
def test_grid_ragged_sizes_vs_oracle(ctx, orc, synth, ns, nt, sort_q, lanes):
    src, _ = synth.kitti_like_pair(max(ns, 64), seed_target=7 + ns, seed_pair=11 + nt)
    tgt = synth.kitti_like_scan(max(nt, 64), seed=13 + nt)
    _grid_vs_oracle(ctx, orc, np.ascontiguousarray(src[:, :ns]), np.ascontiguousarray(tgt[:, :nt]),
                    grid_sort_queries=sort_q, grid_lanes=lanes)


@pytest.mark.parametrize("cell_um", [20000, 250000, 3000000, 50000000])   # 2 cm ... 50 m cells
def test_grid_any_cell_size_is_exact(ctx, orc, synth, cell_um):
    src, tgt = synth.kitti_like_pair(5000, seed_target=41, seed_pair=42)
    _grid_vs_oracle(ctx, orc, src, tgt, grid_cell_um=cell_um)


@pytest.mark.parametrize("method", [2])
def test_grid_ties_duplicates_and_far_queries(ctx, orc, synth, method):
    lat = synth.lattice_cloud(6000, 3, 10.0, seed=5, levels=10).astype(np.float32)
    q = synth.lattice_cloud(3000, 3, 10.0, seed=6, levels=10).astype(np.float32)
    tgt, src = np.ascontiguousarray(lat.T), np.ascontiguousarray(q.T)
    assert (orc.nn1_tiecount_f32(tgt, src) > 1).sum() > 1000
    _grid_vs_oracle(ctx, orc, src, tgt, method=method)
    # queries far outside the target's bounding box (ring fast-forward), on its faces and in empty regions
    far = src.copy()
    far[0, :1000] += 500.0; far[1, 1000:2000] -= 73.5; far[2, 2000:] *= 40.0
    _grid_vs_oracle(ctx, orc, far, tgt, method=method)
    # degenerate targets: all identical / collinear / coplanar (zero-extent bounding boxes)
    same = np.repeat(np.array([[1.5], [2.5], [-3.0]], np.float32), 3000, axis=1)
    _grid_vs_oracle(ctx, orc, src, same, method=method)
    line = np.zeros((3, 3000), np.float32); line[0] = np.linspace(-5, 5, 3000, dtype=np.float32)
    _grid_vs_oracle(ctx, orc, src, line, method=method)
    plane = tgt.copy(); plane[2] = 0.25
    _grid_vs_oracle(ctx, orc, src, plane, method=method)


@pytest.mark.parametrize("method", [2])
def test_grid_non_finite_and_extreme_inputs(ctx, orc, synth, method):
    src, tgt = synth.kitti_like_pair(4000, seed_target=43, seed_pair=44)
    tgt = tgt.copy(); src = src.copy()
    tgt[0, 5] = np.nan; tgt[1, 6] = np.inf; tgt[2, 7] = -np.inf
    src[0, 0] = np.nan; src[1, 1] = np.inf; src[2, 2] = -np.inf; src[:, 3] = 3e30; src[:, 4] = -1e-30
    _grid_vs_oracle(ctx, orc, src, tgt, method=method)
    # a few huge outliers blow the bounding box up: cells get coarse, answers stay exact
    tgt2 = tgt.copy(); tgt2[:, 100] = [1e6, -1e6, 1e5]; tgt2[:, 101] = [-3e5, 2e5, 9e5]
    _grid_vs_oracle(ctx, orc, src, tgt2, method=method)


@pytest.mark.parametrize("method", [2])
def test_grid_index_invalidated_by_transform(ctx, orc, synth, method):
    src, tgt = synth.kitti_like_pair(4000, seed_target=45, seed_pair=46)
    ctx.tune("nn_method", method)
    cs, ct = ctx.cloud(src), ctx.cloud(tgt)
    ctx.nn1(ct, cs)                                   # builds the index on ct
    T = synth.gt_pose().astype(np.float32)
    ctx.transform(ct, T)                              # must drop it
    idx, d2 = ctx.nn1(ct, cs)
    ctx.tune("nn_method", 0)
    moved = orc.transform_f32(tgt, T[:3, :3], T[:3, 3])
    oidx, od2 = orc.nn1_f32(moved, src)
    assert np.array_equal(idx, oidx) and np.array_equal(bits32(d2), bits32(od2))
    cs.free(); ct.free()


def test_grid_equals_brute_force_at_120k(ctx, synth):
    # BASELINE config 2 size: both exact methods must agree bit for bit on every query (pre- and post-alignment)
    src, tgt = synth.kitti_like_pair(120000)
    cs, ct = ctx.cloud(src), ctx.cloud(tgt)
    ctx.tune("nn_method", 1); bi, bd = ctx.nn1(ct, cs)
    ctx.tune("grid_order", 2); cm = ctx.cloud(tgt); ctx.tune("nn_method", 2); ctx.nn1(cm, cs); ctx.tune("grid_order", 0)   # Morton-ordered twin
    for target, mode in ((ct, 1), (ct, 2), (ct, 3), (cm, 3), (cm, 1)):       # plain / x-window / spheres kernels on both record orders
        ctx.tune("nn_method", 2); ctx.tune("grid_mode", mode); gi, gd = ctx.nn1(target, cs)
        assert np.array_equal(bi, gi) and np.array_equal(bits32(bd), bits32(gd)), mode
    ctx.transform(cs, synth.gt_pose().astype(np.float32))
    ctx.tune("nn_method", 1); bi, bd = ctx.nn1(ct, cs)
    for target, mode in ((ct, 1), (ct, 2), (ct, 3), (cm, 3), (cm, 1)):
        ctx.tune("nn_method", 2); ctx.tune("grid_mode", mode); gi, gd = ctx.nn1(target, cs)
        assert np.array_equal(bi, gi) and np.array_equal(bits32(bd), bits32(gd)), mode
    ctx.tune("nn_method", 0); ctx.tune("grid_mode", 0)
    assert np.median(bd) < 1e-3
    cs.free(); ct.free(); cm.free()


@pytest.mark.parametrize("method", [1, 2])
def test_icp_same_pose_with_either_nn_method(ctx, orc, synth, method):
    src, tgt = synth.kitti_like_pair(9000, seed_target=47, seed_pair=48)
    ctx.tune("nn_method", method)
    cs, ct = ctx.cloud(src), ctx.cloud(tgt)
    T, st = ctx.icp_point2point(cs, ct, max_corr=1.0, max_iter=10, eps=1e-8)
    ctx.tune("nn_method", 0)
    oT, ost = orc.icp_p2p_f32(src, tgt, max_corr=1.0, max_iter=10, eps=1e-8)
    assert st["iters_run"] == ost["iters_run"] and st["last_pairs"] == ost["last_pairs"]
    assert np.linalg.norm(T.astype(np.float64) - oT) <= 1e-5
    cs.free(); ct.free()


@pytest.mark.parametrize("n", [9000, 70001])
def test_icp_sphere_walk_gate_as_bound_and_any_working_order(ctx, synth, n):
    """The sphere walk of large targets, forced onto a small pair.  Inside an ICP the walk starts from the caller's gate as a
    bound (pseudo-candidate (max_corr, none)) and runs over a working cloud sorted into record order or into coarse bins.  Neither
    may change a kept correspondence: the sums are exact, so pose, pair count and loss have to be bit-equal to the exhaustive
    search's for every iteration count, gate and loop flavour."""
    src, tgt = synth.kitti_like_pair(n, seed_target=147, seed_pair=148)
    src = src.copy(); src[:, 5] = np.nan; src[2, 77] = np.inf; src[0, 100:110] += 300.0; src[2, 200:260] += 1.3   # non-finite, far and gated-out queries
    cs, ct = ctx.cloud(src), ctx.cloud(tgt)
    gates = (1.0, 0.05, 4.0, 1e-9)
    ref = {}
    ctx.tune("nn_method", 1)
    for gate in gates:
        for it in (1, 2, 7):
            ref[gate, it] = ctx.icp_point2point(cs, ct, max_corr=gate, max_iter=it, eps=0.0)
    assert ref[1.0, 7][1]["last_pairs"] < n - 20 and ref[0.05, 1][1]["last_pairs"] < n - n // 4
    ctx.tune("nn_method", 2); ctx.tune("grid_order", 2); ctx.tune("grid_mode", 3)
    cm = ctx.cloud(tgt)                                  # a fresh cloud: its index is built Morton-ordered
    for fine, pipe, bounded in ((0, 0, 0), (1, 0, 0), (1, -1, 0), (2, 0, 0), (2, 1, 0), (1, 0, 2)):
        ctx.tune("grid_sort_fine", fine); ctx.tune("icp_pipeline", pipe); ctx.tune("icp_bounded_search", bounded)
        for gate in gates:
            for it in (1, 2, 7):
                T, st = ctx.icp_point2point(cs, cm, max_corr=gate, max_iter=it, eps=0.0)
                r = ref[gate, it]
                assert np.array_equal(T.view(np.uint32), r[0].view(np.uint32)), (fine, pipe, bounded, gate, it)
                for k in ("iters_run", "last_pairs", "empty_pairs"):
                    assert st[k] == r[1][k], (k, fine, pipe, bounded, gate, it)
                assert np.float32(st["last_loss"]).view(np.uint32) == np.float32(r[1]["last_loss"]).view(np.uint32)
    for k in ("nn_method", "grid_order", "grid_mode", "grid_sort_fine", "icp_pipeline", "icp_bounded_search"):
        ctx.tune(k, 0)
    cs.free(); ct.free(); cm.free()


@pytest.mark.parametrize("flavour", ["stile", "walk"])
@pytest.mark.parametrize("n", [30000, 200000])
def test_icp_tile_search_equals_cell_walk_and_brute_force(ctx, synth, n, flavour):
    """flavour "stile": the SIGN tile search (csrc/grid_stile.hpp, round 4: 64 queries per wave over the target's Morton-ordered matrix-core
    index, the sign form of the f16 filter; far queries deferred to the cell walk in list mode); "walk": the same loops with it switched off
    (tune grid_stile = 2: the cell walk alone, what a device whose f16 arithmetic fails the check gets).  Forced onto small pairs: pose bits, pair count and loss of every iteration
    count, gate and ball limit equal the exhaustive search's — with non-finite, far-away, gated-out and duplicated points in the
    clouds, and with the limits so tight that whole groups overflow a wave and take the deferral path."""
    src, tgt = synth.kitti_like_pair(n, seed_target=247, seed_pair=248)
    src = src.copy(); tgt = tgt.copy()
    src[:, 5] = np.nan; src[2, 77] = np.inf; src[0, 100:110] += 300.0; src[2, 200:260] += 1.3
    tgt[:, 1000:1040] = tgt[:, 2000:2001]                  # 40 exact duplicates of one target: ties, lowest original index wins
    src[:, 3000:3010] = tgt[:, 2000:2001]                  # ... and queries ON them
    cs, ct = ctx.cloud(src), ctx.cloud(tgt)
    gates = (1.0, 0.02, 1e-9)
    its = (1, 2, 3, 6, 11)
    ref = {}
    ctx.tune("nn_method", 1)
    for gate in gates:
        for it in its:
            ref[gate, it] = ctx.icp_point2point(cs, ct, max_corr=gate, max_iter=it, eps=0.0)
    ctx.tune("nn_method", 2); ctx.tune("grid_order", 2); ctx.tune("grid_mode", 3); ctx.tune("grid_tile", 1)
    ctx.tune("grid_stile", 2 if flavour == "walk" else 0)
    cm = ctx.cloud(tgt)                                    # a fresh cloud: its index is built Morton-ordered
    used = set()
    # (ball limit, loop, extra knobs): for the sign tile search also lists flushed after every entry / chunks always evaluated in place /
    # never in place, passes limited to 2 tiles or 1 coarse cell (whole waves overflow and take the deferral path), coarser / finer cells
    arms = ((0, 0, {}), (400, -1, {}))
    if flavour == "stile":
        arms = ((0, 0, {}), (1, 0, {}), (30, 0, {"grid_stile_flush": 1}), (200, -1, {"grid_stile_dense": 1}), (10, 1, {"grid_stile_dense": 65}),
                (100, 0, {"grid_stile_keep": 2}), (100, 0, {"grid_stile_cells": 1}), (300, 0, {"grid_stile_cells": 4096, "grid_tile_min_members": 1}),
                (20, 0, {"grid_stile_queue": 2}), (20, 0, {"grid_stile_list_wgs": 1}), (60, 0, {"grid_stile_split_mm": 1}),
                (60, 0, {"grid_stile_cold": 2}), (60, 0, {"grid_stile_l1": 2}), (100, 0, {"grid_stile_cold_own": 1, "grid_stile_cold_per": 0}), (100, 0, {"grid_stile_cold_per": 64}))
    for bmax, pipe, extra in arms:
        ctx.tune("grid_stile_bmax_cm", bmax); ctx.tune("icp_pipeline", pipe)
        for k in ("grid_stile_flush", "grid_stile_dense", "grid_stile_keep", "grid_stile_cells", "grid_tile_min_members", "grid_stile_queue", "grid_stile_list_wgs",
                  "grid_stile_split_mm", "grid_stile_cold", "grid_stile_cold_own", "grid_stile_cold_per", "grid_stile_l1"):
            ctx.tune(k, extra.get(k, 0))
        for gate in gates:
            for it in its:
                T, st = ctx.icp_point2point(cs, cm, max_corr=gate, max_iter=it, eps=0.0)
                if it > 2:
                    used.add(ctx.mfma_check()["last_nn1_kernel"])
                r = ref[gate, it]
                assert np.array_equal(T.view(np.uint32), r[0].view(np.uint32)), (bmax, pipe, gate, it)
                for k in ("iters_run", "last_pairs", "empty_pairs"):
                    assert st[k] == r[1][k], (k, bmax, pipe, gate, it)
                assert np.float32(st["last_loss"]).view(np.uint32) == np.float32(r[1]["last_loss"]).view(np.uint32)
    assert used == {"grid-stile" if flavour == "stile" else "grid"}, used
    # the exits of the state machine with the tile search's two launches in the enqueued tail: convergence at the 16th iteration
    # (registration.cpp:948-958 with eps large), no pair at all (the loop stops at once), max_iter 0
    for k in ("grid_stile_bmax_cm", "grid_stile_flush", "grid_stile_dense", "grid_stile_keep", "grid_stile_cells", "grid_tile_min_members",
              "grid_stile_queue", "grid_stile_list_wgs", "grid_stile_split_mm", "grid_stile_cold", "grid_stile_cold_own", "grid_stile_cold_per", "grid_stile_l1"):
        ctx.tune(k, 0)
    for pipe in (0, -1):
        ctx.tune("icp_pipeline", pipe)
        for kw in (dict(max_corr=1.0, max_iter=40, eps=1e30), dict(max_corr=1e-12, max_iter=9, eps=0.0), dict(max_corr=1.0, max_iter=0, eps=0.0)):
            ctx.tune("nn_method", 1)
            Tb, sb = ctx.icp_point2point(cs, ct, **kw)
            ctx.tune("nn_method", 2)
            Tt, stt = ctx.icp_point2point(cs, cm, **kw)
            assert np.array_equal(Tt.view(np.uint32), Tb.view(np.uint32)), (pipe, kw)
            for k in ("iters_run", "converged", "empty_pairs", "last_pairs"):
                assert stt[k] == sb[k], (k, pipe, kw)
    assert sb["iters_run"] == 0
    # the diagnostics: a loop at the converged pose serves nearly every query from the tiles
    ctx.tune("icp_pipeline", 0)
    T, _ = ctx.icp_point2point(cs, cm, max_corr=1.0, max_iter=25, eps=0.0)
    ca = cs.clone(); ctx.transform(ca, T)
    ctx.tune("grid_stats", 1)
    ctx.icp_point2point(ca, cm, max_corr=1.0, max_iter=3, eps=0.0)
    w = ctx.nn1_stats()
    ctx.tune("grid_stats", 0)
    assert w[0] > 0 and (flavour == "walk" or w[6] < n), w      # (sparse clouds: balls of the size of a cell — most queries take the walk at 30 000 points)
    for k in ("nn_method", "grid_order", "grid_mode", "grid_tile", "icp_pipeline", "grid_stile"):
        ctx.tune(k, 0)
    cs.free(); ct.free(); cm.free(); ca.free()


def test_cloud_layouts_roundtrip(ctx, pcr, synth):
    src, _ = synth.kitti_like_pair(1234)
    c = ctx.cloud(src)
    assert np.array_equal(c.numpy(), src)
    aos3 = np.ascontiguousarray(src.T)
    aos4 = np.concatenate([aos3, np.ones((1234, 1), np.float32)], axis=1)
    aos6 = np.concatenate([aos3, np.full((1234, 3), 0.5, np.float32)], axis=1)
    for arr, lay in ((aos3, pcr.PCR_AOS3), (aos4, pcr.PCR_AOS4), (aos6, pcr.PCR_AOS6)):
        c2 = ctx.cloud(arr, lay)
        assert np.array_equal(c2.numpy(), src)
        c2.free()
    c.free()


# ------------------------------------------------------------------ A8 transform, A7 sums
@pytest.mark.parametrize("n", [1, 3, 4, 1021, 4096, 50001])
def test_transform_bit_exact(ctx, orc, synth, n):
    src, _ = synth.kitti_like_pair(max(n, 64))
    src = np.ascontiguousarray(src[:, :n])
    T = synth.gt_pose().astype(np.float32)
    c = ctx.cloud(src)
    ctx.transform(c, T)
    got = c.numpy()
    want = orc.transform_f32(src, T[:3, :3], T[:3, 3])
    assert np.array_equal(bits32(got), bits32(want))
    # a transformed cloud is still a valid target (padding invariant restored)
    q = ctx.cloud(want[:, : min(n, 50)])
    idx, d2 = ctx.nn1(c, q)
    assert np.array_equal(idx, np.arange(min(n, 50), dtype=np.uint32)) and (d2 == 0).all()
    c.free(); q.free()


def test_kabsch_sums_vs_oracle(ctx, orc, synth):
    src, tgt = synth.kitti_like_pair(20000)
    cs, ct = ctx.cloud(src), ctx.cloud(tgt)
    ctx.nn1_async(ct, cs)
    sums, last, last_d2 = ctx.kabsch_sums(ct, cs, 0.05)
    idx, d2 = ctx.nn1_fetch(20000)
    osums, olast = orc.kabsch_accumulate(src, tgt, idx, d2, 0.05)
    assert sums[15] == osums[15] and last == olast and 0 < sums[15] < 20000
    assert np.float32(last_d2) == d2[olast]
    assert np.allclose(sums, osums, rtol=1e-13, atol=0)           # exact sums (integer limbs) vs the oracle's running f64 sums
    rc, R, t = orc.kabsch_solve(osums)
    import importlib
    rc2, R2, t2 = importlib.import_module("hands-on-point-cloud-processing_amd").kabsch_solve(sums)
    assert rc == rc2 == 0 and np.array_equal(R, R2) and np.array_equal(t, t2)
    cs.free(); ct.free()


def test_kabsch_reduction_paths_agree_bit_for_bit(ctx, synth):
    """The workgroup reduction of the Kabsch pass has two forms — the halving butterfly over the 40 live limbs (at most four pairs per
    thread) and one shuffle tree per limb (tune kabsch_bfly = 2; always beyond ~1 M points) — and the pass several launch geometries
    (tune kabsch_max_blocks).  The sums are exact integers, so every combination must give the same bits; sizes sit on the edges of a
    wave, a workgroup, the one-pair-per-thread rule (128 workgroups) and the four-pairs rule (1 024 workgroups)."""
    rng = np.random.default_rng(23)
    for n in (1, 2, 63, 64, 65, 255, 256, 257, 1000, 32768, 32769, 40001, 140000, 1048576, 1100003):
        tgt = rng.normal(0, 5, (3, min(n, 50000))).astype(np.float32)
        src = (tgt[:, rng.integers(0, tgt.shape[1], n)] + rng.normal(0, 0.05, (3, n))).astype(np.float32)
        ct, cs = ctx.cloud(tgt), ctx.cloud(src)
        ctx.nn1_async(ct, cs)
        got = []
        for bfly, cap in ((1, 0), (2, 0), (1, 64), (2, 7)):
            ctx.tune("kabsch_bfly", bfly); ctx.tune("kabsch_max_blocks", cap)
            sums, last, last_d2 = ctx.kabsch_sums(ct, cs, 0.004)
            got.append((bits64(sums).tobytes(), last, np.float32(last_d2).tobytes()))
        ctx.tune("kabsch_bfly", 0); ctx.tune("kabsch_max_blocks", 0)
        assert all(g == got[0] for g in got[1:]), n
        kept = np.frombuffer(got[0][0], np.float64)[15]
        assert 0 <= kept <= n and (n < 1000 or 0 < kept < n), (n, kept)
        cs.free(); ct.free()


def test_kabsch_sums_are_exact_and_order_independent(ctx, pcr, orc, synth):
    """The 16 moments are accumulated as integer limbs (csrc/numerics.hpp): exactly the rounded value of the true sum, whatever the
    order of the pairs, the launch geometry or the magnitude of the coordinates."""
    from fractions import Fraction
    rng = np.random.default_rng(17)
    for n, scale, shift in ((3000, 1.0, 0.0), (7001, 1e-6, 0.0), (5000, 1e7, 3e8), (257, 1.0, 0.0)):
        src, tgt = synth.kitti_like_pair(n, seed_target=11, seed_pair=12)
        src = (src * np.float32(scale) + np.float32(shift)).astype(np.float32)
        tgt = (tgt * np.float32(scale) + np.float32(shift)).astype(np.float32)
        gate = float(np.float32(scale * scale))                                  # keeps a strict subset of the pairs
        ct, cs = ctx.cloud(tgt), ctx.cloud(src)
        ctx.nn1_async(ct, cs)
        sums, last, last_d2 = ctx.kabsch_sums(ct, cs, gate)
        idx, d2 = ctx.nn1_fetch(n)
        keep = d2 < np.float32(gate)
        assert 0 < keep.sum() < n or scale != 1.0
        p64, q64 = src[:, keep].astype(np.float64), tgt[:, idx[keep]].astype(np.float64)
        want = np.zeros(16)
        for c in range(3):
            want[c] = float(sum((Fraction(float(v)) for v in p64[c]), Fraction(0)))
            want[3 + c] = float(sum((Fraction(float(v)) for v in q64[c]), Fraction(0)))
            for r in range(3):
                want[6 + 3 * r + c] = float(sum((Fraction(float(v)) for v in q64[r] * p64[c]), Fraction(0)))
        want[15] = keep.sum()
        assert np.array_equal(bits64(sums), bits64(want)), (n, scale, np.abs(sums - want).max())
        assert last == int(np.flatnonzero(keep)[-1]) and np.float32(last_d2) == d2[last]
        # the same pairs visited in another order (shuffled source cloud): the same bits
        perm = rng.permutation(n)
        cp = ctx.cloud(np.ascontiguousarray(src[:, perm]))
        ctx.nn1_async(ct, cp)
        sums2, last2, _ = ctx.kabsch_sums(ct, cp, gate)
        assert np.array_equal(bits64(sums2), bits64(sums))
        cp.free(); cs.free(); ct.free()
    # a kept pair whose source lies beyond the accumulation grid (unbounded gate, source 2^30 target extents away) is refused
    tgt = rng.normal(0, 1, (3, 500)).astype(np.float32)
    far = (tgt * np.float32(2.0 ** 40)).astype(np.float32)
    ct, cf = ctx.cloud(tgt), ctx.cloud(far)
    ctx.nn1_async(ct, cf)
    with pytest.raises(pcr.PcrError):
        ctx.kabsch_sums(ct, cf, 3e38)
    cf.free(); ct.free()


# ------------------------------------------------------------------ A9 ICP
def test_icp_vs_selfgolden_and_oracle(ctx, orc, golden):
    g = golden("icp_selfgolden.npz")
    cs, ct = ctx.cloud(g["src"]), ctx.cloud(g["tgt"])
    T, st = ctx.icp_point2point(cs, ct, max_corr=1.0, max_iter=20, eps=1e-8)
    assert np.linalg.norm(T.astype(np.float64) - g["T"]) <= 1e-5
    assert st["iters_run"] == int(g["iters_run"]) and st["converged"] == int(g["converged"])
    assert st["last_pairs"] == int(g["per_iter_pairs"][st["iters_run"] - 1 if st["converged"] == 0 else st["iters_run"]])
    assert np.array_equal(cs.numpy(), g["src"])                   # the input cloud is not modified
    cs.free(); ct.free()


def test_icp_state_machine_matches_oracle(ctx, orc, synth):
    src, tgt = synth.kitti_like_pair(2500, seed_target=31, seed_pair=32)
    cs, ct = ctx.cloud(src), ctx.cloud(tgt)
    init = np.eye(4, dtype=np.float32); init[0, 3] = 0.1
    for kw in (dict(max_iter=40, eps=1e30), dict(max_iter=7, eps=0.0), dict(max_iter=12, eps=1e-8, max_corr=0.3)):
        T, st = ctx.icp_point2point(cs, ct, init_T=init, **kw)
        oT, ost = orc.icp_p2p_f32(src, tgt, init_T=init, **kw)
        assert st["iters_run"] == ost["iters_run"] and st["converged"] == ost["converged"]
        assert st["last_pairs"] == ost["last_pairs"]
        assert np.linalg.norm(T.astype(np.float64) - oT) <= 1e-5
    far = ctx.cloud(src + np.float32(1000.0))
    T, st = ctx.icp_point2point(far, ct, max_iter=5)
    assert st["empty_pairs"] == 1 and np.array_equal(T, np.eye(4, dtype=np.float32))
    cs.free(); ct.free(); far.free()


def test_matrix_core_search_switches_change_no_bit(ctx, synth):
    """The launch / ordering switches of the matrix-core brute-force search are speed only: XCD-aware launch off / 1 / 2 / 4 (nn1_xcd), the
    loop's sorted working cloud off (bt_sort_work), the seed of the next search written by the move or by its own kernel
    (icp_seed_in_move), solve + move in one launch or two (icp_fused_move), slice length (nn1_supers_per_slice), query groups per wave
    (nn1_btrack_qg), each for the f16 and the bf16 form — same keys as the exact-only kernel for a one-shot search, same pose /
    statistics / loss for a 10-iteration loop."""
    n = 21000                                                       # > 8 192: the matrix-core kernels answer; 83 super-tiles
    src, tgt = synth.kitti_like_pair(n, seed_target=601, seed_pair=602)
    ctx.tune("nn_method", 1)
    cs, ct = ctx.cloud(src), ctx.cloud(tgt)
    ctx.tune("nn1_variant", 2)
    ref_idx, ref_d2 = ctx.nn1(ct, cs)
    ctx.tune("nn1_variant", 0)
    T0, st0 = ctx.icp_point2point(cs, ct, max_corr=1.0, max_iter=10, eps=0.0)
    switches = [dict(nn1_xcd=-1), dict(nn1_xcd=1), dict(nn1_xcd=2), dict(bt_sort_work=2), dict(icp_seed_in_move=2), dict(icp_fused_move=2),
                dict(icp_seed_in_move=2, icp_fused_move=2), dict(icp_fused_max=4096),
                dict(nn1_supers_per_slice=1), dict(nn1_supers_per_slice=5), dict(nn1_supers_per_slice=40), dict(nn1_btrack_qg=2), dict(nn1_btrack_qg=4),
                dict(nn1_f16=2), dict(nn1_f16=2, nn1_btrack_qg=4), dict(nn1_f16=2, icp_seed_in_move=2, nn1_supers_per_slice=2),
                # the sign form of the f16 filter (STRACK): never, its list flushed after every super-tile or only when full, no cold seed
                # (cold searches then take HTRACK, stale seeds are not merged with a cold one), one-slice launches too (variant 8)
                dict(nn1_sign=2), dict(nn1_sign_flush=1), dict(nn1_sign_flush=100000), dict(nn1_cold_seed=2), dict(nn1_cold_seed=2, icp_seed_in_move=2),
                dict(nn1_sign_flush=1, nn1_supers_per_slice=1), dict(nn1_supers_per_slice=100, nn1_btrack_qg=2),
                dict(nn1_variant=8), dict(nn1_variant=8, nn1_xcd=-1, nn1_btrack_qg=2),
                # the sphere form on this small target (groups per wave, evaluation cadence)
                dict(nn1_variant=10), dict(nn1_variant=10, nn1_sphere_qg=2, nn1_sign_flush=1), dict(nn1_variant=10, nn1_sphere_qg=4, nn1_sphere_flush_end=128),
                dict(nn1_variant=10, nn1_cold_seed=2, icp_seed_in_move=2),
                # where cold seeds come from (centre of the nearest super-tile / Morton neighbour / both), stale seeds kept as they are, a coarser working-cloud sort
                dict(nn1_seed_mode=1), dict(nn1_seed_mode=1, nn1_seed_levels=1, nn1_variant=10), dict(nn1_seed_mode=2), dict(nn1_seed_mode=3, nn1_variant=8), dict(nn1_seed_mode=2, nn1_variant=10, nn1_sphere_reseed=2),
                dict(nn1_variant=10, nn1_sphere_reseed=2, icp_seed_in_move=2), dict(bt_sort_begin_bit=12), dict(bt_sort_begin_bit=24, nn1_variant=10)]
    for sw in switches:
        for k, v in sw.items():
            ctx.tune(k, v)
        fresh = ctx.cloud(tgt)                                      # cold, unseeded search of a target without an index
        idx, d2 = ctx.nn1(fresh, cs)
        assert np.array_equal(idx, ref_idx) and np.array_equal(bits32(d2), bits32(ref_d2)), sw
        T, st = ctx.icp_point2point(cs, fresh, max_corr=1.0, max_iter=10, eps=0.0)
        assert np.array_equal(T.view(np.uint32), T0.view(np.uint32)), sw
        for k in ("iters_run", "converged", "empty_pairs", "last_pairs"):
            assert st[k] == st0[k], (k, sw)
        assert np.float32(st["last_loss"]).view(np.uint32) == np.float32(st0["last_loss"]).view(np.uint32), sw
        fresh.free()
        for k in sw:
            ctx.tune(k, 0)
    ctx.tune("nn_method", 0)
    cs.free(); ct.free()


@pytest.mark.parametrize("n", [100, 128, 300, 1500, 2047, 2048])
def test_icp_auto_dispatch_of_small_targets_changes_no_bit(ctx, synth, n):
    """Inside a loop the dispatcher sends targets from 128 points on to the exact grid (api.cpp nn1_auto_grid: max_iter >= 5 or an
    existing index), below that and for short loops to the exhaustive kernels: pose, statistics and loss are those of either search
    forced, bit for bit — for long and short loops, fresh targets and targets whose index exists."""
    src, tgt = synth.kitti_like_pair(n, seed_target=501 + n, seed_pair=502 + n)
    for kw in (dict(max_iter=14, eps=1e-8), dict(max_iter=3, eps=0.0), dict(max_iter=30, eps=1e30)):
        got = []
        for method in (1, 2, 0, 0):
            ctx.tune("nn_method", method)
            cs = ctx.cloud(src)
            ct = ctx.cloud(tgt) if len(got) < 3 else ct        # the fourth run searches the target the third one indexed (or did not)
            T, st = ctx.icp_point2point(cs, ct, **kw)
            got.append((T.view(np.uint32).tobytes(), st["iters_run"], st["converged"], st["empty_pairs"], st["last_pairs"],
                        np.float32(st["last_loss"]).tobytes()))
            cs.free()
            if len(got) < 3: ct.free()
        ct.free()
        ctx.tune("nn_method", 0)
        assert all(g == got[0] for g in got[1:]), (n, kw)


@pytest.mark.parametrize("method", [1, 2])
def test_icp_pipelined_equals_synchronous_loop(ctx, synth, method):
    """The device-resident pipelined loop (default) and the synchronous host loop share one numerics header:
    pose, iteration count, flags and loss must be identical, bit for bit, in every exit path."""
    src, tgt = synth.kitti_like_pair(7000, seed_target=81, seed_pair=82)
    cs, ct = ctx.cloud(src), ctx.cloud(tgt)
    far = ctx.cloud(src + np.float32(1000.0))
    init = np.eye(4, dtype=np.float32); init[1, 3] = -0.07
    ctx.tune("nn_method", method)
    cases = [(cs, dict(max_iter=12, eps=1e-8)), (cs, dict(max_iter=40, eps=1e30)), (cs, dict(max_iter=1, eps=0.0)),
             (cs, dict(max_iter=0, eps=0.0)), (cs, dict(max_iter=9, eps=1e-8, max_corr=0.2, init_T=init)),
             (far, dict(max_iter=5, eps=1e-8)), (cs, dict(max_iter=23, eps=0.0))]
    for cloud, kw in cases:
        ctx.tune("icp_pipeline", -1)
        Ts, ss = ctx.icp_point2point(cloud, ct, **kw)
        # (fused: 0 = default — solve + move in ONE launch at this size, state double-buffered; 2 = the separate update and move kernels;
        #  slots 1 = the multi-rank kernels reduce_slots + update_from_sums)
        for chunk, slots, fused in ((1, 0, 0), (4, 0, 0), (7, 0, 0), (4, 1, 0), (4, 0, 2), (3, 0, 2)):
            ctx.tune("icp_pipeline", 1); ctx.tune("icp_chunk", chunk); ctx.tune("icp_force_slots", slots); ctx.tune("icp_fused_move", fused)
            Tp, sp = ctx.icp_point2point(cloud, ct, **kw)
            assert np.array_equal(Ts.view(np.uint32), Tp.view(np.uint32)), (kw, chunk, slots, fused)
            for k in ("iters_run", "converged", "empty_pairs", "last_pairs"):
                assert ss[k] == sp[k], (k, kw, chunk, slots, fused, ss, sp)
            assert np.float32(ss["last_loss"]).view(np.uint32) == np.float32(sp["last_loss"]).view(np.uint32)
    for k in ("icp_pipeline", "icp_chunk", "icp_force_slots", "icp_fused_move", "nn_method"):
        ctx.tune(k, 0)
    cs.free(); ct.free(); far.free()


def test_icp_full_size_120k_recovers_ground_truth(ctx, synth):
    # BASELINE config 3 (120k pair, 20 iterations): the oracle is O(N^2), so check properties instead:
    # pose moves monotonically towards the known ground truth and all pairs are kept at the end
    src, tgt = synth.kitti_like_pair(120000)
    cs, ct = ctx.cloud(src), ctx.cloud(tgt)
    gt = synth.gt_pose()
    errs = []
    for it in (5, 20):
        T, st = ctx.icp_point2point(cs, ct, max_corr=1.0, max_iter=it, eps=1e-8)
        errs.append(np.linalg.norm(T - gt))
    assert errs[1] < errs[0] < np.linalg.norm(np.eye(4) - gt) and errs[1] < 0.01
    assert st["last_pairs"] == 120000 and st["iters_run"] == 20
    cs.free(); ct.free()


def test_nn1_full_size_120k_properties(ctx, orc, synth):
    # BASELINE config 2: 120k x 120k. (a) self-query returns identity with d2 == 0 (sortedness/idempotence);
    # (b) a 2 000-query sample agrees bit-for-bit with the oracle.
    src, tgt = synth.kitti_like_pair(120000)
    ct = ctx.cloud(tgt)
    idx, d2 = ctx.nn1(ct, ct)
    assert (d2 == 0).all()
    assert (idx <= np.arange(120000)).all()                       # lowest index among exact duplicates
    moved = idx != np.arange(120000)
    assert np.array_equal(tgt[:, idx[moved]], tgt[:, moved])      # ... which are bit-identical points
    cs = ctx.cloud(src)
    idx, d2 = ctx.nn1(ct, cs)
    sel = np.arange(0, 120000, 60)
    oidx, od2 = orc.nn1_f32(tgt, np.ascontiguousarray(src[:, sel]))
    assert np.array_equal(idx[sel], oidx) and np.array_equal(bits32(d2[sel]), bits32(od2))
    cs.free(); ct.free()


# ------------------------------------------------------------------ A10 plane-inlier count
def test_plane_count_vs_hw4_golden(ctx, golden):
    g = golden("plane_hw4.npz")
    c = ctx.cloud(g["pts_f32"], 1)
    counts = ctx.plane_count(c, g["params"], float(g["thr"]))
    assert np.array_equal(counts, g["counts"])
    masks = np.unpackbits(g["masks"], axis=1)[:, : g["pts_f32"].shape[0]]
    for h in range(16):
        mask, cnt = ctx.plane_mask(c, g["params"][h], float(g["thr"]))
        assert np.array_equal(mask, masks[h]) and cnt == g["counts"][h]
    c.free()


def test_plane_count_120k_80_hypotheses_vs_oracle(ctx, orc, synth):
    # BASELINE config 4: one 120k scan, 80 hypotheses (40 per x-segment, ground_detection_ransac.py:54,71-72)
    scan = synth.kitti_like_scan(120000)
    ground = np.where(np.abs(scan[2] + 1.73) < 0.3)[0]
    pick = (synth.splitmix64(9, np.arange(240, dtype=np.uint64)) % np.uint64(ground.size)).astype(np.int64).reshape(80, 3)
    planes = np.stack([orc.plane_from_3pts(scan[:, ground[p]].T.astype(np.float64)) for p in pick])
    planes = planes[np.isfinite(planes).all(axis=1)]
    c = ctx.cloud(scan)
    counts = ctx.plane_count(c, planes, 0.15)
    assert np.array_equal(counts, orc.plane_count(scan, planes, 0.15))
    assert counts.max() > 20000
    # more than one launch worth of hypotheses, and none
    many = np.tile(planes, (3, 1))[:200]
    assert np.array_equal(ctx.plane_count(c, many, 0.15), orc.plane_count(scan, many, 0.15))
    assert ctx.plane_count(c, np.zeros((0, 4)), 0.15).size == 0
    # the hypotheses in groups of any size across the rows of the launch (default 20): groups that straddle lane 64, one per row, all in one
    for group in (1, 7, 33, 64, 96):
        ctx.tune("plane_group", group)
        assert np.array_equal(ctx.plane_count(c, many, 0.15), orc.plane_count(scan, many, 0.15)), group
    ctx.tune("plane_group", 0)
    c.free()


# ------------------------------------------------------------------ A2/A4/A11 k-NN + radius (hw2 contract)
def test_knn_kat_query5(ctx, golden):
    g = golden("kat_kitti_q5.npz")
    db = g["db_f32"].astype(np.float64)
    idx, dist = ctx.knn_f64(db, db[5:6], 8)
    assert idx[0].tolist() == [5, 1972, 6, 1971, 1970, 3946, 8, 3945]          # result_cpp.txt:13-20
    assert np.array_equal(bits64(dist[0]), bits64(g["knn_dist"]))
    row, ridx, rdist = ctx.radius_f64(db, db[5:6], 1.66)
    assert ridx.tolist() == [5, 6, 8, 1970, 1971, 1972, 3945, 3946]            # result_cpp.txt:26-33
    o = np.argsort(g["radius_idx_visit_order"])
    assert np.array_equal(bits64(rdist), bits64(g["radius_dist_visit_order"][o]))


@pytest.mark.parametrize("case", ["synth1000", "kitti4096", "lattice"])
@pytest.mark.parametrize("k", [1, 8])
def test_knn_vs_hw2_golden(ctx, orc, golden, case, k):
    g = golden(f"knn_hw2_{case}.npz")
    idx, dist = ctx.knn_f64(g["db"], g["q"], k)
    assert np.array_equal(bits64(dist), bits64(g[f"dist_k{k}"]))               # distances bit-equal to hw2's
    oidx, odist = orc.knn_f64(g["db"], g["q"], k)
    assert np.array_equal(idx, oidx)                                           # canonical order (= oracle)
    if case != "lattice":
        assert np.array_equal(idx, g[f"idx_k{k}"])                             # tie-free: equal to hw2's indices


@pytest.mark.parametrize("k", [1, 3, 8, 13, 32])
def test_knn_other_k_and_short_db(ctx, orc, synth, k):
    db = synth.uniform_cloud(700, seed=3)
    q = synth.uniform_cloud(300, seed=4)
    idx, dist = ctx.knn_f64(db, q, k)
    oidx, odist = orc.knn_f64(db, q, k)
    assert np.array_equal(idx, oidx) and np.array_equal(bits64(dist), bits64(odist))
    idx, dist = ctx.knn_f64(db[:5], q[:7], k)                                  # n < k: slots keep (1e10, 0)
    oidx, odist = orc.knn_f64(db[:5], q[:7], k)
    assert np.array_equal(idx, oidx) and np.array_equal(bits64(dist), bits64(odist))


@pytest.mark.parametrize("case", ["synth1000", "kitti4096"])
@pytest.mark.parametrize("r", [0.5, 1.0])
def test_radius_vs_hw2_golden(ctx, golden, case, r):
    g = golden(f"radius_hw2_{case}.npz")
    tag = str(r).replace(".", "p")
    row, idx, dist = ctx.radius_f64(g["db"], g["q"], r)
    assert np.array_equal(row, g[f"row_r{tag}"]) and np.array_equal(idx, g[f"idx_r{tag}"])
    assert np.array_equal(bits64(dist), bits64(g[f"dist_r{tag}"]))


def test_radius_boundary_is_inclusive_and_sqrt_exact(ctx, orc, synth):
    # radii that coincide with realised distances: d <= r must include them (resultSet.hpp:133)
    db = synth.lattice_cloud(2000, 3, 10.0, seed=8, levels=20)
    q = synth.lattice_cloud(200, 3, 10.0, seed=9, levels=20)
    for r in (0.5, 1.0, float(np.sqrt(0.5)), 1.5, 0.0):
        row, idx, dist = ctx.radius_f64(db, q, r)
        orow, oidx, odist = orc.radius_f64(db, q, r)
        assert np.array_equal(row, orow) and np.array_equal(idx, oidx)
        assert np.array_equal(bits64(dist), bits64(odist))


def test_radius_self_query_full_scan(ctx, synth):
    # BASELINE config 4 (radius-NN r = 1.0, every point queries its own scan) at 30k (the 120k run is the
    # bench's job): every row contains its own index, rows are ascending, symmetric membership
    scan = synth.kitti_like_scan(30000).T.astype(np.float64)
    row, idx, dist = ctx.radius_f64(scan, scan, 1.0)
    assert row[-1] == idx.size and (np.diff(row) >= 1).all()
    own = np.repeat(np.arange(30000), np.diff(row)) == idx
    assert own.sum() >= 30000 and (dist[own] == 0).all()
    seg_start = np.zeros(idx.size, bool); seg_start[row[:-1]] = True
    assert (np.diff(idx)[~seg_start[1:]] > 0).all()
    assert (dist <= 1.0).all()


def test_context_owns_its_handles(pcr, synth):
    """close() (or leaving a with-block) frees the clouds / databases still alive; later free() calls are no-ops."""
    src, tgt = synth.kitti_like_pair(500)
    with pcr.Context(0) as ctx:
        cs, ct = ctx.cloud(src), ctx.cloud(tgt)
        db = ctx.db64(np.ascontiguousarray(tgt.T.astype(np.float64)))
        idx, _ = ctx.nn1(ct, cs)
        assert idx.shape == (500,)
    assert ctx.h is None and cs.h is None and ct.h is None and db.h is None
    cs.free(); db.free(); ctx.close()          # idempotent
    ctx2 = pcr.Context(0)
    c = ctx2.cloud(src)
    del ctx2                                     # the cloud keeps its context alive
    assert len(c) == 500
    c.free()


def test_working_copies_are_parked_callers_clouds_are_freed(pcr, synth):
    """only the ICP loops' own working copies are parked on the context (for the next call's clone); a cloud the CALLER
    destroys goes back to the device at once, parked buffers of another size are dropped by the next allocation, pcr_ctx_trim frees them."""
    with pcr.Context(0) as ctx:
        assert ctx.parked_bytes() == 0
        a, b = ctx.cloud(synth.kitti_like_pair(3001)[0]), ctx.cloud(synth.kitti_like_pair(7777)[0])
        a.free(); b.free()
        assert ctx.parked_bytes() == 0                               # odd-sized caller clouds: nothing stays behind
        src, tgt = synth.kitti_like_pair(5000)
        cs, ct = ctx.cloud(src), ctx.cloud(tgt)
        T1, _ = ctx.icp_point2point(cs, ct, max_iter=3)
        p1 = ctx.parked_bytes()
        assert 0 < p1 <= 2 * 3 * 4 * (5000 + 2048)                   # the loop's clone and its sorted copy, nothing else
        T2, _ = ctx.icp_point2point(cs, ct, max_iter=3)              # the reuse path: same buffers, same result
        assert ctx.parked_bytes() == p1 and np.array_equal(T1.view(np.uint32), T2.view(np.uint32))
        s2, t2 = synth.kitti_like_pair(9000)
        c2, d2 = ctx.cloud(s2), ctx.cloud(t2)                        # another size: the stale spares are dropped on the miss
        assert ctx.parked_bytes() == 0
        ctx.icp_point2point(c2, d2, max_iter=2)
        assert 0 < ctx.parked_bytes() <= 2 * 3 * 4 * (9000 + 2048)
        ctx.trim()
        assert ctx.parked_bytes() == 0


def test_spatial_shards_are_a_partition_and_keep_the_scene_dense(pcr, synth):
    """pcr_cloud_shard_spatial: for every rank count the shards are disjoint and complete, keep ascending original order and the points'
    coordinates; sizes differ by at most one run; a shard is spatially compact — the mean distance between CONSECUTIVE points of its
    index-sorted working order stays that of the whole cloud, where a contiguous block of the shuffled source is N times sparser."""
    n = 60000
    src, tgt = synth.kitti_like_pair(n)
    with pcr.Context(0) as ctx:
        ct, full = ctx.cloud(tgt), ctx.cloud(src)
        for world, chunks in ((1, 0), (2, 0), (3, 7), (8, 0), (8, 1)):
            seen = np.zeros(n, np.int32)
            sizes = []
            for r in range(world):
                sh = ctx.shard_spatial(ct, full, world, r, chunks)
                gi = ctx.global_index(sh)
                assert gi.size == len(sh) and (np.diff(gi.astype(np.int64)) > 0).all()
                assert np.array_equal(sh.numpy(), src[:, gi])
                seen[gi] += 1
                sizes.append(gi.size)
                sh.free()
            assert (seen == 1).all(), (world, chunks)
            per = -(-n // (world * (chunks or 64)))
            assert max(sizes) - min(sizes) <= per, (world, chunks, sizes)
        with pytest.raises(pcr.PcrError):
            ctx.global_index(full)                                   # not a shard
        e = ctx.shard_spatial(ct, ctx.cloud(np.zeros((3, 0), np.float32)), 4, 1)
        assert len(e) == 0 and ctx.global_index(e).size == 0


def test_caller_stepped_loop_equals_icp_call(pcr, synth):
    """pcr_cloud_sort_for_target + pcr_nn1_f32_loop + pcr_kabsch_sums + pcr_kabsch_solve + pcr_transform_f32 stepped by the caller
    = pcr_icp_p2p_f32, pose bits and kept pairs, with the exact grid and with the exhaustive search (registration.cpp:917-1006)."""
    for n, method in ((30000, 2), (30000, 1), (3000, 0)):
        src, tgt = synth.kitti_like_pair(n)
        with pcr.Context(0) as ctx:
            ctx.tune("nn_method", method)
            cs, ct = ctx.cloud(src), ctx.cloud(tgt)
            T_ref, st = ctx.icp_point2point(cs, ct, max_corr=1.0, max_iter=6, eps=0.0)
            work = cs.clone()
            orig = ctx.sort_for_target(ct, work)
            assert np.array_equal(np.sort(orig), np.arange(n, dtype=np.uint32))
            assert np.array_equal(work.numpy(), src[:, orig])
            T = np.eye(4, dtype=np.float32)
            for it in range(6):
                ctx.nn1_loop(ct, work, 1.0)
                sums, last, _ = ctx.kabsch_sums(ct, work, 1.0)
                rc, R, t = pcr.kabsch_solve(sums)
                assert rc == 0
                Td = np.eye(4, dtype=np.float32); Td[:3, :3], Td[:3, 3] = R, t
                T = pcr_mat4(Td, T)
                ctx.transform(work, Td)
            assert int(sums[15]) == st["last_pairs"]
            assert np.array_equal(T.view(np.uint32), T_ref.view(np.uint32)), (n, method)


def pcr_mat4(A, B):
    """T_delta . T_total in f32, row by row as mat4_mul_f32 composes it (registration.cpp:1000-1002)"""
    out = np.zeros((4, 4), np.float32)
    for i in range(4):
        for j in range(4):
            acc = np.float32(0.0)
            for k in range(4):
                acc = np.float32(acc + np.float32(A[i, k] * B[k, j]))
            out[i, j] = acc
    return out


def test_icp_bounded_search_with_outliers(pcr, synth):
    """Partial overlap: 10 % of the source has no target within the max_corres_dist gate.  The grid search bounded by that gate
    (default inside ICP loops), the unbounded exact search and brute force give the same pose bits and statistics."""
    n = 20000
    src, tgt = synth.kitti_like_pair(n)
    rng = np.random.default_rng(11)
    out = rng.choice(n, n // 10, replace=False)
    src = src.copy()
    src[:, out] += rng.uniform(15.0, 40.0, (1, out.size)).astype(np.float32) * np.array([[0.3], [0.2], [1.0]], np.float32)
    res = []
    for method, bounded in ((2, 1), (2, 2), (1, 1)):
        with pcr.Context(0) as ctx:
            ctx.tune("nn_method", method); ctx.tune("icp_bounded_search", bounded)
            for mc, it in ((1.0, 6), (0.05, 4), (400.0, 3)):
                T, st = ctx.icp_point2point(ctx.cloud(src), ctx.cloud(tgt), max_corr=mc, max_iter=it, eps=0.0)
                res.append((method, bounded, mc, T.view(np.uint32).copy(), st["last_pairs"], st["iters_run"], st["empty_pairs"]))
    ref = [r for r in res if r[0] == 1]
    for r in res:
        m = [x for x in ref if x[2] == r[2]][0]
        assert np.array_equal(r[3], m[3]) and r[4:] == m[4:], (r[0], r[1], r[2])
    assert ref[0][4] <= n - n // 10                      # the outliers were indeed rejected at max_corr = 1

"""Device-resident radius rows (include/pcr.h pcr_rows): the CSR of a radius search kept in HBM.  Same rows as pcr_db64_radius bit for
bit (pinned to the hw2 kd-tree by tests/test_gpu_parity.py::test_radius_vs_hw2_golden), fetched whole, in blocks and reduced on the
GPU.  Reference consumers: Homework7/hw7/src/iss_detector.cpp:48-76 (counts, weights), Homework1 pca_normal.py:89-103 (moments),
Homework2/hw2/include/benchmark.hpp:66-70 (the self-query walk)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits64(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


@pytest.fixture(scope="module")
def ctx(pcr):
    c = pcr.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("n,r,exact_f32", [(20000, 0.5, True), (6000, 0.8, True), (3000, 0.7, False), (100, 5.0, True)])
def test_rows_handle_equals_host_rows_and_reduces(ctx, synth, n, r, exact_f32):
    scan = synth.kitti_like_scan(max(n, 64))[:, :n]
    db = np.ascontiguousarray(scan.T.astype(np.float64))
    if not exact_f32:
        db = db + 1e-9 * np.arange(n)[:, None]             # not f32-representable: the exhaustive route, rows uploaded once
    d = ctx.db64(db)
    row, idx, dist = d.radius(db, r)
    R = d.radius_rows(None, r)                             # every point queries the database
    assert R.m == n and R.total == row[-1]
    assert np.array_equal(R.row_ptr(), row)
    i2, d2 = R.fetch(0, n)
    assert np.array_equal(i2, idx) and np.array_equal(bits64(d2), bits64(dist))
    rng = np.random.default_rng(5)
    for _ in range(6):                                     # blocks of whole rows
        a = int(rng.integers(0, n)); b = int(rng.integers(a, min(n, a + 400) + 1))
        ib, db_ = R.fetch(a, b, row)
        assert np.array_equal(ib, idx[row[a]:row[b]]) and np.array_equal(bits64(db_), bits64(dist[row[a]:row[b]]))
    cnt = R.reduce(R.COUNT)
    assert np.array_equal(cnt, np.diff(row).astype(np.float64))
    nz = np.diff(row) > 0
    mx = R.reduce(R.MAX_DIST)
    assert np.array_equal(mx[nz], np.maximum.reduceat(dist, row[:-1][nz]))        # exact: a maximum does not round
    sm = R.reduce(R.SUM_DIST)
    want = np.add.reduceat(dist, row[:-1][nz])
    assert np.allclose(sm[nz], want, rtol=1e-12, atol=0) and (sm[~nz] == 0).all()
    mean, cov = R.moments()
    for i in rng.integers(0, n, 40):
        nb = db[idx[row[i]:row[i + 1]]]
        m = nb.mean(axis=0)
        c = (nb - m).T @ (nb - m) / nb.shape[0]
        assert np.allclose(mean[i], m, rtol=1e-12, atol=1e-12)
        assert np.allclose(cov[i], [c[0, 0], c[0, 1], c[0, 2], c[1, 1], c[1, 2], c[2, 2]], rtol=1e-9, atol=1e-12)
    # queries that are not the database
    q = db[rng.integers(0, n, 300)] + 0.25
    rq, iq, dq = d.radius(q, r)
    Rq = d.radius_rows(q, r)
    assert np.array_equal(Rq.row_ptr(), rq)
    i3, d3 = Rq.fetch(0, 300)
    assert np.array_equal(i3, iq) and np.array_equal(bits64(d3), bits64(dq))
    Rq.free(); R.free(); d.free()


def test_rows_of_an_empty_search(ctx):
    db = np.array([[0.0, 0.0, 0.0], [10.0, 0.0, 0.0]])
    d = ctx.db64(db)
    R = d.radius_rows(np.array([[100.0, 100.0, 100.0]]), 1.0)
    assert R.m == 1 and R.total == 0 and np.array_equal(R.row_ptr(), [0, 0])
    assert np.array_equal(R.reduce(R.COUNT), [0.0]) and np.array_equal(R.reduce(R.SUM_DIST), [0.0])
    i, dd = R.fetch(0, 1)
    assert i.size == 0 and dd.size == 0
    R.free(); d.free()

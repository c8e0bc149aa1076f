"""BASELINE config 1: "Homework2 NNSearchTree brute-force 1-NN on 1 000 random 3-D points, CPU reference path".
The reference's brute force exists only in Python: Homework2/hw2/benchmark.py:69-71
    diff = np.linalg.norm(np.expand_dims(query, 0) - db_np, axis=1); nn_idx = np.argsort(diff); nn_dist = diff[nn_idx]
on float32 points; the inputs follow its generator (test.hpp:142: range * (rand() % 1000) / 1000.0, glibc rand())."""
import numpy as np
import pytest


def numpy_brute_force(db32, query32):
    diff = np.linalg.norm(np.expand_dims(query32, 0) - db32, axis=1)      # benchmark.py:69
    nn_idx = np.argsort(diff)                                            # :70
    return nn_idx, diff[nn_idx]                                          # :71


@pytest.fixture(scope="module")
def c1_data(synth):
    pts = synth.glibc_rand_lattice(2000, 3, 10.0, seed=1)                # hw2 never seeds: glibc default seed 1
    return pts[:1000].astype(np.float32), pts[1000:].astype(np.float32)  # database, second draw = queries


def check_against_numpy(db, q, idx, d2):
    ties = 0
    for i in range(q.shape[0]):
        nn_idx, nn_dist = numpy_brute_force(db, q[i])
        assert nn_dist.dtype == np.float32
        # distance parity: numpy's f32 norm == sqrt_f32 of the A1 squared distance, bit for bit
        assert np.sqrt(d2[i]).view(np.uint32) == nn_dist[0].view(np.uint32)
        # index parity under the tie-set rule: both indices attain the minimum distance; ours is the lowest index
        # among the minimisers of the squared distance
        full = np.linalg.norm(q[i][None, :] - db, axis=1)
        tie_set = np.where(full == nn_dist[0])[0]
        assert nn_idx[0] in tie_set and idx[i] in tie_set
        ties += tie_set.size > 1
    return ties


def test_c1_oracle_matches_numpy_brute_force(orc, c1_data):
    db, q = c1_data
    idx, d2 = orc.nn1_f32(np.ascontiguousarray(db.T), np.ascontiguousarray(q.T))
    check_against_numpy(db, q, idx, d2)


def test_c1_reference_kdtree_matches_on_the_same_points(orc, c1_data):
    if not orc.have_ref():
        pytest.skip("oracle/_ref not built")
    db, q = c1_data
    udb = np.unique(db.astype(np.float64), axis=0)
    ridx, rdist = orc.ref_hw2_kd_knn(udb, q.astype(np.float64), 1, leaf=32)    # leaf 32: see DESIGN.md §2 (lattice inputs)
    oidx, odist = orc.knn_f64(udb, q.astype(np.float64), 1)
    assert np.array_equal(rdist.view(np.uint64), odist.view(np.uint64))


@pytest.mark.gpu
def test_c1_gpu_matches_numpy_brute_force(pcr, c1_data):
    db, q = c1_data
    ctx = pcr.Context(0)
    try:
        for method in (1, 2):
            ctx.tune("nn_method", method)
            ct, cs = ctx.cloud(np.ascontiguousarray(db.T)), ctx.cloud(np.ascontiguousarray(q.T))
            idx, d2 = ctx.nn1(ct, cs)
            check_against_numpy(db, q, idx, d2)
            ct.free(); cs.free()
    finally:
        ctx.close()

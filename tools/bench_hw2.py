#!/usr/bin/env python3
"""The hw2 benchmark() protocol (Homework2/hw2/include/benchmark.hpp:6-78: first 10 000 points, k = 8, leaf 1, every point
queries its own cloud) on the GPU drop-in vs the reference's own kd-tree (oracle/_ref) on this box's host CPU."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
pcr = importlib.import_module("hands-on-point-cloud-processing_amd")
g = np.load(os.path.join(ROOT, "tests", "golden", "kat_kitti_q5.npz"))
for npts in (10000, 100000):
    db = np.unique(g["db_f32"][:npts].astype(np.float64), axis=0)   # the reference's build cannot take duplicates
    ctx = pcr.Context(0); ctx.tune("prof", 2)
    t0 = time.perf_counter(); h = ctx.db64(db); tb = time.perf_counter() - t0
    # the first batched call pays one-time costs (code objects, sort workspace, the index over the database): reported apart
    t0 = time.perf_counter(); idx, dist = h.knn(db, 8); tfirst = time.perf_counter() - t0
    tq = 1e9
    for _ in range(3):
        ctx.prof_reset()
        t0 = time.perf_counter(); idx, dist = h.knn(db, 8); tq = min(tq, time.perf_counter() - t0)
    k, ms = ctx.prof_get("knn_grid")
    print(f"GPU  n={db.shape[0]}: upload {tb*1e3:.2f} ms, first batched 8-NN call {tfirst*1e3:.2f} ms; steady state {tq*1e3:.2f} ms for all points "
          f"({tq*1e3/db.shape[0]:.6f} ms/query; search kernel {ms/max(k,1):.3f} ms)")
    h.radius(db, 1.0)
    tr = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); row, ri, rd = h.radius(db, 1.0); tr = min(tr, time.perf_counter() - t0)
    print(f"     radius r=1.0 of all points {tr*1e3:.2f} ms ({tr*1e3/db.shape[0]:.6f} ms/query), {row[-1]} neighbours ({row[-1]*12/1e6:.0f} MB of results over PCIe)")
    # ONE arbitrary query per call (KDTreeKNNSearch / findNeighbors with a query that is not a database point): wall time per call
    rng = np.random.default_rng(3)
    qs = (db[rng.integers(0, db.shape[0], 400)] + rng.normal(0, 0.2, (400, 3))).astype(np.float32).astype(np.float64)
    for coop in (1, 2):
        ctx.tune("knn_coop", coop)
        h.knn(qs[:1], 8)
        t0 = time.perf_counter()
        for i in range(400):
            h.knn(qs[i:i + 1], 8)
        t1 = (time.perf_counter() - t0) / 400
        t0 = time.perf_counter()
        for i in range(0, 400, 8):
            h.knn(qs[i:i + 8], 8)
        t8 = (time.perf_counter() - t0) / 50
        print(f"     one 8-NN query per call: {t1*1e3:.4f} ms per call; eight per call: {t8*1e3:.4f} ms per call "
              f"({'one wave per query, completion polled' if coop == 1 else 'one lane per query, stream synchronisation (round 2)'}; incl. the Python / ctypes call)")
    ctx.tune("knn_coop", 0)
    try:
        import orc
        if orc.have_ref():
            ridx, rdist, cmp, bms, qms = orc.ref_hw2_kd_knn(db, db, 8, leaf=1, want_cmp=True)
            print(f"CPU  reference hw2 kd-tree (1 thread): build {bms:.2f} ms, 8-NN {qms/db.shape[0]:.6f} ms/query; distances bit-equal: "
                  f"{np.array_equal(rdist.view(np.uint64), dist.view(np.uint64))}")
    except Exception as e:  # noqa: BLE001
        print("reference unavailable:", e)
    ctx.close()

#!/usr/bin/env python3
"""ICP on the REAL KITTI scan subset held as a fixture (tests/golden/kat_kitti_q5.npz: the first 100 000 points of
Homework2/hw2/000000.bin) against its own perturbed copy (1 deg / 0.3 m), grid and brute force.
usage: run_real_scan.py [iters=20]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
pcr = importlib.import_module("hands-on-point-cloud-processing_amd")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
g = np.load(os.path.join(ROOT, "tests", "golden", "kat_kitti_q5.npz"))
tgt = np.ascontiguousarray(g["db_f32"][:, :3].T.astype(np.float32))
n = tgt.shape[1]
rng = np.random.default_rng(5)
a = np.deg2rad(1.0)
R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
src = (R @ tgt[:, rng.permutation(n)].astype(np.float64) + np.array([[0.3], [0.1], [0.02]]) + rng.normal(0, 0.01, (3, n))).astype(np.float32)
src = np.ascontiguousarray(src)
print(f"real scan subset: {n} points, extent {np.ptp(tgt, axis=1)}")
out = {}
for name, method in (("grid", 2), ("brute", 1)):
    ctx = pcr.Context(0); ctx.tune("nn_method", method)
    cs, ct = ctx.cloud(src), ctx.cloud(tgt)
    ctx.icp_point2point(cs, ct, max_corr=1.0, max_iter=2, eps=0.0)
    ctx.sync(); t0 = time.perf_counter()
    T, st = ctx.icp_point2point(cs, ct, max_corr=1.0, max_iter=iters, eps=0.0)
    dt = time.perf_counter() - t0
    out[name] = T
    print(f"{name:6s}: {iters} iterations {dt*1e3:.2f} ms = {dt*1e3/iters:.3f} ms/iter, {n*iters/dt/1e6:.0f} M corr/s, kept {st['last_pairs']}, "
          f"recovered yaw {np.rad2deg(np.arctan2(T[1,0], T[0,0])):+.3f} deg, t = {T[:3,3]}", flush=True)
    ctx.tune("prof", 1)
    for al, c in (("initial pose", cs),):
        ctx.prof_reset(); ctx.nn1_async(ct, c); ctx.nn1_async(ct, c); ctx.sync()
        k, ms = ctx.prof_get("nn1_grid" if method == 2 else "nn1_brute")
        print(f"        one-shot search at the {al}: {ms/max(k,1):.3f} ms")
    ctx.close()
print("grid pose == brute pose (bits):", np.array_equal(out["grid"].view(np.uint32), out["brute"].view(np.uint32)))

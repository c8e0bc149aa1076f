#!/usr/bin/env python3
"""ICPpoint2plane (hw9 registration.cpp:710-860) at the BASELINE config 3 size: 20 iterations on the 120 k pair, target normals
from the k-NN service; per-iteration time with both correspondence searches."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
pcr = importlib.import_module("hands-on-point-cloud-processing_amd")
synth = importlib.import_module("hands-on-point-cloud-processing_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
src, tgt = synth.kitti_like_pair(n)
ctx = pcr.Context(0); ctx.tune("prof", 2)
cs, ct = ctx.cloud(src), ctx.cloud(tgt)
t0 = time.perf_counter(); nrm = ctx.normals(ct, 10, 5.0).astype(np.float32); tn = time.perf_counter() - t0
cn = ctx.cloud(np.ascontiguousarray(nrm.T))
gt = synth.gt_pose()
print(f"target normals (k = 10, r = 5) through the k-NN service: {tn*1e3:.1f} ms")
for name, method in (("brute force", 1), ("exact grid", 2)):
    ctx.tune("nn_method", method)
    ctx.icp_point2plane(cs, ct, cn, max_iter=3, eps=0.0); ctx.prof_reset()
    t0 = time.perf_counter(); T, st = ctx.icp_point2plane(cs, ct, cn, max_corr=1.0, max_iter=20, eps=0.0); dt = time.perf_counter() - t0
    k, ms = ctx.prof_get("p2plane_partial")
    print(f"point-to-plane ICP, {name}: 20 iterations {dt*1e3:.2f} ms = {dt*1e3/20:.3f} ms/iter (normal-equation pass {ms/k*1e3:.1f} us, "
          f"{(12+8+24)*st['last_pairs']/(ms/k*1e-3)/1e9:.0f} GB/s of 44 B/pair), kept {st['last_pairs']}, pose error {np.linalg.norm(T-gt):.2e} "
          f"(point-to-point after 20: see bench)")
T2, st2 = ctx.icp_point2point(cs, ct, max_corr=1.0, max_iter=20, eps=0.0)
print(f"point-to-point ICP, exact grid, 20 iterations: pose error {np.linalg.norm(T2-gt):.2e}")

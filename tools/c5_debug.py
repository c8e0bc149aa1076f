#!/usr/bin/env python3
"""debugging aid: per-query results of the k-th search of a caller-stepped 10 M loop under two tune settings"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
pcr = importlib.import_module("hands-on-point-cloud-processing_amd")
synth = importlib.import_module("hands-on-point-cloud-processing_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
K = int(os.environ.get("K", "3"))
armA = dict(kv.split("=") for kv in (sys.argv[2].split() if len(sys.argv) > 2 else []))
armB = dict(kv.split("=") for kv in (sys.argv[3].split() if len(sys.argv) > 3 else []))
src, tgt = synth.kitti_like_pair(n)
ctx = pcr.Context(0)
ctx.tune("nn_method", 2)
cs, ct = ctx.cloud(src), ctx.cloud(tgt)
def run(arm):
    for k, v in arm.items():
        ctx.tune(k, int(v))
    work = cs.clone()
    ctx.sort_for_target(ct, work)
    res = []
    for it in range(K):
        ctx.nn1_loop(ct, work, 1.0)
        res.append(ctx.nn1_fetch(n) + (ctx.mfma_check()["last_nn1_kernel"],))
        sums, last, _ = ctx.kabsch_sums(ct, work, 1.0)
        rc, R, t = pcr.kabsch_solve(sums)
        Td = np.eye(4, dtype=np.float32); Td[:3, :3], Td[:3, 3] = R, t
        ctx.transform(work, Td)
    cur = work.numpy()
    work.free()
    for k in arm:
        ctx.tune(k, 0)
    return res
A, B = run(armA), run(armB)
for it in range(K):
    ia, da, ka = A[it]; ib, db, kb = B[it]
    bad = np.flatnonzero((ia != ib) | (da.view(np.uint32) != db.view(np.uint32)))
    print(f"search {it}: kernels {ka} / {kb}: {bad.size} queries differ", bad[:10], flush=True)
    for q in bad[:6]:
        print("   q", q, "A:", ia[q], da[q], " B:", ib[q], db[q])
    if bad.size:
        break
ctx.close()

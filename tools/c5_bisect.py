#!/usr/bin/env python3
"""10 M pair: the sign tile search against the cell walk alone, pose bits per arm of tunes (debugging aid)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
pcr = importlib.import_module("hands-on-point-cloud-processing_amd")
synth = importlib.import_module("hands-on-point-cloud-processing_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
its = [int(v) for v in os.environ.get("ITS", "3,4,9").split(",")]
arms = [dict(kv.split("=") for kv in a.split()) for a in sys.argv[2:]] or [{}]
src, tgt = synth.kitti_like_pair(n)
ctx = pcr.Context(0)
ctx.tune("nn_method", 2)
cs = ctx.cloud(src)
ref = {}
ct = ctx.cloud(tgt)
ctx.tune("grid_tile", 2)
for it in its:
    ref[it] = ctx.icp_point2point(cs, ct, max_corr=1.0, max_iter=it, eps=0.0)
ctx.tune("grid_tile", 0)
ct.free()
for arm in arms:
    for k, v in arm.items():
        ctx.tune(k, int(v))
    ct = ctx.cloud(tgt)          # a fresh target: index knobs take effect
    for it in its:
        T, st = ctx.icp_point2point(cs, ct, max_corr=1.0, max_iter=it, eps=0.0)
        ok = np.array_equal(T.view(np.uint32), ref[it][0].view(np.uint32))
        print(arm, "iters", it, "kernel", ctx.mfma_check()["last_nn1_kernel"], "OK" if ok else "MISMATCH", "pairs", st["last_pairs"], ref[it][1]["last_pairs"], flush=True)
    ct.free()
    for k in arm:
        ctx.tune(k, 0)
ctx.close()

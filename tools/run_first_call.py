#!/usr/bin/env python3
"""The first ICP call on a pair nobody has searched yet: HIP-event duration of its searches (the first one is cold: it seeds itself) and the wall time of the
whole 20-iteration call, on a context whose code objects are loaded (a throw-away pair of another size first).  usage: run_first_call.py [n] [key=value ...]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pcr = importlib.import_module("hands-on-point-cloud-processing_amd")
synth = importlib.import_module("hands-on-point-cloud-processing_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
ctx = pcr.Context(0); ctx.tune("nn_method", 1)
for kv in sys.argv[2:]:
    k, v = kv.split("="); ctx.tune(k, int(v))
s0, t0 = synth.kitti_like_pair(n - 7000, seed_target=5, seed_pair=6)
ctx.icp_point2point(ctx.cloud(s0), ctx.cloud(t0), max_corr=1.0, max_iter=4, eps=0.0)
for rep in range(3):
    src, tgt = synth.kitti_like_pair(n, seed_target=100 + rep, seed_pair=200 + rep)
    cs, ct = ctx.cloud(src), ctx.cloud(tgt)
    ctx.nn1(ct, ctx.cloud(src[:, :4096]))                  # the target's index exists (its build is not what this measures)
    ctx.tune("prof", 1); ctx.prof_reset()
    ctx.sync(); w0 = time.perf_counter()
    T, st = ctx.icp_point2point(cs, ct, max_corr=1.0, max_iter=20, eps=0.0)
    wall = (time.perf_counter() - w0) * 1e3
    each = ctx.prof_get_each("nn1_brute"); ctx.tune("prof", 0)
    print(f"pair {rep}: searches {each[0]:.3f} {each[1]:.3f} {each[2]:.3f} ... {each[-1]:.3f} ms; the call {wall:.3f} ms = {wall / 20:.4f} ms per iteration ({ctx.mfma_check()['last_nn1_kernel']})")

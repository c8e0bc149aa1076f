#!/usr/bin/env python3
"""The COLD exhaustive search of an indexed target (BASELINE configs[1] read literally: one 1-NN search, no earlier correspondences):
HTRACK (tune nn1_sign = 2) against STRACK (default), both behind the cold seed of bt_seed_kernel (the best of 32 records of the
nearest super-tile).  HIP-event time of the search scope (seed kernel included), median of 9; keys compared with the exact-only kernel.
usage: run_cold.py [n]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pcr = importlib.import_module("hands-on-point-cloud-processing_amd")
synth = importlib.import_module("hands-on-point-cloud-processing_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
src, tgt = synth.kitti_like_pair(n)
ctx = pcr.Context(0)
ctx.tune("nn_method", 1)
for kv in sys.argv[2:]:
    k__, v__ = kv.split("="); ctx.tune(k__, int(v__))
cs, ct = ctx.cloud(src), ctx.cloud(tgt)
ctx.tune("nn1_variant", 2); ri, rd = ctx.nn1(ct, cs); ctx.tune("nn1_variant", 0)
ctx.nn1(ct, cs)                                              # index, code objects
for label, tunes in (("HTRACK", {"nn1_sign": 2}), ("HTRACK r3 rule", {"nn1_cold_seed": 3}),
                     ("STRACK", {}),
                     ("HTRACK", {"nn1_sign": 2}), ("STRACK", {})):
    for k_, v_ in tunes.items():
        ctx.tune(k_, v_)
    idx, d2 = ctx.nn1(ct, cs)
    same = bool(np.array_equal(idx, ri) and np.array_equal(d2.view(np.uint32), rd.view(np.uint32)))
    ctx.tune("prof", 1); ctx.prof_reset()
    for _ in range(9):
        ctx.nn1_async(ct, cs)
    ctx.sync()
    each = np.sort(ctx.prof_get_each("nn1_brute"))
    ctx.tune("prof", 0)
    ctx.tune("grid_stats", 1); ctx.nn1_async(ct, cs); ctx.sync(); w = ctx.nn1_stats(); ctx.tune("grid_stats", 0)
    print(f"{label:22s} ({ctx.mfma_check()['last_nn1_kernel']}): median {each[len(each) // 2]:.4f} ms (min {each[0]:.4f}) = {n / each[len(each) // 2] / 1e3:.1f} M corr/s; "
          f"chunks evaluated exactly per query {w[6] / n:.2f}; keys = exact-only kernel: {same}")
    for k_ in tunes:
        ctx.tune(k_, 0)

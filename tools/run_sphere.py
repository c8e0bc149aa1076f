#!/usr/bin/env python3
"""STRACK3 (the sign filter over three levels of bounding spheres, default from 32 768 target points) against STRACK (tune nn1_sphere = 2): HIP-event duration of every search of
a 20-iteration brute-force ICP loop, wall time per iteration without event pairs, flagged statistics at the final pose, pose bits.
usage: python tools/run_sphere.py [n] [key=value ...]   (arms: default, nn1_sphere=2, plus the given tunes on top of the default)"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
pcr = importlib.import_module("hands-on-point-cloud-processing_amd")
synth = importlib.import_module("hands-on-point-cloud-processing_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
extra = dict(kv.split("=") for kv in sys.argv[2:])
src, tgt = synth.kitti_like_pair(n)
ctx = pcr.Context(0)
ctx.tune("nn_method", 1)
cs, ct = ctx.cloud(src), ctx.cloud(tgt)
arms = [("STRACK3 (default)", {}), ("STRACK (nn1_sphere=2)", {"nn1_sphere": 2})]
if extra:
    arms.append((f"STRACK3 + {extra}", {k: int(v) for k, v in extra.items()}))
ref = None
for name, tunes in arms + arms[:2]:
    for k, v in tunes.items():
        ctx.tune(k, v)
    ctx.icp_point2point(cs, ct, max_corr=1.0, max_iter=3, eps=0.0)
    ctx.tune("prof", 1); ctx.prof_reset()
    T, st = ctx.icp_point2point(cs, ct, max_corr=1.0, max_iter=20, eps=0.0)
    each = ctx.prof_get_each("nn1_brute")
    ctx.tune("prof", 0)
    best = None
    for _ in range(3):
        ctx.sync(); t0 = time.perf_counter()
        ctx.icp_point2point(cs, ct, max_corr=1.0, max_iter=20, eps=0.0)
        dt = (time.perf_counter() - t0) * 1e3 / 20
        best = dt if best is None else min(best, dt)
    bits = "".join(f"{int(v):08x}" for v in T.view(np.uint32).ravel())
    ref = ref or bits
    ctx.tune("grid_stats", 1)
    ctx.icp_point2point(cs, ct, init_T=T, max_corr=1.0, max_iter=9, eps=0.0)
    w = ctx.nn1_stats()
    ctx.tune("grid_stats", 0)
    print(f"{name:28s} kernel {ctx.mfma_check()['last_nn1_kernel']}: search ms", " ".join(f"{v:.3f}" for v in each[:4]), "...", " ".join(f"{v:.3f}" for v in each[-3:]),
          f"| mean {each[1:].mean():.4f} | wall {best:.4f} ms/iteration = {n / best / 1e3:.0f} M corr/s | pose bits {'same' if bits == ref else 'DIFFERENT'}"
          f" | evals/q {w[6] / n:.2f} l0 mfma {w[7]} l1 tiles flagged {w[3]} l1 mfma {w[8]} l2 tiles flagged {w[9]} l2 mfma {w[10]} | waves {w[13]} life us: mean {w[12] / max(w[13], 1) / 100:.1f} max {w[11] / 100:.1f}; most l2 mfma in a wave {w[14]}, waves over 200: {w[15]}")
    for k in tunes:
        ctx.tune(k, 0)
ctx.close()

#!/usr/bin/env python3
"""Steady state of the 10 M search at the converged pose, one number per library / tune arm: the searches of a loop that starts at the
final pose (the cold first one and the first seeded one skipped).  Timing builds (PCR_SL_T_*) give wrong answers by design: only the
duration counts.   usage: python tools/c5_conv.py [n] [key=value ...]   (PCR_LIB_PATH selects the library)"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
pcr = importlib.import_module("hands-on-point-cloud-processing_amd")
synth = importlib.import_module("hands-on-point-cloud-processing_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
tunes = dict(kv.split("=") for kv in sys.argv[2:])
src, tgt = synth.kitti_like_pair(n)
T = synth.gt_pose().astype(np.float32)
ctx = pcr.Context(0)
ctx.tune("nn_method", 2)
for k, v in tunes.items():
    ctx.tune(k, int(v))
cs, ct = ctx.cloud(src), ctx.cloud(tgt)
ctx.icp_point2point(cs, ct, init_T=T, max_corr=1.0, max_iter=2, eps=0.0)
ctx.tune("prof", 1)
out = []
for _ in range(3):
    ctx.prof_reset()
    ctx.icp_point2point(cs, ct, init_T=T, max_corr=1.0, max_iter=8, eps=0.0)
    each = ctx.prof_get_each("nn1_grid")
    out.append(float(np.mean(each[2:])))
print(f"{os.environ.get('PCR_LIB_PATH', 'default').split('/')[-1]:24s} {tunes} steady search {np.median(out):.3f} ms  ({ctx.mfma_check()['last_nn1_kernel']})")
ctx.close()

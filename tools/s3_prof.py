#!/usr/bin/env python3
"""Where a STRACK3 wave's life goes (profile build: tools/ab_build.sh prof nn1_brute.hip -DPCR_S2_PROF; PCR_LIB_PATH=.../libpcr_prof.so): stamps of one
wave in sixteen, last search of a 9-iteration loop from the final pose of a 20-iteration one.  usage: s3_prof.py [n] [key=value ...]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pcr = importlib.import_module("hands-on-point-cloud-processing_amd")
synth = importlib.import_module("hands-on-point-cloud-processing_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
src, tgt = synth.kitti_like_pair(n)
ctx = pcr.Context(0); ctx.tune("nn_method", 1)
for kv in sys.argv[2:]:
    k, v = kv.split("="); ctx.tune(k, int(v))
cs, ct = ctx.cloud(src), ctx.cloud(tgt)
T, st = ctx.icp_point2point(cs, ct, max_corr=1.0, max_iter=20, eps=0.0)
ctx.tune("grid_stats", 1); ctx.tune("prof", 1); ctx.prof_reset()
ctx.icp_point2point(cs, ct, init_T=T, max_corr=1.0, max_iter=9, eps=0.0)
print("search ms:", " ".join(f"{v:.4f}" for v in ctx.prof_get_each("nn1_brute")))
w = ctx.nn1_stats(); nw = max(w[5], 1)
print(f"waves sampled {w[5]}: mean life {w[6] / nw / 100:.2f} us (longest {w[7] / 100:.2f}) = prologue + level-0 setup {w[0] / nw / 100:.2f} + level 0 {w[1] / nw / 100:.2f} + level 1 {w[2] / nw / 100:.2f} "
      f"+ level 2 and evaluation {w[3] / nw / 100:.2f} + epilogue {w[4] / nw / 100:.2f}; lives by 5 us bins {w[8:16]}")

#!/usr/bin/env python3
"""Where a STRACK2 wave's life goes (profile build: tools/ab_build.sh prof nn1_brute.hip -DPCR_S2_PROF; PCR_LIB_PATH=.../libpcr_prof.so)."""
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
ctx.tune("prof", 1); ctx.prof_reset()
ctx.icp_point2point(cs, ct, max_corr=1.0, max_iter=20, eps=0.0)
print("search ms of a plain loop (last 3):", ctx.prof_get_each("nn1_brute")[-3:])
ctx.prof_reset()
ctx.icp_point2point(cs, ct, init_T=T, max_corr=1.0, max_iter=9, eps=0.0)
print("search ms of the 9-iteration loop from the final pose, no diagnostics:", ctx.prof_get_each("nn1_brute"))
ctx.tune("grid_stats", 1)
ctx.tune("prof", 1); ctx.prof_reset()
ctx.icp_point2point(cs, ct, init_T=T, max_corr=1.0, max_iter=9, eps=0.0)
print("search ms of the stats loop:", ctx.prof_get_each("nn1_brute"))
w = ctx.nn1_stats()
nw = max(w[3], 1)
print(f"waves {w[3]}: mean life {w[13] / nw / 100:.1f} us (max {w[12] / 100:.1f}); setup {w[0] / nw / 100:.1f} us, level 1 {w[1] / nw / 100:.1f} us (max {w[15] / 100:.1f}), level 2 + evaluation {w[2] / nw / 100:.1f} us (max {w[14] / 100:.1f}); lives by 10 us bins {w[4:12]}")

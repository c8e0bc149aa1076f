#!/usr/bin/env python3
"""Where a wave of the sign tile search spends its life at the settled pose of the 10 M pair (profile build: tools/ab_build.sh slprof grid.hip -DPCR_SL_PROF;
PCR_LIB_PATH=.../libpcr_slprof.so): one wave in 16 stamps its phases with s_memrealtime.  usage: stile_prof.py [n] [key=value ...]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pcr = importlib.import_module("hands-on-point-cloud-processing_amd")
synth = importlib.import_module("hands-on-point-cloud-processing_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
src, tgt = synth.kitti_like_pair(n)
T = synth.gt_pose().astype(np.float32)
ctx = pcr.Context(0); ctx.tune("nn_method", 2)
for kv in sys.argv[2:]:
    k, v = kv.split("="); ctx.tune(k, int(v))
cs, ct = ctx.cloud(src), ctx.cloud(tgt)
ctx.icp_point2point(cs, ct, init_T=T, max_corr=1.0, max_iter=6, eps=0.0)
ctx.tune("prof", 1); ctx.prof_reset()
ctx.tune("grid_stats", 1)
ctx.icp_point2point(cs, ct, init_T=T, max_corr=1.0, max_iter=6, eps=0.0)
print("search ms:", " ".join(f"{v:.3f}" for v in ctx.prof_get_each("nn1_grid")), ctx.mfma_check()["last_nn1_kernel"])
w = ctx.nn1_stats(); nw = max(w[4], 1)
print(f"waves sampled {w[4]}: mean life {w[5] / nw / 100:.1f} us (longest {w[6] / 100:.1f}) = prologue + seed run {w[0] / nw / 100:.1f} + boxes, cells, spheres, list {w[1] / nw / 100:.1f} "
      f"+ tile loops and evaluations {w[2] / nw / 100:.1f} + write-back and deferral {w[3] / nw / 100:.1f}")

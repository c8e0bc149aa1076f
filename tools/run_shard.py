#!/usr/bin/env python3
"""One rank's shard of the 10 M pair on this GPU: kernel family and HIP-event duration of every search of a loop.
usage: python tools/run_shard.py [n] [nranks] [rank] [spatial|contiguous] [key=value ...]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
pcr = importlib.import_module("hands-on-point-cloud-processing_amd")
synth = importlib.import_module("hands-on-point-cloud-processing_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
r = int(sys.argv[3]) if len(sys.argv) > 3 else 0
how = sys.argv[4] if len(sys.argv) > 4 else "spatial"
tunes = dict(kv.split("=") for kv in sys.argv[5:])
src, tgt = synth.kitti_like_pair(n)
ctx = pcr.Context(0)
ctx.tune("nn_method", 2)
for k, v in tunes.items():
    ctx.tune(k, int(v))
ct = ctx.cloud(tgt)
if how == "spatial":
    full = ctx.cloud(src); cs = ctx.shard_spatial(ct, full, N, r); full.free()
else:
    b, e = pcr.shard_range(n, N, r); cs = ctx.cloud(np.ascontiguousarray(src[:, b:e]))
T = synth.gt_pose().astype(np.float32)
ctx.icp_point2point(cs, ct, max_corr=1.0, max_iter=2, eps=0.0)
ctx.tune("prof", 1)
for init, name in ((None, "from the start pose"), (T, "from the final pose")):
    ctx.prof_reset()
    ctx.icp_point2point(cs, ct, init_T=init, max_corr=1.0, max_iter=12, eps=0.0)
    each = ctx.prof_get_each("nn1_grid")
    print(f"{how} shard {r}/{N} ({len(cs)} queries) {name}: kernel {ctx.mfma_check()['last_nn1_kernel']}; nn1 ms", " ".join(f"{v:.3f}" for v in each))
ctx.tune("prof", 0)
best = 1e9
for _ in range(3):
    ctx.sync(); t0 = time.perf_counter(); ctx.icp_point2point(cs, ct, init_T=T, max_corr=1.0, max_iter=40, eps=0.0); best = min(best, (time.perf_counter() - t0) * 1e3 / 40)
print(f"whole iteration from the final pose (wall of a 40-iteration call / 40, no event pairs): {best:.4f} ms")
ctx.tune("grid_stats", 1)
ctx.icp_point2point(cs, ct, init_T=T, max_corr=1.0, max_iter=3, eps=0.0)
w = ctx.nn1_stats(); nq = len(cs)
print(f"stats: cand/q {w[0] / nq:.1f} cells/q {w[1] / nq:.3f} spheres/q {w[2] / nq:.2f} setups/wave {w[3] / (nq / 64):.1f} deferred {w[6]} passes/wave {w[7] / (nq / 64):.2f} evals/q {w[8] / nq:.2f} mfma/wave {w[11] / (nq / 64):.1f} max tiles/pass {w[12]}; deferred because: beyond the ball limit {w[13]}, fourth cluster {w[14]}, pass overflow {w[15]}")
ctx.close()

#!/usr/bin/env python3
"""Per-iteration picture of the 10 M ICP (BASELINE configs[4], one GPU): the HIP-event duration of every search of a loop (median of
three runs of the same loop) — and, with STATS=1, the diagnostics of a search at the converged pose.  A/B over tunes given as KEY=VALUE arguments, e.g.
    run_c5_iters.py 10000000 20 grid_tile=2      (the cell walk alone)
    run_c5_iters.py 10000000 20 grid_stile_bmax_cm=50
Prints: nn1 ms per iteration (1 .. iters), the average over the first 10 / 20, wall per iteration of the 20-iteration run, pose bits."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
pcr = importlib.import_module("hands-on-point-cloud-processing_amd")
synth = importlib.import_module("hands-on-point-cloud-processing_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
tunes = dict(kv.split("=") for kv in sys.argv[3:])
src, tgt = synth.kitti_like_pair(n)
if os.environ.get("SHARD"):           # what ONE rank of an N-rank run holds: a contiguous block of the source (pcr_shard_range), the whole target
    if os.environ.get("SHARD_SPATIAL"):   # a spatially compact shard instead: the source ordered along x first (a slab keeps the scene's local density)
        src = np.ascontiguousarray(src[:, np.argsort(src[0], kind="stable")])
    b, e = pcr.shard_range(n, int(os.environ["SHARD"]), 0)
    src = np.ascontiguousarray(src[:, b:e])
    print(f"source shard 1 / {os.environ['SHARD']}: {src.shape[1]} queries against {n} targets")
nq = src.shape[1]
ctx = pcr.Context(0)
ctx.tune("nn_method", 2)
for k, v in tunes.items():
    ctx.tune(k, int(v))
cs, ct = ctx.cloud(src), ctx.cloud(tgt)
ctx.icp_point2point(cs, ct, max_corr=1.0, max_iter=2, eps=0.0)           # index, code objects
ctx.tune("prof", 1)
reps = []
for _ in range(3):
    ctx.prof_reset()
    T, st = ctx.icp_point2point(cs, ct, max_corr=1.0, max_iter=iters, eps=0.0)
    reps.append(ctx.prof_get_each("nn1_grid"))
    assert reps[-1].size == iters, reps[-1].size
per = list(np.median(np.stack(reps), axis=0))
print("tunes", tunes, "kernel", ctx.mfma_check()["last_nn1_kernel"])
print("nn1 ms per iteration:", " ".join(f"{v:.2f}" for v in per))
print(f"avg first 10: {sum(per[:10]) / min(10, iters):.3f} ms   avg first 20: {sum(per[:20]) / min(20, iters):.3f} ms   last: {per[-1]:.3f} ms")
ctx.tune("prof", 0)
t0 = time.perf_counter(); T, st = ctx.icp_point2point(cs, ct, max_corr=1.0, max_iter=iters, eps=0.0); dt = time.perf_counter() - t0
print(f"ICP {iters} iterations: {dt * 1e3 / iters:.3f} ms per iteration (wall, no event pairs) = {nq * iters / dt / 1e6:.0f} M corr/s; kept {st['last_pairs']}; pose bits",
      "".join(f"{int(v):08x}" for v in T.view(np.uint32).ravel())[:48], "...")
if os.environ.get("STATS_AT"):
    # diagnostics of the LAST search of a k-iteration loop from the identity (what iteration k of the timed loop does)
    ctx.tune("grid_stats", 1)
    for k in [int(v) for v in os.environ["STATS_AT"].split(",")]:
        ctx.icp_point2point(cs, ct, max_corr=1.0, max_iter=k, eps=0.0)
        w = ctx.nn1_stats()
        print(f"search {k}: cand/q {w[0] / nq:.1f} spheres/q {w[2] / nq:.1f} deferred {100.0 * w[6] / nq:.1f} % passes/group {w[7] / (nq / 32):.2f} filter passes {w[8]} unsettled {w[9]} setups {w[3]}; deferred because: beyond the ball limit {w[13]}, more clusters than passes {w[14]}, pass overflow {w[15]}; most tiles in a pass {w[12]}")
    ctx.tune("grid_stats", 0)
if os.environ.get("STATS"):
    ca = cs.clone(); ctx.transform(ca, T)
    ctx.tune("prof", 1); ctx.tune("nn1_async_in_loop", 1)
    # (the tile search needs the loop's sorted working cloud: a one-iteration ICP from the converged pose shows its steady state)
    ctx.tune("grid_stats", 1)
    T1, st1 = ctx.icp_point2point(ca, ct, max_corr=1.0, max_iter=3, eps=0.0)
    w = ctx.nn1_stats()
    print(f"last search of a 3-iteration loop from the converged pose: cand/q {w[0] / nq:.1f} rows/q {w[1] / nq:.2f} spheres/q {w[2] / nq:.1f} deferred {w[6]} ({100.0 * w[6] / nq:.2f} %) passes/group {w[7] / (nq / 32):.2f} box edge {w[4] / max(w[7], 1) / 1e4:.2f} cm ball max {w[5] / max(w[7], 1) / 1e4:.2f} cm (per served pass, upper estimates); filter passes {w[8]} of which unsettled {w[9]}")
ctx.close()

#!/usr/bin/env python3
"""Launch the brute-force 1-NN kernel a few times on the BASELINE pair (for rocprofv3 runs).
usage: run_nn1.py [n] [launches] [variant] [unused] [tiles_per_slice] [order: scan|shuffle]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pcr = importlib.import_module("hands-on-point-cloud-processing_amd")
synth = importlib.import_module("hands-on-point-cloud-processing_amd.synth")
a = sys.argv[1:]
n = int(a[0]) if len(a) > 0 else 120000
launches = int(a[1]) if len(a) > 1 else 5
variant = int(a[2]) if len(a) > 2 else None   # None: library default
qpl = int(a[3]) if len(a) > 3 else 0
tps = int(a[4]) if len(a) > 4 else 0
order = a[5] if len(a) > 5 else "scan"
src, tgt = synth.kitti_like_pair(n)
if order == "shuffle":
    perm = np.argsort(synth.splitmix64(4242, np.arange(n, dtype=np.uint64)), kind="stable")
    tgt = np.ascontiguousarray(tgt[:, perm])
ctx = pcr.Context(0)
if variant is not None:
    ctx.tune("nn1_variant", variant)
ctx.tune("nn1_tiles_per_slice", tps)
method = int(os.environ.get("NN_METHOD", "1"))     # 1 = brute force, 2 = exact grid
ctx.tune("nn_method", method)
if os.environ.get("GRID_MODE"):     # 1 = plain kernel, 2 = x-window kernel, 3 = bounding spheres (0 / unset: by target size / index order)
    ctx.tune("grid_mode", int(os.environ["GRID_MODE"]))
if os.environ.get("GRID_CELL_UM"):
    ctx.tune("grid_cell_um", int(os.environ["GRID_CELL_UM"]))
for kv in os.environ.get("PCR_TUNE", "").split(","):      # any other knob: PCR_TUNE="key=value,..."
    if "=" in kv:
        k_, v_ = kv.split("="); ctx.tune(k_, int(v_))
aligned = os.environ.get("ALIGNED", "0") == "1"
cs, ct = ctx.cloud(src), ctx.cloud(tgt)
if aligned:
    ctx.transform(cs, synth.gt_pose().astype(np.float32))
ctx.tune("prof", 2)
ctx.nn1_async(ct, cs); ctx.sync(); print("first call:", {nm: ctx.prof_get(nm) for nm in ("grid_build",) if ctx.prof_get(nm)[0]}); ctx.prof_reset()
if os.environ.get("IN_LOOP"):       # repeated searches seeded by the previous result, as inside a converged ICP loop (grid: record-position warm start)
    ctx.tune("nn1_async_in_loop", 1)
    ctx.nn1_async(ct, cs); ctx.sync(); ctx.prof_reset()
if os.environ.get("ICP_LOOP"):      # inside an ICP loop: iterations after the first run the warm-start kernel (ETRACK for brute force)
    ctx.icp_point2point(cs, ct, max_corr=1.0, max_iter=launches, eps=0.0)
else:
    for _ in range(launches):
        ctx.nn1_async(ct, cs)
if os.environ.get("GRID_STATS"):
    ctx.tune("grid_stats", 1); ctx.nn1_async(ct, cs); ctx.sync(); gs = ctx.grid_stats(); ctx.tune("grid_stats", 0)
    print("grid stats per query:", {k: round(v / n, 2) for k, v in gs.items()})
names = ["nn1_brute", "nn1_grid", "grid_sort_queries", "grid_build"]
ctx.tune("prof", 2)
out = {nm: ctx.prof_get(nm) for nm in names}
desc = ", ".join(f"{nm}: {ms/k:.4f} ms x{k}" for nm, (k, ms) in out.items() if k)
print(f"n={n} method={method} aligned={aligned} variant={variant} qpl={qpl} tps={tps} order={order}: {desc}")

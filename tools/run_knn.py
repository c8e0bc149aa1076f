#!/usr/bin/env python3
"""Grid k-NN service + PCA normals (next row N1, second consumer): kernel times at 120 k points next to the brute-force
f64 k-NN kernel and the CPU oracle on a sample."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
pcr = importlib.import_module("hands-on-point-cloud-processing_amd")
synth = importlib.import_module("hands-on-point-cloud-processing_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
scan = synth.kitti_like_scan(n)
ctx = pcr.Context(0)
ctx.tune("prof", 2)
c = ctx.cloud(scan)
for scale in ([int(a) for a in sys.argv[2:]] or [0]):
    ctx.tune("knn_cell_scale_x100", scale)
    for k in (1, 8, 10, 16, 32):
        ctx.cloud_knn(c, c, k); ctx.prof_reset()
        t0 = time.perf_counter(); idx, s, found = ctx.cloud_knn(c, c, k); dt = time.perf_counter() - t0
        kk, ms = ctx.prof_get("knn_grid")
        print(f"grid k-NN k={k:2d} (cell scale {scale or 'auto'}), {n} x {n}: kernel {ms/kk:.3f} ms -> {n/(ms/kk*1e-3)/1e6:.1f} M queries/s; call incl. grid + D2H {dt*1e3:.1f} ms", flush=True)
ctx.tune("knn_cell_scale_x100", 0)
ctx.normals(c, 10, 5.0); ctx.prof_reset()
t0 = time.perf_counter(); nrm = ctx.normals(c, 10, 5.0); dt = time.perf_counter() - t0
k1, m1 = ctx.prof_get("knn_grid"); k2, m2 = ctx.prof_get("normals_pca")
print(f"normals (hybrid k=10, r=5): knn {m1/k1:.3f} ms + PCA {m2/k2:.3f} ms, call {dt*1e3:.1f} ms; |n|=1 for {(np.abs(np.linalg.norm(nrm,axis=1)-1)<1e-9).mean()*100:.1f}% of points")
db64 = np.ascontiguousarray(scan.T.astype(np.float64))
h = ctx.db64(db64)
t0 = time.perf_counter(); bi, bd = h.knn(db64, 8); dt = time.perf_counter() - t0
kb, mb = ctx.prof_get("knn_f64")
print(f"brute-force f64 k-NN k=8 (same contract, non-squared): kernel {mb/max(kb,1):.1f} ms, call {dt*1e3:.1f} ms")
gi, gd, _ = ctx.cloud_knn(c, c, 8, squared=False)
print("  grid == brute force (indices, distance bits):", np.array_equal(gi, bi), np.array_equal(gd.view(np.uint64), bd.view(np.uint64)))
try:
    import orc
    m = 2000
    t0 = time.perf_counter(); orc.knn_sq_f32pts(scan, np.ascontiguousarray(scan[:, :m]), 10); tc = (time.perf_counter() - t0) * n / m
    print(f"  CPU oracle brute force (1 thread), extrapolated to {n} queries: {tc:.1f} s")
except Exception as e:  # noqa: BLE001
    print("  oracle unavailable:", e)

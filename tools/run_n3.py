#!/usr/bin/env python3
"""Next row N3: voxel-grid down-sampling (Homework1 voxel_filter.py) — GPU call next to the CPU oracle and, in the build
container, the reference function itself is what tests/golden/voxel_filter_hw1.npz was generated with."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
pcr = importlib.import_module("hands-on-point-cloud-processing_amd")
synth = importlib.import_module("hands-on-point-cloud-processing_amd.synth")
ctx = pcr.Context(0)
ctx.tune("prof", 2)
for n, leaf in ((120000, 0.3), (120000, 1.0), (2000000, 0.3), (10000000, 0.3)):
    scan = synth.kitti_like_scan(n)
    c = ctx.cloud(scan)
    ctx.voxel_filter(c, leaf).free(); ctx.sync()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        f = ctx.voxel_filter(c, leaf)
        m = len(f); f.free()
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    line = f"voxel_filter n={n} leaf={leaf}: {m} voxels, call {dt*1e3:.2f} ms ({n/dt/1e6:.0f} M points/s, result stays in HBM)"
    if n <= 2000000:
        import orc
        t0 = time.perf_counter(); want = orc.voxel_filter_f32(scan, leaf); tc = time.perf_counter() - t0
        got = ctx.voxel_filter(c, leaf).numpy()
        line += f"; CPU oracle (1 thread) {tc*1e3:.0f} ms, bit-equal: {got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))}"
    print(line, flush=True)
    c.free()

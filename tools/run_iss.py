#!/usr/bin/env python3
"""Next row N1: ISS keypoints (Homework7/hw7) — GPU time per pass next to the CPU oracle / the reference's kd-tree."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
pcr = importlib.import_module("hands-on-point-cloud-processing_amd")
synth = importlib.import_module("hands-on-point-cloud-processing_amd.synth")
ctx = pcr.Context(0)
ctx.tune("prof", 2)
g = np.load(os.path.join(ROOT, "tests", "golden", "iss_hw7.npz"))
cases = [("airplane_0001 10k (hw7 driver radii)", np.ascontiguousarray(g["xyz_airplane_0001"].T), float(g["local_r"]), float(g["nms_r"])),
         ("kitti-like 120k r=1.2/0.8", synth.kitti_like_scan(120000), 1.2, 0.8),
         ("kitti-like 120k r=0.6/0.4", synth.kitti_like_scan(120000), 0.6, 0.4)]
lanes_list = [int(a) for a in sys.argv[1:]] or [8]
for name, soa, rl, rn in cases:
    c = ctx.cloud(soa)
    for lanes in lanes_list:
        ctx.tune("iss_lanes", lanes)
        ctx.iss_keypoints(c, rl, rn); ctx.prof_reset()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            idx, l3, cnt = ctx.iss_keypoints(c, rl, rn)
        dt = (time.perf_counter() - t0) / reps
        parts = {k: ctx.prof_get(k) for k in ("iss_grid_build", "iss_count", "iss_cov", "iss_eig", "iss_nms")}
        pairs = float(cnt.astype(np.float64).sum())
        ms = {k: (v[1] / max(v[0], 1)) for k, v in parts.items()}
        print(f"{name}: lanes {lanes}: call {dt*1e3:.2f} ms (grid {ms['iss_grid_build']:.2f} count {ms['iss_count']:.3f} cov {ms['iss_cov']:.3f} eig {ms['iss_eig']:.3f} nms {ms['iss_nms']:.3f}), "
              f"{idx.size} keypoints, mean |N| {cnt.mean():.1f}, {pairs/ (ms['iss_count']*1e-3)/1e9:.2f} G neighbours/s in pass 1", flush=True)
    if soa.shape[1] <= 20000:
        import orc
        t0 = time.perf_counter(); okey, ol3 = orc.iss_f32(soa, rl, rn); tc = time.perf_counter() - t0
        print(f"  CPU oracle (brute force, 1 thread): {tc*1e3:.0f} ms; keypoints equal: {np.array_equal(np.flatnonzero(okey), idx)}")
        if orc.have_hw7():
            pts = np.ascontiguousarray(soa.T)
            t0 = time.perf_counter(); orc.ref_hw7_radius(pts, pts, rl); orc.ref_hw7_radius(pts, pts, rn); tr = time.perf_counter() - t0
            print(f"  reference hw7 kd-tree, both radius passes only (1 thread, each run twice by the harness): {tr*1e3/2:.0f} ms")

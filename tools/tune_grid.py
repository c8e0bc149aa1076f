#!/usr/bin/env python3
"""Sweep lanes-per-query and occupancy target of the exact grid 1-NN on the BASELINE pair (misaligned + aligned)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pcr = importlib.import_module("hands-on-point-cloud-processing_amd")
synth = importlib.import_module("hands-on-point-cloud-processing_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
src, tgt = synth.kitti_like_pair(n)
ctx = pcr.Context(0)
ctx.tune("nn_method", 2)
for aligned in (0, 1):
    for occ in (10, 20, 40, 80):
        for lanes in (1, 4, 8, 16):
            ctx.tune("grid_occupancy_x10", occ); ctx.tune("grid_lanes", lanes)
            cs, ct = ctx.cloud(src), ctx.cloud(tgt)
            if aligned:
                ctx.transform(cs, synth.gt_pose().astype(np.float32))
            ctx.nn1_async(ct, cs); ctx.sync(); ctx.prof_reset()
            for _ in range(10):
                ctx.nn1_async(ct, cs)
            k, ms = ctx.prof_get("nn1_grid"); k2, ms2 = ctx.prof_get("grid_sort_queries")
            ctx.prof_reset()
            ct2 = ctx.cloud(tgt); ctx.nn1_async(ct2, cs); kb, msb = ctx.prof_get("grid_build")
            print(f"aligned={aligned} occ={occ/10:.1f} lanes={lanes:2d}: nn1_grid {ms/k*1e3:8.1f} us  sort_q {ms2/max(k2,1)*1e3:6.1f} us  build {msb/max(kb,1)*1e3:7.1f} us", flush=True)
            cs.free(); ct.free(); ct2.free()

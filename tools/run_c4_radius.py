#!/usr/bin/env python3
"""configs[3] radius leg alone, for --pmc passes: every point of the 120 000-point scan queries its scan with r = 1 (count-only call, then the
filled call); the kernels of the LAST filled call are what tools/summarize_c4_pmc.py sums."""
import importlib, os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
pcr = importlib.import_module("hands-on-point-cloud-processing_amd")
synth = importlib.import_module("hands-on-point-cloud-processing_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
scan = synth.kitti_like_scan(n)
db = np.ascontiguousarray(scan.T.astype(np.float64))
ctx = pcr.Context(0)
d = ctx.db64(db)
L = pcr.lib()
row = np.zeros(n + 1, np.int64)
L.pcr_db64_radius(ctx.h, d.h, db.ctypes.data, n, C.c_double(1.0), row.ctypes.data, None, None)
total = int(row[-1])
idx = np.zeros(total, np.int32); dist = np.zeros(total, np.float64)
for _ in range(2):
    rc = L.pcr_db64_radius(ctx.h, d.h, db.ctypes.data, n, C.c_double(1.0), row.ctypes.data, idx.ctypes.data, dist.ctypes.data)
    assert rc == 0
print("neighbours", total)
ctx.close()

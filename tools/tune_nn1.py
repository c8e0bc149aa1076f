#!/usr/bin/env python3
"""Sweep the launch geometry of the brute-force 1-NN kernel on the BASELINE 120k x 120k pair.
Prints one line per configuration: median HIP-event kernel time over `reps` interleaved rounds."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pcr = importlib.import_module("hands-on-point-cloud-processing_amd")
synth = importlib.import_module("hands-on-point-cloud-processing_amd.synth")

n = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
src, tgt = synth.kitti_like_pair(n)
ctx = pcr.Context(0)
ctx.tune("nn_method", 1)
cs, ct = ctx.cloud(src), ctx.cloud(tgt)
configs = []
for var in ((1, 8), (1, 16), (2, 16), (3, 16)):
    for qpl in (1, 2, 4):
        for tps in (2, 4, 8):
            configs.append((var, qpl, tps))
res = {c: [] for c in configs}
def setcfg(c):
    var, qpl, tps = c
    ctx.tune("nn1_variant", var[0]); ctx.tune("nn1_chunk", var[1]); ctx.tune("nn1_qpl", qpl); ctx.tune("nn1_tiles_per_slice", tps)
for c in configs:      # warm
    setcfg(c); ctx.nn1_async(ct, cs)
ctx.sync()
for r in range(reps):
    for c in configs:
        setcfg(c)
        ctx.prof_reset()
        ctx.nn1_async(ct, cs)
        k, ms = ctx.prof_get("nn1_brute")
        res[c].append(ms)
print("lib", pcr.LIB_PATH)
for (var, qpl, tps), v in sorted(res.items(), key=lambda kv: np.median(kv[1])):
    med = float(np.median(v))
    print(f"variant={var} qpl={qpl} tps={tps:4d} median={med:.4f} ms min={min(v):.4f}  -> {9*n*n/med/1e9:.1f} TFLOP/s(9-op)  {n/med/1e3:.1f} M corr/s")

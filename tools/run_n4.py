#!/usr/bin/env python3
"""Next row N4: hw9 global-registration front half — descriptor matching + RANSAC consensus at the driver's size
(80 000 iterations, main.cpp:86), GPU kernels timed next to the CPU oracle / the reference's nanoflann."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
pcr = importlib.import_module("hands-on-point-cloud-processing_amd")
from test_global_registration import scene
ctx = pcr.Context(0)
ctx.tune("prof", 2)
src, tgt, dsrc, dtgt, R, t = scene(41, 3000, 2500, 900)
ctx.match_union(dsrc, dtgt, 0.5); ctx.prof_reset()
t0 = time.perf_counter(); pairs, dist = ctx.match_union(dsrc, dtgt, 0.5); dt = time.perf_counter() - t0
k, ms = ctx.prof_get("nn1_desc")
pe = 2 * 3000 * 2500
print(f"match_union 3000 x 2500 x 33-D: call {dt*1e3:.2f} ms, two NN kernels {ms/k*1e3:.1f} us each -> {pe/2/(ms/k*1e-3)/1e9:.1f} G pair-evals/s, {pairs.shape[0]} pairs kept", flush=True)
t0 = time.perf_counter(); quads = pcr.ransac_sample_quads(src, pairs, 80000, 2020); ts = time.perf_counter() - t0
ctx.ransac_global(src, tgt, pairs, quads, 0.3); ctx.prof_reset()
reps = 5
t0 = time.perf_counter()
for _ in range(reps):
    win, Rg, tg, best, counts = ctx.ransac_global(src, tgt, pairs, quads, 0.3)
dt = (time.perf_counter() - t0) / reps
kh, msh = ctx.prof_get("ransac_hypotheses"); kc, msc = ctx.prof_get("consensus_count")
ev = 80000 * pairs.shape[0]
print(f"RANSAC 80000 hypotheses x {pairs.shape[0]} correspondences: sampling (host) {ts*1e3:.1f} ms, call {dt*1e3:.2f} ms: hypotheses kernel {msh/kh*1e3:.1f} us, "
      f"consensus kernel {msc/kc*1e3:.1f} us -> {ev/(msc/kc*1e-3)/1e12:.2f} T evals/s ({27*ev/(msc/kc*1e-3)/1e12:.1f} T lane-ops/s at 27 ops/eval); best {best}, pose error {np.linalg.norm(Rg-R):.2e}", flush=True)
try:
    import orc
    t0 = time.perf_counter(); orc.match_union_f32(dsrc, dtgt, 0.5); tm = time.perf_counter() - t0
    t0 = time.perf_counter(); ow = orc.ransac_global_f32(src, tgt, pairs, quads[:4000], 0.3); tr = (time.perf_counter() - t0) * 20
    print(f"  CPU oracle (1 thread): match_union {tm*1e3:.0f} ms; RANSAC {tr*1e3:.0f} ms (extrapolated from 4000 hypotheses); counts equal on those: {np.array_equal(ow[4], counts[:4000])}")
    if orc.have_ref():
        t0 = time.perf_counter(); orc.ref_nano_nn1_dim_f32(dsrc, dtgt); orc.ref_nano_nn1_dim_f32(dtgt, dsrc); tn = time.perf_counter() - t0
        print(f"  reference nanoflann, both directions (1 thread): {tn*1e3:.0f} ms")
except Exception as e:  # noqa: BLE001
    print("  oracle unavailable:", e)

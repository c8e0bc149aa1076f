#!/usr/bin/env python3
"""One-off soak of the correspondence kernels on random inputs: the default path (FTRACK on a target's first search, the bf16
matrix-core filters HTRACK / BTRACK from 8 192 target points on), the cold ETRACK, BTRACK and HTRACK searches, the exact grid search and the tile
search of large-target loops (forced onto these pairs, with its default limits and with limits so tight that most passes overflow and are
deferred) must give the bits of the exact-only brute-force kernel (nn1_variant = 2) on every input.
usage: soak_nn1.py [cases=60] [seed0=1]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pcr = importlib.import_module("hands-on-point-cloud-processing_amd")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1


def make(rng):
    kind = rng.integers(0, 7)
    nt = int(rng.choice([2048, 3000, 4097, 8192, 20000, 50000]))
    ns = int(rng.choice([1, 63, 1000, 4096, 9999, 30000]))
    scale = float(10.0 ** rng.integers(-3, 4))
    if kind == 0:      # uniform box
        t = rng.uniform(-1, 1, (3, nt))
    elif kind == 1:    # lattice: ties everywhere
        t = rng.integers(0, 12, (3, nt)).astype(np.float64) / 4
    elif kind == 2:    # thin plane + noise
        t = rng.uniform(-1, 1, (3, nt)); t[2] *= 1e-4
    elif kind == 3:    # clusters + duplicates
        c = rng.uniform(-1, 1, (3, 20)); t = c[:, rng.integers(0, 20, nt)] + rng.normal(0, 0.01, (3, nt)); t[:, ::7] = t[:, :1]
    elif kind == 4:    # far from the origin
        t = rng.uniform(-1, 1, (3, nt)) + 1000.0
    elif kind == 5:    # line
        t = np.outer(rng.normal(size=3), rng.uniform(-1, 1, nt))
    else:              # rings like a LiDAR scan
        a = rng.uniform(0, 2 * np.pi, nt); r = rng.integers(1, 30, nt).astype(np.float64)
        t = np.stack([r * np.cos(a), r * np.sin(a), 0.05 * r])
    t = (t * scale).astype(np.float32)
    ang = rng.normal(0, 0.03, 3); tr = rng.normal(0, 0.05, 3) * scale
    cz, sz_ = np.cos(ang[2]), np.sin(ang[2])
    R = np.array([[cz, -sz_, 0], [sz_, cz, 0], [0, 0, 1]])
    s = (R @ t[:, rng.integers(0, nt, ns)].astype(np.float64) + tr[:, None] + rng.normal(0, 0.002 * scale, (3, ns))).astype(np.float32)
    if rng.integers(0, 4) == 0:
        s[:, :: max(1, ns // 5)] = t[:, : len(s[0, :: max(1, ns // 5)])]      # exact coincidences
    if rng.integers(0, 3) == 0:                                              # partial overlap: sources with no target inside the gate
        m = rng.random(ns) < 0.15
        s[:, m] += (rng.uniform(3.0, 60.0, (1, int(m.sum()))) * scale * rng.choice([-1.0, 1.0], (3, 1))).astype(np.float32)
    return kind, np.ascontiguousarray(t), np.ascontiguousarray(s), scale


bad = 0
t0 = time.time()
for case in range(cases):
    rng = np.random.default_rng(seed0 + case)
    kind, t, s, scale = make(rng)
    res = {}
    for name, tunes in (("exact", {"nn1_variant": 2}), ("default", {}), ("etrack_cold", {"nn1_variant": 4}), ("btrack_cold", {"nn1_variant": 6}), ("htrack_cold", {"nn1_variant": 7}),
                        ("strack", {"nn1_variant": 8}), ("strack_flush1", {"nn1_variant": 8, "nn1_sign_flush": 1, "nn1_supers_per_slice": 2}),
                        ("strack3", {"nn1_variant": 10}), ("strack3_qg4_flush1", {"nn1_variant": 10, "nn1_sign_flush": 1, "nn1_sphere_qg": 4}),
                        ("strack3_qg2_late", {"nn1_variant": 10, "nn1_sphere_flush_end": 128, "nn1_sphere_qg": 2}), ("grid", {"nn_method": 2}),
                        ("grid_stile_tight", {"nn_method": 2, "grid_order": 2, "grid_mode": 3, "grid_tile": 1, "grid_stile_bmax_cm": 100000, "grid_stile_keep": 6, "grid_stile_flush": 1, "grid_stile_split_mm": 1, "grid_stile_cells": 3}),
                        ("grid_tile", {"nn_method": 2, "grid_order": 2, "grid_mode": 3, "grid_tile": 1})):
        ctx = pcr.Context(0)
        ctx.tune("nn_method", 1)
        for k, v in tunes.items():
            ctx.tune(k, v)
        ct, cs = ctx.cloud(t), ctx.cloud(s)
        idx, d2 = ctx.nn1(ct, cs)
        out = [idx.copy(), d2.view(np.uint32).copy()]
        for iters in (1, 2, 3, 6):
            T, st = ctx.icp_point2point(ctx.cloud(s), ct, max_corr=float(scale * scale), max_iter=iters, eps=0.0)
            out.append(T.view(np.uint32).copy()); out.append(np.array([st["iters_run"], st["last_pairs"]]))
        res[name] = out
        ctx.close()
    for name in ("default", "etrack_cold", "btrack_cold", "htrack_cold", "strack", "strack_flush1", "strack3", "strack3_qg4_flush1", "strack3_qg2_late", "grid", "grid_tile", "grid_stile_tight"):
        ok = all(np.array_equal(a, b) for a, b in zip(res["exact"], res[name]))
        if not ok:
            bad += 1
            print(f"MISMATCH case {case} seed {seed0 + case} kind {kind} nt {t.shape[1]} ns {s.shape[1]} scale {scale} vs {name}", flush=True)
    if case % 10 == 9:
        print(f"{case + 1} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print("soak done:", cases, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)

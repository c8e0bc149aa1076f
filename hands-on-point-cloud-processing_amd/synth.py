"""Deterministic synthetic inputs for the k-NN correspondence + ICP hot path.

No dataset ships with the repo and the GPU box has no network, so every test / bench input is
regenerated from a counter-based PRNG (SplitMix64 -> uniform / Box-Muller).  Shapes follow what the
reference feeds its hot path:

* ``kitti_like_scan``   — a 64-beam scan like Homework2/hw2/000000.bin (N x 4 f32 on disk,
  test.hpp:11-33) / Homework4/test/*.bin, ground at z = -1.73 (ground_detection_SVD.py:47).
* ``kitti_like_pair``   — the (source, target) pair ICPpoint2point receives
  (Homework9/hw9/src/registration.cpp:862).
* ``lattice_cloud``     — generateRandomPointCloud (Homework2/hw2/include/test.hpp:132-159):
  ``range * (rand() % 1000) / 1000.0`` with glibc rand(); creates exact ties and duplicates.
"""
from __future__ import annotations

import ctypes
import math

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = np.uint64(0x9E3779B97F4A7C15)

SEED_TARGET = 0x5EED0001
SEED_PAIR = 0x5EED0002


def splitmix64(seed: int, counter: np.ndarray) -> np.ndarray:
    """SplitMix64 output for state = seed + (counter+1)*golden, vectorised (uint64 wrap-around)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + (counter.astype(np.uint64) + np.uint64(1)) * _GOLD
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed: int, counter: np.ndarray) -> np.ndarray:
    """Uniform in (0, 1], 53-bit, f64."""
    return ((splitmix64(seed, counter) >> np.uint64(11)).astype(np.float64) + 1.0) * (1.0 / 9007199254740992.0)


def normal(seed: int, counter: np.ndarray) -> np.ndarray:
    """Standard normal via Box-Muller on two independent streams, f64."""
    u1 = uniform01(seed, counter)
    u2 = uniform01(seed ^ 0xA5A5A5A5A5A5A5A5, counter)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * math.pi * u2)


def _scene_boxes(seed: int, n_boxes: int = 40):
    c = np.arange(n_boxes, dtype=np.uint64)
    cx = (uniform01(seed + 11, c) * 2.0 - 1.0) * 38.0
    cy = (uniform01(seed + 12, c) * 2.0 - 1.0) * 38.0
    # keep the sensor itself outside every box
    near = (np.abs(cx) < 5.0) & (np.abs(cy) < 5.0)
    cx = np.where(near, cx + np.sign(cx + 1e-9) * 8.0, cx)
    sx = 1.0 + 5.0 * uniform01(seed + 13, c)
    sy = 1.0 + 5.0 * uniform01(seed + 14, c)
    sz = 1.0 + 5.0 * uniform01(seed + 15, c)
    lo = np.stack([cx - sx / 2, cy - sy / 2, np.full(n_boxes, -1.73)], axis=1)
    hi = np.stack([cx + sx / 2, cy + sy / 2, -1.73 + sz], axis=1)
    return lo, hi


def kitti_like_scan(n: int, seed: int = SEED_TARGET, chunk: int = 1 << 20) -> np.ndarray:
    """(3, n) f32 SoA scan: 64 beams (+2.0 deg .. -24.8 deg), ceil(n/64) azimuth steps, sensor at the
    origin 1.73 m above the ground plane z = -1.73; every ray hits the nearest of ground / 40 boxes /
    4 walls at +-40 m; range clamped to 80 m; Gaussian range noise sigma = 0.02 m."""
    if n == 0:
        return np.zeros((3, 0), dtype=np.float32)
    beams = 64
    A = (n + beams - 1) // beams
    lo, hi = _scene_boxes(seed)
    out = np.empty((3, n), dtype=np.float32)
    for start in range(0, n, chunk):
        stop = min(n, start + chunk)
        i = np.arange(start, stop, dtype=np.uint64)
        beam = (i // np.uint64(A)).astype(np.float64)
        az_i = (i % np.uint64(A)).astype(np.float64)
        elev = np.deg2rad(2.0 - beam * (26.8 / (beams - 1)))
        az = (az_i + 0.5 * (beam % 2)) * (2.0 * math.pi / A)
        d = np.stack([np.cos(elev) * np.cos(az), np.cos(elev) * np.sin(az), np.sin(elev)], axis=0)  # (3, m)
        t_best = np.full(stop - start, 80.0)
        # ground plane z = -1.73
        with np.errstate(divide="ignore", invalid="ignore"):
            tg = np.where(d[2] < -1e-9, -1.73 / d[2], np.inf)
            t_best = np.minimum(t_best, tg)
            # walls x = +-40, y = +-40
            for ax in (0, 1):
                tw = np.where(np.abs(d[ax]) > 1e-12, 40.0 / np.abs(d[ax]), np.inf)
                t_best = np.minimum(t_best, tw)
            # boxes: slab test from the origin
            inv = 1.0 / d  # (3, m), inf where d == 0 (handled by min/max below)
            for b in range(lo.shape[0]):
                t1 = lo[b][:, None] * inv
                t2 = hi[b][:, None] * inv
                tmin = np.max(np.minimum(t1, t2), axis=0)
                tmax = np.min(np.maximum(t1, t2), axis=0)
                hit = (tmax >= np.maximum(tmin, 0.0)) & (tmin > 0.0)
                t_best = np.where(hit & (tmin < t_best), tmin, t_best)
        r = t_best + 0.02 * normal(seed + 21, i)
        out[:, start:stop] = (d * r).astype(np.float32)
    return out


def gt_pose() -> np.ndarray:
    """Ground-truth 4x4 (f64): yaw 2.0 deg, pitch 0.3 deg, roll -0.2 deg, t = (0.50, -0.20, 0.05)."""
    y, p, r = np.deg2rad(2.0), np.deg2rad(0.3), np.deg2rad(-0.2)
    Rz = np.array([[math.cos(y), -math.sin(y), 0], [math.sin(y), math.cos(y), 0], [0, 0, 1]])
    Ry = np.array([[math.cos(p), 0, math.sin(p)], [0, 1, 0], [-math.sin(p), 0, math.cos(p)]])
    Rx = np.array([[1, 0, 0], [0, math.cos(r), -math.sin(r)], [0, math.sin(r), math.cos(r)]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = [0.50, -0.20, 0.05]
    return T


def kitti_like_pair(n: int, seed_target: int = SEED_TARGET, seed_pair: int = SEED_PAIR,
                    n_src: int | None = None, shard: int = 0):
    """(src, tgt), both (3, n) f32 SoA.  tgt = kitti_like_scan(n); the ICP solution maps src onto tgt:
    src = T_gt^-1 applied to (tgt[perm] + N(0, 0.01^2)), so that ICP(src, tgt) -> T_gt.
    ``n_src``/``shard``: draw a different permutation + noise stream per shard (multi-GPU weak scaling:
    every rank registers its own n_src-point piece of a denser source scan against the same target)."""
    tgt = kitti_like_scan(n, seed_target)
    m = n if n_src is None else n_src
    sp = seed_pair + 7919 * shard
    c = np.arange(n, dtype=np.uint64)
    perm = np.argsort(splitmix64(sp + 1, c), kind="stable")[:m] if n else np.zeros(0, dtype=np.int64)
    cm = np.arange(m, dtype=np.uint64)
    noise = np.stack([normal(sp + 2, cm), normal(sp + 3, cm), normal(sp + 4, cm)], axis=0) * 0.01
    q = tgt[:, perm].astype(np.float64) + noise
    T = gt_pose()
    Rinv = T[:3, :3].T
    src = Rinv @ (q - T[:3, 3:4])
    return src.astype(np.float32), tgt


def glibc_rand_lattice(n: int, dim: int, max_range: float, seed: int = 1) -> np.ndarray:
    """generateRandomPointCloud (test.hpp:142): max_range * (rand() % 1000) / 1000.0 with glibc rand()
    after srand(seed) (hw2 never seeds -> glibc default seed 1).  (n, dim) f64 AoS."""
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(ctypes.c_uint(seed))
    out = np.empty((n, dim), dtype=np.float64)
    for i in range(n):
        for d in range(dim):
            out[i, d] = max_range * (libc.rand() % 1000) / 1000.0
    return out


def lattice_cloud(n: int, dim: int = 3, max_range: float = 10.0, seed: int = 1234,
                  levels: int = 1000) -> np.ndarray:
    """Same lattice as test.hpp:142 (levels = 1000) but from SplitMix64 (fast, any n): (n, dim) f64.
    A small ``levels`` makes exact ties and exact duplicates frequent."""
    c = np.arange(n * dim, dtype=np.uint64)
    lvl = (splitmix64(seed, c) % np.uint64(levels)).astype(np.float64)
    return (max_range * lvl / float(levels)).reshape(n, dim)


def uniform_cloud(n: int, dim: int = 3, max_range: float = 10.0, seed: int = 4321) -> np.ndarray:
    """Continuous-uniform (tie-free) variant, (n, dim) f64."""
    c = np.arange(n * dim, dtype=np.uint64)
    return (max_range * uniform01(seed, c)).reshape(n, dim)

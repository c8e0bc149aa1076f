"""Host-side mirror of the Homework4 RANSAC ground-plane interface, running the inlier counts on the GPU.

Mirrors (same names, argument meaning and return values):
  estimate_plane_params   Homework4/ground_detection_ransac.py:158-169   (host, f64; 3 points)
  my_ransac               Homework4/ground_detection_ransac.py:104-155
  ransac_on_segments      Homework4/ground_detection_ransac.py:54-73
  ransac_on_segments_v2   Homework4/ground_detection_ransac.py:76-101
  extract_initial_seeds   Homework4/ground_detection_SVD.py:46-71        (GPU: radix-select of the lowest z, pcr_ground_seeds_f64)
  ground_detection        Homework4/ground_detection_SVD.py:88-101       (GPU: pcr_ground_detection_f64, PCA refit loop)
  ground_detection_on3segs  Homework4/ground_detection_SVD.py:104-126

The hot loop of my_ransac — `dists = |[X 1] . params|; inliers = sum(dists < thr)` evaluated once per hypothesis
(:138-139) — becomes ONE launch of pcr_plane_count_f64 over all `max_iteration` hypotheses: the points are read
from HBM once instead of `max_iteration` times.  The final inlier mask (:152-153) is pcr_plane_mask_f64.

Call signatures are the reference's, positionally: `my_ransac(data, indices, max_iteration, threshold)`,
`ransac_on_segments(data, segment_x=0, max_iteration=40, threshold=0.15)`, `ground_segmentation(data)`, ... — a caller written
against ground_detection_ransac.py / ground_detection_SVD.py runs unchanged (tests/test_hw4_gpu.py calls them exactly as
ground_detection_ransac.py:42-73 does).  Two keyword-only extras exist on every function that reaches the GPU:
  ctx=  a pcr Context; default: one lazily created module-level context on device $PCR_DEVICE (default 0) — no CPU fallback;
  rng=  the reference draws its 3-point samples from a fresh, unseeded `np.random.default_rng()` per iteration (:132); an
        explicit generator makes runs reproducible (and equal to the reference's for an identical sample sequence).
"""
from __future__ import annotations

import math
import os
import struct

import numpy as np

_default_ctx = None


def default_context():
    """The module-level context the reference-signature calls run on (created on first use; raises without an MI355X)."""
    global _default_ctx
    if _default_ctx is None or not _default_ctx.h:
        from . import Context
        _default_ctx = Context(int(os.environ.get("PCR_DEVICE", "0")))
    return _default_ctx


def set_default_context(ctx):
    """Use `ctx` for every call that does not name one (None: drop it; the next call creates a fresh one)."""
    global _default_ctx
    _default_ctx = ctx


def read_velodyne_bin(path):
    """ground_detection_ransac.py:23-34: KITTI .bin (N x 4 f32) -> N x 3 float32 array (x, y, z)."""
    with open(path, "rb") as f:
        content = f.read()
    n = len(content) // struct.calcsize("ffff")        # iter_unpack raises on a ragged tail; the files are whole rows
    return np.frombuffer(content, dtype=np.float32, count=4 * n).reshape(n, 4)[:, :3].copy()


def estimate_plane_params(selected_points: np.ndarray) -> np.ndarray:
    """Plane (a, b, c, d)/|n| through 3 points — ground_detection_ransac.py:158-169."""
    p = np.asarray(selected_points)
    vector1 = p[1, :] - p[0, :]
    vector2 = p[2, :] - p[0, :]
    a = (vector1[1] * vector2[2]) - (vector1[2] * vector2[1])
    b = (vector1[2] * vector2[0]) - (vector1[0] * vector2[2])
    c = (vector1[0] * vector2[1]) - (vector1[1] * vector2[0])
    d = -(a * p[0, 0] + b * p[0, 1] + c * p[0, 2])
    n = math.sqrt(a ** 2 + b ** 2 + c ** 2)
    return np.array([a / n, b / n, c / n, d / n])


def extract_initial_seeds(pcd_points: np.ndarray, LPR_size: int, threshold_seeds: float, *, ctx=None) -> np.ndarray:
    """ground_detection_SVD.py:46-71: the points below z_high whose z is below LPR.z + threshold_seeds (input order kept).
    The selection of the LPR_size lowest z runs on the GPU (pcr_ground_seeds_f64)."""
    if pcd_points.shape[0] == 0:
        return pcd_points[:0]
    ctx = ctx or default_context()
    cloud = ctx.cloud(np.ascontiguousarray(pcd_points[:, :3], np.float32), 1)
    try:
        mask, _ = ctx.ground_seeds(cloud, LPR_size, threshold_seeds)
    finally:
        cloud.free()
    return pcd_points[mask, :]


def ground_detection(pcd_points: np.ndarray, pcd_indices: np.ndarray, max_iter: int, LPR_size: int, threshold_dist: float, *, ctx=None,
                     return_params: bool = False):
    """ground_detection_SVD.py:88-101 -> (seeds, ground indices, foreground indices), the reference's 3-tuple; the plane the
    reference prints (:100) is appended as a 4th value with return_params=True."""
    ctx = ctx or default_context()
    cloud = ctx.cloud(np.ascontiguousarray(pcd_points[:, :3], np.float32), 1)
    try:
        params, inliers_filter = ctx.ground_detection(cloud, max_iter, LPR_size, threshold_dist)
    finally:
        cloud.free()
    out = (pcd_points[inliers_filter], pcd_indices[inliers_filter], pcd_indices[np.logical_not(inliers_filter)])
    return out + (params,) if return_params else out


def ground_detection_on3segs(pcd_points: np.ndarray, main_dist=20, max_iter=6, threshold_dist=0.18, *, ctx=None):
    """ground_detection_SVD.py:104-126: three x-segments [x_min, -main_dist, main_dist, x_max], open intervals as written."""
    ctx = ctx or default_context()
    x_min, x_max = np.min(pcd_points[:, 0]), np.max(pcd_points[:, 0])
    segments_x = [x_min, -main_dist, main_dist, x_max]
    total_indices = np.array(range(pcd_points.shape[0]))
    stacked_ground_idx = np.empty(0, dtype=int)
    stacked_foregr_idx = np.empty(0, dtype=int)
    for i in range(len(segments_x) - 1):
        range_filter = np.logical_and(pcd_points[:, 0] < segments_x[i + 1], pcd_points[:, 0] > segments_x[i])
        if not range_filter.any():
            continue                                                    # the reference would fail on an empty segment
        _, ground, foreground = ground_detection(pcd_points[range_filter], total_indices[range_filter], max_iter,
                                                 LPR_size=10000, threshold_dist=threshold_dist, ctx=ctx)
        stacked_ground_idx = np.r_[stacked_ground_idx, ground]
        stacked_foregr_idx = np.r_[stacked_foregr_idx, foreground]
    return stacked_ground_idx, stacked_foregr_idx


def my_ransac(data: np.ndarray, indices: np.ndarray, max_iteration: int, threshold: float, *, ctx=None, rng=None):
    """ground_detection_ransac.py:104-155 -> (inliers_idx, best_model_params)."""
    assert data.shape[0] == indices.shape[0]
    ctx = ctx or default_context()
    rng = np.random.default_rng() if rng is None else rng
    filtered_data = extract_initial_seeds(data, 40000, 1, ctx=ctx)              # :125
    if filtered_data.shape[0] < 3:
        return indices[:0], []
    hyps = np.zeros((max_iteration, 4), np.float64)
    for it in range(max_iteration):                                             # :131-135
        sel = rng.choice(filtered_data.shape[0], 3, replace=False)          # same draw as choice(range(n), ...) without building the list
        hyps[it] = estimate_plane_params(filtered_data[sel, :].astype(np.float64))
    valid = np.isfinite(hyps).all(axis=1)       # collinear samples give n = 0 -> NaN params; NaN < thr is False (:139)
    seeds = ctx.cloud(np.ascontiguousarray(filtered_data[:, :3], np.float32), 1)
    try:
        counts = np.zeros(max_iteration, np.int64)
        if valid.any():
            counts[valid] = ctx.plane_count(seeds, hyps[valid], float(threshold))   # :138-139, all hypotheses at once
    finally:
        seeds.free()
    best_set_size, best_model_params = 0, []
    for it in range(max_iteration):                                             # :140-142 (strict >: first maximum wins)
        if counts[it] > best_set_size:
            best_set_size = counts[it]
            best_model_params = hyps[it]
    if len(best_model_params) == 0:
        return indices[:0], []
    allpts = ctx.cloud(np.ascontiguousarray(data[:, :3], np.float32), 1)
    try:
        mask, _ = ctx.plane_mask(allpts, best_model_params, float(threshold))   # :152-153
    finally:
        allpts.free()
    return indices[mask.astype(bool)], best_model_params


def ground_segmentation(data, *, ctx=None, rng=None):
    """ground_detection_ransac.py:42-51: indices of the ground points of one full scan (prints what the reference prints)."""
    ground_indices = ransac_on_segments(data, ctx=ctx, rng=rng)
    print('origin data points num:', data.shape[0])
    print('segmented data points num:', ground_indices.shape[0])
    return ground_indices


def ransac_on_segments(data: np.ndarray, segment_x=0, max_iteration=40, threshold=0.15, *, ctx=None, rng=None):
    """ground_detection_ransac.py:54-73: RANSAC on the two x-segments, stacked inlier indices."""
    ctx = ctx or default_context()
    total = np.array(range(data.shape[0]))
    fwd = data[:, 0] >= segment_x
    idx1, _ = my_ransac(data[fwd], total[fwd], max_iteration, threshold, ctx=ctx, rng=rng)
    idx2, _ = my_ransac(data[np.logical_not(fwd)], total[np.logical_not(fwd)], max_iteration, threshold, ctx=ctx, rng=rng)
    return np.r_[idx1, idx2]


def ransac_on_segments_v2(data: np.ndarray, segments_num=5, max_iteration=40, threshold=0.15, *, ctx=None, rng=None):
    """ground_detection_ransac.py:76-101: RANSAC on `segments_num` uniform x-segments (open intervals, as written)."""
    ctx = ctx or default_context()
    total = np.array(range(data.shape[0]))
    x_min, x_max = np.min(data[:, 0]), np.max(data[:, 0])
    seg_bound = x_min + np.array(range(segments_num + 1)) * ((x_max - x_min) / segments_num)
    stacked = np.empty(0, dtype=int)
    for i in range(segments_num):
        flt = np.logical_and(data[:, 0] < seg_bound[i + 1], data[:, 0] > seg_bound[i])
        if not flt.any():
            continue
        idx, _ = my_ransac(data[flt], total[flt], max_iteration, threshold, ctx=ctx, rng=rng)
        if idx is None:
            continue
        stacked = np.r_[stacked, idx]
    return stacked

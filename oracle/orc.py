"""ctypes binding of the CPU ORACLE (oracle/liborc.so) and of the compiled reference harness
(oracle/_ref/libpcr_ref.so).  TEST INFRASTRUCTURE ONLY: importable from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg — never from the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORC = None
_REF = None

f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


class IcpParams(C.Structure):
    _fields_ = [("max_corr", C.c_float), ("max_iter", C.c_uint64), ("eps", C.c_float)]


class IcpStats(C.Structure):
    _fields_ = [("iters_run", C.c_uint64), ("converged", C.c_int), ("empty_pairs", C.c_int),
                ("last_pairs", C.c_uint64), ("last_loss", C.c_float)]


def build(ref: bool = True) -> None:
    """Compile liborc.so and, when /root/reference is present, _ref/libpcr_ref.so."""
    subprocess.run(["make", "-C", _HERE, "all"], check=True, capture_output=True)
    if ref and os.path.isdir("/root/reference"):
        subprocess.run(["make", "-C", _HERE, "ref"], check=True, capture_output=True)


def lib():
    global _ORC
    if _ORC is None:
        path = os.environ.get("ORC_LIB_PATH") or os.path.join(_HERE, "liborc.so")   # override: the sanitizer build (make sanitize)
        if not os.path.exists(path):
            build(ref=False)
        L = C.CDLL(path)
        sz = C.c_size_t
        L.orc_d2_f32.restype = C.c_float
        L.orc_d2_f32.argtypes = [C.c_float] * 6
        L.orc_nn1_f32.argtypes = [f32p, f32p, f32p, sz, f32p, f32p, f32p, sz, u32p, f32p]
        L.orc_nn1_f32_mt.argtypes = [f32p, f32p, f32p, sz, f32p, f32p, f32p, sz, u32p, f32p, C.c_int]
        L.orc_nn1_tiecount_f32.argtypes = [f32p, f32p, f32p, sz, f32p, f32p, f32p, sz, u32p]
        L.orc_knn_f64.argtypes = [f64p, sz, C.c_int, f64p, sz, C.c_int, i32p, f64p]
        L.orc_radius_f64.argtypes = [f64p, sz, C.c_int, f64p, sz, C.c_double, i64p, C.c_void_p, C.c_void_p]
        L.orc_radius_count_f64_mt.argtypes = [f64p, sz, C.c_int, f64p, sz, C.c_double, i64p, C.c_int]
        L.orc_radius_f32.argtypes = [f32p, sz, C.c_int, f32p, sz, C.c_float, i64p, C.c_void_p, C.c_void_p]
        L.orc_transform_f32.argtypes = [f32p, f32p, f32p, sz, f32p, f32p]
        L.orc_kabsch_accumulate.restype = C.c_int64
        L.orc_kabsch_accumulate.argtypes = [f32p, f32p, f32p, sz, f32p, f32p, f32p, u32p, f32p, C.c_float, f64p]
        L.orc_svd3.argtypes = [f64p, f64p, f64p, f64p]
        L.orc_kabsch_solve.restype = C.c_int
        L.orc_kabsch_solve.argtypes = [f64p, f32p, f32p]
        L.orc_kabsch_solve_ransac.restype = C.c_int
        L.orc_kabsch_solve_ransac.argtypes = [f64p, f32p, f32p]
        L.orc_mat4_mul_f32.argtypes = [f32p, f32p, f32p]
        L.orc_icp_p2p_f32.argtypes = [f32p, f32p, f32p, sz, f32p, f32p, f32p, sz, f32p,
                                      C.POINTER(IcpParams), f32p, C.POINTER(IcpStats), C.c_void_p, C.c_void_p]
        L.orc_plane_count_f32pts.argtypes = [f32p, f32p, f32p, sz, f64p, sz, C.c_double, i64p]
        L.orc_plane_count_f64pts.argtypes = [f64p, f64p, f64p, sz, f64p, sz, C.c_double, i64p]
        L.orc_plane_mask_f32pts.argtypes = [f32p, f32p, f32p, sz, f64p, C.c_double, u8p]
        L.orc_plane_from_3pts.argtypes = [f64p, f64p]
        L.orc_voxel_filter_f32.restype = sz
        L.orc_voxel_filter_f32.argtypes = [f32p, f32p, f32p, sz, C.c_double, f32p, f32p, f32p]
        L.orc_d2_dim_f32.argtypes = [f32p, f32p, C.c_int]
        L.orc_d2_dim_f32.restype = C.c_float
        L.orc_nn1_dim_f32.argtypes = [f32p, sz, f32p, sz, C.c_int, u32p, f32p]
        L.orc_match_union_f32.argtypes = [f32p, sz, f32p, sz, C.c_int, C.c_float, u32p, f32p]
        L.orc_match_union_f32.restype = sz
        L.orc_match_inter_f32.argtypes = [f32p, sz, f32p, sz, C.c_int, C.c_float, u32p, f32p]
        L.orc_match_inter_f32.restype = sz
        L.orc_ransac_hypothesis.argtypes = [f32p, f32p, u32p, u32p, f32p, f32p]
        L.orc_consensus_count_f32.argtypes = [f32p, f32p, u32p, sz, f32p, f32p, C.c_float]
        L.orc_consensus_count_f32.restype = C.c_uint32
        L.orc_ransac_global_f32.argtypes = [f32p, f32p, u32p, sz, u32p, sz, C.c_float, f32p, f32p, u32p, C.c_void_p]
        L.orc_ransac_global_f32.restype = C.c_int64
        L.orc_fast_eigen3x3.argtypes = [f64p, f64p]
        L.orc_ground_seeds_f64.argtypes = [f32p, f32p, f32p, sz, sz, C.c_double, u8p, f64p]
        L.orc_ground_seeds_f64.restype = sz
        L.orc_estimate_plane_f64.argtypes = [f32p, f32p, f32p, sz, u8p, f64p]
        L.orc_estimate_plane_f64.restype = sz
        L.orc_ground_detection_f64.argtypes = [f32p, f32p, f32p, sz, C.c_int, sz, C.c_double, f64p, u8p]
        L.orc_ground_detection_f64.restype = sz
        L.orc_knn_sq_f32pts.argtypes = [f32p, f32p, f32p, sz, f32p, f32p, f32p, sz, C.c_int, C.c_double, i32p, f64p, u32p]
        L.orc_normals_knn_f64.argtypes = [f32p, f32p, f32p, sz, C.c_int, C.c_double, f64p]
        L.orc_icp_p2plane_f32.argtypes = [f32p, f32p, f32p, sz, f32p, f32p, f32p, sz, f32p, f32p, f32p, f32p, C.POINTER(IcpParams), f32p, C.POINTER(IcpStats)]
        L.orc_solve6.argtypes = [f64p, f64p, f64p]
        L.orc_iss_f32.argtypes = [f32p, f32p, f32p, sz, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, u8p, f32p]
        L.orc_hw2_knn_add.argtypes = [f64p, i32p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double),
                                      C.c_double, C.c_int]
        L.orc_nano_knn_add.argtypes = [f32p, u64p, sz, C.POINTER(sz), C.c_float, sz]
        _ORC = L
    return _ORC


def ref_path() -> str:
    return os.path.join(_HERE, "_ref", "libpcr_ref.so")


def have_ref() -> bool:
    return os.path.exists(ref_path())


def ref():
    """The REFERENCE's own code (hw2 kd-tree/octree, vendored nanoflann) behind a C harness."""
    global _REF
    if _REF is None:
        L = C.CDLL(ref_path())
        sz = C.c_size_t
        dp = C.POINTER(C.c_double)
        L.ref_nano_nn1_f32.argtypes = [f32p, f32p, f32p, sz, f32p, f32p, f32p, sz, C.c_int, C.c_int,
                                       u32p, f32p, dp, dp]
        L.ref_nano_nn1_dim_f32.argtypes = [f32p, sz, f32p, sz, C.c_int, C.c_int, u32p, f32p]
        L.ref_nano_knn_f64.argtypes = [f64p, sz, C.c_int, f64p, sz, C.c_int, C.c_int, u64p, f64p]
        L.ref_hw2_kd_knn.argtypes = [f64p, sz, C.c_int, f64p, sz, C.c_int, C.c_int, i32p, f64p,
                                     C.c_void_p, dp, dp]
        L.ref_hw2_kd_radius.argtypes = [f64p, sz, C.c_int, f64p, sz, C.c_double, C.c_int, i64p,
                                        C.c_void_p, C.c_void_p]
        L.ref_hw2_oct_knn.argtypes = [f64p, sz, C.c_int, f64p, sz, C.c_int, C.c_int, C.c_double, i32p, f64p]
        L.ref_hw2_read_binary.restype = C.c_int64
        L.ref_hw2_read_binary.argtypes = [C.c_char_p, f64p, sz]
        _REF = L
    return _REF


# ----------------------------------------------------------------------------- numpy-level helpers
def _soa(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[0] == 3
    return a[0].copy(), a[1].copy(), a[2].copy()


def nn1_f32(tgt_soa, src_soa):
    """Brute-force canonical 1-NN. tgt/src: (3, n) f32. Returns (idx u32, d2 f32)."""
    tx, ty, tz = _soa(tgt_soa)
    sx, sy, sz = _soa(src_soa)
    idx = np.empty(sx.size, np.uint32)
    d2 = np.empty(sx.size, np.float32)
    lib().orc_nn1_f32(tx, ty, tz, tx.size, sx, sy, sz, sx.size, idx, d2)
    return idx, d2


def nn1_f32_mt(tgt_soa, src_soa, threads=8):
    """orc_nn1_f32 with the queries split over host threads (identical results)."""
    tx, ty, tz = _soa(tgt_soa)
    sx, sy, sz = _soa(src_soa)
    idx = np.empty(sx.size, np.uint32)
    d2 = np.empty(sx.size, np.float32)
    lib().orc_nn1_f32_mt(tx, ty, tz, tx.size, sx, sy, sz, sx.size, idx, d2, int(threads))
    return idx, d2


def nn1_tiecount_f32(tgt_soa, src_soa):
    tx, ty, tz = _soa(tgt_soa)
    sx, sy, sz = _soa(src_soa)
    cnt = np.empty(sx.size, np.uint32)
    lib().orc_nn1_tiecount_f32(tx, ty, tz, tx.size, sx, sy, sz, sx.size, cnt)
    return cnt


def knn_f64(db, q, k):
    db = np.ascontiguousarray(db, np.float64)
    q = np.ascontiguousarray(q, np.float64)
    m, dim = q.shape
    idx = np.empty((m, k), np.int32)
    dist = np.empty((m, k), np.float64)
    lib().orc_knn_f64(db, db.shape[0], dim, q, m, k, idx, dist)
    return idx, dist


def radius_f64(db, q, r):
    db = np.ascontiguousarray(db, np.float64)
    q = np.ascontiguousarray(q, np.float64)
    m, dim = q.shape
    row = np.empty(m + 1, np.int64)
    lib().orc_radius_f64(db, db.shape[0], dim, q, m, r, row, None, None)
    idx = np.empty(max(1, int(row[-1])), np.int32)
    dist = np.empty(max(1, int(row[-1])), np.float64)
    lib().orc_radius_f64(db, db.shape[0], dim, q, m, r, row, idx.ctypes.data, dist.ctypes.data)
    return row, idx[: row[-1]], dist[: row[-1]]


def radius_count_f64_mt(db, q, r, threads=8):
    """row counts only, on host threads (the same comparison pair by pair as radius_f64)"""
    db = np.ascontiguousarray(db, np.float64)
    q = np.ascontiguousarray(q, np.float64)
    m, dim = q.shape
    cnt = np.empty(m, np.int64)
    lib().orc_radius_count_f64_mt(db, db.shape[0], dim, q, m, r, cnt, threads)
    return cnt


def radius_f32(db, q, r):
    db = np.ascontiguousarray(db, np.float32)
    q = np.ascontiguousarray(q, np.float32)
    m, dim = q.shape
    row = np.empty(m + 1, np.int64)
    lib().orc_radius_f32(db, db.shape[0], dim, q, m, r, row, None, None)
    idx = np.empty(max(1, int(row[-1])), np.int32)
    dist = np.empty(max(1, int(row[-1])), np.float32)
    lib().orc_radius_f32(db, db.shape[0], dim, q, m, r, row, idx.ctypes.data, dist.ctypes.data)
    return row, idx[: row[-1]], dist[: row[-1]]


def transform_f32(soa, R, t):
    x, y, z = _soa(soa)
    lib().orc_transform_f32(x, y, z, x.size, np.ascontiguousarray(R, np.float32).reshape(9),
                            np.ascontiguousarray(t, np.float32).reshape(3))
    return np.stack([x, y, z])


def kabsch_accumulate(src_soa, tgt_soa, idx, d2, max_corr):
    sx, sy, sz = _soa(src_soa)
    tx, ty, tz = _soa(tgt_soa)
    sums = np.zeros(16, np.float64)
    last = lib().orc_kabsch_accumulate(sx, sy, sz, sx.size, tx, ty, tz,
                                       np.ascontiguousarray(idx, np.uint32),
                                       np.ascontiguousarray(d2, np.float32), max_corr, sums)
    return sums, int(last)


def kabsch_solve(sums):
    R = np.zeros(9, np.float32)
    t = np.zeros(3, np.float32)
    rc = lib().orc_kabsch_solve(np.ascontiguousarray(sums, np.float64), R, t)
    return rc, R.reshape(3, 3), t


def kabsch_solve_ransac(sums):
    """registration.cpp:372-392: t from U V^T, the det < 0 repair replaces R only."""
    R = np.zeros(9, np.float32)
    t = np.zeros(3, np.float32)
    rc = lib().orc_kabsch_solve_ransac(np.ascontiguousarray(sums, np.float64), R, t)
    return rc, R.reshape(3, 3), t


def svd3(A):
    U = np.zeros(9); S = np.zeros(3); V = np.zeros(9)
    lib().orc_svd3(np.ascontiguousarray(A, np.float64).reshape(9), U, S, V)
    return U.reshape(3, 3), S, V.reshape(3, 3)


def icp_p2p_f32(src_soa, tgt_soa, init_T=None, max_corr=1.0, max_iter=20, eps=1e-8, trace=False):
    sx, sy, sz = _soa(src_soa)
    tx, ty, tz = _soa(tgt_soa)
    T0 = np.eye(4, dtype=np.float32) if init_T is None else np.ascontiguousarray(init_T, np.float32)
    prm = IcpParams(max_corr, max_iter, eps)
    st = IcpStats()
    out = np.zeros(16, np.float32)
    per_T = np.zeros((max_iter, 16), np.float32) if trace else None
    per_n = np.zeros(max_iter, np.uint64) if trace else None
    lib().orc_icp_p2p_f32(sx, sy, sz, sx.size, tx, ty, tz, tx.size, T0.reshape(16), C.byref(prm), out,
                          C.byref(st), per_T.ctypes.data if trace else None,
                          per_n.ctypes.data if trace else None)
    stats = dict(iters_run=int(st.iters_run), converged=int(st.converged), empty_pairs=int(st.empty_pairs),
                 last_pairs=int(st.last_pairs), last_loss=float(st.last_loss))
    if trace:
        return out.reshape(4, 4), stats, per_T.reshape(max_iter, 4, 4), per_n
    return out.reshape(4, 4), stats


def icp_p2plane_f32(src_soa, tgt_soa, tgt_normals_soa, init_T=None, max_corr=1.0, max_iter=20, eps=1e-8):
    sx, sy, sz = _soa(src_soa)
    tx, ty, tz = _soa(tgt_soa)
    nx, ny, nz = _soa(tgt_normals_soa)
    T0 = np.eye(4, dtype=np.float32) if init_T is None else np.ascontiguousarray(init_T, np.float32)
    prm = IcpParams(max_corr, max_iter, eps)
    st = IcpStats()
    out = np.zeros(16, np.float32)
    lib().orc_icp_p2plane_f32(sx, sy, sz, sx.size, tx, ty, tz, tx.size, nx, ny, nz, T0.reshape(16), C.byref(prm), out, C.byref(st))
    return out.reshape(4, 4), dict(iters_run=int(st.iters_run), converged=int(st.converged), empty_pairs=int(st.empty_pairs),
                                   last_pairs=int(st.last_pairs), last_loss=float(st.last_loss))


def plane_count(soa, planes4, thr):
    planes4 = np.ascontiguousarray(planes4, np.float64).reshape(-1, 4)
    counts = np.zeros(planes4.shape[0], np.int64)
    a = np.asarray(soa)
    if a.dtype == np.float64:
        x, y, z = (np.ascontiguousarray(a[i]) for i in range(3))
        lib().orc_plane_count_f64pts(x, y, z, x.size, planes4, planes4.shape[0], thr, counts)
    else:
        x, y, z = _soa(a)
        lib().orc_plane_count_f32pts(x, y, z, x.size, planes4, planes4.shape[0], thr, counts)
    return counts


def plane_mask(soa, plane4, thr):
    x, y, z = _soa(soa)
    mask = np.zeros(x.size, np.uint8)
    lib().orc_plane_mask_f32pts(x, y, z, x.size, np.ascontiguousarray(plane4, np.float64).reshape(4), thr, mask)
    return mask


def plane_from_3pts(p):
    out = np.zeros(4)
    lib().orc_plane_from_3pts(np.ascontiguousarray(p, np.float64).reshape(9), out)
    return out


def voxel_filter_f32(soa, leaf_size):
    """Homework1 voxel_filter (centroid mode) -> (3, m) f32."""
    x, y, z = _soa(soa)
    ox, oy, oz = np.empty_like(x), np.empty_like(y), np.empty_like(z)
    m = lib().orc_voxel_filter_f32(x, y, z, x.size, float(leaf_size), ox, oy, oz)
    return np.stack([ox[:m], oy[:m], oz[:m]])


def nn1_dim_f32(db, q):
    db = np.ascontiguousarray(db, np.float32)
    q = np.ascontiguousarray(q, np.float32)
    idx = np.empty(q.shape[0], np.uint32)
    d2 = np.empty(q.shape[0], np.float32)
    lib().orc_nn1_dim_f32(db, db.shape[0], q, q.shape[0], db.shape[1], idx, d2)
    return idx, d2


def match_union_f32(desc_src, desc_tgt, rejection_rate):
    """findRANSACCorrespondencesUnion -> (pairs [K, 2] u32 (src, tgt), dist f32[K])."""
    a = np.ascontiguousarray(desc_src, np.float32)
    b = np.ascontiguousarray(desc_tgt, np.float32)
    total = a.shape[0] + b.shape[0]
    pairs = np.zeros((max(total, 1), 2), np.uint32)
    dist = np.zeros(max(total, 1), np.float32)
    k = lib().orc_match_union_f32(a, a.shape[0], b, b.shape[0], a.shape[1], rejection_rate, pairs.reshape(-1), dist)
    return pairs[:k], dist[:k]


def match_inter_f32(desc_src, desc_tgt, rejection_rate):
    """findRANSACCorrespondencesInter -> (pairs [K, 2] u32 (src, tgt), dist f32[K])."""
    a = np.ascontiguousarray(desc_src, np.float32)
    b = np.ascontiguousarray(desc_tgt, np.float32)
    pairs = np.zeros((max(a.shape[0], 1), 2), np.uint32)
    dist = np.zeros(max(a.shape[0], 1), np.float32)
    k = lib().orc_match_inter_f32(a, a.shape[0], b, b.shape[0], a.shape[1], rejection_rate, pairs.reshape(-1), dist)
    return pairs[:k], dist[:k]


def ransac_hypothesis(src_xyz, tgt_xyz, pairs, quad):
    R = np.zeros(9, np.float32)
    t = np.zeros(3, np.float32)
    rc = lib().orc_ransac_hypothesis(np.ascontiguousarray(src_xyz, np.float32).reshape(-1), np.ascontiguousarray(tgt_xyz, np.float32).reshape(-1),
                                     np.ascontiguousarray(pairs, np.uint32).reshape(-1), np.ascontiguousarray(quad, np.uint32), R, t)
    return rc, R.reshape(3, 3), t


def consensus_count_f32(src_xyz, tgt_xyz, pairs, R, t, thr):
    p = np.ascontiguousarray(pairs, np.uint32)
    return int(lib().orc_consensus_count_f32(np.ascontiguousarray(src_xyz, np.float32).reshape(-1), np.ascontiguousarray(tgt_xyz, np.float32).reshape(-1),
                                             p.reshape(-1), p.shape[0], np.ascontiguousarray(R, np.float32).reshape(9),
                                             np.ascontiguousarray(t, np.float32).reshape(3), thr))


def ransac_global_f32(src_xyz, tgt_xyz, pairs, quads, thr):
    """Registration::RANSAC over given quads -> (winner index, R, t, best count, counts[n_hyp])."""
    p = np.ascontiguousarray(pairs, np.uint32)
    qd = np.ascontiguousarray(quads, np.uint32)
    R = np.zeros(9, np.float32)
    t = np.zeros(3, np.float32)
    best = np.zeros(1, np.uint32)
    counts = np.zeros(max(qd.shape[0], 1), np.uint32)
    w = lib().orc_ransac_global_f32(np.ascontiguousarray(src_xyz, np.float32).reshape(-1), np.ascontiguousarray(tgt_xyz, np.float32).reshape(-1),
                                    p.reshape(-1), p.shape[0], qd.reshape(-1), qd.shape[0], thr, R, t, best, counts.ctypes.data)
    return int(w), R.reshape(3, 3), t, int(best[0]), counts[: qd.shape[0]]


def fast_eigen3x3(A):
    """mylib.FastEigen3x3: unit eigenvector of the smallest eigenvalue of a symmetric 3x3."""
    out = np.zeros(3, np.float64)
    lib().orc_fast_eigen3x3(np.ascontiguousarray(A, np.float64).reshape(9), out)
    return out


def ground_seeds_f64(soa, lpr_size, threshold_seeds):
    """extract_initial_seeds -> (seed mask u8[n], upper bound)."""
    x, y, z = _soa(soa)
    mask = np.zeros(max(x.size, 1), np.uint8)
    ub = np.zeros(1, np.float64)
    lib().orc_ground_seeds_f64(x, y, z, x.size, lpr_size, threshold_seeds, mask, ub)
    return mask[: x.size], float(ub[0])


def estimate_plane_f64(soa, mask):
    x, y, z = _soa(soa)
    params = np.zeros(4, np.float64)
    m = lib().orc_estimate_plane_f64(x, y, z, x.size, np.ascontiguousarray(mask, np.uint8), params)
    return params, int(m)


def ground_detection_f64(soa, max_iter, lpr_size, threshold_dist):
    """ground_detection -> (params f64[4], ground mask u8[n], count or -1)."""
    x, y, z = _soa(soa)
    params = np.zeros(4, np.float64)
    mask = np.zeros(max(x.size, 1), np.uint8)
    c = lib().orc_ground_detection_f64(x, y, z, x.size, max_iter, lpr_size, threshold_dist, params, mask)
    return params, mask[: x.size], (-1 if c == C.c_size_t(-1).value else int(c))


def knn_sq_f32pts(db_soa, q_soa, k, radius=-1.0):
    """squared-distance k-NN (FLANN / open3d contract), optional hybrid cap s < radius^2 -> (idx [m,k], s [m,k], found [m])."""
    x, y, z = _soa(db_soa)
    qx, qy, qz = _soa(q_soa)
    m = qx.size
    idx = np.zeros((max(m, 1), k), np.int32)
    s = np.zeros((max(m, 1), k), np.float64)
    found = np.zeros(max(m, 1), np.uint32)
    cap = float("inf") if radius < 0 else float(radius) * float(radius)
    lib().orc_knn_sq_f32pts(x, y, z, x.size, qx, qy, qz, m, k, cap, idx.reshape(-1), s.reshape(-1), found)
    return idx[:m], s[:m], found[:m]


def normals_knn_f64(soa, k, radius):
    x, y, z = _soa(soa)
    out = np.zeros((max(x.size, 1), 3), np.float64)
    lib().orc_normals_knn_f64(x, y, z, x.size, k, radius, out.reshape(-1))
    return out[: x.size]


def iss_f32(soa, local_radius, non_max_radius, gamma21=0.9, gamma32=0.9, min_neighbors=5, weighted=True):
    """ISSKeypoint::compute (hw7) -> (is_key uint8[n], lambda3 f32[n])."""
    x, y, z = _soa(soa)
    key = np.zeros(x.size, np.uint8)
    l3 = np.zeros(x.size, np.float32)
    lib().orc_iss_f32(x, y, z, x.size, local_radius, non_max_radius, gamma21, gamma32, min_neighbors, int(weighted), key, l3)
    return key, l3


# ----------------------------------------------------------------------------- reference harness
def ref_nano_nn1_f32(tgt_soa, src_soa, leaf=2, threads=1):
    """vendored nanoflann, f32, as ICPpoint2point instantiates it. Returns (idx, d2, build_ms, query_ms)."""
    tx, ty, tz = _soa(tgt_soa)
    sx, sy, sz = _soa(src_soa)
    idx = np.empty(sx.size, np.uint32)
    d2 = np.empty(sx.size, np.float32)
    b = C.c_double(); q = C.c_double()
    ref().ref_nano_nn1_f32(tx, ty, tz, tx.size, sx, sy, sz, sx.size, leaf, threads, idx, d2,
                           C.byref(b), C.byref(q))
    return idx, d2, b.value, q.value


def ref_nano_nn1_dim_f32(db, q, leaf=2):
    """vendored nanoflann at any dim (row-major f32), as findRANSACCorrespondencesUnion uses it -> (idx, d2)."""
    db = np.ascontiguousarray(db, np.float32)
    q = np.ascontiguousarray(q, np.float32)
    idx = np.empty(q.shape[0], np.uint32)
    d2 = np.empty(q.shape[0], np.float32)
    ref().ref_nano_nn1_dim_f32(db, db.shape[0], q, q.shape[0], db.shape[1], leaf, idx, d2)
    return idx, d2


def ref_nano_knn_f64(db, q, k, leaf=10):
    db = np.ascontiguousarray(db, np.float64)
    q = np.ascontiguousarray(q, np.float64)
    m, dim = q.shape
    idx = np.empty((m, k), np.uint64)
    d2 = np.empty((m, k), np.float64)
    ref().ref_nano_knn_f64(db, db.shape[0], dim, q, m, k, leaf, idx, d2)
    return idx, d2


def ref_hw2_kd_knn(db, q, k, leaf=1, want_cmp=False):
    db = np.ascontiguousarray(db, np.float64)
    q = np.ascontiguousarray(q, np.float64)
    m, dim = q.shape
    idx = np.empty((m, k), np.int32)
    dist = np.empty((m, k), np.float64)
    cmp = np.zeros(m, np.int32)
    b = C.c_double(); t = C.c_double()
    ref().ref_hw2_kd_knn(db, db.shape[0], dim, q, m, k, leaf, idx, dist, cmp.ctypes.data,
                         C.byref(b), C.byref(t))
    if want_cmp:
        return idx, dist, cmp, b.value, t.value
    return idx, dist


def ref_hw2_kd_radius(db, q, r, leaf=1):
    db = np.ascontiguousarray(db, np.float64)
    q = np.ascontiguousarray(q, np.float64)
    m, dim = q.shape
    row = np.empty(m + 1, np.int64)
    ref().ref_hw2_kd_radius(db, db.shape[0], dim, q, m, r, leaf, row, None, None)
    idx = np.empty(max(1, int(row[-1])), np.int32)
    dist = np.empty(max(1, int(row[-1])), np.float64)
    ref().ref_hw2_kd_radius(db, db.shape[0], dim, q, m, r, leaf, row, idx.ctypes.data, dist.ctypes.data)
    return row, idx[: row[-1]], dist[: row[-1]]


def ref_hw2_oct_knn(db, q, k, leaf=32, min_extent=0.0001):
    db = np.ascontiguousarray(db, np.float64)
    q = np.ascontiguousarray(q, np.float64)
    m, dim = q.shape
    idx = np.empty((m, k), np.int32)
    dist = np.empty((m, k), np.float64)
    ref().ref_hw2_oct_knn(db, db.shape[0], dim, q, m, k, leaf, min_extent, idx, dist)
    return idx, dist


def hw7_path() -> str:
    return os.path.join(_HERE, "_ref", "libhw7_ref.so")


def have_hw7() -> bool:
    return os.path.exists(hw7_path())


_HW7 = None


def ref_hw7_radius(db, q, r, leaf=12):
    """The reference's hw7 float kd-tree (KDTreeRadiusNNSearch) -> CSR (row_ptr, idx, dist), tree-visit order."""
    global _HW7
    if _HW7 is None:
        _HW7 = C.CDLL(hw7_path())
        _HW7.ref_hw7_radius.argtypes = [f32p, C.c_size_t, f32p, C.c_size_t, C.c_float, C.c_int, i64p, C.c_void_p, C.c_void_p]
    db = np.ascontiguousarray(db, np.float32)
    q = np.ascontiguousarray(q, np.float32)
    m = q.shape[0]
    row = np.empty(m + 1, np.int64)
    _HW7.ref_hw7_radius(db, db.shape[0], q, m, r, leaf, row, None, None)
    idx = np.empty(max(1, int(row[-1])), np.int32)
    dist = np.empty(max(1, int(row[-1])), np.float32)
    _HW7.ref_hw7_radius(db, db.shape[0], q, m, r, leaf, row, idx.ctypes.data, dist.ctypes.data)
    return row, idx[: row[-1]], dist[: row[-1]]


def ref_hw2_read_binary(path, cap=200000):
    out = np.zeros((cap, 3), np.float64)
    n = ref().ref_hw2_read_binary(path.encode(), out, cap)
    return out[: min(n, cap)], int(n)

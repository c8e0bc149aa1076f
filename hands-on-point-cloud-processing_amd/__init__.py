"""hands-on-point-cloud-processing_amd — MI355X-native (gfx950) k-NN correspondence + ICP + plane-inlier hot
path of yf26/Hands-On-Point-Cloud-Processing.

The product is ``libpcr_hip.so`` (hand-written HIP kernels behind the C ABI of ``include/pcr.h``); this
package is the thin Python mirror over that ABI used by the tests, the bench and Python callers
(Homework4 is Python in the reference).  There is NO CPU fallback: every compute entry point raises if the
HIP library is missing or no GPU is present.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PCR_LIB_PATH") or os.path.join(_HERE, "libpcr_hip.so")   # override: A/B builds of the same ABI
INCLUDE_DIR = os.path.join(os.path.dirname(_HERE), "include")

PCR_SOA, PCR_AOS3, PCR_AOS4, PCR_AOS6 = 0, 1, 2, 6
ERRORS = {0: "ok", -1: "bad argument", -2: "HIP error", -3: "out of memory", -4: "bad state",
          -5: "RCCL/collective error", -6: "no correspondence kept"}

_lib = None


class PcrError(RuntimeError):
    pass


class IcpParams(C.Structure):
    _fields_ = [("max_corr", C.c_float), ("max_iter", C.c_uint64), ("eps", C.c_float)]


class IssParams(C.Structure):
    _fields_ = [("local_radius", C.c_float), ("non_max_radius", C.c_float), ("gamma21", C.c_float), ("gamma32", C.c_float),
                ("min_neighbors", C.c_int), ("weighted_covariance", C.c_int)]


class IcpStats(C.Structure):
    _fields_ = [("iters_run", C.c_uint64), ("converged", C.c_int32), ("empty_pairs", C.c_int32),
                ("last_pairs", C.c_uint64), ("last_loss", C.c_float), ("reserved", C.c_float),
                ("ms_total", C.c_double), ("ms_nn", C.c_double), ("nn_launches", C.c_uint64)]


class MfmaCheck(C.Structure):
    _fields_ = [("f16_ok", C.c_int32), ("bf16_ok", C.c_int32), ("f16_worst", C.c_double * 4), ("bf16_worst", C.c_double * 4),
                ("check_ms", C.c_double), ("last_nn1_kernel", C.c_char * 16)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int)

# every symbol include/pcr.h declares (checked by tests/test_abi.py against the header text)
ABI_SYMBOLS = [
    "pcr_device_count", "pcr_ctx_create", "pcr_ctx_destroy", "pcr_ctx_sync", "pcr_ctx_last_error", "pcr_version", "pcr_ctx_device_info",
    "pcr_cloud_create", "pcr_cloud_clone", "pcr_cloud_assign", "pcr_cloud_read", "pcr_cloud_size", "pcr_cloud_destroy",
    "pcr_nn1_f32", "pcr_nn1_f32_async", "pcr_nn1_fetch", "pcr_transform_f32", "pcr_kabsch_sums", "pcr_kabsch_solve", "pcr_kabsch_grid_exponent", "pcr_kabsch_limbs_to_sums",
    "pcr_icp_p2p_f32", "pcr_plane_count_f64", "pcr_plane_mask_f64", "pcr_knn_f64", "pcr_radius_f64",
    "pcr_comm_unique_id", "pcr_comm_init_rccl", "pcr_comm_init_callback", "pcr_comm_destroy", "pcr_comm_selftest", "pcr_shard_range",
    "pcr_prof_reset", "pcr_prof_get", "pcr_prof_get_each", "pcr_tune_set",
    "pcr_grid_stats", "pcr_nn1_stats", "pcr_selftest_mfma_bf16", "pcr_selftest_mfma_f16", "pcr_selftest_mfma_bf16_v2", "pcr_selftest_mfma_f16_v2", "pcr_selftest_sign_f16", "pcr_selftest_sphere_f16", "pcr_ctx_mfma_check", "pcr_voxel_filter_f32", "pcr_iss_keypoints_f32", "pcr_icp_p2plane_f32", "pcr_cloud_knn_f64", "pcr_normals_knn_f64", "pcr_cloud_pca_f64", "pcr_fast_eigen3x3", "pcr_ground_seeds_f64", "pcr_ground_detection_f64",
    "pcr_nn1_desc_f32", "pcr_match_union_f32", "pcr_match_inter_f32", "pcr_ransac_sample_quads", "pcr_consensus_count_f32", "pcr_ransac_global_f32", "pcr_db64_create", "pcr_db64_destroy", "pcr_db64_size", "pcr_db64_knn", "pcr_db64_radius",
    "pcr_ctx_trim", "pcr_ctx_parked_bytes", "pcr_cloud_shard_spatial", "pcr_cloud_global_index", "pcr_cloud_sort_for_target", "pcr_nn1_f32_loop",
    "pcr_db64_radius_rows", "pcr_rows_destroy", "pcr_rows_info", "pcr_rows_row_ptr", "pcr_rows_fetch", "pcr_rows_reduce", "pcr_rows_moments",
]


def lib():
    """Load libpcr_hip.so (once).  Fails loudly when it has not been built — no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PcrError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, sz, f32p, f64p = C.c_void_p, C.c_size_t, C.POINTER(C.c_float), C.POINTER(C.c_double)
    L.pcr_version.restype = C.c_char_p
    L.pcr_ctx_last_error.restype = C.c_char_p
    L.pcr_ctx_last_error.argtypes = [vp]
    L.pcr_device_count.argtypes = [C.POINTER(C.c_int)]
    L.pcr_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.pcr_ctx_destroy.argtypes = [vp]
    L.pcr_ctx_sync.argtypes = [vp]
    L.pcr_ctx_device_info.argtypes = [vp, C.c_char_p, sz, C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
    L.pcr_cloud_create.argtypes = [vp, vp, sz, C.c_int, C.POINTER(vp)]
    L.pcr_cloud_clone.argtypes = [vp, vp, C.POINTER(vp)]
    L.pcr_cloud_assign.argtypes = [vp, vp, vp]
    L.pcr_cloud_read.argtypes = [vp, vp, vp, C.c_int]
    L.pcr_cloud_size.restype = sz
    L.pcr_cloud_size.argtypes = [vp]
    L.pcr_cloud_destroy.argtypes = [vp, vp]
    L.pcr_nn1_f32.argtypes = [vp, vp, vp, vp, vp]
    L.pcr_nn1_f32_async.argtypes = [vp, vp, vp]
    L.pcr_nn1_fetch.argtypes = [vp, sz, vp, vp]
    L.pcr_ctx_trim.argtypes = [vp]
    L.pcr_cloud_shard_spatial.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.pcr_cloud_global_index.argtypes = [vp, vp, vp]
    L.pcr_ctx_parked_bytes.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.pcr_cloud_sort_for_target.argtypes = [vp, vp, vp, vp]
    L.pcr_nn1_f32_loop.argtypes = [vp, vp, vp, C.c_float]
    L.pcr_transform_f32.argtypes = [vp, vp, vp]
    L.pcr_kabsch_sums.argtypes = [vp, vp, vp, C.c_float, vp, C.POINTER(C.c_int64), C.POINTER(C.c_float)]
    L.pcr_kabsch_solve.argtypes = [vp, vp, vp]
    L.pcr_kabsch_grid_exponent.argtypes = [C.c_float, C.c_float]
    L.pcr_kabsch_limbs_to_sums.argtypes = [vp, C.c_int, vp]
    L.pcr_icp_p2p_f32.argtypes = [vp, vp, vp, vp, C.POINTER(IcpParams), vp, C.POINTER(IcpStats)]
    L.pcr_plane_count_f64.argtypes = [vp, vp, vp, sz, C.c_double, vp]
    L.pcr_plane_mask_f64.argtypes = [vp, vp, vp, C.c_double, vp, C.POINTER(C.c_int64)]
    L.pcr_knn_f64.argtypes = [vp, vp, sz, vp, sz, C.c_int, vp, vp]
    L.pcr_radius_f64.argtypes = [vp, vp, sz, vp, sz, C.c_double, vp, vp, vp]
    L.pcr_db64_create.argtypes = [vp, vp, sz, C.POINTER(vp)]
    L.pcr_db64_destroy.argtypes = [vp, vp]
    L.pcr_db64_size.restype = sz
    L.pcr_db64_size.argtypes = [vp]
    L.pcr_db64_knn.argtypes = [vp, vp, vp, sz, C.c_int, C.c_int, vp, vp]
    L.pcr_db64_radius.argtypes = [vp, vp, vp, sz, C.c_double, vp, vp, vp]
    L.pcr_db64_radius_rows.argtypes = [vp, vp, vp, sz, C.c_double, C.POINTER(vp)]
    L.pcr_rows_destroy.argtypes = [vp, vp]
    L.pcr_rows_info.argtypes = [vp, C.POINTER(sz), C.POINTER(C.c_uint64)]
    L.pcr_rows_row_ptr.argtypes = [vp, vp]
    L.pcr_rows_fetch.argtypes = [vp, vp, sz, sz, vp, vp]
    L.pcr_rows_reduce.argtypes = [vp, vp, C.c_int, vp]
    L.pcr_rows_moments.argtypes = [vp, vp, vp, vp]
    L.pcr_comm_unique_id.argtypes = [vp]
    L.pcr_comm_init_rccl.argtypes = [vp, C.c_int, C.c_int, vp]
    L.pcr_comm_init_callback.argtypes = [vp, C.c_int, C.c_int, ALLREDUCE_FN, vp]
    L.pcr_comm_destroy.argtypes = [vp]
    L.pcr_comm_selftest.argtypes = [vp]
    L.pcr_shard_range.restype = None
    L.pcr_shard_range.argtypes = [sz, C.c_int, C.c_int, C.POINTER(sz), C.POINTER(sz)]
    L.pcr_prof_reset.argtypes = [vp]
    L.pcr_prof_get.argtypes = [vp, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
    L.pcr_prof_get_each.argtypes = [vp, C.c_char_p, vp, sz, C.POINTER(sz)]
    L.pcr_tune_set.argtypes = [vp, C.c_char_p, C.c_int64]
    L.pcr_grid_stats.argtypes = [vp, vp]
    L.pcr_nn1_stats.argtypes = [vp, vp]
    L.pcr_selftest_mfma_bf16.argtypes = [vp, C.c_int, vp]
    L.pcr_selftest_mfma_f16.argtypes = [vp, C.c_int, vp]
    L.pcr_selftest_mfma_bf16_v2.argtypes = [vp, C.c_int, vp]
    L.pcr_selftest_mfma_f16_v2.argtypes = [vp, C.c_int, vp]
    L.pcr_selftest_sign_f16.argtypes = [vp, C.c_int, vp]
    L.pcr_selftest_sphere_f16.argtypes = [vp, C.c_int, vp]
    L.pcr_ctx_mfma_check.argtypes = [vp, C.c_int, C.POINTER(MfmaCheck)]
    L.pcr_voxel_filter_f32.argtypes = [vp, vp, C.c_double, C.POINTER(vp)]
    L.pcr_iss_keypoints_f32.argtypes = [vp, vp, C.POINTER(IssParams), vp, vp, vp, C.POINTER(C.c_uint64)]
    L.pcr_icp_p2plane_f32.argtypes = [vp, vp, vp, vp, vp, C.POINTER(IcpParams), vp, C.POINTER(IcpStats)]
    L.pcr_cloud_knn_f64.argtypes = [vp, vp, vp, C.c_int, C.c_double, C.c_int, vp, vp, vp]
    L.pcr_normals_knn_f64.argtypes = [vp, vp, C.c_int, C.c_double, vp]
    L.pcr_cloud_pca_f64.argtypes = [vp, vp, vp, vp, vp]
    L.pcr_fast_eigen3x3.argtypes = [vp, vp]
    L.pcr_ground_seeds_f64.argtypes = [vp, vp, sz, C.c_double, vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.pcr_ground_detection_f64.argtypes = [vp, vp, C.c_int, sz, C.c_double, vp, vp, C.POINTER(C.c_uint64)]
    L.pcr_nn1_desc_f32.argtypes = [vp, vp, sz, vp, sz, C.c_int, vp, vp]
    L.pcr_match_union_f32.argtypes = [vp, vp, sz, vp, sz, C.c_int, C.c_float, vp, vp, C.POINTER(sz)]
    L.pcr_match_inter_f32.argtypes = [vp, vp, sz, vp, sz, C.c_int, C.c_float, vp, vp, C.POINTER(sz)]
    L.pcr_ransac_sample_quads.argtypes = [vp, sz, vp, sz, sz, C.c_uint64, vp]
    L.pcr_consensus_count_f32.argtypes = [vp, vp, sz, vp, sz, vp, sz, vp, sz, C.c_float, vp]
    L.pcr_ransac_global_f32.argtypes = [vp, vp, sz, vp, sz, vp, sz, vp, sz, C.c_float, vp, vp, C.POINTER(C.c_uint32), C.POINTER(C.c_int64), vp]
    _lib = L
    return L


def device_count() -> int:
    """GPUs visible to this process (0 when there is none or the HIP runtime cannot start)."""
    n = C.c_int()
    return n.value if lib().pcr_device_count(C.byref(n)) == 0 else 0


def shard_range(n: int, nranks: int, rank: int):
    """Contiguous shard [begin, end) of n source points for `rank` (pcr_shard_range; host logic, no GPU)."""
    b, e = C.c_size_t(), C.c_size_t()
    lib().pcr_shard_range(n, nranks, rank, C.byref(b), C.byref(e))
    return b.value, e.value


def fast_eigen3x3(A):
    """mylib.FastEigen3x3 (Homework1/.../mylib.cpp:105-189): eigenvector of the smallest eigenvalue (host, no GPU)."""
    a = np.ascontiguousarray(A, np.float64).reshape(9)
    out = np.zeros(3, np.float64)
    lib().pcr_fast_eigen3x3(a.ctypes.data, out.ctypes.data)
    return out


def ransac_sample_quads(src_xyz, pairs, n_hyp, seed):
    """The sampling loop of Registration::RANSAC (registration.cpp:318-352), explicit seed (host logic, no GPU)."""
    src = np.ascontiguousarray(src_xyz, np.float32)
    p = np.ascontiguousarray(pairs, np.uint32)
    quads = np.zeros((n_hyp, 4), np.uint32)
    rc = lib().pcr_ransac_sample_quads(src.ctypes.data, src.shape[0], p.ctypes.data, p.shape[0], n_hyp, seed, quads.ctypes.data)
    if rc != 0:
        raise PcrError(f"pcr_ransac_sample_quads failed (rc = {rc})")
    return quads


def kabsch_grid_exponent(target_absmax: float, max_corr: float) -> int:
    """e of the fixed-point grid of the exact Kabsch sums: 2^e bounds every coordinate of a kept pair (host logic, no GPU)."""
    return int(lib().pcr_kabsch_grid_exponent(float(target_absmax), float(max_corr)))


def kabsch_limbs_to_sums(row, e: int):
    """A (summed) row of 55 limbs -> the 16 moments; carries are propagated in place on a copy (host logic, no GPU)."""
    r = np.ascontiguousarray(row, np.float64)[:55].copy()
    sums = np.zeros(16, np.float64)
    rc = lib().pcr_kabsch_limbs_to_sums(r.ctypes.data, int(e), sums.ctypes.data)
    if rc != 0:
        raise PcrError(f"pcr_kabsch_limbs_to_sums failed (rc = {rc})")
    return sums


def kabsch_solve(sums):
    """(R 3x3 f32, t f32[3]) from the 16 f64 moments — registration.cpp:979-998 (host, no GPU needed)."""
    s = np.ascontiguousarray(sums, np.float64)
    R = np.zeros(9, np.float32)
    t = np.zeros(3, np.float32)
    rc = lib().pcr_kabsch_solve(s.ctypes.data, R.ctypes.data, t.ctypes.data)
    return rc, R.reshape(3, 3), t


class Cloud:
    """An N-point f32 cloud resident in HBM (SoA)."""

    def __init__(self, ctx: "Context", handle):
        self.ctx = ctx
        self.h = handle
        ctx._handles.add(self)        # freed with the context at the latest

    def __del__(self):
        try:
            self.free()
        except Exception:   # noqa: BLE001  (interpreter shutdown)
            pass

    def __len__(self):
        return int(lib().pcr_cloud_size(self.h))

    def numpy(self) -> np.ndarray:
        out = np.empty((3, len(self)), np.float32)
        self.ctx._ck(lib().pcr_cloud_read(self.ctx.h, self.h, out.ctypes.data, PCR_SOA))
        return out

    def clone(self) -> "Cloud":
        h = C.c_void_p()
        self.ctx._ck(lib().pcr_cloud_clone(self.ctx.h, self.h, C.byref(h)))
        return Cloud(self.ctx, h)

    def assign(self, other: "Cloud"):
        self.ctx._ck(lib().pcr_cloud_assign(self.ctx.h, self.h, other.h))

    def free(self):
        if self.h and self.ctx.h:
            lib().pcr_cloud_destroy(self.ctx.h, self.h)
        self.h = None


class Db64:
    """An n x 3 f64 database resident in HBM: built once, queried many times (k-NN / radius)."""

    def __init__(self, ctx: "Context", handle):
        self.ctx = ctx
        self.h = handle
        ctx._handles.add(self)

    def __del__(self):
        try:
            self.free()
        except Exception:   # noqa: BLE001
            pass

    def __len__(self):
        return int(lib().pcr_db64_size(self.h))

    def knn(self, q, k: int, squared: bool = False):
        q = np.ascontiguousarray(q, np.float64).reshape(-1, 3)
        idx = np.zeros((q.shape[0], k), np.int32)
        dist = np.zeros((q.shape[0], k), np.float64)
        self.ctx._ck(lib().pcr_db64_knn(self.ctx.h, self.h, q.ctypes.data, q.shape[0], k, int(squared),
                                        idx.ctypes.data, dist.ctypes.data))
        return idx, dist

    def radius(self, q, r: float):
        q = np.ascontiguousarray(q, np.float64).reshape(-1, 3)
        row = np.zeros(q.shape[0] + 1, np.int64)
        self.ctx._ck(lib().pcr_db64_radius(self.ctx.h, self.h, q.ctypes.data, q.shape[0], r, row.ctypes.data, None, None))
        total = int(row[-1])
        idx = np.zeros(max(total, 1), np.int32)
        dist = np.zeros(max(total, 1), np.float64)
        if total:
            self.ctx._ck(lib().pcr_db64_radius(self.ctx.h, self.h, q.ctypes.data, q.shape[0], r, row.ctypes.data,
                                               idx.ctypes.data, dist.ctypes.data))
        return row, idx[:total], dist[:total]

    def radius_rows(self, q, r: float) -> "Rows":
        """The same search with its rows kept in HBM (q = None: every point of the database queries the database)."""
        h = C.c_void_p()
        if q is None:
            self.ctx._ck(lib().pcr_db64_radius_rows(self.ctx.h, self.h, None, 0, r, C.byref(h)))
        else:
            q = np.ascontiguousarray(q, np.float64).reshape(-1, 3)
            self.ctx._ck(lib().pcr_db64_radius_rows(self.ctx.h, self.h, q.ctypes.data, q.shape[0], r, C.byref(h)))
        return Rows(self.ctx, h, self)

    def free(self):
        if self.h and self.ctx.h:
            lib().pcr_db64_destroy(self.ctx.h, self.h)
        self.h = None


class Rows:
    """Device-resident CSR rows of a radius search (include/pcr.h pcr_rows): reduce them on the GPU or fetch them block by block."""
    COUNT, SUM_DIST, MAX_DIST = 0, 1, 2

    def __init__(self, ctx: "Context", handle, db: "Db64"):
        self.ctx, self.h, self.db = ctx, handle, db          # (db kept alive: the indices refer to it)
        ctx._handles.add(self)
        m, total = C.c_size_t(), C.c_uint64()
        lib().pcr_rows_info(self.h, C.byref(m), C.byref(total))
        self.m, self.total = m.value, total.value

    def __del__(self):
        try:
            self.free()
        except Exception:   # noqa: BLE001
            pass

    def row_ptr(self):
        row = np.zeros(self.m + 1, np.int64)
        self.ctx._ck(lib().pcr_rows_row_ptr(self.h, row.ctypes.data))
        return row

    def fetch(self, row_begin: int, row_end: int, row_ptr=None):
        row = self.row_ptr() if row_ptr is None else row_ptr
        cnt = int(row[row_end] - row[row_begin])
        idx, dist = np.zeros(max(cnt, 1), np.int32), np.zeros(max(cnt, 1), np.float64)
        self.ctx._ck(lib().pcr_rows_fetch(self.ctx.h, self.h, row_begin, row_end, idx.ctypes.data, dist.ctypes.data))
        return idx[:cnt], dist[:cnt]

    def reduce(self, op: int):
        out = np.zeros(self.m, np.float64)
        self.ctx._ck(lib().pcr_rows_reduce(self.ctx.h, self.h, int(op), out.ctypes.data))
        return out

    def moments(self):
        mean, cov = np.zeros((self.m, 3), np.float64), np.zeros((self.m, 6), np.float64)
        self.ctx._ck(lib().pcr_rows_moments(self.ctx.h, self.h, mean.ctypes.data, cov.ctypes.data))
        return mean, cov

    def free(self):
        if self.h and self.ctx.h:
            lib().pcr_rows_destroy(self.ctx.h, self.h)
        self.h = None
        self.ctx._handles.discard(self)


def read_kitti_bin(path: str, floats_per_point: int = 4) -> np.ndarray:
    """A velodyne .bin as the reference reads it: N x 4 f32 rows x, y, z, intensity (read_velodyne_bin,
    Homework4/ground_detection_ransac.py:23-34; Homework2/hw2/include/test.hpp:26-28 without its EOF duplicate) or the hw9
    registration format N x 6 f32 xyz + normal (Homework9/hw9/src/registration.cpp:25-26).  Returns the (n, k) f32 rows;
    pass them to Context.cloud(rows, PCR_AOS4 / PCR_AOS6)."""
    a = np.fromfile(path, dtype=np.float32)
    return a[: a.size // floats_per_point * floats_per_point].reshape(-1, floats_per_point)


class Context:
    """One GPU, one HIP stream, one workspace (pcr_ctx)."""

    def __init__(self, device: int = 0):
        h = C.c_void_p()
        rc = lib().pcr_ctx_create(device, C.byref(h))
        if rc != 0:
            raise PcrError(f"pcr_ctx_create(device={device}) failed: {ERRORS.get(rc, rc)} — an MI355X is required "
                           "(no CPU fallback)")
        self.h = h
        self._cb_keepalive = None
        self._handles = weakref.WeakSet()      # clouds / databases created on this context

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001  (interpreter shutdown)
            pass

    def _ck(self, rc: int):
        if rc != 0:
            raise PcrError(f"{ERRORS.get(rc, rc)}: {lib().pcr_ctx_last_error(self.h).decode()}")

    def close(self):
        """Frees every cloud / database still alive on this context, then the context (stream, workspace, communicator)."""
        if self.h:
            for obj in list(self._handles):
                obj.free()
            lib().pcr_ctx_destroy(self.h)
            self.h = None

    def sync(self):
        self._ck(lib().pcr_ctx_sync(self.h))

    def device_info(self):
        arch = C.create_string_buffer(64)
        ncu = C.c_int()
        hbm = C.c_uint64()
        self._ck(lib().pcr_ctx_device_info(self.h, arch, 64, C.byref(ncu), C.byref(hbm)))
        return {"arch": arch.value.decode(), "cus": ncu.value, "hbm_bytes": hbm.value, "mfma_check": self.mfma_check()}

    def tune(self, key: str, value: int):
        self._ck(lib().pcr_tune_set(self.h, key.encode(), int(value)))

    def grid_stats(self):
        out = (C.c_uint64 * 4)()
        self._ck(lib().pcr_grid_stats(self.h, out))
        return {"candidates": out[0], "fine_rows": out[1], "coarse_rows": out[2], "far_stages": out[3]}

    def nn1_stats(self):
        """the sixteen diagnostics words of the last 1-NN launch made with tune grid_stats = 1 (include/pcr.h)"""
        out = (C.c_uint64 * 16)()
        self._ck(lib().pcr_nn1_stats(self.h, out))
        return [int(v) for v in out]

    def selftest_mfma_bf16(self, trials: int = 64):
        """(accumulation error on random operands in 2^-24 sum|a b|, filter-value error in 2^-24 (|r|^2 + |t|^2), absolute error in 2^-24
        in the small-magnitude regime, accumulation error on the structured tiles) measured on this device — include/pcr.h"""
        out = (C.c_double * 4)()
        self._ck(lib().pcr_selftest_mfma_bf16_v2(self.h, int(trials), out))
        return tuple(float(v) for v in out)

    def selftest_mfma_f16(self, trials: int = 64):
        """the same for the f16 form (one MFMA per tile, two-piece scaled operands)"""
        out = (C.c_double * 4)()
        self._ck(lib().pcr_selftest_mfma_f16_v2(self.h, int(trials), out))
        return tuple(float(v) for v in out)

    def selftest_sign_f16(self, trials: int = 64):
        """STRACK's decision checked on this device (include/pcr.h): (pairs at or below their query's threshold, of those without the
        sign set — must be 0 —, pairs with the sign set, pairs in all)"""
        out = (C.c_uint64 * 4)()
        self._ck(lib().pcr_selftest_sign_f16(self.h, int(trials), out))
        return tuple(int(v) for v in out)

    def selftest_sphere_f16(self, trials: int = 64):
        """(pairs that must be flagged, of those missed, pairs flagged, pairs) of the chunk-sphere rows of the sign filter (STRACK3)"""
        out = (C.c_uint64 * 4)()
        self._ck(lib().pcr_selftest_sphere_f16(self.h, int(trials), out))
        return tuple(int(v) for v in out)

    def mfma_check(self, run_now: bool = False):
        """The verdicts of the library's own once-per-context check of the matrix-core arithmetic (-1 = not run yet), the figures
        behind them, the host time they took and the kernel family of the last 1-NN search."""
        m = MfmaCheck()
        self._ck(lib().pcr_ctx_mfma_check(self.h, 1 if run_now else 0, C.byref(m)))
        return {"f16_ok": m.f16_ok, "bf16_ok": m.bf16_ok, "f16_worst": [float(v) for v in m.f16_worst], "bf16_worst": [float(v) for v in m.bf16_worst],
                "check_ms": m.check_ms, "last_nn1_kernel": m.last_nn1_kernel.decode()}

    def prof_reset(self):
        self._ck(lib().pcr_prof_reset(self.h))

    def prof_get(self, kernel: str):
        n, ms = C.c_uint64(), C.c_double()
        self._ck(lib().pcr_prof_get(self.h, kernel.encode(), C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def prof_get_each(self, kernel: str):
        """the individual durations (ms) of the named scope since the last prof_reset, in launch order"""
        buf = np.zeros(4096, np.float64)
        n = C.c_size_t()
        self._ck(lib().pcr_prof_get_each(self.h, kernel.encode(), buf.ctypes.data, buf.size, C.byref(n)))
        return buf[: min(n.value, buf.size)].copy()

    # ---- clouds
    def cloud(self, xyz: np.ndarray, layout: int = PCR_SOA) -> Cloud:
        """SOA: (3, n) f32; AOS3: (n, 3); AOS4: (n, 4) (KITTI .bin rows, pcl::PointXYZ); AOS6: (n, 6) (hw9 xyz + normal rows)."""
        a = np.ascontiguousarray(xyz, np.float32)
        n = a.shape[1] if layout == PCR_SOA else a.shape[0]
        if a.size == 0:
            n = 0
        h = C.c_void_p()
        self._ck(lib().pcr_cloud_create(self.h, a.ctypes.data if n else None, n, layout, C.byref(h)))
        return Cloud(self, h)

    # ---- A6 search
    def nn1(self, tgt: Cloud, src: Cloud):
        n = len(src)
        idx = np.empty(n, np.uint32)
        d2 = np.empty(n, np.float32)
        self._ck(lib().pcr_nn1_f32(self.h, tgt.h, src.h, idx.ctypes.data, d2.ctypes.data))
        return idx, d2

    def nn1_async(self, tgt: Cloud, src: Cloud):
        self._ck(lib().pcr_nn1_f32_async(self.h, tgt.h, src.h))

    def nn1_loop(self, tgt: Cloud, src: Cloud, max_corr: float):
        """one search of a caller's own ICP-style loop: seeded by the previous call, bounded by the gate (registration.cpp:936)"""
        self._ck(lib().pcr_nn1_f32_loop(self.h, tgt.h, src.h, C.c_float(max_corr)))

    def sort_for_target(self, tgt: Cloud, cloud: Cloud) -> np.ndarray:
        """re-orders `cloud` in place into the order of the target's index; returns the original index of every position"""
        orig = np.empty(len(cloud), np.uint32)
        self._ck(lib().pcr_cloud_sort_for_target(self.h, tgt.h, cloud.h, orig.ctypes.data if len(cloud) else None))
        return orig

    def shard_spatial(self, tgt: Cloud, full: Cloud, nranks: int, rank: int, chunks_per_rank: int = 0) -> Cloud:
        """this rank's share of `full` under the spatially coherent partition (pcr_cloud_shard_spatial)"""
        h = C.c_void_p()
        self._ck(lib().pcr_cloud_shard_spatial(self.h, tgt.h, full.h, nranks, rank, chunks_per_rank, C.byref(h)))
        return Cloud(self, h)

    def global_index(self, shard: Cloud) -> np.ndarray:
        idx = np.empty(len(shard), np.uint32)
        self._ck(lib().pcr_cloud_global_index(self.h, shard.h, idx.ctypes.data if len(shard) else None))
        return idx

    def trim(self):
        self._ck(lib().pcr_ctx_trim(self.h))

    def parked_bytes(self) -> int:
        b = C.c_uint64()
        self._ck(lib().pcr_ctx_parked_bytes(self.h, C.byref(b)))
        return int(b.value)

    def nn1_fetch(self, n: int):
        idx = np.empty(n, np.uint32)
        d2 = np.empty(n, np.float32)
        self._ck(lib().pcr_nn1_fetch(self.h, n, idx.ctypes.data, d2.ctypes.data))
        return idx, d2

    # ---- N3
    def voxel_filter(self, cloud: Cloud, leaf_size: float) -> Cloud:
        """Homework1 voxel_filter(point_cloud, leaf_size) (centroid mode) -> new device cloud."""
        h = C.c_void_p()
        self._ck(lib().pcr_voxel_filter_f32(self.h, cloud.h, float(leaf_size), C.byref(h)))
        return Cloud(self, h)

    def icp_point2plane(self, src: Cloud, tgt: Cloud, tgt_normals: Cloud, init_T=None, max_corr=1.0, max_iter=20, eps=1e-8):
        """Registration::ICPpoint2plane (registration.cpp:710-860) on already-sampled clouds -> (T 4x4, stats)."""
        T0 = np.eye(4, dtype=np.float32) if init_T is None else np.ascontiguousarray(init_T, np.float32)
        out = np.zeros(16, np.float32)
        prm = IcpParams(max_corr, max_iter, eps)
        st = IcpStats()
        self._ck(lib().pcr_icp_p2plane_f32(self.h, src.h, tgt.h, tgt_normals.h, T0.ctypes.data, C.byref(prm), out.ctypes.data, C.byref(st)))
        return out.reshape(4, 4), {k: getattr(st, k) for k, _ in IcpStats._fields_}

    # ---- N1
    def iss_keypoints(self, cloud: Cloud, local_radius, non_max_radius, gamma21=0.9, gamma32=0.9, min_neighbors=5, weighted=True):
        """ISSKeypoint::compute (hw7 iss_detector.cpp:38-110) -> (keypoint indices ascending, lambda3 f32[n], |N_local| u32[n])."""
        n = len(cloud)
        key = np.zeros(max(n, 1), np.uint8)
        l3 = np.zeros(max(n, 1), np.float32)
        cn = np.zeros(max(n, 1), np.uint32)
        prm = IssParams(local_radius, non_max_radius, gamma21, gamma32, int(min_neighbors), int(bool(weighted)))
        cnt = C.c_uint64()
        self._ck(lib().pcr_iss_keypoints_f32(self.h, cloud.h, C.byref(prm), key.ctypes.data, l3.ctypes.data, cn.ctypes.data, C.byref(cnt)))
        idx = np.flatnonzero(key[:n])
        assert idx.size == cnt.value
        return idx, l3[:n], cn[:n]

    def cloud_knn(self, db: Cloud, queries: Cloud, k: int, radius: float = -1.0, squared: bool = True):
        """Exact grid k-NN between resident clouds -> (idx i32 [m,k], dist f64 [m,k], found u32 [m]); see pcr_cloud_knn_f64."""
        m = len(queries)
        idx = np.zeros((max(m, 1), k), np.int32)
        dist = np.zeros((max(m, 1), k), np.float64)
        found = np.zeros(max(m, 1), np.uint32)
        self._ck(lib().pcr_cloud_knn_f64(self.h, db.h, queries.h, int(k), float(radius), int(bool(squared)), idx.ctypes.data, dist.ctypes.data,
                                         found.ctypes.data))
        return idx[:m], dist[:m], found[:m]

    def normals(self, cloud: Cloud, k: int = 10, radius: float = 5.0):
        """pca_normal.py:89-103 -> normals f64 [n,3] (hybrid search radius / max_nn = k, FastEigen3x3 eigenvector)."""
        n = len(cloud)
        out = np.zeros((max(n, 1), 3), np.float64)
        self._ck(lib().pcr_normals_knn_f64(self.h, cloud.h, int(k), float(radius), out.ctypes.data))
        return out[:n]

    def pca(self, cloud: Cloud):
        """pca_normal.py PCA(data) -> (eigenvalues descending f64[3], eigenvectors in columns f64[3,3], centre f64[3])."""
        w = np.zeros(3, np.float64); v = np.zeros(9, np.float64); c = np.zeros(3, np.float64)
        self._ck(lib().pcr_cloud_pca_f64(self.h, cloud.h, w.ctypes.data, v.ctypes.data, c.ctypes.data))
        return w, v.reshape(3, 3), c

    # ---- N2
    def ground_seeds(self, cloud: Cloud, lpr_size: int, threshold_seeds: float):
        """extract_initial_seeds (ground_detection_SVD.py:46-71) -> (seed mask bool[n], LPR_z + threshold)."""
        n = len(cloud)
        mask = np.zeros(max(n, 1), np.uint8)
        ub = C.c_double()
        cnt = C.c_uint64()
        self._ck(lib().pcr_ground_seeds_f64(self.h, cloud.h, int(lpr_size), float(threshold_seeds), mask.ctypes.data, C.byref(ub), C.byref(cnt)))
        return mask[:n].astype(bool), ub.value

    def ground_detection(self, cloud: Cloud, max_iter: int, lpr_size: int, threshold_dist: float):
        """ground_detection (ground_detection_SVD.py:88-101) -> (params f64[4], inlier mask bool[n])."""
        n = len(cloud)
        mask = np.zeros(max(n, 1), np.uint8)
        params = np.zeros(4, np.float64)
        cnt = C.c_uint64()
        self._ck(lib().pcr_ground_detection_f64(self.h, cloud.h, int(max_iter), int(lpr_size), float(threshold_dist), params.ctypes.data,
                                                mask.ctypes.data, C.byref(cnt)))
        return params, mask[:n].astype(bool)

    # ---- N4
    def nn1_desc(self, db, q):
        """1-NN between descriptor sets (rows), nanoflann L2 arithmetic at any dim -> (idx u32, d2 f32)."""
        db = np.ascontiguousarray(db, np.float32)
        q = np.ascontiguousarray(q, np.float32)
        idx = np.zeros(max(q.shape[0], 1), np.uint32)
        d2 = np.zeros(max(q.shape[0], 1), np.float32)
        self._ck(lib().pcr_nn1_desc_f32(self.h, db.ctypes.data, db.shape[0], q.ctypes.data, q.shape[0], db.shape[1], idx.ctypes.data, d2.ctypes.data))
        return idx[: q.shape[0]], d2[: q.shape[0]]

    def match_union(self, desc_src, desc_tgt, rejection_rate):
        """findRANSACCorrespondencesUnion (registration.cpp:535-615) -> (pairs [K, 2] (src, tgt), dist [K])."""
        a = np.ascontiguousarray(desc_src, np.float32)
        b = np.ascontiguousarray(desc_tgt, np.float32)
        total = a.shape[0] + b.shape[0]
        pairs = np.zeros((max(total, 1), 2), np.uint32)
        dist = np.zeros(max(total, 1), np.float32)
        k = C.c_size_t()
        self._ck(lib().pcr_match_union_f32(self.h, a.ctypes.data, a.shape[0], b.ctypes.data, b.shape[0], a.shape[1], rejection_rate,
                                           pairs.ctypes.data, dist.ctypes.data, C.byref(k)))
        return pairs[: k.value], dist[: k.value]

    def match_inter(self, desc_src, desc_tgt, rejection_rate):
        """findRANSACCorrespondencesInter (registration.cpp:437-533) -> (pairs [K, 2] (src, tgt), dist [K])."""
        a = np.ascontiguousarray(desc_src, np.float32)
        b = np.ascontiguousarray(desc_tgt, np.float32)
        pairs = np.zeros((max(a.shape[0], 1), 2), np.uint32)
        dist = np.zeros(max(a.shape[0], 1), np.float32)
        k = C.c_size_t()
        self._ck(lib().pcr_match_inter_f32(self.h, a.ctypes.data, a.shape[0], b.ctypes.data, b.shape[0], a.shape[1], rejection_rate,
                                           pairs.ctypes.data, dist.ctypes.data, C.byref(k)))
        return pairs[: k.value], dist[: k.value]

    def consensus_count(self, src_xyz, tgt_xyz, pairs, Rt, thr):
        """consensus-set sizes (registration.cpp:395-421) of poses Rt [H, 12] = (R row-major, t)."""
        s = np.ascontiguousarray(src_xyz, np.float32)
        t = np.ascontiguousarray(tgt_xyz, np.float32)
        p = np.ascontiguousarray(pairs, np.uint32)
        rt = np.ascontiguousarray(Rt, np.float32).reshape(-1, 12)
        counts = np.zeros(max(rt.shape[0], 1), np.uint32)
        self._ck(lib().pcr_consensus_count_f32(self.h, s.ctypes.data, s.shape[0], t.ctypes.data, t.shape[0], p.ctypes.data, p.shape[0],
                                               rt.ctypes.data, rt.shape[0], thr, counts.ctypes.data))
        return counts[: rt.shape[0]]

    def ransac_global(self, src_xyz, tgt_xyz, pairs, quads, thr):
        """Registration::RANSAC over given quads -> (winner, R 3x3, t, best count, counts [H])."""
        s = np.ascontiguousarray(src_xyz, np.float32)
        t = np.ascontiguousarray(tgt_xyz, np.float32)
        p = np.ascontiguousarray(pairs, np.uint32)
        qd = np.ascontiguousarray(quads, np.uint32).reshape(-1, 4)
        R = np.zeros(9, np.float32)
        tv = np.zeros(3, np.float32)
        best = C.c_uint32()
        win = C.c_int64()
        counts = np.zeros(max(qd.shape[0], 1), np.uint32)
        self._ck(lib().pcr_ransac_global_f32(self.h, s.ctypes.data, s.shape[0], t.ctypes.data, t.shape[0], p.ctypes.data, p.shape[0],
                                             qd.ctypes.data, qd.shape[0], thr, R.ctypes.data, tv.ctypes.data, C.byref(best), C.byref(win),
                                             counts.ctypes.data))
        return win.value, R.reshape(3, 3), tv, best.value, counts[: qd.shape[0]]

    # ---- A8 / A7
    def transform(self, cloud: Cloud, T):
        T = np.ascontiguousarray(T, np.float32).reshape(16)
        self._ck(lib().pcr_transform_f32(self.h, cloud.h, T.ctypes.data))

    def kabsch_sums(self, tgt: Cloud, src: Cloud, max_corr: float):
        sums = np.zeros(16, np.float64)
        last = C.c_int64()
        d2 = C.c_float()
        self._ck(lib().pcr_kabsch_sums(self.h, tgt.h, src.h, max_corr, sums.ctypes.data, C.byref(last), C.byref(d2)))
        return sums, last.value, d2.value

    # ---- A9
    def icp_point2point(self, src: Cloud, tgt: Cloud, init_T=None, max_corr=1.0, max_iter=20, eps=1e-8):
        """Registration::ICPpoint2point (registration.cpp:862-1011) on already-sampled clouds -> (T 4x4, stats)."""
        T0 = np.eye(4, dtype=np.float32) if init_T is None else np.ascontiguousarray(init_T, np.float32)
        out = np.zeros(16, np.float32)
        prm = IcpParams(max_corr, max_iter, eps)
        st = IcpStats()
        self._ck(lib().pcr_icp_p2p_f32(self.h, src.h, tgt.h, T0.ctypes.data, C.byref(prm), out.ctypes.data, C.byref(st)))
        stats = {f: getattr(st, f) for f, _ in IcpStats._fields_ if f != "reserved"}
        return out.reshape(4, 4), stats

    # ---- A10
    def plane_count(self, pts: Cloud, planes4, thr: float) -> np.ndarray:
        p = np.ascontiguousarray(planes4, np.float64).reshape(-1, 4)
        counts = np.zeros(p.shape[0], np.int64)
        self._ck(lib().pcr_plane_count_f64(self.h, pts.h, p.ctypes.data, p.shape[0], thr, counts.ctypes.data))
        return counts

    def plane_mask(self, pts: Cloud, plane4, thr: float):
        p = np.ascontiguousarray(plane4, np.float64).reshape(4)
        mask = np.zeros(len(pts), np.uint8)
        cnt = C.c_int64()
        self._ck(lib().pcr_plane_mask_f64(self.h, pts.h, p.ctypes.data, thr, mask.ctypes.data, C.byref(cnt)))
        return mask, cnt.value

    # ---- A2/A4/A11
    def knn_f64(self, db, q, k: int):
        db = np.ascontiguousarray(db, np.float64).reshape(-1, 3)
        q = np.ascontiguousarray(q, np.float64).reshape(-1, 3)
        idx = np.zeros((q.shape[0], k), np.int32)
        dist = np.zeros((q.shape[0], k), np.float64)
        self._ck(lib().pcr_knn_f64(self.h, db.ctypes.data, db.shape[0], q.ctypes.data, q.shape[0], k,
                                   idx.ctypes.data, dist.ctypes.data))
        return idx, dist

    def radius_f64(self, db, q, r: float):
        db = np.ascontiguousarray(db, np.float64).reshape(-1, 3)
        q = np.ascontiguousarray(q, np.float64).reshape(-1, 3)
        row = np.zeros(q.shape[0] + 1, np.int64)
        self._ck(lib().pcr_radius_f64(self.h, db.ctypes.data, db.shape[0], q.ctypes.data, q.shape[0], r,
                                      row.ctypes.data, None, None))
        total = int(row[-1])
        idx = np.zeros(max(total, 1), np.int32)
        dist = np.zeros(max(total, 1), np.float64)
        if total:
            self._ck(lib().pcr_radius_f64(self.h, db.ctypes.data, db.shape[0], q.ctypes.data, q.shape[0], r,
                                          row.ctypes.data, idx.ctypes.data, dist.ctypes.data))
        return row, idx[:total], dist[:total]

    def db64(self, db) -> "Db64":
        """Keep an n x 3 f64 database resident in HBM (the GPU-side 'tree')."""
        a = np.ascontiguousarray(db, np.float64).reshape(-1, 3)
        h = C.c_void_p()
        self._ck(lib().pcr_db64_create(self.h, a.ctypes.data if a.size else None, a.shape[0], C.byref(h)))
        return Db64(self, h)

    # ---- multi-GPU
    def comm_init_rccl(self, nranks: int, rank: int, unique_id: bytes):
        buf = C.create_string_buffer(unique_id, 128)
        self._ck(lib().pcr_comm_init_rccl(self.h, nranks, rank, buf))

    def comm_init_callback(self, nranks: int, rank: int, fn):
        """fn(np.ndarray f64 view) must all-reduce(sum) in place; any exception aborts the collective."""
        def _cb(user, buf, n):
            try:
                arr = np.ctypeslib.as_array(buf, shape=(n,))
                fn(arr)
                return 0
            except Exception:   # noqa: BLE001 - reported through the C status
                import traceback
                traceback.print_exc()
                return 1
        self._cb_keepalive = ALLREDUCE_FN(_cb)
        self._ck(lib().pcr_comm_init_callback(self.h, nranks, rank, self._cb_keepalive, None))

    def comm_selftest(self):
        self._ck(lib().pcr_comm_selftest(self.h))

    def comm_destroy(self):
        lib().pcr_comm_destroy(self.h)
        self._cb_keepalive = None


def comm_unique_id() -> bytes:
    buf = C.create_string_buffer(128)
    rc = lib().pcr_comm_unique_id(buf)
    if rc != 0:
        raise PcrError("pcr_comm_unique_id failed (RCCL not loadable)")
    return buf.raw

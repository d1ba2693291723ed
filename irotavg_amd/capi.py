"""ctypes view of the C ABI in include/irotavg_hip.h (plumbing for tests, bench.py and the Python
mirror of the RAL API in irotavg_amd/ral.py). All numerics live in libirotavg_hip.so; this module
never computes rotations itself and raises if the library is missing."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libirotavg_hip.so")

OK = 0
ERR_BAD_ARG, ERR_NOT_SPANNING, ERR_SOLVER, ERR_UNKNOWN_COST = -1, -2, -3, -4
ERR_NOMEM, ERR_HIP, ERR_NO_DEVICE, ERR_NOT_CONVERGED = -5, -6, -7, -8

# symbols declared in include/irotavg_hip.h (kept in sync by tests/test_abi.py)
SYMBOLS = [
    "irotavg_default_options", "irotavg_init_mst", "irotavg_make_A", "irotavg_l1ra", "irotavg_irls",
    "irotavg_quat_normalised", "irotavg_graph_create", "irotavg_graph_destroy",
    "irotavg_graph_set_rotations", "irotavg_graph_get_rotations",
    "irotavg_graph_snapshot_rotations", "irotavg_graph_restore_rotations", "irotavg_graph_get_weights",
    "irotavg_graph_set_weights", "irotavg_graph_irls", "irotavg_graph_l1ra",
    "irotavg_graph_quat_normalised", "irotavg_graph_get_stats", "irotavg_graph_reset_stats",
    "irotavg_graph_synchronize", "irotavg_graph_edge_residual", "irotavg_graph_get_residuals",
    "irotavg_graph_ls_solve", "irotavg_graph_update_weights", "irotavg_graph_apply_step",
    "irotavg_graph_l1decode_pd", "irotavg_graph_time_kernel", "irotavg_graph_fingerprint", "irotavg_version",
    "irotavg_device_count", "irotavg_error_string",
    "irotavg_viewgraph_create", "irotavg_viewgraph_destroy", "irotavg_viewgraph_add_view",
    "irotavg_viewgraph_num_views", "irotavg_viewgraph_connect", "irotavg_viewgraph_fix_pose",
    "irotavg_viewgraph_is_pose_fixed", "irotavg_viewgraph_count_fixed_poses",
    "irotavg_viewgraph_get_pose", "irotavg_viewgraph_set_pose", "irotavg_viewgraph_rot_avg",
    "irotavg_viewgraph_rot_avg_batch", "irotavg_viewgraph_prepare",
    "irotavg_dist_unique_id", "irotavg_dist_create", "irotavg_dist_destroy",
    "irotavg_dist_set_rotations", "irotavg_dist_get_rotations", "irotavg_dist_get_weights",
    "irotavg_dist_snapshot_rotations", "irotavg_dist_restore_rotations",
    "irotavg_dist_irls", "irotavg_dist_get_stats", "irotavg_dist_info", "irotavg_dist_plan", "irotavg_dist_plan_host",
    "irotavg_dist_l1ra", "irotavg_dist_create_hosted",
    "irotavg_graph_direct_info",
    "irotavg_graph_direct_residual",
    "irotavg_window_solve", "irotavg_window_solve_kernel", "irotavg_trim_memory", "irotavg_rmat2quat", "irotavg_quat2rmat", "irotavg_viewgraph_save_poses",
    "irotavg_oneshot_cache", "irotavg_oneshot_cache_clear", "irotavg_oneshot_cache_stats",
    "irotavg_dist_timing",
]


class Options(C.Structure):
    _fields_ = [("pcg_rtol", C.c_double), ("pcg_max_iters", C.c_int), ("pcg_check_every", C.c_int),
                ("mg_levels_max", C.c_int), ("mg_agg0", C.c_int), ("mg_agg", C.c_int),
                ("mg_dense_max", C.c_int), ("mg_omega", C.c_double), ("mg_kc", C.c_double),
                ("device", C.c_int), ("mg_multiplicative_top", C.c_int), ("dense_always_refresh", C.c_int),
                ("no_window_kernel", C.c_int), ("pcg_stall_accept", C.c_int), ("no_fused_pspmv", C.c_int), ("no_lowrank_repair", C.c_int), ("pcg_classic", C.c_int),
                ("band_direct", C.c_int), ("inexact_outer", C.c_int)]


class Stats(C.Structure):
    _fields_ = [("pcg_solves", C.c_int64), ("pcg_iters", C.c_int64), ("pcg_iters_last", C.c_int64),
                ("outer_iters", C.c_int64), ("edge_updates", C.c_int64),
                ("seconds_irls", C.c_double), ("seconds_l1ra", C.c_double), ("levels", C.c_int),
                ("level_rows", C.c_int64 * 16), ("level_nnz", C.c_int64 * 16),
                ("last_relres", C.c_double * 3), ("pcg_stagnated", C.c_int64),
                ("dense_inversions", C.c_int64), ("dense_repairs", C.c_int64),
                ("pcg_handed_over", C.c_int64), ("direct_solves", C.c_int64), ("band", C.c_int64),
                ("band_block", C.c_int64), ("direct_guarded", C.c_int64), ("direct_dead_pivots", C.c_int64),
                ("direct_up_fallbacks", C.c_int64)]


class RotAvgInfo(C.Structure):
    _fields_ = [("skipped", C.c_int), ("n_views", C.c_int), ("n_edges", C.c_int), ("n_fixed", C.c_int),
                ("l1_iters", C.c_int), ("irls_iters", C.c_int), ("l1_runtime", C.c_double),
                ("irls_runtime", C.c_double)]


class IrotavgError(RuntimeError):
    def __init__(self, code, where=""):
        self.code = code
        msg = lib().irotavg_error_string(code).decode()
        super().__init__("%s: %s (code %d)" % (where, msg, code))


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_LIB = None


def lib():
    """Loads libirotavg_hip.so; fails loudly if it has not been built (no fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError("libirotavg_hip.so is missing: run `python -c 'import __graft_entry__ as g; "
                          "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.irotavg_version.restype = C.c_char_p
    L.irotavg_error_string.restype = C.c_char_p
    L.irotavg_error_string.argtypes = [C.c_int]
    L.irotavg_device_count.restype = C.c_int
    L.irotavg_default_options.argtypes = [C.POINTER(Options)]
    L.irotavg_default_options.restype = None
    L.irotavg_init_mst.argtypes = [C.c_int64, C.c_int64, _dp, C.c_int64, _dp, C.c_int64, _ip, C.c_int]
    L.irotavg_make_A.restype = C.c_int64
    L.irotavg_make_A.argtypes = [C.c_int, C.c_int, C.c_int64, _ip, _i64p, _i64p, _dp]
    L.irotavg_l1ra.argtypes = [C.c_int64, C.c_int64, C.c_int, _ip, _dp, C.c_int64, _dp, C.c_int64,
                               C.c_int, C.c_double, C.POINTER(C.c_int), _dp]
    L.irotavg_irls.argtypes = [C.c_int64, C.c_int64, C.c_int, _ip, _dp, C.c_int64, C.c_int,
                               C.c_double, _dp, C.c_int64, C.c_int, C.c_double, _dp,
                               C.POINTER(C.c_int), _dp]
    L.irotavg_quat_normalised.argtypes = [C.c_int64, _dp, C.c_int64, C.c_int]
    L.irotavg_graph_create.argtypes = [C.POINTER(vp), C.c_int64, C.c_int64, C.c_int, _ip, _dp,
                                       C.c_int64, C.POINTER(Options)]
    L.irotavg_graph_destroy.argtypes = [vp]
    L.irotavg_graph_destroy.restype = None
    L.irotavg_graph_set_rotations.argtypes = [vp, _dp, C.c_int64]
    L.irotavg_graph_get_rotations.argtypes = [vp, _dp, C.c_int64]
    L.irotavg_graph_snapshot_rotations.argtypes = [vp]
    L.irotavg_graph_restore_rotations.argtypes = [vp]
    L.irotavg_graph_get_weights.argtypes = [vp, _dp]
    L.irotavg_graph_set_weights.argtypes = [vp, _dp]
    L.irotavg_graph_irls.argtypes = [vp, C.c_int, C.c_double, C.c_int, C.c_double,
                                     C.POINTER(C.c_int), _dp, _dp]
    L.irotavg_graph_l1ra.argtypes = [vp, C.c_int, C.c_double, C.POINTER(C.c_int), _dp, _dp]
    L.irotavg_graph_quat_normalised.argtypes = [vp]
    L.irotavg_graph_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.irotavg_graph_reset_stats.argtypes = [vp]
    L.irotavg_graph_reset_stats.restype = None
    L.irotavg_graph_synchronize.argtypes = [vp]
    L.irotavg_graph_edge_residual.argtypes = [vp]
    L.irotavg_graph_get_residuals.argtypes = [vp, _dp, C.c_int64]
    L.irotavg_graph_ls_solve.argtypes = [vp, _dp, C.c_int64]
    L.irotavg_graph_update_weights.argtypes = [vp, C.c_int, C.c_double]
    L.irotavg_graph_apply_step.argtypes = [vp, _dp]
    L.irotavg_graph_l1decode_pd.argtypes = [vp, _dp, C.c_int, _dp, C.POINTER(C.c_int)]
    L.irotavg_graph_time_kernel.argtypes = [vp, C.c_int, C.c_int, _dp]
    L.irotavg_graph_fingerprint.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int]
    L.irotavg_graph_direct_info.argtypes = [vp, _i64p, C.c_int]
    L.irotavg_graph_direct_residual.argtypes = [vp, _dp]
    L.irotavg_viewgraph_create.argtypes = [C.POINTER(vp), C.POINTER(Options)]
    L.irotavg_viewgraph_destroy.argtypes = [vp]
    L.irotavg_viewgraph_destroy.restype = None
    L.irotavg_viewgraph_add_view.argtypes = [vp, _dp]
    L.irotavg_viewgraph_num_views.argtypes = [vp]
    L.irotavg_viewgraph_connect.argtypes = [vp, C.c_int, C.c_int, _dp]
    L.irotavg_viewgraph_fix_pose.argtypes = [vp, C.c_int, _dp]
    L.irotavg_viewgraph_is_pose_fixed.argtypes = [vp, C.c_int]
    L.irotavg_viewgraph_count_fixed_poses.argtypes = [vp]
    L.irotavg_viewgraph_get_pose.argtypes = [vp, C.c_int, _dp]
    L.irotavg_viewgraph_set_pose.argtypes = [vp, C.c_int, _dp]
    L.irotavg_viewgraph_rot_avg.argtypes = [vp, C.c_int, C.POINTER(RotAvgInfo)]
    L.irotavg_window_solve.argtypes = [C.c_int64, C.c_int64, C.c_int, _ip, _dp, C.c_int64, _dp, C.c_int64,
                                       C.c_int, C.c_double, C.c_int, C.c_int, C.c_double, _dp,
                                       C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.irotavg_window_solve_kernel.argtypes = [C.c_int64, C.c_int64, C.c_int, _ip, _dp, C.c_int64, _dp, C.c_int64,
                                       C.c_int, C.c_double, C.c_int, C.c_int, C.c_double, _dp,
                                       C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    L.irotavg_trim_memory.argtypes = []
    L.irotavg_trim_memory.restype = C.c_int64
    L.irotavg_oneshot_cache.argtypes = [C.c_int]
    L.irotavg_oneshot_cache.restype = None
    L.irotavg_oneshot_cache_clear.argtypes = []
    L.irotavg_oneshot_cache_clear.restype = None
    L.irotavg_oneshot_cache_stats.argtypes = [_i64p, _i64p]
    L.irotavg_oneshot_cache_stats.restype = None
    L.irotavg_rmat2quat.argtypes = [_dp, _dp]
    L.irotavg_rmat2quat.restype = None
    L.irotavg_quat2rmat.argtypes = [_dp, _dp]
    L.irotavg_quat2rmat.restype = None
    L.irotavg_viewgraph_save_poses.argtypes = [vp, C.c_char_p, _dp]
    L.irotavg_dist_unique_id.argtypes = [C.c_void_p]
    L.irotavg_dist_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_int64,
                                      C.c_int, _ip, _dp, C.c_int64, C.POINTER(Options)]
    L.irotavg_dist_destroy.argtypes = [vp]
    L.irotavg_dist_destroy.restype = None
    L.irotavg_dist_set_rotations.argtypes = [vp, _dp, C.c_int64]
    L.irotavg_dist_get_rotations.argtypes = [vp, _dp, C.c_int64]
    L.irotavg_dist_get_weights.argtypes = [vp, _dp]
    L.irotavg_dist_snapshot_rotations.argtypes = [vp]
    L.irotavg_dist_restore_rotations.argtypes = [vp]
    L.irotavg_dist_irls.argtypes = [vp, C.c_int, C.c_double, C.c_int, C.c_double, C.POINTER(C.c_int),
                                    _dp, _dp]
    L.irotavg_dist_l1ra.argtypes = [vp, C.c_int, C.c_double, C.POINTER(C.c_int), _dp, _dp]
    L.irotavg_dist_create_hosted.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.POINTER(Transport), C.c_int64,
                                             C.c_int64, C.c_int, _ip, _dp, C.c_int64, C.POINTER(Options)]
    L.irotavg_dist_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.irotavg_dist_info.argtypes = [vp, C.POINTER(C.c_int64)]
    L.irotavg_dist_timing.argtypes = [vp, C.c_int, _dp, C.POINTER(C.c_int64)]
    L.irotavg_dist_plan.argtypes = [vp, C.c_int, _i64p, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                    C.POINTER(C.c_int), C.c_int]
    L.irotavg_dist_plan_host.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int, _ip, _i64p,
                                         _ip, C.c_int64, _ip, C.c_int64, _ip, C.c_int64,
                                         C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                         C.c_int]
    _LIB = L
    return L


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def fmat(a):
    """float64 column-major copy: the reference's Eigen `Mat` layout."""
    return np.array(a, dtype=np.float64, order="F", copy=True)


def edges(I):
    I = np.ascontiguousarray(I, dtype=np.int32)
    if I.ndim != 2 or I.shape[1] != 2:
        raise ValueError("I must be (m, 2)")
    return I


def default_options(**kw):
    o = Options()
    lib().irotavg_default_options(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def check(rc, where):
    if rc != OK:
        raise IrotavgError(rc, where)


class Graph:
    """Device-resident view-graph handle (irotavg_graph_* of include/irotavg_hip.h)."""

    def __init__(self, I, QQ, n_total, f, **opts):
        I = edges(I)
        QQ = fmat(QQ)
        self.m, self.n_total, self.f = len(I), int(n_total), int(f)
        self.nu = self.n_total - self.f
        self._h = C.c_void_p()
        o = default_options(**opts)
        rc = lib().irotavg_graph_create(C.byref(self._h), self.m, self.n_total, self.f, _i(I),
                                        _d(QQ), QQ.shape[0], C.byref(o))
        if rc != OK:
            self._h = C.c_void_p()
            raise IrotavgError(rc, "irotavg_graph_create")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().irotavg_graph_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_rotations(self, Q):
        Q = fmat(Q)
        assert Q.shape == (self.n_total, 4)
        check(lib().irotavg_graph_set_rotations(self._h, _d(Q), Q.shape[0]), "set_rotations")

    def get_rotations(self):
        Q = np.zeros((self.n_total, 4), order="F")
        check(lib().irotavg_graph_get_rotations(self._h, _d(Q), Q.shape[0]), "get_rotations")
        return Q

    def snapshot_rotations(self):
        check(lib().irotavg_graph_snapshot_rotations(self._h), "snapshot_rotations")

    def restore_rotations(self):
        check(lib().irotavg_graph_restore_rotations(self._h), "restore_rotations")

    def get_weights(self):
        w = np.zeros(self.m)
        check(lib().irotavg_graph_get_weights(self._h, _d(w)), "get_weights")
        return w

    def set_weights(self, w):
        w = np.ascontiguousarray(w, dtype=np.float64)
        assert w.shape == (self.m,)
        check(lib().irotavg_graph_set_weights(self._h, _d(w)), "set_weights")

    def irls(self, cost=4, sigma=5 * np.pi / 180, max_iters=50, change_th=1e-3, allow_rc=()):
        iters = C.c_int(0)
        rt = C.c_double(0)
        trace = np.full(max(max_iters, 1), np.nan)
        rc = lib().irotavg_graph_irls(self._h, cost, sigma, max_iters, change_th, C.byref(iters),
                                      C.byref(rt), _d(trace))
        if rc != OK and rc not in allow_rc:
            raise IrotavgError(rc, "irotavg_graph_irls")
        return dict(rc=rc, iters=iters.value, runtime=rt.value, scores=trace[:iters.value].copy())

    def l1ra(self, max_iters=5, change_th=1e-3, allow_rc=()):
        iters = C.c_int(0)
        rt = C.c_double(0)
        trace = np.full(max(max_iters, 1), np.nan)
        rc = lib().irotavg_graph_l1ra(self._h, max_iters, change_th, C.byref(iters), C.byref(rt),
                                      _d(trace))
        if rc != OK and rc not in allow_rc:
            raise IrotavgError(rc, "irotavg_graph_l1ra")
        return dict(rc=rc, iters=iters.value, runtime=rt.value, scores=trace[:iters.value].copy())

    def quat_normalised(self):
        check(lib().irotavg_graph_quat_normalised(self._h), "quat_normalised")

    def stats(self):
        s = Stats()
        check(lib().irotavg_graph_get_stats(self._h, C.byref(s)), "get_stats")
        d = {k: getattr(s, k) for k, _ in Stats._fields_
             if k not in ("level_rows", "level_nnz", "last_relres")}
        d["level_rows"] = list(s.level_rows)[:s.levels]
        d["level_nnz"] = list(s.level_nnz)[:s.levels]
        d["last_relres"] = list(s.last_relres)
        return d

    def reset_stats(self):
        lib().irotavg_graph_reset_stats(self._h)

    def synchronize(self):
        check(lib().irotavg_graph_synchronize(self._h), "synchronize")

    # stage-level entry points
    def edge_residual(self):
        check(lib().irotavg_graph_edge_residual(self._h), "edge_residual")

    def get_residuals(self):
        r = np.zeros((self.m, 3), order="F")
        check(lib().irotavg_graph_get_residuals(self._h, _d(r), self.m), "get_residuals")
        return r

    def ls_solve(self, allow_rc=()):
        X = np.zeros((self.nu, 3), order="F")
        rc = lib().irotavg_graph_ls_solve(self._h, _d(X), self.nu)
        if rc != OK and rc not in allow_rc:
            raise IrotavgError(rc, "irotavg_graph_ls_solve")
        return X

    def update_weights(self, cost, sigma):
        check(lib().irotavg_graph_update_weights(self._h, cost, sigma), "update_weights")

    def apply_step(self):
        s = C.c_double(0)
        check(lib().irotavg_graph_apply_step(self._h, C.byref(s)), "apply_step")
        return s.value

    def l1decode_pd(self, y, pdmaxiter=2):
        y = np.ascontiguousarray(y, dtype=np.float64)
        assert y.shape == (self.m,)
        x = np.zeros(self.nu)
        stuck = C.c_int(0)
        check(lib().irotavg_graph_l1decode_pd(self._h, _d(y), pdmaxiter, _d(x), C.byref(stuck)),
              "l1decode_pd")
        return x, stuck.value

    def direct_info(self):
        """irotavg_graph_direct_info: dict(block, levels=[dict(blocks, chunks, reduced)]) of the banded direct solver."""
        out = (C.c_int64 * 64)()
        k = lib().irotavg_graph_direct_info(self._h, out, 64)
        if k < 0:
            raise IrotavgError(k, "direct_info")
        if out[0] == 0:
            return dict(block=0, levels=[], closures=0)
        nl = int(out[1])
        return dict(block=int(out[0]), levels=[dict(blocks=int(out[2 + 3 * l]), chunks=int(out[3 + 3 * l]),
                                                    reduced=int(out[4 + 3 * l])) for l in range(nl)],
                    closures=int(out[2 + 3 * nl]))

    def direct_residual(self):
        """irotavg_graph_direct_residual: ||b - A x|| / ||b|| per coordinate of the most recent direct solve."""
        out = np.zeros(3)
        check(lib().irotavg_graph_direct_residual(self._h, _d(out)), "direct_residual")
        return out

    def fingerprint(self):
        """Hashes of every structural array + the kernel-choosing scalars (irotavg_graph_fingerprint)."""
        out = (C.c_uint64 * 512)()
        n = lib().irotavg_graph_fingerprint(self._h, out, 512)
        if n < 0:
            raise IrotavgError(n, "fingerprint")
        return [int(out[i]) for i in range(n)]

    def time_kernel(self, which, reps=20):
        ms = C.c_double(0)
        check(lib().irotavg_graph_time_kernel(self._h, which, reps, C.byref(ms)), "time_kernel")
        return ms.value


def oneshot_cache(enable):
    """irotavg_oneshot_cache: the one-shot calls keep the handle of their last call (on) or build one per call (off)."""
    lib().irotavg_oneshot_cache(1 if enable else 0)


def oneshot_cache_clear():
    lib().irotavg_oneshot_cache_clear()


def oneshot_cache_stats():
    """(hits, misses) of the kept handle since the library was loaded."""
    h, m = C.c_int64(0), C.c_int64(0)
    lib().irotavg_oneshot_cache_stats(C.byref(h), C.byref(m))
    return int(h.value), int(m.value)


def trim_memory():
    """irotavg_trim_memory: release the cached device buffers of destroyed handles; bytes freed."""
    return int(lib().irotavg_trim_memory())


def window_solve(I, QQ, Q, f, cost=4, sigma=5 * np.pi / 180, l1_iters=100, irls_iters=100,
                 change_th=1e-3, kernel=0):
    """irotavg_window_solve[_kernel]: l1ra + irls of a small problem in one kernel launch
    (kernel: 0 automatic, 1 general LDS kernel, 2 wave-resident kernel)."""
    I = edges(I)
    QQ = fmat(QQ)
    Q = fmat(Q)
    w = np.zeros(len(I))
    a, b = C.c_int(0), C.c_int(0)
    rc = lib().irotavg_window_solve_kernel(len(I), Q.shape[0], f, _i(I), _d(QQ), QQ.shape[0], _d(Q),
                                           Q.shape[0], cost, sigma, l1_iters, irls_iters, change_th,
                                           _d(w), C.byref(a), C.byref(b), kernel)
    check(rc, "irotavg_window_solve")
    return dict(Q=Q, weights=w, l1_iters=a.value, irls_iters=b.value)


def plan_host(world, rank, I, n_total, f):
    """Host-only partition plan of one rank (irotavg_dist_plan_host): dict with the owned range,
    ghost ids, per-peer send ids (GLOBAL view ids), local edge ids."""
    I = edges(I)
    m = len(I)
    counts = (C.c_int64 * 6)()
    cap = 2 * m + 8
    ghosts = np.zeros(cap, dtype=np.int32)
    send = np.zeros(cap, dtype=np.int32)
    led = np.zeros(m + 8, dtype=np.int32)
    pc = max(world, 1)
    peers = (C.c_int * pc)()
    sc = (C.c_int * pc)()
    rc_ = (C.c_int * pc)()
    rc = lib().irotavg_dist_plan_host(world, rank, m, n_total, f, _i(I), counts, _i(ghosts), cap, _i(send),
                                      cap, _i(led), m + 8, peers, sc, rc_, pc)
    check(rc, "dist_plan_host")
    lo, hi, ng, ml, npeers, ns = [int(x) for x in counts]
    out = dict(lo=lo, hi=hi, ghosts=ghosts[:ng].copy(), edges=led[:ml].copy(), peers=list(peers)[:npeers],
               send_cnt=list(sc)[:npeers], recv_cnt=list(rc_)[:npeers])
    off, per = 0, {}
    for q, h in enumerate(out["peers"]):
        per[h] = send[off:off + out["send_cnt"][q]].copy()
        off += out["send_cnt"][q]
    out["send"] = per
    return out


_AR_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int)
_I64P = C.POINTER(C.c_int64)
_EX_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), _I64P, _I64P,
                     C.POINTER(C.c_double), _I64P, _I64P)


class Transport(C.Structure):
    """irotavg_transport (include/irotavg_hip.h): the host-staged wire of the sharded solve."""
    _fields_ = [("ctx", C.c_void_p), ("allreduce", _AR_FN), ("exchange", _EX_FN)]


def torch_transport(group=None):
    """An irotavg_transport over torch.distributed (any backend that moves CPU tensors, e.g. gloo).
    Returns (Transport, keepalive): keep `keepalive` referenced for the life of the handle."""
    import torch
    import torch.distributed as dist
    ops = {0: dist.ReduceOp.SUM, 1: dist.ReduceOp.MIN, 2: dist.ReduceOp.MAX}

    def allreduce(_ctx, buf, n, op):
        try:
            a = np.ctypeslib.as_array(buf, shape=(n,))
            t = torch.from_numpy(a)
            dist.all_reduce(t, op=ops[op], group=group)
            return 0
        except Exception as e:  # never raise through the C frames
            print("[irotavg transport] allreduce failed:", e, flush=True)
            return 1

    def exchange(_ctx, npeers, peers, send, so, sc, recv, ro, rc):
        try:
            reqs, keep = [], []
            me = dist.get_rank(group)
            order = sorted(range(npeers), key=lambda q: peers[q])
            for q in order:
                h = peers[q]
                srt = torch.from_numpy(np.ctypeslib.as_array(send, shape=(so[q] + sc[q],))[so[q]:].copy()) \
                    if sc[q] > 0 else None
                rcv = torch.empty(rc[q], dtype=torch.float64) if rc[q] > 0 else None
                keep.append((q, rcv))
                # deadlock-free pairing: the lower rank sends first
                steps = [("s", srt), ("r", rcv)] if me < h else [("r", rcv), ("s", srt)]
                for kind, t in steps:
                    if t is None:
                        continue
                    if kind == "s":
                        dist.send(t, dst=h, group=group)
                    else:
                        dist.recv(t, src=h, group=group)
            for q, rcv in keep:
                if rcv is not None:
                    np.ctypeslib.as_array(recv, shape=(ro[q] + rc[q],))[ro[q]:] = rcv.numpy()
            return 0
        except Exception as e:
            print("[irotavg transport] exchange failed:", e, flush=True)
            return 1

    ar, ex = _AR_FN(allreduce), _EX_FN(exchange)
    return Transport(None, ar, ex), (ar, ex)


class DistGraph:
    """Sharded IRLS (irotavg_dist_*). unique_id=None: all `world` shards in this process on one GPU
    (loopback transport); otherwise this process holds shard `rank` and talks RCCL."""

    def __init__(self, I, QQ, n_total, f, world, rank=0, unique_id=None, transport=None, **opts):
        """transport: (Transport, keepalive) from torch_transport(): host-staged wire instead of RCCL."""
        I = edges(I)
        QQ = fmat(QQ)
        self.m, self.n_total, self.f, self.world, self.rank = len(I), int(n_total), int(f), world, rank
        self._h = C.c_void_p()
        o = default_options(**opts)
        uid = None
        if unique_id is not None:
            uid = (C.c_char * 128).from_buffer_copy(bytes(unique_id))
        if transport is not None:
            self._transport = transport
            rc = lib().irotavg_dist_create_hosted(C.byref(self._h), world, rank, C.byref(transport[0]), self.m,
                                                  self.n_total, self.f, _i(I), _d(QQ), QQ.shape[0], C.byref(o))
        else:
            rc = lib().irotavg_dist_create(C.byref(self._h), world, rank, uid, self.m, self.n_total, self.f,
                                           _i(I), _d(QQ), QQ.shape[0], C.byref(o))
        if rc != OK:
            self._h = C.c_void_p()
            raise IrotavgError(rc, "irotavg_dist_create")

    @staticmethod
    def unique_id():
        buf = (C.c_char * 128)()
        check(lib().irotavg_dist_unique_id(buf), "dist_unique_id")
        return bytes(buf)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().irotavg_dist_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_rotations(self, Q):
        Q = fmat(Q)
        check(lib().irotavg_dist_set_rotations(self._h, _d(Q), Q.shape[0]), "dist_set_rotations")

    def get_rotations(self, into=None):
        Q = np.zeros((self.n_total, 4), order="F") if into is None else fmat(into)
        check(lib().irotavg_dist_get_rotations(self._h, _d(Q), Q.shape[0]), "dist_get_rotations")
        return Q

    def snapshot_rotations(self):
        check(lib().irotavg_dist_snapshot_rotations(self._h), "dist_snapshot_rotations")

    def restore_rotations(self):
        check(lib().irotavg_dist_restore_rotations(self._h), "dist_restore_rotations")

    def get_weights(self):
        w = np.full(self.m, np.nan)
        check(lib().irotavg_dist_get_weights(self._h, _d(w)), "dist_get_weights")
        return w

    def irls(self, cost=4, sigma=5 * np.pi / 180, max_iters=50, change_th=1e-3, allow_rc=()):
        iters = C.c_int(0)
        rt = C.c_double(0)
        trace = np.full(max(max_iters, 1), np.nan)
        rc = lib().irotavg_dist_irls(self._h, cost, sigma, max_iters, change_th, C.byref(iters),
                                     C.byref(rt), _d(trace))
        if rc != OK and rc not in allow_rc:
            raise IrotavgError(rc, "irotavg_dist_irls")
        return dict(rc=rc, iters=iters.value, runtime=rt.value, scores=trace[:iters.value].copy())

    def l1ra(self, max_iters=5, change_th=1e-3, allow_rc=()):
        iters = C.c_int(0)
        rt = C.c_double(0)
        trace = np.full(max(max_iters, 1), np.nan)
        rc = lib().irotavg_dist_l1ra(self._h, max_iters, change_th, C.byref(iters), C.byref(rt), _d(trace))
        if rc != OK and rc not in allow_rc:
            raise IrotavgError(rc, "irotavg_dist_l1ra")
        return dict(rc=rc, iters=iters.value, runtime=rt.value, scores=trace[:iters.value].copy())

    def info(self):
        """wire / RCCL communicator size / local shards / world / ghost views / block size of the sharded direct solver
        (0: the sharded PCG) (irotavg_dist_info)."""
        v = (C.c_int64 * 8)()
        check(lib().irotavg_dist_info(self._h, v), "dist_info")
        return dict(wire=["loopback", "rccl", "host-staged"][v[0] & 15],
                    halo="all-gather of boundary records" if v[0] & 16 else "point-to-point", rccl_comm_ranks=int(v[1]),
                    local_shards=int(v[2]), world=int(v[3]), ghost_views=int(v[4]), peers=int(v[5]),
                    direct_block=int(v[6]), closures=int(v[7]))

    TIMING_PHASES = ["local_edge_kernels_and_assembly", "local_reductions", "gather_of_separators", "closure_sum",
                     "separator_system_and_ways_back", "halo_of_the_step", "weights_and_rotation_update",
                     "score_allreduce"]

    def timing(self, enable):
        """irotavg_dist_timing: the per-phase means (us per IRLS iteration) collected so far, then the phase clock is
        switched on (cleared) or off. A diagnostic: calls made under it are slower and must not be timed."""
        v = (C.c_double * 8)()
        n = C.c_int64(0)
        check(lib().irotavg_dist_timing(self._h, 1 if enable else 0, v, C.byref(n)), "dist_timing")
        return dict(iterations=int(n.value), us_per_iteration={k: float(v[i]) for i, k in enumerate(self.TIMING_PHASES)})

    def stats(self):
        s = Stats()
        check(lib().irotavg_dist_get_stats(self._h, C.byref(s)), "dist_get_stats")
        return dict(pcg_solves=s.pcg_solves, pcg_iters=s.pcg_iters, pcg_iters_last=s.pcg_iters_last,
                    outer_iters=s.outer_iters, edge_updates=s.edge_updates, seconds_irls=s.seconds_irls,
                    levels=s.levels, level_rows=list(s.level_rows)[:s.levels],
                    pcg_handed_over=s.pcg_handed_over, direct_solves=s.direct_solves,
                    direct_guarded=s.direct_guarded, direct_dead_pivots=s.direct_dead_pivots,
                    last_relres=list(s.last_relres))

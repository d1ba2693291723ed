"""ctypes binding of the CPU ORACLE (oracle/liboracle.so) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
PARITY UNPINNED: see oracle/irotavg_oracle.h.

Array conventions mirror the reference API (ral/l1_irls.hpp:40-51): Q and QQ are (rows, 4)
float64 arrays in *Fortran* (column-major) order with columns [x, y, z, w]; I is an (m, 2)
int32 C-order array of (i, j) pairs.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

COSTS = ["L2", "L1", "L1.5", "L0.5", "Geman-McClure", "Huber", "Pseudo-Huber", "Andrews",
         "Bisquare", "Cauchy", "Fair", "Logistic", "Talwar", "Welsch"]  # ral/l1_irls.hpp:56-79

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_lp = C.POINTER(C.c_long)


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in
            ("ral_oracle.c", "sparse_chol.c", "sparse_pcg.c", "irotavg_oracle.h", "sparse_chol.h",
             "sparse_pcg.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so)
                                               for s in srcs if os.path.exists(s)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        L = _LIB
        L.ora_irls.restype = C.c_int
        L.ora_irls.argtypes = [C.c_long, C.c_long, C.c_int, _ip, _dp, C.c_long, _dp, C.c_long,
                               C.c_int, C.c_double, C.c_int, C.c_double, _dp, _ip, _dp, _dp]
        L.ora_l1ra.restype = C.c_int
        L.ora_l1ra.argtypes = [C.c_long, C.c_long, C.c_int, _ip, _dp, C.c_long, _dp, C.c_long,
                               C.c_int, C.c_double, _ip, _dp, _dp]
        L.ora_init_mst.restype = C.c_int
        L.ora_init_mst.argtypes = [C.c_long, C.c_long, _dp, C.c_long, _dp, C.c_long, _ip, C.c_int]
        L.ora_quat_normalised.restype = None
        L.ora_quat_normalised.argtypes = [C.c_long, _dp, C.c_long, C.c_int]
        L.ora_delta_rel.restype = None
        L.ora_delta_rel.argtypes = [C.c_long, _ip, _dp, C.c_long, _dp, C.c_long, _dp, C.c_long]
        L.ora_log_map.restype = None
        L.ora_log_map.argtypes = [C.c_long, _dp, C.c_long]
        L.ora_exp_map.restype = None
        L.ora_exp_map.argtypes = [C.c_long, _dp, C.c_long]
        L.ora_quat_mult.restype = None
        L.ora_quat_mult.argtypes = [_dp, _dp, _dp]
        L.ora_make_A.restype = C.c_long
        L.ora_make_A.argtypes = [C.c_int, C.c_int, C.c_long, _ip, _lp, _lp, _dp]
        L.ora_l1decode_pd.restype = C.c_int
        L.ora_l1decode_pd.argtypes = [C.c_long, C.c_long, C.c_int, _ip, _dp, C.c_int, _dp, _ip]
        L.ora_ls_solve.restype = C.c_int
        L.ora_ls_solve.argtypes = [C.c_long, C.c_long, C.c_int, _ip, _dp, _dp, C.c_long, _dp]
        L.ora_normal_matvec.restype = None
        L.ora_normal_matvec.argtypes = [C.c_long, C.c_long, C.c_int, _ip, _dp, _dp, _dp]
        L.ora_rmat2quat.restype = None
        L.ora_rmat2quat.argtypes = [_dp, _dp]
        L.ora_quat2rmat.restype = None
        L.ora_quat2rmat.argtypes = [_dp, _dp]
        L.ora_set_solver.restype = None
        L.ora_set_solver.argtypes = [C.c_int]
        L.ora_solver_stats.restype = None
        L.ora_solver_stats.argtypes = [_lp, _lp, _lp, _dp, C.c_int]
    return _LIB


SOLVER_AUTO, SOLVER_CHOLESKY, SOLVER_PCG = 0, 1, 2


def set_solver(mode):
    """Which stand-in for SuiteSparse the oracle's solves use (irotavg_oracle.h: ora_set_solver)."""
    lib().ora_set_solver(int(mode))


def solver_stats(reset=False):
    a, b, c = C.c_long(0), C.c_long(0), C.c_long(0)
    d = C.c_double(0)
    lib().ora_solver_stats(C.byref(a), C.byref(b), C.byref(c), C.byref(d), 1 if reset else 0)
    return dict(chol_solves=a.value, pcg_solves=b.value, pcg_iters=c.value, pcg_worst_relres=d.value)


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def fmat(a):
    """float64 column-major copy (Eigen Mat layout)."""
    return np.array(a, dtype=np.float64, order="F", copy=True)


def edges(I):
    I = np.ascontiguousarray(I, dtype=np.int32)
    assert I.ndim == 2 and I.shape[1] == 2
    return I


def quat_mult(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    o = np.zeros(4)
    lib().ora_quat_mult(_d(a), _d(b), _d(o))
    return o


def delta_rel(I, QQ, Q):
    I = edges(I); QQ = fmat(QQ); Q = fmat(Q)
    m = len(I)
    out = np.zeros((m, 4), order="F")
    lib().ora_delta_rel(m, _i(I), _d(QQ), QQ.shape[0], _d(Q), Q.shape[0], _d(out), m)
    return out


def log_map(w):
    w = fmat(w)
    lib().ora_log_map(w.shape[0], _d(w), w.shape[0])
    return w


def exp_map(W):
    W = fmat(W)
    lib().ora_exp_map(W.shape[0], _d(W), W.shape[0])
    return W


def make_A(n, f, I):
    """Returns scipy CSC matrix of ral/l1_irls.cpp:755-780 (m x (n-f))."""
    import scipy.sparse as sp
    I = edges(I)
    m = len(I)
    nu = n - f
    colptr = np.zeros(nu + 1, dtype=np.int64)
    rowidx = np.zeros(2 * m + 1, dtype=np.int64)
    vals = np.zeros(2 * m + 1)
    nnz = lib().ora_make_A(n, f, m, _i(I), colptr.ctypes.data_as(_lp), rowidx.ctypes.data_as(_lp),
                           _d(vals))
    if nnz < 0:
        raise ValueError("ora_make_A error %d" % nnz)
    return sp.csc_matrix((vals[:nnz], rowidx[:nnz], colptr), shape=(m, nu))


def init_mst(Q, QQ, I, f):
    Q = fmat(Q); QQ = fmat(QQ); I = edges(I)
    rc = lib().ora_init_mst(Q.shape[0], len(I), _d(Q), Q.shape[0], _d(QQ), QQ.shape[0], _i(I), f)
    return rc, Q


def quat_normalised(Q, f):
    Q = fmat(Q)
    lib().ora_quat_normalised(Q.shape[0], _d(Q), Q.shape[0], f)
    return Q


def irls(QQ, I, Q, f, cost=4, sigma=5 * np.pi / 180, max_iters=50, change_th=1e-3):
    """ral/l1_irls.cpp:559-752. Returns dict(rc, Q, weights, iters, runtime, scores)."""
    QQ = fmat(QQ); Q = fmat(Q); I = edges(I)
    m = len(I)
    weights = np.zeros(m)
    iters = C.c_int(0)
    rt = C.c_double(0)
    trace = np.full(max(max_iters, 1), np.nan)
    rc = lib().ora_irls(m, Q.shape[0], f, _i(I), _d(QQ), QQ.shape[0], _d(Q), Q.shape[0], cost,
                        sigma, max_iters, change_th, _d(weights), C.byref(iters), C.byref(rt),
                        _d(trace))
    return dict(rc=rc, Q=Q, weights=weights, iters=iters.value, runtime=rt.value,
                scores=trace[:iters.value].copy())


def l1ra(QQ, I, Q, f, max_iters=5, change_th=1e-3):
    """ral/l1_irls.cpp:851-912."""
    QQ = fmat(QQ); Q = fmat(Q); I = edges(I)
    m = len(I)
    iters = C.c_int(0)
    rt = C.c_double(0)
    trace = np.full(max(max_iters, 1), np.nan)
    rc = lib().ora_l1ra(m, Q.shape[0], f, _i(I), _d(QQ), QQ.shape[0], _d(Q), Q.shape[0],
                        max_iters, change_th, C.byref(iters), C.byref(rt), _d(trace))
    return dict(rc=rc, Q=Q, iters=iters.value, runtime=rt.value, scores=trace[:iters.value].copy())


def l1decode_pd(n_total, f, I, y, pdmaxiter=2):
    I = edges(I)
    y = np.ascontiguousarray(y, dtype=np.float64)
    x = np.zeros(n_total - f)
    stuck = C.c_int(0)
    rc = lib().ora_l1decode_pd(len(I), n_total, f, _i(I), _d(y), pdmaxiter, _d(x), C.byref(stuck))
    return rc, x, stuck.value


def ls_solve(n_total, f, I, weights, w):
    """X (n_u x 3) of the weighted LS at ral/l1_irls.cpp:596-612."""
    I = edges(I)
    weights = np.ascontiguousarray(weights, dtype=np.float64)
    w = fmat(w)
    X = np.zeros((n_total - f, 3), order="F")
    rc = lib().ora_ls_solve(len(I), n_total, f, _i(I), _d(weights), _d(w), w.shape[0], _d(X))
    return rc, X


def normal_matvec(n_total, f, I, weights, X):
    I = edges(I)
    weights = np.ascontiguousarray(weights, dtype=np.float64)
    X = fmat(X)
    Y = np.zeros_like(X, order="F")
    lib().ora_normal_matvec(len(I), n_total, f, _i(I), _d(weights), _d(X), _d(Y))
    return Y


def rmat2quat(R):
    R = np.ascontiguousarray(R, dtype=np.float64).reshape(9)
    q = np.zeros(4)
    lib().ora_rmat2quat(_d(R), _d(q))
    return q


def quat2rmat(q):
    q = np.ascontiguousarray(q, dtype=np.float64)
    R = np.zeros(9)
    lib().ora_quat2rmat(_d(q), _d(R))
    return R.reshape(3, 3)

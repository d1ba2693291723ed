"""Python mirror of the reference's RAL free-function API (ral/l1_irls.hpp:81-112) on top of the
C ABI -- same names, argument meaning and mutation-in-place behaviour, so tests and drivers read
like ral/test.cpp:285-302. Q and QQ are (rows, 4) float64 arrays, columns [x y z w]; I is (m, 2)
int32. Errors the reference turns into exit(-1) are raised as capi.IrotavgError.
"""
import ctypes as C
import warnings

import numpy as np

from . import capi

# ral/l1_irls.hpp:56-57
L2, L1, L15, L05, Geman_McClure, Huber, Pseudo_Huber, Andrews, Bisquare, Cauchy, Fair, Logistic, \
    Talwar, Welsch = range(14)
COST_NAMES = ["L2", "L1", "L1.5", "L0.5", "Geman-McClure", "Huber", "Pseudo-Huber", "Andrews",
              "Bisquare", "Cauchy", "Fair", "Logistic", "Talwar", "Welsch"]  # operator<< :59-79


def parse_cost(name):
    """ral/test.cpp:35-72 (case-insensitive)."""
    low = [c.lower() for c in COST_NAMES]
    if name is None or name.lower() not in low:
        raise ValueError("Unknown string. %s" % name)
    return low.index(name.lower())


def init_mst(Q, QQ, I, f):
    """ral/l1_irls.hpp:89. Q is updated in place (rows >= f)."""
    Qf, QQf, Ie = capi.fmat(Q), capi.fmat(QQ), capi.edges(I)
    rc = capi.lib().irotavg_init_mst(Qf.shape[0], len(Ie), capi._d(Qf), Qf.shape[0], capi._d(QQf),
                                     QQf.shape[0], capi._i(Ie), f)
    capi.check(rc, "init_mst")
    Q[...] = Qf
    return Q


def make_A(n, f, I):
    """ral/l1_irls.hpp:91. Returns (colptr, rowidx, vals) CSC arrays of the m x (n-f) matrix."""
    Ie = capi.edges(I)
    m = len(Ie)
    colptr = np.zeros(n - f + 1, dtype=np.int64)
    rowidx = np.zeros(2 * m + 1, dtype=np.int64)
    vals = np.zeros(2 * m + 1)
    p64 = C.POINTER(C.c_int64)
    nnz = capi.lib().irotavg_make_A(n, f, m, capi._i(Ie), colptr.ctypes.data_as(p64),
                                    rowidx.ctypes.data_as(p64), capi._d(vals))
    if nnz < 0:
        raise capi.IrotavgError(int(nnz), "make_A")
    return colptr, rowidx[:nnz].copy(), vals[:nnz].copy()


def _check_soft(rc, where):
    """ERR_NOT_CONVERGED (inner PCG at its iteration cap) is a warning, like the reference's
    " Max Iteration" message (ral/l1_irls.cpp:746-749): the C call has written Q / weights back and
    the reference's direct solvers always return a result. Everything else raises."""
    if rc == capi.ERR_NOT_CONVERGED:
        warnings.warn("%s: inner PCG did not converge within pcg_max_iters (result kept)" % where,
                      RuntimeWarning)
        return
    capi.check(rc, where)


def l1ra(QQ, I, A, Q, f, max_iters, change_th):
    """ral/l1_irls.hpp:100-102. `A` is accepted for signature parity and ignored (derivable from
    n, f, I). Q updated in place. Returns (iter, runtime)."""
    Qf, QQf, Ie = capi.fmat(Q), capi.fmat(QQ), capi.edges(I)
    it, rt = C.c_int(0), C.c_double(0)
    rc = capi.lib().irotavg_l1ra(len(Ie), Qf.shape[0], f, capi._i(Ie), capi._d(QQf), QQf.shape[0],
                                 capi._d(Qf), Qf.shape[0], max_iters, change_th, C.byref(it),
                                 C.byref(rt))
    _check_soft(rc, "l1ra")
    Q[...] = Qf
    return it.value, rt.value


def irls(QQ, I, A, cost, sigma, Q, f, max_iters, change_th, weights):
    """ral/l1_irls.hpp:104-107. Q and weights (length m) updated in place. Returns (iters, runtime)."""
    Qf, QQf, Ie = capi.fmat(Q), capi.fmat(QQ), capi.edges(I)
    if weights.shape != (len(Ie),):
        raise ValueError("weights must be pre-sized to m (ral/test.cpp:299)")
    w = np.zeros(len(Ie))
    it, rt = C.c_int(0), C.c_double(0)
    rc = capi.lib().irotavg_irls(len(Ie), Qf.shape[0], f, capi._i(Ie), capi._d(QQf), QQf.shape[0],
                                 int(cost), float(sigma), capi._d(Qf), Qf.shape[0], max_iters,
                                 change_th, capi._d(w), C.byref(it), C.byref(rt))
    _check_soft(rc, "irls")
    Q[...] = Qf
    weights[...] = w
    return it.value, rt.value


def quat_normalised(Q, f):
    """ral/l1_irls.hpp:112."""
    Qf = capi.fmat(Q)
    capi.check(capi.lib().irotavg_quat_normalised(Qf.shape[0], capi._d(Qf), Qf.shape[0], f),
               "quat_normalised")
    Q[...] = Qf
    return Q

"""What pins the oracle's linear solves against the reference's FORMULATION.

The reference solves the weighted least-squares problem min ||D A X - D w|| in its least-squares
form by sparse QR (SuiteSparseQR, ral/l1_irls.cpp:536-556, 596-612) and the primal-dual system by LU
(UMFPACK, :131-184); the oracle (and its NumPy twin, and the GPU path) solve the NORMAL equations
A'D^2A X = A'D^2 w. Neither SuiteSparse nor the reference can be built here, so the LS form is
checked against dense orthogonal factorisations of the same D A (NumPy's LAPACK: Householder QR and
the SVD-based minimum-norm lstsq):

* full column rank (connected graph, f >= 1, positive weights): the LS solution is unique, so QR,
  minimum-norm lstsq and the oracle must agree to round-off -- fixture and a seeded synthetic graph;
* rank-deficient cases the reference can produce (Talwar's exact zero weights, ral/l1_irls.cpp:712;
  rows emptied by make_A's edge-drop quirk, :770-771): the solution set is an affine space and the
  reference returns SPQR's *basic* solution, which depends on SPQR's column ordering (COLAMD) and is
  therefore not reproducible without SuiteSparse. What IS solver-independent and is asserted here:
  every least-squares solution has the same A X (hence the same residuals, robust weights and edge
  errors, :614-727); an isolated view gets X = 0 from the oracle's dead-pivot rule, from a basic
  solution and from the minimum-norm solution alike; on a floating component solutions differ by one
  constant row per component -- the oracle pins one view of it to 0 (as a basic solution does, for
  SOME view), minimum-norm centres it. The per-view step, the `score` (:729) and hence the
  iteration count of such an ill-posed input are NOT pinned; DESIGN.md says so.
"""
import numpy as np
import pytest

from irotavg_amd import synth
from oracle import oracle as O

SIG = 5 * np.pi / 180


def dense_DA(n_total, f, I, weights):
    A = O.make_A(n_total, f, I).toarray()                 # m x n_u, incl. the edge-drop quirk
    return weights[:, None] * A, A


def residual_rows(I, QQ, Q):
    return O.log_map(O.delta_rel(I, QQ, Q))[:, :3]        # w(:, :3) of ral/l1_irls.cpp:592-594


def qr_solve(M, B):
    Qm, R = np.linalg.qr(M)                               # Householder QR (LAPACK dgeqrf)
    return np.linalg.solve(R, Qm.T @ B)


@pytest.mark.parametrize("which", ["fixture", "synth"])
def test_ls_form_agrees_with_normal_equations_when_full_rank(which, fixture_graph):
    if which == "fixture":
        g = fixture_graph
        n, f, I, QQ = g["n"], g["f"], g["I"], g["QQ"]
        rc, Q = O.init_mst(g["Q"], QQ, I, max(g["n_abs_read"], f))
        assert rc == 0
    else:
        S = synth.make_graph(400, 3000, 0.2, seed=5)
        n, f, I, QQ = 400, 1, S["I"], S["QQ"]
        Q = np.zeros((n, 4)); Q[:, 3] = 1; Q[0] = S["Qgt"][0]
        rc, Q = O.init_mst(Q, QQ, I, f)
    w3 = residual_rows(I, QQ, Q)
    rng = np.random.default_rng(1)
    for weights in (np.ones(len(I)), np.exp(rng.normal(scale=1.5, size=len(I)))):   # IRLS iteration 1 / spread weights
        DA, _ = dense_DA(n, f, I, weights)
        DB = weights[:, None] * w3
        rc, X = O.ls_solve(n, f, I, weights, w3)
        assert rc == 0
        Xqr = qr_solve(DA, DB)
        Xmn = np.linalg.lstsq(DA, DB, rcond=None)[0]
        scale = np.abs(Xqr).max()
        assert np.abs(X - Xqr).max() <= 1e-9 * scale
        assert np.abs(X - Xmn).max() <= 1e-9 * scale
        # and the LS optimality condition itself, evaluated in the LS form: (DA)'(DA X - DB) = 0
        g_ = DA.T @ (DA @ X - DB)
        assert np.abs(g_).max() <= 1e-9 * np.abs(DA.T @ DB).max()


def _components_of_unknowns(n_total, f, I, live):
    """Connected components of the free views under the rows of A that are non-zero and carry a
    non-zero weight; `anchored[c]` = some edge of the component touches a fixed view through a
    surviving coefficient."""
    nu = n_total - f
    parent = list(range(nu))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a
    A = O.make_A(n_total, f, I).tocsr()
    anchored_v = np.zeros(nu, dtype=bool)
    for k in np.flatnonzero(live):
        cols = A.indices[A.indptr[k]:A.indptr[k + 1]]
        if len(cols) == 2:
            parent[find(cols[0])] = find(cols[1])
        elif len(cols) == 1:
            anchored_v[cols[0]] = True
    comp = np.array([find(v) for v in range(nu)])
    anchored = {c: bool(anchored_v[comp == c].any()) for c in np.unique(comp)}
    return comp, anchored


def test_rank_deficient_cases_are_characterised_not_pinned():
    """Talwar zeros + make_A-dropped rows: an isolated view and a floating component."""
    # views 0,1 fixed; chain 2-3-4 tied to view 0; views 5,6,7 connected among themselves and tied to
    # the rest ONLY by an edge whose SECOND endpoint is fixed (make_A empties that row, :770-771);
    # view 8 hangs on an edge that Talwar zeroes (:712) -> isolated.
    I = np.array([[0, 2], [2, 3], [3, 4], [1, 4], [5, 6], [6, 7], [5, 7], [5, 1], [4, 8]], dtype=np.int32)
    n, f = 9, 2
    rng = np.random.default_rng(3)
    Qgt = rng.normal(size=(n, 4)); Qgt /= np.linalg.norm(Qgt, axis=1, keepdims=True)
    QQ = synth.qmul(synth.qexp(rng.normal(scale=0.02, size=(len(I), 3))),
                    synth.qmul(Qgt[I[:, 1]], synth.qconj(Qgt[I[:, 0]])))
    Q = Qgt.copy()
    Q[f:] = synth.qmul(Q[f:], synth.qexp(rng.normal(scale=0.05, size=(n - f, 3))))
    w3 = residual_rows(I, QQ, Q)
    weights = np.array([1.0, 1.0001, 1.0001, 1.0, 1.0001, 1.0, 1.0001, 1.0, 0.0])   # last: Talwar zero
    DA, A = dense_DA(n, f, I, weights)
    assert not A[7].any()                                   # the quirk emptied the (5, 1) row
    DB = weights[:, None] * w3
    assert np.linalg.matrix_rank(DA) == (n - f) - 2         # one floating component + one isolated view
    rc, X = O.ls_solve(n, f, I, weights, w3)
    assert rc == 0
    Xmn = np.linalg.lstsq(DA, DB, rcond=None)[0]
    # 1. what every LS solution shares: D A X (residuals -> weights, :614-727)
    np.testing.assert_allclose(DA @ X, DA @ Xmn, atol=1e-12)
    # 2. the isolated view (8 -> unknown 6): zero in the oracle, in the minimum-norm solution, and in
    #    any basic solution (its column of D A is zero)
    assert not DA[:, 8 - f].any()
    np.testing.assert_array_equal(X[8 - f], 0.0)
    np.testing.assert_allclose(Xmn[8 - f], 0.0, atol=1e-14)
    # 3. the floating component {5, 6, 7}: solutions differ by ONE constant row on the component; the
    #    oracle pins one of its views to exactly 0 (a basic solution does that for some view, which
    #    one depends on SPQR's ordering), minimum-norm makes the component's mean zero
    comp, anchored = _components_of_unknowns(n, f, I, weights != 0.0)
    floating = [c for c, a in anchored.items() if not a and (comp == c).sum() > 1]
    assert len(floating) == 1
    rows = np.flatnonzero(comp == floating[0])
    assert sorted(rows + f) == [5, 6, 7]
    d = X[rows] - Xmn[rows]
    np.testing.assert_allclose(d, np.broadcast_to(d[0], d.shape), atol=1e-12)
    assert (np.abs(X[rows]).max(axis=1) == 0.0).sum() == 1
    np.testing.assert_allclose(Xmn[rows].mean(axis=0), 0.0, atol=1e-13)
    # 4. the anchored part is unique: everything agrees
    rest = np.setdiff1d(np.arange(n - f), np.concatenate([rows, [8 - f]]))
    np.testing.assert_allclose(X[rest], Xmn[rest], atol=1e-12)

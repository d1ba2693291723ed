"""NumPy/SciPy twin of the CPU oracle -- TEST INFRASTRUCTURE ONLY (PARITY UNPINNED, see
oracle/irotavg_oracle.h).

An independent, vectorised restatement of the same reference semantics (ral/l1_irls.cpp) that
shares no code with oracle/ral_oracle.c: sparse matrices are built with scipy.sparse exactly as
`make_A` (:755-780) and `make_AtA` (:811-848) define them, and the two SuiteSparse solves are
replaced by SuperLU (`scipy.sparse.linalg.splu`) on the normal matrix. It exists so the two
restatements can be checked against each other (tests/test_oracle_twin.py) and is used only in
this container (SciPy is not needed on the GPU box).
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as sla

EPS = 2.2204e-16  # ral/l1_irls.hpp:40


def quat_mult(a, b):  # :99-105, rows [x y z w]
    ax, ay, az, aw = a.T
    bx, by, bz, bw = b.T
    return np.stack([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz], axis=1)


def delta_rel(I, QQ, Q):  # :109-127
    Qinv = Q.copy()
    Qinv[:, 3] *= -1
    return quat_mult(Qinv[I[:, 1]], quat_mult(QQ, Q[I[:, 0]]))


def log_map(w):  # :498-532
    w = w.copy()
    s2 = np.linalg.norm(w[:, :3], axis=1)
    th = 2 * np.arctan2(s2, w[:, 3])
    th = np.where(th < -np.pi, th + 2 * np.pi, np.where(th >= np.pi, th - 2 * np.pi, th))
    with np.errstate(invalid="ignore", divide="ignore"):
        w[:, :3] *= (th / s2)[:, None]
    w[:, 3] = th
    w[s2 < EPS, :3] = 0
    return w


def exp_map(W):  # :471-492
    th = np.linalg.norm(W[:, :3], axis=1)
    with np.errstate(invalid="ignore", divide="ignore"):
        c = np.sin(th / 2) / th
    out = np.concatenate([W[:, :3] * c[:, None], np.cos(th / 2)[:, None]], axis=1)
    out[~np.isfinite(out)] = 0
    return out


def make_A(n, f, I):  # :755-780 incl. the edge-drop quirk
    m = len(I)
    A = sp.lil_matrix((m, n - f))
    for k, (e1, e2) in enumerate(I):
        j = e2 - f
        if j < 0:
            continue
        A[k, j] = 1
        i = e1 - f
        if i < 0:
            continue
        A[k, i] = -1
    return A.tocsc()


def make_AtA(nu, f, I):  # :811-848, (nu*nu) x m
    rows, cols, vals = {}, None, None
    ent = {}
    for k, (e1, e2) in enumerate(I):
        i, j = e1 - f, e2 - f
        if i >= 0:
            ent[(nu * i + i, k)] = 1
        if j >= 0:
            ent[(nu * j + j, k)] = 1
        if i >= 0 and j >= 0:
            ent[(nu * i + j, k)] = -1
            ent[(nu * j + i, k)] = -1
    keys = np.array(list(ent.keys()), dtype=np.int64).reshape(-1, 2)
    v = np.array(list(ent.values()), dtype=np.float64)
    return sp.csc_matrix((v, (keys[:, 0], keys[:, 1])), shape=(nu * nu, len(I)))


def H_from_AtA(AtA, sigx, nu):  # :308-317 + sp_vec_to_squared_mat :187-225
    h = (AtA @ sigx)
    return sp.csc_matrix(h.reshape(nu, nu, order="F")) if nu <= 2000 else None


def weights_update(cost, sigma, E, weights):  # :617-727
    e2 = np.sum(E * E, axis=1)
    e = np.sqrt(e2)
    w = weights.copy()
    with np.errstate(invalid="ignore", divide="ignore"):
        if cost == 0:
            pass
        elif cost == 3:
            w = np.minimum(1.0 / e2 ** (3. / 8.), 1e4)
        elif cost == 1:
            w = np.minimum(1.0 / np.sqrt(e), 1e4)
        elif cost == 2:
            w = np.minimum(1.0 / np.sqrt(np.sqrt(e)), 1e4)
        elif cost == 4:
            w = 1.0 / (e2 + sigma * sigma)
        elif cost == 5:
            t = 1.345 * sigma
            r = e / t
            w = np.where(r >= 1, np.sqrt(1. / r), w)
        elif cost == 6:
            w = 1.0 / np.sqrt(np.sqrt(1.0 + e2 / (sigma * sigma)))
        elif cost == 7:
            t = 1.339 * sigma
            r = e / t
            w = np.sqrt(np.sin(r) / r)
            w = np.where(r >= np.pi, 0.0, np.where(r < 1e-4, 1.0, w))
            w = np.where(w < 1e-4, 1e-4, w)
        elif cost == 8:
            t = 4.685 * sigma
            w = np.maximum(1.0 - e2 / (t * t), 1e-4)
        elif cost == 9:
            t = 2.385 * sigma
            w = 1.0 / np.sqrt(1.0 + e2 / (t * t))
        elif cost == 10:
            t = 1.400 * sigma
            w = 1.0 / np.sqrt(1.0 + e / t)
        elif cost == 11:
            t = 1.205 * sigma
            r = e / t
            w = np.where(r < 1e-4, 1.0, np.sqrt(np.tanh(r) / r))
        elif cost == 12:
            t = 2.795 * sigma
            w = np.where(e2 < t * t, 1.0001, 0.0)
        elif cost == 13:
            t = 2.985 * sigma
            w = np.maximum(np.exp(-.5 * e2 / (t * t)), 1e-4)
        else:
            raise ValueError("Unknown cost!!")
    return w


def irls(QQ, I, Q, f, cost=4, sigma=5 * np.pi / 180, max_iters=50, change_th=1e-3):  # :559-752
    Q = Q.copy()
    m, n = len(I), len(Q) - f
    A = make_A(len(Q), f, I)
    weights = np.ones(m)
    score, iters, scores = np.inf, 0, []
    while score > change_th and iters < max_iters:
        w = log_map(delta_rel(I, QQ, Q))
        D = sp.diags(weights)
        DA = (D @ A).tocsc()
        DB = weights[:, None] * w[:, :3]
        X = sla.splu((DA.T @ DA).tocsc()).solve(DA.T @ DB)
        E = A @ X - w[:, :3]
        weights = weights_update(cost, sigma, E, weights)
        score = np.linalg.norm(X, axis=1).mean()
        scores.append(score)
        Q[f:] = quat_mult(Q[f:], exp_map(np.concatenate([X, np.zeros((n, 1))], axis=1)))
        iters += 1
    return dict(Q=Q, weights=weights, iters=iters, scores=np.array(scores))


def l1decode_pd(A, Hfun, y, pdmaxiter=2):  # :228-468, x0 = 0
    PDTOL, alpha, beta, mu = 1e-3, 0.01, 0.5, 10
    m, n = A.shape
    x = np.zeros(n)
    Ax = A @ x
    r = np.abs(y - Ax)
    u = 0.95 * r + 0.10 * r.max()
    fu1, fu2 = Ax - y - u, -Ax + y - u
    l1, l2 = -1 / fu1, -1 / fu2
    Atv = A.T @ (l1 - l2)
    sdg = -(fu1 @ l1 + fu2 @ l2)
    tau = mu * 2 * m / sdg
    rcent = np.concatenate([-l1 * fu1, -l2 * fu2]) - 1 / tau
    rdual = np.concatenate([Atv, 1 - l1 - l2])
    resnorm = np.sqrt(rdual @ rdual + rcent @ rcent)
    pditer = 0
    done = sdg < PDTOL or pditer >= pdmaxiter
    stuck = False
    while not done:
        pditer += 1
        w2 = -1 - 1 / tau * (1 / fu1 + 1 / fu2)
        sig1 = -l1 / fu1 - l2 / fu2
        sig2 = l1 / fu1 - l2 / fu2
        sigx = sig1 - sig2 ** 2 / sig1
        w1 = -1 / tau * (A.T @ (-1 / fu1 + 1 / fu2))
        w1p = w1 - A.T @ ((sig2 / sig1) * w2)
        dx = sla.splu(Hfun(sigx)).solve(w1p)
        Adx = A @ dx
        du = (w2 - sig2 * Adx) / sig1
        dl1 = -(l1 / fu1) * (Adx - du) - l1 - (1 / tau) / fu1
        dl2 = (l2 / fu2) * (Adx + du) - l2 - (1 / tau) / fu2
        Atdv = A.T @ (dl1 - dl2)
        s = 1.0
        for num, den in ((-l1, dl1), (-l2, dl2)):
            sel = den < 0
            if sel.any():
                s = min(s, (num[sel] / den[sel]).min())
        for num, den in ((-fu1, Adx - du), (-fu2, -Adx - du)):
            sel = den > 0
            if sel.any():
                s = min(s, (num[sel] / den[sel]).min())
        s *= 0.99
        suff, back = False, 0
        while not suff:
            xp, up = x + s * dx, u + s * du
            Axp, Atvp = Ax + s * Adx, Atv + s * Atdv
            l1p, l2p = l1 + s * dl1, l2 + s * dl2
            fu1p, fu2p = Axp - y - up, -Axp + y - up
            rdp = np.concatenate([Atvp, 1 + (-l1p - l2p)])
            rcp = np.concatenate([-l1p * fu1p, -l2p * fu2p]) - 1 / tau
            suff = np.sqrt(rdp @ rdp + rcp @ rcp) <= (1 - alpha * s) * resnorm
            s *= beta
            back += 1
            if back > 32:
                return x, True
        x, u, Ax, Atv, l1, l2, fu1, fu2 = xp, up, Axp, Atvp, l1p, l2p, fu1p, fu2p
        sdg = -(fu1 @ l1 + fu2 @ l2)
        tau = mu * 2 * m / sdg
        rcent = np.concatenate([-l1 * fu1, -l2 * fu2]) - 1 / tau
        resnorm = np.sqrt(rdp @ rdp + rcent @ rcent)
        done = sdg < PDTOL or pditer >= pdmaxiter
    return x, stuck


def _H_builder(n_total, f, I):
    """H(sigx) = reshape(AtA*sigx): endpoints skipped independently (:825-843)."""
    nu = n_total - f
    i = I[:, 0].astype(np.int64) - f
    j = I[:, 1].astype(np.int64) - f
    fi, fj, both = i >= 0, j >= 0, (i >= 0) & (j >= 0)
    Si = sp.csr_matrix((np.ones(fi.sum()), (i[fi], np.flatnonzero(fi))), shape=(nu, len(I)))
    Sj = sp.csr_matrix((np.ones(fj.sum()), (j[fj], np.flatnonzero(fj))), shape=(nu, len(I)))

    def H(sigx):
        d = Si @ sigx + Sj @ sigx
        off = sp.coo_matrix((-sigx[both], (i[both], j[both])), shape=(nu, nu))
        return (sp.diags(d) + off + off.T).tocsc()
    return H


def l1ra(QQ, I, Q, f, max_iters=5, change_th=1e-3):  # :851-912
    Q = Q.copy()
    n = len(Q) - f
    A = make_A(len(Q), f, I)
    H = _H_builder(len(Q), f, I)
    score, it, scores = np.inf, 0, []
    while score >= change_th and it < max_iters:
        w = log_map(delta_rel(I, QQ, Q))
        X = np.stack([l1decode_pd(A, H, w[:, c], 2)[0] for c in range(3)], axis=1)
        score = np.linalg.norm(X, axis=1).mean()
        scores.append(score)
        Q[f:] = quat_mult(Q[f:], exp_map(np.concatenate([X, np.zeros((n, 1))], axis=1)))
        it += 1
    return dict(Q=Q, iters=it, scores=np.array(scores))


def init_mst(Q, QQ, I, f):  # :915-979
    Q = Q.copy()
    n = len(Q)
    flags = np.zeros(n, dtype=bool)
    flags[0] = True
    count = 1
    while count < n:
        span = False
        for k, (e1, e2) in enumerate(I):
            if flags[e1] and not flags[e2]:
                if e2 >= f:
                    Q[e2] = quat_mult(QQ[k:k + 1], Q[e1:e1 + 1])[0]
                count += 1
                flags[e2] = True
                span = True
            if (not flags[e1]) and flags[e2]:
                if e1 >= f:
                    qi = QQ[k:k + 1].copy()
                    qi[0, 3] *= -1
                    Q[e1] = quat_mult(qi, Q[e2:e2 + 1])[0]
                count += 1
                flags[e1] = True
                span = True
        if not span and count < n:
            raise RuntimeError("Relative rotations DO NOT SPAN all the nodes in the VIEW GRAPH")
    return Q

"""Seeded synthetic SO(3) view-graph generator; data plumbing for tests and
bench.py, not part of the solve path.

Conventions follow the reference: quaternion rows are [x, y, z, w]; an edge (i, j), i < j,
carries QQ with Q_j ~= QQ (x) Q_i (ral/l1_irls.cpp:941, src/ViewGraph.cpp:1282-1307 stores each
connection once under its newer view j).
"""
import numpy as np


def qmul(a, b):
    """Row-wise Hamilton product, rows [x y z w] (same convention as ral/l1_irls.cpp:99-105)."""
    ax, ay, az, aw = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bx, by, bz, bw = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz], axis=-1)


def qconj(a):
    return a * np.array([-1.0, -1.0, -1.0, 1.0])


def qexp(r):
    th = np.linalg.norm(r, axis=-1, keepdims=True)
    with np.errstate(invalid="ignore", divide="ignore"):
        c = np.where(th > 0, np.sin(th / 2) / th, 0.5)
    return np.concatenate([r * c, np.cos(th / 2)], axis=-1)


def angular_distance(q1, q2):
    """Sign-invariant rotation angle between quaternion rows (normalised here). Uses the chord
    form 4*asin(min(|q1-q2|,|q1+q2|)/2), which stays accurate for tiny angles (acos does not)."""
    q1 = q1 / np.linalg.norm(q1, axis=-1, keepdims=True)
    q2 = q2 / np.linalg.norm(q2, axis=-1, keepdims=True)
    d = np.minimum(np.linalg.norm(q1 - q2, axis=-1), np.linalg.norm(q1 + q2, axis=-1))
    return 4 * np.arcsin(np.clip(d / 2, 0.0, 1.0))


def band_loop_topology(n, m, p_loop, rng):
    """Edge list (m,2) int32, i<j, grouped by the newer view j.

    Every view is linked to its w = floor(m_band/n) predecessors (m_band = m - round(p_loop*m));
    the remainder of m_band is spent on (w+1)-th predecessor links of evenly spaced views, so the
    total is exactly m; the loop edges are uniform random pairs with |i-j| > w+1, de-duplicated.
    """
    n_loop = int(round(p_loop * m))
    m_band = m - n_loop
    w = m_band // n
    while w > 0 and n * w - w * (w + 1) // 2 > m_band:
        w -= 1
    ii, jj = [], []
    for d in range(1, w + 1):
        j = np.arange(d, n, dtype=np.int64)
        ii.append(j - d)
        jj.append(j)
    have = sum(len(x) for x in ii)
    extra = m_band - have
    if extra > 0:
        d = w + 1
        cand = np.arange(d, n, dtype=np.int64)
        if extra > len(cand):
            raise ValueError("m too large for n")
        pick = cand[np.linspace(0, len(cand) - 1, extra).round().astype(np.int64)]
        pick = np.unique(pick)
        while len(pick) < extra:  # rounding collisions: top up deterministically
            rest = np.setdiff1d(cand, pick)
            pick = np.sort(np.concatenate([pick, rest[:extra - len(pick)]]))
        ii.append(pick - d)
        jj.append(pick)
    ii = np.concatenate(ii) if ii else np.zeros(0, np.int64)
    jj = np.concatenate(jj) if jj else np.zeros(0, np.int64)
    is_loop = np.zeros(len(ii), dtype=bool)
    if n_loop > 0:
        got = np.zeros((0, 2), np.int64)
        while len(got) < n_loop:
            a = rng.integers(0, n, size=2 * n_loop + 64)
            b = rng.integers(0, n, size=2 * n_loop + 64)
            lo, hi = np.minimum(a, b), np.maximum(a, b)
            ok = (hi - lo) > (w + 1)
            pairs = np.stack([lo[ok], hi[ok]], axis=1)
            got = np.concatenate([got, pairs])
            _, first = np.unique(got[:, 0] * n + got[:, 1], return_index=True)
            got = got[np.sort(first)]
        got = got[:n_loop]
        ii = np.concatenate([ii, got[:, 0]])
        jj = np.concatenate([jj, got[:, 1]])
        is_loop = np.concatenate([is_loop, np.ones(n_loop, dtype=bool)])
    # group by newer view j; within a view: band edges by increasing i distance, loops last
    order = np.lexsort((jj - ii, is_loop, jj))
    I = np.stack([ii[order], jj[order]], axis=1).astype(np.int32)
    return I, is_loop[order], w


def make_graph(n, m, p_loop=0.0, sigma_n=0.01, p_out=0.05, seed=0, p_band_out=0.0, band_out_scale=0.3):
    """Returns dict(I, QQ, Qgt, is_loop, is_outlier, w) -- QQ, Qgt are (rows,4) [x y z w].

    p_band_out > 0 (not part of SURVEY.md 8(d)'s generator, whose outliers are loop edges only): that
    share of ALL edges additionally carries a rotation error of N(0, band_out_scale^2 I) rad -- the
    robust weights of a band-only graph then move non-uniformly. Drawn from its own generator, so the
    rest of the graph is the same as without it."""
    rng = np.random.default_rng(seed)
    Qgt = rng.normal(size=(n, 4))
    Qgt /= np.linalg.norm(Qgt, axis=1, keepdims=True)
    I, is_loop, w = band_loop_topology(n, m, p_loop, rng)
    mm = len(I)
    eps = rng.normal(scale=sigma_n, size=(mm, 3))
    QQ = qmul(qexp(eps), qmul(Qgt[I[:, 1]], qconj(Qgt[I[:, 0]])))
    is_out = np.zeros(mm, dtype=bool)
    loop_idx = np.flatnonzero(is_loop)
    n_out = int(round(p_out * len(loop_idx)))
    if n_out > 0:
        sel = rng.choice(loop_idx, size=n_out, replace=False)
        R = rng.normal(size=(n_out, 4))
        R /= np.linalg.norm(R, axis=1, keepdims=True)
        QQ[sel] = R
        is_out[sel] = True
    if p_band_out > 0:
        rng2 = np.random.default_rng([seed, 0xBAD])
        bad = rng2.choice(mm, size=int(round(p_band_out * mm)), replace=False)
        QQ[bad] = qmul(qexp(rng2.normal(scale=band_out_scale, size=(len(bad), 3))), QQ[bad])
        is_out[bad] = True
    return dict(I=I, QQ=QQ, Qgt=Qgt, is_loop=is_loop, is_outlier=is_out, w=w, n=n, m=mm,
                p_loop=p_loop, seed=seed)


def add_closures(S, nclose, seed=7, wrong=0):
    """A view sequence (make_graph with p_loop 0) plus `nclose` loop closures between views 100 ... n/2 apart,
    `wrong` of them with a random relative rotation (a false loop detection); edges stay grouped by the newer view,
    as ViewGraph::rotAvg lists them (src/ViewGraph.cpp:1290-1300)."""
    n = S["n"]
    rng = np.random.default_rng(seed + 100)
    a = rng.integers(0, n - 200, nclose)
    b = np.minimum(n - 1, a + rng.integers(100, n // 2, nclose))
    eps = rng.normal(scale=0.01, size=(nclose, 3))
    QQc = qmul(qexp(eps), qmul(S["Qgt"][b], qconj(S["Qgt"][a])))
    if wrong:
        R = rng.normal(size=(wrong, 4))
        R /= np.linalg.norm(R, axis=1, keepdims=True)
        QQc[:wrong] = R
    I = np.concatenate([S["I"], np.stack([a, b], 1)]).astype(np.int32)
    QQ = np.concatenate([S["QQ"], QQc])
    order = np.lexsort((np.arange(len(I)), I[:, 1]))
    return dict(S, I=I[order], QQ=QQ[order], m=len(I))


def closure_graph(n, m, nclose, seed, wrong=0, f=1):
    """a view sequence (band) + nclose long-range edges, `wrong` of them with a random rotation -- the graph a SLAM
    front end produces (sequence + loop closures, src/IRotAvg.cpp:371-378)"""
    S = make_graph(n, m, 0.0, seed=seed)
    rng = np.random.default_rng(seed + 100)
    a = rng.integers(f, n - 200, nclose)
    b = np.minimum(n - 1, a + rng.integers(100, n // 2, nclose))
    QQc = qmul(qexp(rng.normal(scale=0.01, size=(nclose, 3))), qmul(S["Qgt"][b], qconj(S["Qgt"][a])))
    if wrong:
        R = rng.normal(size=(wrong, 4))
        QQc[:wrong] = R / np.linalg.norm(R, axis=1, keepdims=True)
    I = np.concatenate([S["I"], np.stack([a, b], 1)]).astype(np.int32)
    QQ = np.concatenate([S["QQ"], QQc])
    order = np.lexsort((np.arange(len(I)), I[:, 1]))      # stored under the later view, as the reference does
    return dict(S, I=I[order], QQ=QQ[order], m=len(I))

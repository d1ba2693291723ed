"""Would a warm-started Newton-Schulz inverse X <- X (2I - D X) converge from the previous IRLS iteration's inverse?
Measured: || I - D_{t+1} D_t^{-1} ||_2 over the 24 x 24 diagonal blocks of the level-0 operator (the blocks the first round
of every chunk inverts) for consecutive IRLS iterations, on the headline graph and on the one with 2 % of all edges off by
0.3 rad (bench.py's also_band_outliers). Newton-Schulz converges iff that norm is < 1 and needs ~log2(log(eps)/log(norm))
steps; DESIGN.md section 5a prices a step. CPU only (SciPy); the outer iteration is oracle/np_twin.py's.

    python tools/dev/ns_norm.py [views edges]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as sla
from irotavg_amd import synth
from oracle import np_twin as T, oracle as O

n, m = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100000, 2000000)
B, NBLK = 24, 400
SIG = 5 * np.pi / 180
for name, kw in (("headline", {}), ("band outliers 2 %", {"p_band_out": 0.02})):
    S = synth.make_graph(n, m, 0.0, seed=0, **kw)
    Q = np.zeros((n, 4)); Q[:, 3] = 1; Q[0] = S["Qgt"][0]
    rc, Q = O.init_mst(Q, S["QQ"], S["I"], 1)
    I, QQ, f = S["I"], S["QQ"], 1
    i, j = I[:, 0].astype(np.int64) - f, I[:, 1].astype(np.int64) - f
    keep = j >= 0
    rows = np.concatenate([np.flatnonzero(keep), np.flatnonzero(keep & (i >= 0))])
    cols = np.concatenate([j[keep], i[keep & (i >= 0)]])
    vals = np.concatenate([np.ones(keep.sum()), -np.ones((keep & (i >= 0)).sum())])
    A = sp.csr_matrix((vals, (rows, cols)), shape=(len(I), n - f)).tocsc()
    w = np.ones(len(I))
    Dprev, score, it = None, np.inf, 0
    print("%s: %d views, %d edges" % (name, n, len(I)))
    while score > 1e-3 and it < 15:
        r = T.log_map(T.delta_rel(I, QQ, Q))
        DA = (sp.diags(w) @ A).tocsc()
        H = (DA.T @ DA).tocsc()
        X = sla.splu(H).solve(DA.T @ (w[:, None] * r[:, :3]))
        blocks = np.stack([H[b * B:(b + 1) * B, b * B:(b + 1) * B].toarray() for b in range(100, 100 + NBLK)])
        if Dprev is not None:
            nr = np.array([np.linalg.norm(np.eye(B) - blocks[k] @ np.linalg.inv(Dprev[k]), 2) for k in range(NBLK)])
            # ... and after the best uniform rescaling of the old inverse (a scalar per block is one multiply)
            sc = np.array([np.trace(blocks[k]) / np.trace(Dprev[k]) for k in range(NBLK)])
            nr2 = np.array([np.linalg.norm(np.eye(B) - blocks[k] @ np.linalg.inv(Dprev[k]) / sc[k], 2) for k in range(NBLK)])
            print("  iteration %d -> %d: ||I - D_new D_old^-1||_2 median %.3g max %.3g; with the old inverse rescaled by tr(D_new)/tr(D_old): "
                  "median %.3g max %.3g; blocks with norm >= 0.5 (sweep instead): %d of %d" % (
                      it, it + 1, np.median(nr), nr.max(), np.median(nr2), nr2.max(), int((nr2 >= 0.5).sum()), NBLK))
        Dprev = blocks
        E = A @ X - r[:, :3]
        w = T.weights_update(4, SIG, E, w)
        score = np.linalg.norm(X, axis=1).mean()
        Q[f:] = T.quat_mult(Q[f:], T.exp_map(np.concatenate([X, np.zeros((n - f, 1))], axis=1)))
        it += 1
        print("  iteration %d: score %.3e" % (it, score), flush=True)

import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irotavg_amd import capi, synth, ral
SIG = 5 * np.pi / 180
n, m, f = 16390, 81950, 3
S = synth.make_graph(n, m, 0.0, seed=4)
Q0 = np.zeros((n, 4)); Q0[:, 3] = 1; Q0[:f] = S["Qgt"][:f]
ral.init_mst(Q0, S["QQ"], S["I"], f)
for kw in (dict(), dict(pcg_classic=1), dict(pcg_max_iters=300)):
    with capi.Graph(S["I"], S["QQ"], n, f, **kw) as G:
        G.set_rotations(Q0)
        r = G.irls(4, SIG, 100, 1e-3, allow_rc=(-8, -3))
        st = G.stats()
        print(kw, "rc", r["rc"], "iters", r["iters"], "scores", r["scores"][:3], {k: st[k] for k in ("pcg_solves", "pcg_iters", "pcg_stagnated", "dense_inversions")}, st["level_rows"][:3], st.get("last_relres"))

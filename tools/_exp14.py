import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irotavg_amd import capi, synth, ral
SIG = 5 * np.pi / 180
n, m = 75000, 300000
for pl in (0.0, 0.0003):
    S = synth.make_graph(n, m, pl, seed=0)
    Q0 = np.zeros((n, 4)); Q0[:, 3] = 1; Q0[:1] = S["Qgt"][:1]
    ral.init_mst(Q0, S["QQ"], S["I"], 1)
    for kw in (dict(), dict(mg_kc=1.6), dict(mg_kc=2.4), dict(mg_kc=3.0), dict(mg_agg0=4), dict(mg_agg0=4, mg_agg=4), dict(mg_agg0=4, mg_kc=1.6), dict(mg_agg0=2, mg_agg=8), dict(mg_omega=0.9), dict(mg_omega=0.5)):
        with capi.Graph(S["I"], S["QQ"], n, 1, **kw) as G:
            ts = []
            for rep in range(2):
                G.set_rotations(Q0)
                t = time.perf_counter()
                r = G.irls(4, SIG, 100, 1e-3, allow_rc=(-8,))
                ts.append(time.perf_counter() - t)
            st = G.stats()
            print(pl, kw, "ms", round(1e3 * min(ts), 2), "iters", r["iters"], st["level_rows"][:st["levels"]], "pcg/solve", round(st["pcg_iters"] / max(st["pcg_solves"], 1), 1), "inv", st["dense_inversions"], flush=True)

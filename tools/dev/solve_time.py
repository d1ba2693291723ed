"""Time of one direct solve (and of its ways back) on a few graphs."""
import sys
sys.path.insert(0, ".")
import numpy as np
from irotavg_amd import capi, synth, ral
for n, m in ((100000, 2000000), (10000, 150000), (75000, 300000)):
    S = synth.make_graph(n, m, 0.0, seed=0)
    Q = np.zeros((n, 4)); Q[:, 3] = 1; Q[0] = S["Qgt"][0]
    ral.init_mst(Q, S["QQ"], S["I"], 1)
    with capi.Graph(S["I"], S["QQ"], n, 1) as G:
        G.set_rotations(Q)
        r = G.irls(4, 5 * np.pi / 180, 50, 1e-3)
        rr = G.direct_residual()
        s = min(G.time_kernel(19, 30) for _ in range(3))
        nl = len(G.direct_info()["levels"])
        back = [min(G.time_kernel(40 + l, 50) for _ in range(3)) for l in range(nl)]
        print(n, m, "iters", r["iters"], "score %.9e" % r["scores"][-1], "solve %.1f us" % (1e3 * s), "relres %.1e" % rr.max(), "back", ["%.1f" % (1e3 * b) for b in back], flush=True)

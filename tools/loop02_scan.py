"""100k / 2M with 2 % random loop edges on the iterative path: ms per irls for a few hierarchy shapes (development)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from irotavg_amd import capi, ral, synth
SIG = 5 * np.pi / 180
n, m = 100000, 2000000
S = synth.make_graph(n, m, 0.02, seed=0)
Q0 = np.zeros((n, 4)); Q0[:, 3] = 1; Q0[0] = S["Qgt"][0]
ral.init_mst(Q0, S["QQ"], S["I"], 1)
ref = None
for opts in [dict(), dict(mg_dense_max=1024), dict(mg_dense_max=512), dict(mg_dense_max=256), dict(mg_agg=4), dict(mg_agg=16),
             dict(mg_agg=16, mg_dense_max=1024), dict(inexact_outer=1), dict(inexact_outer=1, mg_dense_max=512),
             dict(inexact_outer=1, mg_agg=16)]:
    try:
        with capi.Graph(S["I"], S["QQ"], n, 1, **opts) as G:
            G.set_rotations(Q0); G.snapshot_rotations()
            for _ in range(3):
                G.restore_rotations(); r = G.irls(4, SIG, 100, 1e-3)
            G.synchronize(); G.reset_stats()
            t = time.perf_counter()
            for _ in range(5):
                G.restore_rotations(); r = G.irls(4, SIG, 100, 1e-3)
            G.synchronize()
            dt = (time.perf_counter() - t) / 5
            st = G.stats(); Q = G.get_rotations()
        if ref is None: ref = Q
        print("%-45s %.2f ms  iters %d  pcg/solve %.1f  rows %s  diff %.2e" % (opts, 1e3 * dt, r["iters"], st["pcg_iters"] / max(st["pcg_solves"], 1),
              st["level_rows"], synth.angular_distance(Q, ref).max()), flush=True)
    except Exception as e:
        print(opts, "failed:", e, flush=True)

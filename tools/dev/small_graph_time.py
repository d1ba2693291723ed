"""irls on graphs that are ONE dense level (<= 2048 views): ms per call (what solving to the attainable residual costs;
run once with IROTAVG_NO_DENSE_REFINE=1 for round 5's behaviour)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from irotavg_amd import capi, ral, synth
for n, m in ((300, 2400), (1000, 8000), (2000, 30000)):
    S = synth.make_graph(n, m, 0.02, seed=1)
    Q0 = np.zeros((n, 4)); Q0[:, 3] = 1; Q0[0] = S["Qgt"][0]
    ral.init_mst(Q0, S["QQ"], S["I"], 1)
    with capi.Graph(S["I"], S["QQ"], n, 1) as G:
        G.set_rotations(Q0); G.snapshot_rotations()
        for rep in range(12):
            if rep == 2:
                G.synchronize(); G.reset_stats(); t0 = time.perf_counter()
            G.restore_rotations(); r = G.irls(4, 5 * np.pi / 180, 50, 1e-3)
        G.synchronize(); ms = 1e3 * (time.perf_counter() - t0) / 10; st = G.stats()
    print(n, m, "irls %.3f ms, %d iterations, %.1f PCG iterations per solve, last relres %.1e" % (
        ms, r["iters"], st["pcg_iters"] / max(1, st["pcg_solves"]), max(st["last_relres"])), flush=True)

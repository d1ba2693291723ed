"""Robustness sweep: every robust cost on the 100k/2M graphs (PCG iterations, time, result codes)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from irotavg_amd import capi, ral, synth
n, m = 100000, 2000000
for pl in (0.0, 0.02):
    S = synth.make_graph(n, m, pl, seed=0)
    Q = np.zeros((n, 4)); Q[:, 3] = 1; Q[0] = S["Qgt"][0]
    ral.init_mst(Q, S["QQ"], S["I"], 1)
    G = capi.Graph(S["I"], S["QQ"], n, 1)
    G.set_rotations(Q); G.snapshot_rotations()
    for cost in range(14):
        G.restore_rotations(); G.reset_stats()
        t = time.perf_counter()
        r = G.irls(cost, 5*np.pi/180, 100, 1e-3, allow_rc=(-8, -3))
        dt = time.perf_counter() - t
        st = G.stats()
        err = synth.angular_distance(G.get_rotations(), S["Qgt"]).mean()
        print("p_loop %.2f %-14s rc %2d iters %3d pcg/solve %6.1f  %.1f ms  err %.4f" % (pl, ral.COST_NAMES[cost], r["rc"], r["iters"], st["pcg_iters"]/max(st["pcg_solves"],1), dt*1e3, err), flush=True)
    G.close()

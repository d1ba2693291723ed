"""One workload under rocprofv3: `irls` on a view sequence with loop closures (the banded direct solver's closure path)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from irotavg_amd import capi, ral
from tools.dev.bcr_closures import graph
n, m, nc = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
S = graph(n, m, nc, 7, max(1, nc // 33))
Q0 = np.zeros((n, 4)); Q0[:, 3] = 1; Q0[0] = S["Qgt"][0]
ral.init_mst(Q0, S["QQ"], S["I"], 1)
with capi.Graph(S["I"], S["QQ"], n, 1, band_direct=1) as G:
    G.set_rotations(Q0); G.snapshot_rotations()
    for rep in range(4):
        G.restore_rotations()
        t0 = time.perf_counter(); r = G.irls(4, 5 * np.pi / 180, 50, 1e-3); G.synchronize(); t1 = time.perf_counter()
    print("irls", r["iters"], "iterations", 1e3 * (t1 - t0), "ms", G.stats()["direct_solves"])

import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from irotavg_amd import capi, ral, synth
n, m = 100000, 2000000
for pl in (0.0, 0.02):
    S = synth.make_graph(n, m, pl, seed=0)
    Q = np.zeros((n, 4)); Q[:, 3] = 1; Q[0] = S["Qgt"][0]
    ral.init_mst(Q, S["QQ"], S["I"], 1)
    for pol in (0, 1):
        o = capi.default_options()
        res = (capi.C.c_int * 7)(0, pol, 0, 0, 0, 0, 0)
        G = capi.Graph(S["I"], S["QQ"], n, 1, reserved=res)
        G.set_rotations(Q); G.snapshot_rotations()
        G.irls(4, 5*np.pi/180, 100, 1e-3)
        t = time.perf_counter()
        for _ in range(5):
            G.restore_rotations(); r = G.irls(4, 5*np.pi/180, 100, 1e-3)
        G.synchronize(); dt = (time.perf_counter() - t) / 5
        st = G.stats()
        print(pl, "always-refresh" if pol else "adaptive", "ms/step", dt*1e3, "iters", r["iters"], "pcg/solve", st["pcg_iters"]/st["pcg_solves"], "inv ms", G.time_kernel(7, 5), flush=True)
        G.close()

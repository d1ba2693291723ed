import sys, numpy as np
sys.path.insert(0, '.')
import bench
from irotavg_amd import capi
for n, m in ((100000, 2000000), (10000, 150000), (1000000, 20000000)):
    S, Q0 = bench.build_problem(n, m, 0.0, 0)
    G = capi.Graph(S["I"], S["QQ"], S["n"], 1); G.set_rotations(Q0)
    r = G.irls(4, 5 * np.pi / 180, 100, 1e-3)
    print(n, m, r["iters"], ["%.2e" % s for s in r["scores"]])
    G.close()

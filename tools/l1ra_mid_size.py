import sys, time, numpy as np
sys.path.insert(0, '.')
import bench
from irotavg_amd import capi
n, m = int(sys.argv[1]), int(sys.argv[2])
S, Q0 = bench.build_problem(n, m, 0.0, 0)
G = capi.Graph(S["I"], S["QQ"], S["n"], 1); G.set_rotations(Q0)
print(G.direct_info())
for rep in range(3):
    t = time.perf_counter()
    try:
        r = G.l1ra(3, 1e-3); G.synchronize()
        print("l1ra ok", r["iters"], "%.1f ms" % (1e3 * (time.perf_counter() - t)), flush=True)
    except Exception as e:
        print("l1ra FAILED", e, "%.1f ms" % (1e3 * (time.perf_counter() - t)), flush=True)

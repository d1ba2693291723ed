"""l1ra(5) at 100k/2M, mean of N runs (A/B aid)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import bench
from irotavg_amd import capi
S, Q0 = bench.build_problem(100000, 2000000, 0.0, 0)
G = capi.Graph(S["I"], S["QQ"], S["n"], 1); G.set_rotations(Q0); G.snapshot_rotations()
G.l1ra(1, 1e-3)
ts = []
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    G.restore_rotations(); G.synchronize()
    t = time.perf_counter(); r = G.l1ra(5, 1e-3); G.synchronize(); ts.append(1e3 * (time.perf_counter() - t) / r["iters"])
print("l1ra ms per outer iteration: mean %.3f median %.3f min %.3f max %.3f" % (np.mean(ts), np.median(ts), min(ts), max(ts)))

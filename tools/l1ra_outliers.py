"""l1ra(5) repeated: mean / max of the call time (development aid: looks for stalls)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import bench
from irotavg_amd import capi
n, m, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
S, Q0 = bench.build_problem(n, m, 0.0, 0)
G = capi.Graph(S["I"], S["QQ"], S["n"], 1); G.set_rotations(Q0); G.snapshot_rotations()
G.l1ra(1, 1e-3)
ts = []
for _ in range(reps):
    G.restore_rotations(); G.synchronize()
    t = time.perf_counter(); r = G.l1ra(5, 1e-3); G.synchronize(); ts.append(1e3 * (time.perf_counter() - t))
ts = np.array(ts)
print("first calls:", np.round(ts[:4], 2), "argmax", int(ts.argmax()))
print("l1ra(5) ms: mean %.3f median %.3f max %.3f; calls above 3x the median: %d of %d" % (ts.mean(), np.median(ts), ts.max(), (ts > 3 * np.median(ts)).sum(), reps))

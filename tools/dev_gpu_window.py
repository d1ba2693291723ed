import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from irotavg_amd import capi, synth
from irotavg_amd.viewgraph import ViewGraph
from oracle import oracle as O
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_viewgraph import build_sequence, rot
n = 400
Qgt, rel = build_sequence(n, seed=3, n_loops=0)
vg = ViewGraph()
by_new = {}
for (i, j), R in rel.items():
    by_new.setdefault(j, []).append((i, R))
ts = []
for v in range(n):
    if v == 0:
        R0 = rot(Qgt[0])
    else:
        i, R = sorted(by_new[v], key=lambda t: -t[0])[0]
        R0 = R @ vg.R(i)
    vg.addView(R0)
    for (i, R) in by_new.get(v, []):
        vg.connect(i, v, R)
    if v % 20 == 0:
        vg.fixPose(v, rot(Qgt[v]))
    t = time.perf_counter(); a = vg.rotAvg(10); ts.append(time.perf_counter() - t)
ts = np.array(ts[20:])
print("rotAvg(10) per call: mean %.3f ms median %.3f ms max %.3f ms" % (ts.mean()*1e3, np.median(ts)*1e3, ts.max()*1e3), a)
t = time.perf_counter(); a = vg.rotAvg(5000000); print("global rotAvg on %d views: %.3f ms" % (n, (time.perf_counter()-t)*1e3), a)

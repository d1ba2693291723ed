#!/usr/bin/env python3
"""Steady-state cost of the loop-closure branch of the reference's policy (rotAvg(5000000), src/IRotAvg.cpp:371-378)
at 75k views / 300k connections + a few loop closures: IROTAVG_ROTAVG_TIMING=1 prints the phases of the last call."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irotavg_amd import synth
from irotavg_amd.viewgraph import ViewGraph
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_incremental import quat2rmat
n = int(sys.argv[1]) if len(sys.argv) > 1 else 75000
rng = np.random.default_rng(0)
Qgt = rng.normal(size=(n, 4)); Qgt /= np.linalg.norm(Qgt, axis=1, keepdims=True)
Rgt = quat2rmat(Qgt)
def rel(i, j):
    e = synth.qexp(rng.normal(scale=0.01, size=(1, 3)))[0]
    return quat2rmat(synth.qmul(e, synth.qmul(Qgt[j], synth.qconj(Qgt[i]))))
vg = ViewGraph()
for v in range(n):
    vg.addView(quat2rmat(synth.qmul(synth.qexp(rng.normal(scale=0.01, size=(1, 3)))[0], Qgt[v])))
    for d in range(1, min(4, v) + 1):
        vg.connect(v - d, v, rel(v - d, v))
    if v % 20 == 0:
        vg.fixPose(v, Rgt[v])
for v in rng.choice(np.arange(2000, n), 8, replace=False):
    vg.connect(int(rng.integers(0, v - 1000)), int(v), rel(int(rng.integers(0, v - 1000)), int(v)))
ts = []
for rep in range(6):
    if rep == 5:
        os.environ["IROTAVG_ROTAVG_TIMING"] = "1"
        os.environ["IROTAVG_BUILD_TIMING"] = "1"
        os.environ["IROTAVG_PCG_TRACE"] = "1"
    t = time.perf_counter()
    info = vg.rotAvg(5000000)
    ts.append(1e3 * (time.perf_counter() - t))
print("global rotAvg at %d views: ms per call %s; info %s" % (n, np.round(ts, 2), info))

#!/usr/bin/env python3
"""Per-phase timing of the one-shot drop-in call (irotavg_irls from host buffers) at 100k/2M:
IROTAVG_BUILD_TIMING=1 prints the build's phases on stderr."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irotavg_amd import capi, ral, synth
n, m = 100000, 2000000
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
S = synth.make_graph(n, m, p, seed=0)
Q0 = np.zeros((n, 4)); Q0[:, 3] = 1; Q0[0] = S["Qgt"][0]
ral.init_mst(Q0, S["QQ"], S["I"], 1)
SIG = 5 * np.pi / 180
for rep in range(4):
    if rep == 3:
        os.environ["IROTAVG_BUILD_TIMING"] = "1"
    Qh = Q0.copy(); wh = np.zeros(m)
    t = time.perf_counter()
    it, rt = ral.irls(S["QQ"], S["I"], None, 4, SIG, Qh, 1, 100, 1e-3, wh)
    dt = time.perf_counter() - t
    print("call %d: %.2f ms total, irls runtime %.2f ms, iters %d" % (rep, 1e3 * dt, 1e3 * rt, it), flush=True)
os.environ.pop("IROTAVG_BUILD_TIMING")
t = time.perf_counter()
G = capi.Graph(S["I"], S["QQ"], n, 1)
t1 = time.perf_counter()
G.set_rotations(Q0)
t2 = time.perf_counter()
r = G.irls(4, SIG, 100, 1e-3)
t3 = time.perf_counter()
Q = G.get_rotations(); w = G.get_weights()
t4 = time.perf_counter()
G.close()
t5 = time.perf_counter()
print("create %.2f set %.2f irls %.2f get %.2f destroy %.2f ms" % tuple(1e3 * x for x in (t1 - t, t2 - t1, t3 - t2, t4 - t3, t5 - t4)))
# the C call alone, on arrays that already have the reference's layout (column-major Mat, int32 pairs)
import ctypes as C
QQf, Ie = capi.fmat(S["QQ"]), capi.edges(S["I"])
for rep in range(3):
    Qf = capi.fmat(Q0); w = np.zeros(m)
    it, rt = C.c_int(0), C.c_double(0)
    t = time.perf_counter()
    rc = capi.lib().irotavg_irls(m, n, 1, capi._i(Ie), capi._d(QQf), m, 4, SIG, capi._d(Qf), n, 100, 1e-3, capi._d(w), C.byref(it), C.byref(rt))
    dt = time.perf_counter() - t
    print("C call alone: %.2f ms (rc %d, iters %d, irls %.2f ms)" % (1e3 * dt, rc, it.value, 1e3 * rt.value))
t = time.perf_counter(); x = capi.fmat(S["QQ"]); print("fmat(QQ) %.2f ms" % (1e3 * (time.perf_counter() - t)))

"""Where a one-shot irotavg_irls call (host pointers in, host pointers out) spends its time at the headline size:
run with IROTAVG_BUILD_TIMING=1 for the laps of the device build."""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from irotavg_amd import capi, ral, synth  # noqa: E402

n, m = 100000, 2000000
S = synth.make_graph(n, m, 0.0, seed=0)
Q0 = np.zeros((n, 4)); Q0[:, 3] = 1; Q0[0] = S["Qgt"][0]
ral.init_mst(Q0, S["QQ"], S["I"], 1)
Ie, QQf = capi.edges(S["I"]), capi.fmat(S["QQ"])
SIG = 5 * np.pi / 180
for rep in range(4):
    Qf, wh = capi.fmat(Q0), np.zeros(m)
    it_c, rt_c = C.c_int(0), C.c_double(0)
    t = time.perf_counter()
    rc = capi.lib().irotavg_irls(m, n, 1, capi._i(Ie), capi._d(QQf), m, 4, SIG, capi._d(Qf), n, 100, 1e-3, capi._d(wh),
                                 C.byref(it_c), C.byref(rt_c))
    print("call %d: rc %d, %.3f ms (irls inside %.3f ms, %d iterations)"
          % (rep, rc, 1e3 * (time.perf_counter() - t), 1e3 * rt_c.value, it_c.value), flush=True)

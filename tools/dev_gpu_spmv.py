import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from irotavg_amd import capi, ral, synth
n, m = 100000, 2000000
S = synth.make_graph(n, m, 0.0, seed=0)
Q = np.zeros((n, 4)); Q[:, 3] = 1; Q[0] = S["Qgt"][0]
ral.init_mst(Q, S["QQ"], S["I"], 1)
G = capi.Graph(S["I"], S["QQ"], n, 1)
G.set_rotations(Q)
if not os.environ.get("IROTAVG_SPMV_PROBE"):
    G.irls(4, 5*np.pi/180, 1, 1e-3)
print("probe", os.environ.get("IROTAVG_SPMV_PROBE"), "spmv ms", G.time_kernel(4, 100), flush=True)

"""Development aid: per-phase wall-clock stamps of k_cg_apply (one mid-grid workgroup)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from irotavg_amd import capi, synth, ral
n, m = 100000, 2000000
S = synth.make_graph(n, m, 0.0, seed=0)
Q0 = np.zeros((n, 4)); Q0[:, 3] = 1; Q0[0] = S["Qgt"][0]
ral.init_mst(Q0, S["QQ"], S["I"], 1)
G = capi.Graph(S["I"], S["QQ"], n, 1)
G.set_rotations(Q0)
G.irls(4, 5 * np.pi / 180, 100, 1e-3)
names = ["entry", "loads issued", "pcg_check done", "sb staged+barrier", "dense slice done", "x1' done", "y1 done",
         "u window done", "spmv done", "reduced+stored", "dense fma done", "matrix/L1 loads issued"]
for rep in range(2):
    print(" | ".join("%s %.2f" % (names[k], G.time_kernel(100 + k, 1)) for k in (1, 2, 3, 10, 11, 4, 5, 6, 7, 8, 9)))
print("grid: first workgroup start %.2f, last workgroup end %.2f (us, relative to the stamped workgroup's entry)" % (G.time_kernel(112, 1), G.time_kernel(113, 1)))
print("apply us", 1e3 * G.time_kernel(9, 50), "update us", 1e3 * G.time_kernel(10, 50))

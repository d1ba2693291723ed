import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ctypes as C
from irotavg_amd import capi, synth, ral
n, m = int(sys.argv[1]), int(sys.argv[2])
G0 = synth.make_graph(n, m, 0.0, seed=0)
Q0 = np.zeros((n, 4)); Q0[:, 3] = 1; Q0[0] = G0["Qgt"][0]
ral.init_mst(Q0, G0["QQ"], G0["I"], 1)
lib = capi.lib()
with capi.Graph(G0["I"], G0["QQ"], n, 1) as G:
    G.set_rotations(Q0); G.edge_residual(); G.ls_solve()
    ms = C.c_double(0)
    for dbg in [0, 1, 2, 4, 6, 7]:
        os.environ["IROTAVG_BCR_DBG"] = str(dbg)
        parts = []
        for which in [19] + list(range(20, 26)):
            if lib.irotavg_graph_time_kernel(G._h, which, 50, C.byref(ms)) == 0:
                parts.append("%d:%.1f" % (which, 1e3 * ms.value))
        print("dbg=%d" % dbg, " ".join(parts), flush=True)

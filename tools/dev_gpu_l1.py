import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from irotavg_amd import capi, graphio, synth
from oracle import oracle as O
g = graphio.read_ravg_input(os.path.join(ROOT, "tests/golden/ravg_input.txt"))
rc, Q0 = O.init_mst(g["Q"], g["QQ"], g["I"], 1)
G = capi.Graph(g["I"], g["QQ"], g["n"], 1)
ro = O.log_map(O.delta_rel(g["I"], g["QQ"], Q0))[:, :3]
for c in range(3):
    x, stuck = G.l1decode_pd(np.ascontiguousarray(ro[:, c]), 2)
    rc, xo, so = O.l1decode_pd(g["n"], 1, g["I"], ro[:, c], 2)
    print("pd coord", c, "diff", np.abs(x - xo).max(), "scale", np.abs(xo).max(), stuck, so, G.stats()["pcg_iters_last"], flush=True)
G.set_rotations(Q0)
a = G.l1ra(5, 1e-3); ra = O.l1ra(g["QQ"], g["I"], Q0, 1, 5, 1e-3)
print("l1ra", a, ra["iters"], ra["scores"], "ang", synth.angular_distance(G.get_rotations(), ra["Q"]).max(), flush=True)
G.close()
S = synth.make_graph(2000, 20000, 0.1, seed=5)
n = 2000
Q = np.zeros((n, 4)); Q[:, 3] = 1; Q[0] = S["Qgt"][0]
rc, Qm = O.init_mst(Q, S["QQ"], S["I"], 1)
G = capi.Graph(S["I"], S["QQ"], n, 1); G.set_rotations(Qm)
t=time.time(); a = G.l1ra(5, 1e-3); t1=time.time()-t
ra = O.l1ra(S["QQ"], S["I"], Qm, 1, 5, 1e-3)
print("l1ra synth", a, t1, ra["iters"], ra["scores"], ra["runtime"], "ang", synth.angular_distance(G.get_rotations(), ra["Q"]).max(), G.stats()["pcg_iters"], flush=True)
b = G.irls(4, 5*np.pi/180, 50, 1e-3); rb = O.irls(S["QQ"], S["I"], ra["Q"], 1, 4, 5*np.pi/180, 50, 1e-3)
print("irls synth", b, rb["iters"], rb["scores"], "ang", synth.angular_distance(G.get_rotations(), rb["Q"]).max(), "w", np.abs(G.get_weights()-rb["weights"]).max())
G.close()
import __graft_entry__ as ge
ge.smoke()
for (n, m, pl) in [(100000, 2000000, 0.0), (100000, 2000000, 0.02)]:
    S = synth.make_graph(n, m, pl, seed=0)
    Q = np.zeros((n, 4)); Q[:, 3] = 1; Q[0] = S["Qgt"][0]
    rc, Qm = O.init_mst(Q, S["QQ"], S["I"], 1)
    G = capi.Graph(S["I"], S["QQ"], n, 1); G.set_rotations(Qm)
    t=time.time(); a = G.l1ra(5, 1e-3, allow_rc=(capi.ERR_NOT_CONVERGED,)); t1=time.time()-t
    print("l1ra C2", pl, a, t1, G.stats()["pcg_iters"], G.stats()["pcg_solves"], flush=True)
    G.close()

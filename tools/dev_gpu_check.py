"""Developer smoke on the GPU box: parity of each stage vs the oracle + first timings."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from irotavg_amd import capi, graphio, synth
from oracle import oracle as O

print(capi.lib().irotavg_version(), "devices", capi.lib().irotavg_device_count(), flush=True)
g = graphio.read_ravg_input(os.path.join(ROOT, "tests/golden/ravg_input.txt"))
rc, Q0 = O.init_mst(g["Q"], g["QQ"], g["I"], 1)
G = capi.Graph(g["I"], g["QQ"], g["n"], 1)
print("stats", G.stats(), flush=True)
G.set_rotations(Q0)
G.edge_residual()
r = G.get_residuals()
ro = O.log_map(O.delta_rel(g["I"], g["QQ"], Q0))[:, :3]
print("K1 max abs diff", np.abs(r - ro).max(), flush=True)
X = G.ls_solve()
rc, Xo = O.ls_solve(g["n"], 1, g["I"], np.ones(g["m"]), ro)
print("LS max abs diff", np.abs(X - Xo).max(), "scale", np.abs(Xo).max(), G.stats()["pcg_iters_last"], G.stats()["last_relres"], flush=True)
G.set_rotations(Q0)
res = G.irls(4, 5 * np.pi / 180, 50, 1e-3)
ref = O.irls(g["QQ"], g["I"], Q0, 1, 4, 5 * np.pi / 180, 50, 1e-3)
print("IRLS iters", res["iters"], ref["iters"], res["scores"], ref["scores"])
print("IRLS ang", synth.angular_distance(G.get_rotations(), ref["Q"]).max(), "w", np.abs(G.get_weights() - ref["weights"]).max(), flush=True)
G.close()

for (n, m, pl) in [(10000, 150000, 0.0), (10000, 150000, 0.02), (100000, 2000000, 0.0), (100000, 2000000, 0.02)]:
    S = synth.make_graph(n, m, pl, seed=0)
    Q = np.zeros((n, 4)); Q[:, 3] = 1; Q[0] = S["Qgt"][0]
    t = time.time(); rc, Qm = O.init_mst(Q, S["QQ"], S["I"], 1); t_mst = time.time() - t
    t = time.time(); G = capi.Graph(S["I"], S["QQ"], n, 1); t_create = time.time() - t
    G.set_rotations(Qm)
    t = time.time(); res = G.irls(4, 5 * np.pi / 180, 100, 1e-3, allow_rc=(capi.ERR_NOT_CONVERGED,)); t_irls = time.time() - t
    st = G.stats()
    err = synth.angular_distance(G.get_rotations(), S["Qgt"])
    print(json.dumps(dict(n=n, m=S["m"], p_loop=pl, t_mst=t_mst, t_create=t_create, t_irls=t_irls, rc=res["rc"], iters=res["iters"], scores=list(res["scores"]), pcg_iters=st["pcg_iters"], levels=st["level_rows"], nnz=st["level_nnz"], relres=st["last_relres"], err_mean=float(err.mean()), eups=S["m"] * res["iters"] / res["runtime"])), flush=True)
    for which in (1, 2, 3, 4, 5, 6):
        print("  kernel", which, "ms", G.time_kernel(which, 20), flush=True)
    G.close()

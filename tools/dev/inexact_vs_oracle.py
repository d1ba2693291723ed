"""options.inexact_outer against the ORACLE at the sizes the general-topology leg is quoted on: iteration counts, mean / max
angle of the final rotations, time per irls call -- the measurement behind the default of the option."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from irotavg_amd import capi, ral, synth
from oracle import oracle as O
SIG = 5 * np.pi / 180
for n, m in ((100000, 2000000), (10000, 150000)):
    S = synth.make_graph(n, m, 0.02, seed=0)
    Q0 = np.zeros((n, 4)); Q0[:, 3] = 1; Q0[0] = S["Qgt"][0]
    ral.init_mst(Q0, S["QQ"], S["I"], 1)
    ro = O.irls(S["QQ"], S["I"], Q0, 1, 4, SIG, 100, 1e-3)
    out = {"n": n, "m": m, "oracle_iters": int(ro["iters"])}
    Qs = {}
    for inexact in (0, 1):
        with capi.Graph(S["I"], S["QQ"], n, 1, inexact_outer=inexact) as G:
            G.set_rotations(Q0); G.snapshot_rotations()
            for rep in range(6):
                if rep == 1:
                    G.synchronize(); G.reset_stats(); t0 = time.perf_counter()
                G.restore_rotations()
                r = G.irls(4, SIG, 100, 1e-3)
            G.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / 5
            Q = G.get_rotations(); st = G.stats()
        ang = synth.angular_distance(Q, ro["Q"])
        Qs[inexact] = Q
        out["inexact_%d" % inexact] = dict(iters=int(r["iters"]), ms=ms, G_edge_updates_per_s=m * r["iters"] / ms / 1e6,
                                           mean_rad_vs_oracle=float(ang.mean()), max_rad_vs_oracle=float(ang.max()),
                                           pcg_iters_per_solve=st["pcg_iters"] / max(1, st["pcg_solves"]),
                                           score_rel_err=[float(abs(a - b) / b) for a, b in zip(r["scores"], ro["scores"])])
    a = synth.angular_distance(Qs[0], Qs[1])
    out["inexact_vs_exact_gpu"] = dict(mean=float(a.mean()), max=float(a.max()))
    print(json.dumps(out), flush=True)

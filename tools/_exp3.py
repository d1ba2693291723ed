import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irotavg_amd import capi, synth, ral
SIG = 5 * np.pi / 180
n, m = 100000, 2000000
for pl in (0.0, 0.02):
    S = synth.make_graph(n, m, pl, seed=0)
    Q0 = np.zeros((n, 4)); Q0[:, 3] = 1; Q0[:1] = S["Qgt"][:1]
    ral.init_mst(Q0, S["QQ"], S["I"], 1)
    res = {}
    for classic in (0, 1):
        if classic: os.environ["IROTAVG_ASM_CLASSIC"] = "1"
        else: os.environ.pop("IROTAVG_ASM_CLASSIC", None)
        with capi.Graph(S["I"], S["QQ"], n, 1) as G:
            ts = []
            for rep in range(4):
                G.set_rotations(Q0)
                t = time.perf_counter()
                r = G.irls(4, SIG, 100, 1e-3)
                ts.append(time.perf_counter() - t)
            st = G.stats()
            res[classic] = G.get_rotations()
            print("p_loop", pl, "classic" if classic else "windowed", "iters", r["iters"], "ms", [round(1e3 * x, 2) for x in ts], "asm us", 1e3 * G.time_kernel(3, 50), {k: st[k] for k in ("pcg_iters", "dense_inversions")}, r["scores"], flush=True)
    print("  max angle between the two", synth.angular_distance(res[0], res[1]).max())

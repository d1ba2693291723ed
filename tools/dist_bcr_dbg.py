import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from irotavg_amd import capi, synth, ral
SIG = 5 * np.pi / 180
n, m, f, world = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
S = synth.make_graph(n, m, 0.0, seed=0)
Q0 = np.zeros((n, 4)); Q0[:, 3] = 1; Q0[:f] = S["Qgt"][:f]
ral.init_mst(Q0, S["QQ"], S["I"], f)
with capi.Graph(S["I"], S["QQ"], n, f) as G:
    G.set_rotations(Q0); a = G.irls(4, SIG, 1, 1e-3); Qa = G.get_rotations()
with capi.DistGraph(S["I"], S["QQ"], n, f, world) as D:
    print("info", D.info())
    D.set_rotations(Q0); b = D.irls(4, SIG, 1, 1e-3); Qb = D.get_rotations(into=Q0.copy())
ang = synth.angular_distance(Qa, Qb)
bad = np.flatnonzero(ang > 1e-9)
print("scores", a["scores"], b["scores"], "max angle %.3e, views off %d" % (ang.max(), len(bad)))
if len(bad):
    print("first/last off", bad[:10], bad[-10:])
    # histogram by shard
    nu = n - f
    chunk = ((nu + world - 1) // world + 191) // 192 * 192
    print("chunk", chunk, "off per shard", np.bincount((bad - f) // chunk, minlength=world))
    for r in range(world):
        sel = bad[(bad - f) // chunk == r] - f - r * chunk
        if len(sel): print("  shard", r, "local rows off: min", sel.min(), "max", sel.max(), "count", len(sel), "max ang", ang[bad[(bad - f) // chunk == r]].max())

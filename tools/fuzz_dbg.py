import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fuzz_band_direct import make_case, SIG
from irotavg_amd import capi, synth
from oracle import oracle as O
seed, want = int(sys.argv[1]), [int(x) for x in sys.argv[2:]]
rng = np.random.default_rng(seed)
for case in range(max(want) + 1):
    n, f, I, QQ, Q0 = make_case(rng)
    cost = int(rng.integers(0, 14)); l1 = int(rng.choice([0, 1, 2]))
    if case not in want:
        continue
    ra = O.l1ra(QQ, I, Q0, f, l1, 1e-3) if l1 else dict(rc=0, Q=Q0, iters=0)
    rb = O.irls(QQ, I, ra["Q"], f, cost, SIG, 15, 1e-3)
    # connectivity of the free graph + fixed
    import scipy.sparse as sp, scipy.sparse.csgraph as cg
    A = sp.coo_matrix((np.ones(len(I)), (I[:, 0], I[:, 1])), shape=(n, n))
    # merge all fixed views into one node
    lab = np.arange(n); lab[:f] = 0
    A2 = sp.coo_matrix((np.ones(len(I)), (lab[I[:, 0]], lab[I[:, 1]])), shape=(n, n))
    nc, comp = cg.connected_components(A2, directed=False)
    deg = np.bincount(I.ravel(), minlength=n)
    print("case %d: n %d f %d m %d cost %d l1 %d: components (fixed merged) %d (isolated views %d)" % (case, n, f, len(I), cost, l1, nc - (f - 1), (deg == 0).sum()))
    res = {}
    for bd in (1, -1):
        with capi.Graph(I, QQ, n, f, band_direct=bd) as G:
            G.set_rotations(Q0)
            ga = G.l1ra(l1, 1e-3) if l1 else dict(iters=0)
            gb = G.irls(cost, SIG, 15, 1e-3)
            res[bd] = (ga["iters"], gb["iters"], G.get_rotations(), gb["scores"], G.stats()["band_block"], G.direct_info())
    for bd in (1, -1):
        ang = synth.angular_distance(res[bd][2], rb["Q"])
        badv = np.flatnonzero(ang > 1e-6)
        print("   band_direct %2d: iters %s oracle %s; max angle %.2e; views off: %d (first %s) comps of those %s; block %d info %s" % (
            bd, res[bd][:2], (ra["iters"], rb["iters"]), ang.max(), len(badv), badv[:8], np.unique(comp[lab[badv]])[:6], res[bd][4], str(res[bd][5])[:150]))
        print("      scores gpu", res[bd][3][:6], "oracle", rb["scores"][:6])

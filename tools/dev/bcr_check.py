"""Development aid: the banded direct solver (bcr.hip) against the oracle and against the PCG path, and its timing."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from irotavg_amd import capi, synth
from oracle import oracle as O

SIG = 5 * np.pi / 180


def mst_init(G, n, f=1):
    Q = np.zeros((n, 4)); Q[:, 3] = 1; Q[:f] = G["Qgt"][:f]
    rc, Qm = O.init_mst(Q, G["QQ"], G["I"], f)
    assert rc == 0
    return Qm


def small(n, m, pbo=0.0):
    G0 = synth.make_graph(n, m, 0.0, seed=3, p_band_out=pbo)
    Qm = mst_init(G0, n)
    rng = np.random.default_rng(0)
    w = rng.uniform(0.1, 5.0, size=len(G0["I"]))
    w[rng.choice(len(w), len(w) // 40, replace=False)] *= 1e-4
    ro = O.log_map(O.delta_rel(G0["I"], G0["QQ"], Qm))[:, :3]
    rc, Xo = O.ls_solve(n, 1, G0["I"], w, ro)
    out = {}
    for bd in (1, -1):
        with capi.Graph(G0["I"], G0["QQ"], n, 1, band_direct=bd) as G:
            G.set_rotations(Qm)
            G.edge_residual()
            G.set_weights(w)
            X = G.ls_solve()
            st = G.stats()
            G.set_rotations(Qm)
            r = G.irls(4, SIG, 50, 1e-3)
            Qg = G.get_rotations()
            G.set_rotations(Qm)
            r1 = G.l1ra(2, 1e-3)
            Ql = G.get_rotations()
        out[bd] = (X, st, r, Qg, r1, Ql)
        print("n=%d m=%d band_direct=%2d band=%d B=%d direct=%d pcg=%d  ls err %.2e  irls iters %d  l1 iters %d" % (
            n, m, bd, st["band"], st["band_block"], st["direct_solves"], st["pcg_solves"],
            np.abs(X - Xo).max() / np.abs(Xo).max(), r["iters"], r1["iters"]), flush=True)
    res = O.irls(G0["QQ"], G0["I"], Qm.copy(), 1, cost=4, sigma=SIG, max_iters=50, change_th=1e-3)
    print("   oracle irls:", res["iters"], " max angle bcr vs oracle %.2e  pcg vs oracle %.2e  scores bcr %s oracle %s" % (
        synth.angular_distance(out[1][3], res["Q"]).max(), synth.angular_distance(out[-1][3], res["Q"]).max(),
        out[1][2]["scores"], res["scores"]), flush=True)
    l1 = O.l1ra(G0["QQ"], G0["I"], Qm.copy(), 1, max_iters=2, change_th=1e-3)
    print("   oracle l1ra:", l1["iters"], " max angle bcr vs oracle %.2e  pcg vs oracle %.2e" % (
        synth.angular_distance(out[1][5], l1["Q"]).max(), synth.angular_distance(out[-1][5], l1["Q"]).max()), flush=True)


def big(n, m, reps=10, pbo=0.0):
    G0 = synth.make_graph(n, m, 0.0, seed=0, p_band_out=pbo)
    Q0 = np.zeros((n, 4)); Q0[:, 3] = 1; Q0[0] = G0["Qgt"][0]
    lib = capi.lib()
    import ctypes as C
    Qm = capi.fmat(Q0)
    rc = lib.irotavg_init_mst(n, len(G0["I"]), Qm.ctypes.data_as(capi._dp), n, capi.fmat(G0["QQ"]).ctypes.data_as(capi._dp),
                              len(G0["I"]), capi.edges(G0["I"]).ctypes.data_as(capi._ip), 1)
    assert rc == 0
    res = {}
    for bd in (0, -1):
        with capi.Graph(G0["I"], G0["QQ"], n, 1, band_direct=bd) as G:
            G.set_rotations(Qm); G.snapshot_rotations()
            ts = []
            for k in range(reps):
                G.restore_rotations()
                r = G.irls(4, SIG, 50, 1e-3)
                ts.append(r["runtime"])
            Qg = G.get_rotations()
            st = G.stats()
            G.restore_rotations()
            t0 = time.time(); r1 = G.l1ra(5, 1e-3); t1 = time.time() - t0
            G.restore_rotations()
            t0 = time.time(); r1 = G.l1ra(5, 1e-3); t1 = time.time() - t0
            if st["band_block"]:
                G.restore_rotations(); G.edge_residual(); G.ls_solve()
                ms = C.c_double(0)
                parts = []
                for which in [1, 3, 2, 19] + list(range(20, 30)) + list(range(40, 50)):
                    if lib.irotavg_graph_time_kernel(G._h, which, 50, C.byref(ms)) == 0:
                        parts.append("%d:%.1f" % (which, 1e3 * ms.value))
                print("   kernel us:", " ".join(parts), flush=True)
        res[bd] = Qg
        err = synth.angular_distance(Qg, G0["Qgt"] if False else Qg).max()
        print("n=%d m=%d pbo=%g band_direct=%2d B=%d: irls iters %d, ms/solve min %.3f med %.3f  -> %.3f G edge-updates/s; l1ra(5) %.2f ms (%d iters)" % (
            n, m, pbo, bd, st["band_block"], r["iters"], 1e3 * min(ts), 1e3 * np.median(ts),
            len(G0["I"]) * r["iters"] / np.median(ts) / 1e9, 1e3 * t1, r1["iters"]), flush=True)
    print("   bcr vs pcg: max angle %.2e rad" % synth.angular_distance(res[0], res[-1]).max(), flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("small", "all"):
        small(3000, 12000)
        small(3000, 45000)
        small(3000, 63000, 0.02)
        small(3000, 90000)
        small(2500, 50000)
    if what in ("big", "all"):
        big(100000, 2000000)
        big(100000, 2000000, pbo=0.02)
        big(75000, 300000)
        big(10000, 150000)

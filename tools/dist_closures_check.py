"""Development check: closures on the sharded direct solver (loopback shards on one GPU) against the unsharded handle
and the oracle. usage: python tools/dist_closures_check.py [n m nclose wrong world]..."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from irotavg_amd import capi, ral, synth
from oracle import oracle as O
sys.path.insert(0, "tests")
from test_gpu_band_direct import closure_graph

SIG = 5 * np.pi / 180


def run(n, m, nclose, wrong, world, cost=4, with_oracle=True, l1=True):
    S = closure_graph(n, m, nclose, 7, wrong)
    Q = np.zeros((n, 4)); Q[:, 3] = 1; Q[0] = S["Qgt"][0]
    ral.init_mst(Q, S["QQ"], S["I"], 1)
    with capi.Graph(S["I"], S["QQ"], n, 1, band_direct=1) as G:
        G.set_rotations(Q)
        a1 = G.l1ra(2, 1e-3) if l1 else None
        a = G.irls(cost, SIG, 50, 1e-3)
        Qa, wa = G.get_rotations(), G.get_weights()
    with capi.DistGraph(S["I"], S["QQ"], n, 1, world, band_direct=1) as D:
        info = D.info()
        D.set_rotations(Q)
        b1 = D.l1ra(2, 1e-3) if l1 else None
        t0 = time.time()
        b = D.irls(cost, SIG, 50, 1e-3)
        t1 = time.time()
        Qb, wb = D.get_rotations(into=Q.copy()), D.get_weights()
        st = D.stats()
    print("n %d m %d closures %d (wrong %d) world %d: block %s iters %s / %s  l1 %s / %s  direct %d pcg %d  irls %.2f ms"
          % (n, S["m"], nclose, wrong, world, info["direct_block"], a["iters"], b["iters"],
             a1 and a1["iters"], b1 and b1["iters"], st["direct_solves"], st["pcg_iters"], 1e3 * (t1 - t0)))
    print("   scores", np.array(a["scores"]), np.array(b["scores"]))
    print("   max angular diff sharded vs unsharded %.3e  weights %.3e" %
          (synth.angular_distance(Qa, Qb).max(), np.abs(wa - wb).max()))
    if with_oracle:
        r = {"Q": Q}
        if l1:
            r = O.l1ra(S["QQ"], S["I"], Q, 1, 2, 1e-3)
        rb = O.irls(S["QQ"], S["I"], r["Q"], 1, cost, SIG, 50, 1e-3)
        print("   oracle iters %d  max angular diff sharded vs oracle %.3e  unsharded vs oracle %.3e" %
              (rb["iters"], synth.angular_distance(Qb, rb["Q"]).max(), synth.angular_distance(Qa, rb["Q"]).max()))


if __name__ == "__main__":
    no_oracle = "--no-oracle" in sys.argv
    args = [int(x) for x in sys.argv[1:] if not x.startswith("--")]
    if not args:
        cases = [(3000, 12000, 5, 1, 2), (3000, 45000, 20, 3, 4), (4000, 80000, 40, 4, 3), (6000, 60000, 300, 20, 8),
                 (20000, 300000, 1000, 40, 8)]
    else:
        cases = [tuple(args[i:i + 5]) for i in range(0, len(args), 5)]
    for c in cases:
        run(*c, with_oracle=not no_oracle)

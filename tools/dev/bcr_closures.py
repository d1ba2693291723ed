"""Development aid: band graphs with a few loop closures on the direct solver (Woodbury) against the oracle / the PCG."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from irotavg_amd import capi, synth, ral
from oracle import oracle as O
SIG = 5 * np.pi / 180


def graph(n, m, nclose, seed, wrong=0):
    return synth.add_closures(synth.make_graph(n, m, 0.0, seed=seed), nclose, seed, wrong)


def run(n, m, nclose, oracle=True, wrong=0, reps=3):
    S = graph(n, m, nclose, 7, wrong)
    Q0 = np.zeros((n, 4)); Q0[:, 3] = 1; Q0[0] = S["Qgt"][0]
    ral.init_mst(Q0, S["QQ"], S["I"], 1)
    out = {}
    for bd in (1, -1):
        with capi.Graph(S["I"], S["QQ"], n, 1, band_direct=bd) as G:
            G.set_rotations(Q0); G.snapshot_rotations()
            for _ in range(reps):
                G.restore_rotations()
                t0 = time.perf_counter(); a = G.l1ra(2, 1e-3); G.synchronize(); t1 = time.perf_counter()
                b = G.irls(4, SIG, 50, 1e-3); G.synchronize(); t2 = time.perf_counter()
            st = G.stats()
            out[bd] = (a, b, G.get_rotations(), G.get_weights())
            print("n=%d m=%d closures=%d band_direct=%2d: block %d direct %d pcg %d  l1ra %d it %.2f ms  irls %d it %.2f ms" % (
                n, S["m"], nclose, bd, st["band_block"], st["direct_solves"], st["pcg_solves"], a["iters"], 1e3 * (t1 - t0),
                b["iters"], 1e3 * (t2 - t1)), flush=True)
    print("   direct vs pcg: max angle %.2e  iters %s vs %s" % (synth.angular_distance(out[1][2], out[-1][2]).max(),
          (out[1][0]["iters"], out[1][1]["iters"]), (out[-1][0]["iters"], out[-1][1]["iters"])), flush=True)
    if oracle:
        ra = O.l1ra(S["QQ"], S["I"], Q0, 1, 2, 1e-3)
        rb = O.irls(S["QQ"], S["I"], ra["Q"], 1, 4, SIG, 50, 1e-3)
        print("   oracle iters %s; direct vs oracle %.2e rad, weights %.2e; pcg vs oracle %.2e" % (
            (ra["iters"], rb["iters"]), synth.angular_distance(out[1][2], rb["Q"]).max(),
            np.abs(out[1][3] / rb["weights"] - 1).max(), synth.angular_distance(out[-1][2], rb["Q"]).max()), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "many":   # round 4: hundreds of closures (the oracle up to 20k views)
        run(3000, 45000, 100, wrong=5)
        run(20000, 300000, 300, wrong=10)
        run(20000, 300000, 1000, wrong=30)
        run(100000, 2000000, 100, oracle=False, wrong=5)
        run(100000, 2000000, 1000, oracle=False, wrong=30)
        sys.exit(0)
    run(3000, 12000, 5)
    run(3000, 45000, 20, wrong=3)
    run(4000, 80000, 40, wrong=4)
    run(20000, 300000, 17, wrong=2)
    run(75000, 300000, 8, oracle=False, wrong=1)
    run(100000, 2000000, 12, oracle=False, wrong=3)

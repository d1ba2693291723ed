#!/usr/bin/env python3
"""One case of tools/fuzz_parity.py in detail (the campaign's random stream is replayed up to it): iteration scores of
l1ra and irls from the handle and from the oracle, per-stage angular distances.

    python tools/fuzz_case.py --seed 301 --case 386 [--lib path/to/another/libirotavg_hip.so]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, required=True)
    ap.add_argument("--case", type=int, required=True)
    ap.add_argument("--nmax", type=int, default=400)
    ap.add_argument("--small-share", type=float, default=0.4)
    ap.add_argument("--lib", default=None)
    a = ap.parse_args()
    from irotavg_amd import capi, synth
    if a.lib:
        capi.LIB_PATH = os.path.abspath(a.lib)
    import fuzz_parity as F
    from oracle import oracle as O
    rng = np.random.default_rng(a.seed)
    for k in range(a.case + 1):
        c = F.random_case(rng, 20 if rng.random() < a.small_share else a.nmax)
    n, f, I, QQ, Q0, cost = c["n"], c["f"], c["I"], c["QQ"], c["Q0"], c["cost"]
    sig = 5 * np.pi / 180
    print("case %d: n %d f %d m %d cost %d" % (a.case, n, f, len(I), cost))
    for l1_iters in (1, 2, 3):
        ra = O.l1ra(QQ, I, Q0, f, l1_iters, 1e-3)
        with capi.Graph(I, QQ, n, f) as G:
            G.set_rotations(Q0)
            ga = G.l1ra(l1_iters, 1e-3)
            Qa = G.get_rotations()
            st = G.stats()
        print("l1ra(%d): iters %d vs %d, scores %s vs %s, angle %.3e (direct solves %d, pcg solves %d)"
              % (l1_iters, ga["iters"], ra["iters"], np.array2string(np.asarray(ga["scores"]), precision=6),
                 np.array2string(np.asarray(ra["scores"]), precision=6), synth.angular_distance(Qa, ra["Q"]).max(),
                 st["direct_solves"], st["pcg_solves"]))
    rb = O.irls(QQ, I, ra["Q"], f, cost, sig, 15, 1e-3)
    with capi.Graph(I, QQ, n, f) as G:
        G.set_rotations(ra["Q"])
        gb = G.irls(cost, sig, 15, 1e-3)
        Qb = G.get_rotations()
    print("irls from the oracle's l1ra result: iters %d vs %d, angle %.3e" % (gb["iters"], rb["iters"],
                                                                               synth.angular_distance(Qb, rb["Q"]).max()))
    print("scores", np.asarray(gb["scores"]), np.asarray(rb["scores"]))


if __name__ == "__main__":
    main()

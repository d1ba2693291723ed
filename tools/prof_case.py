#!/usr/bin/env python3
"""One workload under rocprofv3: `irls` (or `l1ra`) repeated on a synthetic graph, nothing else in the process.
    python tools/prof_case.py --views 100000 --edges 2000000 --p-loop 0.02 --what irls --reps 5"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irotavg_amd import capi, ral, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--views", type=int, default=100000)
ap.add_argument("--edges", type=int, default=2000000)
ap.add_argument("--p-loop", type=float, default=0.0)
ap.add_argument("--band-outliers", type=float, default=0.0)
ap.add_argument("--what", default="irls")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--l1-iters", type=int, default=5)
ap.add_argument("--classic", type=int, default=0)
ap.add_argument("--dense-max", type=int, default=0)
ap.add_argument("--agg", type=int, default=0)
ap.add_argument("--agg0", type=int, default=0)
ap.add_argument("--kc", type=float, default=0.0)
ap.add_argument("--omega", type=float, default=0.0)
ap.add_argument("--fixed-every", type=int, default=0)
ap.add_argument("--band-direct", type=int, default=0)
a = ap.parse_args()
S = synth.make_graph(a.views, a.edges, a.p_loop, seed=0)
Q0 = np.zeros((a.views, 4)); Q0[:, 3] = 1; Q0[0] = S["Qgt"][0]
ral.init_mst(Q0, S["QQ"], S["I"], 1)
SIG = 5 * np.pi / 180
with capi.Graph(S["I"], S["QQ"], a.views, 1, pcg_classic=a.classic, mg_dense_max=a.dense_max, mg_agg=a.agg, mg_agg0=a.agg0, mg_kc=a.kc, mg_omega=a.omega, band_direct=a.band_direct) as G:
    G.set_rotations(Q0)
    G.snapshot_rotations()
    out = {}
    for rep in range(a.reps + 2):
        if rep == 2:
            G.synchronize()
            G.reset_stats()
            t0 = time.perf_counter()
        G.restore_rotations()
        if a.what == "irls":
            r = G.irls(4, SIG, 100, 1e-3)
        else:
            r = G.l1ra(a.l1_iters, 1e-3)
    G.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    st = G.stats()
    print(json.dumps(dict(what=a.what, ms=1e3 * dt, iters=r["iters"], pcg_iters_per_solve=st["pcg_iters"] / max(1, st["pcg_solves"]),
                          pcg_solves=st["pcg_solves"] / a.reps, dense_inversions=st["dense_inversions"] / a.reps,
                          dense_repairs=st["dense_repairs"] / a.reps, levels=st["level_rows"], scores=[float(x) for x in r["scores"]])))

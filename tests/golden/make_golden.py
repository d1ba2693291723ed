"""Regenerates the golden outputs under tests/golden/ from the CPU oracle (oracle/liboracle.so).

The reference ships only the INPUT fixture ral/data/ravg_input.txt (copied verbatim as
tests/golden/ravg_input.txt -- it is data, not source) and no expected outputs, and it cannot be
built in this image, so these goldens are ORACLE outputs (parity unpinned, see
oracle/irotavg_oracle.h). They pin the oracle against regressions and carry the sanity values
SURVEY.md 8(c) recorded from an independent throw-away restatement.

Usage: python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from irotavg_amd import graphio, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


def run_pipeline(I, QQ, Q, f, n_abs, cost=4, sigma=5 * np.pi / 180, irls_iters=50, l1_iters=5,
                 th=1e-3):
    """ral/test.cpp:277-302 with its default arguments (:250-272)."""
    rc, Q0 = O.init_mst(Q, QQ, I, max(n_abs, f))
    assert rc == 0
    a = O.l1ra(QQ, I, Q0, f, l1_iters, th)
    b = O.irls(QQ, I, a["Q"], f, cost, sigma, irls_iters, th)
    Qf = O.quat_normalised(b["Q"], f)
    return Q0, a, b, Qf


def main():
    g = graphio.read_ravg_input(os.path.join(HERE, "ravg_input.txt"))
    Q0, a, b, Qf = run_pipeline(g["I"], g["QQ"], g["Q"], g["f"], g["n_abs_read"])
    r0 = np.linalg.norm(O.log_map(O.delta_rel(g["I"], g["QQ"], Q0))[:, :3], axis=1)
    r1 = np.linalg.norm(O.log_map(O.delta_rel(g["I"], g["QQ"], Qf))[:, :3], axis=1)
    meta = dict(m=g["m"], n=g["n"], f=g["f"], l1ra_iters=a["iters"], l1ra_scores=list(a["scores"]),
                irls_iters=b["iters"], irls_scores=list(b["scores"]),
                mst_residual_mean=float(r0.mean()), mst_residual_max=float(r0.max()),
                final_residual_mean=float(r1.mean()), final_residual_max=float(r1.max()),
                weights_min=float(b["weights"].min()), weights_max=float(b["weights"].max()))
    np.savez_compressed(os.path.join(HERE, "fixture_expected.npz"), Q=np.ascontiguousarray(Qf),
                        weights=b["weights"], Q_mst=np.ascontiguousarray(Q0))
    # small seeded synthetic graph, every cost
    S = synth.make_graph(400, 4000, 0.1, seed=11)
    n = 400
    Q = np.zeros((n, 4)); Q[:, 3] = 1; Q[0] = S["Qgt"][0]
    rc, Qm = O.init_mst(Q, S["QQ"], S["I"], 1)
    l1 = O.l1ra(S["QQ"], S["I"], Qm, 1, 5, 1e-3)
    out = dict(I=S["I"], QQ=S["QQ"], Qgt=S["Qgt"], Q_mst=np.ascontiguousarray(Qm),
               Q_l1=np.ascontiguousarray(l1["Q"]))
    meta["synth_l1ra_iters"] = l1["iters"]
    meta["synth_l1ra_scores"] = list(l1["scores"])
    meta["synth_irls"] = {}
    for c in range(14):
        r = O.irls(S["QQ"], S["I"], l1["Q"], 1, c, 5 * np.pi / 180, 20, 1e-3)
        out["Q_cost%d" % c] = np.ascontiguousarray(r["Q"])
        out["w_cost%d" % c] = r["weights"]
        meta["synth_irls"][O.COSTS[c]] = dict(iters=r["iters"], scores=list(r["scores"]))
    np.savez_compressed(os.path.join(HERE, "synth400_expected.npz"), **out)
    with open(os.path.join(HERE, "expected.json"), "w") as fh:
        json.dump(meta, fh, indent=1)
    print(json.dumps(meta, indent=1)[:1500])


if __name__ == "__main__":
    main()

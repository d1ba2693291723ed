"""The cases of the reference-golden recipe (tools/ref_golden/README.md): name -> (input builder,
CLI arguments after the output file: cost sigma[deg] irls_iters l1_iters change_th -- the argument
order of ral/test.cpp:88-132). Shared by make_cases.py (writes the inputs) and
tests/test_ref_golden.py (replays them through the oracle)."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CASES = {
    # the reference's own fixture with its default arguments (ral/test.cpp:250-272)
    "fixture_default": dict(kind="fixture", args=["Geman-McClure", "5", "50", "5", "0.001"]),
    "fixture_L1": dict(kind="fixture", args=["L1", "5", "50", "5", "0.001"]),
    "fixture_Huber": dict(kind="fixture", args=["Huber", "5", "50", "5", "0.001"]),
    # seeded synthetic graphs (irotavg_amd/synth.py): band + loop closures, 5 % outliers among the loops
    "synth400_GM": dict(kind="synth", n=400, m=3000, p_loop=0.2, seed=5,
                        args=["Geman-McClure", "5", "50", "5", "0.001"]),
    "synth400_Cauchy": dict(kind="synth", n=400, m=3000, p_loop=0.2, seed=5,
                            args=["Cauchy", "5", "50", "5", "0.001"]),
    "synth2000_GM": dict(kind="synth", n=2000, m=24000, p_loop=0.05, seed=7,
                         args=["Geman-McClure", "5", "50", "5", "0.001"]),
}


def build_case(name):
    """dict(m, n, f, I, QQ, Q, n_abs_read) in the reader's conventions (irotavg_amd/graphio.py)."""
    import sys
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from irotavg_amd import graphio, synth
    c = CASES[name]
    if c["kind"] == "fixture":
        return graphio.read_ravg_input(os.path.join(ROOT, "tests", "golden", "ravg_input.txt"))
    S = synth.make_graph(c["n"], c["m"], c["p_loop"], seed=c["seed"])
    Q = np.zeros((c["n"], 4))
    Q[:, 3] = 1
    Q[0] = S["Qgt"][0]
    return dict(m=len(S["I"]), n=c["n"], f=1, I=S["I"], QQ=S["QQ"], Q=Q, n_abs_read=1)

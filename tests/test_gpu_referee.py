"""The refereed near-tree cases on the GPU (see tests/test_referee.py for the CPU side and the fixtures).

Round 5's library took 15 IRLS iterations and ended 0.11 rad off on seed 603 case 163 where four exact CPU solves agree on
13 iterations and on the rotations to 1.5e-5 rad: the single dense level was solved to pcg_rtol = 1e-10 and the error
that leaves (cond 4e6) was amplified by an outer iteration that does not contract. Such a level is now solved to the
residual it can attain (solver.hip, pcg_solve_classic: `refine`); these tests hold the handle to the referees.
"""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from irotavg_amd import capi, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu
SIG = 5 * np.pi / 180


def load(name):
    c = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    with open(os.path.join(ROOT, "tests", "golden", name + "_referee.json")) as fh:
        ref = json.load(fh)
    return dict(n=int(c["n"]), f=int(c["f"]), I=c["I"], QQ=c["QQ"], Q0=c["Q0"], cost=int(c["cost"])), ref


def run_gpu(c):
    with capi.Graph(c["I"], c["QQ"], c["n"], c["f"]) as G:
        G.set_rotations(c["Q0"])
        a = G.l1ra(3, 1e-3)
        Qa = G.get_rotations()
        b = G.irls(c["cost"], SIG, 15, 1e-3)
        return a, Qa, b, G.get_rotations(), G.get_weights(), G.stats()


@pytest.mark.parametrize("name,angle_bar", [("neartree_seed603_case163", 3e-5), ("neartree_seed501_case196", 2e-8)])
def test_near_tree_case_matches_the_referees(name, angle_bar):
    c, ref = load(name)
    ra = O.l1ra(c["QQ"], c["I"], c["Q0"], c["f"], 3, 1e-3)
    rb = O.irls(c["QQ"], c["I"], ra["Q"], c["f"], c["cost"], SIG, 15, 1e-3)
    a, Qa, b, Q, w, st = run_gpu(c)
    # the l1ra part is reproducible to round-off
    assert a["iters"] == ra["iters"] and synth.angular_distance(Qa, ra["Q"]).max() < 1e-12
    # the iteration count is every referee's
    assert b["iters"] == rb["iters"] == ref["iters"]["ld"] == 13
    # the score trace is the long-double referee's up to where the fp64 referees themselves leave it (1e-6 relative) ...
    fork = ref["first_fork"]
    upto = b["iters"] if fork < 0 else fork
    np.testing.assert_allclose(np.asarray(b["scores"])[:upto], ref["scores"]["ld"][:upto], rtol=2e-6)
    # ... and to 1e-4 relative over the whole run (round 5: 4e-2 at iteration 12, then two more iterations)
    np.testing.assert_allclose(np.asarray(b["scores"])[:b["iters"]], ref["scores"]["ld"], rtol=1e-4)
    # the rotations: as close to the oracle as the referees are to each other (their spread: the recorded maximum)
    spread = max(ref["final_angle_rad"].values())
    d = synth.angular_distance(Q, rb["Q"]).max()
    assert d < angle_bar and d < 3 * spread + 1e-8, (d, spread)
    assert st["pcg_solves"] > 0 and st["direct_solves"] == 0 and st["levels"] == 1   # the dense single level


def test_a_dense_single_level_is_solved_to_its_attainable_residual():
    """The mechanism itself: on the single dense level the relative residual a solve ends with is far below pcg_rtol
    (it was just below 1e-10), and IROTAVG_NO_DENSE_REFINE=1 gives round 5's behaviour back (15 iterations)."""
    c, ref = load("neartree_seed603_case163")
    *_, st = run_gpu(c)
    assert max(st["last_relres"]) < 1e-12, st["last_relres"]
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); import numpy as np; from tests.test_gpu_referee import load, run_gpu; "
            "c, _ = load('neartree_seed603_case163'); a, Qa, b, *_ = run_gpu(c); print(b['iters'])" % ROOT)
    env = dict(os.environ, IROTAVG_NO_DENSE_REFINE="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert int(r.stdout.strip().splitlines()[-1]) != 13      # the amplified inner tolerance: what the fix removed

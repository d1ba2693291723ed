"""Consumes reference outputs produced by tools/ref_golden/run_reference.sh (the UNMODIFIED reference
demo built with Eigen + SuiteSparse on a machine that has them) when they are committed under
tests/golden/ref_<case>.out. None can be produced in this repository's image (no Eigen, no
SuiteSparse), so without the files every case is SKIPPED -- parity stays "unpinned by the reference"
(oracle/irotavg_oracle.h) and nothing passes vacuously."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tools", "ref_golden"))
from cases import CASES, build_case  # noqa: E402

from irotavg_amd import ral, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


def read_reference_output(path, n, m):
    """l1_irls_out.txt (ral/test.cpp:314-326): n lines `w x y z`, then m weights."""
    vals = np.loadtxt(path, ndmin=1) if False else None
    with open(path) as fh:
        tok = fh.read().split()
    a = np.array(tok, dtype=np.float64)
    assert len(a) == 4 * n + m, "unexpected size of %s" % path
    Qw = a[:4 * n].reshape(n, 4)
    return Qw[:, [1, 2, 3, 0]], a[4 * n:]                # -> [x y z w]


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_reference_output(name):
    path = os.path.join(HERE, "golden", "ref_%s.out" % name)
    if not os.path.exists(path):
        pytest.skip("no reference output committed for %s (tools/ref_golden/README.md)" % name)
    c = CASES[name]
    g = build_case(name)
    cost = ral.parse_cost(c["args"][0])
    sigma = float(c["args"][1]) * np.pi / 180
    irls_iters, l1_iters, th = int(c["args"][2]), int(c["args"][3]), float(c["args"][4])
    f = g["f"]
    rc, Q = O.init_mst(g["Q"], g["QQ"], g["I"], max(g["n_abs_read"], f))
    assert rc == 0
    a = O.l1ra(g["QQ"], g["I"], Q, f, l1_iters, th)
    b = O.irls(g["QQ"], g["I"], a["Q"], f, cost, sigma, irls_iters, th)
    Qo = O.quat_normalised(b["Q"], f)
    Qr, wr = read_reference_output(path, g["n"], g["m"])
    assert synth.angular_distance(Qo, Qr).max() < 1e-6
    np.testing.assert_allclose(b["weights"], wr, rtol=1e-5, atol=1e-9)

"""C oracle vs the independent NumPy/SciPy twin (both restate ral/l1_irls.cpp; they share no code)."""
import numpy as np
import pytest

from irotavg_amd import synth
from oracle import np_twin as T
from oracle import oracle as O


@pytest.fixture(scope="module")
def small():
    G = synth.make_graph(150, 1200, 0.15, seed=21)
    Q = np.zeros((150, 4)); Q[:, 3] = 1; Q[0] = G["Qgt"][0]
    return G, Q


def test_stagewise_agreement(small):
    G, Q = small
    rc, Qm = O.init_mst(Q, G["QQ"], G["I"], 1)
    np.testing.assert_array_equal(Qm, T.init_mst(Q, G["QQ"], G["I"], 1))
    d = O.delta_rel(G["I"], G["QQ"], Qm)
    np.testing.assert_allclose(d, T.delta_rel(G["I"], G["QQ"], Qm), atol=1e-15)
    np.testing.assert_allclose(O.log_map(d), T.log_map(d), atol=1e-14)
    np.testing.assert_allclose(O.exp_map(d * 0.1), T.exp_map(d * 0.1), atol=1e-15)
    np.testing.assert_array_equal(O.make_A(150, 3, G["I"]).toarray(), T.make_A(150, 3, G["I"]).toarray())


def test_fixture_pipeline_agreement(fixture_graph):
    g = fixture_graph
    rc, Q0 = O.init_mst(g["Q"], g["QQ"], g["I"], 1)
    a, b = O.l1ra(g["QQ"], g["I"], Q0, 1, 5, 1e-3), T.l1ra(g["QQ"], g["I"], Q0, 1, 5, 1e-3)
    assert a["iters"] == b["iters"]
    np.testing.assert_allclose(a["scores"], b["scores"], rtol=1e-9)
    assert synth.angular_distance(a["Q"], b["Q"]).max() < 1e-10
    a2, b2 = O.irls(g["QQ"], g["I"], a["Q"], 1), T.irls(g["QQ"], g["I"], b["Q"], 1)
    assert a2["iters"] == b2["iters"]
    assert synth.angular_distance(a2["Q"], b2["Q"]).max() < 1e-10
    np.testing.assert_allclose(a2["weights"], b2["weights"], rtol=1e-9)


@pytest.mark.parametrize("cost", range(14))
def test_every_cost_agreement(small, cost):
    G, Q = small
    rc, Qm = O.init_mst(Q, G["QQ"], G["I"], 1)
    a = O.irls(G["QQ"], G["I"], Qm, 1, cost, max_iters=8)
    b = T.irls(G["QQ"], G["I"], Qm, 1, cost, max_iters=8)
    assert a["iters"] == b["iters"]
    assert synth.angular_distance(a["Q"], b["Q"]).max() < 1e-9
    np.testing.assert_allclose(a["weights"], b["weights"], rtol=1e-7, atol=1e-12)


def test_l1decode_and_l1ra_agreement(small):
    G, Q = small
    rc, Qm = O.init_mst(Q, G["QQ"], G["I"], 1)
    w = O.log_map(O.delta_rel(G["I"], G["QQ"], Qm))
    A = T.make_A(150, 1, G["I"])
    H = T._H_builder(150, 1, G["I"])
    for c in range(3):
        rc, x, stuck = O.l1decode_pd(150, 1, G["I"], w[:, c], 2)
        xt, st = T.l1decode_pd(A, H, w[:, c], 2)
        assert rc == 0 and stuck == int(st)
        np.testing.assert_allclose(x, xt, atol=1e-10)
    a, b = O.l1ra(G["QQ"], G["I"], Qm, 1, 5, 1e-3), T.l1ra(G["QQ"], G["I"], Qm, 1, 5, 1e-3)
    assert a["iters"] == b["iters"]
    assert synth.angular_distance(a["Q"], b["Q"]).max() < 1e-9


def test_make_AtA_semantics_against_literal_construction():
    """The oracle's Hessian fill follows make_AtA (:811-848): endpoints skipped independently."""
    I = np.array([[0, 2], [3, 1], [2, 3], [2, 4], [3, 4]], dtype=np.int32)   # f = 2
    nu, f = 3, 2
    AtA = T.make_AtA(nu, f, I)
    s = np.array([1.5, 2.5, 3.5, 4.5, 5.5])
    H_lit = (AtA @ s).reshape(nu, nu, order="F")
    H = T._H_builder(5, f, I)(s).toarray()
    np.testing.assert_allclose(H, H_lit)
    # edge (3,1): second endpoint fixed, first free -> still adds to (i,i) here, unlike make_A
    assert H[1, 1] == pytest.approx(2.5 + 3.5 + 5.5)

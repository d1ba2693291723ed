"""GPU parity tests: every stage and the full drivers of the HIP path (through the C ABI) against
the CPU oracle on the same seeded inputs, at sizes the oracle finishes in seconds.

Tolerances (fp64 path): stage outputs 1e-12 absolute; linear-solve outputs 1e-8 relative to the
step size (PCG stops at ||r|| <= 1e-10 ||b||); final rotations within 1e-8 rad of the oracle --
four orders inside the 1e-4 rad the north star asks for -- with IDENTICAL outer iteration counts.
"""
import os

import numpy as np
import pytest

from irotavg_amd import capi, ral, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu
SIG = 5 * np.pi / 180


def mst_init(G, n, f=1):
    Q = np.zeros((n, 4)); Q[:, 3] = 1; Q[:f] = G["Qgt"][:f]
    rc, Qm = O.init_mst(Q, G["QQ"], G["I"], f)
    assert rc == 0
    return Qm


@pytest.fixture(scope="module")
def syn():
    G = synth.make_graph(3000, 36000, 0.05, seed=2)
    return G, mst_init(G, 3000)


def test_device_present():
    assert capi.lib().irotavg_device_count() > 0, "GPU tests need a HIP device (no CPU fallback)"


def test_k1_edge_residual(fixture_graph, syn):
    for I, QQ, Q, n in [(fixture_graph["I"], fixture_graph["QQ"],
                         O.init_mst(fixture_graph["Q"], fixture_graph["QQ"], fixture_graph["I"], 1)[1], 1832),
                        (syn[0]["I"], syn[0]["QQ"], syn[1], 3000)]:
        with capi.Graph(I, QQ, n, 1) as G:
            G.set_rotations(Q)
            G.edge_residual()
            r = G.get_residuals()
        ro = O.log_map(O.delta_rel(I, QQ, Q))[:, :3]
        np.testing.assert_allclose(r, ro, atol=1e-12, rtol=0)


def test_k1_edge_cases():
    """theta -> 0 (s2 < EPS guard), theta -> pi (half-open wrap), un-normalised inputs, w < 0."""
    Q = np.array([[0, 0, 0, 1.0], [0, 0, 0, 1.0], [0, 0, 0, -1.0], [0, 0, 0, 2.0]])
    I = np.array([[0, 1], [0, 1], [0, 1], [0, 2], [0, 3], [1, 0]], dtype=np.int32)
    QQ = np.array([[0, 0, 0, 1.0],                      # identity -> zero vector
                   [1e-17, 0, 0, 1.0],                  # below EPS -> zeroed
                   [1.0, 0, 0, 0.0],                    # exactly pi -> wraps to -pi
                   [np.sin(.2), 0, 0, np.cos(.2)],      # Q_j = -identity
                   [0, np.sin(.3), 0, np.cos(.3)],      # Q_j not unit
                   [0, 0, np.sin(1.5), -np.cos(1.5)]])  # negative w measurement, i > j
    with capi.Graph(I, QQ, 4, 1) as G:
        G.set_rotations(Q)
        G.edge_residual()
        r = G.get_residuals()
    ro = O.log_map(O.delta_rel(I, QQ, Q))[:, :3]
    np.testing.assert_allclose(r, ro, atol=1e-14, rtol=1e-14)
    # Q_j = identity: Qinv_j = -identity, d = -QQ = (-1,0,0,-0): theta = pi wraps to -pi, xyz = (-1)(-pi)
    assert (r[0] == 0).all() and (r[1] == 0).all() and r[2, 0] == pytest.approx(np.pi)


def test_ls_solve_matches_direct_solver(syn):
    G0, Qm = syn
    rng = np.random.default_rng(0)
    w = rng.uniform(0.1, 5.0, size=len(G0["I"]))
    with capi.Graph(G0["I"], G0["QQ"], 3000, 1) as G:
        G.set_rotations(Qm)
        G.edge_residual()
        G.set_weights(w)
        X = G.ls_solve()
        st = G.stats()
    ro = O.log_map(O.delta_rel(G0["I"], G0["QQ"], Qm))[:, :3]
    rc, Xo = O.ls_solve(3000, 1, G0["I"], w, ro)
    assert rc == 0
    assert np.abs(X - Xo).max() < 1e-8 * np.abs(Xo).max()
    assert max(st["last_relres"]) <= 1e-10 and st["pcg_iters_last"] < 200
    assert st["levels"] >= 2                       # the multigrid hierarchy is in use


def test_plain_jacobi_pcg_gives_the_same_solution(syn):
    G0, Qm = syn
    with capi.Graph(G0["I"], G0["QQ"], 3000, 1, mg_levels_max=1) as G:
        G.set_rotations(Qm)
        G.edge_residual()
        X1 = G.ls_solve()
        it1 = G.stats()["pcg_iters_last"]
    with capi.Graph(G0["I"], G0["QQ"], 3000, 1) as G:
        G.set_rotations(Qm)
        G.edge_residual()
        X2 = G.ls_solve()
        it2 = G.stats()["pcg_iters_last"]
    assert np.abs(X1 - X2).max() < 1e-8 * np.abs(X2).max()
    assert it2 < it1                                # the hierarchy pays


@pytest.mark.parametrize("cost", range(14))
def test_weight_update_every_cost(syn, cost):
    G0, Qm = syn
    with capi.Graph(G0["I"], G0["QQ"], 3000, 1) as G:
        G.set_rotations(Qm)
        G.edge_residual()
        X = G.ls_solve()
        prev = np.random.default_rng(1).uniform(0.5, 2, size=len(G0["I"]))
        G.set_weights(prev)
        G.update_weights(cost, SIG)
        w = G.get_weights()
    ro = O.log_map(O.delta_rel(G0["I"], G0["QQ"], Qm))[:, :3]
    A = O.make_A(3000, 1, G0["I"])
    E = A @ X - ro
    from oracle import np_twin as T
    wo = T.weights_update(cost, SIG, E, prev)
    np.testing.assert_allclose(w, wo, rtol=1e-11, atol=1e-300)


def test_apply_step_matches_oracle(syn):
    G0, Qm = syn
    with capi.Graph(G0["I"], G0["QQ"], 3000, 1) as G:
        G.set_rotations(Qm)
        G.edge_residual()
        X = G.ls_solve()
        score = G.apply_step()
        Q1 = G.get_rotations()
    W = O.exp_map(np.concatenate([X, np.zeros((2999, 1))], axis=1))
    want = synth.qmul(Qm[1:], W)
    np.testing.assert_allclose(Q1[1:], want, atol=1e-14)
    np.testing.assert_array_equal(Q1[0], Qm[0])                # gauge: fixed row bit-unchanged
    assert score == pytest.approx(np.linalg.norm(X, axis=1).mean(), rel=1e-13)


def test_fixture_full_pipeline_matches_golden(fixture_graph):
    """ral/test.cpp:277-302 with default arguments, through the drop-in functions."""
    g = fixture_graph
    f = g["f"]
    Q = g["Q"].copy()
    ral.init_mst(Q, g["QQ"], g["I"], max(g["n_abs_read"], f))
    l1_it, _ = ral.l1ra(g["QQ"], g["I"], None, Q, f, 5, 1e-3)
    w = np.zeros(g["m"])
    ir_it, _ = ral.irls(g["QQ"], g["I"], None, ral.Geman_McClure, SIG, Q, f, 50, 1e-3, w)
    ral.quat_normalised(Q, f)
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "fixture_expected.npz"))
    assert (l1_it, ir_it) == (1, 2)
    assert synth.angular_distance(Q, gold["Q"]).max() < 1e-8
    np.testing.assert_allclose(w, gold["weights"], rtol=1e-8)
    np.testing.assert_array_equal(Q[0], g["Q"][0])


@pytest.mark.parametrize("cost", range(14))
def test_irls_every_cost_matches_golden(cost):
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "synth400_expected.npz"))
    I, QQ, Q0 = gold["I"], gold["QQ"], gold["Q_l1"]
    with capi.Graph(I, QQ, 400, 1) as G:
        G.set_rotations(Q0)
        r = G.irls(cost, SIG, 20, 1e-3)
        Q = G.get_rotations()
        w = G.get_weights()
    ro = O.irls(QQ, I, Q0, 1, cost, SIG, 20, 1e-3)
    assert r["iters"] == ro["iters"]
    np.testing.assert_allclose(r["scores"], ro["scores"], rtol=1e-7)
    assert synth.angular_distance(Q, gold["Q_cost%d" % cost]).max() < 1e-8
    np.testing.assert_allclose(w, gold["w_cost%d" % cost], rtol=1e-6, atol=1e-12)


def test_l1decode_pd_matches_oracle(syn):
    G0, Qm = syn
    ro = O.log_map(O.delta_rel(G0["I"], G0["QQ"], Qm))[:, :3]
    with capi.Graph(G0["I"], G0["QQ"], 3000, 1) as G:
        for c in range(3):
            x, stuck = G.l1decode_pd(np.ascontiguousarray(ro[:, c]), 2)
            rc, xo, so = O.l1decode_pd(3000, 1, G0["I"], ro[:, c], 2)
            assert rc == 0 and stuck == so
            assert np.abs(x - xo).max() < 1e-8 * max(np.abs(xo).max(), 1e-300)


def test_l1ra_then_irls_matches_oracle(syn):
    G0, Qm = syn
    with capi.Graph(G0["I"], G0["QQ"], 3000, 1) as G:
        G.set_rotations(Qm)
        a = G.l1ra(5, 1e-3)
        Qa = G.get_rotations()
        b = G.irls(4, SIG, 50, 1e-3)
        G.quat_normalised()
        Qb = G.get_rotations()
        w = G.get_weights()
    ra = O.l1ra(G0["QQ"], G0["I"], Qm, 1, 5, 1e-3)
    rb = O.irls(G0["QQ"], G0["I"], ra["Q"], 1, 4, SIG, 50, 1e-3)
    assert a["iters"] == ra["iters"] and b["iters"] == rb["iters"]
    np.testing.assert_allclose(a["scores"], ra["scores"], rtol=1e-7)
    assert synth.angular_distance(Qa, ra["Q"]).max() < 1e-8
    assert synth.angular_distance(Qb, O.quat_normalised(rb["Q"], 1)).max() < 1e-8
    np.testing.assert_allclose(w, rb["weights"], rtol=1e-6)
    np.testing.assert_allclose(np.linalg.norm(Qb[1:], axis=1), 1, atol=1e-15)


def test_multiple_fixed_views_and_quirk_edges():
    """f = 5 fixed views; edges in both orientations so that some have their SECOND endpoint fixed
    (dropped by make_A, ral/l1_irls.cpp:770-771, but kept by make_AtA :825-835), plus duplicates."""
    S = synth.make_graph(600, 6000, 0.2, seed=13)
    rng = np.random.default_rng(5)
    I, QQ = S["I"].copy(), S["QQ"].copy()
    flip = rng.random(len(I)) < 0.3
    I[flip] = I[flip][:, ::-1]
    QQ[flip] = synth.qconj(QQ[flip])
    I = np.concatenate([I, I[:50]]).astype(np.int32)        # duplicate edges
    QQ = np.concatenate([QQ, QQ[:50]])
    f = 5
    assert ((I[:, 1] < f) & (I[:, 0] >= f)).sum() > 0       # quirk edges exist
    Q = np.zeros((600, 4)); Q[:, 3] = 1; Q[:f] = S["Qgt"][:f]
    rc, Qm = O.init_mst(Q, QQ, I, f)
    with capi.Graph(I, QQ, 600, f) as G:
        G.set_rotations(Qm)
        a = G.l1ra(5, 1e-3)
        b = G.irls(4, SIG, 50, 1e-3)
        Qg = G.get_rotations()
        w = G.get_weights()
    ra = O.l1ra(QQ, I, Qm, f, 5, 1e-3)
    rb = O.irls(QQ, I, ra["Q"], f, 4, SIG, 50, 1e-3)
    assert a["iters"] == ra["iters"] and b["iters"] == rb["iters"]
    assert synth.angular_distance(Qg, rb["Q"]).max() < 1e-8
    np.testing.assert_allclose(w, rb["weights"], rtol=1e-6)
    np.testing.assert_array_equal(Qg[:f], Qm[:f])


def test_tiny_window_graph():
    """The sliding-window solve of ViewGraph::rotAvg(10) is this small (src/ViewGraph.cpp:1282-1363)."""
    S = synth.make_graph(14, 30, 0.0, seed=3)
    f = 4
    Q = np.zeros((14, 4)); Q[:, 3] = 1; Q[:f] = S["Qgt"][:f]
    rc, Qm = O.init_mst(Q, S["QQ"], S["I"], f)
    with capi.Graph(S["I"], S["QQ"], 14, f) as G:
        G.set_rotations(Qm)
        a = G.l1ra(100, 1e-3)
        b = G.irls(4, SIG, 100, 1e-3)
        Qg = G.get_rotations()
    ra = O.l1ra(S["QQ"], S["I"], Qm, f, 100, 1e-3)
    rb = O.irls(S["QQ"], S["I"], ra["Q"], f, 4, SIG, 100, 1e-3)
    assert (a["iters"], b["iters"]) == (ra["iters"], rb["iters"])
    assert synth.angular_distance(Qg, rb["Q"]).max() < 1e-8


def test_max_iteration_cap_and_error_codes(syn):
    G0, Qm = syn
    with capi.Graph(G0["I"], G0["QQ"], 3000, 1) as G:
        G.set_rotations(Qm)
        r = G.irls(4, SIG, 1, 1e-9)
        assert r["iters"] == 1                               # " Max Iteration" path (:746-749)
        with pytest.raises(capi.IrotavgError) as e:
            G.irls(14, SIG, 5, 1e-3)
        assert e.value.code == capi.ERR_UNKNOWN_COST
        r0 = G.irls(4, SIG, 0, 1e-3)
        assert r0["iters"] == 0


def test_determinism_bitwise(syn):
    G0, Qm = syn
    outs = []
    for _ in range(2):
        with capi.Graph(G0["I"], G0["QQ"], 3000, 1) as G:
            G.set_rotations(Qm)
            G.irls(1, SIG, 10, 1e-3)
            outs.append((G.get_rotations(), G.get_weights(), G.stats()["pcg_iters"]))
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    assert outs[0][2] == outs[1][2]


def test_isolated_view_single_unknown_and_self_loop():
    """Ragged inputs: a free view without edges (zero diagonal -> stays put, like a dead SPQR
    column / the oracle's dead pivot), a single unknown, and a self-loop edge (make_A keeps the
    -1 coefficient, ral/l1_irls.cpp:770-776)."""
    rng = np.random.default_rng(4)
    # views 0 (fixed), 1..4 free; view 4 has no edges at all; edge (2,2) is a self loop
    I = np.array([[0, 1], [1, 2], [0, 2], [2, 2], [1, 3], [0, 3]], dtype=np.int32)
    Qgt = rng.normal(size=(5, 4)); Qgt /= np.linalg.norm(Qgt, axis=1, keepdims=True)
    QQ = synth.qmul(synth.qexp(rng.normal(scale=0.01, size=(6, 3))), synth.qmul(Qgt[I[:, 1]], synth.qconj(Qgt[I[:, 0]])))
    Q0 = synth.qmul(synth.qexp(rng.normal(scale=0.03, size=(5, 3))), Qgt); Q0[0] = Qgt[0]
    for cost in (0, 4):
        with capi.Graph(I, QQ, 5, 1) as G:
            G.set_rotations(Q0)
            r = G.irls(cost, SIG, 20, 1e-6)
            Q = G.get_rotations(); w = G.get_weights()
        ro = O.irls(QQ, I, Q0, 1, cost, SIG, 20, 1e-6)
        assert r["iters"] == ro["iters"]
        assert synth.angular_distance(Q, ro["Q"]).max() < 1e-9
        np.testing.assert_allclose(w, ro["weights"], rtol=1e-8)
        np.testing.assert_array_equal(Q[4], Q0[4])             # the isolated view never moves
    # one unknown
    I1 = np.array([[0, 1], [0, 1]], dtype=np.int32)
    with capi.Graph(I1, QQ[:2], 2, 1) as G:
        G.set_rotations(Q0[:2])
        r = G.irls(4, SIG, 20, 1e-9)
        Q = G.get_rotations()
    ro = O.irls(QQ[:2], I1, Q0[:2], 1, 4, SIG, 20, 1e-9)
    assert r["iters"] == ro["iters"] and synth.angular_distance(Q, ro["Q"]).max() < 1e-10


def test_bad_inputs_are_rejected():
    QQ = np.tile([0, 0, 0, 1.0], (2, 1))
    with pytest.raises(capi.IrotavgError) as e:
        capi.Graph(np.array([[0, 5], [1, 2]], dtype=np.int32), QQ, 3, 1)   # endpoint out of range
    assert e.value.code == capi.ERR_BAD_ARG
    with pytest.raises(capi.IrotavgError) as e:
        capi.Graph(np.zeros((0, 2), dtype=np.int32), np.zeros((0, 4)), 3, 1)  # no edges
    assert e.value.code == capi.ERR_BAD_ARG
    with pytest.raises(capi.IrotavgError) as e:
        capi.Graph(np.array([[0, 1]], dtype=np.int32), QQ[:1], 2, 2)          # no free view
    assert e.value.code == capi.ERR_BAD_ARG


def test_handle_churn_reuses_device_memory_without_changing_results(syn):
    """Buffers of destroyed handles are cached and handed to the next handle (DevPool): a solve on
    recycled (dirty) memory gives bitwise the result of the first one; irotavg_trim_memory returns
    the cache to the driver."""
    G0, Qm = syn
    m = len(G0["I"])

    def one_shot(cost):
        Q, w = Qm.copy(order="F"), np.ones(m)
        it, _ = ral.irls(G0["QQ"], G0["I"], None, cost, SIG, Q, 1, 100, 1e-3, w)
        return it, Q, w

    capi.trim_memory()
    first = one_shot(4)
    outs = [one_shot(c) for c in (1, 4)]               # the L1 solve dirties the recycled blocks
    assert outs[1][0] == first[0]
    np.testing.assert_array_equal(outs[1][1], first[1])
    np.testing.assert_array_equal(outs[1][2], first[2])
    Qa, Qb = Qm.copy(order="F"), Qm.copy(order="F")
    ral.l1ra(G0["QQ"], G0["I"], None, Qa, 1, 100, 1e-3)
    ral.l1ra(G0["QQ"], G0["I"], None, Qb, 1, 100, 1e-3)
    np.testing.assert_array_equal(Qa, Qb)
    assert capi.trim_memory() > 0                      # the one-shot calls left their buffers cached
    assert capi.trim_memory() == 0


def test_floating_component_is_gauge_fixed_not_an_error():
    """Edge (2,0) has its SECOND endpoint fixed, so make_A drops its row (ral/l1_irls.cpp:770-771)
    and nothing ties the free views to the fixed one: A'D^2A is singular. SPQR returns a basic
    solution for that (rank detection with a relative tolerance); here the eliminations declare the
    pivot that cancels to rounding level dead (kDeadTol x the largest diagonal entry) -- on the
    handle path, in both window kernels and in the oracle -- so the solve runs instead of dying in
    IROTAVG_ERR_NOT_CONVERGED. Which view gets pinned is solver-defined: only sanity is checked."""
    I = np.array([[1, 5], [2, 0], [4, 1], [1, 5], [1, 3], [4, 3], [3, 2]], dtype=np.int32)
    rng = np.random.default_rng(3)
    Qgt = rng.normal(size=(6, 4)); Qgt /= np.linalg.norm(Qgt, axis=1, keepdims=True)
    QQ = synth.qmul(synth.qexp(rng.normal(scale=0.01, size=(7, 3))), synth.qmul(Qgt[I[:, 1]], synth.qconj(Qgt[I[:, 0]])))
    Q0 = synth.qmul(synth.qexp(rng.normal(scale=0.05, size=(6, 3))), Qgt); Q0[0] = Qgt[0]
    for cost in (0, 3, 4):
        with capi.Graph(I, QQ, 6, 1) as G:
            G.set_rotations(Q0)
            r = G.irls(cost, SIG, 20, 1e-3)
            Q = G.get_rotations()
        assert r["iters"] >= 1 and np.isfinite(Q).all()
        np.testing.assert_array_equal(Q[0], Q0[0])
        ro = O.irls(QQ, I, Q0, 1, cost, SIG, 20, 1e-3)
        assert ro["rc"] == 0 and np.isfinite(ro["Q"]).all()
        # the residuals of the surviving edges are gauge independent: both solvers fit them equally well
        res_g = O.log_map(O.delta_rel(I, QQ, Q))[np.arange(7) != 1, :3]
        res_o = O.log_map(O.delta_rel(I, QQ, ro["Q"]))[np.arange(7) != 1, :3]
        assert abs(np.linalg.norm(res_g) - np.linalg.norm(res_o)) < 1e-3
        for kern in (1, 2):
            w = capi.window_solve(I, QQ, Q0, 1, cost, SIG, 0, 20, 1e-3, kernel=kern)
            assert np.isfinite(w["Q"]).all()


def test_graph_build_is_independent_of_the_host_thread_count(syn, monkeypatch):
    """irotavg_graph_create builds adjacency, coarse patterns and SELL layout on several host threads
    (slots claimed with atomic counters, then sorted on the edge id): a solve on a handle built by
    1 thread and one built by the default thread count must agree bit for bit."""
    G0, Qm = syn
    outs = []
    for threads in ("1", "7", None):
        if threads is None:
            monkeypatch.delenv("IROTAVG_BUILD_THREADS", raising=False)
        else:
            monkeypatch.setenv("IROTAVG_BUILD_THREADS", threads)
        with capi.Graph(G0["I"], G0["QQ"], 3000, 1) as G:
            G.set_rotations(Qm)
            G.l1ra(2, 1e-3)
            G.irls(4, SIG, 20, 1e-3)
            outs.append((G.get_rotations(), G.get_weights(), G.stats()["pcg_iters"]))
    for o in outs[1:]:
        np.testing.assert_array_equal(o[0], outs[0][0])
        np.testing.assert_array_equal(o[1], outs[0][1])
        assert o[2] == outs[0][2]


def test_ill_conditioned_small_graph_still_returns_the_oracle_answer():
    """tools/fuzz_parity.py seed 31, case 1037 (tests/golden/illcond_case_seed31_1037.npz): 281 views, 319
    edges (almost a tree), cost L1.5 whose weights hit the 1e4 cap -- the operator spreads over ~13 decades, the
    explicit inverse of the single level is useless as a preconditioner and the PCG diverges. The reference's
    factorisations return an answer there; so does the handle: re-inversion, then the Cholesky solve of
    dense.hip (k_chol_solve). Here the answer even agrees with the oracle's."""
    import os
    c = np.load(os.path.join(os.path.dirname(__file__), "golden", "illcond_case_seed31_1037.npz"))
    n, f, I, QQ, Q0, cost = int(c["n"]), int(c["f"]), c["I"], c["QQ"], c["Q0"], int(c["cost"])
    ra = O.l1ra(QQ, I, Q0, f, 3, 1e-3)
    rb = O.irls(QQ, I, ra["Q"], f, cost, SIG, 15, 1e-3)
    with capi.Graph(I, QQ, n, f) as G:
        G.set_rotations(Q0)
        a = G.l1ra(3, 1e-3)
        b = G.irls(cost, SIG, 15, 1e-3)
        Q = G.get_rotations()
        st = G.stats()
    assert (a["iters"], b["iters"]) == (ra["iters"], rb["iters"])
    assert st["pcg_stagnated"] >= 1                      # at least one solve went past the PCG
    assert synth.angular_distance(Q, rb["Q"]).max() < 1e-6

"""The single-kernel window solve (irotavg_amd/csrc/window.hip): l1ra + irls of a sliding-window
sized problem in ONE launch, against the oracle pipeline on the same inputs."""
import time

import numpy as np
import pytest

from irotavg_amd import capi, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu
SIG = 5 * np.pi / 180


def small(n, m, f, seed, p=0.1, init_noise=0.05):
    S = synth.make_graph(n, m, p, seed=seed)
    rng = np.random.default_rng(seed)
    Q = synth.qmul(synth.qexp(rng.normal(scale=init_noise, size=(n, 3))), S["Qgt"])
    Q[:f] = S["Qgt"][:f]
    return S, Q


@pytest.mark.parametrize("n,m,f", [(14, 40, 4), (30, 120, 1), (70, 400, 8), (64, 300, 1), (12, 30, 2)])
def test_window_pipeline_matches_oracle(n, m, f):
    S, Q0 = small(n, m, f, seed=n)
    r = capi.window_solve(S["I"], S["QQ"], Q0, f, 4, SIG, 100, 100, 1e-3)
    a = O.l1ra(S["QQ"], S["I"], Q0, f, 100, 1e-3)
    b = O.irls(S["QQ"], S["I"], a["Q"], f, 4, SIG, 100, 1e-3)
    assert (r["l1_iters"], r["irls_iters"]) == (a["iters"], b["iters"])
    assert synth.angular_distance(r["Q"], b["Q"]).max() < 1e-9
    np.testing.assert_allclose(r["weights"], b["weights"], rtol=1e-7)
    np.testing.assert_array_equal(r["Q"][:f], Q0[:f])


@pytest.mark.parametrize("n,m,f", [(14, 40, 4), (12, 30, 2), (17, 64, 1), (40, 60, 30), (3, 3, 1)])
def test_wave_kernel_matches_general_kernel_and_oracle(n, m, f):
    """<= 16 free views / <= 64 edges: the wave-resident kernel (kernel=2) vs the LDS kernel (1)."""
    S, Q0 = small(n, m, f, seed=7 * n + m)
    w = capi.window_solve(S["I"], S["QQ"], Q0, f, 4, SIG, 100, 100, 1e-3, kernel=2)
    g = capi.window_solve(S["I"], S["QQ"], Q0, f, 4, SIG, 100, 100, 1e-3, kernel=1)
    auto = capi.window_solve(S["I"], S["QQ"], Q0, f, 4, SIG, 100, 100, 1e-3)
    a = O.l1ra(S["QQ"], S["I"], Q0, f, 100, 1e-3)
    b = O.irls(S["QQ"], S["I"], a["Q"], f, 4, SIG, 100, 1e-3)
    assert (w["l1_iters"], w["irls_iters"]) == (g["l1_iters"], g["irls_iters"]) == (a["iters"], b["iters"])
    assert synth.angular_distance(w["Q"], b["Q"]).max() < 1e-9
    assert synth.angular_distance(w["Q"], g["Q"]).max() < 1e-11
    np.testing.assert_allclose(w["weights"], b["weights"], rtol=1e-7)
    np.testing.assert_array_equal(auto["Q"], w["Q"])           # automatic choice = wave kernel here
    np.testing.assert_array_equal(w["Q"][:f], Q0[:f])


@pytest.mark.parametrize("cost", range(14))
def test_wave_kernel_every_cost(cost):
    S, Q0 = small(16, 60, 2, seed=300 + cost)
    r = capi.window_solve(S["I"], S["QQ"], Q0, 2, cost, SIG, 3, 12, 1e-3, kernel=2)
    a = O.l1ra(S["QQ"], S["I"], Q0, 2, 3, 1e-3)
    b = O.irls(S["QQ"], S["I"], a["Q"], 2, cost, SIG, 12, 1e-3)
    assert (r["l1_iters"], r["irls_iters"]) == (a["iters"], b["iters"])
    assert synth.angular_distance(r["Q"], b["Q"]).max() < 1e-8
    np.testing.assert_allclose(r["weights"], b["weights"], rtol=1e-6, atol=1e-12)


def test_wave_kernel_quirk_edges_and_limits():
    S, Q0 = small(20, 60, 5, seed=19)
    I, QQ = S["I"].copy(), S["QQ"].copy()
    flip = np.random.default_rng(1).random(len(I)) < 0.3      # edges whose 2nd endpoint is fixed
    I[flip] = I[flip][:, ::-1]
    QQ[flip] = synth.qconj(QQ[flip])
    r = capi.window_solve(I, QQ, Q0, 5, 4, SIG, 100, 100, 1e-3, kernel=2)
    a = O.l1ra(QQ, I, Q0, 5, 100, 1e-3)
    b = O.irls(QQ, I, a["Q"], 5, 4, SIG, 100, 1e-3)
    assert (r["l1_iters"], r["irls_iters"]) == (a["iters"], b["iters"])
    assert synth.angular_distance(r["Q"], b["Q"]).max() < 1e-9
    big, Qb = small(30, 100, 1, seed=2)                       # fits the LDS kernel, not the wave kernel
    with pytest.raises(capi.IrotavgError) as e:
        capi.window_solve(big["I"], big["QQ"], Qb, 1, kernel=2)
    assert e.value.code == capi.ERR_BAD_ARG
    capi.window_solve(big["I"], big["QQ"], Qb, 1, kernel=0)


@pytest.mark.parametrize("cost", range(14))
def test_window_every_cost(cost):
    S, Q0 = small(40, 200, 2, seed=100 + cost)
    r = capi.window_solve(S["I"], S["QQ"], Q0, 2, cost, SIG, 3, 12, 1e-3)
    a = O.l1ra(S["QQ"], S["I"], Q0, 2, 3, 1e-3)
    b = O.irls(S["QQ"], S["I"], a["Q"], 2, cost, SIG, 12, 1e-3)
    assert (r["l1_iters"], r["irls_iters"]) == (a["iters"], b["iters"])
    assert synth.angular_distance(r["Q"], b["Q"]).max() < 1e-8
    np.testing.assert_allclose(r["weights"], b["weights"], rtol=1e-6, atol=1e-12)


def test_window_quirk_edges_and_limits():
    S, Q0 = small(50, 300, 5, seed=9)
    I, QQ = S["I"].copy(), S["QQ"].copy()
    flip = np.random.default_rng(1).random(len(I)) < 0.3      # edges whose 2nd endpoint is fixed
    I[flip] = I[flip][:, ::-1]
    QQ[flip] = synth.qconj(QQ[flip])
    r = capi.window_solve(I, QQ, Q0, 5, 4, SIG, 100, 100, 1e-3)
    a = O.l1ra(QQ, I, Q0, 5, 100, 1e-3)
    b = O.irls(QQ, I, a["Q"], 5, 4, SIG, 100, 1e-3)
    assert (r["l1_iters"], r["irls_iters"]) == (a["iters"], b["iters"])
    assert synth.angular_distance(r["Q"], b["Q"]).max() < 1e-9
    big, Qb = small(200, 1000, 1, seed=2)
    with pytest.raises(capi.IrotavgError) as e:                # 199 free views: does not fit
        capi.window_solve(big["I"], big["QQ"], Qb, 1)
    assert e.value.code == capi.ERR_BAD_ARG


def test_rotavg_window_path_equals_general_path_and_is_fast():
    """rotAvg(10) through the window kernel vs the graph-handle path (no_window_kernel = 1)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from test_viewgraph import build_sequence, rot
    from irotavg_amd.viewgraph import ViewGraph
    n = 120
    Qgt, rel = build_sequence(n, seed=5, n_loops=0)
    by_new = {}
    for (i, j), R in rel.items():
        by_new.setdefault(j, []).append((i, R))
    fast, slow = ViewGraph(), ViewGraph(no_window_kernel=1)
    tf = ts = 0.0
    for v in range(n):
        R0 = rot(Qgt[0]) if v == 0 else sorted(by_new[v], key=lambda t: -t[0])[0][1] @ fast.R(v - 1)
        for g in (fast, slow):
            g.addView(R0)
            for (i, R) in by_new.get(v, []):
                g.connect(i, v, R)
            if v % 20 == 0:
                g.fixPose(v, rot(Qgt[v]))
        t = time.perf_counter(); a = fast.rotAvg(10); tf += time.perf_counter() - t
        t = time.perf_counter(); b = slow.rotAvg(10); ts += time.perf_counter() - t
        assert a["skipped"] == b["skipped"]
        if not a["skipped"]:
            assert (a["l1_iters"], a["irls_iters"]) == (b["l1_iters"], b["irls_iters"])
    for v in range(n):
        np.testing.assert_allclose(fast.R(v), slow.R(v), atol=1e-8)
    assert tf < ts / 3, (tf, ts)
    print("window kernel %.3f ms/call vs general path %.3f ms/call" % (1e3 * tf / n, 1e3 * ts / n))

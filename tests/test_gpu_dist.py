"""Sharded IRLS on the GPU: all `world` shards in one process on one GPU (loopback transport)
against the unsharded handle, and the RCCL transport with a 1-rank communicator."""
import numpy as np
import pytest

from irotavg_amd import capi, ral, synth

pytestmark = pytest.mark.gpu
SIG = 5 * np.pi / 180


def problem(n, m, p, f=1, seed=0):
    S = synth.make_graph(n, m, p, seed=seed)
    Q = np.zeros((n, 4)); Q[:, 3] = 1; Q[:f] = S["Qgt"][:f]
    ral.init_mst(Q, S["QQ"], S["I"], f)
    return S, Q


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("p_loop,band_direct", [(0.0, 0), (0.0, -1), (0.02, 0)])
def test_loopback_sharded_equals_unsharded(world, p_loop, band_direct):
    """band graph: the sharded DIRECT solver (one gather per linear solve; band_direct 0) and the sharded PCG (-1);
    with loop closures: the sharded PCG"""
    n, m, f = 20000, 300000, 3
    S, Q0 = problem(n, m, p_loop, f)
    with capi.Graph(S["I"], S["QQ"], n, f, band_direct=band_direct) as G:
        G.set_rotations(Q0)
        a = G.irls(4, SIG, 50, 1e-3)
        Qa, wa = G.get_rotations(), G.get_weights()
    with capi.DistGraph(S["I"], S["QQ"], n, f, world, band_direct=band_direct) as D:
        D.set_rotations(Q0)
        D.snapshot_rotations()
        D.irls(4, SIG, 2, 1e-3)          # ... a first run, thrown away: restore brings the start back
        D.restore_rotations()
        b = D.irls(4, SIG, 50, 1e-3)
        Qb, wb = D.get_rotations(into=Q0), D.get_weights()
        st = D.stats()
        direct = D.info()["direct_block"]
    assert a["iters"] == b["iters"]
    np.testing.assert_allclose(a["scores"], b["scores"], rtol=1e-6, atol=1e-9)  # rad; the last score is ~1e-5
    assert synth.angular_distance(Qa, Qb).max() < 1e-8
    assert not np.isnan(wb).any()                      # every edge belongs to some shard
    np.testing.assert_allclose(wa, wb, rtol=1e-6)
    np.testing.assert_array_equal(Qb[:f], Q0[:f])
    if p_loop == 0.0 and band_direct == 0:
        assert direct == 16 and st["direct_solves"] == b["iters"] + 2 and st["pcg_iters"] == 0
    else:
        assert direct == 0 and st["pcg_iters"] > 0 and st["direct_solves"] == 0


def test_loopback_every_cost_family():
    n, m = 8000, 80000
    S, Q0 = problem(n, m, 0.05, 1, seed=3)
    for cost in (1, 5, 12):                            # L1, Huber (keeps old weights), Talwar (zeros)
        with capi.Graph(S["I"], S["QQ"], n, 1) as G:
            G.set_rotations(Q0)
            a = G.irls(cost, SIG, 8, 1e-3)
            Qa, wa = G.get_rotations(), G.get_weights()
        with capi.DistGraph(S["I"], S["QQ"], n, 1, 4) as D:
            D.set_rotations(Q0)
            b = D.irls(cost, SIG, 8, 1e-3)
            Qb, wb = D.get_rotations(into=Q0), D.get_weights()
        assert a["iters"] == b["iters"]
        assert synth.angular_distance(Qa, Qb).max() < 1e-7
        np.testing.assert_allclose(wa, wb, rtol=1e-5, atol=1e-9)


def test_rccl_transport_single_rank():
    """ncclCommInitRank / ncclAllReduce / send-recv groups with a 1-rank communicator: exercises
    the RCCL code path on the one GPU a test box has."""
    n, m = 20000, 300000
    S, Q0 = problem(n, m, 0.02)
    uid = capi.DistGraph.unique_id()
    with capi.DistGraph(S["I"], S["QQ"], n, 1, 1, rank=0, unique_id=uid) as D:
        D.set_rotations(Q0)
        b = D.irls(4, SIG, 50, 1e-3)
        Qb = D.get_rotations(into=Q0)
    with capi.Graph(S["I"], S["QQ"], n, 1) as G:
        G.set_rotations(Q0)
        a = G.irls(4, SIG, 50, 1e-3)
        Qa = G.get_rotations()
    assert a["iters"] == b["iters"]
    assert synth.angular_distance(Qa, Qb).max() < 1e-8


def test_rccl_transport_single_rank_direct_solver_gathers_records():
    """The sharded DIRECT solver on the RCCL wire: the ranks' separator data travels as one record per rank through
    ncclAllGather (round 5; a sum-all-reduce of the zero-filled buffer until then) -- a 1-rank communicator on the one
    GPU of a test box runs that code: to-record, all-gather in place, from-records, the separator system."""
    n, m = 20000, 300000
    S, Q0 = problem(n, m, 0.0)
    uid = capi.DistGraph.unique_id()
    with capi.DistGraph(S["I"], S["QQ"], n, 1, 1, rank=0, unique_id=uid, band_direct=1) as D:
        assert D.info()["direct_block"] > 0
        D.set_rotations(Q0)
        b = D.irls(4, SIG, 50, 1e-3)
        Qb = D.get_rotations(into=Q0.copy())
        assert D.stats()["direct_solves"] > 0
    with capi.Graph(S["I"], S["QQ"], n, 1) as G:
        G.set_rotations(Q0)
        a = G.irls(4, SIG, 50, 1e-3)
        Qa = G.get_rotations()
    assert a["iters"] == b["iters"]
    assert synth.angular_distance(Qa, Qb).max() < 1e-9


def test_bad_world_size():
    S, Q0 = problem(200, 1200, 0.0)
    with pytest.raises(capi.IrotavgError):
        capi.DistGraph(S["I"], S["QQ"], 200, 1, 8)     # 199 free views cannot feed 8 shards


@pytest.mark.parametrize("world,p_loop", [(2, 0.0), (4, 0.02), (3, 0.01)])
def test_loopback_sharded_l1ra_then_irls_equals_unsharded(world, p_loop):
    """The callers' pipeline (l1ra then irls: src/ViewGraph.cpp:1400-1417, ral/test.cpp:295-301) on the
    sharded graph against the single-GPU handle: same outer iteration counts, scores, rotations."""
    n, m, f = 6000, 60000, 2
    S, Q0 = problem(n, m, p_loop, f, seed=2)
    with capi.Graph(S["I"], S["QQ"], n, f) as G:
        G.set_rotations(Q0)
        a1 = G.l1ra(3, 1e-3)
        Q1 = G.get_rotations()
        a2 = G.irls(4, SIG, 20, 1e-3)
        Qa, wa = G.get_rotations(), G.get_weights()
    Qw = Q0.copy()
    with capi.DistGraph(S["I"], S["QQ"], n, f, world) as D:
        D.set_rotations(Q0)
        b1 = D.l1ra(3, 1e-3)
        Qb1 = D.get_rotations(into=Qw).copy()
        b2 = D.irls(4, SIG, 20, 1e-3)
        Qb, wb = D.get_rotations(into=Qw), D.get_weights()
    assert a1["iters"] == b1["iters"] and a2["iters"] == b2["iters"]
    # p_loop 0: the Hessian solves of the sharded l1ra and the systems of the sharded irls are direct solves
    np.testing.assert_allclose(a1["scores"], b1["scores"], rtol=1e-6)
    assert synth.angular_distance(Q1, Qb1).max() < 1e-8
    assert synth.angular_distance(Qa, Qb).max() < 1e-8
    np.testing.assert_allclose(wa, wb, rtol=1e-6)


@pytest.mark.parametrize("p_loop", [0.0, 0.02])
def test_config4_size_eight_shards_match_the_oracle_checked_handle(p_loop):
    """BASELINE config 4's workload (the 100k-view / 2M-edge graph in 8 vertex-range shards), all shards on
    the one GPU of a test box (loopback transport): same IRLS iteration count, scores, rotations and weights
    as the unsharded handle, which tests/test_gpu_fullsize.py checks against the oracle on this very graph
    (full IRLS for p_loop = 0, normal-equation residual by the oracle's mat-vec for 0.02)."""
    n, m = 100000, 2000000
    S, Q0 = problem(n, m, p_loop, 1)
    with capi.Graph(S["I"], S["QQ"], n, 1) as G:
        G.set_rotations(Q0)
        a = G.irls(4, SIG, 100, 1e-3)
        Qa, wa = G.get_rotations(), G.get_weights()
    with capi.DistGraph(S["I"], S["QQ"], n, 1, 8) as D:
        D.set_rotations(Q0)
        b = D.irls(4, SIG, 100, 1e-3)
        Qb, wb = D.get_rotations(into=Q0.copy()), D.get_weights()
        st = D.stats()
    assert a["iters"] == b["iters"]
    np.testing.assert_allclose(a["scores"], b["scores"], rtol=1e-6, atol=1e-9)
    assert synth.angular_distance(Qa, Qb).max() < 1e-7
    np.testing.assert_allclose(wa, wb, rtol=1e-5, atol=1e-9)
    assert (st["direct_solves"] > 0) == (p_loop == 0.0) and (st["pcg_iters"] > 0) == (p_loop > 0.0)


def test_sharded_single_reduction_solve_hands_over_to_the_classic_recurrences(monkeypatch):
    """The sharded Chronopoulos-Gear PCG (one all-reduce per iteration) carries r and s = Lp by recurrence;
    a solve that stalls (or passes the give-up limit) restarts with the classic sharded recurrences from
    the saved right-hand side -- the decision comes from all-reduced values, so every rank takes it.
    Forced by a limit of 6 iterations; the result must still be the ORACLE's."""
    from oracle import oracle as O
    n, m = 20000, 300000
    S, Q0 = problem(n, m, 0.0, 1, seed=2)
    ro = O.irls(S["QQ"], S["I"], Q0, 1, 4, SIG, 50, 1e-3)
    out = []
    for limit in (None, "6"):
        if limit:
            monkeypatch.setenv("IROTAVG_CG2_GIVEUP", limit)
        with capi.DistGraph(S["I"], S["QQ"], n, 1, 4, band_direct=-1) as D:   # the sharded PCG (this is a band graph)
            D.set_rotations(Q0)
            r = D.irls(4, SIG, 50, 1e-3)
            out.append((r, D.get_rotations(into=Q0.copy()), D.get_weights(), D.stats()))
    assert out[0][3]["pcg_handed_over"] == 0
    assert 1 <= out[1][3]["pcg_handed_over"] <= out[1][3]["pcg_solves"]   # the limit acts at a poll of the host
    assert out[1][3]["pcg_iters"] > out[0][3]["pcg_iters"]      # the abandoned iterations are counted
    for r, Q, w, _ in out:
        assert r["iters"] == ro["iters"]
        np.testing.assert_allclose(r["scores"], ro["scores"], rtol=1e-5)
        assert synth.angular_distance(Q, ro["Q"]).max() < 1e-7
        np.testing.assert_allclose(w, ro["weights"], rtol=1e-5)


@pytest.mark.parametrize("n,m,f,world", [(5000, 20000, 1, 3), (4000, 16000, 4, 6), (9000, 180000, 2, 5),
                                         (2500, 75000, 1, 2), (30000, 120000, 1, 7)])
def test_sharded_direct_solver_shapes_match_the_oracle(n, m, f, world):
    """The sharded direct solver on shapes that stress its bookkeeping: blocks of 8 / 24 / 32, odd world sizes, shards
    of a few chunks only (the local top level is reached at once; partial chunks place their blocks so that the
    rank's last block stays the separator), several fixed views, a last shard much smaller than the others. l1ra then
    irls against the ORACLE."""
    from oracle import oracle as O
    S, Q0 = problem(n, m, 0.0, f, seed=5)
    Qw = Q0.copy()
    with capi.DistGraph(S["I"], S["QQ"], n, f, world, band_direct=1) as D:
        assert D.info()["direct_block"] in (8, 24, 32)
        D.set_rotations(Q0)
        a = D.l1ra(2, 1e-3)
        Qa = D.get_rotations(into=Qw).copy()
        b = D.irls(4, SIG, 30, 1e-3)
        Qb, wb = D.get_rotations(into=Qw), D.get_weights()
        st = D.stats()
    ra = O.l1ra(S["QQ"], S["I"], Q0, f, 2, 1e-3)
    rb = O.irls(S["QQ"], S["I"], ra["Q"], f, 4, SIG, 30, 1e-3)
    assert (a["iters"], b["iters"]) == (ra["iters"], rb["iters"])
    assert synth.angular_distance(Qa, ra["Q"]).max() < 1e-9
    assert synth.angular_distance(Qb, rb["Q"]).max() < 1e-9
    np.testing.assert_allclose(wb, rb["weights"], rtol=1e-7)
    assert st["direct_solves"] > 0 and st["pcg_iters"] == 0


# ---- loop closures on the sharded direct solver (round 5) ---------------------------------------------------------
def closure_problem(n, m, nclose, wrong, seed=7):
    S = synth.closure_graph(n, m, nclose, seed, wrong)
    Q = np.zeros((n, 4)); Q[:, 3] = 1; Q[0] = S["Qgt"][0]
    ral.init_mst(Q, S["QQ"], S["I"], 1)
    return S, Q


@pytest.mark.parametrize("n,m,nclose,wrong,world", [(3000, 12000, 5, 1, 2), (3000, 45000, 20, 3, 4), (4000, 80000, 40, 4, 3),
                                                    (5000, 20000, 64, 6, 6), (5000, 20000, 65, 6, 5), (5000, 20000, 96, 6, 4), (5000, 20000, 97, 6, 3),
                                                    (6000, 60000, 300, 20, 8), (2511, 74830, 100, 8, 2),
                                                    (20000, 300000, 1000, 40, 8), (20000, 300000, 1000, 40, 1)])
def test_loopback_sharded_sequence_with_closures_matches_the_oracle(n, m, nclose, wrong, world):
    """A view sequence WITH loop closures (what src/IRotAvg.cpp:371-378 re-solves on every closure) in 1 ... 8 shards:
    the sharded direct solver carries them by the Woodbury correction across the ranks -- a rank's step programs on its
    own levels, one summed buffer, the separator system with the closures' columns (bcr.hip). Until round 5 one closure
    sent the shards to the sharded PCG. 5 ... 64 closures: the Woodbury system in LDS; more: the blocked sweep; blocks
    of 8 / 16 / 24 / 32; some closures wrong. l1ra then irls against the ORACLE."""
    from oracle import oracle as O
    S, Q0 = closure_problem(n, m, nclose, wrong)
    with capi.DistGraph(S["I"], S["QQ"], n, 1, world, band_direct=1) as D:
        info = D.info()
        assert info["direct_block"] in (8, 16, 24, 32) and info["closures"] == nclose
        D.set_rotations(Q0)
        a = D.l1ra(2, 1e-3)
        Qa = D.get_rotations(into=Q0.copy())
        b = D.irls(4, SIG, 50, 1e-3)
        Qb, wb = D.get_rotations(into=Q0.copy()), D.get_weights()
        st = D.stats()
    assert st["direct_solves"] > 0 and st["pcg_iters"] == 0, st
    ra = O.l1ra(S["QQ"], S["I"], Q0, 1, 2, 1e-3)
    rb = O.irls(S["QQ"], S["I"], ra["Q"], 1, 4, SIG, 50, 1e-3)
    assert (a["iters"], b["iters"]) == (ra["iters"], rb["iters"])
    np.testing.assert_allclose(a["scores"], ra["scores"], rtol=1e-7)
    np.testing.assert_allclose(b["scores"], rb["scores"], rtol=1e-7, atol=1e-12)
    assert synth.angular_distance(Qa, ra["Q"]).max() < 1e-9
    assert synth.angular_distance(Qb, rb["Q"]).max() < 1e-9
    np.testing.assert_allclose(wb, rb["weights"], rtol=1e-7, atol=1e-12)


def test_sharded_closures_at_the_shard_boundaries_match_the_oracle():
    """Closures placed where the ranks' bookkeeping has its corners (4 shards of 1536 views, blocks of 8): an endpoint
    in a rank's LAST block (the separator: no local elimination at all), in block 0 behind a rank boundary, both
    endpoints on one rank, neighbouring ranks, the first and the last rank, two closures sharing a view (one row's
    diagonal loses two ghost weights), a closure from the block next to a boundary to the previous rank's last block
    (a ghost that is ALSO a separator row), a wrong closure."""
    from oracle import oracle as O
    n, world = 6000, 4
    S = synth.make_graph(n, 24000, 0.0, seed=3)
    c = 1536                                                # chunk_of(5999, 4, 192)
    pairs = [(c - 3, 3 * c + 40), (c + 2, 2 * c + 700), (100, 900), (c - 700, c + 650), (5, n - 2), (200, 3 * c + 40),
             (c - 2, c + 35), (2 * c + 5, 2 * c + 1400), (3 * c - 1, n - 700), (50, 2 * c - 1)]
    a = np.array([p[0] for p in pairs]) + 1
    b = np.array([p[1] for p in pairs]) + 1                 # (+ 1: free view r is view r + f)
    QQc = synth.qmul(S["Qgt"][b], synth.qconj(S["Qgt"][a]))
    QQc[3] = QQc[6] = np.array([0.5, -0.5, 0.5, 0.5])      # two wrong ones: a long one (the chain bends) and a short one
    I = np.concatenate([S["I"], np.stack([a, b], 1)]).astype(np.int32)
    QQ = np.concatenate([S["QQ"], QQc])
    order = np.lexsort((np.arange(len(I)), I[:, 1]))
    I, QQ = I[order], QQ[order]
    Q0 = np.zeros((n, 4)); Q0[:, 3] = 1; Q0[0] = S["Qgt"][0]
    ral.init_mst(Q0, QQ, I, 1)
    for cost in (4, 12):                                    # Geman-McClure; Talwar (weights of exactly 0: dead closures)
        with capi.DistGraph(I, QQ, n, 1, world, band_direct=1) as D:
            assert D.info()["direct_block"] == 8 and D.info()["closures"] == len(pairs)
            D.set_rotations(Q0)
            r = D.irls(cost, SIG, 50, 1e-3)
            Q, w = D.get_rotations(into=Q0.copy()), D.get_weights()
            st = D.stats()
        assert st["direct_solves"] == r["iters"] and st["pcg_iters"] == 0
        ro = O.irls(QQ, I, Q0, 1, cost, SIG, 50, 1e-3)
        assert r["iters"] == ro["iters"]
        np.testing.assert_allclose(r["scores"], ro["scores"], rtol=1e-7, atol=1e-12)
        assert synth.angular_distance(Q, ro["Q"]).max() < 1e-9
        np.testing.assert_allclose(w, ro["weights"], rtol=1e-7, atol=1e-12)
        if cost == 12:                                      # ... which Talwar switches off: a dead row of the Woodbury system
            k6 = np.flatnonzero((I[:, 0] == a[6]) & (I[:, 1] == b[6]))
            assert len(k6) == 1 and w[k6[0]] == 0


def test_sharded_closures_limits_and_switch(monkeypatch):
    """2049 closures are one too many (the single-GPU plan's limit): the shards take the sharded PCG, as they do when
    IROTAVG_DIST_NO_CLOSURES is set; same answers either way."""
    S, Q0 = closure_problem(12000, 48000, 2049, 0, seed=11)
    with capi.DistGraph(S["I"], S["QQ"], 12000, 1, 2, band_direct=1) as D:
        assert D.info()["direct_block"] == 0 and D.info()["closures"] == 0
    S, Q0 = closure_problem(6000, 60000, 30, 3)
    res = []
    for off in (False, True):
        if off:
            monkeypatch.setenv("IROTAVG_DIST_NO_CLOSURES", "1")
        with capi.DistGraph(S["I"], S["QQ"], 6000, 1, 3, band_direct=1) as D:
            assert (D.info()["direct_block"] == 0) == off
            D.set_rotations(Q0)
            r = D.irls(4, SIG, 50, 1e-3)
            res.append((r["iters"], D.get_rotations(into=Q0.copy()), D.stats()["pcg_iters"]))
    assert res[0][0] == res[1][0] and res[0][2] == 0 and res[1][2] > 0
    assert synth.angular_distance(res[0][1], res[1][1]).max() < 1e-8


def test_rccl_transport_single_rank_closures_sum_one_buffer():
    """The closures' exchange on the RCCL wire: ncclAllReduce (sum) of the one buffer behind the records' all-gather --
    a 1-rank communicator on the one GPU of a test box runs that code."""
    n, m = 8000, 120000
    S, Q0 = closure_problem(n, m, 80, 6)
    uid = capi.DistGraph.unique_id()
    with capi.DistGraph(S["I"], S["QQ"], n, 1, 1, rank=0, unique_id=uid, band_direct=1) as D:
        assert D.info()["direct_block"] > 0 and D.info()["closures"] == 80
        D.set_rotations(Q0)
        b = D.irls(4, SIG, 50, 1e-3)
        Qb = D.get_rotations(into=Q0.copy())
        assert D.stats()["direct_solves"] > 0 and D.stats()["pcg_iters"] == 0
    with capi.Graph(S["I"], S["QQ"], n, 1, band_direct=1) as G:
        G.set_rotations(Q0)
        a = G.irls(4, SIG, 50, 1e-3)
        Qa = G.get_rotations()
    assert a["iters"] == b["iters"]
    assert synth.angular_distance(Qa, Qb).max() < 1e-9


def _fuzz_case(seed, case, closures_max):
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("fuzz_sharded_direct", os.path.join(os.path.dirname(__file__), "..", "tools",
                                                                                      "fuzz_sharded_direct.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.case_of(seed, case, closures_max)


def test_a_closure_solve_that_lost_digits_is_repaired_on_both_handles():
    """fuzz campaign (seed 21, --closures-max 400), case 46: 38k views, band 24, 365 closures (18 of them wrong), Welsch.
    In the third iteration whole stretches of the band sit at the weight floor (w^2 = 1e-8) and the closures hold them:
    the Woodbury form cancels digits -- relative residual 4e-7, rotations 5e-5 rad (one GPU) / 3e-4 rad (5 shards) off
    the oracle until round 5. Now the residual of the FULL system gates the step: the single-GPU handle repeats the
    solve by conjugate gradients with the direct solve as the preconditioner, the shards refine it -- both within
    5e-7 rad of the oracle (what the conditioning of this graph leaves: 1.4e-7 measured)."""
    from oracle import oracle as O
    c = _fuzz_case(21, 46, 400)
    n, f, I, QQ, Q0 = c["n"], c["f"], c["I"], c["QQ"], c["Q0"]
    assert (c["cost"], c["nclose"], c["world"]) == (13, 366, 5)
    with capi.Graph(I, QQ, n, f, band_direct=1) as G:
        G.set_rotations(Q0)
        ra = G.irls(13, SIG, 3, 1e-3)
        Qa, sa = G.get_rotations(), G.stats()
    with capi.DistGraph(I, QQ, n, f, 5, band_direct=1) as D:
        assert D.info()["closures"] == 365   # (one of the 366 ends at a fixed view)
        D.set_rotations(Q0)
        rb = D.irls(13, SIG, 3, 1e-3)
        Qb, sb = D.get_rotations(into=Q0.copy()), D.stats()
    ro = O.irls(QQ, I, Q0, f, 13, SIG, 3, 1e-3)
    assert ra["iters"] == rb["iters"] == ro["iters"] == 3
    assert sa["direct_guarded"] >= 1 and sa["direct_dead_pivots"] == 0     # the gate was the residual, not a dead pivot
    assert synth.angular_distance(Qa, ro["Q"]).max() < 5e-7
    assert synth.angular_distance(Qb, ro["Q"]).max() < 5e-7


def test_a_band_part_next_to_singular_is_an_error_not_a_wrong_answer():
    """fuzz campaign (seed 22, --closures-max 1000), case 69: a THIN chain (band 3, 30k views) with 928 closures, 46 of
    them wrong, Welsch: from the third iteration on the band part alone is next to singular (the closures are the
    structure) and the Woodbury solve is no approximate inverse -- until round 5 the direct path returned rotations
    0.08 rad (one GPU) off the oracle without a word. Now both handles say IROTAVG_ERR_SOLVER and leave the rotations as
    they were. (The multigrid-PCG does not converge on this chain either -- DESIGN.md section 2 lists the class; the
    fall-back of the one-shot calls to the iterative solver is tested on a graph it takes, test_gpu_band_direct.py.)"""
    from oracle import oracle as O
    c = _fuzz_case(22, 69, 1000)
    n, f, I, QQ, Q0 = c["n"], c["f"], c["I"], c["QQ"], c["Q0"]
    assert (c["cost"], c["nclose"], c["world"]) == (13, 938, 8)
    for make in (lambda: capi.Graph(I, QQ, n, f, band_direct=1), lambda: capi.DistGraph(I, QQ, n, f, 8, band_direct=1)):
        with make() as H:
            get = (lambda: H.get_rotations()) if isinstance(H, capi.Graph) else (lambda: H.get_rotations(into=Q0.copy()))
            H.set_rotations(Q0)
            H.irls(13, SIG, 2, 1e-3)                            # two iterations are fine ...
            Q2 = get()
            H.set_rotations(Q0)
            with pytest.raises(capi.IrotavgError) as ei:        # ... the third one's system is not
                H.irls(13, SIG, 3, 1e-3)
            assert ei.value.code == capi.ERR_SOLVER
            # the rotations are what the two good iterations left: the failed step was not taken
            assert synth.angular_distance(Q2, get()).max() < 1e-9


@pytest.mark.parametrize("world,n,m", [(2, 20000, 300000), (8, 100000, 2000000), (5, 30000, 120000)])
def test_gathered_halo_equals_the_point_to_point_one_bit_for_bit(world, n, m):
    """Round 6: the halo of a closure-free sharded sequence is ONE all-gather of a fixed boundary record per rank
    ([to the lower neighbour | to the upper neighbour], block-size slots each) instead of point-to-point messages between
    neighbours (IROTAVG_DIST_HALO_P2P=1 in a process of its own gives those back). Same values into the same ghost slots:
    l1ra (halo of dx) then irls (halo of the step) bit for bit, in the loopback -- the RCCL form differs from it by the
    ncclAllGather call alone, which the 1-rank communicator test below runs."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, hashlib; sys.path.insert(0, %r); import numpy as np\n"
            "from tests.test_gpu_dist import problem, SIG\n"
            "from irotavg_amd import capi\n"
            "S, Q0 = problem(%d, %d, 0.0)\n"
            "with capi.DistGraph(S['I'], S['QQ'], %d, 1, %d) as D:\n"
            "    info = D.info(); D.set_rotations(Q0)\n"
            "    a = D.l1ra(1, 1e-3); b = D.irls(4, SIG, 50, 1e-3)\n"
            "    Q, w = D.get_rotations(into=Q0.copy()), D.get_weights()\n"
            "print(info['halo'].split()[0], info['direct_block'], a['iters'], b['iters'],\n"
            "      hashlib.sha256(np.ascontiguousarray(Q).tobytes() + np.ascontiguousarray(w).tobytes()).hexdigest())\n"
            % (root, n, m, n, world))
    outs = []
    for p2p in (False, True):
        env = dict(os.environ)
        env.pop("IROTAVG_DIST_HALO_P2P", None)
        if p2p:
            env["IROTAVG_DIST_HALO_P2P"] = "1"
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=root, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append(r.stdout.strip().splitlines()[-1].split())
    assert outs[0][0] == "all-gather" and outs[1][0] == "point-to-point"
    assert int(outs[0][1]) > 0                         # the sharded direct solver
    assert outs[0][1:] == outs[1][1:]                  # block size, iteration counts, every bit of rotations and weights


def test_rccl_single_rank_keeps_the_point_to_point_form_and_closures_do_too():
    """world = 1 has no halo; a sharded sequence WITH closures has ghosts on distant ranks and keeps the point-to-point
    exchange (the gathered record holds the two neighbours' boundaries only)."""
    n, m = 20000, 300000
    S, Q0 = problem(n, m, 0.0)
    S2 = synth.add_closures(S, 12, 3, 1)
    with capi.DistGraph(S2["I"], S2["QQ"], n, 1, 4, band_direct=1) as D:
        info = D.info()
        assert info["direct_block"] > 0 and info["closures"] > 0 and info["halo"] == "point-to-point"
    with capi.DistGraph(S["I"], S["QQ"], n, 1, 4, band_direct=1) as D:
        assert D.info()["halo"].startswith("all-gather")

"""GPU parity tests of the banded direct solver (irotavg_amd/csrc/bcr.hip): the HIP path through the C ABI against the
CPU oracle where the linear systems of `irls` (ral/l1_irls.cpp:536-556, SuiteSparseQR in the reference) and of
`l1decode_pd` (:131-184, UMFPACK) are solved by block cyclic reduction instead of the PCG.

A graph takes this path when every edge between two free views spans at most 32 views (a sequence without loop
closures): block sizes 8 / 16 / 24 / 32 by the half-bandwidth. Tolerances: a direct fp64 solve -- linear-solve outputs
1e-9 relative, final rotations 1e-9 rad with IDENTICAL outer iteration counts, weights 1e-7.
"""
import ctypes as C

import numpy as np
import pytest

from irotavg_amd import capi, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu
SIG = 5 * np.pi / 180


def mst_init(G, n, f=1):
    Q = np.zeros((n, 4)); Q[:, 3] = 1; Q[:f] = G["Qgt"][:f]
    rc, Qm = O.init_mst(Q, G["QQ"], G["I"], f)
    assert rc == 0
    return Qm


def direct_stats(st, block):
    assert st["band_block"] == block, st
    assert st["direct_solves"] > 0 and st["pcg_solves"] == 0, st


# band 5 -> blocks of 8, 16 -> 16, 22 -> 24, 31 -> 32; 2510 / 2999 rows: neither a multiple of the block size
# (round 4: every multiple of four -- band 12 -> 12, 20 -> 20, 28 -> 28)
@pytest.mark.parametrize("n,m,block", [(3000, 12000, 8), (3000, 45000, 16), (3000, 63000, 24), (2511, 74830, 32),
                                       (3000, 33000, 12), (3000, 57000, 20), (3000, 81000, 28)])
def test_direct_solver_matches_oracle(n, m, block):
    S = synth.make_graph(n, m, 0.0, seed=3, p_band_out=0.02)
    Qm = mst_init(S, n)
    rng = np.random.default_rng(0)
    w = rng.uniform(0.1, 5.0, size=len(S["I"]))
    w[rng.choice(len(w), len(w) // 40, replace=False)] *= 1e-4   # weights over five decades
    ro = O.log_map(O.delta_rel(S["I"], S["QQ"], Qm))[:, :3]
    rc, Xo = O.ls_solve(n, 1, S["I"], w, ro)
    assert rc == 0
    with capi.Graph(S["I"], S["QQ"], n, 1, band_direct=1) as G:
        G.set_rotations(Qm)
        G.edge_residual()
        G.set_weights(w)
        X = G.ls_solve()
        direct_stats(G.stats(), block)
        assert np.abs(X - Xo).max() < 1e-9 * np.abs(Xo).max()
        G.set_rotations(Qm)
        a = G.l1ra(2, 1e-3)
        Qa = G.get_rotations()
        b = G.irls(4, SIG, 50, 1e-3)
        Qb = G.get_rotations()
        wb = G.get_weights()
        direct_stats(G.stats(), block)
    ra = O.l1ra(S["QQ"], S["I"], Qm, 1, 2, 1e-3)
    rb = O.irls(S["QQ"], S["I"], ra["Q"], 1, 4, SIG, 50, 1e-3)
    assert (a["iters"], b["iters"]) == (ra["iters"], rb["iters"])
    np.testing.assert_allclose(a["scores"], ra["scores"], rtol=1e-8)
    np.testing.assert_allclose(b["scores"], rb["scores"], rtol=1e-7)
    assert synth.angular_distance(Qa, ra["Q"]).max() < 1e-9
    assert synth.angular_distance(Qb, rb["Q"]).max() < 1e-9
    np.testing.assert_allclose(wb, rb["weights"], rtol=1e-7)


@pytest.mark.parametrize("cost", list(range(14)))
def test_every_cost_on_the_direct_path(cost):
    n, m = 2600, 39000
    S = synth.make_graph(n, m, 0.0, seed=8, p_band_out=0.03)
    Qm = mst_init(S, n)
    with capi.Graph(S["I"], S["QQ"], n, 1, band_direct=1) as G:
        G.set_rotations(Qm)
        r = G.irls(cost, SIG, 15, 1e-3)
        Q = G.get_rotations()
        w = G.get_weights()
        direct_stats(G.stats(), 16)
    ro = O.irls(S["QQ"], S["I"], Qm, 1, cost, SIG, 15, 1e-3)
    assert r["iters"] == ro["iters"]
    np.testing.assert_allclose(r["scores"], ro["scores"], rtol=1e-7)
    assert synth.angular_distance(Q, ro["Q"]).max() < 1e-9
    np.testing.assert_allclose(w, ro["weights"], rtol=1e-6, atol=1e-12)


def test_l1decode_pd_on_the_direct_path():
    n, m = 3000, 45000
    S = synth.make_graph(n, m, 0.0, seed=2)
    Qm = mst_init(S, n)
    ro = O.log_map(O.delta_rel(S["I"], S["QQ"], Qm))[:, :3]
    with capi.Graph(S["I"], S["QQ"], n, 1, band_direct=1) as G:
        for c in range(3):
            x, stuck = G.l1decode_pd(np.ascontiguousarray(ro[:, c]), 2)
            rc, xo, so = O.l1decode_pd(n, 1, S["I"], ro[:, c], 2)
            assert rc == 0 and stuck == so
            assert np.abs(x - xo).max() < 1e-9 * max(np.abs(xo).max(), 1e-300)
        direct_stats(G.stats(), 16)


@pytest.mark.parametrize("n", [9, 60, 64, 65, 190, 520, 4100])
def test_sizes_around_the_chunk_boundaries(n):
    """One block, one chunk (the top kernel at level 0), one chunk + one block, two and three levels; band 3 -> blocks
    of 8 rows, chunks of 64."""
    S = synth.make_graph(n, 3 * n - 6, 0.0, seed=n)
    Qm = mst_init(S, n)
    with capi.Graph(S["I"], S["QQ"], n, 1, band_direct=1) as G:
        G.set_rotations(Qm)
        a = G.l1ra(3, 1e-3)
        b = G.irls(4, SIG, 30, 1e-4)
        Q = G.get_rotations()
        direct_stats(G.stats(), 8)
    ra = O.l1ra(S["QQ"], S["I"], Qm, 1, 3, 1e-3)
    rb = O.irls(S["QQ"], S["I"], ra["Q"], 1, 4, SIG, 30, 1e-4)
    assert (a["iters"], b["iters"]) == (ra["iters"], rb["iters"])
    assert synth.angular_distance(Q, rb["Q"]).max() < 1e-9


def test_mixed_level_one():
    """36000 views in blocks of 8 = 563 chunks on 512 resident slots: the 51 surplus chunks are not reduced at level
    0 -- their 404 blocks enter level 1 next to the 512 separators (bcr_alloc)."""
    n, m = 36000, 4 * 36000 - 10
    S = synth.make_graph(n, m, 0.0, seed=12, p_band_out=0.01)
    Qm = mst_init(S, n)
    res = []
    for env in (None, "1"):
        import os
        if env:
            os.environ["IROTAVG_BCR_NO_MIXED"] = env
        try:
            with capi.Graph(S["I"], S["QQ"], n, 1) as G:
                G.set_rotations(Qm)
                b = G.irls(4, SIG, 30, 1e-3)
                res.append((b["iters"], G.get_rotations(), G.get_weights()))
                direct_stats(G.stats(), 8)
        finally:
            os.environ.pop("IROTAVG_BCR_NO_MIXED", None)
    rb = O.irls(S["QQ"], S["I"], Qm, 1, 4, SIG, 30, 1e-3)
    for it, Q, w in res:
        assert it == rb["iters"]
        assert synth.angular_distance(Q, rb["Q"]).max() < 1e-9
        np.testing.assert_allclose(w, rb["weights"], rtol=1e-7)


closure_graph = synth.closure_graph


# 5 ... 96 closures: the Woodbury system is solved in LDS (round 5; 64 until then); 97, 300, 1000: by the blocked Gauss-Jordan sweep (round 4:
# the closures' forward eliminations follow their paths up the elimination tree, bcr.hip; round 3 took at most 64,
# sixteen per re-factorisation); blocks of 8, 12, 16, 24 and 32
@pytest.mark.parametrize("n,m,nclose,wrong,block", [(3000, 12000, 5, 1, 8), (3000, 45000, 20, 3, 16),
                                                    (4000, 80000, 40, 4, 24), (5000, 20000, 64, 6, 8),
                                                    (5000, 20000, 65, 6, 8), (5000, 20000, 96, 6, 8), (5000, 20000, 97, 6, 8),
                                                    (6000, 60000, 300, 20, 12),
                                                    (2511, 74830, 100, 8, 32), (20000, 300000, 1000, 40, 16)])
def test_loop_closures_on_the_direct_path_match_oracle(n, m, nclose, wrong, block):
    """A sequence with a few loop closures -- the SLAM case (src/IRotAvg.cpp:371-378 re-solves the whole graph on
    every closure): the band part is factorised, the closures re-enter by the Woodbury correction (bcr_solve).
    Some closures are wrong: the robust weights must switch them off."""
    S = closure_graph(n, m, nclose, 7, wrong)
    Qm = mst_init(S, n)
    with capi.Graph(S["I"], S["QQ"], n, 1, band_direct=1) as G:
        assert G.direct_info()["closures"] == nclose
        G.set_rotations(Qm)
        a = G.l1ra(2, 1e-3)
        Qa = G.get_rotations()
        b = G.irls(4, SIG, 50, 1e-3)
        Qb, w = G.get_rotations(), G.get_weights()
        direct_stats(G.stats(), block)
    ra = O.l1ra(S["QQ"], S["I"], Qm, 1, 2, 1e-3)
    rb = O.irls(S["QQ"], S["I"], ra["Q"], 1, 4, SIG, 50, 1e-3)
    assert (a["iters"], b["iters"]) == (ra["iters"], rb["iters"])
    np.testing.assert_allclose(a["scores"], ra["scores"], rtol=1e-7)
    assert synth.angular_distance(Qa, ra["Q"]).max() < 1e-9
    assert synth.angular_distance(Qb, rb["Q"]).max() < 1e-9
    np.testing.assert_allclose(w, rb["weights"], rtol=1e-7, atol=1e-12)


def test_closure_with_zero_weight_and_too_many_closures():
    """Talwar sets the weight of a large residual to exactly 0 (ral/l1_irls.cpp:700-707): such a closure is absent
    from the operator -- its row of the Woodbury system is dead. 2049 closures are one too many: the handle solves
    iteratively."""
    n = 3000
    S = closure_graph(n, 30000, 6, 11, wrong=3)
    Qm = mst_init(S, n)
    with capi.Graph(S["I"], S["QQ"], n, 1, band_direct=1) as G:
        G.set_rotations(Qm)
        r = G.irls(12, SIG, 30, 1e-3)           # Talwar
        Q, w = G.get_rotations(), G.get_weights()
        direct_stats(G.stats(), 12)
    ro = O.irls(S["QQ"], S["I"], Qm, 1, 12, SIG, 30, 1e-3)
    assert (w == 0).sum() >= 3 and r["iters"] == ro["iters"]
    assert synth.angular_distance(Q, ro["Q"]).max() < 1e-9
    np.testing.assert_array_equal(w == 0, ro["weights"] == 0)
    S = closure_graph(12000, 48000, 2049, 11)
    with capi.Graph(S["I"], S["QQ"], 12000, 1, band_direct=1) as G:
        assert G.stats()["band_block"] == 0 and G.direct_info()["block"] == 0


@pytest.mark.parametrize("stretch", [1, 3])
def test_views_held_only_by_a_closure_match_oracle(stretch):
    """The known limit of round 3, guarded in round 4: Talwar's exact-zero weights (ral/l1_irls.cpp:700-707) cut a view
    (or a stretch of views) off ALL its band neighbours while a loop closure still holds it. The band part of the operator
    is then singular and its factor has a dead pivot, but the full system -- which the reference always solves,
    ral/l1_irls.cpp:536-556 -- is not: the handle notices (dead pivots are counted on the device), repeats that solve by
    conjugate gradients on the full operator preconditioned with the regularised direct solve, and says so in its stats."""
    n = 3000
    S = closure_graph(n, 12000, 6, 5)
    I, QQ = S["I"].copy(), S["QQ"].copy()
    v0 = 1500
    cut = np.arange(v0, v0 + stretch)
    rng = np.random.default_rng(9)
    inside = np.isin(I[:, 0], cut) & np.isin(I[:, 1], cut)
    touching = (np.isin(I[:, 0], cut) | np.isin(I[:, 1], cut)) & ~inside & (np.abs(I[:, 0] - I[:, 1]) <= 32)
    assert touching.sum() >= 4
    R = rng.normal(size=(int(touching.sum()), 4))
    QQ[touching] = R / np.linalg.norm(R, axis=1, keepdims=True)        # every band edge out of the stretch is wrong ...
    far = np.array([[200, v0]], dtype=np.int32)                        # ... and one correct closure holds it
    QQf = synth.qmul(S["Qgt"][v0:v0 + 1], synth.qconj(S["Qgt"][200:201]))
    I = np.concatenate([I, far]).astype(np.int32)
    QQ = np.concatenate([QQ, QQf])
    order = np.lexsort((np.arange(len(I)), I[:, 1]))
    I, QQ = I[order], QQ[order]
    Q = np.zeros((n, 4)); Q[:, 3] = 1; Q[0] = S["Qgt"][0]
    rc, Qm = O.init_mst(Q, QQ, I, 1)
    assert rc == 0
    with capi.Graph(I, QQ, n, 1, band_direct=1) as G:
        assert G.direct_info()["closures"] == 7
        G.set_rotations(Qm)
        r = G.irls(12, SIG, 30, 1e-3)           # Talwar
        Qg, w = G.get_rotations(), G.get_weights()
        st = G.stats()
    ro = O.irls(QQ, I, Qm, 1, 12, SIG, 30, 1e-3)
    assert st["direct_guarded"] >= 1 and st["direct_dead_pivots"] >= 1, st
    assert st["band_block"] == 8 and st["direct_solves"] > 0
    assert r["iters"] == ro["iters"], (r["iters"], ro["iters"], r["scores"], ro["scores"])
    np.testing.assert_array_equal(w == 0, ro["weights"] == 0)
    assert synth.angular_distance(Qg, ro["Q"]).max() < 1e-8
    # the stretch really hangs on the closure alone: all its band edges to the outside ended at weight 0
    out = (np.isin(I[:, 0], cut) ^ np.isin(I[:, 1], cut)) & (np.abs(I[:, 0] - I[:, 1]) <= 32)
    assert (w[out] == 0).all() and w[(I[:, 0] == 200) & (I[:, 1] == v0)][0] > 0
    # the same on SHARDS (round 5: the closure crosses from rank 0 to rank 1 of two, and both endpoints lie on one rank of
    # three): the dead pivot is noticed by every rank (the counts are summed with the closures' buffer), the solve is
    # repeated with the regularised factor and refined on the sharded operator (M = A + E, E positive semi-definite:
    # the rounds contract)
    for world in (2, 3):
        with capi.DistGraph(I, QQ, n, 1, world, band_direct=1) as D:
            assert D.info()["direct_block"] == 8 and D.info()["closures"] == 7
            D.set_rotations(Qm)
            rd = D.irls(12, SIG, 30, 1e-3)
            Qd, wd = D.get_rotations(into=Qm.copy()), D.get_weights()
            sd = D.stats()
        assert sd["direct_solves"] > 0 and sd["pcg_iters"] == 0
        assert rd["iters"] == ro["iters"]
        np.testing.assert_array_equal(wd == 0, ro["weights"] == 0)
        assert synth.angular_distance(Qd, ro["Q"]).max() < 1e-8


def test_blocks_of_32_with_a_mixed_level_match_the_iterative_solver():
    """70k views, band 29 -> blocks of 32 (one workgroup per CU: LDS), 274 chunks on 256 slots: 18 chunks' blocks enter
    level 1 unreduced. Too large for the oracle's Cholesky to be quick: both solvers of the handle against each
    other (each is held against the oracle elsewhere), and the normal equations by the oracle's mat-vec."""
    n, m = 70000, 2000000
    S = synth.make_graph(n, m, 0.0, seed=3, p_band_out=0.01)
    Q0 = np.zeros((n, 4)); Q0[:, 3] = 1; Q0[0] = S["Qgt"][0]
    from irotavg_amd import ral
    ral.init_mst(Q0, S["QQ"], S["I"], 1)
    out = {}
    for bd in (0, -1):
        with capi.Graph(S["I"], S["QQ"], n, 1, band_direct=bd) as G:
            if bd == 0:
                info = G.direct_info()
                assert info["block"] == 32 and info["levels"][1]["reduced"] < info["levels"][1]["blocks"], info
                G.set_rotations(Q0)
                G.edge_residual()
                w = np.random.default_rng(1).uniform(0.2, 3.0, size=m)
                G.set_weights(w)
                X = G.ls_solve()
                ro = O.log_map(O.delta_rel(S["I"], S["QQ"], Q0))[:, :3]
                b = O.make_A(n, 1, S["I"]).T @ ((w * w)[:, None] * ro)
                rel = np.linalg.norm(O.normal_matvec(n, 1, S["I"], w, X) - b, axis=0) / np.linalg.norm(b, axis=0)
                assert rel.max() < 5e-10, rel
            G.set_rotations(Q0)
            r = G.irls(4, SIG, 50, 1e-3)
            out[bd] = (r["iters"], G.get_rotations(), G.get_weights())
    assert out[0][0] == out[-1][0]
    assert synth.angular_distance(out[0][1], out[-1][1]).max() < 1e-8
    np.testing.assert_allclose(out[0][2], out[-1][2], rtol=1e-6)


def test_fixed_views_flipped_and_duplicate_edges():
    """f = 4 fixed views (rows = views - f; edges to fixed views only reach the diagonal and the right-hand side),
    30 % of the edges given as (j, i) -- some then have their SECOND endpoint fixed, which make_A drops
    (ral/l1_irls.cpp:770-771) and make_AtA keeps (:825-835) --, duplicate edges."""
    n, f = 1500, 4
    S = synth.make_graph(n, 15000, 0.0, seed=13, p_band_out=0.02)
    rng = np.random.default_rng(5)
    I, QQ = S["I"].copy(), S["QQ"].copy()
    flip = rng.random(len(I)) < 0.3
    I[flip] = I[flip][:, ::-1]
    QQ[flip] = synth.qconj(QQ[flip])
    I = np.concatenate([I, I[:80]]).astype(np.int32)
    QQ = np.concatenate([QQ, QQ[:80]])
    assert ((I[:, 1] < f) & (I[:, 0] >= f)).sum() > 0
    Q = np.zeros((n, 4)); Q[:, 3] = 1; Q[:f] = S["Qgt"][:f]
    rc, Qm = O.init_mst(Q, QQ, I, f)
    assert rc == 0
    with capi.Graph(I, QQ, n, f, band_direct=1) as G:
        G.set_rotations(Qm)
        a = G.l1ra(3, 1e-3)
        b = G.irls(4, SIG, 50, 1e-3)
        Qg = G.get_rotations()
        w = G.get_weights()
        direct_stats(G.stats(), 12)
    ra = O.l1ra(QQ, I, Qm, f, 3, 1e-3)
    rb = O.irls(QQ, I, ra["Q"], f, 4, SIG, 50, 1e-3)
    assert (a["iters"], b["iters"]) == (ra["iters"], rb["iters"])
    assert synth.angular_distance(Qg, rb["Q"]).max() < 1e-9
    np.testing.assert_allclose(w, rb["weights"], rtol=1e-7)
    np.testing.assert_array_equal(Qg[:f], Qm[:f])


def test_isolated_views_are_dead_pivots():
    """A free view without any edge has an empty row: its pivot is dead, its unknown solves to 0 and the view stays
    where it was -- what SPQR's rank detection and the oracle's dead pivot do."""
    n = 900
    S = synth.make_graph(n, 9000, 0.0, seed=21)
    lonely = [17, 400, 401, 899]
    keep = ~np.isin(S["I"], lonely).any(axis=1)
    I = np.concatenate([S["I"][keep], [[500, 500]]]).astype(np.int32)   # and a self loop (make_A keeps its -1, :770-776)
    QQ = np.concatenate([S["QQ"][keep], [[0, 0, 0, 1.0]]])
    rng = np.random.default_rng(1)
    Q0 = synth.qmul(synth.qexp(rng.normal(scale=0.03, size=(n, 3))), S["Qgt"]); Q0[0] = S["Qgt"][0]
    with capi.Graph(I, QQ, n, 1, band_direct=1) as G:
        G.set_rotations(Q0)
        r = G.irls(4, SIG, 30, 1e-6)
        Q = G.get_rotations()
        w = G.get_weights()
        direct_stats(G.stats(), 12)
    ro = O.irls(QQ, I, Q0, 1, 4, SIG, 30, 1e-6)
    assert r["iters"] == ro["iters"]
    assert synth.angular_distance(Q, ro["Q"]).max() < 1e-9
    np.testing.assert_allclose(w, ro["weights"], rtol=1e-7)
    np.testing.assert_array_equal(Q[lonely], Q0[lonely])


def test_a_non_finite_solve_leaves_the_rotations_alone():
    """A relative rotation that is not a number poisons the normal equations; the direct solve then returns non-finite
    steps, irls reports IROTAVG_ERR_SOLVER -- and the handle still holds the rotations it was given (the step of a view is
    applied only when it is finite; the reference would store zero quaternions, ral/l1_irls.cpp:491, and exit)."""
    n = 3000
    S = synth.make_graph(n, 45000, 0.0, seed=8)
    QQ = S["QQ"].copy()
    QQ[1234] = np.nan
    rng = np.random.default_rng(2)
    Q0 = synth.qmul(synth.qexp(rng.normal(scale=0.03, size=(n, 3))), S["Qgt"]); Q0[0] = S["Qgt"][0]
    with capi.Graph(S["I"], QQ, n, 1, band_direct=1) as G:
        G.set_rotations(Q0)
        with pytest.raises(capi.IrotavgError) as e:
            G.irls(4, SIG, 30, 1e-6)
        assert e.value.code == capi.ERR_SOLVER
        assert G.stats()["direct_solves"] >= 1
        np.testing.assert_array_equal(G.get_rotations(), Q0)


@pytest.mark.parametrize("n,m,nclose", [(3000, 45000, 0), (20000, 400000, 0), (4000, 80000, 40), (20000, 300000, 1000)])
def test_residual_of_the_last_direct_solve(n, m, nclose):
    """The direct solver has no residual test of its own; irotavg_graph_direct_residual computes ||b - A x|| / ||b|| of the
    handle's most recent direct solve on demand (one pass over level 0, the closures' entries included). A solve of a
    well-conditioned IRLS system is accurate to rounding; a handle whose systems run through the PCG says BAD_ARG."""
    S = closure_graph(n, m, nclose, 7, nclose // 10) if nclose else synth.make_graph(n, m, 0.0, seed=5)
    Qm = mst_init(S, n)
    with capi.Graph(S["I"], S["QQ"], n, 1, band_direct=1) as G:
        G.set_rotations(Qm)
        with pytest.raises(capi.IrotavgError) as e:   # nothing solved yet
            G.direct_residual()
        assert e.value.code == capi.ERR_BAD_ARG
        G.irls(4, SIG, 50, 1e-3)
        rr = G.direct_residual()
        assert rr.shape == (3,) and np.all(np.isfinite(rr)) and rr.max() < (1e-9 if nclose else 1e-11), rr
        np.testing.assert_array_equal(G.stats()["last_relres"], rr)
    with capi.Graph(S["I"], S["QQ"], n, 1, band_direct=-1) as G:
        G.set_rotations(Qm)
        G.irls(4, SIG, 3, 1e-3)
        with pytest.raises(capi.IrotavgError) as e:
            G.direct_residual()
        assert e.value.code == capi.ERR_BAD_ARG


def test_single_launch_upper_reduction_is_admitted_by_reservation(monkeypatch):
    """The workgroups of the single-launch upper reduction wait for each other, so every such launch reserves its share of
    the device first (bcr.hip, bcr_up_reserve); one that is not admitted runs level by level. Both routes are the same
    arithmetic: with the capacity forced to nothing (IROTAVG_BCR_UP_CAP=0), and with a capacity that admits only the first
    of l1ra's three chains, rotations and weights are bit-identical to the default."""
    n, m = 20000, 400000            # five levels: levels 1-4 in one launch
    S = synth.make_graph(n, m, 0.0, seed=6)
    Qm = mst_init(S, n)

    def run():
        with capi.Graph(S["I"], S["QQ"], n, 1) as G:
            assert len(G.direct_info()["levels"]) >= 3
            G.set_rotations(Qm)
            G.l1ra(2, 1e-3)
            G.irls(4, SIG, 50, 1e-3)
            return G.get_rotations(), G.get_weights()

    ref = run()
    for cap in ("0", "70", "1024"):   # 20000 views: level 1 has 14 chunks = 55 / 1024 of the device per chain
        monkeypatch.setenv("IROTAVG_BCR_UP_CAP", cap)
        out = run()
        np.testing.assert_array_equal(out[0], ref[0])
        np.testing.assert_array_equal(out[1], ref[1])
    monkeypatch.delenv("IROTAVG_BCR_UP_CAP")
    # two handles alive at the same time give their shares back when they are idle: nothing is left reserved
    with capi.Graph(S["I"], S["QQ"], n, 1) as A, capi.Graph(S["I"], S["QQ"], n, 1) as Bh:
        for G in (A, Bh):
            G.set_rotations(Qm)
            G.irls(4, SIG, 50, 1e-3)
        np.testing.assert_array_equal(A.get_rotations(), Bh.get_rotations())
    monkeypatch.setenv("IROTAVG_BCR_UP_CAP", "56")   # exactly one launch of this size
    out = run()
    np.testing.assert_array_equal(out[0], ref[0])


# round 5: the last level is ONE workgroup of up to sixteen blocks that also makes its way back (bcr_top_body). Band 3 ->
# blocks of 8 rows: n - 1 rows = 8 blocks per chunk x 8 rows; two levels (level 0 + a top of N = 2 ... 16 blocks, a launch
# of its own) and three (level 0, a level of chunks, the top: the single-launch upper reduction runs both)
@pytest.mark.parametrize("n", [65 + 64 * k for k in (1, 2, 3, 7, 8, 9, 12, 14, 15)] + [1025 + 512 * k for k in (0, 3, 8, 9, 13, 14)])
def test_single_workgroup_top_of_up_to_sixteen_blocks_matches_oracle(n, monkeypatch):
    S = synth.make_graph(n, 3 * n - 6, 0.0, seed=n, p_band_out=0.02)
    Qm = mst_init(S, n)
    rng = np.random.default_rng(n)
    w = rng.uniform(0.1, 5.0, size=len(S["I"]))
    w[rng.choice(len(w), len(w) // 30, replace=False)] *= 1e-4
    ro = O.log_map(O.delta_rel(S["I"], S["QQ"], Qm))[:, :3]
    rc, Xo = O.ls_solve(n, 1, S["I"], w, ro)
    assert rc == 0
    out = []
    for no16 in (False, True):
        if no16:
            monkeypatch.setenv("IROTAVG_BCR_NO_TOP16", "1")
        with capi.Graph(S["I"], S["QQ"], n, 1, band_direct=1) as G:
            lev = G.direct_info()["levels"]
            if not no16:   # the top holds what the level below left: its chunks (<= 16)
                assert 2 <= lev[-1]["blocks"] <= 16 and lev[-1]["blocks"] == lev[-2]["chunks"], lev
                assert lev[-2]["blocks"] > 8, lev
            G.set_rotations(Qm)
            G.edge_residual()
            G.set_weights(w)
            X = G.ls_solve()
            assert np.abs(X - Xo).max() < 1e-9 * np.abs(Xo).max()
            G.set_rotations(Qm)
            a = G.l1ra(2, 1e-3)
            b = G.irls(4, SIG, 30, 1e-4)
            out.append((a["iters"], b["iters"], G.get_rotations(), G.get_weights()))
            direct_stats(G.stats(), 8)
    ra = O.l1ra(S["QQ"], S["I"], Qm, 1, 2, 1e-3)
    rb = O.irls(S["QQ"], S["I"], ra["Q"], 1, 4, SIG, 30, 1e-4)
    for ia, ib, Q, wts in out:
        assert (ia, ib) == (ra["iters"], rb["iters"])
        assert synth.angular_distance(Q, rb["Q"]).max() < 1e-9
        np.testing.assert_allclose(wts, rb["weights"], rtol=1e-7)


def test_a_wait_that_gives_up_is_repeated_level_by_level(monkeypatch):
    """k_bcr_reduce_up's workgroups wait for each other; the reservation (bcr_up_reserve) only knows this process. A
    workgroup whose wait does not end says so in a device word, the top workgroup then poisons the whole solution, no view
    takes a step, the kernel behind the solve leaves weights and residuals alone, and the host repeats the iteration level
    by level (stats.direct_up_fallbacks) -- in irls and inside l1decode_pd. IROTAVG_BCR_FAKE_UP_FAIL sets the word as a
    timed-out wait would; the results are bit-identical to an undisturbed run (same arithmetic on both routes)."""
    n, m = 20000, 400000
    S = synth.make_graph(n, m, 0.0, seed=16, p_band_out=0.01)
    Qm = mst_init(S, n)

    def run(l1):
        with capi.Graph(S["I"], S["QQ"], n, 1) as G:
            G.set_rotations(Qm)
            if l1:
                G.l1ra(2, 1e-3)
            r = G.irls(4, SIG, 50, 1e-3)
            return r["iters"], G.get_rotations(), G.get_weights(), G.stats()

    for l1 in (False, True):
        monkeypatch.delenv("IROTAVG_BCR_FAKE_UP_FAIL", raising=False)
        ref = run(l1)
        assert ref[3]["direct_up_fallbacks"] == 0
        monkeypatch.setenv("IROTAVG_BCR_FAKE_UP_FAIL", "1")
        out = run(l1)
        assert out[3]["direct_up_fallbacks"] == 1, out[3]
        assert out[0] == ref[0]
        np.testing.assert_array_equal(out[1], ref[1])
        np.testing.assert_array_equal(out[2], ref[2])


def test_direct_path_is_bitwise_reproducible():
    S = synth.make_graph(5000, 100000, 0.0, seed=4, p_band_out=0.02)
    Qm = mst_init(S, 5000)
    outs = []
    for _ in range(2):
        with capi.Graph(S["I"], S["QQ"], 5000, 1) as G:
            G.set_rotations(Qm)
            G.l1ra(1, 1e-3)
            G.irls(1, SIG, 10, 1e-3)
            outs.append((G.get_rotations(), G.get_weights()))
            direct_stats(G.stats(), 24)   # chosen without the option: more than 2048 free views
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])


def test_option_and_environment_switch_the_path(monkeypatch):
    S = synth.make_graph(3000, 30000, 0.0, seed=5)
    Qm = mst_init(S, 3000)
    out = {}
    for tag, kw, env in [("direct", dict(), None), ("never", dict(band_direct=-1), None), ("env", dict(), "-1")]:
        if env:
            monkeypatch.setenv("IROTAVG_BAND_DIRECT", env)
        else:
            monkeypatch.delenv("IROTAVG_BAND_DIRECT", raising=False)
        with capi.Graph(S["I"], S["QQ"], 3000, 1, **kw) as G:
            G.set_rotations(Qm)
            r = G.irls(4, SIG, 50, 1e-3)
            out[tag] = (r["iters"], G.get_rotations(), G.stats())
    assert out["direct"][2]["band_block"] == 12 and out["direct"][2]["pcg_solves"] == 0
    for tag in ("never", "env"):
        assert out[tag][2]["band_block"] == 0 and out[tag][2]["direct_solves"] == 0 and out[tag][2]["pcg_solves"] > 0
        assert out[tag][0] == out["direct"][0]
        assert synth.angular_distance(out[tag][1], out["direct"][1]).max() < 1e-8
    # a loop closure stays on the direct path (Woodbury) unless the environment says otherwise
    I = np.concatenate([S["I"], [[10, 2900]]]).astype(np.int32)
    QQ = np.concatenate([S["QQ"], synth.qmul(S["Qgt"][2900:2901], synth.qconj(S["Qgt"][10:11]))])
    monkeypatch.delenv("IROTAVG_BAND_DIRECT", raising=False)
    with capi.Graph(I, QQ, 3000, 1, band_direct=1) as G:
        st = G.stats()
        assert st["band"] == 2890 and st["band_block"] == 12 and G.direct_info()["closures"] == 1
    monkeypatch.setenv("IROTAVG_BCR_NO_CLOSURES", "1")
    with capi.Graph(I, QQ, 3000, 1, band_direct=1) as G:
        assert G.stats()["band_block"] == 0


def test_one_shot_calls_take_the_direct_path_and_match_the_oracle():
    """irotavg_l1ra / irotavg_irls with host pointers (the reference's signatures, ral/l1_irls.hpp:100-107)."""
    from irotavg_amd import ral
    n, m = 5000, 60000
    S = synth.make_graph(n, m, 0.0, seed=31)
    Qm = mst_init(S, n)
    Q = capi.fmat(Qm)
    wts = np.zeros(len(S["I"]))
    it1, _ = ral.l1ra(S["QQ"], S["I"], None, Q, 1, 2, 1e-3)
    it2, _ = ral.irls(S["QQ"], S["I"], None, 4, SIG, Q, 1, 50, 1e-3, wts)
    ra = O.l1ra(S["QQ"], S["I"], Qm, 1, 2, 1e-3)
    rb = O.irls(S["QQ"], S["I"], ra["Q"], 1, 4, SIG, 50, 1e-3)
    assert (it1, it2) == (ra["iters"], rb["iters"])
    assert synth.angular_distance(Q, rb["Q"]).max() < 1e-9
    np.testing.assert_allclose(wts, rb["weights"], rtol=1e-7)


def test_time_kernel_reports_every_launch_of_a_direct_solve():
    S = synth.make_graph(20000, 300000, 0.0, seed=1)
    Qm = mst_init(S, 20000)
    ms = C.c_double(0)
    with capi.Graph(S["I"], S["QQ"], 20000, 1) as G:
        G.set_rotations(Qm)
        G.edge_residual()
        X = G.ls_solve()
        lib = capi.lib()
        assert lib.irotavg_graph_time_kernel(G._h, 19, 3, C.byref(ms)) == 0 and ms.value > 0
        total, levels = 0.0, 0
        for which in list(range(20, 30)) + list(range(40, 50)):
            if lib.irotavg_graph_time_kernel(G._h, which, 3, C.byref(ms)) == 0:
                total += ms.value
                levels += which < 40
        assert levels >= 3 and total > 0
        # the timed launches do not disturb the handle: the same solve again
        G.edge_residual()
        np.testing.assert_array_equal(G.ls_solve(), X)
    with capi.Graph(S["I"], S["QQ"], 20000, 1, band_direct=-1) as G:
        assert capi.lib().irotavg_graph_time_kernel(G._h, 19, 1, C.byref(ms)) == capi.ERR_BAD_ARG


def test_one_shot_call_and_rotavg_fall_back_to_the_iterative_solver_when_the_direct_one_gives_up(monkeypatch):
    """A direct solve with closures whose residual does not pass the gate and that conjugate gradients cannot repair
    ends irls with IROTAVG_ERR_SOLVER (run_irls; tests/test_gpu_dist.py holds a graph that does that by itself). The
    one-shot calls -- what the drop-in header makes of irotavg::irls -- and ViewGraph::rotAvg then solve the graph
    iteratively, like the reference's single code path (ral/l1_irls.cpp:536-556). Here the give-up is forced
    (IROTAVG_BCR_FAKE_GIVE_UP) on a graph the iterative solver takes."""
    n = 6000
    S = closure_graph(n, 60000, 30, 7, 3)
    Qm = mst_init(S, n)
    ro = O.irls(S["QQ"], S["I"], Qm, 1, 4, SIG, 50, 1e-3)
    monkeypatch.setenv("IROTAVG_BCR_FAKE_GIVE_UP", "1")
    with capi.Graph(S["I"], S["QQ"], n, 1, band_direct=1) as G:
        G.set_rotations(Qm)
        with pytest.raises(capi.IrotavgError) as ei:
            G.irls(4, SIG, 50, 1e-3)
        assert ei.value.code == capi.ERR_SOLVER
        np.testing.assert_array_equal(G.get_rotations(), Qm)       # the handle's rotations are untouched
    from irotavg_amd import ral
    Q, w = Qm.copy(), np.zeros(S["m"])
    iters, _ = ral.irls(S["QQ"], S["I"], None, 4, SIG, Q, 1, 50, 1e-3, w)
    assert iters == ro["iters"]
    assert synth.angular_distance(Q, ro["Q"]).max() < 1e-7         # (the iterative solver's bar)
    np.testing.assert_allclose(w, ro["weights"], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("nclose", [7, 30, 120])
def test_round6_closure_kernels_equal_the_round5_ones(nclose):
    """Round 6 changed HOW two things in a closure solve are computed, not what: the Woodbury system by a wave per pair of
    closures (k_bcr_closure_S_pairs; IROTAVG_BCR_S_TILES=1: round 5's tile kernel) -- the same entries up to the order of
    the additions -- and K6 / K2 / the next K1 behind the gate as two launches with one publication
    (IROTAVG_NO_FUSED_CL=1: round 5's four) -- the same statements. Each variant in a process of its own (the switches are
    read once): iteration and solve counts equal, rotations within 1e-11 rad, weights to 1e-7."""
    import hashlib
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, hashlib; sys.path.insert(0, %r); import numpy as np\n"
            "from irotavg_amd import capi, synth, ral\n"
            "S = synth.add_closures(synth.make_graph(20000, 400000, 0.0, seed=5), %d, 11, max(1, %d // 30))\n"
            "Q0 = np.zeros((20000, 4)); Q0[:, 3] = 1; Q0[0] = S['Qgt'][0]\n"
            "ral.init_mst(Q0, S['QQ'], S['I'], 1)\n"
            "with capi.Graph(S['I'], S['QQ'], 20000, 1, band_direct=1) as G:\n"
            "    G.set_rotations(Q0); r = G.irls(4, 5 * np.pi / 180, 50, 1e-3)\n"
            "    Q, w, st = G.get_rotations(), G.get_weights(), G.stats()\n"
            "np.save(sys.argv[1], np.concatenate([Q.ravel(), w]))\n"
            "print(r['iters'], st['direct_solves'], st['direct_guarded'], G.direct_info()['closures'] if False else 0)\n"
            % (root, nclose, nclose))
    import tempfile
    outs = {}
    with tempfile.TemporaryDirectory() as td:
        for name, env_add in (("new", {}), ("tiles", {"IROTAVG_BCR_S_TILES": "1"}), ("unfused", {"IROTAVG_NO_FUSED_CL": "1"})):
            env = dict(os.environ)
            for k in ("IROTAVG_BCR_S_TILES", "IROTAVG_NO_FUSED_CL"):
                env.pop(k, None)
            env.update(env_add)
            f = os.path.join(td, name + ".npy")
            r = subprocess.run([sys.executable, "-c", code, f], env=env, capture_output=True, text=True, cwd=root, timeout=600)
            assert r.returncode == 0, r.stderr[-3000:]
            outs[name] = (r.stdout.strip().splitlines()[-1].split(), np.load(f))
    assert outs["new"][0] == outs["tiles"][0] == outs["unfused"][0]            # iteration and solve counts
    assert int(outs["new"][0][1]) > 0
    n4 = 4 * 20000
    for other in ("unfused", "tiles"):
        # (not bit for bit: the residuals of the next iteration come from k_weights_then_residual in one variant and from
        # k_edge_residual in the other -- the same statements, contracted into multiply-adds differently: 4e-9 relative in a
        # weight, 1e-12 in a rotation)
        Qa, Qb = outs["new"][1][:n4].reshape(-1, 4), outs[other][1][:n4].reshape(-1, 4)
        assert synth.angular_distance(Qa, Qb).max() < 1e-11, other
        np.testing.assert_allclose(outs["new"][1][n4:], outs[other][1][n4:], rtol=1e-7, err_msg=other)

"""The N > 1 path on the CPU (no GPU): (1) the host-side partition plan of every rank is
self-consistent (what A sends to B is exactly B's ghost list owned by A, without a handshake);
(2) two `gloo` processes run the sharded PCG protocol of irotavg_amd/csrc/dist.hip -- halo exchange
of the search direction of ghost views + all-reduced dot products -- with NumPy as the local
compute (the oracle's normal-matrix semantics) and reproduce the unsharded solution; (3) two and three `gloo`
processes run the protocol of the sharded DIRECT solver (bcr_dist: every rank reduces its range of a banded system to
its last block, one all-reduce gathers the separators, every rank solves the separator system and walks back)."""
import os
import socket

import numpy as np
import pytest

from irotavg_amd import capi, synth
from oracle import oracle as O


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("p_loop", [0.0, 0.05])
def test_partition_plan_consistency(world, p_loop):
    n, m, f = 3000, 30000, 3
    S = synth.make_graph(n, m, p_loop, seed=world)
    I = S["I"]
    plans = [capi.plan_host(world, r, I, n, f) for r in range(world)]
    nu = n - f
    assert plans[0]["lo"] == 0 and plans[-1]["hi"] == nu
    for a, b in zip(plans[:-1], plans[1:]):
        assert a["hi"] == b["lo"] and a["lo"] % 64 == 0          # contiguous, slice-aligned ranges
    owner = np.zeros(n, dtype=np.int64) - 1
    for r, p in enumerate(plans):
        owner[f + p["lo"]: f + p["hi"]] = r
    covered = np.zeros(m, dtype=int)
    for r, p in enumerate(plans):
        covered[p["edges"]] += 1
        e = I[p["edges"]]
        assert ((owner[e[:, 0]] == r) | (owner[e[:, 1]] == r) | ((r == 0) & (e < f).all(axis=1))).all()
        assert (owner[p["ghosts"]] != r).all() and (p["ghosts"] >= f).all()
        assert (np.diff(p["ghosts"]) > 0).all()
        # my ghosts = free endpoints of my edges that I do not own
        ends = np.unique(e[(e >= f)])
        np.testing.assert_array_equal(p["ghosts"], ends[owner[ends] != r])
    cross = (owner[I[:, 0]] >= 0) & (owner[I[:, 1]] >= 0) & (owner[I[:, 0]] != owner[I[:, 1]])
    np.testing.assert_array_equal(covered, np.where(cross, 2, 1))   # cross edges live on both shards
    for a in range(world):
        for q, b in enumerate(plans[a]["peers"]):
            sent = plans[a]["send"][b]
            want = plans[b]["ghosts"][owner[plans[b]["ghosts"]] == a]
            np.testing.assert_array_equal(sent, want)               # no handshake needed
            rb = plans[b]["peers"].index(a)
            assert plans[b]["recv_cnt"][rb] == plans[a]["send_cnt"][q]


def test_plan_rejects_too_many_ranks():
    I = np.array([[0, 1], [1, 2], [2, 3]], dtype=np.int32)
    with pytest.raises(capi.IrotavgError):
        capi.plan_host(4, 0, I, 4, 1)            # 3 free views cannot feed 4 shards of >= 64


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, m, f, seed, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    S = synth.make_graph(n, m, 0.03, seed=seed)
    I = S["I"]
    rng = np.random.default_rng(seed)
    w = rng.uniform(0.3, 2.0, size=m)               # IRLS weights d_k
    r = rng.normal(scale=0.02, size=(m, 3))          # edge residuals
    P = capi.plan_host(world, rank, I, n, f)
    lo, hi, ghosts = P["lo"], P["hi"], P["ghosts"]
    no, ng = hi - lo, len(ghosts)
    gid = {int(v): q for q, v in enumerate(ghosts)}
    e = I[P["edges"]]
    we, re_ = w[P["edges"]] ** 2, r[P["edges"]]
    own = lambda v: (v >= f + lo) & (v < f + hi)
    # local operator on owned rows: y = L_oo x_o + L_og x_g  (make_A semantics: drop edges with j fixed)
    rows_i, rows_j = e[:, 0], e[:, 1]
    keep = rows_j >= f
    diag = np.zeros(no)
    b = np.zeros((no, 3))
    for k in np.flatnonzero(keep):
        i, j = int(rows_i[k]), int(rows_j[k])
        if own(j):
            diag[j - f - lo] += we[k]; b[j - f - lo] += we[k] * re_[k]
        if i >= f and own(i):
            diag[i - f - lo] += we[k]; b[i - f - lo] -= we[k] * re_[k]

    def matvec(xo, xg):
        y = diag[:, None] * xo
        for k in np.flatnonzero(keep):
            i, j = int(rows_i[k]), int(rows_j[k])
            if i < f:
                continue
            xi = xo[i - f - lo] if own(i) else xg[gid[i]]
            xj = xo[j - f - lo] if own(j) else xg[gid[j]]
            if own(j):
                y[j - f - lo] -= we[k] * xi
            if own(i):
                y[i - f - lo] -= we[k] * xj
        return y

    def halo(xo):                                    # the protocol of halo_exchange() in dist.hip
        xg = np.zeros((ng, 3))
        reqs, bufs = [], {}
        for q, h in enumerate(P["peers"]):
            if P["send_cnt"][q]:
                t = torch.from_numpy(np.ascontiguousarray(xo[P["send"][h] - f - lo]))
                reqs.append(dist.isend(t, h))
            if P["recv_cnt"][q]:
                bufs[h] = torch.zeros(P["recv_cnt"][q], 3, dtype=torch.float64)
                reqs.append(dist.irecv(bufs[h], h))
        for rq in reqs:
            rq.wait()
        off = 0
        for q, h in enumerate(P["peers"]):               # ghosts are grouped by owner, ascending
            c = P["recv_cnt"][q]
            if c:
                xg[off:off + c] = bufs[h].numpy()
                off += c
        return xg

    def allsum(v):
        t = torch.from_numpy(np.array(v, dtype=np.float64))
        dist.all_reduce(t)
        return t.numpy()

    x = np.zeros((no, 3)); res = b.copy(); z = res / diag[:, None]; p = z.copy()
    rz = allsum((res * z).sum(0)); bb = allsum((b * b).sum(0))
    for it in range(2000):
        q = matvec(p, halo(p))
        al = rz / allsum((p * q).sum(0))
        x += al * p; res -= al * q
        if (allsum((res * res).sum(0)) <= 1e-24 * bb).all():
            break
        z = res / diag[:, None]
        rzn = allsum((res * z).sum(0))
        p = z + (rzn / rz) * p
        rz = rzn
    np.save(os.path.join(out, "x%d.npy" % rank), x)
    np.save(os.path.join(out, "range%d.npy" % rank), np.array([lo, hi]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharded_pcg_matches_unsharded(tmp_path):
    import torch.multiprocessing as mp
    n, m, f, seed, world = 600, 6000, 2, 5, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, m, f, seed, str(tmp_path)), nprocs=world, join=True)
    S = synth.make_graph(n, m, 0.03, seed=seed)
    rng = np.random.default_rng(seed)
    w = rng.uniform(0.3, 2.0, size=m)
    r = rng.normal(scale=0.02, size=(m, 3))
    rc, X = O.ls_solve(n, f, S["I"], w, r)
    assert rc == 0
    got = np.zeros_like(X)
    for rk in range(world):
        lo, hi = np.load(tmp_path / ("range%d.npy" % rk))
        got[lo:hi] = np.load(tmp_path / ("x%d.npy" % rk))
    assert np.abs(got - X).max() < 1e-9 * np.abs(X).max()


def _direct_worker(rank, world, port, n, m, f, B, seed, out):
    """One rank of the sharded DIRECT solver's protocol (irotavg_amd/csrc/dist.hip, bcr_dist): reduce the own range of
    the banded normal equations to its last block (dense Schur complements in NumPy stand in for the chunk kernels),
    ONE sum-all-reduce of a zeroed buffer = the gather of the separators, the separator system solved by every rank,
    the way back through the own range."""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    S = synth.make_graph(n, m, 0.0, seed=seed)
    rng = np.random.default_rng(seed)
    w = rng.uniform(0.3, 2.0, size=len(S["I"]))
    r = rng.normal(scale=0.02, size=(len(S["I"]), 3))
    import scipy.sparse as sp
    A = O.make_A(n, f, S["I"])
    H = (A.T @ sp.diags(w * w) @ A).toarray()         # every rank holds the graph; it USES its own rows only
    b = A.T @ ((w * w)[:, None] * r)
    nu = n - f
    chunk = ((nu + world - 1) // world + 191) // 192 * 192       # chunk_of(nu, world, 192)
    lo, hi = rank * chunk, min(nu, (rank + 1) * chunk)
    sep = np.arange(hi - B, hi) if rank < world - 1 else np.arange(hi - B, hi)   # the range's LAST block
    inner = np.arange(lo, hi - B)
    ext = np.arange(lo - B, lo) if rank > 0 else np.arange(0)                   # the previous rank's last block
    Hii = H[np.ix_(inner, inner)]
    Yi = np.linalg.solve(Hii, np.concatenate([H[np.ix_(inner, ext)], H[np.ix_(inner, sep)], b[inner]], axis=1))
    ne = len(ext)
    Ye, Ys, yb = Yi[:, :ne], Yi[:, ne:ne + B], Yi[:, ne + B:]
    sepD = H[np.ix_(sep, sep)] - H[np.ix_(sep, inner)] @ Ys
    sepR = b[sep] - H[np.ix_(sep, inner)] @ yb
    extD = -H[np.ix_(ext, inner)] @ Ye if ne else np.zeros((B, B))
    extR = -H[np.ix_(ext, inner)] @ yb if ne else np.zeros((B, 3))
    extG = (H[np.ix_(ext, sep)] - H[np.ix_(ext, inner)] @ Ys) if ne else np.zeros((B, B))
    # the gather: every rank writes its slices of a zeroed buffer, one all-reduce (what ncclAllReduce does in dist.hip)
    buf = np.zeros((world, 3 * B * B + 2 * B * 3))
    buf[rank] = np.concatenate([sepD.ravel(), extD.ravel(), extG.ravel(), sepR.ravel(), extR.ravel()])
    t = torch.from_numpy(buf)
    dist.all_reduce(t)
    buf = t.numpy()
    sl = lambda k, a, z, shape: buf[k, a:z].reshape(shape)
    oD, oXD, oXG, oR, oXR = 0, B * B, 2 * B * B, 3 * B * B, 3 * B * B + 3 * B
    # separator system: block tridiagonal, `world` blocks; D_k = sepD[k] + extD[k + 1], coupling k-1 -> k = extG[k]
    T = np.zeros((world * B, world * B)); R = np.zeros((world * B, 3))
    for k in range(world):
        Dk = sl(k, oD, oXD, (B, B)).copy(); Rk = sl(k, oR, oXR, (B, 3)).copy()
        if k + 1 < world:
            Dk += sl(k + 1, oXD, oXG, (B, B)); Rk += sl(k + 1, oXR, oXR + 3 * B, (B, 3))
        T[k * B:(k + 1) * B, k * B:(k + 1) * B] = Dk
        R[k * B:(k + 1) * B] = Rk
        if k > 0:
            G = sl(k, oXG, oR, (B, B))
            T[(k - 1) * B:k * B, k * B:(k + 1) * B] = G
            T[k * B:(k + 1) * B, (k - 1) * B:k * B] = G.T
    xs = np.linalg.solve(T, R)                          # every rank solves it (redundantly, identically)
    x_sep = xs[rank * B:(rank + 1) * B]
    x_ext = xs[(rank - 1) * B:rank * B] if rank > 0 else np.zeros((0, 3))
    x_in = yb - Ys @ x_sep - (Ye @ x_ext if ne else 0.0)
    np.save(os.path.join(out, "dx%d.npy" % rank), np.concatenate([x_in, x_sep]))
    np.save(os.path.join(out, "drange%d.npy" % rank), np.array([lo, hi]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_sharded_direct_solver_protocol_matches_unsharded(tmp_path, world):
    """the N > 1 form of the banded direct solver: local reduction, ONE collective, separator system, way back"""
    import torch.multiprocessing as mp
    n, m, f, B, seed = 1300, 13000, 2, 16, 9        # band 10 <= B; ranges of 768 / 576 views, the last one shorter
    port = _free_port()
    mp.spawn(_direct_worker, args=(world, port, n, m, f, B, seed, str(tmp_path)), nprocs=world, join=True)
    S = synth.make_graph(n, m, 0.0, seed=seed)
    rng = np.random.default_rng(seed)
    w = rng.uniform(0.3, 2.0, size=len(S["I"]))
    r = rng.normal(scale=0.02, size=(len(S["I"]), 3))
    rc, X = O.ls_solve(n, f, S["I"], w, r)
    assert rc == 0
    got = np.zeros_like(X)
    for rk in range(world):
        lo, hi = np.load(tmp_path / ("drange%d.npy" % rk))
        got[lo:hi] = np.load(tmp_path / ("dx%d.npy" % rk))
    assert np.abs(got - X).max() < 1e-9 * np.abs(X).max()


def _closure_system(n, m, f, B, nclose, seed):
    """a view sequence + closures (span > 32, blocks not neighbours), weights, right-hand side: what every rank holds"""
    S = synth.add_closures(synth.make_graph(n, m, 0.0, seed=seed), nclose, seed=seed, wrong=0)
    I = S["I"]
    rng = np.random.default_rng(seed)
    w = rng.uniform(0.3, 2.0, size=len(I))
    r = rng.normal(scale=0.02, size=(len(I), 3))
    i, j = I[:, 0].astype(np.int64), I[:, 1].astype(np.int64)
    far = (i >= f) & (j >= f) & (np.abs(i - j) > 32) & (np.abs((i - f) // B - (j - f) // B) >= 2)
    return S, I, w, r, far


def _direct_closure_worker(rank, world, port, n, m, f, B, nclose, seed, out):
    """One rank of the sharded direct solver's protocol WITH loop closures (dist.hip bcr_dist, bcr.hip "loop closures on
    a sharded sequence"): the band part reduced to the range's last block as above; a closure's incidence column is
    eliminated through the rank's own rows, what it leaves on the rank's separator and on the one before it, the rank's
    share of S = C^-1 + V' A_b^-1 V and of T = V' Y go into ONE buffer [dep | S | T] that the ranks SUM next to the
    separators' gather; the separator system carries the columns, lambda is solved by every rank, the corrections are
    local. Dense NumPy algebra stands in for the kernels; the messages are the real ones."""
    import torch
    import torch.distributed as dist
    import scipy.sparse as sp
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    S, I, w, r, far = _closure_system(n, m, f, B, nclose, seed)
    A = O.make_A(n, f, I).tocsr()
    Ab = A[np.flatnonzero(~far)]
    H = (Ab.T @ sp.diags((w * w)[~far]) @ Ab).toarray()           # the BAND operator
    b = A.T @ ((w * w)[:, None] * r)                               # the right-hand side has every edge in it
    V = A[np.flatnonzero(far)].T.toarray()                         # nu x r: +-1 at a closure's endpoints
    wc = (w * w)[far]
    nc = V.shape[1]
    nu = n - f
    chunk = ((nu + world - 1) // world + 191) // 192 * 192
    lo, hi = rank * chunk, min(nu, (rank + 1) * chunk)
    sep = np.arange(hi - B, hi)
    inner = np.arange(lo, hi - B)
    ext = np.arange(lo - B, lo) if rank > 0 else np.arange(0)
    ne = len(ext)
    Hii = H[np.ix_(inner, inner)]
    Yi = np.linalg.solve(Hii, np.concatenate([H[np.ix_(inner, ext)], H[np.ix_(inner, sep)], b[inner], V[inner]], axis=1))
    Ye, Ys, yb, Z = Yi[:, :ne], Yi[:, ne:ne + B], Yi[:, ne + B:ne + B + 3], Yi[:, ne + B + 3:]
    sepD = H[np.ix_(sep, sep)] - H[np.ix_(sep, inner)] @ Ys
    sepR = b[sep] - H[np.ix_(sep, inner)] @ yb
    extD = -H[np.ix_(ext, inner)] @ Ye if ne else np.zeros((B, B))
    extR = -H[np.ix_(ext, inner)] @ yb if ne else np.zeros((B, 3))
    extG = (H[np.ix_(ext, sep)] - H[np.ix_(ext, inner)] @ Ys) if ne else np.zeros((B, B))
    buf = np.zeros((world, 3 * B * B + 2 * B * 3))
    buf[rank] = np.concatenate([sepD.ravel(), extD.ravel(), extG.ravel(), sepR.ravel(), extR.ravel()])
    # the closures' buffer: dep[q][block of the separator system][B] | S | T  (BcrTop::xbuf)
    dep = np.zeros((nc, world, B))
    dep[:, rank, :] += (V[sep] - H[np.ix_(sep, inner)] @ Z).T      # what the columns leave on this rank's separator
    if ne:
        dep[:, rank - 1, :] += (-H[np.ix_(ext, inner)] @ Z).T      # ... and on the one before it
    Sx = V[inner].T @ Z                                            # sum over this rank's eliminated rows
    first = np.array([np.flatnonzero(V[:, q] > 0)[0] for q in range(nc)])   # I[:, 0]'s row: +1
    owner = np.minimum(first // chunk, world - 1)
    Sx[np.arange(nc), np.arange(nc)] += np.where(owner == rank, 1.0 / wc, 0.0)
    Tx = V[inner].T @ yb
    xbuf = np.concatenate([dep.ravel(), Sx.ravel(), Tx.ravel()])
    t = torch.from_numpy(np.concatenate([buf.ravel(), xbuf]))
    dist.all_reduce(t)                                             # (the hosted wire: one all-reduce of both buffers)
    tot = t.numpy()
    buf = tot[:buf.size].reshape(buf.shape)
    xbuf = tot[buf.size:]
    dep = xbuf[:dep.size].reshape(nc, world * B).T                 # world B x r: the columns' separator right-hand sides
    Sx = xbuf[dep.size:dep.size + nc * nc].reshape(nc, nc)
    Tx = xbuf[dep.size + nc * nc:].reshape(nc, 3)
    sl = lambda k, a, z, shape: buf[k, a:z].reshape(shape)
    oD, oXD, oXG, oR, oXR = 0, B * B, 2 * B * B, 3 * B * B, 3 * B * B + 3 * B
    T = np.zeros((world * B, world * B)); R = np.zeros((world * B, 3))
    for k in range(world):
        Dk = sl(k, oD, oXD, (B, B)).copy(); Rk = sl(k, oR, oXR, (B, 3)).copy()
        if k + 1 < world:
            Dk += sl(k + 1, oXD, oXG, (B, B)); Rk += sl(k + 1, oXR, oXR + 3 * B, (B, 3))
        T[k * B:(k + 1) * B, k * B:(k + 1) * B] = Dk
        R[k * B:(k + 1) * B] = Rk
        if k > 0:
            G = sl(k, oXG, oR, (B, B))
            T[(k - 1) * B:k * B, k * B:(k + 1) * B] = G
            T[k * B:(k + 1) * B, (k - 1) * B:k * B] = G.T
    sol = np.linalg.solve(T, np.concatenate([R, dep], axis=1))     # every rank, redundantly
    xs0, Zt = sol[:, :3], sol[:, 3:]
    lam = np.linalg.solve(Sx + dep.T @ Zt, Tx + dep.T @ xs0)
    xs = xs0 - Zt @ lam
    x_sep = xs[rank * B:(rank + 1) * B]
    x_ext = xs[(rank - 1) * B:rank * B] if rank > 0 else np.zeros((0, 3))
    x_in = (yb - Z @ lam) - Ys @ x_sep - (Ye @ x_ext if ne else 0.0)
    np.save(os.path.join(out, "cx%d.npy" % rank), np.concatenate([x_in, x_sep]))
    np.save(os.path.join(out, "crange%d.npy" % rank), np.array([lo, hi]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_sharded_direct_solver_protocol_with_closures_matches_unsharded(tmp_path, world):
    """closures on the sharded direct solver: ONE more summed buffer per solve ([dep | S | T]), everything else local
    or redundant -- against the oracle's solve of the FULL system (band + closures)"""
    import torch.multiprocessing as mp
    n, m, f, B, nclose, seed = 1300, 13000, 2, 16, 25, 9
    port = _free_port()
    mp.spawn(_direct_closure_worker, args=(world, port, n, m, f, B, nclose, seed, str(tmp_path)), nprocs=world, join=True)
    S, I, w, r, far = _closure_system(n, m, f, B, nclose, seed)
    assert far.sum() >= 20
    rc, X = O.ls_solve(n, f, I, w, r)
    assert rc == 0
    got = np.zeros_like(X)
    for rk in range(world):
        lo, hi = np.load(tmp_path / ("crange%d.npy" % rk))
        got[lo:hi] = np.load(tmp_path / ("cx%d.npy" % rk))
    assert np.abs(got - X).max() < 1e-9 * np.abs(X).max()

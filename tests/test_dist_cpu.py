"""The N > 1 path on the CPU (no GPU): (1) the host-side partition plan of every rank is
self-consistent (what A sends to B is exactly B's ghost list owned by A, without a handshake);
(2) two `gloo` processes run the sharded PCG protocol of irotavg_amd/csrc/dist.hip -- halo exchange
of the search direction of ghost views + all-reduced dot products -- with NumPy as the local
compute (the oracle's normal-matrix semantics) and reproduce the unsharded solution."""
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

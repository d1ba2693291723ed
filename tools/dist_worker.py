#!/usr/bin/env python3
"""One rank of the multi-PROCESS test of the sharded path (tests/test_gpu_dist_mp.py): launched by
torch.distributed.run with 2+ ranks that all use GPU 0 (a test box has one GPU). Wire:
  --wire hosted : torch.distributed over gloo behind irotavg_transport (host-staged);
  --wire rccl   : ncclCommInitRank with the ranks' unique id (RCCL may refuse ranks that share a
                  device; the caller treats that as "not available here").
Every rank builds the same seeded graph, holds its shard, runs l1ra then irls; rank 0 gathers the
rotations and compares with the single-GPU handle. Prints one line `DIST_WORKER_OK ...` on success."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--wire", default="hosted")
    ap.add_argument("--views", type=int, default=6000)
    ap.add_argument("--edges", type=int, default=60000)
    ap.add_argument("--p-loop", type=float, default=0.01)
    ap.add_argument("--closures", type=int, default=0, help="loop closures added to the sequence (a tenth of them wrong)")
    ap.add_argument("--cut-stretch", action="store_true",
                    help="Talwar + a view whose band edges are all wrong and that one correct closure holds: a dead pivot of the "
                         "band part, the shards' guarded re-solve (conjugate gradients across the processes)")
    ap.add_argument("--expect-direct", action="store_true", help="fail unless the sharded direct solver ran")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    from irotavg_amd import capi, ral, synth
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    SIG = 5 * np.pi / 180
    n, m, f = args.views, args.edges, 2
    S = synth.make_graph(n, m, args.p_loop, seed=2)
    if args.closures:
        S = synth.add_closures(S, args.closures, seed=7, wrong=args.closures // 10)
    cost = 4
    if args.cut_stretch:
        cost = 12                                                  # Talwar: weights of exactly 0
        I, QQ = S["I"].copy(), S["QQ"].copy()
        v0 = n // 2 + 37
        rng = np.random.default_rng(9)
        touching = ((I[:, 0] == v0) | (I[:, 1] == v0)) & (np.abs(I[:, 0] - I[:, 1]) <= 32)
        R = rng.normal(size=(int(touching.sum()), 4))
        QQ[touching] = R / np.linalg.norm(R, axis=1, keepdims=True)
        far = np.array([[200, v0]], dtype=np.int32)
        QQf = synth.qmul(S["Qgt"][v0:v0 + 1], synth.qconj(S["Qgt"][200:201]))
        I = np.concatenate([I, far]).astype(np.int32)
        QQ = np.concatenate([QQ, QQf])
        order = np.lexsort((np.arange(len(I)), I[:, 1]))
        S = dict(S, I=I[order], QQ=QQ[order])
    Q0 = np.zeros((n, 4)); Q0[:, 3] = 1; Q0[:f] = S["Qgt"][:f]
    ral.init_mst(Q0, S["QQ"], S["I"], f)
    if args.wire == "hosted":
        D = capi.DistGraph(S["I"], S["QQ"], n, f, world, rank=rank, transport=capi.torch_transport(), device=0)
    else:
        uid = [capi.DistGraph.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        D = capi.DistGraph(S["I"], S["QQ"], n, f, world, rank=rank, unique_id=uid[0], device=0)
    D.set_rotations(Q0)
    b1 = D.l1ra(3, 1e-3)
    b2 = D.irls(cost, SIG, 20, 1e-3)
    st = D.stats()
    if args.expect_direct and (st["direct_solves"] == 0 or st["pcg_iters"] != 0 or D.info()["direct_block"] == 0):
        print("DIST_WORKER_FAIL rank %d: the sharded direct solver did not run: %r" % (rank, st), flush=True)
        sys.exit(1)
    Qmine = D.get_rotations(into=np.zeros_like(Q0))        # rows this rank owns, zeros elsewhere
    wmine = np.nan_to_num(D.get_weights(), nan=-1.0)
    D.close()
    Qt = torch.from_numpy(np.ascontiguousarray(Qmine))
    dist.all_reduce(Qt)                                     # owned rows are disjoint: the sum assembles them
    wt = torch.from_numpy(wmine)
    dist.all_reduce(wt, op=dist.ReduceOp.MAX)               # an edge lives on 1 or 2 shards, same value
    ok = True
    if rank == 0:
        Qd = Qt.numpy().copy()
        Qd[:f] = Q0[:f]
        with capi.Graph(S["I"], S["QQ"], n, f, device=0) as G:
            G.set_rotations(Q0)
            a1 = G.l1ra(3, 1e-3)
            a2 = G.irls(cost, SIG, 20, 1e-3)
            ga = G.stats()
            Qa, wa = G.get_rotations(), G.get_weights()
        err = synth.angular_distance(Qa, Qd).max()
        werr = np.abs(wa - wt.numpy()).max() / np.abs(wa).max()
        ok = (a1["iters"], a2["iters"]) == (b1["iters"], b2["iters"]) and err < 1e-8 and werr < 1e-6
        if args.cut_stretch:   # both handles must have met the dead pivot and repaired the solve
            ok = ok and ga["direct_guarded"] >= 1 and st["direct_guarded"] >= 1
        print("DIST_WORKER_%s wire=%s world=%d l1ra_iters=%d irls_iters=%d max_angle=%.2e weights_rel=%.2e"
              % ("OK" if ok else "FAIL", args.wire, world, b1["iters"], b2["iters"], err, werr), flush=True)
    flag = torch.tensor([1 if ok else 0])
    dist.broadcast(flag, src=0)
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()

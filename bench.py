#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config.

metric : IRLS edge-updates/s (= m * IRLS iterations / wall time of the `irls` call, inputs
         resident in HBM) + iterations-to-converge, on the synthetic 100k-view / 2M-edge SO(3)
         view-graph (SURVEY.md 8(d) generator, seed 0), Geman-McClure sigma = 5 deg,
         change_th = 1e-3, max_iters = 100 -- the reference's defaults (src/ViewGraph.cpp:1402-1414).
step   : one complete `irls` solve of that graph from its `init_mst` initialisation (rotations
         restored on the device before every step; the restore is inside the timed region).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--p-loop P] [--views n] [--edges m]

For N > 1 the driver launches one process per GPU through torch.distributed.run; rank 0 prints
ONE JSON line. The line also carries `roofline` (dominant kernel + the edge-residual kernel the
north star names, HIP-event timed on the handle's stream) and `cpu_baseline` (the CPU oracle --
a port, not Eigen+SuiteSparse, which cannot be built in this image -- on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SIG = 5 * np.pi / 180
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def build_problem(n, m, p_loop, seed):
    from irotavg_amd import ral, synth
    S = synth.make_graph(n, m, p_loop, seed=seed)
    Q0 = np.zeros((n, 4))
    Q0[:, 3] = 1
    Q0[0] = S["Qgt"][0]
    ral.init_mst(Q0, S["QQ"], S["I"], 1)   # host, outside the timed region (as in the reference)
    return S, Q0


def kernel_rooflines(G, S, st, sharded=False):
    """HIP-event timed kernels -> algorithmic GB/s (formulas: SURVEY.md 8(d))."""
    from irotavg_amd import capi
    m, n_t = S["m"], S["n"]
    nu = n_t - 1
    nnz0 = st["level_nnz"][0]
    alg = {
        "edge_residual": m * (8 + 32 + 24) + 32 * n_t,
        "update_weights": m * (8 + 24 + 8) + 24 * nu,
        "assemble": m * (8 + 8 + 24) + nu * (8 + 24) + 8 * nnz0,
        "spmv": nnz0 * (8 + 4) + 4 * (nu + 1) + 2 * 24 * nu,
    }
    which = {"edge_residual": 1, "update_weights": 2, "assemble": 3, "spmv": 4}
    out = {}
    di = G.direct_info() if not sharded else dict(block=0, levels=[])
    if di["block"]:
        # the handle's systems are solved by the banded direct solver (bcr.hip): its launches instead of the PCG's
        del which["spmv"]
        alg["assemble"] = m * (8 + 8 + 24) + nu * (8 + 24) + 8 * nnz0   # level 0 only: same formula, no coarse level
    for name, w in which.items():
        ms = G.time_kernel(w, 50)
        out[name] = dict(ms=ms, bytes=alg[name], gbs=alg[name] / (ms * 1e-3) / 1e9)
    if di["block"]:
        B = di["block"]
        NC = 2 * B + 3
        lev = di["levels"]
        tot_ms, tot_by, tot_fl = 0.0, 0, 0
        for l, L in enumerate(lev):
            top = l == len(lev) - 1
            elim = sum(min(7, max(0, L["blocks"] - 8 * c)) for c in range(L["chunks"]))   # blocks eliminated
            wbytes = 8 * elim * B * NC
            sep = 0 if top else 8 * L["chunks"] * (3 * B * B + 6 * B)
            if l == 0:
                rd = nnz0 * (8 + 4) + 4 * (nu // 64 + 2) + nu * (8 + 32)    # SELL entries, slice offsets, diagonal, rhs
            else:
                P = lev[l - 1]
                rd = 8 * P["chunks"] * (3 * B * B + 6 * B)
                if L["reduced"] < L["blocks"]:   # mixed level 1: the level-0 rows no chunk reduced
                    raw = (L["blocks"] - L["reduced"]) * B
                    rd += int(nnz0 * (8 + 4) * raw / max(nu, 1)) + raw * (8 + 32)
            # flops issued on useful entries: D^-1 (2 B^3), W = D^-1 [P' Q R], P W, Q' W_QR
            fl = elim * (2 * B ** 3 + 2 * B * B * NC * 2 + 2 * B * B * (B + 3))
            ms = G.time_kernel(20 + l, 50)
            out["bcr_reduce_l%d" % l] = dict(ms=ms, bytes=rd + wbytes + sep, gbs=(rd + wbytes + sep) / (ms * 1e-3) / 1e9,
                                             flops=fl, tflops=fl / (ms * 1e-3) / 1e12, workgroups=L["chunks"])
            msb = G.time_kernel(40 + l, 50)
            bb = wbytes + 8 * (L["chunks"] * 8 * B * 3) + (32 * nu if l == 0 else 0)
            out["bcr_back_l%d" % l] = dict(ms=msb, bytes=bb, gbs=bb / (msb * 1e-3) / 1e9, workgroups=L["chunks"])
            tot_ms += ms + msb
            tot_by += rd + wbytes + sep + bb
            tot_fl += fl
        ms = G.time_kernel(19, 50)
        out["bcr_solve"] = dict(ms=ms, bytes=tot_by, gbs=tot_by / (ms * 1e-3) / 1e9, flops=tot_fl,
                                tflops=tot_fl / (ms * 1e-3) / 1e12, launches=2 * len(lev), sum_of_launches_ms=tot_ms,
                                block=B, levels=lev)
        return out
    try:  # band-only graphs on one GPU run the p-update fused into the SpMV (k_pspmv_dot)
        if sharded:   # the sharded PCG exchanges p between its p-update and its SpMV: unfused kernels
            raise capi.IrotavgError(capi.ERR_BAD_ARG, "sharded")
        ms = G.time_kernel(8, 50)
        by = nnz0 * (8 + 4) + 4 * (nu + 1) + nu * (24 + 24 + 8 + 3 + 24 + 24)
        out["pspmv"] = dict(ms=ms, bytes=by, gbs=by / (ms * 1e-3) / 1e9)
    except capi.IrotavgError:
        pass
    try:  # the two-launch Chronopoulos-Gear iteration (cgcg.hip): band-only graphs with >= 3 levels on one GPU
        if sharded or st["levels"] < 3:
            raise capi.IrotavgError(capi.ERR_BAD_ARG, "not this graph's PCG")
        n1, nd = st["level_rows"][1], st["level_rows"][2]
        nnz1 = st["level_nnz"][1]
        coarse = 2 * 24 * n1 + 12 * nnz1 + 24 * nd + (8 * nd * nd if st["levels"] == 3 else 0)
        ms = G.time_kernel(9, 50)
        # matrix once, r / idg / diag in, u / w out, plus the coarse data every tile slice comes from
        by = nnz0 * (8 + 4) + 4 * (nu + 1) + nu * (24 + 8 + 8 + 24 + 24) + coarse
        out["cg_apply"] = dict(ms=ms, bytes=by, gbs=by / (ms * 1e-3) / 1e9)
        ms = G.time_kernel(10, 50)
        by = nu * 24 * 10 + 2 * 24 * n1 + 12 * nnz1 + 2 * 24 * nd   # r w s u p x in, p s x r out, b1 x1 b2 x2 out
        out["cg_update"] = dict(ms=ms, bytes=by, gbs=by / (ms * 1e-3) / 1e9)
    except capi.IrotavgError:
        pass
    out["precondition"] = dict(ms=G.time_kernel(5, 50))
    out["dense_inversion"] = dict(ms=G.time_kernel(7, 5))
    return out


PMC_SUMMARY = "r03_pmc_summary.json"   # (the PCG-era profiles of this round: profiles/r03_pcg_*)
KERNEL_STATS = "r03_bench_kernel_stats.csv"


def pmc_traffic(kernel, workload):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of THIS workload
    (tools/profile_counters.sh + tools/summarize_pmc.py -> profiles/r02_pmc_summary.json). The counters
    need their own rocprofv3 passes, so they cannot be collected inside a plain bench run; the summary
    carries the workload string of the bench line it was taken on, and a run on any other workload
    reports null instead of a number that belongs to another size or topology."""
    path = os.path.join(ROOT, "profiles", PMC_SUMMARY)
    try:
        with open(path) as fh:
            table = json.load(fh)
        if table.get("_meta", {}).get("workload") != workload:
            return None
        for name, row in table.items():   # template kernels appear as "name<args>"
            if name == kernel or name.startswith(kernel + "<") or _is_l0_reduce(kernel, name):
                return float(row["traffic_bytes"])
        return None
    except Exception:
        return None


def _is_l0_reduce(kernel, name):
    """k_bcr_reduce is one template for every level; the level-0 instantiation is k_bcr_reduce<B, NR, true, ...>"""
    n = name.replace(" ", "")
    return kernel == "k_bcr_reduce_l0" and n.split("<")[0].endswith("k_bcr_reduce") and "<" in n and \
        n.split("<")[1].split(",")[2:3] == ["true"]


def in_situ_ms(kernel, workload):
    """Average duration (ms) of `kernel` inside real solves, from the committed `rocprofv3 --kernel-trace --stats`
    summary of this bench command (profiles/r03_bench_kernel_stats.csv, stamped with the workload string by
    tools/summarize_pmc.py in the PMC summary next to it). The HIP-event figure of `roofline.ms_per_launch` is a
    back-to-back loop of one kernel; inside a solve the same kernel runs 2-7 % slower (dependent launches, cold
    L2 after other kernels). None when no profile of THIS workload is committed."""
    import csv
    try:
        with open(os.path.join(ROOT, "profiles", PMC_SUMMARY)) as fh:
            if json.load(fh).get("_meta", {}).get("workload") != workload:
                return None
        with open(os.path.join(ROOT, "profiles", KERNEL_STATS)) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Name", "")
                if name.split("(")[0].split("<")[0].split("::")[-1].strip() == kernel or \
                        _is_l0_reduce(kernel, name.split("(")[0]):
                    return float(row["AverageNs"]) * 1e-6
    except Exception:
        return None
    return None


def suitesparse_baseline(S, Q0, budget_s=20.0):
    """if the GPU box has SuiteSparse (it is not in this image), time the reference's
    own library call -- SuiteSparseQR X = A \\ B per IRLS iteration (ral/l1_irls.cpp:536-556) -- through
    oracle/spqr_harness.cpp on the same graph. Returns a dict, or a reason string."""
    import subprocess
    import tempfile
    src = os.path.join(ROOT, "oracle", "spqr_harness.cpp")
    exe = os.path.join(tempfile.gettempdir(), "irotavg_spqr_harness")
    probe = subprocess.run(["g++", "-O2", "-std=c++11", src, "-o", exe, "-I/usr/include/suitesparse", "-lspqr",
                            "-lcholmod", "-lsuitesparseconfig"], capture_output=True, text=True)
    if probe.returncode != 0:
        return "SuiteSparse (SuiteSparseQR.hpp / libspqr) not found on this box: " + \
               (probe.stderr.strip().splitlines() or ["compile failed"])[0][:160]
    from oracle import oracle as O
    w3 = O.log_map(O.delta_rel(S["I"], S["QQ"], Q0))[:, :3]
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as fh:
        fh.write("%d %d %d\n" % (S["m"], S["n"], 1))
        for (i, j), r in zip(S["I"], w3):
            fh.write("%d %d %.17g %.17g %.17g\n" % (i, j, r[0], r[1], r[2]))
        path = fh.name
    try:
        out = subprocess.run([exe, path, str(budget_s)], capture_output=True, text=True, timeout=10 * budget_s + 60)
        sec, reps = [float(x) for x in out.stdout.split()[:2]]
        return dict(value=S["m"] * reps / sec, unit="edge-updates/s", cores=1, kind="reference-library",
                    sample="%d SuiteSparseQR least-squares solves (one per IRLS iteration, unit weights) of the same "
                           "%d-view/%d-edge graph, %.1f s" % (int(reps), S["n"], S["m"], sec))
    except Exception as e:
        return "SuiteSparse harness failed: %s" % e
    finally:
        os.unlink(path)


def cpu_baseline(S, Q0, p_loop, budget_s=25.0):
    """The CPU oracle (oracle/, single thread, own sparse Cholesky) on a bounded sample."""
    from oracle import oracle as O
    from irotavg_amd import synth
    cores = 1
    if p_loop == 0.0 or S["m"] <= 200000:
        # the identical workload: complete IRLS solves (to convergence) of the same graph, repeated
        # until ~10 s of CPU time have been spent
        t = time.time()
        updates, runs, iters = 0, 0, 0
        while time.time() - t < 10.0 and runs < 50:
            r = O.irls(S["QQ"], S["I"], Q0, 1, 4, SIG, 100, 1e-3)
            updates += S["m"] * r["iters"]
            iters = r["iters"]
            runs += 1
        dt = time.time() - t
        return dict(value=updates / dt, unit="edge-updates/s", cores=cores, kind="port",
                    sample="%d complete IRLS solves (%d iterations each, incl. symbolic analysis) of the "
                           "same %d-view/%d-edge graph, %.1f s" % (runs, iters, S["n"], S["m"], dt),
                    iters_to_converge=iters,
                    note="oracle = C restatement + own sparse Cholesky; NOT Eigen+SuiteSparse "
                         "(unbuildable here)")
    # loop-closure-rich graphs: fill of the direct factorisation explodes; sample a 10x smaller graph
    n2, m2 = S["n"] // 10, S["m"] // 10
    S2 = synth.make_graph(n2, m2, p_loop, seed=0)
    Q2 = np.zeros((n2, 4)); Q2[:, 3] = 1; Q2[0] = S2["Qgt"][0]
    rc, Q2 = O.init_mst(Q2, S2["QQ"], S2["I"], 1)
    t = time.time()
    r = O.irls(S2["QQ"], S2["I"], Q2, 1, 4, SIG, 1, 1e-3)
    dt = time.time() - t
    return dict(value=S2["m"] * r["iters"] / dt, unit="edge-updates/s", cores=cores, kind="port",
                sample="1 IRLS iteration of a %d-view/%d-edge graph with the same p_loop "
                       "(the full-size factorisation does not finish in minutes), %.1f s" % (n2, m2, dt),
                note="oracle = C restatement + own sparse Cholesky; NOT Eigen+SuiteSparse")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--ramp", type=int, default=40, help="untimed ramp-up steps ahead of --warmup")
    ap.add_argument("--views", type=int, default=100000)
    ap.add_argument("--edges", type=int, default=2000000)
    ap.add_argument("--p-loop", type=float, default=0.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--rtol", type=float, default=1e-10)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--classic", action="store_true",
                    help="A/B: the classic PCG recurrences (separate launches) instead of the two-launch iteration")
    ap.add_argument("--force-dist", action="store_true",
                    help="use the sharded (RCCL) path even with one rank (exercises it on a 1-GPU box)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        # IROTAVG_BENCH_SHARE_GPU=1 (tests on a one-GPU box): every rank on device 0, gloo for the control
        # messages; RCCL refuses two ranks on one device, so the shards then talk over the hosted transport
        share = os.environ.get("IROTAVG_BENCH_SHARE_GPU") == "1"
        if share:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from irotavg_amd import capi

    S, Q0 = build_problem(args.views, args.edges, args.p_loop, args.seed)
    dev = local_rank if dist is not None else -1

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # N = 1: the resident single-GPU handle. N > 1: the SAME graph sharded by contiguous view
    # ranges, one shard per process/GPU, RCCL over xGMI (halo exchange of the PCG direction +
    # all-reduced dot products; see DESIGN.md "Multi-GPU") -- strong scaling.
    G = D = None
    wire = None
    if dist is None:
        G = capi.Graph(S["I"], S["QQ"], S["n"], 1, pcg_rtol=args.rtol, device=dev,
                       pcg_classic=1 if args.classic else 0)
        G.set_rotations(Q0)
        G.snapshot_rotations()

        def step():
            G.restore_rotations()
            r = G.irls(4, SIG, 100, 1e-3)
            G.synchronize()
            return r
    else:
        wire = "RCCL"
        try:
            uid = [capi.DistGraph.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            D = capi.DistGraph(S["I"], S["QQ"], S["n"], 1, world, rank=rank, unique_id=uid[0],
                               pcg_rtol=args.rtol, device=dev)
        except Exception as e:  # e.g. the library's communicator cannot be formed on this node
            print("[bench] rank %d: RCCL shard handle failed (%s)" % (rank, e), file=sys.stderr, flush=True)
            D = None
        ok = torch.tensor([1 if D is not None else 0], dtype=torch.int32, device="cpu" if share else "cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            # every rank falls back together: the same sharded solver over the hosted transport
            # (irotavg_dist_create_hosted: halo + all-reduce staged through host buffers and moved by
            # torch.distributed/gloo) -- slower per exchange, same arithmetic
            if D is not None:
                D.close()
            hosted = dist.new_group(backend="gloo")
            D = capi.DistGraph(S["I"], S["QQ"], S["n"], 1, world, rank=rank,
                               transport=capi.torch_transport(hosted), pcg_rtol=args.rtol, device=dev)
            wire = "host-staged torch.distributed/gloo (RCCL communicator unavailable)"

        D.set_rotations(Q0)
        D.snapshot_rotations()

        def step():
            D.restore_rotations()        # device copy, as the single-GPU step
            return D.irls(4, SIG, 100, 1e-3)

    res = None
    # a fresh box starts at idle clocks and with cold caches / allocator pools: the first solves of a process
    # measured ~8 % slower than the steady state (6.24 vs 5.75 ms). Ramp-up steps ahead of the W warm-up
    # steps the contract names -- untimed, like them
    for _ in range(args.ramp):
        res = step()
    for _ in range(args.warmup):
        res = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if share else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    dstats = D.stats() if D is not None else None
    if D is not None:
        dinfo = D.info()   # wire actually used + ranks of the RCCL communicator as RCCL counts them
        D.close()
        if rank == 0:   # kernel rooflines are measured on a single-GPU handle of the same graph
            G = capi.Graph(S["I"], S["QQ"], S["n"], 1, pcg_rtol=args.rtol, device=dev)
            G.set_rotations(Q0)
            G.irls(4, SIG, 100, 1e-3)

    if rank == 0:
        iters = res["iters"]
        st = G.stats()
        if dstats is not None:
            st = dict(st, pcg_iters=dstats["pcg_iters"], pcg_solves=dstats["pcg_solves"])
        value = S["m"] * iters * args.steps / dt
        line = {
            "metric": "IRLS edge-updates/sec (+ iters-to-converge)",
            "value": value, "unit": "edge-updates/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ramp": args.ramp, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "iters_to_converge": iters,
            "config": {"workload": "synthetic SO(3) view-graph, %d views / %d edges, p_loop=%g, "
                                   "sigma_n=0.01 rad, 5%% outliers among loop edges, seed %d; "
                                   "f=1, init_mst start, Geman-McClure sigma=5deg, change_th=1e-3, "
                                   "max_iters=100" % (S["n"], S["m"], args.p_loop, args.seed),
                       "linear_solver": ("banded direct solver (block cyclic reduction, blocks of %d; band %d)" % (
                                             st["band_block"], st["band"]) if st.get("direct_solves", 0) > 0 and dstats is None
                                         else ("sharded banded direct solver (blocks of %d: local reductions, ONE gather of the "
                                               "%d separators per linear solve, separator system on every rank)" % (
                                                   dinfo["direct_block"], world)
                                               if dstats is not None and dinfo.get("direct_block") else
                                               "multigrid-preconditioned CG")),
                       "direct_solves": st.get("direct_solves", 0) if dstats is None else dstats.get("direct_solves", 0),
                       "pcg_rtol": args.rtol, "pcg_iters_per_solve": st["pcg_iters"] / max(st["pcg_solves"], 1),
                       "mg_level_rows": st["level_rows"] if dstats is None else dstats["level_rows"],
                       "parallelism": "1 GPU" if world == 1 else
                       "views sharded in %d contiguous ranges, 1 shard/GPU, %s %s" % (
                           world, wire, "gather of the separators (1 per linear solve) + halo of X and score all-reduce "
                           "(1 per IRLS iteration)" if dstats is not None and dinfo.get("direct_block") else "halo + all-reduce")},
            "final_scores": [float(x) for x in res["scores"]],
            "timing_note": "steady state: %d untimed ramp-up solves precede the --warmup solves (a fresh box starts at "
                           "idle clocks; the first solves of a process run ~8 %% slower); every step repeats the "
                           "identical solve (on the PCG path the predictive poll schedule and the cached coarse inverse "
                           "are then warm; the direct solver keeps nothing between solves) -- not a first-call figure "
                           "(that is also_one_shot_host_buffers)" % args.ramp,
        }
        if dstats is not None:
            line["config"]["dist"] = dinfo
        kr = kernel_rooflines(G, S, st, sharded=dstats is not None)
        dom = "bcr_reduce_l0" if "bcr_reduce_l0" in kr else \
            ("cg_apply" if "cg_apply" in kr else ("pspmv" if "pspmv" in kr else "spmv"))
        dname = {"bcr_reduce_l0": "k_bcr_reduce, level 0 (banded direct solver: blocks gathered from the SELL-64 operator, "
                                  "7 of 8 blocks per chunk eliminated on the matrix cores, W written for the way back; the "
                                  "longest launch of a solve)",
                 "spmv": "k_spmv_dot (level-0 SELL-64 SpMV + fused dot, dominant PCG kernel)",
                 "pspmv": "k_pspmv_dot (PCG p-update fused into the level-0 SELL-64 SpMV + dot, dominant PCG kernel)",
                 "cg_apply": "k_cg_apply (u = M^-1 r incl. the tile's slice of the dense coarse solve and the level-1 "
                             "up-sweep, then the level-0 SELL-64 SpMV w = L u + dots; dominant PCG kernel)"}[dom]
        line["roofline"] = {"kernel": dname,
                            "bound": "hbm", "achieved": kr[dom]["gbs"], "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": kr[dom]["gbs"] / HBM_PEAK_GBS,
                            "traffic": pmc_traffic({"pspmv": "k_pspmv_dot", "spmv": "k_spmv_dot",
                                                    "cg_apply": "k_cg_apply", "bcr_reduce_l0": "k_bcr_reduce_l0"}[dom],
                                                   line["config"]["workload"]),
                            "ms_per_launch": kr[dom]["ms"], "algorithmic_bytes": kr[dom]["bytes"]}
        kname = {"pspmv": "k_pspmv_dot", "spmv": "k_spmv_dot", "cg_apply": "k_cg_apply",
                 "bcr_reduce_l0": "k_bcr_reduce_l0"}[dom]
        insitu = in_situ_ms(kname, line["config"]["workload"])
        line["roofline"]["ms_per_launch_in_situ"] = insitu
        line["roofline"]["frac_in_situ"] = (kr[dom]["bytes"] / (insitu * 1e-3) / 1e9 / HBM_PEAK_GBS) if insitu else None
        if dom == "bcr_reduce_l0":
            FP64_PEAK_TF = 78.6   # AMD's MI355X data sheet (fp64 vector = matrix); the guide lists no fp64 figure
            line["roofline"]["note"] = (
                "a direct solve: its ~14 dependent block eliminations (Gauss-Jordan sweep of a %d x %d block + five "
                "products on the matrix cores each) set the time, not the bytes -- see roofline_direct_solve for both "
                "ceilings of the whole solve" % (kr["bcr_solve"]["block"], kr["bcr_solve"]["block"]))
            line["roofline"]["mfma"] = {"bound": "mfma", "achieved": kr[dom]["tflops"], "peak": FP64_PEAK_TF,
                                        "unit": "TFLOP/s", "frac": kr[dom]["tflops"] / FP64_PEAK_TF,
                                        "flops_per_launch": kr[dom]["flops"]}
            bs = kr["bcr_solve"]
            line["roofline_direct_solve"] = {
                "kernel": "all %d launches of one direct solve (k_bcr_reduce x %d, k_bcr_back x %d)" % (
                    bs["launches"], bs["launches"] // 2, bs["launches"] // 2),
                "bound": "hbm", "algorithmic_bytes": bs["bytes"], "ms_per_solve": bs["ms"],
                "achieved": bs["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bs["gbs"] / HBM_PEAK_GBS,
                "mfma": {"achieved": bs["tflops"], "peak": FP64_PEAK_TF, "unit": "TFLOP/s",
                         "frac": bs["tflops"] / FP64_PEAK_TF, "flops": bs["flops"]},
                "block": bs["block"], "levels": bs["levels"]}
        if "cg_apply" in kr and "cg_update" in kr:
            # the whole PCG iteration (both launches) against SURVEY.md 8(d)'s own K4 + K5 bytes: the dense inverse
            # every tile slice re-reads and the coarse vectors are this design's cost, not algorithmic traffic
            nu_ = S["n"] - 1
            nnz0_ = st["level_nnz"][0]
            k45 = nnz0_ * 12 + 4 * (nu_ + 1) + 2 * 24 * nu_ + 10 * 24 * nu_
            ms_it = kr["cg_apply"]["ms"] + kr["cg_update"]["ms"]
            ia, iu = in_situ_ms("k_cg_apply", line["config"]["workload"]), in_situ_ms("k_cg_update", line["config"]["workload"])
            line["roofline_pcg_iteration"] = {
                "kernel": "k_cg_apply + k_cg_update (one PCG iteration)", "bound": "hbm",
                "algorithmic_bytes": k45, "formula": "K4 + K5 of SURVEY.md 8(d): nnz0*(8+4) + 4(n+1) + 2*24n + 10*24n",
                "ms_per_iteration": ms_it, "achieved": k45 / (ms_it * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": k45 / (ms_it * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "ms_per_iteration_in_situ": (ia + iu) if ia and iu else None,
                "frac_in_situ": (k45 / ((ia + iu) * 1e-3) / 1e9 / HBM_PEAK_GBS) if ia and iu else None,
                "own_bytes_both_kernels": kr["cg_apply"]["bytes"] + kr["cg_update"]["bytes"]}
        line["roofline_edge_residual"] = {
            "kernel": "k_edge_residual (K1, the kernel north_star names)", "bound": "hbm",
            "achieved": kr["edge_residual"]["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": kr["edge_residual"]["gbs"] / HBM_PEAK_GBS,
            "traffic": pmc_traffic("k_edge_residual", line["config"]["workload"]),
            "ms_per_launch": kr["edge_residual"]["ms"], "algorithmic_bytes": kr["edge_residual"]["bytes"]}
        ta = [pmc_traffic(k, line["config"]["workload"]) for k in ("k_assemble0w", "k_coarse_level")]
        line["roofline_assembly"] = {
            "kernel": "K3: k_assemble0w (level 0 from the LDS-staged run of the edge list; on the PCG path also level 1 "
                      "and k_coarse_level for level 2)",
            "bound": "hbm", "achieved": kr["assemble"]["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": kr["assemble"]["gbs"] / HBM_PEAK_GBS,
            "traffic": (ta[0] + (ta[1] or 0.0)) if ta[0] is not None else None,
            "ms_per_launch": kr["assemble"]["ms"], "algorithmic_bytes": kr["assemble"]["bytes"]}
        line["kernels"] = {k: {kk: (vv if isinstance(vv, (int, list, dict, str)) else float(vv)) for kk, vv in v.items()}
                           for k, v in kr.items()}
        if not args.no_extra and world == 1 and args.p_loop == 0.0 and args.views == 100000:
            # the other topology SURVEY.md 8(d) asks for: 2 % random loop-closure edges
            S2, Q2 = build_problem(args.views, args.edges, 0.02, args.seed)
            with capi.Graph(S2["I"], S2["QQ"], S2["n"], 1, pcg_rtol=args.rtol) as G2:
                G2.set_rotations(Q2)
                G2.snapshot_rotations()
                G2.irls(4, SIG, 100, 1e-3)
                t1 = time.perf_counter()
                reps = 5
                for _ in range(reps):
                    G2.restore_rotations()
                    r2 = G2.irls(4, SIG, 100, 1e-3)
                G2.synchronize()
                d2 = time.perf_counter() - t1
                s2 = G2.stats()
            line["also_p_loop_0.02"] = {"value": S2["m"] * r2["iters"] * reps / d2, "unit": "edge-updates/s",
                                        "iters_to_converge": r2["iters"], "ms_per_step": 1e3 * d2 / reps,
                                        "pcg_iters_per_solve": s2["pcg_iters"] / max(s2["pcg_solves"], 1),
                                        "dense_inversions_per_solve_call": s2["dense_inversions"] / (reps + 1),
                                        "dense_repairs_per_solve_call": s2["dense_repairs"] / (reps + 1)}
        if not args.no_extra and world == 1 and args.p_loop == 0.0 and args.views == 100000:
            # the headline topology with a NON-uniform re-weighting: 2 % of the band edges carry a 0.3 rad
            # error (the workload of test_every_cost_on_the_two_launch_path_matches_oracle at full size); the
            # headline graph itself has no outliers at all (SURVEY's generator puts them among loop edges)
            from irotavg_amd import ral, synth
            S4 = synth.make_graph(args.views, args.edges, 0.0, seed=args.seed, p_band_out=0.02)
            Q4 = np.zeros((args.views, 4)); Q4[:, 3] = 1; Q4[0] = S4["Qgt"][0]
            ral.init_mst(Q4, S4["QQ"], S4["I"], 1)
            with capi.Graph(S4["I"], S4["QQ"], S4["n"], 1, pcg_rtol=args.rtol) as G4:
                G4.set_rotations(Q4)
                G4.snapshot_rotations()
                for _ in range(3):
                    G4.restore_rotations()
                    G4.irls(4, SIG, 100, 1e-3)
                G4.synchronize()
                G4.reset_stats()
                t1 = time.perf_counter()
                reps = 5
                for _ in range(reps):
                    G4.restore_rotations()
                    r4 = G4.irls(4, SIG, 100, 1e-3)
                G4.synchronize()
                d4 = time.perf_counter() - t1
                s4 = G4.stats()
            line["also_band_outliers"] = {"value": S4["m"] * r4["iters"] * reps / d4, "unit": "edge-updates/s",
                                          "iters_to_converge": r4["iters"], "ms_per_step": 1e3 * d4 / reps,
                                          "direct_solves_per_solve_call": s4.get("direct_solves", 0) / reps,
                                          "pcg_iters_per_solve": s4["pcg_iters"] / max(s4["pcg_solves"], 1),
                                          "dense_inversions_per_solve_call": s4["dense_inversions"] / reps,
                                          "note": "p_loop=0, 2 % of ALL edges off by N(0, 0.3^2) rad, init_mst start"}
        if not args.no_extra and world == 1 and st.get("direct_solves", 0) > 0:
            # the same workload through the handle's OTHER solver: the multigrid-PCG (what every graph with loop
            # closures and every shard runs; the headline of rounds 1 and 2)
            with capi.Graph(S["I"], S["QQ"], S["n"], 1, pcg_rtol=args.rtol, band_direct=-1) as G5:
                G5.set_rotations(Q0)
                G5.snapshot_rotations()
                for _ in range(10):
                    G5.restore_rotations()
                    G5.irls(4, SIG, 100, 1e-3)
                G5.synchronize()
                G5.reset_stats()
                t1 = time.perf_counter()
                reps = 5
                for _ in range(reps):
                    G5.restore_rotations()
                    r5 = G5.irls(4, SIG, 100, 1e-3)
                G5.synchronize()
                d5 = time.perf_counter() - t1
                s5 = G5.stats()
            line["also_pcg_path"] = {"value": S["m"] * r5["iters"] * reps / d5, "unit": "edge-updates/s",
                                     "iters_to_converge": r5["iters"], "ms_per_step": 1e3 * d5 / reps,
                                     "pcg_iters_per_solve": s5["pcg_iters"] / max(s5["pcg_solves"], 1),
                                     "note": "band_direct = -1: two-launch multigrid-PCG (cgcg.hip), pcg_rtol %g" % args.rtol}
        if not args.no_extra and world == 1 and args.rtol == 1e-10 and st.get("direct_solves", 0) == 0:
            # the same workload with the inner tolerance at the accuracy a direct fp64 factorisation of
            # these normal equations reaches itself (kappa*eps ~ 1e-9): fewer PCG iterations, same result
            with capi.Graph(S["I"], S["QQ"], S["n"], 1, pcg_rtol=1e-8) as G3:
                G3.set_rotations(Q0)
                G3.snapshot_rotations()
                G3.irls(4, SIG, 100, 1e-3)
                t1 = time.perf_counter()
                reps = 5
                for _ in range(reps):
                    G3.restore_rotations()
                    r3 = G3.irls(4, SIG, 100, 1e-3)
                G3.synchronize()
                d3 = time.perf_counter() - t1
                s3 = G3.stats()
            line["also_pcg_rtol_1e-8"] = {"value": S["m"] * r3["iters"] * reps / d3, "unit": "edge-updates/s",
                                          "iters_to_converge": r3["iters"], "ms_per_step": 1e3 * d3 / reps,
                                          "pcg_iters_per_solve": s3["pcg_iters"] / max(s3["pcg_solves"], 1)}
        if not args.no_extra and world == 1:
            # the callers' real pipeline: l1ra THEN irls (ral/test.cpp:295-301 with its defaults: 5 L1RA and
            # 50 IRLS iterations, change_th 1e-3; src/ViewGraph.cpp:1400-1417 allows 100 L1RA iterations)
            G.restore_rotations()
            G.l1ra(1, 1e-3)            # allocates the primal-dual planes and the solver clones
            # (three host threads drive the three coordinates: a single timing moved by +-25 % from box to box and
            # from call to call; the pipeline runs five times and the MEAN is quoted, the spread next to it)
            reps_p, tl, ti = 5, [], []
            for _ in range(reps_p):
                G.restore_rotations()
                G.synchronize()
                t1 = time.perf_counter()
                ra = G.l1ra(5, 1e-3)
                G.synchronize()
                t2 = time.perf_counter()
                rb = G.irls(4, SIG, 50, 1e-3)
                G.synchronize()
                t3 = time.perf_counter()
                tl.append(t2 - t1)
                ti.append(t3 - t2)
            ml, mi = sum(tl) / reps_p, sum(ti) / reps_p
            line["also_l1ra_then_irls"] = {
                "l1ra_iters": ra["iters"], "l1ra_ms": 1e3 * ml,
                "l1ra_ms_per_outer_iteration": 1e3 * ml / max(ra["iters"], 1),
                "l1ra_ms_min_max": [1e3 * min(tl), 1e3 * max(tl)],
                "irls_iters": rb["iters"], "irls_ms": 1e3 * mi,
                "edge_updates_per_s_whole_pipeline": S["m"] * (ra["iters"] + rb["iters"]) / (ml + mi),
                "reps": reps_p,
                "note": "l1ra(5) then irls(50): the reference demo's defaults; l1ra = 3 coordinate LPs x 2 primal-dual "
                        "iterations per outer iteration, each a Hessian solve by the handle's linear solver; mean of "
                        "5 runs of the pipeline"}
            G.restore_rotations()
            # what a caller of the drop-in irotavg_irls pays with HOST buffers: graph build (adjacency,
            # hierarchy, SELL) + upload + the same solve + download, per call (the resident
            # figure above is the amortised / incremental case)
            # The C call itself on arrays that already have the reference's layout (Eigen column-major Mat, int32
            # pairs): what a C++ caller of irotavg::irls pays. (Until round 2 this went through irotavg_amd/ral.py,
            # whose NumPy layout conversions -- 10-15 ms at this size -- were inside the timed region.)
            import ctypes as C
            QQf, Ie = capi.fmat(S["QQ"]), capi.edges(S["I"])
            reps = 5
            d1 = 0.0
            for rep in range(reps + 1):
                Qf, wh = capi.fmat(Q0), np.zeros(S["m"])
                it_c, rt_c = C.c_int(0), C.c_double(0)
                t1 = time.perf_counter()
                rc1 = capi.lib().irotavg_irls(S["m"], S["n"], 1, capi._i(Ie), capi._d(QQf), S["m"], 4, SIG, capi._d(Qf),
                                              S["n"], 100, 1e-3, capi._d(wh), C.byref(it_c), C.byref(rt_c))
                if rep > 0:
                    d1 += time.perf_counter() - t1
                assert rc1 == 0
            it1 = it_c.value
            line["also_one_shot_host_buffers"] = {
                "value": S["m"] * it1 * reps / d1, "unit": "edge-updates/s", "ms_per_call": 1e3 * d1 / reps,
                "iters_to_converge": it1, "irls_ms_inside": 1e3 * rt_c.value,
                "note": "irotavg_irls from host pointers (pageable memory): handle creation by the device build "
                        "(gbuild.hip) + PCIe both ways + the solve inside the timed region; one untimed call first"}
        if not args.no_cpu and world == 1:
            line["cpu_baseline"] = cpu_baseline(S, Q0, args.p_loop)
            ss = suitesparse_baseline(S, Q0)
            if isinstance(ss, dict):
                line["cpu_baseline_suitesparse"] = ss
            else:
                line["cpu_baseline"]["suitesparse_probe"] = ss
        G.close()
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

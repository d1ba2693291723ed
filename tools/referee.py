#!/usr/bin/env python3
"""Referee for the inputs on which the GPU path and the oracle part ways (TEST INFRASTRUCTURE).

An `irls` run (ral/l1_irls.cpp:559-752) is a fixed-point iteration whose inner step is a linear
least-squares solve (:536-556, SuiteSparseQR in the reference). On a barely connected view-graph
that iteration need not contract, and then the rounding of the INNER solves decides after how many
iterations the strict `score > change_th` test (:590) ends the loop. This tool replays one case
of tools/fuzz_parity.py and runs the same outer iteration -- every statement outside the solve in
plain fp64 exactly as oracle/np_twin.py restates it -- with FOUR exact CPU solves of each system:

  chol   the oracle's sparse Cholesky on the normal equations      (oracle/ral_oracle.c)
  splu   SuperLU on the normal equations                            (oracle/np_twin.py)
  qr     dense Householder QR of the LS form  D A X = D B           (what SPQR factorises, :550)
  ld     dense Cholesky in 80-bit long double + two refinements, rounded to fp64 once
         (the correctly rounded solution of the fp64 normal equations, to ~1e-19 relative)

and, with --gpu, the handle path of libirotavg_hip.so. It prints the score traces, where they fork
(first iteration at which two traces differ by more than --fork-rtol), iteration counts and the
pairwise angular distances of the final rotations, and the conditioning of the last system.

    python tools/referee.py --seed 603 --case 163 [--gpu] [--json out.json]
    python tools/referee.py --npz tests/golden/neartree_seed603_case163.npz --gpu
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from oracle import np_twin as T  # noqa: E402


def chol_ld(H, B):
    """Dense Cholesky of H (fp64 entries) in long double, two rounds of refinement, -> fp64."""
    L = np.array(H, dtype=np.longdouble)
    n = L.shape[0]
    for j in range(n):
        L[j, j] = np.sqrt(L[j, j] - L[j, :j] @ L[j, :j])
        if j + 1 < n:
            L[j + 1:, j] = (L[j + 1:, j] - L[j + 1:, :j] @ L[j, :j]) / L[j, j]
    L = np.tril(L)

    def solve(R):
        Y = np.array(R, dtype=np.longdouble)
        for j in range(n):
            Y[j] = (Y[j] - L[j, :j] @ Y[:j]) / L[j, j]
        for j in range(n - 1, -1, -1):
            Y[j] = (Y[j] - L[j + 1:, j] @ Y[j + 1:]) / L[j, j]
        return Y
    Hl, Bl = np.array(H, dtype=np.longdouble), np.array(B, dtype=np.longdouble)
    X = solve(Bl)
    for _ in range(2):
        X = X + solve(Bl - Hl @ X)
    return np.array(X, dtype=np.float64)


def make_solver(kind, A):
    import scipy.sparse as sp
    import scipy.sparse.linalg as sla
    Ad = A.toarray()

    def splu(weights, w3):
        DA = (sp.diags(weights) @ A).tocsc()
        return sla.splu((DA.T @ DA).tocsc()).solve(DA.T @ (weights[:, None] * w3))

    def qr(weights, w3):
        DA = weights[:, None] * Ad
        Qf, R = np.linalg.qr(DA)           # LAPACK Householder QR of the LS form
        import scipy.linalg as la
        return la.solve_triangular(R, Qf.T @ (weights[:, None] * w3))

    def ld(weights, w3):
        DA = weights[:, None] * Ad
        return chol_ld(DA.T @ DA, DA.T @ (weights[:, None] * w3))
    return dict(splu=splu, qr=qr, ld=ld)[kind]


def irls_with(solver, QQ, I, Q, f, cost, sigma, max_iters, change_th):
    """oracle/np_twin.py's irls with the linear solve handed in (everything else statement for statement)."""
    Q = Q.copy()
    m, n = len(I), len(Q) - f
    A = T.make_A(len(Q), f, I)
    solve = make_solver(solver, A)
    weights = np.ones(m)
    score, iters, scores, Ws = np.inf, 0, [], []
    while score > change_th and iters < max_iters:
        w = T.log_map(T.delta_rel(I, QQ, Q))
        X = solve(weights, w[:, :3])
        E = A @ X - w[:, :3]
        weights = T.weights_update(cost, sigma, E, weights)
        score = np.linalg.norm(X, axis=1).mean()
        scores.append(score)
        Ws.append(weights.copy())
        Q[f:] = T.quat_mult(Q[f:], T.exp_map(np.concatenate([X, np.zeros((n, 1))], axis=1)))
        iters += 1
    return dict(Q=Q, weights=weights, iters=iters, scores=np.array(scores), W=Ws)


def conditioning(n, f, I, weights):
    """lambda_min / kappa of the Jacobi-scaled normal matrix (what fuzz_parity.py's ill-posedness test looks at)."""
    A = T.make_A(n, f, I).toarray()
    H = A.T @ (A * (weights ** 2)[:, None])
    d = np.diag(H).copy()
    if (d <= 0).any():
        return dict(lam_min=0.0, kappa=np.inf, kappa_unscaled=np.inf)
    ev = np.linalg.eigvalsh(H / np.sqrt(np.outer(d, d)))
    evu = np.linalg.eigvalsh(H)
    return dict(lam_min=float(ev[0]), kappa=float(ev[-1] / max(ev[0], 1e-300)),
                kappa_unscaled=float(evu[-1] / max(evu[0], 1e-300)))


def contraction(scores):
    """largest ratio score[k+1] / score[k] over the tail of the run (>= 1: the fixed-point iteration is not contracting)"""
    s = np.asarray(scores)
    if len(s) < 4:
        return 0.0
    return float(np.max(s[3:] / s[2:-1]))


def fork(a, b, rtol):
    k = min(len(a), len(b))
    d = np.nonzero(np.abs(a[:k] - b[:k]) > rtol * np.abs(b[:k]))[0]
    return int(d[0]) if len(d) else (k if len(a) != len(b) else -1)


def load_case(a):
    if a.npz:
        z = np.load(a.npz)
        return dict(n=int(z["n"]), f=int(z["f"]), I=z["I"], QQ=z["QQ"], Q0=z["Q0"], cost=int(z["cost"]))
    import fuzz_parity as F
    rng = np.random.default_rng(a.seed)
    for _ in range(a.case + 1):
        c = F.random_case(rng, 20 if rng.random() < a.small_share else a.nmax)
    return c


def run(c, a):
    from irotavg_amd import synth
    from oracle import oracle as O
    n, f, I, QQ, Q0, cost = c["n"], c["f"], c["I"], c["QQ"], c["Q0"], c["cost"]
    sig = 5 * np.pi / 180
    out = dict(n=n, f=f, m=len(I), cost=cost, cost_name=O.COSTS[cost], l1_iters=a.l1_iters, max_iters=a.max_iters)
    ra = O.l1ra(QQ, I, Q0, f, a.l1_iters, 1e-3)
    rt = T.l1ra(np.asfortranarray(QQ), I, Q0, f, a.l1_iters, 1e-3)
    out["l1ra"] = dict(iters=[int(ra["iters"]), int(rt["iters"])],
                       oracle_vs_twin_rad=float(synth.angular_distance(ra["Q"], rt["Q"]).max()))
    Qa = ra["Q"]
    runs = {"chol": O.irls(QQ, I, Qa, f, cost, sig, a.max_iters, 1e-3)}
    for k in ("splu", "qr", "ld"):
        runs[k] = irls_with(k, QQ, I, np.array(Qa), f, cost, sig, a.max_iters, 1e-3)
    if a.gpu:
        from irotavg_amd import capi
        with capi.Graph(I, QQ, n, f) as G:
            G.set_rotations(Q0)
            ga = G.l1ra(a.l1_iters, 1e-3)
            Qga = G.get_rotations()
            out["l1ra"]["gpu_iters"] = int(ga["iters"])
            out["l1ra"]["gpu_vs_oracle_rad"] = float(synth.angular_distance(Qga, Qa).max())
        with capi.Graph(I, QQ, n, f) as G:     # from the ORACLE's l1ra result: the irls part on its own
            G.set_rotations(Qa)
            gb = G.irls(cost, sig, a.max_iters, 1e-3)
            runs["gpu"] = dict(Q=G.get_rotations(), weights=G.get_weights(), iters=gb["iters"],
                               scores=np.asarray(gb["scores"]))
            st = G.stats()
            out["gpu_stats"] = {k: int(st[k]) for k in ("pcg_solves", "direct_solves", "dense_inversions", "levels")}
    names = list(runs)
    out["iters"] = {k: int(runs[k]["iters"]) for k in names}
    out["scores"] = {k: [float(s) for s in runs[k]["scores"]] for k in names}
    out["fork_vs_ld"] = {k: fork(np.asarray(runs[k]["scores"]), np.asarray(runs["ld"]["scores"]), a.fork_rtol)
                         for k in names if k != "ld"}
    out["final_angle_rad"] = {"%s-%s" % (p, q): float(synth.angular_distance(runs[p]["Q"], runs[q]["Q"]).max())
                              for i, p in enumerate(names) for q in names[i + 1:]}
    out["contraction_tail_max_ratio"] = contraction(runs["ld"]["scores"])
    out["conditioning_last"] = conditioning(n, f, I, runs["ld"]["weights"])
    # the worst system of the run (weights after every iteration of the long-double run)
    ks = [conditioning(n, f, I, w) for w in runs["ld"]["W"]]
    out["conditioning_worst"] = dict(lam_min=min(k["lam_min"] for k in ks), kappa=max(k["kappa"] for k in ks),
                                     kappa_unscaled=max(k["kappa_unscaled"] for k in ks))
    # agreement up to the fork: the part of the traces that IS reproducible
    kf = [v for v in out["fork_vs_ld"].values() if v >= 0]
    out["first_fork"] = min(kf) if kf else -1
    return out, runs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int)
    ap.add_argument("--case", type=int)
    ap.add_argument("--npz")
    ap.add_argument("--nmax", type=int, default=400)
    ap.add_argument("--small-share", type=float, default=0.4)
    ap.add_argument("--l1-iters", type=int, default=3)
    ap.add_argument("--max-iters", type=int, default=15)
    ap.add_argument("--fork-rtol", type=float, default=1e-6)
    ap.add_argument("--gpu", action="store_true")
    ap.add_argument("--json")
    ap.add_argument("--save-npz", help="store the case (inputs only) as a fixture")
    a = ap.parse_args()
    c = load_case(a)
    if a.save_npz:
        np.savez_compressed(a.save_npz, n=c["n"], f=c["f"], I=c["I"], QQ=c["QQ"], Q0=c["Q0"], cost=c["cost"])
    out, _ = run(c, a)
    np.set_printoptions(precision=9, linewidth=200)
    print("case: n %d f %d m %d cost %d (%s); l1ra oracle vs twin %.2e rad" %
          (out["n"], out["f"], out["m"], out["cost"], out["cost_name"], out["l1ra"]["oracle_vs_twin_rad"]))
    for k, s in out["scores"].items():
        print("%-5s iters %2d scores %s" % (k, out["iters"][k], np.asarray(s)))
    print("fork vs ld (first iteration whose score differs by > %g rel; -1: never):" % a.fork_rtol, out["fork_vs_ld"])
    print("final angles (rad):", {k: "%.2e" % v for k, v in out["final_angle_rad"].items()})
    print("contraction (max score ratio in the tail): %.3f; conditioning last %s worst %s" %
          (out["contraction_tail_max_ratio"], out["conditioning_last"], out["conditioning_worst"]))
    if a.json:
        with open(a.json, "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Randomised parity campaign: GPU (handle path, window kernels) vs the CPU oracle on random
view-graphs -- random spanning tree + extra edges in random orientation, duplicate edges, self
loops, several fixed views (incl. edges whose second endpoint is fixed: make_A's dropped rows,
ral/l1_irls.cpp:770-771), every robust cost, outliers. Prints one line per failing case and a
summary; exit code 1 if anything disagreed.

    python tools/fuzz_parity.py --cases 400 --seed 1
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from irotavg_amd import capi, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


STAG = [0, 0]  # cases with a stagnation-accepted solve: [compared, ill-conditioned]
CAPPED = [0]   # runs at the IRLS iteration cap that agree within 1e-4 but not within --tol
DIVERGING = [0]  # capped runs whose scores grow again (not compared beyond the turning point)
AMPLIFYING = [0]  # runs below the cap on an input that amplifies inner rounding (cond > 1e4, tail ratio > 0.7): held to 1e-4


def random_case(rng, nmax):
    n = int(rng.integers(3, nmax + 1))
    f = int(rng.integers(1, max(2, n // 3)))
    f = min(f, n - 2)
    extra = int(rng.integers(0, 6 * n))
    # spanning tree over a random permutation, then extra edges
    perm = rng.permutation(n)
    E = [(int(perm[rng.integers(0, k)]), int(perm[k])) for k in range(1, n)]
    for _ in range(extra):
        a, b = int(rng.integers(0, n)), int(rng.integers(0, n))
        if a == b and rng.random() > 0.05:       # self loops only occasionally
            continue
        E.append((a, b))
    if rng.random() < 0.3 and len(E) > 4:        # duplicates
        E += [E[int(k)] for k in rng.integers(0, len(E), size=max(1, len(E) // 20))]
    I = np.array(E, dtype=np.int32)
    rng.shuffle(I)
    Qgt = rng.normal(size=(n, 4))
    Qgt /= np.linalg.norm(Qgt, axis=1, keepdims=True)
    noise = float(rng.choice([1e-3, 0.01, 0.05]))
    QQ = synth.qmul(synth.qexp(rng.normal(scale=noise, size=(len(I), 3))),
                    synth.qmul(Qgt[I[:, 1]], synth.qconj(Qgt[I[:, 0]])))
    out = rng.random(len(I)) < float(rng.choice([0.0, 0.02, 0.1]))
    if out.any():
        R = rng.normal(size=(int(out.sum()), 4))
        QQ[out] = R / np.linalg.norm(R, axis=1, keepdims=True)
    Q0 = synth.qmul(synth.qexp(rng.normal(scale=float(rng.choice([0.01, 0.1])), size=(n, 3))), Qgt)
    Q0[:f] = Qgt[:f]
    cost = int(rng.integers(0, 14))
    return dict(n=n, f=f, I=I, QQ=QQ, Q0=Q0, cost=cost)


def large_case(rng):
    """Multi-level path (> 2048 free views): band + loop topology of SURVEY.md 8(d), well posed by
    construction (every view keeps band edges), random orientation, duplicates, several fixed."""
    n = int(rng.integers(2100, 3500))
    deg = int(rng.integers(3, 12))
    p = float(rng.choice([0.0, 0.01]))  # the oracle's Cholesky fill explodes with more loop edges
    S = synth.make_graph(n, n * deg, p, seed=int(rng.integers(0, 1 << 30)))
    I, QQ = S["I"].copy(), S["QQ"].copy()
    f = int(rng.choice([1, 1, 3, 40]))
    flip = (rng.random(len(I)) < 0.3) & (I[:, 0] >= f)     # never make a fixed view the 2nd endpoint
    I[flip] = I[flip][:, ::-1]
    QQ[flip] = synth.qconj(QQ[flip])
    if rng.random() < 0.3:
        k = rng.integers(0, len(I), size=len(I) // 50)
        I, QQ = np.concatenate([I, I[k]]).astype(np.int32), np.concatenate([QQ, QQ[k]])
    Q0 = synth.qmul(synth.qexp(rng.normal(scale=0.05, size=(n, 3))), S["Qgt"])
    Q0[:f] = S["Qgt"][:f]
    cost = int(rng.choice([0, 1, 2, 4, 5, 6, 7, 9, 10, 11, 13]))
    return dict(n=n, f=f, I=I, QQ=QQ, Q0=Q0, cost=cost, skip_eig=True)


def check(c, tol, sig):
    """returns a list of failure strings"""
    bad = []
    n, f, I, QQ, Q0, cost = c["n"], c["f"], c["I"], c["QQ"], c["Q0"], c["cost"]
    l1_iters = 3
    ra = O.l1ra(QQ, I, Q0, f, l1_iters, 1e-3)
    rb = O.irls(QQ, I, ra["Q"], f, cost, sig, 15, 1e-3)
    if ra.get("rc", 0) != 0 or rb.get("rc", 0) != 0 or not np.isfinite(rb["Q"]).all():
        # the oracle itself gave up (what makes the reference exit(-1): a solver breakdown inside
        # l1decode_pd, NaNs): not a parity case; record what the GPU path reports for it
        try:
            with capi.Graph(I, QQ, n, f) as G:
                G.set_rotations(Q0)
                G.l1ra(l1_iters, 1e-3)
                G.irls(cost, sig, 15, 1e-3)
            return ["oracle-failed", "gpu ok", "oracle rc %s/%s" % (ra.get("rc"), rb.get("rc"))]
        except capi.IrotavgError as e:
            return ["oracle-failed", "gpu error %d" % e.code, "oracle rc %s/%s" % (ra.get("rc"), rb.get("rc"))]
    # Is the weighted LS problem of the last IRLS iteration well posed? H = A' D^2 A (make_A's A,
    # final weights), Jacobi-scaled; a smallest eigenvalue near zero means that some view or group
    # of views is held by (almost) nothing -- isolated by dropped rows (make_A's quirk), by Talwar's
    # exact zeros or by floor weights. The answer is then defined by the solver's rank decision /
    # rounding (SPQR basic solution, dead pivots, minimum norm), which SURVEY.md 8(c) lists as NOT
    # pinned: such cases must still run without an error, their rotations are not compared.
    lam_min = 1.0
    if c.get("skip_eig"):
        ill = False
    else:
        A = O.make_A(n, f, I).toarray()
        H = A.T @ (A * (rb["weights"] ** 2)[:, None])
        d = np.diag(H).copy()
        if (d <= 0).any():
            ill = True
        else:
            Hs = H / np.sqrt(np.outer(d, d))
            lam_min = float(np.linalg.eigvalsh(Hs)[0])
            ill = lam_min < 1e-7
    if ill:
        try:
            with capi.Graph(I, QQ, n, f) as G:
                G.set_rotations(Q0)
                G.l1ra(l1_iters, 1e-3)
                G.irls(cost, sig, 15, 1e-3)
                ok = np.isfinite(G.get_rotations()).all()
                STAG[1] += int(G.stats()["pcg_stagnated"] > 0)
            return ["ill-conditioned"] if ok else ["ill-conditioned case gave non-finite rotations"]
        except capi.IrotavgError as e:
            return ["ill-conditioned case failed with error %d" % e.code]
    try:
        with capi.Graph(I, QQ, n, f) as G:
            G.set_rotations(Q0)
            a = G.l1ra(l1_iters, 1e-3)
            b = G.irls(cost, sig, 15, 1e-3)
            Q, w = G.get_rotations(), G.get_weights()
            STAG[0] += int(G.stats()["pcg_stagnated"] > 0)
        if (a["iters"], b["iters"]) != (ra["iters"], rb["iters"]):
            bad.append("handle iters %s vs oracle %s" % ((a["iters"], b["iters"]), (ra["iters"], rb["iters"])))
        d = synth.angular_distance(Q, rb["Q"]).max()
        # an IRLS run that hits its iteration cap has not converged: where it is not even contracting (the
        # scores stop falling) the 1e-10 of the inner solves is amplified from iteration to iteration and any
        # two solvers differ -- such runs are held to the north star's 1e-4 rad, counted
        capped = rb["iters"] >= 15
        # ... and so is a run that is AMPLIFYING by a measurable criterion (round 6, tools/referee.py): the scaled normal
        # matrix of the last iteration has lambda_min < 1e-4 (cond > 1e4) AND the oracle's own score trace falls by less
        # than a factor 0.7 somewhere after its third iteration (the outer fixed-point iteration hardly contracts there). On such an input four EXACT
        # CPU solves of every system -- sparse Cholesky, SuperLU, Householder QR of the LS form, long-double Cholesky --
        # end 1.5e-5 rad apart from each other (seed 603 case 163; tests/test_referee.py), so no fp64 solver can be held
        # to 1e-6 there; the iteration COUNT must still be the oracle's (checked above), the angle is held to 1e-4.
        # (seed 701 case 421 moved the numbers: lambda_min 1.3e-5, tail ratio 0.80, 14 iterations -- the QR referee ends
        # 4.9e-6 rad from the three Cholesky-type referees, the GPU 5.6e-6 from the oracle: the class is "cond > 1e4 and the
        # tail of the outer iteration contracts by less than 0.7 per step", rising scores being its extreme)
        sc_all = np.asarray(rb["scores"])[:rb["iters"]]
        amplifying = lam_min < 1e-4 and len(sc_all) >= 5 and float(np.max(sc_all[3:] / sc_all[2:-1])) > 0.7
        if amplifying and not capped:
            AMPLIFYING[0] += 1
            capped = True
        # ... and a capped run whose scores GROW again (to more than twice the score at the first turning point: IRLS is
        # moving away from its fixed point, e.g. Geman-McClure on a tree-like graph, seed 301 case 386; Welsch with
        # scores going up and down by decades, seed 401 case 841) amplifies without bound: the
        # handle of the commit before and after a change, and the oracle, all differ by 1e-3..1e-2 rad there.
        # Counted, held to the first iterations only (the scores up to the smallest one must agree to 1e-6).
        sc_o, sc_g = np.asarray(rb["scores"]), np.asarray(b["scores"])
        rise = np.nonzero(np.diff(sc_o) > 0)[0]          # first iteration whose score is above the one before it
        if capped and d >= 1e-4 and len(rise) and sc_o[rise[0] + 1:].max() > 2 * sc_o[rise[0]]:
            kmin = int(rise[0])
            DIVERGING[0] += 1
            if not np.allclose(sc_g[:kmin + 1], sc_o[:kmin + 1], rtol=1e-4):
                bad.append("diverging IRLS run: scores differ before the turning point")
            return bad
        if capped and d >= tol:
            CAPPED[0] += 1
        if not d < (1e-4 if capped else tol):
            bad.append("handle angular distance %.3e" % d)
        # weights of the L1-type costs blow up (cap 1e4) on edges that are fitted exactly: there the
        # residual is rounding noise and so is the weight -- compare the others
        cmp = rb["weights"] < 1e2
        # (a run held to 1e-4 rad has edge residuals that differ by as much: its weights are held to 1e-2 relative --
        # Welsch at three times its scale turns 5e-5 rad into 9e-4 of a weight, seed 708 case 1787)
        if not np.allclose(w[cmp], rb["weights"][cmp], rtol=1e-2 if capped else 1e-4, atol=1e-9):
            bad.append("handle weights differ (max rel %.2e)" % np.max(np.abs(w - rb["weights"])[cmp] / (np.abs(rb["weights"][cmp]) + 1e-300)))
    except capi.IrotavgError as e:
        bad.append("handle error %d" % e.code)
    nu, m = n - f, len(I)
    for kern, ok in ((1, nu <= 64 and m <= 640 and n <= 320), (2, nu <= 16 and m <= 64 and n <= 320)):
        if not ok:
            continue
        try:
            r = capi.window_solve(I, QQ, Q0, f, cost, sig, l1_iters, 15, 1e-3, kernel=kern)
            if (r["l1_iters"], r["irls_iters"]) != (ra["iters"], rb["iters"]):
                bad.append("window%d iters %s vs %s" % (kern, (r["l1_iters"], r["irls_iters"]), (ra["iters"], rb["iters"])))
            d = synth.angular_distance(r["Q"], rb["Q"]).max()
            if not d < tol:
                bad.append("window%d angular distance %.3e" % (kern, d))
        except capi.IrotavgError as e:
            bad.append("window%d error %d" % (kern, e.code))
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--nmax", type=int, default=400)
    ap.add_argument("--small-share", type=float, default=0.4, help="share of cases with n <= 20 (window kernels)")
    ap.add_argument("--tol", type=float, default=1e-6)
    ap.add_argument("--large", action="store_true", help="multi-level graphs (2100..3500 views) instead")
    ap.add_argument("--device-build", action="store_true",
                    help="build every handle on the device (gbuild.hip); by default graphs below 20000 edges are "
                         "built on the host")
    ap.add_argument("--max-capped", type=int, default=-1,
                    help="runs at the IRLS iteration cap that may lie between --tol and 1e-4 rad before the "
                         "campaign fails; default: 2 + cases // 500 (recorded baseline: 4 in 5300 cases, DESIGN.md 2)")
    a = ap.parse_args()
    if a.device_build:
        os.environ["IROTAVG_HOST_BUILD"] = "0"
    import warnings
    warnings.simplefilter("error")   # a NOT_CONVERGED that ral.py would soften to a warning is a failure here
    rng = np.random.default_rng(a.seed)
    sig = 5 * np.pi / 180
    fails, skipped, ill = 0, 0, 0
    gave_up = {}
    for k in range(a.cases):
        c = large_case(rng) if a.large else random_case(rng, 20 if rng.random() < a.small_share else a.nmax)
        bad = check(c, a.tol, sig)
        if bad[:1] == ["oracle-failed"]:
            skipped += 1
            gave_up[" ".join(bad[1:])] = gave_up.get(" ".join(bad[1:]), 0) + 1
            continue
        if bad == ["ill-conditioned"]:
            ill += 1
            continue
        if bad:
            fails += 1
            print("case %d (n=%d f=%d m=%d cost=%d): %s" % (k, c["n"], c["f"], len(c["I"]), c["cost"], "; ".join(bad)))
    print(json.dumps({"cases": a.cases, "failed": fails, "oracle_gave_up": skipped,
                      "oracle_gave_up_detail": gave_up, "cases_with_stagnation_accepted_solves": STAG,
                      "ill_conditioned_ran_ok_not_compared": ill,
                      "capped_runs_between_tol_and_1e-4": CAPPED[0],
                      "capped_runs_with_growing_scores": DIVERGING[0],
                      "amplifying_runs_held_to_1e-4": AMPLIFYING[0], "seed": a.seed}))
    max_capped = a.max_capped if a.max_capped >= 0 else 2 + a.cases // 500
    if CAPPED[0] > max_capped:
        print("FAILED: %d capped runs between tol and 1e-4 rad (allowed %d)" % (CAPPED[0], max_capped))
        return 1
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())

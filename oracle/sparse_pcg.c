/*
 * sparse_pcg.c -- ORACLE-ONLY iterative solve (test infrastructure, not the product).
 *
 * Second stand-in for the two SuiteSparse calls on the reference path (SuiteSparseQR at
 * ral/l1_irls.cpp:550, UMFPACK at ral/l1_irls.cpp:147-169), for the graphs on which the oracle's
 * own sparse Cholesky (sparse_chol.c: natural / reverse Cuthill-McKee ordering) cannot work: a
 * view sequence with thousands of random loop closures has an envelope of order n^2 under both
 * orderings (100k views / 2M edges with 2 % loop edges: > 10^9 factor entries).
 *
 * Method: conjugate gradients on the same normal equations H x = b, preconditioned by symmetric
 * Gauss-Seidel  M = (D + L) D^-1 (D + L')  (SPD whenever D > 0; needs no factorisation and no fill),
 * run to a TRUE relative residual of 1e-13 -- three decades below the product's default 1e-10, so
 * that the oracle's solve error is negligible next to the product's. The recurrence residual is
 * replaced by b - Hx whenever it claims convergence, and the solve only ends when the replaced
 * residual agrees (or the attainable accuracy is reached: no progress over two restarts).
 * Textbook algorithms (Hestenes-Stiefel; Saad, "Iterative Methods for Sparse Linear Systems",
 * ch. 9-10), written from scratch; nothing here comes from the product's PCG (different
 * preconditioner, different recurrences, CSC with duplicates; the three right-hand sides of an
 * IRLS solve share the passes over the matrix but run independent recurrences).
 *
 * Rank deficiency: an unknown whose diagonal is not above ORA_DEAD_TOL x the largest diagonal (an
 * isolated view) solves to 0 like a dead pivot of sparse_chol.c. A FLOATING component (singular
 * but consistent system) converges to the solution with zero mean start -- a different member of
 * the solution family than the Cholesky's "first dead pivot pinned": solver-defined in the
 * reference as well (Oracle), so such inputs are not compared across solvers.
 */
#include "sparse_pcg.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORA_DEAD_TOL 1e-13
#define ORA_PCG_RTOL 1e-13
#define ORA_PCG_MAXIT 100000

static double *dalloc(long n) { return (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double)); }

double ora_envelope(long n, const long *Ap, const long *Ai) {
    double env = 0.0;
    for (long j = 0; j < n; j++) {
        long lo = j;
        for (long p = Ap[j]; p < Ap[j + 1]; p++)
            if (Ai[p] < lo) lo = Ai[p];
        env += (double)(j - lo + 1) * (double)(j - lo + 1);
    }
    return env;
}

ora_pcg *ora_pcg_alloc(long n) {
    ora_pcg *c = (ora_pcg *)calloc(1, sizeof(ora_pcg));
    if (!c) return NULL;
    c->n = n;
    c->diag = dalloc(n);
    c->dead = (char *)calloc((size_t)(n > 0 ? n : 1), 1);
    c->r = dalloc(3 * n);
    c->z = dalloc(3 * n);
    c->p = dalloc(3 * n);
    c->q = dalloc(3 * n);
    c->xi = dalloc(3 * n);
    if (!c->diag || !c->dead || !c->r || !c->z || !c->p || !c->q || !c->xi) {
        ora_pcg_free(c);
        return NULL;
    }
    return c;
}

void ora_pcg_free(ora_pcg *c) {
    if (!c) return;
    free(c->diag);
    free(c->dead);
    free(c->r);
    free(c->z);
    free(c->p);
    free(c->q);
    free(c->xi);
    free(c);
}

long ora_pcg_setup(ora_pcg *c, const long *Ap, const long *Ai, const double *Ax) {
    const long n = c->n;
    double dmax = 0.0;
    for (long j = 0; j < n; j++) {
        double d = 0.0;
        for (long p = Ap[j]; p < Ap[j + 1]; p++)
            if (Ai[p] == j) d += Ax[p];
        c->diag[j] = d;
        if (d > dmax) dmax = d;
    }
    c->ndead = 0;
    for (long j = 0; j < n; j++) {
        const double d = c->diag[j];
        c->dead[j] = (!(d > ORA_DEAD_TOL * dmax) || !(d > 0.0) || !isfinite(d)) ? 1 : 0;
        c->ndead += c->dead[j];
    }
    return c->ndead;
}

/* Up to ORA_PCG_NRHS right-hand sides share every pass over the matrix: vectors are interleaved
 * (entry k of column c at [k * nr + c]); every column runs its OWN conjugate-gradient recurrence
 * (own alpha, beta, residual norm) -- the sharing is of memory traffic only. */
#define ORA_PCG_NRHS 3

/* y = H x over the live unknowns (column j of the symmetric CSC = row j) */
static void spmv(const ora_pcg *c, const long *Ap, const long *Ai, const double *Ax, int nr,
                 const double *x, double *y) {
    const long n = c->n;
    for (long j = 0; j < n; j++) {
        double s[ORA_PCG_NRHS] = {0.0, 0.0, 0.0};
        if (!c->dead[j])
            for (long p = Ap[j]; p < Ap[j + 1]; p++) {
                const double a = Ax[p];
                const double *xi = x + Ai[p] * nr; /* x of a dead unknown is 0 */
                for (int k = 0; k < nr; k++) s[k] += a * xi[k];
            }
        for (int k = 0; k < nr; k++) y[j * nr + k] = s[k];
    }
}

/* z = M^-1 r, M = (D + L) D^-1 (D + L'): forward sweep, scale, backward sweep */
static void sgs(const ora_pcg *c, const long *Ap, const long *Ai, const double *Ax, int nr,
                const double *r, double *z) {
    const long n = c->n;
    for (long j = 0; j < n; j++) { /* (D + L) y = r */
        double s[ORA_PCG_NRHS] = {0.0, 0.0, 0.0};
        if (c->dead[j]) {
            for (int k = 0; k < nr; k++) z[j * nr + k] = 0.0;
            continue;
        }
        for (int k = 0; k < nr; k++) s[k] = r[j * nr + k];
        for (long p = Ap[j]; p < Ap[j + 1]; p++) {
            const long i = Ai[p];
            if (i < j) {
                const double a = Ax[p];
                for (int k = 0; k < nr; k++) s[k] -= a * z[i * nr + k];
            }
        }
        for (int k = 0; k < nr; k++) z[j * nr + k] = s[k] / c->diag[j];
    }
    for (long j = n - 1; j >= 0; j--) { /* (D + L') z = D y */
        if (c->dead[j]) continue;
        double s[ORA_PCG_NRHS] = {0.0, 0.0, 0.0};
        for (long p = Ap[j]; p < Ap[j + 1]; p++) {
            const long i = Ai[p];
            if (i > j) {
                const double a = Ax[p];
                for (int k = 0; k < nr; k++) s[k] += a * z[i * nr + k];
            }
        }
        for (int k = 0; k < nr; k++) z[j * nr + k] -= s[k] / c->diag[j];
    }
}

static void dots(long n, int nr, const double *a, const double *b, double *out) {
    for (int k = 0; k < nr; k++) out[k] = 0.0;
    for (long j = 0; j < n; j++)
        for (int k = 0; k < nr; k++) out[k] += a[j * nr + k] * b[j * nr + k];
}

int ora_pcg_solve_multi(ora_pcg *c, const long *Ap, const long *Ai, const double *Ax, int nr,
                        const double *b, long ldb, double *x, long ldx) {
    const long n = c->n;
    if (nr < 1 || nr > ORA_PCG_NRHS) return -2;
    double *r = c->r, *z = c->z, *p = c->p, *q = c->q, *xi = c->xi;
    double bb[ORA_PCG_NRHS], tol2[ORA_PCG_NRHS], rz[ORA_PCG_NRHS], rr[ORA_PCG_NRHS];
    double best_true[ORA_PCG_NRHS];
    int settled[ORA_PCG_NRHS], no_gain[ORA_PCG_NRHS];
    for (int k = 0; k < nr; k++) bb[k] = 0.0;
    for (long j = 0; j < n; j++)
        for (int k = 0; k < nr; k++) {
            xi[j * nr + k] = 0.0;
            const double v = c->dead[j] ? 0.0 : b[k * ldb + j];
            r[j * nr + k] = v;
            bb[k] += v * v;
        }
    c->solves++;
    c->iters_last = 0;
    c->relres_last = 0.0;
    int rc = 0, open = 0;
    for (int k = 0; k < nr; k++) {
        tol2[k] = ORA_PCG_RTOL * ORA_PCG_RTOL * bb[k];
        best_true[k] = INFINITY;
        no_gain[k] = 0;
        settled[k] = !(bb[k] > 0.0); /* zero right-hand side: x = 0 */
        open += !settled[k];
    }
    long it = 0;
    while (open > 0) {
        /* (re)start from the true residual */
        sgs(c, Ap, Ai, Ax, nr, r, z);
        memcpy(p, z, sizeof(double) * (size_t)(n * nr));
        dots(n, nr, r, z, rz);
        dots(n, nr, r, r, rr);
        for (;;) {
            int busy = 0;
            for (int k = 0; k < nr; k++) busy += !settled[k] && rr[k] > tol2[k];
            if (!busy || it >= ORA_PCG_MAXIT) break;
            spmv(c, Ap, Ai, Ax, nr, p, q);
            double pq[ORA_PCG_NRHS], alpha[ORA_PCG_NRHS];
            dots(n, nr, p, q, pq);
            int broke = 0;
            for (int k = 0; k < nr; k++) {
                /* a column that is done (or has broken down) stops moving: alpha = 0 */
                const int live = !settled[k] && rr[k] > tol2[k];
                if (live && (!(pq[k] > 0.0) || !isfinite(pq[k]))) broke = 1;
                alpha[k] = (live && pq[k] > 0.0 && isfinite(pq[k])) ? rz[k] / pq[k] : 0.0;
            }
            for (long j = 0; j < n; j++)
                for (int k = 0; k < nr; k++) {
                    xi[j * nr + k] += alpha[k] * p[j * nr + k];
                    r[j * nr + k] -= alpha[k] * q[j * nr + k];
                }
            it++;
            dots(n, nr, r, r, rr);
            if (broke) break;
            sgs(c, Ap, Ai, Ax, nr, r, z);
            double rz_new[ORA_PCG_NRHS];
            dots(n, nr, r, z, rz_new);
            double beta[ORA_PCG_NRHS];
            for (int k = 0; k < nr; k++) {
                beta[k] = (rz[k] > 0.0 && alpha[k] != 0.0) ? rz_new[k] / rz[k] : 0.0;
                if (alpha[k] != 0.0) rz[k] = rz_new[k];
            }
            for (long j = 0; j < n; j++)
                for (int k = 0; k < nr; k++)
                    if (alpha[k] != 0.0) p[j * nr + k] = z[j * nr + k] + beta[k] * p[j * nr + k];
        }
        /* replace the recurrence residuals by the true ones */
        if (getenv("ORA_PCG_DEBUG")) fprintf(stderr, "pcg: it %ld rr/bb %.2e %.2e %.2e\n", it, sqrt(rr[0]/bb[0]), nr>1?sqrt(rr[1]/bb[1]):0., nr>2?sqrt(rr[2]/bb[2]):0.);
        spmv(c, Ap, Ai, Ax, nr, xi, q);
        double tt[ORA_PCG_NRHS] = {0.0, 0.0, 0.0};
        for (long j = 0; j < n; j++)
            for (int k = 0; k < nr; k++) {
                const double v = c->dead[j] ? 0.0 : b[k * ldb + j] - q[j * nr + k];
                r[j * nr + k] = v;
                tt[k] += v * v;
            }
        for (int k = 0; k < nr; k++) {
            if (settled[k]) continue;
            const double rel = sqrt(tt[k] / bb[k]);
            int done = tt[k] <= 4.0 * tol2[k]; /* the true residual agrees (factor 2 in norm) */
            if (!done) {
                if (!(tt[k] < 0.25 * best_true[k])) no_gain[k]++;
                if (tt[k] < best_true[k]) best_true[k] = tt[k];
                if (no_gain[k] >= 2 || it >= ORA_PCG_MAXIT) { /* attainable accuracy, or the cap */
                    done = 1;
                    if (rel > 1e-9) rc = -1;
                }
            }
            if (done) {
                settled[k] = 1;
                open--;
                if (rel > c->relres_last) c->relres_last = rel;
            }
        }
    }
    for (long j = 0; j < n; j++)
        for (int k = 0; k < nr; k++) x[k * ldx + j] = xi[j * nr + k];
    c->iters_last = it;
    c->iters_total += it;
    return rc;
}

int ora_pcg_solve(ora_pcg *c, const long *Ap, const long *Ai, const double *Ax, const double *b,
                  double *x) {
    return ora_pcg_solve_multi(c, Ap, Ai, Ax, 1, b, c->n, x, c->n);
}

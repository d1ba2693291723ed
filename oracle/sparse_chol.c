/*
 * sparse_chol.c -- ORACLE-ONLY sparse Cholesky (test infrastructure, not the product).
 *
 * Stands in for the two SuiteSparse calls on the reference path, neither of which is vendored
 * by the reference nor installed in this image:
 *   - SuiteSparseQR<double>(A,B)  (ral/l1_irls.cpp:550)  -> solved here through the normal
 *     equations (A'D^2A) X = A'D^2 w, which have the same unique solution when A has full
 *     column rank;
 *   - umfpack_dl_{symbolic,numeric,solve} (ral/l1_irls.cpp:147-169) on the SPD matrix
 *     A' diag(sigx) A.
 *
 * Method: fill-reducing ordering chosen between natural and reverse Cuthill-McKee by exact
 * symbolic fill count; elimination tree + up-looking (row-by-row) numeric factorisation
 * L L' = P H P'; two triangular solves per right-hand side. Textbook algorithm
 * (George & Liu; Davis, "Direct Methods for Sparse Linear Systems", ch. 4), written from
 * scratch.
 *
 * A pivot that is not above ORA_DEAD_TOL times the LARGEST diagonal entry of the matrix (or is
 * NaN) marks the variable "dead": its row/column is dropped and the variable solves to 0, mimicking the basic
 * solution SPQR returns for a column it finds dependent (rank detection with a relative
 * tolerance): an isolated view, or one view of a component that no surviving edge ties to a
 * fixed view. The number of dead pivots is reported so callers can treat it as a breakdown.
 */
#include "sparse_chol.h"

#define ORA_DEAD_TOL 1e-13

#include <math.h>
#include <stdlib.h>
#include <string.h>

static long *lalloc(long n) { return (long *)malloc(sizeof(long) * (size_t)(n > 0 ? n : 1)); }

/* ---- reverse Cuthill-McKee on the pattern of a symmetric matrix (both triangles, CSC) ---- */
typedef struct {
    long deg, v;
} degv;
static int cmp_degv(const void *a, const void *b) {
    const degv *x = (const degv *)a, *y = (const degv *)b;
    if (x->deg != y->deg) return x->deg < y->deg ? -1 : 1;
    return x->v < y->v ? -1 : (x->v > y->v);
}

static void rcm_order(long n, const long *Ap, const long *Ai, long *perm) {
    long *queue = lalloc(n), *mark = lalloc(n);
    degv *buf = (degv *)malloc(sizeof(degv) * (size_t)(n > 0 ? n : 1));
    long head = 0, tail = 0;
    for (long v = 0; v < n; v++) mark[v] = 0;
    for (long s0 = 0; s0 < n; s0++) {
        if (mark[s0]) continue;
        /* pseudo-peripheral start: two BFS passes from s0 */
        long start = s0;
        for (int pass = 0; pass < 2; pass++) {
            long h = tail, t = tail;
            queue[t++] = start;
            mark[start] = 2;
            long last = start;
            while (h < t) {
                long v = queue[h++];
                last = v;
                for (long p = Ap[v]; p < Ap[v + 1]; p++) {
                    long u = Ai[p];
                    if (u != v && mark[u] == 0) {
                        mark[u] = 2;
                        queue[t++] = u;
                    }
                }
            }
            for (long q = tail; q < t; q++) mark[queue[q]] = 0;
            /* among the last BFS level pick the last visited (far end) */
            start = last;
        }
        /* Cuthill-McKee BFS from start, neighbours by increasing degree */
        queue[tail++] = start;
        mark[start] = 1;
        while (head < tail) {
            long v = queue[head++];
            long nb = 0;
            for (long p = Ap[v]; p < Ap[v + 1]; p++) {
                long u = Ai[p];
                if (u != v && !mark[u]) {
                    mark[u] = 1;
                    buf[nb].deg = Ap[u + 1] - Ap[u];
                    buf[nb].v = u;
                    nb++;
                }
            }
            qsort(buf, (size_t)nb, sizeof(degv), cmp_degv);
            for (long q = 0; q < nb; q++) queue[tail++] = buf[q].v;
        }
    }
    for (long k = 0; k < n; k++) perm[k] = queue[n - 1 - k]; /* reverse */
    free(queue);
    free(mark);
    free(buf);
}

/* upper triangle (rows <= col) of C = P A P' in CSC; perm[new] = old. Values optional. */
static void permute_upper(long n, const long *Ap, const long *Ai, const double *Ax,
                          const long *iperm, long *Cp, long *Ci, double *Cx) {
    long *cnt = lalloc(n);
    for (long k = 0; k < n; k++) cnt[k] = 0;
    for (long j = 0; j < n; j++) {
        long j2 = iperm[j];
        for (long p = Ap[j]; p < Ap[j + 1]; p++) {
            long i2 = iperm[Ai[p]];
            if (i2 <= j2) cnt[j2]++; /* entry (i2,j2) with i2<=j2; symmetric twin skipped */
        }
    }
    Cp[0] = 0;
    for (long k = 0; k < n; k++) Cp[k + 1] = Cp[k] + cnt[k];
    for (long k = 0; k < n; k++) cnt[k] = Cp[k];
    for (long j = 0; j < n; j++) {
        long j2 = iperm[j];
        for (long p = Ap[j]; p < Ap[j + 1]; p++) {
            long i2 = iperm[Ai[p]];
            if (i2 <= j2) {
                long q = cnt[j2]++;
                Ci[q] = i2;
                if (Cx) Cx[q] = Ax[p];
            }
        }
    }
    free(cnt);
}

/* elimination tree of C (upper CSC) */
static void etree(long n, const long *Cp, const long *Ci, long *parent) {
    long *anc = lalloc(n);
    for (long k = 0; k < n; k++) {
        parent[k] = -1;
        anc[k] = -1;
        for (long p = Cp[k]; p < Cp[k + 1]; p++) {
            long i = Ci[p];
            while (i != -1 && i < k) {
                long next = anc[i];
                anc[i] = k;
                if (next == -1) parent[i] = k;
                i = next;
            }
        }
    }
    free(anc);
}

/* pattern of row k of L: nodes reachable from the entries of column k of C through the etree.
 * Returns top; pattern is stack[top..n-1] in topological order. flag[] uses stamp k. */
static long row_reach(long k, const long *Cp, const long *Ci, const long *parent, long n,
                      long *stack, long *path, long *flag) {
    long top = n;
    flag[k] = k;
    for (long p = Cp[k]; p < Cp[k + 1]; p++) {
        long i = Ci[p];
        if (i >= k) continue;
        long len = 0;
        while (flag[i] != k) {
            path[len++] = i;
            flag[i] = k;
            i = parent[i];
        }
        while (len > 0) stack[--top] = path[--len];
    }
    return top;
}

static long symbolic_counts(long n, const long *Cp, const long *Ci, const long *parent,
                            long *colcount) {
    long *stack = lalloc(n), *path = lalloc(n), *flag = lalloc(n);
    long total = 0;
    for (long k = 0; k < n; k++) {
        colcount[k] = 1;
        flag[k] = -1;
    }
    for (long k = 0; k < n; k++) {
        long top = row_reach(k, Cp, Ci, parent, n, stack, path, flag);
        for (long q = top; q < n; q++) colcount[stack[q]]++;
    }
    for (long k = 0; k < n; k++) total += colcount[k];
    free(stack);
    free(path);
    free(flag);
    return total;
}

ora_chol *ora_chol_analyze(long n, const long *Ap, const long *Ai) {
    ora_chol *c = (ora_chol *)calloc(1, sizeof(ora_chol));
    if (!c) return NULL;
    c->n = n;
    long nnzA = Ap[n];
    long *perm_nat = lalloc(n), *perm_rcm = lalloc(n), *iperm = lalloc(n);
    long *Cp = lalloc(n + 1), *Ci = lalloc(nnzA), *parent = lalloc(n), *cc = lalloc(n);
    long best = -1;
    long *best_perm = NULL;
    for (int cand = 0; cand < 2; cand++) {
        long *perm = cand == 0 ? perm_nat : perm_rcm;
        if (cand == 0)
            for (long k = 0; k < n; k++) perm[k] = k;
        else
            rcm_order(n, Ap, Ai, perm);
        for (long k = 0; k < n; k++) iperm[perm[k]] = k;
        permute_upper(n, Ap, Ai, NULL, iperm, Cp, Ci, NULL);
        etree(n, Cp, Ci, parent);
        long fill = symbolic_counts(n, Cp, Ci, parent, cc);
        if (best < 0 || fill < best) {
            best = fill;
            best_perm = perm;
        }
    }
    c->perm = lalloc(n);
    c->iperm = lalloc(n);
    memcpy(c->perm, best_perm, sizeof(long) * (size_t)n);
    for (long k = 0; k < n; k++) c->iperm[c->perm[k]] = k;
    /* final symbolic with the chosen ordering */
    c->Cp = Cp;
    c->Ci = Ci;
    c->Cx = (double *)malloc(sizeof(double) * (size_t)(nnzA > 0 ? nnzA : 1));
    permute_upper(n, Ap, Ai, NULL, c->iperm, c->Cp, c->Ci, NULL);
    c->parent = parent;
    etree(n, c->Cp, c->Ci, c->parent);
    c->lnz = symbolic_counts(n, c->Cp, c->Ci, c->parent, cc);
    c->Lp = lalloc(n + 1);
    c->Lp[0] = 0;
    for (long k = 0; k < n; k++) c->Lp[k + 1] = c->Lp[k] + cc[k];
    c->Li = lalloc(c->lnz);
    c->Lx = (double *)malloc(sizeof(double) * (size_t)(c->lnz > 0 ? c->lnz : 1));
    c->dead = (char *)calloc((size_t)(n > 0 ? n : 1), 1);
    free(perm_nat);
    free(perm_rcm);
    free(iperm);
    free(cc);
    if (!c->Li || !c->Lx || !c->Cx) {
        ora_chol_free(c);
        return NULL;
    }
    return c;
}

long ora_chol_factor(ora_chol *c, const long *Ap, const long *Ai, const double *Ax) {
    const long n = c->n;
    long *stack = lalloc(n), *path = lalloc(n), *flag = lalloc(n), *fillpos = lalloc(n);
    double *x = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
    long ndead = 0;
    permute_upper(n, Ap, Ai, Ax, c->iperm, c->Cp, c->Ci, c->Cx);
    /* permute_upper may list duplicate (i,j) entries; they are summed by the scatter below */
    double dmax = 0.0; /* largest diagonal entry: the scale of the rounding in every pivot */
    for (long k = 0; k < n; k++) {
        flag[k] = -1;
        fillpos[k] = c->Lp[k];
        c->dead[k] = 0;
        double d = 0.0;
        for (long p = c->Cp[k]; p < c->Cp[k + 1]; p++)
            if (c->Ci[p] == k) d += c->Cx[p];
        if (d > dmax) dmax = d;
    }
    for (long k = 0; k < n; k++) {
        long top = row_reach(k, c->Cp, c->Ci, c->parent, n, stack, path, flag);
        double d = 0.0;
        for (long p = c->Cp[k]; p < c->Cp[k + 1]; p++) {
            long i = c->Ci[p];
            if (i < k)
                x[i] += c->Cx[p];
            else if (i == k)
                d += c->Cx[p];
        }
        for (long q = top; q < n; q++) {
            long i = stack[q];
            double lki;
            if (c->dead[i]) {
                lki = 0.0;
            } else {
                lki = x[i] / c->Lx[c->Lp[i]];
            }
            x[i] = 0.0;
            for (long p = c->Lp[i] + 1; p < fillpos[i]; p++) x[c->Li[p]] -= c->Lx[p] * lki;
            d -= lki * lki;
            long p = fillpos[i]++;
            c->Li[p] = k;
            c->Lx[p] = lki;
        }
        long p = fillpos[k]++;
        c->Li[p] = k;
        if (!(d > ORA_DEAD_TOL * dmax) || !(d > 0.0) || !isfinite(d)) { /* dead pivot */
            c->dead[k] = 1;
            c->Lx[p] = 1.0;
            ndead++;
        } else {
            c->Lx[p] = sqrt(d);
        }
    }
    c->ndead = ndead;
    free(stack);
    free(path);
    free(flag);
    free(fillpos);
    free(x);
    return ndead;
}

void ora_chol_solve(const ora_chol *c, const double *b, double *xout) {
    const long n = c->n;
    double *y = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    for (long k = 0; k < n; k++) y[k] = b[c->perm[k]];
    /* L y = b : column-oriented forward substitution */
    for (long j = 0; j < n; j++) {
        if (c->dead[j]) {
            y[j] = 0.0;
            continue;
        }
        y[j] /= c->Lx[c->Lp[j]];
        const double yj = y[j];
        for (long p = c->Lp[j] + 1; p < c->Lp[j + 1]; p++) y[c->Li[p]] -= c->Lx[p] * yj;
    }
    /* L' x = y */
    for (long j = n - 1; j >= 0; j--) {
        if (c->dead[j]) {
            y[j] = 0.0;
            continue;
        }
        double s = y[j];
        for (long p = c->Lp[j] + 1; p < c->Lp[j + 1]; p++) s -= c->Lx[p] * y[c->Li[p]];
        y[j] = s / c->Lx[c->Lp[j]];
    }
    for (long k = 0; k < n; k++) xout[c->perm[k]] = y[k];
    free(y);
}

void ora_chol_free(ora_chol *c) {
    if (!c) return;
    free(c->perm);
    free(c->iperm);
    free(c->parent);
    free(c->Cp);
    free(c->Ci);
    free(c->Cx);
    free(c->Lp);
    free(c->Li);
    free(c->Lx);
    free(c->dead);
    free(c);
}

/*
 * ral_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, NOT THE PRODUCT). PARITY UNPINNED
 * (see irotavg_oracle.h for why and for what pins it instead).
 *
 * Restates, function by function, the arithmetic of the reference's RAL library. Each function
 * cites the reference lines it follows; the code itself is written from those semantics.
 */
#include "irotavg_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "sparse_chol.h"
#include "sparse_pcg.h"

#define ORA_PI 3.141592653589793238462643383279502884 /* EIGEN_PI */

/* ---------------------------------------------------------------------------------------------
 * quaternion algebra
 * ------------------------------------------------------------------------------------------ */

/* ral/l1_irls.cpp:99-105 -- Hamilton product through Eigen::Quaterniond(w,x,y,z) *= ; no
 * normalisation. Rows are [x y z w]. */
void ora_quat_mult(const double a[4], const double b[4], double out[4]) {
    const double ax = a[0], ay = a[1], az = a[2], aw = a[3];
    const double bx = b[0], by = b[1], bz = b[2], bw = b[3];
    const double w = aw * bw - ax * bx - ay * by - az * bz;
    const double x = aw * bx + ax * bw + ay * bz - az * by;
    const double y = aw * by + ay * bw + az * bx - ax * bz;
    const double z = aw * bz + az * bw + ax * by - ay * bx;
    out[0] = x;
    out[1] = y;
    out[2] = z;
    out[3] = w;
}

static void load_row(const double *M, long ld, long r, double q[4]) {
    q[0] = M[r];
    q[1] = M[ld + r];
    q[2] = M[2 * ld + r];
    q[3] = M[3 * ld + r];
}
static void store_row(double *M, long ld, long r, const double q[4]) {
    M[r] = q[0];
    M[ld + r] = q[1];
    M[2 * ld + r] = q[2];
    M[3 * ld + r] = q[3];
}

/* ral/l1_irls.cpp:109-127 -- the "inverse" of Q_j is Q_j with ONLY w negated (:114-115),
 * i.e. -conj(Q_j); the sign is absorbed by the wrap in log_map. */
void ora_delta_rel(long m, const int *I, const double *QQ, long ldqq, const double *Q, long ldq,
                   double *out, long ldo) {
    for (long k = 0; k < m; k++) {
        const long i = I[2 * k], j = I[2 * k + 1];
        double qi[4], qj[4], qq[4], t[4], r[4];
        load_row(Q, ldq, i, qi);
        load_row(Q, ldq, j, qj);
        qj[3] = -qj[3];
        load_row(QQ, ldqq, k, qq);
        ora_quat_mult(qq, qi, t);
        ora_quat_mult(qj, t, r);
        store_row(out, ldo, k, r);
    }
}

/* ral/l1_irls.cpp:498-532 -- theta = 2 atan2(|xyz|, w) wrapped into [-pi, pi) (:510-517),
 * xyz *= theta/|xyz|, column 3 <- theta, rows with |xyz| < EPS get xyz = 0 (:527-531). */
void ora_log_map(long m, double *w, long ld) {
    for (long k = 0; k < m; k++) {
        const double x = w[k], y = w[ld + k], z = w[2 * ld + k];
        const double s2 = sqrt(x * x + y * y + z * z);
        double theta = 2 * atan2(s2, w[3 * ld + k]);
        if (theta < -ORA_PI)
            theta += 2 * ORA_PI;
        else if (theta >= ORA_PI)
            theta -= 2 * ORA_PI;
        w[3 * ld + k] = theta;
        const double aux = theta / s2;
        w[k] = x * aux;
        w[ld + k] = y * aux;
        w[2 * ld + k] = z * aux;
        if (s2 < ORA_EPS) {
            w[k] = 0;
            w[ld + k] = 0;
            w[2 * ld + k] = 0;
        }
    }
}

/* ral/l1_irls.cpp:471-492 -- theta = |xyz|; xyz *= sin(theta/2)/theta; w = cos(theta/2);
 * every non-finite entry -> 0 (:491), so theta = 0 yields (0,0,0,1). */
void ora_exp_map(long n, double *W, long ld) {
    for (long k = 0; k < n; k++) {
        const double x = W[k], y = W[ld + k], z = W[2 * ld + k];
        const double theta = sqrt(x * x + y * y + z * z);
        const double coef = sin(theta / 2.0) / theta;
        double o[4] = {x * coef, y * coef, z * coef, cos(theta / 2.0)};
        for (int c = 0; c < 4; c++)
            if (!isfinite(o[c])) o[c] = 0.0;
        store_row(W, ld, k, o);
    }
}

/* ral/l1_irls.cpp:982-991 -- Eigen normalized() */
void ora_quat_normalised(long n, double *Q, long ldq, int f) {
    for (long i = f; i < n; i++) {
        double q[4];
        load_row(Q, ldq, i, q);
        const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
        if (n2 > 0.0) {
            const double nn = sqrt(n2);
            for (int c = 0; c < 4; c++) q[c] /= nn;
        }
        store_row(Q, ldq, i, q);
    }
}

/* ---------------------------------------------------------------------------------------------
 * incidence structure (make_A semantics)
 * ------------------------------------------------------------------------------------------ */

/* Row k of A as up to two (column, value) entries. ral/l1_irls.cpp:764-777:
 *   j = e2-f; if (j<0) continue;      -> nothing at all for this edge (even if e1 is free)
 *   A(k,j) = 1;
 *   i = e1-f; if (i<0) continue;
 *   A(k,i) = -1;                      -> overwrites the +1 when i == j (self loop)
 * Returns the number of entries (0,1,2). */
static int edge_row(const int *I, long k, int f, long col[2], double val[2]) {
    const long j = (long)I[2 * k + 1] - f;
    const long i = (long)I[2 * k] - f;
    if (j < 0) return 0;
    if (i < 0) {
        col[0] = j;
        val[0] = 1.0;
        return 1;
    }
    if (i == j) {
        col[0] = i;
        val[0] = -1.0;
        return 1;
    }
    col[0] = j;
    val[0] = 1.0;
    col[1] = i;
    val[1] = -1.0;
    return 2;
}

long ora_make_A(int n, int f, long m, const int *I, long *colptr, long *rowidx, double *vals) {
    if (n < 0 || f < 0 || n - f <= 1) return ORA_ERR_BAD_ARG; /* asserts at :757-758 */
    const long nu = (long)n - f;
    long *cnt = (long *)calloc((size_t)nu + 1, sizeof(long));
    if (!cnt) return ORA_ERR_NOMEM;
    long col[2];
    double val[2];
    for (long k = 0; k < m; k++) {
        int ne = edge_row(I, k, f, col, val);
        for (int e = 0; e < ne; e++) cnt[col[e]]++;
    }
    colptr[0] = 0;
    for (long c = 0; c < nu; c++) colptr[c + 1] = colptr[c] + cnt[c];
    for (long c = 0; c < nu; c++) cnt[c] = colptr[c];
    /* k ascending => rows sorted within each column, as Eigen's compressed storage holds them */
    for (long k = 0; k < m; k++) {
        int ne = edge_row(I, k, f, col, val);
        for (int e = 0; e < ne; e++) {
            long p = cnt[col[e]]++;
            rowidx[p] = k;
            vals[p] = val[e];
        }
    }
    free(cnt);
    return colptr[nu];
}

/* y (m) = A x (n_u) */
static void A_mul(long m, int f, const int *I, const double *x, double *y) {
    long col[2];
    double val[2];
    for (long k = 0; k < m; k++) {
        int ne = edge_row(I, k, f, col, val);
        double s = 0.0;
        for (int e = 0; e < ne; e++) s += val[e] * x[col[e]];
        y[k] = s;
    }
}
/* x (n_u) = A' y (m) */
static void At_mul(long m, long nu, int f, const int *I, const double *y, double *x) {
    long col[2];
    double val[2];
    for (long c = 0; c < nu; c++) x[c] = 0.0;
    for (long k = 0; k < m; k++) {
        int ne = edge_row(I, k, f, col, val);
        for (int e = 0; e < ne; e++) x[col[e]] += val[e] * y[k];
    }
}

/* ---------------------------------------------------------------------------------------------
 * weighted Laplacian assembly + solve (stands in for SPQR / UMFPACK)
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    long nu, nnz;
    long *Ap, *Ai; /* CSC, full symmetric pattern: one diagonal entry per column + 2 per coupled edge */
    double *Ax;
    long *diag_pos;  /* position of the diagonal entry of column c */
    long *edge_pos;  /* 2 per edge: positions of (i,j) and (j,i) entries, or -1 */
    ora_chol *chol;  /* direct solve (sparse_chol.c) ... */
    ora_pcg *pcg;    /* ... or, when its fill would be prohibitive, the iterative one (sparse_pcg.c) */
} lap_sys;

/* which of the two stand-ins for SuiteSparse solves the systems: 0 = by the graph (Cholesky unless
 * there are more than ORA_PCG_MIN_ROWS unknowns AND the envelope work of the natural ordering -- sum
 * of squared envelope heights -- exceeds ORA_ENVELOPE_MAX: band graph of 1M views, 20 wide: 4e8;
 * 10k views with 3000 loop closures: 3e10, 105 s per l1ra(2)+irls by Cholesky, 7 s by PCG),
 * 1 = Cholesky, 2 = PCG. Tests force 2 to validate the PCG against the Cholesky on sizes both can do. */
#define ORA_ENVELOPE_MAX 1.0e10
#define ORA_PCG_MIN_ROWS 5000
static int g_solver_mode = 0;
static long g_pcg_solves = 0, g_pcg_iters = 0, g_chol_solves = 0;
static double g_pcg_worst_relres = 0.0;
void ora_set_solver(int mode) { g_solver_mode = mode; }
void ora_solver_stats(long *chol_solves, long *pcg_solves, long *pcg_iters, double *pcg_worst_relres,
                      int reset) {
    if (chol_solves) *chol_solves = g_chol_solves;
    if (pcg_solves) *pcg_solves = g_pcg_solves;
    if (pcg_iters) *pcg_iters = g_pcg_iters;
    if (pcg_worst_relres) *pcg_worst_relres = g_pcg_worst_relres;
    if (reset) {
        g_pcg_solves = g_pcg_iters = g_chol_solves = 0;
        g_pcg_worst_relres = 0.0;
    }
}

static void lap_free(lap_sys *S) {
    if (!S) return;
    free(S->Ap);
    free(S->Ai);
    free(S->Ax);
    free(S->diag_pos);
    free(S->edge_pos);
    ora_chol_free(S->chol);
    ora_pcg_free(S->pcg);
    free(S);
}

static lap_sys *lap_build(long m, long n_total, int f, const int *I) {
    lap_sys *S = (lap_sys *)calloc(1, sizeof(lap_sys));
    if (!S) return NULL;
    const long nu = n_total - f;
    S->nu = nu;
    long *cnt = (long *)calloc((size_t)nu + 1, sizeof(long));
    for (long c = 0; c < nu; c++) cnt[c] = 1; /* diagonal */
    for (long k = 0; k < m; k++) {
        const long i = (long)I[2 * k] - f, j = (long)I[2 * k + 1] - f;
        if (i >= 0 && j >= 0 && i != j) {
            cnt[i]++;
            cnt[j]++;
        }
    }
    S->Ap = (long *)malloc(sizeof(long) * (size_t)(nu + 1));
    S->Ap[0] = 0;
    for (long c = 0; c < nu; c++) S->Ap[c + 1] = S->Ap[c] + cnt[c];
    S->nnz = S->Ap[nu];
    S->Ai = (long *)malloc(sizeof(long) * (size_t)(S->nnz > 0 ? S->nnz : 1));
    S->Ax = (double *)malloc(sizeof(double) * (size_t)(S->nnz > 0 ? S->nnz : 1));
    S->diag_pos = (long *)malloc(sizeof(long) * (size_t)(nu > 0 ? nu : 1));
    S->edge_pos = (long *)malloc(sizeof(long) * (size_t)(2 * m > 0 ? 2 * m : 1));
    for (long c = 0; c < nu; c++) {
        cnt[c] = S->Ap[c];
        S->diag_pos[c] = cnt[c];
        S->Ai[cnt[c]++] = c;
    }
    for (long k = 0; k < m; k++) {
        const long i = (long)I[2 * k] - f, j = (long)I[2 * k + 1] - f;
        if (i >= 0 && j >= 0 && i != j) {
            long p = cnt[j]++; /* entry (row i, col j) */
            S->Ai[p] = i;
            S->edge_pos[2 * k] = p;
            p = cnt[i]++; /* entry (row j, col i) */
            S->Ai[p] = j;
            S->edge_pos[2 * k + 1] = p;
        } else {
            S->edge_pos[2 * k] = S->edge_pos[2 * k + 1] = -1;
        }
    }
    free(cnt);
    int use_pcg = g_solver_mode == 2;
    if (g_solver_mode == 0) {
        const char *e = getenv("ORA_SOLVER");
        if (e && e[0] == 'p')
            use_pcg = 1;
        else if (!(e && e[0] == 'c'))
            use_pcg = nu > ORA_PCG_MIN_ROWS && ora_envelope(nu, S->Ap, S->Ai) > ORA_ENVELOPE_MAX;
    }
    if (use_pcg)
        S->pcg = ora_pcg_alloc(nu);
    else
        S->chol = ora_chol_analyze(nu, S->Ap, S->Ai);
    if (!S->chol && !S->pcg) {
        lap_free(S);
        return NULL;
    }
    return S;
}

/* "factorise" the values in S->Ax; returns the number of dead unknowns */
static long lap_factor(lap_sys *S) {
    if (S->pcg) return ora_pcg_setup(S->pcg, S->Ap, S->Ai, S->Ax);
    return ora_chol_factor(S->chol, S->Ap, S->Ai, S->Ax);
}

static int lap_solve(lap_sys *S, const double *b, double *x) {
    if (S->pcg) {
        int rc = ora_pcg_solve(S->pcg, S->Ap, S->Ai, S->Ax, b, x);
        g_pcg_solves++;
        g_pcg_iters += S->pcg->iters_last;
        if (S->pcg->relres_last > g_pcg_worst_relres) g_pcg_worst_relres = S->pcg->relres_last;
        return rc ? ORA_ERR_SOLVER : ORA_OK;
    }
    ora_chol_solve(S->chol, b, x);
    g_chol_solves++;
    return ORA_OK;
}

/* H = A' diag(s) A with A from make_A (edge-drop quirk included): what SPQR implicitly
 * factorises at ral/l1_irls.cpp:604,612 with s = weights^2. */
static void lap_fill_AtSA(lap_sys *S, long m, int f, const int *I, const double *s) {
    long col[2];
    double val[2];
    for (long p = 0; p < S->nnz; p++) S->Ax[p] = 0.0;
    for (long k = 0; k < m; k++) {
        int ne = edge_row(I, k, f, col, val);
        for (int e = 0; e < ne; e++) S->Ax[S->diag_pos[col[e]]] += s[k] * val[e] * val[e];
        if (ne == 2) {
            S->Ax[S->edge_pos[2 * k]] += s[k] * val[0] * val[1];
            S->Ax[S->edge_pos[2 * k + 1]] += s[k] * val[0] * val[1];
        }
    }
}

/* H = reshape(AtA * s) with AtA from make_AtA, ral/l1_irls.cpp:811-848: endpoints below f are
 * skipped INDEPENDENTLY (:825-843) -- unlike make_A, an edge whose 2nd endpoint is fixed still
 * adds +s to (i,i). For i == j the four coeffRef assignments land on one entry whose final
 * value is -1. */
static void lap_fill_AtA_times(lap_sys *S, long m, int f, const int *I, const double *s) {
    for (long p = 0; p < S->nnz; p++) S->Ax[p] = 0.0;
    for (long k = 0; k < m; k++) {
        const long i = (long)I[2 * k] - f, j = (long)I[2 * k + 1] - f;
        if (i >= 0 && j >= 0 && i == j) {
            S->Ax[S->diag_pos[i]] += -s[k];
            continue;
        }
        if (i >= 0) S->Ax[S->diag_pos[i]] += s[k];
        if (j >= 0) S->Ax[S->diag_pos[j]] += s[k];
        if (i >= 0 && j >= 0) {
            S->Ax[S->edge_pos[2 * k]] += -s[k];
            S->Ax[S->edge_pos[2 * k + 1]] += -s[k];
        }
    }
}

/* ---------------------------------------------------------------------------------------------
 * IRLS
 * ------------------------------------------------------------------------------------------ */

/* ral/l1_irls.cpp:617-727 -- weight update from the residual rows E (e2 = |E|^2). Returns
 * ORA_ERR_UNKNOWN_COST for an enum value outside 0..13 (:723-726). */
static int update_weights(int cost, double sigma, long m, const double *e2v, double *weights) {
    switch (cost) {
    case ORA_L2:
        break;
    case ORA_L05:
        for (long k = 0; k < m; k++) {
            double w = 1.0 / pow(e2v[k], 3. / 8.);
            if (w > 1e4) w = 1e4;
            weights[k] = w;
        }
        break;
    case ORA_L1:
        for (long k = 0; k < m; k++) {
            double w = 1.0 / sqrt(sqrt(e2v[k]));
            if (w > 1e4) w = 1e4;
            weights[k] = w;
        }
        break;
    case ORA_L15:
        for (long k = 0; k < m; k++) {
            double w = 1.0 / sqrt(sqrt(sqrt(e2v[k])));
            if (w > 1e4) w = 1e4;
            weights[k] = w;
        }
        break;
    case ORA_GEMAN_MCCLURE: {
        const double tun = sigma;
        for (long k = 0; k < m; k++) weights[k] = 1.0 / (e2v[k] + tun * tun);
        break;
    }
    case ORA_HUBER: { /* only e >= 1 is touched; the rest keep their previous value (:647-649) */
        const double tun = 1.345 * sigma;
        for (long k = 0; k < m; k++) {
            const double e = sqrt(e2v[k]) / tun;
            if (e >= 1) weights[k] = sqrt(1. / e);
        }
        break;
    }
    case ORA_PSEUDO_HUBER: {
        const double tun = sigma;
        for (long k = 0; k < m; k++) weights[k] = 1.0 / sqrt(sqrt(1.0 + e2v[k] / (tun * tun)));
        break;
    }
    case ORA_ANDREWS: {
        const double tun = 1.339 * sigma;
        for (long k = 0; k < m; k++) {
            const double e = sqrt(e2v[k]) / tun;
            double w = sqrt(sin(e) / e);
            if (e >= ORA_PI)
                w = 0;
            else if (e < .0001)
                w = 1;
            if (w < 0.0001) w = 0.0001;
            weights[k] = w;
        }
        break;
    }
    case ORA_BISQUARE: {
        const double tun = 4.685 * sigma;
        for (long k = 0; k < m; k++) {
            double w = 1.0 - e2v[k] / (tun * tun);
            if (w < 0.0001) w = 0.0001;
            weights[k] = w;
        }
        break;
    }
    case ORA_CAUCHY: {
        const double tun = 2.385 * sigma;
        for (long k = 0; k < m; k++) weights[k] = 1.0 / sqrt(1.0 + e2v[k] / (tun * tun));
        break;
    }
    case ORA_FAIR: {
        const double tun = 1.400 * sigma;
        for (long k = 0; k < m; k++) weights[k] = 1.0 / sqrt(1.0 + sqrt(e2v[k]) / tun);
        break;
    }
    case ORA_LOGISTIC: {
        const double tun = 1.205 * sigma;
        for (long k = 0; k < m; k++) {
            const double e = sqrt(e2v[k]) / tun;
            double w = sqrt(tanh(e) / e);
            if (e < 0.0001) w = 1;
            weights[k] = w;
        }
        break;
    }
    case ORA_TALWAR: {
        const double tun = 2.795 * sigma;
        for (long k = 0; k < m; k++) weights[k] = (e2v[k] < tun * tun) ? 1.0001 : 0;
        break;
    }
    case ORA_WELSCH: {
        const double tun = 2.985 * sigma;
        for (long k = 0; k < m; k++) {
            double w = exp(-.5 * e2v[k] / (tun * tun));
            if (w < 0.0001) w = 0.0001;
            weights[k] = w;
        }
        break;
    }
    default:
        return ORA_ERR_UNKNOWN_COST;
    }
    return ORA_OK;
}

/* solve (A'D^2A) X = A'D^2 w for the 3 coordinate columns; S must hold the factor-ready
 * pattern. X: n_u x 3 col-major (ld = n_u). */
static int weighted_ls(lap_sys *S, long m, int f, const int *I, const double *weights,
                       const double *w, long ldw, double *X, double *s_tmp, double *y_tmp,
                       double *b_tmp) {
    const long nu = S->nu;
    for (long k = 0; k < m; k++) s_tmp[k] = weights[k] * weights[k];
    lap_fill_AtSA(S, m, f, I, s_tmp);
    lap_factor(S);
    if (S->pcg) { /* the three columns share the passes over the matrix */
        double *B = (double *)malloc(sizeof(double) * (size_t)(3 * nu > 0 ? 3 * nu : 1));
        if (!B) return ORA_ERR_NOMEM;
        for (int c = 0; c < 3; c++) {
            for (long k = 0; k < m; k++) y_tmp[k] = s_tmp[k] * w[c * ldw + k];
            At_mul(m, nu, f, I, y_tmp, B + c * nu);
        }
        int rc = ora_pcg_solve_multi(S->pcg, S->Ap, S->Ai, S->Ax, 3, B, nu, X, nu);
        free(B);
        g_pcg_solves += 3;
        g_pcg_iters += S->pcg->iters_last;
        if (S->pcg->relres_last > g_pcg_worst_relres) g_pcg_worst_relres = S->pcg->relres_last;
        (void)b_tmp;
        return rc ? ORA_ERR_SOLVER : ORA_OK;
    }
    for (int c = 0; c < 3; c++) {
        for (long k = 0; k < m; k++) y_tmp[k] = s_tmp[k] * w[c * ldw + k];
        At_mul(m, nu, f, I, y_tmp, b_tmp);
        int rc = lap_solve(S, b_tmp, X + c * nu);
        if (rc != ORA_OK) return rc;
    }
    return ORA_OK;
}

int ora_ls_solve(long m, long n_total, int f, const int *I, const double *weights,
                 const double *w, long ldw, double *X) {
    lap_sys *S = lap_build(m, n_total, f, I);
    if (!S) return ORA_ERR_NOMEM;
    const long nu = n_total - f;
    double *s = (double *)malloc(sizeof(double) * (size_t)(m + 1));
    double *y = (double *)malloc(sizeof(double) * (size_t)(m + 1));
    double *b = (double *)malloc(sizeof(double) * (size_t)(nu + 1));
    int rc = weighted_ls(S, m, f, I, weights, w, ldw, X, s, y, b);
    free(s);
    free(y);
    free(b);
    lap_free(S);
    return rc;
}

void ora_normal_matvec(long m, long n_total, int f, const int *I, const double *weights,
                       const double *X, double *Y) {
    const long nu = n_total - f;
    double *t = (double *)malloc(sizeof(double) * (size_t)(m + 1));
    for (int c = 0; c < 3; c++) {
        A_mul(m, f, I, X + c * nu, t);
        for (long k = 0; k < m; k++) t[k] *= weights[k] * weights[k];
        At_mul(m, nu, f, I, t, Y + c * nu);
    }
    free(t);
}

/* shared tail of irls/l1ra: score = mean row norm of W(:,0:3) (:729 / :894), exp map (:731 /
 * :896), Q(f+i) <- Q(f+i) (x) W(i) (:734-737 / :899-902; right-multiply, no renormalisation) */
static double step_and_update(long nu, int f, double *W, double *Q, long ldq) {
    double sum = 0.0;
    for (long i = 0; i < nu; i++) {
        const double x = W[i], y = W[nu + i], z = W[2 * nu + i];
        sum += sqrt(x * x + y * y + z * z);
    }
    const double score = sum / (double)nu;
    ora_exp_map(nu, W, nu);
    for (long i = 0; i < nu; i++) {
        double q[4], wq[4], r[4];
        load_row(Q, ldq, i + f, q);
        load_row(W, nu, i, wq);
        ora_quat_mult(q, wq, r);
        store_row(Q, ldq, i + f, r);
    }
    return score;
}

/* ral/l1_irls.cpp:559-752 */
int ora_irls(long m, long n_total, int f, const int *I, const double *QQ, long ldqq, double *Q,
             long ldq, int cost, double sigma, int max_iters, double change_th, double *weights,
             int *iters, double *runtime, double *score_trace) {
    if (m <= 0 || n_total - f < 1 || f < 0) return ORA_ERR_BAD_ARG;
    if (cost < ORA_L2 || cost > ORA_WELSCH) return ORA_ERR_UNKNOWN_COST;
    clock_t tic = clock(), toc;
    const long nu = n_total - f;
    int rc = ORA_OK;
    lap_sys *S = lap_build(m, n_total, f, I);
    if (!S) return ORA_ERR_NOMEM;
    double *w = (double *)calloc((size_t)(4 * m), sizeof(double));
    double *W = (double *)calloc((size_t)(4 * nu), sizeof(double));
    double *s_tmp = (double *)malloc(sizeof(double) * (size_t)m);
    double *y_tmp = (double *)malloc(sizeof(double) * (size_t)m);
    double *b_tmp = (double *)malloc(sizeof(double) * (size_t)nu);
    double *e2v = (double *)malloc(sizeof(double) * (size_t)m);
    double score = HUGE_VAL; /* DBL_MAX in the reference (:574) */
    *iters = 0;
    for (long k = 0; k < m; k++) weights[k] = 1.0; /* :577 */
    toc = clock();
    while (score > change_th && *iters < max_iters) { /* :590, strict > */
        ora_delta_rel(m, I, QQ, ldqq, Q, ldq, w, m);
        ora_log_map(m, w, m);
        /* :596-612 -- least squares with rows scaled by `weights` */
        rc = weighted_ls(S, m, f, I, weights, w, m, W, s_tmp, y_tmp, b_tmp);
        if (rc != ORA_OK) break; /* the iterative stand-in gave up (never seen; the direct one cannot) */
        /* :614 -- E = A*W3 - w(:,0:3), with make_A's A */
        for (long k = 0; k < m; k++) e2v[k] = 0.0;
        for (int c = 0; c < 3; c++) {
            A_mul(m, f, I, W + c * nu, y_tmp);
            for (long k = 0; k < m; k++) {
                const double e = y_tmp[k] - w[c * m + k];
                e2v[k] += e * e;
            }
        }
        rc = update_weights(cost, sigma, m, e2v, weights);
        if (rc != ORA_OK) break;
        score = step_and_update(nu, f, W, Q, ldq);
        if (score_trace) score_trace[*iters] = score;
        (*iters)++;
        toc = clock();
    }
    *runtime = (double)(toc - tic) / CLOCKS_PER_SEC; /* :751, CPU seconds */
    free(w);
    free(W);
    free(s_tmp);
    free(y_tmp);
    free(b_tmp);
    free(e2v);
    lap_free(S);
    return rc;
}

/* ---------------------------------------------------------------------------------------------
 * L1RA: primal-dual interior point per coordinate
 * ------------------------------------------------------------------------------------------ */

typedef struct {
    long m, nu;
    double *Ax, *u, *fu1, *fu2, *lamu1, *lamu2, *Atv, *rdual_x;
    double *w2, *sig1, *sig2, *sigx, *tm, *w1, *w1p, *dx, *Adx, *du, *dlamu1, *dlamu2, *Atdv;
    double *xp, *up, *Axp, *Atvp, *lamu1p, *lamu2p, *fu1p, *fu2p, *x;
    double *pool;
} pd_work;

static pd_work *pd_alloc(long m, long nu) {
    pd_work *P = (pd_work *)calloc(1, sizeof(pd_work));
    const long nm = 20, nn = 10;
    P->pool = (double *)calloc((size_t)(nm * m + nn * nu + 1), sizeof(double));
    if (!P->pool) {
        free(P);
        return NULL;
    }
    double *p = P->pool;
    P->m = m;
    P->nu = nu;
#define TAKE_M(name) P->name = p, p += m
#define TAKE_N(name) P->name = p, p += nu
    TAKE_M(Ax); TAKE_M(u); TAKE_M(fu1); TAKE_M(fu2); TAKE_M(lamu1); TAKE_M(lamu2);
    TAKE_M(w2); TAKE_M(sig1); TAKE_M(sig2); TAKE_M(sigx); TAKE_M(tm); TAKE_M(Adx); TAKE_M(du);
    TAKE_M(dlamu1); TAKE_M(dlamu2); TAKE_M(up); TAKE_M(Axp); TAKE_M(lamu1p); TAKE_M(lamu2p);
    TAKE_M(fu1p);
    TAKE_N(Atv); TAKE_N(w1); TAKE_N(w1p); TAKE_N(dx); TAKE_N(Atdv); TAKE_N(xp); TAKE_N(Atvp);
    TAKE_N(x); TAKE_N(rdual_x);
#undef TAKE_M
#undef TAKE_N
    P->fu2p = (double *)calloc((size_t)m + 1, sizeof(double));
    return P;
}
static void pd_free(pd_work *P) {
    if (!P) return;
    free(P->fu2p);
    free(P->pool);
    free(P);
}

/* ral/l1_irls.cpp:228-468 with x0 = 0. Constants :231-238. */
static int l1decode_pd_core(lap_sys *S, pd_work *P, long m, int f, const int *I, const double *y,
                            int pdmaxiter, double *xout, int *stuck) {
    const double PDTOL = 1e-3, alpha = 0.01, beta = 0.5, mu = 10;
    const long nu = S->nu;
    if (stuck) *stuck = 0;
    for (long i = 0; i < nu; i++) P->x[i] = 0.0; /* x0 = 0 (:889-892 pass zeroed columns) */
    A_mul(m, f, I, P->x, P->Ax);
    /* :248-253 */
    double maxabs = -HUGE_VAL;
    for (long k = 0; k < m; k++) {
        const double a = fabs(y[k] - P->Ax[k]);
        P->u[k] = a * 0.95;
        if (a > maxabs) maxabs = a;
    }
    for (long k = 0; k < m; k++) P->u[k] += maxabs * 0.10;
    /* :255-259 */
    for (long k = 0; k < m; k++) {
        P->fu1[k] = P->Ax[k] - y[k] - P->u[k];
        P->fu2[k] = -P->Ax[k] + y[k] - P->u[k];
        P->lamu1[k] = -(1.0 / P->fu1[k]);
        P->lamu2[k] = -(1.0 / P->fu2[k]);
        P->tm[k] = P->lamu1[k] - P->lamu2[k];
    }
    At_mul(m, nu, f, I, P->tm, P->Atv); /* :262 */
    double dot = 0.0, dot2 = 0.0;
    for (long k = 0; k < m; k++) {
        dot += P->fu1[k] * P->lamu1[k];
        dot2 += P->fu2[k] * P->lamu2[k];
    }
    double sdg = -(dot + dot2);        /* :264 */
    double tau = mu * 2 * (double)m / sdg; /* :265 */
    /* :267-281 -- resnorm = || [rdual; rcent] ||, rdual = [Atv; 1 - lamu1 - lamu2] */
    double acc = 0.0;
    for (long i = 0; i < nu; i++) {
        P->rdual_x[i] = P->Atv[i];
        acc += P->Atv[i] * P->Atv[i];
    }
    for (long k = 0; k < m; k++) {
        const double rd = 1.0 - P->lamu1[k] - P->lamu2[k];
        acc += rd * rd;
    }
    for (long k = 0; k < m; k++) {
        const double rc1 = -P->lamu1[k] * P->fu1[k] - (1.0 / tau);
        const double rc2 = -P->lamu2[k] * P->fu2[k] - (1.0 / tau);
        acc += rc1 * rc1 + rc2 * rc2;
    }
    double resnorm = sqrt(acc);

    int pditer = 0;
    int done = (sdg < PDTOL) || (pditer >= pdmaxiter); /* :284 */
    int have_xp = 0;
    while (!done) {
        pditer++;
        const double itau = 1.0 / tau;
        for (long k = 0; k < m; k++) { /* :292-305 */
            const double if1 = 1.0 / P->fu1[k], if2 = 1.0 / P->fu2[k];
            P->w2[k] = -1 - itau * (if1 + if2);
            const double a = P->lamu1[k] / P->fu1[k], b = P->lamu2[k] / P->fu2[k];
            P->sig1[k] = -a - b;
            P->sig2[k] = a - b;
            P->sigx[k] = P->sig1[k] - (P->sig2[k] * P->sig2[k]) / P->sig1[k];
            P->tm[k] = -if1 + if2;
        }
        At_mul(m, nu, f, I, P->tm, P->w1);
        for (long i = 0; i < nu; i++) P->w1[i] = -itau * P->w1[i]; /* :302 */
        for (long k = 0; k < m; k++) P->tm[k] = (P->sig2[k] / P->sig1[k]) * P->w2[k];
        At_mul(m, nu, f, I, P->tm, P->w1p);
        for (long i = 0; i < nu; i++) P->w1p[i] = P->w1[i] - P->w1p[i]; /* :306 */
        /* :308-319 -- H11p = AtA*sigx reshaped, solved by UMFPACK in the reference */
        lap_fill_AtA_times(S, m, f, I, P->sigx);
        long ndead = lap_factor(S);
        if (ndead > 0) return ORA_ERR_SOLVER;
        if (lap_solve(S, P->w1p, P->dx) != ORA_OK) return ORA_ERR_SOLVER;
        for (long i = 0; i < nu; i++)
            if (!isfinite(P->dx[i])) return ORA_ERR_SOLVER;
        A_mul(m, f, I, P->dx, P->Adx); /* :324 */
        for (long k = 0; k < m; k++) { /* :327-339 */
            P->du[k] = (P->w2[k] - P->sig2[k] * P->Adx[k]) / P->sig1[k];
            double d1 = -P->lamu1[k] / P->fu1[k];
            d1 *= (P->Adx[k] - P->du[k]);
            d1 -= P->lamu1[k];
            d1 -= itau * (1.0 / P->fu1[k]);
            P->dlamu1[k] = d1;
            double d2 = P->lamu2[k] / P->fu2[k];
            d2 *= (P->Adx[k] + P->du[k]);
            d2 -= P->lamu2[k];
            d2 -= itau * (1.0 / P->fu2[k]);
            P->dlamu2[k] = d2;
            P->tm[k] = d1 - d2;
        }
        At_mul(m, nu, f, I, P->tm, P->Atdv); /* :342 */
        /* :347-381 -- largest feasible step */
        double s = 1;
        for (long k = 0; k < m; k++) {
            if (P->dlamu1[k] < 0) s = fmin(s, -P->lamu1[k] / P->dlamu1[k]);
            if (P->dlamu2[k] < 0) s = fmin(s, -P->lamu2[k] / P->dlamu2[k]);
        }
        for (long k = 0; k < m; k++) {
            const double a = P->Adx[k] - P->du[k];
            if (a > 0) s = fmin(s, -P->fu1[k] / a);
            const double b = -P->Adx[k] - P->du[k];
            if (b > 0) s = fmin(s, -P->fu2[k] / b);
        }
        s *= 0.99;
        /* :384-429 -- backtracking */
        int suffdec = 0, backiter = 0;
        double rdp2 = 0.0;
        while (!suffdec) {
            for (long i = 0; i < nu; i++) {
                P->xp[i] = P->x[i] + s * P->dx[i];
                P->Atvp[i] = P->Atv[i] + s * P->Atdv[i];
            }
            double rd = 0.0, rcp = 0.0;
            for (long i = 0; i < nu; i++) rd += P->Atvp[i] * P->Atvp[i];
            for (long k = 0; k < m; k++) {
                P->up[k] = P->u[k] + s * P->du[k];
                P->Axp[k] = P->Ax[k] + s * P->Adx[k];
                P->lamu1p[k] = P->lamu1[k] + s * P->dlamu1[k];
                P->lamu2p[k] = P->lamu2[k] + s * P->dlamu2[k];
                P->fu1p[k] = P->Axp[k] - y[k] - P->up[k];
                P->fu2p[k] = -P->Axp[k] + y[k] - P->up[k];
                const double r = 1.0 + (-P->lamu1p[k] - P->lamu2p[k]);
                rd += r * r;
                const double c1 = -P->lamu1p[k] * P->fu1p[k] - 1.0 / tau;
                const double c2 = -P->lamu2p[k] * P->fu2p[k] - 1.0 / tau;
                rcp += c1 * c1 + c2 * c2;
            }
            rdp2 = rd;
            suffdec = sqrt(rd + rcp) <= (1 - alpha * s) * resnorm; /* :419 */
            s *= beta;
            backiter++;
            if (backiter > 32) { /* :423-428 -- returns the PREVIOUS iterate x */
                if (stuck) *stuck = 1;
                memcpy(xout, P->x, sizeof(double) * (size_t)nu);
                return ORA_OK;
            }
        }
        /* :432-442 */
        memcpy(P->x, P->xp, sizeof(double) * (size_t)nu);
        memcpy(P->Atv, P->Atvp, sizeof(double) * (size_t)nu);
        memcpy(P->u, P->up, sizeof(double) * (size_t)m);
        memcpy(P->Ax, P->Axp, sizeof(double) * (size_t)m);
        memcpy(P->lamu1, P->lamu1p, sizeof(double) * (size_t)m);
        memcpy(P->lamu2, P->lamu2p, sizeof(double) * (size_t)m);
        memcpy(P->fu1, P->fu1p, sizeof(double) * (size_t)m);
        memcpy(P->fu2, P->fu2p, sizeof(double) * (size_t)m);
        have_xp = 1;
        /* :446-458 */
        dot = 0.0;
        dot2 = 0.0;
        for (long k = 0; k < m; k++) {
            dot += P->fu1[k] * P->lamu1[k];
            dot2 += P->fu2[k] * P->lamu2[k];
        }
        sdg = -(dot + dot2);
        tau = mu * 2 * (double)m / sdg;
        double rc2 = 0.0;
        for (long k = 0; k < m; k++) {
            const double c1 = -P->lamu1[k] * P->fu1[k] - (1.0 / tau);
            const double c2 = -P->lamu2[k] * P->fu2[k] - (1.0 / tau);
            rc2 += c1 * c1 + c2 * c2;
        }
        resnorm = sqrt(rdp2 + rc2);
        done = (sdg < PDTOL) || (pditer >= pdmaxiter); /* :460 */
    }
    /* :467 returns xp; if the loop never ran the reference returns an unsized vector (UB) --
     * cannot happen for m >= 1 because sdg = 2m at entry. */
    (void)have_xp;
    memcpy(xout, P->x, sizeof(double) * (size_t)nu);
    return ORA_OK;
}

int ora_l1decode_pd(long m, long n_total, int f, const int *I, const double *y, int pdmaxiter,
                    double *x, int *stuck) {
    lap_sys *S = lap_build(m, n_total, f, I);
    if (!S) return ORA_ERR_NOMEM;
    pd_work *P = pd_alloc(m, n_total - f);
    if (!P) {
        lap_free(S);
        return ORA_ERR_NOMEM;
    }
    int rc = l1decode_pd_core(S, P, m, f, I, y, pdmaxiter, x, stuck);
    pd_free(P);
    lap_free(S);
    return rc;
}

/* ral/l1_irls.cpp:851-912 */
int ora_l1ra(long m, long n_total, int f, const int *I, const double *QQ, long ldqq, double *Q,
             long ldq, int max_iters, double change_th, int *iters, double *runtime,
             double *score_trace) {
    if (m <= 0 || n_total - f < 1 || f < 0) return ORA_ERR_BAD_ARG;
    clock_t tic = clock(), toc;
    const long nu = n_total - f;
    int rc = ORA_OK;
    lap_sys *S = lap_build(m, n_total, f, I);
    if (!S) return ORA_ERR_NOMEM;
    pd_work *P = pd_alloc(m, nu);
    double *w = (double *)calloc((size_t)(4 * m), sizeof(double));
    double *W = (double *)calloc((size_t)(4 * nu), sizeof(double));
    double score = HUGE_VAL;
    int l1_step = 2; /* :868 */
    *iters = 0;
    toc = clock();
    while (((score >= change_th) || (l1_step < 2)) && (*iters < max_iters)) { /* :877, >= */
        if (score < change_th) { /* :879-883, unreachable given the guard above; kept literal */
            l1_step *= 4;
            change_th /= 100.0;
        }
        ora_delta_rel(m, I, QQ, ldqq, Q, ldq, w, m);
        ora_log_map(m, w, m);
        for (int c = 0; c < 3; c++) { /* :889-892 */
            rc = l1decode_pd_core(S, P, m, f, I, w + c * m, l1_step, W + c * nu, NULL);
            if (rc != ORA_OK) break;
        }
        if (rc != ORA_OK) break;
        score = step_and_update(nu, f, W, Q, ldq); /* :894-902 */
        if (score_trace) score_trace[*iters] = score;
        (*iters)++;
        toc = clock();
    }
    *runtime = (double)(toc - tic) / CLOCKS_PER_SEC;
    free(w);
    free(W);
    pd_free(P);
    lap_free(S);
    return rc;
}

/* ---------------------------------------------------------------------------------------------
 * init_mst
 * ------------------------------------------------------------------------------------------ */

/* ral/l1_irls.cpp:915-979 -- NOT a minimum spanning tree: repeated in-order sweeps over the
 * edge list starting from flags[0] = true. Forward: Q[e2] = QQ_k (x) Q[e1] (:941); backward:
 * Q[e1] = QQ_k^{-1} (x) Q[e2] with the inverse again formed by negating w (:956-958). Rows
 * below f are flagged but never written (:939,954). A sweep that adds nothing while vertices
 * remain unvisited is the reference's exit(-1) (:970-977). */
int ora_init_mst(long n, long m, double *Q, long ldq, const double *QQ, long ldqq, const int *I,
                 int f) {
    if (f <= 0 || n <= 0) return ORA_ERR_BAD_ARG; /* assert(f>0) :917 */
    char *flags = (char *)calloc((size_t)n, 1);
    if (!flags) return ORA_ERR_NOMEM;
    flags[0] = 1;
    long count = 1;
    int rc = ORA_OK;
    while (count < n) {
        int span_flag = 0;
        for (long k = 0; k < m; k++) {
            const long e1 = I[2 * k], e2 = I[2 * k + 1];
            if (flags[e1] && !flags[e2]) {
                if (e2 >= f) {
                    double a[4], b[4], r[4];
                    load_row(QQ, ldqq, k, a);
                    load_row(Q, ldq, e1, b);
                    ora_quat_mult(a, b, r);
                    store_row(Q, ldq, e2, r);
                }
                count++;
                flags[e2] = 1;
                span_flag = 1;
            }
            if (!flags[e1] && flags[e2]) {
                if (e1 >= f) {
                    double a[4], b[4], r[4];
                    load_row(QQ, ldqq, k, a);
                    a[3] = -a[3];
                    load_row(Q, ldq, e2, b);
                    ora_quat_mult(a, b, r);
                    store_row(Q, ldq, e1, r);
                }
                count++;
                flags[e1] = 1;
                span_flag = 1;
            }
        }
        if (!span_flag && count < n) {
            rc = ORA_ERR_NOT_SPANNING;
            break;
        }
    }
    free(flags);
    return rc;
}

/* ---------------------------------------------------------------------------------------------
 * rotation-matrix side of the caller (src/ViewGraph.cpp)
 * ------------------------------------------------------------------------------------------ */

/* src/ViewGraph.cpp:1175-1203. R row-major; q = [x y z w]. */
void ora_rmat2quat(const double R[9], double q[4]) {
#define RM(r, c) R[3 * (r) + (c)]
    const double trace = RM(0, 0) + RM(1, 1) + RM(2, 2);
    if (trace > 0.0) {
        double s = sqrt(trace + 1.0);
        q[3] = s * 0.5;
        s = 0.5 / s;
        q[0] = (RM(2, 1) - RM(1, 2)) * s;
        q[1] = (RM(0, 2) - RM(2, 0)) * s;
        q[2] = (RM(1, 0) - RM(0, 1)) * s;
    } else {
        const int i = RM(0, 0) < RM(1, 1) ? (RM(1, 1) < RM(2, 2) ? 2 : 1)
                                          : (RM(0, 0) < RM(2, 2) ? 2 : 0);
        const int j = (i + 1) % 3, k = (i + 2) % 3;
        double s = sqrt(RM(i, i) - RM(j, j) - RM(k, k) + 1.0);
        q[i] = s * 0.5;
        s = 0.5 / s;
        q[3] = (RM(k, j) - RM(j, k)) * s;
        q[j] = (RM(j, i) + RM(i, j)) * s;
        q[k] = (RM(k, i) + RM(i, k)) * s;
    }
#undef RM
}

/* src/ViewGraph.cpp:1426-1433 -- q.normalized().toRotationMatrix(), written back row-major */
void ora_quat2rmat(const double qin[4], double R[9]) {
    double x = qin[0], y = qin[1], z = qin[2], w = qin[3];
    const double n2 = x * x + y * y + z * z + w * w;
    if (n2 > 0.0) {
        const double nn = sqrt(n2);
        x /= nn;
        y /= nn;
        z /= nn;
        w /= nn;
    }
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz);
    R[1] = txy - twz;
    R[2] = txz + twy;
    R[3] = txy + twz;
    R[4] = 1 - (txx + tzz);
    R[5] = tyz - twx;
    R[6] = txz - twy;
    R[7] = tyz + twx;
    R[8] = 1 - (txx + tyy);
}

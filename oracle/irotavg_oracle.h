/*
 * irotavg_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, NOT THE PRODUCT).
 *
 * A plain-C, single-threaded, fp64 restatement of the rotation-averaging path of
 * ajparra/iRotAvg (ral/l1_irls.cpp, ral/test.cpp, src/ViewGraph.cpp::rotAvg), written from the
 * algorithm's semantics (including its quirks), with our own sparse Cholesky in place of
 * SuiteSparse (SPQR at ral/l1_irls.cpp:550, UMFPACK at ral/l1_irls.cpp:147-169), which is
 * not vendored by the reference and not installed here.
 *
 * PARITY UNPINNED: the reference cannot be compiled in this image (no Eigen, no SuiteSparse:
 * ral/l1_irls.hpp:30,32 fail to include) and ships no expected outputs for its one fixture
 * (ral/data/ravg_input.txt), no unit tests and no known-answer vectors. This oracle is pinned
 * only by (1) analytic known answers (tests/test_oracle_kat.py), (2) an independent
 * NumPy/SciPy twin (oracle/np_twin.py, SuperLU solves), (3) sanity values recorded in
 * SURVEY.md 8(c) for the fixture and (4) a check of its normal-equation solves against the
 * reference's FORMULATION -- the least-squares form by dense Householder QR / minimum-norm lstsq,
 * incl. a characterisation of the rank-deficient cases (tests/test_oracle_lsform.py).
 * tools/ref_golden/ is the recipe that would pin it (build the unmodified reference where Eigen and
 * SuiteSparse exist, commit its outputs, tests/test_ref_golden.py consumes them). See DESIGN.md
 * "Oracle".
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this library.
 *
 * Layout conventions follow the reference API (ral/l1_irls.hpp:40-51,81-107):
 *   Mat  = column-major fp64 with leading dimension ld (Eigen MatrixXd), quaternion columns
 *          are [x, y, z, w] (w LAST);
 *   I_t  = m pairs of int32 (first=i, second=j), 0-based;
 *   first f rows of Q are fixed.
 */
#ifndef IROTAVG_ORACLE_H
#define IROTAVG_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* ral/l1_irls.hpp:56-57 -- integer values are ABI */
enum ora_cost {
    ORA_L2 = 0, ORA_L1, ORA_L15, ORA_L05, ORA_GEMAN_MCCLURE, ORA_HUBER, ORA_PSEUDO_HUBER,
    ORA_ANDREWS, ORA_BISQUARE, ORA_CAUCHY, ORA_FAIR, ORA_LOGISTIC, ORA_TALWAR, ORA_WELSCH
};

/* error codes (the reference calls exit(-1) in these situations) */
#define ORA_OK 0
#define ORA_ERR_BAD_ARG (-1)
#define ORA_ERR_NOT_SPANNING (-2)  /* ral/l1_irls.cpp:970-977 */
#define ORA_ERR_SOLVER (-3)        /* ral/l1_irls.cpp:149-177 */
#define ORA_ERR_UNKNOWN_COST (-4)  /* ral/l1_irls.cpp:723-726 */
#define ORA_ERR_NOMEM (-5)

#define ORA_EPS 2.2204e-16 /* ral/l1_irls.hpp:40 */

/* ral/l1_irls.cpp:99-105 : out = q1 (x) q2, [x y z w] */
void ora_quat_mult(const double q1[4], const double q2[4], double out[4]);

/* ral/l1_irls.cpp:109-127 : out(k,:) = Qinv(j) (x) (QQ(k) (x) Q(i)), Qinv = Q with w negated */
void ora_delta_rel(long m, const int *I, const double *QQ, long ldqq, const double *Q, long ldq,
                   double *out, long ldo);

/* ral/l1_irls.cpp:498-532 : in-place log map of m rows [x y z w] -> [r1 r2 r3 theta] */
void ora_log_map(long m, double *w, long ld);

/* ral/l1_irls.cpp:471-492 : in-place exp map of n rows [r1 r2 r3 *] -> unit quaternion [x y z w] */
void ora_exp_map(long n, double *W, long ld);

/* ral/l1_irls.cpp:755-780 : incidence matrix in CSC (Eigen column-major, 64-bit indices,
 * rows sorted within a column). colptr has n-f+1 entries; rowidx/vals have at most 2m entries.
 * Returns nnz (>=0) or a negative error. Reproduces the edge-drop quirk of :770-771. */
long ora_make_A(int n, int f, long m, const int *I, long *colptr, long *rowidx, double *vals);

/* ral/l1_irls.cpp:915-979 : sweep propagation from vertex 0; rows < f never overwritten */
int ora_init_mst(long n, long m, double *Q, long ldq, const double *QQ, long ldqq, const int *I,
                 int f);

/* ral/l1_irls.cpp:982-991 */
void ora_quat_normalised(long n, double *Q, long ldq, int f);

/* ral/l1_irls.cpp:559-752. weights must hold m doubles (overwritten with ones first, :577).
 * score_trace (optional, may be NULL) receives the per-iteration `score`, up to max_iters. */
int ora_irls(long m, long n_total, int f, const int *I, const double *QQ, long ldqq, double *Q,
             long ldq, int cost, double sigma, int max_iters, double change_th, double *weights,
             int *iters, double *runtime, double *score_trace);

/* ral/l1_irls.cpp:851-912 */
int ora_l1ra(long m, long n_total, int f, const int *I, const double *QQ, long ldqq, double *Q,
             long ldq, int max_iters, double change_th, int *iters, double *runtime,
             double *score_trace);

/* ral/l1_irls.cpp:228-468 : one coordinate of the primal-dual LP, x0 = 0, A from make_A, H from
 * make_AtA (:811-848). x has n_total-f entries. Returns ORA_OK or an error;
 * *stuck (optional) set to 1 if "Stuck backtracking" fired (:423-428). */
int ora_l1decode_pd(long m, long n_total, int f, const int *I, const double *y, int pdmaxiter,
                    double *x, int *stuck);

/* one weighted least-squares solve as in ral/l1_irls.cpp:596-612 (3 right-hand sides):
 * X (n_u x 3, col-major ld n_u) = argmin || D A X - D w ||, D = diag(weights). */
int ora_ls_solve(long m, long n_total, int f, const int *I, const double *weights,
                 const double *w, long ldw, double *X);

/* y = L x for the IRLS normal matrix L = A' D^2 A (3 columns, col-major, ld n_u) -- used by the
 * full-size residual checks in tests */
void ora_normal_matvec(long m, long n_total, int f, const int *I, const double *weights,
                       const double *X, double *Y);

/* Which stand-in for SuiteSparse solves the linear systems (both solve the same normal equations):
 * 0 = chosen by the graph (sparse Cholesky, sparse_chol.c, unless there are more than 5000 unknowns
 * and the envelope work of the natural ordering exceeds 1e10 -- view sequences with thousands of
 * loop closures -- then Gauss-Seidel-preconditioned CG to a true relative residual of 1e-13,
 * sparse_pcg.c),
 * 1 = Cholesky, 2 = PCG. Environment ORA_SOLVER=chol|pcg does the same when the mode is 0. */
void ora_set_solver(int mode);
void ora_solver_stats(long *chol_solves, long *pcg_solves, long *pcg_iters, double *pcg_worst_relres,
                      int reset);

/* src/ViewGraph.cpp:1175-1203 : row-major 3x3 -> [x y z w] */
void ora_rmat2quat(const double R[9], double q[4]);
/* src/ViewGraph.cpp:1426-1433 : normalised quaternion [x y z w] -> row-major 3x3 */
void ora_quat2rmat(const double q[4], double R[9]);

#ifdef __cplusplus
}
#endif
#endif

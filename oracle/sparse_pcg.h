/* sparse_pcg.h -- ORACLE-ONLY iterative solve (see sparse_pcg.c). Test infrastructure. */
#ifndef ORA_SPARSE_PCG_H
#define ORA_SPARSE_PCG_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ora_pcg {
    long n;
    double *diag;  /* summed diagonal of the current matrix */
    char *dead;    /* diagonal not above ORA_DEAD_TOL x the largest one: unknown solves to 0 */
    long ndead;
    double *r, *z, *p, *q, *xi; /* work, 3 interleaved columns each */
    long iters_last, iters_total, solves; /* statistics for the tests */
    double relres_last;                    /* TRUE relative residual ||b - Hx|| / ||b|| of the last solve */
} ora_pcg;

ora_pcg *ora_pcg_alloc(long n);
/* Ap/Ai/Ax: CSC of the full symmetric matrix (both triangles; duplicate entries are summed,
 * rows unsorted). Returns the number of dead unknowns. */
long ora_pcg_setup(ora_pcg *c, const long *Ap, const long *Ai, const double *Ax);
/* conjugate gradients preconditioned by symmetric Gauss-Seidel, to a TRUE relative residual of
 * ORA_PCG_RTOL (1e-13) or the attainable accuracy; x starts at 0. Returns 0, or -1 when the
 * iteration cap was reached above 1e-9. */
int ora_pcg_solve(ora_pcg *c, const long *Ap, const long *Ai, const double *Ax, const double *b,
                  double *x);
/* nr <= 3 right-hand sides (columns of b, leading dimension ldb) in shared passes over the matrix,
 * every column with its own recurrence and its own stopping test */
int ora_pcg_solve_multi(ora_pcg *c, const long *Ap, const long *Ai, const double *Ax, int nr,
                        const double *b, long ldb, double *x, long ldx);
void ora_pcg_free(ora_pcg *c);

/* upper bound on the Cholesky work of the natural ordering: sum over the columns of the squared
 * envelope height, O(nnz) */
double ora_envelope(long n, const long *Ap, const long *Ai);

#ifdef __cplusplus
}
#endif
#endif

/* sparse_chol.h -- ORACLE-ONLY sparse Cholesky (see sparse_chol.c). Test infrastructure. */
#ifndef ORA_SPARSE_CHOL_H
#define ORA_SPARSE_CHOL_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ora_chol {
    long n;
    long *perm, *iperm; /* perm[new] = old */
    long *parent;       /* elimination tree */
    long *Cp, *Ci;      /* upper triangle of P A P' (CSC) */
    double *Cx;
    long *Lp, *Li; /* L in CSC, diagonal first in each column */
    double *Lx;
    long lnz;
    char *dead;
    long ndead;
} ora_chol;

/* Ap/Ai: CSC pattern of the full symmetric matrix (both triangles, duplicates allowed) */
ora_chol *ora_chol_analyze(long n, const long *Ap, const long *Ai);
/* numeric factorisation with the same pattern; returns the number of dead (non-positive) pivots */
long ora_chol_factor(ora_chol *c, const long *Ap, const long *Ai, const double *Ax);
void ora_chol_solve(const ora_chol *c, const double *b, double *x);
void ora_chol_free(ora_chol *c);

#ifdef __cplusplus
}
#endif
#endif

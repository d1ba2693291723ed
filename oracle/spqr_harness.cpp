/*
 * spqr_harness.cpp -- TEST / MEASUREMENT INFRASTRUCTURE (oracle/), never part of the product.
 *
 * when the GPU box has SuiteSparse, bench.py's cpu_baseline leg times the
 * reference's own LIBRARY CALL for one IRLS iteration -- the least-squares solve X = A \ B by
 * SuiteSparseQR<double>(&A, &B, cc) with default ordering and tolerance and a fresh cholmod_common
 * per call, exactly what ls_solve does at ral/l1_irls.cpp:536-556 -- on the same graph. This image
 * has no SuiteSparse (bench.py probes by compiling this file and reports what it found), so this
 * file has never been compiled here; it is written against the documented CHOLMOD / SPQR C++ API.
 *
 * input : text file "m n f" then m lines "i j rx ry rz" (0-based views, i < j; rotation-vector
 *         residual rows = the right-hand side of the first IRLS iteration, unit weights)
 * output: "<seconds> <solves>" on stdout. Repeats the solve until `budget` seconds have passed.
 */
#include <SuiteSparseQR.hpp>
#include <cholmod.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const double budget = argc > 2 ? std::atof(argv[2]) : 10.0;
    std::FILE *fh = std::fopen(argv[1], "r");
    if (!fh) return 2;
    long m, n, f;
    if (std::fscanf(fh, "%ld %ld %ld", &m, &n, &f) != 3) return 2;
    std::vector<long> ei(m), ej(m);
    std::vector<double> r(3 * (size_t)m);
    for (long k = 0; k < m; k++)
        if (std::fscanf(fh, "%ld %ld %lf %lf %lf", &ei[k], &ej[k], &r[k], &r[m + k], &r[2 * m + k]) != 5) return 2;
    std::fclose(fh);
    const long nu = n - f;
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    long solves = 0;
    double sec = 0.0;
    do {
        cholmod_common cc;  /* a fresh common per solve, as the reference (ral/l1_irls.cpp:541-546) */
        cholmod_l_start(&cc);
        /* A = make_A(n, f, I) (ral/l1_irls.cpp:755-780): +1 at column j-f, -1 at column i-f, the whole
         * row dropped when j < f; unit weights: D A = A */
        cholmod_triplet *T = cholmod_l_allocate_triplet(m, nu, 2 * m, 0, CHOLMOD_REAL, &cc);
        long *Ti = (long *)T->i, *Tj = (long *)T->j;
        double *Tx = (double *)T->x;
        size_t nz = 0;
        for (long k = 0; k < m; k++) {
            if (ej[k] - f < 0) continue;
            Ti[nz] = k; Tj[nz] = ej[k] - f; Tx[nz++] = 1.0;
            if (ei[k] - f >= 0) { Ti[nz] = k; Tj[nz] = ei[k] - f; Tx[nz++] = -1.0; }
        }
        T->nnz = nz;
        cholmod_sparse *A = cholmod_l_triplet_to_sparse(T, nz, &cc);
        cholmod_dense *B = cholmod_l_allocate_dense(m, 3, m, CHOLMOD_REAL, &cc);
        double *Bx = (double *)B->x;
        for (size_t q = 0; q < 3 * (size_t)m; q++) Bx[q] = r[q];
        cholmod_dense *X = SuiteSparseQR<double>(A, B, &cc); /* ral/l1_irls.cpp:550 */
        if (!X) return 3;
        cholmod_l_free_dense(&X, &cc);
        cholmod_l_free_dense(&B, &cc);
        cholmod_l_free_sparse(&A, &cc);
        cholmod_l_free_triplet(&T, &cc);
        cholmod_l_finish(&cc);
        solves++;
        sec = std::chrono::duration<double>(clk::now() - t0).count();
    } while (sec < budget);
    std::printf("%.6f %ld\n", sec, solves);
    return 0;
}

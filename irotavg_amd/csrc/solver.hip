// solver.hip -- kernels and drivers of the IRLS path:
//   K1 edge_residual      delta_rel + log_map            (ral/l1_irls.cpp:109-127, 498-532)
//   K2 update_weights     E = A X - w, robust weights    (ral/l1_irls.cpp:614-727)
//   K3 assemble           weighted Laplacian A'D^2A + rhs (what SPQR factorises, :596-612)
//   K4 spmv / V-cycle     PCG with an aggregation-multigrid preconditioner replacing
//                         SuiteSparseQR (:550) and UMFPACK (:147-169)
//   K6 apply_step         score, exp_map, Q update       (ral/l1_irls.cpp:729-737, 471-492)
#include "graph.hpp"
#include "kernels.hpp"

namespace irh {

#define IRH_PI 3.141592653589793238462643383279502884
#define IRH_EPS 2.2204e-16  // ral/l1_irls.hpp:40

// =============================================================================================
// K1 -- edge residual. Two edges per thread so every SoA stream moves as 16 B per lane.
// Algorithmic traffic: 8 B indices + 32 B QQ + 24 B out per edge, 32 B per view.
// =============================================================================================
__device__ __forceinline__ void edge_log(const double4 qi, double4 qj, const double4 qq,
                                         double &ox, double &oy, double &oz) {
    qj.w = -qj.w;  // the reference's "inverse": only w negated (ral/l1_irls.cpp:114-115)
    const double4 d = qmul(qj, qmul(qq, qi));
    const double s2 = sqrt(d.x * d.x + d.y * d.y + d.z * d.z);
    double th = 2.0 * atan2(s2, d.w);
    if (th < -IRH_PI)  // wrap into [-pi, pi) (ral/l1_irls.cpp:510-517)
        th += 2.0 * IRH_PI;
    else if (th >= IRH_PI)
        th -= 2.0 * IRH_PI;
    const double aux = th / s2;
    ox = d.x * aux;
    oy = d.y * aux;
    oz = d.z * aux;
    if (s2 < IRH_EPS) {  // ral/l1_irls.cpp:527-531
        ox = 0.0;
        oy = 0.0;
        oz = 0.0;
    }
}

__global__ __launch_bounds__(256) void k_edge_residual(long long mpad, const int *__restrict__ ei,
                                                       const int *__restrict__ ej,
                                                       const double *__restrict__ qq,
                                                       const double4 *__restrict__ Q,
                                                       double *__restrict__ er) {
    const long long k = 2ll * ((long long)blockIdx.x * blockDim.x + threadIdx.x);
    if (k >= mpad) return;
    const int2 ii = *reinterpret_cast<const int2 *>(ei + k);
    const int2 jj = *reinterpret_cast<const int2 *>(ej + k);
    const double2 qx = *reinterpret_cast<const double2 *>(qq + k);
    const double2 qy = *reinterpret_cast<const double2 *>(qq + mpad + k);
    const double2 qz = *reinterpret_cast<const double2 *>(qq + 2 * mpad + k);
    const double2 qw = *reinterpret_cast<const double2 *>(qq + 3 * mpad + k);
    const double4 qi0 = Q[ii.x], qj0 = Q[jj.x], qi1 = Q[ii.y], qj1 = Q[jj.y];
    double2 rx, ry, rz;
    edge_log(qi0, qj0, make_double4(qx.x, qy.x, qz.x, qw.x), rx.x, ry.x, rz.x);
    edge_log(qi1, qj1, make_double4(qx.y, qy.y, qz.y, qw.y), rx.y, ry.y, rz.y);
    *reinterpret_cast<double2 *>(er + k) = rx;
    *reinterpret_cast<double2 *>(er + mpad + k) = ry;
    *reinterpret_cast<double2 *>(er + 2 * mpad + k) = rz;
}

void launch_edge_residual(Graph &g) {
    const long long threads = g.mpad / 2;
    const int grid = (int)((threads + 255) / 256);
    hipLaunchKernelGGL(k_edge_residual, dim3(grid), dim3(256), 0, g.stream, (long long)g.mpad,
                       g.ei.p, g.ej.p, g.qq.p, g.Q.p, g.er.p);
}

// =============================================================================================
// K2 -- residual of the linearised system and robust weight update (one edge per thread).
// =============================================================================================
__device__ __forceinline__ double robust_weight(int cost, double sigma, double e2, double prev) {
    switch (cost) {
    case IROTAVG_L2:
        return prev;
    case IROTAVG_L05: {
        double w = 1.0 / pow(e2, 3. / 8.);
        return w > 1e4 ? 1e4 : w;
    }
    case IROTAVG_L1: {
        double w = 1.0 / sqrt(sqrt(e2));
        return w > 1e4 ? 1e4 : w;
    }
    case IROTAVG_L15: {
        double w = 1.0 / sqrt(sqrt(sqrt(e2)));
        return w > 1e4 ? 1e4 : w;
    }
    case IROTAVG_GEMAN_MCCLURE:
        return 1.0 / (e2 + sigma * sigma);
    case IROTAVG_HUBER: {  // weights of inliers keep their previous value (:647-649)
        const double e = sqrt(e2) / (1.345 * sigma);
        return e >= 1 ? sqrt(1. / e) : prev;
    }
    case IROTAVG_PSEUDO_HUBER:
        return 1.0 / sqrt(sqrt(1.0 + e2 / (sigma * sigma)));
    case IROTAVG_ANDREWS: {
        const double e = sqrt(e2) / (1.339 * sigma);
        double w = sqrt(sin(e) / e);
        if (e >= IRH_PI)
            w = 0;
        else if (e < .0001)
            w = 1;
        if (w < 0.0001) w = 0.0001;
        return w;
    }
    case IROTAVG_BISQUARE: {
        const double t = 4.685 * sigma;
        double w = 1.0 - e2 / (t * t);
        return w < 0.0001 ? 0.0001 : w;
    }
    case IROTAVG_CAUCHY: {
        const double t = 2.385 * sigma;
        return 1.0 / sqrt(1.0 + e2 / (t * t));
    }
    case IROTAVG_FAIR:
        return 1.0 / sqrt(1.0 + sqrt(e2) / (1.400 * sigma));
    case IROTAVG_LOGISTIC: {
        const double e = sqrt(e2) / (1.205 * sigma);
        return e < 0.0001 ? 1.0 : sqrt(tanh(e) / e);
    }
    case IROTAVG_TALWAR: {
        const double t = 2.795 * sigma;
        return e2 < t * t ? 1.0001 : 0.0;
    }
    default: {  // IROTAVG_WELSCH
        const double t = 2.985 * sigma;
        double w = exp(-.5 * e2 / (t * t));
        return w < 0.0001 ? 0.0001 : w;
    }
    }
}

__global__ __launch_bounds__(256) void k_update_weights(long long m, long long mpad, int f,
                                                        const int *__restrict__ ei,
                                                        const int *__restrict__ ej,
                                                        const uint8_t *__restrict__ eflag,
                                                        const double *__restrict__ er,
                                                        const double4 *__restrict__ X, int cost,
                                                        double sigma, double *__restrict__ dw) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    const uint8_t fl = eflag[k];
    double e0 = 0.0, e1 = 0.0, e2c = 0.0;
    if (fl & EF_CJ) {
        const double4 xj = X[ej[k] - f];
        e0 += xj.x;
        e1 += xj.y;
        e2c += xj.z;
    }
    if (fl & EF_CI) {
        const double4 xi = X[ei[k] - f];
        e0 -= xi.x;
        e1 -= xi.y;
        e2c -= xi.z;
    }
    e0 -= er[k];
    e1 -= er[mpad + k];
    e2c -= er[2 * mpad + k];
    const double e2 = e0 * e0 + e1 * e1 + e2c * e2c;
    dw[k] = robust_weight(cost, sigma, e2, dw[k]);
}

void launch_update_weights(Graph &g, int cost, double sigma) {
    const int grid = (int)((g.m + 255) / 256);
    hipLaunchKernelGGL(k_update_weights, dim3(grid), dim3(256), 0, g.stream, (long long)g.m,
                       (long long)g.mpad, g.f, g.ei.p, g.ej.p, g.eflag.p, g.er.p, g.X.p, cost,
                       sigma, g.dw.p);
}

// =============================================================================================
// K3 -- level-0 assembly. One G-lane group per free view walks the view's incident-edge slots:
// off-diagonal value -w, diagonal sum, Dirichlet excess and (IRLS) the right-hand side
// b_v = sum_k +-w_k r_k. MODE 0: w = d_k^2 from the IRLS weights, rhs built.
// MODE 1: w = s[k] (sigx of the primal-dual step), no rhs, make_AtA boundary rule.
// =============================================================================================
template <int G, int MODE>
__global__ __launch_bounds__(kBlock) void k_assemble0(
    int n, const int *__restrict__ rowptr, const uint32_t *__restrict__ slot_eid,
    const int *__restrict__ bptr, const uint32_t *__restrict__ beid,
    const uint8_t *__restrict__ bflag, const double *__restrict__ wsrc,
    const double *__restrict__ er, long long mpad, double *__restrict__ val,
    double *__restrict__ excess, double *__restrict__ diag, double *__restrict__ idg,
    double4 *__restrict__ rhs) {
    constexpr int R = kBlock / G;
    const int grp = threadIdx.x / G, l = threadIdx.x % G;
    const int ntiles = (n + R - 1) / R;
    int t0, t1;
    tile_range(ntiles, t0, t1);
    for (int t = t0; t < t1; t++) {
        const int row = t * R + grp;
        double sw = 0.0, ex = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0;
        if (row < n) {
            const int beg = rowptr[row], end = rowptr[row + 1];
            for (int s = beg + l; s < end; s += G) {
                const uint32_t se = slot_eid[s];
                const uint32_t k = se >> 1;
                double w = wsrc[k];
                if (MODE == 0) w = w * w;
                val[s] = -w;
                sw += w;
                if (MODE == 0) {
                    const double sg = (se & 1u) ? w : -w;
                    b0 += sg * er[k];
                    b1 += sg * er[mpad + k];
                    b2 += sg * er[2 * mpad + k];
                }
            }
            const int bb = bptr[row], be = bptr[row + 1];
            for (int s = bb + l; s < be; s += G) {
                const uint8_t fl = bflag[s];
                if (!(fl & (MODE == 0 ? BF_IRLS : BF_L1H))) continue;
                const uint32_t se = beid[s];
                const uint32_t k = se >> 1;
                double w = wsrc[k];
                if (MODE == 0) w = w * w;
                if (MODE == 1 && (fl & BF_NEG)) w = -w;
                ex += w;
                if (MODE == 0) {
                    const double sg = (se & 1u) ? w : -w;
                    b0 += sg * er[k];
                    b1 += sg * er[mpad + k];
                    b2 += sg * er[2 * mpad + k];
                }
            }
        }
        sw = group_sum<G>(sw);
        ex = group_sum<G>(ex);
        if (MODE == 0) {
            b0 = group_sum<G>(b0);
            b1 = group_sum<G>(b1);
            b2 = group_sum<G>(b2);
        }
        if (l == 0 && row < n) {
            const double d = sw + ex;
            excess[row] = ex;
            diag[row] = d;
            idg[row] = d > 0.0 ? 1.0 / d : 0.0;
            if (MODE == 0) rhs[row] = make_double4(b0, b1, b2, 0.0);
        }
    }
}

// coarse off-diagonal values: val_c[c] = sum of the finer slots listed for c
template <int G>
__global__ __launch_bounds__(kBlock) void k_coarse_vals(int nslots, const int *__restrict__ cptr,
                                                        const int *__restrict__ cidx,
                                                        const double *__restrict__ fval,
                                                        double *__restrict__ cval) {
    constexpr int R = kBlock / G;
    const int grp = threadIdx.x / G, l = threadIdx.x % G;
    const int ntiles = (nslots + R - 1) / R;
    int t0, t1;
    tile_range(ntiles, t0, t1);
    for (int t = t0; t < t1; t++) {
        const int c = t * R + grp;
        double s = 0.0;
        if (c < nslots) {
            const int beg = cptr[c], end = cptr[c + 1];
            for (int q = beg + l; q < end; q += G) s += fval[cidx[q]];
        }
        s = group_sum<G>(s);
        if (l == 0 && c < nslots) cval[c] = s;
    }
}

// coarse diagonal: excess_c = sum of the aggregate's excess, diag_c = excess_c - sum(val_c)
__global__ __launch_bounds__(256) void k_coarse_diag(int nc, int nf, int agg,
                                                     const double *__restrict__ fexcess,
                                                     const int *__restrict__ rowptr,
                                                     const double *__restrict__ cval,
                                                     double *__restrict__ cexcess,
                                                     double *__restrict__ cdiag,
                                                     double *__restrict__ cidg) {
    const int I = blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= nc) return;
    double ex = 0.0;
    const int v0 = I * agg, v1 = min(nf, v0 + agg);
    for (int v = v0; v < v1; v++) ex += fexcess[v];
    double sv = 0.0;
    for (int s = rowptr[I]; s < rowptr[I + 1]; s++) sv += cval[s];
    const double d = ex - sv;
    cexcess[I] = ex;
    cdiag[I] = d;
    cidg[I] = d > 0.0 ? 1.0 / d : 0.0;
}

// dense inverse of the coarsest level (n <= 128) by in-place Gauss-Jordan in LDS. The matrix is
// SPD for a connected graph with f >= 1; a non-positive pivot (isolated coarse vertex) is
// replaced by 1 after its row/column were zeroed, i.e. that unknown solves to 0.
__global__ __launch_bounds__(1024) void k_dense_invert(int n, const int *__restrict__ rowptr,
                                                       const int *__restrict__ col,
                                                       const double *__restrict__ val,
                                                       const double *__restrict__ diag,
                                                       double *__restrict__ inv) {
    extern __shared__ double A[];  // n*n + 2n
    double *rowk = A + n * n, *colk = rowk + n;
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int e = tid; e < n * n; e += nt) A[e] = 0.0;
    __syncthreads();
    for (int i = tid; i < n; i += nt) {
        A[i * n + i] = diag[i];
        for (int s = rowptr[i]; s < rowptr[i + 1]; s++) A[i * n + col[s]] += val[s];
    }
    __syncthreads();
    for (int k = 0; k < n; k++) {
        const double piv = A[k * n + k];
        const bool dead = !(piv > 0.0);
        const double ip = dead ? 0.0 : 1.0 / piv;
        for (int j = tid; j < n; j += nt) {
            rowk[j] = (j == k) ? ip : A[k * n + j] * ip;
            colk[j] = (j == k) ? 0.0 : A[j * n + k];
        }
        __syncthreads();
        for (int e = tid; e < n * n; e += nt) {
            const int i = e / n, j = e - i * n;
            if (i == k)
                A[e] = rowk[j];
            else
                A[e] = (j == k ? 0.0 : A[e]) - colk[i] * rowk[j];
        }
        __syncthreads();
    }
    for (int e = tid; e < n * n; e += nt) inv[e] = A[e];
}

__global__ __launch_bounds__(256) void k_dense_solve(int n, const double *__restrict__ inv,
                                                     const double4 *__restrict__ b,
                                                     double4 *__restrict__ x,
                                                     const int *__restrict__ flags) {
    if (flags[FL_DONE]) return;
    extern __shared__ double4 sb[];
    for (int i = threadIdx.x; i < n; i += blockDim.x) sb[i] = b[i];
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        double s0 = 0, s1 = 0, s2 = 0;
        for (int j = 0; j < n; j++) {
            const double a = inv[i * n + j];
            s0 += a * sb[j].x;
            s1 += a * sb[j].y;
            s2 += a * sb[j].z;
        }
        x[i] = make_double4(s0, s1, s2, 0.0);
    }
}

// =============================================================================================
// K4 -- row kernels on a level: y = L x variants. One G-lane group per row.
// =============================================================================================
template <int G>
__device__ __forceinline__ void row_offdiag(const LevelView &L, int row, int l,
                                            const double4 *__restrict__ x, double &s0, double &s1,
                                            double &s2) {
    s0 = s1 = s2 = 0.0;
    if (row < L.n) {
        const int beg = L.rowptr[row], end = L.rowptr[row + 1];
        for (int s = beg + l; s < end; s += G) {
            const int c = L.col[s];
            const double v = L.val[s];
            const double4 xc = x[c];
            s0 += v * xc.x;
            s1 += v * xc.y;
            s2 += v * xc.z;
        }
    }
    s0 = group_sum<G>(s0);
    s1 = group_sum<G>(s1);
    s2 = group_sum<G>(s2);
}

// q = L p, partial dot products p.q
template <int G>
__global__ __launch_bounds__(kBlock) void k_spmv_dot(LevelView L, const double4 *__restrict__ p,
                                                     double4 *__restrict__ q,
                                                     double *__restrict__ part_pq,
                                                     const int *__restrict__ flags) {
    if (flags[FL_DONE]) return;
    constexpr int R = kBlock / G;
    const int grp = threadIdx.x / G, l = threadIdx.x % G;
    const int ntiles = (L.n + R - 1) / R;
    int t0, t1;
    tile_range(ntiles, t0, t1);
    double a0 = 0, a1 = 0, a2 = 0;
    for (int t = t0; t < t1; t++) {
        const int row = t * R + grp;
        double s0, s1, s2;
        row_offdiag<G>(L, row, l, p, s0, s1, s2);
        if (l == 0 && row < L.n) {
            const double4 pr = p[row];
            const double d = L.diag[row];
            s0 += d * pr.x;
            s1 += d * pr.y;
            s2 += d * pr.z;
            q[row] = make_double4(s0, s1, s2, 0.0);
            a0 += pr.x * s0;
            a1 += pr.y * s1;
            a2 += pr.z * s2;
        }
    }
    block_sum3_store(a0, a1, a2, part_pq + 4 * blockIdx.x);
}

// Down-sweep on level l: r = b - L x (x = omega D^-1 b already stored), restricted by summing
// each aggregate of `agg` consecutive rows: bc = P' r, and the coarse pre-smoothed iterate
// xc = omega Dc^-1 bc. On level 0 (CHECK) the prologue turns the ||r||^2 partials of the last
// PCG update into the convergence decision.
template <int G, bool CHECK>
__global__ __launch_bounds__(kBlock) void k_residual_restrict(
    LevelView L, const double4 *__restrict__ b, const double4 *__restrict__ x,
    double4 *__restrict__ bc, double4 *__restrict__ xc, const double *__restrict__ cidg, int nc,
    double omega, const double *__restrict__ part_rr, int nparts, int first, double rtol2,
    double *__restrict__ scal, int *__restrict__ flags) {
    if (flags[FL_DONE]) return;
    if (CHECK) {
        double rr[3];
        load_reduced3(part_rr, nparts, rr);
        double bb[3];
        if (first) {
            bb[0] = rr[0];
            bb[1] = rr[1];
            bb[2] = rr[2];
        } else {
            bb[0] = scal[SC_BB];
            bb[1] = scal[SC_BB + 1];
            bb[2] = scal[SC_BB + 2];
        }
        const bool finite = isfinite(rr[0]) && isfinite(rr[1]) && isfinite(rr[2]);
        const bool conv = rr[0] <= rtol2 * bb[0] && rr[1] <= rtol2 * bb[1] && rr[2] <= rtol2 * bb[2];
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            if (first) {
                scal[SC_BB] = bb[0];
                scal[SC_BB + 1] = bb[1];
                scal[SC_BB + 2] = bb[2];
            }
            for (int c = 0; c < 3; c++) scal[SC_RELRES + c] = bb[c] > 0.0 ? sqrt(rr[c] / bb[c]) : 0.0;
            if (!finite)
                flags[FL_DONE] = 2;
            else if (conv)
                flags[FL_DONE] = 1;
        }
        if (!finite || conv) return;
    }
    constexpr int R = kBlock / G;
    __shared__ double sr[3][R];
    const int grp = threadIdx.x / G, l = threadIdx.x % G;
    const int ntiles = (L.n + R - 1) / R;
    int t0, t1;
    tile_range(ntiles, t0, t1);
    for (int t = t0; t < t1; t++) {
        const int row = t * R + grp;
        double s0, s1, s2;
        row_offdiag<G>(L, row, l, x, s0, s1, s2);
        if (l == 0) {
            double r0 = 0, r1 = 0, r2 = 0;
            if (row < L.n) {
                const double4 xr = x[row], br = b[row];
                const double d = L.diag[row];
                r0 = br.x - (s0 + d * xr.x);
                r1 = br.y - (s1 + d * xr.y);
                r2 = br.z - (s2 + d * xr.z);
            }
            sr[0][grp] = r0;
            sr[1][grp] = r1;
            sr[2][grp] = r2;
        }
        __syncthreads();
        const int nagg = R / L.agg;
        if ((int)threadIdx.x < nagg) {
            const int I = (t * R) / L.agg + threadIdx.x;
            if (I < nc) {
                double c0 = 0, c1 = 0, c2 = 0;
                for (int q = 0; q < L.agg; q++) {
                    c0 += sr[0][threadIdx.x * L.agg + q];
                    c1 += sr[1][threadIdx.x * L.agg + q];
                    c2 += sr[2][threadIdx.x * L.agg + q];
                }
                bc[I] = make_double4(c0, c1, c2, 0.0);
                const double w = omega * cidg[I];
                xc[I] = make_double4(w * c0, w * c1, w * c2, 0.0);
            }
        }
        __syncthreads();
    }
}

// Up-sweep on level l: x' = x + kc * P xc; y = x' + omega D^-1 (b - L x'). On level 0 (DOT) the
// partial dot products r.z (r = b) are produced for the PCG beta.
template <int G, bool DOT>
__global__ __launch_bounds__(kBlock) void k_prolong_smooth(
    LevelView L, const double4 *__restrict__ b, const double4 *__restrict__ x,
    const double4 *__restrict__ xc, double4 *__restrict__ y, double omega, double kc,
    double *__restrict__ part_rz, const int *__restrict__ flags) {
    if (flags[FL_DONE]) return;
    constexpr int R = kBlock / G;
    const int grp = threadIdx.x / G, l = threadIdx.x % G;
    const int ntiles = (L.n + R - 1) / R;
    int t0, t1;
    tile_range(ntiles, t0, t1);
    const int sh = __ffs(L.agg) - 1;  // agg is a power of two
    double a0 = 0, a1 = 0, a2 = 0;
    for (int t = t0; t < t1; t++) {
        const int row = t * R + grp;
        double s0 = 0, s1 = 0, s2 = 0;
        if (row < L.n) {
            const int beg = L.rowptr[row], end = L.rowptr[row + 1];
            for (int s = beg + l; s < end; s += G) {
                const int c = L.col[s];
                const double v = L.val[s];
                const double4 xf = x[c];
                const double4 xk = xc[c >> sh];
                s0 += v * (xf.x + kc * xk.x);
                s1 += v * (xf.y + kc * xk.y);
                s2 += v * (xf.z + kc * xk.z);
            }
        }
        s0 = group_sum<G>(s0);
        s1 = group_sum<G>(s1);
        s2 = group_sum<G>(s2);
        if (l == 0 && row < L.n) {
            const double4 xf = x[row], xk = xc[row >> sh], br = b[row];
            const double d = L.diag[row], w = omega * L.idg[row];
            const double p0 = xf.x + kc * xk.x, p1 = xf.y + kc * xk.y, p2 = xf.z + kc * xk.z;
            const double y0 = p0 + w * (br.x - (s0 + d * p0));
            const double y1 = p1 + w * (br.y - (s1 + d * p1));
            const double y2 = p2 + w * (br.z - (s2 + d * p2));
            y[row] = make_double4(y0, y1, y2, 0.0);
            if (DOT) {
                a0 += br.x * y0;
                a1 += br.y * y1;
                a2 += br.z * y2;
            }
        }
    }
    if (DOT) block_sum3_store(a0, a1, a2, part_rz + 4 * blockIdx.x);
}

// single-level preconditioner (plain Jacobi): z = D^-1 r, with the convergence prologue
__global__ __launch_bounds__(kBlock) void k_jacobi_z(int n, const double *__restrict__ idg,
                                                     const double4 *__restrict__ r,
                                                     double4 *__restrict__ z,
                                                     double *__restrict__ part_rz,
                                                     const double *__restrict__ part_rr,
                                                     int nparts, int first, double rtol2,
                                                     double *__restrict__ scal,
                                                     int *__restrict__ flags) {
    if (flags[FL_DONE]) return;
    double rr[3], bb[3];
    load_reduced3(part_rr, nparts, rr);
    for (int c = 0; c < 3; c++) bb[c] = first ? rr[c] : scal[SC_BB + c];
    const bool finite = isfinite(rr[0]) && isfinite(rr[1]) && isfinite(rr[2]);
    const bool conv = rr[0] <= rtol2 * bb[0] && rr[1] <= rtol2 * bb[1] && rr[2] <= rtol2 * bb[2];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (first)
            for (int c = 0; c < 3; c++) scal[SC_BB + c] = bb[c];
        for (int c = 0; c < 3; c++) scal[SC_RELRES + c] = bb[c] > 0.0 ? sqrt(rr[c] / bb[c]) : 0.0;
        if (!finite)
            flags[FL_DONE] = 2;
        else if (conv)
            flags[FL_DONE] = 1;
    }
    if (!finite || conv) return;
    double a0 = 0, a1 = 0, a2 = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double w = idg[i];
        const double4 ri = r[i];
        const double4 zi = make_double4(w * ri.x, w * ri.y, w * ri.z, 0.0);
        z[i] = zi;
        a0 += ri.x * zi.x;
        a1 += ri.y * zi.y;
        a2 += ri.z * zi.z;
    }
    block_sum3_store(a0, a1, a2, part_rz + 4 * blockIdx.x);
}

// =============================================================================================
// K5 -- PCG vector kernels (three independent columns share the matrix and the preconditioner)
// =============================================================================================
// INIT: x = 0, r = b (already in R), x0 = omega D^-1 r, partials of ||r||^2.
// else: alpha = rz/pq; x += alpha p; r -= alpha q; x0 = omega D^-1 r; partials of ||r||^2.
template <bool INIT>
__global__ __launch_bounds__(kBlock) void k_pcg_update(
    int n, const double *__restrict__ scal, int par, const double *__restrict__ part_pq, int nparts,
    double4 *__restrict__ X, double4 *__restrict__ R, const double4 *__restrict__ P,
    const double4 *__restrict__ AP, const double *__restrict__ idg, double4 *__restrict__ x0,
    double omega, double *__restrict__ part_rr, int *__restrict__ flags) {
    if (flags[FL_DONE]) return;
    double al[3] = {0, 0, 0};
    if (!INIT) {
        double pq[3];
        load_reduced3(part_pq, nparts, pq);
        for (int c = 0; c < 3; c++) {
            const double rz = scal[(par ? SC_RZ1 : SC_RZ0) + c];
            al[c] = pq[c] > 0.0 ? rz / pq[c] : 0.0;  // pq = 0: column already solved exactly
        }
    }
    double a0 = 0, a1 = 0, a2 = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        double4 r = R[i];
        if (INIT) {
            X[i] = make_double4(0, 0, 0, 0);
        } else {
            const double4 p = P[i], q = AP[i];
            double4 x = X[i];
            x.x += al[0] * p.x;
            x.y += al[1] * p.y;
            x.z += al[2] * p.z;
            X[i] = x;
            r.x -= al[0] * q.x;
            r.y -= al[1] * q.y;
            r.z -= al[2] * q.z;
            R[i] = r;
        }
        const double w = omega * idg[i];
        x0[i] = make_double4(w * r.x, w * r.y, w * r.z, 0.0);
        a0 += r.x * r.x;
        a1 += r.y * r.y;
        a2 += r.z * r.z;
    }
    block_sum3_store(a0, a1, a2, part_rr + 4 * blockIdx.x);
    if (!INIT && blockIdx.x == 0 && threadIdx.x == 0) flags[FL_ITERS] += 1;
}

// beta = rz_new / rz_old (0 on the first pass); p = z + beta p; rz_new stored under the other parity
__global__ __launch_bounds__(kBlock) void k_pcg_pupdate(int n, double *__restrict__ scal, int par,
                                                        int first,
                                                        const double *__restrict__ part_rz,
                                                        int nparts, const double4 *__restrict__ Z,
                                                        double4 *__restrict__ P,
                                                        int *__restrict__ flags) {
    if (flags[FL_DONE]) return;
    double rzn[3], be[3];
    load_reduced3(part_rz, nparts, rzn);
    bool finite = true;
    for (int c = 0; c < 3; c++) {
        const double rzo = scal[(par ? SC_RZ1 : SC_RZ0) + c];
        be[c] = (first || !(rzo > 0.0)) ? 0.0 : rzn[c] / rzo;
        finite = finite && isfinite(rzn[c]);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (int c = 0; c < 3; c++) scal[(par ? SC_RZ0 : SC_RZ1) + c] = rzn[c];
        if (!finite) flags[FL_DONE] = 2;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double4 z = Z[i];
        if (first) {
            P[i] = z;
        } else {
            double4 p = P[i];
            p.x = z.x + be[0] * p.x;
            p.y = z.y + be[1] * p.y;
            p.z = z.z + be[2] * p.z;
            P[i] = p;
        }
    }
}

// =============================================================================================
// K6 -- score, exp map and rotation update (one free view per thread)
// =============================================================================================
__global__ __launch_bounds__(kBlock) void k_apply_step(int n, int f, const double4 *__restrict__ X,
                                                       double4 *__restrict__ Q,
                                                       double *__restrict__ part_score,
                                                       int write) {
    double acc = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double4 x = X[i];
        const double th = sqrt(x.x * x.x + x.y * x.y + x.z * x.z);
        acc += th;  // score = mean ||W3 row|| BEFORE the exp map (ral/l1_irls.cpp:729)
        double sn, cs;
        sincos(th / 2.0, &sn, &cs);
        const double coef = sn / th;
        double4 w = make_double4(x.x * coef, x.y * coef, x.z * coef, cs);
        if (!isfinite(w.x)) w.x = 0.0;  // ral/l1_irls.cpp:491
        if (!isfinite(w.y)) w.y = 0.0;
        if (!isfinite(w.z)) w.z = 0.0;
        if (!isfinite(w.w)) w.w = 0.0;
        const double4 q = qmul(Q[i + f], w);  // right-multiply, no renormalisation (:734-737)
        if (write) Q[i + f] = q;
    }
    block_sum3_store(acc, 0.0, 0.0, part_score + 4 * blockIdx.x);
}

__global__ __launch_bounds__(256) void k_normalise(int n_total, int f, double4 *__restrict__ Q) {
    const int i = f + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total) return;
    double4 q = Q[i];
    const double n2 = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
    if (n2 > 0.0) {  // Eigen normalized() (ral/l1_irls.cpp:982-991)
        const double nn = sqrt(n2);
        q.x /= nn;
        q.y /= nn;
        q.z /= nn;
        q.w /= nn;
    }
    Q[i] = q;
}

__global__ __launch_bounds__(256) void k_fill(long long n, double v, double *__restrict__ p) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
void fill(Graph &g, double *p, long long n, double v) {
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, g.stream, n, v, p);
}

void normalise_rotations(Graph &g) {
    const int cnt = (int)(g.n_total - g.f);
    if (cnt <= 0) return;
    hipLaunchKernelGGL(k_normalise, dim3((cnt + 255) / 256), dim3(256), 0, g.stream,
                       (int)g.n_total, g.f, g.Q.p);
}

// one-shot variant for host-resident rows (irotavg_quat_normalised): staged through HBM so the
// arithmetic is the same kernel as the resident path
int normalise_host_rows(int64_t n, double *Q, int64_t ldq, int f) {
    if (n - f <= 0) return IROTAVG_OK;
    DevBuf<double4> d;
    std::vector<double4> h((size_t)n);
    for (int64_t i = 0; i < n; i++) h[i] = make_double4(Q[i], Q[ldq + i], Q[2 * ldq + i], Q[3 * ldq + i]);
    d.alloc((size_t)n);
    IRH_CHECK(hipMemcpy(d.p, h.data(), sizeof(double4) * (size_t)n, hipMemcpyHostToDevice));
    const int cnt = (int)(n - f);
    hipLaunchKernelGGL(k_normalise, dim3((cnt + 255) / 256), dim3(256), 0, 0, (int)n, f, d.p);
    IRH_CHECK(hipMemcpy(h.data(), d.p, sizeof(double4) * (size_t)n, hipMemcpyDeviceToHost));
    for (int64_t i = f; i < n; i++) {
        Q[i] = h[i].x;
        Q[ldq + i] = h[i].y;
        Q[2 * ldq + i] = h[i].z;
        Q[3 * ldq + i] = h[i].w;
    }
    return IROTAVG_OK;
}

// =============================================================================================
// host drivers
// =============================================================================================
static LevelView view_of(const Level &L) {
    return LevelView{L.n, L.nnz, L.agg, L.rowptr.p, L.col.p, L.val.p, L.diag.p, L.idg.p};
}

static int grid_for_rows(int n, int lanes) {
    const int R = kBlock / lanes;
    const int ntiles = (n + R - 1) / R;
    int gsz = std::min(ntiles, (int)kMaxParts);
    if (gsz >= 8) gsz &= ~7;
    return std::max(gsz, 1);
}
static int grid_for_elems(int n) {
    int gsz = std::min((n + kBlock - 1) / kBlock, (int)kMaxParts);
    if (gsz >= 8) gsz &= ~7;
    return std::max(gsz, 1);
}

#define DISPATCH_LANES(lanes, CALL)      \
    switch (lanes) {                     \
    case 2: { constexpr int G = 2; CALL; } break;   \
    case 4: { constexpr int G = 4; CALL; } break;   \
    case 8: { constexpr int G = 8; CALL; } break;   \
    case 16: { constexpr int G = 16; CALL; } break; \
    case 32: { constexpr int G = 32; CALL; } break; \
    default: { constexpr int G = 64; CALL; } break; \
    }

// refresh all matrix values from per-edge weights: mode 0 = IRLS (d^2, rhs), mode 1 = L1 Hessian
void assemble(Graph &g, int mode, const double *wsrc) {
    Level &L0 = g.levels[0];
    const int grid = grid_for_rows(L0.n, L0.lanes);
    if (mode == 0) {
        DISPATCH_LANES(L0.lanes,
                       hipLaunchKernelGGL((k_assemble0<G, 0>), dim3(grid), dim3(kBlock), 0, g.stream,
                                          L0.n, L0.rowptr.p, g.slot_eid.p, g.bptr.p, g.beid.p,
                                          g.bflag.p, wsrc, g.er.p, (long long)g.mpad, L0.val.p,
                                          L0.excess.p, L0.diag.p, L0.idg.p, L0.b.p));
    } else {
        DISPATCH_LANES(L0.lanes,
                       hipLaunchKernelGGL((k_assemble0<G, 1>), dim3(grid), dim3(kBlock), 0, g.stream,
                                          L0.n, L0.rowptr.p, g.slot_eid.p, g.bptr.p, g.beid.p,
                                          g.bflag.p, wsrc, g.er.p, (long long)g.mpad, L0.val.p,
                                          L0.excess.p, L0.diag.p, L0.idg.p, L0.b.p));
    }
    for (size_t l = 1; l < g.levels.size(); l++) {
        Level &F = g.levels[l - 1];
        Level &C = g.levels[l];
        if (C.nnz > 0) {
            constexpr int G = 8;
            const int grid2 = grid_for_rows(C.nnz, G);
            hipLaunchKernelGGL((k_coarse_vals<G>), dim3(grid2), dim3(kBlock), 0, g.stream, C.nnz,
                               C.cptr.p, C.cidx.p, F.val.p, C.val.p);
        }
        hipLaunchKernelGGL(k_coarse_diag, dim3((C.n + 255) / 256), dim3(256), 0, g.stream, C.n,
                           F.n, F.agg, F.excess.p, C.rowptr.p, C.val.p, C.excess.p, C.diag.p,
                           C.idg.p);
    }
    if (g.ndense > 0) {
        Level &C = g.levels.back();
        const size_t shm = sizeof(double) * ((size_t)g.ndense * g.ndense + 2 * (size_t)g.ndense);
        hipLaunchKernelGGL(k_dense_invert, dim3(1), dim3(1024), shm, g.stream, g.ndense,
                           C.rowptr.p, C.col.p, C.val.p, C.diag.p, g.dense_inv.p);
    }
}

// one V-cycle: z = M^-1 r with r = levels[0].b, x0 = levels[0].x (pre-smoothed) -> levels[0].y
static void vcycle(Graph &g, int first, double rtol2) {
    const int nl = (int)g.levels.size();
    const double omega = g.opt.mg_omega, kc = g.opt.mg_kc;
    Level &L0 = g.levels[0];
    const int np_rr = grid_for_elems(L0.n);
    if (nl == 1) {
        const int grid = grid_for_elems(L0.n);
        hipLaunchKernelGGL(k_jacobi_z, dim3(grid), dim3(kBlock), 0, g.stream, L0.n, L0.idg.p,
                           L0.b.p, L0.y.p, g.part_rz.p, g.part_rr.p, np_rr, first, rtol2,
                           g.scal.p, g.flags.p);
        return;
    }
    for (int l = 0; l < nl - 1; l++) {
        Level &F = g.levels[l];
        Level &C = g.levels[l + 1];
        const int grid = grid_for_rows(F.n, F.lanes);
        LevelView V = view_of(F);
        if (l == 0) {
            DISPATCH_LANES(F.lanes, hipLaunchKernelGGL((k_residual_restrict<G, true>), dim3(grid),
                                                       dim3(kBlock), 0, g.stream, V, F.b.p, F.x.p,
                                                       C.b.p, C.x.p, C.idg.p, C.n, omega,
                                                       g.part_rr.p, np_rr, first, rtol2, g.scal.p,
                                                       g.flags.p));
        } else {
            DISPATCH_LANES(F.lanes, hipLaunchKernelGGL((k_residual_restrict<G, false>), dim3(grid),
                                                       dim3(kBlock), 0, g.stream, V, F.b.p, F.x.p,
                                                       C.b.p, C.x.p, C.idg.p, C.n, omega,
                                                       g.part_rr.p, np_rr, first, rtol2, g.scal.p,
                                                       g.flags.p));
        }
    }
    Level &CL = g.levels[nl - 1];
    if (g.ndense > 0) {
        hipLaunchKernelGGL(k_dense_solve, dim3(1), dim3(256), sizeof(double4) * (size_t)g.ndense,
                           g.stream, g.ndense, g.dense_inv.p, CL.b.p, CL.y.p, g.flags.p);
    } else {
        // no dense inverse (level cap reached): the coarsest correction is its Jacobi sweep
        IRH_CHECK(hipMemcpyAsync(CL.y.p, CL.x.p, sizeof(double4) * (size_t)CL.n,
                                 hipMemcpyDeviceToDevice, g.stream));
    }
    for (int l = nl - 2; l >= 0; l--) {
        Level &F = g.levels[l];
        Level &C = g.levels[l + 1];
        const int grid = grid_for_rows(F.n, F.lanes);
        LevelView V = view_of(F);
        if (l == 0) {
            DISPATCH_LANES(F.lanes, hipLaunchKernelGGL((k_prolong_smooth<G, true>), dim3(grid),
                                                       dim3(kBlock), 0, g.stream, V, F.b.p, F.x.p,
                                                       C.y.p, F.y.p, omega, kc, g.part_rz.p,
                                                       g.flags.p));
        } else {
            DISPATCH_LANES(F.lanes, hipLaunchKernelGGL((k_prolong_smooth<G, false>), dim3(grid),
                                                       dim3(kBlock), 0, g.stream, V, F.b.p, F.x.p,
                                                       C.y.p, F.y.p, omega, kc, g.part_rz.p,
                                                       g.flags.p));
        }
    }
}

static int nparts_rz(Graph &g) {
    Level &L0 = g.levels[0];
    return g.levels.size() == 1 ? grid_for_elems(L0.n) : grid_for_rows(L0.n, L0.lanes);
}

// PCG on L X = levels[0].b (three columns). Matrix values must be assembled. Result in g.X.
int pcg_solve(Graph &g) {
    Level &L0 = g.levels[0];
    const int n = L0.n;
    const double rtol2 = g.opt.pcg_rtol * g.opt.pcg_rtol;
    const int ge = grid_for_elems(n);
    const int gr = grid_for_rows(n, L0.lanes);
    LevelView V0 = view_of(L0);
    IRH_CHECK(hipMemsetAsync(g.flags.p, 0, sizeof(int) * FL_COUNT, g.stream));
    hipLaunchKernelGGL((k_pcg_update<true>), dim3(ge), dim3(kBlock), 0, g.stream, n, g.scal.p, 0,
                       g.part_pq.p, gr, g.X.p, L0.b.p, g.P.p, g.AP.p, L0.idg.p, L0.x.p,
                       g.opt.mg_omega, g.part_rr.p, g.flags.p);
    const double omega1 = g.levels.size() == 1 ? 1.0 : g.opt.mg_omega;
    (void)omega1;
    int h_flags[FL_COUNT] = {0, 0, 0, 0};
    int it = 0;
    const int check = std::max(1, g.opt.pcg_check_every);
    const int maxit = std::max(1, g.opt.pcg_max_iters);
    const int np_rz = nparts_rz(g);
    while (true) {
        for (int c = 0; c < check; c++, it++) {
            const int first = (it == 0);
            const int par = it & 1;
            vcycle(g, first, rtol2);
            hipLaunchKernelGGL(k_pcg_pupdate, dim3(ge), dim3(kBlock), 0, g.stream, n, g.scal.p, par,
                               first, g.part_rz.p, np_rz, L0.y.p, g.P.p, g.flags.p);
            DISPATCH_LANES(L0.lanes,
                           hipLaunchKernelGGL((k_spmv_dot<G>), dim3(gr), dim3(kBlock), 0, g.stream,
                                              V0, g.P.p, g.AP.p, g.part_pq.p, g.flags.p));
            hipLaunchKernelGGL((k_pcg_update<false>), dim3(ge), dim3(kBlock), 0, g.stream, n,
                               g.scal.p, par ^ 1, g.part_pq.p, gr, g.X.p, L0.b.p, g.P.p, g.AP.p,
                               L0.idg.p, L0.x.p, g.opt.mg_omega, g.part_rr.p, g.flags.p);
        }
        // the convergence test of the last update runs in the next V-cycle prologue; enqueue a
        // bare check so the flag is current when the host reads it
        vcycle(g, it == 0, rtol2);
        IRH_CHECK(hipMemcpyAsync(h_flags, g.flags.p, sizeof(int) * FL_COUNT, hipMemcpyDeviceToHost,
                                 g.stream));
        IRH_CHECK(hipStreamSynchronize(g.stream));
        if (h_flags[FL_DONE] != 0) break;
        if (it >= maxit) break;
        // not converged: the V-cycle just run is exactly the one the next iteration needs
        // -> continue with its p-update
        {
            const int first = (it == 0);
            const int par = it & 1;
            hipLaunchKernelGGL(k_pcg_pupdate, dim3(ge), dim3(kBlock), 0, g.stream, n, g.scal.p, par,
                               first, g.part_rz.p, np_rz, L0.y.p, g.P.p, g.flags.p);
            DISPATCH_LANES(L0.lanes,
                           hipLaunchKernelGGL((k_spmv_dot<G>), dim3(gr), dim3(kBlock), 0, g.stream,
                                              V0, g.P.p, g.AP.p, g.part_pq.p, g.flags.p));
            hipLaunchKernelGGL((k_pcg_update<false>), dim3(ge), dim3(kBlock), 0, g.stream, n,
                               g.scal.p, par ^ 1, g.part_pq.p, gr, g.X.p, L0.b.p, g.P.p, g.AP.p,
                               L0.idg.p, L0.x.p, g.opt.mg_omega, g.part_rr.p, g.flags.p);
            it++;
        }
    }
    double h_scal[SC_COUNT];
    IRH_CHECK(hipMemcpyAsync(h_scal, g.scal.p, sizeof(double) * SC_COUNT, hipMemcpyDeviceToHost,
                             g.stream));
    IRH_CHECK(hipStreamSynchronize(g.stream));
    g.stats.pcg_solves += 1;
    g.stats.pcg_iters += h_flags[FL_ITERS];
    g.stats.pcg_iters_last = h_flags[FL_ITERS];
    for (int c = 0; c < 3; c++) g.stats.last_relres[c] = h_scal[SC_RELRES + c];
    if (h_flags[FL_DONE] == 2) return IROTAVG_ERR_SOLVER;
    if (h_flags[FL_DONE] == 0) return IROTAVG_ERR_NOT_CONVERGED;
    return IROTAVG_OK;
}

int ls_solve(Graph &g) {
    assemble(g, 0, g.dw.p);
    return pcg_solve(g);
}

double apply_step(Graph &g) {
    const int n = g.nu;
    const int grid = grid_for_elems(n);
    hipLaunchKernelGGL(k_apply_step, dim3(grid), dim3(kBlock), 0, g.stream, n, g.f, g.X.p, g.Q.p,
                       g.part_score.p, 1);
    IRH_CHECK(hipMemcpyAsync(g.h_part.data(), g.part_score.p, sizeof(double) * 4 * (size_t)grid,
                             hipMemcpyDeviceToHost, g.stream));
    IRH_CHECK(hipStreamSynchronize(g.stream));
    double s = 0.0;
    for (int b = 0; b < grid; b++) s += g.h_part[4 * (size_t)b];
    return s / (double)n;
}

// ral/l1_irls.cpp:559-752
int run_irls(Graph &g, int cost, double sigma, int max_iters, double change_th, int *iters,
             double *runtime, double *trace) {
    if (cost < IROTAVG_L2 || cost > IROTAVG_WELSCH) return IROTAVG_ERR_UNKNOWN_COST;
    const double tic = now_seconds();
    double score = HUGE_VAL;
    int it = 0, rc = IROTAVG_OK;
    fill(g, g.dw.p, (long long)g.mpad, 1.0);  // weights.setOnes() (:577)
    while (score > change_th && it < max_iters) {  // :590, strict >
        launch_edge_residual(g);
        rc = ls_solve(g);
        if (rc != IROTAVG_OK) break;
        launch_update_weights(g, cost, sigma);
        score = apply_step(g);
        if (trace) trace[it] = score;
        it++;
    }
    IRH_CHECK(hipStreamSynchronize(g.stream));
    const double toc = now_seconds();
    *iters = it;
    *runtime = toc - tic;
    g.stats.outer_iters += it;
    g.stats.edge_updates += (int64_t)it * g.m;
    g.stats.seconds_irls += toc - tic;
    return rc;
}

// ---------------------------------------------------------------------------------------------
// kernel timing for the roofline leg of bench.py (HIP events on the handle's stream)
// ---------------------------------------------------------------------------------------------
int time_kernel(Graph &g, int which, int reps, double *ms) {
    hipEvent_t e0, e1;
    IRH_CHECK(hipEventCreate(&e0));
    IRH_CHECK(hipEventCreate(&e1));
    Level &L0 = g.levels[0];
    const int gr = grid_for_rows(L0.n, L0.lanes);
    LevelView V0 = view_of(L0);
    auto once = [&]() {
        switch (which) {
        case 1: launch_edge_residual(g); break;
        case 2: launch_update_weights(g, IROTAVG_GEMAN_MCCLURE, 5 * IRH_PI / 180.0); break;
        case 3: assemble(g, 0, g.dw.p); break;
        case 4:
            DISPATCH_LANES(L0.lanes,
                           hipLaunchKernelGGL((k_spmv_dot<G>), dim3(gr), dim3(kBlock), 0, g.stream,
                                              V0, g.P.p, g.AP.p, g.part_pq.p, g.flags.p));
            break;
        case 5: vcycle(g, 0, -1.0); break;
        case 6:
            hipLaunchKernelGGL(k_apply_step, dim3(grid_for_elems(g.nu)), dim3(kBlock), 0, g.stream,
                               g.nu, g.f, g.X.p, g.Q.p, g.part_score.p, 0);
            break;
        default: break;
        }
    };
    if (which < 1 || which > 6) return IROTAVG_ERR_BAD_ARG;
    IRH_CHECK(hipMemsetAsync(g.flags.p, 0, sizeof(int) * FL_COUNT, g.stream));
    once();  // warm-up
    IRH_CHECK(hipEventRecord(e0, g.stream));
    for (int r = 0; r < reps; r++) once();
    IRH_CHECK(hipEventRecord(e1, g.stream));
    IRH_CHECK(hipEventSynchronize(e1));
    float t = 0.f;
    IRH_CHECK(hipEventElapsedTime(&t, e0, e1));
    *ms = (double)t / std::max(reps, 1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return IROTAVG_OK;
}

}  // namespace irh

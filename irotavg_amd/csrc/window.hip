// window.hip -- the sliding-window solve of ViewGraph::rotAvg(10) (src/ViewGraph.cpp:1263-1435,
// called once per admitted frame, src/IRotAvg.cpp:371-378) as ONE kernel launch.
//
// A window problem has <= 64 free views and a few hundred edges: far too small for the
// multi-kernel path (one l1ra + irls there is ~150 launches and ~40 host round trips, ~6 ms),
// so the complete pipeline -- l1ra (primal-dual LP per coordinate, ral/l1_irls.cpp:228-468,
// 851-912) followed by irls (:559-752) -- runs inside a single workgroup with every vector and
// the dense normal matrix in LDS: one H2D, one launch, one D2H. The arithmetic follows the same
// reference statements as solver.hip / l1pd.hip; the linear solves are exact (dense Gauss-Jordan
// in LDS, dead pivots -> 0 like the oracle).
#include "graph.hpp"
#include "kernels.hpp"

namespace irh {

constexpr int WIN_MAX_NU = 64;    // free views
constexpr int WIN_MAX_NV = 320;   // all views of the sub-problem
constexpr int WIN_MAX_NE = 640;   // edges
constexpr int WIN_THREADS = 256;

struct WinParams {
    int nv, f, ne;
    int l1_max, irls_max, cost;
    double change_th, sigma;
    int seq;  // k_window_wave stores it into WinResult::seq LAST (system scope): the host polls for it
};
struct WinResult {
    int l1_iters, irls_iters, status, seq;
    double l1_score, irls_score;
    long long stamp[8];  // development aid (IROTAVG_WINDOW_STAMPS=1 prints them): s_memtime at the phase boundaries of k_window_wave
};

#define W_PI 3.141592653589793238462643383279502884
#define W_EPS 2.2204e-16

struct WinShared {
    double4 *Q;       // nv
    double *r;        // 3 * ne (planes)
    double *d;        // ne   IRLS weights
    double *W;        // 3 * nu (planes): the step X
    double *H;        // nu * (nu + 1)
    double *B;        // nu * 3
    double *pd;       // 12 * ne  primal-dual edge vectors
    double *pn;       // 5 * nu   primal-dual view vectors
    double *red;      // 16 reduction scratch
    int2 *I;          // ne
    unsigned char *fl;  // ne
};

// ---- workgroup reductions (result to every thread) -------------------------------------------
__device__ __forceinline__ double wg_sum(double v, double *red) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const double s = ((red[0] + red[1]) + red[2]) + red[3];
    __syncthreads();
    return s;
}
__device__ __forceinline__ double wg_min(double v, double *red) {
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_down(v, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const double s = fmin(fmin(red[0], red[1]), fmin(red[2], red[3]));
    __syncthreads();
    return s;
}
__device__ __forceinline__ double wg_max(double v, double *red) {
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const double s = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    __syncthreads();
    return s;
}

// ---- dense SPD solve in LDS: H (n x n, leading dim n+1) X = B (n x nrhs, row-major) ----------
// Gauss-Jordan without pivoting; a non-positive pivot marks a dead variable (row/column dropped,
// solution 0), like the oracle's sparse Cholesky. Returns false if a non-finite value appears.
__device__ bool dense_solve(double *H, int n, double *B, int nrhs, double *red) {
    const int ld = n + 1;
    // dead-pivot threshold: kDeadTol (common.hpp) x the largest original diagonal entry
    __syncthreads();
    double dm = 0.0;
    for (int k = threadIdx.x; k < n; k += blockDim.x) dm = fmax(dm, H[k * ld + k]);
    const double thr0 = kDeadTol * wg_max(dm, red);
    for (int k = 0; k < n; k++) {
        const double piv = H[k * ld + k];
        const bool dead = !(piv > thr0) || !(piv > 0.0);
        const double ip = dead ? 0.0 : 1.0 / piv;
        __syncthreads();
        // scale row k (entries j > k and the right-hand sides)
        for (int j = threadIdx.x; j < n + nrhs; j += blockDim.x) {
            if (j < n) {
                if (j > k) H[k * ld + j] *= ip;
            } else {
                B[k * nrhs + (j - n)] *= ip;
            }
        }
        __syncthreads();
        // eliminate column k from every other row
        const int cols = n - k - 1 + nrhs;
        for (int e = threadIdx.x; e < n * cols; e += blockDim.x) {
            const int i = e / cols, c = e - i * cols;
            if (i == k) continue;
            const double fk = H[i * ld + k];
            if (fk == 0.0) continue;
            if (c < n - k - 1)
                H[i * ld + k + 1 + c] -= fk * H[k * ld + k + 1 + c];
            else
                B[i * nrhs + (c - (n - k - 1))] -= fk * B[k * nrhs + (c - (n - k - 1))];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += blockDim.x)
            if (i != k) H[i * ld + k] = 0.0;
    }
    __syncthreads();
    double bad = 0.0;
    for (int e = threadIdx.x; e < n * nrhs; e += blockDim.x)
        if (!isfinite(B[e])) bad = 1.0;
    return wg_sum(bad, red) == 0.0;
}

// ---- K1 on the window: r_k = log(Qinv_j (x) QQ_k (x) Q_i)  (ral/l1_irls.cpp:109-127,498-532) --
__device__ void win_residual(const WinParams &P, const WinShared &S, const double4 *__restrict__ QQ) {
    for (int k = threadIdx.x; k < P.ne; k += blockDim.x) {
        double4 qj = S.Q[S.I[k].y];
        qj.w = -qj.w;
        const double4 d = qmul(qj, qmul(QQ[k], S.Q[S.I[k].x]));
        const double s2 = sqrt(d.x * d.x + d.y * d.y + d.z * d.z);
        double th = 2.0 * atan2(s2, d.w);
        if (th < -W_PI)
            th += 2.0 * W_PI;
        else if (th >= W_PI)
            th -= 2.0 * W_PI;
        const double aux = th / s2;
        double ox = d.x * aux, oy = d.y * aux, oz = d.z * aux;
        if (s2 < W_EPS) ox = oy = oz = 0.0;
        S.r[k] = ox;
        S.r[P.ne + k] = oy;
        S.r[2 * P.ne + k] = oz;
    }
    __syncthreads();
}

// (A' t)_v with make_A's coefficients, v a free-view index
__device__ __forceinline__ double at_dot(const WinParams &P, const WinShared &S, const double *t, int v) {
    double s = 0.0;
    for (int k = 0; k < P.ne; k++) {
        const unsigned char fl = S.fl[k];
        if ((fl & EF_CJ) && S.I[k].y - P.f == v) s += t[k];
        if ((fl & EF_CI) && S.I[k].x - P.f == v) s -= t[k];
    }
    return s;
}

// score = mean ||W row|| (before the exp map), exp map, Q_{f+i} <- Q_{f+i} (x) W_i
// (ral/l1_irls.cpp:729-737 / :894-902, :471-492)
__device__ double win_apply_step(const WinParams &P, const WinShared &S) {
    const int nu = P.nv - P.f;
    double acc = 0.0;
    for (int i = threadIdx.x; i < nu; i += blockDim.x) {
        const double x = S.W[i], y = S.W[nu + i], z = S.W[2 * nu + i];
        const double th = sqrt(x * x + y * y + z * z);
        acc += th;
        const double sn = sin(th / 2.0), cs = cos(th / 2.0);
        const double coef = sn / th;
        double4 w = make_double4(x * coef, y * coef, z * coef, cs);
        if (!isfinite(w.x)) w.x = 0.0;
        if (!isfinite(w.y)) w.y = 0.0;
        if (!isfinite(w.z)) w.z = 0.0;
        if (!isfinite(w.w)) w.w = 0.0;
        S.Q[i + P.f] = qmul(S.Q[i + P.f], w);
    }
    const double s = wg_sum(acc, S.red);
    return s / (double)nu;
}

// ---- one coordinate of the primal-dual LP, x0 = 0 (ral/l1_irls.cpp:228-468) -------------------
// y: LDS vector of ne entries; result written to xout (nu entries). Returns 0 ok, 1 solver error.
__device__ int win_l1decode(const WinParams &P, const WinShared &S, const double *y, int pdmaxiter,
                            double *xout) {
    const double PDTOL = 1e-3, alpha = 0.01, beta = 0.5, mu = 10;
    const int m = P.ne, nu = P.nv - P.f, f = P.f;
    double *u = S.pd, *Ax = u + m, *f1 = Ax + m, *f2 = f1 + m, *l1 = f2 + m, *l2 = l1 + m;
    double *sigx = l2 + m, *t1 = sigx + m, *t2 = t1 + m, *Adx = t2 + m, *du = Adx + m, *dl1 = du + m;
    double *dl2 = S.r + 3 * m;  // one spare plane behind the residuals (allocated 4 * ne)
    double *x = S.pn, *Atv = x + nu, *Atdv = Atv + nu, *dx = Atdv + nu;
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int i = tid; i < nu; i += nt) x[i] = 0.0;
    double loc = -HUGE_VAL;
    for (int k = tid; k < m; k += nt) {
        Ax[k] = 0.0;
        loc = fmax(loc, fabs(y[k]));
    }
    const double maxabs = wg_max(loc, S.red);
    double a0 = 0, a1 = 0, a2 = 0;
    for (int k = tid; k < m; k += nt) {  // :252-276
        const double uu = fabs(y[k] - Ax[k]) * 0.95 + maxabs * 0.10;
        const double g1 = Ax[k] - y[k] - uu, g2 = -Ax[k] + y[k] - uu;
        const double m1 = -(1.0 / g1), m2 = -(1.0 / g2);
        u[k] = uu;
        f1[k] = g1;
        f2[k] = g2;
        l1[k] = m1;
        l2[k] = m2;
        t1[k] = m1 - m2;
        a0 += g1 * m1;
        a1 += g2 * m2;
        const double rd = 1.0 - m1 - m2;
        a2 += rd * rd;
    }
    const double sa = wg_sum(a0, S.red), sb = wg_sum(a1, S.red), rd_tail2 = wg_sum(a2, S.red);
    double sdg = -(sa + sb);
    double tau = mu * 2 * (double)m / sdg;
    double acc = 0.0;
    for (int v = tid; v < nu; v += nt) {
        const double s = at_dot(P, S, t1, v);
        Atv[v] = s;
        acc += s * s;
    }
    const double atv2 = wg_sum(acc, S.red);
    acc = 0.0;
    for (int k = tid; k < m; k += nt) {
        const double c1 = -l1[k] * f1[k] - 1.0 / tau, c2 = -l2[k] * f2[k] - 1.0 / tau;
        acc += c1 * c1 + c2 * c2;
    }
    double resnorm = sqrt(atv2 + rd_tail2 + wg_sum(acc, S.red));
    int pditer = 0;
    bool done = (sdg < PDTOL) || (pditer >= pdmaxiter);
    while (!done) {
        pditer++;
        const double itau = 1.0 / tau;
        for (int k = tid; k < m; k += nt) {  // :292-305
            const double if1 = 1.0 / f1[k], if2 = 1.0 / f2[k];
            const double w2 = -1 - itau * (if1 + if2);
            const double a = l1[k] / f1[k], b = l2[k] / f2[k];
            const double s1 = -a - b, s2 = a - b;
            sigx[k] = s1 - (s2 * s2) / s1;
            t1[k] = -if1 + if2;
            t2[k] = (s2 / s1) * w2;
        }
        __syncthreads();
        // H11p = reshape(AtA * sigx) (make_AtA: endpoints skipped independently, :825-843),
        // right-hand side w1p = -(1/tau) A't1 - A't2
        const int ld = nu + 1;
        for (int e = tid; e < nu * ld; e += nt) S.H[e] = 0.0;
        __syncthreads();
        for (int v = tid; v < nu; v += nt) {
            double *row = S.H + v * ld;
            for (int k = 0; k < m; k++) {
                const int i = S.I[k].x - f, j = S.I[k].y - f;
                const double s = sigx[k];
                if (i >= 0 && i == j) {
                    if (i == v) row[v] -= s;
                    continue;
                }
                if (i == v) {
                    row[v] += s;
                    if (j >= 0) row[j] -= s;
                }
                if (j == v) {
                    row[v] += s;
                    if (i >= 0) row[i] -= s;
                }
            }
            const double w1 = -itau * at_dot(P, S, t1, v);
            dx[v] = w1 - at_dot(P, S, t2, v);
        }
        __syncthreads();
        if (!dense_solve(S.H, nu, dx, 1, S.red)) return 1;
        double smin = HUGE_VAL;
        for (int k = tid; k < m; k += nt) {  // :324-381
            const unsigned char fl = S.fl[k];
            double adx = 0.0;
            if (fl & EF_CJ) adx += dx[S.I[k].y - f];
            if (fl & EF_CI) adx -= dx[S.I[k].x - f];
            const double g1 = f1[k], g2 = f2[k], m1 = l1[k], m2 = l2[k];
            const double if1 = 1.0 / g1, if2 = 1.0 / g2;
            const double w2 = -1 - itau * (if1 + if2);
            const double a = m1 / g1, b = m2 / g2;
            const double s1 = -a - b, s2 = a - b;
            const double d_u = (w2 - s2 * adx) / s1;
            double d1 = -m1 / g1;
            d1 *= (adx - d_u);
            d1 -= m1;
            d1 -= itau * if1;
            double d2 = m2 / g2;
            d2 *= (adx + d_u);
            d2 -= m2;
            d2 -= itau * if2;
            Adx[k] = adx;
            du[k] = d_u;
            dl1[k] = d1;
            dl2[k] = d2;
            t1[k] = d1 - d2;
            if (d1 < 0) smin = fmin(smin, -m1 / d1);
            if (d2 < 0) smin = fmin(smin, -m2 / d2);
            const double p = adx - d_u;
            if (p > 0) smin = fmin(smin, -g1 / p);
            const double q = -adx - d_u;
            if (q > 0) smin = fmin(smin, -g2 / q);
        }
        double s = fmin(1.0, wg_min(smin, S.red));
        if (!(s == s)) return 1;
        s *= 0.99;
        for (int v = tid; v < nu; v += nt) Atdv[v] = at_dot(P, S, t1, v);
        __syncthreads();
        bool suffdec = false;
        int backiter = 0;
        double s_acc = s, rdp2 = 0.0;
        while (!suffdec) {  // :384-429
            double b0 = 0, b1 = 0;
            for (int v = tid; v < nu; v += nt) {
                const double q = Atv[v] + s * Atdv[v];
                b0 += q * q;
            }
            for (int k = tid; k < m; k += nt) {
                const double up = u[k] + s * du[k];
                const double axp = Ax[k] + s * Adx[k];
                const double m1 = l1[k] + s * dl1[k], m2 = l2[k] + s * dl2[k];
                const double g1 = axp - y[k] - up, g2 = -axp + y[k] - up;
                const double r = 1.0 + (-m1 - m2);
                b0 += r * r;
                const double c1 = -m1 * g1 - itau, c2 = -m2 * g2 - itau;
                b1 += c1 * c1 + c2 * c2;
            }
            rdp2 = wg_sum(b0, S.red);
            const double rcp2 = wg_sum(b1, S.red);
            suffdec = sqrt(rdp2 + rcp2) <= (1 - alpha * s) * resnorm;
            s_acc = s;
            s *= beta;
            backiter++;
            if (backiter > 32) {  // "Stuck backtracking": the previous iterate is returned
                for (int v = tid; v < nu; v += nt) xout[v] = x[v];
                __syncthreads();
                return 0;
            }
        }
        for (int v = tid; v < nu; v += nt) {
            x[v] += s_acc * dx[v];
            Atv[v] += s_acc * Atdv[v];
        }
        a0 = a1 = 0;
        for (int k = tid; k < m; k += nt) {
            const double up = u[k] + s_acc * du[k];
            const double axp = Ax[k] + s_acc * Adx[k];
            const double m1 = l1[k] + s_acc * dl1[k], m2 = l2[k] + s_acc * dl2[k];
            const double g1 = axp - y[k] - up, g2 = -axp + y[k] - up;
            u[k] = up;
            Ax[k] = axp;
            l1[k] = m1;
            l2[k] = m2;
            f1[k] = g1;
            f2[k] = g2;
            a0 += g1 * m1;
            a1 += g2 * m2;
        }
        sdg = -(wg_sum(a0, S.red) + wg_sum(a1, S.red));
        tau = mu * 2 * (double)m / sdg;
        acc = 0.0;
        for (int k = tid; k < m; k += nt) {
            const double c1 = -l1[k] * f1[k] - 1.0 / tau, c2 = -l2[k] * f2[k] - 1.0 / tau;
            acc += c1 * c1 + c2 * c2;
        }
        resnorm = sqrt(rdp2 + wg_sum(acc, S.red));
        done = (sdg < PDTOL) || (pditer >= pdmaxiter);
    }
    for (int v = tid; v < nu; v += nt) xout[v] = x[v];
    __syncthreads();
    return 0;
}

__device__ __forceinline__ double win_weight(int cost, double sigma, double e2, double prev);

__global__ __launch_bounds__(WIN_THREADS) void k_window_solve(WinParams P, const int2 *__restrict__ Ig,
                                                              const double4 *__restrict__ QQ,
                                                              double4 *__restrict__ Qg,
                                                              double *__restrict__ weights,
                                                              WinResult *__restrict__ out) {
    extern __shared__ double4 smem4[];
    const int nv = P.nv, ne = P.ne, f = P.f, nu = nv - f;
    WinShared S;
    S.Q = smem4;
    double *p = reinterpret_cast<double *>(S.Q + nv);
    S.r = p;          p += 4 * ne;
    S.d = p;          p += ne;
    S.W = p;          p += 3 * nu;
    S.H = p;          p += nu * (nu + 1);
    S.B = p;          p += 3 * nu;
    S.pd = p;         p += 12 * ne;
    S.pn = p;         p += 5 * nu;
    S.red = p;        p += 16;
    S.I = reinterpret_cast<int2 *>(p);
    S.fl = reinterpret_cast<unsigned char *>(S.I + ne);
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int i = tid; i < nv; i += nt) S.Q[i] = Qg[i];
    for (int k = tid; k < ne; k += nt) {
        const int2 e = Ig[k];
        S.I[k] = e;
        unsigned char fl = 0;  // make_A coefficients (ral/l1_irls.cpp:764-777)
        if (e.y >= f) {
            if (e.x >= f && e.x == e.y) {
                fl = EF_CI;
            } else {
                fl |= EF_CJ;
                if (e.x >= f) fl |= EF_CI;
            }
        }
        S.fl[k] = fl;
    }
    __syncthreads();
    int status = 0;
    // ---------------- l1ra (ral/l1_irls.cpp:851-912) ----------------
    double score = HUGE_VAL, change_th = P.change_th;
    int iter = 0, l1_step = 2;
    while (((score >= change_th) || (l1_step < 2)) && (iter < P.l1_max) && status == 0) {
        if (score < change_th) {  // unreachable under the guard above; kept literal (:879-883)
            l1_step *= 4;
            change_th /= 100.0;
        }
        win_residual(P, S, QQ);
        for (int c = 0; c < 3 && status == 0; c++)
            status = win_l1decode(P, S, S.r + c * ne, l1_step, S.W + c * nu);
        if (status) break;
        score = win_apply_step(P, S);
        iter++;
    }
    const int l1_iters = iter;
    const double l1_score = score;
    // ---------------- irls (ral/l1_irls.cpp:559-752) ----------------
    for (int k = tid; k < ne; k += nt) S.d[k] = 1.0;
    __syncthreads();
    score = HUGE_VAL;
    iter = 0;
    while (score > P.change_th && iter < P.irls_max && status == 0) {
        win_residual(P, S, QQ);
        const int ld = nu + 1;
        for (int e = tid; e < nu * ld; e += nt) S.H[e] = 0.0;
        __syncthreads();
        for (int v = tid; v < nu; v += nt) {  // A'D^2A and A'D^2 r with make_A's A
            double *row = S.H + v * ld;
            double b0 = 0, b1 = 0, b2 = 0;
            for (int k = 0; k < ne; k++) {
                const unsigned char fl = S.fl[k];
                if (!fl) continue;
                const int i = S.I[k].x - f, j = S.I[k].y - f;
                const double s = S.d[k] * S.d[k];
                if ((fl & EF_CJ) && j == v) {
                    row[v] += s;
                    if (fl & EF_CI) row[i] -= s;
                    b0 += s * S.r[k];
                    b1 += s * S.r[ne + k];
                    b2 += s * S.r[2 * ne + k];
                }
                if ((fl & EF_CI) && i == v) {
                    row[v] += s;
                    if (fl & EF_CJ) row[j] -= s;
                    b0 -= s * S.r[k];
                    b1 -= s * S.r[ne + k];
                    b2 -= s * S.r[2 * ne + k];
                }
            }
            S.B[3 * v] = b0;
            S.B[3 * v + 1] = b1;
            S.B[3 * v + 2] = b2;
        }
        __syncthreads();
        if (!dense_solve(S.H, nu, S.B, 3, S.red)) {
            status = IROTAVG_ERR_SOLVER;
            break;
        }
        for (int v = tid; v < nu; v += nt) {
            S.W[v] = S.B[3 * v];
            S.W[nu + v] = S.B[3 * v + 1];
            S.W[2 * nu + v] = S.B[3 * v + 2];
        }
        __syncthreads();
        for (int k = tid; k < ne; k += nt) {  // E = A W3 - w, weights (:614-727)
            const unsigned char fl = S.fl[k];
            double e0 = 0, e1 = 0, e2c = 0;
            if (fl & EF_CJ) {
                const int j = S.I[k].y - f;
                e0 += S.W[j];
                e1 += S.W[nu + j];
                e2c += S.W[2 * nu + j];
            }
            if (fl & EF_CI) {
                const int i = S.I[k].x - f;
                e0 -= S.W[i];
                e1 -= S.W[nu + i];
                e2c -= S.W[2 * nu + i];
            }
            e0 -= S.r[k];
            e1 -= S.r[ne + k];
            e2c -= S.r[2 * ne + k];
            S.d[k] = win_weight(P.cost, P.sigma, e0 * e0 + e1 * e1 + e2c * e2c, S.d[k]);
        }
        __syncthreads();
        score = win_apply_step(P, S);
        iter++;
    }
    if (status == 1) status = IROTAVG_ERR_SOLVER;
    __syncthreads();
    for (int i = tid; i < nv; i += nt) Qg[i] = S.Q[i];
    for (int k = tid; k < ne; k += nt) weights[k] = S.d[k];
    if (tid == 0) {
        out->l1_iters = l1_iters;
        out->irls_iters = iter;
        out->status = status;
        out->l1_score = l1_score;
        out->irls_score = score;
    }
}

// the 14 robust weights (ral/l1_irls.cpp:617-727); same statements as robust_weight() in solver.hip
__device__ __forceinline__ double win_weight(int cost, double sigma, double e2, double prev) {
    switch (cost) {
    case IROTAVG_L2: return prev;
    case IROTAVG_L05: { double w = 1.0 / pow(e2, 3. / 8.); return w > 1e4 ? 1e4 : w; }
    case IROTAVG_L1: { double w = 1.0 / sqrt(sqrt(e2)); return w > 1e4 ? 1e4 : w; }
    case IROTAVG_L15: { double w = 1.0 / sqrt(sqrt(sqrt(e2))); return w > 1e4 ? 1e4 : w; }
    case IROTAVG_GEMAN_MCCLURE: return 1.0 / (e2 + sigma * sigma);
    case IROTAVG_HUBER: { const double e = sqrt(e2) / (1.345 * sigma); return e >= 1 ? sqrt(1. / e) : prev; }
    case IROTAVG_PSEUDO_HUBER: return 1.0 / sqrt(sqrt(1.0 + e2 / (sigma * sigma)));
    case IROTAVG_ANDREWS: {
        const double e = sqrt(e2) / (1.339 * sigma);
        double w = sqrt(sin(e) / e);
        if (e >= W_PI) w = 0; else if (e < .0001) w = 1;
        return w < 0.0001 ? 0.0001 : w;
    }
    case IROTAVG_BISQUARE: { const double t = 4.685 * sigma; double w = 1.0 - e2 / (t * t); return w < 0.0001 ? 0.0001 : w; }
    case IROTAVG_CAUCHY: { const double t = 2.385 * sigma; return 1.0 / sqrt(1.0 + e2 / (t * t)); }
    case IROTAVG_FAIR: return 1.0 / sqrt(1.0 + sqrt(e2) / (1.400 * sigma));
    case IROTAVG_LOGISTIC: { const double e = sqrt(e2) / (1.205 * sigma); return e < 0.0001 ? 1.0 : sqrt(tanh(e) / e); }
    case IROTAVG_TALWAR: { const double t = 2.795 * sigma; return e2 < t * t ? 1.0001 : 0.0; }
    default: { const double t = 2.985 * sigma; double w = exp(-.5 * e2 / (t * t)); return w < 0.0001 ? 0.0001 : w; }
    }
}

// =================================================================================================
// The wave-resident variant for the usual rotAvg(10) sub-problem (<= 64 edges, <= 16 free views,
// src/IRotAvg.cpp:158-161,371-378: each view is linked to its 4 predecessors). The general kernel
// above spends its time in LDS round trips and workgroup barriers; here every edge vector of the
// primal-dual LP lives in the registers of ONE lane of a wave (lane k = edge k, lane v = free view
// v), the dense normal matrix is a 16-register row per lane (Gauss-Jordan by v_readlane
// broadcasts), reductions are wave butterflies, and the three coordinate LPs of l1ra / the three
// right-hand sides of irls run on three waves at once. Two workgroup barriers per outer iteration.
// Statement order inside the sums follows the general kernel (k ascending), so both give the same
// normal matrices bit for bit; only the order of the wave reductions differs.
// =================================================================================================
constexpr int SM_MAX_NE = 64;
constexpr int SM_MAX_NU = 16;
constexpr int SM_THREADS = 192;
constexpr unsigned ADJ_NEG = 1u << 16;   // the edge enters the view's row with coefficient -1
constexpr unsigned ADJ_SELF = 1u << 17;  // make_AtA self-loop entry

__device__ __forceinline__ double rl_d(double x, int k) {  // k wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), k);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), k);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int rl_i(int x, int k) { return __builtin_amdgcn_readlane(x, k); }

// wave64 reductions on the DPP network (no LDS traffic): four steps inside each row of 16 lanes,
// two row broadcasts, the total is read from lane 63 -> identical bits in every lane
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_d(double ident, double x) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(ident), __double2loint(x), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(ident), __double2hiint(x), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
struct OpSum {
    static __device__ __forceinline__ double id() { return 0.0; }
    static __device__ __forceinline__ double f(double a, double b) { return a + b; }
};
struct OpMin {
    static __device__ __forceinline__ double id() { return HUGE_VAL; }
    static __device__ __forceinline__ double f(double a, double b) { return fmin(a, b); }
};
struct OpMax {
    static __device__ __forceinline__ double id() { return -HUGE_VAL; }
    static __device__ __forceinline__ double f(double a, double b) { return fmax(a, b); }
};
template <class Op>
__device__ __forceinline__ double wv_reduce(double v) {
    v = Op::f(v, dpp_d<0xb1, 0xf>(Op::id(), v));   // quad_perm [1,0,3,2]
    v = Op::f(v, dpp_d<0x4e, 0xf>(Op::id(), v));   // quad_perm [2,3,0,1]
    v = Op::f(v, dpp_d<0x124, 0xf>(Op::id(), v));  // row_ror 4
    v = Op::f(v, dpp_d<0x128, 0xf>(Op::id(), v));  // row_ror 8
    v = Op::f(v, dpp_d<0x142, 0xa>(Op::id(), v));  // row_bcast 15 -> rows 1, 3
    v = Op::f(v, dpp_d<0x143, 0xc>(Op::id(), v));  // row_bcast 31 -> rows 2, 3
    return rl_d(v, 63);
}
__device__ __forceinline__ double wv_sum(double v) { return wv_reduce<OpSum>(v); }
__device__ __forceinline__ double wv_min(double v) { return wv_reduce<OpMin>(v); }
__device__ __forceinline__ double wv_max(double v) { return wv_reduce<OpMax>(v); }

// what a lane knows about "its" edge (lane k < m) -- and, as lane v < nu, about its free view
struct SmLane {
    int m, nu, lane;
    int cj, ci;  // make_A columns of the edge (+1 at cj, -1 at ci), -1 = absent (ral/l1_irls.cpp:764-777)
    // incidence lists of view `lane`, ascending edge id; entry = k | (other column + 1) << 8 | flags
    const unsigned *adjA;  // make_A's A: rows that touch the view (A' products, the IRLS normal matrix)
    const unsigned *adjH;  // make_AtA's endpoint rule (:825-843): the primal-dual normal matrix
    int degA, degH, maxA, maxH;  // own list lengths, wave-wide maxima
    double *Hrow;                // this wave's LDS row of the view (16 entries)
};

// (A' t)_v on lane v: k ascending, +t_k through cj then -t_k through ci (same order as at_dot)
__device__ __forceinline__ double sm_at(const SmLane &E, double t) {
    double s = 0.0;
    // four entries per step: the list loads and the lane exchanges of a step are independent of each other (one by one
    // each entry was an LDS load -> ds_bpermute -> add chain); the sum keeps its order
    for (int d0 = 0; d0 < E.maxA; d0 += 4) {
        unsigned e[4];
        double tk[4];
#pragma unroll
        for (int u = 0; u < 4; u++) e[u] = d0 + u < E.degA ? E.adjA[d0 + u] : 0u;
#pragma unroll
        for (int u = 0; u < 4; u++) tk[u] = __shfl(t, (int)(e[u] & 63u), 64);
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (d0 + u < E.degA) s += (e[u] & ADJ_NEG) ? -tk[u] : tk[u];
    }
    return s;
}
// (A x)_k on lane k
__device__ __forceinline__ double sm_ax(const SmLane &E, double x) {
    const double xj = __shfl(x, E.cj < 0 ? 0 : E.cj, 64), xi = __shfl(x, E.ci < 0 ? 0 : E.ci, 64);
    double s = 0.0;
    if (E.cj >= 0) s += xj;
    if (E.ci >= 0) s -= xi;
    return s;
}

// Gauss-Jordan on rows held in registers: lane i owns row i (h[c], c < 16) and its right-hand
// side. Pivot rows stay unscaled (row i -= (h_ik / h_kk) * row k, x_i = b_i / h_ii at the end): one
// broadcast + one fma per column. A non-positive pivot marks a dead variable (never eliminates,
// solution 0), like dense_solve above / the oracle's Cholesky. False if a non-finite value appears.
// NUP: nu rounded up to a multiple of four -- the loops unroll to straight-line code (with nu as a run-time bound every
// column update sat behind a scalar branch: ~12 clocks per instruction, the solve 30 % of the kernel). Rows and
// columns nu .. NUP - 1 are zero: their pivots are dead, their multipliers 0.
template <int NUP>
__device__ __forceinline__ void sm_solve_t(double (&h)[SM_MAX_NU], double &b, double thr, int lane, double &myip) {
#pragma unroll
    for (int k = 0; k < NUP; k++) {
        // the reciprocal -- the only chain from pivot to pivot -- by v_rcp_f64 + two Newton steps (a third of the IEEE
        // division's dependent instructions)
        const double piv = rl_d(h[k], k);
        double ip = __builtin_amdgcn_rcp(piv);
        ip = fma(fma(-piv, ip, 1.0), ip, ip);
        ip = fma(fma(-piv, ip, 1.0), ip, ip);
        ip = (piv > thr && piv > 0.0) ? ip : 0.0;
        const bool me = lane == k;
        if (me) myip = ip;
        const double mult = me ? 0.0 : -(h[k] * ip);
#pragma unroll
        for (int c = k + 1; c < NUP; c++) h[c] = fma(mult, rl_d(h[c], k), h[c]);
        b = fma(mult, rl_d(b, k), b);
    }
}
__device__ __forceinline__ bool sm_solve(double (&h)[SM_MAX_NU], double &b, int nu, int lane) {
    double myip = 0.0;
    double dg = 0.0;  // own original diagonal entry
#pragma unroll
    for (int c = 0; c < SM_MAX_NU; c++)
        if (c == lane) dg = h[c];
    const double thr = kDeadTol * wv_max(lane < nu ? dg : 0.0);  // relative to the largest one
    if (lane >= nu) b = 0.0;
    if (nu <= 4)
        sm_solve_t<4>(h, b, thr, lane, myip);
    else if (nu <= 8)
        sm_solve_t<8>(h, b, thr, lane, myip);
    else if (nu <= 12)
        sm_solve_t<12>(h, b, thr, lane, myip);
    else
        sm_solve_t<16>(h, b, thr, lane, myip);
    b *= myip;
    return __ballot(lane < nu && !isfinite(b)) == 0ull;
}

// one coordinate of the primal-dual LP on one wave, x0 = 0 (ral/l1_irls.cpp:228-468); y on edge
// lanes, result on view lanes. Returns 0 ok, 1 solver error.
__device__ int sm_l1decode(const SmLane &E, const double y, const int pdmaxiter, double &xout) {
    const double PDTOL = 1e-3, alpha = 0.01, beta = 0.5, mu = 10;
    const int m = E.m, lane = E.lane;
    const bool ek = lane < m, vk = lane < E.nu;
    double x = 0.0, Ax = 0.0;
    const double maxabs = wv_max(ek ? fabs(y) : -HUGE_VAL);
    double u = 0, f1 = 0, f2 = 0, l1 = 0, l2 = 0, t1 = 0;
    double a0 = 0, a1 = 0, a2 = 0;
    if (ek) {  // :252-276
        u = fabs(y - Ax) * 0.95 + maxabs * 0.10;
        f1 = Ax - y - u;
        f2 = -Ax + y - u;
        l1 = -(1.0 / f1);
        l2 = -(1.0 / f2);
        t1 = l1 - l2;
        a0 = f1 * l1;
        a1 = f2 * l2;
        const double rd = 1.0 - l1 - l2;
        a2 = rd * rd;
    }
    const double sa = wv_sum(a0), sb = wv_sum(a1), rd_tail2 = wv_sum(a2);
    double sdg = -(sa + sb);
    double tau = mu * 2 * (double)m / sdg;
    double Atv = sm_at(E, t1);
    const double atv2 = wv_sum(Atv * Atv);
    double acc = 0.0;
    if (ek) {
        const double c1 = -l1 * f1 - 1.0 / tau, c2 = -l2 * f2 - 1.0 / tau;
        acc = c1 * c1 + c2 * c2;
    }
    double resnorm = sqrt(atv2 + rd_tail2 + wv_sum(acc));
    int pditer = 0;
    bool done = (sdg < PDTOL) || (pditer >= pdmaxiter);
    while (!done) {
        pditer++;
        const double itau = 1.0 / tau;
        double sigx = 0.0, t2 = 0.0, if1 = 0, if2 = 0, w2 = 0, s1 = 1, s2 = 0, qa = 0, qb = 0;
        t1 = 0.0;
        if (ek) {  // :292-305
            if1 = 1.0 / f1;
            if2 = 1.0 / f2;
            w2 = -1 - itau * (if1 + if2);
            qa = l1 / f1;
            qb = l2 / f2;
            s1 = -qa - qb;
            s2 = qa - qb;
            sigx = s1 - (s2 * s2) / s1;
            t1 = -if1 + if2;
            t2 = (s2 / s1) * w2;
        }
        // H11p = A' diag(sigx) A with make_AtA's endpoint rule: lane v accumulates row v in LDS
        // (k ascending), then holds it in registers for the elimination
        double h[SM_MAX_NU];
        {
            double hd = 0.0;
            if (vk) {
#pragma unroll
                for (int c = 0; c < SM_MAX_NU; c++) E.Hrow[c] = 0.0;
            }
            for (int d0 = 0; d0 < E.maxH; d0 += 4) {  // (four entries per step, as sm_at)
                unsigned e[4];
                double sk[4];
#pragma unroll
                for (int u = 0; u < 4; u++) e[u] = d0 + u < E.degH ? E.adjH[d0 + u] : 0u;
#pragma unroll
                for (int u = 0; u < 4; u++) sk[u] = __shfl(sigx, (int)(e[u] & 63u), 64);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (d0 + u < E.degH) {
                        const double s = sk[u];
                        if (e[u] & ADJ_SELF) {
                            hd -= s;
                        } else {
                            hd += s;
                            const int o = (int)((e[u] >> 8) & 255u);
                            if (o) atomicAdd(&E.Hrow[o - 1], -s);  // ds_add_f64 without return: no read-modify-write round trip per entry; same bits as -= s
                        }
                    }
                }
            }
            if (vk) E.Hrow[lane] = hd;
#pragma unroll
            for (int c = 0; c < SM_MAX_NU; c++) h[c] = vk ? E.Hrow[c] : 0.0;
        }
        const double w1 = -itau * sm_at(E, t1);
        double dx = w1 - sm_at(E, t2);
        if (!sm_solve(h, dx, E.nu, lane)) return 1;
        if (!vk) dx = 0.0;
        const double adx = sm_ax(E, dx);
        double du = 0, dl1 = 0, dl2 = 0, Adx = 0, smin = HUGE_VAL;
        t1 = 0.0;
        if (ek) {  // :324-381
            const double d_u = (w2 - s2 * adx) / s1;
            double d1 = -qa;
            d1 *= (adx - d_u);
            d1 -= l1;
            d1 -= itau * if1;
            double d2 = qb;
            d2 *= (adx + d_u);
            d2 -= l2;
            d2 -= itau * if2;
            Adx = adx;
            du = d_u;
            dl1 = d1;
            dl2 = d2;
            t1 = d1 - d2;
            if (d1 < 0) smin = fmin(smin, -l1 / d1);
            if (d2 < 0) smin = fmin(smin, -l2 / d2);
            const double p = adx - d_u;
            if (p > 0) smin = fmin(smin, -f1 / p);
            const double q = -adx - d_u;
            if (q > 0) smin = fmin(smin, -f2 / q);
        }
        double s = fmin(1.0, wv_min(smin));
        if (!(s == s)) return 1;
        s *= 0.99;
        const double Atdv = sm_at(E, t1);
        bool suffdec = false;
        int backiter = 0;
        double s_acc = s, rdp2 = 0.0;
        while (!suffdec) {  // :384-429
            double b0 = 0, b1 = 0;
            if (vk) {
                const double q = Atv + s * Atdv;
                b0 = q * q;
            }
            if (ek) {
                const double up = u + s * du;
                const double axp = Ax + s * Adx;
                const double m1 = l1 + s * dl1, m2 = l2 + s * dl2;
                const double g1 = axp - y - up, g2 = -axp + y - up;
                const double r = 1.0 + (-m1 - m2);
                b0 += r * r;
                const double c1 = -m1 * g1 - itau, c2 = -m2 * g2 - itau;
                b1 = c1 * c1 + c2 * c2;
            }
            rdp2 = wv_sum(b0);
            const double rcp2 = wv_sum(b1);
            suffdec = sqrt(rdp2 + rcp2) <= (1 - alpha * s) * resnorm;
            s_acc = s;
            s *= beta;
            backiter++;
            if (backiter > 32) {  // "Stuck backtracking": the previous iterate is returned
                xout = x;
                return 0;
            }
        }
        x += s_acc * dx;
        Atv += s_acc * Atdv;
        a0 = a1 = 0;
        if (ek) {
            u = u + s_acc * du;
            Ax = Ax + s_acc * Adx;
            l1 = l1 + s_acc * dl1;
            l2 = l2 + s_acc * dl2;
            f1 = Ax - y - u;
            f2 = -Ax + y - u;
            a0 = f1 * l1;
            a1 = f2 * l2;
        }
        sdg = -(wv_sum(a0) + wv_sum(a1));
        tau = mu * 2 * (double)m / sdg;
        acc = 0.0;
        if (ek) {
            const double c1 = -l1 * f1 - 1.0 / tau, c2 = -l2 * f2 - 1.0 / tau;
            acc = c1 * c1 + c2 * c2;
        }
        resnorm = sqrt(rdp2 + wv_sum(acc));
        done = (sdg < PDTOL) || (pditer >= pdmaxiter);
    }
    xout = x;
    return 0;
}

// BATCH: workgroup b solves the b-th of several INDEPENDENT windows (one per view-graph of a multi-session
// server: irotavg_viewgraph_rot_avg_batch) -- its parameters come from Pb[b] and its arrays lie b * stride bytes
// behind the first problem's. A single window is the kernel-argument form (no extra round trip for P).
template <bool BATCH>
__global__ __launch_bounds__(SM_THREADS) void k_window_wave(WinParams Pk, const WinParams *__restrict__ Pb,
                                                           size_t stride, const int2 *__restrict__ Ig,
                                                           const double4 *__restrict__ QQ,
                                                           double4 *__restrict__ Qg,
                                                           double *__restrict__ weights,
                                                           WinResult *__restrict__ out) {
    const WinParams P = BATCH ? *reinterpret_cast<const WinParams *>(reinterpret_cast<const unsigned char *>(Pb) + stride * blockIdx.x) : Pk;
    if (BATCH) {
        const size_t off = stride * blockIdx.x;
        Ig = reinterpret_cast<const int2 *>(reinterpret_cast<const unsigned char *>(Ig) + off);
        QQ = reinterpret_cast<const double4 *>(reinterpret_cast<const unsigned char *>(QQ) + off);
        Qg = reinterpret_cast<double4 *>(reinterpret_cast<unsigned char *>(Qg) + off);
        weights = reinterpret_cast<double *>(reinterpret_cast<unsigned char *>(weights) + off);
        out = reinterpret_cast<WinResult *>(reinterpret_cast<unsigned char *>(out) + off);
    }
    __shared__ double4 sQ[WIN_MAX_NV];
    __shared__ double sW[3][SM_MAX_NU];
    __shared__ double sH[3][SM_MAX_NU][SM_MAX_NU + 1];
    __shared__ unsigned sAdjA[SM_MAX_NU][SM_MAX_NE], sAdjH[SM_MAX_NU][SM_MAX_NE];
    __shared__ int sDeg[2][SM_MAX_NU];
    __shared__ int sStatus[3];
    const int nv = P.nv, ne = P.ne, f = P.f, nu = nv - f;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    long long stamp[6];
    stamp[0] = (long long)__builtin_amdgcn_s_memtime();
    for (int i = tid; i < nv; i += SM_THREADS) sQ[i] = Qg[i];
    SmLane E;
    E.m = ne;
    E.nu = nu;
    E.lane = lane;
    E.cj = E.ci = -1;
    int ai = -1, aj = -2;  // make_AtA endpoints i - f, j - f (negative = fixed)
    int ex = 0, ey = 0;
    double4 qq = make_double4(0, 0, 0, 1);
    if (lane < ne) {
        const int2 e = Ig[lane];
        qq = QQ[lane];
        ex = e.x;
        ey = e.y;
        ai = e.x - f;
        aj = e.y - f;
        if (e.y >= f) {  // make_A coefficients (ral/l1_irls.cpp:764-777)
            if (e.x >= f && e.x == e.y) {
                E.ci = e.x - f;
            } else {
                E.cj = e.y - f;
                if (e.x >= f) E.ci = e.x - f;
            }
        }
    }
    const bool ek = lane < ne, vk = lane < nu;
    if (wave == 0) {
        // Incidence lists of the free views, ascending edge id. The edges that touch view v are the set bits of wave
        // ballots (lane k = edge k), taken for all views in turn and kept by lane v; lane v then walks ITS bits in
        // ascending order -- a handful of steps -- and fetches the other endpoint of each edge from the edge's lane.
        // (Until round 4 every lane walked all ne edges through readlanes: 40 serial steps, a fifth of the kernel.)
        const bool self = ek && ai >= 0 && ai == aj;
        unsigned long long mAj = 0, mAi = 0, mHs = 0, mHi = 0, mHj = 0;
        for (int v = 0; v < nu; v++) {
            const unsigned long long bj = __ballot(ek && E.cj == v), bi = __ballot(ek && E.ci == v);
            const unsigned long long hs = __ballot(self && ai == v), hi = __ballot(ek && !self && ai == v),
                                     hj = __ballot(ek && !self && aj == v);
            if (lane == v) {
                mAj = bj;
                mAi = bi;
                mHs = hs;
                mHi = hi;
                mHj = hj;
            }
        }
        const int row = vk ? lane : 0;
        int da = 0, dh = 0;
        unsigned long long ma = mAj | mAi, mh = mHs | mHi | mHj;
        while (__ballot(ma != 0ull || mh != 0ull) != 0ull) {  // wave-uniform trip count: the shuffles need every lane
            const int ka = ma ? (int)__builtin_ctzll(ma) : 0, kh = mh ? (int)__builtin_ctzll(mh) : 0;
            const int kj = __shfl(E.cj, ka, 64), ki = __shfl(E.ci, ka, 64);
            const int hi_ = __shfl(ai, kh, 64), hj_ = __shfl(aj, kh, 64);
            if (ma) {  // (an edge is in at most one of the two make_A lists of a view: cj == ci only for a self loop, whose cj is -1)
                const unsigned long long bit = 1ull << ka;
                if (mAj & bit)
                    sAdjA[row][da++] = (unsigned)ka | ((unsigned)(ki + 1) << 8);
                else
                    sAdjA[row][da++] = (unsigned)ka | ((unsigned)(kj + 1) << 8) | ADJ_NEG;
                ma &= ma - 1;
            }
            if (mh) {
                const unsigned long long bit = 1ull << kh;
                if (mHs & bit)
                    sAdjH[row][dh++] = (unsigned)kh | ADJ_SELF;
                else if (mHi & bit)
                    sAdjH[row][dh++] = (unsigned)kh | ((unsigned)(hj_ >= 0 ? hj_ + 1 : 0) << 8);
                else
                    sAdjH[row][dh++] = (unsigned)kh | ((unsigned)(hi_ >= 0 ? hi_ + 1 : 0) << 8);
                mh &= mh - 1;
            }
        }
        if (vk) {
            sDeg[0][lane] = da;
            sDeg[1][lane] = dh;
        }
    }
    if (tid < 3) sStatus[tid] = 0;
    __syncthreads();
    stamp[1] = (long long)__builtin_amdgcn_s_memtime();
    E.adjA = sAdjA[vk ? lane : 0];
    E.adjH = sAdjH[vk ? lane : 0];
    E.degA = vk ? sDeg[0][lane] : 0;
    E.degH = vk ? sDeg[1][lane] : 0;
    E.maxA = (int)wv_max((double)E.degA);
    E.maxH = (int)wv_max((double)E.degH);
    E.Hrow = sH[wave][vk ? lane : 0];
    double r[3] = {0, 0, 0};
    auto residual = [&]() {  // K1 on lane k (ral/l1_irls.cpp:109-127,498-532)
        if (ek) {
            double4 qj = sQ[ey];
            qj.w = -qj.w;
            const double4 d = qmul(qj, qmul(qq, sQ[ex]));
            const double s2 = sqrt(d.x * d.x + d.y * d.y + d.z * d.z);
            double th = 2.0 * atan2(s2, d.w);
            if (th < -W_PI)
                th += 2.0 * W_PI;
            else if (th >= W_PI)
                th -= 2.0 * W_PI;
            const double aux = th / s2;
            r[0] = d.x * aux;
            r[1] = d.y * aux;
            r[2] = d.z * aux;
            if (s2 < W_EPS) r[0] = r[1] = r[2] = 0.0;
        }
    };
    // score = mean ||W row|| before the exp map; wave 0 updates Q (:729-737 / :894-902, :471-492)
    auto apply_step = [&]() -> double {
        double acc = 0.0;
        if (vk) {
            const double x = sW[0][lane], y = sW[1][lane], z = sW[2][lane];
            const double th = sqrt(x * x + y * y + z * z);
            acc = th;
            if (wave == 0) {
                double sn, cs;
                sincos(th / 2.0, &sn, &cs);  // (one argument reduction for both; k_apply_step of solver.hip does the same)
                const double coef = sn / th;
                double4 w = make_double4(x * coef, y * coef, z * coef, cs);
                if (!isfinite(w.x)) w.x = 0.0;
                if (!isfinite(w.y)) w.y = 0.0;
                if (!isfinite(w.z)) w.z = 0.0;
                if (!isfinite(w.w)) w.w = 0.0;
                sQ[lane + f] = qmul(sQ[lane + f], w);
            }
        }
        const double s = wv_sum(acc) / (double)nu;
        __syncthreads();
        return s;
    };
    int status = 0;
    // ---------------- l1ra (ral/l1_irls.cpp:851-912): wave c solves coordinate c ----------------
    double score = HUGE_VAL, change_th = P.change_th;
    int iter = 0, l1_step = 2;
    while (((score >= change_th) || (l1_step < 2)) && (iter < P.l1_max) && status == 0) {
        if (score < change_th) {  // unreachable under the guard above; kept literal (:879-883)
            l1_step *= 4;
            change_th /= 100.0;
        }
        residual();
        double xo = 0.0;
        const int st = sm_l1decode(E, wave == 0 ? r[0] : (wave == 1 ? r[1] : r[2]), l1_step, xo);
        if (lane == 0) sStatus[wave] = st;
        if (vk) sW[wave][lane] = xo;
        __syncthreads();
        status = sStatus[0] | sStatus[1] | sStatus[2];
        if (status) break;
        score = apply_step();
        iter++;
    }
    const int l1_iters = iter;
    const double l1_score = score;
    stamp[2] = (long long)__builtin_amdgcn_s_memtime();
    // ---------------- irls (ral/l1_irls.cpp:559-752): wave c solves right-hand side c ------------
    double d = 1.0;
    score = HUGE_VAL;
    iter = 0;
    while (score > P.change_th && iter < P.irls_max && status == 0) {
        residual();
        const double rc = wave == 0 ? r[0] : (wave == 1 ? r[1] : r[2]);
        // A'D^2A and A'D^2 r with make_A's A: row v on lane v (k ascending)
        double h[SM_MAX_NU];
        double hd = 0.0, b = 0.0;
        if (vk) {
#pragma unroll
            for (int c = 0; c < SM_MAX_NU; c++) E.Hrow[c] = 0.0;
        }
        for (int q0 = 0; q0 < E.maxA; q0 += 4) {  // (four entries per step, as sm_at)
            unsigned e[4];
            double dk[4], rk[4];
#pragma unroll
            for (int u = 0; u < 4; u++) e[u] = q0 + u < E.degA ? E.adjA[q0 + u] : 0u;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                dk[u] = __shfl(d, (int)(e[u] & 63u), 64);
                rk[u] = __shfl(rc, (int)(e[u] & 63u), 64);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (q0 + u < E.degA) {
                    const double s = dk[u] * dk[u];
                    hd += s;
                    const int o = (int)((e[u] >> 8) & 255u);
                    if (o) atomicAdd(&E.Hrow[o - 1], -s);  // ds_add_f64 without return: no read-modify-write round trip per entry; same bits as -= s
                    if (e[u] & ADJ_NEG)
                        b -= s * rk[u];
                    else
                        b += s * rk[u];
                }
            }
        }
        if (vk) E.Hrow[lane] = hd;
#pragma unroll
        for (int c = 0; c < SM_MAX_NU; c++) h[c] = vk ? E.Hrow[c] : 0.0;
        const bool ok = sm_solve(h, b, nu, lane);
        if (lane == 0) sStatus[wave] = ok ? 0 : 1;
        if (vk) sW[wave][lane] = b;
        __syncthreads();
        status = sStatus[0] | sStatus[1] | sStatus[2];
        if (status) break;
        if (ek) {  // E = A W3 - w, weights (:614-727)
            double e0 = 0, e1 = 0, e2c = 0;
            if (E.cj >= 0) {
                e0 += sW[0][E.cj];
                e1 += sW[1][E.cj];
                e2c += sW[2][E.cj];
            }
            if (E.ci >= 0) {
                e0 -= sW[0][E.ci];
                e1 -= sW[1][E.ci];
                e2c -= sW[2][E.ci];
            }
            e0 -= r[0];
            e1 -= r[1];
            e2c -= r[2];
            d = win_weight(P.cost, P.sigma, e0 * e0 + e1 * e1 + e2c * e2c, d);
        }
        score = apply_step();
        iter++;
    }
    if (status) status = IROTAVG_ERR_SOLVER;
    __syncthreads();
    stamp[3] = (long long)__builtin_amdgcn_s_memtime();
    for (int i = tid; i < nv; i += SM_THREADS) Qg[i] = sQ[i];
    if (wave == 0 && ek) weights[lane] = d;
    // The outputs live in pinned host memory; the sequence number goes out last, after a system-scope fence
    // of every thread: a host that sees it sees everything (window_solve polls it instead of waiting for
    // the runtime's completion signal)
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
        stamp[4] = (long long)__builtin_amdgcn_s_memtime();
        for (int q = 0; q < 5; q++) out->stamp[q] = stamp[q];
        out->l1_iters = l1_iters;
        out->irls_iters = iter;
        out->status = status;
        out->l1_score = l1_score;
        out->irls_score = score;
        __hip_atomic_store(&out->seq, P.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---- host side: persistent staging, one H2D / launch / D2H per solve ---------------------------
struct WindowSolver {
    hipStream_t stream = nullptr;
    DevBuf<unsigned char> dev;   // [I | QQ | Q | weights | result]
    unsigned char *host = nullptr;   // pinned, device-visible
    unsigned char *hdev = nullptr;   // the same block as the device sees it
    size_t cap = 0;
    int seq = 0;       // sequence number of the last wave-kernel launch
    bool attr_set = false;
    unsigned char *bhost = nullptr, *bhdev = nullptr;  // pinned block of the batched form (window_solve_batch)
    size_t bcap = 0;
    ~WindowSolver() {
        if (bhost) (void)hipHostFree(bhost);
        if (host) (void)hipHostFree(host);
        if (stream) StreamPool::get().give(stream);
    }
};

static size_t win_lds_bytes(int nv, int ne, int nu) {
    return sizeof(double4) * (size_t)nv +
           sizeof(double) * ((size_t)4 * ne + ne + 3 * nu + (size_t)nu * (nu + 1) + 3 * nu + 12 * (size_t)ne +
                             5 * nu + 16) +
           sizeof(int2) * (size_t)ne + (size_t)ne + 64;
}

bool window_fits_wave(int nv, int f, int ne) {
    const int nu = nv - f;
    return nu >= 1 && nu <= SM_MAX_NU && nv <= WIN_MAX_NV && ne >= 1 && ne <= SM_MAX_NE;
}

bool window_fits(int nv, int f, int ne) {
    const int nu = nv - f;
    return nu >= 1 && nu <= WIN_MAX_NU && nv <= WIN_MAX_NV && ne >= 1 && ne <= WIN_MAX_NE &&
           win_lds_bytes(nv, ne, nu) <= 160 * 1024;
}

int window_solve(WindowSolver &ws, int nv, int f, int ne, const int32_t *I, const double *QQ_aos,
                 double *Q_aos, double *weights, int l1_max, int irls_max, int cost, double sigma,
                 double change_th, int *l1_iters, int *irls_iters, int kernel) {
    if (!window_fits(nv, f, ne)) return IROTAVG_ERR_BAD_ARG;
    if (kernel == 2 && !window_fits_wave(nv, f, ne)) return IROTAVG_ERR_BAD_ARG;
    const bool wave = kernel == 2 || (kernel == 0 && window_fits_wave(nv, f, ne));
    if (!ws.stream) ws.stream = StreamPool::get().take();
    const size_t oI = 0, oQQ = oI + sizeof(int2) * (size_t)WIN_MAX_NE;
    const size_t oQ = oQQ + sizeof(double4) * (size_t)WIN_MAX_NE;
    const size_t oW = oQ + sizeof(double4) * (size_t)WIN_MAX_NV;
    const size_t oR = oW + sizeof(double) * (size_t)WIN_MAX_NE;
    const size_t total = oR + sizeof(WinResult);
    if (ws.cap < total) {
        ws.dev.alloc(total);
        if (ws.host) (void)hipHostFree(ws.host);
        IRH_CHECK(hipHostMalloc((void **)&ws.host, total, hipHostMallocMapped | hipHostMallocCoherent));  // fine-grained: the kernel's stores are visible to the polling host while it runs
        IRH_CHECK(hipHostGetDevicePointer((void **)&ws.hdev, ws.host, 0));
        ws.cap = total;
    }
    std::memcpy(ws.host + oI, I, sizeof(int32_t) * 2 * (size_t)ne);
    std::memcpy(ws.host + oQQ, QQ_aos, sizeof(double) * 4 * (size_t)ne);
    std::memcpy(ws.host + oQ, Q_aos, sizeof(double) * 4 * (size_t)nv);
    WinParams P{nv, f, ne, l1_max, irls_max, cost, change_th, sigma, 0};
    if (wave) {
        P.seq = ++ws.seq == 0 ? ++ws.seq : ws.seq;  // never 0 ...
        reinterpret_cast<WinResult *>(ws.host + oR)->seq = 0;  // ... which is what the host leaves there
        // the wave kernel touches its inputs once and its outputs once: it works directly on the
        // pinned (device-visible) staging block -- launch + synchronise, no copy commands
        hipLaunchKernelGGL((k_window_wave<false>), dim3(1), dim3(SM_THREADS), 0, ws.stream, P,
                           (const WinParams *)nullptr, (size_t)0,
                           (const int2 *)(ws.hdev + oI), (const double4 *)(ws.hdev + oQQ),
                           (double4 *)(ws.hdev + oQ), (double *)(ws.hdev + oW),
                           (WinResult *)(ws.hdev + oR));
        IRH_CHECK(hipGetLastError());
        // completion: poll the sequence number the kernel stores last into the pinned block (the runtime's
        // own wait costs 5-10 us of a 60 us call); after 2 ms, or if the kernel died, the stream is synchronised
        volatile int *seqp = &reinterpret_cast<WinResult *>(ws.host + oR)->seq;
        const double t0 = now_seconds();
        bool seen = false;
        while (!(seen = __atomic_load_n(const_cast<int *>(seqp), __ATOMIC_ACQUIRE) == P.seq)) {
            if (now_seconds() - t0 > 2e-3) break;
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
        if (!seen) IRH_CHECK(hipStreamSynchronize(ws.stream));
    } else {
        IRH_CHECK(hipMemcpyAsync(ws.dev.p, ws.host, oW, hipMemcpyHostToDevice, ws.stream));
        const size_t shm = win_lds_bytes(nv, ne, nv - f);
        if (!ws.attr_set) {
            IRH_CHECK(hipFuncSetAttribute((const void *)k_window_solve,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            ws.attr_set = true;
        }
        hipLaunchKernelGGL(k_window_solve, dim3(1), dim3(WIN_THREADS), shm, ws.stream, P,
                           (const int2 *)(ws.dev.p + oI), (const double4 *)(ws.dev.p + oQQ),
                           (double4 *)(ws.dev.p + oQ), (double *)(ws.dev.p + oW),
                           (WinResult *)(ws.dev.p + oR));
        IRH_CHECK(hipMemcpyAsync(ws.host + oQ, ws.dev.p + oQ, total - oQ, hipMemcpyDeviceToHost, ws.stream));
        IRH_CHECK(hipStreamSynchronize(ws.stream));
    }
    WinResult R;
    std::memcpy(&R, ws.host + oR, sizeof(R));
    static const bool stamps = getenv("IROTAVG_WINDOW_STAMPS") != nullptr;
    if (stamps && wave)  // s_memtime counts at 100 MHz
        std::fprintf(stderr, "[window] nv %d f %d ne %d l1 %d irls %d: load+lists %.2f us, l1ra %.2f us, irls %.2f us, store %.2f us\n", nv,
                     f, ne, R.l1_iters, R.irls_iters, (R.stamp[1] - R.stamp[0]) * 1e-2, (R.stamp[2] - R.stamp[1]) * 1e-2,
                     (R.stamp[3] - R.stamp[2]) * 1e-2, (R.stamp[4] - R.stamp[3]) * 1e-2);
    std::memcpy(Q_aos, ws.host + oQ, sizeof(double) * 4 * (size_t)nv);
    if (weights) std::memcpy(weights, ws.host + oW, sizeof(double) * (size_t)ne);
    if (l1_iters) *l1_iters = R.l1_iters;
    if (irls_iters) *irls_iters = R.irls_iters;
    return R.status;
}

// Several independent window problems in ONE launch (one workgroup each): every problem must fit the
// wave-resident kernel. Layout of the pinned block: nb slots of `stride` bytes [I | QQ | Q | weights | result | params].
int window_solve_batch(WindowSolver &ws, int nb, WinBatchItem *items, int l1_max, int irls_max, int cost, double sigma,
                       double change_th) {
    if (nb <= 0) return IROTAVG_OK;
    for (int b = 0; b < nb; b++)
        if (!window_fits_wave(items[b].nv, items[b].f, items[b].ne)) return IROTAVG_ERR_BAD_ARG;
    if (!ws.stream) ws.stream = StreamPool::get().take();
    const size_t oI = 0, oQQ = oI + sizeof(int2) * (size_t)SM_MAX_NE;
    const size_t oQ = oQQ + sizeof(double4) * (size_t)SM_MAX_NE;
    const size_t oW = oQ + sizeof(double4) * (size_t)WIN_MAX_NV;
    const size_t oR = oW + sizeof(double) * (size_t)SM_MAX_NE;
    const size_t oP = oR + sizeof(WinResult);
    const size_t stride = (oP + sizeof(WinParams) + 255) & ~(size_t)255;
    const size_t total = stride * (size_t)nb;
    if (ws.bcap < total) {
        if (ws.bhost) (void)hipHostFree(ws.bhost);
        ws.bhost = nullptr;
        IRH_CHECK(hipHostMalloc((void **)&ws.bhost, total + total / 2, hipHostMallocMapped | hipHostMallocCoherent));
        IRH_CHECK(hipHostGetDevicePointer((void **)&ws.bhdev, ws.bhost, 0));
        ws.bcap = total + total / 2;
    }
    ws.seq = ++ws.seq == 0 ? ++ws.seq : ws.seq;
    for (int b = 0; b < nb; b++) {
        unsigned char *h = ws.bhost + stride * (size_t)b;
        const WinBatchItem &it = items[b];
        std::memcpy(h + oI, it.I, sizeof(int32_t) * 2 * (size_t)it.ne);
        std::memcpy(h + oQQ, it.QQ_aos, sizeof(double) * 4 * (size_t)it.ne);
        std::memcpy(h + oQ, it.Q_aos, sizeof(double) * 4 * (size_t)it.nv);
        WinParams P{it.nv, it.f, it.ne, l1_max, irls_max, cost, change_th, sigma, ws.seq};
        std::memcpy(h + oP, &P, sizeof(P));
        reinterpret_cast<WinResult *>(h + oR)->seq = 0;
    }
    hipLaunchKernelGGL((k_window_wave<true>), dim3(nb), dim3(SM_THREADS), 0, ws.stream, WinParams{},
                       (const WinParams *)(ws.bhdev + oP), stride, (const int2 *)(ws.bhdev + oI),
                       (const double4 *)(ws.bhdev + oQQ), (double4 *)(ws.bhdev + oQ), (double *)(ws.bhdev + oW),
                       (WinResult *)(ws.bhdev + oR));
    IRH_CHECK(hipGetLastError());
    // completion: every workgroup stores the launch's sequence number last (see window_solve)
    const double t0 = now_seconds();
    bool all = false;
    int next = 0;
    while (!all) {
        while (next < nb && __atomic_load_n(&reinterpret_cast<WinResult *>(ws.bhost + stride * (size_t)next + oR)->seq,
                                            __ATOMIC_ACQUIRE) == ws.seq)
            next++;
        all = next == nb;
        if (!all && now_seconds() - t0 > 5e-3) break;
#if defined(__x86_64__)
        if (!all) __builtin_ia32_pause();
#endif
    }
    if (!all) IRH_CHECK(hipStreamSynchronize(ws.stream));
    int rc = IROTAVG_OK;
    for (int b = 0; b < nb; b++) {
        unsigned char *h = ws.bhost + stride * (size_t)b;
        WinBatchItem &it = items[b];
        WinResult R;
        std::memcpy(&R, h + oR, sizeof(R));
        std::memcpy(it.Q_aos, h + oQ, sizeof(double) * 4 * (size_t)it.nv);
        it.l1_iters = R.l1_iters;
        it.irls_iters = R.irls_iters;
        it.status = R.status;
        if (R.status != IROTAVG_OK && rc == IROTAVG_OK) rc = R.status;
    }
    return rc;
}

WindowSolver *window_solver_new() { return new WindowSolver(); }
void window_solver_delete(WindowSolver *w) { delete w; }

}  // namespace irh

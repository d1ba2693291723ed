// l1pd.hip -- L1RA: the primal-dual interior-point LP of ral/l1_irls.cpp:228-468 (one coordinate
// of min ||A x - y||_1 per call, x0 = 0) and its outer loop, ral/l1_irls.cpp:851-912.
//
// Edge-length vectors (u, Ax, fu1, fu2, lamu1, lamu2, ...) live as planes of `pd`; the ~25
// element-wise statements of one primal-dual iteration are grouped into a handful of streaming
// kernels; A x is an edge-parallel gather, A' y a view-parallel walk over the incident-edge
// slots (no atomics), and H11p dx = w1p (UMFPACK in the reference, :308-319) is solved by the
// same multigrid-PCG as the IRLS step with the matrix values refreshed from sigx under
// make_AtA's boundary rule (:825-843). Control flow (step length, backtracking, stopping) runs on
// the host from fixed-order reductions; the decisions are the reference's, statement for statement, the kernels are
// grouped so that a primal-dual iteration needs two host round trips.
#include <atomic>
#include <thread>

#include "graph.hpp"
#include "kernels.hpp"

namespace irh {

constexpr int kPdSlots = 2;  // parked dense inverses per solver (pdmaxiter of l1ra is 2)
constexpr double kPdSpread = 8.0;  // staleness band of a parked primal-dual inverse (IRLS: 1.1)


enum PdPlane : int {
    P_Y = 0, P_U, P_AX, P_F1, P_F2, P_L1, P_L2, P_SIGX, P_T1, P_T2, P_ADX,
    // the trial point of the back-tracking loop (u, Ax, lamu1, lamu2, fu1, fu2 at step s): accepting it is an exchange
    // of plane numbers, not another pass over the edges
    P_U2, P_AX2, P_L12, P_L22, P_F12, P_F22,
    P_COUNT
};
constexpr int kPdPartSlots = 3;  // partial arrays of reductions that are fetched with ONE host round trip
enum PdnPlane : int { N_X = 0, N_ATV, N_ATDV, N_X0, N_X1, N_X2, N_COUNT };

static void pd_prepare(Graph &g) {
    if (g.pd_ready) return;
    g.pd.alloc((size_t)P_COUNT * g.mpad);
    g.pd.zero(g.stream);
    g.pdn.alloc((size_t)N_COUNT * g.nu);
    g.pdn.zero(g.stream);
    g.pd_part.alloc((size_t)kMaxParts * 4 * kPdPartSlots);
    g.pd_part.zero(g.stream);
    g.pd_ready = true;
}

__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_down(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    return v;
}
// workgroup min/max -> part[0]
template <bool MAX>
__device__ __forceinline__ void block_ext_store(double v, double *part) {
    __shared__ double sm[16];
    v = MAX ? wave_max(v) : wave_min(v);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = sm[0];
        for (int i = 1; i < (int)(blockDim.x >> 6); i++) r = MAX ? fmax(r, sm[i]) : fmin(r, sm[i]);
        part[0] = r;
        part[1] = part[2] = part[3] = 0.0;
    }
    __syncthreads();
}

// workgroup sum of four accumulators -> part[0..3] (fixed order: waves in order)
__device__ __forceinline__ void block_sum4_store(double a0, double a1, double a2, double a3, double *part) {
    __shared__ double sm4[4][16];
    a0 = wave_sum(a0);
    a1 = wave_sum(a1);
    a2 = wave_sum(a2);
    a3 = wave_sum(a3);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
        sm4[0][w] = a0;
        sm4[1][w] = a1;
        sm4[2][w] = a2;
        sm4[3][w] = a3;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        const int nw = blockDim.x >> 6;
        for (int i = 0; i < nw; i++) {
            s0 += sm4[0][i];
            s1 += sm4[1][i];
            s2 += sm4[2][i];
            s3 += sm4[3][i];
        }
        part[0] = s0;
        part[1] = s1;
        part[2] = s2;
        part[3] = s3;
    }
    __syncthreads();
}

// The edge kernels of the primal-dual iteration take TWO edges per thread (round 4: every plane moves as 16 B per lane,
// as in K1; with one edge and 8-byte loads they ran at 0.4 of the HBM rate where K1 reaches 0.77). Planes are padded to
// mpad (a multiple of 64), so the pair of the last edge of an odd m lies inside every plane; its values are never summed
// and what is stored there is never read as an edge.
#define EDGE_LOOP2(k) \
    for (long long k = 2 * ((long long)blockIdx.x * blockDim.x + threadIdx.x); k < m; k += 2 * (long long)gridDim.x * blockDim.x)
__device__ __forceinline__ double2 ld2(const double *p, long long k) { return *reinterpret_cast<const double2 *>(p + k); }
__device__ __forceinline__ void st2(double *p, long long k, double a, double b) { *reinterpret_cast<double2 *>(p + k) = make_double2(a, b); }

// max_k |y - Ax|   (ral/l1_irls.cpp:250-253)
// (Ax == nullptr: x0 = 0, so Ax = A x0 is the zero vector -- l1ra always starts there, ral/l1_irls.cpp:889-892 -- and the
// plane is neither cleared nor read: y - 0.0 and the other statements below give the same bits as with a stored zero)
__global__ __launch_bounds__(kRowBlock) void k_pd_absmax(long long m, const double *__restrict__ y,
                                                      const double *__restrict__ Ax,
                                                      double *__restrict__ part) {
    double v = -HUGE_VAL;
    EDGE_LOOP2(k) {
        const double2 yy = ld2(y, k), ax = Ax ? ld2(Ax, k) : make_double2(0.0, 0.0);
        v = fmax(v, fabs(yy.x - ax.x));
        if (k + 1 < m) v = fmax(v, fabs(yy.y - ax.y));
    }
    block_ext_store<true>(v, part + 4 * blockIdx.x);
}

// :252-259, 262 (operand of A'), 264, 272-276 (tail of rdual)
__global__ __launch_bounds__(kRowBlock) void k_pd_init(long long m, const double *__restrict__ y,
                                                    const double *__restrict__ Ax, double maxabs,
                                                    double *__restrict__ u, double *__restrict__ f1,
                                                    double *__restrict__ f2, double *__restrict__ l1,
                                                    double *__restrict__ l2, double *__restrict__ t,
                                                    double *__restrict__ part,
                                                    const uint8_t *__restrict__ own) {
    double a0 = 0, a1 = 0, a2 = 0;
    EDGE_LOOP2(k) {
        const double2 y2 = ld2(y, k), ax2 = Ax ? ld2(Ax, k) : make_double2(0.0, 0.0);
        double uu[2], g1[2], g2[2], m1[2], m2[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const double ax = h ? ax2.y : ax2.x, yy = h ? y2.y : y2.x;
            uu[h] = fabs(yy - ax) * 0.95 + maxabs * 0.10;
            g1[h] = ax - yy - uu[h];
            g2[h] = -ax + yy - uu[h];
            m1[h] = -(1.0 / g1[h]);
            m2[h] = -(1.0 / g2[h]);
        }
        st2(u, k, uu[0], uu[1]);
        st2(f1, k, g1[0], g1[1]);
        st2(f2, k, g2[0], g2[1]);
        st2(l1, k, m1[0], m1[1]);
        st2(l2, k, m2[0], m2[1]);
        st2(t, k, m1[0] - m2[0], m1[1] - m2[1]);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if (k + h >= m) continue;
            if (own != nullptr && !own[k + h]) continue;  // sharded: a cross-shard edge is summed by one shard only
            a0 += g1[h] * m1[h];
            a1 += g2[h] * m2[h];
            const double rd = 1.0 - m1[h] - m2[h];
            a2 += rd * rd;
        }
    }
    block_sum3_store(a0, a1, a2, part + 4 * blockIdx.x);
}

// :292-305 -- sigx and the operand of A' for the right-hand side: w1p = -(1/tau) A' t1 - A' t2 with t1 = -1/fu1 + 1/fu2,
// t2 = (sig12/sig11) w2 is A'(-(1/tau) t1 - t2), one plane and one gather per entry for k_pd_rhs. On the way, from the
// same four loads, the sum of squares of rcent = [-lamu1.*fu1; -lamu2.*fu2] - 1/tau (:267-270, 450-453) that the
// residual norm of THIS iteration's back-tracking test needs (the reference forms it at the end of the previous
// iteration, from the same fu, lamu and tau; a pass of its own, with its own host round trip, until round 3)
__global__ __launch_bounds__(kRowBlock) void k_pd_sig(long long m, const double *__restrict__ f1,
                                                   const double *__restrict__ f2,
                                                   const double *__restrict__ l1,
                                                   const double *__restrict__ l2, double itau,
                                                   double *__restrict__ sigx, double *__restrict__ t12,
                                                   double *__restrict__ part, const uint8_t *__restrict__ own) {
    double a0 = 0;
    const double c1 = -itau, c2 = -1.0;
    EDGE_LOOP2(k) {
        const double2 g1v = ld2(f1, k), g2v = ld2(f2, k), m1v = ld2(l1, k), m2v = ld2(l2, k);
        double sx[2], tt[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const double g1 = h ? g1v.y : g1v.x, g2 = h ? g2v.y : g2v.x, m1 = h ? m1v.y : m1v.x, m2 = h ? m2v.y : m2v.x;
            const double if1 = 1.0 / g1, if2 = 1.0 / g2;
            const double w2 = -1 - itau * (if1 + if2);
            const double a = m1 / g1, b = m2 / g2;
            const double s1 = -a - b, s2 = a - b;
            sx[h] = s1 - (s2 * s2) / s1;
            const double t1 = -if1 + if2;
            const double t2 = (s2 / s1) * w2;
            tt[h] = c1 * t1 + c2 * t2;
            if (k + h >= m) continue;
            if (own != nullptr && !own[k + h]) continue;  // sharded: a cross-shard edge is summed by one shard only
            const double r1 = -m1 * g1 - itau, r2 = -m2 * g2 - itau;
            a0 += r1 * r1 + r2 * r2;
        }
        st2(sigx, k, sx[0], sx[1]);
        st2(t12, k, tt[0], tt[1]);
    }
    block_sum3_store(a0, 0.0, 0.0, part + 4 * blockIdx.x);
}

// view-parallel A' y: the lane that owns a view walks its incident-edge entries (SELL layout of
// level 0; make_A coefficients: +1 for the j endpoint, -1 for the i endpoint; boundary slots
// only when make_A kept the coefficient).
// The walk is a chain of dependent loads (slot -> edge value); eight entries are in flight per lane (with four the
// kernels were latency-bound: 90-200 us for 0.4 M rows once the direct solver had removed the PCG around them).
__device__ __forceinline__ double at_row(int row, int n, const int *__restrict__ sl_off,
                                         const uint32_t *__restrict__ slot_eid, const int *__restrict__ bptr,
                                         const uint32_t *__restrict__ beid, const uint8_t *__restrict__ bflag,
                                         const double *__restrict__ t) {
    const int sl = row >> 6, lane = row & 63;
    const int o0 = sl_off[sl], w = sl_off[sl + 1] - o0;   // a multiple of 8
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    const v2u *__restrict__ sp = reinterpret_cast<const v2u *>(slot_eid) + (size_t)(o0 / 2) * 64 + lane;
    double s = 0.0;
    for (int k0 = 0; k0 < w; k0 += 8) {
        v2u se[4];
#pragma unroll
        for (int u = 0; u < 4; u++) se[u] = sp[(size_t)(k0 / 2 + u) * 64];
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const unsigned e = (u & 1) ? se[u >> 1].y : se[u >> 1].x;
            const bool ok = e != 0xffffffffu;
            const size_t idx = ok ? (size_t)(e >> 1) : 0;
            double x = t[idx];
            x = (e & 1u) ? x : -x;
            v[u] = ok ? x : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) s += v[u];
    }
    if (row < n) {
        for (int q = bptr[row]; q < bptr[row + 1]; q++) {
            if (!(bflag[q] & BF_IRLS)) continue;
            const uint32_t se = beid[q];
            const double x = t[se >> 1];
            s += (se & 1u) ? x : -x;
        }
    }
    return s;
}

#define PD_ROW_LOOP(nsl)                                       \
    const int ntiles_ = ((nsl) + 3) / 4;                       \
    int t0_, t1_;                                              \
    tile_range(ntiles_, t0_, t1_);                             \
    for (int t_ = t0_; t_ < t1_; t_++)                         \
        if (const int sl_ = t_ * 4 + (threadIdx.x >> 6); sl_ < (nsl))

// out = A' t, partial sum of out^2
__global__ __launch_bounds__(kRowBlock) void k_at_mul(int n, int nsl, const int *__restrict__ sl_off,
                                                      const uint32_t *__restrict__ slot_eid,
                                                      const int *__restrict__ bptr,
                                                      const uint32_t *__restrict__ beid,
                                                      const uint8_t *__restrict__ bflag,
                                                      const double *__restrict__ t,
                                                      double *__restrict__ out,
                                                      double *__restrict__ part) {
    double acc = 0.0;
    PD_ROW_LOOP(nsl) {
        const int row = sl_ * 64 + (threadIdx.x & 63);
        const double s = at_row(row, n, sl_off, slot_eid, bptr, beid, bflag, t);
        if (row < n) {
            out[row] = s;
            acc += s * s;
        }
    }
    block_sum3_store(acc, 0.0, 0.0, part + 4 * blockIdx.x);
}

// The same product with the entries of a slice dealt to the FOUR waves of the workgroup (round 4). k_at_mul keeps one
// lane per row and walks its ~40 entries in batches of eight behind dependent loads (slot id -> edge value): with ~390
// workgroups that is six waves per CU and five serial round trips of ~2 us, 41 us under l1ra's three chains for 32 MB.
// Here wave w takes the entry pairs w, w + 4, ... of every row of the slice -- all its slot ids are requested at once, then
// all its edge values --, the four partial sums meet in LDS ((p0 + p1) + p2) + p3, then the boundary slots. A workgroup
// walks a contiguous range of slices (at most kMaxParts workgroups: one partial row each for the sum of squares). The
// order of the sum differs from at_row's; it is fixed.
__global__ __launch_bounds__(kRowBlock) void k_at_mul4(int n, int nsl, const int *__restrict__ sl_off,
                                                       const uint32_t *__restrict__ slot_eid,
                                                       const int *__restrict__ bptr,
                                                       const uint32_t *__restrict__ beid,
                                                       const uint8_t *__restrict__ bflag,
                                                       const double *__restrict__ t,
                                                       double *__restrict__ out,
                                                       double *__restrict__ part) {
    __shared__ double sp4[4][64];
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int s0, s1;
    tile_range(nsl, s0, s1);
    double acc = 0.0;
    for (int sl = s0; sl < s1; sl++) {
        const int row = sl * 64 + lane;
        const int o0 = sl_off[sl], np = (sl_off[sl + 1] - o0) / 2;
        const v2u *__restrict__ sp = reinterpret_cast<const v2u *>(slot_eid) + (size_t)(o0 / 2) * 64 + lane;
        double s = 0.0;
        constexpr int PB = 6;  // pairs of this wave per batch (rows of up to 48 entries in one batch)
        for (int p0 = wave; p0 < np; p0 += 4 * PB) {
            v2u se[PB];
            double v[2 * PB];
#pragma unroll
            for (int u = 0; u < PB; u++) {
                const int p = p0 + 4 * u;
                se[u] = p < np ? sp[(size_t)p * 64] : v2u{0xffffffffu, 0xffffffffu};
            }
#pragma unroll
            for (int u = 0; u < 2 * PB; u++) {
                const unsigned e = (u & 1) ? se[u >> 1].y : se[u >> 1].x;
                const bool ok = e != 0xffffffffu;
                double x = t[ok ? (size_t)(e >> 1) : 0];
                x = (e & 1u) ? x : -x;
                v[u] = ok ? x : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 2 * PB; u++) s += v[u];
        }
        sp4[wave][lane] = s;
        __syncthreads();
        if (wave == 0 && row < n) {
            double tot = ((sp4[0][lane] + sp4[1][lane]) + sp4[2][lane]) + sp4[3][lane];
            for (int q = bptr[row]; q < bptr[row + 1]; q++) {
                if (!(bflag[q] & BF_IRLS)) continue;
                const uint32_t se = beid[q];
                const double x = t[se >> 1];
                tot += (se & 1u) ? x : -x;
            }
            out[row] = tot;
            acc += tot * tot;
        }
        __syncthreads();
    }
    block_sum3_store(acc, 0.0, 0.0, part + 4 * blockIdx.x);
}

// rhs = w1p = -(1/tau) A' t1 - A' t2 = A' t12  (:300-306; t12 from k_pd_sig), stored in component 0 of the solver's rhs
__global__ __launch_bounds__(kRowBlock) void k_pd_rhs(int n, int nsl, const int *__restrict__ sl_off,
                                                      const uint32_t *__restrict__ slot_eid,
                                                      const int *__restrict__ bptr,
                                                      const uint32_t *__restrict__ beid,
                                                      const uint8_t *__restrict__ bflag,
                                                      const double *__restrict__ t12, double4 *__restrict__ rhs) {
    PD_ROW_LOOP(nsl) {
        const int row = sl_ * 64 + (threadIdx.x & 63);
        const double w1p = at_row(row, n, sl_off, slot_eid, bptr, beid, bflag, t12);
        if (row < n) rhs[row] = make_double4(w1p, 0.0, 0.0, 0.0);
    }
}

// :324-345 -- du, dlamu1, dlamu2 of one edge from A dx and the iterate. k_pd_dir needs them for the step bounds,
// k_pd_trial_edge for the trial point: both call this (the same statements in the same order), so the three vectors
// are never stored.
__device__ __forceinline__ void pd_direction(double adx, double g1, double g2, double m1, double m2, double itau,
                                             double &d_u, double &d1, double &d2) {
    const double if1 = 1.0 / g1, if2 = 1.0 / g2;
    const double w2 = -1 - itau * (if1 + if2);
    const double a = m1 / g1, b = m2 / g2;
    const double s1 = -a - b, s2 = a - b;
    d_u = (w2 - s2 * adx) / s1;
    d1 = -m1 / g1;
    d1 *= (adx - d_u);
    d1 -= m1;
    d1 -= itau * if1;
    d2 = m2 / g2;
    d2 *= (adx + d_u);
    d2 -= m2;
    d2 -= itau * if2;
}

// :324-381 -- Adx, the operand of A' (Atdv), and the four guarded step bounds. (The make_A coefficients of an edge follow
// from its endpoints and f -- the rule of the builds, build.cpp / gbuild.hip -- so the flag byte is not read.)
__global__ __launch_bounds__(kRowBlock) void k_pd_dir(
    long long m, int f, const int *__restrict__ ei, const int *__restrict__ ej,
    const uint8_t *__restrict__ eflag, const double4 *__restrict__ DX,
    const double *__restrict__ f1, const double *__restrict__ f2, const double *__restrict__ l1,
    const double *__restrict__ l2, double itau, double *__restrict__ Adx, double *__restrict__ t3,
    double *__restrict__ part) {
    (void)eflag;
    double smin = HUGE_VAL;
    EDGE_LOOP2(k) {
        const int2 ii = *reinterpret_cast<const int2 *>(ei + k), jj = *reinterpret_cast<const int2 *>(ej + k);
        const double2 g1v = ld2(f1, k), g2v = ld2(f2, k), m1v = ld2(l1, k), m2v = ld2(l2, k);
        double ad[2], tt[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int i = h ? ii.y : ii.x, j = h ? jj.y : jj.x;
            double adx = 0.0;
            if (j >= f && i != j) adx += DX[j - f].x;   // EF_CJ
            if (j >= f && i >= f) adx -= DX[i - f].x;   // EF_CI
            const double g1 = h ? g1v.y : g1v.x, g2 = h ? g2v.y : g2v.x, m1 = h ? m1v.y : m1v.x, m2 = h ? m2v.y : m2v.x;
            double d_u, d1, d2;
            pd_direction(adx, g1, g2, m1, m2, itau, d_u, d1, d2);
            ad[h] = adx;
            tt[h] = d1 - d2;
            if (k + h >= m) continue;
            // (a direction that is not a number -- the solve in front of this kernel failed -- passes every guard below
            // and fmin drops it: the host would take the full step and back-track 32 times on NaN sums. No step bound is
            // negative, so -1 says it)
            if (!(adx == adx)) smin = -1.0;
            if (d1 < 0) smin = fmin(smin, -m1 / d1);
            if (d2 < 0) smin = fmin(smin, -m2 / d2);
            const double p = adx - d_u;
            if (p > 0) smin = fmin(smin, -g1 / p);
            const double q = -adx - d_u;
            if (q > 0) smin = fmin(smin, -g2 / q);
        }
        st2(Adx, k, ad[0], ad[1]);
        st2(t3, k, tt[0], tt[1]);
    }
    block_ext_store<false>(smin, part + 4 * blockIdx.x);
}

// trial point at step s: sums of squares of the m-tail of rdp (:407-410) and of rcp (:412-416). The point itself is
// stored (planes u2 ... f22) together with the two sums of the surrogate duality gap (:446) it would have: when the
// host accepts the step (:432-442) nothing is left to do on the edges. fu1, fu2 of the iterate are re-formed from
// Ax, y, u by the statement that produced the stored ones (bit for bit), the direction by pd_direction.
__global__ __launch_bounds__(kRowBlock) void k_pd_trial_edge(
    long long m, const double *__restrict__ y, double s, double itau, const double *__restrict__ u,
    const double *__restrict__ Ax, const double *__restrict__ Adx, const double *__restrict__ l1,
    const double *__restrict__ l2, double *__restrict__ u2, double *__restrict__ Ax2, double *__restrict__ l12,
    double *__restrict__ l22, double *__restrict__ f12, double *__restrict__ f22, double *__restrict__ part,
    const uint8_t *__restrict__ own, int nv, const double *__restrict__ Atv, const double *__restrict__ Atdv,
    double *__restrict__ part_v) {
    // the views' share of the trial residual (:407-410: |Atv + s Atdv|^2) rides along -- a launch of its own
    // (k_pd_trial_vert, 9 us for 100k views) until round 4; its partial rows keep their slot and their order
    if (part_v != nullptr) {
        double b0 = 0;
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += gridDim.x * blockDim.x) {
            const double v = Atv[i] + s * Atdv[i];
            b0 += v * v;
        }
        block_sum3_store(b0, 0.0, 0.0, part_v + 4 * blockIdx.x);
    }
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    EDGE_LOOP2(k) {
        const double2 yv = ld2(y, k), uv = ld2(u, k), axv = Ax ? ld2(Ax, k) : make_double2(0.0, 0.0), adv = ld2(Adx, k),
                      m1v = ld2(l1, k), m2v = ld2(l2, k);
        double up[2], axp[2], m1[2], m2[2], g1[2], g2[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const double yy = h ? yv.y : yv.x, u0 = h ? uv.y : uv.x, ax0 = h ? axv.y : axv.x, adx = h ? adv.y : adv.x,
                         m10 = h ? m1v.y : m1v.x, m20 = h ? m2v.y : m2v.x;
            const double g10 = ax0 - yy - u0, g20 = -ax0 + yy - u0;
            double d_u, d1, d2;
            pd_direction(adx, g10, g20, m10, m20, itau, d_u, d1, d2);
            up[h] = u0 + s * d_u;
            axp[h] = ax0 + s * adx;
            m1[h] = m10 + s * d1;
            m2[h] = m20 + s * d2;
            g1[h] = axp[h] - yy - up[h];
            g2[h] = -axp[h] + yy - up[h];
            if (k + h >= m) continue;
            if (own != nullptr && !own[k + h]) continue;
            const double r = 1.0 + (-m1[h] - m2[h]);
            a0 += r * r;
            const double c1 = -m1[h] * g1[h] - itau, c2 = -m2[h] * g2[h] - itau;
            a1 += c1 * c1 + c2 * c2;
            a2 += g1[h] * m1[h];
            a3 += g2[h] * m2[h];
        }
        st2(u2, k, up[0], up[1]);
        st2(Ax2, k, axp[0], axp[1]);
        st2(l12, k, m1[0], m1[1]);
        st2(l22, k, m2[0], m2[1]);
        st2(f12, k, g1[0], g1[1]);
        st2(f22, k, g2[0], g2[1]);
    }
    block_sum4_store(a0, a1, a2, a3, part + 4 * blockIdx.x);
}

// accept the trial point (:432-442): the views (the edges were stored by the trial kernel)
__global__ __launch_bounds__(kRowBlock) void k_pd_commit_vert(int n, double s, double *__restrict__ x,
                                                           const double4 *__restrict__ DX,
                                                           double *__restrict__ Atv,
                                                           const double *__restrict__ Atdv) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        x[i] += s * DX[i].x;
        Atv[i] += s * Atdv[i];
    }
}

__global__ __launch_bounds__(kRowBlock) void k_pack3(int n, const double *__restrict__ x0,
                                                  const double *__restrict__ x1,
                                                  const double *__restrict__ x2,
                                                  double4 *__restrict__ X) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        X[i] = make_double4(x0[i], x1[i], x2[i], 0.0);
}

// ---------------------------------------------------------------------------------------------
static int grid_edges(long long m) {  // (two edges per thread)
    long long gsz = std::min<long long>(((m + 1) / 2 + kRowBlock - 1) / kRowBlock, kMaxParts);
    if (gsz >= 8) gsz &= ~7ll;
    return (int)std::max<long long>(gsz, 1);
}
static int grid_elems(int n) { return grid_edges(n); }
static int grid_rows(const Level &L) {
    long long gsz = std::min<long long>((L.nsl + 3) / 4, kMaxParts);
    if (gsz >= 8) gsz &= ~7ll;
    return (int)std::max<long long>(gsz, 1);
}

// Reductions come back as per-workgroup partials (rows of 4 doubles) and are summed on the host in fixed order. A
// graph has kPdPartSlots partial arrays, so that kernels whose results are needed at the same decision are fetched with
// ONE host round trip: pd_publish sends them to the pinned block, the caller waits once, pd_sum / pd_ext read it.
static double *pd_part_slot(Graph &g, int slot) { return g.pd_part.p + (size_t)slot * 4 * kMaxParts; }
static double *pd_host_slot(Graph &g, int slot) {  // the pinned block has room behind the PCG's own staging
    static_assert(4096 + (kPdPartSlots - 1) * 4 * kMaxParts <= 8192, "pinned block: [8192] is the sequence number");
    return slot == 0 ? g.h_part() : g.hpin + 4096 + (size_t)(slot - 1) * 4 * kMaxParts;
}
// the partial arrays of up to three slots go to the pinned block by one publishing kernel (solver.hip)
static void pd_publish(Graph &g, std::initializer_list<std::pair<int, int>> slot_nparts) {
    PubPart parts[3];
    int k = 0;
    for (const auto &sn : slot_nparts) parts[k++] = PubPart{pd_part_slot(g, sn.first), pd_host_slot(g, sn.first), 4 * sn.second};
    publish_parts(g, parts, k);
}
static void pd_sum(Graph &g, int slot, int nparts, double out[4]) {
    const double *h = pd_host_slot(g, slot);
    out[0] = out[1] = out[2] = out[3] = 0.0;
    for (int b = 0; b < nparts; b++)
        for (int c = 0; c < 4; c++) out[c] += h[4 * (size_t)b + c];
}
static double pd_ext(Graph &g, int slot, int nparts, bool is_max) {
    const double *h = pd_host_slot(g, slot);
    double r = h[0];
    for (int b = 1; b < nparts; b++) r = is_max ? std::max(r, h[4 * (size_t)b]) : std::min(r, h[4 * (size_t)b]);
    return r;
}

// One coordinate of ral/l1_irls.cpp:228-468 over a GROUP of solvers: one member on a single GPU, one
// member per local shard in a sharded run (dist.hip). Every member runs the same kernels on its own
// edges and views; sums over edges count a cross-shard edge once (PdMember::eown), sums over views
// run over owned views; `combine` adds what other processes hold. y: device pointer per member
// (plane of er, or P_Y). Result in pdn plane `xplane` of every member (owned views).
//
// Host round trips (round 3): the decisions of the reference's loop need, per primal-dual iteration, the step bound
// (:347-380) and then, per back-tracking trial, the norms of the trial residuals (:407-419). Everything else is a
// consequence of an accepted trial and is produced by the trial kernels themselves (the trial point and its duality
// gap) or by the first kernel of the next iteration (rcent): 2 round trips (and, sharded, 2 combines) per iteration
// with the first trial accepted, where the literal statement order had 5; 2 instead of 4 before the loop.
void pd_prepare_graph(Graph &g) { pd_prepare(g); }

int l1decode_group(PdGroup &G, int pdmaxiter, int xplane, int *stuck) {
    struct SlotGuard {  // whatever PD slot was last live, the IRLS inverse (slot 0) is live on return
        PdGroup &G;
        ~SlotGuard() {
            for (auto &M : G.mem) dense_select_slot(*M.g, 0);
        }
    } slot_guard{G};
    const double PDTOL = 1e-3, alpha = 0.01, beta = 0.5, mu = 10;  // :231-238
    const double mglob = (double)G.m_global;
    if (stuck) *stuck = 0;
    int pmap[P_COUNT];  // plane numbers: the accepted trial point becomes the iterate by exchange
    for (int i = 0; i < P_COUNT; i++) pmap[i] = i;
    auto pl = [&pmap](Graph &g, int i) { return g.pd.p + (size_t)pmap[i] * g.mpad; };
    auto xv = [&](Graph &g) { return g.pdn.p + (size_t)xplane * g.no; };
    auto atv = [](Graph &g) { return g.pdn.p + (size_t)N_ATV * g.no; };
    auto atdv = [](Graph &g) { return g.pdn.p + (size_t)N_ATDV * g.no; };
    auto ge = [](Graph &g) { return grid_edges(g.m); };
    auto gv = [](Graph &g) { return grid_elems(g.no); };
    auto gr = [](Graph &g) { return grid_rows(g.levels[0]); };
    auto sync_all = [&]() {  // every member's published partials have arrived
        for (auto &M : G.mem) wait_published(*M.g);
    };
    // fixed-order sum over the members' partial arrays (members in order) -- after pd_publish + sync_all
    auto sum_members = [&](int slot, auto nparts_of, double out[4]) {
        out[0] = out[1] = out[2] = out[3] = 0.0;
        for (auto &M : G.mem) {
            double t[4];
            pd_sum(*M.g, slot, nparts_of(*M.g), t);
            for (int c = 0; c < 4; c++) out[c] += t[c];
        }
    };
    auto ext_members = [&](int slot, auto nparts_of, bool is_max) {
        double r = is_max ? -HUGE_VAL : HUGE_VAL;
        for (auto &M : G.mem) {
            const double t = pd_ext(*M.g, slot, nparts_of(*M.g), is_max);
            r = is_max ? std::max(r, t) : std::min(r, t);
            if (t != t) r = t;  // NaN must surface (breakdown)
        }
        if (G.combine) G.combine(&r, 1, is_max ? 2 : 1);
        return r;
    };
    for (auto &M : G.mem) {
        Graph &g = *M.g;
        hipStream_t st = g.stream;
        IRH_CHECK(hipMemsetAsync(xv(g), 0, sizeof(double) * (size_t)g.no, st));           // x0 = 0
        // Ax = A x0 = 0: never stored (the kernels of the first iteration take a null plane for it)
        hipLaunchKernelGGL(k_pd_absmax, dim3(ge(g)), dim3(kRowBlock), 0, st, (long long)g.m, M.y, (const double *)nullptr,
                           pd_part_slot(g, 0));
        pd_publish(g, {{0, ge(g)}});
    }
    sync_all();
    const double maxabs = ext_members(0, ge, true);
    static const bool at_classic = getenv("IROTAVG_AT_MUL_CLASSIC") != nullptr;  // A/B: one lane per row (k_at_mul)
    auto gr4 = [](Graph &g) {  // k_at_mul4: a workgroup per slice, at most kMaxParts of them
        long long gsz = std::min<long long>(g.levels[0].nsl, kMaxParts);
        if (gsz >= 8) gsz &= ~7ll;
        return (int)std::max<long long>(gsz, 1);
    };
    auto gat = [&](Graph &g) { return at_classic ? gr(g) : gr4(g); };  // partial rows of the A' t product
    auto at_mul = [&](int tplane, bool into_atdv, int slot) {
        for (auto &M : G.mem) {
            Graph &g = *M.g;
            Level &L0 = g.levels[0];
            if (at_classic)
                hipLaunchKernelGGL(k_at_mul, dim3(gr(g)), dim3(kRowBlock), 0, g.stream, g.no, L0.nsl, L0.sl_off.p,
                                   g.slot_eid.p, g.bptr.p, g.beid.p, g.bflag.p, pl(g, tplane),
                                   into_atdv ? atdv(g) : atv(g), pd_part_slot(g, slot));
            else
                hipLaunchKernelGGL(k_at_mul4, dim3(gr4(g)), dim3(kRowBlock), 0, g.stream, g.no, L0.nsl, L0.sl_off.p,
                                   g.slot_eid.p, g.bptr.p, g.beid.p, g.bflag.p, pl(g, tplane),
                                   into_atdv ? atdv(g) : atv(g), pd_part_slot(g, slot));
        }
    };
    for (auto &M : G.mem) {
        Graph &g = *M.g;
        hipLaunchKernelGGL(k_pd_init, dim3(ge(g)), dim3(kRowBlock), 0, g.stream, (long long)g.m, M.y, (const double *)nullptr,
                           maxabs, pl(g, P_U), pl(g, P_F1), pl(g, P_F2), pl(g, P_L1), pl(g, P_L2), pl(g, P_T1),
                           pd_part_slot(g, 0), M.eown);
    }
    at_mul(P_T1, false, 1);  // Atv = A'(lamu1 - lamu2) (:262) and its sum of squares
    for (auto &M : G.mem) pd_publish(*M.g, {{0, ge(*M.g)}, {1, gat(*M.g)}});
    sync_all();
    double s4[4], v4[4];
    sum_members(0, ge, s4);
    sum_members(1, gat, v4);
    s4[3] = v4[0];
    if (G.combine) G.combine(s4, 4, 0);
    double sdg = -(s4[0] + s4[1]);      // :264
    double tau = mu * 2 * mglob / sdg;  // :265
    // resnorm^2 (:278-281, 455-458) = res2 + |rcent|^2; the rcent part arrives with the first fetch of the iteration
    // that uses the norm (k_pd_sig)
    double res2 = s4[3] + s4[2];

    int pditer = 0;
    bool ax_zero = true;  // Ax of the iterate is still A x0 = 0 (no plane holds it)
    bool done = (sdg < PDTOL) || (pditer >= pdmaxiter);  // :284
    while (!done) {
        pditer++;
        const double itau = 1.0 / tau;
        for (auto &M : G.mem) {
            Graph &g = *M.g;
            Level &L0 = g.levels[0];
            hipStream_t st = g.stream;
            hipLaunchKernelGGL(k_pd_sig, dim3(ge(g)), dim3(kRowBlock), 0, st, (long long)g.m, pl(g, P_F1),
                               pl(g, P_F2), pl(g, P_L1), pl(g, P_L2), itau, pl(g, P_SIGX), pl(g, P_T2), pd_part_slot(g, 1),
                               M.eown);
            // inverse of the same PD iteration of the previous outer iteration, if still close enough
            // (measured at 100k/2M: the entry ratios against (p, t-1) span 2-20x in the first outer
            // iterations and <1.5x from the ~7th on; accepting up to kPdSpread costs ~1 PCG iteration
            // per solve and saves most inversions of a long l1ra run)
            if (!g.bcr_B) dense_select_slot(g, std::min(pditer - 1, kPdSlots - 1));
            {
                const double keep = g.stale_spread;
                g.stale_spread = kPdSpread;
                g.pd_rhs_src = pl(g, P_T2);  // the windowed assembly forms the right-hand side A' t12 in the same walk
                g.pd_rhs_done = false;
                assemble(g, 1, pl(g, P_SIGX), g.opt.dense_always_refresh == 1);
                g.pd_rhs_src = nullptr;
                g.stale_spread = keep;
            }
            if (!g.pd_rhs_done)
                hipLaunchKernelGGL(k_pd_rhs, dim3(gr(g)), dim3(kRowBlock), 0, st, g.no, L0.nsl, L0.sl_off.p,
                                   g.slot_eid.p, g.bptr.p, g.beid.p, g.bflag.p, pl(g, P_T2), L0.b.p);
        }
        // the primal-dual Hessians (weights 1/f^2 spread over decades) want less over-correction than
        // the IRLS systems of a band graph: 1.6 measured best on both topologies
        // ... and a stronger Jacobi damping (round 3, 100k/2M, iterations per Hessian solve at omega 0.7 / 0.9:
        // band-only 60.3 / 56.8, 2 % loop edges 28.6 / 26.3; kc 1.0 .. 2.8 scanned again: 1.6 stays)
        std::vector<double> kc_keep, om_keep;
        for (auto &M : G.mem) {
            kc_keep.push_back(M.g->opt.mg_kc);
            om_keep.push_back(M.g->opt.mg_omega);
            if (M.g->kc_auto) {
                M.g->opt.mg_kc = std::min(M.g->opt.mg_kc, 1.6);
                if (M.g->opt.mg_omega == 0.7) M.g->opt.mg_omega = 0.9;  // only the library's default is replaced
            }
        }
        double s = 0.0;
        for (int attempt = 0;; attempt++) {
            int rc = G.solve();  // dx in X component 0 of every member (owned views)
            if (rc != IROTAVG_OK) {
                for (size_t q = 0; q < G.mem.size(); q++) {
                    G.mem[q].g->opt.mg_kc = kc_keep[q];
                    G.mem[q].g->opt.mg_omega = om_keep[q];
                }
                return rc == IROTAVG_ERR_NOT_CONVERGED ? rc : IROTAVG_ERR_SOLVER;
            }
            if (G.halo_x) G.halo_x();  // A dx needs dx of the ghost views
            for (auto &M : G.mem) {
                Graph &g = *M.g;
                hipLaunchKernelGGL(k_pd_dir, dim3(ge(g)), dim3(kRowBlock), 0, g.stream, (long long)g.m, g.f, g.ei.p,
                                   g.ej.p, g.eflag.p, g.X.p, pl(g, P_F1), pl(g, P_F2), pl(g, P_L1), pl(g, P_L2), itau,
                                   pl(g, P_ADX), pl(g, P_T1), pd_part_slot(g, 0));
                pd_publish(g, {{0, ge(g)}, {1, ge(g)}});
            }
            at_mul(P_T1, true, 2);  // Atdv (:383); its sum of squares is not used
            sync_all();
            s = std::fmin(1.0, ext_members(0, ge, false));  // :347-380
            // a direct solve whose single-launch upper reduction gave up on a wait (another process on the device held
            // its workgroups back) poisons its solution: once more, level by level -- nothing but derived planes was
            // written since the solve
            bool again = false;
            if ((!(s == s) || s < 0.0) && attempt == 0)
                for (auto &M : G.mem) again = bcr_up_failed(*M.g) || again;
            if (!again) break;
        }
        for (size_t q = 0; q < G.mem.size(); q++) {
            G.mem[q].g->opt.mg_kc = kc_keep[q];
            G.mem[q].g->opt.mg_omega = om_keep[q];
        }
        if (!(s == s) || s < 0.0) return IROTAVG_ERR_SOLVER;  // (k_pd_dir: -1 = the direction is not a number)
        s *= 0.99;  // :381
        double rc4[4];
        sum_members(1, ge, rc4);  // |rcent|^2 at (fu, lamu, tau) of this iteration; combined with the first trial's sums
        // backtracking (:384-429)
        bool suffdec = false;
        int backiter = 0;
        double s_acc = s, rdp2 = 0.0, resnorm = 0.0, sdg_trial = 0.0;
        while (!suffdec) {
            for (auto &M : G.mem) {
                Graph &g = *M.g;
                hipLaunchKernelGGL(k_pd_trial_edge, dim3(ge(g)), dim3(kRowBlock), 0, g.stream, (long long)g.m, M.y, s,
                                   itau, pl(g, P_U), ax_zero ? (const double *)nullptr : (const double *)pl(g, P_AX), pl(g, P_ADX),
                                   pl(g, P_L1), pl(g, P_L2), pl(g, P_U2),
                                   pl(g, P_AX2), pl(g, P_L12), pl(g, P_L22), pl(g, P_F12), pl(g, P_F22),
                                   pd_part_slot(g, 1), M.eown, g.no, atv(g), atdv(g), pd_part_slot(g, 0));
                pd_publish(g, {{0, ge(g)}, {1, ge(g)}});
            }
            sync_all();
            double tv[4], te[4], all[6];
            sum_members(0, ge, tv);
            sum_members(1, ge, te);
            all[0] = tv[0];
            for (int c = 0; c < 4; c++) all[1 + c] = te[c];
            all[5] = rc4[0];
            if (G.combine) G.combine(all, backiter == 0 ? 6 : 5, 0);
            if (backiter == 0) resnorm = std::sqrt(res2 + all[5]);
            rdp2 = all[0] + all[1];
            suffdec = std::sqrt(rdp2 + all[2]) <= (1 - alpha * s) * resnorm;  // :419
            s_acc = s;
            sdg_trial = -(all[3] + all[4]);  // :446 at this trial point
            s *= beta;
            backiter++;
            if (backiter > 32) {  // :423-428 -- return the previous iterate
                if (stuck) *stuck = 1;
                return IROTAVG_OK;
            }
        }
        // :432-442 -- the trial point is the new iterate
        for (auto &M : G.mem) {
            Graph &g = *M.g;
            hipLaunchKernelGGL(k_pd_commit_vert, dim3(gv(g)), dim3(kRowBlock), 0, g.stream, g.no, s_acc, xv(g),
                               g.X.p + g.ng, atv(g), atdv(g));
        }
        ax_zero = false;
        std::swap(pmap[P_U], pmap[P_U2]);
        std::swap(pmap[P_AX], pmap[P_AX2]);
        std::swap(pmap[P_L1], pmap[P_L12]);
        std::swap(pmap[P_L2], pmap[P_L22]);
        std::swap(pmap[P_F1], pmap[P_F12]);
        std::swap(pmap[P_F2], pmap[P_F22]);
        sdg = sdg_trial;               // :446
        tau = mu * 2 * mglob / sdg;    // :448
        res2 = rdp2;                   // :455-458: resnorm^2 = rdp2 + |rcent(tau)|^2, completed by the next k_pd_sig
        done = (sdg < PDTOL) || (pditer >= pdmaxiter);  // :460
    }
    for (auto &M : G.mem) publish_parts(*M.g, nullptr, 0);  // x is complete when this returns
    sync_all();
    return IROTAVG_OK;
}

// single-GPU form: a group of one
static int l1decode_core(Graph &g, const double *y, int pdmaxiter, int xplane, int *stuck) {
    PdGroup G;
    G.mem.push_back(PdMember{&g, nullptr, y});
    G.m_global = g.m;
    G.solve = [&g]() { return g.bcr_B ? bcr_solve(g) : pcg_solve(g); };
    return l1decode_group(G, pdmaxiter, xplane, stuck);
}

int l1decode_pd_dev(Graph &g, int er_plane, const double *y_host, int pdmaxiter, double *x_host,
                    int *stuck, int /*unused*/) {
    pd_prepare(g);
    const double *y;
    if (er_plane >= 0) {
        y = g.er.p + (size_t)er_plane * g.mpad;
    } else {
        double *yp = g.pd.p + (size_t)P_Y * g.mpad;
        IRH_CHECK(hipMemcpyAsync(yp, y_host, sizeof(double) * (size_t)g.m, hipMemcpyHostToDevice,
                                 g.stream));
        y = yp;
    }
    const int rc = l1decode_core(g, y, pdmaxiter, N_X, stuck);
    if (x_host) {
        IRH_CHECK(hipMemcpyAsync(x_host, g.pdn.p + (size_t)N_X * g.nu, sizeof(double) * (size_t)g.nu,
                                 hipMemcpyDeviceToHost, g.stream));
        IRH_CHECK(hipStreamSynchronize(g.stream));
    }
    return rc;
}

// A solver clone shares the handle's static structure (edges, patterns, maps) and owns every
// array a solve writes: one clone per coordinate lets the three primal-dual LPs of an outer
// iteration (ral/l1_irls.cpp:889-892, independent by construction) run concurrently on three
// streams. Every kernel of this latency-bound path leaves most of the chip idle, so the three
// chains overlap almost perfectly.
static std::unique_ptr<Graph> make_solver_clone(Graph &g, hipStream_t stream) {
    std::unique_ptr<Graph> c(new Graph());
    Graph &q = *c;
    q.is_clone = true;
    q.m = g.m; q.n_total = g.n_total; q.mpad = g.mpad;
    q.f = g.f; q.nu = g.nu; q.ng = g.ng; q.no = g.no;
    q.opt = g.opt;
    q.device = g.device;
    q.stream = stream;
    hipStream_t s = q.stream;
    q.ei.alias(g.ei); q.ej.alias(g.ej); q.eflag.alias(g.eflag);
    q.qq.alias(g.qq); q.er.alias(g.er); q.dw.alias(g.dw); q.Q.alias(g.Q);
    q.slot_eid.alias(g.slot_eid); q.bptr.alias(g.bptr); q.beid.alias(g.beid);
    q.bflag.alias(g.bflag); q.bghost.alias(g.bghost);
    q.slot_cs.alias(g.slot_cs); q.tile_e0.alias(g.tile_e0);
    q.asm_windowed = g.asm_windowed; q.asm_l1_fused = g.asm_l1_fused;
    q.bval.alloc_like(g.bval, s);
    q.PG.alloc_like(g.PG, s);
    q.levels.resize(g.levels.size());
    for (size_t l = 0; l < g.levels.size(); l++) {
        Level &A = g.levels[l];
        Level &B = q.levels[l];
        B.n = A.n; B.nnz = A.nnz; B.agg = A.agg; B.nsl = A.nsl; B.sell_len = A.sell_len;
        B.max_near = A.max_near; B.uni_w = A.uni_w;
        B.sl_off.alias(A.sl_off); B.sl_near.alias(A.sl_near); B.col.alias(A.col);
        B.cptr.alias(A.cptr); B.cidx.alias(A.cidx); B.cpos.alias(A.cpos); B.crow.alias(A.crow);
        B.max_row = A.max_row;
        B.val.alloc_like(A.val, s); B.excess.alloc_like(A.excess, s);
        B.diag.alloc_like(A.diag, s); B.idg.alloc_like(A.idg, s);
        B.b.alloc_like(A.b, s); B.x.alloc_like(A.x, s); B.y.alloc_like(A.y, s); B.e.alloc_like(A.e, s);
    }
    q.ndense = g.ndense; q.ndense_pad = g.ndense_pad; q.dense_bw = g.dense_bw; q.additive_top = g.additive_top;
    q.stale_spread = g.stale_spread;
    q.l0_far_entries = g.l0_far_entries;
    q.l1_fused = g.l1_fused;
    q.cg2 = g.cg2;
    q.bcr_B = g.bcr_B;
    q.band0 = g.band0;
    q.bcr_far_i = g.bcr_far_i;
    q.bcr_far_j = g.bcr_far_j;
    q.bcr_far_e = g.bcr_far_e;
    q.dense32 = g.dense32;
    q.b2p.alloc_like(g.b2p, s);
    q.kc_auto = g.kc_auto;
    q.dense_inv.alloc_like(g.dense_inv, s); q.dense_wr.alloc_like(g.dense_wr, s);
    q.dense_wc.alloc_like(g.dense_wc, s); q.dense_ref_diag.alloc_like(g.dense_ref_diag, s);
    q.X.alloc_like(g.X, s); q.P.alloc_like(g.P, s); q.P2.alloc_like(g.P2, s); q.R2.alloc_like(g.R2, s); q.AP.alloc_like(g.AP, s);
    q.part_pq.alloc_like(g.part_pq, s); q.part_rr.alloc_like(g.part_rr, s);
    q.part_rz.alloc_like(g.part_rz, s); q.part_rz2.alloc_like(g.part_rz2, s);
    q.part_score.alloc_like(g.part_score, s);
    alloc_state(q);
    q.stats = g.stats;
    IRH_CHECK(hipStreamSynchronize(s));
    return c;
}

static void destroy_clone_streams(Graph &g) {
    if (g.l1_clones.size() != 3) return;
    StreamPool::Trio t{{nullptr, nullptr, nullptr}};
    for (int c = 0; c < 3; c++) {
        Graph *q = g.l1_clones[c].get();
        if (!q || !q->stream) return;  // (never partially built: see run_l1ra)
        (void)hipStreamSynchronize(q->stream);
        t.s[c] = q->stream;
        q->stream = nullptr;
    }
    StreamPool::get().give_trio(t, g.device);
}

// the three coordinates' solutions (pdn planes N_X0..N_X2, owned views) -> X as double4 rows
void pd_pack_solution(Graph &g) {
    const int n = g.no;
    hipLaunchKernelGGL(k_pack3, dim3(grid_elems(n)), dim3(kRowBlock), 0, g.stream, n,
                       g.pdn.p + (size_t)N_X0 * n, g.pdn.p + (size_t)N_X1 * n, g.pdn.p + (size_t)N_X2 * n,
                       g.X.p + g.ng);
}

void release_l1_clones(Graph &g) {
    destroy_clone_streams(g);
    g.l1_clones.clear();
}

// The host threads of coordinates 1 and 2 (coordinate 0 runs on the caller): started once per l1ra call and handed one
// job per outer iteration, instead of three threads created and joined per outer iteration (their creation delayed the
// last chain by the time two creations take, every iteration). A waiting thread spins, then yields.
namespace {
struct L1Crew {
    std::thread th[2];
    std::atomic<int> go{0}, done{0};
    std::atomic<bool> quit{false};
    std::function<void(int)> job;  // set by the caller before go is raised
    bool started = false;
    static void pause_some(int &spins) {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
        if (++spins > 4000) std::this_thread::yield();
    }
    void start() {
        if (started) return;
        started = true;
        for (int w = 0; w < 2; w++)
            th[w] = std::thread([this, w]() {
                int seen = 0;
                for (;;) {
                    int spins = 0;
                    while (go.load(std::memory_order_acquire) == seen && !quit.load(std::memory_order_acquire)) pause_some(spins);
                    if (go.load(std::memory_order_acquire) == seen) return;  // quit
                    seen++;
                    job(w + 1);
                    done.fetch_add(1, std::memory_order_release);
                }
            });
    }
    // job(0) on the caller, job(1), job(2) on the crew; returns when all three have ended
    void run(const std::function<void(int)> &fn) {
        start();
        job = fn;
        const int target = done.load(std::memory_order_acquire) + 2;
        go.fetch_add(1, std::memory_order_release);
        fn(0);
        int spins = 0;
        while (done.load(std::memory_order_acquire) != target) pause_some(spins);
    }
    ~L1Crew() {
        quit.store(true, std::memory_order_release);
        for (auto &t : th)
            if (t.joinable()) t.join();
    }
};
}  // namespace

// ral/l1_irls.cpp:851-912
int run_l1ra(Graph &g, int max_iters, double change_th, int *iters, double *runtime,
             double *trace) {
    pd_prepare(g);
    L1Crew crew;
    const bool use_crew = !std::getenv("IROTAVG_L1_THREADS_PER_ITERATION");
    const double tic = now_seconds();
    double score = HUGE_VAL;
    int l1_step = 2;  // :868
    int it = 0, rc = IROTAVG_OK;
    const int n = g.no;
    while (((score >= change_th) || (l1_step < 2)) && (it < max_iters)) {  // :877, >=
        if (score < change_th) {  // :879-883 -- unreachable under the guard above; kept literal
            l1_step *= 4;
            change_th /= 100.0;
        }
        launch_edge_residual(g);
        IRH_CHECK(hipStreamSynchronize(g.stream));  // the clones read the residual planes
        // :889-892 -- the three coordinates are independent LPs: one solver clone and one host
        // thread per coordinate, three streams
        if (g.l1_clones.empty()) {
            const StreamPool::Trio trio = StreamPool::get().take_trio();  // three hardware queues (common.hpp)
            std::vector<std::unique_ptr<Graph>> cl;
            try {
                for (int c = 0; c < 3; c++) cl.push_back(make_solver_clone(g, trio.s[c]));
            } catch (...) {
                for (auto &q : cl) q->stream = nullptr;
                for (int c = 0; c < 3; c++) (void)hipStreamSynchronize(trio.s[c]);
                StreamPool::get().give_trio(trio, g.device);
                throw;
            }
            g.l1_clones = std::move(cl);
        }
        int rcs[3] = {IROTAVG_OK, IROTAVG_OK, IROTAVG_OK};
        const std::function<void(int)> chain = [&](int c) {
            try {
                (void)hipSetDevice(g.device);
                DevPool::HeadroomScope hs(g.pool_headroom);  // (a chain's thread allocates for its clone)
                Graph &q = *g.l1_clones[c];
                pd_prepare(q);
                rcs[c] = l1decode_core(q, g.er.p + (size_t)c * g.mpad, l1_step, N_X0, nullptr);
            } catch (...) {
                rcs[c] = IROTAVG_ERR_HIP;
            }
        };
        if (use_crew) {
            crew.run(chain);
        } else {
            std::thread th[3];
            for (int c = 0; c < 3; c++) th[c] = std::thread(chain, c);
            for (int c = 0; c < 3; c++) th[c].join();
        }
        for (int c = 0; c < 3; c++) {
            if (rcs[c] != IROTAVG_OK && rc == IROTAVG_OK) rc = rcs[c];
            Graph &q = *g.l1_clones[c];
            bcr_up_release(q);  // (its solves lie before the last decision it waited for)
            g.stats.pcg_solves += q.stats.pcg_solves;
            g.stats.pcg_iters += q.stats.pcg_iters;
            g.stats.pcg_iters_last = q.stats.pcg_iters_last;
            g.stats.pcg_stagnated += q.stats.pcg_stagnated;
            g.stats.direct_solves += q.stats.direct_solves;
            q.stats.direct_solves = 0;
            g.stats.dense_inversions += q.stats.dense_inversions;
            q.stats.dense_inversions = 0;
            q.stats.pcg_solves = 0;
            q.stats.pcg_iters = 0;
            q.stats.pcg_stagnated = 0;
        }
        if (rc != IROTAVG_OK) break;
        hipLaunchKernelGGL(k_pack3, dim3(grid_elems(n)), dim3(kRowBlock), 0, g.stream, n,
                           g.l1_clones[0]->pdn.p + (size_t)N_X0 * n, g.l1_clones[1]->pdn.p + (size_t)N_X0 * n,
                           g.l1_clones[2]->pdn.p + (size_t)N_X0 * n, g.X.p + g.ng);
        score = apply_step(g);  // :894-902
        if (trace) trace[it] = score;
        it++;
    }
    IRH_CHECK(hipStreamSynchronize(g.stream));
    bcr_up_release(g);
    const double toc = now_seconds();
    *iters = it;
    *runtime = toc - tic;
    g.stats.outer_iters += it;
    g.stats.seconds_l1ra += toc - tic;
    return rc;
}

}  // namespace irh

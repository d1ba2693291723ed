// cgcg.hip -- the PCG iteration as TWO launches (one GPU, band-dominated view-graphs).
//
// What SuiteSparseQR does at ral/l1_irls.cpp:550 and UMFPACK at :147-169 is a multigrid-
// preconditioned CG here (solver.hip). Its iteration was four dependent launches (p-update + SpMV |
// x/r update + restriction + level-1 down-sweep | dense coarse solve | level-1 up-sweep), and every
// one of them is latency-bound at 100k views: ~3 us of launch + a chain of ~2 us memory round trips
// before the first useful byte. This file cuts the iteration to the minimum number of global
// synchronisation points a preconditioned CG with a global coarse solve can have -- two:
//
//   k_cg_update : alpha, beta from the partial dots; p = u + beta p, s = w + beta s, x += alpha p,
//                 r -= alpha s; restriction of r to level 1, level-1 down-sweep, restriction to
//                 level 2 (all tile-local: r_new is formed for the tile's window)
//   k_cg_apply  : u = M^-1 r and w = L u.  The workgroup of a 256-row tile computes ITS OWN slice of
//                 the coarse solve (8 rows of the dense inverse x the level-2 right-hand side, or 8
//                 loads when level 2 is not the dense level), the level-1 up-sweep of the 64 level-1
//                 rows under and around its window, forms u on the window in LDS and runs the SELL
//                 SpMV from it; partial dots r.u and u.w.
//
// The recurrences are Chronopoulos & Gear's (s = L p is carried by recurrence, so the only reduction
// of an iteration -- gamma = r.u, delta = u.w -- is consumed by the NEXT launch and nothing has to
// wait inside a kernel):
//   beta = gamma / gamma_old, alpha = gamma / (delta - beta gamma / alpha_old).
// Same preconditioner, same arithmetic per row and the same fixed-order reductions as the classic
// path; results are bitwise reproducible run to run.
//
// Eligibility (build.cpp, Graph::cg2): one GPU, additive top level, aggregates of 8 on levels 0 and
// 1, no far (loop-closure) entries on level 0, every level-1 neighbour within 8 rows (band graphs),
// at least three levels.
#include <algorithm>
#include <cmath>

#include "graph.hpp"
#include "kernels.hpp"

namespace irh {

constexpr int kCg2GiveUp = 400;  // iterations after which a two-launch solve is handed to the classic recurrences
constexpr int kL1Pre = 8;   // entries of a level-1 row requested up front (a band graph's row has <= 8)

// ---------------------------------------------------------------------------------------------
// k_cg_update. MODE 0: start of a solve (x = 0, r = b: restriction only). MODE 1: first iteration
// (beta = 0). MODE 2: general. Thread tid holds window slots tid and tid + 256; exactly one of them
// is an own row of the tile. r and s ping-pong between two buffers (neighbouring workgroups read the
// old values of these rows as their halo).
// ---------------------------------------------------------------------------------------------
// every thread obtains the fixed-order sums of TWO partial arrays in one pass (one round trip to memory
// and one pair of barriers instead of two: this sits at the head of k_cg_update's dependency chain)
struct Part3x2 {
    double v[6];
};
// the loads (issued with the kernel's other loads, at the top) ...
__device__ __forceinline__ Part3x2 load_parts3x2(const double *pa, const double *pb, int nparts) {
    Part3x2 P;
#pragma unroll
    for (int c = 0; c < 6; c++) P.v[c] = 0.0;
    const int t = threadIdx.x;
    if (t < 256) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int q = t + 256 * h;
            if (q < nparts) {
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    P.v[c] += pa[4 * q + c];
                    P.v[3 + c] += pb[4 * q + c];
                }
            }
        }
    }
    return P;
}
// ... and the fixed-order reduction, where the scalars are needed
__device__ __forceinline__ void reduce_parts3x2(Part3x2 P, double a[3], double b[3]) {
    __shared__ double sm[6][4];
    const int t = threadIdx.x;
    if (t < 256) {
#pragma unroll
        for (int c = 0; c < 6; c++) P.v[c] = wave_sum(P.v[c]);
        if ((t & 63) == 0) {
#pragma unroll
            for (int c = 0; c < 6; c++) sm[c][t >> 6] = P.v[c];
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 3; c++) {
        a[c] = ((sm[c][0] + sm[c][1]) + sm[c][2]) + sm[c][3];
        b[c] = ((sm[3 + c][0] + sm[3 + c][1]) + sm[3 + c][2]) + sm[3 + c][3];
    }
    __syncthreads();
}

// Tile geometry (both kernels): the workgroup owns the slices [s0, s0 + ns), ns <= 4, i.e. the rows
// [r0, r0 + 64 ns), r0 = 64 s0; its window is [r0 - 64, r0 + 64 ns + 64) clipped to the level (<= 384
// rows); level-1 window rows [8 s0 - 8, 8 (s0 + ns) + 8) (<= 48), extended level-1 rows
// [8 s0 - 16, 8 s0 + 48) (64: the window's rows and every neighbour of theirs), level-2 rows
// [s0 - 2, s0 + 6) (8). One tile per workgroup: straight-line code (a tile loop makes the register
// allocator keep the prologue alive across its back edge: 80-280 spilled VGPRs).
template <int MODE>
__global__ __launch_bounds__(kRowBlock, 2) void k_cg_update(
    int n, int nsl, double *__restrict__ scal, int par, const double *__restrict__ part_g,
    const double *__restrict__ part_d, int nparts, double4 *__restrict__ X,
    const double4 *__restrict__ Rin, double4 *__restrict__ Rout, const double4 *__restrict__ Sin,
    double4 *__restrict__ Sout, double4 *__restrict__ P, const double4 *__restrict__ U,
    const double4 *__restrict__ W, LevelView L1, int uw1, double4 *__restrict__ b1, double4 *__restrict__ x1,
    double4 *__restrict__ b2, double4 *__restrict__ x2, const double *__restrict__ idg2, double omega,
    double *__restrict__ part_rr, int *__restrict__ flags, double *__restrict__ b2p, int ndpad,
    double4 *__restrict__ bsave) {
    // MODE 0 starts a solve: it does not look at the done flag of the previous one, it clears it (and the
    // iteration counter), and it leaves a copy of the right-hand side in `bsave` -- a memset and a copy command less
    const int done = MODE == 0 ? 0 : flags[FL_DONE];
    __shared__ double wb[3][kL1Win], wxv[3][kL1Win];
    int s0, ns;
    slice_range(nsl, s0, ns);
    const int tid = threadIdx.x;
    const int r0 = 64 * s0, nown = 64 * ns;
    const int wlo = max(0, r0 - kWinHalo), off = r0 - wlo;
    const int wlen = max(min(n, r0 + nown + kWinHalo) - wlo, 1);  // rows in the window
    // the thread's own row: window slot tid or tid + 256 (at most one of them lies in [off, off + nown))
    const int uo = (tid >= off && tid < off + nown) ? 0 : ((tid + kRowBlock >= off && tid + kRowBlock < off + nown) ? 1 : -1);
    const int io = uo < 0 ? -1 : wlo + tid + uo * kRowBlock;
    const bool has_own = io >= 0 && io < n;
    // everything the tile reads is requested at the top, BEFORE the scalars are reduced from the partials
    double4 vr[2], vw[2], vs[2];  // both window slots (slots beyond the window re-read its last row)
    double w1[2];                 // omega-less 1 / diagonal of the level-1 rows this thread restricts to
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int iw = tid + u * kRowBlock;
        const int i = wlo + min(iw, wlen - 1);
        vr[u] = Rin[i];
        vw[u] = MODE != 0 ? W[i] : make_double4(0, 0, 0, 0);
        vs[u] = MODE == 2 ? Sin[i] : make_double4(0, 0, 0, 0);
        w1[u] = L1.idg[i >> 3];
    }
    const int ioc = has_own ? io : min(wlo, n - 1);
    const double4 vu = MODE != 0 ? U[ioc] : make_double4(0, 0, 0, 0);
    const double4 vx = MODE != 0 ? X[ioc] : make_double4(0, 0, 0, 0);
    const double4 vp = MODE == 2 ? P[ioc] : make_double4(0, 0, 0, 0);
    // lanes 0..8 ns - 1: level-1 row (entries, diagonal) of an own level-1 row
    const int row1 = min(8 * s0 + (tid & 31), L1.n - 1);
    const int l1o0 = uw1 > 0 ? (row1 >> 6) * uw1 : L1.sl_off[row1 >> 6];
    const int l1wd = uw1 > 0 ? uw1 : L1.sl_off[(row1 >> 6) + 1] - l1o0;
    const double l1d = L1.diag[row1];
    const double i2 = idg2[row1 >> 3];
    double l1v[kL1Pre];
    int l1c[kL1Pre];
#pragma unroll
    for (int k = 0; k < kL1Pre; k++) {
        const size_t pos = sell_pos(l1o0, min(k, l1wd - 1), row1 & 63);  // slice widths are >= 8 = kL1Pre
        l1v[k] = L1.val[pos];
        l1c[k] = L1.col[pos];
    }
    Part3x2 parts;
    if (MODE != 0) parts = load_parts3x2(part_g, part_d, nparts);  // in flight with everything above
    if (done) return;
    double al[3] = {0, 0, 0}, be[3] = {0, 0, 0};
    if (MODE != 0) {
        double g3[3], d3[3];
        reduce_parts3x2(parts, g3, d3);
        bool finite = true;
        for (int c = 0; c < 3; c++) {
            const double go = scal[(par ? SC_GAM1 : SC_GAM0) + c], ao = scal[(par ? SC_ALF1 : SC_ALF0) + c];
            const bool chain = MODE == 2 && go > 0.0 && ao > 0.0;
            be[c] = chain ? g3[c] / go : 0.0;
            double den = d3[c] - (chain ? be[c] * g3[c] / ao : 0.0);
            if (chain && !(den > 0.0)) {  // p'Lp <= 0 from the recurrence: rounding on an ill-conditioned
                be[c] = 0.0;              // system -- restart the direction (p = u: delta = u'Lu > 0)
                den = d3[c];
            }
            al[c] = den > 0.0 ? g3[c] / den : 0.0;  // gamma = 0: the column is solved exactly
            finite = finite && isfinite(g3[c]) && isfinite(d3[c]);
        }
        if (blockIdx.x == 0 && tid == 0) {
            for (int c = 0; c < 3; c++) {
                scal[(par ? SC_GAM0 : SC_GAM1) + c] = g3[c];
                scal[(par ? SC_ALF0 : SC_ALF1) + c] = al[c];
            }
            if (!finite) flags[FL_DONE] = 2;
        }
    }
    double a0 = 0, a1 = 0, a2 = 0;
    if (ns > 0) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int iw = tid + u * kRowBlock, i = wlo + iw;
            const bool in = iw < wlen;
            double4 r = vr[u];
            double4 s = vw[u];
            if (MODE != 0) {
                if (MODE == 2) {
                    s.x += be[0] * vs[u].x;
                    s.y += be[1] * vs[u].y;
                    s.z += be[2] * vs[u].z;
                }
                r.x -= al[0] * s.x;
                r.y -= al[1] * s.y;
                r.z -= al[2] * s.z;
            }
            if (!in) r = make_double4(0, 0, 0, 0);
            if (u == uo && has_own) {  // own row: p, s, x, r and ||r||^2
                if (MODE == 0) {
                    X[i] = make_double4(0, 0, 0, 0);
                    if (bsave != nullptr) bsave[i] = r;
                } else {
                    double4 p = vu;
                    if (MODE == 2) {
                        p.x += be[0] * vp.x;
                        p.y += be[1] * vp.y;
                        p.z += be[2] * vp.z;
                    }
                    P[i] = p;
                    Sout[i] = s;  // never in place: neighbouring workgroups read Sin of these rows
                    double4 x = vx;
                    x.x += al[0] * p.x;
                    x.y += al[1] * p.y;
                    x.z += al[2] * p.z;
                    X[i] = x;
                }
                Rout[i] = r;
                a0 += r.x * r.x;
                a1 += r.y * r.y;
                a2 += r.z * r.z;
            }
            // restriction to level 1 (aggregates of 8 consecutive rows = 8 consecutive lanes)
            const double c0 = seg_sum(r.x, 8), c1 = seg_sum(r.y, 8), c2 = seg_sum(r.z, 8);
            if ((iw & 7) == 0 && iw < kWinLen) {
                const int I = i >> 3, Iw = iw >> 3;
                const double w = in ? omega * w1[u] : 0.0;
                wb[0][Iw] = c0;
                wb[1][Iw] = c1;
                wb[2][Iw] = c2;
                wxv[0][Iw] = w * c0;
                wxv[1][Iw] = w * c1;
                wxv[2][Iw] = w * c2;
                if (in && i >= r0 && i < r0 + nown) {
                    b1[I] = make_double4(c0, c1, c2, 0.0);
                    x1[I] = make_double4(w * c0, w * c1, w * c2, 0.0);
                }
            }
        }
    }
    __syncthreads();
    // level-1 residual of the tile's own 8 ns level-1 rows, restricted to level 2
    if (tid < 32) {
        const int W1lo = wlo >> 3;
        const int row = 8 * s0 + tid;
        const bool mine = tid < 8 * ns && row < L1.n;
        double s0_ = 0, s1 = 0, s2 = 0, e0 = 0, e1 = 0, e2 = 0;
        if (mine) {
#pragma unroll
            for (int k = 0; k < kL1Pre; k++) {  // entries requested at the top
                const int ci = min(max(l1c[k] - W1lo, 0), kL1Win - 1);  // padding: v = 0
                s0_ += l1v[k] * wxv[0][ci];
                s1 += l1v[k] * wxv[1][ci];
                s2 += l1v[k] * wxv[2][ci];
            }
            for (int k = kL1Pre; k < l1wd; k++) {
                const size_t pos = sell_pos(l1o0, k, row & 63);
                const double v = L1.val[pos];
                const int ci = min(max(L1.col[pos] - W1lo, 0), kL1Win - 1);
                s0_ += v * wxv[0][ci];
                s1 += v * wxv[1][ci];
                s2 += v * wxv[2][ci];
            }
            const int me = row - W1lo;
            e0 = wb[0][me] - (s0_ + l1d * wxv[0][me]);
            e1 = wb[1][me] - (s1 + l1d * wxv[1][me]);
            e2 = wb[2][me] - (s2 + l1d * wxv[2][me]);
        }
        e0 = seg_sum(e0, 8);
        e1 = seg_sum(e1, 8);
        e2 = seg_sum(e2, 8);
        if ((tid & 7) == 0 && mine) {
            const int J = row >> 3;
            b2[J] = make_double4(e0, e1, e2, 0.0);
            if (b2p != nullptr) {  // planar copy: k_cg_apply stages it into LDS by LDS-DMA
                b2p[J] = e0;
                b2p[ndpad + J] = e1;
                b2p[2 * ndpad + J] = e2;
            }
            const double w = omega * i2;
            x2[J] = make_double4(w * e0, w * e1, w * e2, 0.0);
        }
    }
    block_sum3_store(a0, a1, a2, part_rr + 4 * blockIdx.x);
    if (blockIdx.x == 0 && tid == 0) {
        if (MODE == 0) {
            flags[FL_DONE] = 0;
            flags[FL_ITERS] = 0;
        } else {
            flags[FL_ITERS] += 1;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_cg_apply. YMODE 2: level 2 is the dense level; the workgroup multiplies the 8 rows of the
// explicit inverse it needs with the level-2 right-hand side (staged in LDS by LDS-DMA). YMODE 1:
// level 2 has levels below it; its correction y2 was computed by the generic cycle and is loaded.
// ET / NJ: element type of the inverse the slices read (double, or float for the fp32 copy) and the
// 16-byte steps per lane over one of its rows; NB: batches of 8 entries of the lane's matrix row
// held in registers (0: the row streams through the pipelined loop).
// ---------------------------------------------------------------------------------------------
template <int YMODE, typename ET, int NJ, int NB>
__global__ __launch_bounds__(kRowBlock, 2) void k_cg_apply(
    LevelView L, int uw, LevelView L1, int uw1, const double4 *__restrict__ R, double4 *__restrict__ U,
    double4 *__restrict__ Wv, const double4 *__restrict__ b1, const double4 *__restrict__ x1,
    const double *__restrict__ b2p, const double4 *__restrict__ y2g, const ET *__restrict__ Einv,
    int nd, int ndpad, double dscale_host, int dscale_dev, double omega, double kc, double *__restrict__ part_g,
    double *__restrict__ part_d, const double *__restrict__ part_rr, int np_rr, int first, double rtol2,
    double *__restrict__ scal, int *__restrict__ flags, long long *dbg) {
#define CG_STAMP(k) \
    if (dbg != nullptr && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) dbg[k] = wall_clock64()
    CG_STAMP(0);
    if (dbg != nullptr && threadIdx.x == 0)  // span of the whole grid: first workgroup start, last end
        atomicMin(reinterpret_cast<unsigned long long *>(dbg + 12), (unsigned long long)wall_clock64());
    const int done = flags[FL_DONE];
    // scale of the re-used dense inverse: known to the host, or decided on the device just before this solve
    const double dscale = dscale_dev ? scal[SC_DSCALE] : dscale_host;
    __shared__ double wx[kWinLen], wy[kWinLen], wz[kWinLen];
    __shared__ double ex[3][kL1Ext], ey[3][kL1Win], y2s[3][8];
    extern __shared__ double sb[];  // YMODE 2: 3 * ndpad, the level-2 right-hand side
    int s0, ns;
    slice_range(L.nsl, s0, ns);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    typedef double v2d __attribute__((ext_vector_type(2)));
    typedef int v2i __attribute__((ext_vector_type(2)));
    constexpr int CPL = 16 / (int)sizeof(ET);  // columns of a dense-inverse row per 16-byte lane load (2 or 4)
    typedef ET vE __attribute__((ext_vector_type(CPL)));
    const int r0 = 64 * s0, nown = 64 * ns;
    const int wlo = max(0, r0 - kWinHalo), whi = max(min(L.n, r0 + nown + kWinHalo), wlo + 1);
    const int base2 = s0 - 2, baseE = 8 * s0 - 16, base1 = 8 * s0 - 8;
    const int sl = min(s0 + wv, L.nsl - 1);
    const bool live = wv < ns;
    const int row = sl * 64 + lane;
    // ---- everything the coarse slice and the window need is requested here, before anything waits
    if (YMODE == 2) {
        // level-2 right-hand side (planar copy written by k_cg_update) -> LDS by LDS-DMA: no staging
        // registers, no ds_write pass. A wave moves 64 x 16 B per instruction to a wave-uniform LDS base.
#pragma unroll
        for (int pl = 0; pl < 3; pl++)
#pragma unroll
            for (int it = 0; it < 4; it++) {  // 4 x 512 doubles >= npad (<= 2048)
                const int base = it * 512 + wv * 128;
                if (base + 2 * lane < ndpad)
                    __builtin_amdgcn_global_load_lds(
                        (const void __attribute__((address_space(1))) *)(b2p + (size_t)pl * ndpad + base + 2 * lane),
                        (void __attribute__((address_space(3))) *)(sb + pl * ndpad + base), 16, 0, 0);
            }
    }
    double4 ar[2];  // window slots: residual (slots beyond the window re-read its last row; never used)
    double aw[2];   //               1 / diagonal
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int ic = min(wlo + tid + u * kRowBlock, whi - 1);
        ar[u] = R[ic];
        aw[u] = L.idg[ic];
    }
    vE ea[NJ], eb[NJ];  // YMODE 2: this wave's two rows of the dense inverse, 16 bytes per lane and step
    if (YMODE == 2) {
        // branch-free: a row outside the level reads row 0 (its result is zeroed when y2 is stored),
        // a step beyond npad re-reads the last one (skipped by the fma loop)
        const int J0 = base2 + 2 * wv, J1 = J0 + 1;
        const ET *__restrict__ e0p = Einv + (size_t)(J0 >= 0 && J0 < nd ? J0 : 0) * ndpad;
        const ET *__restrict__ e1p = Einv + (size_t)(J1 >= 0 && J1 < nd ? J1 : 0) * ndpad;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int c = min(CPL * (lane + 64 * j), ndpad - CPL);  // ndpad is a multiple of 64
            ea[j] = *reinterpret_cast<const vE *>(e0p + c);
            eb[j] = *reinterpret_cast<const vE *>(e1p + c);
        }
    }
    double4 x1v = make_double4(0, 0, 0, 0), y2v = make_double4(0, 0, 0, 0);
    if (tid < kL1Ext) {
        const int I = baseE + tid;
        if (I >= 0 && I < L1.n) x1v = x1[I];
    }
    if (YMODE == 1 && tid < 8) {
        const int J = base2 + tid;
        if (J >= 0 && J < nd) y2v = y2g[J];
    }
    const int o0 = uw > 0 ? sl * uw : L.sl_off[sl];
    const int wn = uw > 0 ? uw : L.sl_near[sl];
    const double dg = L.diag[row];
    v2i mc[NB > 0 ? NB : 1][kSellUnroll / 2];
    v2d mv[NB > 0 ? NB : 1][kSellUnroll / 2];
    NearBatch nb;
    {
        const v2i *cs = reinterpret_cast<const v2i *>(L.col) + (size_t)(o0 / 2) * 64 + lane;
        const v2d *vs = reinterpret_cast<const v2d *>(L.val) + (size_t)(o0 / 2) * 64 + lane;
        if (NB > 0) {
#pragma unroll
            for (int bq = 0; bq < (NB > 0 ? NB : 1); bq++)
#pragma unroll
                for (int u = 0; u < kSellUnroll / 2; u++) {
                    mc[bq][u] = v2i{sl * 64, sl * 64};  // a near column, value 0
                    mv[bq][u] = v2d{0.0, 0.0};
                    if (live && bq * kSellUnroll < wn) {
                        mc[bq][u] = __builtin_nontemporal_load(&cs[(size_t)(bq * (kSellUnroll / 2) + u) * 64]);
                        mv[bq][u] = __builtin_nontemporal_load(&vs[(size_t)(bq * (kSellUnroll / 2) + u) * 64]);
                    }
                }
        } else {
            near_prefetch(L, o0, live ? wn : 0, lane, nb);
        }
    }
    if (done) return;
    CG_STAMP(1);
    if (pcg_check(part_rr, np_rr, first, rtol2, scal, flags)) return;  // uniform across the grid
    CG_STAMP(2);
    double g0 = 0, g1 = 0, g2 = 0, d0 = 0, d1 = 0, d2 = 0;
    __syncthreads();  // sb staged (the barrier waits for the LDS-DMA: vmcnt(0))
    CG_STAMP(3);
    // ---- the tile's slice of the coarse solve: y2 on 8 level-2 rows
    if (YMODE == 2) {
        double p0 = 0, p1 = 0, p2 = 0, q0 = 0, q1 = 0, q2 = 0;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int c = CPL * (lane + 64 * j);
            // keep the scheduler from hoisting all the LDS reads above the first fma
            if ((j & 1) == 0) __builtin_amdgcn_sched_barrier(0);
            if (c < ndpad) {
#pragma unroll
                for (int q = 0; q < CPL; q += 2) {
                    const v2d bx = *reinterpret_cast<const v2d *>(sb + c + q);
                    const v2d by = *reinterpret_cast<const v2d *>(sb + ndpad + c + q);
                    const v2d bz = *reinterpret_cast<const v2d *>(sb + 2 * ndpad + c + q);
                    const double a0 = (double)ea[j][q], a1 = (double)ea[j][q + 1];
                    const double c0 = (double)eb[j][q], c1 = (double)eb[j][q + 1];
                    p0 += a0 * bx.x + a1 * bx.y;
                    p1 += a0 * by.x + a1 * by.y;
                    p2 += a0 * bz.x + a1 * bz.y;
                    q0 += c0 * bx.x + c1 * bx.y;
                    q1 += c0 * by.x + c1 * by.y;
                    q2 += c0 * bz.x + c1 * bz.y;
                }
            }
        }
        p0 = wave_sum(p0);
        p1 = wave_sum(p1);
        p2 = wave_sum(p2);
        q0 = wave_sum(q0);
        q1 = wave_sum(q1);
        q2 = wave_sum(q2);
        if (lane == 0) {
            const int J0 = base2 + 2 * wv, J1 = J0 + 1;
            const double z0 = J0 >= 0 && J0 < nd ? dscale : 0.0, z1 = J1 >= 0 && J1 < nd ? dscale : 0.0;
            y2s[0][2 * wv] = p0 * z0;
            y2s[1][2 * wv] = p1 * z0;
            y2s[2][2 * wv] = p2 * z0;
            y2s[0][2 * wv + 1] = q0 * z1;
            y2s[1][2 * wv + 1] = q1 * z1;
            y2s[2][2 * wv + 1] = q2 * z1;
        }
    } else if (tid < 8) {
        y2s[0][tid] = y2v.x;
        y2s[1][tid] = y2v.y;
        y2s[2][tid] = y2v.z;
    }
    CG_STAMP(10);
    // ---- now that the dense-inverse registers are free: lanes < 48 request their level-1 row. It is
    // read-only and unaliased, so the compiler would hoist the loads to the top of the kernel (next to
    // the dense rows: spills); passing the base pointers through an empty asm pins them here.
    double l1v[kL1Pre], l1d = 0.0, l1w = 0.0;
    int l1c[kL1Pre], l1o0 = 0, l1wd = 0;
    double4 b1v = make_double4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < kL1Pre; k++) {
        l1v[k] = 0.0;
        l1c[k] = 0;
    }
    if (tid < kL1Win) {
        const int I = base1 + tid;
        if (I >= 0 && I < L1.n) {
            const int s1l = I >> 6, ln = I & 63;
            const double4 *b1p = b1;
            const double *v1p = L1.val;
            const int *c1p = L1.col;
            asm volatile("" : "+v"(b1p), "+v"(v1p), "+v"(c1p) : : "memory");
            b1v = b1p[I];
            l1d = L1.diag[I];
            l1w = omega * L1.idg[I];
            l1o0 = uw1 > 0 ? s1l * uw1 : L1.sl_off[s1l];
            l1wd = uw1 > 0 ? uw1 : L1.sl_off[s1l + 1] - l1o0;
#pragma unroll
            for (int k = 0; k < kL1Pre; k++)
                if (k < l1wd) {
                    const size_t pos = sell_pos(l1o0, k, ln);
                    l1v[k] = v1p[pos];
                    l1c[k] = c1p[pos];
                }
        }
    }
    CG_STAMP(11);
    __syncthreads();
    CG_STAMP(4);
    // ---- level-1 up-sweep, part 1: x1' = x1 + kc P1 y2 on the extended rows
    if (tid < kL1Ext) {
        const int I = baseE + tid;
        double4 v = x1v;
        if (I >= 0 && I < L1.n) {
            const int J = min(max((I >> 3) - base2, 0), 7);
            v.x += kc * y2s[0][J];
            v.y += kc * y2s[1][J];
            v.z += kc * y2s[2][J];
        }
        ex[0][tid] = v.x;
        ex[1][tid] = v.y;
        ex[2][tid] = v.z;
    }
    __syncthreads();
    CG_STAMP(5);
    // ---- part 2: y1 = x1' + omega D1^-1 (b1 - L1 x1') on the window's level-1 rows
    if (tid < kL1Win) {
        const int I = base1 + tid;
        double y0 = 0, y1 = 0, y2 = 0;
        if (I >= 0 && I < L1.n) {
            double t0 = 0, t1 = 0, t2 = 0;
#pragma unroll
            for (int k = 0; k < kL1Pre; k++) {
                const int ci = min(max(l1c[k] - baseE, 0), kL1Ext - 1);  // padding: v = 0
                t0 += l1v[k] * ex[0][ci];
                t1 += l1v[k] * ex[1][ci];
                t2 += l1v[k] * ex[2][ci];
            }
            for (int k = kL1Pre; k < l1wd; k++) {
                const size_t pos = sell_pos(l1o0, k, I & 63);
                const double v = L1.val[pos];
                const int ci = min(max(L1.col[pos] - baseE, 0), kL1Ext - 1);
                t0 += v * ex[0][ci];
                t1 += v * ex[1][ci];
                t2 += v * ex[2][ci];
            }
            const int me = I - baseE;
            const double p0 = ex[0][me], p1 = ex[1][me], p2 = ex[2][me];
            y0 = p0 + l1w * (b1v.x - (t0 + l1d * p0));
            y1 = p1 + l1w * (b1v.y - (t1 + l1d * p1));
            y2 = p2 + l1w * (b1v.z - (t2 + l1d * p2));
        }
        ey[0][tid] = y0;
        ey[1][tid] = y1;
        ey[2][tid] = y2;
    }
    __syncthreads();
    CG_STAMP(6);
    // ---- u = omega D^-1 r + kc P0 y1 on the window; own rows stored, r.u accumulated
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int iw = tid + u * kRowBlock;
        if (iw < kWinLen) {
            const int i = wlo + iw;
            double4 uv = make_double4(0, 0, 0, 0);
            if (i < whi && ns > 0) {
                const double w = omega * aw[u];
                const int I1 = min(max((i >> 3) - base1, 0), kL1Win - 1);
                uv.x = w * ar[u].x + kc * ey[0][I1];
                uv.y = w * ar[u].y + kc * ey[1][I1];
                uv.z = w * ar[u].z + kc * ey[2][I1];
                if (i >= r0 && i < r0 + nown) {  // own rows of this tile
                    U[i] = uv;
                    g0 += ar[u].x * uv.x;
                    g1 += ar[u].y * uv.y;
                    g2 += ar[u].z * uv.z;
                }
            }
            wx[iw] = uv.x;
            wy[iw] = uv.y;
            wz[iw] = uv.z;
        }
    }
    __syncthreads();
    CG_STAMP(7);
    // ---- w = L u for the own rows, u.w accumulated
    if (live) {
        double t0 = 0, t1 = 0, t2 = 0;
        if (NB > 0) {
#pragma unroll
            for (int bq = 0; bq < (NB > 0 ? NB : 1); bq++)
#pragma unroll
                for (int u = 0; u < kSellUnroll / 2; u++) {
                    const int i0 = mc[bq][u].x - wlo, i1 = mc[bq][u].y - wlo;
                    t0 += mv[bq][u].x * wx[i0] + mv[bq][u].y * wx[i1];
                    t1 += mv[bq][u].x * wy[i0] + mv[bq][u].y * wy[i1];
                    t2 += mv[bq][u].x * wz[i0] + mv[bq][u].y * wz[i1];
                }
            if (wn > NB * kSellUnroll) {  // a slice wider than the register-resident part
                double f0, f1, f2;
                NearBatch nb2;
                near_prefetch_at(L, o0, NB * kSellUnroll, wn, lane, nb2);
                near_window_row_from(L, o0, NB * kSellUnroll, wn, lane, wlo, wx, wy, wz, nb2, f0, f1, f2);
                t0 += f0;
                t1 += f1;
                t2 += f2;
            }
        } else {
            near_window_row(L, o0, wn, lane, wlo, wx, wy, wz, nb, t0, t1, t2);
        }
        if (row < L.n) {
            const int ir = row - wlo;
            const double ux = wx[ir], uy = wy[ir], uz = wz[ir];
            t0 += dg * ux;
            t1 += dg * uy;
            t2 += dg * uz;
            Wv[row] = make_double4(t0, t1, t2, 0.0);
            d0 += ux * t0;
            d1 += uy * t1;
            d2 += uz * t2;
        }
    }
    CG_STAMP(8);
    block_sum3_store(g0, g1, g2, part_g + 4 * blockIdx.x);
    block_sum3_store(d0, d1, d2, part_d + 4 * blockIdx.x);
    CG_STAMP(9);
    if (dbg != nullptr && threadIdx.x == 0)
        atomicMax(reinterpret_cast<unsigned long long *>(dbg + 13), (unsigned long long)wall_clock64());
#undef CG_STAMP
}

// ---------------------------------------------------------------------------------------------
// host driver
// ---------------------------------------------------------------------------------------------
namespace {
struct Cg2Bufs {
    double4 *R[2], *S[2], *P, *U, *W;
};
Cg2Bufs cg2_bufs(Graph &g) {
    Level &L0 = g.levels[0];
    // residual: levels[0].b (the right-hand side on entry) <-> R2; s: P2 <-> levels[0].e;
    // u: levels[0].y; w: AP; p: P
    return Cg2Bufs{{L0.b.p, g.R2.p}, {g.P2.p, L0.e.p}, g.P.p, L0.y.p, g.AP.p};
}
}  // namespace

// workgroups of the two kernels: 4 slices each (3 for some, so that the count is a multiple of 8), at
// most kMaxParts partials. IROTAVG_CG2_SLICES=3 deals ~3 slices per workgroup (512 workgroups at 100k
// views: every CU hosts two) -- measured 1 us per iteration SLOWER: a CU's two workgroups then each
// carry the per-tile overhead (8 rows of the dense inverse, the level-2 right-hand side).
int cg2_grid(const Level &L0) {
    static const int per = [] {
        const char *e = std::getenv("IROTAVG_CG2_SLICES");
        return e ? std::min(4, std::max(1, std::atoi(e))) : 4;
    }();
    long long gsz = (L0.nsl + per - 1) / per;
    gsz = (gsz + 7) & ~7ll;
    gsz = std::min<long long>(gsz, kMaxParts);
    return (int)std::max<long long>(gsz, 1);
}

void cg2_launch_update(Graph &g, int mode, int par, int rcur) {
    Level &L0 = g.levels[0];
    Level &L1 = g.levels[1];
    Level &L2 = g.levels[2];
    const Cg2Bufs B = cg2_bufs(g);
    const int grid = cg2_grid(L0);
#define CG_UPD(M)                                                                                          \
    hipLaunchKernelGGL((k_cg_update<M>), dim3(grid), dim3(kRowBlock), 0, g.stream, L0.n, L0.nsl, g.scal.p,   \
                       par, g.part_rz.p, g.part_pq.p, grid, g.X.p, B.R[rcur], B.R[rcur ^ 1], B.S[rcur],      \
                       B.S[rcur ^ 1], B.P, B.U, B.W, view_of(L1), L1.uni_w, L1.b.p, L1.x.p, L2.b.p, L2.x.p,  \
                       L2.idg.p, g.opt.mg_omega, g.part_rr.p, g.flags.p,                                     \
                       g.levels.size() == 3 ? g.b2p.p : (double *)nullptr, g.ndense_pad, L0.x.p)
    if (mode == 0)
        CG_UPD(0);
    else if (mode == 1)
        CG_UPD(1);
    else
        CG_UPD(2);
#undef CG_UPD
}

void cg2_launch_apply(Graph &g, int first, int rcur, double rtol2, long long *dbg = nullptr, bool dev_scale = false) {
    Level &L0 = g.levels[0];
    Level &L1 = g.levels[1];
    Level &L2 = g.levels[2];
    const Cg2Bufs B = cg2_bufs(g);
    const int grid = cg2_grid(L0);
    const int nl = (int)g.levels.size();
    if (nl == 3) {
        const size_t shm = sizeof(double) * 3 * (size_t)g.ndense_pad;
#define CG_APPLY2(ET, EP, NJ, NB)                                                                             \
    hipLaunchKernelGGL((k_cg_apply<2, ET, NJ, NB>), dim3(grid), dim3(kRowBlock), shm, g.stream, view_of(L0),    \
                       L0.uni_w, view_of(L1), L1.uni_w, B.R[rcur], B.U, B.W, L1.b.p, L1.x.p, g.b2p.p,           \
                       (const double4 *)nullptr, EP, g.ndense, g.ndense_pad, g.dense_scale, dev_scale ? 1 : 0, g.opt.mg_omega, \
                       g.opt.mg_kc, g.part_rz.p, g.part_pq.p, g.part_rr.p, grid, first, rtol2, g.scal.p,        \
                       g.flags.p, dbg)
        if (g.dense32) {
            // the fp32 copy of the inverse (cg2_refresh_inv32): NJ = ceil(npad / 256) 16-byte steps per lane;
            // next to it the lane's whole matrix row fits in registers (band graph: 5 batches of 8 entries)
            const float *E = g.dense_inv32.p;
            if (L0.max_near <= 40) {
                if (g.ndense_pad <= 1024)
                    CG_APPLY2(float, E, 4, 5);
                else if (g.ndense_pad <= 1792)
                    CG_APPLY2(float, E, 7, 5);
                else
                    CG_APPLY2(float, E, 8, 5);
            } else {
                CG_APPLY2(float, E, 8, 0);
            }
        } else {  // fp64 slices (default): NJ = ceil(npad / 128)
            const double *E = g.dense_inv.p;
            if (g.ndense_pad <= 1024)
                CG_APPLY2(double, E, 8, 0);
            else if (g.ndense_pad <= 1664)
                CG_APPLY2(double, E, 13, 0);
            else
                CG_APPLY2(double, E, 16, 0);
        }
#undef CG_APPLY2
    } else {
        // level 2 has levels below it: its correction comes from the generic cycle (solver.hip)
        cycle_levels(g, 2);
        hipLaunchKernelGGL((k_cg_apply<1, double, 1, 0>), dim3(grid), dim3(kRowBlock), 0, g.stream, view_of(L0),
                           L0.uni_w, view_of(L1), L1.uni_w, B.R[rcur], B.U, B.W, L1.b.p, L1.x.p,
                           (const double *)nullptr, L2.y.p, (const double *)nullptr, L2.n, 0, 1.0, 0, g.opt.mg_omega,
                           g.opt.mg_kc, g.part_rz.p, g.part_pq.p, g.part_rr.p, grid, first, rtol2, g.scal.p,
                           g.flags.p, dbg);
    }
}

// fp32 copy of the explicit inverse of the dense level for k_cg_apply: the coarse correction is a
// preconditioner component, its rounding (6e-8 relative) does not touch the accuracy of the solve,
// and half the bytes is half the registers a tile slice occupies while it is in flight.
__global__ __launch_bounds__(256) void k_inv_to_f32(long long n, const double *__restrict__ a, float *__restrict__ b) {
    const long long i = 2 * ((long long)blockIdx.x * blockDim.x + threadIdx.x);
    if (i < n) {
        const double2 v = *reinterpret_cast<const double2 *>(a + i);
        *reinterpret_cast<float2 *>(b + i) = make_float2((float)v.x, (float)v.y);
    }
}
static void cg2_refresh_inv32(Graph &g) {
    if (!g.dense32 || g.inv32_epoch == g.dense_epoch) return;
    const long long n = (long long)g.ndense_pad * g.ndense_pad;
    if (g.dense_inv32.n < (size_t)n) g.dense_inv32.alloc((size_t)n);
    hipLaunchKernelGGL(k_inv_to_f32, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, g.stream, n, g.dense_inv.p,
                       g.dense_inv32.p);
    g.inv32_epoch = g.dense_epoch;
}

// PCG on L X = levels[0].b (three columns), two launches per iteration. Same contract as pcg_solve.
int pcg_solve_cg2(Graph &g, const std::function<void()> *tail, bool *tail_ran, bool device_scale) {
    if (tail_ran) *tail_ran = false;
    const double rtol2 = g.opt.pcg_rtol * g.opt.pcg_rtol;
    if (g.levels.size() == 3) cg2_refresh_inv32(g);
    // k_cg_update<0> clears FL_DONE / FL_ITERS (not FL_STALE) and keeps a copy of the right-hand side in
    // levels[0].x (the residual ping-pongs through its buffer): a speculative solve may be handed back
    // (run_irls), and a system on which the Chronopoulos-Gear recurrences stall is solved again by the classic ones
    int rcur = 0;  // which of the two r / s buffers is current
    cg2_launch_update(g, 0, 0, rcur);
    rcur ^= 1;
    int *h_flags = g.h_flags();
    for (int c = 0; c < FL_COUNT; c++) h_flags[c] = 0;
    int it = 0;
    const int check = std::max(1, g.opt.pcg_check_every);
    const int maxit = std::max(1, g.opt.pcg_max_iters);
    auto apply = [&]() { cg2_launch_apply(g, it == 0, rcur, rtol2, nullptr, device_scale); };
    auto update = [&]() {
        cg2_launch_update(g, it == 0 ? 1 : 2, it & 1, rcur);
        rcur ^= 1;
        it++;
    };
    // poll schedule, stagnation rule: as in pcg_solve (solver.hip)
    // with a tail the first poll is placed one iteration past the prediction: a solve that needs one
    // more than the last still comes back in one round trip (surplus launches return at once)
    int chunk = g.stats.pcg_iters_last > 2 ? (int)std::min<int64_t>(g.stats.pcg_iters_last + (tail ? 1 : 0), maxit) : check;
    constexpr int kStallIters = 64;
    const int ax = g.opt.pcg_stall_accept;
    const double accept = ax < 0 ? -1.0 : (ax == 0 ? 1e-6 : std::pow(10.0, -(double)ax));
    double best = HUGE_VAL;
    int best_it = 0;
    bool stagnated = false;
    double *h_scal = g.h_scal();
    bool first_poll = true;
    while (true) {
        for (int c = 0; c < chunk; c++) {
            apply();
            update();
        }
        chunk = std::max(2, check / 2);
        apply();  // its prologue tests the convergence of the last update
        if (first_poll && tail) (*tail)();
        read_back_state(g);
        if (first_poll && tail && tail_ran) *tail_ran = h_flags[FL_DONE] == 1;
        if (first_poll && device_scale && h_flags[FL_DONE] == 0 && h_flags[FL_STALE] != 0) {
            // speculation on a re-used dense inverse that turned out stale, and the solve is slower than
            // predicted: hand it back (run_irls re-inverts and solves again)
            g.stats.pcg_iters += it;
            return IROTAVG_RETRY_STALE;
        }
        first_poll = false;
        if (h_flags[FL_DONE] != 0) break;
        const double cur = std::max(h_scal[SC_RELRES], std::max(h_scal[SC_RELRES + 1], h_scal[SC_RELRES + 2]));
        if (std::getenv("IROTAVG_PCG_TRACE") && (it < 80 || it % 100 < 4))
            std::fprintf(stderr, "[pcg cg2]   it %d relres %.3e %.3e %.3e alpha %.3e %.3e gamma %.3e\n", it, h_scal[SC_RELRES],
                         h_scal[SC_RELRES + 1], h_scal[SC_RELRES + 2], h_scal[SC_ALF0], h_scal[SC_ALF1], h_scal[SC_GAM0]);
        if (cur < 0.5 * best) {
            best = cur;
            best_it = it;
        } else if (it - best_it >= kStallIters && cur <= accept) {
            stagnated = true;
            break;
        }
        if (it >= maxit) break;
        const char *ge = std::getenv("IROTAVG_CG2_GIVEUP");  // tests force the hand-over
        const int giveup = ge ? std::max(1, std::atoi(ge)) : kCg2GiveUp;
        if (it >= giveup) {
            // s = L p and r are carried by recurrences here; on a badly conditioned system their rounding
            // can stall the iteration far above the tolerance. The classic recurrences (one more launch
            // per iteration, q = L p computed) are the robust path: start over with them.
            if (std::getenv("IROTAVG_PCG_TRACE")) std::fprintf(stderr, "[pcg cg2] giving up at %d iterations -> classic\n", it);
            g.stats.pcg_iters += it;
            g.stats.pcg_handed_over += 1;
            // the classic launches take the scale of a re-used inverse from the host copy; in device_scale
            // mode the kernels above read the device's SC_DSCALE, which the host copy lags behind
            if (device_scale && h_scal[SC_DSCALE] > 0.0) g.dense_scale = h_scal[SC_DSCALE];
            IRH_CHECK(hipMemcpyAsync(g.levels[0].b.p, g.levels[0].x.p, sizeof(double4) * (size_t)g.levels[0].n,
                                     hipMemcpyDeviceToDevice, g.stream));
            return pcg_solve_classic(g);
        }
        update();  // not converged: that application is the next iteration's
    }
    g.stats.pcg_solves += 1;
    g.stats.pcg_iters += stagnated ? it : h_flags[FL_ITERS];
    g.stats.pcg_iters_last = stagnated ? it : h_flags[FL_ITERS];
    for (int c = 0; c < 3; c++) g.stats.last_relres[c] = h_scal[SC_RELRES + c];
    if (std::getenv("IROTAVG_PCG_TRACE"))
        std::fprintf(stderr, "[pcg cg2] clone %d iters %d enqueued %d\n", g.is_clone ? 1 : 0, h_flags[FL_ITERS], it);
    if (h_flags[FL_DONE] == 2) return IROTAVG_ERR_SOLVER;
    if (stagnated) {
        g.stats.pcg_stagnated += 1;
        return IROTAVG_OK;
    }
    if (h_flags[FL_DONE] == 0) return IROTAVG_ERR_NOT_CONVERGED;
    return IROTAVG_OK;
}

// kernel timing for bench.py's roofline leg (time_kernel, solver.hip): one launch of each kernel in
// its steady-state form on whatever the buffers hold
void cg2_time_once(Graph &g, int which) {
    if (g.levels.size() == 3) cg2_refresh_inv32(g);
    if (which == 0)
        cg2_launch_apply(g, 0, 0, -1.0);
    else
        cg2_launch_update(g, 2, 0, 0);
}

// development aid: wall-clock stamps (100 MHz) of the phases of k_cg_apply in one mid-grid workgroup;
// out[k] = microseconds from kernel entry to stamp k (IROTAVG_CG2_STAMPS, tools/)
int cg2_phase_stamps(Graph &g, double *out, int n) {
    DevBuf<long long> d;
    d.alloc(16);
    IRH_CHECK(hipMemsetAsync(d.p, 0, sizeof(long long) * 16, g.stream));
    IRH_CHECK(hipMemsetAsync(d.p + 12, 0xff, sizeof(long long), g.stream));  // atomicMin target
    IRH_CHECK(hipMemsetAsync(g.flags.p, 0, sizeof(int) * FL_COUNT, g.stream));
    if (g.levels.size() == 3) cg2_refresh_inv32(g);
    for (int r = 0; r < 3; r++) {
        if (r == 2) {  // the grid-span slots refer to the last launch only
            IRH_CHECK(hipMemsetAsync(d.p + 12, 0xff, sizeof(long long), g.stream));
            IRH_CHECK(hipMemsetAsync(d.p + 13, 0, sizeof(long long), g.stream));
        }
        cg2_launch_apply(g, 0, 0, -1.0, d.p);
    }
    long long h[16];
    IRH_CHECK(hipMemcpyAsync(h, d.p, sizeof(h), hipMemcpyDeviceToHost, g.stream));
    IRH_CHECK(hipStreamSynchronize(g.stream));
    for (int k = 0; k < n && k < 16; k++) out[k] = (double)(h[k] - h[0]) * 0.01;
    if (n > 13) {  // grid span: first start and last end relative to the stamped workgroup's entry
        out[12] = (double)((long long)((unsigned long long)h[12]) - h[0]) * 0.01;
        out[13] = (double)(h[13] - h[0]) * 0.01;
    }
    return IROTAVG_OK;
}

}  // namespace irh

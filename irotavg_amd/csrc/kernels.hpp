// kernels.hpp -- gfx950 device helpers of the rotation-averaging core. Wave = 64 lanes.
//
// Row kernels: one lane owns one matrix row (SELL-64, see common.hpp), 256-thread workgroups =
// 4 slices. The grid is at most kMaxParts workgroups; each takes a CONTIGUOUS chunk of
// 4-slice tiles. Workgroups that share blockIdx % 8 sit on one XCD (observed dispatch order),
// so chunks are dealt such that one XCD's workgroups cover adjacent row ranges: neighbour
// gathers of the band-dominated view-graph then hit that XCD's own L2. Placement only affects
// speed, never results.
//
// Every reduction is evaluated in a fixed order (lane -> wave -> workgroup -> partial array ->
// fixed-order re-reduction in the consumer), so results are bitwise reproducible run to run.
#pragma once
#include "common.hpp"

namespace irh {

struct LevelView {
    int n, nsl, agg;
    const int *sl_off, *sl_near, *col;
    const double *val, *diag, *idg;
};

inline LevelView view_of(const Level &L) {
    return LevelView{L.n, L.nsl, L.agg, L.sl_off.p, L.sl_near.p, L.col.p, L.val.p, L.diag.p, L.idg.p};
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;  // lane 0
}

// sum over aligned groups of `g` consecutive lanes (g a power of two <= 64); every lane gets it
__device__ __forceinline__ double seg_sum(double v, int g) {
    for (int o = 1; o < g; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// workgroup sum of three accumulators -> part[0..2] (written by thread 0). Any blockDim <= 1024.
__device__ __forceinline__ void block_sum3_store(double a0, double a1, double a2, double *part) {
    __shared__ double sm[3][16];
    a0 = wave_sum(a0);
    a1 = wave_sum(a1);
    a2 = wave_sum(a2);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
        sm[0][w] = a0;
        sm[1][w] = a1;
        sm[2][w] = a2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s0 = 0, s1 = 0, s2 = 0;
        const int nw = blockDim.x >> 6;
        for (int i = 0; i < nw; i++) {
            s0 += sm[0][i];
            s1 += sm[1][i];
            s2 += sm[2][i];
        }
        part[0] = s0;
        part[1] = s1;
        part[2] = s2;
        part[3] = 0.0;
    }
    __syncthreads();
}

// every thread of the workgroup obtains the fixed-order sum of the partial array
// (nparts <= kMaxParts = 512 rows of 4 doubles); blockDim >= 256
__device__ __forceinline__ void load_reduced3(const double *part, int nparts, double out[3]) {
    __shared__ double sm[3][4];
    const int t = threadIdx.x;
    if (t < 256) {
        double a0 = 0, a1 = 0, a2 = 0;
        if (t < nparts) {
            a0 = part[4 * t + 0];
            a1 = part[4 * t + 1];
            a2 = part[4 * t + 2];
        }
        if (t + 256 < nparts) {
            a0 += part[4 * (t + 256) + 0];
            a1 += part[4 * (t + 256) + 1];
            a2 += part[4 * (t + 256) + 2];
        }
        a0 = wave_sum(a0);
        a1 = wave_sum(a1);
        a2 = wave_sum(a2);
        if ((t & 63) == 0) {
            sm[0][t >> 6] = a0;
            sm[1][t >> 6] = a1;
            sm[2][t >> 6] = a2;
        }
    }
    __syncthreads();
    out[0] = ((sm[0][0] + sm[0][1]) + sm[0][2]) + sm[0][3];
    out[1] = ((sm[1][0] + sm[1][1]) + sm[1][2]) + sm[1][3];
    out[2] = ((sm[2][0] + sm[2][1]) + sm[2][2]) + sm[2][3];
    __syncthreads();
}

// convergence decision from the ||r||^2 partials of the last PCG update (every workgroup takes
// the same decision; workgroup 0 publishes it)
__device__ __forceinline__ bool pcg_check(const double *part_rr, int nparts, int first,
                                          double rtol2, double *scal, int *flags) {
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
    return !finite || conv;
}

// contiguous tile chunk of this workgroup, XCD-aware (see file header)
__device__ __forceinline__ void tile_range(int ntiles, int &t0, int &t1) {
    const int nb = gridDim.x, b = blockIdx.x;
    int lb = b;
    if ((nb & 7) == 0) lb = (b & 7) * (nb >> 3) + (b >> 3);
    t0 = (int)(((long long)ntiles * lb) / nb);
    t1 = (int)(((long long)ntiles * (lb + 1)) / nb);
}

// the two-launch PCG iteration (cgcg.hip) deals SLICES, not 256-row tiles: workgroup b owns the
// slices [nsl b / G, nsl (b + 1) / G) -- at most four, G >= nsl / 4 -- so that every CU carries the
// same number of rows (391 tiles of four slices on 256 CUs leave the CUs that host two of them with
// twice the bytes: they set the kernel time). Same XCD-aware block order as tile_range.
__device__ __forceinline__ void slice_range(int nsl, int &s0, int &ns) {
    const int nb = gridDim.x, b = blockIdx.x;
    int lb = b;
    if ((nb & 7) == 0) lb = (b & 7) * (nb >> 3) + (b >> 3);
    s0 = (int)(((long long)nsl * lb) / nb);
    ns = (int)(((long long)nsl * (lb + 1)) / nb) - s0;
}

// Off-diagonal part of (L x)_row for the lane's own row. The slice width is a multiple of
// kSellUnroll and uniform across the wave. The loop is software-pipelined by hand TWO batches
// deep: while the gathers of batch k are in flight, the col/val loads of batch k+2 are issued
// and those of batch k+1 are already underway. The matrix stream does not fit the 4 MB L2 of an
// XCD, so every batch is a ~2 us trip to the Infinity Cache / HBM; with only ~6 waves per CU
// (one lane per row) the bytes in flight per wave set the bandwidth (Little's law).
template <bool PROLONG>
__device__ __forceinline__ void row_offdiag_t(const LevelView &L, int row, const double4 *__restrict__ x,
                                              const double4 *__restrict__ xc, int sh, double kc,
                                              double &s0, double &s1, double &s2, int kbeg = 0) {
    constexpr int U = kSellUnroll, H = kSellUnroll / 2;  // a batch = U entries = H pairs
    const int sl = row >> 6, lane = row & 63;
    const int o0 = L.sl_off[sl], w = L.sl_off[sl + 1] - o0;
    // entry pairs: (o0 + 2q) * 64 + 2 lane  ==  int2/double2 index (o0/2 + q) * 64 + lane
    const int2 *__restrict__ c = reinterpret_cast<const int2 *>(L.col) + (size_t)(o0 / 2) * 64 + lane;
    const double2 *__restrict__ v = reinterpret_cast<const double2 *>(L.val) + (size_t)(o0 / 2) * 64 + lane;
    s0 = s1 = s2 = 0.0;
    if (kbeg >= w) return;
    int2 ca[H], cb[H], cn[H];
    double2 va[H], vb[H], vn[H];
#pragma unroll
    for (int u = 0; u < H; u++) {
        ca[u] = c[(size_t)(kbeg / 2 + u) * 64];
        va[u] = v[(size_t)(kbeg / 2 + u) * 64];
    }
    if (kbeg + U < w) {
#pragma unroll
        for (int u = 0; u < H; u++) {
            cb[u] = c[(size_t)((kbeg + U) / 2 + u) * 64];
            vb[u] = v[(size_t)((kbeg + U) / 2 + u) * 64];
        }
    }
    for (int k0 = kbeg; k0 < w; k0 += U) {
        double4 xx[U], xk[U];
#pragma unroll
        for (int u = 0; u < H; u++) {
            xx[2 * u] = x[ca[u].x];
            xx[2 * u + 1] = x[ca[u].y];
            if (PROLONG) {
                xk[2 * u] = xc[ca[u].x >> sh];
                xk[2 * u + 1] = xc[ca[u].y >> sh];
            }
        }
        if (k0 + 2 * U < w) {
#pragma unroll
            for (int u = 0; u < H; u++) {
                cn[u] = c[(size_t)((k0 + 2 * U) / 2 + u) * 64];
                vn[u] = v[(size_t)((k0 + 2 * U) / 2 + u) * 64];
            }
        }
#pragma unroll
        for (int u = 0; u < H; u++) {
            if (PROLONG) {
                s0 += va[u].x * (xx[2 * u].x + kc * xk[2 * u].x) + va[u].y * (xx[2 * u + 1].x + kc * xk[2 * u + 1].x);
                s1 += va[u].x * (xx[2 * u].y + kc * xk[2 * u].y) + va[u].y * (xx[2 * u + 1].y + kc * xk[2 * u + 1].y);
                s2 += va[u].x * (xx[2 * u].z + kc * xk[2 * u].z) + va[u].y * (xx[2 * u + 1].z + kc * xk[2 * u + 1].z);
            } else {
                s0 += va[u].x * xx[2 * u].x + va[u].y * xx[2 * u + 1].x;
                s1 += va[u].x * xx[2 * u].y + va[u].y * xx[2 * u + 1].y;
                s2 += va[u].x * xx[2 * u].z + va[u].y * xx[2 * u + 1].z;
            }
        }
#pragma unroll
        for (int u = 0; u < H; u++) {
            ca[u] = cb[u];
            va[u] = vb[u];
            cb[u] = cn[u];
            vb[u] = vn[u];
        }
    }
}

__device__ __forceinline__ void row_offdiag(const LevelView &L, int row, const double4 *__restrict__ x,
                                            double &s0, double &s1, double &s2) {
    row_offdiag_t<false>(L, row, x, nullptr, 0, 0.0, s0, s1, s2);
}

// same with the prolongated iterate x + kc * xc[col >> sh] gathered on the fly
__device__ __forceinline__ void row_offdiag_prolong(const LevelView &L, int row,
                                                    const double4 *__restrict__ x,
                                                    const double4 *__restrict__ xc, int sh, double kc,
                                                    double &s0, double &s1, double &s2) {
    row_offdiag_t<true>(L, row, x, xc, sh, kc, s0, s1, s2);
}

// Near part of one SELL row (the first `wn` entry-columns of its slice: columns inside the tile
// window): a pure 16 B/lane matrix stream -- the gathered vector comes from the LDS copy of the
// window (wx, wy, wz; index = column - wlo). wn is a multiple of 8: batches of 4 pairs, all 8 loads
// of a batch issued before use and the next batch's loads issued before the current one is consumed.
struct NearBatch {  // the first batch of a row's near entries (4 column pairs, 4 value pairs)
    typedef int v2i __attribute__((ext_vector_type(2)));
    typedef double v2d __attribute__((ext_vector_type(2)));
    v2i c[kSellUnroll / 2];
    v2d v[kSellUnroll / 2];
};
// issue the loads of a row's first near batch (so that they fly during the reductions / the LDS
// window fill that precede the row loop)
// (kbeg: first entry-column of the part of the row to walk, a multiple of 8; 0 = the whole near part)
__device__ __forceinline__ void near_prefetch_at(const LevelView &L, int o0, int kbeg, int wn, int lane, NearBatch &B) {
    constexpr int HB = kSellUnroll / 2;
    const NearBatch::v2i *__restrict__ cs = reinterpret_cast<const NearBatch::v2i *>(L.col) + (size_t)(o0 / 2) * 64 + lane;
    const NearBatch::v2d *__restrict__ vs = reinterpret_cast<const NearBatch::v2d *>(L.val) + (size_t)(o0 / 2) * 64 + lane;
#pragma unroll
    for (int u = 0; u < HB; u++) {
        B.c[u] = NearBatch::v2i{0, 0};
        B.v[u] = NearBatch::v2d{0.0, 0.0};
    }
    if (wn > kbeg) {
#pragma unroll
        for (int u = 0; u < HB; u++) {
            B.c[u] = __builtin_nontemporal_load(&cs[(size_t)(kbeg / 2 + u) * 64]);
            B.v[u] = __builtin_nontemporal_load(&vs[(size_t)(kbeg / 2 + u) * 64]);
        }
    }
}
__device__ __forceinline__ void near_prefetch(const LevelView &L, int o0, int wn, int lane, NearBatch &B) {
    near_prefetch_at(L, o0, 0, wn, lane, B);
}
__device__ __forceinline__ void near_window_row_from(const LevelView &L, int o0, int kbeg, int wn, int lane, int wlo,
                                                     const double *wx, const double *wy, const double *wz,
                                                     const NearBatch &first, double &s0, double &s1, double &s2) {
    constexpr int HB = kSellUnroll / 2;
    typedef NearBatch::v2i v2i;
    typedef NearBatch::v2d v2d;
    const v2i *__restrict__ cs = reinterpret_cast<const v2i *>(L.col) + (size_t)(o0 / 2) * 64 + lane;
    const v2d *__restrict__ vs = reinterpret_cast<const v2d *>(L.val) + (size_t)(o0 / 2) * 64 + lane;
    s0 = s1 = s2 = 0.0;
    v2i cc[HB], cn[HB];
    v2d vv[HB], vn[HB];
#pragma unroll
    for (int u = 0; u < HB; u++) {
        cc[u] = first.c[u];
        vv[u] = first.v[u];
    }
    for (int q0 = kbeg / 2; q0 < wn / 2; q0 += HB) {
        if (q0 + HB < wn / 2) {
#pragma unroll
            for (int u = 0; u < HB; u++) {
                cn[u] = __builtin_nontemporal_load(&cs[(size_t)(q0 + HB + u) * 64]);
                vn[u] = __builtin_nontemporal_load(&vs[(size_t)(q0 + HB + u) * 64]);
            }
        }
#pragma unroll
        for (int u = 0; u < HB; u++) {
            const int i0 = cc[u].x - wlo, i1 = cc[u].y - wlo;
            s0 += vv[u].x * wx[i0] + vv[u].y * wx[i1];
            s1 += vv[u].x * wy[i0] + vv[u].y * wy[i1];
            s2 += vv[u].x * wz[i0] + vv[u].y * wz[i1];
        }
#pragma unroll
        for (int u = 0; u < HB; u++) {
            cc[u] = cn[u];
            vv[u] = vn[u];
        }
    }
}
__device__ __forceinline__ void near_window_row(const LevelView &L, int o0, int wn, int lane, int wlo,
                                                const double *wx, const double *wy, const double *wz,
                                                const NearBatch &first, double &s0, double &s1, double &s2) {
    near_window_row_from(L, o0, 0, wn, lane, wlo, wx, wy, wz, first, s0, s1, s2);
}

// Hamilton product, [x y z w] (ral/l1_irls.cpp:99-105)
__device__ __forceinline__ double4 qmul(const double4 a, const double4 b) {
    double4 r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    return r;
}

// One view's share of K6 (ral/l1_irls.cpp:729-737, exp_map :471-492): returns ||x|| (the view's term of the score, taken
// BEFORE the exp map), Q[idx] <- Q[idx] (x) exp(x) (right-multiply, no renormalisation; every non-finite entry of the
// exponential -> 0, :491). A step that is not finite leaves its rotation alone (the score turns non-finite instead).
__device__ __forceinline__ double step_apply(double x0, double x1, double x2, double4 *__restrict__ Q, int idx, bool write) {
    const double th = sqrt(x0 * x0 + x1 * x1 + x2 * x2);
    double sn, cs;
    sincos(th / 2.0, &sn, &cs);
    const double coef = sn / th;
    double4 w = make_double4(x0 * coef, x1 * coef, x2 * coef, cs);
    if (!isfinite(w.x)) w.x = 0.0;
    if (!isfinite(w.y)) w.y = 0.0;
    if (!isfinite(w.z)) w.z = 0.0;
    if (!isfinite(w.w)) w.w = 0.0;
    const double4 q = qmul(Q[idx], w);
    if (write && isfinite(th)) Q[idx] = q;
    return th;
}

}  // namespace irh

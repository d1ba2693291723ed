// dense.hip -- the coarsest multigrid level as a DENSE explicit inverse.
//
// Many tiny sparse levels are launch/latency bound on a 256-CU GPU (each costs ~5 us per sweep
// whatever its size), while one dense n_c x n_c mat-vec with n_c <= 2048 streams at most 32 MB and
// costs about the same as ONE of them. So the hierarchy stops early (n_c ~ 1-2k rows) and the
// coarse operator E = P'LP is inverted explicitly once per matrix refresh by a blocked, in-place
// Gauss-Jordan sweep (E is SPD: no pivoting), and applied exactly in every V-cycle.
//
// Blocked in-place Gauss-Jordan, block size B = 32, for block step k (D = A_kk):
//   A'_kk = D^-1,  A'_kj = D^-1 A_kj,  A'_ik = -A_ik D^-1,  A'_ij = A_ij - A_ik D^-1 A_kj.
// With Rt = [D^-1 A_k,: with its block k replaced by D^-1] and C = A_:,k this is
//   rows of block k:  A' = Rt;     other rows:  A'_ij = (j in block k ? 0 : A_ij) - C_i Rt_j,
// i.e. a panel (Rt and a copy of C) and a rank-32 update per step. The update kernel of step k also
// produces the panel of step k+1 (look-ahead, k_gj_update_la): the serial 32 x 32 inversion then
// overlaps the bandwidth-bound bulk of the update -- one launch per step instead of two.
//
// Also here: the banded inverse of a coarsest operator without loop closures (k_band_inverse), the
// low-rank repair of an inverse whose operator changed in a few long-range entries (dense_lowrank_repair),
// the staleness tests (dense_is_stale, dense_check_async) and the last resort of a small single-level
// system whose explicit inverse has lost its accuracy (k_chol_solve).
#include <algorithm>
#include <cmath>
#include <vector>

#include "graph.hpp"
#include "kernels.hpp"

namespace irh {

constexpr int GJB = 32;   // block size of the Gauss-Jordan sweep
constexpr int GJT = 64;   // tile edge of the update kernel

// broadcast of lane `lane` (wave-uniform, here a compile-time constant) through SGPRs
__device__ __forceinline__ double readlane_d(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

// In-register Gauss-Jordan inversion of a 32 x 32 SPD block: lane l (and its mirror l + 32) holds
// row l in d[0..31]; pivot rows are broadcast by lane reads. Per pivot and column: two v_readlane
// and ONE fma (the pivot scaling is folded into the lane's multiplier and into a per-lane row
// scale); the reciprocal is v_rcp_f64 + two Newton steps
// instead of an IEEE division. (Splitting a row over lanes l and l + 32 halves the instruction
// count but needs ds_bpermute broadcasts: measured 10 % slower; publishing the scaled pivot row in
// LDS and reading it back as broadcast loads: 75 % slower -- the LDS round trip sits on the
// dependent chain. 13 us for the 32 pivots, measured with s_memtime.) A pivot not above `thr` is dead: its row and column become zero.
__device__ __forceinline__ double rcp_newton(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
    r = fma(r, fma(-x, r, 1.0), r);
    return r;
}
__device__ __forceinline__ void gj_invert32(double (&d)[GJB], int l, double thr) {
    // Row l is held as sc * d[] (sc = this lane's scale): scaling the pivot row by 1/pivot is then
    // two scalar operations on lane k instead of 31 multiplies executed for one active lane, and
    // the rows are scaled once, in parallel, at the end.
    double sc = 1.0;
#pragma unroll
    for (int k = 0; k < GJB; k++) {
        const double sk = readlane_d(sc, k);
        const double piv = sk * readlane_d(d[k], k);  // true pivot
        const double ip = (piv > thr && piv > 0.0) ? rcp_newton(piv) : 0.0;
        const double skn = sk * ip;                   // scale of the pivot row after the step
        const bool me = l == k;
        const double m = me ? 0.0 : d[k] * skn;       // multiple of the (unscaled) pivot row to subtract
#pragma unroll
        for (int j = 0; j < GJB; j++) {
            if (j != k) d[j] = fma(-m, readlane_d(d[j], k), d[j]);
        }
        // column k: true value ip on the pivot row (= skn * (1 / sk)), -a_lk * ip elsewhere
        d[k] = me ? (sk != 0.0 ? 1.0 / sk : 0.0) : -(d[k] * ip);
        if (me) sc = skn;
    }
#pragma unroll
    for (int j = 0; j < GJB; j++) d[j] *= sc;
}

// The same inversion with TWO lanes per row and the pivot data broadcast through LDS: lane l + 32 h
// (l = row, h = 0 / 1) holds columns [16 h, 16 h + 16) of row l. Per pivot k the two lanes of row k
// publish their halves of the pivot row, the 32 lanes that own column k publish a(r, k), and after one
// wave-local LDS round trip every lane has its row's multiplier and its half of the pivot row: 16 fma
// per pivot instead of 32 and no v_readlane (two per fma above; they, not the fma, set the 13 us).
// Row k is scaled by the same fma as the others (multiplier (p - 1) / p: a_kj - ((p - 1) / p) a_kj =
// a_kj / p), so the wave never diverges; column k is patched afterwards by its owners. LDS operations of a
// wave execute in order, so one buffer serves every pivot. `buf`: 64 doubles.
__device__ __forceinline__ void gj_invert32_split(double (&d)[GJB / 2], int l, int h, double thr,
                                                  double *buf) {
    double *rowb = buf, *colb = buf + GJB;
#pragma unroll
    for (int k = 0; k < GJB; k++) {
        constexpr int HALF = GJB / 2;
        const int hk = k / HALF, jk = k % HALF;
        if (h == hk) colb[l] = d[jk];
        if (l == k) {
#pragma unroll
            for (int j = 0; j < HALF; j++) rowb[HALF * h + j] = d[j];
        }
        asm volatile("" ::: "memory");  // the wave's LDS operations execute in order: a compiler fence is enough
        __builtin_amdgcn_wave_barrier();
        const double ark = colb[l];
        const double piv = rowb[k];
        double pr[HALF];
#pragma unroll
        for (int j = 0; j < HALF; j++) pr[j] = rowb[HALF * h + j];
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const double ip = (piv > thr && piv > 0.0) ? rcp_newton(piv) : 0.0;
        const bool me = l == k;
        const double m = me ? (ip != 0.0 ? (piv - 1.0) * ip : 1.0) : ark * ip;
#pragma unroll
        for (int j = 0; j < HALF; j++) d[j] = fma(-m, pr[j], d[j]);
        if (h == hk) d[jk] = me ? ip : -(ark * ip);
    }
}

// E (npad x npad, row-major, zeroed beforehand) from the SELL level: E = diag + offdiag;
// identity on the padding rows. One lane per row: a row's entries are written by its owner only.
__global__ __launch_bounds__(kRowBlock) void k_dense_build(LevelView C, int npad,
                                                           double *__restrict__ E,
                                                           double *__restrict__ maxdiag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    {   // largest diagonal entry (zeroed beforehand; non-negative doubles order like their bits)
        double dm = (i < C.n) ? fmax(C.diag[i], 0.0) : 0.0;
        for (int o = 32; o > 0; o >>= 1) dm = fmax(dm, __shfl_xor(dm, o, 64));
        if ((threadIdx.x & 63) == 0 && dm > 0.0)
            atomicMax(reinterpret_cast<unsigned long long *>(maxdiag), (unsigned long long)__double_as_longlong(dm));
    }
    if (i >= npad) return;
    double *row = E + (size_t)i * npad;
    if (i >= C.n) {
        row[i] = 1.0;
        return;
    }
    row[i] = C.diag[i];
    const int sl = i >> 6, lane = i & 63;
    const int o0 = C.sl_off[sl], w = C.sl_off[sl + 1] - o0;
    for (int k = 0; k < w; k++) {
        const size_t p = sell_pos(o0, k, lane);
        const double v = C.val[p];
        if (v != 0.0) row[C.col[p]] += v;
    }
}

// The same for a COARSE level (every (row, column) pair occurs once: the coarse patterns are merged at build
// time, so plain stores into the zeroed matrix suffice -- level 0 can hold duplicate edges and keeps the
// read-modify-write kernel above): one workgroup per slice, one thread per SELL position. The row-per-lane
// kernel spent 80 us at 1563 rows on its dependent load / store chains, once per inversion.
__global__ __launch_bounds__(kRowBlock) void k_dense_build_coarse(LevelView C, int npad, double *__restrict__ E,
                                                                  double *__restrict__ maxdiag) {
    const int sl = blockIdx.x, tid = threadIdx.x;
    if (sl >= C.nsl) {  // the padding rows: identity
        for (int i = C.n + (sl - C.nsl) * kRowBlock + tid; i < npad; i += (gridDim.x - C.nsl) * kRowBlock) E[(size_t)i * npad + i] = 1.0;
        return;
    }
    const int o0 = C.sl_off[sl], w = C.sl_off[sl + 1] - o0;
    if (tid < 64) {
        const int i = sl * 64 + tid;
        double dm = 0.0;
        if (i < C.n) {
            const double d = C.diag[i];
            E[(size_t)i * npad + i] = d;
            dm = fmax(d, 0.0);
        }
        for (int o = 32; o > 0; o >>= 1) dm = fmax(dm, __shfl_xor(dm, o, 64));
        if (tid == 0 && dm > 0.0)
            atomicMax(reinterpret_cast<unsigned long long *>(maxdiag), (unsigned long long)__double_as_longlong(dm));
    }
    for (int idx = tid; idx < w * 64; idx += kRowBlock) {
        const int k = idx >> 6, lane = idx & 63, row = sl * 64 + lane;
        const size_t p = sell_pos(o0, k, lane);
        const double v = C.val[p];
        if (v != 0.0 && row < C.n) E[(size_t)row * npad + C.col[p]] = v;
    }
}

// panel kernel of block step k: every workgroup inverts D = A_kk (32 x 32) by itself -- ONE wave,
// one lane per row, the row in registers, 32 fully unrolled scalar Gauss-Jordan steps with the
// pivot row broadcast by lane reads: no barriers, ~3 us (a barrier-per-step LDS version costs
// ten times that, and this sits on the critical path of all npad/32 steps). A pivot not above
// kDeadTol x the largest diagonal entry zeroes its row/column: that unknown solves to 0. Then the workgroup builds a 32 x 64 chunk of
// Rt, or copies a 64 x 32 chunk of the column panel C.
__global__ __launch_bounds__(256) void k_gj_panel(int npad, int k0, const double *__restrict__ A,
                                                  double *__restrict__ Wr, double *__restrict__ Wc,
                                                  const double *__restrict__ maxdiag) {
    const int nchunk = npad / GJT;
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= nchunk) {  // column panel copy
        const int r0 = ((int)blockIdx.x - nchunk) * GJT;
        for (int e = tid; e < GJT * GJB; e += 256) {
            const int r = e / GJB, q = e % GJB;
            Wc[(size_t)(r0 + r) * GJB + q] = A[(size_t)(r0 + r) * npad + k0 + q];
        }
        return;
    }
    __shared__ double D[GJB][GJB + 1];
    __shared__ double Ak[GJB][GJT + 1];
    const int c0 = blockIdx.x * GJT;
    for (int e = tid; e < GJB * GJT; e += 256) Ak[e / GJT][e % GJT] = A[(size_t)(k0 + e / GJT) * npad + c0 + e % GJT];
    if (tid < 64) {
        const int l = tid & 31;  // lanes 32..63 mirror lanes 0..31 (keeps the wave uniform)
        double d[GJB];
#pragma unroll
        for (int j = 0; j < GJB; j++) d[j] = A[(size_t)(k0 + l) * npad + k0 + j];
        // dead-pivot threshold: kDeadTol (common.hpp) x the largest diagonal entry of E
        const double thr = kDeadTol * maxdiag[0];
        gj_invert32(d, l, thr);
        if (tid < GJB) {
#pragma unroll
            for (int j = 0; j < GJB; j++) D[l][j] = d[j];
        }
    }
    __syncthreads();
    // Rt chunk = D^-1 * A_k,chunk, except columns inside block k, which receive D^-1 itself
    for (int e = tid; e < GJB * GJT; e += 256) {
        const int q = e / GJT, c = e % GJT;
        const int gc = c0 + c;
        double s;
        if (gc >= k0 && gc < k0 + GJB) {
            s = D[q][gc - k0];
        } else {
            s = 0.0;
#pragma unroll 8
            for (int t = 0; t < GJB; t++) s += D[q][t] * Ak[t][c];
        }
        Wr[(size_t)q * npad + gc] = s;
    }
}

// rank-32 update of one 64 x 64 tile (see file header); 256 threads, 4 x 4 outputs each
__global__ __launch_bounds__(256) void k_gj_update(int npad, int k0, double *__restrict__ A,
                                                   const double *__restrict__ Wr,
                                                   const double *__restrict__ Wc) {
    __shared__ double Cs[GJT][GJB + 1];
    __shared__ double Rs[GJB][GJT + 4];
    const int nt = npad / GJT;
    const int ti = blockIdx.x / nt, tj = blockIdx.x % nt;
    const int r0 = ti * GJT, c0 = tj * GJT;
    const int tid = threadIdx.x;
    for (int e = tid; e < GJT * GJB; e += 256) Cs[e / GJB][e % GJB] = Wc[(size_t)(r0 + e / GJB) * GJB + e % GJB];
    for (int e = tid; e < GJB * GJT; e += 256) Rs[e / GJT][e % GJT] = Wr[(size_t)(e / GJT) * npad + c0 + e % GJT];
    __syncthreads();
    const int ty = tid / 16, tx = tid % 16;
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = 0.0;
#pragma unroll 4
    for (int q = 0; q < GJB; q++) {
        double cv[4], rv[4];
#pragma unroll
        for (int a = 0; a < 4; a++) cv[a] = Cs[ty * 4 + a][q];
#pragma unroll
        for (int b = 0; b < 4; b++) rv[b] = Rs[q][tx * 4 + b];
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) acc[a][b] += cv[a] * rv[b];
    }
#pragma unroll
    for (int a = 0; a < 4; a++) {
        const int r = r0 + ty * 4 + a;
        const bool in_k_row = r >= k0 && r < k0 + GJB;
        double *arow = A + (size_t)r * npad + c0 + tx * 4;
        double out[4];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int c = c0 + tx * 4 + b;
            if (in_k_row) {
                out[b] = Rs[r - k0][tx * 4 + b];
            } else {
                const double old = (c >= k0 && c < k0 + GJB) ? 0.0 : arow[b];
                out[b] = old - acc[a][b];
            }
        }
        *reinterpret_cast<double2 *>(arow) = make_double2(out[0], out[1]);
        *reinterpret_cast<double2 *>(arow + 2) = make_double2(out[2], out[3]);
    }
}

// Update of step k WITH LOOK-AHEAD: the same rank-32 tile update, and in the same launch the panel
// of step k+1 (what k_gj_panel would compute next), so that the serial 32 x 32 inversion overlaps
// the update instead of preceding it. No workgroup waits for another:
//   * every workgroup of block row k+1 (dispatched first) rebuilds D' = A'_{k+1,k+1} itself --
//     D' = S - C_k[rows k+1] Rt_k[cols k+1], where S is the snapshot of that block taken by the
//     previous launch (the live block is being overwritten by its owner in this launch) -- and
//     inverts it on a FIFTH wave while waves 0-3 update the tile (measured: the tile update of
//     these workgroups competes with 600 others for memory and takes as long as the inversion;
//     done one after the other the look-ahead gained nothing). Then the workgroup writes its
//     32 x 64 chunk of Rt' = D'^-1 A'_{k+1,cols} (block k+1's own columns receive D'^-1) to WrN;
//   * the workgroups of block column k+1 copy their updated 64 x 32 strip into WcN;
//   * the owner of block (k+2, k+2) stores its updated block as the snapshot for the next launch.
// 42.7 KB of LDS and <= 128 VGPRs: three 5-wave workgroups per CU, all tiles of a 2048^2 matrix
// resident at once. In the other workgroups the fifth wave exits at once.
constexpr int GJ_LA_THREADS = 320;
__global__ __launch_bounds__(GJ_LA_THREADS, 4) void k_gj_update_la(
    int npad, int k0, double *__restrict__ A, const double *__restrict__ Wr,
    const double *__restrict__ Wc, double *__restrict__ WrN, double *__restrict__ WcN,
    const double *__restrict__ Sr, double *__restrict__ Sw, const double *__restrict__ maxdiag) {
    __shared__ double Cs[GJT][GJB + 1];
    __shared__ double Rs[GJB][GJT + 4];   // Rt_k chunk; later the updated rows of block k+1
    __shared__ double Dv[GJB][GJB + 1];   // Rt_k[:, cols k+1], then D', then D'^-1
    __shared__ double Pv[2 * GJB];        // pivot row / pivot column of the inversion in flight
    const int nt = npad / GJT;
    const int k1 = k0 + GJB, k2 = k0 + 2 * GJB;  // first row/column of blocks k+1, k+2
    const int tn = k1 / GJT;                     // tile index of block k+1 (rows and columns)
    int ti, tj;
    {
        const int bid = blockIdx.x;
        if (bid == 0) {
            ti = tn;
            tj = tn;
        } else if (bid < nt) {
            ti = tn;
            tj = bid - 1;
            if (tj >= tn) tj++;
        } else if (bid < 2 * nt - 1) {
            tj = tn;
            ti = bid - nt;
            if (ti >= tn) ti++;
        } else {
            const int r = bid - (2 * nt - 1);
            ti = r / (nt - 1);
            tj = r % (nt - 1);
            if (ti >= tn) ti++;
            if (tj >= tn) tj++;
        }
    }
    const int r0 = ti * GJT, c0 = tj * GJT;
    const int tid = threadIdx.x;
    const bool brow = ti == tn;
    const bool inverter = tid >= 256;  // the fifth wave
    if (inverter && !brow) return;
    const int ty = (tid & 255) / 16, tx = tid % 16;
    // Loads, in the order their consumers sit on the critical path (a wave's loads return in issue
    // order): what D' needs -- Rt_k[cols k+1], the snapshot, the column panel -- then the row panel
    // chunk and the tile's own old values, both of which are only waited for after D' is formed.
    double dvr[4] = {0, 0, 0, 0}, snap[4] = {0, 0, 0, 0}, csr[8], rsr[8];
    double out[4][4];
    if (!inverter) {
        if (brow) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int e = tid * 4 + u;
                dvr[u] = Wr[(size_t)(e / GJB) * npad + k1 + e % GJB];
                snap[u] = Sr[e];
            }
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int e = tid + 256 * u;
            csr[u] = Wc[(size_t)(r0 + e / GJB) * GJB + e % GJB];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int e = tid + 256 * u;
            rsr[u] = Wr[(size_t)(e / GJT) * npad + c0 + e % GJT];
        }
#pragma unroll
        for (int a = 0; a < 4; a++) {
            const double *arow = A + (size_t)(r0 + ty * 4 + a) * npad + c0 + tx * 4;
            const double2 v0 = *reinterpret_cast<const double2 *>(arow);
            const double2 v1 = *reinterpret_cast<const double2 *>(arow + 2);
            out[a][0] = v0.x;
            out[a][1] = v0.y;
            out[a][2] = v1.x;
            out[a][3] = v1.y;
        }
        if (brow) {
#pragma unroll
            for (int u = 0; u < 4; u++) Dv[(tid * 4 + u) / GJB][(tid * 4 + u) % GJB] = dvr[u];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) Cs[(tid + 256 * u) / GJB][(tid + 256 * u) % GJB] = csr[u];
        if (!brow) {
#pragma unroll
            for (int u = 0; u < 8; u++) Rs[(tid + 256 * u) / GJT][(tid + 256 * u) % GJT] = rsr[u];
        }
    }
    __syncthreads();
    if (brow) {
        // D' = S - C_k[rows k+1] * Rt_k[cols k+1]: 4 entries per thread of waves 0-3, same
        // summation order as the tile update
        double dn[4] = {0, 0, 0, 0};
        const int l = (tid & 255) / 8, j0 = (tid % 8) * 4;  // entry (l, j0..j0+3); snap[] matches
        if (!inverter) {
            double accd[4] = {0, 0, 0, 0};
#pragma unroll 4
            for (int t = 0; t < GJB; t++) {
                const double cv = Cs[k1 - r0 + l][t];
#pragma unroll
                for (int u = 0; u < 4; u++) accd[u] += cv * Dv[t][j0 + u];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) dn[u] = snap[u] - accd[u];
        }
        __syncthreads();  // all reads of Rt_k[cols k+1] in Dv done
        if (!inverter) {
#pragma unroll
            for (int u = 0; u < 4; u++) Dv[l][j0 + u] = dn[u];
#pragma unroll
            for (int u = 0; u < 8; u++) Rs[(tid + 256 * u) / GJT][(tid + 256 * u) % GJT] = rsr[u];
        }
        __syncthreads();
    }
    if (inverter) {  // invert D' in registers, two lanes per row (gj_invert32_split)
        const int l = tid & 31, h = (tid >> 5) & 1;
        double d[GJB / 2];
#pragma unroll
        for (int j = 0; j < GJB / 2; j++) d[j] = Dv[l][GJB / 2 * h + j];
        gj_invert32_split(d, l, h, kDeadTol * maxdiag[0], Pv);
        __syncthreads();  // (1) the update waves are done with the Rt_k chunk in Rs
#pragma unroll
        for (int j = 0; j < GJB / 2; j++) Dv[l][GJB / 2 * h + j] = d[j];
    } else {
        double acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) acc[a][b] = 0.0;
#pragma unroll 4
        for (int q = 0; q < GJB; q++) {
            double cv[4], rv[4];
#pragma unroll
            for (int a = 0; a < 4; a++) cv[a] = Cs[ty * 4 + a][q];
#pragma unroll
            for (int b = 0; b < 4; b++) rv[b] = Rs[q][tx * 4 + b];
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) acc[a][b] += cv[a] * rv[b];
        }
#pragma unroll
        for (int a = 0; a < 4; a++) {
            const int r = r0 + ty * 4 + a;
            const bool in_k_row = r >= k0 && r < k0 + GJB;
            double *arow = A + (size_t)r * npad + c0 + tx * 4;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int c = c0 + tx * 4 + b;
                if (in_k_row) {
                    out[a][b] = Rs[r - k0][tx * 4 + b];
                } else {
                    const double old = (c >= k0 && c < k0 + GJB) ? 0.0 : out[a][b];
                    out[a][b] = old - acc[a][b];
                }
            }
            *reinterpret_cast<double2 *>(arow) = make_double2(out[a][0], out[a][1]);
            *reinterpret_cast<double2 *>(arow + 2) = make_double2(out[a][2], out[a][3]);
            // hand-overs to the next launch: column panel of block k+1, snapshot of block (k+2, k+2)
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int c = c0 + tx * 4 + b;
                if (c >= k1 && c < k1 + GJB) WcN[(size_t)r * GJB + (c - k1)] = out[a][b];
                if (r >= k2 && r < k2 + GJB && c >= k2 && c < k2 + GJB) Sw[(r - k2) * GJB + (c - k2)] = out[a][b];
            }
        }
        if (!brow) return;
        __syncthreads();  // (1)
        // the updated rows of block k+1 go where the Rt_k chunk was
#pragma unroll
        for (int a = 0; a < 4; a++) {
            const int r = r0 + ty * 4 + a;
            if (r >= k1 && r < k1 + GJB) {
#pragma unroll
                for (int b = 0; b < 4; b++) Rs[r - k1][tx * 4 + b] = out[a][b];
            }
        }
    }
    __syncthreads();  // (2) D'^-1 in Dv, rows of block k+1 in Rs
    // Rt' chunk = D'^-1 * A'_{k+1, chunk}; the columns of block k+1 receive D'^-1 itself.
    // Thread -> row q, columns c8, c8 + 8, ... (conflict-free LDS reads, D'^-1 entry read once per 8)
    if (!inverter) {
        const int q = tid / 8, c8 = tid % 8;
        double rt[8];
#pragma unroll
        for (int b = 0; b < 8; b++) rt[b] = 0.0;
#pragma unroll 8
        for (int t = 0; t < GJB; t++) {
            const double dq = Dv[q][t];
#pragma unroll
            for (int b = 0; b < 8; b++) rt[b] += dq * Rs[t][c8 + 8 * b];
        }
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const int gc = c0 + c8 + 8 * b;
            if (gc >= k1 && gc < k1 + GJB) rt[b] = Dv[q][gc - k1];
            WrN[(size_t)q * npad + gc] = rt[b];
        }
    }
}

// snapshot of a 32 x 32 diagonal block (the look-ahead's stable copy of A_{b,b})
__global__ void k_gj_snapshot(int npad, int b0, const double *__restrict__ A, double *__restrict__ S) {
    for (int e = threadIdx.x; e < GJB * GJB; e += blockDim.x)
        S[e] = A[(size_t)(b0 + e / GJB) * npad + b0 + e % GJB];
}

// min / max of now[i] / ref[i] over an array (entries that are zero in both are skipped; an entry
// that is zero in only one of them forces a refresh). One workgroup, fixed-order reduction.
__global__ __launch_bounds__(1024) void k_value_ratio(long long n, const double *__restrict__ now,
                                                      const double *__restrict__ ref,
                                                      double *__restrict__ out, int accumulate) {
    __shared__ double smin[16], smax[16];
    double lo = HUGE_VAL, hi = 0.0;
    for (long long i = threadIdx.x; i < n; i += blockDim.x) {
        const double r = ref[i], v = now[i];
        if (r != 0.0 && v != 0.0) {
            const double q = v / r;
            if (q > 0.0) {
                lo = fmin(lo, q);
                hi = fmax(hi, q);
            } else {
                hi = HUGE_VAL;
            }
        } else if ((r != 0.0) != (v != 0.0)) {
            hi = HUGE_VAL;  // an entry appeared or vanished: force a refresh
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        lo = fmin(lo, __shfl_down(lo, o, 64));
        hi = fmax(hi, __shfl_down(hi, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        smin[threadIdx.x >> 6] = lo;
        smax[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); w++) {
            lo = fmin(lo, smin[w]);
            hi = fmax(hi, smax[w]);
        }
        if (accumulate) {
            lo = fmin(lo, out[0]);
            hi = fmax(hi, out[1]);
        }
        out[0] = lo;
        out[1] = hi;
    }
}

// ---- low-rank repair of the inverse ---------------------------------------------------------------
// A graph with a handful of loop closures re-weights almost uniformly (E_now ~ c E_ref entry by
// entry) EXCEPT for the few long-range coarse entries those closures own, which move by orders of
// magnitude in every IRLS iteration. Measured at 100k/2M with 20 closures: 20 deviating entry
// pairs out of 3164, all others within 3 %. The coarse operator is a weighted graph Laplacian, so
//     E_now ~ c E_ref + sum_k g_k v_k v_k',   v_k = e_i - e_j,  g_k = -(now_ij - c ref_ij),
// and with Z = E_ref^-1 V, S = c G^-1 + V'Z (r x r):  E_now^-1 ~ (E_ref^-1 - Z S^-1 Z') / c
// (Sherman-Morrison-Woodbury): an n^2 r update instead of the n^3 sweep.
constexpr int kWbMax = 64;  // deviating entry pairs repaired this way; more -> full inversion
struct WbEntry {
    int i, j;      // coarse row < coarse column
    double delta;  // now - c * ref of the off-diagonal entry (i, j)
};

// row of SELL entry position p on level C
__device__ __forceinline__ int sell_row_of(const LevelView &C, long long p) {
    const int q2 = (int)(p / 128) * 2;  // entry-column pair index -> first entry column
    const int lane = (int)(p % 128) / 2;
    int sl = 0;
    while (sl + 1 < C.nsl && C.sl_off[sl + 1] <= q2) sl++;
    return sl * 64 + lane;
}

// deviating off-diagonal entries (|now / (c ref) - 1| beyond `band`), each pair once (col > row).
// out: cnt[0] = number found (may exceed cap), cnt[1] = structural changes (zero <-> non-zero)
__global__ __launch_bounds__(256) void k_wb_collect(LevelView C, long long len, const double *__restrict__ ref,
                                                    double c, double band, WbEntry *__restrict__ list,
                                                    int cap, int *__restrict__ cnt) {
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < len;
         p += (long long)gridDim.x * blockDim.x) {
        const double r = ref[p], v = C.val[p];
        if (r == 0.0 && v == 0.0) continue;
        if ((r != 0.0) != (v != 0.0)) {
            atomicAdd(cnt + 1, 1);
            continue;
        }
        const double q = v / (c * r);
        if (q <= band && q * band >= 1.0) continue;
        // an entry that moved by more than four decades (an L1-type weight hitting its 1e4 cap: 1e8 in the
        // operator) makes S = c G^-1 + V'Z as ill-conditioned as that ratio and the repaired inverse useless
        // as a preconditioner (fuzz seed 31 case 511: the PCG stalled at 1e-5): counted like a structural
        // change -> full re-inversion
        if (!(q < 1e4 && q > 1e-4)) {
            atomicAdd(cnt + 1, 1);
            continue;
        }
        const int row = sell_row_of(C, p), col = C.col[p];
        if (col <= row) continue;
        const int k = atomicAdd(cnt, 1);
        if (k < cap) list[k] = WbEntry{row, col, v - c * r};
    }
}

// Z (r x npad, row k contiguous) = rows i_k - j_k of the symmetric inverse
__global__ __launch_bounds__(256) void k_wb_z(int npad, int r, const double *__restrict__ Minv,
                                              const WbEntry *__restrict__ list, double *__restrict__ Z) {
    const int k = blockIdx.y;
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < r && a < npad)
        Z[(size_t)k * npad + a] = Minv[(size_t)list[k].i * npad + a] - Minv[(size_t)list[k].j * npad + a];
}
// S = c G^-1 + V'Z  (r x r, row-major, ld = kWbMax)
__global__ void k_wb_s(int npad, int r, double c, const WbEntry *__restrict__ list, const double *__restrict__ Z,
                       double *__restrict__ S) {
    const int k = threadIdx.x / kWbMax, l = threadIdx.x % kWbMax;
    for (int kk = k; kk < r; kk += blockDim.x / kWbMax) {
        if (l < r) {
            double v = Z[(size_t)l * npad + list[kk].i] - Z[(size_t)l * npad + list[kk].j];
            if (kk == l) v += c / (-list[kk].delta);
            S[kk * kWbMax + l] = v;
        }
    }
}
// W = Sinv * Z  (r x npad)
__global__ __launch_bounds__(256) void k_wb_w(int npad, int r, const double *__restrict__ Sinv,
                                              const double *__restrict__ Z, double *__restrict__ W) {
    const int k = blockIdx.y;
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= r || a >= npad) return;
    double s = 0.0;
    for (int l = 0; l < r; l++) s += Sinv[k * kWbMax + l] * Z[(size_t)l * npad + a];
    W[(size_t)k * npad + a] = s;
}
// Minv <- (Minv - Z' W) / c
__global__ __launch_bounds__(256) void k_wb_apply(int npad, int r, double inv_c, const double *__restrict__ Z,
                                                  const double *__restrict__ W, double *__restrict__ Minv) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    const int a = blockIdx.y;
    if (b >= npad) return;
    double s = 0.0;
    for (int k = 0; k < r; k++) s += Z[(size_t)k * npad + a] * W[(size_t)k * npad + b];
    Minv[(size_t)a * npad + b] = (Minv[(size_t)a * npad + b] - s) * inv_c;
}
// the operator the repaired inverse belongs to: c E_ref + V G V'
__global__ __launch_bounds__(256) void k_wb_ref_val(long long len, const double *__restrict__ now, double c,
                                                    double band, double *__restrict__ ref) {
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < len;
         p += (long long)gridDim.x * blockDim.x) {
        const double r = ref[p], v = now[p];
        if (r == 0.0 || v == 0.0) continue;
        const double q = v / (c * r);
        ref[p] = (q <= band && q * band >= 1.0) ? c * r : v;
    }
}
__global__ void k_wb_ref_diag(int n, int r, double c, const WbEntry *__restrict__ list, double *__restrict__ ref) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) ref[i] *= c;
    __syncthreads();
    if (threadIdx.x == 0)  // sequential over the (sorted) list: fixed summation order
        for (int k = 0; k < r; k++) {
            ref[list[k].i] -= list[k].delta;
            ref[list[k].j] -= list[k].delta;
        }
}

// dense LU with partial pivoting on the host: S (r x r, ld = kWbMax) -> Sinv. False if singular.
static bool host_invert(int r, const double *S, double *Sinv) {
    std::vector<double> A((size_t)r * 2 * r);
    for (int i = 0; i < r; i++)
        for (int j = 0; j < r; j++) {
            A[(size_t)i * 2 * r + j] = S[i * kWbMax + j];
            A[(size_t)i * 2 * r + r + j] = i == j ? 1.0 : 0.0;
        }
    for (int k = 0; k < r; k++) {
        int piv = k;
        for (int i = k + 1; i < r; i++)
            if (std::fabs(A[(size_t)i * 2 * r + k]) > std::fabs(A[(size_t)piv * 2 * r + k])) piv = i;
        const double p = A[(size_t)piv * 2 * r + k];
        if (!(std::fabs(p) > 0.0) || !std::isfinite(p)) return false;
        if (piv != k)
            for (int j = 0; j < 2 * r; j++) std::swap(A[(size_t)k * 2 * r + j], A[(size_t)piv * 2 * r + j]);
        const double ip = 1.0 / p;
        for (int j = 0; j < 2 * r; j++) A[(size_t)k * 2 * r + j] *= ip;
        for (int i = 0; i < r; i++) {
            if (i == k) continue;
            const double f = A[(size_t)i * 2 * r + k];
            if (f == 0.0) continue;
            for (int j = 0; j < 2 * r; j++) A[(size_t)i * 2 * r + j] -= f * A[(size_t)k * 2 * r + j];
        }
    }
    for (int i = 0; i < r; i++)
        for (int j = 0; j < r; j++) {
            const double v = A[(size_t)i * 2 * r + r + j];
            if (!std::isfinite(v)) return false;
            Sinv[i * kWbMax + j] = v;
        }
    return true;
}

// Tries the low-rank repair for E_now ~ c E_ref + (few entries). True on success (the live inverse
// and its reference operator are updated, dense_scale = 1); false -> the caller re-inverts.
static bool dense_lowrank_repair(Graph &g, double c) {
    Level &C = g.levels.back();
    const int npad = g.ndense_pad;
    const size_t zlen = (size_t)kWbMax * npad;
    if (g.dense_wb.n < 2 * zlen + 2 * (size_t)kWbMax * kWbMax + 2 * kWbMax + 8)
        g.dense_wb.alloc(2 * zlen + 2 * (size_t)kWbMax * kWbMax + 2 * kWbMax + 8);
    double *Z = g.dense_wb.p, *W = Z + zlen, *S = W + zlen, *Sinv = S + kWbMax * kWbMax;
    WbEntry *list = reinterpret_cast<WbEntry *>(Sinv + kWbMax * kWbMax);
    int *cnt = reinterpret_cast<int *>(list + kWbMax);
    LevelView V{C.n, C.nsl, C.agg, C.sl_off.p, C.sl_near.p, C.col.p, C.val.p, C.diag.p, C.idg.p};
    IRH_CHECK(hipMemsetAsync(cnt, 0, 2 * sizeof(int), g.stream));
    hipLaunchKernelGGL(k_wb_collect, dim3(64), dim3(256), 0, g.stream, V, C.sell_len, g.dense_ref_val.p, c,
                       g.stale_spread, list, kWbMax, cnt);
    int hc[2];
    WbEntry hl[kWbMax];
    IRH_CHECK(hipMemcpyAsync(hc, cnt, sizeof(hc), hipMemcpyDeviceToHost, g.stream));
    IRH_CHECK(hipMemcpyAsync(hl, list, sizeof(hl), hipMemcpyDeviceToHost, g.stream));
    IRH_CHECK(hipStreamSynchronize(g.stream));
    const int r = hc[0];
    if (hc[1] != 0 || r <= 0 || r > kWbMax) return false;
    std::sort(hl, hl + r, [](const WbEntry &a, const WbEntry &b) { return a.i != b.i ? a.i < b.i : a.j < b.j; });
    for (int k = 0; k < r; k++)
        if (!(hl[k].delta != 0.0) || !std::isfinite(hl[k].delta)) return false;
    IRH_CHECK(hipMemcpyAsync(list, hl, sizeof(WbEntry) * r, hipMemcpyHostToDevice, g.stream));
    const dim3 gz((npad + 255) / 256, r);
    hipLaunchKernelGGL(k_wb_z, gz, dim3(256), 0, g.stream, npad, r, g.dense_inv.p, list, Z);
    hipLaunchKernelGGL(k_wb_s, dim3(1), dim3(1024), 0, g.stream, npad, r, c, list, Z, S);
    std::vector<double> hS((size_t)kWbMax * kWbMax), hSinv((size_t)kWbMax * kWbMax, 0.0);
    IRH_CHECK(hipMemcpyAsync(hS.data(), S, sizeof(double) * kWbMax * kWbMax, hipMemcpyDeviceToHost, g.stream));
    IRH_CHECK(hipStreamSynchronize(g.stream));
    if (!host_invert(r, hS.data(), hSinv.data())) return false;
    IRH_CHECK(hipMemcpyAsync(Sinv, hSinv.data(), sizeof(double) * kWbMax * kWbMax, hipMemcpyHostToDevice, g.stream));
    hipLaunchKernelGGL(k_wb_w, gz, dim3(256), 0, g.stream, npad, r, Sinv, Z, W);
    hipLaunchKernelGGL(k_wb_apply, dim3((npad + 255) / 256, npad), dim3(256), 0, g.stream, npad, r, 1.0 / c, Z, W,
                       g.dense_inv.p);
    hipLaunchKernelGGL(k_wb_ref_val, dim3(64), dim3(256), 0, g.stream, C.sell_len, C.val.p, c, g.stale_spread,
                       g.dense_ref_val.p);
    hipLaunchKernelGGL(k_wb_ref_diag, dim3(1), dim3(1024), 0, g.stream, C.n, r, c, list, g.dense_ref_diag.p);
    IRH_CHECK(hipStreamSynchronize(g.stream));  // hSinv / hl leave scope
    g.dense_scale = 1.0;
    g.dense_epoch++;  // dense_inv changed: copies of it (cgcg.hip) are stale
    g.stats.dense_repairs++;
    return true;
}

// Is the inverse computed at the last refresh still a good coarse solver for the CURRENT coarse
// operator? Every entry of E (diagonal AND off-diagonal: a re-weighted loop closure barely moves
// a diagonal that sums ~2500 edges, but changes its own long-range entry by orders of magnitude)
// is compared with the operator the inverse belongs to. If all ratios lie within a narrow band
// around one factor c, E_now ~ c E_ref and E_ref^-1 / c is reused; if only the diagonal does and
// few off-diagonal entries deviate (`allow_repair`), the inverse is repaired by a low-rank update;
// otherwise the caller re-inverts. Returns true when a full refresh is needed.
// The same test in ONE launch with the decision taken on the device, for callers that do not want to
// wait for it (run_irls on the two-launch PCG path): ratios now / ref over the coarse diagonal and the
// coarse off-diagonal values; if they all lie within `spread` of one factor the scale of the re-used
// inverse goes to scal[SC_DSCALE] (k_cg_apply reads it there) and flags[FL_STALE] = 0, otherwise
// FL_STALE = 1 and the scale stays. A stale inverse costs PCG iterations, never accuracy, so the host
// reads the verdict with the solve's own read-back and re-inverts before the NEXT solve.
__global__ __launch_bounds__(1024) void k_stale_check(long long nd, const double *__restrict__ dnow,
                                                      const double *__restrict__ dref, long long nv,
                                                      const double *__restrict__ vnow,
                                                      const double *__restrict__ vref, double spread,
                                                      double *__restrict__ scal, int *__restrict__ flags) {
    __shared__ double smin[16], smax[16];
    double lo = HUGE_VAL, hi = 0.0;
    auto visit = [&](double v, double r) {
        if (r != 0.0 && v != 0.0) {
            const double q = v / r;
            if (q > 0.0) {
                lo = fmin(lo, q);
                hi = fmax(hi, q);
            } else {
                hi = HUGE_VAL;
            }
        } else if ((r != 0.0) != (v != 0.0)) {
            hi = HUGE_VAL;  // an entry appeared or vanished
        }
    };
    for (long long i = threadIdx.x; i < nd; i += blockDim.x) visit(dnow[i], dref[i]);
    for (long long i = threadIdx.x; i < nv; i += blockDim.x) visit(vnow[i], vref[i]);
    for (int o = 32; o > 0; o >>= 1) {
        lo = fmin(lo, __shfl_down(lo, o, 64));
        hi = fmax(hi, __shfl_down(hi, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        smin[threadIdx.x >> 6] = lo;
        smax[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); w++) {
            lo = fmin(lo, smin[w]);
            hi = fmax(hi, smax[w]);
        }
        // the scale follows the operator even when the verdict is 'stale' (the geometric mean of the
        // ratios keeps the coarse correction at the right magnitude: a stale inverse at scale 1 after the
        // weights went from 1 to 1e4 is no preconditioner at all -- measured 2000 iterations)
        if (lo > 0.0 && hi < HUGE_VAL) scal[SC_DSCALE] = 1.0 / sqrt(lo * hi);
        flags[FL_STALE] = (lo > 0.0 && hi < HUGE_VAL && hi <= spread * lo) ? 0 : 1;
    }
}

void dense_check_async(Graph &g) {
    Level &C = g.levels.back();
    hipLaunchKernelGGL(k_stale_check, dim3(1), dim3(1024), 0, g.stream, (long long)C.n, C.diag.p, g.dense_ref_diag.p,
                       (long long)(g.dense_ref_val.n >= (size_t)C.sell_len ? C.sell_len : 0), C.val.p,
                       g.dense_ref_val.p, g.stale_spread, g.scal.p, g.flags.p);
}

// Keep the current inverse whatever the entry ratios say and only follow the operator's scale (geometric mean of
// the diagonal ratios): for a solve whose operator is known to have moved little (assemble(), solver.hip: the last
// IRLS step was tiny, so the robust weights have settled). false: no usable scale (a diagonal entry vanished).
bool dense_rescale_only(Graph &g) {
    if (g.ndense <= 0 || !g.dense_valid) return false;
    Level &C = g.levels.back();
    hipLaunchKernelGGL(k_value_ratio, dim3(1), dim3(1024), 0, g.stream, (long long)C.n, C.diag.p, g.dense_ref_diag.p,
                       g.part_score.p, 0);
    double *h = g.h_part();
    IRH_CHECK(hipMemcpyAsync(h, g.part_score.p, sizeof(double) * 2, hipMemcpyDeviceToHost, g.stream));
    IRH_CHECK(hipStreamSynchronize(g.stream));
    if (!(h[0] > 0.0) || !(h[1] < HUGE_VAL)) return false;
    g.dense_scale = 1.0 / std::sqrt(h[0] * h[1]);
    return true;
}

bool dense_is_stale(Graph &g, bool allow_repair) {
    if (g.ndense <= 0) return false;
    if (!g.dense_valid) return true;
    Level &C = g.levels.back();
    // [0,1]: all entries; [2,3]: diagonal only
    hipLaunchKernelGGL(k_value_ratio, dim3(1), dim3(1024), 0, g.stream, (long long)C.n, C.diag.p,
                       g.dense_ref_diag.p, g.part_score.p + 2, 0);
    hipLaunchKernelGGL(k_value_ratio, dim3(1), dim3(1024), 0, g.stream, (long long)C.n, C.diag.p,
                       g.dense_ref_diag.p, g.part_score.p, 0);
    if (C.sell_len > 0)
        hipLaunchKernelGGL(k_value_ratio, dim3(1), dim3(1024), 0, g.stream, C.sell_len, C.val.p,
                           g.dense_ref_val.p, g.part_score.p, 1);
    double *h = g.h_part();  // pinned staging
    IRH_CHECK(hipMemcpyAsync(h, g.part_score.p, sizeof(double) * 4, hipMemcpyDeviceToHost, g.stream));
    IRH_CHECK(hipStreamSynchronize(g.stream));
    const double lo = h[0], hi = h[1], lod = h[2], hid = h[3];
    if ((lo > 0.0) && (hi < HUGE_VAL) && hi <= g.stale_spread * lo) {
        g.dense_scale = 1.0 / std::sqrt(lo * hi);
        return false;
    }
    // not uniform: a few deviating long-range entries on top of a uniform change? (at most
    // kWbRepairsMax repairs in a row: rounding accumulates in the repaired inverse)
    constexpr int kWbRepairsMax = 12;
    if (allow_repair && g.opt.no_lowrank_repair != 1 && (lod > 0.0) && (hid < HUGE_VAL) &&
        hid <= g.stale_spread * lod && C.sell_len > 0 && g.dense_repairs_in_a_row < kWbRepairsMax) {
        if (dense_lowrank_repair(g, std::sqrt(lod * hid))) {
            g.dense_repairs_in_a_row++;
            return false;
        }
    }
    return true;
}

// Make `slot` the live inverse (dense_inv, dense_ref_*, dense_scale, dense_valid): the live one is
// parked under its own number, the requested one (empty = never computed) is brought in.
void dense_select_slot(Graph &g, int slot) {
    if (g.ndense <= 0 || slot == g.dense_slot) return;
    const size_t need = (size_t)std::max(slot, g.dense_slot) + 1;
    if (g.dense_parked.size() < need) g.dense_parked.resize(need);
    auto exchange = [&](Graph::DenseSlot &S) {
        std::swap(S.inv, g.dense_inv);
        std::swap(S.ref_diag, g.dense_ref_diag);
        std::swap(S.ref_val, g.dense_ref_val);
        std::swap(S.scale, g.dense_scale);
        std::swap(S.valid, g.dense_valid);
    };
    exchange(g.dense_parked[g.dense_slot]);  // park the live one
    Graph::DenseSlot &T = g.dense_parked[slot];
    if (T.inv.n == 0) {  // first use: same sizes as the parked original
        const Graph::DenseSlot &O = g.dense_parked[g.dense_slot];
        T.inv.alloc(O.inv.n);
        T.ref_diag.alloc(O.ref_diag.n);
        T.valid = false;
        T.scale = 1.0;
    }
    exchange(T);
    g.dense_slot = slot;
    g.dense_epoch++;
}

// ---- last resort for a small single-level system: dense Cholesky + iterative refinement -----------------
// A graph of a few hundred views is ONE level whose operator is inverted explicitly and used as the PCG's
// preconditioner. When the operator spreads over ~13 decades (a near-tree graph under an L1-type cost whose
// weights hit the 1e4 cap: 1e8 in the operator) the explicit inverse carries a relative error of order 1, is
// no longer positive definite and the PCG stalls or diverges (fuzz seed 31, cases 162 / 1037) -- where the
// reference's factorisations still return an answer, because a Cholesky SOLVE is backward stable whatever
// the condition number. This kernel is that: one workgroup factorises E = L L' in place (right-looking,
// the pivot column staged in LDS), substitutes for the three right-hand sides and refines twice with the
// residual of the unfactorised copy. Dead pivots (not above kDeadTol x the largest diagonal entry) give 0,
// as everywhere else. ~10 ms at 400 views; only reached when the PCG has failed on a fresh inverse.
constexpr int kDirectMax = 1024;
constexpr int kDirectThreads = 1024;
__device__ void chol_substitute(int n, int npad, const double *__restrict__ A, double (*sol)[kDirectMax]) {
    const int tid = threadIdx.x;
    for (int k = 0; k < n; k++) {  // L y = b
        if (tid < 3) {
            const double d = A[(size_t)k * npad + k];
            sol[tid][k] = d > 0.0 ? sol[tid][k] / d : 0.0;
        }
        __syncthreads();
        for (int i = k + 1 + tid; i < n; i += kDirectThreads) {
            const double l = A[(size_t)i * npad + k];
            sol[0][i] -= l * sol[0][k];
            sol[1][i] -= l * sol[1][k];
            sol[2][i] -= l * sol[2][k];
        }
        __syncthreads();
    }
    for (int k = n - 1; k >= 0; k--) {  // L' x = y
        if (tid < 3) {
            const double d = A[(size_t)k * npad + k];
            sol[tid][k] = d > 0.0 ? sol[tid][k] / d : 0.0;
        }
        __syncthreads();
        for (int i = tid; i < k; i += kDirectThreads) {
            const double l = A[(size_t)k * npad + i];
            sol[0][i] -= l * sol[0][k];
            sol[1][i] -= l * sol[1][k];
            sol[2][i] -= l * sol[2][k];
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(kDirectThreads) void k_chol_solve(int n, int npad, double *__restrict__ A,
                                                             const double *__restrict__ E0,
                                                             const double4 *__restrict__ b, double4 *__restrict__ x,
                                                             const double *__restrict__ maxdiag) {
    __shared__ double col[kDirectMax];
    __shared__ double sol[3][kDirectMax];
    __shared__ double xs[3][kDirectMax];
    __shared__ double sinv;
    const int tid = threadIdx.x;
    const double thr = kDeadTol * maxdiag[0];
    for (int k = 0; k < n; k++) {
        if (tid == 0) {
            const double d = A[(size_t)k * npad + k];
            const bool dead = !(d > thr);
            const double r = dead ? 0.0 : sqrt(d);
            A[(size_t)k * npad + k] = r;
            sinv = dead ? 0.0 : 1.0 / r;
        }
        __syncthreads();
        const double inv = sinv;
        for (int i = k + 1 + tid; i < n; i += kDirectThreads) {
            const double l = A[(size_t)i * npad + k] * inv;  // a dead pivot: its column of L is zero
            A[(size_t)i * npad + k] = l;
            col[i] = l;
        }
        __syncthreads();
        const int m = n - k - 1;
        for (int idx = tid; idx < m * m; idx += kDirectThreads) {
            const int i = k + 1 + idx / m, j = k + 1 + idx % m;
            if (j <= i) A[(size_t)i * npad + j] -= col[i] * col[j];
        }
        __syncthreads();
    }
    for (int i = tid; i < n; i += kDirectThreads) {
        const double4 v = b[i];
        sol[0][i] = v.x;
        sol[1][i] = v.y;
        sol[2][i] = v.z;
        xs[0][i] = xs[1][i] = xs[2][i] = 0.0;
    }
    __syncthreads();
    for (int round = 0; round < 3; round++) {  // solve, then two refinements with r = b - E x
        chol_substitute(n, npad, A, sol);
        for (int i = tid; i < n; i += kDirectThreads)
            for (int c = 0; c < 3; c++) xs[c][i] += sol[c][i];
        __syncthreads();
        if (round == 2) break;
        const int lane = tid & 63, wv = tid >> 6;
        for (int i = wv; i < n; i += kDirectThreads / 64) {
            double s0 = 0, s1 = 0, s2 = 0;
            for (int j = lane; j < n; j += 64) {
                const double e = E0[(size_t)i * npad + j];
                s0 += e * xs[0][j];
                s1 += e * xs[1][j];
                s2 += e * xs[2][j];
            }
            s0 = wave_sum(s0);
            s1 = wave_sum(s1);
            s2 = wave_sum(s2);
            if (lane == 0) {
                const double4 v = b[i];
                sol[0][i] = v.x - s0;
                sol[1][i] = v.y - s1;
                sol[2][i] = v.z - s2;
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < n; i += kDirectThreads) {
        const double4 v = make_double4(xs[0][i], xs[1][i], xs[2][i], 0.0);
        x[i] = (isfinite(v.x) && isfinite(v.y) && isfinite(v.z)) ? v : make_double4(0, 0, 0, 0);
    }
}

// levels[0].b -> X for a single-level graph (see k_chol_solve). False if the graph does not qualify.
bool dense_direct_solve(Graph &g) {
    if (g.levels.size() != 1 || g.ndense <= 0 || g.ndense > kDirectMax || g.ng != 0) return false;
    Level &C = g.levels[0];
    const int npad = g.ndense_pad;
    const size_t sz = (size_t)npad * npad;
    if (g.dense_chol.n < 2 * sz) g.dense_chol.alloc(2 * sz);
    if (g.dense_maxdiag.n < 1) g.dense_maxdiag.alloc(1);
    IRH_CHECK(hipMemsetAsync(g.dense_chol.p, 0, sizeof(double) * 2 * sz, g.stream));
    IRH_CHECK(hipMemsetAsync(g.dense_maxdiag.p, 0, sizeof(double), g.stream));
    LevelView V{C.n, C.nsl, C.agg, C.sl_off.p, C.sl_near.p, C.col.p, C.val.p, C.diag.p, C.idg.p};
    for (int h = 0; h < 2; h++)
        hipLaunchKernelGGL(k_dense_build, dim3((npad + kRowBlock - 1) / kRowBlock), dim3(kRowBlock), 0, g.stream, V,
                           npad, g.dense_chol.p + h * sz, g.dense_maxdiag.p);
    hipLaunchKernelGGL(k_chol_solve, dim3(1), dim3(kDirectThreads), 0, g.stream, C.n, npad, g.dense_chol.p,
                       g.dense_chol.p + sz, C.b.p, g.X.p, g.dense_maxdiag.p);
    IRH_CHECK(hipStreamSynchronize(g.stream));
    return true;
}

// ---- banded coarsest operator: LDL' + one substitution per column --------------------------------
// A view sequence without loop closures coarsens to a BANDED operator (half-bandwidth 1-3 at 100k
// views). The blocked Gauss-Jordan sweep above spends npad / 32 dependent block steps (~20 us each:
// 1.0 ms at 1563 rows) whatever the structure; for a band of half-width BW the same explicit inverse
// is n columns of E X = I, each a forward and a backward substitution of BW multiply-adds per row:
//   * every workgroup factorises E = L D L' itself (one lane, the last BW rows of L and D in
//     registers: n dependent steps of ~BW^2 fma; redundant across the ~n/256 workgroups, but they run
//     side by side and nothing has to be published),
//   * then one thread per column runs the two substitutions from the LDS copy of L (broadcast reads)
//     with the column's running values in registers, storing rows as it goes (coalesced across the
//     workgroup's columns).
// A pivot not above kDeadTol x the largest diagonal entry is dead: that unknown solves to 0 (the
// same rule as the dense sweep). ~0.1 ms instead of 1.0 ms.
template <int BW>
__global__ __launch_bounds__(256) void k_band_inverse(LevelView C, int npad, double *__restrict__ X) {
    // LDS: one record per row, W doubles (16-byte aligned): l[0..BW-1] (l[d-1] = L(r, r-d)), then 1 / d(r)
    constexpr int W = ((BW + 1) + 1) & ~1;
    extern __shared__ double lds[];
    const int n = C.n;
    double *rec = lds;
    const int tid = threadIdx.x;
    __shared__ double smax[4];
    // band of E into LDS: slot BW holds the diagonal for now, slots 0..BW-1 the sub-diagonals
    for (int e = tid; e < n * W; e += 256) rec[e] = 0.0;
    __syncthreads();
    double dm = 0.0;
    for (int i = tid; i < n; i += 256) {
        const double d = C.diag[i];
        rec[(size_t)i * W + BW] = d;
        dm = fmax(dm, d);
        const int sl = i >> 6, ln = i & 63;
        const int o0 = C.sl_off[sl], w = C.sl_off[sl + 1] - o0;
        for (int k = 0; k < w; k++) {
            const size_t p = sell_pos(o0, k, ln);
            const double v = C.val[p];
            const int c = C.col[p];
            if (v != 0.0 && c < i && i - c <= BW) rec[(size_t)i * W + (i - c - 1)] += v;
        }
    }
    for (int o = 32; o > 0; o >>= 1) dm = fmax(dm, __shfl_xor(dm, o, 64));
    if ((tid & 63) == 0) smax[tid >> 6] = dm;
    __syncthreads();
    const double thr = kDeadTol * fmax(fmax(smax[0], smax[1]), fmax(smax[2], smax[3]));
    if (tid == 0) {
        // row-wise LDL': u(i, j) = a(i, j) - sum_p u(i, p) l(j, p), l(i, j) = u(i, j) / d(j),
        // d(i) = a(i, i) - sum_j u(i, j) l(i, j); pl[q][.] = row i-1-q of L, pd[q] = 1 / d(i-1-q)
        double pl[BW][BW], pd[BW];
#pragma unroll
        for (int q = 0; q < BW; q++) {
            pd[q] = 0.0;
#pragma unroll
            for (int d = 0; d < BW; d++) pl[q][d] = 0.0;
        }
        constexpr int FB = 8;  // rows per batch: their band entries are read from LDS before the
                               // dependent chain starts (an LDS round trip per row otherwise)
        for (int i0 = 0; i0 < n; i0 += FB) {
            double ab[FB][W];
#pragma unroll
            for (int b = 0; b < FB; b++) {
                const int i = min(i0 + b, n - 1);
#pragma unroll
                for (int d = 0; d < W; d++) ab[b][d] = rec[(size_t)i * W + d];  // a(i, i-1-d); [BW] = a(i, i)
            }
#pragma unroll
            for (int b = 0; b < FB; b++) {
                const int i = i0 + b;
                if (i >= n) break;
                double u[BW], l[BW];
                double dd = ab[b][BW];
                // columns j = i-BW .. i-1, i.e. d = BW-1 .. 0 (ascending j)
#pragma unroll
                for (int d = BW - 1; d >= 0; d--) {
                    // u(i, j) with j = i-1-d: subtract sum over p = j-1-e (e >= 0) of u(i, p) l(j, p);
                    // p = i-1-(d+1+e): u index d+1+e, l(j, p) = pl[d][e]
                    double t = ab[b][d];
#pragma unroll
                    for (int e = 0; e + d + 1 < BW; e++) t -= u[d + 1 + e] * pl[d][e];
                    u[d] = t;
                    l[d] = t * pd[d];
                    dd -= t * l[d];
                }
                const bool dead = !(dd > thr);
                // reciprocal: hardware estimate + two Newton steps (an fp64 division is a ~150-cycle
                // routine, and there is one per row on this serial path)
                double idd = __builtin_amdgcn_rcp(dd);
                idd = fma(fma(-dd, idd, 1.0), idd, idd);
                idd = fma(fma(-dd, idd, 1.0), idd, idd);
                if (dead) idd = 0.0;
#pragma unroll
                for (int d = 0; d < BW; d++) {
                    if (dead) l[d] = 0.0;  // the unknown solves to 0: no coupling through its row
                    rec[(size_t)i * W + d] = l[d];
                }
                rec[(size_t)i * W + BW] = idd;
                // shift the register window
#pragma unroll
                for (int q = BW - 1; q > 0; q--) {
                    pd[q] = pd[q - 1];
#pragma unroll
                    for (int d = 0; d < BW; d++) pl[q][d] = pl[q - 1][d];
                }
                pd[0] = idd;
#pragma unroll
                for (int d = 0; d < BW; d++) pl[0][d] = l[d];
            }
        }
    }
    __syncthreads();
    // one column per thread
    const int c = blockIdx.x * 256 + tid;
    const int c0 = blockIdx.x * 256;
    if (c0 >= n) return;
    const bool live = c < n;
    // The substitutions are ISSUE-bound, not latency-bound: 28 waves on the whole chip, one per SIMD, a
    // few cycles per instruction -- so the row loops are stripped to the recurrence itself (full
    // batches of U rows without bounds checks, running pointers, masked columns write to a dummy row).
    double yy[BW];  // y(r-1) .. y(r-BW)
#pragma unroll
    for (int d = 0; d < BW; d++) yy[d] = 0.0;
    constexpr int U = 8;
    // columns beyond n ride along on the last live column's address arithmetic and store nothing
    const int cc = live ? c : n - 1;
    double *xp = X + (size_t)c0 * npad + cc;
    auto fwd_rows = [&](int rb, int cnt) {
        double lr[U][W];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int r = min(rb + u, n - 1);
#pragma unroll
            for (int d = 0; d < W; d++) lr[u][d] = rec[(size_t)r * W + d];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (u < cnt) {
                double y = (rb + u == cc) ? 1.0 : 0.0;
#pragma unroll
                for (int d = 0; d < BW; d++) y -= lr[u][d] * yy[d];
                y = (rb + u < cc) ? 0.0 : y;
#pragma unroll
                for (int d = BW - 1; d > 0; d--) yy[d] = yy[d - 1];
                yy[0] = y;
                if (live) *xp = y * lr[u][BW];
                xp += npad;
            }
        }
    };
    int rb = c0;
    for (; rb + U <= n; rb += U) fwd_rows(rb, U);  // forward: L y = e_c, z = D^-1 y stored in place of X(r, c)
    if (rb < n) fwd_rows(rb, n - rb);
    double xx[BW];  // x(r+1) .. x(r+BW)
#pragma unroll
    for (int d = 0; d < BW; d++) xx[d] = 0.0;
    // backward: L' x = z, rows n-1 .. 0. z comes back from global memory: the NEXT batch's rows are
    // requested before the current batch's dependent chain runs. Rows above the workgroup's first column
    // hold z = 0 (never stored).
    xp = X + (size_t)(n - 1) * npad + cc;
    auto load_z = [&](int rtop, double (&z)[U]) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int r = rtop - u;
            z[u] = (live && r >= c0) ? X[(size_t)r * npad + c] : 0.0;
        }
    };
    auto bwd_rows = [&](int rtop, int cnt, const double (&z)[U]) {
        double lc[U][BW];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int r = rtop - u;
#pragma unroll
            for (int d = 0; d < BW; d++) {
                const int q = min(max(r + d + 1, 0), n - 1);  // L(r+d+1, r); beyond the last row: times xx = 0
                lc[u][d] = rec[(size_t)q * W + d];
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (u < cnt) {
                double x = z[u];
#pragma unroll
                for (int d = 0; d < BW; d++) x -= lc[u][d] * xx[d];
#pragma unroll
                for (int d = BW - 1; d > 0; d--) xx[d] = xx[d - 1];
                xx[0] = x;
                if (live) *xp = x;
                xp -= npad;
            }
        }
    };
    double zc[U], zn[U];
    int rt = n - 1;
    load_z(rt, zc);
    for (; rt - U + 1 >= 0; rt -= U) {
        load_z(rt - U, zn);
        bwd_rows(rt, U, zc);
#pragma unroll
        for (int u = 0; u < U; u++) zc[u] = zn[u];
    }
    if (rt >= 0) bwd_rows(rt, rt + 1, zc);
}

template <int BW>
static void band_inverse_launch(Graph &g, const LevelView &V) {
    const int n = g.ndense, npad = g.ndense_pad;
    const size_t shm = sizeof(double) * (size_t)n * (((BW + 1) + 1) & ~1);
    if (shm > 48 * 1024)
        IRH_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_band_inverse<BW>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipLaunchKernelGGL((k_band_inverse<BW>), dim3((n + 255) / 256), dim3(256), shm, g.stream, V, npad, g.dense_inv.p);
}

void dense_refresh(Graph &g) {
    if (g.ndense <= 0) return;
    g.dense_epoch++;
    g.dense_scale = 1.0;
    g.dense_repairs_in_a_row = 0;
    g.stats.dense_inversions++;
    IRH_CHECK(hipMemcpyAsync(g.dense_ref_diag.p, g.levels.back().diag.p,
                             sizeof(double) * (size_t)g.levels.back().n, hipMemcpyDeviceToDevice,
                             g.stream));
    if (g.levels.back().sell_len > 0) {
        if (g.dense_ref_val.n < (size_t)g.levels.back().sell_len)
            g.dense_ref_val.alloc((size_t)g.levels.back().sell_len);
        IRH_CHECK(hipMemcpyAsync(g.dense_ref_val.p, g.levels.back().val.p,
                                 sizeof(double) * (size_t)g.levels.back().sell_len,
                                 hipMemcpyDeviceToDevice, g.stream));
    }
    Level &C = g.levels.back();
    const int npad = g.ndense_pad;
    LevelView V{C.n, C.nsl, C.agg, C.sl_off.p, C.sl_near.p, C.col.p, C.val.p, C.diag.p, C.idg.p};
    IRH_CHECK(hipMemsetAsync(g.dense_inv.p, 0, sizeof(double) * (size_t)npad * npad, g.stream));
    if (g.dense_bw >= 1 && g.dense_bw <= kBandMax && !std::getenv("IROTAVG_NO_BAND_INVERSE")) {
        switch (g.dense_bw) {
        case 1: band_inverse_launch<1>(g, V); break;
        case 2: band_inverse_launch<2>(g, V); break;
        case 3: band_inverse_launch<3>(g, V); break;
        default: band_inverse_launch<4>(g, V); break;
        }
        return;
    }
    if (g.dense_maxdiag.n < 1) g.dense_maxdiag.alloc(1);
    IRH_CHECK(hipMemsetAsync(g.dense_maxdiag.p, 0, sizeof(double), g.stream));
    if (g.levels.size() > 1)
        hipLaunchKernelGGL(k_dense_build_coarse, dim3(C.nsl + 1), dim3(kRowBlock), 0, g.stream, V, npad, g.dense_inv.p,
                           g.dense_maxdiag.p);
    else
        hipLaunchKernelGGL(k_dense_build, dim3((npad + kRowBlock - 1) / kRowBlock), dim3(kRowBlock), 0,
                           g.stream, V, npad, g.dense_inv.p, g.dense_maxdiag.p);
    const int nchunk = npad / GJT;
    if (nchunk < 2 || std::getenv("IROTAVG_GJ_NO_LOOKAHEAD")) {
        for (int k0 = 0; k0 < npad; k0 += GJB) {
            hipLaunchKernelGGL(k_gj_panel, dim3(2 * nchunk), dim3(256), 0, g.stream, npad, k0,
                               g.dense_inv.p, g.dense_wr.p, g.dense_wc.p, g.dense_maxdiag.p);
            hipLaunchKernelGGL(k_gj_update, dim3(nchunk * nchunk), dim3(256), 0, g.stream, npad, k0,
                               g.dense_inv.p, g.dense_wr.p, g.dense_wc.p);
        }
        return;
    }
    // look-ahead sweep: the update of step k also produces the panel of step k+1 (ping-pong panels)
    // and the snapshot of block (k+2, k+2) for the launch after it (ping-pong snapshots)
    const size_t pan = (size_t)32 * npad;
    if (g.dense_wr.n < 2 * pan) g.dense_wr.alloc(2 * pan);
    if (g.dense_wc.n < 2 * pan) g.dense_wc.alloc(2 * pan);
    if (g.dense_la.n < (size_t)2 * GJB * GJB) g.dense_la.alloc((size_t)2 * GJB * GJB);
    hipLaunchKernelGGL(k_gj_panel, dim3(2 * nchunk), dim3(256), 0, g.stream, npad, 0, g.dense_inv.p,
                       g.dense_wr.p, g.dense_wc.p, g.dense_maxdiag.p);
    // step 0 reads the snapshot of block 1 (slot 1) and writes that of block 2 (slot 0)
    hipLaunchKernelGGL(k_gj_snapshot, dim3(1), dim3(256), 0, g.stream, npad, GJB, g.dense_inv.p,
                       g.dense_la.p + GJB * GJB);
    int cur = 0;
    for (int k0 = 0; k0 < npad; k0 += GJB) {
        double *wr = g.dense_wr.p + cur * pan, *wc = g.dense_wc.p + cur * pan;
        double *wrn = g.dense_wr.p + (cur ^ 1) * pan, *wcn = g.dense_wc.p + (cur ^ 1) * pan;
        const int step = k0 / GJB;
        double *sr = g.dense_la.p + ((step + 1) & 1) * GJB * GJB, *sw = g.dense_la.p + (step & 1) * GJB * GJB;
        if (k0 + GJB < npad)
            hipLaunchKernelGGL(k_gj_update_la, dim3(nchunk * nchunk), dim3(GJ_LA_THREADS), 0, g.stream, npad, k0,
                               g.dense_inv.p, wr, wc, wrn, wcn, sr, sw, g.dense_maxdiag.p);
        else
            hipLaunchKernelGGL(k_gj_update, dim3(nchunk * nchunk), dim3(256), 0, g.stream, npad, k0,
                               g.dense_inv.p, wr, wc);
        cur ^= 1;
    }
}

// In-place inverse of a dense symmetric positive definite matrix that is not a handle's coarse operator (the Woodbury
// system of the banded direct solver's closures, bcr.hip): the same blocked Gauss-Jordan sweep with look-ahead. npad: a
// multiple of 64, padding rows / columns = the identity's. Dead-pivot scale: the largest diagonal entry.
__global__ __launch_bounds__(256) void k_diag_max(int n, int npad, const double *__restrict__ A, double *__restrict__ maxdiag) {
    double dm = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) dm = fmax(dm, A[(size_t)i * npad + i]);
    for (int o = 32; o > 0; o >>= 1) dm = fmax(dm, __shfl_xor(dm, o, 64));
    if ((threadIdx.x & 63) == 0 && dm > 0.0)
        atomicMax(reinterpret_cast<unsigned long long *>(maxdiag), (unsigned long long)__double_as_longlong(dm));
}
void dense_invert_spd(Graph &g, double *A, int npad) {
    const size_t pan = (size_t)32 * npad;
    if (g.dense_wr.n < 2 * pan) g.dense_wr.alloc(2 * pan);
    if (g.dense_wc.n < 2 * pan) g.dense_wc.alloc(2 * pan);
    if (g.dense_la.n < (size_t)2 * GJB * GJB) g.dense_la.alloc((size_t)2 * GJB * GJB);
    if (g.dense_maxdiag.n < 1) g.dense_maxdiag.alloc(1);
    IRH_CHECK(hipMemsetAsync(g.dense_maxdiag.p, 0, sizeof(double), g.stream));
    hipLaunchKernelGGL(k_diag_max, dim3(4), dim3(256), 0, g.stream, npad, npad, A, g.dense_maxdiag.p);
    const int nchunk = npad / GJT;
    if (nchunk < 2) {
        for (int k0 = 0; k0 < npad; k0 += GJB) {
            hipLaunchKernelGGL(k_gj_panel, dim3(2 * nchunk), dim3(256), 0, g.stream, npad, k0, A, g.dense_wr.p,
                               g.dense_wc.p, g.dense_maxdiag.p);
            hipLaunchKernelGGL(k_gj_update, dim3(nchunk * nchunk), dim3(256), 0, g.stream, npad, k0, A, g.dense_wr.p,
                               g.dense_wc.p);
        }
        return;
    }
    hipLaunchKernelGGL(k_gj_panel, dim3(2 * nchunk), dim3(256), 0, g.stream, npad, 0, A, g.dense_wr.p, g.dense_wc.p,
                       g.dense_maxdiag.p);
    hipLaunchKernelGGL(k_gj_snapshot, dim3(1), dim3(256), 0, g.stream, npad, GJB, A, g.dense_la.p + GJB * GJB);
    int cur = 0;
    for (int k0 = 0; k0 < npad; k0 += GJB) {
        double *wr = g.dense_wr.p + cur * pan, *wc = g.dense_wc.p + cur * pan;
        double *wrn = g.dense_wr.p + (cur ^ 1) * pan, *wcn = g.dense_wc.p + (cur ^ 1) * pan;
        const int step = k0 / GJB;
        double *sr = g.dense_la.p + ((step + 1) & 1) * GJB * GJB, *sw = g.dense_la.p + (step & 1) * GJB * GJB;
        if (k0 + GJB < npad)
            hipLaunchKernelGGL(k_gj_update_la, dim3(nchunk * nchunk), dim3(GJ_LA_THREADS), 0, g.stream, npad, k0, A, wr,
                               wc, wrn, wcn, sr, sw, g.dense_maxdiag.p);
        else
            hipLaunchKernelGGL(k_gj_update, dim3(nchunk * nchunk), dim3(256), 0, g.stream, npad, k0, A, wr, wc);
        cur ^= 1;
    }
}

// y = E^-1 b on the dense level (3 columns). One wave per row, 16 rows per workgroup; b lives in
// LDS (one wave per row, 4 rows per workgroup so that ~n/4 workgroups spread over the chip). CHECK: PCG convergence prologue (when this is the first kernel after the PCG update).
// DOT: partial sums of b.y (the coarse part of r.z in the additive variant / the whole r.z when
// the dense level is level 0).
template <bool CHECK, bool DOT>
__global__ __launch_bounds__(kRowBlock) void k_dense_apply(
    int n, int npad, const double *__restrict__ Einv, double scale, const double4 *__restrict__ b,
    double4 *__restrict__ y, double *__restrict__ part_dot, const double *__restrict__ part_rr,
    int nparts, int first, double rtol2, double *__restrict__ scal, int *__restrict__ flags) {
    if (flags[FL_DONE]) return;
    if (CHECK && pcg_check(part_rr, nparts, first, rtol2, scal, flags)) return;
    extern __shared__ double sb[];  // 3 * npad
    for (int i = threadIdx.x; i < npad; i += blockDim.x) {
        double4 v = make_double4(0, 0, 0, 0);
        if (i < n) v = b[i];
        sb[i] = v.x;
        sb[npad + i] = v.y;
        sb[2 * npad + i] = v.z;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double a0 = 0, a1 = 0, a2 = 0;
    for (int rr = 0; rr < 1; rr++) {
        const int r = blockIdx.x * 4 + wv;
        double s0 = 0, s1 = 0, s2 = 0;
        if (r < n) {
            const double *row = Einv + (size_t)r * npad;
#pragma unroll 4
            for (int c = lane; c < npad; c += 64) {
                const double e = row[c];
                s0 += e * sb[c];
                s1 += e * sb[npad + c];
                s2 += e * sb[2 * npad + c];
            }
        }
        s0 = wave_sum(s0) * scale;
        s1 = wave_sum(s1) * scale;
        s2 = wave_sum(s2) * scale;
        if (lane == 0 && r < n) {
            y[r] = make_double4(s0, s1, s2, 0.0);
            if (DOT) {
                a0 += sb[r] * s0;
                a1 += sb[npad + r] * s1;
                a2 += sb[2 * npad + r] * s2;
            }
        }
    }
    if (DOT) block_sum3_store(a0, a1, a2, part_dot + 4 * blockIdx.x);
}

int dense_apply_grid(const Graph &g) { return (g.ndense + 3) / 4; }

void dense_apply(Graph &g, const double4 *b, double4 *y, bool check, bool dot, double *part_dot,
                 int np_rr, int first, double rtol2) {
    const int grid = dense_apply_grid(g);
    const size_t shm = sizeof(double) * 3 * (size_t)g.ndense_pad;
#define DA_LAUNCH(CH, DT)                                                                            \
    hipLaunchKernelGGL((k_dense_apply<CH, DT>), dim3(grid), dim3(kRowBlock), shm, g.stream, g.ndense, \
                       g.ndense_pad, g.dense_inv.p, g.dense_scale, b, y, part_dot, g.part_rr.p, np_rr, first, rtol2, \
                       g.scal.p, g.flags.p)
    if (check && dot)
        DA_LAUNCH(true, true);
    else if (check)
        DA_LAUNCH(true, false);
    else if (dot)
        DA_LAUNCH(false, true);
    else
        DA_LAUNCH(false, false);
#undef DA_LAUNCH
}

}  // namespace irh

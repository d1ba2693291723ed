// sweep_mfma.hip -- the blocked symmetric sweep on the matrix cores (built, measured, withdrawn)
// next to the scalar sweep of irotavg_amd/csrc/bcr.hip (bcr_invert, copied below as it stood), with a stand-alone test:
// random SPD blocks of 8 / 16 / 24 / 32, with and without a floating pair (dead pivot), both sweeps against each other
// and against A X = I, and the time per inversion of one wave.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/sweep_mfma.hip -o /tmp/sweep_mfma && /tmp/sweep_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include <cstdlib>
constexpr double kDeadTol = 1e-13;
__device__ __forceinline__ double bcr_readlane(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

template <int B>
__device__ __forceinline__ void bcr_invert(double *Dm, int lane) {
    constexpr int T = B / 8;
    const int a = lane >> 3, b = lane & 7;
    double t[T][T], od[T];
#pragma unroll
    for (int tr = 0; tr < T; tr++)
#pragma unroll
        for (int tc = 0; tc < T; tc++) t[tr][tc] = Dm[(a + 8 * tr) * B + b + 8 * tc];
#pragma unroll
    for (int tr = 0; tr < T; tr++) od[tr] = t[tr][tr];  // the diagonal where a == b
#pragma unroll
    for (int k = 0; k < B; k++) {
        const int kb = k & 7, kt = k >> 3;
        const int dl = (kb << 3) | kb;
        const double p = bcr_readlane(t[kt][kt], dl);
        const double ref = bcr_readlane(od[kt], dl);
        // reciprocal by v_rcp_f64 + two Newton steps (the IEEE division sequence is three times as long and
        // sits on the chain from pivot to pivot)
        double x = __builtin_amdgcn_rcp(p);
        x = fma(fma(-p, x, 1.0), x, x);
        x = fma(fma(-p, x, 1.0), x, x);
        const double pinv = (p > kDeadTol * ref) ? x : 0.0;
        double cr[T], cc[T];
#pragma unroll
        for (int tr = 0; tr < T; tr++) cr[tr] = __shfl(t[tr][kt], (a << 3) | kb, 64);
#pragma unroll
        for (int tc = 0; tc < T; tc++) cc[tc] = __shfl(t[tc][kt], (b << 3) | kb, 64);
        // row k lives in row tile kt of the lanes a == kb, column k in column tile kt of the lanes b == kb: only
        // those tiles need the selects (a 64-bit select is two VALU operations; with selects on all T x T
        // elements they, not the arithmetic, set the time per pivot)
        const bool rowk = a == kb, colk = b == kb;
        double crp[T];
#pragma unroll
        for (int tr = 0; tr < T; tr++) crp[tr] = cr[tr] * pinv;
#pragma unroll
        for (int tr = 0; tr < T; tr++)
#pragma unroll
            for (int tc = 0; tc < T; tc++) {
                const double upd = fma(-crp[tr], cc[tc], t[tr][tc]);
                if (tr == kt && tc == kt)
                    t[tr][tc] = rowk ? (colk ? -pinv : cc[tc] * pinv) : (colk ? crp[tr] : upd);
                else if (tr == kt)
                    t[tr][tc] = rowk ? cc[tc] * pinv : upd;
                else if (tc == kt)
                    t[tr][tc] = colk ? crp[tr] : upd;
                else
                    t[tr][tc] = upd;
            }
    }
#pragma unroll
    for (int tr = 0; tr < T; tr++)
#pragma unroll
        for (int tc = 0; tc < T; tc++) Dm[(a + 8 * tr) * B + b + 8 * tc] = -t[tr][tc];
}
// (experiment) blocked symmetric sweep on the matrix cores
typedef double v4d_ __attribute__((ext_vector_type(4)));
template <int B>
__device__ __forceinline__ void bcr_invert_mfma(double *Dm, int lane) {
    constexpr int MT = (B + 15) / 16, NG = B / 4;
    const int lj = lane & 15, lk = lane >> 4;
    v4d_ acc[MT][MT], od[MT];
    // rows beyond B are whole registers (B is a multiple of 4), columns beyond B are lanes of the last tile column
    const bool colin = 16 * (MT - 1) + lj < B;
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < MT; nt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = 16 * mt + 4 * r + lk, col = 16 * nt + lj;
                if (16 * mt + 4 * r < B) {
                    const int cc = (nt == MT - 1 && !colin) ? B - 1 : col;
                    const double v = Dm[row * B + cc];
                    acc[mt][nt][r] = (nt == MT - 1 && !colin) ? 0.0 : v;
                } else {
                    acc[mt][nt][r] = row == col ? 1.0 : 0.0;
                }
            }
#pragma unroll
    for (int mt = 0; mt < MT; mt++) od[mt] = acc[mt][mt];
    // 0 / 1 masks instead of nested selects (the compiler turns those into branches)
    double mk[4], mj[16];
#pragma unroll
    for (int k = 0; k < 4; k++) mk[k] = lk == k ? 1.0 : 0.0;
#pragma unroll
    for (int j = 0; j < 16; j++) mj[j] = lj == j ? 1.0 : 0.0;
#pragma unroll
    for (int g = 0; g < NG; g++) {
        constexpr int dummy = 0;
        (void)dummy;
        const int gm = g >> 2, gr = g & 3;
        double brow[MT], U[MT][4];
#pragma unroll
        for (int nt = 0; nt < MT; nt++) brow[nt] = acc[gm][nt][gr];
#pragma unroll
        for (int nt = 0; nt < MT; nt++)
#pragma unroll
            for (int kk = 0; kk < 4; kk++) U[nt][kk] = __shfl(brow[nt], lj + 16 * kk, 64);
        // the 4 x 4 pivot block (uniform) and the reference diagonal
        double s[4][4], ref[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            ref[k] = bcr_readlane(od[gm][gr], 16 * k + 4 * gr + k);
#pragma unroll
            for (int kk = k; kk < 4; kk++) {
                s[k][kk] = bcr_readlane(brow[gm], 16 * k + 4 * gr + kk);
                s[kk][k] = s[k][kk];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const double p = s[k][k];
            double x = __builtin_amdgcn_rcp(p);
            x = fma(fma(-p, x, 1.0), x, x);
            x = fma(fma(-p, x, 1.0), x, x);
            const double pinv = (p > kDeadTol * ref[k]) ? x : 0.0;
            double crp[4];
#pragma unroll
            for (int i = 0; i < 4; i++) crp[i] = s[i][k] * pinv;
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = i; j < 4; j++) {
                    if (i == k || j == k) continue;
                    s[i][j] = fma(-crp[i], s[k][j], s[i][j]);
                    s[j][i] = s[i][j];
                }
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (i != k) {
                    s[i][k] = crp[i];
                    s[k][i] = crp[i];
                }
            s[k][k] = -pinv;
        }
        // this lane's row of S
        double srow[4];
#pragma unroll
        for (int kk = 0; kk < 4; kk++)
            srow[kk] = fma(mk[3], s[3][kk], fma(mk[2], s[2][kk], fma(mk[1], s[1][kk], mk[0] * s[0][kk])));
        double aop[MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            double a = srow[0] * U[mt][0];
            a = fma(srow[1], U[mt][1], a);
            a = fma(srow[2], U[mt][2], a);
            a = fma(srow[3], U[mt][3], a);
            aop[mt] = a;
        }
        // columns of the group: accumulator cleared, B operand = -identity there
        const int cj = lj - 4 * gr;
        const bool colp = cj >= 0 && cj < 4;
        double bop[MT];
#pragma unroll
        for (int nt = 0; nt < MT; nt++) bop[nt] = brow[nt];
        const double negdelta = (cj == lk) ? -1.0 : 0.0;
        bop[gm] = colp ? negdelta : bop[gm];
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int r = 0; r < 4; r++) acc[mt][gm][r] = colp ? 0.0 : acc[mt][gm][r];
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int nt = 0; nt < MT; nt++)
                acc[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[mt], bop[nt], acc[mt][nt], 0, 0, 0);
        // rows of the group
#pragma unroll
        for (int nt = 0; nt < MT; nt++) acc[gm][nt][gr] = -aop[nt];
        {
            // (the masks of the group's columns: lanes lj == 4 gr + k')
            double sv = 0.0;
#pragma unroll
            for (int kk = 0; kk < 4; kk++) sv = fma(mj[4 * gr + kk], srow[kk], sv);
            acc[gm][gm][gr] = colp ? sv : acc[gm][gm][gr];
        }
    }
#pragma unroll
    for (int nt = 0; nt < MT; nt++) {
        if (nt == MT - 1 && !colin) continue;
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (16 * mt + 4 * r < B) Dm[(16 * mt + 4 * r + lk) * B + 16 * nt + lj] = -acc[mt][nt][r];
    }
}
template <int B, int WHICH> __global__ __launch_bounds__(512) void k(double *g, int reps) {
    __shared__ double Dm[B * B];
    for (int rep = 0; rep < reps; rep++) {
        for (int e = threadIdx.x; e < B * B; e += 64) Dm[e] = g[(size_t)blockIdx.x * B * B + e];
        __syncthreads();
        if (WHICH == 0) bcr_invert<B>(Dm, threadIdx.x); else bcr_invert_mfma<B>(Dm, threadIdx.x);
        __syncthreads();
    }
    for (int e = threadIdx.x; e < B * B; e += 64) g[(size_t)blockIdx.x * B * B + e] = Dm[e];
}
template <int B> int run(int dead) {
    const int nb = 256;
    std::vector<double> A((size_t)nb * B * B), X0, X1;
    srand(1);
    for (int b = 0; b < nb; b++) {
        std::vector<double> M(B * B);
        for (auto &v : M) v = rand() / (double)RAND_MAX - 0.5;
        for (int i = 0; i < B; i++) for (int j = 0; j < B; j++) { double s = 0; for (int k = 0; k < B; k++) s += M[i*B+k]*M[j*B+k]; A[(size_t)b*B*B + i*B + j] = s + (i==j ? 0.1 : 0); }
        if (dead && b % 2 == 0) { // a floating pair: rows/cols d, d+1 form a singular 2x2 laplacian, decoupled from the rest
            int d = (b / 2) % (B - 1);
            for (int i = 0; i < B; i++) { A[(size_t)b*B*B + i*B + d] = A[(size_t)b*B*B + d*B + i] = 0; A[(size_t)b*B*B + i*B + d+1] = A[(size_t)b*B*B + (d+1)*B + i] = 0; }
            A[(size_t)b*B*B + d*B + d] = 2.0; A[(size_t)b*B*B + (d+1)*B + d+1] = 2.0; A[(size_t)b*B*B + d*B + d+1] = A[(size_t)b*B*B + (d+1)*B + d] = -2.0;
        }
    }
    double *d; hipMalloc(&d, A.size() * 8);
    X0.resize(A.size()); X1.resize(A.size());
    hipMemcpy(d, A.data(), A.size()*8, hipMemcpyHostToDevice); k<B,0><<<nb,64>>>(d, 1); hipMemcpy(X0.data(), d, A.size()*8, hipMemcpyDeviceToHost);
    hipMemcpy(d, A.data(), A.size()*8, hipMemcpyHostToDevice); k<B,1><<<nb,64>>>(d, 1); hipMemcpy(X1.data(), d, A.size()*8, hipMemcpyDeviceToHost);
    double e01 = 0, eres0 = 0, eres1 = 0, xmax = 0;
    for (int b = 0; b < nb; b++) for (int i = 0; i < B; i++) for (int j = 0; j < B; j++) {
        size_t o = (size_t)b*B*B;
        e01 = fmax(e01, fabs(X0[o+i*B+j] - X1[o+i*B+j])); xmax = fmax(xmax, fabs(X0[o+i*B+j]));
        if (!dead) { double s0 = 0, s1 = 0; for (int k = 0; k < B; k++) { s0 += A[o+i*B+k]*X0[o+k*B+j]; s1 += A[o+i*B+k]*X1[o+k*B+j]; } eres0 = fmax(eres0, fabs(s0 - (i==j))); eres1 = fmax(eres1, fabs(s1 - (i==j))); }
    }
    // timing: one workgroup, many reps
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms0, ms1;
    hipMemcpy(d, A.data(), A.size()*8, hipMemcpyHostToDevice);
    k<B,0><<<1,64>>>(d, 10); hipEventRecord(e0); k<B,0><<<1,64>>>(d, 2000); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms0, e0, e1);
    hipMemcpy(d, A.data(), A.size()*8, hipMemcpyHostToDevice);
    k<B,1><<<1,64>>>(d, 10); hipEventRecord(e0); k<B,1><<<1,64>>>(d, 2000); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms1, e0, e1);
    printf("B %d dead %d: scalar vs mfma max diff %.3e (max |x| %.3e), |AX-I| scalar %.3e mfma %.3e; us per inversion (incl. LDS load): scalar %.2f mfma %.2f\n", B, dead, e01, xmax, eres0, eres1, ms0 / 2000 * 1e3, ms1 / 2000 * 1e3);
    hipFree(d);
    return 0;
}
int main() { run<8>(0); run<16>(0); run<24>(0); run<32>(0); run<8>(1); run<16>(1); run<24>(1); run<32>(1); return 0; }

// sweep_cols.hip -- the symmetric sweep (in-place inverse of an SPD B x B block by ONE wave) in a column-per-lane
// layout, next to the 8 x 8-lane register-tile sweep it replaces (bcr_invert of round 3, copied below as it stood).
// Stand-alone test: random SPD blocks with and without a floating pair (dead pivots), both sweeps against each other and
// against A X = I, and the time per inversion of one wave alone / of four waves of a workgroup.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/sweep_cols.hip -o /tmp/sweep_cols && /tmp/sweep_cols
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include <cstdlib>
constexpr double kDeadTol = 1e-13;
typedef double v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double bcr_readlane(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

// ---- round 3 ----
template <int B>
__device__ __forceinline__ void bcr_invert_old(double *Dm, int lane) {
    constexpr int T = B / 8;
    const int a = lane >> 3, b = lane & 7;
    double t[T][T], od[T];
#pragma unroll
    for (int tr = 0; tr < T; tr++)
#pragma unroll
        for (int tc = 0; tc < T; tc++) t[tr][tc] = Dm[(a + 8 * tr) * B + b + 8 * tc];
#pragma unroll
    for (int tr = 0; tr < T; tr++) od[tr] = t[tr][tr];
#pragma unroll
    for (int k = 0; k < B; k++) {
        const int kb = k & 7, kt = k >> 3;
        const int dl = (kb << 3) | kb;
        const double p = bcr_readlane(t[kt][kt], dl);
        const double ref = bcr_readlane(od[kt], dl);
        double x = __builtin_amdgcn_rcp(p);
        x = fma(fma(-p, x, 1.0), x, x);
        x = fma(fma(-p, x, 1.0), x, x);
        const double pinv = (p > kDeadTol * ref) ? x : 0.0;
        double cr[T], cc[T];
#pragma unroll
        for (int tr = 0; tr < T; tr++) cr[tr] = __shfl(t[tr][kt], (a << 3) | kb, 64);
#pragma unroll
        for (int tc = 0; tc < T; tc++) cc[tc] = __shfl(t[tc][kt], (b << 3) | kb, 64);
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

// ---- round 4: a column per lane ----
// Lane (c, h) = (lane % 32, lane / 32) holds the rows h HB .. h HB + HB - 1 of column c (HB = B / 2). The sweep stays
// symmetric, so column k IS row k: per pivot the lanes that hold row k write it to LDS (one store), and every lane reads
// back its own column's entry (the multiplier of the pivot row) and the HB entries of its rows (the pivot column) --
// broadcast reads, no lane permutes, no address arithmetic. The pivot itself and its dead-pivot reference travel through
// SGPRs (v_readlane), so the reciprocal -- the longest dependent piece of a pivot -- starts before the LDS round trip of
// the row has finished (VAR >= 1). LDS scratch: the first 2 B doubles of the block itself (it lives in registers
// during the sweep).
template <int B, int VAR>
__device__ __forceinline__ void bcr_invert_cols(double *Dm, int lane) {
    constexpr int HB = B / 2;
    const int c = lane & 31, h = lane >> 5;
    const bool act = c < B;
    const int cl = act ? c : B - 1;
    double t[HB];
#pragma unroll
    for (int i = 0; i < HB; i++) t[i] = Dm[(h * HB + i) * B + cl];
    const double dg = Dm[cl * B + cl];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < B; k++) {
        constexpr int dummy = 0;
        (void)dummy;
        const int kh = k / HB, ki = k - kh * HB;
        // row k -> LDS
        if (h == kh && act) Dm[c] = t[ki];
        __builtin_amdgcn_wave_barrier();
        double p, ref;
        if (VAR >= 1) {
            p = bcr_readlane(t[ki], k + 32 * kh);
        } else {
            p = Dm[k];
        }
        ref = bcr_readlane(dg, k);
        double x = __builtin_amdgcn_rcp(p);
        x = fma(fma(-p, x, 1.0), x, x);
        x = fma(fma(-p, x, 1.0), x, x);
        const double pinv = (p > kDeadTol * ref) ? x : 0.0;
        const double rowk = Dm[cl];
        double colk[HB];
#pragma unroll
        for (int i = 0; i < HB; i += 2) {
            const v2d v = *reinterpret_cast<const v2d *>(&Dm[h * HB + i]);
            colk[i] = v.x;
            colk[i + 1] = v.y;
        }
        __builtin_amdgcn_wave_barrier();
        const bool isk = c == k;
        const double f = rowk * pinv;
        const double g = isk ? -pinv : f;
#pragma unroll
        for (int i = 0; i < HB; i++) {
            const double a = isk ? 0.0 : t[i];
            double u = fma(-colk[i], g, a);
            if (i == ki) u = (h == kh) ? g : u;   // row k: f, and -pinv at (k, k)
            t[i] = u;
        }
    }
#pragma unroll
    for (int i = 0; i < HB; i++)
        if (act) Dm[(h * HB + i) * B + c] = -t[i];
}


// VAR 2: the same sweep software-pipelined by hand. A wave issues in order, so the chain of a pivot (row -> LDS ->
// back, reciprocal + two Newton steps) and the issue time of its HB updates ADD unless independent work stands between
// the links: the element of row k + 1 is updated first, written, its reciprocal started and the reads of row k + 1
// issued; the other HB - 1 updates of pivot k follow and cover those latencies.
template <int B>
__device__ __forceinline__ void bcr_invert_pipe(double *Dm, int lane) {
    constexpr int HB = B / 2;
    const int c = lane & 31, h = lane >> 5;
    const bool act = c < B;
    const int cl = act ? c : B - 1;
    double *Dv = Dm;
    const v2d *Dv2 = reinterpret_cast<const v2d *>(Dm + h * HB);
    double t[HB];
#pragma unroll
    for (int i = 0; i < HB; i++) t[i] = Dm[(h * HB + i) * B + cl];
    const double dg = Dm[cl * B + cl];
    __builtin_amdgcn_wave_barrier();
    // prologue: row 0
    if (h == 0 && act) Dv[c] = t[0];
    asm volatile("" ::: "memory");
    double pinv;
    {
        const double p = bcr_readlane(t[0], 0), ref = bcr_readlane(dg, 0);
        double x = __builtin_amdgcn_rcp(p);
        x = fma(fma(-p, x, 1.0), x, x);
        x = fma(fma(-p, x, 1.0), x, x);
        pinv = (p > kDeadTol * ref) ? x : 0.0;
    }
    double rowk = Dv[cl], colk[HB];
#pragma unroll
    for (int i = 0; i < HB; i += 2) {
        const v2d v = Dv2[i / 2];
        colk[i] = v.x;
        colk[i + 1] = v.y;
    }
#pragma unroll
    for (int k = 0; k < B; k++) {
        const int kh = k / HB, ki = k - kh * HB;
        const int k1 = k + 1, kh1 = k1 / HB, ki1 = k1 - kh1 * HB;   // next pivot (k1 < B)
        const bool isk = c == k;
        const double f = rowk * pinv;
        const double g = isk ? -pinv : f;
        auto upd = [&](int i) {
            const double a = isk ? 0.0 : t[i];
            double u = fma(-colk[i], g, a);
            if (i == ki) u = (h == kh) ? g : u;
            t[i] = u;
        };
        double pinv1 = 0.0, rowk1 = 0.0, colk1[HB];
        if (k1 < B) {
            upd(ki1);
            if (h == kh1 && act) Dv[c] = t[ki1];
            asm volatile("" ::: "memory");
            const double p = bcr_readlane(t[ki1], k1 + 32 * kh1), ref = bcr_readlane(dg, k1);
            rowk1 = Dv[cl];
#pragma unroll
            for (int i = 0; i < HB; i += 2) {
                const v2d v = Dv2[i / 2];
                colk1[i] = v.x;
                colk1[i + 1] = v.y;
            }
            double x = __builtin_amdgcn_rcp(p);
            x = fma(fma(-p, x, 1.0), x, x);
            x = fma(fma(-p, x, 1.0), x, x);
            pinv1 = (p > kDeadTol * ref) ? x : 0.0;
        }
#pragma unroll
        for (int i = 0; i < HB; i++)
            if (!(k1 < B && i == ki1)) upd(i);
        // pin the right-looking order: without it the compiler defers an element's updates until its row is the
        // pivot row -- k dependent FMAs in a row on the critical path instead of HB independent ones per pivot
#pragma unroll
        for (int i = 0; i < HB; i++) asm volatile("" : "+v"(t[i]));
        if (k1 < B) {
            pinv = pinv1;
            rowk = rowk1;
#pragma unroll
            for (int i = 0; i < HB; i++) colk[i] = colk1[i];
        }
    }
#pragma unroll
    for (int i = 0; i < HB; i++)
        if (act) Dm[(h * HB + i) * B + c] = -t[i];
}

template <int B, int WHICH>
__global__ __launch_bounds__(256) void k(double *g, int reps) {
    __shared__ double Dm[4][B * B];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t blk = (size_t)blockIdx.x * (blockDim.x >> 6) + wave;
    for (int rep = 0; rep < reps; rep++) {
        for (int e = lane; e < B * B; e += 64) Dm[wave][e] = g[blk * B * B + e];
        __syncthreads();
        if (WHICH == 0) bcr_invert_old<B>(Dm[wave], lane);
        else if (WHICH == 3) bcr_invert_pipe<B>(Dm[wave], lane);
        else bcr_invert_cols<B, WHICH - 1>(Dm[wave], lane);
        __syncthreads();
    }
    for (int e = lane; e < B * B; e += 64) g[blk * B * B + e] = Dm[wave][e];
}

template <int B, int WHICH>
static float time_it(double *d, int nwg, int nthr) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms;
    k<B, WHICH><<<nwg, nthr>>>(d, 10);
    hipEventRecord(e0);
    k<B, WHICH><<<nwg, nthr>>>(d, 2000);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 2000 * 1e3f;
}

template <int B>
int run(int dead) {
    const int nb = 256;
    std::vector<double> A((size_t)nb * B * B), X0, X1, X2;
    srand(1);
    for (int b = 0; b < nb; b++) {
        std::vector<double> M(B * B);
        for (auto &v : M) v = rand() / (double)RAND_MAX - 0.5;
        for (int i = 0; i < B; i++)
            for (int j = 0; j < B; j++) {
                double s = 0;
                for (int kk = 0; kk < B; kk++) s += M[i * B + kk] * M[j * B + kk];
                A[(size_t)b * B * B + i * B + j] = s + (i == j ? 0.1 : 0);
            }
        if (dead && b % 2 == 0) {
            int d = (b / 2) % (B - 1);
            for (int i = 0; i < B; i++) {
                A[(size_t)b * B * B + i * B + d] = A[(size_t)b * B * B + d * B + i] = 0;
                A[(size_t)b * B * B + i * B + d + 1] = A[(size_t)b * B * B + (d + 1) * B + i] = 0;
            }
            A[(size_t)b * B * B + d * B + d] = 2.0;
            A[(size_t)b * B * B + (d + 1) * B + d + 1] = 2.0;
            A[(size_t)b * B * B + d * B + d + 1] = A[(size_t)b * B * B + (d + 1) * B + d] = -2.0;
        }
    }
    double *d;
    hipMalloc(&d, A.size() * 8);
    X0.resize(A.size());
    X1.resize(A.size());
    X2.resize(A.size());
    hipMemcpy(d, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    k<B, 0><<<nb, 64>>>(d, 1);
    hipMemcpy(X0.data(), d, A.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(d, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    k<B, 3><<<nb, 64>>>(d, 1);
    hipMemcpy(X1.data(), d, A.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(d, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    k<B, 2><<<nb / 4, 256>>>(d, 1);
    hipMemcpy(X2.data(), d, A.size() * 8, hipMemcpyDeviceToHost);
    double e01 = 0, e02 = 0, eres0 = 0, eres1 = 0, eres2 = 0, xmax = 0, asym = 0;
    for (int b = 0; b < nb; b++)
        for (int i = 0; i < B; i++)
            for (int j = 0; j < B; j++) {
                size_t o = (size_t)b * B * B;
                e01 = fmax(e01, fabs(X0[o + i * B + j] - X1[o + i * B + j]));
                e02 = fmax(e02, fabs(X0[o + i * B + j] - X2[o + i * B + j]));
                asym = fmax(asym, fabs(X2[o + i * B + j] - X2[o + j * B + i]));
                xmax = fmax(xmax, fabs(X0[o + i * B + j]));
                if (!dead) {
                    double s0 = 0, s1 = 0, s2 = 0;
                    for (int kk = 0; kk < B; kk++) {
                        s0 += A[o + i * B + kk] * X0[o + kk * B + j];
                        s1 += A[o + i * B + kk] * X1[o + kk * B + j];
                        s2 += A[o + i * B + kk] * X2[o + kk * B + j];
                    }
                    eres0 = fmax(eres0, fabs(s0 - (i == j)));
                    eres1 = fmax(eres1, fabs(s1 - (i == j)));
                    eres2 = fmax(eres2, fabs(s2 - (i == j)));
                }
            }
    hipMemcpy(d, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    const float a0 = time_it<B, 0>(d, 1, 64), a1 = time_it<B, 3>(d, 1, 64), a2 = time_it<B, 2>(d, 1, 64);
    hipMemcpy(d, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    const float b0 = time_it<B, 0>(d, 1, 256), b1 = time_it<B, 3>(d, 1, 256), b2 = time_it<B, 2>(d, 1, 256);
    hipMemcpy(d, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    const float c0 = time_it<B, 0>(d, 64, 256), c2 = time_it<B, 2>(d, 64, 256);
    printf("B %2d dead %d: old vs pipe %.3e cols1 %.3e (max |x| %.3e, asym %.1e) |AX-I| old %.3e pipe %.3e cols1 %.3e; "
           "us/inversion (incl. LDS load + 2 barriers): 1 wave old %.2f pipe %.2f cols1 %.2f; 4 waves old %.2f pipe %.2f "
           "cols1 %.2f; 64 WGs x 4 waves old %.2f cols1 %.2f\n",
           B, dead, e01, e02, xmax, asym, eres0, eres1, eres2, a0, a1, a2, b0, b1, b2, c0, c2);
    hipFree(d);
    return 0;
}
int main() {
    run<8>(0);
    run<16>(0);
    run<24>(0);
    run<32>(0);
    run<8>(1);
    run<16>(1);
    run<24>(1);
    run<32>(1);
    return 0;
}

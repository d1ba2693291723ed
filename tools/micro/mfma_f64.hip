// mfma_f64.hip -- what v_mfma_f64_16x16x4_f64 costs on gfx950: issue interval of independent and of dependent MFMAs of one
// wave, of 1 / 2 waves per SIMD, next to f64 FMAs of a second wave on the same SIMD (do they share the FP64 lanes?).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_f64.hip -o build/micro/mfma_f64
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
// mode 0: 8 independent accumulators; 1: one dependent chain; 2: VALU f64 FMAs (8 independent chains) instead;
// 3: even waves MFMA (independent), odd waves VALU; 4: v_mfma_f64_4x4x4 (four blocks), 8 independent accumulators;
// 5: 4x4x4, one dependent chain; 6: waves 0-3 VALU f64 (one per SIMD), waves 4-7 MFMA on the SAME SIMDs (does a sweep
// slow down when another wave of its SIMD issues products?); 7: waves 0-3 VALU alone in an eight-wave workgroup (reference for 6)
__global__ __launch_bounds__(1024) void k(double *out, long long *clk, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    v4d acc[8];
    for (int i = 0; i < 8; i++) acc[i] = v4d{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    double f[8];
    for (int i = 0; i < 8; i++) f[i] = i;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    const bool valu = mode == 2 || (mode == 3 && wave >= (int)(blockDim.x >> 7)) || ((mode == 6 || mode == 7) && wave < 4);
    double c4[8];
    for (int i = 0; i < 8; i++) c4[i] = 0.0;
    if (mode == 7 && !valu) {
    } else if (!valu && (mode == 4 || mode == 5)) {
        if (mode == 5) {
            for (int it = 0; it < iters; it++)
#pragma unroll
                for (int i = 0; i < 8; i++) c4[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4[0], 0, 0, 0);
        } else {
            for (int it = 0; it < iters; it++)
#pragma unroll
                for (int i = 0; i < 8; i++) c4[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4[i], 0, 0, 0);
        }
    } else if (!valu) {
        if (mode == 1) {
            for (int it = 0; it < iters; it++)
#pragma unroll
                for (int i = 0; i < 8; i++) acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[0], 0, 0, 0);
        } else {
            for (int it = 0; it < iters; it++)
#pragma unroll
                for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
    } else {
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int i = 0; i < 8; i++) f[i] = fma(f[i], b, a);
    }
    const long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + f[i] + c4[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 16 + wave] = t1 - t0;
}
int main() {
    double *out;
    long long *clk, h[16];
    hipMalloc(&out, 8 * 1024 * 16);
    hipMalloc(&clk, 16 * 8 * 16);
    const int iters = 2000;
    const char *names[] = {"mfma, 8 independent accumulators", "mfma, one dependent chain", "v_fma_f64, 8 independent chains",
                           "first half mfma / second half v_fma_f64", "mfma 4x4x4, 8 independent accumulators",
                           "mfma 4x4x4, one dependent chain", "waves 0-3 v_fma_f64, waves 4.. mfma 16x16x4",
                           "waves 0-3 v_fma_f64, the others idle"};
    for (int mode = 0; mode < 8; mode++)
        for (int nw : {1, 4, 8, 16}) {
            if (mode >= 6 && nw < 8) continue;
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            k<<<1, nw * 64>>>(out, clk, 10, mode);
            hipEventRecord(e0);
            k<<<1, nw * 64>>>(out, clk, iters, mode);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
            printf("%-40s %2d waves: %.1f clocks per op per wave (wave 0), %.1f (wave 1), %.1f (last); kernel %.1f us -> %.2f GHz\n", names[mode], nw,
                   h[0] / (8.0 * iters), nw > 1 ? h[1] / (8.0 * iters) : 0.0, h[nw - 1] / (8.0 * iters), ms * 1e3, h[0] / (ms * 1e-3) / 1e9);
        }
    return 0;
}

// wave_place.hip -- do the first four waves of an eight-wave workgroup run their fp64 chains at full speed whatever the
// kernel's register allocation, LDS size and scratch use? (round 5: the sweeps of k_bcr_reduce_up ran 3.5 x slower when
// four of them ran at the same time than in the stand-alone kernel of the same body.)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/wave_place.hip -o build/micro/wave_place
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ double dyn[];
template <int MODE>
__device__ __forceinline__ void work(long long *clk, unsigned *hw, int iters, int nactive) {
    const int wave = threadIdx.x >> 6;
    double f[8];
    for (int i = 0; i < 8; i++) f[i] = i + threadIdx.x * 1e-3;
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1e-3;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    if (wave < nactive) {
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int i = 0; i < 8; i++) f[i] = fma(f[i], a, b);
    }
    const long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    double s = 0;
    for (int i = 0; i < 8; i++) s += f[i];
    if (s == 123.456) dyn[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        clk[wave] = t1 - t0;
        hw[wave] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
    }
}
__global__ __launch_bounds__(512, 1) void k_plain(long long *clk, unsigned *hw, int iters, int nactive) { work<0>(clk, hw, iters, nactive); }
__global__ __launch_bounds__(512, 1) void k_bigvgpr(long long *clk, unsigned *hw, int iters, int nactive) {
    asm volatile("v_mov_b32 v250, 0" ::: "v250");
    work<1>(clk, hw, iters, nactive);
}
__device__ __noinline__ void callee(long long *clk, unsigned *hw, int iters, int nactive) {
    asm volatile("v_mov_b32 v240, 0" ::: "v240");
    work<2>(clk, hw, iters, nactive);
}
__global__ __launch_bounds__(512, 1) void k_call(long long *clk, unsigned *hw, int iters, int nactive) { callee(clk, hw, iters, nactive); }
__global__ __launch_bounds__(512, 1) void k_scratch(long long *clk, unsigned *hw, int iters, int nactive) {
    volatile double buf[64];
    for (int i = 0; i < 64; i++) buf[i] = i;
    asm volatile("v_mov_b32 v250, 0" ::: "v250");
    work<3>(clk, hw, iters, nactive);
    if (buf[threadIdx.x & 63] == -1.0) clk[9] = 1;
}
int main() {
    long long *clk, h[16];
    unsigned *hw, hh[16];
    hipMalloc(&clk, 16 * 8);
    hipMalloc(&hw, 16 * 4);
    const int iters = 2000;
    const char *names[] = {"plain (few registers)", "256 registers", "noinline callee, 240+ registers", "256 registers + scratch"};
    for (int mode = 0; mode < 4; mode++)
        for (size_t lds : {(size_t)0, (size_t)100 * 1024})
            for (int na : {1, 2, 4, 8}) {
                if (lds) {
                    hipFuncSetAttribute((const void *)k_plain, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                    hipFuncSetAttribute((const void *)k_bigvgpr, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                    hipFuncSetAttribute((const void *)k_call, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                    hipFuncSetAttribute((const void *)k_scratch, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                }
                for (int rep = 0; rep < 2; rep++) {
                    if (mode == 0) k_plain<<<1, 512, lds>>>(clk, hw, iters, na);
                    if (mode == 1) k_bigvgpr<<<1, 512, lds>>>(clk, hw, iters, na);
                    if (mode == 2) k_call<<<1, 512, lds>>>(clk, hw, iters, na);
                    if (mode == 3) k_scratch<<<1, 512, lds>>>(clk, hw, iters, na);
                }
                hipDeviceSynchronize();
                hipMemcpy(h, clk, sizeof(long long) * 8, hipMemcpyDeviceToHost);
                hipMemcpy(hh, hw, sizeof(unsigned) * 8, hipMemcpyDeviceToHost);
                printf("%-34s lds %6zu, %d active: clocks per fma of waves 0..%d:", names[mode], lds, na, na - 1);
                for (int w = 0; w < na; w++) printf(" %.1f", h[w] / (8.0 * iters));
                printf("   hw_id[7:0] of waves 0..7:");
                for (int w = 0; w < 8; w++) printf(" %02x", hh[w] & 0xff);
                printf("\n");
            }
    return 0;
}

// micro-benchmark: cost of a device-wide barrier between co-resident workgroups on MI355X (gfx950),
// against the cost of a dependent kernel launch. Development aid for the persistent PCG kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ void grid_barrier(unsigned *ctr, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

// variant: per-XCD counters first (blockIdx % 8), then one global counter of 8 arrivals
__device__ __forceinline__ void grid_barrier2(unsigned *ctr, unsigned gen, unsigned nwg) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned x = blockIdx.x & 7;
        const unsigned per = (nwg + 7 - x) / 8;  // workgroups with this residue
        unsigned old = __hip_atomic_fetch_add(ctr + 32 * (1 + x), 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == gen * per) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gen * 8) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

__global__ __launch_bounds__(256, 2) void k_bar(unsigned *ctr, int n, double *data, int work, int variant) {
    double acc = 0;
    for (int i = 1; i <= n; i++) {
        for (int w = 0; w < work; w++) acc += data[(blockIdx.x * 256 + threadIdx.x + 4096 * w) & 0xfffff];
        if (variant == 0) grid_barrier(ctr, (unsigned)i * gridDim.x);
        else grid_barrier2(ctr, (unsigned)i, gridDim.x);
    }
    if (acc == 12345.678) data[0] = acc;
}
__global__ __launch_bounds__(256, 2) void k_empty(double *data, int work) {
    double acc = 0;
    for (int w = 0; w < work; w++) acc += data[(blockIdx.x * 256 + threadIdx.x + 4096 * w) & 0xfffff];
    if (acc == 12345.678) data[0] = acc;
}

int main(int argc, char **argv) {
    int grid = argc > 1 ? atoi(argv[1]) : 392;
    int n = 2000;
    unsigned *ctr; double *data;
    CK(hipMalloc(&ctr, 4096)); CK(hipMalloc(&data, 8 << 20));
    CK(hipMemset(data, 0, 8 << 20));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_bar, 256, 0));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("CUs %d, occupancy %d blocks/CU -> %d co-resident, grid %d\n", p.multiProcessorCount, occ, occ * p.multiProcessorCount, grid);
    if (grid > occ * p.multiProcessorCount) { printf("grid not co-resident: refusing\n"); return 2; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int variant = 0; variant < 2; variant++)
    for (int work = 0; work <= 8; work += 8) {
        CK(hipMemset(ctr, 0, 4096));
        hipLaunchKernelGGL(k_bar, dim3(grid), dim3(256), 0, 0, ctr, 10, data, work, variant);
        CK(hipDeviceSynchronize());
        CK(hipMemset(ctr, 0, 4096));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_bar, dim3(grid), dim3(256), 0, 0, ctr, n, data, work, variant);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("variant %d work %d: %.3f us per barrier-iteration\n", variant, work, 1e3 * ms / n);
    }
    for (int work = 0; work <= 8; work += 8) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, 0, data, work);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("dependent launches, work %d: %.3f us per launch\n", work, 1e3 * ms / n);
    }
    return 0;
}

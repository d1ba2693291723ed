// micro-benchmark: where should the inputs of a single-workgroup kernel live? (a) pinned host memory, read by the
// kernel over PCIe; (b) fine-grained DEVICE memory that the host writes through the PCIe BAR (large BAR).
// hipcc --offload-arch=gfx950 -O3 hostwrite.hip -o hostwrite && ./hostwrite
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_read(const double *in, int n, double *out, volatile int *seq, int s) {
    __shared__ double sm[2048];
    for (int i = threadIdx.x; i < n; i += blockDim.x) sm[i] = in[i];
    __syncthreads();
    double a = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) a += sm[i];
    out[threadIdx.x] = a;
    __threadfence_system();
    if (threadIdx.x == 0) *seq = s;
}
int main() {
    int large = 0;
    CK(hipDeviceGetAttribute(&large, hipDeviceAttributeIsLargeBar, 0));
    printf("large BAR: %d\n", large);
    const int n = 1664;  // 13 KB of doubles
    double *hp, *hpd, *dv = nullptr, *out;
    int *seqh, *seqd;
    CK(hipHostMalloc((void **)&hp, n * 8, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostGetDevicePointer((void **)&hpd, hp, 0));
    CK(hipHostMalloc((void **)&seqh, 64, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostGetDevicePointer((void **)&seqd, seqh, 0));
    CK(hipMalloc((void **)&out, 256 * 8));
    hipError_t e = hipExtMallocWithFlags((void **)&dv, n * 8, hipDeviceMallocFinegrained);
    printf("hipExtMallocWithFlags(finegrained): %s\n", hipGetErrorString(e));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int variant = 0; variant < (large && e == hipSuccess ? 2 : 1); variant++) {
        double *src = variant ? dv : hp;
        const double *ksrc = variant ? dv : hpd;
        double best = 1e9, tw = 0;
        for (int rep = 0; rep < 200; rep++) {
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < n; i++) src[i] = rep + i;   // the host fills the inputs
            auto t1 = std::chrono::steady_clock::now();
            *seqh = 0;
            hipLaunchKernelGGL(k_read, dim3(1), dim3(256), 0, st, ksrc, n, out, seqd, rep + 1);
            while (*(volatile int *)seqh != rep + 1) {}
            auto t2 = std::chrono::steady_clock::now();
            const double us = std::chrono::duration<double, std::micro>(t2 - t1).count();
            if (rep > 20 && us < best) best = us;
            tw = std::chrono::duration<double, std::micro>(t1 - t0).count();
        }
        printf("%s: host fill %.2f us, launch..result visible %.2f us (best)\n", variant ? "device memory written through the BAR" : "pinned host memory", tw, best);
    }
    return 0;
}

// micro-test (round 4): which workgroups of a launch share a CU, and on which SIMD each of their waves sits. 512 workgroups
// of 256 threads with 79 KB of LDS each (the footprint of the level-0 reduction: two per CU) stay resident until all have
// started, and report XCC_ID and HW_ID (gfx9: wave [3:0], SIMD [5:4], CU [11:8], SH [12], SE [15:13]) per wave.
//   hipcc --offload-arch=gfx950 -O3 cu_pairs.hip -o cu_pairs && ./cu_pairs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ __launch_bounds__(256, 2) void k_where(unsigned *out, unsigned *ctr, int G) {
    extern __shared__ double lds[];
    lds[threadIdx.x] = 1.0;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = (hw & 0xffff) | ((xcc & 0xf) << 16);
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int spin = 0; spin < (1 << 20); spin++) {
            if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)G) break;
            __builtin_amdgcn_s_sleep(4);
        }
    }
    __syncthreads();
    if (lds[threadIdx.x] != 1.0) out[0] = 0;
}

int main() {
    const int G = 512;
    unsigned *out, *ctr;
    CK(hipMalloc(&out, sizeof(unsigned) * G * 4));
    CK(hipMalloc(&ctr, sizeof(unsigned)));
    CK(hipMemset(ctr, 0, sizeof(unsigned)));
    CK(hipFuncSetAttribute((const void *)k_where, hipFuncAttributeMaxDynamicSharedMemorySize, 79 * 1024));
    hipLaunchKernelGGL(k_where, dim3(G), dim3(256), 79 * 1024, 0, out, ctr, G);
    CK(hipDeviceSynchronize());
    std::vector<unsigned> h(G * 4);
    CK(hipMemcpy(h.data(), out, sizeof(unsigned) * G * 4, hipMemcpyDeviceToHost));
    std::map<unsigned, std::vector<int>> cu;   // (xcc, se, sh, cu) -> workgroups
    for (int b = 0; b < G; b++) {
        const unsigned v = h[b * 4];
        const unsigned key = ((v >> 16) << 16) | (v & 0xff00);
        cu[key].push_back(b);
    }
    printf("%zu distinct CUs for %d workgroups\n", cu.size(), G);
    int shown = 0;
    for (auto &kv : cu) {
        if (shown++ >= 12) break;
        printf("xcc %u se %u sh %u cu %2u:", kv.first >> 16, (kv.first >> 13) & 7, (kv.first >> 12) & 1, (kv.first >> 8) & 15);
        for (int b : kv.second) {
            printf("  wg %3d simd of waves 0-3:", b);
            for (int w = 0; w < 4; w++) printf(" %u", (h[b * 4 + w] >> 4) & 3);
        }
        printf("\n");
    }
    // how the partner's index relates
    std::map<int, int> diff;
    for (auto &kv : cu)
        if (kv.second.size() == 2) diff[kv.second[1] - kv.second[0]]++;
    for (auto &d : diff) printf("partner distance %d: %d CUs\n", d.first, d.second);
    return 0;
}

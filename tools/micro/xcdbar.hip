// micro-benchmark (round 4): a barrier + data exchange among workgroups that all sit on ONE XCD (the grid is 8 x the
// workgroups wanted, only blockIdx % 8 == 0 takes part: workgroups are dealt to the XCDs round-robin), with and without
// device-scope fences, against the same among workgroups on all XCDs. Every round each workgroup publishes a value and
// reads its neighbour's; wrong reads are counted.   hipcc --offload-arch=gfx950 -O3 xcdbar.hip -o xcdbar && ./xcdbar [G]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

// mode 0: release / acquire at device scope (fences); mode 1: relaxed atomics only (no fence); mode 2: plain store of the
// value, relaxed counter, plain load (what is coherent inside one L2 if the reader's L1 does not hold the line)
template <int MODE>
__global__ __launch_bounds__(256) void k_xbar(unsigned *ctr, unsigned long long *slot, int rounds, int stride, int G,
                                              unsigned *bad, unsigned *xccs) {
    if (blockIdx.x % stride != 0) return;
    const int me = blockIdx.x / stride;
    if (threadIdx.x == 0) xccs[me] = xcc_id();
    unsigned wrong = 0;
    for (int r = 1; r <= rounds; r++) {
        if (threadIdx.x == 0) {
            const unsigned long long v = ((unsigned long long)r << 32) | (unsigned)me;
            unsigned long long *mine = slot + 16 * ((size_t)me + (size_t)G * (r & 1));
            if (MODE == 0) {
                *mine = v;
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)r * G) __builtin_amdgcn_s_sleep(1);
            } else if (MODE == 1) {
                __hip_atomic_store(mine, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_s_waitcnt(0);
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)r * G) __builtin_amdgcn_s_sleep(1);
            } else {
                *mine = v;
                __builtin_amdgcn_s_waitcnt(0);
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)r * G) __builtin_amdgcn_s_sleep(1);
            }
            const int nb = (me + 1) % G;
            const unsigned long long *theirs = slot + 16 * ((size_t)nb + (size_t)G * (r & 1));
            unsigned long long got;
            if (MODE == 1) got = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (MODE == 2) got = __builtin_nontemporal_load(theirs);
            else got = *theirs;
            if (got != (((unsigned long long)r << 32) | (unsigned)nb)) wrong++;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && wrong) atomicAdd(bad, wrong);
}

template <int MODE>
static void run(const char *what, int G, int stride, unsigned *ctr, unsigned long long *slot, unsigned *bad, unsigned *xccs) {
    const int rounds = 2000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMemset(ctr, 0, 64)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(slot, 0, 16 * 8 * 2 * 512));
    hipLaunchKernelGGL(k_xbar<MODE>, dim3(G * stride), dim3(256), 0, 0, ctr, slot, 10, stride, G, bad, xccs);
    CK(hipDeviceSynchronize());
    CK(hipMemset(ctr, 0, 64)); CK(hipMemset(bad, 0, 4));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_xbar<MODE>, dim3(G * stride), dim3(256), 0, 0, ctr, slot, rounds, stride, G, bad, xccs);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned hb = 0, hx[512];
    CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hx, xccs, 4 * G, hipMemcpyDeviceToHost));
    unsigned mask = 0;
    for (int i = 0; i < G; i++) mask |= 1u << hx[i];
    printf("%-44s G %3d stride %d: %7.3f us per round, wrong reads %u of %d, XCC ids seen 0x%02x\n", what, G, stride,
           1e3 * ms / rounds, hb, rounds * G, mask);
}

int main(int argc, char **argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 16;
    unsigned *ctr, *bad, *xccs; unsigned long long *slot;
    CK(hipMalloc(&ctr, 64)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&xccs, 4 * 512)); CK(hipMalloc(&slot, 16 * 8 * 2 * 512));
    for (int stride = 8; stride >= 1; stride -= 7) {
        run<0>("release/acquire at device scope", G, stride, ctr, slot, bad, xccs);
        run<1>("relaxed device-scope atomics, no fence", G, stride, ctr, slot, bad, xccs);
        run<2>("plain store, relaxed counter, nt load", G, stride, ctr, slot, bad, xccs);
    }
    return 0;
}

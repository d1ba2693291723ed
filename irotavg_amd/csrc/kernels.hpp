// kernels.hpp -- gfx950 device code of the rotation-averaging core. Wave = 64 lanes; all row
// kernels run 1024-thread workgroups (16 waves) on a grid of <= kMaxParts workgroups, each
// looping over a CONTIGUOUS chunk of row tiles. Workgroups that share blockIdx % 8 sit on one
// XCD (observed dispatch order), so chunks are assigned such that one XCD's workgroups cover
// adjacent row ranges: neighbour gathers of the band-dominated view-graph then hit that XCD's
// own L2. Placement only affects speed, never results.
//
// Every reduction is evaluated in a fixed order (lane -> wave -> workgroup -> partial array ->
// fixed-order re-reduction in the consumer), so results are bitwise reproducible run to run.
#pragma once
#include "common.hpp"

namespace irh {

struct LevelView {
    int n, nnz, agg;
    const int *rowptr, *col;
    const double *val, *diag, *idg;
};

// ---------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;  // lane 0
}

template <int G>
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;  // all lanes of the G-lane group
}

// workgroup sum of three accumulators -> part[0..2] (written by thread 0)
__device__ __forceinline__ void block_sum3_store(double a0, double a1, double a2, double *part) {
    __shared__ double sm[3][kBlock / 64];
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
// (nparts <= kMaxParts rows of 4 doubles)
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

// contiguous tile chunk of this workgroup, XCD-aware (see file header)
__device__ __forceinline__ void tile_range(int ntiles, int &t0, int &t1) {
    const int nb = gridDim.x, b = blockIdx.x;
    int lb = b;
    if ((nb & 7) == 0) lb = (b & 7) * (nb >> 3) + (b >> 3);
    t0 = (int)(((long long)ntiles * lb) / nb);
    t1 = (int)(((long long)ntiles * (lb + 1)) / nb);
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

}  // namespace irh

// common.hpp -- shared declarations of libirotavg_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/irotavg_hip.h"

namespace irh {

struct HipError {
    hipError_t e;
};

#define IRH_CHECK(expr)                                                                        \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            std::fprintf(stderr, "[irotavg_hip] %s failed: %s (%s:%d)\n", #expr,               \
                         hipGetErrorString(_e), __FILE__, __LINE__);                           \
            throw ::irh::HipError{_e};                                                         \
        }                                                                                      \
    } while (0)

// Device buffer (plain hipMalloc; sized once per graph, HBM-resident for the handle's life).
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    bool own = true;  // false: a non-owning alias of another handle's buffer (read-only data)
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n), own(o.own) {
        o.p = nullptr;
        o.n = 0;
    }
    DevBuf &operator=(DevBuf &&o) noexcept {
        if (this != &o) {
            release();
            p = o.p;
            n = o.n;
            own = o.own;
            o.p = nullptr;
            o.n = 0;
        }
        return *this;
    }
    ~DevBuf() { release(); }
    void release() {
        if (p && own) (void)hipFree(p);
        p = nullptr;
        n = 0;
        own = true;
    }
    void alias(const DevBuf &o) {
        release();
        p = o.p;
        n = o.n;
        own = false;
    }
    void alloc_like(const DevBuf &o, hipStream_t s) {
        alloc(o.n);
        if (o.n) IRH_CHECK(hipMemsetAsync(p, 0, o.n * sizeof(T), s));
    }
    void alloc(size_t count) {
        release();
        n = count;
        if (count == 0) count = 1;
        IRH_CHECK(hipMalloc((void **)&p, count * sizeof(T)));
    }
    void upload(const T *h, size_t count, hipStream_t s) {
        if (count > n) alloc(count);
        if (count) IRH_CHECK(hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, s));
    }
    void upload(const std::vector<T> &h, hipStream_t s) { upload(h.data(), h.size(), s); }
    void zero(hipStream_t s) {
        if (n) IRH_CHECK(hipMemsetAsync(p, 0, n * sizeof(T), s));
    }
};

constexpr int kBlock = 1024;   // threads of the single-workgroup coarse-cycle kernel
constexpr int kRowBlock = 256;  // threads per workgroup of the row / edge / vector kernels (4 waves)
constexpr int kMaxParts = 512;  // dot-product partials (= max grid of a reducing kernel)
constexpr int kMaxLevels = 16;
constexpr int kSellUnroll = 8;  // slice widths are multiples of this (batch size of the row loops)

// per-edge flag bits (host-built; see build.cpp)
enum : uint8_t {
    EF_CJ = 1,  // A(k, j-f) = +1 present   (ral/l1_irls.cpp:770-772)
    EF_CI = 2,  // A(k, i-f) = -1 present   (ral/l1_irls.cpp:774-776)
};
// boundary-slot flag bits
enum : uint8_t {
    BF_IRLS = 1,  // contributes to A'D^2A (the make_A matrix has a coefficient for this row)
    BF_L1H = 2,   // contributes to the make_AtA matrix (ral/l1_irls.cpp:825-835)
    BF_NEG = 4,   // self loop: make_AtA ends with -1 on the diagonal (see oracle lap_fill_AtA_times)
};

// scalar block shared by the PCG kernels (device memory, doubles)
enum ScalIdx : int {
    SC_RZ0 = 0,   // rz of parity 0 (3 values, padded to 4)
    SC_RZ1 = 4,   // rz of parity 1
    SC_BB = 8,    // ||b||^2 per column
    SC_RELRES = 12,  // ||r||/||b|| per column at the last check
    SC_COUNT = 16
};
// int flags block
enum FlagIdx : int {
    FL_DONE = 0,   // 0 running, 1 converged, 2 breakdown (non-finite scalar)
    FL_ITERS = 1,  // PCG iterations performed
    FL_COUNT = 4
};

// One multigrid level. The off-diagonal part of the weighted Laplacian is held in SELL-64 with
// 2-entry interleave: rows are cut into slices of 64 consecutive rows (one wavefront), slice s is
// `sl_off[s+1] - sl_off[s]` entry-columns wide (a multiple of 8), and entries 2q, 2q+1 of the row
// in lane l sit next to each other at (sl_off[s] + 2q) * 64 + 2l (+1): one lane owns one row and
// reads its entries two at a time, so a wave-level load of `val` is one contiguous 1 KiB run of
// 16 B per lane (`col`: 512 B of 8 B per lane), the inner loop has a wave-uniform trip count and
// no cross-lane reduction is needed. Within a row the entries whose column lies in the window of
// the row's 256-row workgroup tile ([tile - 64, tile + 320)) come first -- the first `sl_near[s]`
// entry-columns of the slice hold only such entries -- so the SpMV can serve them from an LDS copy
// of that window instead of gathering from global memory. Padding: val = 0, col = a row of the slice.
__host__ __device__ inline size_t sell_pos(int o0, int k, int lane) {
    return (size_t)(o0 + (k & ~1)) * 64 + (size_t)lane * 2 + (size_t)(k & 1);
}
constexpr int kWinHalo = 64;   // rows on either side of a 256-row tile held in the LDS window
constexpr int kWinLen = 256 + 2 * kWinHalo;
struct Level {
    int n = 0;        // rows
    int nnz = 0;      // real off-diagonal entries
    int agg = 0;      // rows per aggregate towards the next (coarser) level; 0 on the coarsest
    int nsl = 0;      // slices
    long long sell_len = 0;  // 64 * total entry-columns
    DevBuf<int> sl_off, sl_near, col;
    DevBuf<double> val;     // off-diagonal values (<= 0 for a Laplacian)
    DevBuf<double> excess;  // diag - sum|offdiag| : Dirichlet mass from fixed neighbours
    DevBuf<double> diag, idg;  // diagonal and its inverse (0 where the diagonal is 0)
    // value refresh from the finer level: coarse entry c (CSR order) sums the finer SELL
    // positions cidx[cptr[c]..cptr[c+1]) and is stored at SELL position cpos[c]
    DevBuf<int> cptr, cidx, cpos;
    // multigrid work vectors (double4 with 3 active components)
    DevBuf<double4> b, x, y, e;
};

}  // namespace irh

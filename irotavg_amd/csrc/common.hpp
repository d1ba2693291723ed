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
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n) {
        o.p = nullptr;
        o.n = 0;
    }
    DevBuf &operator=(DevBuf &&o) noexcept {
        if (this != &o) {
            release();
            p = o.p;
            n = o.n;
            o.p = nullptr;
            o.n = 0;
        }
        return *this;
    }
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
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

constexpr int kBlock = 1024;  // threads per workgroup of the row/edge kernels (16 waves)
constexpr int kMaxParts = 256;  // dot-product partials (= max grid of a reducing kernel)
constexpr int kMaxLevels = 16;

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

struct Level {
    int n = 0;        // rows
    int nnz = 0;      // off-diagonal entries
    int agg = 0;      // rows per aggregate towards the next (coarser) level; 0 on the coarsest
    int lanes = 16;   // lanes cooperating on one row in the row kernels
    DevBuf<int> rowptr, col;
    DevBuf<double> val;     // off-diagonal values (<= 0 for a Laplacian)
    DevBuf<double> excess;  // diag - sum|offdiag| : Dirichlet mass from fixed neighbours
    DevBuf<double> diag, idg;  // diagonal and its inverse (0 where the diagonal is 0)
    // value refresh from the finer level: coarse slot c sums finer slots cidx[cptr[c]..cptr[c+1])
    DevBuf<int> cptr, cidx;
    // multigrid work vectors (double4 with 3 active components)
    DevBuf<double4> b, x, y;
};

}  // namespace irh

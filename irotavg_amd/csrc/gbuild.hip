// gbuild.hip -- construction of the device-resident graph ON THE DEVICE: the edge streams, the vertex
// adjacency of the weighted Laplacian (SELL-64), the aggregation hierarchy with its value-refresh maps.
//
// build.cpp does the same on the host (up to 16 threads): ~43 ms at 100k views / 2M edges, seven times the
// solve it prepares, which made the one-shot drop-ins irotavg_irls / irotavg_l1ra (the signature the
// reference's callers use, ral/l1_irls.hpp:100-107, with make_A's work -- ral/l1_irls.cpp:755-780 -- inside)
// and every global re-solve of rot_avg host-bound. Here the caller's arrays are uploaded once and every
// pattern is derived by kernels: stable radix sorts (rocPRIM) where the host sorted, scans where it summed
// counters, one lane per row / one wave per slice where it looped. The result is the SAME structure, array for
// array (tests/test_gpu_build.py compares them bit for bit and the solves that run on them), so everything
// build.cpp documents about the layout holds here:
//  * level-0 entries of a row ordered by (column, edge id); boundary slots of a row in edge order;
//  * coarse entry (I, J) sums the finer entries in ascending finer-slot order (cptr / cidx);
//  * SELL-64 with 2-entry interleave, near entries (|column - row| <= 64) first in every row.
// Orders that the host obtained from serial loops come from STABLE sorts of elements generated in edge
// (or slot) order; counters are integer atomics (order-independent): the build is deterministic.
//
// The host keeps what is arithmetic on a handful of numbers: the shape of the hierarchy (plan_hierarchy,
// build.cpp) from n, nnz and the `far` flag, and the tail of the build (finish_build). Seven small
// read-backs (sizes needed to allocate the next arrays) synchronise the stream; ~2.5 ms of kernels at 100k / 2M.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>

#include "graph.hpp"
#include "kernels.hpp"

namespace irh {

namespace {

constexpr int kT = 256;
inline unsigned grid_of(long long n, int per = kT) { return (unsigned)std::max<long long>(1, (n + per - 1) / per); }
inline int bits_for(unsigned long long v) {  // bits needed to represent values 0..v
    int b = 1;
    while (b < 64 && (v >> b) != 0) b++;
    return b;
}

// ---- edges ---------------------------------------------------------------------------------------
// I (m pairs) -> ei, ej, eflag over [0, mpad); range check -> info[0]
// relabel (a resident view-graph's edge list, resident.hip): the pairs hold the caller's view ids, the rows of this
// problem are relabel[id]
__global__ __launch_bounds__(kT) void k_gb_edges(long long m, long long mpad, int n_total, int f,
                                                 const int2 *__restrict__ I, const int *__restrict__ relabel,
                                                 int *__restrict__ ei, int *__restrict__ ej,
                                                 uint8_t *__restrict__ eflag, int *__restrict__ info) {
    const long long k = (long long)blockIdx.x * kT + threadIdx.x;
    if (k >= mpad) return;
    int i = 0, j = 0;
    uint8_t fl = 0;
    if (k < m) {
        const int2 e = I[k];
        i = e.x;
        j = e.y;
        if (relabel && i >= 0 && j >= 0 && i < n_total && j < n_total) {
            i = relabel[i];
            j = relabel[j];
        }
        if (i < 0 || j < 0 || i >= n_total || j >= n_total) {
            atomicOr(info, 1);
            i = j = 0;
        } else if (j >= f) {
            if (i >= f && i == j) {
                fl = EF_CI;  // self loop: the -1 overwrites the +1
            } else {
                fl |= EF_CJ;
                if (i >= f) fl |= EF_CI;
            }
        }
    }
    ei[k] = i;
    ej[k] = j;
    eflag[k] = fl;
}

// relative rotations held on the device as one double4 per edge -> the four planes of the edge kernels
__global__ __launch_bounds__(kT) void k_gb_qq_planes(long long m, long long mpad, const double4 *__restrict__ q,
                                                     double *__restrict__ planes) {
    const long long k = (long long)blockIdx.x * kT + threadIdx.x;
    if (k >= mpad) return;
    const double4 v = k < m ? q[k] : make_double4(0, 0, 0, 0);
    planes[k] = v.x;
    planes[mpad + k] = v.y;
    planes[2 * mpad + k] = v.z;
    planes[3 * mpad + k] = v.w;
}

// level-0 matrix entries and boundary slots, generated in EDGE order: edge k yields the entries 2k (row j:
// the +1 coefficient) and 2k + 1 (row i) when both endpoints are free and differ, otherwise at most one
// boundary slot. Keys of what an edge does not yield are `inv` (sorted to the end).
__global__ __launch_bounds__(kT) void k_gb_gen0(long long m, int f, int nu, const int *__restrict__ ei,
                                                const int *__restrict__ ej, unsigned long long *__restrict__ key,
                                                unsigned *__restrict__ val, unsigned *__restrict__ bkey,
                                                unsigned *__restrict__ bval, int *__restrict__ rowcnt,
                                                int *__restrict__ bcnt, int *__restrict__ info) {
    const long long k = (long long)blockIdx.x * kT + threadIdx.x;
    if (k >= m) return;
    const int i = ei[k] - f, j = ej[k] - f;  // single GPU: no ghosts, free = index >= f
    const unsigned long long inv = (unsigned long long)nu * (unsigned long long)nu;
    unsigned long long k0 = inv, k1 = inv;
    unsigned bk = (unsigned)nu;
    if (i >= 0 && j >= 0 && i != j) {
        k0 = (unsigned long long)j * nu + (unsigned long long)i;
        k1 = (unsigned long long)i * nu + (unsigned long long)j;
        atomicAdd(rowcnt + j, 1);
        atomicAdd(rowcnt + i, 1);
        if (i < j - kWinHalo || i > j + kWinHalo) atomicOr(info + 1, 1);  // a far entry
    } else if (i >= 0 && j >= 0) {
        bk = (unsigned)i;  // self loop
    } else if (j >= 0) {
        bk = (unsigned)j;  // i fixed
    } else if (i >= 0) {
        bk = (unsigned)i;  // j fixed (dropped by make_A, kept by make_AtA)
    }
    key[2 * k] = k0;
    key[2 * k + 1] = k1;
    val[2 * k] = ((unsigned)k << 1) | 1u;
    val[2 * k + 1] = (unsigned)k << 1;
    bkey[k] = bk;
    bval[k] = (unsigned)k;
    if (bk != (unsigned)nu) atomicAdd(bcnt + bk, 1);
}

// sorted keys -> CSR columns (key = row * ncols + col)
__global__ __launch_bounds__(kT) void k_gb_cols(long long nnz, int ncols, const unsigned long long *__restrict__ key,
                                                int *__restrict__ col) {
    const long long t = (long long)blockIdx.x * kT + threadIdx.x;
    if (t < nnz) col[t] = (int)(key[t] % (unsigned long long)ncols);
}

// sorted boundary slots (by row, edge order inside a row) -> beid / bflag / bghost
__global__ __launch_bounds__(kT) void k_gb_boundary(long long nb, int f, const unsigned *__restrict__ bval,
                                                    const int *__restrict__ ei, const int *__restrict__ ej,
                                                    uint32_t *__restrict__ beid, uint8_t *__restrict__ bflag,
                                                    int *__restrict__ bghost) {
    const long long t = (long long)blockIdx.x * kT + threadIdx.x;
    if (t >= nb) return;
    const unsigned k = bval[t];
    const int i = ei[k] - f, j = ej[k] - f;
    uint32_t e;
    uint8_t fl;
    if (i >= 0 && j >= 0) {  // self loop
        e = k << 1;
        fl = (uint8_t)(BF_IRLS | BF_L1H | BF_NEG);
    } else if (j >= 0) {  // row j, other endpoint fixed
        e = (k << 1) | 1u;
        fl = (uint8_t)(BF_IRLS | BF_L1H);
    } else {  // row i, j fixed: make_A drops the row, make_AtA keeps the diagonal term
        e = k << 1;
        fl = (uint8_t)BF_L1H;
    }
    beid[t] = e;
    bflag[t] = fl;
    bghost[t] = -1;
}

// ---- SELL map of a CSR pattern ---------------------------------------------------------------------
// one wave per slice: widths of the near / far parts (rounded up to kSellUnroll)
__global__ __launch_bounds__(kT) void k_gb_slice_width(int n, int nsl, const int *__restrict__ rowptr,
                                                       const int *__restrict__ col, int *__restrict__ sl_near,
                                                       int *__restrict__ width) {
    const int sl = blockIdx.x * (kT / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (sl >= nsl) return;
    const int r = sl * 64 + lane;
    int nn = 0, nf = 0;
    if (r < n) {
        for (int t = rowptr[r]; t < rowptr[r + 1]; t++) {
            const int c = col[t];
            if (c >= r - kWinHalo && c <= r + kWinHalo)
                nn++;
            else
                nf++;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        nn = max(nn, __shfl_xor(nn, o, 64));
        nf = max(nf, __shfl_xor(nf, o, 64));
    }
    if (lane == 0) {
        const int wn = max((nn + kSellUnroll - 1) / kSellUnroll * kSellUnroll, kSellUnroll);  // see build.cpp
        const int wf = (nf + kSellUnroll - 1) / kSellUnroll * kSellUnroll;
        sl_near[sl] = wn;
        width[sl] = wn + wf;
    }
    if (sl == nsl - 1 && lane == 0) width[nsl] = 0;  // the scan runs over nsl + 1 elements
}

// summary of a SELL map (one workgroup): out[0] = widest near part, out[1] = uniform width (0: none),
// out[2..3] = far entry-columns (64-bit), out[4] = longest row
__global__ __launch_bounds__(1024) void k_gb_sell_summary(int n, int nsl, const int *__restrict__ sl_off,
                                                          const int *__restrict__ sl_near,
                                                          const int *__restrict__ rowptr, int *__restrict__ out) {
    __shared__ int smax[16], sbad[16], srow[16];
    __shared__ long long sfar[16];
    int mx = 0, bad = 0, mr = 0;
    long long far = 0;
    const int w0 = nsl > 0 ? sl_off[1] - sl_off[0] : 0;
    for (int sl = threadIdx.x; sl < nsl; sl += 1024) {
        const int w = sl_off[sl + 1] - sl_off[sl], nr = sl_near[sl];
        mx = max(mx, nr);
        far += w - nr;
        if (w != w0 || nr != w0) bad = 1;
    }
    for (int r = threadIdx.x; r < n; r += 1024) mr = max(mr, rowptr[r + 1] - rowptr[r]);
    for (int o = 32; o > 0; o >>= 1) {
        mx = max(mx, __shfl_xor(mx, o, 64));
        mr = max(mr, __shfl_xor(mr, o, 64));
        bad |= __shfl_xor(bad, o, 64);
        far += __shfl_xor(far, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        smax[threadIdx.x >> 6] = mx;
        sbad[threadIdx.x >> 6] = bad;
        srow[threadIdx.x >> 6] = mr;
        sfar[threadIdx.x >> 6] = far;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; w++) {
            mx = max(mx, smax[w]);
            mr = max(mr, srow[w]);
            bad |= sbad[w];
            far += sfar[w];
        }
        out[0] = mx;
        out[1] = bad ? 0 : w0;
        out[2] = (int)(far & 0xffffffffll);
        out[3] = (int)(far >> 32);
        out[4] = mr;
    }
}

// one lane per row: SELL position and in-row index of every CSR slot (near entries first)
__global__ __launch_bounds__(kT) void k_gb_positions(int n, const int *__restrict__ rowptr,
                                                     const int *__restrict__ col, const int *__restrict__ sl_off,
                                                     const int *__restrict__ sl_near, int *__restrict__ pos,
                                                     int *__restrict__ kidx) {
    const int r = blockIdx.x * kT + threadIdx.x;
    if (r >= n) return;
    const int sl = r >> 6, lane = r & 63;
    const int o0 = sl_off[sl];
    int kn = 0, kf = sl_near[sl];
    for (int t = rowptr[r]; t < rowptr[r + 1]; t++) {
        const int c = col[t];
        const int k = (c >= r - kWinHalo && c <= r + kWinHalo) ? kn++ : kf++;
        pos[t] = (int)sell_pos(o0, k, lane);
        kidx[t] = k;
    }
}

// padding of the SELL column array: a valid near column (row 0 of the slice); one workgroup per slice
__global__ __launch_bounds__(kT) void k_gb_fill_col(int nsl, const int *__restrict__ sl_off, int *__restrict__ scol) {
    const int sl = blockIdx.x;
    if (sl >= nsl) return;
    const long long a = 64ll * sl_off[sl], b = 64ll * sl_off[sl + 1];
    for (long long p = a + threadIdx.x; p < b; p += kT) scol[p] = sl * 64;
}

__global__ __launch_bounds__(kT) void k_gb_scatter(long long nnz, const int *__restrict__ pos,
                                                   const int *__restrict__ col, int *__restrict__ scol,
                                                   const unsigned *__restrict__ eid, uint32_t *__restrict__ seid) {
    const long long t = (long long)blockIdx.x * kT + threadIdx.x;
    if (t >= nnz) return;
    const int p = pos[t];
    scol[p] = col[t];
    if (seid != nullptr) seid[p] = eid[t];
}

// per level-0 slice: lowest edge id among its near entries (first edge of the run k_assemble0w stages)
__global__ __launch_bounds__(kT) void k_gb_tile_e0(int n, int nsl, const int *__restrict__ rowptr,
                                                   const int *__restrict__ col, const unsigned *__restrict__ eid,
                                                   int *__restrict__ te0) {
    const int sl = blockIdx.x * (kT / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (sl >= nsl) return;
    const int r = sl * 64 + lane;
    unsigned lo = 0xffffffffu;
    if (r < n)
        for (int t = rowptr[r]; t < rowptr[r + 1]; t++) {
            const int c = col[t];
            if (c >= r - kWinHalo && c <= r + kWinHalo) lo = min(lo, eid[t] >> 1);
        }
    for (int o = 32; o > 0; o >>= 1) lo = min(lo, (unsigned)__shfl_xor((int)lo, o, 64));
    if (lane == 0) te0[sl] = lo == 0xffffffffu ? 0 : (int)lo;
}

// ---- coarse level from a finer CSR pattern -----------------------------------------------------------
// key of every finer slot t (row v, column c): (v / agg) * Cn + c / agg, or `inv` inside an aggregate
__global__ __launch_bounds__(kT) void k_gb_coarse_keys(int n, int agg, int Cn, const int *__restrict__ rowptr,
                                                       const int *__restrict__ col,
                                                       unsigned long long *__restrict__ key,
                                                       unsigned *__restrict__ val, int *__restrict__ info) {
    const int v = blockIdx.x * kT + threadIdx.x;
    if (v >= n) return;
    const unsigned long long inv = (unsigned long long)Cn * (unsigned long long)Cn;
    const int Ic = v / agg;
    int used = 0;
    for (int t = rowptr[v]; t < rowptr[v + 1]; t++) {
        const int Jc = col[t] / agg;
        key[t] = Jc != Ic ? (unsigned long long)Ic * Cn + (unsigned long long)Jc : inv;
        val[t] = (unsigned)t;
        used += Jc != Ic;
    }
    for (int o = 32; o > 0; o >>= 1) used += __shfl_xor(used, o, 64);
    if ((threadIdx.x & 63) == 0 && used) atomicAdd(info, used);
}

// head flag of every sorted slot q < nv (first slot of a coarse entry)
__global__ __launch_bounds__(kT) void k_gb_heads(long long nv, const unsigned long long *__restrict__ key,
                                                 int *__restrict__ head) {
    const long long q = (long long)blockIdx.x * kT + threadIdx.x;
    if (q > nv) return;
    head[q] = (q < nv && (q == 0 || key[q] != key[q - 1])) ? 1 : 0;  // head[nv] = 0: the scan's total lands there
}

// entry index of every slot (exclusive scan of the head flags) -> coarse columns, cptr, per-row counts, cidx
__global__ __launch_bounds__(kT) void k_gb_coarse_entries(long long nv, int Cn, const unsigned long long *__restrict__ key,
                                                          const unsigned *__restrict__ val,
                                                          const int *__restrict__ head, const int *__restrict__ eidx,
                                                          int *__restrict__ ccol, int *__restrict__ cptr,
                                                          int *__restrict__ crowcnt, int *__restrict__ cidx) {
    const long long q = (long long)blockIdx.x * kT + threadIdx.x;
    if (q >= nv) return;
    cidx[q] = (int)val[q];
    if (head[q]) {
        const int c = eidx[q];
        const unsigned long long k = key[q];
        ccol[c] = (int)(k % (unsigned long long)Cn);
        cptr[c] = (int)q;
        atomicAdd(crowcnt + (int)(k / (unsigned long long)Cn), 1);
    }
}

__global__ void k_gb_set(int *p, int v) { *p = v; }

// cidx: finer CSR slot -> finer SELL position
__global__ __launch_bounds__(kT) void k_gb_remap(long long nv, const int *__restrict__ prevpos, int *__restrict__ cidx) {
    const long long q = (long long)blockIdx.x * kT + threadIdx.x;
    if (q < nv) cidx[q] = prevpos[cidx[q]];
}

// level 1 only: level-1 entry index (within its SELL row) of every level-0 SELL position; info: any index >= 8
__global__ __launch_bounds__(kT) void k_gb_slot_cs(int nent, const int *__restrict__ cptr, const int *__restrict__ cidx_sell,
                                                   const int *__restrict__ kidx, uint8_t *__restrict__ cs,
                                                   int *__restrict__ info) {
    const int c = blockIdx.x * kT + threadIdx.x;
    if (c >= nent) return;
    const int k = kidx[c];
    if (k >= 8) atomicOr(info, 1);
    for (int q = cptr[c]; q < cptr[c + 1]; q++) cs[cidx_sell[q]] = (uint8_t)min(k, 255);
}

// pattern statistics of a level: info[0] = max |col - row|, info[1] |= a column outside the 48-row window of
// the row's 32-row block
__global__ __launch_bounds__(kT) void k_gb_band_stats(int n, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                      int *__restrict__ info) {
    const int r = blockIdx.x * kT + threadIdx.x;
    int bw = 0, out = 0;
    if (r < n) {
        const int lo = (r / 32) * 32 - 8, hi = (r / 32) * 32 + 40;
        for (int t = rowptr[r]; t < rowptr[r + 1]; t++) {
            const int c = col[t];
            bw = max(bw, abs(c - r));
            if (c < lo || c >= hi) out = 1;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        bw = max(bw, __shfl_xor(bw, o, 64));
        out |= __shfl_xor(out, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (bw) atomicMax(info, bw);
        if (out) atomicOr(info + 1, 1);
    }
}

// ---- host helpers --------------------------------------------------------------------------------------
struct Scratch {  // temporary storage of the rocPRIM calls, grown on demand
    DevBuf<char> buf;
    void *need(size_t bytes) {
        if (buf.n < bytes) buf.alloc(bytes + bytes / 4 + 4096);
        return buf.p;
    }
};

void sort_pairs64(Scratch &S, unsigned long long *kin, unsigned long long *kout, unsigned *vin, unsigned *vout,
                  size_t n, int bits, hipStream_t s) {
    size_t bytes = 0;
    IRH_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, 0u, (unsigned)bits, s));
    IRH_CHECK(rocprim::radix_sort_pairs(S.need(bytes), bytes, kin, kout, vin, vout, n, 0u, (unsigned)bits, s));
}
void sort_pairs32(Scratch &S, unsigned *kin, unsigned *kout, unsigned *vin, unsigned *vout, size_t n, int bits,
                  hipStream_t s) {
    size_t bytes = 0;
    IRH_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, 0u, (unsigned)bits, s));
    IRH_CHECK(rocprim::radix_sort_pairs(S.need(bytes), bytes, kin, kout, vin, vout, n, 0u, (unsigned)bits, s));
}
// out[0..n] = exclusive prefix sums of in[0..n-1] followed by the total (in must hold n + 1 elements, in[n] = 0)
void scan_with_total(Scratch &S, int *in, int *out, size_t n, hipStream_t s) {
    size_t bytes = 0;
    IRH_CHECK(rocprim::exclusive_scan(nullptr, bytes, in, out, 0, n + 1, rocprim::plus<int>(), s));
    IRH_CHECK(rocprim::exclusive_scan(S.need(bytes), bytes, in, out, 0, n + 1, rocprim::plus<int>(), s));
}

struct DevCsr {  // a level's CSR pattern on the device (temporary: the solve uses the SELL form)
    int n = 0;
    long long nnz = 0;
    DevBuf<int> rowptr, col;
};
struct DevSell {
    int nsl = 0;
    long long len = 0;
    int max_near = 0, uni_w = 0, max_row = 0;
    long long far_cols = 0;
    DevBuf<int> sl_off, sl_near, pos, kidx;
};

// SELL map of a CSR pattern; one read-back (length and summary)
void sell_map_device(Graph &g, Scratch &S, const DevCsr &A, DevSell &M, int *h_info) {
    hipStream_t s = g.stream;
    M.nsl = (A.n + 63) / 64;
    DevBuf<int> width, summary;
    width.alloc((size_t)M.nsl + 1);
    M.sl_near.alloc((size_t)std::max(M.nsl, 1));
    M.sl_off.alloc((size_t)M.nsl + 1);
    summary.alloc(8);
    if (M.nsl > 0)
        hipLaunchKernelGGL(k_gb_slice_width, dim3(grid_of(M.nsl, kT / 64)), dim3(kT), 0, s, A.n, M.nsl, A.rowptr.p,
                           A.col.p, M.sl_near.p, width.p);
    else
        hipLaunchKernelGGL(k_gb_set, dim3(1), dim3(1), 0, s, width.p, 0);
    scan_with_total(S, width.p, M.sl_off.p, (size_t)M.nsl, s);
    hipLaunchKernelGGL(k_gb_sell_summary, dim3(1), dim3(1024), 0, s, A.n, M.nsl, M.sl_off.p, M.sl_near.p, A.rowptr.p,
                       summary.p);
    IRH_CHECK(hipMemcpyAsync(h_info, summary.p, sizeof(int) * 5, hipMemcpyDeviceToHost, s));
    IRH_CHECK(hipMemcpyAsync(h_info + 5, M.sl_off.p + M.nsl, sizeof(int), hipMemcpyDeviceToHost, s));
    IRH_CHECK(hipStreamSynchronize(s));
    M.max_near = h_info[0];
    M.uni_w = h_info[1];
    M.far_cols = ((long long)(unsigned)h_info[2]) | ((long long)h_info[3] << 32);
    M.max_row = h_info[4];
    M.len = 64ll * h_info[5];
    M.pos.alloc((size_t)std::max<long long>(A.nnz, 1));
    M.kidx.alloc((size_t)std::max<long long>(A.nnz, 1));
    if (A.n > 0)
        hipLaunchKernelGGL(k_gb_positions, dim3(grid_of(A.n)), dim3(kT), 0, s, A.n, A.rowptr.p, A.col.p, M.sl_off.p,
                           M.sl_near.p, M.pos.p, M.kidx.p);
}

}  // namespace

int build_graph_device(Graph &g, const int32_t *I, const double *QQ, int64_t ldqq, const DevEdgeSrc *src) {
    const bool timing = getenv("IROTAVG_BUILD_TIMING") != nullptr;
    double tlast = now_seconds();
    auto lap = [&](const char *what) {
        if (!timing) return;
        (void)hipStreamSynchronize(g.stream);
        const double t = now_seconds();
        std::fprintf(stderr, "[irotavg_hip device build] %-26s %8.2f ms\n", what, 1e3 * (t - tlast));
        tlast = t;
    };
    const int64_t m = g.m;
    const int f = g.f;
    const int nu = g.no;
    hipStream_t s = g.stream;
    if (g.ng != 0 || m > 0x3fffffffLL) return IROTAVG_ERR_BAD_ARG;
    Scratch S;
    int *h_info = reinterpret_cast<int *>(PinPool::get().take());  // pinned read-back block (16 ints used)
    struct PinGuard {
        int *p;
        ~PinGuard() { PinPool::get().give(p); }
    } pin_guard{h_info};

    // ---- edge streams ----------------------------------------------------------------------------------
    g.mpad = (m + 63) / 64 * 64;
    DevBuf<int> dI;
    if (!src) {
        dI.alloc((size_t)2 * m);
        IRH_CHECK(hipMemcpyAsync(dI.p, I, sizeof(int) * 2 * (size_t)m, hipMemcpyHostToDevice, s));
    }
    g.qq.alloc((size_t)4 * g.mpad);
    if (src)  // the edge list is resident on the device already (resident.hip): planes from its records, no upload
        hipLaunchKernelGGL(k_gb_qq_planes, dim3(grid_of(g.mpad)), dim3(kT), 0, s, (long long)m, (long long)g.mpad, src->QQ,
                           g.qq.p);
    else if (g.mpad > m)
        for (int c = 0; c < 4; c++)
            IRH_CHECK(hipMemsetAsync(g.qq.p + (size_t)c * g.mpad + m, 0, sizeof(double) * (size_t)(g.mpad - m), s));
    // The relative rotations (4 m doubles: 64 MB at 2M edges, the bulk of the upload) are needed by the first solve
    // only: helper threads copy them (a plane each, a stream each) while this thread builds the patterns from I (a copy
    // from pageable memory keeps its caller busy, and one caller reaches ~25 GB/s only).
    // Joined before the build returns, on every path.
    struct QQUpload {
        std::thread th[4];
        hipError_t err[4] = {hipSuccess, hipSuccess, hipSuccess, hipSuccess};
        void join() {
            for (auto &t : th)
                if (t.joinable()) t.join();
        }
        ~QQUpload() { join(); }
    } qq_up;
    const char *upenv = getenv("IROTAVG_UPLOAD_THREADS");
    const int n_up = (m >= 100000 && !src) ? (upenv ? std::min(4, std::max(0, atoi(upenv))) : 4) : 0;
    if (src) {
    } else if (n_up > 0) {
        double *dst = g.qq.p;
        const size_t mp = (size_t)g.mpad;
        const int dev = g.device;
        for (int t = 0; t < n_up; t++)
            qq_up.th[t] = std::thread([&qq_up, dst, mp, QQ, ldqq, m, dev, t, n_up]() {
                hipError_t e = hipSetDevice(dev);
                hipStream_t s2 = nullptr;
                try {
                    if (e == hipSuccess) s2 = StreamPool::get().take();  // (creating a stream costs milliseconds)
                } catch (...) {
                    e = hipErrorUnknown;
                }
                for (int c = t; c < 4 && e == hipSuccess; c += n_up)
                    e = hipMemcpyAsync(dst + (size_t)c * mp, QQ + (size_t)c * ldqq, sizeof(double) * (size_t)m,
                                       hipMemcpyHostToDevice, s2);
                if (e == hipSuccess) e = hipStreamSynchronize(s2);
                if (s2 && e == hipSuccess)
                    StreamPool::get().give(s2, dev);
                else if (s2)
                    (void)hipStreamDestroy(s2);  // (a stream that failed is not handed to the next build)
                qq_up.err[t] = e;
            });
    } else {
        for (int c = 0; c < 4; c++)
            IRH_CHECK(hipMemcpyAsync(g.qq.p + (size_t)c * g.mpad, QQ + (size_t)c * ldqq, sizeof(double) * (size_t)m,
                                     hipMemcpyHostToDevice, s));
    }
    g.ei.alloc((size_t)g.mpad);
    g.ej.alloc((size_t)g.mpad);
    g.eflag.alloc((size_t)g.mpad);
    DevBuf<int> info;  // [0] bad index, [1] far entry on level 0, [2..] scratch counters
    info.alloc(16);
    info.zero(s);
    hipLaunchKernelGGL(k_gb_edges, dim3(grid_of(g.mpad)), dim3(kT), 0, s, (long long)m, (long long)g.mpad,
                       (int)g.n_total, f, src ? src->I : reinterpret_cast<const int2 *>(dI.p), src ? src->relabel : nullptr,
                       g.ei.p, g.ej.p, g.eflag.p, info.p);
    g.er.alloc((size_t)3 * g.mpad);
    g.er.zero(s);
    g.dw.alloc((size_t)g.mpad);
    fill(g, g.dw.p, (long long)g.mpad, 1.0);
    g.Q.alloc((size_t)g.n_total);
    g.Q.zero(s);
    lap("uploads + edge streams");

    // ---- level-0 adjacency --------------------------------------------------------------------------------
    DevCsr A0;
    A0.n = nu;
    DevBuf<unsigned> eid0;  // per level-0 CSR slot: (edge id << 1) | (row is the j endpoint)
    DevBuf<int> bptr_cnt;
    long long nb = 0;
    {
        DevBuf<unsigned long long> key, key2;
        DevBuf<unsigned> val, val2, bkey, bkey2, bval, bval2;
        DevBuf<int> rowcnt;
        key.alloc((size_t)2 * m);
        key2.alloc((size_t)2 * m);
        val.alloc((size_t)2 * m);
        val2.alloc((size_t)2 * m);
        bkey.alloc((size_t)m);
        bkey2.alloc((size_t)m);
        bval.alloc((size_t)m);
        bval2.alloc((size_t)m);
        rowcnt.alloc((size_t)nu + 1);
        bptr_cnt.alloc((size_t)nu + 1);
        rowcnt.zero(s);
        bptr_cnt.zero(s);
        hipLaunchKernelGGL(k_gb_gen0, dim3(grid_of(m)), dim3(kT), 0, s, (long long)m, f, nu, g.ei.p, g.ej.p, key.p,
                           val.p, bkey.p, bval.p, rowcnt.p, bptr_cnt.p, info.p);
        A0.rowptr.alloc((size_t)nu + 1);
        g.bptr.alloc((size_t)nu + 1);
        scan_with_total(S, rowcnt.p, A0.rowptr.p, (size_t)nu, s);
        scan_with_total(S, bptr_cnt.p, g.bptr.p, (size_t)nu, s);
        IRH_CHECK(hipMemcpyAsync(h_info, info.p, sizeof(int) * 2, hipMemcpyDeviceToHost, s));
        IRH_CHECK(hipMemcpyAsync(h_info + 2, A0.rowptr.p + nu, sizeof(int), hipMemcpyDeviceToHost, s));
        IRH_CHECK(hipMemcpyAsync(h_info + 3, g.bptr.p + nu, sizeof(int), hipMemcpyDeviceToHost, s));
        // the sorts do not depend on the read-back: enqueue them first
        const int kbits = bits_for((unsigned long long)nu * (unsigned long long)nu);
        sort_pairs64(S, key.p, key2.p, val.p, val2.p, (size_t)2 * m, kbits, s);
        sort_pairs32(S, bkey.p, bkey2.p, bval.p, bval2.p, (size_t)m, bits_for((unsigned long long)nu), s);
        IRH_CHECK(hipStreamSynchronize(s));
        if (h_info[0]) return IROTAVG_ERR_BAD_ARG;
        A0.nnz = h_info[2];
        nb = h_info[3];
        if (A0.nnz < 0 || (long long)2 * m > 0x7fffffffLL) return IROTAVG_ERR_BAD_ARG;
        A0.col.alloc((size_t)std::max<long long>(A0.nnz, 1));
        eid0.alloc((size_t)std::max<long long>(A0.nnz, 1));
        if (A0.nnz > 0) {
            hipLaunchKernelGGL(k_gb_cols, dim3(grid_of(A0.nnz)), dim3(kT), 0, s, A0.nnz, nu, key2.p, A0.col.p);
            IRH_CHECK(hipMemcpyAsync(eid0.p, val2.p, sizeof(unsigned) * (size_t)A0.nnz, hipMemcpyDeviceToDevice, s));
        }
        g.beid.alloc((size_t)nb);
        g.bflag.alloc((size_t)nb);
        g.bghost.alloc((size_t)nb);
        if (nb > 0)
            hipLaunchKernelGGL(k_gb_boundary, dim3(grid_of(nb)), dim3(kT), 0, s, nb, f, bval2.p, g.ei.p, g.ej.p,
                               g.beid.p, g.bflag.p, g.bghost.p);
        g.bval.alloc((size_t)nb);
        g.bval.zero(s);
        g.PG.alloc((size_t)g.ng + 1);
        g.PG.zero(s);
        IRH_CHECK(hipStreamSynchronize(s));  // the temporaries of this block go back to the pool
    }
    const bool far0 = h_info[1] != 0;
    lap("level-0 adjacency");

    // ---- hierarchy ------------------------------------------------------------------------------------------
    const HierPlan plan = plan_hierarchy(g, nu, A0.nnz, far0);
    const size_t nlev = plan.n.size();
    g.levels.clear();
    g.levels.resize(nlev);
    g.stats.levels = (int)nlev;
    g.asm_windowed = getenv("IROTAVG_ASM_CLASSIC") ? 0 : 1;
    g.asm_l1_fused = 0;
    BuildTail T;
    T.nlev = (int)nlev;
    for (size_t l = 0; l < nlev && l < (size_t)kMaxLevels; l++) T.agg[l] = plan.agg[l];

    DevCsr cur = std::move(A0);
    DevSell prev;
    for (size_t lev = 0; lev < nlev; lev++) {
        Level &L = g.levels[lev];
        DevSell M;
        DevBuf<int> cptr, cidx;  // lev > 0: value-refresh map of this level (cidx in finer CSR slots at first)
        long long nv = 0;
        if (lev > 0) {
            // pattern of this level from the finer one
            const int agg = plan.agg[lev - 1], Cn = plan.n[lev];
            DevCsr C;
            C.n = Cn;
            DevBuf<unsigned long long> key, key2;
            DevBuf<unsigned> val, val2;
            DevBuf<int> head, eidx, crowcnt;
            const long long fn = cur.nnz;
            key.alloc((size_t)std::max<long long>(fn, 1));
            key2.alloc((size_t)std::max<long long>(fn, 1));
            val.alloc((size_t)std::max<long long>(fn, 1));
            val2.alloc((size_t)std::max<long long>(fn, 1));
            IRH_CHECK(hipMemsetAsync(info.p + 2, 0, sizeof(int), s));
            hipLaunchKernelGGL(k_gb_coarse_keys, dim3(grid_of(cur.n)), dim3(kT), 0, s, cur.n, agg, Cn, cur.rowptr.p,
                               cur.col.p, key.p, val.p, info.p + 2);
            IRH_CHECK(hipMemcpyAsync(h_info, info.p + 2, sizeof(int), hipMemcpyDeviceToHost, s));
            if (fn > 0)
                sort_pairs64(S, key.p, key2.p, val.p, val2.p, (size_t)fn,
                             bits_for((unsigned long long)Cn * (unsigned long long)Cn), s);
            IRH_CHECK(hipStreamSynchronize(s));
            nv = h_info[0];  // finer slots that land in a coarse off-diagonal entry
            head.alloc((size_t)nv + 1);
            eidx.alloc((size_t)nv + 1);
            crowcnt.alloc((size_t)Cn + 1);
            crowcnt.zero(s);
            hipLaunchKernelGGL(k_gb_heads, dim3(grid_of(nv + 1)), dim3(kT), 0, s, nv, key2.p, head.p);
            {
                size_t bytes = 0;
                IRH_CHECK(rocprim::exclusive_scan(nullptr, bytes, head.p, eidx.p, 0, (size_t)nv + 1, rocprim::plus<int>(), s));
                IRH_CHECK(rocprim::exclusive_scan(S.need(bytes), bytes, head.p, eidx.p, 0, (size_t)nv + 1,
                                                  rocprim::plus<int>(), s));
            }
            IRH_CHECK(hipMemcpyAsync(h_info, eidx.p + nv, sizeof(int), hipMemcpyDeviceToHost, s));
            IRH_CHECK(hipStreamSynchronize(s));
            C.nnz = h_info[0];  // coarse off-diagonal entries
            C.col.alloc((size_t)std::max<long long>(C.nnz, 1));
            C.rowptr.alloc((size_t)Cn + 1);
            cptr.alloc((size_t)C.nnz + 1);
            cidx.alloc((size_t)std::max<long long>(nv, 1));
            if (nv > 0)
                hipLaunchKernelGGL(k_gb_coarse_entries, dim3(grid_of(nv)), dim3(kT), 0, s, nv, Cn, key2.p, val2.p, head.p,
                                   eidx.p, C.col.p, cptr.p, crowcnt.p, cidx.p);
            hipLaunchKernelGGL(k_gb_set, dim3(1), dim3(1), 0, s, cptr.p + C.nnz, (int)nv);
            scan_with_total(S, crowcnt.p, C.rowptr.p, (size_t)Cn, s);
            IRH_CHECK(hipStreamSynchronize(s));  // temporaries of this block
            cur = std::move(C);
        }
        sell_map_device(g, S, cur, M, h_info);
        L.n = cur.n;
        L.nnz = (int)cur.nnz;
        L.agg = plan.agg[lev];
        L.nsl = M.nsl;
        L.sell_len = M.len;
        L.max_near = M.max_near;
        L.uni_w = M.uni_w;
        L.sl_off = std::move(M.sl_off);
        L.sl_near = std::move(M.sl_near);
        L.col.alloc((size_t)std::max<long long>(M.len, 1));
        if (M.nsl > 0) hipLaunchKernelGGL(k_gb_fill_col, dim3(M.nsl), dim3(kT), 0, s, M.nsl, L.sl_off.p, L.col.p);
        L.val.alloc((size_t)M.len);
        L.val.zero(s);
        if (lev == 0) {
            g.l0_far_entries = M.far_cols;
            g.kc_auto = g.opt.mg_kc <= 0;
            if (g.kc_auto) g.opt.mg_kc = g.l0_far_entries == 0 ? 2.0 : 1.6;
            g.slot_eid.alloc((size_t)std::max<long long>(M.len, 1));
            if (M.len > 0) IRH_CHECK(hipMemsetAsync(g.slot_eid.p, 0xff, sizeof(uint32_t) * (size_t)M.len, s));
            if (cur.nnz > 0)
                hipLaunchKernelGGL(k_gb_scatter, dim3(grid_of(cur.nnz)), dim3(kT), 0, s, cur.nnz, M.pos.p, cur.col.p,
                                   L.col.p, eid0.p, g.slot_eid.p);
            g.tile_e0.alloc((size_t)std::max(M.nsl, 1));
            if (g.asm_windowed && M.nsl > 0)
                hipLaunchKernelGGL(k_gb_tile_e0, dim3(grid_of(M.nsl, kT / 64)), dim3(kT), 0, s, cur.n, M.nsl,
                                   cur.rowptr.p, cur.col.p, eid0.p, g.tile_e0.p);
            else
                g.tile_e0.zero(s);
            if (nlev < 2) {  // no level 1: the assembly still reads a (dummy) level-1 index per pair
                g.slot_cs.alloc((size_t)std::max<long long>(M.len, 1));
                if (M.len > 0) IRH_CHECK(hipMemsetAsync(g.slot_cs.p, 0xff, (size_t)M.len, s));
            }
        } else {
            if (cur.nnz > 0)
                hipLaunchKernelGGL(k_gb_scatter, dim3(grid_of(cur.nnz)), dim3(kT), 0, s, cur.nnz, M.pos.p, cur.col.p,
                                   L.col.p, (const unsigned *)nullptr, (uint32_t *)nullptr);
            if (nv > 0) hipLaunchKernelGGL(k_gb_remap, dim3(grid_of(nv)), dim3(kT), 0, s, nv, prev.pos.p, cidx.p);
            L.max_row = M.max_row;
            if (lev == 1 && g.asm_windowed) {
                g.slot_cs.alloc((size_t)std::max<long long>(prev.len, 1));
                if (prev.len > 0) IRH_CHECK(hipMemsetAsync(g.slot_cs.p, 0xff, (size_t)prev.len, s));
                IRH_CHECK(hipMemsetAsync(info.p + 3, 0, sizeof(int), s));
                if (cur.nnz > 0)
                    hipLaunchKernelGGL(k_gb_slot_cs, dim3(grid_of(cur.nnz)), dim3(kT), 0, s, (int)cur.nnz, cptr.p, cidx.p,
                                       M.kidx.p, g.slot_cs.p, info.p + 3);
                IRH_CHECK(hipMemcpyAsync(h_info + 8, info.p + 3, sizeof(int), hipMemcpyDeviceToHost, s));
                IRH_CHECK(hipStreamSynchronize(s));
                const bool wide = h_info[8] != 0;
                g.asm_l1_fused = (g.asm_windowed && plan.agg[0] == 8 && !wide) ? 1 : 0;
            }
            L.crow.alloc((size_t)cur.n + 1);
            IRH_CHECK(hipMemcpyAsync(L.crow.p, cur.rowptr.p, sizeof(int) * ((size_t)cur.n + 1), hipMemcpyDeviceToDevice, s));
            L.cptr = std::move(cptr);
            L.cptr.n = (size_t)cur.nnz + 1;
            L.cidx = std::move(cidx);
            L.cidx.n = (size_t)nv;
            L.cpos.alloc((size_t)std::max<long long>(cur.nnz, 1));
            L.cpos.n = (size_t)cur.nnz;
            if (cur.nnz > 0)
                IRH_CHECK(hipMemcpyAsync(L.cpos.p, M.pos.p, sizeof(int) * (size_t)cur.nnz, hipMemcpyDeviceToDevice, s));
        }
        // pattern statistics the tail of the build asks for: level 1 (window tests), last level (bandwidth)
        if (lev == 1 || lev + 1 == nlev) {
            IRH_CHECK(hipMemsetAsync(info.p + 4, 0, 2 * sizeof(int), s));
            if (cur.n > 0)
                hipLaunchKernelGGL(k_gb_band_stats, dim3(grid_of(cur.n)), dim3(kT), 0, s, cur.n, cur.rowptr.p, cur.col.p,
                                   info.p + 4);
            IRH_CHECK(hipMemcpyAsync(h_info + 9, info.p + 4, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
            IRH_CHECK(hipStreamSynchronize(s));
            if (lev == 1) {
                T.l1_window_ok = h_info[10] == 0;
                T.l1_band8 = h_info[9] <= 8;
            }
            if (lev + 1 == nlev) T.dense_bw = h_info[9];
        }
        L.excess.alloc((size_t)M.nsl * 64);
        L.diag.alloc((size_t)M.nsl * 64);
        L.idg.alloc((size_t)M.nsl * 64);
        L.excess.zero(s);
        L.diag.zero(s);
        L.idg.zero(s);
        const size_t nvec = (size_t)M.nsl * 64 + 64;
        L.b.alloc(nvec);
        L.x.alloc(nvec);
        L.y.alloc(nvec);
        L.e.alloc(nvec);
        L.b.zero(s);
        L.x.zero(s);
        L.y.zero(s);
        L.e.zero(s);
        if (lev < (size_t)kMaxLevels) {
            g.stats.level_rows[lev] = L.n;
            g.stats.level_nnz[lev] = L.nnz;
        }
        IRH_CHECK(hipStreamSynchronize(s));  // temporaries of this level
        prev = std::move(M);
    }
    lap("hierarchy");
    const int rc = finish_build(g, T);
    lap("PCG state");
    if (n_up > 0) {
        qq_up.join();
        for (int t = 0; t < n_up; t++) IRH_CHECK(qq_up.err[t]);
        lap("join of the rotation upload");
    }
    return rc;
}

}  // namespace irh

// resident.hip -- the device-resident, GROWING copy of a view-graph ("device-resident growing
// graph: append views/edges, pattern update, warm Q") behind irotavg_viewgraph_rot_avg's global re-solves
// (rotAvg(5000000) on every loop closure, src/IRotAvg.cpp:371-378; ViewGraph::rotAvg, src/ViewGraph.cpp:1263-1435).
//
// What lives in HBM between calls, in the CALLER's view ids (nothing is relabelled in place):
//   edge records   int2 (i, j) + double4 relative rotation, in the order rotAvg walks them (view ascending, lower
//                  endpoint ascending, src/ViewGraph.cpp:1290-1300) -- append-only as long as new connections attach
//                  to the newest views (what a stream does); a connection to an older view re-sends the tail behind it;
//   poses          3 x 3 row-major per view (Pose::R) and the fixed mask (m_fixed_mask).
// A global re-solve then costs the host the records of what changed since the last one (the views admitted since,
// the few poses the sliding windows moved) instead of the whole graph: the fixed-first relabelling of
// src/ViewGraph.cpp:1323-1363 is a scan over the mask ON THE DEVICE, the patterns of the solver (SELL-64 operator,
// block-cyclic-reduction plan) are rebuilt on the device from the resident records (gbuild.hip, bcr_plan_dev -- every
// newly fixed view renumbers the rows behind it, so the patterns cannot be kept), R -> quaternion
// (src/ViewGraph.cpp:1175-1203) and quaternion -> R (:1426-1433) run as kernels with the host's arithmetic (no
// contraction into FMAs), and the solved rotations come back as one copy of 32 bytes per view.
#include <hip/hip_runtime.h>

#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <cmath>

#include "graph.hpp"

namespace irh {

namespace {

constexpr int kT = 256;
inline unsigned grid_of(long long n) { return (unsigned)std::max<long long>(1, (n + kT - 1) / kT); }

// src/ViewGraph.cpp:1175-1203, statement for statement as viewgraph.cpp's host function (same roundings: every
// operation is a single IEEE add / multiply / divide / square root)
__device__ inline double4 d_rmat2quat(const double *__restrict__ R) {
#pragma clang fp contract(off)
    double q[4];
    const double trace = R[0] + R[4] + R[8];
    if (trace > 0.0) {
        double s = sqrt(trace + 1.0);
        q[3] = s * 0.5;
        s = 0.5 / s;
        q[0] = (R[7] - R[5]) * s;
        q[1] = (R[2] - R[6]) * s;
        q[2] = (R[3] - R[1]) * s;
    } else {
        const int i = R[0] < R[4] ? (R[4] < R[8] ? 2 : 1) : (R[0] < R[8] ? 2 : 0);
        const int j = (i + 1) % 3, k = (i + 2) % 3;
        double s = sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
        q[i] = s * 0.5;
        s = 0.5 / s;
        q[3] = (R[3 * k + j] - R[3 * j + k]) * s;
        q[j] = (R[3 * j + i] + R[3 * i + j]) * s;
        q[k] = (R[3 * k + i] + R[3 * i + k]) * s;
    }
    return make_double4(q[0], q[1], q[2], q[3]);
}

// q.normalized().toRotationMatrix() (src/ViewGraph.cpp:1426-1433), as viewgraph.cpp's quat2rmat
__device__ inline void d_quat2rmat(double4 qin, double *__restrict__ R) {
#pragma clang fp contract(off)
    double x = qin.x, y = qin.y, z = qin.z, w = qin.w;
    const double n2 = x * x + y * y + z * z + w * w;
    if (n2 > 0.0) {
        const double nn = sqrt(n2);
        x /= nn;
        y /= nn;
        z /= nn;
        w /= nn;
    }
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz);
    R[1] = txy - twz;
    R[2] = txz + twy;
    R[3] = txy + twz;
    R[4] = 1 - (txx + tzz);
    R[5] = tyz - twx;
    R[6] = txz - twy;
    R[7] = tyz + twx;
    R[8] = 1 - (txx + tyy);
}

__global__ __launch_bounds__(kT) void k_res_flags(int n, const uint8_t *__restrict__ fixed, int *__restrict__ flag) {
    const int x = blockIdx.x * kT + threadIdx.x;
    if (x < n) flag[x] = fixed[x] ? 1 : 0;
}
// fixed views first, both groups in ascending view id (src/ViewGraph.cpp:1340-1363)
__global__ __launch_bounds__(kT) void k_res_relabel(int n, int f, const uint8_t *__restrict__ fixed,
                                                    const int *__restrict__ before, int *__restrict__ v2i) {
    const int x = blockIdx.x * kT + threadIdx.x;
    if (x < n) v2i[x] = fixed[x] ? before[x] : f + (x - before[x]);
}
// Q of the call (src/ViewGraph.cpp:1365-1380): row v2i[x] = rmat2quat(pose of view x)
__global__ __launch_bounds__(kT) void k_res_gather(int n, const double *__restrict__ R, const int *__restrict__ v2i,
                                                   double4 *__restrict__ Q) {
    const int x = blockIdx.x * kT + threadIdx.x;
    if (x < n) Q[v2i[x]] = d_rmat2quat(R + 9 * (size_t)x);
}
// write-back for the free views (src/ViewGraph.cpp:1420-1434): the solved quaternion goes to the host as it is
// (which forms the same matrix), the resident pose takes q.normalized().toRotationMatrix()
__global__ __launch_bounds__(kT) void k_res_writeback(int n, int f, const int *__restrict__ v2i,
                                                      const double4 *__restrict__ Q, double4 *__restrict__ qout,
                                                      double *__restrict__ R) {
    const int x = blockIdx.x * kT + threadIdx.x;
    if (x >= n) return;
    const int r = v2i[x];
    double4 q = make_double4(0, 0, 0, 0);
    if (r >= f) {
        q = Q[r];
        d_quat2rmat(q, R + 9 * (size_t)x);
    }
    qout[x] = q;
}

// pinned, growing host staging
struct PinBuf {
    void *p = nullptr;
    size_t cap = 0;
    void *need(size_t bytes) {
        if (bytes <= cap) return p;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 2 + 4096;
        IRH_CHECK(hipHostMalloc(&p, want, hipHostMallocDefault));
        cap = want;
        return p;
    }
    ~PinBuf() {
        if (p) (void)hipHostFree(p);
    }
};

template <class T>
void grow_keep(DevBuf<T> &b, size_t keep, size_t need, hipStream_t s) {
    if (b.n >= need) return;
    DevBuf<T> nb;
    nb.alloc(need + need / 2 + 1024);
    if (keep > 0 && b.p) IRH_CHECK(hipMemcpyAsync(nb.p, b.p, sizeof(T) * keep, hipMemcpyDeviceToDevice, s));
    IRH_CHECK(hipStreamSynchronize(s));  // the old block goes back to the pool
    b = std::move(nb);
}

}  // namespace

struct Resident {
    int device = -1;
    hipStream_t stream = nullptr;
    // resident records (caller ids)
    DevBuf<int2> I;
    DevBuf<double4> QQ;
    DevBuf<double> R;       // 9 per view
    DevBuf<uint8_t> fixed;  // per view
    DevBuf<int> flag, before, v2i;
    DevBuf<double4> qout;
    DevBuf<char> scan_tmp;
    long n_views = 0, n_edges = 0;  // what the device holds
    PinBuf hR, hfixed, hI, hqq, hQ;
    ~Resident() {
        if (stream) {
            (void)hipStreamSynchronize(stream);
            StreamPool::get().give(stream, device);
        }
    }
};

Resident *resident_new() { return new Resident(); }
void resident_delete(Resident *r) { delete r; }
void resident_invalidate(Resident &r) {
    r.n_views = 0;
    r.n_edges = 0;
}
long resident_views(const Resident &r) { return r.n_views; }
long resident_edges(const Resident &r) { return r.n_edges; }

ResidentStage resident_stage(Resident &r, long n_views, long view_lo, long n_edges, long edge_lo) {
    ResidentStage st{};
    const size_t dv = (size_t)std::max<long>(0, n_views - view_lo), de = (size_t)std::max<long>(0, n_edges - edge_lo);
    st.R = static_cast<double *>(r.hR.need(sizeof(double) * 9 * dv + 64));
    st.fixed = static_cast<uint8_t *>(r.hfixed.need(dv + 64));
    st.I = static_cast<int32_t *>(r.hI.need(sizeof(int32_t) * 2 * de + 64));
    st.qq = static_cast<double *>(r.hqq.need(sizeof(double) * 4 * de + 64));
    st.Q = static_cast<double *>(r.hQ.need(sizeof(double) * 4 * (size_t)n_views + 64));
    return st;
}

// One global re-solve on the resident graph. The staging blocks of resident_stage hold the records of the views
// [view_lo, n_views) and of the edges [edge_lo, n_edges); everything below those marks must be what the device holds.
// f = fixed views (> 0). On success st.Q (n_views x 4, AoS) holds the solved quaternion of every free view.
// dry: everything runs (allocations, kernels, the solve) but no pose changes, on the device or for the caller -- what
// irotavg_viewgraph_prepare uses to take the one-time costs of a process out of the first loop closure's latency.
// A dry run also solves with ONE made-up loop closure (dry_a, dry_b: two free views far apart, identity relative
// rotation; < 0: none) behind the real records: the closure path of the direct solver (its plan, its kernels) is what
// the first real call after a loop closure needs, and the graph a caller prepares usually has none yet.
int resident_rot_avg(Resident &r, long n_views, long view_lo, long n_edges, long edge_lo, int f,
                     const irotavg_options &opt, irotavg_rotavg_info &loc, bool timing, bool dry, int dry_a, int dry_b) {
    if (view_lo > r.n_views || edge_lo > r.n_edges || view_lo < 0 || edge_lo < 0 || f <= 0 || f >= n_views)
        return IROTAVG_ERR_BAD_ARG;
    double tl = now_seconds();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const double t = now_seconds();
        std::fprintf(stderr, "[rot_avg resident] %-24s %8.3f ms\n", what, 1e3 * (t - tl));
        tl = t;
    };
    irotavg_graph *h = nullptr;
    try {
        if (opt.device >= 0) IRH_CHECK(hipSetDevice(opt.device));
        if (!r.stream) {
            IRH_CHECK(hipGetDevice(&r.device));
            r.stream = StreamPool::get().take();
        }
        hipStream_t s = r.stream;
        const size_t nv = (size_t)n_views, ne = (size_t)n_edges;
        grow_keep(r.I, (size_t)edge_lo, ne + 1, s);
        grow_keep(r.QQ, (size_t)edge_lo, ne + 1, s);
        grow_keep(r.R, 9 * (size_t)view_lo, 9 * nv, s);
        grow_keep(r.fixed, (size_t)view_lo, nv, s);
        if (r.flag.n < nv + 1) {
            r.flag.alloc(nv + nv / 2 + 1024);
            r.before.alloc(r.flag.n);
            r.v2i.alloc(r.flag.n);
            r.qout.alloc(r.flag.n);
        }
        const size_t dv = nv - (size_t)view_lo, de = ne - (size_t)edge_lo;
        if (de > 0) {
            IRH_CHECK(hipMemcpyAsync(r.I.p + edge_lo, r.hI.p, sizeof(int2) * de, hipMemcpyHostToDevice, s));
            IRH_CHECK(hipMemcpyAsync(r.QQ.p + edge_lo, r.hqq.p, sizeof(double4) * de, hipMemcpyHostToDevice, s));
        }
        if (dv > 0) {
            IRH_CHECK(hipMemcpyAsync(r.R.p + 9 * (size_t)view_lo, r.hR.p, sizeof(double) * 9 * dv, hipMemcpyHostToDevice, s));
            IRH_CHECK(hipMemcpyAsync(r.fixed.p + view_lo, r.hfixed.p, dv, hipMemcpyHostToDevice, s));
        }
        r.n_views = n_views;
        r.n_edges = n_edges;
        long ne_solve = n_edges;
        if (dry && dry_a >= 0 && dry_b > dry_a && dry_b < n_views) {  // the made-up closure (never part of the resident records)
            const int2 e = make_int2(dry_a, dry_b);
            const double4 q = make_double4(0.0, 0.0, 0.0, 1.0);
            IRH_CHECK(hipMemcpyAsync(r.I.p + ne, &e, sizeof(e), hipMemcpyHostToDevice, s));
            IRH_CHECK(hipMemcpyAsync(r.QQ.p + ne, &q, sizeof(q), hipMemcpyHostToDevice, s));
            IRH_CHECK(hipStreamSynchronize(s));
            ne_solve = n_edges + 1;
        }
        // ---- fixed-first relabelling (src/ViewGraph.cpp:1323-1363): a scan over the mask
        hipLaunchKernelGGL(k_res_flags, dim3(grid_of(n_views)), dim3(kT), 0, s, (int)n_views, r.fixed.p, r.flag.p);
        {
            size_t bytes = 0;
            IRH_CHECK(rocprim::exclusive_scan(nullptr, bytes, r.flag.p, r.before.p, 0, nv, rocprim::plus<int>(), s));
            if (r.scan_tmp.n < bytes) r.scan_tmp.alloc(bytes + 4096);
            IRH_CHECK(rocprim::exclusive_scan(r.scan_tmp.p, bytes, r.flag.p, r.before.p, 0, nv, rocprim::plus<int>(), s));
        }
        hipLaunchKernelGGL(k_res_relabel, dim3(grid_of(n_views)), dim3(kT), 0, s, (int)n_views, f, r.fixed.p, r.before.p,
                           r.v2i.p);
        IRH_CHECK(hipStreamSynchronize(s));  // the build runs on the handle's own stream
        lap("delta upload + relabel");
        // ---- the solver's handle, built on the device from the resident records
        DevEdgeSrc src;
        src.I = r.I.p;
        src.QQ = r.QQ.p;
        src.relabel = r.v2i.p;
        // (the handle of a growing graph and everything its solves allocate: blocks half as large again, so that the next
        // re-solves find them in the pool)
        DevPool::HeadroomScope headroom;  // this thread; the handle's worker threads follow its flag (Graph::pool_headroom)
        int rc = graph_create_dev(&h, ne_solve, n_views, f, src, &opt);
        if (rc != IROTAVG_OK) return rc;
        Graph &g = graph_of(h);
        g.pool_headroom = true;
        hipLaunchKernelGGL(k_res_gather, dim3(grid_of(n_views)), dim3(kT), 0, g.stream, (int)n_views, r.R.p, r.v2i.p, g.Q.p);
        if (timing) (void)hipStreamSynchronize(g.stream);
        lap("handle (device build)");
        // ---- solve (src/ViewGraph.cpp:1396-1417)
        const double change_th = .001;
        rc = irotavg_graph_l1ra(h, 100, change_th, &loc.l1_iters, &loc.l1_runtime, nullptr);
        lap("l1ra");
        if (rc == IROTAVG_OK)
            rc = irotavg_graph_irls(h, IROTAVG_GEMAN_MCCLURE, 5 * M_PI / 180.0, 100, change_th, &loc.irls_iters,
                                    &loc.irls_runtime, nullptr);
        lap("irls");
        if (rc == IROTAVG_OK && dry) {
            (void)hipStreamSynchronize(g.stream);
        } else if (rc == IROTAVG_OK) {
            hipLaunchKernelGGL(k_res_writeback, dim3(grid_of(n_views)), dim3(kT), 0, g.stream, (int)n_views, f, r.v2i.p,
                               g.Q.p, r.qout.p, r.R.p);
            IRH_CHECK(hipMemcpyAsync(r.hQ.p, r.qout.p, sizeof(double4) * nv, hipMemcpyDeviceToHost, g.stream));
            IRH_CHECK(hipStreamSynchronize(g.stream));
            lap("write-back + download");
        } else {
            (void)hipStreamSynchronize(g.stream);
        }
        irotavg_graph_destroy(h);
        h = nullptr;
        lap("destroy");
        if (rc != IROTAVG_OK) resident_invalidate(r);  // the poses on the device may be half-way: send everything again
        return rc;
    } catch (const HipError &) {
        if (h) irotavg_graph_destroy(h);
        resident_invalidate(r);
        return IROTAVG_ERR_HIP;
    } catch (const std::bad_alloc &) {
        if (h) irotavg_graph_destroy(h);
        resident_invalidate(r);
        return IROTAVG_ERR_NOMEM;
    } catch (...) {
        if (h) irotavg_graph_destroy(h);
        resident_invalidate(r);
        return IROTAVG_ERR_HIP;
    }
}

}  // namespace irh

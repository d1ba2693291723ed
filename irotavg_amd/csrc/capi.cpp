// capi.cpp -- extern "C" boundary of libirotavg_hip.so (see include/irotavg_hip.h).
// No exception crosses this file; there is no CPU fallback for the compute entry points.
#include <atomic>
#include <cmath>
#include <mutex>
#include <new>

#include "graph.hpp"

using namespace irh;

struct irotavg_graph {
    Graph g;
};

#define API_TRY try {
#define API_CATCH                          \
    }                                      \
    catch (const HipError &) {             \
        return IROTAVG_ERR_HIP;            \
    }                                      \
    catch (const std::bad_alloc &) {       \
        return IROTAVG_ERR_NOMEM;          \
    }                                      \
    catch (...) {                          \
        return IROTAVG_ERR_HIP;            \
    }

extern "C" {

const char *irotavg_version(void) { return "irotavg_hip 0.1 (gfx950)"; }

int irotavg_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *irotavg_error_string(int code) {
    switch (code) {
    case IROTAVG_OK: return "ok";
    case IROTAVG_ERR_BAD_ARG: return "bad argument";
    case IROTAVG_ERR_NOT_SPANNING: return "relative rotations do not span all the nodes in the view graph";
    case IROTAVG_ERR_SOLVER: return "linear solver breakdown (non-finite values)";
    case IROTAVG_ERR_UNKNOWN_COST: return "unknown cost";
    case IROTAVG_ERR_NOMEM: return "out of memory";
    case IROTAVG_ERR_HIP: return "HIP runtime error";
    case IROTAVG_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU fallback)";
    case IROTAVG_ERR_NOT_CONVERGED: return "inner PCG did not converge within pcg_max_iters";
    default: return "unknown error";
    }
}

void irotavg_default_options(irotavg_options *o) {
    if (!o) return;
    std::memset(o, 0, sizeof(*o));
    o->pcg_rtol = 1e-10;
    o->pcg_max_iters = 2000;
    o->pcg_check_every = 8;
    o->mg_levels_max = 16;
    o->mg_agg0 = 0;
    o->mg_agg = 0;
    o->mg_dense_max = 0;  // choose from the graph structure (build.cpp): 2048, or 1100 with loop closures
    o->mg_omega = 0.7;
    o->mg_kc = 0.0;  // choose from the graph structure (build.cpp)
    o->device = -1;
}

// ---------------------------------------------------------------------------------------------
// host-side pieces of the RAL API
// ---------------------------------------------------------------------------------------------
static inline void h_qmul(const double a[4], const double b[4], double o[4]) {
    // [x y z w] Hamilton product (ral/l1_irls.cpp:99-105)
    const double w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    const double x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    const double y = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    const double z = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
    o[0] = x;
    o[1] = y;
    o[2] = z;
    o[3] = w;
}

// init_mst is a sequential, edge-order-dependent propagation (ral/l1_irls.cpp:915-979): each
// sweep visits the edge list in order and a view takes its value from the first edge that
// reaches it, so it is evaluated on the host exactly in that order. It is outside the
// reference's timed region and O(sweeps * m).
int irotavg_init_mst(int64_t n, int64_t m, double *Q, int64_t ldq, const double *QQ,
                     int64_t ldqq, const int32_t *I, int f) {
    if (!Q || !QQ || !I || n <= 0 || m < 0 || f <= 0 || ldq < n || ldqq < m)
        return IROTAVG_ERR_BAD_ARG;  // assert(f>0) at :917
    std::vector<char> seen((size_t)n, 0);
    seen[0] = 1;
    int64_t count = 1;
    while (count < n) {
        bool grew = false;
        for (int64_t k = 0; k < m; k++) {
            const int64_t a = I[2 * k], b = I[2 * k + 1];
            if (a < 0 || b < 0 || a >= n || b >= n) return IROTAVG_ERR_BAD_ARG;
            if (seen[a] && !seen[b]) {
                if (b >= f) {  // forward: Q_b = QQ_k (x) Q_a (:941)
                    const double qq[4] = {QQ[k], QQ[ldqq + k], QQ[2 * ldqq + k], QQ[3 * ldqq + k]};
                    const double qa[4] = {Q[a], Q[ldq + a], Q[2 * ldq + a], Q[3 * ldq + a]};
                    double r[4];
                    h_qmul(qq, qa, r);
                    for (int c = 0; c < 4; c++) Q[c * ldq + b] = r[c];
                }
                seen[b] = 1;
                count++;
                grew = true;
            } else if (!seen[a] && seen[b]) {
                if (a >= f) {  // backward: inverse formed by negating w only (:956-958)
                    const double qq[4] = {QQ[k], QQ[ldqq + k], QQ[2 * ldqq + k], -QQ[3 * ldqq + k]};
                    const double qb[4] = {Q[b], Q[ldq + b], Q[2 * ldq + b], Q[3 * ldq + b]};
                    double r[4];
                    h_qmul(qq, qb, r);
                    for (int c = 0; c < 4; c++) Q[c * ldq + a] = r[c];
                }
                seen[a] = 1;
                count++;
                grew = true;
            }
        }
        if (!grew && count < n) return IROTAVG_ERR_NOT_SPANNING;  // exit(-1) at :970-977
    }
    return IROTAVG_OK;
}

int64_t irotavg_make_A(int n, int f, int64_t m, const int32_t *I, int64_t *colptr,
                       int64_t *rowidx, double *vals) {
    if (n < 0 || f < 0 || n - f <= 1 || !I || !colptr || !rowidx || !vals)
        return IROTAVG_ERR_BAD_ARG;  // asserts at ral/l1_irls.cpp:757-758
    const int64_t nu = (int64_t)n - f;
    std::vector<int64_t> cnt((size_t)nu + 1, 0);
    auto row = [&](int64_t k, int64_t c[2], double v[2]) -> int {
        const int64_t j = (int64_t)I[2 * k + 1] - f, i = (int64_t)I[2 * k] - f;
        if (j < 0) return 0;  // :770-771 -- edge dropped even if i is free
        if (i < 0) {
            c[0] = j;
            v[0] = 1.0;
            return 1;
        }
        if (i == j) {  // the second coeffRef overwrites the first
            c[0] = i;
            v[0] = -1.0;
            return 1;
        }
        c[0] = j;
        v[0] = 1.0;
        c[1] = i;
        v[1] = -1.0;
        return 2;
    };
    int64_t c[2];
    double v[2];
    for (int64_t k = 0; k < m; k++) {
        const int ne = row(k, c, v);
        for (int e = 0; e < ne; e++) {
            if (c[e] >= nu) return IROTAVG_ERR_BAD_ARG;
            cnt[c[e]]++;
        }
    }
    colptr[0] = 0;
    for (int64_t q = 0; q < nu; q++) colptr[q + 1] = colptr[q] + cnt[q];
    for (int64_t q = 0; q < nu; q++) cnt[q] = colptr[q];
    for (int64_t k = 0; k < m; k++) {
        const int ne = row(k, c, v);
        for (int e = 0; e < ne; e++) {
            const int64_t p = cnt[c[e]]++;
            rowidx[p] = k;
            vals[p] = v[e];
        }
    }
    return colptr[nu];
}

// ---------------------------------------------------------------------------------------------
// handle API
// ---------------------------------------------------------------------------------------------
}  // extern "C"

// the handle behind irotavg_graph_create; src != nullptr: the edge list lives on the device already (resident.hip)
static int graph_create_impl(irotavg_graph **out, int64_t m, int64_t n_total, int f, const int32_t *I,
                             const double *QQ, int64_t ldqq, const irotavg_options *opt, const DevEdgeSrc *src) {
    if (!out || (!src && (!I || !QQ || ldqq < m)) || m <= 0 || n_total <= 0 || f < 0 || n_total - f < 1 ||
        n_total > 0x7fffffffLL)
        return IROTAVG_ERR_BAD_ARG;
    *out = nullptr;
    if (irotavg_device_count() <= 0) return IROTAVG_ERR_NO_DEVICE;
    irotavg_graph *h = nullptr;
    API_TRY
    h = new irotavg_graph();
    Graph &g = h->g;
    if (opt)
        g.opt = *opt;
    else
        irotavg_default_options(&g.opt);
    if (g.opt.pcg_rtol <= 0) g.opt.pcg_rtol = 1e-10;
    if (g.opt.mg_omega <= 0) g.opt.mg_omega = 0.7;
    if (g.opt.mg_levels_max <= 0) g.opt.mg_levels_max = 16;
    if (g.opt.pcg_max_iters <= 0) g.opt.pcg_max_iters = 2000;
    if (g.opt.pcg_check_every <= 0) g.opt.pcg_check_every = 8;
    if (g.opt.device >= 0) IRH_CHECK(hipSetDevice(g.opt.device));
    IRH_CHECK(hipGetDevice(&g.device));
    g.stream = StreamPool::get().take();
    g.m = m;
    g.n_total = n_total;
    g.f = f;
    g.nu = (int)(n_total - f);
    g.ng = 0;
    g.no = g.nu;
    if (const char *e = std::getenv("IROTAVG_STALE_SPREAD")) g.stale_spread = std::max(1.0, std::atof(e));  // experiments
    // a banded operator (+ a few loop closures) is solved directly (bcr.hip): level 0 is all such a handle needs
    // (no coarse patterns, no dense level: a third of the build)
    if (src)
        bcr_plan_dev(g, *src);
    else
        bcr_plan(g, I);  // (an edge list with an index out of range plans nothing; the build rejects it)
    if (g.bcr_B) g.opt.mg_levels_max = 1;
    const int rc = src ? build_graph_device(g, nullptr, nullptr, 0, src) : build_graph(g, I, QQ, ldqq);
    if (rc != IROTAVG_OK) {
        irotavg_graph_destroy(h);
        return rc;
    }
    *out = h;
    return IROTAVG_OK;
    }
    catch (const HipError &) {
        if (h) irotavg_graph_destroy(h);
        return IROTAVG_ERR_HIP;
    }
    catch (const std::bad_alloc &) {
        if (h) irotavg_graph_destroy(h);
        return IROTAVG_ERR_NOMEM;
    }
    catch (...) {
        if (h) irotavg_graph_destroy(h);
        return IROTAVG_ERR_HIP;
    }
}

namespace irh {
int graph_create_dev(irotavg_graph **out, int64_t m, int64_t n_total, int f, const DevEdgeSrc &src,
                     const irotavg_options *opt) {
    if (!src.I || !src.QQ) return IROTAVG_ERR_BAD_ARG;
    return graph_create_impl(out, m, n_total, f, nullptr, nullptr, 0, opt, &src);
}
Graph &graph_of(irotavg_graph *h) { return h->g; }
}  // namespace irh

extern "C" {

int irotavg_graph_create(irotavg_graph **out, int64_t m, int64_t n_total, int f, const int32_t *I,
                         const double *QQ, int64_t ldqq, const irotavg_options *opt) {
    return graph_create_impl(out, m, n_total, f, I, QQ, ldqq, opt, nullptr);
}

// Device buffers of destroyed handles are cached for reuse (common.hpp, DevPool): hand them back.
int64_t irotavg_trim_memory(void) {
    try {
        StreamPool::get().trim();
        return (int64_t)DevPool::get().trim();
    } catch (...) {
        return 0;
    }
}

void irotavg_graph_destroy(irotavg_graph *h) {
    if (!h) return;
    const bool timing = std::getenv("IROTAVG_BUILD_TIMING") != nullptr;
    double t0 = now_seconds();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const double t = now_seconds();
        std::fprintf(stderr, "[irotavg_hip destroy] %-24s %8.3f ms\n", what, 1e3 * (t - t0));
        t0 = t;
    };
    hipStream_t s = h->g.stream;
    const int dev = h->g.device;
    if (s) (void)hipStreamSynchronize(s);
    lap("sync");
    release_l1_clones(h->g);
    lap("clones");
    h->g.stream = nullptr;
    delete h;
    lap("delete");
    if (s) StreamPool::get().give(s, dev);
    lap("stream destroy");
}

int irotavg_graph_set_rotations(irotavg_graph *h, const double *Q, int64_t ldq) {
    if (!h || !Q || ldq < h->g.n_total) return IROTAVG_ERR_BAD_ARG;
    API_TRY
    Graph &g = h->g;
    std::vector<double4> aos((size_t)g.n_total);
    parallel_for(g.n_total, 8192, [&](int64_t a, int64_t b, int) {
        for (int64_t i = a; i < b; i++) aos[i] = make_double4(Q[i], Q[ldq + i], Q[2 * ldq + i], Q[3 * ldq + i]);
    });
    g.Q.upload(aos.data(), aos.size(), g.stream);
    IRH_CHECK(hipStreamSynchronize(g.stream));
    return IROTAVG_OK;
    API_CATCH
}

int irotavg_graph_get_rotations(irotavg_graph *h, double *Q, int64_t ldq) {
    if (!h || !Q || ldq < h->g.n_total) return IROTAVG_ERR_BAD_ARG;
    API_TRY
    Graph &g = h->g;
    std::vector<double4> aos((size_t)g.n_total);
    IRH_CHECK(hipMemcpyAsync(aos.data(), g.Q.p, sizeof(double4) * aos.size(), hipMemcpyDeviceToHost,
                             g.stream));
    IRH_CHECK(hipStreamSynchronize(g.stream));
    parallel_for(g.n_total, 8192, [&](int64_t a, int64_t b, int) {
        for (int64_t i = a; i < b; i++) {
            Q[i] = aos[i].x;
            Q[ldq + i] = aos[i].y;
            Q[2 * ldq + i] = aos[i].z;
            Q[3 * ldq + i] = aos[i].w;
        }
    });
    return IROTAVG_OK;
    API_CATCH
}

int irotavg_graph_snapshot_rotations(irotavg_graph *h) {
    if (!h) return IROTAVG_ERR_BAD_ARG;
    API_TRY
    Graph &g = h->g;
    if (g.Qsnap.n < (size_t)g.n_total) g.Qsnap.alloc((size_t)g.n_total);
    IRH_CHECK(hipMemcpyAsync(g.Qsnap.p, g.Q.p, sizeof(double4) * (size_t)g.n_total,
                             hipMemcpyDeviceToDevice, g.stream));
    IRH_CHECK(hipStreamSynchronize(g.stream));
    return IROTAVG_OK;
    API_CATCH
}

int irotavg_graph_restore_rotations(irotavg_graph *h) {
    if (!h || h->g.Qsnap.n < (size_t)h->g.n_total) return IROTAVG_ERR_BAD_ARG;
    API_TRY
    Graph &g = h->g;
    IRH_CHECK(hipMemcpyAsync(g.Q.p, g.Qsnap.p, sizeof(double4) * (size_t)g.n_total,
                             hipMemcpyDeviceToDevice, g.stream));
    return IROTAVG_OK;
    API_CATCH
}

int irotavg_graph_get_weights(irotavg_graph *h, double *w) {
    if (!h || !w) return IROTAVG_ERR_BAD_ARG;
    API_TRY
    Graph &g = h->g;
    IRH_CHECK(hipMemcpyAsync(w, g.dw.p, sizeof(double) * (size_t)g.m, hipMemcpyDeviceToHost, g.stream));
    IRH_CHECK(hipStreamSynchronize(g.stream));
    return IROTAVG_OK;
    API_CATCH
}

int irotavg_graph_set_weights(irotavg_graph *h, const double *w) {
    if (!h || !w) return IROTAVG_ERR_BAD_ARG;
    API_TRY
    Graph &g = h->g;
    IRH_CHECK(hipMemcpyAsync(g.dw.p, w, sizeof(double) * (size_t)g.m, hipMemcpyHostToDevice, g.stream));
    IRH_CHECK(hipStreamSynchronize(g.stream));
    return IROTAVG_OK;
    API_CATCH
}

int irotavg_graph_irls(irotavg_graph *h, int cost, double sigma, int max_iters, double change_th,
                       int *iters, double *runtime, double *trace) {
    if (!h || !iters || !runtime) return IROTAVG_ERR_BAD_ARG;
    API_TRY
    return run_irls(h->g, cost, sigma, max_iters, change_th, iters, runtime, trace);
    API_CATCH
}

int irotavg_graph_l1ra(irotavg_graph *h, int max_iters, double change_th, int *iter, double *runtime,
                       double *trace) {
    if (!h || !iter || !runtime) return IROTAVG_ERR_BAD_ARG;
    API_TRY
    return run_l1ra(h->g, max_iters, change_th, iter, runtime, trace);
    API_CATCH
}

int irotavg_graph_quat_normalised(irotavg_graph *h) {
    if (!h) return IROTAVG_ERR_BAD_ARG;
    API_TRY
    normalise_rotations(h->g);
    IRH_CHECK(hipStreamSynchronize(h->g.stream));
    return IROTAVG_OK;
    API_CATCH
}

int irotavg_graph_get_stats(irotavg_graph *h, irotavg_stats *out) {
    if (!h || !out) return IROTAVG_ERR_BAD_ARG;
    *out = h->g.stats;
    out->band = h->g.band0;
    out->band_block = h->g.bcr_B;
    return IROTAVG_OK;
}

int irotavg_graph_direct_info(irotavg_graph *h, int64_t *info, int cap) {
    if (!h || !info || cap < 1) return IROTAVG_ERR_BAD_ARG;
    API_TRY
    return bcr_info(h->g, info, cap);
    API_CATCH
}

int irotavg_graph_direct_residual(irotavg_graph *h, double *relres) {
    if (!h || !relres) return IROTAVG_ERR_BAD_ARG;
    API_TRY
    return bcr_residual(h->g, relres);
    API_CATCH
}

void irotavg_graph_reset_stats(irotavg_graph *h) {
    if (!h) return;
    irotavg_stats keep = h->g.stats;
    std::memset(&h->g.stats, 0, sizeof(irotavg_stats));
    h->g.stats.levels = keep.levels;
    for (int i = 0; i < 16; i++) {
        h->g.stats.level_rows[i] = keep.level_rows[i];
        h->g.stats.level_nnz[i] = keep.level_nnz[i];
    }
}

int irotavg_graph_synchronize(irotavg_graph *h) {
    if (!h) return IROTAVG_ERR_BAD_ARG;
    if (hipStreamSynchronize(h->g.stream) != hipSuccess) return IROTAVG_ERR_HIP;
    bcr_up_release(h->g);
    return IROTAVG_OK;
}

int irotavg_graph_edge_residual(irotavg_graph *h) {
    if (!h) return IROTAVG_ERR_BAD_ARG;
    API_TRY
    launch_edge_residual(h->g);
    IRH_CHECK(hipStreamSynchronize(h->g.stream));
    return IROTAVG_OK;
    API_CATCH
}

int irotavg_graph_get_residuals(irotavg_graph *h, double *out, int64_t ld) {
    if (!h || !out || ld < h->g.m) return IROTAVG_ERR_BAD_ARG;
    API_TRY
    Graph &g = h->g;
    for (int c = 0; c < 3; c++)
        IRH_CHECK(hipMemcpyAsync(out + (size_t)c * ld, g.er.p + (size_t)c * g.mpad,
                                 sizeof(double) * (size_t)g.m, hipMemcpyDeviceToHost, g.stream));
    IRH_CHECK(hipStreamSynchronize(g.stream));
    return IROTAVG_OK;
    API_CATCH
}

int irotavg_graph_ls_solve(irotavg_graph *h, double *X, int64_t ldx) {
    if (!h || (X && ldx < h->g.nu)) return IROTAVG_ERR_BAD_ARG;
    API_TRY
    Graph &g = h->g;
    const int rc = ls_solve(g);
    if (X) {
        std::vector<double4> aos((size_t)g.nu);
        IRH_CHECK(hipMemcpyAsync(aos.data(), g.X.p, sizeof(double4) * aos.size(),
                                 hipMemcpyDeviceToHost, g.stream));
        IRH_CHECK(hipStreamSynchronize(g.stream));
        for (int i = 0; i < g.nu; i++) {
            X[i] = aos[i].x;
            X[ldx + i] = aos[i].y;
            X[2 * ldx + i] = aos[i].z;
        }
    }
    return rc;
    API_CATCH
}

int irotavg_graph_update_weights(irotavg_graph *h, int cost, double sigma) {
    if (!h) return IROTAVG_ERR_BAD_ARG;
    if (cost < IROTAVG_L2 || cost > IROTAVG_WELSCH) return IROTAVG_ERR_UNKNOWN_COST;
    API_TRY
    launch_update_weights(h->g, cost, sigma);
    IRH_CHECK(hipStreamSynchronize(h->g.stream));
    return IROTAVG_OK;
    API_CATCH
}

int irotavg_graph_apply_step(irotavg_graph *h, double *score) {
    if (!h || !score) return IROTAVG_ERR_BAD_ARG;
    API_TRY
    *score = apply_step(h->g);
    return IROTAVG_OK;
    API_CATCH
}

int irotavg_graph_l1decode_pd(irotavg_graph *h, const double *y, int pdmaxiter, double *x,
                              int *stuck) {
    if (!h || !y || !x) return IROTAVG_ERR_BAD_ARG;
    API_TRY
    return l1decode_pd_dev(h->g, -1, y, pdmaxiter, x, stuck, 0);
    API_CATCH
}

int irotavg_graph_fingerprint(irotavg_graph *h, uint64_t *out, int cap) {
    if (!h || !out || cap < 0) return IROTAVG_ERR_BAD_ARG;
    API_TRY
    Graph &g = h->g;
    IRH_CHECK(hipStreamSynchronize(g.stream));
    std::vector<uint64_t> v;
    std::vector<unsigned char> host;
    auto fnv = [](const unsigned char *p, size_t n) {
        uint64_t hsh = 1469598103934665603ull;
        for (size_t i = 0; i < n; i++) hsh = (hsh ^ p[i]) * 1099511628211ull;
        return hsh;
    };
    auto arr = [&](const void *dev, size_t bytes) {
        if (!dev || bytes == 0) {
            v.push_back(0);
            return;
        }
        host.resize(bytes);
        IRH_CHECK(hipMemcpy(host.data(), dev, bytes, hipMemcpyDeviceToHost));
        v.push_back(fnv(host.data(), bytes));
    };
    auto last_int = [&](const int *dev, size_t idx) {
        int x = 0;
        if (dev) IRH_CHECK(hipMemcpy(&x, dev + idx, sizeof(int), hipMemcpyDeviceToHost));
        return x;
    };
    const size_t mp = (size_t)g.mpad;
    arr(g.ei.p, 4 * mp);
    arr(g.ej.p, 4 * mp);
    arr(g.eflag.p, mp);
    arr(g.qq.p, 8 * 4 * mp);
    arr(g.bptr.p, 4 * ((size_t)g.no + 1));
    const size_t nb = (size_t)last_int(g.bptr.p, (size_t)g.no);
    arr(g.beid.p, 4 * nb);
    arr(g.bflag.p, nb);
    arr(g.bghost.p, 4 * nb);
    const Level &L0 = g.levels[0];
    arr(g.slot_eid.p, 4 * (size_t)L0.sell_len);
    arr(g.slot_cs.p, g.slot_cs.p ? (size_t)L0.sell_len : 0);
    arr(g.tile_e0.p, 4 * (size_t)L0.nsl);
    v.push_back((uint64_t)g.levels.size());
    for (size_t l = 0; l < g.levels.size(); l++) {
        const Level &L = g.levels[l];
        for (long long x : {(long long)L.n, (long long)L.nnz, (long long)L.agg, (long long)L.nsl, L.sell_len,
                            (long long)L.max_near, (long long)L.uni_w, (long long)(l > 0 ? L.max_row : 0)})
            v.push_back((uint64_t)x);
        arr(L.sl_off.p, 4 * ((size_t)L.nsl + 1));
        arr(L.sl_near.p, 4 * (size_t)L.nsl);
        arr(L.col.p, 4 * (size_t)L.sell_len);
        if (l > 0) {
            arr(L.crow.p, 4 * ((size_t)L.n + 1));
            arr(L.cptr.p, 4 * ((size_t)L.nnz + 1));
            const size_t nv = (size_t)last_int(L.cptr.p, (size_t)L.nnz);
            arr(L.cidx.p, 4 * nv);
            arr(L.cpos.p, 4 * (size_t)L.nnz);
        }
    }
    for (long long x : {g.l0_far_entries, (long long)g.asm_windowed, (long long)g.asm_l1_fused, (long long)g.l1_fused,
                        (long long)g.cg2, (long long)g.ndense, (long long)g.ndense_pad, (long long)g.dense_bw,
                        (long long)g.opt.mg_dense_max, (long long)(g.opt.mg_kc * 1000.0 + 0.5), (long long)g.additive_top})
        v.push_back((uint64_t)x);
    const int nout = (int)std::min<size_t>(v.size(), (size_t)cap);
    for (int i = 0; i < nout; i++) out[i] = v[(size_t)i];
    return nout;
    API_CATCH
}

int irotavg_graph_time_kernel(irotavg_graph *h, int which, int reps, double *ms) {
    if (!h || !ms || reps <= 0) return IROTAVG_ERR_BAD_ARG;
    API_TRY
    return time_kernel(h->g, which, reps, ms);
    API_CATCH
}

// ---------------------------------------------------------------------------------------------
// one-shot drop-ins
// ---------------------------------------------------------------------------------------------
}  // extern "C"

// The reference's callers hand the SAME edge list and relative rotations to l1ra and then to irls
// (src/ViewGraph.cpp:1400-1417, ral/test.cpp:295-301). Until round 5 each of the two one-shot calls built its own
// handle and uploaded the 80 MB again. Now the last handle is kept: the next one-shot call whose (I, QQ, m, n_total, f,
// ldqq) and CONTENT match takes it and uploads nothing but Q. The content is a 64-bit hash of every word of the edge
// list and of the m x 4 relative rotations as the caller holds them (all host cores, ~1 ms for 2M edges) -- an array
// that was changed in place between two calls, in a single entry, is a different graph (the pointers are only the
// cheap first test). IROTAVG_ONESHOT_CACHE=0 or irotavg_oneshot_cache(0) switch it off; irotavg_oneshot_cache_clear()
// gives the kept handle's memory back. Thread safety: the cache is one slot behind a mutex, a handle is taken OUT of
// the slot for the duration of a call (a second thread that calls meanwhile builds its own), the last one to finish
// stays. The kept handle is deliberately not destroyed at process exit (the HIP runtime may be gone by then).
namespace {
struct OneShotKey {
    const void *I = nullptr, *QQ = nullptr;
    int64_t m = 0, n_total = 0, ldqq = 0;
    int f = 0, dev = -1;
    uint64_t hI = 0, hQQ = 0;
    bool same(const OneShotKey &o) const {
        return I == o.I && QQ == o.QQ && m == o.m && n_total == o.n_total && ldqq == o.ldqq && f == o.f && dev == o.dev &&
               hI == o.hI && hQQ == o.hQQ;
    }
};
std::mutex g_os_mu;
irotavg_graph *g_os_handle = nullptr;
OneShotKey g_os_key;
std::atomic<int> g_os_enabled{-1};  // -1: ask the environment
std::atomic<int64_t> g_os_hits{0}, g_os_misses{0};

bool oneshot_cache_on() {
    int e = g_os_enabled.load();
    if (e < 0) {
        const char *v = std::getenv("IROTAVG_ONESHOT_CACHE");
        e = (v && std::atoi(v) == 0) ? 0 : 1;
        g_os_enabled.store(e);
    }
    return e != 0;
}

// 64-bit hash of n 8-byte words (chunks of 64K words hashed on all cores, combined in order)
uint64_t hash_words(const void *base, int64_t n) {
    // (the edge list is an int32 array: 4-byte aligned only -- words are fetched by memcpy, which compiles to one
    // unaligned load)
    struct Words {
        const unsigned char *b;
        uint64_t operator[](int64_t k) const {
            uint64_t v;
            std::memcpy(&v, b + 8 * k, 8);
            return v;
        }
    } p{static_cast<const unsigned char *>(base)};
    const int64_t chunk = 1 << 16;
    const int64_t nch = (n + chunk - 1) / chunk;
    std::vector<uint64_t> part((size_t)std::max<int64_t>(nch, 1), 0);
    parallel_for(nch, 1, [&](int64_t c0, int64_t c1, int) {
        for (int64_t c = c0; c < c1; c++) {
            const int64_t a = c * chunk, b = std::min(n, a + chunk);
            uint64_t h0 = 0x9E3779B97F4A7C15ull ^ (uint64_t)c, h1 = 0xC2B2AE3D27D4EB4Full, h2 = 0x165667B19E3779F9ull,
                     h3 = 0x27D4EB2F165667C5ull;
            int64_t k = a;
            for (; k + 4 <= b; k += 4) {  // four independent lanes: the multiplies pipeline
                h0 = (h0 ^ p[k]) * 0xFF51AFD7ED558CCDull;
                h1 = (h1 ^ p[k + 1]) * 0xC4CEB9FE1A85EC53ull;
                h2 = (h2 ^ p[k + 2]) * 0x9FB21C651E98DF25ull;
                h3 = (h3 ^ p[k + 3]) * 0xD6E8FEB86659FD93ull;
                h0 ^= h0 >> 29;
                h1 ^= h1 >> 31;
                h2 ^= h2 >> 30;
                h3 ^= h3 >> 28;
            }
            for (; k < b; k++) h0 = ((h0 ^ p[k]) * 0xFF51AFD7ED558CCDull) ^ (h0 >> 29);
            part[(size_t)c] = (h0 * 31 + h1) * 31 + (h2 * 31 + h3);
        }
    });
    uint64_t h = 0x2545F4914F6CDD1Dull ^ (uint64_t)n;
    for (int64_t c = 0; c < nch; c++) h = (h ^ part[(size_t)c]) * 0x9E3779B97F4A7C15ull + (h >> 27);
    return h;
}

OneShotKey oneshot_key(int64_t m, int64_t n_total, int f, const int32_t *I, const double *QQ, int64_t ldqq) {
    OneShotKey k;
    k.I = I;
    k.QQ = QQ;
    k.m = m;
    k.n_total = n_total;
    k.ldqq = ldqq;
    k.f = f;
    (void)hipGetDevice(&k.dev);
    k.hI = hash_words(I, m);  // an edge = two int32 = one word
    uint64_t h = 0;
    for (int c = 0; c < 4; c++)  // the four columns (ldqq may exceed m: the padding is not the graph)
        h = h * 0x100000001B3ull ^ hash_words(QQ + (size_t)c * (size_t)ldqq, m);
    k.hQQ = h;
    return k;
}

// the kept handle if it is this graph (taken out of the slot), else nullptr
irotavg_graph *oneshot_take(const OneShotKey &k) {
    std::lock_guard<std::mutex> lk(g_os_mu);
    if (g_os_handle && g_os_key.same(k)) {
        irotavg_graph *h = g_os_handle;
        g_os_handle = nullptr;
        g_os_hits++;
        return h;
    }
    g_os_misses++;
    return nullptr;
}
void oneshot_keep(irotavg_graph *h, const OneShotKey &k) {
    irotavg_graph *old = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_os_mu);
        old = g_os_handle;
        g_os_handle = h;
        g_os_key = k;
    }
    if (old) irotavg_graph_destroy(old);
}

// handle for a one-shot call: the kept one or a new one; `cached` says whether it goes back to the slot afterwards
int oneshot_open(irotavg_graph **h, OneShotKey *key, bool *cached, int64_t m, int64_t n_total, int f, const int32_t *I,
                 const double *QQ, int64_t ldqq, bool *hit = nullptr) {
    *cached = oneshot_cache_on() && I && QQ && m > 0 && ldqq >= m && irotavg_device_count() > 0;
    *h = nullptr;
    if (hit) *hit = false;
    if (*cached) {
        *key = oneshot_key(m, n_total, f, I, QQ, ldqq);
        *h = oneshot_take(*key);
        if (*h) {
            if (hit) *hit = true;
            return IROTAVG_OK;
        }
    }
    return irotavg_graph_create(h, m, n_total, f, I, QQ, ldqq, nullptr);
}
// A call that failed with a device or memory error on a KEPT handle (one that a device reset invalidated, or whose
// pinned buffers are what the new work could not allocate next to) would have succeeded before handles were kept: the
// kept one is destroyed, the pools are trimmed and the call gets ONE more attempt on a fresh handle (advisor, round 5).
bool oneshot_retry_fresh(irotavg_graph **h, bool hit, int rc, int64_t m, int64_t n_total, int f, const int32_t *I,
                         const double *QQ, int64_t ldqq) {
    if (!hit || (rc != IROTAVG_ERR_HIP && rc != IROTAVG_ERR_NOMEM)) return false;
    if (*h) irotavg_graph_destroy(*h);
    *h = nullptr;
    (void)irotavg_trim_memory();
    return irotavg_graph_create(h, m, n_total, f, I, QQ, ldqq, nullptr) == IROTAVG_OK;
}
// the handle of a failed one-shot call replaced by one that solves iteratively (band_direct = -1) -- when the failed one
// was a direct-solver handle with loop closures; false: nothing to retry with
bool oneshot_retry_iterative(irotavg_graph **h, int64_t m, int64_t n_total, int f, const int32_t *I, const double *QQ,
                             int64_t ldqq) {
    if (!*h || bcr_closures((*h)->g) == 0) return false;
    irotavg_options opt;
    irotavg_default_options(&opt);
    opt.band_direct = -1;
    irotavg_graph *it = nullptr;
    if (irotavg_graph_create(&it, m, n_total, f, I, QQ, ldqq, &opt) != IROTAVG_OK) return false;
    irotavg_graph_destroy(*h);
    *h = it;
    return true;
}
void oneshot_close(irotavg_graph *h, const OneShotKey &key, bool cached, int rc) {
    // (a handle whose call failed is not kept: whatever state it is in, the next call starts from a fresh one)
    if (cached && (rc == IROTAVG_OK || rc == IROTAVG_ERR_NOT_CONVERGED)) oneshot_keep(h, key);
    else irotavg_graph_destroy(h);
}
}  // namespace

extern "C" {

void irotavg_oneshot_cache(int enable) { g_os_enabled.store(enable ? 1 : 0); if (!enable) irotavg_oneshot_cache_clear(); }
void irotavg_oneshot_cache_clear(void) {
    irotavg_graph *old = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_os_mu);
        old = g_os_handle;
        g_os_handle = nullptr;
    }
    if (old) irotavg_graph_destroy(old);
}
void irotavg_oneshot_cache_stats(int64_t *hits, int64_t *misses) {
    if (hits) *hits = g_os_hits.load();
    if (misses) *misses = g_os_misses.load();
}

int irotavg_irls(int64_t m, int64_t n_total, int f, const int32_t *I, const double *QQ,
                 int64_t ldqq, int cost, double sigma, double *Q, int64_t ldq, int max_iters,
                 double change_th, double *weights, int *iters, double *runtime) {
    if (!Q || !weights || !iters || !runtime) return IROTAVG_ERR_BAD_ARG;
    if (cost < IROTAVG_L2 || cost > IROTAVG_WELSCH) return IROTAVG_ERR_UNKNOWN_COST;
    const bool timing = std::getenv("IROTAVG_BUILD_TIMING") != nullptr;
    double t0 = now_seconds();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const double t = now_seconds();
        std::fprintf(stderr, "[irotavg_hip one-shot] %-22s %8.2f ms\n", what, 1e3 * (t - t0));
        t0 = t;
    };
    irotavg_graph *h = nullptr;
    OneShotKey key;
    bool cached = false, hit = false;
    int rc = oneshot_open(&h, &key, &cached, m, n_total, f, I, QQ, ldqq, &hit);
    if (rc != IROTAVG_OK) return rc;
    lap("handle (kept or created)");
    for (int attempt = 0; attempt < 2; attempt++) {
        rc = irotavg_graph_set_rotations(h, Q, ldq);
        lap("set_rotations");
        if (rc == IROTAVG_OK)
            rc = irotavg_graph_irls(h, cost, sigma, max_iters, change_th, iters, runtime, nullptr);
        lap("irls");
        if (attempt == 1 || !oneshot_retry_fresh(&h, hit, rc, m, n_total, f, I, QQ, ldqq)) break;
    }
    if (!h) return rc;
    if (rc == IROTAVG_ERR_SOLVER && oneshot_retry_iterative(&h, m, n_total, f, I, QQ, ldqq)) {
        // the direct solver gave the graph up (closures on a band part that is next to singular, run_irls): once more
        // on the iterative solver, from the caller's rotations (nothing was written back) -- the reference has one
        // solver for every graph (ral/l1_irls.cpp:536-556)
        rc = irotavg_graph_set_rotations(h, Q, ldq);
        if (rc == IROTAVG_OK)
            rc = irotavg_graph_irls(h, cost, sigma, max_iters, change_th, iters, runtime, nullptr);
        lap("irls, iterative solver");
    }
    if (rc == IROTAVG_OK || rc == IROTAVG_ERR_NOT_CONVERGED) {
        (void)irotavg_graph_get_rotations(h, Q, ldq);
        lap("get_rotations");
        (void)irotavg_graph_get_weights(h, weights);
        lap("get_weights");
    }
    oneshot_close(h, key, cached, rc);
    lap("keep / destroy");
    return rc;
}

int irotavg_l1ra(int64_t m, int64_t n_total, int f, const int32_t *I, const double *QQ,
                 int64_t ldqq, double *Q, int64_t ldq, int max_iters, double change_th, int *iter,
                 double *runtime) {
    if (!Q || !iter || !runtime) return IROTAVG_ERR_BAD_ARG;
    irotavg_graph *h = nullptr;
    OneShotKey key;
    bool cached = false, hit = false;
    int rc = oneshot_open(&h, &key, &cached, m, n_total, f, I, QQ, ldqq, &hit);
    if (rc != IROTAVG_OK) return rc;
    for (int attempt = 0; attempt < 2; attempt++) {
        rc = irotavg_graph_set_rotations(h, Q, ldq);
        if (rc == IROTAVG_OK) rc = irotavg_graph_l1ra(h, max_iters, change_th, iter, runtime, nullptr);
        if (attempt == 1 || !oneshot_retry_fresh(&h, hit, rc, m, n_total, f, I, QQ, ldqq)) break;
    }
    if (!h) return rc;
    if (rc == IROTAVG_OK || rc == IROTAVG_ERR_NOT_CONVERGED) (void)irotavg_graph_get_rotations(h, Q, ldq);
    oneshot_close(h, key, cached, rc);
    return rc;
}

// replaces irotavg::quat_normalised for host-resident rows: Eigen normalized() per row
// (ral/l1_irls.cpp:982-991), evaluated by the same device kernel as the resident variant.
int irotavg_quat_normalised(int64_t n, double *Q, int64_t ldq, int f) {
    if (!Q || n < 0 || ldq < n || f < 0 || n > 0x7fffffffLL) return IROTAVG_ERR_BAD_ARG;
    if (irotavg_device_count() <= 0) return IROTAVG_ERR_NO_DEVICE;
    API_TRY
    return normalise_host_rows(n, Q, ldq, f);
    API_CATCH
}

}  // extern "C"

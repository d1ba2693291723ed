// viewgraph.cpp -- OpenCV-free counterpart of the rotation side of the reference's ViewGraph /
// Pose API (src/ViewGraph.hpp:54-75, src/Pose.hpp:35-59): views hold a row-major 3x3 rotation and
// a fixed flag, connections hold the relative rotation R_ij (R_j = R_ij R_i, stored once with
// i < j), and rot_avg(win_size) reproduces ViewGraph::rotAvg (src/ViewGraph.cpp:1263-1435):
// window extraction, fixed/free relabelling, R -> quaternion, l1ra + irls on the GPU core,
// quaternion -> R write-back. The vision front-end that produces the edges is out of scope.
//
// One deliberate difference: the reference walks `std::map<View*, ViewConnection*>`, i.e. in
// pointer order (run-to-run non-deterministic edge order, src/View.hpp:66); here a view's
// connections are walked by ascending neighbour id.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "graph.hpp"

namespace {

struct Mat3 {
    double m[9];
};

// src/ViewGraph.cpp:1175-1203, row-major R -> [x y z w]
void rmat2quat(const double *R, double q[4]) {
    auto at = [&](int r, int c) { return R[3 * r + c]; };
    const double trace = at(0, 0) + at(1, 1) + at(2, 2);
    if (trace > 0.0) {
        double s = std::sqrt(trace + 1.0);
        q[3] = s * 0.5;
        s = 0.5 / s;
        q[0] = (at(2, 1) - at(1, 2)) * s;
        q[1] = (at(0, 2) - at(2, 0)) * s;
        q[2] = (at(1, 0) - at(0, 1)) * s;
    } else {
        const int i = at(0, 0) < at(1, 1) ? (at(1, 1) < at(2, 2) ? 2 : 1) : (at(0, 0) < at(2, 2) ? 2 : 0);
        const int j = (i + 1) % 3, k = (i + 2) % 3;
        double s = std::sqrt(at(i, i) - at(j, j) - at(k, k) + 1.0);
        q[i] = s * 0.5;
        s = 0.5 / s;
        q[3] = (at(k, j) - at(j, k)) * s;
        q[j] = (at(j, i) + at(i, j)) * s;
        q[k] = (at(k, i) + at(i, k)) * s;
    }
}

// src/ViewGraph.cpp:1426-1433: q.normalized().toRotationMatrix(), row-major
void quat2rmat(const double qin[4], double *R) {
    double x = qin[0], y = qin[1], z = qin[2], w = qin[3];
    const double n2 = x * x + y * y + z * z + w * w;
    if (n2 > 0.0) {
        const double nn = std::sqrt(n2);
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

}  // namespace

struct irotavg_viewgraph {
    std::vector<Mat3> pose;                      // absolute rotation per view (Pose::R)
    std::vector<char> fixed;                     // m_fixed_mask
    std::vector<int> mark;                       // rot_avg scratch: view id -> row, -1 outside a call
    // rot_avg's work arrays, kept between calls: a global re-solve at 75k views spent 5 ms returning 25 MB to
    // the OS on exit and as much again faulting the pages back in on the next call
    struct Scratch {
        std::vector<int32_t> I;
        std::vector<double> qq, Q, QQ, Qa;
        std::vector<int> vertices, i2v;
        std::vector<long> off;
    } scratch;
    // per view j: its connections to LOWER ids i < j (the only direction rot_avg walks, :1290-1300),
    // ascending i, with R_ij and its quaternion (converted once, at connect time)
    struct Conn {
        int i;
        Mat3 R;
        double q[4];
    };
    std::vector<std::vector<Conn>> conn;
    irotavg_options opt;
    irotavg_rotavg_info last{};
    irh::WindowSolver *win = nullptr;  // persistent staging of the single-kernel window solve
    // The device-resident growing copy of the graph that the global re-solves run on (resident.hip) and what the
    // host has changed since the device last saw it: poses / fixed flags of views >= res_pose_lo, edge records of
    // views >= res_edge_view (a connection is filed under its HIGHER view); eoff[t] = edges of the views below t,
    // valid up to res_edge_view. Counters that decide rot_avg's early-outs without walking the graph.
    irh::Resident *res = nullptr;
    long res_pose_lo = 0, res_edge_view = 0;
    std::vector<long> eoff;
    std::vector<char> touched;  // view has at least one connection
    long n_touched = 0, n_fixed = 0, n_conn = 0;
    void pose_changed(long idx) { res_pose_lo = std::min(res_pose_lo, idx); }
    // A global re-solve on the resident graph leaves the QUATERNION of every free view here and marks the view: its
    // rotation matrix (src/ViewGraph.cpp:1420-1434: q.normalized().toRotationMatrix()) is formed when somebody reads the
    // pose -- get_pose, a window that holds the view, save_poses, the next delta for the device. Converting all 75k
    // views at the end of the call was a third of it (round 4: 1.5 of 4.8 ms) for matrices that mostly nobody reads
    // before the next global re-solve replaces them.
    mutable std::vector<double> qlazy;  // 4 per view
    mutable std::vector<char> lazy;
    const double *pose_m(size_t x) const {  // (callers: the non-const paths of rot_avg; one thread per view)
        if (x < lazy.size() && lazy[x]) {
            quat2rmat(&qlazy[4 * x], const_cast<double *>(pose[x].m));
            lazy[x] = 0;
        }
        return pose[x].m;
    }
    // the same for the readers that hold a CONST handle (get_pose, save_poses): the matrix is formed into the caller's
    // buffer and nothing of the handle is written, so concurrent readers of one view-graph do not race (advisor, round 5)
    void pose_read(size_t x, double *out) const {
        if (x < lazy.size() && lazy[x])
            quat2rmat(&qlazy[4 * x], out);
        else
            std::copy(pose[x].m, pose[x].m + 9, out);
    }
    double *pose_w(size_t x) {  // the pose is about to be overwritten
        if (x < lazy.size()) lazy[x] = 0;
        return pose[x].m;
    }
    // a window extracted by rot_avg whose solve was deferred to a batched launch (irotavg_viewgraph_rot_avg_batch)
    struct Pending {
        bool on = false;
        int f = 0;
        long nv = 0, ne = 0;
        irotavg_rotavg_info loc{};
    } pending;
    ~irotavg_viewgraph() {
        if (win) irh::window_solver_delete(win);
        if (res) irh::resident_delete(res);
    }
};

namespace {
constexpr int kRotAvgDeferred = 1000;      // internal return code of rot_avg: window packed, solve deferred
thread_local bool tl_defer_windows = false;  // set by irotavg_viewgraph_rot_avg_batch around its extraction calls

// write-back for k >= f (src/ViewGraph.cpp:1420-1434): q.normalized().toRotationMatrix() into the views' poses
void rotavg_writeback(irotavg_viewgraph *vg, int f, long nv) {
    const std::vector<double> &Q = vg->scratch.Q;
    const std::vector<int> &i2v = vg->scratch.i2v;
    irh::parallel_for((int64_t)(nv - f), 4096, [&](int64_t a, int64_t b, int) {
        for (int64_t r = f + a; r < f + b; r++) {
            const double q[4] = {Q[(size_t)r], Q[(size_t)(nv + r)], Q[(size_t)(2 * nv + r)], Q[(size_t)(3 * nv + r)]};
            quat2rmat(q, vg->pose_w((size_t)i2v[(size_t)r]));
        }
    });
    // free views are relabelled in ascending id: the lowest one is row f
    if (nv > f) vg->pose_changed(i2v[(size_t)f]);
}
}  // namespace

extern "C" {

int irotavg_viewgraph_create(irotavg_viewgraph **vg, const irotavg_options *opt) {
    if (!vg) return IROTAVG_ERR_BAD_ARG;
    try {
        *vg = new irotavg_viewgraph();
        if (opt)
            (*vg)->opt = *opt;
        else
            irotavg_default_options(&(*vg)->opt);
    } catch (...) {
        return IROTAVG_ERR_NOMEM;
    }
    return IROTAVG_OK;
}

void irotavg_viewgraph_destroy(irotavg_viewgraph *vg) { delete vg; }

int irotavg_viewgraph_add_view(irotavg_viewgraph *vg, const double R[9]) {
    if (!vg) return IROTAVG_ERR_BAD_ARG;
    Mat3 M;
    if (R) {
        std::copy(R, R + 9, M.m);
    } else {  // Pose() default: identity (src/Pose.hpp)
        const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        std::copy(I, I + 9, M.m);
    }
    vg->pose.push_back(M);
    vg->fixed.push_back(0);
    vg->mark.push_back(-1);
    vg->conn.emplace_back();
    vg->touched.push_back(0);
    return (int)vg->pose.size() - 1;
}

int irotavg_viewgraph_num_views(const irotavg_viewgraph *vg) { return vg ? (int)vg->pose.size() : 0; }

// View::connect (src/ViewGraph.cpp:1438-1455): undirected, one ViewConnection per pair, a second
// connect of the same pair is refused. Rij relates a to b: R_b = R_ab R_a. The pair is stored under
// (lo, hi) with R_hi = R R_lo, so a call with a > b stores the transpose (the reference only ever
// calls connect(prev, curr) with prev < curr).
int irotavg_viewgraph_connect(irotavg_viewgraph *vg, int a, int b, const double Rij[9]) {
    if (!vg || !Rij || a == b || a < 0 || b < 0 || a >= (int)vg->pose.size() || b >= (int)vg->pose.size())
        return IROTAVG_ERR_BAD_ARG;
    const int lo = std::min(a, b), hi = std::max(a, b);
    auto &list = vg->conn[hi];
    auto it = std::lower_bound(list.begin(), list.end(), lo,
                               [](const irotavg_viewgraph::Conn &c, int key) { return c.i < key; });
    if (it != list.end() && it->i == lo) return 0;
    irotavg_viewgraph::Conn c;
    c.i = lo;
    if (a < b) {
        std::copy(Rij, Rij + 9, c.R.m);
    } else {  // R_ab given with a > b: R_lo->hi = R_ab^T
        for (int r = 0; r < 3; r++)
            for (int q = 0; q < 3; q++) c.R.m[3 * r + q] = Rij[3 * q + r];
    }
    rmat2quat(c.R.m, c.q);
    list.insert(it, c);
    vg->n_conn++;
    for (int x : {lo, hi})
        if (!vg->touched[(size_t)x]) {
            vg->touched[(size_t)x] = 1;
            vg->n_touched++;
        }
    vg->res_edge_view = std::min<long>(vg->res_edge_view, hi);
    return 1;
}

// ViewGraph::fixPose / isPoseFixed / countFixedPoses (src/ViewGraph.cpp:1234-1260)
int irotavg_viewgraph_fix_pose(irotavg_viewgraph *vg, int idx, const double R[9]) {
    if (!vg || !R || idx < 0 || idx >= (int)vg->pose.size()) return IROTAVG_ERR_BAD_ARG;
    if (!vg->fixed[idx]) vg->n_fixed++;
    vg->fixed[idx] = 1;
    std::copy(R, R + 9, vg->pose_w((size_t)idx));
    vg->pose_changed(idx);
    return IROTAVG_OK;
}
int irotavg_viewgraph_is_pose_fixed(const irotavg_viewgraph *vg, int idx) {
    if (!vg || idx < 0 || idx >= (int)vg->pose.size()) return IROTAVG_ERR_BAD_ARG;
    return vg->fixed[idx] ? 1 : 0;
}
int irotavg_viewgraph_count_fixed_poses(const irotavg_viewgraph *vg) {
    if (!vg) return IROTAVG_ERR_BAD_ARG;
    int c = 0;
    for (char f : vg->fixed) c += f ? 1 : 0;
    return c;
}
int irotavg_viewgraph_get_pose(const irotavg_viewgraph *vg, int idx, double R[9]) {
    if (!vg || !R || idx < 0 || idx >= (int)vg->pose.size()) return IROTAVG_ERR_BAD_ARG;
    vg->pose_read((size_t)idx, R);
    return IROTAVG_OK;
}
int irotavg_viewgraph_set_pose(irotavg_viewgraph *vg, int idx, const double R[9]) {
    if (!vg || !R || idx < 0 || idx >= (int)vg->pose.size()) return IROTAVG_ERR_BAD_ARG;
    std::copy(R, R + 9, vg->pose_w((size_t)idx));
    vg->pose_changed(idx);
    return IROTAVG_OK;
}

namespace {
// A GLOBAL re-solve (the window holds every view: rotAvg(5000000) after a loop closure, src/IRotAvg.cpp:371-378) on
// the device-resident copy of the graph (resident.hip). With every view in the window the quantities of
// src/ViewGraph.cpp:1282-1363 are counters: the edges are all connections, the vertices all views that have one
// (fewer than the window -> early-out 3, so a call that goes on has ALL views), f = the fixed views, relabelled
// fixed-first in ascending id. Returns false when the call is not of that kind (the caller takes the general path:
// windows, graphs of fewer than 20000 connections, f == 0, no device); true with *rc set otherwise.
bool rotavg_resident(irotavg_viewgraph *vg, irotavg_rotavg_info &loc, bool timing, int *rc, bool dry = false) {
    const long m = (long)vg->pose.size();
    if (std::getenv("IROTAVG_NO_RESIDENT")) return false;
    long min_edges = 20000;  // below: the host build of the general path (build.cpp) is the faster one
    if (const char *e = std::getenv("IROTAVG_RESIDENT_MIN_EDGES")) min_edges = std::atol(e);
    if (vg->n_conn < min_edges || vg->n_conn < m || vg->n_touched < m || vg->n_fixed < 1 || m - vg->n_fixed < 1) return false;
    if (vg->n_conn > 0x3fffffffL) return false;
    if (irotavg_device_count() <= 0) return false;
    const double t0 = irh::now_seconds();
    try {
        if (!vg->res) vg->res = irh::resident_new();
        irh::Resident &R = *vg->res;
        // what the device holds is valid below these marks
        long view_lo = std::min({vg->res_pose_lo, irh::resident_views(R), m});
        long ev = std::min({vg->res_edge_view, m});
        if (irh::resident_views(R) == 0 || irh::resident_edges(R) == 0) {
            view_lo = 0;
            ev = 0;
        }
        std::vector<long> &eoff = vg->eoff;
        if ((long)eoff.size() < m + 1) eoff.resize((size_t)m + 1 + (size_t)m / 2, 0);
        for (long t = ev; t < m; t++) eoff[(size_t)t + 1] = eoff[(size_t)t] + (long)vg->conn[(size_t)t].size();
        const long ne = eoff[(size_t)m], edge_lo = eoff[(size_t)ev];
        if (edge_lo > irh::resident_edges(R)) {  // (cannot happen: the marks only move down between calls)
            irh::resident_invalidate(R);
            vg->res_pose_lo = 0;
            vg->res_edge_view = 0;
            return false;
        }
        irh::ResidentStage st = irh::resident_stage(R, m, view_lo, ne, edge_lo);
        irh::parallel_for((int64_t)(m - ev), 2048, [&](int64_t a, int64_t b, int) {
            for (int64_t t = ev + a; t < ev + b; t++) {
                size_t e = (size_t)(eoff[(size_t)t] - edge_lo);
                for (const auto &c : vg->conn[(size_t)t]) {
                    st.I[2 * e] = c.i;
                    st.I[2 * e + 1] = (int32_t)t;
                    for (int q = 0; q < 4; q++) st.qq[4 * e + q] = c.q[q];
                    e++;
                }
            }
        });
        irh::parallel_for((int64_t)(m - view_lo), 4096, [&](int64_t a, int64_t b, int) {
            for (int64_t x = view_lo + a; x < view_lo + b; x++) {
                const double *P = vg->pose_m((size_t)x);
                std::copy(P, P + 9, st.R + 9 * (size_t)(x - view_lo));
                st.fixed[(size_t)(x - view_lo)] = (uint8_t)(vg->fixed[(size_t)x] ? 1 : 0);
            }
        });
        if (timing) std::fprintf(stderr, "[rot_avg resident] %-24s %8.3f ms (%ld views, %ld edges sent)\n", "delta packing",
                                 1e3 * (irh::now_seconds() - t0), m - view_lo, ne - edge_lo);
        const int f = (int)vg->n_fixed;
        // the marks move up BEFORE the call: whatever fails inside invalidates the resident copy as a whole
        vg->res_pose_lo = m;
        vg->res_edge_view = m;
        int dry_a = -1, dry_b = -1;
        if (dry) {  // two free views far apart for the dry run's made-up closure
            for (long x = 0; x < m && dry_a < 0; x++)
                if (!vg->fixed[(size_t)x]) dry_a = (int)x;
            for (long x = m - 1; x > dry_a + 1000 && dry_b < 0; x--)
                if (!vg->fixed[(size_t)x]) dry_b = (int)x;
        }
        *rc = irh::resident_rot_avg(R, m, view_lo, ne, edge_lo, f, vg->opt, loc, timing, dry, dry_a, dry_b);
        loc.n_views = (int)m;
        loc.n_edges = (int)ne;
        loc.n_fixed = f;
        if (*rc != IROTAVG_OK) {
            vg->res_pose_lo = 0;
            vg->res_edge_view = 0;
            irh::resident_invalidate(R);
            // out of device memory (the resident copy and the enlarged blocks of its handles cost more than the general
            // path's one-off handle) or a runtime error inside this path: the copy is dropped, the cached blocks go back
            // to the driver and the call takes the general path, which may well succeed -- it reports on its own if not
            // (... or the direct solver gave the graph up: the general path repeats that and then solves iteratively)
            if (*rc == IROTAVG_ERR_NOMEM || *rc == IROTAVG_ERR_HIP || *rc == IROTAVG_ERR_SOLVER) {
                (void)hipGetLastError();
                (void)irotavg_trim_memory();
                *rc = IROTAVG_OK;
                return false;
            }
            return true;
        }
        if (dry) return true;
        const double t1 = irh::now_seconds();
        // the quaternions are kept, the matrices are formed on demand (pose_m)
        if ((long)vg->lazy.size() < m) {
            vg->lazy.resize((size_t)m + (size_t)m / 2, 0);
            vg->qlazy.resize(4 * vg->lazy.size());
        }
        std::memcpy(vg->qlazy.data(), st.Q, sizeof(double) * 4 * (size_t)m);
        for (long x = 0; x < m; x++) vg->lazy[(size_t)x] = vg->fixed[(size_t)x] ? 0 : 1;
        if (timing) std::fprintf(stderr, "[rot_avg resident] %-24s %8.3f ms\n", "poses (host)", 1e3 * (irh::now_seconds() - t1));
        return true;
    } catch (...) {
        if (vg->res) irh::resident_invalidate(*vg->res);
        vg->res_pose_lo = 0;
        vg->res_edge_view = 0;
        (void)hipGetLastError();
        (void)irotavg_trim_memory();
        *rc = IROTAVG_OK;
        return false;  // (the general path: see above)
    }
}
}  // namespace

// ViewGraph::rotAvg(winSize), src/ViewGraph.cpp:1263-1435. Returns IROTAVG_OK also for the
// reference's silent early-outs (info->skipped tells which).
int irotavg_viewgraph_rot_avg(irotavg_viewgraph *vg, int win_size, irotavg_rotavg_info *info) {
    if (!vg || win_size <= 2) return IROTAVG_ERR_BAD_ARG;  // assert(winSize > 2) :1265
    irotavg_rotavg_info loc{};
    const bool timing = std::getenv("IROTAVG_ROTAVG_TIMING") != nullptr;
    struct Total {
        bool on;
        double t0;
        ~Total() {
            if (on) std::fprintf(stderr, "[rot_avg] %-28s %8.3f ms\n", "total (incl. clean-up)", 1e3 * (irh::now_seconds() - t0));
        }
    } total{timing, irh::now_seconds()};
    double tl = irh::now_seconds();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const double t = irh::now_seconds();
        std::fprintf(stderr, "[rot_avg] %-28s %8.3f ms\n", what, 1e3 * (t - tl));
        tl = t;
    };
    const long m = (long)vg->pose.size();
    int win = (int)std::min<long>(m, win_size);  // :1269
    if (win < 2) {
        loc.skipped = 1;
        if (info) *info = loc;
        return IROTAVG_OK;  // :1270-1273
    }
    if (win == m) {  // a global re-solve: on the device-resident copy of the graph when it is of that kind
        int rrc = IROTAVG_OK;
        if (rotavg_resident(vg, loc, timing, &rrc)) {
            if (rrc == IROTAVG_OK) vg->last = loc;
            if (info) *info = loc;
            return rrc;
        }
    }
    // ---- local connections (:1282-1307): for the last `win` views, edges with i < j.
    // `vertices` of the reference is a std::set<int> (ascending ids); here: a mark array + sort.
    std::vector<int32_t> &I = vg->scratch.I;
    std::vector<double> &qq = vg->scratch.qq;  // per edge [x y z w]
    std::vector<int> &vertices = vg->scratch.vertices;
    I.clear();
    qq.clear();
    vertices.clear();
    std::vector<int> &v2i = vg->mark;  // -1 unseen, -2 seen, >= 0 row in Q after relabelling
    struct Unmark {  // the scratch map is persistent (O(window) work per call): restore on exit
        std::vector<int> &map;
        std::vector<int> &touched;
        ~Unmark() {
            for (int x : touched) map[x] = -1;
        }
    } unmark{v2i, vertices};
    auto touch = [&](int x) {
        if (v2i[x] == -1) {
            v2i[x] = -2;
            vertices.push_back(x);
        }
    };
    // edge k of the call = the k-th connection in (view ascending, lower endpoint ascending) order: offsets
    // first, then the views fill their runs side by side (a global re-solve walks 300k connections)
    std::vector<long> &off = vg->scratch.off;
    off.assign((size_t)win + 1, 0);
    for (long t = 0; t < win; t++) off[(size_t)t + 1] = off[(size_t)t] + (long)vg->conn[(size_t)(m - win + t)].size();
    // a stream's global re-solves grow from call to call: half as much again, so that most calls find room
    auto room = [](auto &v, size_t need) {
        if (v.capacity() < need) v.reserve(need + need / 2);
    };
    room(I, (size_t)2 * off[(size_t)win]);
    room(qq, (size_t)4 * off[(size_t)win]);
    room(vg->scratch.QQ, (size_t)4 * off[(size_t)win]);
    room(vertices, (size_t)std::min<long>(m, 2 * off[(size_t)win]));
    room(vg->scratch.i2v, (size_t)std::min<long>(m, 2 * off[(size_t)win]));
    room(vg->scratch.Q, (size_t)4 * std::min<long>(m, 2 * off[(size_t)win]));
    I.resize((size_t)2 * off[(size_t)win]);
    qq.resize((size_t)4 * off[(size_t)win]);
    irh::parallel_for((int64_t)win, 4096, [&](int64_t a, int64_t b, int) {
        for (int64_t t = a; t < b; t++) {
            const int j = (int)(m - win + t);  // frame id == view index (src/IRotAvg.cpp:280-284)
            size_t e = (size_t)off[(size_t)t];
            for (const auto &c : vg->conn[(size_t)j]) {
                I[2 * e] = c.i;
                I[2 * e + 1] = j;
                for (int q = 0; q < 4; q++) qq[4 * e + q] = c.q[q];
                e++;
            }
        }
    });
    for (long t = m - win; t < m; t++)
        for (const auto &c : vg->conn[(size_t)t]) {
            touch(c.i);
            touch((int)t);
        }
    std::sort(vertices.begin(), vertices.end());
    const long ne = (long)qq.size() / 4, nv = (long)vertices.size();
    if (ne < win) {  // :1313-1316
        loc.skipped = 2;
        if (info) *info = loc;
        return IROTAVG_OK;
    }
    if (nv < win) {  // :1318-1321
        loc.skipped = 3;
        if (info) *info = loc;
        return IROTAVG_OK;
    }
    // ---- fixed count and relabelling (:1323-1363)
    int f = (int)nv - win;
    for (int x : vertices)
        if (x >= m - win && vg->fixed[x]) f++;
    std::vector<int> &i2v = vg->scratch.i2v;
    i2v.assign((size_t)nv, 0);
    int t = 0, k = f;
    for (int x : vertices) {
        if (x >= m - win && !vg->fixed[x]) {
            i2v[k] = x;
            v2i[x] = k++;
        } else {
            i2v[t] = x;
            v2i[x] = t++;
        }
    }
    irh::parallel_for((int64_t)I.size(), 65536, [&](int64_t a, int64_t b, int) {
        for (int64_t q = a; q < b; q++) I[(size_t)q] = v2i[I[(size_t)q]];
    });
    // ---- Q (:1365-1386)
    std::vector<double> &Q = vg->scratch.Q;
    Q.resize((size_t)4 * nv);  // every entry is written below
    irh::parallel_for((int64_t)nv, 4096, [&](int64_t a, int64_t b, int) {
        for (int64_t p = a; p < b; p++) {
            const int x = vertices[(size_t)p];
            double q[4];
            rmat2quat(vg->pose_m((size_t)x), q);
            const int r = v2i[x];
            for (int c = 0; c < 4; c++) Q[(size_t)c * nv + r] = q[c];
        }
    });
    if (f == 0) {  // :1382-1386
        Q[0] = 0;
        Q[nv] = 0;
        Q[2 * nv] = 0;
        Q[3 * nv] = 1;
        f = 1;
    }
    // make_A asserts n - f > 1 (ral/l1_irls.cpp:758); fewer unknowns cannot be solved
    if (nv - f < 1) {
        loc.skipped = 4;
        if (info) *info = loc;
        return IROTAVG_OK;
    }
    std::vector<double> &QQ = vg->scratch.QQ;
    QQ.resize((size_t)4 * ne);
    irh::parallel_for((int64_t)ne, 32768, [&](int64_t a, int64_t b, int) {
        for (int64_t e = a; e < b; e++)
            for (int c = 0; c < 4; c++) QQ[(size_t)c * ne + e] = qq[(size_t)4 * e + c];
    });
    lap("window extraction + packing");
    // ---- solve (:1396-1417): no init_mst (refine from the current poses); l1ra 100 iterations,
    // then irls Geman-McClure, sigma 5 deg, 100 iterations, change_th 1e-3
    const double change_th = .001;
    int rc = IROTAVG_OK;
    if (vg->opt.no_window_kernel != 1 && irh::window_fits((int)nv, f, (int)ne)) {
        // small (sliding-window) problem: the whole l1ra + irls pipeline in ONE kernel launch
        if (irotavg_device_count() <= 0) return IROTAVG_ERR_NO_DEVICE;
        try {
            if (!vg->win) vg->win = irh::window_solver_new();
            std::vector<double> &Qa = vg->scratch.Qa;
            Qa.resize((size_t)4 * nv);
            for (long r = 0; r < nv; r++)
                for (int c = 0; c < 4; c++) Qa[(size_t)4 * r + c] = Q[(size_t)c * nv + r];
            if (tl_defer_windows && irh::window_fits_wave((int)nv, f, (int)ne)) {
                // one of several independent windows: the caller solves them in ONE launch and finishes this call
                loc.n_views = (int)nv;
                loc.n_edges = (int)ne;
                loc.n_fixed = f;
                vg->pending.on = true;
                vg->pending.f = f;
                vg->pending.nv = nv;
                vg->pending.ne = ne;
                vg->pending.loc = loc;
                return kRotAvgDeferred;
            }
            const double t0 = irh::now_seconds();
            rc = irh::window_solve(*vg->win, (int)nv, f, (int)ne, I.data(), qq.data(), Qa.data(), nullptr,
                                   100, 100, IROTAVG_GEMAN_MCCLURE, 5 * M_PI / 180.0, change_th,
                                   &loc.l1_iters, &loc.irls_iters);
            loc.irls_runtime = irh::now_seconds() - t0;  // both stages run in the one launch
            for (long r = 0; r < nv; r++)
                for (int c = 0; c < 4; c++) Q[(size_t)c * nv + r] = Qa[(size_t)4 * r + c];
        } catch (...) {
            return IROTAVG_ERR_HIP;
        }
    } else {
        irotavg_graph *g = nullptr;
        rc = irotavg_graph_create(&g, ne, nv, f, I.data(), QQ.data(), ne, &vg->opt);
        if (rc != IROTAVG_OK) return rc;
        rc = irotavg_graph_set_rotations(g, Q.data(), nv);
        lap("graph create + upload");
        if (rc == IROTAVG_OK)
            rc = irotavg_graph_l1ra(g, 100, change_th, &loc.l1_iters, &loc.l1_runtime, nullptr);
        lap("l1ra");
        if (rc == IROTAVG_OK)
            rc = irotavg_graph_irls(g, IROTAVG_GEMAN_MCCLURE, 5 * M_PI / 180.0, 100, change_th,
                                    &loc.irls_iters, &loc.irls_runtime, nullptr);
        lap("irls");
        if (rc == IROTAVG_ERR_SOLVER && vg->opt.band_direct >= 0) {
            // the direct solver gave the graph up (loop closures on a band part that is next to singular, run_irls): the
            // iterative solver takes every graph, like the reference's one code path (ral/l1_irls.cpp:536-556)
            irotavg_graph_destroy(g);
            g = nullptr;
            irotavg_options it = vg->opt;
            it.band_direct = -1;
            rc = irotavg_graph_create(&g, ne, nv, f, I.data(), QQ.data(), ne, &it);
            if (rc != IROTAVG_OK) return rc;
            rc = irotavg_graph_set_rotations(g, Q.data(), nv);
            if (rc == IROTAVG_OK) rc = irotavg_graph_l1ra(g, 100, change_th, &loc.l1_iters, &loc.l1_runtime, nullptr);
            if (rc == IROTAVG_OK)
                rc = irotavg_graph_irls(g, IROTAVG_GEMAN_MCCLURE, 5 * M_PI / 180.0, 100, change_th, &loc.irls_iters,
                                        &loc.irls_runtime, nullptr);
            lap("l1ra + irls, iterative solver");
        }
        if (rc == IROTAVG_OK) rc = irotavg_graph_get_rotations(g, Q.data(), nv);
        lap("download");
        irotavg_graph_destroy(g);
        lap("destroy");
    }
    loc.n_views = (int)nv;
    loc.n_edges = (int)ne;
    loc.n_fixed = f;
    if (rc != IROTAVG_OK) {
        if (info) *info = loc;
        return rc;
    }
    // ---- write-back for k >= f (:1420-1434)
    rotavg_writeback(vg, f, nv);
    lap("write-back");
    vg->last = loc;
    if (info) *info = loc;
    return IROTAVG_OK;
}

// Takes what a process pays ONCE for its first global re-solve -- the resident copy of the graph on the device,
// the device allocations of a solver handle of this size, kernel code loads, the streams and pinned blocks of
// l1ra's three solver chains -- out of that call's latency: the pipeline of a global rotAvg runs on the graph as it
// is and its result is dropped (no pose changes). For callers that load a graph and then stream onto it
// (BASELINE.json config 5); never needed for correctness. Returns IROTAVG_OK also when there was nothing to do.
int irotavg_viewgraph_prepare(irotavg_viewgraph *vg) {
    if (!vg) return IROTAVG_ERR_BAD_ARG;
    irotavg_rotavg_info loc{};
    int rc = IROTAVG_OK;
    const bool timing = std::getenv("IROTAVG_ROTAVG_TIMING") != nullptr;
    if (!rotavg_resident(vg, loc, timing, &rc, true)) return IROTAVG_OK;
    return rc;
}

// rotAvg for SEVERAL independent view-graphs at once (a server tracking many sequences): each graph's window is
// extracted as irotavg_viewgraph_rot_avg does, the windows that fit the wave-resident kernel (every rotAvg(10) of
// a sequence linked to <= 4 predecessors) are solved by ONE launch with a workgroup per window -- a single window
// keeps one of the 256 compute units busy --, the others (global re-solves) run one by one, then every graph's poses
// are written back. Results are those of n separate calls. infos: n entries or NULL. Returns the first error.
int irotavg_viewgraph_rot_avg_batch(irotavg_viewgraph *const *vgs, int n, int win_size, irotavg_rotavg_info *infos) {
    if (!vgs || n < 0) return IROTAVG_ERR_BAD_ARG;
    for (int b = 0; b < n; b++)
        if (!vgs[b]) return IROTAVG_ERR_BAD_ARG;
    for (int a = 0; a < n; a++)
        for (int b = a + 1; b < n; b++)
            if (vgs[a] == vgs[b]) return IROTAVG_ERR_BAD_ARG;  // windows of one graph depend on each other
    int first_err = IROTAVG_OK;
    std::vector<irh::WinBatchItem> items;
    std::vector<int> owner;
    struct Undefer {
        ~Undefer() { tl_defer_windows = false; }
    } undefer;
    tl_defer_windows = true;
    for (int b = 0; b < n; b++) {
        irotavg_viewgraph *vg = vgs[b];
        vg->pending.on = false;
        irotavg_rotavg_info loc{};
        const int rc = irotavg_viewgraph_rot_avg(vg, win_size, &loc);
        if (rc == kRotAvgDeferred) {
            irh::WinBatchItem it{};
            it.nv = (int)vg->pending.nv;
            it.f = vg->pending.f;
            it.ne = (int)vg->pending.ne;
            it.I = vg->scratch.I.data();
            it.QQ_aos = vg->scratch.qq.data();
            it.Q_aos = vg->scratch.Qa.data();
            items.push_back(it);
            owner.push_back(b);
        } else {
            if (infos) infos[b] = loc;
            if (rc != IROTAVG_OK && first_err == IROTAVG_OK) first_err = rc;
        }
    }
    tl_defer_windows = false;
    if (items.empty()) return first_err;
    if (irotavg_device_count() <= 0) return IROTAVG_ERR_NO_DEVICE;
    int rc = IROTAVG_OK;
    try {
        irotavg_viewgraph *host = vgs[owner[0]];
        if (!host->win) host->win = irh::window_solver_new();
        const double t0 = irh::now_seconds();
        rc = irh::window_solve_batch(*host->win, (int)items.size(), items.data(), 100, 100, IROTAVG_GEMAN_MCCLURE,
                                     5 * M_PI / 180.0, .001);
        const double dt = irh::now_seconds() - t0;
        for (size_t k = 0; k < items.size(); k++) {
            irotavg_viewgraph *vg = vgs[owner[k]];
            irotavg_rotavg_info loc = vg->pending.loc;
            vg->pending.on = false;
            loc.l1_iters = items[k].l1_iters;
            loc.irls_iters = items[k].irls_iters;
            loc.irls_runtime = dt;  // the whole batch ran in the one launch
            if (items[k].status == IROTAVG_OK) {
                const long nv = items[k].nv;
                std::vector<double> &Q = vg->scratch.Q;
                const std::vector<double> &Qa = vg->scratch.Qa;
                for (long r = 0; r < nv; r++)
                    for (int c = 0; c < 4; c++) Q[(size_t)c * nv + r] = Qa[(size_t)4 * r + c];
                rotavg_writeback(vg, items[k].f, nv);
                vg->last = loc;
            } else if (first_err == IROTAVG_OK) {
                first_err = items[k].status;
            }
            if (infos) infos[owner[k]] = loc;
        }
    } catch (...) {
        return IROTAVG_ERR_HIP;
    }
    if (rc != IROTAVG_OK && first_err == IROTAVG_OK) first_err = rc;
    return first_err;
}

// rmat2quat (src/ViewGraph.cpp:1175-1203) and q.normalized().toRotationMatrix() (:1426-1433) for
// callers that hold Pose-style 3x3 matrices; row-major R, q = [x y z w].
void irotavg_rmat2quat(const double R[9], double q[4]) {
    if (R && q) rmat2quat(R, q);
}
void irotavg_quat2rmat(const double q[4], double R[9]) {
    if (R && q) quat2rmat(q, R);
}

// ViewGraph::savePoses (src/ViewGraph.cpp:1206-1231): one line per view,
// `id \t qw \t qx \t qy \t qz \t tx \t ty \t tz`, 17 significant digits, scientific. Translations
// belong to the vision front-end and are not tracked here: t (3 doubles per view) may be NULL (zeros).
int irotavg_viewgraph_save_poses(const irotavg_viewgraph *vg, const char *filename, const double *t) {
    if (!vg || !filename) return IROTAVG_ERR_BAD_ARG;
    std::FILE *fs = std::fopen(filename, "w");
    if (!fs) return IROTAVG_ERR_BAD_ARG;  // "Unable to save results."
    for (size_t v = 0; v < vg->pose.size(); v++) {
        double q[4], Rm[9];
        vg->pose_read(v, Rm);
        rmat2quat(Rm, q);
        const double tx = t ? t[3 * v] : 0.0, ty = t ? t[3 * v + 1] : 0.0, tz = t ? t[3 * v + 2] : 0.0;
        std::fprintf(fs, "%zu\t%.17e\t%.17e\t%.17e\t%.17e\t%.17e\t%.17e\t%.17e\n", v, q[3], q[0], q[1],
                     q[2], tx, ty, tz);
    }
    std::fclose(fs);
    return IROTAVG_OK;
}

// The single-kernel window pipeline on caller data (same layout as irotavg_l1ra / irotavg_irls):
// l1ra(l1_iters) then irls(cost, sigma, irls_iters) in one launch. Only for problems that fit
// (<= 64 free views, <= 640 edges, <= 320 views): IROTAVG_ERR_BAD_ARG otherwise.
// kernel: 0 = pick (wave-resident variant when <= 16 free views and <= 64 edges), 1 = the general
// LDS kernel, 2 = the wave-resident kernel (BAD_ARG if the problem is too large for it).
int irotavg_window_solve_kernel(int64_t m, int64_t n_total, int f, const int32_t *I, const double *QQ,
                                int64_t ldqq, double *Q, int64_t ldq, int cost, double sigma, int l1_iters,
                                int irls_iters, double change_th, double *weights, int *l1_out,
                                int *irls_out, int kernel) {
    if (!I || !QQ || !Q || m <= 0 || n_total <= 0 || ldqq < m || ldq < n_total) return IROTAVG_ERR_BAD_ARG;
    if (kernel < 0 || kernel > 2) return IROTAVG_ERR_BAD_ARG;
    if (cost < IROTAVG_L2 || cost > IROTAVG_WELSCH) return IROTAVG_ERR_UNKNOWN_COST;
    // the single-workgroup kernels index LDS / global arrays with f and the edge endpoints unguarded
    if (f < 0 || f >= n_total) return IROTAVG_ERR_BAD_ARG;
    if (!irh::window_fits((int)n_total, f, (int)m)) return IROTAVG_ERR_BAD_ARG;
    for (int64_t k = 0; k < 2 * m; k++)
        if (I[k] < 0 || I[k] >= n_total) return IROTAVG_ERR_BAD_ARG;
    if (kernel == 2 && !irh::window_fits_wave((int)n_total, f, (int)m)) return IROTAVG_ERR_BAD_ARG;
    if (irotavg_device_count() <= 0) return IROTAVG_ERR_NO_DEVICE;
    try {
        std::vector<double> qa((size_t)4 * m), Qa((size_t)4 * n_total);
        for (int64_t k = 0; k < m; k++)
            for (int c = 0; c < 4; c++) qa[(size_t)4 * k + c] = QQ[(size_t)c * ldqq + k];
        for (int64_t r = 0; r < n_total; r++)
            for (int c = 0; c < 4; c++) Qa[(size_t)4 * r + c] = Q[(size_t)c * ldq + r];
        irh::WindowSolver *ws = irh::window_solver_new();
        const int rc = irh::window_solve(*ws, (int)n_total, f, (int)m, I, qa.data(), Qa.data(), weights,
                                         l1_iters, irls_iters, cost, sigma, change_th, l1_out, irls_out,
                                         kernel);
        irh::window_solver_delete(ws);
        for (int64_t r = 0; r < n_total; r++)
            for (int c = 0; c < 4; c++) Q[(size_t)c * ldq + r] = Qa[(size_t)4 * r + c];
        return rc;
    } catch (...) {
        return IROTAVG_ERR_HIP;
    }
}

int irotavg_window_solve(int64_t m, int64_t n_total, int f, const int32_t *I, const double *QQ,
                         int64_t ldqq, double *Q, int64_t ldq, int cost, double sigma, int l1_iters,
                         int irls_iters, double change_th, double *weights, int *l1_out, int *irls_out) {
    return irotavg_window_solve_kernel(m, n_total, f, I, QQ, ldqq, Q, ldq, cost, sigma, l1_iters, irls_iters,
                                       change_th, weights, l1_out, irls_out, 0);
}

}  // extern "C"

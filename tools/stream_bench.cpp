// stream_bench -- BASELINE.json config 5 (incremental mode) driven the way the reference drives it: a native loop
// around the view-graph API (src/IRotAvg.cpp:280-378: a frame is admitted, connected to its <= 4 predecessors
// -- vg_win_size, :158-161 --, a ground-truth correction fixes a pose every 20 frames, rotAvg(10) runs on every
// frame and rotAvg(5000000) on a loop closure). tools/bench_incremental.py is the same experiment with a Python
// loop, whose interpreter overhead (~20 us per view) is a quarter of its time; this program measures the library.
//
//   stream_bench [warm_views [streamed_views [loop_closures [seed [sessions [prepare]]]]]]      -> one JSON line
//
// prepare (default 1): irotavg_viewgraph_prepare after the warm graph is loaded, outside the timed loop like the
// loading itself -- the one-time costs of a process (device allocations, kernel code loads) are then not part of the
// first loop closure; 0 leaves them there (global_rotavg_ms lists every global re-solve either way).
//
// sessions > 1: that many INDEPENDENT sequences (a server tracking several cameras) advanced in lock-step; the
// rotAvg(10) windows of a step are solved by ONE launch (irotavg_viewgraph_rot_avg_batch, a workgroup per window),
// loop-closure re-solves one by one. views_per_s is then the aggregate over all sessions.
//
// Synthetic front-end: ground-truth rotations ~ uniform, relative rotations with N(0, 0.01^2) rad of noise, the
// initial pose of a new view chained from its predecessor; measurements are generated before the timed loop.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <set>
#include <vector>

#include "../include/irotavg_hip.h"

struct Q4 {
    double x, y, z, w;
};
static Q4 qmul(const Q4 &a, const Q4 &b) {
    return Q4{a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
              a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
static Q4 qconj(const Q4 &a) { return Q4{-a.x, -a.y, -a.z, a.w}; }
static Q4 qexp(double rx, double ry, double rz) {
    const double th = std::sqrt(rx * rx + ry * ry + rz * rz);
    const double c = th > 0 ? std::sin(th / 2) / th : 0.5;
    return Q4{rx * c, ry * c, rz * c, std::cos(th / 2)};
}
struct R9 {
    double m[9];
};
static R9 rmat(const Q4 &q) {
    R9 r;
    const double v[4] = {q.x, q.y, q.z, q.w};
    irotavg_quat2rmat(v, r.m);
    return r;
}
static R9 matmul(const R9 &a, const R9 &b) {
    R9 c;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) c.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
    return c;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Session {
    irotavg_viewgraph *vg = nullptr;
    std::vector<Q4> gt;
    std::vector<R9> rrel;                       // 4 per streamed view
    std::vector<std::pair<int, R9>> loop_edge;  // per view: (from, R) or (-1, .)
};

int main(int argc, char **argv) {
    const int warm = argc > 1 ? std::atoi(argv[1]) : 50000, stream = argc > 2 ? std::atoi(argv[2]) : 50000;
    const int loops = argc > 3 ? std::atoi(argv[3]) : 10;
    const unsigned seed = argc > 4 ? (unsigned)std::atoi(argv[4]) : 0u;
    const int sessions = argc > 5 ? std::max(1, std::atoi(argv[5])) : 1;
    const int prepare = argc > 6 ? std::atoi(argv[6]) : 1;
    const int n = warm + stream, fix_every = 20;
    const double noise = 0.01;
    if (irotavg_device_count() <= 0) {
        std::fprintf(stderr, "stream_bench: no HIP device (the library has no CPU path)\n");
        return 2;
    }
    std::vector<Session> S((size_t)sessions);
    double warm_s = 0;
    for (int sx = 0; sx < sessions; sx++) {
        Session &Z = S[(size_t)sx];
        std::mt19937_64 rng(seed + 7919u * (unsigned)sx);
        std::normal_distribution<double> nd(0.0, 1.0);
        Z.gt.resize((size_t)n);
        for (auto &q : Z.gt) {
            q = Q4{nd(rng), nd(rng), nd(rng), nd(rng)};
            const double s = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
            q = Q4{q.x / s, q.y / s, q.z / s, q.w / s};
        }
        auto rel = [&](int i, int j) {  // R_ij with R_j = R_ij R_i, noisy
            return rmat(qmul(qexp(noise * nd(rng), noise * nd(rng), noise * nd(rng)),
                             qmul(Z.gt[(size_t)j], qconj(Z.gt[(size_t)i]))));
        };
        std::set<int> loop_at;
        if (stream > 200) {
            std::uniform_int_distribution<int> pick(warm + 100, n - 1);
            while ((int)loop_at.size() < std::min(loops, stream - 100)) loop_at.insert(pick(rng));
        }
        if (irotavg_viewgraph_create(&Z.vg, nullptr) != IROTAVG_OK) return 3;
        const double tb = now();
        for (int v = 0; v < warm; v++) {  // poses as a converged run would have left them
            const R9 r0 = rmat(qmul(qexp(noise * nd(rng), noise * nd(rng), noise * nd(rng)), Z.gt[(size_t)v]));
            irotavg_viewgraph_add_view(Z.vg, r0.m);
            for (int d = 1; d <= std::min(4, v); d++) {
                const R9 rij = rel(v - d, v);
                irotavg_viewgraph_connect(Z.vg, v - d, v, rij.m);
            }
            if (v % fix_every == 0) {
                const R9 g = rmat(Z.gt[(size_t)v]);
                irotavg_viewgraph_fix_pose(Z.vg, v, g.m);
            }
        }
        warm_s += now() - tb;
        // the front-end's measurements of the streamed part, generated up front
        Z.rrel.resize((size_t)stream * 4);
        for (int v = warm; v < n; v++)
            for (int d = 1; d <= 4; d++) Z.rrel[(size_t)(v - warm) * 4 + (d - 1)] = rel(v - d, v);
        Z.loop_edge.assign((size_t)n, {-1, R9{}});
        for (int v : loop_at) {
            std::uniform_int_distribution<int> from(0, std::max(1, v - 1000) - 1);  // at least 1000 views back when there are that many
            const int u = from(rng);
            Z.loop_edge[(size_t)v] = {u, rel(u, v)};
        }
    }
    double prepare_s = 0;
    if (prepare) {
        const double tp = now();
        for (auto &Z : S)
            if (irotavg_viewgraph_prepare(Z.vg) != IROTAVG_OK) return 5;
        prepare_s = now() - tp;
    }
    std::vector<double> lat_local, lat_global;
    lat_local.reserve((size_t)stream);
    long long edges_solved = 0, l1_sum = 0, irls_sum = 0, win_solved = 0;
    std::vector<irotavg_viewgraph *> batch((size_t)sessions);
    std::vector<irotavg_rotavg_info> infos((size_t)sessions);
    const double t0 = now();
    for (int v = warm; v < n; v++) {
        int nb = 0;
        for (int sx = 0; sx < sessions; sx++) {
            Session &Z = S[(size_t)sx];
            R9 prev;
            irotavg_viewgraph_get_pose(Z.vg, v - 1, prev.m);
            const R9 &r1 = Z.rrel[(size_t)(v - warm) * 4];
            const R9 init = matmul(r1, prev);  // the front-end's initial pose
            irotavg_viewgraph_add_view(Z.vg, init.m);
            for (int d = 1; d <= 4; d++)
                irotavg_viewgraph_connect(Z.vg, v - d, v, Z.rrel[(size_t)(v - warm) * 4 + (d - 1)].m);
            const bool loop = Z.loop_edge[(size_t)v].first >= 0;
            if (loop) irotavg_viewgraph_connect(Z.vg, Z.loop_edge[(size_t)v].first, v, Z.loop_edge[(size_t)v].second.m);
            if (v % fix_every == 0) {
                const R9 g = rmat(Z.gt[(size_t)v]);
                irotavg_viewgraph_fix_pose(Z.vg, v, g.m);
            }
            if (loop || sessions == 1) {
                irotavg_rotavg_info info{};
                const double t = now();
                const int rc = irotavg_viewgraph_rot_avg(Z.vg, loop ? 5000000 : 10, &info);
                (loop ? lat_global : lat_local).push_back(now() - t);
                if (rc != IROTAVG_OK) {
                    std::fprintf(stderr, "stream_bench: rot_avg failed at view %d: %s\n", v, irotavg_error_string(rc));
                    return 4;
                }
                if (!info.skipped) edges_solved += (long long)info.n_edges * std::max(info.irls_iters, 1);
                if (!info.skipped && !loop) {
                    l1_sum += info.l1_iters;
                    irls_sum += info.irls_iters;
                    win_solved++;
                }
            } else {
                batch[(size_t)nb++] = Z.vg;
            }
        }
        if (nb > 0) {  // the windows of the other sessions: one launch
            const double t = now();
            const int rc = irotavg_viewgraph_rot_avg_batch(batch.data(), nb, 10, infos.data());
            lat_local.push_back(now() - t);
            if (rc != IROTAVG_OK) {
                std::fprintf(stderr, "stream_bench: rot_avg_batch failed at view %d: %s\n", v, irotavg_error_string(rc));
                return 4;
            }
            for (int b = 0; b < nb; b++)
                if (!infos[(size_t)b].skipped) edges_solved += (long long)infos[(size_t)b].n_edges * std::max(infos[(size_t)b].irls_iters, 1);
        }
    }
    const double dt = now() - t0;
    double err_sum = 0, err_max = 0;
    int err_n = 0;
    for (int sx = 0; sx < sessions; sx++)
        for (int v = warm; v < n; v += 97) {
            R9 r;
            irotavg_viewgraph_get_pose(S[(size_t)sx].vg, v, r.m);
            const R9 g = rmat(S[(size_t)sx].gt[(size_t)v]);
            double tr = 0;
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) tr += r.m[3 * i + j] * g.m[3 * i + j];
            const double a = std::acos(std::min(1.0, std::max(-1.0, (tr - 1) / 2)));
            err_sum += a;
            err_max = std::max(err_max, a);
            err_n++;
        }
    auto mean = [](const std::vector<double> &x) {
        double s = 0;
        for (double v : x) s += v;
        return x.empty() ? 0.0 : s / (double)x.size();
    };
    std::sort(lat_local.begin(), lat_local.end());
    const double p99 = lat_local.empty() ? 0.0 : lat_local[(size_t)(0.99 * (double)(lat_local.size() - 1))];
    std::printf("{\"mode\": \"incremental\", \"driver\": \"native (tools/stream_bench.cpp)\", \"sessions\": %d, \"warm_views\": %d, "
                "\"streamed_views\": %d, \"loop_closures\": %d, \"views_per_s\": %.1f, \"seconds\": %.4f, "
                "\"warm_build_seconds\": %.3f, \"prepare_seconds\": %.3f, \"local_rotavg_ms_mean\": %.5f, \"local_rotavg_ms_p99\": %.5f, "
                "\"global_rotavg_ms_mean\": %.3f, \"irls_edge_updates_per_s\": %.1f, \"mean_angular_error_rad\": %.6f, "
                "\"max_angular_error_rad\": %.6f, \"window_l1_iters_mean\": %.3f, \"window_irls_iters_mean\": %.3f, \"global_rotavg_ms\": [",
                sessions, warm, stream, (int)lat_global.size(), (double)stream * sessions / dt, dt, warm_s, prepare_s, 1e3 * mean(lat_local),
                1e3 * p99, 1e3 * mean(lat_global), (double)edges_solved / dt, err_sum / std::max(err_n, 1), err_max,
                (double)l1_sum / std::max(win_solved, 1LL), (double)irls_sum / std::max(win_solved, 1LL));
    for (size_t k = 0; k < lat_global.size(); k++) std::printf("%s%.3f", k ? ", " : "", 1e3 * lat_global[k]);
    std::printf("]}\n");
    for (auto &Z : S) irotavg_viewgraph_destroy(Z.vg);
    return 0;
}

// build.cpp -- host-side construction of the device-resident graph: edge streams, the vertex
// adjacency (CSR of the weighted Laplacian A'WA restricted to the free views) and the
// aggregation hierarchy used by the multigrid preconditioner.
//
// Reference semantics encoded here:
//  * make_A (ral/l1_irls.cpp:755-780): row k of A has +1 at column j-f and -1 at column i-f; if
//    j < f the row is EMPTY even when i is free (:770-771); if only i < f the +1 stays.
//  * make_AtA (ral/l1_irls.cpp:811-848): the Hessian pattern used by l1decode_pd skips fixed
//    endpoints independently (:825-843), so an edge with i free, j fixed still adds to (i,i).
// The CSR pattern is static for the life of the handle; only values are refreshed per solve.
#include <algorithm>
#include <atomic>
#include <numeric>
#include <thread>

#include "graph.hpp"

namespace irh {

// parallel_for (common.hpp): the host phases below are loops over edges, rows or slices with independent
// iterations: they run on up to 16 host threads (contiguous chunks; every result is independent of the thread
// count -- where a serial loop defined an order, the order is restored by sorting on the edge id).
static int pow2floor(int v) {
    int p = 1;
    while (2 * p <= v) p *= 2;
    return p;
}

// SELL-64 positions of a CSR pattern (layout: common.hpp): pos[t] for every CSR slot t, slice
// offsets, near widths, padded length. Near = column inside the LDS window of the row's tile.
struct SellMap {
    int nsl = 0;
    std::vector<int> sl_off, sl_near;  // nsl + 1, nsl
    std::vector<int> pos;              // per CSR slot
    std::vector<int> kidx;             // per CSR slot: entry index within its SELL row
    long long len = 0;                 // 64 * sl_off[nsl]
};
static SellMap sell_map(int n, const std::vector<int> &rowptr, const std::vector<int> &col) {
    SellMap M;
    M.nsl = (n + 63) / 64;
    M.sl_off.assign((size_t)M.nsl + 1, 0);
    M.sl_near.assign((size_t)M.nsl, 0);
    // near = within kWinHalo rows of the row itself: inside the LDS window of EVERY tile that contains
    // the row, whatever the tiling (256-row tiles of the classic kernels, slice ranges of cgcg.hip)
    auto is_near = [&](int r, int c) { return c >= r - kWinHalo && c <= r + kWinHalo; };
    auto roundup = [](int w) { return (w + kSellUnroll - 1) / kSellUnroll * kSellUnroll; };
    std::vector<int> width((size_t)M.nsl, 0);
    parallel_for(M.nsl, 256, [&](int64_t s0, int64_t s1, int) {
        for (int sl = (int)s0; sl < (int)s1; sl++) {
            int wn = 0, wf = 0;
            for (int r = sl * 64; r < std::min(n, sl * 64 + 64); r++) {
                int nn = 0;
                for (int t = rowptr[r]; t < rowptr[r + 1]; t++) nn += is_near(r, col[t]) ? 1 : 0;
                wn = std::max(wn, nn);
                wf = std::max(wf, rowptr[r + 1] - rowptr[r] - nn);
            }
            // never narrower than one batch: the row kernels request the first kSellUnroll entries of a row before
            // they know its length (a slice of views without a free neighbour -- e.g. a star around the fixed view
            // -- would otherwise be zero wide and those requests would land in front of the array)
            wn = std::max(roundup(wn), kSellUnroll);
            wf = roundup(wf);
            M.sl_near[sl] = wn;
            width[sl] = wn + wf;
        }
    });
    for (int sl = 0; sl < M.nsl; sl++) M.sl_off[sl + 1] = M.sl_off[sl] + width[sl];
    M.len = 64ll * M.sl_off[M.nsl];
    M.pos.resize((size_t)rowptr[n]);
    M.kidx.resize((size_t)rowptr[n]);
    parallel_for(n, 4096, [&](int64_t r0, int64_t r1, int) {
        for (int r = (int)r0; r < (int)r1; r++) {
            const int sl = r >> 6, lane = r & 63;
            int kn = 0, kf = M.sl_near[sl];
            for (int t = rowptr[r]; t < rowptr[r + 1]; t++) {
                const int k = is_near(r, col[t]) ? kn++ : kf++;
                M.pos[t] = (int)sell_pos(M.sl_off[sl], k, lane);
                M.kidx[t] = k;
            }
        }
    });
    return M;
}

// The shape of the hierarchy: rows and aggregation factor of every level, from the size of the graph, its
// level-0 entry count and whether level 0 has far (loop-closure) entries. Also fixes g.opt.mg_dense_max when
// it was left to the library. Pure arithmetic: the host build and the device build (gbuild.hip) share it.
HierPlan plan_hierarchy(Graph &g, int n0, int64_t nnz0, bool far0) {
    HierPlan P;
    P.n.push_back(n0);
    const int max_levels = std::max(1, std::min(g.opt.mg_levels_max, (int)kMaxLevels));
    const bool tune = getenv("IROTAVG_NO_SMALL_TUNING") == nullptr;
    if (g.opt.mg_dense_max <= 0) {
        // The dense level is re-inverted whenever the weights move non-uniformly. For a band graph that is the
        // cheap banded inverse and rare; with loop closures it is the full Gauss-Jordan sweep (n/32 block
        // steps of ~21 us) in almost every IRLS iteration, and up to ~70k views a level of 550..1100 rows
        // behind an aggregation factor of 16..64 costs 2-5 more PCG iterations but half the sweep
        // (10k/150k: 5.9 -> 4.7 ms, 16k: 9.1 -> 6.0 ms, 30k: 11.5 -> 8.5 ms per irls call; at 100k views the
        // factor would be 128 and the iterations double: 2048 stays)
        g.opt.mg_dense_max = (tune && n0 <= 70400 && far0) ? 1100 : 2048;
    }
    // A graph without far (loop-closure) entries on one GPU can run its PCG iteration as two launches
    // (cgcg.hip), which needs aggregates of 8 on levels 0 AND 1. The rule below would stop level 1 short
    // (aggregates of 2 or 4 that just reach the dense level: 20k views -> 2500 -> 1250) and leave such a
    // graph on the slowest path (six launches per iteration): measured 389 vs 667 M edge-updates/s at
    // 20k / 400k, 1223 vs 1669 M at 60k / 1.2M, for two more PCG iterations per solve.
    const bool band0 = g.ng == 0 && g.opt.mg_multiplicative_top != 1 && g.opt.no_fused_pspmv != 1 &&
                       g.opt.pcg_classic != 1 && (n0 + 63) / 64 <= 4 * kMaxParts && !far0;
    // A dense level of more than ~1500 rows costs more per PCG iteration (its 8 n^2-byte apply) than a third
    // level with the two-launch iteration saves: a band graph of >= 18 edges per view whose level 1 has
    // 1536..2048 rows coarsens once more (16k views / 320k edges: 348 -> 557 M edge-updates/s; at 15 edges per
    // view the extra iterations of three levels eat the gain, at 4 they double)
    auto go_on = [&]() {
        if (P.n.back() > g.opt.mg_dense_max) return true;
        return tune && band0 && P.n.size() == 2 && P.n.back() >= 1536 && (P.n.back() + 7) / 8 >= 64 &&
               nnz0 >= 36ll * n0;
    };
    while ((int)P.n.size() < max_levels && go_on()) {
        const int Fn = P.n.back();
        const size_t depth = P.n.size();
        // aggregate = `agg` contiguous rows, a power of two <= 64 so it never straddles a slice
        int agg = (depth == 1) ? g.opt.mg_agg0 : g.opt.mg_agg;
        if (agg <= 0) {
            agg = 8;  // few, large steps: every extra level costs two latency-bound sweeps per cycle
            // do not overshoot the dense level: the smallest factor that reaches it -- except on level 1
            // of a graph that can take the two-launch iteration (see band0 above)
            const bool keep8 = band0 && depth == 2 && (Fn + 7) / 8 >= 64;
            // ... and on level 0 when aggregates of 8 still leave >= 512 dense rows (5k views: 625 dense rows
            // instead of 1250: 125 -> 157 M)
            const bool keep8_0 = tune && depth == 1 && (Fn + 7) / 8 >= 512;
            for (int s2 = 2; s2 < agg && !keep8 && !keep8_0; s2 *= 2)
                if ((Fn + s2 - 1) / s2 <= g.opt.mg_dense_max) {
                    agg = s2;
                    break;
                }
        }
        // loop-closure graphs (dense level capped at 1100 above): two levels with aggregates of 16 beat three
        // with 8 x 2 by ~5 % where both reach the cap (8.8k-17.6k views)
        if (tune && depth == 1 && g.opt.mg_agg0 <= 0 && g.opt.mg_dense_max == 1100 && (Fn + 7) / 8 > 1100 &&
            (Fn + 15) / 16 <= 1100)
            agg = 16;
        agg = std::min(pow2floor(std::max(agg, 2)), 64);
        P.agg.push_back(agg);
        P.n.push_back((Fn + agg - 1) / agg);
    }
    P.agg.push_back(0);  // the coarsest level aggregates no further
    return P;
}

int build_graph(Graph &g, const int32_t *I, const double *QQ, int64_t ldqq) {
    // Patterns on the device (gbuild.hip) for a single-GPU graph that is large enough to fill it; on the host for
    // shards (ghost views) and small graphs (sliding windows that miss the single-kernel path: a few hundred
    // edges are built faster than a dozen launches are issued). IROTAVG_HOST_BUILD=1 / =0 force either.
    const char *e = getenv("IROTAVG_HOST_BUILD");
    const bool host = e ? atoi(e) != 0 : (g.ng > 0 || g.m < 20000);
    return host ? build_graph_host(g, I, QQ, ldqq) : build_graph_device(g, I, QQ, ldqq);
}

int build_graph_host(Graph &g, const int32_t *I, const double *QQ, int64_t ldqq) {
    const bool timing = getenv("IROTAVG_BUILD_TIMING") != nullptr;
    double tlast = now_seconds();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const double t = now_seconds();
        (void)hipStreamSynchronize(g.stream);  // attribute queued uploads to the phase that issued them
        const double t2 = now_seconds();
        std::fprintf(stderr, "[irotavg_hip build] %-28s %8.2f ms host + %6.2f ms queued\n", what, 1e3 * (t - tlast),
                     1e3 * (t2 - t));
        tlast = t2;
    };
    const int64_t m = g.m;
    const int f = g.f;
    const int fo = g.f + g.ng;  // operator offset: fixed and ghost views have no row
    const int nu = g.no;        // rows of the operator (owned free views)
    hipStream_t s = g.stream;

    // ---- edge streams -------------------------------------------------------------------
    g.mpad = (m + 63) / 64 * 64;  // streams are padded so kernels may read whole 16-B pairs
    std::vector<int> ei((size_t)g.mpad, 0), ej((size_t)g.mpad, 0);
    std::vector<uint8_t> eflag((size_t)g.mpad, 0);
    std::atomic<bool> bad(false);
    parallel_for(m, 65536, [&](int64_t k0, int64_t k1, int) {
        for (int64_t k = k0; k < k1; k++) {
            const int i = I[2 * k], j = I[2 * k + 1];
            if (i < 0 || j < 0 || i >= g.n_total || j >= g.n_total) {
                bad = true;
                return;
            }
            ei[k] = i;
            ej[k] = j;
            uint8_t fl = 0;
            if (j >= f) {
                if (i >= f && i == j) {
                    fl = EF_CI;  // self loop: the -1 overwrites the +1
                } else {
                    fl |= EF_CJ;
                    if (i >= f) fl |= EF_CI;
                }
            }
            eflag[k] = fl;
        }
    });
    if (bad) return IROTAVG_ERR_BAD_ARG;
    g.ei.upload(ei, s);
    g.ej.upload(ej, s);
    g.eflag.upload(eflag, s);
    g.qq.alloc((size_t)4 * g.mpad);
    g.qq.zero(s);
    for (int c = 0; c < 4; c++)
        IRH_CHECK(hipMemcpyAsync(g.qq.p + (size_t)c * g.mpad, QQ + (size_t)c * ldqq,
                                 sizeof(double) * (size_t)m, hipMemcpyHostToDevice, s));
    g.er.alloc((size_t)3 * g.mpad);
    g.er.zero(s);
    g.dw.alloc((size_t)g.mpad);
    fill(g, g.dw.p, (long long)g.mpad, 1.0);  // default weights (irls resets them anyway, :577)
    g.Q.alloc((size_t)g.n_total);
    g.Q.zero(s);

    lap("edge streams + upload");
    // ---- level-0 adjacency --------------------------------------------------------------
    // Endpoint classes by local index: [0,f) fixed, [f,fo) ghost (free, owned by another shard),
    // [fo, n_total) owned. Owned-owned edges become matrix entries; an edge from an owned view
    // to a fixed or ghost view is a "boundary slot" of the owned row.
    auto cls = [&](int v) { return v < f ? 0 : (v < fo ? 1 : 2); };
    std::vector<int> rowptr((size_t)nu + 1, 0), bptr((size_t)nu + 1, 0);
    auto bump = [](int &x) { return __atomic_fetch_add(&x, 1, __ATOMIC_RELAXED); };
    parallel_for(m, 65536, [&](int64_t k0, int64_t k1, int) {
        for (int64_t k = k0; k < k1; k++) {
            const int ci = cls(ei[k]), cj = cls(ej[k]);
            const int i = ei[k] - fo, j = ej[k] - fo;
            if (ci == 2 && cj == 2 && i != j) {
                bump(rowptr[i + 1]);
                bump(rowptr[j + 1]);
            } else if (ci == 2 && cj == 2) {
                bump(bptr[i + 1]);  // self loop
            } else if (cj == 2) {
                bump(bptr[j + 1]);  // i fixed or ghost
            } else if (ci == 2) {
                bump(bptr[i + 1]);  // j fixed (dropped by make_A, kept by make_AtA) or ghost
            }
        }
    });
    for (int v = 0; v < nu; v++) {
        rowptr[v + 1] += rowptr[v];
        bptr[v + 1] += bptr[v];
    }
    const int64_t nnz0 = rowptr[nu];
    const int64_t nb = bptr[nu];
    if (nnz0 > 0x7fffffffLL || m > 0x7fffffffLL) return IROTAVG_ERR_BAD_ARG;
    struct Ent {
        int col;
        uint32_t eid;
    };
    std::vector<Ent> ents((size_t)nnz0);
    struct BEnt {
        uint32_t eid;
        int ghost;
        uint8_t flag;
    };
    std::vector<BEnt> bents((size_t)nb);
    {
        // slots are claimed with atomic counters (arbitrary order inside a row); the sorts below
        // restore the order a serial pass over the edges would give
        std::vector<int> pos(rowptr.begin(), rowptr.end() - 1), bpos(bptr.begin(), bptr.end() - 1);
        parallel_for(m, 65536, [&](int64_t k0, int64_t k1, int) {
            for (int64_t k = k0; k < k1; k++) {
                const int ci = cls(ei[k]), cj = cls(ej[k]);
                const int i = ei[k] - fo, j = ej[k] - fo;
                if (ci == 2 && cj == 2 && i != j) {
                    ents[bump(pos[j])] = Ent{i, (uint32_t)(k << 1) | 1u};  // row j: +1 coefficient
                    ents[bump(pos[i])] = Ent{j, (uint32_t)(k << 1)};       // row i: -1 coefficient
                } else if (ci == 2 && cj == 2) {
                    bents[bump(bpos[i])] = BEnt{(uint32_t)(k << 1), -1, (uint8_t)(BF_IRLS | BF_L1H | BF_NEG)};
                } else if (cj == 2) {  // row j, other endpoint i fixed or ghost
                    bents[bump(bpos[j])] = BEnt{(uint32_t)(k << 1) | 1u, ci == 1 ? ei[k] - f : -1,
                                                (uint8_t)(BF_IRLS | BF_L1H)};
                } else if (ci == 2) {  // row i, other endpoint j fixed (make_A drops it) or ghost
                    bents[bump(bpos[i])] = BEnt{(uint32_t)(k << 1), cj == 1 ? ej[k] - f : -1,
                                                (uint8_t)(cj == 1 ? (BF_IRLS | BF_L1H) : BF_L1H)};
                }
            }
        });
    }
    lap("adjacency count/fill");
    // sort each row by column, edge order breaking ties (locality for the gathers, a canonical
    // order for the coarse-level maps); boundary slots of a row by edge order
    std::vector<int> col((size_t)nnz0);
    std::vector<uint32_t> slot_eid((size_t)nnz0);
    std::vector<uint32_t> beid((size_t)nb);
    std::vector<uint8_t> bflag((size_t)nb);
    std::vector<int> bghost((size_t)nb, -1);
    parallel_for(nu, 2048, [&](int64_t v0, int64_t v1, int) {
        for (int v = (int)v0; v < (int)v1; v++) {
            std::sort(ents.begin() + rowptr[v], ents.begin() + rowptr[v + 1], [](const Ent &a, const Ent &b) {
                return a.col != b.col ? a.col < b.col : a.eid < b.eid;
            });
            for (int t = rowptr[v]; t < rowptr[v + 1]; t++) {
                col[t] = ents[t].col;
                slot_eid[t] = ents[t].eid;
            }
            std::sort(bents.begin() + bptr[v], bents.begin() + bptr[v + 1],
                      [](const BEnt &a, const BEnt &b) { return a.eid < b.eid; });
            for (int t = bptr[v]; t < bptr[v + 1]; t++) {
                beid[t] = bents[t].eid;
                bflag[t] = bents[t].flag;
                bghost[t] = bents[t].ghost;
            }
        }
    });
    ents.clear();
    ents.shrink_to_fit();

    g.bptr.upload(bptr, s);
    g.beid.upload(beid, s);
    g.bflag.upload(bflag, s);
    g.bghost.upload(bghost, s);
    g.bval.alloc((size_t)nb);
    g.bval.zero(s);
    g.PG.alloc((size_t)g.ng + 1);
    g.PG.zero(s);

    lap("row sort + boundary upload");
    // ---- hierarchy ----------------------------------------------------------------------
    // Host CSR patterns of every level first, SELL conversion + uploads second.
    struct HostLevel {
        int n = 0, agg = 0;
        std::vector<int> rowptr, col;  // off-diagonal pattern (CSR, host only)
        std::vector<int> cptr, cidx;   // value refresh from the finer level (CSR slots of the finer level)
    };
    std::vector<HostLevel> H;
    H.emplace_back();
    H[0].n = nu;
    H[0].rowptr = std::move(rowptr);
    H[0].col = std::move(col);
    // far (loop-closure) entries on level 0? -- decides the dense level's size and which PCG kernels can run
    bool far0 = false;
    {
        std::atomic<bool> far(false);
        const HostLevel &h0 = H[0];
        parallel_for(h0.n, 4096, [&](int64_t r0, int64_t r1, int) {
            for (int r = (int)r0; r < (int)r1 && !far; r++)
                for (int t = h0.rowptr[r]; t < h0.rowptr[r + 1]; t++)
                    if (h0.col[t] < r - kWinHalo || h0.col[t] > r + kWinHalo) {
                        far = true;
                        break;
                    }
        });
        far0 = far;
    }
    const HierPlan plan = plan_hierarchy(g, H[0].n, (int64_t)H[0].rowptr[H[0].n], far0);
    while (H.size() < plan.n.size()) {
        HostLevel &F = H.back();
        int agg = plan.agg[H.size() - 1];
        F.agg = agg;
        HostLevel C;
        C.n = (F.n + agg - 1) / agg;
        C.rowptr.assign((size_t)C.n + 1, 0);
        // every thread builds the rows of a contiguous range of coarse rows into its own vectors;
        // they are concatenated in range order afterwards
        struct Part {
            int64_t c0 = 0, c1 = 0;
            std::vector<int> col, cidx, cptr_local, rowlen;  // cptr_local: offsets into this part's cidx
        };
        std::vector<Part> parts(16);
        parallel_for(C.n, 512, [&](int64_t c0, int64_t c1, int tid) {
            Part &P = parts[(size_t)tid];
            P.c0 = c0;
            P.c1 = c1;
            P.rowlen.assign((size_t)(c1 - c0), 0);
            std::vector<std::pair<int, int>> tmp;  // (coarse col, fine slot)
            for (int Ic = (int)c0; Ic < (int)c1; Ic++) {
                tmp.clear();
                const int v0 = Ic * agg, v1 = std::min(F.n, v0 + agg);
                for (int v = v0; v < v1; v++)
                    for (int t = F.rowptr[v]; t < F.rowptr[v + 1]; t++) {
                        const int Jc = F.col[t] / agg;
                        if (Jc != Ic) tmp.emplace_back(Jc, t);
                    }
                std::sort(tmp.begin(), tmp.end());
                int nrow = 0;
                for (size_t q = 0; q < tmp.size(); q++) {
                    if (q == 0 || tmp[q].first != tmp[q - 1].first) {
                        P.cptr_local.push_back((int)P.cidx.size());
                        P.col.push_back(tmp[q].first);
                        nrow++;
                    }
                    P.cidx.push_back(tmp[q].second);
                }
                P.rowlen[(size_t)(Ic - c0)] = nrow;
            }
        });
        std::sort(parts.begin(), parts.end(), [](const Part &a, const Part &b) {
            return (a.c1 > a.c0) != (b.c1 > b.c0) ? (a.c1 > a.c0) : a.c0 < b.c0;
        });
        size_t ncol = 0, nidx = 0;
        for (const Part &P : parts) {
            ncol += P.col.size();
            nidx += P.cidx.size();
        }
        C.col.reserve(ncol);
        C.cidx.reserve(nidx);
        C.cptr.reserve(ncol + 1);
        for (const Part &P : parts) {
            if (P.c1 <= P.c0) continue;
            const int base = (int)C.cidx.size();
            for (int o : P.cptr_local) C.cptr.push_back(base + o);
            C.col.insert(C.col.end(), P.col.begin(), P.col.end());
            C.cidx.insert(C.cidx.end(), P.cidx.begin(), P.cidx.end());
            for (int64_t Ic = P.c0; Ic < P.c1; Ic++)
                C.rowptr[(size_t)Ic + 1] = C.rowptr[(size_t)Ic] + P.rowlen[(size_t)(Ic - P.c0)];
        }
        C.cptr.push_back((int)C.cidx.size());
        H.push_back(std::move(C));
    }

    lap("coarse patterns (host)");
    g.levels.clear();
    g.levels.resize(H.size());
    g.stats.levels = (int)H.size();
    SellMap prev;
    for (size_t lev = 0; lev < H.size(); lev++) {
        Level &L = g.levels[lev];
        HostLevel &h = H[lev];
        SellMap M = sell_map(h.n, h.rowptr, h.col);
        L.n = h.n;
        L.nnz = h.rowptr[h.n];
        L.agg = h.agg;
        L.nsl = M.nsl;
        L.sell_len = M.len;
        L.max_near = 0;
        L.uni_w = M.nsl > 0 ? M.sl_off[1] - M.sl_off[0] : 0;
        for (int sl = 0; sl < M.nsl; sl++) {
            L.max_near = std::max(L.max_near, M.sl_near[sl]);
            if (M.sl_off[sl + 1] - M.sl_off[sl] != L.uni_w || M.sl_near[sl] != L.uni_w) L.uni_w = 0;
        }
        std::vector<int> scol((size_t)M.len);
        parallel_for(M.nsl, 128, [&](int64_t s0, int64_t s1, int) {
            for (int sl = (int)s0; sl < (int)s1; sl++)  // padding: a valid near column (row 0 of the slice), value 0
                for (int k = M.sl_off[sl]; k < M.sl_off[sl + 1]; k++)
                    for (int lane = 0; lane < 64; lane++) scol[(size_t)k * 64 + lane] = sl * 64;
        });
        parallel_for((int64_t)h.col.size(), 65536, [&](int64_t a, int64_t b, int) {
            for (int64_t t = a; t < b; t++) scol[M.pos[t]] = h.col[t];
        });
        L.sl_off.upload(M.sl_off, s);
        L.sl_near.upload(M.sl_near, s);
        L.col.upload(scol, s);
        L.val.alloc((size_t)M.len);
        L.val.zero(s);
        if (lev == 0) {
            g.l0_far_entries = 0;
            for (int sl = 0; sl < M.nsl; sl++) g.l0_far_entries += (M.sl_off[sl + 1] - M.sl_off[sl]) - M.sl_near[sl];
            // Coarse-correction scale. Piecewise-constant aggregates under-correct the smooth error
            // of a chain-like graph by about a factor two (the classical over-correction of
            // unsmoothed aggregation); long-range edges carry part of it themselves. Measured at
            // 100k views (PCG iterations per solve, kc = 1.0 / 1.6 / 2.0): band graphs of degree
            // 4..30: 75..27 / 61..20 / 57..19; with 0.2-10 % loop closures: 34..32 / 26..27 / 28..32.
            // The preconditioner stays SPD for every kc > 0 (omega < 1).
            g.kc_auto = g.opt.mg_kc <= 0;
            if (g.kc_auto) g.opt.mg_kc = g.l0_far_entries == 0 ? 2.0 : 1.6;
            std::vector<uint32_t> seid((size_t)M.len, 0xffffffffu);
            parallel_for((int64_t)slot_eid.size(), 65536, [&](int64_t a, int64_t b, int) {
                for (int64_t t = a; t < b; t++) seid[M.pos[t]] = slot_eid[t];
            });
            g.slot_eid.upload(seid, s);
            // k_assemble0w: per slice, the first edge of the run it stages in LDS = the lowest edge id
            // among the slice's near entries (a view sequence: the edges of the slice's first view)
            g.asm_windowed = getenv("IROTAVG_ASM_CLASSIC") ? 0 : 1;
            std::vector<int> te0((size_t)M.nsl, 0);
            if (g.asm_windowed) parallel_for(M.nsl, 256, [&](int64_t s0, int64_t s1, int) {
                for (int sl = (int)s0; sl < (int)s1; sl++) {
                    uint32_t lo = 0xffffffffu;
                    for (int r = sl * 64; r < std::min(h.n, sl * 64 + 64); r++)
                        for (int t = h.rowptr[r]; t < h.rowptr[r + 1]; t++)
                            if (h.col[t] >= r - kWinHalo && h.col[t] <= r + kWinHalo) lo = std::min(lo, slot_eid[t] >> 1);
                    te0[(size_t)sl] = lo == 0xffffffffu ? 0 : (int)lo;
                }
            });
            g.tile_e0.upload(te0, s);
            g.asm_l1_fused = 0;
            if (H.size() < 2) {  // no level 1: the kernel still reads a (dummy) level-1 index per pair
                std::vector<uint8_t> cs((size_t)M.len, 255);
                g.slot_cs.upload(cs, s);
            }
            IRH_CHECK(hipStreamSynchronize(s));
        }
        L.excess.alloc((size_t)M.nsl * 64);
        L.diag.alloc((size_t)M.nsl * 64);
        L.idg.alloc((size_t)M.nsl * 64);
        L.excess.zero(s);
        L.diag.zero(s);
        L.idg.zero(s);
        if (lev > 0) {
            std::vector<int> cidx2(h.cidx.size());
            parallel_for((int64_t)h.cidx.size(), 65536, [&](int64_t a, int64_t b, int) {
                for (int64_t q = a; q < b; q++) cidx2[q] = prev.pos[h.cidx[q]];
            });
            L.crow.upload(h.rowptr, s);
            L.max_row = 0;
            for (int r = 0; r < h.n; r++) L.max_row = std::max(L.max_row, h.rowptr[r + 1] - h.rowptr[r]);
            if (lev == 1 && g.asm_windowed) {
                // level-1 entry index of every level-0 SELL position (k_assemble0w sums level 1 on the way
                // when level 0 aggregates by 8 and no level-1 row has more than 8 entries)
                std::vector<uint8_t> cs((size_t)prev.len, 255);
                std::atomic<bool> wide(false);
                parallel_for((int64_t)h.rowptr[h.n], 16384, [&](int64_t a, int64_t b, int) {
                    for (int64_t c = a; c < b; c++) {
                        const int k = M.kidx[(size_t)c];
                        if (k >= 8) wide = true;
                        for (int q = h.cptr[(size_t)c]; q < h.cptr[(size_t)c + 1]; q++)
                            cs[(size_t)prev.pos[(size_t)h.cidx[(size_t)q]]] = (uint8_t)std::min(k, 255);
                    }
                });
                g.slot_cs.upload(cs, s);
                g.asm_l1_fused = (g.asm_windowed && H[0].agg == 8 && !wide) ? 1 : 0;
                IRH_CHECK(hipStreamSynchronize(s));
            }
            L.cptr.upload(h.cptr, s);
            L.cidx.upload(cidx2, s);
            L.cpos.upload(M.pos, s);
            IRH_CHECK(hipStreamSynchronize(s));
        }
        // vectors are padded to whole slices so that tail lanes may load harmlessly
        const size_t nv = (size_t)M.nsl * 64 + 64;
        L.b.alloc(nv);
        L.x.alloc(nv);
        L.y.alloc(nv);
        L.e.alloc(nv);
        L.b.zero(s);
        L.x.zero(s);
        L.y.zero(s);
        L.e.zero(s);
        if (lev < (size_t)kMaxLevels) {
            g.stats.level_rows[lev] = L.n;
            g.stats.level_nnz[lev] = L.nnz;
        }
        IRH_CHECK(hipStreamSynchronize(s));  // scol goes out of scope
        prev = std::move(M);
    }
    lap("SELL conversion + uploads");
    // what the common tail needs to know about the patterns
    BuildTail T;
    T.nlev = (int)H.size();
    for (size_t l = 0; l < H.size() && l < (size_t)kMaxLevels; l++) T.agg[l] = H[l].agg;
    {
        const HostLevel &hd = H.back();
        int bw = 0;
        for (int r = 0; r < hd.n; r++)
            for (int t = hd.rowptr[r]; t < hd.rowptr[r + 1]; t++) bw = std::max(bw, std::abs(hd.col[t] - r));
        T.dense_bw = bw;
    }
    if (H.size() >= 3) {
        bool ok = true;
        const HostLevel &h1 = H[1];
        for (int r = 0; r < h1.n && ok; r++) {
            const int lo = (r / 32) * 32 - 8, hi = (r / 32) * 32 + 40;
            for (int t = h1.rowptr[r]; t < h1.rowptr[r + 1]; t++)
                if (h1.col[t] < lo || h1.col[t] >= hi) {
                    ok = false;
                    break;
                }
        }
        bool band8 = ok;
        for (int r = 0; r < h1.n && band8; r++)
            for (int t = h1.rowptr[r]; t < h1.rowptr[r + 1]; t++)
                if (h1.col[t] < r - 8 || h1.col[t] > r + 8) {
                    band8 = false;
                    break;
                }
        T.l1_window_ok = ok;
        T.l1_band8 = band8;
    }
    const int rc_tail = finish_build(g, T);
    lap("PCG state");
    return rc_tail;
}

// The tail of a build, shared by the host path above and the device path (gbuild.hip): dense level,
// which PCG kernels this graph takes, PCG state.
int finish_build(Graph &g, const BuildTail &T) {
    hipStream_t s = g.stream;
    // dense inverse of the coarsest level (only when it is small enough and there is a hierarchy)
    g.ndense = 0;
    g.ndense_pad = 0;
    g.dense_bw = 0;
    if (g.levels.back().n <= std::min(g.opt.mg_dense_max, 2048) && g.opt.mg_levels_max > 1) {
        g.ndense = g.levels.back().n;
        g.ndense_pad = (g.ndense + 63) / 64 * 64;
        g.dense_inv.alloc((size_t)g.ndense_pad * g.ndense_pad);
        g.dense_inv.zero(s);
        g.dense_wr.alloc((size_t)64 * g.ndense_pad);  // two 32 x npad panels (look-ahead ping-pong)
        g.dense_wc.alloc((size_t)64 * g.ndense_pad);
        g.dense_ref_diag.alloc((size_t)g.ndense_pad);
        // half-bandwidth of the coarsest operator: a banded one (view sequences without loop closures: 1-3)
        // is inverted by a banded LDL' + one substitution per column instead of the dense sweep (dense.hip)
        g.dense_bw = T.dense_bw;
    }
    g.additive_top = g.opt.mg_multiplicative_top == 1 ? 0 : 1;
    // Can the PCG update also do the down-sweep of level 1 (k_pcg_update_restrict2)? Aggregates of 8
    // on levels 0 and 1, one GPU, and every level-1 neighbour of a tile's 32 level-1 rows inside
    // the 48-row window the update kernel holds in LDS.
    g.l1_fused = 0;
    g.cg2 = 0;
    if (g.additive_top && T.nlev >= 3 && T.agg[0] == 8 && T.agg[1] == 8 && g.ng == 0 &&
        g.l0_far_entries == 0 && g.opt.no_fused_pspmv != 1) {
        const bool ok = T.l1_window_ok;
        g.l1_fused = ok ? 1 : 0;
        // the two-launch iteration (cgcg.hip) also runs the level-1 UP-sweep inside a level-0 kernel:
        // the 48 level-1 rows under a tile window need all their neighbours within the 64 extended
        // rows, i.e. within 8 rows
        const bool band8 = ok && T.l1_band8;
        // ... and a workgroup per tile of <= 4 slices with at most kMaxParts workgroups (up to 131k views).
        // Beyond that the kernels are bandwidth-bound, not latency-bound, and the classic launches
        // measured faster (1M/20M: 1.88 vs 1.70 G)
        const bool one_tile = (g.levels[0].n + 63) / 64 <= 4 * kMaxParts;
        g.cg2 = (band8 && one_tile && g.ndense > 0 && g.opt.pcg_classic != 1) ? 1 : 0;
        g.dense32 = 0;  // tile slices of the coarse solve read the fp64 inverse; IROTAVG_CG2_FP32_DENSE=1: an fp32 copy
        if (const char *e = getenv("IROTAVG_CG2_FP32_DENSE")) g.dense32 = atoi(e) == 1 ? 1 : 0;
        if (g.cg2) {
            g.b2p.alloc((size_t)3 * g.ndense_pad);
            g.b2p.zero(s);
        }
    }
    // ---- PCG state ----------------------------------------------------------------------
    const size_t nv0 = (size_t)g.levels[0].nsl * 64 + 64;
    g.X.alloc(nv0 + (size_t)g.ng);
    g.P.alloc(nv0);
    g.P2.alloc(nv0);
    g.P2.zero(s);
    g.R2.alloc(nv0);
    g.R2.zero(s);
    g.AP.alloc(nv0);
    g.X.zero(s);
    g.P.zero(s);
    g.AP.zero(s);
    g.part_pq.alloc((size_t)kMaxParts * 4);
    g.part_rr.alloc((size_t)kMaxParts * 4);
    g.part_rz.alloc((size_t)kMaxParts * 4);
    g.part_rz2.alloc((size_t)kMaxParts * 4);
    g.part_rz2.zero(s);
    g.part_score.alloc((size_t)kMaxParts * 4);
    g.part_pq.zero(s);
    g.part_rr.zero(s);
    g.part_rz.zero(s);
    g.part_score.zero(s);
    alloc_state(g);
    IRH_CHECK(hipStreamSynchronize(s));  // host vectors of the callers go out of scope
    return IROTAVG_OK;
}

}  // namespace irh

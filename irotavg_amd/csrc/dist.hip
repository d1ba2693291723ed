// dist.hip -- vertex-range sharding of the IRLS solve across the GPUs of one node.
//
// The free views are cut into `world` contiguous ranges (multiples of 64 views). A shard owns its
// range's rows of the normal matrix and every edge with an endpoint in the range; cross-shard
// edges are held by both shards (their residuals and weights are computed redundantly and
// identically), the remote endpoint is a GHOST view: free, but without a local row. Locally the
// vertex order is [fixed | ghost | owned], so K1/K2/K6 run unchanged over X = [ghost | owned].
//
// Per PCG iteration: one halo exchange (the direction p of ghost views, 32 B each) and three small
// all-reduces (p.Lp; ||r||^2 with r.z0; b1.y1) of 3-8 doubles. Per IRLS iteration: one halo
// exchange of the solution X and one all-reduce of the score. The preconditioner is the
// single-GPU one built on the shard's own diagonal block (ghost couplings act as Dirichlet mass):
// block-Jacobi across shards, no communication inside it.
//
// Transport: RCCL over xGMI when each process holds one shard (ncclSend/ncclRecv groups for the
// halos, ncclAllReduce for the scalars, all on the shard's stream), or an in-process loopback when
// one process holds all shards on one GPU (used by the tests to verify the sharded algebra
// against the unsharded solve without multi-GPU hardware).
#include <rccl/rccl.h>

#include <algorithm>
#include <map>
#include <memory>

#include "graph.hpp"
#include "kernels.hpp"

namespace irh {

struct Shard {
    Graph g;
    int rank = 0;
    int64_t lo = 0, hi = 0;          // owned range in global free-view index space
    std::vector<int64_t> gvert;      // local vertex -> global vertex id
    std::vector<int64_t> gedge;      // local edge   -> global edge id
    std::vector<int> peers;          // ranks this shard exchanges halos with
    std::vector<int> send_off, send_cnt, recv_off, recv_cnt;
    DevBuf<int> send_idx;            // owned-local indices to pack, grouped by peer
    DevBuf<double4> sendbuf;
    DevBuf<int> gather_dst;          // gathered halo (Dist::halo_gather): slot of every send entry inside this rank's record
    DevBuf<double> gsum;             // 16 doubles: staging of the all-reduced scalars
    DevBuf<uint8_t> eown;            // per local edge: 1 = this shard counts it in sums over edges (L1RA)
    DevBuf<double4> rf_x, rf_b;      // the refinement of a direct solve with closures: the solution so far, the right-hand side
    int send_total = 0;
};

struct Dist {
    int world = 1;
    int64_t m = 0, n_total = 0, nu = 0, chunk = 0;
    int f = 0;
    std::vector<std::unique_ptr<Shard>> shards;  // local shards (all of them in loopback mode)
    bool use_rccl = false;
    ncclComm_t comm = nullptr;
    bool hosted = false;  // wire = the caller's host-staged transport (irotavg_dist_create_hosted)
    irotavg_transport tr{};
    std::vector<double> hbuf, hbuf2;  // host staging of the hosted transport
    hipStream_t stream = nullptr;
    irotavg_options opt{};
    irotavg_stats stats{};
    // A view sequence (round 5: also with up to 2048 loop closures, cl_edge below) is solved DIRECTLY also when it is
    // sharded (bcr.hip): every rank
    // reduces its range of the banded operator to its last block, ONE gather of the `world` separators replaces
    // the ~22 halo exchanges + all-reduces of a PCG solve, every rank solves the separator system and walks back.
    int bcr_B = 0;  // block size; 0: the sharded PCG
    int band0 = -1;
    BcrTop top;
    // ... and (round 5) also with loop closures -- the sequence the reference's SLAM front end produces
    // (src/IRotAvg.cpp:371-378): the long-range edges whose blocks are not neighbours, in edge order; every process
    // derives the same list from the global graph. Woodbury correction across the ranks: bcr.hip, "loop closures on a
    // sharded sequence".
    std::vector<int64_t> cl_edge;
    // The halo of a sharded SEQUENCE without closures (round 6): a rank's ghosts are its two neighbours' boundary views, at
    // most `band` <= bcr_B of them per side -- so the exchange is ONE all-gather of a fixed record per rank, [to the lower
    // neighbour | to the upper neighbour], halo_w slots each (1.5 KB per rank at B = 24), instead of a group of two
    // ncclSend + two ncclRecv of ~0.5 KB: the same shape as the separators' gather, one collective on the wire per halo.
    // Loopback: the shards write their records into the one buffer. The hosted wire keeps its point-to-point round.
    bool halo_gather = false;
    int halo_w = 0;
    DevBuf<double4> hall;  // world records of 2 halo_w views
    // Diagnostic of a first multi-GPU run (irotavg_dist_timing; OFF in every timed region): wall time per phase of an
    // IRLS iteration, the stream drained at every phase boundary -- which adds to the total, so the phases are to be
    // read against each other, not against the undisturbed step. Phases: kDistPhases below.
    bool timing = false;
    double t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double t_mark = 0.0;
    int64_t t_iters = 0;
};
// 0 local edge kernels + assembly | 1 local reductions (closures' forward eliminations; the whole solve on the sharded
// PCG) | 2 gather of the separators | 3 closure sum | 4 separator system, corrections, ways back | 5 halo of the step |
// 6 weights + rotation update | 7 score all-reduce
static inline void tmark(Dist &D, int phase) {
    if (!D.timing) return;
    (void)hipStreamSynchronize(D.stream);
    const double t = now_seconds();
    if (phase >= 0 && phase < 8) D.t_acc[phase] += t - D.t_mark;
    D.t_mark = t;
}

#define NCCL_CHECK(expr)                                                                      \
    do {                                                                                      \
        ncclResult_t _r = (expr);                                                             \
        if (_r != ncclSuccess) {                                                              \
            std::fprintf(stderr, "[irotavg_hip] %s failed: %s (%s:%d)\n", #expr,              \
                         ncclGetErrorString(_r), __FILE__, __LINE__);                         \
            throw ::irh::HipError{hipErrorUnknown};                                           \
        }                                                                                     \
    } while (0)

// ---- small kernels ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack(int cnt, const int *__restrict__ idx,
                                              const double4 *__restrict__ src,
                                              double4 *__restrict__ dst) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < cnt) dst[t] = src[idx[t]];
}

// the gathered halo: a send entry goes to its slot of the rank's record; the two neighbours' segments come out of theirs
__global__ __launch_bounds__(256) void k_pack_to(int cnt, const int *__restrict__ idx, const int *__restrict__ dstslot,
                                                 const double4 *__restrict__ src, double4 *__restrict__ rec) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < cnt) rec[dstslot[t]] = src[idx[t]];
}
__global__ __launch_bounds__(256) void k_unpack_from(const double4 *__restrict__ seg0, int cnt0, double4 *__restrict__ dst0,
                                                     const double4 *__restrict__ seg1, int cnt1, double4 *__restrict__ dst1) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < cnt0) dst0[t] = seg0[t];
    if (t < cnt1) dst1[t] = seg1[t];
}

// fixed-order reduction of up to two partial arrays IN PLACE: row 0 of each array receives the
// shard's sum (load_reduced3 ends with a barrier, so every row has been read before row 0 is
// overwritten); the all-reduce then runs directly on row 0 and the consumers read it with
// nparts = 1 -- no staging copies.
__global__ __launch_bounds__(256) void k_reduce_parts(double *__restrict__ pa, int na,
                                                      double *__restrict__ pb, int nb,
                                                      double *__restrict__ pc, int nc) {
    double a[3], b[3] = {0, 0, 0}, c3[3] = {0, 0, 0};
    load_reduced3(pa, na, a);
    if (pb) load_reduced3(pb, nb, b);
    if (pc) load_reduced3(pc, nc, c3);
    if (threadIdx.x == 0) {
        for (int c = 0; c < 3; c++) {
            pa[c] = a[c];
            if (pb) pb[c] = b[c];
            if (pc) pc[c] = c3[c];
        }
        pa[3] = 0.0;
        if (pb) pb[3] = 0.0;
        if (pc) pc[3] = 0.0;
    }
}

__global__ void k_add_small(int n, double *__restrict__ dst, const double *__restrict__ src) {
    const int t = threadIdx.x;
    if (t < n) dst[t] += src[t];
}

// ---- collectives ------------------------------------------------------------------------------
// sum over all shards of the 4-double rows `pa(shard)` (and `pb(shard)` when given), in place
typedef double *(*RowOf)(Shard &);
static void hosted_allreduce_rows(Dist &D, double *const *rows, int nrows) {
    // device rows -> host -> the caller's all-reduce -> device (one process = one shard here)
    D.hbuf.assign((size_t)4 * nrows, 0.0);
    for (int w = 0; w < nrows; w++)
        IRH_CHECK(hipMemcpyAsync(D.hbuf.data() + 4 * w, rows[w], sizeof(double) * 4, hipMemcpyDeviceToHost, D.stream));
    IRH_CHECK(hipStreamSynchronize(D.stream));
    if (D.tr.allreduce(D.tr.ctx, D.hbuf.data(), 4 * nrows, 0) != 0) throw HipError{hipErrorUnknown};
    for (int w = 0; w < nrows; w++)
        IRH_CHECK(hipMemcpyAsync(rows[w], D.hbuf.data() + 4 * w, sizeof(double) * 4, hipMemcpyHostToDevice, D.stream));
    IRH_CHECK(hipStreamSynchronize(D.stream));  // hbuf is reused by the next call
}

static void allreduce_rows(Dist &D, const RowOf *rows, int nrows) {
    if (D.hosted) {
        double *r[8];
        for (int w = 0; w < nrows; w++) r[w] = rows[w](*D.shards[0]);
        hosted_allreduce_rows(D, r, nrows);
        return;
    }
    if (D.use_rccl) {  // one group = one fused launch for the nrows x 4 doubles
        Shard &S = *D.shards[0];
        NCCL_CHECK(ncclGroupStart());
        for (int w = 0; w < nrows; w++)
            NCCL_CHECK(ncclAllReduce(rows[w](S), rows[w](S), 4, ncclDouble, ncclSum, D.comm, D.stream));
        NCCL_CHECK(ncclGroupEnd());
        return;
    }
    if (D.shards.size() == 1) return;
    Shard &S0 = *D.shards[0];
    for (int w = 0; w < nrows; w++) {
        double *d0 = rows[w](S0);
        for (size_t s = 1; s < D.shards.size(); s++)  // fixed shard order
            hipLaunchKernelGGL(k_add_small, dim3(1), dim3(64), 0, D.stream, 4, d0, rows[w](*D.shards[s]));
        for (size_t s = 1; s < D.shards.size(); s++)
            IRH_CHECK(hipMemcpyAsync(rows[w](*D.shards[s]), d0, sizeof(double) * 4, hipMemcpyDeviceToDevice,
                                     D.stream));
    }
}
template <typename FA, typename FB>
static void allreduce_rows(Dist &D, FA pa, FB pb, bool two) {
    if (D.hosted) {
        double *r[2] = {pa(*D.shards[0]), pb(*D.shards[0])};
        hosted_allreduce_rows(D, r, two ? 2 : 1);
        return;
    }
    if (D.use_rccl) {
        Shard &S = *D.shards[0];
        NCCL_CHECK(ncclGroupStart());
        NCCL_CHECK(ncclAllReduce(pa(S), pa(S), 4, ncclDouble, ncclSum, D.comm, D.stream));
        if (two) NCCL_CHECK(ncclAllReduce(pb(S), pb(S), 4, ncclDouble, ncclSum, D.comm, D.stream));
        NCCL_CHECK(ncclGroupEnd());
        return;
    }
    if (D.shards.size() == 1) return;
    Shard &S0 = *D.shards[0];
    for (int w = 0; w < (two ? 2 : 1); w++) {
        double *d0 = w == 0 ? pa(S0) : pb(S0);
        for (size_t s = 1; s < D.shards.size(); s++)  // fixed shard order
            hipLaunchKernelGGL(k_add_small, dim3(1), dim3(64), 0, D.stream, 4, d0,
                               w == 0 ? pa(*D.shards[s]) : pb(*D.shards[s]));
        for (size_t s = 1; s < D.shards.size(); s++)
            IRH_CHECK(hipMemcpyAsync(w == 0 ? pa(*D.shards[s]) : pb(*D.shards[s]), d0, sizeof(double) * 4,
                                     hipMemcpyDeviceToDevice, D.stream));
    }
}

// halo exchange: owned values `src_of(shard)` -> ghost slots `dst_of(peer shard)`
enum HaloWhat { HALO_P, HALO_X };
static const double4 *halo_src(Shard &S, HaloWhat w) {
    return w == HALO_P ? S.g.P.p : S.g.X.p + S.g.ng;
}
static double4 *halo_dst(Shard &S, HaloWhat w) { return w == HALO_P ? S.g.PG.p : S.g.X.p; }

static void halo_exchange(Dist &D, HaloWhat what) {
    if (D.halo_gather) {
        const int rec = 2 * D.halo_w;
        for (auto &sp : D.shards) {
            Shard &S = *sp;
            if (S.send_total > 0)
                hipLaunchKernelGGL(k_pack_to, dim3((S.send_total + 255) / 256), dim3(256), 0, D.stream, S.send_total,
                                   S.send_idx.p, S.gather_dst.p, halo_src(S, what), D.hall.p + (size_t)S.rank * rec);
        }
        if (D.use_rccl) {
            const int rank = D.shards[0]->rank;
            NCCL_CHECK(ncclAllGather(D.hall.p + (size_t)rank * rec, D.hall.p, (size_t)rec * 4, ncclDouble, D.comm, D.stream));
        }  // (loopback: the shards of this process share the buffer)
        for (auto &sp : D.shards) {
            Shard &S = *sp;
            const double4 *seg[2] = {nullptr, nullptr};
            double4 *dst[2] = {nullptr, nullptr};
            int cnt[2] = {0, 0};
            for (size_t q = 0; q < S.peers.size() && q < 2; q++) {
                const int peer = S.peers[q];
                // the lower neighbour's segment for me is its UPPER half, the upper neighbour's its LOWER half
                seg[q] = D.hall.p + (size_t)peer * rec + (peer < S.rank ? D.halo_w : 0);
                dst[q] = halo_dst(S, what) + S.recv_off[q];
                cnt[q] = S.recv_cnt[q];
            }
            const int mx = std::max(cnt[0], cnt[1]);
            if (mx > 0)
                hipLaunchKernelGGL(k_unpack_from, dim3((mx + 255) / 256), dim3(256), 0, D.stream, seg[0], cnt[0], dst[0], seg[1],
                                   cnt[1], dst[1]);
        }
        return;
    }
    for (auto &sp : D.shards) {
        Shard &S = *sp;
        if (S.send_total > 0)
            hipLaunchKernelGGL(k_pack, dim3((S.send_total + 255) / 256), dim3(256), 0, D.stream,
                               S.send_total, S.send_idx.p, halo_src(S, what), S.sendbuf.p);
    }
    if (D.hosted) {
        Shard &S = *D.shards[0];
        const int np = (int)S.peers.size();
        if (np == 0) return;
        std::vector<int64_t> so(np), sc(np), ro(np), rc(np);
        int64_t rtot = 0;
        for (int q = 0; q < np; q++) {  // counts in doubles (4 per view)
            so[q] = 4 * (int64_t)S.send_off[q];
            sc[q] = 4 * (int64_t)S.send_cnt[q];
            ro[q] = 4 * (int64_t)S.recv_off[q];
            rc[q] = 4 * (int64_t)S.recv_cnt[q];
            rtot = std::max(rtot, ro[q] + rc[q]);
        }
        D.hbuf.assign((size_t)4 * S.send_total + 4, 0.0);
        D.hbuf2.assign((size_t)rtot + 4, 0.0);
        if (S.send_total > 0)
            IRH_CHECK(hipMemcpyAsync(D.hbuf.data(), S.sendbuf.p, sizeof(double4) * (size_t)S.send_total,
                                     hipMemcpyDeviceToHost, D.stream));
        IRH_CHECK(hipStreamSynchronize(D.stream));
        if (D.tr.exchange(D.tr.ctx, np, S.peers.data(), D.hbuf.data(), so.data(), sc.data(), D.hbuf2.data(),
                          ro.data(), rc.data()) != 0)
            throw HipError{hipErrorUnknown};
        for (int q = 0; q < np; q++)
            if (rc[q] > 0)
                IRH_CHECK(hipMemcpyAsync(halo_dst(S, what) + S.recv_off[q], D.hbuf2.data() + ro[q],
                                         sizeof(double) * (size_t)rc[q], hipMemcpyHostToDevice, D.stream));
        IRH_CHECK(hipStreamSynchronize(D.stream));
        return;
    }
    if (D.use_rccl) {
        Shard &S = *D.shards[0];
        NCCL_CHECK(ncclGroupStart());
        for (size_t q = 0; q < S.peers.size(); q++) {
            if (S.send_cnt[q] > 0)
                NCCL_CHECK(ncclSend(S.sendbuf.p + S.send_off[q], (size_t)S.send_cnt[q] * 4, ncclDouble,
                                    S.peers[q], D.comm, D.stream));
            if (S.recv_cnt[q] > 0)
                NCCL_CHECK(ncclRecv(halo_dst(S, what) + S.recv_off[q], (size_t)S.recv_cnt[q] * 4,
                                    ncclDouble, S.peers[q], D.comm, D.stream));
        }
        NCCL_CHECK(ncclGroupEnd());
        return;
    }
    for (auto &sp : D.shards) {  // loopback: copy each send segment into the peer's ghost slots
        Shard &S = *sp;
        for (size_t q = 0; q < S.peers.size(); q++) {
            Shard &T = *D.shards[S.peers[q]];
            // T's receive slot for data coming from S
            size_t r = 0;
            while (r < T.peers.size() && T.peers[r] != S.rank) r++;
            if (r == T.peers.size() || T.recv_cnt[r] != S.send_cnt[q]) throw HipError{hipErrorUnknown};
            IRH_CHECK(hipMemcpyAsync(halo_dst(T, what) + T.recv_off[r], S.sendbuf.p + S.send_off[q],
                                     sizeof(double4) * (size_t)S.send_cnt[q], hipMemcpyDeviceToDevice,
                                     D.stream));
        }
    }
}

// ---- partition (pure host) ---------------------------------------------------------------------
struct ShardPlan {
    int rank = 0;
    int64_t lo = 0, hi = 0, chunk = 0;
    std::vector<int64_t> ledge;        // global edge ids of the shard's edges (ascending)
    std::vector<int> fixed_used;       // global ids of the fixed views its edges touch (ascending)
    std::vector<int> ghosts;           // global ids of its ghost views (ascending => grouped by owner)
    std::vector<int> peers, send_off, send_cnt, recv_off, recv_cnt;
    std::vector<int> send_idx;         // owned-local indices to pack, grouped by peer, ascending
};

// shard size: a multiple of 64 views (slices / tiles of the row kernels); for the sharded direct solver a multiple of
// 192 = of 64 and of every block size (8, 16, 24, 32: a rank's last block must be a full one for its neighbour's
// coupling to fit it)
static int64_t chunk_of(int64_t nu, int world, int align = 64) {
    return ((nu + world - 1) / world + align - 1) / align * align;
}

static int plan_shard(int world, int rank, int64_t m, int64_t n_total, int f, const int32_t *I,
                      ShardPlan &P, int align = 64) {
    const int64_t nu = n_total - f;
    P.rank = rank;
    P.chunk = chunk_of(nu, world, align);
    if ((int64_t)(world - 1) * P.chunk >= nu) return IROTAVG_ERR_BAD_ARG;  // a shard would be empty
    P.lo = (int64_t)rank * P.chunk;
    P.hi = rank == world - 1 ? nu : (int64_t)(rank + 1) * P.chunk;
    auto owner = [&](int64_t gfree) { return (int)std::min<int64_t>(gfree / P.chunk, world - 1); };
    auto owned = [&](int v) { return v >= f && v - f >= P.lo && v - f < P.hi; };
    std::map<int, std::vector<int>> send_to;  // peer -> my owned-local indices
    for (int64_t k = 0; k < m; k++) {
        const int i = I[2 * k], j = I[2 * k + 1];
        if (i < 0 || j < 0 || i >= n_total || j >= n_total) return IROTAVG_ERR_BAD_ARG;
        // local edges: any endpoint owned; edges without a free endpoint go to rank 0
        const bool mine = owned(i) || owned(j) || (rank == 0 && i < f && j < f);
        if (!mine) continue;
        P.ledge.push_back(k);
        for (int v : {i, j}) {
            if (v < f)
                P.fixed_used.push_back(v);
            else if (!owned(v))
                P.ghosts.push_back(v);
        }
        if (owned(i) && j >= f && !owned(j)) send_to[owner(j - f)].push_back((int)(i - f - P.lo));
        if (owned(j) && i >= f && !owned(i)) send_to[owner(i - f)].push_back((int)(j - f - P.lo));
    }
    std::sort(P.fixed_used.begin(), P.fixed_used.end());
    P.fixed_used.erase(std::unique(P.fixed_used.begin(), P.fixed_used.end()), P.fixed_used.end());
    std::sort(P.ghosts.begin(), P.ghosts.end());
    P.ghosts.erase(std::unique(P.ghosts.begin(), P.ghosts.end()), P.ghosts.end());
    // Halo plan. Ghosts are sorted by global id, hence grouped by owner. What I send to peer h is
    // the set of my views adjacent to h's views, sorted by global id -- exactly h's ghost list
    // for me, so the two sides agree without a handshake.
    std::map<int, std::pair<int, int>> recv_from;  // peer -> (offset, count) in the ghost array
    for (int q = 0; q < (int)P.ghosts.size(); q++) {
        const int h = owner(P.ghosts[q] - f);
        auto it = recv_from.find(h);
        if (it == recv_from.end())
            recv_from[h] = {q, 1};
        else
            it->second.second++;
    }
    std::vector<int> all_peers;
    for (auto &kv : send_to) {
        auto &v = kv.second;
        std::sort(v.begin(), v.end());
        v.erase(std::unique(v.begin(), v.end()), v.end());
        all_peers.push_back(kv.first);
    }
    for (auto &kv : recv_from) all_peers.push_back(kv.first);
    std::sort(all_peers.begin(), all_peers.end());
    all_peers.erase(std::unique(all_peers.begin(), all_peers.end()), all_peers.end());
    for (int h : all_peers) {
        P.peers.push_back(h);
        P.send_off.push_back((int)P.send_idx.size());
        auto it = send_to.find(h);
        const int sc = it == send_to.end() ? 0 : (int)it->second.size();
        if (sc) P.send_idx.insert(P.send_idx.end(), it->second.begin(), it->second.end());
        P.send_cnt.push_back(sc);
        auto ir = recv_from.find(h);
        P.recv_off.push_back(ir == recv_from.end() ? 0 : ir->second.first);
        P.recv_cnt.push_back(ir == recv_from.end() ? 0 : ir->second.second);
    }
    return IROTAVG_OK;
}

static int build_shard(Dist &D, Shard &S, const int32_t *I, const double *QQ, int64_t ldqq) {
    const int f = D.f;
    ShardPlan P;
    int rc = plan_shard(D.world, S.rank, D.m, D.n_total, f, I, P, D.bcr_B ? 192 : 64);
    if (rc != IROTAVG_OK) return rc;
    S.lo = P.lo;
    S.hi = P.hi;
    const int64_t no = S.hi - S.lo;
    const int ft = (int)P.fixed_used.size(), ng = (int)P.ghosts.size();
    std::map<int, int> lid;
    S.gvert.clear();
    for (int v : P.fixed_used) {
        lid[v] = (int)S.gvert.size();
        S.gvert.push_back(v);
    }
    for (int v : P.ghosts) {
        lid[v] = (int)S.gvert.size();
        S.gvert.push_back(v);
    }
    for (int64_t q = 0; q < no; q++) S.gvert.push_back(f + S.lo + q);
    auto owned = [&](int v) { return v >= f && v - f >= S.lo && v - f < S.hi; };
    auto local_id = [&](int v) { return owned(v) ? (int)(ft + ng + (v - f - S.lo)) : lid[v]; };
    const int64_t ml = (int64_t)P.ledge.size();
    if (ml == 0) return IROTAVG_ERR_BAD_ARG;
    std::vector<int32_t> Il((size_t)2 * ml);
    std::vector<double> QQl((size_t)4 * ml);
    for (int64_t t = 0; t < ml; t++) {
        const int64_t k = P.ledge[t];
        Il[2 * t] = local_id(I[2 * k]);
        Il[2 * t + 1] = local_id(I[2 * k + 1]);
        for (int c = 0; c < 4; c++) QQl[(size_t)c * ml + t] = QQ[(size_t)c * ldqq + k];
    }
    S.gedge = P.ledge;
    {   // who sums a cross-shard edge (L1RA reduces over edges): the shard that owns its j endpoint; a
        // fixed j: the owner of i; both fixed: rank 0 (the only shard holding such an edge)
        std::vector<uint8_t> own((size_t)ml, 0);
        for (int64_t t = 0; t < ml; t++) {
            const int64_t k = P.ledge[t];
            const int i = I[2 * k], j = I[2 * k + 1];
            own[t] = (j >= f ? owned(j) : (i >= f ? owned(i) : true)) ? 1 : 0;
        }
        S.eown.upload(own, D.stream);
        IRH_CHECK(hipStreamSynchronize(D.stream));
    }
    Graph &g = S.g;
    g.opt = D.opt;
    g.opt.no_fused_pspmv = 1;  // the sharded PCG exchanges p between its p-update and its SpMV
    g.stream = D.stream;
    g.m = ml;
    g.n_total = ft + ng + no;
    g.f = ft;
    g.ng = ng;
    g.no = (int)no;
    g.nu = ng + (int)no;
    g.force_np = 1;
    if (D.bcr_B) {
        // solved directly: level 0 is all the handle needs; the ghosts owned by the rank before this one are rows of
        // that rank's LAST block (its range is a multiple of the block size, the band is at most a block)
        g.opt.mg_levels_max = 1;
        g.bcr_B = D.bcr_B;
        g.band0 = D.band0;
        g.bcr_shard = true;
        g.bcr_ext0 = S.rank > 0 ? 1 : 0;
        g.bcr_ghost_extcol.assign((size_t)std::max(ng, 1), -1);
        const int64_t plo = S.lo - D.chunk;  // the previous rank's range is [plo, S.lo)
        for (int q = 0; q < ng; q++) {
            const int64_t gfree = (int64_t)P.ghosts[q] - f;
            if (S.rank > 0 && gfree >= plo && gfree < S.lo) {
                const int64_t ec = (gfree - plo) - (D.chunk - D.bcr_B);
                g.bcr_ghost_extcol[(size_t)q] = ec >= 0 ? (int)ec : -1;
            }
        }
        // the closures with an endpoint in this range: local rows (-1: a row of another rank), the local edge, the number
        // in the global list, whether this rank adds 1 / w (the owner of the first endpoint), and the rows whose diagonal
        // holds the weight of a closure to a ghost view
        std::map<int, std::vector<int>> fix;  // local row -> local edges
        auto owner_of = [&](int64_t gfree) { return (int)std::min<int64_t>(gfree / D.chunk, D.world - 1); };
        for (size_t c = 0; c < D.cl_edge.size(); c++) {
            const int64_t k = D.cl_edge[c];
            const int i = I[2 * k], j = I[2 * k + 1];
            const bool oi = owned(i), oj = owned(j);
            if (!oi && !oj) continue;
            const auto it = std::lower_bound(P.ledge.begin(), P.ledge.end(), k);
            if (it == P.ledge.end() || *it != k) return IROTAVG_ERR_BAD_ARG;
            const int e = (int)(it - P.ledge.begin());
            const int li = oi ? (int)(i - f - S.lo) : -1, lj = oj ? (int)(j - f - S.lo) : -1;
            g.bcr_far_i.push_back(li);
            g.bcr_far_j.push_back(lj);
            g.bcr_far_e.push_back(e);
            g.bcr_far_gid.push_back((int)c);
            g.bcr_far_own.push_back(owner_of((int64_t)i - f) == S.rank ? 1 : 0);
            if (oi && !oj) fix[li].push_back(e);
            if (oj && !oi) fix[lj].push_back(e);
        }
        g.bcr_fix_off.assign(1, 0);
        for (auto &kv : fix) {
            g.bcr_fix_row.push_back(kv.first);
            for (int e : kv.second) g.bcr_fix_e.push_back(e);
            g.bcr_fix_off.push_back((int)g.bcr_fix_e.size());
        }
    }
    rc = build_graph(g, Il.data(), QQl.data(), ml);
    if (rc != IROTAVG_OK) return rc;
    S.peers = P.peers;
    S.send_off = P.send_off;
    S.send_cnt = P.send_cnt;
    S.recv_off = P.recv_off;
    S.recv_cnt = P.recv_cnt;
    S.send_total = (int)P.send_idx.size();
    S.send_idx.upload(P.send_idx, D.stream);
    S.sendbuf.alloc((size_t)S.send_total + 1);
    S.gsum.alloc(16);
    S.gsum.zero(D.stream);
    IRH_CHECK(hipStreamSynchronize(D.stream));
    return IROTAVG_OK;
}

// ---- sharded PCG ----------------------------------------------------------------------------------
static void reduce_pair(Dist &D, double *pa, int na, double *pb, int nb, double *pc = nullptr, int nc = 0) {
    hipLaunchKernelGGL(k_reduce_parts, dim3(1), dim3(256), 0, D.stream, pa, na, pb, nb, pc, nc);
}

static int pcg_dist(Dist &D) {
    const double rtol2 = D.opt.pcg_rtol * D.opt.pcg_rtol;
    auto rows_grid = [](Graph &g) { return grid_for_rows(g.levels[0]); };
    auto upd_parts = [](Graph &g) {  // grid of the update kernel = producer count of part_rr / part_rz
        return (g.additive_top && g.levels.size() > 1) ? grid_for_rows(g.levels[0])
                                                        : grid_for_elems(g.levels[0].n);
    };
    auto additive = [](Graph &g) { return g.additive_top && g.levels.size() > 1; };
    auto p_rr = [](Shard &S) { return S.g.part_rr.p; };
    auto p_rz = [](Shard &S) { return S.g.part_rz.p; };
    auto p_pq = [](Shard &S) { return S.g.part_pq.p; };
    auto p_prec = [&](Shard &S) { return additive(S.g) ? S.g.part_rz2.p : S.g.part_rz.p; };
    // init: r = b, x = 0, first restriction
    for (auto &sp : D.shards) {
        Graph &g = sp->g;
        IRH_CHECK(hipMemsetAsync(g.flags.p, 0, sizeof(int) * FL_COUNT, D.stream));
        launch_update(g, true, 0, 1);
        if (!(g.additive_top && g.levels.size() > 1))
            reduce_pair(D, g.part_rr.p, upd_parts(g), g.part_rz.p, upd_parts(g));
    }
    if (!additive(D.shards[0]->g)) allreduce_rows(D, p_rr, p_rz, true);
    int it = 0;
    int h_flags[FL_COUNT] = {0, 0, 0, 0};
    const int check = std::max(1, D.opt.pcg_check_every);
    const int maxit = std::max(1, D.opt.pcg_max_iters);
    // Additive top level (the default): ||r||^2 and the Jacobi part of r.z come out of the update
    // kernel, the coarse part of r.z out of the (shard-local) preconditioner -- all three travel in
    // ONE all-reduce after the preconditioner, and convergence is tested in the p-update that
    // consumes them. Two all-reduces per iteration (this one and p.Lp). Multiplicative top level:
    // the older three-stage flow (||r||^2 is needed by the preconditioner's first kernel).
    const bool merged = additive(D.shards[0]->g);
    const RowOf rows3[3] = {+[](Shard &S) { return S.g.part_rr.p; }, +[](Shard &S) { return S.g.part_rz.p; },
                            +[](Shard &S) { return S.g.part_rz2.p; }};
    auto prec_all = [&]() {
        for (auto &sp : D.shards) {
            Graph &g = sp->g;
            PrecInfo pi = precondition(g, it == 0, rtol2, !merged);
            if (merged)
                reduce_pair(D, g.part_rr.p, it == 0 ? upd_parts(g) : upd_parts(g), g.part_rz.p, upd_parts(g),
                            g.part_rz2.p, pi.np_rz2);
            else if (additive(g))
                reduce_pair(D, g.part_rz2.p, pi.np_rz2, nullptr, 0);
            else
                reduce_pair(D, g.part_rz.p, pi.np_rz, nullptr, 0);
        }
        if (merged)
            allreduce_rows(D, rows3, 3);
        else
            allreduce_rows(D, p_prec, p_prec, false);
    };
    auto pupdate_all = [&]() {  // merged flow: carries the convergence test
        const int first = (it == 0), par = it & 1;
        PrecInfo one;
        one.np_rz = 1;
        one.np_rz2 = 1;
        for (auto &sp : D.shards) launch_pupdate(sp->g, par, first, one, merged, 1, rtol2);
    };
    auto tail_all = [&]() {
        const int par = it & 1;
        halo_exchange(D, HALO_P);
        for (auto &sp : D.shards) {
            Graph &g = sp->g;
            launch_spmv(g);
            reduce_pair(D, g.part_pq.p, rows_grid(g), nullptr, 0);
        }
        allreduce_rows(D, p_pq, p_pq, false);
        for (auto &sp : D.shards) {
            Graph &g = sp->g;
            launch_update(g, false, par ^ 1, 1);
            if (!merged) reduce_pair(D, g.part_rr.p, upd_parts(g), g.part_rz.p, upd_parts(g));
        }
        if (!merged) allreduce_rows(D, p_rr, p_rz, true);
        it++;
    };
    int chunk = D.stats.pcg_iters_last > 2 ? (int)std::min<int64_t>(D.stats.pcg_iters_last, maxit) : check;
    while (true) {
        for (int c = 0; c < chunk; c++) {
            prec_all();
            pupdate_all();
            tail_all();
        }
        chunk = std::max(2, check / 2);
        prec_all();
        if (merged) pupdate_all();  // the convergence test of the merged flow lives in the p-update
        IRH_CHECK(hipMemcpyAsync(h_flags, D.shards[0]->g.flags.p, sizeof(int) * FL_COUNT,
                                 hipMemcpyDeviceToHost, D.stream));
        IRH_CHECK(hipStreamSynchronize(D.stream));
        if (h_flags[FL_DONE] != 0) break;
        if (it >= maxit) break;
        if (!merged) pupdate_all();
        tail_all();
    }
    D.stats.pcg_solves += 1;
    D.stats.pcg_iters += h_flags[FL_ITERS];
    D.stats.pcg_iters_last = h_flags[FL_ITERS];
    if (h_flags[FL_DONE] == 2) return IROTAVG_ERR_SOLVER;
    if (h_flags[FL_DONE] == 0) return IROTAVG_ERR_NOT_CONVERGED;
    return IROTAVG_OK;
}

// n host doubles combined over all processes (op 0 sum, 1 min, 2 max); nothing to do when every
// shard lives in this process (loopback)
static void combine_host(Dist &D, double *v, int n, int op) {
    if (D.hosted) {
        if (D.tr.allreduce(D.tr.ctx, v, n, op) != 0) throw HipError{hipErrorUnknown};
    } else if (D.use_rccl) {
        Shard &S = *D.shards[0];
        if (n > 16) throw HipError{hipErrorUnknown};
        IRH_CHECK(hipMemcpyAsync(S.gsum.p, v, sizeof(double) * n, hipMemcpyHostToDevice, D.stream));
        NCCL_CHECK(ncclAllReduce(S.gsum.p, S.gsum.p, n, ncclDouble, op == 0 ? ncclSum : (op == 1 ? ncclMin : ncclMax),
                                 D.comm, D.stream));
        IRH_CHECK(hipMemcpyAsync(v, S.gsum.p, sizeof(double) * n, hipMemcpyDeviceToHost, D.stream));
        IRH_CHECK(hipStreamSynchronize(D.stream));
    }
}

// Sharded PCG with Chronopoulos-Gear recurrences (additive top level, >= 2 levels): per iteration ONE
// all-reduce group ({||r||^2, r.u, u.w}: 3 rows of 4 doubles) and ONE halo exchange (of u). The
// preconditioner is shard-local (block-Jacobi across shards) as before.
//   u = M^-1 r | halo(u) | w = L u, partials | all-reduce | convergence test on ||r||^2; alpha, beta;
//   p = u + beta p, s = w + beta s, x += alpha p, r -= alpha s, restriction
static int pcg_dist_cg(Dist &D) {
    const double rtol2 = D.opt.pcg_rtol * D.opt.pcg_rtol;
    auto rows_grid = [](Graph &g) { return grid_for_rows(g.levels[0]); };
    // ||r||^2 partials ping-pong between part_rr and part_rz (the update kernel tests one and writes the
    // other); gamma partials live in part_score (free during a solve), delta in part_pq
    auto rr_buf = [](Graph &g, int which) { return which ? g.part_rz.p : g.part_rr.p; };
    int cur = 0;  // which buffer holds ||r||^2 of the current residual
    for (auto &sp : D.shards) {
        Graph &g = sp->g;
        IRH_CHECK(hipMemsetAsync(g.flags.p, 0, sizeof(int) * FL_COUNT, D.stream));
        // the right-hand side is kept in levels[0].x (unused with an additive top level): a solve whose
        // recurrences stall is handed to the classic ones from there (see below)
        IRH_CHECK(hipMemcpyAsync(g.levels[0].x.p, g.levels[0].b.p, sizeof(double4) * (size_t)g.levels[0].n,
                                 hipMemcpyDeviceToDevice, D.stream));
        launch_cgd_update(g, true, 0, 1, rtol2, nullptr, nullptr, nullptr, rr_buf(g, cur));
    }
    int it = 0;
    int h_flags[FL_COUNT] = {0, 0, 0, 0};
    const int check = std::max(1, D.opt.pcg_check_every);
    const int maxit = std::max(1, D.opt.pcg_max_iters);
    const RowOf rows_a[3] = {+[](Shard &S) { return S.g.part_rr.p; }, +[](Shard &S) { return S.g.part_score.p; },
                             +[](Shard &S) { return S.g.part_pq.p; }};
    const RowOf rows_b[3] = {+[](Shard &S) { return S.g.part_rz.p; }, +[](Shard &S) { return S.g.part_score.p; },
                             +[](Shard &S) { return S.g.part_pq.p; }};
    auto iteration = [&]() {
        for (auto &sp : D.shards) {
            Graph &g = sp->g;
            (void)precondition(g, it == 0, rtol2, false);  // levels[1].y = M1^-1 P0' r (its dot output is not used)
            launch_form_u(g, g.part_score.p);
        }
        halo_exchange(D, HALO_P);  // P holds u
        for (auto &sp : D.shards) {
            Graph &g = sp->g;
            launch_spmv(g);  // AP = L u, part_pq = u.w
            reduce_pair(D, rr_buf(g, cur), rows_grid(g), g.part_score.p, grid_for_elems(g.levels[0].n), g.part_pq.p,
                        rows_grid(g));
        }
        allreduce_rows(D, cur ? rows_b : rows_a, 3);
        for (auto &sp : D.shards) {
            Graph &g = sp->g;
            launch_cgd_update(g, false, it & 1, it == 0, rtol2, g.part_score.p, g.part_pq.p, rr_buf(g, cur),
                              rr_buf(g, cur ^ 1));
        }
        cur ^= 1;
        it++;
    };
    // Chronopoulos-Gear carries r and s = Lp by recurrence, without residual replacement: on an
    // ill-conditioned system (weights over many decades) they can stall far above the tolerance where the
    // classic recurrences still converge (cgcg.hip, pcg_solve_cg2). Same safety net as there: after `giveup`
    // iterations, or when the residual has not halved for kStall iterations, the solve restarts with the
    // classic sharded recurrences from the saved right-hand side. The decision is taken from all-reduced
    // values (flags and relative residual are identical on every rank), so all ranks agree.
    const char *ge = std::getenv("IROTAVG_CG2_GIVEUP");  // tests force the hand-over
    const int giveup = ge ? std::max(1, std::atoi(ge)) : 400;
    constexpr int kStall = 96;
    double best = HUGE_VAL;
    int best_it = 0;
    bool hand_over = false;
    double h_rel[4] = {0, 0, 0, 0};
    int chunk = D.stats.pcg_iters_last > 2 ? (int)std::min<int64_t>(D.stats.pcg_iters_last + 1, maxit) : check;
    while (true) {
        for (int c = 0; c < chunk; c++) iteration();  // the update of iteration k tests the residual of k - 1
        chunk = std::max(2, check / 2);
        IRH_CHECK(hipMemcpyAsync(h_flags, D.shards[0]->g.flags.p, sizeof(int) * FL_COUNT, hipMemcpyDeviceToHost,
                                 D.stream));
        IRH_CHECK(hipMemcpyAsync(h_rel, D.shards[0]->g.scal.p + SC_RELRES, sizeof(double) * 3, hipMemcpyDeviceToHost,
                                 D.stream));
        IRH_CHECK(hipStreamSynchronize(D.stream));
        if (h_flags[FL_DONE] != 0) break;
        const double now = std::max(h_rel[0], std::max(h_rel[1], h_rel[2]));
        if (now < 0.5 * best) {
            best = now;
            best_it = it;
        }
        if (it >= giveup || it - best_it >= kStall) {
            hand_over = true;
            break;
        }
        if (it >= maxit) break;
    }
    if (hand_over || h_flags[FL_DONE] == 2) {
        D.stats.pcg_iters += h_flags[FL_ITERS];
        for (auto &sp : D.shards) {
            Graph &g = sp->g;
            IRH_CHECK(hipMemcpyAsync(g.levels[0].b.p, g.levels[0].x.p, sizeof(double4) * (size_t)g.levels[0].n,
                                     hipMemcpyDeviceToDevice, D.stream));
        }
        D.stats.pcg_handed_over += 1;
        return pcg_dist(D);
    }
    D.stats.pcg_solves += 1;
    D.stats.pcg_iters += h_flags[FL_ITERS];
    D.stats.pcg_iters_last = h_flags[FL_ITERS];
    if (h_flags[FL_DONE] == 2) return IROTAVG_ERR_SOLVER;
    if (h_flags[FL_DONE] == 0) return IROTAVG_ERR_NOT_CONVERGED;
    return IROTAVG_OK;
}

static int pcg_dist_any(Dist &D) {
    // every LOCAL shard must have the multi-level additive structure; with one shard per process the
    // choice must also agree across processes: shards are equal-sized ranges of one graph, so the level
    // count only differs when a range falls below mg_dense_max rows -- decided from the global size
    bool ok = D.opt.pcg_classic != 1 && D.chunk > std::max(D.opt.mg_dense_max, 1) &&
              D.nu - (int64_t)(D.world - 1) * D.chunk > std::max(D.opt.mg_dense_max, 1);
    for (auto &sp : D.shards) ok = ok && sp->g.additive_top && sp->g.levels.size() > 1;
    return ok ? pcg_dist_cg(D) : pcg_dist(D);
}

// The direct solve of a sharded sequence: local reductions, one gather, the separator system, the ways back. With loop
// closures (round 5) the ranks also sum ONE buffer -- what the closures' columns leave on the separators and every rank's
// share of the Woodbury system -- and the separator system carries the columns along (bcr.hip).
static int bcr_dist(Dist &D) {
    BcrTop &T = D.top;
    const bool cl = T.r > 0;
    // (the hosted wire sums the ranks' buffers -- every rank writes its own slices into a zeroed one; RCCL gathers
    // records and the loopback shards share the buffer: no clearing needed there)
    if (D.hosted) IRH_CHECK(hipMemsetAsync(T.buf.p, 0, sizeof(double) * T.n_doubles(), D.stream));
    if (cl) IRH_CHECK(hipMemsetAsync(T.xbuf.p, 0, sizeof(double) * T.x_doubles(), D.stream));
    for (auto &sp : D.shards) {
        bcr_shard_reduce(sp->g, T, sp->rank);
        if (cl) bcr_shard_closures_forward(sp->g, T, sp->rank);
    }
    tmark(D, 1);
    if (D.hosted) {
        const size_t n0 = T.n_doubles(), n1 = cl ? T.x_doubles() : 0;
        if (n0 + n1 > 0x7fffffffULL) throw HipError{hipErrorUnknown};
        D.hbuf.resize(n0 + n1);  // (both parts are overwritten whole)
        IRH_CHECK(hipMemcpyAsync(D.hbuf.data(), T.buf.p, sizeof(double) * n0, hipMemcpyDeviceToHost, D.stream));
        if (cl) IRH_CHECK(hipMemcpyAsync(D.hbuf.data() + n0, T.xbuf.p, sizeof(double) * n1, hipMemcpyDeviceToHost, D.stream));
        IRH_CHECK(hipStreamSynchronize(D.stream));
        if (D.tr.allreduce(D.tr.ctx, D.hbuf.data(), (int)(n0 + n1), 0) != 0) throw HipError{hipErrorUnknown};
        IRH_CHECK(hipMemcpyAsync(T.buf.p, D.hbuf.data(), sizeof(double) * n0, hipMemcpyHostToDevice, D.stream));
        if (cl) IRH_CHECK(hipMemcpyAsync(T.xbuf.p, D.hbuf.data() + n0, sizeof(double) * n1, hipMemcpyHostToDevice, D.stream));
        IRH_CHECK(hipStreamSynchronize(D.stream));  // hbuf is reused
    } else if (D.use_rccl) {
        // an ALL-GATHER of one record per rank (its five slices, 15 KB at B = 24) -- until round 5 a sum-all-reduce of
        // the whole zero-filled 120 KB buffer stood in for it: twice the bytes on every link and a memset per solve
        const int rank = D.shards[0]->rank;
        bcr_top_to_record(T, rank, D.stream);
        NCCL_CHECK(ncclAllGather(T.rec.p + (size_t)rank * T.record_doubles(), T.rec.p, T.record_doubles(), ncclDouble, D.comm,
                                 D.stream));
        bcr_top_from_records(T, D.stream);
        tmark(D, 2);
        if (cl) NCCL_CHECK(ncclAllReduce(T.xbuf.p, T.xbuf.p, T.x_doubles(), ncclDouble, ncclSum, D.comm, D.stream));
        if (cl) tmark(D, 3);
    }  // loopback: the shards of this process share the buffers
    tmark(D, 2);  // (the hosted wire moves separators and closure buffer in one sum: all of it under "gather")
    if (cl) {
        bcr_top_solve_closures(D.shards[0]->g, T);
        for (auto &sp : D.shards) bcr_shard_closures_correct(sp->g, T);
    } else {
        bcr_top_solve(D.shards[0]->g, T);
    }
    for (auto &sp : D.shards) bcr_shard_back(sp->g, T, sp->rank);
    tmark(D, 4);
    D.stats.direct_solves += 1;
    if (cl) {
        // a dead pivot of the band part (a view that only a closure ties to the rest after robust weights went to zero):
        // the Woodbury form does not hold. Every rank reads the same two counts (the ranks' own pivots were summed).
        double piv = 0.0;
        int topdead = 0;
        IRH_CHECK(hipMemcpyAsync(&piv, T.xbuf.p + T.x_pivots(), sizeof(double), hipMemcpyDeviceToHost, D.stream));
        IRH_CHECK(hipMemcpyAsync(&topdead, T.dead.p, sizeof(int), hipMemcpyDeviceToHost, D.stream));
        IRH_CHECK(hipStreamSynchronize(D.stream));
        D.stats.direct_dead_pivots = (int64_t)piv + topdead;
        if (piv > 0.0 || topdead > 0) {
            // (regularised, a dead pivot is the row's own diagonal entry and the result an approximate inverse that
            // bcr_dist_checked refines; otherwise the Woodbury form does not hold)
            if (!D.shards[0]->g.bcr_guard) return IROTAVG_ERR_SOLVER;
        }
    }
    return IROTAVG_OK;
}

// r = b - Ax on a shard's own rows (r may be Ax's array), partial sums of ||r||^2 and ||b||^2 per coordinate
__global__ __launch_bounds__(kRowBlock) void k_refine_resid(int n, const double4 *__restrict__ b, const double4 *Ax,
                                                            double4 *r, double *__restrict__ part_rr,
                                                            double *__restrict__ part_bb) {
    double a0 = 0, a1 = 0, a2 = 0, c0 = 0, c1 = 0, c2 = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double4 bi = b[i], ai = Ax[i];
        const double4 ri = make_double4(bi.x - ai.x, bi.y - ai.y, bi.z - ai.z, 0.0);
        r[i] = ri;
        a0 += ri.x * ri.x;
        a1 += ri.y * ri.y;
        a2 += ri.z * ri.z;
        c0 += bi.x * bi.x;
        c1 += bi.y * bi.y;
        c2 += bi.z * bi.z;
    }
    block_sum3_store(a0, a1, a2, part_rr + 4 * blockIdx.x);
    __syncthreads();
    block_sum3_store(c0, c1, c2, part_bb + 4 * blockIdx.x);
}
// per-column scalars of the repair below (three independent columns share the operator)
struct Col3 {
    double v[3];
};
// y += a o x;  p = z + b o p;  partial sums of u.v per column
__global__ __launch_bounds__(256) void k_axpy3(int n, double4 *__restrict__ y, const double4 *__restrict__ x, Col3 a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        double4 yi = y[i];
        const double4 xi = x[i];
        yi.x += a.v[0] * xi.x;
        yi.y += a.v[1] * xi.y;
        yi.z += a.v[2] * xi.z;
        y[i] = yi;
    }
}
__global__ __launch_bounds__(256) void k_xpby3(int n, double4 *__restrict__ p, const double4 *__restrict__ z, Col3 b) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const double4 pi = p[i], zi = z[i];
        p[i] = make_double4(zi.x + b.v[0] * pi.x, zi.y + b.v[1] * pi.y, zi.z + b.v[2] * pi.z, 0.0);
    }
}
__global__ __launch_bounds__(kRowBlock) void k_dot3(int n, const double4 *__restrict__ u, const double4 *__restrict__ v,
                                                    double *__restrict__ part) {
    double a0 = 0, a1 = 0, a2 = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double4 ui = u[i], vi = v[i];
        a0 += ui.x * vi.x;
        a1 += ui.y * vi.y;
        a2 += ui.z * vi.z;
    }
    block_sum3_store(a0, a1, a2, part + 4 * blockIdx.x);
}

// A direct solve with closures, checked. The Woodbury form loses digits where robust weights leave a stretch of the band
// nearly free and the closures hold it (bcr.hip, k_bcr_gate), and it does not hold at all when the band part has a dead
// pivot (a cost with exact-zero weights cut a view off all its band neighbours while a closure still holds it). The
// single-GPU handle repeats such a solve by conjugate gradients on the full operator with the (regularised) direct solve
// as the preconditioner (run_irls); the same here on the SHARDED operator: one residual of the full system per solve
// (one halo exchange of x, the sharded SpMV, one combined sum); above the gate, or after a dead pivot, preconditioned CG
// whose operator is the sharded SpMV (halo exchange of p) and whose preconditioner is bcr_dist itself -- regularised
// when a pivot died: M = A + E with E positive semi-definite and of the rank of the dead pivots, so CG ends after about
// as many iterations. Host-driven (a rare path): three sums per iteration, combined over the processes.
static int bcr_dist_checked(Dist &D) {
    int rc = bcr_dist(D);
    static const bool no_res = std::getenv("IROTAVG_BCR_NO_RESIDUAL_GATE") != nullptr;
    static const bool dbg = std::getenv("IROTAVG_DIST_DEBUG") != nullptr;
    if (D.top.r == 0 || no_res) return rc;
    // (the preconditioner applications of the repair below are not linear systems of the caller's: direct_solves counts
    // this call once, as run_irls does on one GPU -- advisor, round 5)
    struct GuardOff {
        Dist &D;
        const int64_t ds_keep;
        ~GuardOff() {
            for (auto &sp : D.shards) sp->g.bcr_guard = false;
            D.stats.direct_solves = ds_keep;
        }
    } guard_off{D, D.stats.direct_solves};
    const bool dead = rc == IROTAVG_ERR_SOLVER && D.stats.direct_dead_pivots > 0;
    if (rc != IROTAVG_OK && !dead) return rc;
    auto dots = [&](const double4 *(*U)(Shard &), const double4 *(*V)(Shard &), double out[3]) {
        double sums[4] = {0, 0, 0, 0};
        for (auto &sp : D.shards) {
            Graph &g = sp->g;
            const int n = g.levels[0].n, grid = grid_for_elems(n);
            hipLaunchKernelGGL(k_dot3, dim3(grid), dim3(kRowBlock), 0, D.stream, n, U(*sp), V(*sp), g.part_rr.p);
            reduce_pair(D, g.part_rr.p, grid, nullptr, 0);
            double h[4];
            IRH_CHECK(hipMemcpyAsync(h, g.part_rr.p, sizeof(double) * 4, hipMemcpyDeviceToHost, D.stream));
            IRH_CHECK(hipStreamSynchronize(D.stream));
            for (int c = 0; c < 3; c++) sums[c] += h[c];
        }
        combine_host(D, sums, 3, 0);
        for (int c = 0; c < 3; c++) out[c] = sums[c];
    };
    // q = A p on every shard: p lives in g.P (ghost values by the halo exchange), q in g.AP
    auto apply_A = [&]() {
        halo_exchange(D, HALO_P);
        for (auto &sp : D.shards) {
            IRH_CHECK(hipMemsetAsync(sp->g.flags.p, 0, sizeof(int) * FL_COUNT, D.stream));
            launch_spmv(sp->g);
        }
    };
    const auto fB = +[](Shard &S) { return (const double4 *)S.g.levels[0].b.p; };
    const auto fP = +[](Shard &S) { return (const double4 *)S.g.P.p; };
    const auto fQ = +[](Shard &S) { return (const double4 *)S.g.AP.p; };
    const auto fZ = +[](Shard &S) { return (const double4 *)(S.g.X.p + S.g.ng); };
    const auto fRb = +[](Shard &S) { return (const double4 *)S.rf_b.p; };
    double bb[3];
    if (!dead) {
        // the check of a solve that went through: r = b - A x (x and its ghost values straight from X: the callers' halo
        // exchange of x is this one; every shard's kernels first, then ONE wait)
        halo_exchange(D, HALO_X);
        double rr[3] = {0, 0, 0};
        double sums[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        std::vector<double> hs(D.shards.size() * 8, 0.0);
        size_t k = 0;
        for (auto &sp : D.shards) {
            Graph &g = sp->g;
            const int n = g.levels[0].n, grid = grid_for_elems(n);
            IRH_CHECK(hipMemsetAsync(g.flags.p, 0, sizeof(int) * FL_COUNT, D.stream));
            launch_spmv(g, g.X.p + g.ng, g.X.p);
            hipLaunchKernelGGL(k_refine_resid, dim3(grid), dim3(kRowBlock), 0, D.stream, n, (const double4 *)g.levels[0].b.p,
                               g.AP.p, g.AP.p, g.part_rr.p, g.part_rz.p);
            reduce_pair(D, g.part_rr.p, grid, g.part_rz.p, grid);
            IRH_CHECK(hipMemcpyAsync(hs.data() + 8 * k, g.part_rr.p, sizeof(double) * 4, hipMemcpyDeviceToHost, D.stream));
            IRH_CHECK(hipMemcpyAsync(hs.data() + 8 * k + 4, g.part_rz.p, sizeof(double) * 4, hipMemcpyDeviceToHost, D.stream));
            k++;
        }
        IRH_CHECK(hipStreamSynchronize(D.stream));
        for (size_t q = 0; q < k; q++)
            for (int c = 0; c < 8; c++) sums[c] += hs[8 * q + c];
        combine_host(D, sums, 8, 0);
        double worst = 0.0;
        for (int c = 0; c < 3; c++) {
            rr[c] = sums[c];
            bb[c] = sums[4 + c];
            const double rel = bb[c] > 0.0 ? std::sqrt(rr[c] / bb[c]) : (rr[c] > 0.0 ? HUGE_VAL : 0.0);
            D.stats.last_relres[c] = rel;
            worst = std::max(worst, rel);
        }
        if (dbg) std::fprintf(stderr, "[bcr_dist_checked] relres of the direct solve %.3e\n", worst);
        if (!(worst > std::max(D.opt.pcg_rtol, 1e-12))) return IROTAVG_OK;  // (the gate of the single-GPU handle: pcg_rtol)
        if (!std::isfinite(worst)) return IROTAVG_ERR_SOLVER;
    }
    // ---- the repair: CG on A x = b, preconditioner = the sharded direct solve ----
    D.stats.direct_guarded += 1;
    if (dead)
        for (auto &sp : D.shards) sp->g.bcr_guard = true;
    for (auto &sp : D.shards) {
        const int n = sp->g.levels[0].n;
        if (sp->rf_b.n < (size_t)n) {
            sp->rf_b.alloc((size_t)n);
            sp->rf_x.alloc((size_t)n);
        }
        IRH_CHECK(hipMemcpyAsync(sp->rf_b.p, sp->g.levels[0].b.p, sizeof(double4) * (size_t)n, hipMemcpyDeviceToDevice, D.stream));
        IRH_CHECK(hipMemsetAsync(sp->rf_x.p, 0, sizeof(double4) * (size_t)n, D.stream));
    }
    dots(fRb, fRb, bb);
    const double rtol = std::max(D.opt.pcg_rtol, 1e-12);
    const int maxit = dead ? 200 : 12;
    double rz[3] = {0, 0, 0}, worst = HUGE_VAL;
    int it = 0;
    // r lives in L0.b (it IS the right-hand side of the preconditioner's solve), z = the solve's result (X), x in rf_x
    for (;; it++) {
        rc = bcr_dist(D);  // z = M^-1 r
        if (rc != IROTAVG_OK && !(dead && rc == IROTAVG_ERR_SOLVER && D.shards[0]->g.bcr_guard)) break;
        rc = IROTAVG_OK;
        double rz_new[3];
        dots(fB, fZ, rz_new);
        Col3 beta;
        for (int c = 0; c < 3; c++) beta.v[c] = it == 0 || rz[c] == 0.0 ? 0.0 : rz_new[c] / rz[c];
        for (int c = 0; c < 3; c++) rz[c] = rz_new[c];
        for (auto &sp : D.shards) {
            const int n = sp->g.levels[0].n;
            if (it == 0)  // p = z (whatever the array held is not multiplied by a zero)
                IRH_CHECK(hipMemcpyAsync(sp->g.P.p, sp->g.X.p + sp->g.ng, sizeof(double4) * (size_t)n, hipMemcpyDeviceToDevice,
                                         D.stream));
            else
                hipLaunchKernelGGL(k_xpby3, dim3((n + 255) / 256), dim3(256), 0, D.stream, n, sp->g.P.p,
                                   (const double4 *)(sp->g.X.p + sp->g.ng), beta);
        }
        apply_A();
        double pq[3];
        dots(fP, fQ, pq);
        Col3 al, nal;
        for (int c = 0; c < 3; c++) {
            al.v[c] = pq[c] > 0.0 ? rz[c] / pq[c] : 0.0;
            nal.v[c] = -al.v[c];
        }
        for (auto &sp : D.shards) {
            const int n = sp->g.levels[0].n;
            hipLaunchKernelGGL(k_axpy3, dim3((n + 255) / 256), dim3(256), 0, D.stream, n, sp->rf_x.p, (const double4 *)sp->g.P.p, al);
            hipLaunchKernelGGL(k_axpy3, dim3((n + 255) / 256), dim3(256), 0, D.stream, n, sp->g.levels[0].b.p,
                               (const double4 *)sp->g.AP.p, nal);
        }
        double rr[3];
        dots(fB, fB, rr);
        worst = 0.0;
        for (int c = 0; c < 3; c++) {
            const double rel = bb[c] > 0.0 ? std::sqrt(rr[c] / bb[c]) : (rr[c] > 0.0 ? HUGE_VAL : 0.0);
            D.stats.last_relres[c] = rel;
            worst = std::max(worst, rel);
        }
        if (dbg) std::fprintf(stderr, "[bcr_dist_checked] cg %d relres %.3e (dead pivots %lld)\n", it + 1, worst,
                              (long long)D.stats.direct_dead_pivots);
        if (!(worst > rtol) || !std::isfinite(worst) || it + 1 >= maxit) break;
    }
    // the solution where the callers read it, the right-hand side back
    for (auto &sp : D.shards) {
        const size_t bytes = sizeof(double4) * (size_t)sp->g.levels[0].n;
        IRH_CHECK(hipMemcpyAsync(sp->g.X.p + sp->g.ng, sp->rf_x.p, bytes, hipMemcpyDeviceToDevice, D.stream));
        IRH_CHECK(hipMemcpyAsync(sp->g.levels[0].b.p, sp->rf_b.p, bytes, hipMemcpyDeviceToDevice, D.stream));
    }
    if (rc != IROTAVG_OK) return rc;
    // (what the iterations reached must still be a solution -- see run_irls: a band part next to singular under hundreds
    // of closures is no case for the Woodbury form; the caller creates the handle with band_direct = -1)
    if (!std::isfinite(worst) || !(worst <= kClosureRepairAccept)) return IROTAVG_ERR_SOLVER;
    return IROTAVG_OK;
}
static int solve_dist(Dist &D) { return D.bcr_B ? bcr_dist_checked(D) : pcg_dist_any(D); }

static int irls_dist(Dist &D, int cost, double sigma, int max_iters, double change_th, int *iters,
                     double *runtime, double *trace) {
    if (cost < IROTAVG_L2 || cost > IROTAVG_WELSCH) return IROTAVG_ERR_UNKNOWN_COST;
    const double tic = now_seconds();
    double score = HUGE_VAL;
    int it = 0, rc = IROTAVG_OK;
    for (auto &sp : D.shards) fill(sp->g, sp->g.dw.p, (long long)sp->g.mpad, 1.0);
    while (score > change_th && it < max_iters) {
        tmark(D, -1);
        for (auto &sp : D.shards) {
            launch_edge_residual(sp->g);
            assemble(sp->g, 0, sp->g.dw.p, D.opt.dense_always_refresh == 1);
        }
        tmark(D, 0);
        rc = solve_dist(D);
        if (rc != IROTAVG_OK) break;
        tmark(D, 1);  // (the sharded PCG: the whole solve; the direct solver has marked its own phases, this adds ~0)
        halo_exchange(D, HALO_X);  // ghost views receive their owners' steps
        tmark(D, 5);
        double local = 0.0;
        for (auto &sp : D.shards) {
            launch_update_weights(sp->g, cost, sigma);
            (void)apply_step(sp->g);  // updates owned AND ghost rotations; scores owned views only
            local += sp->g.last_score_sum;
        }
        tmark(D, 6);
        combine_host(D, &local, 1, 0);
        tmark(D, 7);
        if (D.timing) D.t_iters += 1;
        score = local / (double)D.nu;
        if (trace) trace[it] = score;
        it++;
    }
    IRH_CHECK(hipStreamSynchronize(D.stream));
    const double toc = now_seconds();
    *iters = it;
    *runtime = toc - tic;
    D.stats.outer_iters += it;
    D.stats.edge_updates += (int64_t)it * D.m;
    D.stats.seconds_irls += toc - tic;
    return rc;
}

// Sharded L1RA (ral/l1_irls.cpp:851-912 with l1decode_pd :228-468 inside). The primal-dual LP of a
// coordinate runs over all shards at once (l1decode_group, l1pd.hip): every shard advances the edge
// vectors of its own edges (a cross-shard edge lives on both shards and is advanced identically on
// both), A'y walks the owned views, sums over edges count an edge once (Shard::eown), the Hessian
// system is the sharded PCG above, A dx needs one halo exchange of dx, and the scalars that steer
// the iteration (duality gap, step bound, back-tracking norms) are combined over the processes --
// every process takes the same decisions. The three coordinates run one after the other.
static int l1ra_dist(Dist &D, int max_iters, double change_th, int *iters, double *runtime, double *trace) {
    const double tic = now_seconds();
    double score = HUGE_VAL;
    int l1_step = 2;  // :868
    int it = 0, rc = IROTAVG_OK;
    for (auto &sp : D.shards) pd_prepare_graph(sp->g);
    while (((score >= change_th) || (l1_step < 2)) && (it < max_iters)) {  // :877, >=
        if (score < change_th) {  // :879-883 -- unreachable under the guard above; kept literal
            l1_step *= 4;
            change_th /= 100.0;
        }
        for (auto &sp : D.shards) launch_edge_residual(sp->g);
        for (int c = 0; c < 3 && rc == IROTAVG_OK; c++) {  // :889-892
            PdGroup G;
            for (auto &sp : D.shards)
                G.mem.push_back(PdMember{&sp->g, sp->eown.p, sp->g.er.p + (size_t)c * sp->g.mpad});
            G.m_global = D.m;
            G.combine = [&D](double *v, int n, int op) { combine_host(D, v, n, op); };
            G.halo_x = [&D]() { halo_exchange(D, HALO_X); };
            G.solve = [&D]() { return solve_dist(D); };
            rc = l1decode_group(G, l1_step, kPdXPlane0 + c, nullptr);
        }
        if (rc != IROTAVG_OK) break;
        for (auto &sp : D.shards) pd_pack_solution(sp->g);
        halo_exchange(D, HALO_X);  // ghost views receive their owners' steps
        double local = 0.0;
        for (auto &sp : D.shards) {
            (void)apply_step(sp->g);  // :894-902
            local += sp->g.last_score_sum;
        }
        combine_host(D, &local, 1, 0);
        score = local / (double)D.nu;
        if (trace) trace[it] = score;
        it++;
    }
    IRH_CHECK(hipStreamSynchronize(D.stream));
    const double toc = now_seconds();
    *iters = it;
    *runtime = toc - tic;
    D.stats.outer_iters += it;
    D.stats.seconds_l1ra += toc - tic;
    return rc;
}

}  // namespace irh

using namespace irh;

struct irotavg_dist {
    Dist D;
};

#define API_TRY try {
#define API_CATCH                     \
    }                                 \
    catch (const HipError &) {        \
        return IROTAVG_ERR_HIP;       \
    }                                 \
    catch (const std::bad_alloc &) {  \
        return IROTAVG_ERR_NOMEM;     \
    }                                 \
    catch (...) {                     \
        return IROTAVG_ERR_HIP;       \
    }

extern "C" {

int irotavg_dist_unique_id(void *out128) {
    if (!out128) return IROTAVG_ERR_BAD_ARG;
    static_assert(sizeof(ncclUniqueId) <= 128, "ncclUniqueId larger than the ABI buffer");
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return IROTAVG_ERR_HIP;
    std::memset(out128, 0, 128);
    std::memcpy(out128, &id, sizeof(id));
    return IROTAVG_OK;
}

static int dist_create_impl(irotavg_dist **out, int world, int rank, const void *unique_id128,
                            const irotavg_transport *tr, int64_t m, int64_t n_total, int f, const int32_t *I,
                            const double *QQ, int64_t ldqq, const irotavg_options *opt) {
    if (!out || !I || !QQ || world < 1 || m <= 0 || n_total <= 0 || f < 0 || n_total - f < 1 ||
        ldqq < m || n_total > 0x7fffffffLL)
        return IROTAVG_ERR_BAD_ARG;
    const bool loopback = unique_id128 == nullptr && tr == nullptr;
    if (!loopback && (rank < 0 || rank >= world)) return IROTAVG_ERR_BAD_ARG;
    *out = nullptr;
    if (irotavg_device_count() <= 0) return IROTAVG_ERR_NO_DEVICE;
    irotavg_dist *h = nullptr;
    try {
        h = new irotavg_dist();
        Dist &D = h->D;
        if (opt)
            D.opt = *opt;
        else
            irotavg_default_options(&D.opt);
        if (D.opt.pcg_rtol <= 0) D.opt.pcg_rtol = 1e-10;
        if (D.opt.mg_omega <= 0) D.opt.mg_omega = 0.7;
        if (D.opt.mg_dense_max <= 0) D.opt.mg_dense_max = 2048;
        if (D.opt.mg_levels_max <= 0) D.opt.mg_levels_max = 16;
        if (D.opt.pcg_max_iters <= 0) D.opt.pcg_max_iters = 2000;
        if (D.opt.pcg_check_every <= 0) D.opt.pcg_check_every = 8;
        if (D.opt.device >= 0) IRH_CHECK(hipSetDevice(D.opt.device));
        D.stream = StreamPool::get().take();
        D.world = world;
        D.m = m;
        D.n_total = n_total;
        D.f = f;
        D.nu = n_total - f;
        {   // the direct solver for a sharded view sequence? Decided from the GLOBAL graph: every process agrees.
            int mode = D.opt.band_direct;
            if (const char *e = std::getenv("IROTAVG_BAND_DIRECT")) mode = std::atoi(e);
            int band = 0, bandall = 0;
            int64_t nfar = 0;
            bool ok = mode >= 0 && world <= 8;
            for (int64_t k = 0; k < m && ok; k++) {
                const int i = I[2 * k], j = I[2 * k + 1];
                if (i < 0 || j < 0 || i >= n_total || j >= n_total) ok = false;
                else if (i >= f && j >= f) {
                    const int d = std::abs(i - j);
                    bandall = std::max(bandall, d);
                    if (d <= 32) band = std::max(band, d);
                    else nfar++;
                }
            }
            D.band0 = ok ? bandall : -1;
            // (the limits of the single-GPU plan, bcr_plan: at most 2048 long-range edges, 1024 on small graphs)
            const bool closures_ok = !std::getenv("IROTAVG_BCR_NO_CLOSURES") && !std::getenv("IROTAVG_DIST_NO_CLOSURES");
            if (ok && nfar > 0 && (!closures_ok || nfar > (D.nu < 8192 ? 1024 : 2048))) ok = false;
            if (ok && (mode > 0 || D.nu > 2048)) {
                const int B = band <= 8 ? 8 : band <= 16 ? 16 : band <= 24 ? 24 : 32;
                const int64_t c192 = chunk_of(D.nu, world, 192);
                if (c192 >= 2 * B && (int64_t)(world - 1) * c192 < D.nu) D.bcr_B = B;
                if (D.bcr_B && nfar > 0) {
                    // closures = long-range edges whose blocks are not neighbours (ranges are multiples of every block
                    // size: a view's block number is the same on every rank). The band part alone must be positive
                    // definite: every free view has a band edge to an earlier view or a kept edge to a fixed one
                    // (bcr_plan's rule); otherwise the sharded PCG.
                    std::vector<uint8_t> tied((size_t)D.nu, 0);
                    for (int64_t k = 0; k < m; k++) {
                        const int i = I[2 * k], j = I[2 * k + 1];
                        if (i >= f && j >= f) {
                            const int d = std::abs(i - j);
                            if (d > 32 && std::abs((i - f) / B - (j - f) / B) >= 2) D.cl_edge.push_back(k);
                            else if (i != j && d <= 32) tied[(size_t)(std::max(i, j) - f)] = 1;
                        } else if (i < f && j >= f) {
                            tied[(size_t)(j - f)] = 1;
                        }
                    }
                    bool all = true;
                    for (int64_t r = 0; r < D.nu && all; r++) all = tied[(size_t)r] != 0;
                    if (!all) all = bcr_band_part_anchored(m, f, D.nu, B, I);  // (the rule is sufficient only: the exact test)
                    if (!all) {
                        D.cl_edge.clear();
                        D.bcr_B = 0;
                    }
                }
            }
        }
        D.chunk = chunk_of(D.nu, world, D.bcr_B ? 192 : 64);
        if ((int64_t)(world - 1) * D.chunk >= D.nu) {  // every shard needs at least one view
            irotavg_dist_destroy(h);
            return IROTAVG_ERR_BAD_ARG;
        }
        if (D.bcr_B) bcr_top_alloc(D.top, D.bcr_B, world);
        D.hosted = tr != nullptr;
        if (D.hosted) D.tr = *tr;
        D.use_rccl = !loopback && !D.hosted;
        if (D.use_rccl) {
            ncclUniqueId id;
            std::memcpy(&id, unique_id128, sizeof(id));
            NCCL_CHECK(ncclCommInitRank(&D.comm, world, id, rank));
        }
        const int first = loopback ? 0 : rank, last = loopback ? world : rank + 1;
        for (int r = first; r < last; r++) {
            D.shards.emplace_back(new Shard());
            D.shards.back()->rank = r;
            const int rc = build_shard(D, *D.shards.back(), I, QQ, ldqq);
            if (rc != IROTAVG_OK) {
                irotavg_dist_destroy(h);
                return rc;
            }
        }
        if (D.bcr_B && !D.cl_edge.empty()) bcr_top_closures_alloc(D.shards[0]->g, D.top, (int)D.cl_edge.size());
        // the gathered halo of a closure-free sequence (Dist::halo_gather): decided from what EVERY process derives alike
        // (the global plan above), then checked against this process's own shards -- a rank whose halo does not have the
        // promised shape fails loudly here instead of disagreeing with the others about the wire
        if (D.bcr_B && D.cl_edge.empty() && !D.hosted && world > 1 && !std::getenv("IROTAVG_DIST_HALO_P2P")) {
            D.halo_gather = true;
            D.halo_w = D.bcr_B;
            D.hall.alloc((size_t)world * 2 * D.halo_w);
            D.hall.zero(D.stream);
            for (auto &sp : D.shards) {
                Shard &S = *sp;
                std::vector<int> dst((size_t)S.send_total + 1, 0);
                if (S.peers.size() > 2) throw HipError{hipErrorUnknown};
                for (size_t q = 0; q < S.peers.size(); q++) {
                    const int peer = S.peers[q];
                    if ((peer != S.rank - 1 && peer != S.rank + 1) || S.send_cnt[q] > D.halo_w || S.recv_cnt[q] > D.halo_w)
                        throw HipError{hipErrorUnknown};
                    for (int k = 0; k < S.send_cnt[q]; k++) dst[(size_t)S.send_off[q] + k] = (peer < S.rank ? 0 : D.halo_w) + k;
                }
                S.gather_dst.upload(dst, D.stream);
                IRH_CHECK(hipStreamSynchronize(D.stream));  // (dst is on this stack)
            }
        }
        *out = h;
        return IROTAVG_OK;
    } catch (const std::bad_alloc &) {
        if (h) irotavg_dist_destroy(h);
        return IROTAVG_ERR_NOMEM;
    } catch (...) {
        if (h) irotavg_dist_destroy(h);
        return IROTAVG_ERR_HIP;
    }
}

int irotavg_dist_create(irotavg_dist **out, int world, int rank, const void *unique_id128, int64_t m,
                        int64_t n_total, int f, const int32_t *I, const double *QQ, int64_t ldqq,
                        const irotavg_options *opt) {
    return dist_create_impl(out, world, rank, unique_id128, nullptr, m, n_total, f, I, QQ, ldqq, opt);
}

int irotavg_dist_create_hosted(irotavg_dist **out, int world, int rank, const irotavg_transport *transport,
                               int64_t m, int64_t n_total, int f, const int32_t *I, const double *QQ,
                               int64_t ldqq, const irotavg_options *opt) {
    if (!transport || !transport->allreduce || !transport->exchange) return IROTAVG_ERR_BAD_ARG;
    return dist_create_impl(out, world, rank, nullptr, transport, m, n_total, f, I, QQ, ldqq, opt);
}

void irotavg_dist_destroy(irotavg_dist *h) {
    if (!h) return;
    Dist &D = h->D;
    if (D.stream) (void)hipStreamSynchronize(D.stream);
    for (auto &sp : D.shards) sp->g.stream = nullptr;  // shards share D.stream
    D.shards.clear();
    if (D.comm) (void)ncclCommDestroy(D.comm);
    if (D.stream) StreamPool::get().give(D.stream);
    delete h;
}

// Q: the GLOBAL n_total x 4 column-major matrix (every process passes the same one)
int irotavg_dist_set_rotations(irotavg_dist *h, const double *Q, int64_t ldq) {
    if (!h || !Q || ldq < h->D.n_total) return IROTAVG_ERR_BAD_ARG;
    API_TRY
    Dist &D = h->D;
    for (auto &sp : D.shards) {
        Shard &S = *sp;
        std::vector<double4> aos(S.gvert.size());
        for (size_t t = 0; t < S.gvert.size(); t++) {
            const int64_t v = S.gvert[t];
            aos[t] = make_double4(Q[v], Q[ldq + v], Q[2 * ldq + v], Q[3 * ldq + v]);
        }
        S.g.Q.upload(aos.data(), aos.size(), D.stream);
        IRH_CHECK(hipStreamSynchronize(D.stream));
    }
    return IROTAVG_OK;
    API_CATCH
}

// device-side snapshot / restore of every local shard's rotations (as irotavg_graph_snapshot_rotations: a bench
// step restarts from the same rotations without a host-to-device copy inside its timed region)
int irotavg_dist_snapshot_rotations(irotavg_dist *h) {
    if (!h) return IROTAVG_ERR_BAD_ARG;
    API_TRY
    for (auto &sp : h->D.shards) {
        Graph &g = sp->g;
        if (g.Qsnap.n < (size_t)g.n_total) g.Qsnap.alloc((size_t)g.n_total);
        IRH_CHECK(hipMemcpyAsync(g.Qsnap.p, g.Q.p, sizeof(double4) * (size_t)g.n_total, hipMemcpyDeviceToDevice,
                                 h->D.stream));
    }
    IRH_CHECK(hipStreamSynchronize(h->D.stream));
    return IROTAVG_OK;
    API_CATCH
}
int irotavg_dist_restore_rotations(irotavg_dist *h) {
    if (!h) return IROTAVG_ERR_BAD_ARG;
    API_TRY
    for (auto &sp : h->D.shards) {
        Graph &g = sp->g;
        if (g.Qsnap.n < (size_t)g.n_total) return IROTAVG_ERR_BAD_ARG;
        IRH_CHECK(hipMemcpyAsync(g.Q.p, g.Qsnap.p, sizeof(double4) * (size_t)g.n_total, hipMemcpyDeviceToDevice,
                                 h->D.stream));
    }
    return IROTAVG_OK;
    API_CATCH
}

// writes the rows OWNED by this process's shards into the global matrix (other rows untouched)
int irotavg_dist_get_rotations(irotavg_dist *h, double *Q, int64_t ldq) {
    if (!h || !Q || ldq < h->D.n_total) return IROTAVG_ERR_BAD_ARG;
    API_TRY
    Dist &D = h->D;
    for (auto &sp : D.shards) {
        Shard &S = *sp;
        std::vector<double4> aos(S.gvert.size());
        IRH_CHECK(hipMemcpyAsync(aos.data(), S.g.Q.p, sizeof(double4) * aos.size(),
                                 hipMemcpyDeviceToHost, D.stream));
        IRH_CHECK(hipStreamSynchronize(D.stream));
        for (size_t t = (size_t)(S.g.f + S.g.ng); t < S.gvert.size(); t++) {
            const int64_t v = S.gvert[t];
            Q[v] = aos[t].x;
            Q[ldq + v] = aos[t].y;
            Q[2 * ldq + v] = aos[t].z;
            Q[3 * ldq + v] = aos[t].w;
        }
    }
    return IROTAVG_OK;
    API_CATCH
}

// writes the weights of this process's local edges into the global m-vector
int irotavg_dist_get_weights(irotavg_dist *h, double *w) {
    if (!h || !w) return IROTAVG_ERR_BAD_ARG;
    API_TRY
    Dist &D = h->D;
    for (auto &sp : D.shards) {
        Shard &S = *sp;
        std::vector<double> loc((size_t)S.g.m);
        IRH_CHECK(hipMemcpyAsync(loc.data(), S.g.dw.p, sizeof(double) * loc.size(),
                                 hipMemcpyDeviceToHost, D.stream));
        IRH_CHECK(hipStreamSynchronize(D.stream));
        for (size_t t = 0; t < loc.size(); t++) w[S.gedge[t]] = loc[t];
    }
    return IROTAVG_OK;
    API_CATCH
}

int irotavg_dist_irls(irotavg_dist *h, int cost, double sigma, int max_iters, double change_th,
                      int *iters, double *runtime, double *trace) {
    if (!h || !iters || !runtime) return IROTAVG_ERR_BAD_ARG;
    API_TRY
    return irls_dist(h->D, cost, sigma, max_iters, change_th, iters, runtime, trace);
    API_CATCH
}

int irotavg_dist_l1ra(irotavg_dist *h, int max_iters, double change_th, int *iters, double *runtime,
                      double *trace) {
    if (!h || !iters || !runtime) return IROTAVG_ERR_BAD_ARG;
    API_TRY
    return l1ra_dist(h->D, max_iters, change_th, iters, runtime, trace);
    API_CATCH
}

int irotavg_dist_info(irotavg_dist *h, int64_t info[8]) {
    if (!h || !info) return IROTAVG_ERR_BAD_ARG;
    API_TRY
    Dist &D = h->D;
    for (int i = 0; i < 8; i++) info[i] = 0;
    info[0] = (D.hosted ? 2 : (D.use_rccl ? 1 : 0)) + (D.halo_gather ? 16 : 0);
    if (D.use_rccl && D.comm) {
        int cnt = 0;
        NCCL_CHECK(ncclCommCount(D.comm, &cnt));
        info[1] = cnt;
    }
    info[2] = (int64_t)D.shards.size();
    info[3] = D.world;
    for (auto &sp : D.shards) info[4] += sp->g.ng;
    info[5] = D.shards.empty() ? 0 : (int64_t)D.shards[0]->peers.size();
    info[6] = D.bcr_B;  // block size of the sharded direct solver (0: the sharded PCG)
    info[7] = D.bcr_B ? (int64_t)D.cl_edge.size() : 0;  // loop closures it carries (Woodbury correction across the ranks)
    return IROTAVG_OK;
    API_CATCH
}

int irotavg_dist_timing(irotavg_dist *h, int enable, double us_per_iteration[8], int64_t *iterations) {
    if (!h) return IROTAVG_ERR_BAD_ARG;
    Dist &D = h->D;
    if (us_per_iteration)
        for (int i = 0; i < 8; i++) us_per_iteration[i] = D.t_iters > 0 ? 1e6 * D.t_acc[i] / (double)D.t_iters : 0.0;
    if (iterations) *iterations = D.t_iters;
    if (enable) {
        for (int i = 0; i < 8; i++) D.t_acc[i] = 0.0;
        D.t_iters = 0;
    }
    D.timing = enable != 0;
    return IROTAVG_OK;
}

int irotavg_dist_get_stats(irotavg_dist *h, irotavg_stats *out) {
    if (!h || !out) return IROTAVG_ERR_BAD_ARG;
    *out = h->D.stats;
    out->levels = h->D.shards.empty() ? 0 : h->D.shards[0]->g.stats.levels;
    for (int i = 0; i < 16; i++) {
        out->level_rows[i] = h->D.shards.empty() ? 0 : h->D.shards[0]->g.stats.level_rows[i];
        out->level_nnz[i] = h->D.shards.empty() ? 0 : h->D.shards[0]->g.stats.level_nnz[i];
    }
    return IROTAVG_OK;
}

// Host-only partition plan of rank `rank` (no GPU needed; used by the CPU tests of the N > 1
// path). counts = {owned lo, owned hi, #ghosts, #local edges, #peers, #send entries}. ghosts_out
// receives the ghost views' GLOBAL ids (ascending), send_out the GLOBAL ids this rank sends,
// grouped by peer in `peers` order; the per-peer counts go to send_cnt / recv_cnt.
int irotavg_dist_plan_host(int world, int rank, int64_t m, int64_t n_total, int f, const int32_t *I,
                           int64_t counts[6], int32_t *ghosts_out, int64_t ghosts_cap,
                           int32_t *send_out, int64_t send_cap, int32_t *edges_out, int64_t edges_cap,
                           int *peers, int *send_cnt, int *recv_cnt, int peers_cap) {
    if (!I || !counts || world < 1 || rank < 0 || rank >= world || n_total - f < 1)
        return IROTAVG_ERR_BAD_ARG;
    try {
        ShardPlan P;
        const int rc = plan_shard(world, rank, m, n_total, f, I, P);
        if (rc != IROTAVG_OK) return rc;
        counts[0] = P.lo;
        counts[1] = P.hi;
        counts[2] = (int64_t)P.ghosts.size();
        counts[3] = (int64_t)P.ledge.size();
        counts[4] = (int64_t)P.peers.size();
        counts[5] = (int64_t)P.send_idx.size();
        if (ghosts_out)
            for (int64_t q = 0; q < (int64_t)P.ghosts.size() && q < ghosts_cap; q++) ghosts_out[q] = P.ghosts[q];
        if (send_out)
            for (int64_t q = 0; q < (int64_t)P.send_idx.size() && q < send_cap; q++)
                send_out[q] = (int32_t)(f + P.lo + P.send_idx[q]);
        if (edges_out)
            for (int64_t q = 0; q < (int64_t)P.ledge.size() && q < edges_cap; q++) edges_out[q] = (int32_t)P.ledge[q];
        for (int q = 0; q < (int)P.peers.size() && q < peers_cap; q++) {
            if (peers) peers[q] = P.peers[q];
            if (send_cnt) send_cnt[q] = P.send_cnt[q];
            if (recv_cnt) recv_cnt[q] = P.recv_cnt[q];
        }
        return IROTAVG_OK;
    } catch (...) {
        return IROTAVG_ERR_NOMEM;
    }
}

// partition plan of shard `local_index` (for the tests): counts[0..5] = rank, owned lo, owned hi,
// ghosts, local edges, peers; the peer / send / recv tables are copied up to `cap` entries
int irotavg_dist_plan(irotavg_dist *h, int local_index, int64_t counts[6], int *peers, int *send_cnt,
                      int *recv_cnt, int cap) {
    if (!h || !counts || local_index < 0 || local_index >= (int)h->D.shards.size())
        return IROTAVG_ERR_BAD_ARG;
    Shard &S = *h->D.shards[local_index];
    counts[0] = S.rank;
    counts[1] = S.lo;
    counts[2] = S.hi;
    counts[3] = S.g.ng;
    counts[4] = S.g.m;
    counts[5] = (int64_t)S.peers.size();
    for (int q = 0; q < (int)S.peers.size() && q < cap; q++) {
        if (peers) peers[q] = S.peers[q];
        if (send_cnt) send_cnt[q] = S.send_cnt[q];
        if (recv_cnt) recv_cnt[q] = S.recv_cnt[q];
    }
    return IROTAVG_OK;
}

}  // extern "C"

// graph.hpp -- the device-resident view-graph handle behind the C ABI.
#pragma once
#include <chrono>
#include <functional>
#include <memory>

#include "common.hpp"

namespace irh {

struct Graph;
// bcr.hip: the share of the device a handle has reserved for its single-launch upper reductions (k_bcr_reduce_up) is
// handed back when the handle is idle -- end of irls / l1ra / l1decode_pd / time_kernel, destruction
void bcr_up_release(Graph &g) noexcept;
struct BcrState;  // bcr.hip: buffers of the direct solve of a banded level-0 operator
struct BcrDeleter {
    void operator()(BcrState *p) const;
};

struct Graph {
    int64_t m = 0, n_total = 0, mpad = 0;
    // Local vertex order is [fixed | ghost | owned]. f = fixed views, ng = ghost views (free
    // views owned by another shard; 0 on a single GPU), nu = n_total - f = ghost + owned (length
    // of X), no = owned views = rows of the operator.
    int f = 0, nu = 0, ng = 0, no = 0;
    irotavg_options opt{};
    hipStream_t stream = nullptr;
    int device = 0;

    // edges: SoA streams for the edge-parallel kernels
    DevBuf<int> ei, ej;
    DevBuf<uint8_t> eflag;
    DevBuf<double> qq;  // 4 planes (x,y,z,w) of mpad doubles: the reference's col-major QQ
    DevBuf<double> er;  // 3 planes (rx,ry,rz) of mpad doubles: rotation-vector residual per edge
    DevBuf<double> dw;  // IRLS weights d_k (m)
    DevBuf<double4> T;  // per edge (w r, w), w = d^2: the assembly's gather record (IRLS mode)
    // views
    DevBuf<double4> Q;  // n_total quaternions [x y z w], gather-friendly AoS
    DevBuf<double4> Qsnap;

    // level-0 adjacency extras (CSR itself lives in levels[0])
    DevBuf<uint8_t> slot_cs;    // per level-0 SELL position: entry index in the level-1 row it sums into (255: none)
    DevBuf<int> tile_e0;        // per level-0 slice: first edge of the window k_assemble0w stages in LDS
    int asm_windowed = 1;       // K3 by k_assemble0w (IROTAVG_ASM_CLASSIC=1: edge_pack + assemble0 + coarse kernels)
    int asm_l1_fused = 0;       // ... which also refreshes level 1 (aggregates of 8, level-1 rows of <= 8 entries)
    DevBuf<uint32_t> slot_eid;  // SELL layout of level 0: (edge id << 1) | (row is the j endpoint); ~0u = padding
    DevBuf<int> bptr;           // per row: boundary slots (other endpoint fixed / self loop)
    DevBuf<uint32_t> beid;
    DevBuf<uint8_t> bflag;
    DevBuf<int> bghost;     // per boundary slot: ghost index of the other endpoint, or -1
    DevBuf<double> bval;    // per boundary slot: current weight (refreshed by the assembly)
    DevBuf<double4> PG;     // ghost values of the PCG direction p (filled by the halo exchange)

    std::vector<Level> levels;
    DevBuf<double> dense_inv;  // explicit inverse of the coarsest level, ndense_pad^2 row-major
    DevBuf<float> dense_inv32;  // fp32 copy for the tile slices of k_cg_apply (cgcg.hip), refreshed lazily
    uint64_t dense_epoch = 1, inv32_epoch = 0;  // dense_epoch advances whenever dense_inv's content changes
    int dense32 = 0;            // 1: k_cg_apply reads an fp32 copy of the inverse (IROTAVG_CG2_FP32_DENSE=1; measured
                                // 2-3 % faster, not worth a preconditioner whose definiteness rests on fp32 rounding)
    DevBuf<double> dense_wr, dense_wc;  // Gauss-Jordan panels (32 x npad, npad x 32), two of each
    DevBuf<double> dense_chol;          // scratch of dense_direct_solve: the operator twice (factor, copy)
    DevBuf<double> dense_wb;            // scratch of the low-rank repair (Z, W, S, S^-1, entry list)
    int dense_repairs_in_a_row = 0;     // since the last full inversion
    DevBuf<double> dense_la;            // look-ahead: two snapshots of upcoming 32 x 32 diagonal blocks
    int ndense = 0, ndense_pad = 0;
    int dense_bw = 0;  // half-bandwidth of the coarsest operator's pattern (build.cpp)
    bool dense_valid = false, dense_fresh = false;
    int64_t its_fresh = 0;             // PCG iterations of the last solve that ran on a freshly inverted coarse operator
    bool asm_check_stale = false, asm_check_done = false;  // run_irls <-> assemble_values: verdict inside k_coarse_level
    DevBuf<unsigned long long> dense_mm;  // min / max / ticket of that verdict
    int stale_streak = 0;              // consecutive 'stale' verdicts of IRLS solves (assemble() in solver.hip)
    unsigned stale_skips = 0;
    bool dense_stale_pending = false;  // an asynchronous check (dense_check_async) asked for a re-inversion
    double dense_scale = 1.0, stale_spread = 1.1;
    DevBuf<double> dense_ref_diag, dense_ref_val;  // coarse operator the current inverse was computed from
    // parked inverses: l1decode keeps one per primal-dual iteration index, because the Hessian of
    // PD iteration p of outer iteration t+1 resembles that of (p, t), not that of (p-1, t+1)
    struct DenseSlot {
        DevBuf<double> inv, ref_diag, ref_val;
        double scale = 1.0;
        bool valid = false;
    };
    std::vector<DenseSlot> dense_parked;
    int dense_slot = 0;  // which slot the live dense_inv / dense_ref_* belong to
    DevBuf<double> dense_maxdiag;                  // largest diagonal entry of that operator (dead-pivot scale)
    int additive_top = 1;  // level 0 enters the preconditioner additively (no fine matrix pass)

    // PCG (level-0 sized). levels[0].b is the residual r, levels[0].x the pre-smoothed
    // iterate, levels[0].y the preconditioned residual z.
    DevBuf<double4> X, P, AP;
    bool kc_auto = false;  // opt.mg_kc was chosen from the structure (build.cpp), not given
    int l1_fused = 0;  // the PCG update kernel also does the level-1 down-sweep (build.cpp)
    int cg2 = 0;       // the PCG iteration runs as two launches (cgcg.hip; build.cpp decides)
    long long l0_far_entries = 0;  // level-0 SELL entry-columns outside the tile windows (loop closures)
    DevBuf<double> b2p;  // planar copy of the dense level's right-hand side (3 x ndense_pad; cgcg.hip)
    DevBuf<double4> R2;  // second residual buffer (l1_fused: the update kernel writes r out of place)
    DevBuf<double4> P2;  // second search-direction buffer: the fused p-update + SpMV ping-pongs P / P2
    DevBuf<double> part_pq, part_rr, part_rz, part_rz2, part_score;  // kMaxParts x 4
    DevBuf<double> scal;
    DevBuf<int> flags;

    // L1RA primal-dual work vectors (allocated on first use)
    DevBuf<double> pd;      // m-length planes
    DevBuf<double> pdn;     // nu-length planes
    DevBuf<double> pd_part; // reduction partials
    bool pd_ready = false;
    // l1decode -> assemble_values: the operand t of the Hessian system's right-hand side A' t; when the windowed assembly
    // takes it along (k_assemble0w<2>) it says so in pd_rhs_done and k_pd_rhs is not launched
    const double *pd_rhs_src = nullptr;
    bool pd_rhs_done = false;
    // L1RA solves its three coordinates concurrently: three solver clones (own stream, own
    // matrix values / vectors / dense level; static structure aliased from this handle)
    std::vector<std::unique_ptr<Graph>> l1_clones;
    bool is_clone = false;

    // host staging: one pinned block (PinPool): [0, 4 kMaxParts) doubles = reduction partials,
    // then SC_COUNT doubles + FL_COUNT ints = the read-back of scal / flags (ONE copy: the flags live
    // at the tail of the scal allocation)
    double *hpin = nullptr;
    double *h_part() { return hpin; }
    double *h_scal() { return hpin + 4 * kMaxParts; }
    int *h_flags() { return reinterpret_cast<int *>(hpin + 4 * kMaxParts + SC_COUNT); }
    // [4096, 4096 + 8 kMaxParts): two more partial arrays (l1pd.hip); [8192]: the sequence number a publishing
    // kernel stores last (publish_parts)
    int *h_seq() { return reinterpret_cast<int *>(hpin + 8192); }
    int pub_seq = 0;
    ~Graph() {
        bcr_up_release(*this);
        if (hpin) PinPool::get().give(hpin);
    }
    Graph() = default;
    Graph(const Graph &) = delete;
    Graph &operator=(const Graph &) = delete;

    // direct solve of a banded level-0 operator by block cyclic reduction (bcr.hip): block size (8 / 16 / 24 / 32;
    // 0 = the solves run through the PCG) and the half-bandwidth found at creation (-1: not looked at)
    int bcr_B = 0, band0 = -1;
    // run_irls, plain direct path: the ways back of the solve also make the step (K6) and leave the score's partial sums
    // in part_score, bcr_apply_slots() doubles; bcr_applied = the last solve did
    bool bcr_apply = false, bcr_applied = false;
    int bcr_up_held = 0;  // 1/1024ths of the device reserved for k_bcr_reduce_up's workgroups (bcr_up_reserve)
    bool pool_headroom = false;  // allocations made for this handle ask the device pool for half as much again (resident.hip)
    bool bcr_no_fused_up = false;  // a wait inside k_bcr_reduce_up gave up once (bcr_up_failed): level by level from then on
    std::unique_ptr<BcrState, BcrDeleter> bcr;
    std::vector<int> bcr_far_i, bcr_far_j, bcr_far_e;  // long-range edges (rows, edge id): Woodbury correction
    const double *bcr_wsrc = nullptr;                  // per-edge weights of the last assembly and whether the
    int bcr_wsquare = 0;                               // operator holds their squares (IRLS) -- assemble_values
    // the guard of the closure path (run_irls, precondition): a direct solve whose BAND part lost a pivot is repeated as
    // a CG solve of the full operator with the regularised direct solve (dead pivot -> the row's own diagonal entry) as
    // its preconditioner: bcr_guard switches bcr_solve to that mode, bcr_out redirects its result (nullptr: X)
    bool bcr_guard = false;
    bool bcr_last_guarded = false;  // the most recent linear solve of this handle was such a guarded one (bcr_residual)
    double4 *bcr_out = nullptr;
    // a shard of a sharded sequence solved directly (dist.hip): bcr_ext0 = a rank lies before this one
    bool bcr_shard = false;
    int bcr_ext0 = 0;
    std::vector<int> bcr_ghost_extcol;                 // per ghost view: row in the previous rank's last block or -1
    // ... whose sequence has loop closures (round 5): bcr_far_* list the closures with an endpoint on THIS rank (a row of
    // -1: that endpoint lives on another rank), bcr_far_gid their numbers in the global list every rank agrees on,
    // bcr_far_own whether this rank adds the closure's 1 / w to the Woodbury system (the owner of its first endpoint);
    // bcr_fix_*: the local rows whose diagonal carries the weight of a closure to a GHOST view (a ghost edge is
    // Dirichlet mass on the shard's diagonal; the band operator has to lose it as bcr_gather_row takes a local far entry out)
    std::vector<int> bcr_far_gid;
    std::vector<uint8_t> bcr_far_own;
    std::vector<int> bcr_fix_row, bcr_fix_off, bcr_fix_e;

    double last_score_sum = 0.0;
    double irls_settle = -1.0;  // run_irls: > 0 while the last step was small enough for the weights to have settled (assemble())
    int force_np = 0;  // sharded runs: consumers read this many pre-reduced partial rows (1)
    irotavg_stats stats{};
};

struct PrecInfo {
    int np_rz = 0, np_rz2 = 0;
};

// build.cpp (patterns on the host) / gbuild.hip (patterns on the device)
struct BuildTail {          // what the common tail of a build needs to know about the patterns
    int nlev = 1;
    int agg[kMaxLevels] = {0};
    int dense_bw = 0;           // half-bandwidth of the coarsest level's pattern
    bool l1_window_ok = false;  // every level-1 neighbour of a row within the 48-row window of its 32-row block
    bool l1_band8 = false;      // ... and within 8 rows
};
struct HierPlan {
    std::vector<int> n, agg;  // rows and aggregation factor per level (agg of the coarsest level: 0)
};
HierPlan plan_hierarchy(Graph &g, int n0, int64_t nnz0, bool far0);
// a build whose edge list lives on the device already (a resident view-graph, resident.hip): m pairs in the caller's
// view ids, one double4 [x y z w] per edge, and the caller id -> row map of this problem (nullptr: identity)
struct DevEdgeSrc {
    const int2 *I = nullptr;
    const double4 *QQ = nullptr;
    const int *relabel = nullptr;
};
int build_graph(Graph &g, const int32_t *I, const double *QQ, int64_t ldqq);
int build_graph_host(Graph &g, const int32_t *I, const double *QQ, int64_t ldqq);
int build_graph_device(Graph &g, const int32_t *I, const double *QQ, int64_t ldqq, const DevEdgeSrc *src = nullptr);
int finish_build(Graph &g, const BuildTail &T);

// solver entry points (solver.hip)
void launch_edge_residual(Graph &g, bool weights_to_one = false);
int ls_solve(Graph &g, const std::function<void()> *tail = nullptr, bool *tail_ran = nullptr);  // assemble (IRLS weights) + PCG; result in g.X
void launch_update_weights(Graph &g, int cost, double sigma, bool gated = false);
void launch_apply_step(Graph &g, bool gated);
double finish_apply_step(Graph &g);
double apply_step(Graph &g, bool gated = false, const std::function<void()> *behind = nullptr);
// Small read-backs that steer a solve without the runtime's copy + wait (~25 us of idle GPU per decision): ONE tiny
// kernel behind the producers copies up to three partial arrays into the handle's pinned block and stores a sequence
// number last (system scope); wait_published polls it (2 ms, then the stream is synchronised: long kernels, faults).
struct PubPart {
    const double *src;  // device
    double *dst;        // inside g.hpin
    int n;              // doubles
};
void publish_parts(Graph &g, const PubPart *parts, int nparts);
void wait_published(Graph &g);
int run_irls(Graph &g, int cost, double sigma, int max_iters, double change_th, int *iters,
             double *runtime, double *trace);
int run_l1ra(Graph &g, int max_iters, double change_th, int *iters, double *runtime,
             double *trace);
int l1decode_pd_dev(Graph &g, int coord_plane_from_er, const double *y_host, int pdmaxiter,
                    double *x_host, int *stuck, int out_component);
int time_kernel(Graph &g, int which, int reps, double *ms);
void release_l1_clones(Graph &g);
void normalise_rotations(Graph &g);
void fill(Graph &g, double *p, long long n, double v);
// device -> pinned host copy of scal + flags (one DMA) and stream synchronisation
void read_back_state(Graph &g);
void alloc_state(Graph &g);  // scal + flags (aliased tail) + the pinned block
void assemble(Graph &g, int mode, const double *wsrc, bool refresh_dense = true);
void assemble_values(Graph &g, int mode, const double *wsrc);  // the value refresh alone (no dense-level decision)
int pcg_solve(Graph &g, const std::function<void()> *tail = nullptr, bool *tail_ran = nullptr);
int pcg_solve_classic(Graph &g, const std::function<void()> *tail = nullptr, bool *tail_ran = nullptr);  // the round-1 recurrences (separate launches), whatever Graph::cg2 says
// AP = L p (p: g.P, ghost values: g.PG, the done word it skips its work behind: g.flags -- unless given)
void launch_spmv(Graph &g, const double4 *p = nullptr, const double4 *pg = nullptr, const int *flags = nullptr);
void launch_update(Graph &g, bool init, int par, int np_pq, const double4 *p = nullptr,
                   const double4 *rin = nullptr, double4 *rout = nullptr);
void launch_pupdate(Graph &g, int par, int first, const PrecInfo &pi, bool check = false, int np_rr = 1,
                    double rtol2 = 0.0);
PrecInfo precondition(Graph &g, int first, double rtol2, bool check = true);
// Chronopoulos-Gear pieces of the sharded PCG (solver.hip; buffers: u = P, w = AP, p = P2, s = levels[0].e)
void launch_form_u(Graph &g, double *part_g);
void launch_cgd_update(Graph &g, bool init, int par, int first, double rtol2, const double *part_g,
                       const double *part_d, const double *rr_in, double *rr_out);
int round_grid(long long gsz);
int grid_for_rows(const Level &L);
int grid_for_elems(long long n);
int normalise_host_rows(int64_t n, double *Q, int64_t ldq, int f);
// l1pd.hip: the primal-dual LP over a group of solvers (one on a single GPU; one per local shard in
// a sharded run, dist.hip)
struct PdMember {
    Graph *g;
    const uint8_t *eown;  // device mask: 1 = this member sums the edge (nullptr: all of them)
    const double *y;      // device pointer: the member's right-hand side of the LP (edge plane)
};
struct PdGroup {
    std::vector<PdMember> mem;
    long long m_global = 0;                           // edges of the whole problem
    std::function<void(double *, int, int)> combine;  // across processes, n host doubles; op 0 sum, 1 min, 2 max
    std::function<void()> halo_x;                     // ghost entries of every member's X <- their owners
    std::function<int()> solve;                       // H dx = rhs on the assembled values, dx -> X component 0
};
int l1decode_group(PdGroup &G, int pdmaxiter, int xplane, int *stuck);
void pd_prepare_graph(Graph &g);
void pd_pack_solution(Graph &g);  // pdn planes kPdXPlane0.. -> X (owned rows)
constexpr int kPdXPlane0 = 3;  // pdn planes 3, 4, 5 hold the three coordinates' solutions (PdnPlane N_X0..)
constexpr int IROTAVG_RETRY_STALE = 100;  // internal: pcg_solve_cg2 gave a speculative solve back (see run_irls)
// cgcg.hip: the two-launch PCG iteration
// tail: work to enqueue before the FIRST read-back of the solver state (kernels gated on the done flag);
// *tail_ran tells whether the solve was done at that read-back, i.e. whether the gated kernels ran
int pcg_solve_cg2(Graph &g, const std::function<void()> *tail = nullptr, bool *tail_ran = nullptr,
                  bool device_scale = false);  // device_scale: the inverse's scale is read from scal[SC_DSCALE]
void cg2_time_once(Graph &g, int which);
int cg2_phase_stamps(Graph &g, double *out, int n);
void cycle_levels(Graph &g, int from);  // solver.hip: levels[from].b/.x -> levels[from].y
// window.hip: single-kernel solve of small (sliding-window) problems
struct WindowSolver;
WindowSolver *window_solver_new();
void window_solver_delete(WindowSolver *w);
bool window_fits(int nv, int f, int ne);
int window_solve(WindowSolver &ws, int nv, int f, int ne, const int32_t *I, const double *QQ_aos,
                 double *Q_aos, double *weights, int l1_max, int irls_max, int cost, double sigma,
                 double change_th, int *l1_iters, int *irls_iters, int kernel = 0);
bool window_fits_wave(int nv, int f, int ne);
struct WinBatchItem {  // one problem of a batched launch: inputs as window_solve takes them, Q_aos updated in place
    int nv, f, ne;
    const int32_t *I;
    const double *QQ_aos;
    double *Q_aos;
    int l1_iters, irls_iters, status;
};
int window_solve_batch(WindowSolver &ws, int nb, WinBatchItem *items, int l1_max, int irls_max, int cost, double sigma,
                       double change_th);
// bcr.hip
void bcr_plan(Graph &g, const int32_t *I);  // sets Graph::bcr_B / band0 (capi.cpp, ahead of the build)
void bcr_plan_dev(Graph &g, const DevEdgeSrc &src);  // the same from an edge list on the device (needs g.stream)
int bcr_solve(Graph &g, int only = -1);     // levels[0] values / diagonal / right-hand side -> g.X, asynchronous
int bcr_levels(Graph &g);
int bcr_info(Graph &g, int64_t *out, int cap);
int bcr_residual(Graph &g, double *relres);  // ||b - A x|| / ||b|| per coordinate of the last direct solve (one pass over level 0)
int bcr_closures(Graph &g);  // loop closures the direct solver of this handle carries
// a closure solve that had to be repaired (CG with the direct solve as the preconditioner) is taken at this relative residual
// or better; above it the system counts as not solved (IROTAVG_ERR_SOLVER), on one GPU and on shards alike
constexpr double kClosureRepairAccept = 1e-8;
bool bcr_band_part_anchored(int64_t m, int f, int64_t nu, int B, const int32_t *I);  // the band part alone is positive definite (exact)
int bcr_apply_slots(Graph &g);
void bcr_gate(Graph &g, double *flags_copy = nullptr);  // flags_copy: two doubles' room for a copy of the four flag words (+ the verdict as bcr_gate_skip_word)
const int *bcr_gate_skip_word(Graph &g);  // flags[FL_DONE] = 1 unless the last direct solve with closures saw a dead pivot (flags[3] = their number)
void dense_invert_spd(Graph &g, double *A, int npad);  // in place, npad a multiple of 64 (dense.hip)
int bcr_stamps(Graph &g, int level, int chunk, double *out);  // development aid
int bcr_stamps_up(Graph &g, double *out);                      // development aid: 32 x 8 stamps of k_bcr_reduce_up
// the last direct solve went through the single-launch upper reduction and a workgroup of it gave up waiting (its solution
// is all NaN): clears the word, switches the handle to level-by-level launches and says so (one small synchronous read)
bool bcr_up_failed(Graph &g);
// the device word behind it while such a solve is the last one (kernels behind the solve skip their work when it is set),
// or nullptr
const int *bcr_fail_word(Graph &g);
// the sharded form (dist.hip): every rank reduces its range to its last block; the `world` separators are one chunk
struct BcrTop {
    int B = 0, world = 0;
    DevBuf<double> buf;         // [sepD | extD | extG | sepR | extR], `world` blocks each: filled by the ranks
    DevBuf<double> rec;         // the same data rank-major (a rank's five slices as one record): what an all-gather moves
    size_t record_doubles() const { return 3 * (size_t)B * B + 2 * (size_t)B * 3; }
    DevBuf<double> W, x, xtop;  // factor and solution of the separator system (8 blocks)
    size_t n_doubles() const { return (size_t)world * (3 * (size_t)B * B + 2 * (size_t)B * 3); }
    // loop closures on the sharded sequence (bcr_top_closures_*): r closures in the global list. xbuf is what the ranks
    // SUM (loopback: one buffer every shard adds to; RCCL / hosted wire: an all-reduce):
    //   [dep: r x world x B | S: npad x npad | T: npad x 3 | dead: npad | dead pivots: 1]
    // dep = what the local forward eliminations of a closure's incidence column leave on the separators (the right-hand
    // side of the separator system for that column), S / T = the ranks' shares of the Woodbury system, dead = closures of
    // weight 0 (flagged by their owner).
    int r = 0, npad = 0, nst = 0;
    DevBuf<double> xbuf, Dinv, topDinv, recR, recW, Ttop, lam;
    DevBuf<int> dead, cl_off, cl_owner, co_off, co_rec;
    DevBuf<int4> cl_step;
    DevBuf<int2> co_elim;
    size_t x_dep() const { return 0; }
    size_t x_S() const { return (size_t)r * world * B; }
    size_t x_T() const { return x_S() + (size_t)npad * npad; }
    size_t x_dead() const { return x_T() + (size_t)npad * 3; }
    size_t x_pivots() const { return x_dead() + (size_t)npad; }
    size_t x_doubles() const { return x_pivots() + 1; }
};
void bcr_top_alloc(BcrTop &T, int B, int world);
void bcr_top_to_record(BcrTop &T, int rank, hipStream_t st);   // buf's slices of `rank` -> rec[rank]
void bcr_top_from_records(BcrTop &T, hipStream_t st);          // every record -> buf
void bcr_shard_reduce(Graph &g, BcrTop &T, int rank);  // local reduction; this rank's slices of T.buf
void bcr_top_solve(Graph &g, BcrTop &T);               // after T.buf holds every rank's slices
void bcr_shard_back(Graph &g, BcrTop &T, int rank);    // -> g.X (owned rows)
// closures on a sharded sequence: r = their number in the global list (0: none). A solve runs
//   every rank: bcr_shard_reduce, bcr_shard_closures_forward (adds its shares to T.xbuf)
//   the wire:   the separators' gather, the SUM of T.xbuf
//   every rank: bcr_top_solve_closures (separator system, Woodbury system, lambda; redundantly),
//               bcr_shard_closures_correct, bcr_shard_back
void bcr_top_closures_alloc(Graph &g0, BcrTop &T, int r);
void bcr_shard_closures_forward(Graph &g, BcrTop &T, int rank);
void bcr_top_solve_closures(Graph &g, BcrTop &T);
void bcr_shard_closures_correct(Graph &g, BcrTop &T);
// dense.hip
void dense_refresh(Graph &g);
void dense_select_slot(Graph &g, int slot);
bool dense_is_stale(Graph &g, bool allow_repair = false);
bool dense_rescale_only(Graph &g);
bool dense_direct_solve(Graph &g);  // last resort of a single-level graph: Cholesky solve of levels[0].b -> X
void dense_check_async(Graph &g);  // same test, decision on the device (scal[SC_DSCALE], flags[FL_STALE])
int dense_apply_grid(const Graph &g);
void dense_apply(Graph &g, const double4 *b, double4 *y, bool check, bool dot, double *part_dot,
                 int np_rr, int first, double rtol2);

}  // namespace irh
struct irotavg_graph;
namespace irh {
// capi.cpp: a handle built from an edge list on the device; the Graph behind a handle
int graph_create_dev(irotavg_graph **out, int64_t m, int64_t n_total, int f, const DevEdgeSrc &src,
                     const irotavg_options *opt);
Graph &graph_of(irotavg_graph *h);

// resident.hip: the device-resident, growing copy of a view-graph behind rot_avg's global re-solves
struct Resident;
struct ResidentStage {  // pinned host blocks the caller fills before resident_rot_avg
    double *R;          // 9 per view of [view_lo, n_views): row-major poses
    uint8_t *fixed;     // their fixed flags
    int32_t *I;         // pairs (view ids) of the edges [edge_lo, n_edges)
    double *qq;         // their relative rotations, 4 per edge [x y z w]
    double *Q;          // OUT: n_views x 4 (AoS), the solved quaternion of every free view
};
Resident *resident_new();
void resident_delete(Resident *r);
void resident_invalidate(Resident &r);
long resident_views(const Resident &r);  // views / edges the device holds
long resident_edges(const Resident &r);
ResidentStage resident_stage(Resident &r, long n_views, long view_lo, long n_edges, long edge_lo);
int resident_rot_avg(Resident &r, long n_views, long view_lo, long n_edges, long edge_lo, int f,
                     const irotavg_options &opt, irotavg_rotavg_info &loc, bool timing, bool dry = false, int dry_a = -1,
                     int dry_b = -1);

inline double now_seconds() {
    using namespace std::chrono;
    return duration<double>(steady_clock::now().time_since_epoch()).count();
}

}  // namespace irh

/*
 * irotavg_hip.h -- C ABI of libirotavg_hip.so, the MI355X (gfx950) rotation-averaging core that
 * replaces the RAL library of ajparra/iRotAvg on the solve path.
 *
 * The reference has no FFI/plugin registry: its boundary is the C++ free-function API of
 * `namespace irotavg` in ral/l1_irls.hpp:81-112, called from ral/test.cpp:285-302 and
 * src/ViewGraph.cpp:1400-1417. Every entry point below names the reference interface it
 * replaces. include/irotavg/l1_irls.hpp is a header-only C++ shim with the reference's names on
 * top of this ABI; INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Data layout (identical to the reference's Eigen types, ral/l1_irls.hpp:43-51):
 *   Mat  : column-major fp64, leading dimension ld >= rows; quaternion columns [x, y, z, w];
 *   I_t  : m pairs of int32 (first = i, second = j), 0-based; the first f rows of Q are fixed;
 *   Vec  : contiguous fp64.
 * All pointers in this header are HOST pointers unless a name ends in `_dev`.
 *
 * Error convention: the reference prints to stderr and calls exit(-1); this ABI returns one of
 * the negative codes below and never throws or exits. There is NO CPU fallback: if no HIP
 * device is usable every compute entry point returns IROTAVG_ERR_NO_DEVICE.
 */
#ifndef IROTAVG_HIP_H
#define IROTAVG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IROTAVG_OK 0
#define IROTAVG_ERR_BAD_ARG (-1)
#define IROTAVG_ERR_NOT_SPANNING (-2) /* ral/l1_irls.cpp:970-977 */
#define IROTAVG_ERR_SOLVER (-3)       /* ral/l1_irls.cpp:149-177 (UMFPACK failure) / PCG breakdown */
#define IROTAVG_ERR_UNKNOWN_COST (-4) /* ral/l1_irls.cpp:723-726 */
#define IROTAVG_ERR_NOMEM (-5)
#define IROTAVG_ERR_HIP (-6)
#define IROTAVG_ERR_NO_DEVICE (-7)
#define IROTAVG_ERR_NOT_CONVERGED (-8) /* inner PCG hit its iteration cap -- after irls has re-inverted a re-used
                                          coarse inverse and, on a single level of <= 1024 views, tried the dense
                                          Cholesky solve; the iterate reached so far is left in place */

/* ral/l1_irls.hpp:56-57 -- the integer values are ABI */
enum irotavg_cost {
    IROTAVG_L2 = 0, IROTAVG_L1, IROTAVG_L15, IROTAVG_L05, IROTAVG_GEMAN_MCCLURE, IROTAVG_HUBER,
    IROTAVG_PSEUDO_HUBER, IROTAVG_ANDREWS, IROTAVG_BISQUARE, IROTAVG_CAUCHY, IROTAVG_FAIR,
    IROTAVG_LOGISTIC, IROTAVG_TALWAR, IROTAVG_WELSCH
};

/* Solver knobs that have no counterpart in the reference (it uses direct factorisations). */
typedef struct irotavg_options {
    double pcg_rtol;     /* stop when ||r||_2 <= pcg_rtol * ||b||_2 for every column; default 1e-10 */
    int pcg_max_iters;   /* default 2000 */
    int pcg_check_every; /* PCG iterations enqueued between host polls of the done flag; default 8 */
    int mg_levels_max;   /* cap on multigrid levels (1 = plain Jacobi-PCG); default 16 */
    int mg_agg0;         /* aggregate size of the finest level (0 = choose: 8, or the smallest power of two
                            that reaches the dense level) */
    int mg_agg;          /* aggregate size of the other levels (0 = choose; default) */
    int mg_dense_max;    /* coarsening stops at <= this many rows; that level is inverted densely
                            (blocked Gauss-Jordan on the GPU) and applied exactly; cap 2048;
                            0 (default) = chosen from the graph: 2048, or 1100 for a graph with
                            loop closures and <= 70k views (the sweep runs in most IRLS iterations) */
    double mg_omega;     /* damped-Jacobi factor; default 0.7 */
    double mg_kc;        /* coarse-correction scale (over-correction of the piecewise-constant aggregates);
                            0 (default) = choose: 2.0 for a graph without loop closures, 1.6 otherwise */
    int device;          /* HIP device ordinal; -1 = current device */
    int mg_multiplicative_top; /* 1: multiplicative V-cycle on level 0 (default 0: additive top level) */
    int dense_always_refresh;  /* 1: re-invert the dense coarse level at every solve (default 0: adaptive) */
    int no_window_kernel;      /* 1: irotavg_viewgraph_rot_avg never uses the single-kernel window path */
    int pcg_stall_accept;      /* a solve whose residual has stalled (not halved in 64 iterations: the
                                  attainable accuracy of an ill-conditioned system) is accepted if
                                  ||r||/||b|| <= 1e-6 (0, default) or <= 10^-k (k > 0); -1 = never: such
                                  a solve ends in IROTAVG_ERR_NOT_CONVERGED at pcg_max_iters */
    int no_fused_pspmv;        /* 1: no kernel fusion inside the PCG iteration (default 0: on one GPU the
                                  p-update runs inside the SpMV and, on graphs without loop closures, the
                                  level-1 down-sweep inside the update kernel) */
    int no_lowrank_repair;     /* 1: a non-uniformly changed coarse operator is always re-inverted (default 0:
                                  <= 64 deviating long-range entries are repaired by a low-rank update) */
    int pcg_classic;           /* 1: the PCG iteration always runs as separate launches (default 0: on one
                                  GPU a graph without loop closures runs it as two launches, cgcg.hip) */
    int band_direct;           /* a graph whose edges between free views all span <= 32 views -- a view sequence --
                                  except for at most 2048 long-range edges (loop closures) has a banded operator plus
                                  a low-rank part; its linear systems are then solved DIRECTLY -- on one GPU and on shards -- (block
                                  cyclic reduction + Woodbury correction, bcr.hip) instead of the PCG:
                                  0 (default) = when it has more than 2048 free views (smaller graphs are one
                                  dense level already), 1 = whenever the band allows, -1 = never */
    int inexact_outer;         /* 1: irls on the ITERATIVE solver solves the linear systems of its early iterations only as
                                  accurately as the outer iteration can use (round 5): while the last step was above
                                  50 x change_th the relative residual asked for is 0.01 change_th / last step (at most
                                  1e-4, never below pcg_rtol), i.e. the step is exact to ~1 % of change_th; the iterations
                                  near the fixed point -- the one that decides the stop and leaves the weights -- and the
                                  last one allowed are solved to pcg_rtol. 100k views / 2M edges with 2 % loop edges:
                                  27.8 -> 16.6 PCG iterations per solve, 13.4 -> 10.3 ms per irls, the same outer
                                  iterations, final rotations within 5e-8 rad (mean) of the all-exact run. 0 (default):
                                  every system to pcg_rtol, like the reference's QR (ral/l1_irls.cpp:536-556).
                                  IROTAVG_INEXACT=1 / 0 in the environment overrides the option. A direct solve
                                  (band_direct) has no tolerance and is not affected. */
} irotavg_options;

void irotavg_default_options(irotavg_options *opt);

/* Cumulative counters of a graph handle (since creation or the last reset). */
typedef struct irotavg_stats {
    int64_t pcg_solves;
    int64_t pcg_iters;       /* total PCG iterations over all solves */
    int64_t pcg_iters_last;  /* iterations of the most recent solve */
    int64_t outer_iters;     /* IRLS + L1RA outer iterations */
    int64_t edge_updates;    /* m * (IRLS outer iterations) */
    double seconds_irls;     /* wall seconds inside irotavg_graph_irls */
    double seconds_l1ra;
    int levels;              /* multigrid levels in use */
    int64_t level_rows[16];
    int64_t level_nnz[16];
    double last_relres[3];   /* ||r||/||b|| per column at the end of the last solve */
    int64_t pcg_stagnated;   /* solves accepted at a stalled residual above pcg_rtol (see pcg_stall_accept) */
    int64_t dense_inversions; /* full inversions of the dense coarse level */
    int64_t dense_repairs;    /* low-rank (Woodbury) repairs of that inverse instead of an inversion */
    int64_t pcg_handed_over;  /* solves whose single-reduction (Chronopoulos-Gear) recurrences stalled and that were
                                 repeated with the classic recurrences from the saved right-hand side */
    int64_t direct_solves;    /* linear systems solved by the banded direct solver (options.band_direct) */
    int64_t band;             /* half-bandwidth of the operator found at creation (-1: not looked at), and ... */
    int64_t band_block;       /* ... the block size of the direct solver (0: the solves run through the PCG) */
    int64_t direct_guarded;   /* direct solves with loop closures whose BAND part lost a pivot (a cost with exact-zero
                                 weights cut a view off its band neighbours while a closure may still hold it): repeated as a
                                 conjugate-gradient solve of the full operator preconditioned by the regularised direct solve */
    int64_t direct_dead_pivots; /* dead pivots of the band factor seen by the last such solve */
    int64_t direct_up_fallbacks; /* direct solves repeated level by level because a workgroup of the single-launch upper
                                    reduction gave up waiting for the others (another process held the device): the handle
                                    keeps the level-by-level launches from then on */
} irotavg_stats;

/* ---------------------------------------------------------------------------------------------
 * One-shot drop-ins: same arguments as the reference functions, host pointers in, host
 * pointers out. `A` of the reference signatures is derivable from (n, f, I) and is not passed.
 * ------------------------------------------------------------------------------------------ */

/* replaces irotavg::init_mst (ral/l1_irls.hpp:89, ral/l1_irls.cpp:915-979). Sequential sweep
 * semantics (result depends on edge order) -- runs on the host by design. */
int irotavg_init_mst(int64_t n, int64_t m, double *Q, int64_t ldq, const double *QQ,
                     int64_t ldqq, const int32_t *I, int f);

/* replaces irotavg::make_A (ral/l1_irls.hpp:91, ral/l1_irls.cpp:755-780): CSC arrays of the
 * m x (n-f) incidence matrix incl. the edge-drop quirk of :770-771. colptr: n-f+1 entries,
 * rowidx/vals: capacity 2m. Returns nnz or a negative error. */
int64_t irotavg_make_A(int n, int f, int64_t m, const int32_t *I, int64_t *colptr,
                       int64_t *rowidx, double *vals);

/* replaces irotavg::l1ra (ral/l1_irls.hpp:100-102, ral/l1_irls.cpp:851-912) */
int irotavg_l1ra(int64_t m, int64_t n_total, int f, const int32_t *I, const double *QQ,
                 int64_t ldqq, double *Q, int64_t ldq, int max_iters, double change_th, int *iter,
                 double *runtime);

/* replaces irotavg::irls (ral/l1_irls.hpp:104-107, ral/l1_irls.cpp:559-752). `weights` must
 * hold m doubles. `runtime` is WALL seconds (the reference reports clock() CPU seconds). */
int irotavg_irls(int64_t m, int64_t n_total, int f, const int32_t *I, const double *QQ,
                 int64_t ldqq, int cost, double sigma, double *Q, int64_t ldq, int max_iters,
                 double change_th, double *weights, int *iters, double *runtime);

/* replaces irotavg::quat_normalised (ral/l1_irls.hpp:112, ral/l1_irls.cpp:982-991) */
int irotavg_quat_normalised(int64_t n, double *Q, int64_t ldq, int f);

/* The reference's callers pass the SAME (I, QQ) to l1ra and then to irls (src/ViewGraph.cpp:1400-1417,
 * ral/test.cpp:295-301). irotavg_l1ra / irotavg_irls therefore KEEP the handle of their last call (one slot per
 * process): a following one-shot call with the same pointers, sizes and f whose content hashes equal (64 bits over every
 * word of I and of the four columns of QQ, computed from the caller's arrays on all host cores at every call -- an array
 * changed in place is a different graph) uploads Q only. A call in flight owns its handle (a concurrent call from another
 * thread builds its own); the kept handle holds device memory until irotavg_oneshot_cache_clear(), a call with another
 * graph, or process exit. irotavg_oneshot_cache(0) or IROTAVG_ONESHOT_CACHE=0 in the environment: every call builds and
 * destroys its handle as before round 5. */
void irotavg_oneshot_cache(int enable);
void irotavg_oneshot_cache_clear(void);
void irotavg_oneshot_cache_stats(int64_t *hits, int64_t *misses);

/* ---------------------------------------------------------------------------------------------
 * Handle API: the graph (edges, relative rotations, adjacency, multigrid hierarchy, work
 * vectors) stays resident in HBM across calls. One HIP stream per handle; a handle is not
 * thread-safe, distinct handles are independent.
 * ------------------------------------------------------------------------------------------ */
typedef struct irotavg_graph irotavg_graph;

/* Uploads I and QQ, builds the adjacency and the hierarchy. opt may be NULL. */
int irotavg_graph_create(irotavg_graph **g, int64_t m, int64_t n_total, int f, const int32_t *I,
                         const double *QQ, int64_t ldqq, const irotavg_options *opt);
void irotavg_graph_destroy(irotavg_graph *g);

int irotavg_graph_set_rotations(irotavg_graph *g, const double *Q, int64_t ldq); /* H2D, all n_total rows */
int irotavg_graph_get_rotations(irotavg_graph *g, double *Q, int64_t ldq);       /* D2H */
/* device-side snapshot / restore of the n_total rotations (e.g. to re-run a solve from the same
 * initial rotations without a host round trip) */
int irotavg_graph_snapshot_rotations(irotavg_graph *g);
int irotavg_graph_restore_rotations(irotavg_graph *g);
int irotavg_graph_get_weights(irotavg_graph *g, double *weights);                /* D2H, m */
int irotavg_graph_set_weights(irotavg_graph *g, const double *weights);          /* H2D, m */

/* irls / l1ra / quat_normalised on the resident graph (same semantics as the one-shot calls).
 * A handle on the banded direct solver that carries loop closures checks every closure solve against the FULL system
 * (relative residual options.pcg_rtol, 1e-10 by default; repaired by conjugate gradients with the direct solve as the preconditioner when it is
 * above). IROTAVG_ERR_SOLVER from irls then means: the band part alone is next to singular under the closures (robust
 * weights at their floor over whole stretches of a thin chain) and the repair stalled above 1e-8 -- the rotations are
 * what the last good iteration left; such a graph belongs to the iterative solver (options.band_direct = -1). The
 * one-shot irotavg_irls and irotavg_viewgraph_rot_avg repeat the call that way by themselves; the handle API and the
 * sharded handle (irotavg_dist_irls: the same check, by iterative refinement on the sharded operator) report it. */
int irotavg_graph_irls(irotavg_graph *g, int cost, double sigma, int max_iters, double change_th,
                       int *iters, double *runtime, double *score_trace /* may be NULL */);
int irotavg_graph_l1ra(irotavg_graph *g, int max_iters, double change_th, int *iter,
                       double *runtime, double *score_trace /* may be NULL */);
int irotavg_graph_quat_normalised(irotavg_graph *g);

int irotavg_graph_get_stats(irotavg_graph *g, irotavg_stats *out);
void irotavg_graph_reset_stats(irotavg_graph *g);
int irotavg_graph_synchronize(irotavg_graph *g);

/* ---------------------------------------------------------------------------------------------
 * Stage-level entry points on the resident graph (used by the parity tests, which check every
 * kernel against the oracle, and by bench.py's roofline leg).
 * ------------------------------------------------------------------------------------------ */

/* K1: r_k = log(Qinv_j (x) QQ_k (x) Q_i) for every edge (ral/l1_irls.cpp:109-127 + :498-532). */
int irotavg_graph_edge_residual(irotavg_graph *g);
/* D2H of the residual rows: out is m x 3 column-major, ld >= m. They are what the last irotavg_graph_edge_residual call
 * computed; irotavg_graph_irls / _l1ra use the planes as scratch and leave them unspecified (the residuals of the
 * rotations before OR after the last step, depending on the solver path). */
int irotavg_graph_get_residuals(irotavg_graph *g, double *out, int64_t ld);
/* one weighted least-squares solve with the current weights and residuals
 * (ral/l1_irls.cpp:596-612); X is n_u x 3 column-major (ld >= n_u), may be NULL. */
int irotavg_graph_ls_solve(irotavg_graph *g, double *X, int64_t ldx);
/* weight update from the current X and residuals (ral/l1_irls.cpp:614-727) */
int irotavg_graph_update_weights(irotavg_graph *g, int cost, double sigma);
/* score + exp map + Q update (ral/l1_irls.cpp:729-737); returns the score */
int irotavg_graph_apply_step(irotavg_graph *g, double *score);
/* one coordinate of the primal-dual LP (ral/l1_irls.cpp:228-468) on the resident graph:
 * y is a host vector of m entries, x receives n_u entries. */
int irotavg_graph_l1decode_pd(irotavg_graph *g, const double *y, int pdmaxiter, double *x,
                              int *stuck);

/* Times `reps` back-to-back launches of one kernel with HIP events on the handle's stream and
 * returns the mean milliseconds per launch. which: 1 = K1 edge_residual, 2 = K2 weight update
 * (Geman-McClure), 3 = matrix assembly (all levels, without the dense inversion), 4 = level-0 SpMV
 * (q = L p), 5 = one preconditioner application, 6 = so(3) step kernel (non-destructive variant),
 * 7 = dense coarse-level inversion (blocked Gauss-Jordan), 8 = the PCG p-update fused into the
 * level-0 SpMV (what a single-GPU solve of a graph without far entries runs instead of 4;
 * IROTAVG_ERR_BAD_ARG if this graph's PCG does not use it), 11 = K2 and the next iteration's K1 in one pass over the
 * edges (what the direct solver's irls loop runs from its second iteration on). */
int irotavg_graph_time_kernel(irotavg_graph *g, int which, int reps, double *ms_per_launch);

/* The banded direct solver of this handle (options.band_direct; irotavg_amd/csrc/bcr.hip): info[0] = block size (0: the
 * handle's systems run through the PCG -- nothing else is written), info[1] = levels L, then per level l < L three
 * values: blocks, chunks of eight blocks (= workgroups of its two launches), blocks that come from the level below
 * (fewer than `blocks` only on a mixed level 1, whose other blocks are level-0 blocks no chunk reduced); after the
 * levels one more value: the long-range edges (loop closures) the solver handles by its Woodbury correction. which of
 * irotavg_graph_time_kernel: 19 = a whole solve (2 L launches), 20 + l = the reduction of level l, 40 + l = its way
 * back. Returns the number of values written (<= cap) or a negative error. */
int irotavg_graph_direct_info(irotavg_graph *g, int64_t *info, int cap);

/* The direct solver has no tolerance and no residual test of its own (SuiteSparseQR / UMFPACK in ral/l1_irls.cpp:131-184,
 * 536-556 have none either). On demand: relres[c] = ||b - A x||_2 / ||b||_2 for the three coordinates of the handle's most
 * recent direct solve -- the level-0 operator as last assembled (loop closures included), its right-hand side, the
 * solution the last step was made from; one pass over level 0 and a host round trip, also stored in
 * irotavg_stats.last_relres. IROTAVG_ERR_BAD_ARG if the handle's systems do not run through the direct solver or none
 * has been solved yet. */
int irotavg_graph_direct_residual(irotavg_graph *g, double *relres);

/* Testing aid: fingerprint of the handle's static structure -- every index array the build produces (edge
 * streams, boundary slots, per level the SELL-64 pattern and the value-refresh maps) as one 64-bit FNV-1a hash
 * each, followed by the scalars that choose kernels (level shapes, far-entry count, fused-assembly / two-launch
 * flags, dense level size and bandwidth). The handle is built on the device (irotavg_amd/csrc/gbuild.hip) or,
 * with IROTAVG_HOST_BUILD=1, for shards and for small graphs, on the host (build.cpp): both must give the same
 * fingerprint (tests/test_gpu_build.py). Returns the number of values written (<= cap) or a negative error. */
int irotavg_graph_fingerprint(irotavg_graph *g, uint64_t *out, int cap);

/* ---------------------------------------------------------------------------------------------
 * View-graph side: OpenCV-free counterpart of ViewGraph / Pose (src/ViewGraph.hpp:54-75,
 * src/Pose.hpp:35-59). Rotations are row-major 3x3 doubles (cv::Matx33d layout). Connections
 * carry R_ij with R_j = R_ij R_i for the pair (i < j). The ORB / essential-matrix front-end that
 * produces them is not part of this library.
 * ------------------------------------------------------------------------------------------ */
typedef struct irotavg_viewgraph irotavg_viewgraph;

typedef struct irotavg_rotavg_info {
    int skipped;       /* 0 solved; 1 fewer than 2 views (:1270); 2 too few edges (:1313);
                          3 too few vertices (:1318); 4 no free view left */
    int n_views, n_edges, n_fixed;  /* size of the extracted sub-problem */
    int l1_iters, irls_iters;
    double l1_runtime, irls_runtime;
} irotavg_rotavg_info;

int irotavg_viewgraph_create(irotavg_viewgraph **vg, const irotavg_options *opt /* may be NULL */);
void irotavg_viewgraph_destroy(irotavg_viewgraph *vg);
/* appends a view with absolute rotation R (NULL = identity, Pose() default); returns its index,
 * which plays the role of frame().id() (ids only advance for admitted frames, src/IRotAvg.cpp:280-284) */
int irotavg_viewgraph_add_view(irotavg_viewgraph *vg, const double R[9]);
int irotavg_viewgraph_num_views(const irotavg_viewgraph *vg);
/* replaces View::connect (src/View.hpp:92, src/ViewGraph.cpp:1438-1455): returns 1 if the
 * connection was added, 0 if the pair was already connected. Rij is the rotation from view i to
 * view j (R_j = R_ij R_i); the pair is kept under (min, max), so a call with i > j stores Rij^T */
int irotavg_viewgraph_connect(irotavg_viewgraph *vg, int i, int j, const double Rij[9]);
/* replace ViewGraph::fixPose / isPoseFixed / countFixedPoses (src/ViewGraph.hpp:69-73) */
int irotavg_viewgraph_fix_pose(irotavg_viewgraph *vg, int idx, const double R[9]);
int irotavg_viewgraph_is_pose_fixed(const irotavg_viewgraph *vg, int idx);
int irotavg_viewgraph_count_fixed_poses(const irotavg_viewgraph *vg);
/* Pose::R() / Pose::setR() (src/Pose.hpp:47-51) */
int irotavg_viewgraph_get_pose(const irotavg_viewgraph *vg, int idx, double R[9]);
int irotavg_viewgraph_set_pose(irotavg_viewgraph *vg, int idx, const double R[9]);
/* replaces ViewGraph::rotAvg(int winSize) (src/ViewGraph.hpp:75, src/ViewGraph.cpp:1263-1435).
 * Sub-problems with <= 64 free views / <= 640 edges (every rotAvg(10) call) run as ONE kernel
 * launch (irotavg_amd/csrc/window.hip); larger ones through the graph handle path. */
int irotavg_viewgraph_rot_avg(irotavg_viewgraph *vg, int win_size, irotavg_rotavg_info *info);
/* Global re-solves (win_size >= the number of views: rotAvg(5000000) after a loop closure, src/IRotAvg.cpp:371-378) of a
 * graph with >= 20000 connections run on a DEVICE-RESIDENT, growing copy of the graph (irotavg_amd/csrc/resident.hip):
 * edge records, poses and the fixed mask stay in HBM between calls and a call sends only what changed since the last one
 * (views admitted, poses moved by sliding windows, new connections); results are bit-identical to extracting and
 * rebuilding the whole problem (IROTAVG_NO_RESIDENT=1 in the environment forces that).
 * irotavg_viewgraph_prepare runs that pipeline once on the graph as it is and drops the result (no pose changes): the
 * one-time costs of a process (device allocations, kernel code loads, streams) are then paid before the first loop
 * closure instead of inside it. Optional; IROTAVG_OK also when there is nothing to prepare. */
int irotavg_viewgraph_prepare(irotavg_viewgraph *vg);
/* The same for n DIFFERENT view-graphs at once (a server that tracks many sequences; the reference has one graph
 * per process, src/IRotAvg.cpp). The windows of different graphs are independent: those that fit the wave-resident
 * kernel (every rotAvg(10) of a sequence linked to <= 4 predecessors) are solved by ONE launch with a workgroup per
 * window -- one window alone keeps 1 of the 256 compute units busy --, the others run one by one. Results are
 * those of n separate irotavg_viewgraph_rot_avg calls. infos: n entries or NULL. Returns the first error. */
int irotavg_viewgraph_rot_avg_batch(irotavg_viewgraph *const *vgs, int n, int win_size, irotavg_rotavg_info *infos);

/* rmat2quat (src/ViewGraph.cpp:1175-1203) / q.normalized().toRotationMatrix() (:1426-1433);
 * row-major R, q = [x y z w] */
void irotavg_rmat2quat(const double R[9], double q[4]);
void irotavg_quat2rmat(const double q[4], double R[9]);
/* replaces ViewGraph::savePoses (src/ViewGraph.hpp:64, src/ViewGraph.cpp:1206-1231): lines
 * `id \t qw \t qx \t qy \t qz \t tx \t ty \t tz`; t = 3 doubles per view or NULL (zeros) */
int irotavg_viewgraph_save_poses(const irotavg_viewgraph *vg, const char *filename, const double *t);

/* The single-kernel window pipeline on caller data (layout as irotavg_l1ra / irotavg_irls):
 * l1ra(l1_iters) then irls(cost, sigma, irls_iters), change_th for both, in one launch.
 * IROTAVG_ERR_BAD_ARG if the problem does not fit the kernel's limits. */
int irotavg_window_solve(int64_t m, int64_t n_total, int f, const int32_t *I, const double *QQ,
                         int64_t ldqq, double *Q, int64_t ldq, int cost, double sigma, int l1_iters,
                         int irls_iters, double change_th, double *weights, int *l1_out, int *irls_out);
/* Same with an explicit kernel choice: 0 = automatic (what irotavg_window_solve and rot_avg do),
 * 1 = the general LDS kernel (<= 64 free views, <= 640 edges, <= 320 views), 2 = the wave-resident
 * kernel for the usual rotAvg(10) size (<= 16 free views, <= 64 edges; BAD_ARG beyond). */
int irotavg_window_solve_kernel(int64_t m, int64_t n_total, int f, const int32_t *I, const double *QQ,
                                int64_t ldqq, double *Q, int64_t ldq, int cost, double sigma,
                                int l1_iters, int irls_iters, double change_th, double *weights,
                                int *l1_out, int *irls_out, int kernel);

/* ---------------------------------------------------------------------------------------------
 * Multi-GPU: the IRLS solve sharded by contiguous ranges of free views; one
 * process per GPU, RCCL over xGMI for the halo exchanges and the scalar all-reduces. Every
 * process passes the SAME global graph; each keeps its shard. Bootstrap: rank 0 calls
 * irotavg_dist_unique_id and broadcasts the 128 bytes (e.g. with torch.distributed), then every
 * rank calls irotavg_dist_create(world, rank, id, ...). With unique_id128 == NULL the handle holds
 * ALL `world` shards in this process on the current GPU and exchanges through an in-process
 * loopback (no RCCL): the sharded algebra can then be checked on a single GPU.
 * ------------------------------------------------------------------------------------------ */
typedef struct irotavg_dist irotavg_dist;
/* A host-staged wire for the sharded solve: the library stages device data through host buffers and
 * calls back into the application, which moves the bytes with whatever it has (e.g.
 * torch.distributed over gloo). One process = one shard, as with RCCL. Used where RCCL cannot run
 * (two ranks on ONE GPU in the multi-process test, tests/test_gpu_dist_mp.py) -- the control flow of
 * the sharded path is the same, only the wire differs. Both callbacks return 0 on success. */
typedef struct irotavg_transport {
    void *ctx;
    /* in-place reduction of n doubles over all ranks; op: 0 sum, 1 min, 2 max */
    int (*allreduce)(void *ctx, double *buf, int n, int op);
    /* one point-to-point round: for q < npeers send send_cnt[q] doubles from send + send_off[q] to rank
     * peers[q] and receive recv_cnt[q] doubles from it into recv + recv_off[q] */
    int (*exchange)(void *ctx, int npeers, const int *peers, const double *send, const int64_t *send_off,
                    const int64_t *send_cnt, double *recv, const int64_t *recv_off, const int64_t *recv_cnt);
} irotavg_transport;
int irotavg_dist_unique_id(void *out128);
int irotavg_dist_create(irotavg_dist **d, int world, int rank, const void *unique_id128, int64_t m,
                        int64_t n_total, int f, const int32_t *I, const double *QQ, int64_t ldqq,
                        const irotavg_options *opt);
int irotavg_dist_create_hosted(irotavg_dist **d, int world, int rank, const irotavg_transport *transport,
                               int64_t m, int64_t n_total, int f, const int32_t *I, const double *QQ,
                               int64_t ldqq, const irotavg_options *opt);
void irotavg_dist_destroy(irotavg_dist *d);
int irotavg_dist_set_rotations(irotavg_dist *d, const double *Q, int64_t ldq); /* GLOBAL n_total x 4 */
int irotavg_dist_get_rotations(irotavg_dist *d, double *Q, int64_t ldq); /* writes the rows this process owns */
/* device-side snapshot / restore of the local shards' rotations (as irotavg_graph_snapshot_rotations) */
int irotavg_dist_snapshot_rotations(irotavg_dist *d);
int irotavg_dist_restore_rotations(irotavg_dist *d);
int irotavg_dist_get_weights(irotavg_dist *d, double *weights);          /* writes its local edges */
int irotavg_dist_irls(irotavg_dist *d, int cost, double sigma, int max_iters, double change_th,
                      int *iters, double *runtime, double *score_trace);
/* replaces irotavg::l1ra (ral/l1_irls.hpp:100-102) on the sharded graph: the callers run l1ra THEN
 * irls (src/ViewGraph.cpp:1400-1417, ral/test.cpp:295-301) */
int irotavg_dist_l1ra(irotavg_dist *d, int max_iters, double change_th, int *iters, double *runtime,
                      double *score_trace);
int irotavg_dist_get_stats(irotavg_dist *d, irotavg_stats *out);
/* What the sharded handle actually runs on, so that a scaling run can be checked: info[0] = wire
 * (0 loopback: all shards in this process, 1 RCCL, 2 the caller's host-staged transport) + 16 when the halo of a
 * closure-free sharded sequence travels as ONE all-gather of a fixed boundary record per rank instead of point-to-point
 * messages between neighbours (round 6; RCCL and loopback), info[1] = ranks of
 * the RCCL communicator as RCCL reports them (ncclCommCount; 0 without one), info[2] = shards held by this
 * process, info[3] = world size the graph is partitioned for, info[4] = ghost views of this process's
 * shards (halo rows received per exchange), info[5] = peers of shard 0, info[6] = block size of the sharded direct
 * solver (a view sequence, also with up to 2048 loop closures: every rank reduces its range of the banded operator to
 * its last block, one gather, the `world` separators solved by every rank; 0: the sharded PCG), info[7] = the loop
 * closures that solver carries (Woodbury correction across the ranks: besides the gather the ranks SUM one buffer of
 * r x world x B + r^2 + 4 r doubles per linear solve). */
int irotavg_dist_info(irotavg_dist *d, int64_t info[8]);
/* Diagnostic of a first multi-GPU run -- where an IRLS iteration of the sharded handle spends its time. enable != 0:
 * clear the counters and switch the phase clock ON for the following irotavg_dist_irls calls (the stream is drained at
 * every phase boundary: such calls are slower than undisturbed ones, never time them); enable == 0: switch it off. Either
 * way the means collected so far are returned first: us_per_iteration[0] local edge kernels + assembly, [1] local
 * reductions (closures' forward eliminations; the whole solve on the sharded PCG), [2] gather of the separators, [3] sum
 * of the closures' buffer, [4] separator system + corrections + ways back, [5] halo of the step, [6] weights + rotation
 * update, [7] score all-reduce; *iterations = IRLS iterations they are means over. Either pointer may be NULL. */
int irotavg_dist_timing(irotavg_dist *d, int enable, double us_per_iteration[8], int64_t *iterations);
int irotavg_dist_plan(irotavg_dist *d, int local_index, int64_t counts[6], int *peers, int *send_cnt,
                      int *recv_cnt, int cap);
/* host-only partition plan of one rank (no GPU): see irotavg_amd/csrc/dist.hip */
int irotavg_dist_plan_host(int world, int rank, int64_t m, int64_t n_total, int f, const int32_t *I,
                           int64_t counts[6], int32_t *ghosts_out, int64_t ghosts_cap,
                           int32_t *send_out, int64_t send_cap, int32_t *edges_out, int64_t edges_cap,
                           int *peers, int *send_cnt, int *recv_cnt, int peers_cap);

/* library / device info */
const char *irotavg_version(void);
int irotavg_device_count(void);
/* Handles draw their device buffers from a per-device cache so that create/destroy cycles (the
 * one-shot calls, rot_avg's global re-solves) do not pay hipMalloc/hipFree each time. This returns
 * the cached blocks to the driver; result = bytes released. IROTAVG_POOL_LIMIT_MB (environment)
 * bounds the cache (default 16384; 0 = no caching). */
int64_t irotavg_trim_memory(void);
const char *irotavg_error_string(int code);

#ifdef __cplusplus
}
#endif
#endif /* IROTAVG_HIP_H */

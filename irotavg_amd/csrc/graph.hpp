// graph.hpp -- the device-resident view-graph handle behind the C ABI.
#pragma once
#include <chrono>

#include "common.hpp"

namespace irh {

// scalar block shared by the PCG kernels (device memory, doubles)
enum ScalIdx : int {
    SC_RZ0 = 0,   // rz of parity 0 (3 values, padded to 4)
    SC_RZ1 = 4,   // rz of parity 1
    SC_BB = 8,    // ||b||^2 per column
    SC_RELRES = 12,  // ||r||/||b|| per column at the last check
    SC_COUNT = 16
};
// int flags block
enum FlagIdx : int {
    FL_DONE = 0,   // 0 running, 1 converged, 2 breakdown (non-finite scalar)
    FL_ITERS = 1,  // PCG iterations performed
    FL_COUNT = 4
};

struct Graph {
    int64_t m = 0, n_total = 0, mpad = 0;
    int f = 0, nu = 0;
    irotavg_options opt{};
    hipStream_t stream = nullptr;
    int device = 0;

    // edges: SoA streams for the edge-parallel kernels
    DevBuf<int> ei, ej;
    DevBuf<uint8_t> eflag;
    DevBuf<double> qq;  // 4 planes (x,y,z,w) of mpad doubles: the reference's col-major QQ
    DevBuf<double> er;  // 3 planes (rx,ry,rz) of mpad doubles: rotation-vector residual per edge
    DevBuf<double> dw;  // IRLS weights d_k (m)
    // views
    DevBuf<double4> Q;  // n_total quaternions [x y z w], gather-friendly AoS
    DevBuf<double4> Qsnap;

    // level-0 adjacency extras (CSR itself lives in levels[0])
    DevBuf<uint32_t> slot_eid;  // per inner slot: (edge id << 1) | (row is the j endpoint)
    DevBuf<int> bptr;           // per row: boundary slots (other endpoint fixed / self loop)
    DevBuf<uint32_t> beid;
    DevBuf<uint8_t> bflag;

    std::vector<Level> levels;
    DevBuf<double> dense_inv;  // coarsest level inverse, n x n row-major
    int ndense = 0;

    // PCG (level-0 sized). levels[0].b is the residual r, levels[0].x the pre-smoothed
    // iterate, levels[0].y the preconditioned residual z.
    DevBuf<double4> X, P, AP;
    DevBuf<double> part_pq, part_rr, part_rz, part_score;  // kMaxParts x 4
    DevBuf<double> scal;
    DevBuf<int> flags;

    // L1RA primal-dual work vectors (allocated on first use)
    DevBuf<double> pd;      // m-length planes
    DevBuf<double> pdn;     // nu-length planes
    DevBuf<double> pd_part; // reduction partials
    bool pd_ready = false;

    // host staging
    std::vector<double> h_part;

    irotavg_stats stats{};
};

// build.cpp
int build_graph(Graph &g, const int32_t *I, const double *QQ, int64_t ldqq);

// solver entry points (solver.hip)
void launch_edge_residual(Graph &g);
int ls_solve(Graph &g);  // assemble (IRLS weights) + PCG; result in g.X
void launch_update_weights(Graph &g, int cost, double sigma);
double apply_step(Graph &g);
int run_irls(Graph &g, int cost, double sigma, int max_iters, double change_th, int *iters,
             double *runtime, double *trace);
int run_l1ra(Graph &g, int max_iters, double change_th, int *iters, double *runtime,
             double *trace);
int l1decode_pd_dev(Graph &g, int coord_plane_from_er, const double *y_host, int pdmaxiter,
                    double *x_host, int *stuck, int out_component);
int time_kernel(Graph &g, int which, int reps, double *ms);
void normalise_rotations(Graph &g);
void fill(Graph &g, double *p, long long n, double v);
void assemble(Graph &g, int mode, const double *wsrc);
int pcg_solve(Graph &g);
int normalise_host_rows(int64_t n, double *Q, int64_t ldq, int f);

inline double now_seconds() {
    using namespace std::chrono;
    return duration<double>(steady_clock::now().time_since_epoch()).count();
}

}  // namespace irh

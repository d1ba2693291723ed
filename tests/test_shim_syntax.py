"""The IROTAVG_SHIM_EIGEN branch of include/irotavg/l1_irls.hpp -- the branch a real iRotAvg build
takes -- cannot be compiled against Eigen here (not installed). This is a SYNTAX check only: the
shim plus callers written (by us) the way the reference's two callers use the interface
(ral/test.cpp:161-326, src/ViewGraph.cpp:1365-1434) parse against a mock of the handful of Eigen
declarations they use (tests/mock_eigen). It says nothing about parity.

The same translation units also compile and LINK against the shim's Eigen-free branch and
libirotavg_hip.so (every name the callers use exists in both branches)."""
import os
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# every name of ral/l1_irls.hpp:40-112 the reference's callers use through `irotavg::`
CALLER = r"""
#include <iostream>
#include "irotavg/l1_irls.hpp"
#if defined(WANT_EIGEN_BRANCH) && !defined(IROTAVG_SHIM_EIGEN)
#error "the Eigen branch of the shim was not selected"
#endif

// the pipeline of the demo binary (ral/test.cpp:273-302), names unqualified as there
static int like_the_demo() {
    using namespace irotavg;
    const int n = 3;
    int f = 1;
    I_t I;
    I.push_back(std::make_pair(0, 1));
    I.push_back(std::make_pair(1, 2));
    I.push_back(std::make_pair(0, 2));            // a cycle that does not close: non-zero residuals
    const long m = (long)I.size();
    Mat QQ(m, 4), Q(n, 4);
    for (long k = 0; k < QQ.rows(); k++) QQ.row(k) << 0.01 * (double)(k + 1), 0.002, -0.005, 1;
    Q.row(0) << 0, 0, 0, 1;                       // ral/test.cpp:279
    init_mst(Q, QQ, I, f);                        // :286
    SpMat A = make_A(n, f, I);                    // :288
    int iters = 0;
    double runtime = 0;
    l1ra(QQ, I, A, Q, f, 100, 1e-3, iters, runtime);            // :295
    Vec weights(m);                                              // :299
    Cost cost = Geman_McClure;
    irls(QQ, I, A, cost, 0.0873, Q, f, 100, 1e-3, weights, iters, runtime);  // :300
    quat_normalised(Q, f);                                       // :302
    std::cout << cost << " " << Q(1, 3) << " " << weights(0) << std::endl;   // :314-326
    return (int)A.rows();
}

// the pipeline of ViewGraph::rotAvg (src/ViewGraph.cpp:1365-1434), names qualified as there
static double like_the_viewgraph(long num_of_vertices, long num_of_edges, int f) {
    std::vector<std::pair<int, int> > edges;
    for (long k = 0; k + 1 < num_of_vertices; k++) edges.push_back(std::make_pair((int)k, (int)k + 1));
    edges.push_back(std::make_pair(0, (int)num_of_vertices - 1));
    irotavg::I_t I(edges.begin(), edges.end());
    irotavg::Mat Q(num_of_vertices, 4);
    for (long k = 0; k < num_of_vertices; k++) Q.row(k) << 0, 0, 0, 1;       // :1378
    if (f == 0) {
        Q.row(0) << 0, 0, 0, 1;                                              // :1384
        f = 1;
    }
    irotavg::Mat QQ(num_of_edges, 4);
    for (long i = 0; i < num_of_edges; i++) QQ.row(i) << 0.01 * (double)(i + 1), 0.004 * (double)(2 - i), 0.003, 1;  // :1392
    // (every coordinate needs a non-zero residual: an all-zero one makes l1decode_pd divide by zero --
    //  in the reference, too: ral/l1_irls.cpp:245-281, then exit(-1) at :153)
    irotavg::SpMat A = irotavg::make_A((int)num_of_vertices, f, I);          // :1400
    const double change_th = .001;
    const int l1_iters = 100;
    int l1_iters_out;
    double l1_runtime;
    irotavg::l1ra(QQ, I, A, Q, f, l1_iters, change_th, l1_iters_out, l1_runtime);   // :1407
    const int irls_iters = 100;
    int irls_iters_out;
    double irls_runtime;
    irotavg::Vec weights(num_of_edges);                                      // :1412
    irotavg::Cost cost = irotavg::Cost::Geman_McClure;                       // :1413
    double sigma = 5 * M_PI / 180.0;
    irotavg::irls(QQ, I, A, cost, sigma, Q, f, irls_iters, change_th, weights, irls_iters_out,
                  irls_runtime);                                             // :1416
    double tr = 0;
    for (long k = f; k < num_of_vertices; k++) {
        irotavg::Quat q(Q(k, 3), Q(k, 0), Q(k, 1), Q(k, 2));                 // :1426
        q = q.normalized();                                                  // :1427
        irotavg::Mat R = q.toRotationMatrix();                               // :1429
        R.transposeInPlace();                                                // :1430
        const double *Rcv = R.data();                                        // :1431
        tr += Rcv[0] + Rcv[4] + Rcv[8];
    }
    return tr;
}

// the remaining typedefs of ral/l1_irls.hpp:40-51
static double the_other_names() {
    irotavg::Vec3 v3;
    irotavg::Vec4 v4;
    v3(0) = 1.0;
    v4(3) = 1.0;
    irotavg::T t(0, 1, 2.0);
    irotavg::Long big = 1;
    irotavg::Quat q(1, 0, 0, 0);
    q.normalize();
    return v3(0) + v4(3) + t.value() + (double)t.row() + (double)t.col() + (double)big + q.w() + EPS +
           (irotavg::DBL_MAX_ > 1 ? 1 : 0);
}

int main() {
    int r = like_the_demo();
    double t = like_the_viewgraph(3, 3, 0);
    std::cout << r << " " << t << " " << the_other_names() << std::endl;
    return 0;
}
"""


def _write(d):
    src = os.path.join(d, "caller.cpp")
    with open(src, "w") as fh:
        fh.write(CALLER)
    return src


def test_eigen_branch_of_the_shim_parses_against_the_mock():
    with tempfile.TemporaryDirectory() as d:
        src = _write(d)
        cmd = ["g++", "-std=c++11", "-fsyntax-only", "-Wall", "-DWANT_EIGEN_BRANCH", "-I",
               os.path.join(ROOT, "tests", "mock_eigen"), "-I", os.path.join(ROOT, "include"), src]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def _build_plain(d):
    import irotavg_amd.buildlib as bl
    bl.build()
    src = _write(d)
    exe = os.path.join(d, "caller")
    libdir = os.path.join(ROOT, "irotavg_amd")
    cmd = ["g++", "-std=c++11", "-Wall", "-DIROTAVG_SHIM_NO_EIGEN", "-I", os.path.join(ROOT, "include"), src,
           "-o", exe, "-L", libdir, "-lirotavg_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_the_same_callers_compile_and_link_against_the_eigen_free_branch():
    with tempfile.TemporaryDirectory() as d:
        _build_plain(d)


@pytest.mark.gpu
def test_the_same_callers_run_on_the_gpu():
    with tempfile.TemporaryDirectory() as d:
        exe = _build_plain(d)
        r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        last = r.stdout.strip().splitlines()[-1].split()
        assert last[0] == "3"                      # rows of A
        assert 5.9 < float(last[1]) <= 6.0 + 1e-12  # two free views, small rotations: trace just below 3 each

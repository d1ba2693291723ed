"""The IROTAVG_SHIM_EIGEN branch of include/irotavg/l1_irls.hpp -- the branch a real iRotAvg build
takes -- cannot be compiled against Eigen here (not installed). This is a SYNTAX check only: the
shim plus a caller written like ral/test.cpp:285-302 parse against a mock of the handful of Eigen
declarations they use (tests/mock_eigen). It says nothing about parity."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CALLER = r"""
#include "irotavg/l1_irls.hpp"
#ifndef IROTAVG_SHIM_EIGEN
#error "the Eigen branch of the shim was not selected"
#endif
using namespace irotavg;
int main() {
    const int n = 3, f = 1;
    I_t I;
    I.push_back(std::make_pair(0, 1));
    I.push_back(std::make_pair(1, 2));
    Mat QQ = Mat::Zero((long)I.size(), 4), Q = Mat::Zero(n, 4);
    for (long k = 0; k < QQ.rows(); k++) QQ(k, 3) = 1.0;
    for (long i = 0; i < n; i++) Q(i, 3) = 1.0;
    init_mst(Q, QQ, I, f);                       // ral/test.cpp:286
    SpMat A = make_A(n, f, I);                   // :288
    int iters = 0;
    double runtime = 0;
    l1ra(QQ, I, A, Q, f, 100, 1e-3, iters, runtime);            // :295
    Vec weights((long)I.size());                                  // :299
    irls(QQ, I, A, Geman_McClure, 0.0873, Q, f, 100, 1e-3, weights, iters, runtime);  // :300
    quat_normalised(Q, f);                                        // :302
    return (int)A.rows();
}
"""


def test_eigen_branch_of_the_shim_parses_against_the_mock():
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "caller.cpp")
        with open(src, "w") as fh:
            fh.write(CALLER)
        cmd = ["g++", "-std=c++11", "-fsyntax-only", "-Wall", "-I", os.path.join(ROOT, "tests", "mock_eigen"),
               "-I", os.path.join(ROOT, "include"), src]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr

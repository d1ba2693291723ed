/*
 * irotavg/l1_irls.hpp -- header-only C++ shim with the names and signatures of the reference's
 * RAL API (ral/l1_irls.hpp:81-112, namespace irotavg) on top of the C ABI of libirotavg_hip.so
 * (include/irotavg_hip.h). A caller of the reference (ral/test.cpp:285-302,
 * src/ViewGraph.cpp:1400-1417) compiles against this header instead of ral/l1_irls.hpp and links
 * -lirotavg_hip instead of ral/l1_irls.cpp + SuiteSparse.
 *
 * Types: when Eigen is available (the reference's own dependency) the shim uses the reference's
 * typedefs (ral/l1_irls.hpp:43-51: Long, SpMat, Mat, Vec, Vec3, Vec4, Quat, T, I_t) so call sites
 * compile unchanged; otherwise it provides minimal stand-ins with the members those call sites use
 * (`Q(i, c)`, `Q.row(i) << ...`, `.rows()`, `.data()`, `Quat(w,x,y,z).normalized().toRotationMatrix()`).
 * Errors the reference answers with exit(-1) do the same here (message on stderr), unless
 * IROTAVG_SHIM_THROW is defined, in which case a std::runtime_error is thrown.
 */
#ifndef IROTAVG_L1_IRLS_SHIM_HPP
#define IROTAVG_L1_IRLS_SHIM_HPP

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <ostream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../irotavg_hip.h"

#if defined(__has_include)
#if __has_include(<Eigen/Dense>) && __has_include(<Eigen/Sparse>) && !defined(IROTAVG_SHIM_NO_EIGEN)
#define IROTAVG_SHIM_EIGEN 1
#include <Eigen/Dense>
#include <Eigen/Sparse>
#endif
#endif

namespace irotavg {

#define EPS 2.2204e-16 /* ral/l1_irls.hpp:40 */
const double DBL_MAX_ = std::numeric_limits<double>::max();

typedef std::vector<std::pair<int, int> > I_t; /* ral/l1_irls.hpp:51 */

/* ral/l1_irls.hpp:56-79 */
enum Cost { L2, L1, L15, L05, Geman_McClure, Huber, Pseudo_Huber, Andrews, Bisquare, Cauchy, Fair,
            Logistic, Talwar, Welsch };

inline std::ostream &operator<<(std::ostream &os, const Cost cost) {
    static const char *names[] = {"L2", "L1", "L1.5", "L0.5", "Geman-McClure", "Huber",
                                  "Pseudo-Huber", "Andrews", "Bisquare", "Cauchy", "Fair",
                                  "Logistic", "Talwar", "Welsch"};
    if (cost >= L2 && cost <= Welsch) os << names[cost];
    return os;
}

#ifdef IROTAVG_SHIM_EIGEN
typedef long Long;
typedef Eigen::SparseMatrix<double, Eigen::ColMajor, Long> SpMat;
typedef Eigen::MatrixXd Mat;
typedef Eigen::VectorXd Vec;
typedef Eigen::Vector3d Vec3;    /* ral/l1_irls.hpp:46 */
typedef Eigen::Vector4d Vec4;    /* :47 */
typedef Eigen::Quaterniond Quat; /* :48 -- used by the caller's write-back, src/ViewGraph.cpp:1426 */
typedef Eigen::Triplet<double> T; /* :49 */
inline long shim_ld(const Mat &M) { return (long)M.outerStride(); }
#else
typedef long Long;
/* column-major dense matrix, the subset of Eigen::MatrixXd the RAL callers use */
class Mat {
public:
    Mat() : r_(0), c_(0) {}
    Mat(long r, long c) : r_(r), c_(c), d_((size_t)(r * c), 0.0) {}
    static Mat Zero(long r, long c) { return Mat(r, c); }
    long rows() const { return r_; }
    long cols() const { return c_; }
    double *data() { return d_.data(); }
    const double *data() const { return d_.data(); }
    double &operator()(long i, long j) { return d_[(size_t)(j * r_ + i)]; }
    double operator()(long i, long j) const { return d_[(size_t)(j * r_ + i)]; }
    /* `Q.row(k) << x, y, z, w;` (src/ViewGraph.cpp:1378,1384,1392) */
    class RowInit {
    public:
        RowInit(Mat &m, long i) : m_(m), i_(i), j_(0) {}
        RowInit &operator<<(double v) { m_(i_, j_++) = v; return *this; }
        RowInit &operator,(double v) { m_(i_, j_++) = v; return *this; }
    private:
        Mat &m_;
        long i_, j_;
    };
    RowInit row(long i) { return RowInit(*this, i); }
    void transposeInPlace() { /* src/ViewGraph.cpp:1430 */
        Mat t(c_, r_);
        for (long i = 0; i < r_; i++)
            for (long j = 0; j < c_; j++) t(j, i) = (*this)(i, j);
        *this = t;
    }
private:
    long r_, c_;
    std::vector<double> d_;
};
class Vec {
public:
    Vec() {}
    explicit Vec(long n) : d_((size_t)n, 0.0) {}
    long size() const { return (long)d_.size(); }
    double *data() { return d_.data(); }
    const double *data() const { return d_.data(); }
    double &operator()(long i) { return d_[(size_t)i]; }
    double operator()(long i) const { return d_[(size_t)i]; }
    void setOnes() { for (auto &v : d_) v = 1.0; }
private:
    std::vector<double> d_;
};
/* ral/l1_irls.hpp:46-49 without Eigen: fixed-size vectors, a (row, col, value) triplet and the part of
 * Eigen::Quaterniond the reference's callers use (src/ViewGraph.cpp:1426-1429: construction from
 * (w, x, y, z), normalized(), toRotationMatrix() as a column-major 3x3 Mat). */
struct Vec3 {
    double v[3];
    Vec3() : v{0, 0, 0} {}
    Vec3(double a, double b, double c) : v{a, b, c} {}
    double &operator()(long i) { return v[i]; }
    double operator()(long i) const { return v[i]; }
    double *data() { return v; }
    const double *data() const { return v; }
};
struct Vec4 {
    double v[4];
    Vec4() : v{0, 0, 0, 0} {}
    Vec4(double a, double b, double c, double d) : v{a, b, c, d} {}
    double &operator()(long i) { return v[i]; }
    double operator()(long i) const { return v[i]; }
    double *data() { return v; }
    const double *data() const { return v; }
};
class T {
public:
    T() : r_(0), c_(0), v_(0.0) {}
    T(long r, long c, double v) : r_(r), c_(c), v_(v) {}
    long row() const { return r_; }
    long col() const { return c_; }
    double value() const { return v_; }
private:
    long r_, c_;
    double v_;
};
class Quat {
public:
    Quat() : w_(1), x_(0), y_(0), z_(0) {}
    Quat(double w, double x, double y, double z) : w_(w), x_(x), y_(y), z_(z) {}
    double w() const { return w_; }
    double x() const { return x_; }
    double y() const { return y_; }
    double z() const { return z_; }
    Quat normalized() const {
        const double n = std::sqrt(w_ * w_ + x_ * x_ + y_ * y_ + z_ * z_);
        return n > 0.0 ? Quat(w_ / n, x_ / n, y_ / n, z_ / n) : *this;
    }
    void normalize() { *this = normalized(); }
    Mat toRotationMatrix() const {
        const double q[4] = {x_, y_, z_, w_};
        double R[9];
        irotavg_quat2rmat(q, R); /* row-major */
        Mat M(3, 3);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) M(i, j) = R[3 * i + j];
        return M;
    }
private:
    double w_, x_, y_, z_;
};
/* CSC incidence matrix as make_A returns it (ral/l1_irls.cpp:755-780) */
struct SpMat {
    long nrows = 0, ncols = 0;
    std::vector<Long> outer; /* ncols + 1 */
    std::vector<Long> inner;
    std::vector<double> values;
    long rows() const { return nrows; }
    long cols() const { return ncols; }
    long nonZeros() const { return (long)values.size(); }
};
inline long shim_ld(const Mat &M) { return M.rows(); }
#endif

namespace detail {
inline void fail(int rc, const char *where) {
    std::string msg = std::string(where) + ": " + irotavg_error_string(rc);
#ifdef IROTAVG_SHIM_THROW
    throw std::runtime_error(msg);
#else
    std::fprintf(stderr, "%s\n", msg.c_str()); /* the reference: std::cerr + std::exit(-1) */
    std::exit(-1);
#endif
}
/* IROTAVG_ERR_NOT_CONVERGED: the inner PCG stopped at its iteration cap. The reference's direct
 * solvers always return a result, so the shim reports it the way the reference reports
 * " Max Iteration" (ral/l1_irls.cpp:746-749): a message, and the rotations / weights the C call has
 * already written back are kept. Everything else is the reference's std::cerr + exit(-1). */
inline bool soft(int rc, const char *where) {
    if (rc != IROTAVG_ERR_NOT_CONVERGED) return false;
    std::fprintf(stderr, "%s: warning: %s (result kept)\n", where, irotavg_error_string(rc));
    return true;
}
/* The edge list as the C ABI wants it: 2 m int32, (i, j) pairs. I_t = std::vector<std::pair<int, int>> already IS that in
 * memory (a pair of two ints is two ints: checked below), so the caller's own array is handed over -- no copy, and the
 * SAME pointer in l1ra and irls, which is what lets the library recognise the graph of the call before
 * (irotavg_oneshot_cache in irotavg_hip.h: the reference's callers pass the same I, QQ to both, src/ViewGraph.cpp:1400-1417). */
struct EdgeView {
    const int32_t *p;
    std::vector<int32_t> copy;  /* only when the layout assumption does not hold on this platform */
    const int32_t *data() const { return copy.empty() ? p : &copy[0]; }
};
inline EdgeView flat(const I_t &I) {
    EdgeView v;
    typedef std::pair<int, int> P;
    if (sizeof(P) == 2 * sizeof(int32_t) && sizeof(int) == sizeof(int32_t) && (I.empty() || (const void *)&I[0].second ==
                                                                                (const void *)((const int32_t *)&I[0] + 1))) {
        v.p = I.empty() ? (const int32_t *)0 : (const int32_t *)(const void *)&I[0];
        return v;
    }
    v.copy.resize(2 * I.size());
    for (size_t k = 0; k < I.size(); k++) {
        v.copy[2 * k] = I[k].first;
        v.copy[2 * k + 1] = I[k].second;
    }
    v.p = v.copy.empty() ? (const int32_t *)0 : &v.copy[0];
    return v;
}
}  // namespace detail

/* ral/l1_irls.hpp:89 */
inline void init_mst(Mat &Q, const Mat &QQ, const I_t &I, const int f) {
    detail::EdgeView e = detail::flat(I);
    int rc = irotavg_init_mst(Q.rows(), (int64_t)I.size(), Q.data(), shim_ld(Q), QQ.data(),
                              shim_ld(QQ), e.data(), f);
    if (rc != IROTAVG_OK) detail::fail(rc, "init_mst");
}

/* ral/l1_irls.hpp:91 */
inline SpMat make_A(const int n, const int f, const I_t &I) {
    detail::EdgeView e = detail::flat(I);
    const long m = (long)I.size();
    std::vector<int64_t> colptr((size_t)(n - f + 1)), rowidx((size_t)(2 * m + 1));
    std::vector<double> vals((size_t)(2 * m + 1));
    int64_t nnz = irotavg_make_A(n, f, m, e.data(), colptr.data(), rowidx.data(), vals.data());
    if (nnz < 0) detail::fail((int)nnz, "make_A");
#ifdef IROTAVG_SHIM_EIGEN
    SpMat A(m, n - f);
    std::vector<Eigen::Triplet<double> > t;
    t.reserve((size_t)nnz);
    for (long c = 0; c < n - f; c++)
        for (int64_t p = colptr[c]; p < colptr[c + 1]; p++) t.emplace_back((long)rowidx[p], c, vals[p]);
    A.setFromTriplets(t.begin(), t.end());
    A.makeCompressed();
    return A;
#else
    SpMat A;
    A.nrows = m;
    A.ncols = n - f;
    A.outer.assign(colptr.begin(), colptr.end());
    A.inner.assign(rowidx.begin(), rowidx.begin() + nnz);
    A.values.assign(vals.begin(), vals.begin() + nnz);
    return A;
#endif
}

/* ral/l1_irls.hpp:100-102. `A` is accepted for signature parity; the core derives it from (n, f, I). */
inline void l1ra(const Mat &QQ, const I_t &I, const SpMat & /*A*/, Mat &Q, const int f,
                 const int max_iters, double change_th, int &iter, double &runtime) {
    detail::EdgeView e = detail::flat(I);
    int rc = irotavg_l1ra((int64_t)I.size(), Q.rows(), f, e.data(), QQ.data(), shim_ld(QQ), Q.data(),
                          shim_ld(Q), max_iters, change_th, &iter, &runtime);
    if (rc != IROTAVG_OK && !detail::soft(rc, "l1ra")) detail::fail(rc, "l1ra");
}

/* ral/l1_irls.hpp:104-107. weights must be pre-sized to m (ral/test.cpp:299). */
inline void irls(const Mat &QQ, const I_t &I, const SpMat & /*A*/, Cost cost, double sigma, Mat &Q,
                 const int f, const int max_iters, double change_th, Vec &weights, int &iteration,
                 double &runtime) {
    detail::EdgeView e = detail::flat(I);
    if ((long)weights.size() != (long)I.size()) detail::fail(IROTAVG_ERR_BAD_ARG, "irls (weights size)");
    int rc = irotavg_irls((int64_t)I.size(), Q.rows(), f, e.data(), QQ.data(), shim_ld(QQ), (int)cost,
                          sigma, Q.data(), shim_ld(Q), max_iters, change_th, weights.data(),
                          &iteration, &runtime);
    if (rc != IROTAVG_OK && !detail::soft(rc, "irls")) detail::fail(rc, "irls");
}

/* ral/l1_irls.hpp:112 */
inline void quat_normalised(Mat &Q, const int f) {
    int rc = irotavg_quat_normalised(Q.rows(), Q.data(), shim_ld(Q), f);
    if (rc != IROTAVG_OK) detail::fail(rc, "quat_normalised");
}

}  // namespace irotavg

#endif /* IROTAVG_L1_IRLS_SHIM_HPP */

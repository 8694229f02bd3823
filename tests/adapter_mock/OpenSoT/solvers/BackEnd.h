// tests/adapter_mock/OpenSoT/solvers/BackEnd.h -- TEST DOUBLE, not OpenSoT and not Eigen.
// The smallest interface adapters/opensot_backend/MI355XBackEnd.cpp needs in order to be COMPILED (and, on the GPU box, RUN) in an image
// that has neither OpenSoT nor Eigen nor Boost: dense column-major / row-major matrix holders with the handful of members the adapter
// uses, and an abstract BackEnd with the members and virtuals of include/OpenSoT/solvers/BackEnd.h:23-171 (names, argument order and
// constness restated; no behaviour).  It exists so that a change of include/osot_mi355x.h that the adapter does not follow fails a test.
#pragma once
#include <boost/any.hpp>
#include <cstddef>
#include <vector>

namespace Eigen {
enum { Dynamic = -1, ColMajor = 0, RowMajor = 1 };
template <typename T, int R, int C, int Opt = ColMajor>
class Matrix {
    std::vector<T> v_;
    long r_ = 0, c_ = 0;
public:
    Matrix() {}
    Matrix(long r, long c) : v_((size_t)(r * c)), r_(r), c_(c) {}
    explicit Matrix(long r) : v_((size_t)r), r_(r), c_(1) {}
    template <int O2> Matrix(const Matrix<T, R, C, O2>& o) { *this = o; }
    template <int O2> Matrix& operator=(const Matrix<T, R, C, O2>& o) {      // storage-order converting copy
        r_ = o.rows(); c_ = o.cols(); v_.resize((size_t)(r_ * c_));
        for (long i = 0; i < r_; ++i) for (long j = 0; j < c_; ++j) (*this)(i, j) = o(i, j);
        return *this;
    }
    long rows() const { return r_; }
    long cols() const { return c_; }
    long size() const { return r_ * c_; }
    void resize(long r) { r_ = r; c_ = 1; v_.resize((size_t)r); }
    void resize(long r, long c) { r_ = r; c_ = c; v_.resize((size_t)(r * c)); }
    T* data() { return v_.data(); }
    const T* data() const { return v_.data(); }
    T& operator()(long i, long j) { return v_[(size_t)(Opt == RowMajor ? i * c_ + j : j * r_ + i)]; }
    const T& operator()(long i, long j) const { return v_[(size_t)(Opt == RowMajor ? i * c_ + j : j * r_ + i)]; }
    T& operator()(long i) { return v_[(size_t)i]; }
    const T& operator()(long i) const { return v_[(size_t)i]; }
};
typedef Matrix<double, Dynamic, Dynamic> MatrixXd;
typedef Matrix<double, Dynamic, 1> VectorXd;
}  // namespace Eigen

namespace OpenSoT {
enum HessianType { HST_ZERO, HST_IDENTITY, HST_POSDEF, HST_POSDEF_NULLSPACE, HST_SEMIDEF, HST_UNKNOWN };
namespace solvers {
class BackEnd {
public:
    BackEnd(const int number_of_variables, const int number_of_constraints)
        : _H(number_of_variables, number_of_variables), _g(number_of_variables), _A(number_of_constraints, number_of_variables),
          _lA(number_of_constraints), _uA(number_of_constraints), _solution(number_of_variables) {}
    virtual ~BackEnd() {}
    const Eigen::VectorXd& getSolution() { return _solution; }
    virtual bool initProblem(const Eigen::MatrixXd& H, const Eigen::VectorXd& g, const Eigen::MatrixXd& A, const Eigen::VectorXd& lA,
                             const Eigen::VectorXd& uA, const Eigen::VectorXd& l, const Eigen::VectorXd& u) = 0;
    virtual bool solve() = 0;
    virtual boost::any getOptions() = 0;
    virtual void setOptions(const boost::any& options) = 0;
    virtual double getObjective() = 0;
    virtual bool setEpsRegularisation(const double eps) { (void)eps; return false; }
    virtual double getEpsRegularisation() { return 0.0; }
    // (the base class's default updateTask / updateConstraints / updateBounds copy into these members: BackEnd.cpp:19-93)
    virtual bool updateTask(const Eigen::MatrixXd& H, const Eigen::VectorXd& g) { _H = H; _g = g; return true; }
    virtual bool updateConstraints(const Eigen::MatrixXd& A, const Eigen::VectorXd& lA, const Eigen::VectorXd& uA) { _A = A; _lA = lA; _uA = uA; return true; }
    virtual bool updateBounds(const Eigen::VectorXd& l, const Eigen::VectorXd& u) { _l = l; _u = u; return true; }
protected:
    Eigen::MatrixXd _H;
    Eigen::VectorXd _g;
    Eigen::MatrixXd _A;
    Eigen::VectorXd _lA, _uA, _l, _u, _solution;
};
}}  // namespace OpenSoT::solvers

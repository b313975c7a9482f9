// Minimal column-major containers for the C++ facade.  The reference exposes
// Eigen::VectorXd / Eigen::MatrixXd in its ProblemData / Variable structs
// (fbstab/fbstab_dense.h:55-92, fbstab/fbstab_mpc.h:67-136).  Eigen is not a
// dependency of this library: when <Eigen/Dense> is available the facade uses
// the Eigen types themselves (same names, same memory layout), otherwise the
// stand-ins below, which offer the subset of the Eigen interface the reference's
// own tests use (size/rows/cols/data/operator()/setZero/fill/Zero/comma-free
// row-major assignment via assign_rowmajor).
#pragma once

#include <cstddef>
#include <initializer_list>
#include <stdexcept>
#include <vector>

#if defined(FBSTAB_USE_EIGEN) || (defined(__has_include) && __has_include(<Eigen/Dense>))
#include <Eigen/Dense>
namespace fbstab {
typedef Eigen::VectorXd VectorXd;
typedef Eigen::MatrixXd MatrixXd;
typedef Eigen::Vector4d Vector4d;
#define FBSTAB_FACADE_HAS_EIGEN 1
}  // namespace fbstab
#else
namespace fbstab {

class VectorXd {
 public:
  VectorXd() {}
  explicit VectorXd(int n) : d_(n > 0 ? n : 0, 0.0) {}
  static VectorXd Zero(int n) { return VectorXd(n); }
  void resize(int n) { d_.assign(n > 0 ? n : 0, 0.0); }
  int size() const { return static_cast<int>(d_.size()); }
  double* data() { return d_.data(); }
  const double* data() const { return d_.data(); }
  double& operator()(int i) { return d_[i]; }
  double operator()(int i) const { return d_[i]; }
  double& operator[](int i) { return d_[i]; }
  double operator[](int i) const { return d_[i]; }
  void setZero() { fill(0.0); }
  void fill(double a) { for (size_t i = 0; i < d_.size(); i++) d_[i] = a; }
  void setConstant(double a) { fill(a); }
  // v = {a, b, c}: the role Eigen's "v << a, b, c" plays in the reference tests
  VectorXd& operator=(std::initializer_list<double> v) {
    if (static_cast<size_t>(size()) != v.size()) throw std::runtime_error("VectorXd: size mismatch");
    size_t i = 0;
    for (double a : v) d_[i++] = a;
    return *this;
  }

 private:
  std::vector<double> d_;
};

// s = (N, nx, nu, nc) of FBstabMpc(const Eigen::Vector4d&) (fbstab_mpc.h:130, :168)
class Vector4d {
 public:
  Vector4d() { v_[0] = v_[1] = v_[2] = v_[3] = 0.0; }
  Vector4d(double a, double b, double c, double d) { v_[0] = a; v_[1] = b; v_[2] = c; v_[3] = d; }
  int size() const { return 4; }
  double& operator()(int i) { return v_[i]; }
  double operator()(int i) const { return v_[i]; }
  const double* data() const { return v_; }

 private:
  double v_[4];
};

class MatrixXd {
 public:
  MatrixXd() : r_(0), c_(0) {}
  MatrixXd(int r, int c) : r_(r), c_(c), d_(static_cast<size_t>(r > 0 ? r : 0) * (c > 0 ? c : 0), 0.0) {}
  static MatrixXd Zero(int r, int c) { return MatrixXd(r, c); }
  void resize(int r, int c) { r_ = r; c_ = c; d_.assign(static_cast<size_t>(r) * c, 0.0); }
  int rows() const { return r_; }
  int cols() const { return c_; }
  int size() const { return r_ * c_; }
  double* data() { return d_.data(); }
  const double* data() const { return d_.data(); }
  double& operator()(int i, int j) { return d_[i + static_cast<size_t>(j) * r_]; }
  double operator()(int i, int j) const { return d_[i + static_cast<size_t>(j) * r_]; }
  void setZero() { for (size_t i = 0; i < d_.size(); i++) d_[i] = 0.0; }
  // Row-major literal, like Eigen's "M << a, b, c, d".
  MatrixXd& operator=(std::initializer_list<double> v) {
    if (static_cast<size_t>(size()) != v.size()) throw std::runtime_error("MatrixXd: size mismatch");
    size_t k = 0;
    for (double a : v) {
      (*this)(static_cast<int>(k / c_), static_cast<int>(k % c_)) = a;
      k++;
    }
    return *this;
  }

 private:
  int r_, c_;
  std::vector<double> d_;
};

}  // namespace fbstab
#endif

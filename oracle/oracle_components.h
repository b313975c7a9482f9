// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_linalg.h header).
//
// CPU restatement of the reference's L1 components, Eigen-free, following the
// reference files operation by operation:
//   Data interface            <- fbstab/components/abstract_components.h:24-62
//   DenseData                 <- fbstab/components/dense_data.h:44-126, dense_data.cc:12-41
//   MpcData                   <- fbstab/components/mpc_data.h:62-98, mpc_data.cc:17-363
//   FullVariable              <- fbstab/components/full_variable.cc:14-95
//   FullResidual              <- fbstab/components/full_residual.cc:13-125
//   FullFeasibility           <- fbstab/components/full_feasibility.cc:12-96
//   DenseCholeskySolver       <- fbstab/components/dense_cholesky_solver.cc:12-155
//   RiccatiLinearSolver       <- fbstab/components/riccati_linear_solver.cc:17-373
// All matrices are column-major, sequences are stage-major
// (tools/matrix_sequence.h:81-83).  Error behaviour (std::runtime_error on
// size mismatch / bad sigma, bool false on factorisation failure) mirrors the
// reference.
#pragma once

#include <cmath>
#include <stdexcept>
#include <vector>

#include "oracle_linalg.h"

namespace fbo {

// ---------------------------------------------------------------------------
// abstract_components.h:24-62
class Data {
 public:
  virtual ~Data() {}
  virtual void gemvH(const Vec& x, double a, double b, Vec* y) const = 0;
  virtual void gemvA(const Vec& x, double a, double b, Vec* y) const = 0;
  virtual void gemvG(const Vec& x, double a, double b, Vec* y) const = 0;
  virtual void gemvAT(const Vec& x, double a, double b, Vec* y) const = 0;
  virtual void gemvGT(const Vec& x, double a, double b, Vec* y) const = 0;
  virtual void axpyf(double a, Vec* y) const = 0;
  virtual void axpyh(double a, Vec* y) const = 0;
  virtual void axpyb(double a, Vec* y) const = 0;
  virtual double ForcingNorm() const = 0;
  virtual int nz() const = 0;
  virtual int nl() const = 0;
  virtual int nv() const = 0;
};

inline void scale_or_zero(double b, Vec* y) {
  if (b == 0.0) {
    std::fill(y->begin(), y->end(), 0.0);
  } else if (b != 1.0) {
    for (size_t i = 0; i < y->size(); i++) (*y)[i] *= b;
  }
}

// ---------------------------------------------------------------------------
// dense_data.h:44-126, dense_data.cc:12-41.  Non-owning views.
class DenseData : public Data {
 public:
  DenseData(const double* H, const double* f, const double* G, const double* h,
            const double* A, const double* b, int nz, int nl, int nv)
      : H_(H), G_(G), A_(A), f_(f), h_(h), b_(b), nz_(nz), nl_(nl), nv_(nv) {
    if (nz <= 0 || nl < 0 || nv <= 0)
      throw std::runtime_error("In DenseData::DenseData: bad sizes.");
    // dense_data.h:72-73
    forcing_norm_ =
        std::sqrt(dot(b, b, nv) + dot(f, f, nz) + (nl ? dot(h, h, nl) : 0.0));
  }
  // dense_data.cc:12-35: *y = a*M*x + b*(*y)  (b*(*y) is evaluated even for
  // b == 0).
  void gemvH(const Vec& x, double a, double b, Vec* y) const override {
    gen(H_, nz_, nz_, false, x, a, b, y);
  }
  void gemvG(const Vec& x, double a, double b, Vec* y) const override {
    gen(G_, nl_, nz_, false, x, a, b, y);
  }
  void gemvGT(const Vec& x, double a, double b, Vec* y) const override {
    gen(G_, nl_, nz_, true, x, a, b, y);
  }
  void gemvA(const Vec& x, double a, double b, Vec* y) const override {
    gen(A_, nv_, nz_, false, x, a, b, y);
  }
  void gemvAT(const Vec& x, double a, double b, Vec* y) const override {
    gen(A_, nv_, nz_, true, x, a, b, y);
  }
  void axpyf(double a, Vec* y) const override {
    for (int i = 0; i < nz_; i++) (*y)[i] += a * f_[i];
  }
  void axpyh(double a, Vec* y) const override {
    for (int i = 0; i < nl_; i++) (*y)[i] += a * h_[i];
  }
  void axpyb(double a, Vec* y) const override {
    for (int i = 0; i < nv_; i++) (*y)[i] += a * b_[i];
  }
  double ForcingNorm() const override { return forcing_norm_; }
  int nz() const override { return nz_; }
  int nl() const override { return nl_; }
  int nv() const override { return nv_; }

  const double *H_, *G_, *A_, *f_, *h_, *b_;

 private:
  void gen(const double* M, int m, int n, bool trans, const Vec& x, double a,
           double b, Vec* y) const {
    const int out = trans ? n : m;
    const int in = trans ? m : n;
    if ((int)x.size() != in || (int)y->size() != out)
      throw std::runtime_error("Size mismatch in DenseData::gemv.");
    Vec t(out, 0.0);
    if (trans)
      gemv_t(M, m, n, a, x.data(), t.data());
    else
      gemv_n(M, m, n, a, x.data(), t.data());
    for (int i = 0; i < out; i++) (*y)[i] = t[i] + b * (*y)[i];
  }
  int nz_, nl_, nv_;
  double forcing_norm_ = 0.0;
};

// ---------------------------------------------------------------------------
// mpc_data.h:62-98, mpc_data.cc:17-289.  Non-owning views of the 11 sequences.
class MpcData : public Data {
 public:
  MpcData(const double* Q, const double* R, const double* S, const double* q,
          const double* r, const double* A, const double* B, const double* c,
          const double* E, const double* L, const double* d, const double* x0,
          int N, int nx, int nu, int nc)
      : Q_(Q), R_(R), S_(S), q_(q), r_(r), A_(A), B_(B), c_(c), E_(E), L_(L),
        d_(d), x0_(x0), N_(N), nx_(nx), nu_(nu), nc_(nc) {
    if (N <= 0 || nx <= 0 || nu <= 0 || nc <= 0)
      throw std::runtime_error("Horizon length must be at least 1.");
    nz_ = (N + 1) * (nx + nu);
    nl_ = (N + 1) * nx;
    nv_ = (N + 1) * nc;
    // mpc_data.h:88-97
    double s = 0.0;
    for (int i = 0; i < N + 1; i++) {
      s += dot(qi(i), qi(i), nx);
      s += dot(ri(i), ri(i), nu);
      s += dot(di(i), di(i), nc);
      s += (i == 0 ? dot(x0, x0, nx) : dot(ci(i - 1), ci(i - 1), nx));
    }
    forcing_norm_ = std::sqrt(s);
  }
  const double* Qi(int i) const { return Q_ + (size_t)i * nx_ * nx_; }
  const double* Ri(int i) const { return R_ + (size_t)i * nu_ * nu_; }
  const double* Si(int i) const { return S_ + (size_t)i * nu_ * nx_; }  // nu x nx
  const double* qi(int i) const { return q_ + (size_t)i * nx_; }
  const double* ri(int i) const { return r_ + (size_t)i * nu_; }
  const double* Ai(int i) const { return A_ + (size_t)i * nx_ * nx_; }
  const double* Bi(int i) const { return B_ + (size_t)i * nx_ * nu_; }  // nx x nu
  const double* ci(int i) const { return c_ + (size_t)i * nx_; }
  const double* Ei(int i) const { return E_ + (size_t)i * nc_ * nx_; }  // nc x nx
  const double* Li(int i) const { return L_ + (size_t)i * nc_ * nu_; }  // nc x nu
  const double* di(int i) const { return d_ + (size_t)i * nc_; }

  // mpc_data.cc:17-65
  void gemvH(const Vec& x, double a, double b, Vec* y) const override {
    check(x, nz_, y, nz_, "gemvH");
    scale_or_zero(b, y);
    const int ns = nx_ + nu_;
    for (int i = 0; i < N_ + 1; i++) {
      double* yx = y->data() + (size_t)i * ns;
      double* yu = yx + nx_;
      const double* vx = x.data() + (size_t)i * ns;
      const double* vu = vx + nx_;
      gemv_n(Qi(i), nx_, nx_, a, vx, yx);
      gemv_t(Si(i), nu_, nx_, a, vu, yx);
      gemv_n(Si(i), nu_, nx_, a, vx, yu);
      gemv_n(Ri(i), nu_, nu_, a, vu, yu);
    }
  }
  // mpc_data.cc:67-105
  void gemvA(const Vec& x, double a, double b, Vec* y) const override {
    check(x, nz_, y, nv_, "gemvA");
    scale_or_zero(b, y);
    const int ns = nx_ + nu_;
    for (int i = 0; i < N_ + 1; i++) {
      double* yi = y->data() + (size_t)i * nc_;
      const double* xi = x.data() + (size_t)i * ns;
      gemv_n(Ei(i), nc_, nx_, a, xi, yi);
      gemv_n(Li(i), nc_, nu_, a, xi + nx_, yi);
    }
  }
  // mpc_data.cc:107-153
  void gemvG(const Vec& x, double a, double b, Vec* y) const override {
    check(x, nz_, y, nl_, "gemvG");
    scale_or_zero(b, y);
    const int ns = nx_ + nu_;
    for (int j = 0; j < nx_; j++) (*y)[j] += -a * x[j];
    for (int i = 1; i < N_ + 1; i++) {
      double* yi = y->data() + (size_t)i * nx_;
      const double* xm1 = x.data() + (size_t)(i - 1) * ns;
      const double* xi = x.data() + (size_t)i * ns;
      gemv_n(Ai(i - 1), nx_, nx_, a, xm1, yi);
      gemv_n(Bi(i - 1), nx_, nu_, a, xm1 + nx_, yi);
      for (int j = 0; j < nx_; j++) yi[j] -= a * xi[j];
    }
  }
  // mpc_data.cc:155-198.  The general-a branch of the reference omits the B'
  // term (mpc_data.cc:192-194); mirrored here.
  void gemvGT(const Vec& x, double a, double b, Vec* y) const override {
    check(x, nl_, y, nz_, "gemvGT");
    scale_or_zero(b, y);
    const int ns = nx_ + nu_;
    for (int i = 0; i < N_; i++) {
      const double* vi = x.data() + (size_t)i * nx_;
      const double* vp1 = vi + nx_;
      double* xi = y->data() + (size_t)i * ns;
      double* ui = xi + nx_;
      for (int j = 0; j < nx_; j++) xi[j] += -a * vi[j];
      gemv_t(Ai(i), nx_, nx_, a, vp1, xi);
      if (a == 1.0 || a == -1.0) gemv_t(Bi(i), nx_, nu_, a, vp1, ui);
    }
    double* xN = y->data() + (size_t)N_ * ns;
    const double* vN = x.data() + (size_t)N_ * nx_;
    for (int j = 0; j < nx_; j++) xN[j] += -a * vN[j];
  }
  // mpc_data.cc:200-238
  void gemvAT(const Vec& x, double a, double b, Vec* y) const override {
    check(x, nv_, y, nz_, "gemvAT");
    scale_or_zero(b, y);
    const int ns = nx_ + nu_;
    for (int i = 0; i < N_ + 1; i++) {
      double* xi = y->data() + (size_t)i * ns;
      const double* vi = x.data() + (size_t)i * nc_;
      gemv_t(Ei(i), nc_, nx_, a, vi, xi);
      gemv_t(Li(i), nc_, nu_, a, vi, xi + nx_);
    }
  }
  // mpc_data.cc:240-258: f = (q,r) stage-interleaved.
  void axpyf(double a, Vec* y) const override {
    if ((int)y->size() != nz_) throw std::runtime_error("Size mismatch in MpcData::axpyf.");
    const int ns = nx_ + nu_;
    for (int i = 0; i < N_ + 1; i++) {
      double* xi = y->data() + (size_t)i * ns;
      for (int j = 0; j < nx_; j++) xi[j] += a * qi(i)[j];
      for (int j = 0; j < nu_; j++) xi[nx_ + j] += a * ri(i)[j];
    }
  }
  // mpc_data.cc:260-274: h = -(x0, c0, ..., c_{N-1}).
  void axpyh(double a, Vec* y) const override {
    if ((int)y->size() != nl_) throw std::runtime_error("Size mismatch in MpcData::axpyh.");
    for (int j = 0; j < nx_; j++) (*y)[j] += -a * x0_[j];
    for (int i = 1; i < N_ + 1; i++)
      for (int j = 0; j < nx_; j++) (*y)[(size_t)i * nx_ + j] += -a * ci(i - 1)[j];
  }
  // mpc_data.cc:276-289: b = -d.
  void axpyb(double a, Vec* y) const override {
    if ((int)y->size() != nv_) throw std::runtime_error("Size mismatch in MpcData::axpyb.");
    for (int i = 0; i < N_ + 1; i++)
      for (int j = 0; j < nc_; j++) (*y)[(size_t)i * nc_ + j] += -a * di(i)[j];
  }
  double ForcingNorm() const override { return forcing_norm_; }
  int nz() const override { return nz_; }
  int nl() const override { return nl_; }
  int nv() const override { return nv_; }
  int N() const { return N_; }
  int nx() const { return nx_; }
  int nu() const { return nu_; }
  int nc() const { return nc_; }

  const double *Q_, *R_, *S_, *q_, *r_, *A_, *B_, *c_, *E_, *L_, *d_, *x0_;

 private:
  void check(const Vec& x, int nxs, Vec* y, int nys, const char* who) const {
    if (y == nullptr || (int)x.size() != nxs || (int)y->size() != nys)
      throw std::runtime_error(std::string("Size mismatch in MpcData::") + who);
  }
  int N_, nx_, nu_, nc_, nz_, nl_, nv_;
  double forcing_norm_ = 0.0;
};

// ---------------------------------------------------------------------------
// full_variable.cc:14-95
class FullVariable {
 public:
  FullVariable(int nz, int nl, int nv) : nz_(nz), nl_(nl), nv_(nv) {
    if (nz <= 0 || nl < 0 || nv <= 0)
      throw std::runtime_error(
          "All size inputs to FullVariable::FullVariable must be >= 1.");
    z_.assign(nz, 0.0);
    l_.assign(nl, 0.0);
    v_.assign(nv, 0.0);
    y_.assign(nv, 0.0);
  }
  void LinkData(const Data* data) { data_ = data; }
  void Fill(double a) {
    std::fill(z_.begin(), z_.end(), a);
    std::fill(l_.begin(), l_.end(), a);
    std::fill(v_.begin(), v_.end(), a);
    InitializeConstraintMargin();
  }
  // y = b - A*z (full_variable.cc:47-53)
  void InitializeConstraintMargin() {
    NullDataCheck();
    std::fill(y_.begin(), y_.end(), 0.0);
    data_->axpyb(1.0, &y_);
    data_->gemvA(z_, -1.0, 1.0, &y_);
  }
  // full_variable.cc:55-65
  void axpy(double a, const FullVariable& x) {
    NullDataCheck();
    for (int i = 0; i < nz_; i++) z_[i] += a * x.z_[i];
    for (int i = 0; i < nl_; i++) l_[i] += a * x.l_[i];
    for (int i = 0; i < nv_; i++) v_[i] += a * x.v_[i];
    for (int i = 0; i < nv_; i++) y_[i] += a * x.y_[i];
    data_->axpyb(-a, &y_);
  }
  void Copy(const FullVariable& x) {
    z_ = x.z_;
    l_ = x.l_;
    v_ = x.v_;
    y_ = x.y_;
    data_ = x.data_;
  }
  void ProjectDuals() {
    for (int i = 0; i < nv_; i++) v_[i] = std::max(v_[i], 0.0);
  }
  double Norm() const {
    const double t1 = norm2(z_), t2 = norm2(l_), t3 = norm2(v_);
    return std::sqrt(t1 * t1 + t2 * t2 + t3 * t3);
  }
  bool SameSize(const FullVariable& x) const {
    return x.nz_ == nz_ && x.nl_ == nl_ && x.nv_ == nv_;
  }
  Vec& z() { return z_; }
  Vec& l() { return l_; }
  Vec& v() { return v_; }
  Vec& y() { return y_; }
  const Vec& z() const { return z_; }
  const Vec& l() const { return l_; }
  const Vec& v() const { return v_; }
  const Vec& y() const { return y_; }
  int nz_, nl_, nv_;

 private:
  void NullDataCheck() const {
    if (data_ == nullptr)
      throw std::runtime_error(
          "FullVariable tried to access problem data before it's linked.");
  }
  Vec z_, l_, v_, y_;
  const Data* data_ = nullptr;
};

// ---------------------------------------------------------------------------
// full_residual.cc:13-125
class FullResidual {
 public:
  FullResidual(int nz, int nl, int nv) : nz_(nz), nl_(nl), nv_(nv) {
    if (nz <= 0 || nl < 0 || nv <= 0)
      throw std::runtime_error(
          "All inputs to FullResidual::FullResidual must be >= 1.");
    z_.assign(nz, 0.0);
    l_.assign(nl, 0.0);
    v_.assign(nv, 0.0);
  }
  void LinkData(const Data* data) { data_ = data; }
  void SetAlpha(double alpha) { alpha_ = alpha; }
  // Fill/Negate do not refresh the cached norms (full_residual.cc:28-38).
  void Fill(double a) {
    std::fill(z_.begin(), z_.end(), a);
    std::fill(l_.begin(), l_.end(), a);
    std::fill(v_.begin(), v_.end(), a);
  }
  void Negate() {
    for (size_t i = 0; i < z_.size(); i++) z_[i] *= -1;
    for (size_t i = 0; i < l_.size(); i++) l_[i] *= -1;
    for (size_t i = 0; i < v_.size(); i++) v_[i] *= -1;
  }
  double Norm() const {
    return std::sqrt(znorm_ * znorm_ + lnorm_ * lnorm_ + vnorm_ * vnorm_);
  }
  double Merit() const {
    const double t = Norm();
    return 0.5 * t * t;
  }
  // full_residual.cc:49-74
  void InnerResidual(const FullVariable& x, const FullVariable& xbar,
                     double sigma) {
    NullDataCheck();
    std::fill(z_.begin(), z_.end(), 0.0);
    data_->axpyf(1.0, &z_);
    data_->gemvH(x.z(), 1.0, 1.0, &z_);
    data_->gemvGT(x.l(), 1.0, 1.0, &z_);
    data_->gemvAT(x.v(), 1.0, 1.0, &z_);
    for (int i = 0; i < nz_; i++) z_[i] += sigma * (x.z()[i] - xbar.z()[i]);

    std::fill(l_.begin(), l_.end(), 0.0);
    data_->axpyh(1.0, &l_);
    data_->gemvG(x.z(), -1.0, 1.0, &l_);
    for (int i = 0; i < nl_; i++) l_[i] += sigma * (x.l()[i] - xbar.l()[i]);

    for (int i = 0; i < nv_; i++) {
      const double ys = x.y()[i] + sigma * (x.v()[i] - xbar.v()[i]);
      v_[i] = pfb(ys, x.v()[i], alpha_);
    }
    UpdateNorms();
  }
  // full_residual.cc:76-97
  void NaturalResidual(const FullVariable& x) {
    NullDataCheck();
    std::fill(z_.begin(), z_.end(), 0.0);
    data_->axpyf(1.0, &z_);
    data_->gemvH(x.z(), 1.0, 1.0, &z_);
    data_->gemvGT(x.l(), 1.0, 1.0, &z_);
    data_->gemvAT(x.v(), 1.0, 1.0, &z_);

    std::fill(l_.begin(), l_.end(), 0.0);
    data_->axpyh(1.0, &l_);
    data_->gemvG(x.z(), -1.0, 1.0, &l_);

    for (int i = 0; i < nv_; i++) v_[i] = std::min(x.y()[i], x.v()[i]);
    UpdateNorms();
  }
  // full_residual.cc:99-109
  void PenalizedNaturalResidual(const FullVariable& x) {
    NaturalResidual(x);
    for (int i = 0; i < nv_; i++) {
      v_[i] = alpha_ * v_[i] + (1 - alpha_) * std::max(0.0, x.y()[i]) *
                                   std::max(0.0, x.v()[i]);
    }
    UpdateNorms();
  }
  // full_residual.cc:115-118
  static double pfb(double a, double b, double alpha) {
    const double fb = a + b - std::sqrt(a * a + b * b);
    return alpha * fb + (1.0 - alpha) * std::max(0.0, a) * std::max(0.0, b);
  }
  bool SameSize(const FullVariable& x) const {
    return x.nz_ == nz_ && x.nl_ == nl_ && x.nv_ == nv_;
  }
  double z_norm() const { return znorm_; }
  double l_norm() const { return lnorm_; }
  double v_norm() const { return vnorm_; }
  Vec& z() { return z_; }
  Vec& l() { return l_; }
  Vec& v() { return v_; }
  const Vec& z() const { return z_; }
  const Vec& l() const { return l_; }
  const Vec& v() const { return v_; }
  int nz_, nl_, nv_;

 private:
  void UpdateNorms() {
    znorm_ = norm2(z_);
    lnorm_ = norm2(l_);
    vnorm_ = norm2(v_);
  }
  void NullDataCheck() const {
    if (data_ == nullptr)
      throw std::runtime_error(
          "FullResidual tried to access problem data before it's linked.");
  }
  Vec z_, l_, v_;
  double alpha_ = 0.95;
  double znorm_ = 0.0, lnorm_ = 0.0, vnorm_ = 0.0;
  const Data* data_ = nullptr;
};

// ---------------------------------------------------------------------------
// full_feasibility.cc:12-96
class FullFeasibility {
 public:
  enum class FeasibilityStatus {
    FEASIBLE = 0,
    PRIMAL_INFEASIBLE = 1,
    DUAL_INFEASIBLE = 2,
    BOTH = 3
  };
  FullFeasibility(int nz, int nl, int nv) : nz_(nz), nl_(nl), nv_(nv) {
    if (nz <= 0 || nl < 0 || nv <= 0)
      throw std::runtime_error("Incorrect size inputs to FullFeasibility.");
    tz_.assign(nz, 0.0);
    tl_.assign(nl, 0.0);
    tv_.assign(nv, 0.0);
  }
  void LinkData(const Data* data) { data_ = data; }
  FeasibilityStatus CheckFeasibility(const FullVariable& x, double tol) {
    if (data_ == nullptr)
      throw std::runtime_error(
          "FullFeasibility tried to access problem data before it's linked.");
    // Workspaces are zero-filled before the b == 0 gemvs (the reference's
    // tv_ is uninitialised there).
    std::fill(tv_.begin(), tv_.end(), 0.0);
    std::fill(tl_.begin(), tl_.end(), 0.0);
    std::fill(tz_.begin(), tz_.end(), 0.0);
    data_->gemvA(x.z(), 1.0, 0.0, &tv_);
    double d1 = tv_[0];
    for (int i = 1; i < nv_; i++) d1 = std::max(d1, tv_[i]);
    data_->gemvG(x.z(), 1.0, 0.0, &tl_);
    const double d2 = inf_norm(tl_);
    data_->gemvH(x.z(), 1.0, 0.0, &tz_);
    const double d3 = inf_norm(tz_);
    std::fill(tz_.begin(), tz_.end(), 0.0);
    data_->axpyf(1.0, &tz_);
    const double d4 = dot(tz_.data(), x.z().data(), nz_);
    const double w = inf_norm(x.z());
    bool dual_feasible = true;
    if ((d1 <= w * tol) && (d2 <= tol * w) && (d3 <= tol * w) && (d4 < 0) &&
        (w > 1e-14))
      dual_feasible = false;

    std::fill(tz_.begin(), tz_.end(), 0.0);
    data_->gemvAT(x.v(), 1.0, 1.0, &tz_);
    data_->gemvGT(x.l(), 1.0, 1.0, &tz_);
    const double p1 = inf_norm(tz_);
    std::fill(tv_.begin(), tv_.end(), 0.0);
    data_->axpyb(1.0, &tv_);
    std::fill(tl_.begin(), tl_.end(), 0.0);
    data_->axpyh(1.0, &tl_);
    const double p2 = dot(tl_.data(), x.l().data(), nl_) +
                      dot(tv_.data(), x.v().data(), nv_);
    const double u = std::max(inf_norm(x.v()), inf_norm(x.l()));
    bool primal_feasible = true;
    if ((p1 <= tol * u) && (p2 < 0)) primal_feasible = false;

    if (primal_feasible && dual_feasible) return FeasibilityStatus::FEASIBLE;
    if (primal_feasible && !dual_feasible)
      return FeasibilityStatus::DUAL_INFEASIBLE;
    if (!primal_feasible && dual_feasible)
      return FeasibilityStatus::PRIMAL_INFEASIBLE;
    return FeasibilityStatus::BOTH;
  }

 private:
  int nz_, nl_, nv_;
  Vec tz_, tl_, tv_;
  const Data* data_ = nullptr;
};

// PFBGradient, duplicated verbatim in both reference solvers
// (riccati_linear_solver.cc:346-365, dense_cholesky_solver.cc:129-148).
inline void pfb_gradient(double a, double b, double alpha, double* g0,
                         double* g1) {
  const double zero_tolerance = 1e-13;
  const double r = std::sqrt(a * a + b * b);
  const double d = 1.0 / std::sqrt(2.0);
  if (r < zero_tolerance) {
    *g0 = alpha * (1.0 - d);
    *g1 = alpha * (1.0 - d);
  } else if ((a > 0) && (b > 0)) {
    *g0 = alpha * (1.0 - a / r) + (1.0 - alpha) * b;
    *g1 = alpha * (1.0 - b / r) + (1.0 - alpha) * a;
  } else {
    *g0 = alpha * (1.0 - a / r);
    *g1 = alpha * (1.0 - b / r);
  }
}

// ---------------------------------------------------------------------------
// dense_cholesky_solver.cc:12-127
class DenseCholeskySolver {
 public:
  DenseCholeskySolver(int nz, int nl, int nv)
      : nz_(nz), nl_(nl), nv_(nv), ldlt_(nz + nl) {
    if (nz <= 0 || nv <= 0 || nl < 0)
      throw std::runtime_error(
          "In DenseCholeskySolver: nz and nv must be > 0 and nl >= 0");
    K_.assign((size_t)(nz + nl) * (nz + nl), 0.0);
    E_.assign((size_t)nz * nz, 0.0);
    r1_.assign(nz + nl, 0.0);
    r2_.assign(nv, 0.0);
    Gamma_.assign(nv, 0.0);
    mus_.assign(nv, 0.0);
    gamma_.assign(nv, 0.0);
    B_.assign((size_t)nv * nz, 0.0);
  }
  void LinkData(const DenseData* data) { data_ = data; }
  void SetAlpha(double alpha) { alpha_ = alpha; }

  // dense_cholesky_solver.cc:32-79
  bool Initialize(const FullVariable& x, const FullVariable& xbar,
                  double sigma) {
    NullDataCheck();
    if (!x.SameSize(xbar))
      throw std::runtime_error(
          "In DenseCholeskySolver::Factor: inputs must be the same size");
    if (xbar.nz_ != nz_ || xbar.nv_ != nv_)
      throw std::runtime_error(
          "In DenseCholeskySolver::Factor: inputs must match object size.");
    if (sigma <= 0)
      throw std::runtime_error(
          "In DenseCholeskySolver::Factor: sigma must be positive.");
    const double* H = data_->H_;
    const double* G = data_->G_;
    const double* A = data_->A_;
    const int nk = nz_ + nl_;
    for (int j = 0; j < nz_; j++)
      for (int i = 0; i < nz_; i++)
        FBO_AT(E_, nz_, i, j) = FBO_AT(H, nz_, i, j) + (i == j ? sigma : 0.0);
    for (int i = 0; i < nv_; i++) {
      const double ys = x.y()[i] + sigma * (x.v()[i] - xbar.v()[i]);
      double g0, g1;
      pfb_gradient(ys, x.v()[i], alpha_, &g0, &g1);
      gamma_[i] = g0;
      mus_[i] = g1 + sigma * g0;
      Gamma_[i] = gamma_[i] / mus_[i];
    }
    // B = diag(Gamma)*A ; E += A'*B
    for (int j = 0; j < nz_; j++)
      for (int i = 0; i < nv_; i++)
        FBO_AT(B_, nv_, i, j) = Gamma_[i] * FBO_AT(A, nv_, i, j);
    for (int j = 0; j < nz_; j++)
      for (int i = 0; i < nz_; i++)
        FBO_AT(E_, nz_, i, j) +=
            dot(A + (size_t)i * nv_, B_.data() + (size_t)j * nv_, nv_);
    // K = [E . ; G -sigma I]; the upper-right block is never written
    // (dense_cholesky_solver.cc:67-69), LDLT reads the lower triangle only.
    for (int j = 0; j < nz_; j++)
      for (int i = 0; i < nz_; i++)
        FBO_AT(K_, nk, i, j) = FBO_AT(E_, nz_, i, j);
    for (int j = 0; j < nz_; j++)
      for (int i = 0; i < nl_; i++)
        FBO_AT(K_, nk, nz_ + i, j) = FBO_AT(G, nl_, i, j);
    for (int j = 0; j < nl_; j++)
      for (int i = 0; i < nl_; i++)
        FBO_AT(K_, nk, nz_ + i, nz_ + j) = (i == j ? -sigma : 0.0);
    return ldlt_.compute(K_.data());
  }

  // dense_cholesky_solver.cc:81-127
  bool Solve(const FullResidual& r, FullVariable* x) {
    if (x == nullptr)
      throw std::runtime_error(
          "In DenseCholeskySolver::Solve: x cannot be null.");
    if (!r.SameSize(*x))
      throw std::runtime_error(
          "In DenseCholeskySolver::Solve residual and variable objects must "
          "be the same size");
    if (x->nz_ != nz_ || x->nv_ != nv_ || x->nl_ != nl_)
      throw std::runtime_error(
          "In DenseCholeskySolver::Factor: inputs must match object size.");
    const double* A = data_->A_;
    const double* b = data_->b_;
    for (int i = 0; i < nv_; i++) r2_[i] = r.v()[i] / mus_[i];
    for (int j = 0; j < nz_; j++)
      r1_[j] = r.z()[j] - dot(A + (size_t)j * nv_, r2_.data(), nv_);
    for (int j = 0; j < nl_; j++) r1_[nz_ + j] = -r.l()[j];
    ldlt_.solve_inplace(r1_.data());
    for (int j = 0; j < nz_; j++) x->z()[j] = r1_[j];
    for (int j = 0; j < nl_; j++) x->l()[j] = r1_[nz_ + j];
    // v = (rv + gamma .* (A z)) ./ mus
    std::fill(r2_.begin(), r2_.end(), 0.0);
    gemv_n(A, nv_, nz_, 1.0, x->z().data(), r2_.data());
    for (int i = 0; i < nv_; i++) r2_[i] = gamma_[i] * r2_[i];
    for (int i = 0; i < nv_; i++) r2_[i] += r.v()[i];
    for (int i = 0; i < nv_; i++) x->v()[i] = r2_[i] / mus_[i];
    // y = b - A z
    Vec t(nv_, 0.0);
    gemv_n(A, nv_, nz_, 1.0, x->z().data(), t.data());
    for (int i = 0; i < nv_; i++) x->y()[i] = b[i] - t[i];
    return true;
  }
  Vec gamma_, mus_, Gamma_;

 private:
  void NullDataCheck() const {
    if (data_ == nullptr)
      throw std::runtime_error(
          "DenseCholeskySolver tried to access problem data before it's "
          "linked.");
  }
  int nz_, nl_, nv_;
  double alpha_ = 0.95;
  const DenseData* data_ = nullptr;
  Vec K_, E_, r1_, r2_, B_;
  Ldlt ldlt_;
};

// ---------------------------------------------------------------------------
// riccati_linear_solver.cc:17-344
class RiccatiLinearSolver {
 public:
  RiccatiLinearSolver(int N, int nx, int nu, int nc)
      : N_(N), nx_(nx), nu_(nu), nc_(nc) {
    if (N <= 0 || nx <= 0 || nu <= 0 || nc <= 0)
      throw std::runtime_error(
          "In RiccatiLinearSolver::RiccatiLinearSolver: all inputs must be "
          "positive.");
    nz_ = (N + 1) * (nx + nu);
    nl_ = (N + 1) * nx;
    nv_ = (N + 1) * nc;
    const int S = N + 1;
    Q_.assign((size_t)S * nx * nx, 0.0);
    S_.assign((size_t)S * nu * nx, 0.0);
    R_.assign((size_t)S * nu * nu, 0.0);
    P_.assign((size_t)S * nx * nu, 0.0);
    SG_.assign((size_t)S * nu * nu, 0.0);
    M_.assign((size_t)S * nx * nx, 0.0);
    L_.assign((size_t)S * nx * nx, 0.0);
    SM_.assign((size_t)S * nu * nx, 0.0);
    AM_.assign((size_t)S * nx * nx, 0.0);
    h_.assign((size_t)S * nx, 0.0);
    th_.assign((size_t)S * nx, 0.0);
    gamma_.assign(nv_, 0.0);
    mus_.assign(nv_, 0.0);
    Gamma_.assign(nv_, 0.0);
    Etemp_.assign((size_t)nc * nx, 0.0);
    Ltemp_.assign((size_t)nc * nu, 0.0);
    Linv_.assign((size_t)nx * nx, 0.0);
    tx_.assign(nx, 0.0);
    tu_.assign(nu, 0.0);
    tl_.assign(nx, 0.0);
    r1_.assign(nz_, 0.0);
    r2_.assign(nl_, 0.0);
    r3_.assign(nv_, 0.0);
  }
  void LinkData(const MpcData* data) { data_ = data; }
  void SetAlpha(double alpha) { alpha_ = alpha; }

  // riccati_linear_solver.cc:77-210
  bool Initialize(const FullVariable& x, const FullVariable& xbar,
                  double sigma) {
    if (!x.SameSize(xbar))
      throw std::runtime_error(
          "In RiccatiLinearSolver::Initialize: x and xbar are not the same "
          "size.");
    if (sigma <= 0)
      throw std::runtime_error(
          "In RiccatiLinearSolver::Initialize: sigma must be positive.");
    NullDataCheck();
    const int nx = nx_, nu = nu_, nc = nc_;
    for (int i = 0; i < nv_; i++) {
      const double ys = x.y()[i] + sigma * (x.v()[i] - xbar.v()[i]);
      double g0, g1;
      pfb_gradient(ys, x.v()[i], alpha_, &g0, &g1);
      gamma_[i] = g0;
      mus_[i] = g1 + sigma * g0;
      Gamma_[i] = gamma_[i] / mus_[i];
    }
    // Barrier-augmented stage Hessians (:101-123); only the lower triangles
    // of Q, R are written.
    for (int i = 0; i < N_ + 1; i++) {
      const double* Ei = data_->Ei(i);
      const double* Li = data_->Li(i);
      const double* Gi = Gamma_.data() + (size_t)i * nc;
      double* Q = Qs(i);
      double* R = Rs(i);
      double* S = Ss(i);
      for (int c = 0; c < nx; c++)
        for (int r = c; r < nx; r++)
          FBO_AT(Q, nx, r, c) =
              FBO_AT(data_->Qi(i), nx, r, c) + (r == c ? sigma : 0.0);
      for (int c = 0; c < nu; c++)
        for (int r = c; r < nu; r++)
          FBO_AT(R, nu, r, c) =
              FBO_AT(data_->Ri(i), nu, r, c) + (r == c ? sigma : 0.0);
      for (int k = 0; k < nu * nx; k++) S[k] = data_->Si(i)[k];
      for (int c = 0; c < nx; c++)
        for (int k = 0; k < nc; k++)
          FBO_AT(Etemp_, nc, k, c) = Gi[k] * FBO_AT(Ei, nc, k, c);
      for (int c = 0; c < nx; c++)
        for (int r = c; r < nx; r++)
          FBO_AT(Q, nx, r, c) +=
              dot(Ei + (size_t)r * nc, Etemp_.data() + (size_t)c * nc, nc);
      for (int c = 0; c < nu; c++)
        for (int k = 0; k < nc; k++)
          FBO_AT(Ltemp_, nc, k, c) = Gi[k] * FBO_AT(Li, nc, k, c);
      for (int c = 0; c < nu; c++)
        for (int r = c; r < nu; r++)
          FBO_AT(R, nu, r, c) +=
              dot(Li + (size_t)r * nc, Ltemp_.data() + (size_t)c * nc, nc);
      for (int c = 0; c < nx; c++)
        for (int r = 0; r < nu; r++)
          FBO_AT(S, nu, r, c) +=
              dot(Li + (size_t)r * nc, Etemp_.data() + (size_t)c * nc, nc);
    }
    // Base case L(0) = sqrt(sigma) I (:127).
    {
      double* L0 = Ls(0);
      std::fill(L0, L0 + (size_t)nx * nx, 0.0);
      for (int j = 0; j < nx; j++) FBO_AT(L0, nx, j, j) = std::sqrt(sigma);
    }
    for (int i = 0; i < N_; i++) {
      if (!FactorM(i)) return false;
      // AM = A*inv(M)' (:149-154)
      double* AM = AMs(i);
      for (int k = 0; k < nx * nx; k++) AM[k] = data_->Ai(i)[k];
      trsm_right_lt(Ms(i), nx, nx, AM, nx, nx);
      if (!FactorSG(i)) return false;
      // P = (AM*SM' - B)*inv(SG)' (:167-175)
      double* P = Ps(i);
      const double* SM = SMs(i);
      for (int c = 0; c < nu; c++)
        for (int r = 0; r < nx; r++) {
          double s = 0.0;
          for (int k = 0; k < nx; k++)
            s += FBO_AT(AM, nx, r, k) * FBO_AT(SM, nu, c, k);
          FBO_AT(P, nx, r, c) = s - FBO_AT(data_->Bi(i), nx, r, c);
        }
      trsm_right_lt(SGs(i), nu, nu, P, nx, nx);
      // L(i+1) = chol(sigma I + P P' + AM AM') (:177-183)
      double* Ln = Ls(i + 1);
      for (int c = 0; c < nx; c++)
        for (int r = 0; r < nx; r++) {
          double s = (r == c ? sigma : 0.0);
          double s1 = 0.0;
          for (int k = 0; k < nu; k++)
            s1 += FBO_AT(P, nx, r, k) * FBO_AT(P, nx, c, k);
          s += s1;
          double s2 = 0.0;
          for (int k = 0; k < nx; k++)
            s2 += FBO_AT(AM, nx, r, k) * FBO_AT(AM, nx, c, k);
          s += s2;
          FBO_AT(Ln, nx, r, c) = s;
        }
      if (llt_inplace_lower(Ln, nx, nx) >= 0) return false;
    }
    // Terminal stage (:186-206).
    if (!FactorM(N_)) return false;
    if (!FactorSG(N_)) return false;
    return true;
  }

  // riccati_linear_solver.cc:212-344
  bool Solve(const FullResidual& r, FullVariable* dx) {
    if (!r.SameSize(*dx))
      throw std::runtime_error(
          "In RiccatiLinearSolver::Solve: r and dx size mismatch.");
    NullDataCheck();
    const int nx = nx_, nu = nu_, ns = nx_ + nu_;
    r1_ = r.z();
    for (int i = 0; i < nv_; i++) r3_[i] = r.v()[i] / mus_[i];
    data_->gemvAT(r3_, -1.0, 1.0, &r1_);
    for (int i = 0; i < nl_; i++) r2_[i] = -r.l()[i];
    const double* r1 = r1_.data();
    const double* r2 = r2_.data();
    // Base case (:230-236)
    for (int j = 0; j < nx; j++) th(0)[j] = r2[j];
    for (int j = 0; j < nx; j++) hh(0)[j] = th(0)[j];
    trsv_l(Ls(0), nx, nx, hh(0));
    trsv_lt(Ls(0), nx, nx, hh(0));
    for (int j = 0; j < nx; j++) hh(0)[j] -= r1[j];
    // Forward sweep (:239-262)
    for (int i = 0; i < N_; i++) {
      for (int j = 0; j < nx; j++) tx_[j] = hh(i)[j];
      trsv_l(Ms(i), nx, nx, tx_.data());
      std::fill(tu_.begin(), tu_.end(), 0.0);
      gemv_n(SMs(i), nu, nx, 1.0, tx_.data(), tu_.data());
      for (int j = 0; j < nu; j++) tu_[j] += r1[(size_t)i * ns + nx + j];
      trsv_l(SGs(i), nu, nu, tu_.data());
      double* thn = th(i + 1);
      std::fill(thn, thn + nx, 0.0);
      gemv_n(Ps(i), nx, nu, 1.0, tu_.data(), thn);
      gemv_n(AMs(i), nx, nx, 1.0, tx_.data(), thn);
      for (int j = 0; j < nx; j++) thn[j] += r2[(size_t)(i + 1) * nx + j];
      double* hn = hh(i + 1);
      for (int j = 0; j < nx; j++) hn[j] = thn[j];
      trsv_l(Ls(i + 1), nx, nx, hn);
      trsv_lt(Ls(i + 1), nx, nx, hn);
      for (int j = 0; j < nx; j++) hn[j] -= r1[(size_t)(i + 1) * ns + j];
    }
    // Terminal step (:267-285)
    double* dz = dx->z().data();
    double* dl = dx->l().data();
    {
      const int i = N_;
      for (int j = 0; j < nx; j++) tx_[j] = hh(i)[j];
      trsv_l(Ms(i), nx, nx, tx_.data());
      std::fill(tu_.begin(), tu_.end(), 0.0);
      gemv_n(SMs(i), nu, nx, 1.0, tx_.data(), tu_.data());
      for (int j = 0; j < nu; j++) tu_[j] += r1[(size_t)i * ns + nx + j];
      trsv_l(SGs(i), nu, nu, tu_.data());
      trsv_lt(SGs(i), nu, nu, tu_.data());
      for (int j = 0; j < nx; j++) tx_[j] = hh(i)[j];
      trsv_l(Ms(i), nx, nx, tx_.data());
      gemv_t(SMs(i), nu, nx, 1.0, tu_.data(), tx_.data());
      trsv_lt(Ms(i), nx, nx, tx_.data());
      for (int j = 0; j < nx; j++) tx_[j] *= -1.0;
      for (int j = 0; j < nx; j++) tl_[j] = tx_[j] + th(i)[j];
      trsv_l(Ls(i), nx, nx, tl_.data());
      trsv_lt(Ls(i), nx, nx, tl_.data());
      for (int j = 0; j < nx; j++) tl_[j] *= -1.0;
      for (int j = 0; j < nx; j++) dz[(size_t)i * ns + j] = tx_[j];
      for (int j = 0; j < nu; j++) dz[(size_t)i * ns + nx + j] = tu_[j];
      for (int j = 0; j < nx; j++) dl[(size_t)i * nx + j] = tl_[j];
    }
    // Backward sweep (:297-327)
    for (int i = N_ - 1; i >= 0; i--) {
      const double* lp1 = dl + (size_t)(i + 1) * nx;
      for (int j = 0; j < nx; j++) tx_[j] = hh(i)[j];
      trsv_l(Ms(i), nx, nx, tx_.data());
      double* ui = dz + (size_t)i * ns + nx;
      std::fill(ui, ui + nu, 0.0);
      gemv_n(SMs(i), nu, nx, 1.0, tx_.data(), ui);
      for (int j = 0; j < nu; j++) ui[j] += r1[(size_t)i * ns + nx + j];
      trsv_l(SGs(i), nu, nu, ui);
      gemv_t(Ps(i), nx, nu, 1.0, lp1, ui);
      trsv_lt(SGs(i), nu, nu, ui);
      double* xi = dz + (size_t)i * ns;
      for (int j = 0; j < nx; j++) xi[j] = hh(i)[j];
      trsv_l(Ms(i), nx, nx, xi);
      gemv_t(SMs(i), nu, nx, 1.0, ui, xi);
      gemv_t(AMs(i), nx, nx, 1.0, lp1, xi);
      trsv_lt(Ms(i), nx, nx, xi);
      for (int j = 0; j < nx; j++) xi[j] *= -1.0;
      double* li = dl + (size_t)i * nx;
      for (int j = 0; j < nx; j++) li[j] = th(i)[j] + xi[j];
      trsv_l(Ls(i), nx, nx, li);
      trsv_lt(Ls(i), nx, nx, li);
      for (int j = 0; j < nx; j++) li[j] *= -1.0;
    }
    // dv = (rv + gamma .* (A dz)) ./ mus (:331-336)
    Vec& dv = dx->v();
    data_->gemvA(dx->z(), 1.0, 0.0, &r3_);
    for (int i = 0; i < nv_; i++) dv[i] = (r.v()[i] + gamma_[i] * r3_[i]) / mus_[i];
    // dy = b - A dz (:339-341)
    Vec& dy = dx->y();
    data_->gemvA(dx->z(), -1.0, 0.0, &dy);
    data_->axpyb(1.0, &dy);
    return true;
  }
  Vec gamma_, mus_, Gamma_;

 private:
  double* Qs(int i) { return Q_.data() + (size_t)i * nx_ * nx_; }
  double* Rs(int i) { return R_.data() + (size_t)i * nu_ * nu_; }
  double* Ss(int i) { return S_.data() + (size_t)i * nu_ * nx_; }
  double* Ps(int i) { return P_.data() + (size_t)i * nx_ * nu_; }
  double* SGs(int i) { return SG_.data() + (size_t)i * nu_ * nu_; }
  double* Ms(int i) { return M_.data() + (size_t)i * nx_ * nx_; }
  double* Ls(int i) { return L_.data() + (size_t)i * nx_ * nx_; }
  double* SMs(int i) { return SM_.data() + (size_t)i * nu_ * nx_; }
  double* AMs(int i) { return AM_.data() + (size_t)i * nx_ * nx_; }
  double* hh(int i) { return h_.data() + (size_t)i * nx_; }
  double* th(int i) { return th_.data() + (size_t)i * nx_; }

  // M(i) = chol(Q(i) + inv(L(i) L(i)')), SM(i) = S(i) inv(M(i))'
  // (:142-146, :156-161 and :187-201).
  bool FactorM(int i) {
    const int nx = nx_, nu = nu_;
    std::fill(Linv_.begin(), Linv_.end(), 0.0);
    for (int j = 0; j < nx; j++) FBO_AT(Linv_, nx, j, j) = 1.0;
    trsm_left_l(Ls(i), nx, nx, Linv_.data(), nx, nx);
    trsm_left_lt(Ls(i), nx, nx, Linv_.data(), nx, nx);
    double* M = Ms(i);
    const double* Q = Qs(i);
    for (int c = 0; c < nx; c++)
      for (int r = c; r < nx; r++)
        FBO_AT(M, nx, r, c) = FBO_AT(Q, nx, r, c) + FBO_AT(Linv_, nx, r, c);
    if (llt_inplace_lower(M, nx, nx) >= 0) return false;
    double* SM = SMs(i);
    for (int k = 0; k < nu * nx; k++) SM[k] = Ss(i)[k];
    trsm_right_lt(M, nx, nx, SM, nu, nu);
    return true;
  }
  // SG(i) = chol(R(i) - SM(i) SM(i)') (:163-165, :203-205); lower only.
  bool FactorSG(int i) {
    const int nx = nx_, nu = nu_;
    double* SG = SGs(i);
    const double* SM = SMs(i);
    const double* R = Rs(i);
    for (int c = 0; c < nu; c++)
      for (int r = c; r < nu; r++) {
        double s = 0.0;
        for (int k = 0; k < nx; k++)
          s += FBO_AT(SM, nu, r, k) * FBO_AT(SM, nu, c, k);
        FBO_AT(SG, nu, r, c) = FBO_AT(R, nu, r, c) - s;
      }
    return llt_inplace_lower(SG, nu, nu) < 0;
  }
  void NullDataCheck() const {
    if (data_ == nullptr)
      throw std::runtime_error(
          "RiccatiLinearSolver tried to access problem data before it's "
          "linked.");
  }
  int N_, nx_, nu_, nc_, nz_, nl_, nv_;
  double alpha_ = 0.95;
  const MpcData* data_ = nullptr;
  Vec Q_, S_, R_, P_, SG_, M_, L_, SM_, AM_, h_, th_;
  Vec Etemp_, Ltemp_, Linv_, tx_, tu_, tl_, r1_, r2_, r3_;
};

}  // namespace fbo

// fbstab::FBstabDense with the reference's interface
// (fbstab/fbstab_dense.h:50-194), backed by the MI355X library through the
// C-ABI.  See fbstab_mpc.h for the conventions.
#pragma once

#include <stdexcept>
#include <string>

#include "../fbstab_hip.h"
#include "dense_types.h"
#include "fbstab_algorithm.h"

namespace fbstab {

class FBstabDense {
 public:
  FBstabDense(const FBstabDense&) = delete;
  void operator=(const FBstabDense&) = delete;

  // fbstab_dense.h:55-64, fbstab_dense.cc:44-52
  struct ProblemData {
    ProblemData() = default;
    ProblemData(int nz, int nl, int nv) : H(nz, nz), G(nl, nz), A(nv, nz), f(nz), h(nl), b(nv) {}
    MatrixXd H, G, A;
    VectorXd f, h, b;
  };
  // Non-owning column-major views (the reference uses Eigen::Map).
  struct MatRef {
    MatRef(const double* p, int r, int c) : ptr(p), r_(r), c_(c) {}
    // from any column-major view with data() / rows() / cols(): Eigen::Map<MatrixXd>
    template <class M>
    MatRef(const M& m, decltype(static_cast<void>(m.data()), static_cast<void>(m.rows()), 0) = 0)
        : ptr(m.data()), r_(static_cast<int>(m.rows())), c_(static_cast<int>(m.cols())) {}
    const double* data() const { return ptr; }
    int rows() const { return r_; }
    int cols() const { return c_; }
    double operator()(int i, int j) const { return ptr[i + static_cast<size_t>(j) * r_]; }
    const double* ptr;
    int r_, c_;
  };
  struct VecRef {
    VecRef(double* p, int n_) : ptr(p), n(n_) {}
    // from any view with data() / size(): Eigen::Map<VectorXd>
    template <class V>
    VecRef(const V& v, decltype(static_cast<void>(v.data()), static_cast<void>(v.size()), 0) = 0)
        : ptr(const_cast<double*>(v.data())), n(static_cast<int>(v.size())) {}
    double* data() const { return ptr; }
    int size() const { return n; }
    double& operator()(int i) const { return ptr[i]; }
    void fill(double a) const { for (int i = 0; i < n; i++) ptr[i] = a; }
    double* ptr;
    int n;
  };
  // fbstab_dense.h:67-82
  struct ProblemDataRef {
    ProblemDataRef() = delete;
    ProblemDataRef(const MatRef* H_, const VecRef* f_, const MatRef* G_, const VecRef* h_,
                   const MatRef* A_, const VecRef* b_)
        : H(*H_), G(*G_), A(*A_), f(*f_), h(*h_), b(*b_) {}
    // the reference's signature: pointers to Eigen::Map<MatrixXd> / Eigen::Map<VectorXd>
    // (fbstab_dense.h:69-74), or to any types with the same accessors
    template <class M, class V>
    ProblemDataRef(const M* H_, const V* f_, const M* G_, const V* h_, const M* A_, const V* b_)
        : H(*H_), G(*G_), A(*A_), f(*f_), h(*h_), b(*b_) {}
    MatRef H, G, A;
    VecRef f, h, b;
  };
  // fbstab_dense.h:85-92
  struct Variable {
    Variable(int nz, int nl, int nv)
        : z(VectorXd::Zero(nz)), l(VectorXd::Zero(nl)), v(VectorXd::Zero(nv)), y(VectorXd::Zero(nv)) {}
    VectorXd z, l, v, y;
  };
  // fbstab_dense.h:95-107
  struct VariableRef {
    VariableRef() = delete;
    VariableRef(VecRef* z_, VecRef* l_, VecRef* v_, VecRef* y_) : z(*z_), l(*l_), v(*v_), y(*y_) {}
    // the reference's signature: pointers to Eigen::Map<VectorXd> (fbstab_dense.h:97-100)
    template <class V>
    VariableRef(V* z_, V* l_, V* v_, V* y_) : z(*z_), l(*l_), v(*v_), y(*y_) {}
    void fill(double a) { z.fill(a); l.fill(a); v.fill(a); y.fill(a); }
    VecRef z, l, v, y;
  };

  struct Options : public AlgorithmParameters {};

  // fbstab_dense.cc:18-42
  FBstabDense(int nz, int nl, int nv, int device = 0) : nz_(nz), nl_(nl), nv_(nv) {
    if (nz < 1 || nv < 1 || nl < 0)
      throw std::runtime_error(
          "In FBstabDense::FBstabDense: nz and nv must be positive, nl nonnegative.");
    if (fbstab_hip_dense_create(nz, nl, nv, 1, device, &h_) != FBSTAB_HIP_OK)
      throw std::runtime_error(std::string("In FBstabDense::FBstabDense: ") + fbstab_hip_last_error());
    opts_ = DefaultOptions();
  }
  ~FBstabDense() { fbstab_hip_dense_destroy(h_); }

  // Where the memory behind qp and x lives (see FBstabMpc::SetMemory).
  enum class Memory { HOST, DEVICE };
  void SetMemory(Memory m) { memory_ = m; }

  // fbstab_dense.h:136-149 (DenseData validation dense_data.h:53-66,
  // ValidateInputs fbstab_dense.h:167-180)
  template <class InputData, class InputVariable, class OutStream>
  SolverOut Solve(const InputData& qp, InputVariable* x, const OutStream& os) {
    const int fz = static_cast<int>(qp.f.size()), hz = static_cast<int>(qp.h.size()),
              bz = static_cast<int>(qp.b.size());
    if (qp.H.rows() != qp.H.cols() || qp.H.rows() != fz)
      throw std::runtime_error("In DenseData::DenseData: H must be square and the same size as f");
    if (qp.A.cols() != qp.H.rows() || qp.A.rows() != bz)
      throw std::runtime_error("In DenseData::DenseData: Sizing of data defining Az <= b is inconsistent.");
    if ((hz > 0 && qp.G.cols() != qp.H.rows()) || qp.G.rows() != hz)
      throw std::runtime_error("In DenseData::DenseData: Sizing of Gz = h is inconsistent.");
    if (nz_ != fz || nv_ != bz || nl_ != hz)
      throw std::runtime_error("In FBstabDense::Solve: mismatch between *this and data dimensions.");
    if (nz_ != static_cast<int>(x->z.size()) || static_cast<int>(x->l.size()) != nl_ ||
        nv_ != static_cast<int>(x->v.size()))
      throw std::runtime_error(
          "In FBstabDense::Solve: mismatch between *this and initial guess dimensions.");
    fbstab_dense_batch_t b;
    const double* p[FBSTAB_DENSE_NARR] = {qp.H.data(), qp.f.data(), qp.G.data(),
                                          qp.h.data(), qp.A.data(), qp.b.data()};
    const long long len[FBSTAB_DENSE_NARR] = {(long long)nz_ * nz_, nz_, (long long)nl_ * nz_, nl_,
                                              (long long)nv_ * nz_, nv_};
    for (int i = 0; i < FBSTAB_DENSE_NARR; i++) { b.base[i] = p[i]; b.stride[i] = len[i]; }
    fbstab_var_batch_t v;
    v.base[0] = x->z.data(); v.base[1] = x->l.data(); v.base[2] = x->v.data(); v.base[3] = x->y.data();
    v.stride[0] = nz_; v.stride[1] = nl_; v.stride[2] = nv_; v.stride[3] = nv_;
    fbstab_solver_out_t out;
    if (memory_ == Memory::DEVICE) {
      if (opts_.display_level > Display::FINAL)
        throw std::runtime_error("In FBstabDense::Solve: the iteration displays need host-resident data.");
      const int fl = FBSTAB_HIP_DEVICE_POINTERS | FBSTAB_HIP_OUT_ON_HOST;
      if (opts_.display_level == Display::OFF) {
        if (fbstab_hip_dense_solve_batch(h_, 1, &b, &v, &out, fl, nullptr) != FBSTAB_HIP_OK)
          throw std::runtime_error(std::string("In FBstabDense::Solve: ") + fbstab_hip_last_error());
        return detail::FromC(out);
      }
      // (FINAL on device-resident data: the summary's residual blocks are always those of the
      // returned point - for an infeasibility certificate or at the proximal iteration limit the
      // reference prints the residual of an earlier iterate, which would need a second solve)
      double nrm[4];
      if (fbstab_hip_dense_solve_batch_final(h_, 1, &b, &v, &out, nrm, fl, nullptr) != FBSTAB_HIP_OK)
        throw std::runtime_error(std::string("In FBstabDense::Solve: ") + fbstab_hip_last_error());
      SolverOut st = detail::FromC(out);
      detail::PrintFinalBlock(nrm, st, opts_, os);
      return st;
    }
    if (opts_.display_level == Display::FINAL) {
      // the reference's default level: the batch kernels, then the summary block.  The guess
      // is kept for the exits whose printed residual belongs to an earlier iterate.
      std::vector<double> gz(x->z.data(), x->z.data() + nz_), gl(x->l.data(), x->l.data() + nl_),
          gv(x->v.data(), x->v.data() + nv_);
      double nrm[4];
      if (fbstab_hip_dense_solve_batch_final(h_, 1, &b, &v, &out, nrm, FBSTAB_HIP_HOST_POINTERS, nullptr) !=
          FBSTAB_HIP_OK)
        throw std::runtime_error(std::string("In FBstabDense::Solve: ") + fbstab_hip_last_error());
      if (detail::FinalNormsAtReturnedPoint(out, opts_) || out.eflag == FBSTAB_DIVERGENCE ||
          out.eflag == FBSTAB_SATURATE_ERROR) {
        SolverOut st = detail::FromC(out);
        detail::PrintFinalBlock(nrm, st, opts_, os);
        return st;
      }
      for (int i = 0; i < nz_; i++) x->z.data()[i] = gz[i];
      for (int i = 0; i < nl_; i++) x->l.data()[i] = gl[i];
      for (int i = 0; i < nv_; i++) x->v.data()[i] = gv[i];
    }
    if (opts_.display_level >= Display::FINAL) {
      // every display level prints component norms (|rz| |rl| |rv|, impl:411-541):
      // the traced solve returns the lines of the reference's display as records
      std::vector<fbstab_trace_record_t> rec(detail::TraceCapacity(opts_));
      int n = 0;
      if (fbstab_hip_dense_solve_traced(h_, &b, &v, &out, rec.data(), static_cast<int>(rec.size()), &n) !=
          FBSTAB_HIP_OK)
        throw std::runtime_error(std::string("In FBstabDense::Solve: ") + fbstab_hip_last_error());
      SolverOut st = detail::FromC(out);
      if (n > static_cast<int>(rec.size())) n = static_cast<int>(rec.size());
      detail::PrintTrace(rec.data(), n, st, opts_, os);
      return st;
    }
    if (fbstab_hip_dense_solve_batch(h_, 1, &b, &v, &out, FBSTAB_HIP_HOST_POINTERS, nullptr) != FBSTAB_HIP_OK)
      throw std::runtime_error(std::string("In FBstabDense::Solve: ") + fbstab_hip_last_error());
    return detail::FromC(out);
  }
  template <class InputData, class InputVariable>
  SolverOut Solve(const InputData& qp, InputVariable* x) {
    StandardOutput os;
    return Solve(qp, x, os);
  }

  void UpdateOptions(const Options& options) {
    opts_ = options;
    opts_.ValidateOptions();
    fbstab_options_t o = opts_.ToC();
    fbstab_hip_dense_set_options(h_, &o);
  }
  static Options DefaultOptions() { Options o; o.DefaultParameters(); return o; }
  static Options ReliableOptions() { Options o; o.ReliableParameters(); return o; }

  // Not in the reference: the elimination order of the KKT factorisation
  // (fbstab_hip_dense_set_factorisation).  PIVOTED - Eigen::LDLT's rule,
  // dense_cholesky_solver.cc:70-79 - is the default and reproduces the reference's iteration
  // counts; NATURAL and AUTO are faster and may take a different number of iterations on
  // ill-conditioned QPs (same solutions to the tolerance).
  enum class FactorisationOrder { AUTO = 0, PIVOTED = 1, NATURAL = 2 };
  void SetFactorisationOrder(FactorisationOrder order, int spread_bits = 0) {
    if (fbstab_hip_dense_set_factorisation(h_, static_cast<int>(order), spread_bits) != FBSTAB_HIP_OK)
      throw std::runtime_error(std::string("In FBstabDense::SetFactorisationOrder: ") + fbstab_hip_last_error());
  }

 private:
  int nz_, nl_, nv_;
  Options opts_;
  Memory memory_ = Memory::HOST;
  fbstab_dense_handle_t h_ = nullptr;
};

class FBstabDenseBatch {
 public:
  FBstabDenseBatch(const FBstabDenseBatch&) = delete;
  void operator=(const FBstabDenseBatch&) = delete;
  FBstabDenseBatch(int nz, int nl, int nv, int max_batch, int device = 0) {
    if (fbstab_hip_dense_create(nz, nl, nv, max_batch, device, &h_) != FBSTAB_HIP_OK)
      throw std::runtime_error(std::string("In FBstabDenseBatch: ") + fbstab_hip_last_error());
  }
  ~FBstabDenseBatch() { fbstab_hip_dense_destroy(h_); }
  void UpdateOptions(const FBstabDense::Options& options) {
    FBstabDense::Options o = options;
    o.ValidateOptions();
    fbstab_options_t c = o.ToC();
    fbstab_hip_dense_set_options(h_, &c);
  }
  void SetFactorisationOrder(FBstabDense::FactorisationOrder order, int spread_bits = 0) {
    if (fbstab_hip_dense_set_factorisation(h_, static_cast<int>(order), spread_bits) != FBSTAB_HIP_OK)
      throw std::runtime_error(std::string("In FBstabDenseBatch::SetFactorisationOrder: ") + fbstab_hip_last_error());
  }
  void Solve(int batch, const fbstab_dense_batch_t& data, const fbstab_var_batch_t& x,
             fbstab_solver_out_t* out, int flags = FBSTAB_HIP_HOST_POINTERS, void* stream = nullptr) {
    if (fbstab_hip_dense_solve_batch(h_, batch, &data, &x, out, flags, stream) != FBSTAB_HIP_OK)
      throw std::runtime_error(std::string("In FBstabDenseBatch::Solve: ") + fbstab_hip_last_error());
  }

 private:
  fbstab_dense_handle_t h_ = nullptr;
};

}  // namespace fbstab

// fbstab::FBstabMpc with the reference's interface (fbstab/fbstab_mpc.h:56-243),
// backed by the MI355X library through the C-ABI (include/fbstab_hip.h).  A
// reference user switches by including this header instead of the reference's
// and linking libfbstab_hip.so; ProblemData / ProblemDataRef / Variable /
// VariableRef / Options / Solve / UpdateOptions keep their names, argument
// meaning and error behaviour.  FBstabMpcBatch is the batched entry point this
// library adds (one call, many QPs, host or device memory).
#pragma once

#include <stdexcept>
#include <string>
#include <vector>

#include "../fbstab_hip.h"
#include "dense_types.h"
#include "fbstab_algorithm.h"
#include "matrix_sequence.h"

namespace fbstab {

class FBstabMpc {
 public:
  FBstabMpc(const FBstabMpc&) = delete;
  void operator=(const FBstabMpc&) = delete;

  // fbstab_mpc.h:67-81
  struct ProblemData {
    ProblemData() = default;
    MatrixSequence Q, R, S, q, r, A, B, c, E, L, d;
    VectorXd x0;
  };

  // Non-owning view of x0 (the reference uses Eigen::Map<const VectorXd>).
  struct ConstVectorRef {
    ConstVectorRef() : ptr(nullptr), n(0) {}
    ConstVectorRef(const double* p, int n_) : ptr(p), n(n_) {}
    const double* data() const { return ptr; }
    int size() const { return n; }
    const double* ptr;
    int n;
  };
  struct VectorRef {
    VectorRef(double* p, int n_) : ptr(p), n(n_) {}
    // from anything that views a vector of doubles through data() / size(), the
    // reference's Eigen::Map<Eigen::VectorXd> in particular (fbstab_mpc.h:139-150)
    template <class V>
    VectorRef(V& v, decltype(static_cast<void>(v.data()), static_cast<void>(v.size()), 0) = 0)
        : ptr(v.data()), n(static_cast<int>(v.size())) {}
    double* data() const { return ptr; }
    int size() const { return n; }
    double& operator()(int i) const { return ptr[i]; }
    void fill(double a) const { for (int i = 0; i < n; i++) ptr[i] = a; }
    double* ptr;
    int n;
  };

  // fbstab_mpc.h:90-120
  struct ProblemDataRef {
    ProblemDataRef() {}
    template <class Vector>
    void SetX0(const Vector& x0_) { x0 = ConstVectorRef(x0_.data(), static_cast<int>(x0_.size())); }
    ProblemDataRef(const MatrixSequence* Q_, const MatrixSequence* R_, const MatrixSequence* S_,
                   const MatrixSequence* q_, const MatrixSequence* r_, const MatrixSequence* A_,
                   const MatrixSequence* B_, const MatrixSequence* c_, const MatrixSequence* E_,
                   const MatrixSequence* L_, const MatrixSequence* d_, const VectorXd* x0_)
        : Q(*Q_), R(*R_), S(*S_), q(*q_), r(*r_), A(*A_), B(*B_), c(*c_), E(*E_), L(*L_), d(*d_),
          x0(x0_->data(), static_cast<int>(x0_->size())) {}
    MapMatrixSequence Q, R, S, q, r, A, B, c, E, L, d;
    ConstVectorRef x0;
  };

  // fbstab_mpc.h:126-136, fbstab_mpc.cc:38-46
  struct Variable {
    Variable(int N, int nx, int nu, int nc)
        : z(VectorXd::Zero((N + 1) * (nx + nu))), l(VectorXd::Zero((N + 1) * nx)),
          v(VectorXd::Zero((N + 1) * nc)), y(VectorXd::Zero((N + 1) * nc)) {}
    // s = (N, nx, nu, nc), sizes carried as doubles like the reference's
    // Eigen::Vector4d (fbstab_mpc.h:130, fbstab_mpc.cc:45)
    explicit Variable(const Vector4d& s)
        : Variable(static_cast<int>(s(0)), static_cast<int>(s(1)), static_cast<int>(s(2)), static_cast<int>(s(3))) {}
    VectorXd z, l, v, y;
  };
  // fbstab_mpc.h:139-150.  The arguments are views (taken by value like the
  // reference's Eigen::Map arguments): VectorRef itself, Eigen::Map<VectorXd>, or any
  // type with data() / size().
  struct VariableRef {
    VariableRef(VectorRef z_, VectorRef l_, VectorRef v_, VectorRef y_) : z(z_), l(l_), v(v_), y(y_) {}
    template <class V>
    VariableRef(V z_, V l_, V v_, V y_, decltype(static_cast<void>(z_.data()), 0) = 0)
        : z(z_.data(), static_cast<int>(z_.size())), l(l_.data(), static_cast<int>(l_.size())),
          v(v_.data(), static_cast<int>(v_.size())), y(y_.data(), static_cast<int>(y_.size())) {}
    void fill(double a) { z.fill(a); l.fill(a); v.fill(a); y.fill(a); }
    VectorRef z, l, v, y;
  };

  struct Options : public AlgorithmParameters {};

  // fbstab_mpc.cc:61-89: allocates the (device) workspace.
  FBstabMpc(int N, int nx, int nu, int nc, int device = 0) : N_(N), nx_(nx), nu_(nu), nc_(nc) {
    if (N < 1 || nx < 1 || nu < 1 || nc < 1)
      throw std::runtime_error("In FBstabMpc::FBstabMpc: problem sizes must be positive.");
    nz_ = (nx + nu) * (N + 1);
    nl_ = nx * (N + 1);
    nv_ = nc * (N + 1);
    if (fbstab_hip_mpc_create(N, nx, nu, nc, 1, device, &h_) != FBSTAB_HIP_OK)
      throw std::runtime_error(std::string("In FBstabMpc::FBstabMpc: ") + fbstab_hip_last_error());
    opts_ = DefaultOptions();
  }
  // s = (N, nx, nu, nc) (fbstab_mpc.h:168, fbstab_mpc.cc:91)
  explicit FBstabMpc(const Vector4d& s, int device = 0)
      : FBstabMpc(static_cast<int>(s(0)), static_cast<int>(s(1)), static_cast<int>(s(2)), static_cast<int>(s(3)),
                  device) {}
  ~FBstabMpc() { fbstab_hip_mpc_destroy(h_); }

  // Where the memory behind qp and x lives.  The reference's ProblemDataRef /
  // VariableRef exist so that a caller's buffers are used without copies
  // (fbstab_mpc.h:90-150); here those buffers may be DEVICE memory: after
  // SetMemory(Memory::DEVICE) Solve hands the pointers to the kernel as they are
  // (no staging copies; the iteration displays ITER / ITER_DETAILED need host memory).
  enum class Memory { HOST, DEVICE };
  void SetMemory(Memory m) { memory_ = m; }

  // fbstab_mpc.h:181-195
  template <class InputData, class InputVariable, class OutStream>
  SolverOut Solve(const InputData& qp, InputVariable* x, const OutStream& os) {
    ValidateData(qp);
    if (x->z.size() != nz_ || x->l.size() != nl_ || x->v.size() != nv_ || x->y.size() != nv_)
      throw std::runtime_error(
          "In FBstabMpc::Solve: mismatch between *this and initial guess dimensions.");
    fbstab_mpc_batch_t b;
    const double* p[FBSTAB_MPC_NSEQ] = {qp.Q.data(), qp.R.data(), qp.S.data(), qp.q.data(),
                                        qp.r.data(), qp.A.data(), qp.B.data(), qp.c.data(),
                                        qp.E.data(), qp.L.data(), qp.d.data(), qp.x0.data()};
    const int len[FBSTAB_MPC_NSEQ] = {qp.Q.size(), qp.R.size(), qp.S.size(), qp.q.size(),
                                      qp.r.size(), qp.A.size(), qp.B.size(), qp.c.size(),
                                      qp.E.size(), qp.L.size(), qp.d.size(),
                                      static_cast<int>(qp.x0.size())};
    for (int i = 0; i < FBSTAB_MPC_NSEQ; i++) { b.base[i] = p[i]; b.stride[i] = len[i]; }
    fbstab_var_batch_t v;
    v.base[0] = x->z.data(); v.base[1] = x->l.data(); v.base[2] = x->v.data(); v.base[3] = x->y.data();
    v.stride[0] = nz_; v.stride[1] = nl_; v.stride[2] = nv_; v.stride[3] = nv_;
    fbstab_solver_out_t out;
    if (memory_ == Memory::DEVICE) {
      if (opts_.display_level > Display::FINAL)
        throw std::runtime_error("In FBstabMpc::Solve: the iteration displays need host-resident data.");
      const int fl = FBSTAB_HIP_DEVICE_POINTERS | FBSTAB_HIP_OUT_ON_HOST;
      if (opts_.display_level == Display::OFF) {
        if (fbstab_hip_mpc_solve_batch(h_, 1, &b, &v, &out, fl, nullptr) != FBSTAB_HIP_OK)
          throw std::runtime_error(std::string("In FBstabMpc::Solve: ") + fbstab_hip_last_error());
        return detail::FromC(out);
      }
      // (FINAL on device-resident data: the summary's residual blocks are always those of the
      // returned point - for an infeasibility certificate or at the proximal iteration limit the
      // reference prints the residual of an earlier iterate, which would need a second solve)
      double nrm[4];
      if (fbstab_hip_mpc_solve_batch_final(h_, 1, &b, &v, &out, nrm, fl, nullptr) != FBSTAB_HIP_OK)
        throw std::runtime_error(std::string("In FBstabMpc::Solve: ") + fbstab_hip_last_error());
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
      if (fbstab_hip_mpc_solve_batch_final(h_, 1, &b, &v, &out, nrm, FBSTAB_HIP_HOST_POINTERS, nullptr) !=
          FBSTAB_HIP_OK)
        throw std::runtime_error(std::string("In FBstabMpc::Solve: ") + fbstab_hip_last_error());
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
      if (fbstab_hip_mpc_solve_traced(h_, &b, &v, &out, rec.data(), static_cast<int>(rec.size()), &n) !=
          FBSTAB_HIP_OK)
        throw std::runtime_error(std::string("In FBstabMpc::Solve: ") + fbstab_hip_last_error());
      SolverOut st = detail::FromC(out);
      if (n > static_cast<int>(rec.size())) n = static_cast<int>(rec.size());
      detail::PrintTrace(rec.data(), n, st, opts_, os);
      return st;
    }
    if (fbstab_hip_mpc_solve_batch(h_, 1, &b, &v, &out, FBSTAB_HIP_HOST_POINTERS, nullptr) != FBSTAB_HIP_OK)
      throw std::runtime_error(std::string("In FBstabMpc::Solve: ") + fbstab_hip_last_error());
    return detail::FromC(out);
  }
  template <class InputData, class InputVariable>
  SolverOut Solve(const InputData& qp, InputVariable* x) {
    StandardOutput os;
    return Solve(qp, x, os);
  }

  // fbstab_mpc.cc:96-100 -> UpdateParameters + ValidateOptions
  void UpdateOptions(const Options& options) {
    opts_ = options;
    opts_.ValidateOptions();
    fbstab_options_t o = opts_.ToC();
    fbstab_hip_mpc_set_options(h_, &o);
  }
  static Options DefaultOptions() { Options o; o.DefaultParameters(); return o; }
  static Options ReliableOptions() { Options o; o.ReliableParameters(); return o; }

 private:
  // MpcData::ValidateInputs (mpc_data.cc:291-363) + ValidateInputSizes
  // (fbstab_mpc.h:229-242).
  template <class InputData>
  void ValidateData(const InputData& qp) const {
    const int N = qp.Q.length();
    if (N <= 0) throw std::runtime_error("Horizon length must be at least 1.");
    bool OK = N == qp.R.length() && N == qp.S.length() && N == qp.q.length() && N == qp.r.length() &&
              (N - 1) == qp.A.length() && (N - 1) == qp.B.length() && (N - 1) == qp.c.length() &&
              N == qp.E.length() && N == qp.L.length() && N == qp.d.length();
    if (!OK) throw std::runtime_error("Sequence length mismatch in input data to MpcData.");
    const int nx = qp.Q.rows();
    if (static_cast<int>(qp.x0.size()) != nx) throw std::runtime_error("Size mismatch in x0 input to MpcData.");
    if (qp.Q.cols() != nx) throw std::runtime_error("Size mismatch in Q input to MpcData.");
    if (qp.S.cols() != nx) throw std::runtime_error("Size mismatch in S input to MpcData.");
    if (qp.q.rows() != nx) throw std::runtime_error("Size mismatch in q input to MpcData.");
    if (qp.E.cols() != nx) throw std::runtime_error("Size mismatch in E input to MpcData.");
    if (qp.A.rows() != nx || qp.A.cols() != nx) throw std::runtime_error("Size mismatch in A input to MpcData.");
    if (qp.B.rows() != nx) throw std::runtime_error("Size mismatch in B input to MpcData.");
    if (qp.c.rows() != nx) throw std::runtime_error("Size mismatch in c input to MpcData.");
    const int nu = qp.R.rows();
    if (qp.R.cols() != nu) throw std::runtime_error("Size mismatch in R input to MpcData.");
    if (qp.S.rows() != nu) throw std::runtime_error("Size mismatch in S input to MpcData.");
    if (qp.r.rows() != nu) throw std::runtime_error("Size mismatch in r input to MpcData.");
    if (qp.L.cols() != nu) throw std::runtime_error("Size mismatch in L input to MpcData.");
    if (qp.B.cols() != nu) throw std::runtime_error("Size mismatch in B input to MpcData.");
    const int nc = qp.E.rows();
    if (qp.L.rows() != nc) throw std::runtime_error("Size mismatch in L input to MpcData.");
    if (qp.d.rows() != nc) throw std::runtime_error("Size mismatch in d input to MpcData.");
    if (qp.B.length() != N_ || nx != nx_ || nu != nu_ || nc != nc_)
      throw std::runtime_error("In FBstabMpc::Solve: mismatch between *this and data dimensions.");
  }

  int N_, nx_, nu_, nc_, nz_, nl_, nv_;
  Options opts_;
  Memory memory_ = Memory::HOST;
  fbstab_mpc_handle_t h_ = nullptr;
};

// Batched solves (new in this library): `batch` QPs of one size in one call.
// Arrays follow fbstab_mpc_batch_t (include/fbstab_hip.h): base pointer +
// stride per sequence, host or device memory.
class FBstabMpcBatch {
 public:
  FBstabMpcBatch(const FBstabMpcBatch&) = delete;
  void operator=(const FBstabMpcBatch&) = delete;
  FBstabMpcBatch(int N, int nx, int nu, int nc, int max_batch, int device = 0) {
    if (N < 1 || nx < 1 || nu < 1 || nc < 1)
      throw std::runtime_error("In FBstabMpc::FBstabMpc: problem sizes must be positive.");
    if (fbstab_hip_mpc_create(N, nx, nu, nc, max_batch, device, &h_) != FBSTAB_HIP_OK)
      throw std::runtime_error(std::string("In FBstabMpcBatch: ") + fbstab_hip_last_error());
  }
  ~FBstabMpcBatch() { fbstab_hip_mpc_destroy(h_); }
  void UpdateOptions(const FBstabMpc::Options& options) {
    FBstabMpc::Options o = options;
    o.ValidateOptions();
    fbstab_options_t c = o.ToC();
    fbstab_hip_mpc_set_options(h_, &c);
  }
  // flags: FBSTAB_HIP_HOST_POINTERS or FBSTAB_HIP_DEVICE_POINTERS [| FBSTAB_HIP_ASYNC]
  void Solve(int batch, const fbstab_mpc_batch_t& data, const fbstab_var_batch_t& x,
             fbstab_solver_out_t* out, int flags = FBSTAB_HIP_HOST_POINTERS, void* stream = nullptr) {
    if (fbstab_hip_mpc_solve_batch(h_, batch, &data, &x, out, flags, stream) != FBSTAB_HIP_OK)
      throw std::runtime_error(std::string("In FBstabMpcBatch::Solve: ") + fbstab_hip_last_error());
  }

 private:
  fbstab_mpc_handle_t h_ = nullptr;
};

// The same batches over several GPUs of one node, from ONE host thread: a solver per
// device, shard d = the next counts[d] QPs of the batch with its arrays resident on
// devices[d]; the shards run side by side, solutions and SolverOut records are gathered
// to devices[root] in one RCCL operation over xGMI (fbstab_hip_mpc_solve_batch_sharded).
class FBstabMpcSharded {
 public:
  FBstabMpcSharded(const FBstabMpcSharded&) = delete;
  void operator=(const FBstabMpcSharded&) = delete;
  FBstabMpcSharded(int N, int nx, int nu, int nc, const std::vector<int>& devices, int max_batch_per_device) {
    if (N < 1 || nx < 1 || nu < 1 || nc < 1)
      throw std::runtime_error("In FBstabMpc::FBstabMpc: problem sizes must be positive.");
    if (fbstab_hip_shard_group_create(static_cast<int>(devices.size()), devices.data(), &g_) != FBSTAB_HIP_OK)
      throw std::runtime_error(std::string("In FBstabMpcSharded: ") + fbstab_hip_last_error());
    h_.assign(devices.size(), nullptr);
    for (size_t d = 0; d < devices.size(); d++)
      if (fbstab_hip_mpc_create(N, nx, nu, nc, max_batch_per_device, devices[d], &h_[d]) != FBSTAB_HIP_OK) {
        const std::string why = fbstab_hip_last_error();
        Release();
        throw std::runtime_error("In FBstabMpcSharded: " + why);
      }
  }
  ~FBstabMpcSharded() { Release(); }
  int devices() const { return static_cast<int>(h_.size()); }
  void UpdateOptions(const FBstabMpc::Options& options) {
    FBstabMpc::Options o = options;
    o.ValidateOptions();
    fbstab_options_t c = o.ToC();
    for (fbstab_mpc_handle_t h : h_) fbstab_hip_mpc_set_options(h, &c);
  }
  // counts, data, x, out: one entry per device (device memory of that device);
  // root_x / root_out: the whole batch on devices[root]
  void Solve(const std::vector<int>& counts, const std::vector<fbstab_mpc_batch_t>& data,
             const std::vector<fbstab_var_batch_t>& x, const std::vector<fbstab_solver_out_t*>& out, int root,
             const fbstab_var_batch_t& root_x, fbstab_solver_out_t* root_out) {
    if (counts.size() != h_.size() || data.size() != h_.size() || x.size() != h_.size() || out.size() != h_.size())
      throw std::runtime_error("In FBstabMpcSharded::Solve: one entry per device is required.");
    if (fbstab_hip_mpc_solve_batch_sharded(g_, h_.data(), counts.data(), data.data(), x.data(), out.data(), root,
                                           &root_x, root_out) != FBSTAB_HIP_OK)
      throw std::runtime_error(std::string("In FBstabMpcSharded::Solve: ") + fbstab_hip_last_error());
  }
  // BASELINE configs[4]: every device sweeps its trajectories, the applied inputs are
  // gathered once at the end (fbstab_hip_mpc_receding_sweep_sharded)
  void RecedingSweep(const std::vector<int>& counts, const std::vector<fbstab_mpc_batch_t>& data,
                     const std::vector<fbstab_var_batch_t>& x, const std::vector<fbstab_solver_out_t*>& out,
                     const std::vector<fbstab_receding_plant_t>& plants, int steps, bool retire,
                     const std::vector<double*>& u_log, int root, double* root_u_log,
                     unsigned long long* stats = nullptr) {
    if (counts.size() != h_.size() || data.size() != h_.size() || x.size() != h_.size() || out.size() != h_.size() ||
        plants.size() != h_.size() || u_log.size() != h_.size())
      throw std::runtime_error("In FBstabMpcSharded::RecedingSweep: one entry per device is required.");
    if (fbstab_hip_mpc_receding_sweep_sharded(g_, h_.data(), counts.data(), data.data(), x.data(), out.data(),
                                              plants.data(), steps, retire ? 1 : 0, u_log.data(), root, root_u_log,
                                              stats) != FBSTAB_HIP_OK)
      throw std::runtime_error(std::string("In FBstabMpcSharded::RecedingSweep: ") + fbstab_hip_last_error());
  }

 private:
  void Release() {
    for (fbstab_mpc_handle_t h : h_) fbstab_hip_mpc_destroy(h);
    h_.clear();
    fbstab_hip_shard_group_destroy(g_);
    g_ = nullptr;
  }
  fbstab_shard_group_t g_ = nullptr;
  std::vector<fbstab_mpc_handle_t> h_;
};

}  // namespace fbstab

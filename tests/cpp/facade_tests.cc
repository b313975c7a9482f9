// End-to-end tests of the C++ facade (include/fbstab/), written to read like
// the reference's own gtest files:
//   fbstab/test/fbstab_dense_unit_tests.cc  (FeasibleQP, FeasibleQPwithEQ,
//       DegenerateQP via ProblemDataRef/VariableRef, InfeasibleQP, UnboundedQP)
//   fbstab/test/fbstab_mpc_unit_tests.cc    (DoubleIntegrator + quadprog golden,
//       DoubleIntegratorLongHorizon via ProblemDataRef, ServoMotor)
// plus the error behaviour of the constructors / Solve.  C++11, no gtest
// (absent from the image): a failed expectation prints and counts.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

// device memory for the zero-copy test: the HIP runtime's C API, no device code here
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include "fbstab/fbstab_dense.h"
#include "fbstab/fbstab_mpc.h"

using namespace fbstab;

static int g_fail = 0;
#define EXPECT_TRUE(c) do { if (!(c)) { printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); g_fail++; } } while (0)
#define EXPECT_NEAR(a, b, tol) EXPECT_TRUE(std::fabs((a) - (b)) <= (tol))
#define EXPECT_THROW(stmt) do { bool t_ = false; try { stmt; } catch (const std::runtime_error&) { t_ = true; } EXPECT_TRUE(t_); } while (0)

static FBstabDense::Options DenseOpts() {
  FBstabDense::Options o = FBstabDense::DefaultOptions();
  o.abs_tol = 1e-8;
  o.display_level = Display::OFF;
  return o;
}

static void FeasibleQP() {
  const int n = 2, m = 0, q = 2;
  FBstabDense::Variable x0(n, m, q);
  FBstabDense::ProblemData data(n, m, q);
  data.H = {3, 1, 1, 1};
  data.f = {10, 5};
  data.A = {-1, 0, 0, 1};
  data.b = {0, 0};
  FBstabDense solver(n, m, q);
  solver.UpdateOptions(DenseOpts());
  SolverOut out = solver.Solve(data, &x0);
  EXPECT_TRUE(out.eflag == ExitFlag::SUCCESS);
  const double zopt[2] = {0, -5}, vopt[2] = {5, 0};
  for (int i = 0; i < n; i++) EXPECT_NEAR(x0.z(i), zopt[i], 1e-8);
  for (int i = 0; i < q; i++) EXPECT_NEAR(x0.v(i), vopt[i], 1e-8);
}

static void FeasibleQPwithEQ() {
  const int n = 2, m = 1, q = 2;
  FBstabDense::Variable x0(n, m, q);
  FBstabDense::ProblemData data(n, m, q);
  data.H = {4, 1, 1, 2};
  data.f = {1, 1};
  data.G = {1, 1};
  data.h = {1};
  data.A = {-1, 0, 0, -1};
  data.b = {0, 0};
  FBstabDense solver(n, m, q);
  solver.UpdateOptions(DenseOpts());
  SolverOut out = solver.Solve(data, &x0);
  EXPECT_TRUE(out.eflag == ExitFlag::SUCCESS);
  EXPECT_NEAR(x0.z(0), 0.25, 1e-8);
  EXPECT_NEAR(x0.z(1), 0.75, 1e-8);
}

// Not in the reference: the opt-in elimination orders give the same solution
// (FBstabDense::SetFactorisationOrder -> fbstab_hip_dense_set_factorisation).
static void FactorisationOrders() {
  const int n = 2, m = 1, q = 2;
  FBstabDense::ProblemData data(n, m, q);
  data.H = {4, 1, 1, 2};
  data.f = {1, 1};
  data.G = {1, 1};
  data.h = {1};
  data.A = {-1, 0, 0, -1};
  data.b = {0, 0};
  const FBstabDense::FactorisationOrder orders[3] = {FBstabDense::FactorisationOrder::PIVOTED,
                                                     FBstabDense::FactorisationOrder::AUTO,
                                                     FBstabDense::FactorisationOrder::NATURAL};
  for (int k = 0; k < 3; k++) {
    FBstabDense::Variable x0(n, m, q);
    FBstabDense solver(n, m, q);
    solver.UpdateOptions(DenseOpts());
    solver.SetFactorisationOrder(orders[k]);
    SolverOut out = solver.Solve(data, &x0);
    EXPECT_TRUE(out.eflag == ExitFlag::SUCCESS);
    EXPECT_NEAR(x0.z(0), 0.25, 1e-8);
    EXPECT_NEAR(x0.z(1), 0.75, 1e-8);
  }
  FBstabDense solver(n, m, q);
  EXPECT_THROW(solver.SetFactorisationOrder(static_cast<FBstabDense::FactorisationOrder>(7)));
}

static void DegenerateQP() {
  const int n = 2, m = 0, q = 5;
  std::unique_ptr<double[]> zmem(new double[n]), lmem(new double[1]), vmem(new double[q]), ymem(new double[q]);
  FBstabDense::VecRef z(zmem.get(), n), l(lmem.get(), m), v(vmem.get(), q), y(ymem.get(), q);
  FBstabDense::VariableRef x0(&z, &l, &v, &y);
  x0.fill(0.0);
  // column-major images of H = [1 0; 0 0], A = [0 0; 1 0; 0 1; -1 0; 0 -1]
  double Hmem[4] = {1, 0, 0, 0}, fmem[2] = {1, 0}, Gmem[1] = {0}, hmem[1] = {0};
  double Amem[10] = {0, 1, 0, -1, 0, 0, 0, 1, 0, -1}, bmem[5] = {0, 3, 3, -1, -1};
  FBstabDense::MatRef H(Hmem, n, n), G(Gmem, m, n), A(Amem, q, n);
  FBstabDense::VecRef f(fmem, n), h(hmem, m), b(bmem, q);
  FBstabDense::ProblemDataRef data(&H, &f, &G, &h, &A, &b);
  FBstabDense solver(n, m, q);
  solver.UpdateOptions(DenseOpts());
  SolverOut out = solver.Solve(data, &x0);
  EXPECT_TRUE(out.eflag == ExitFlag::SUCCESS);
  EXPECT_NEAR(x0.z(0), 1, 1e-8);
  EXPECT_TRUE(x0.z(1) >= 1 && x0.z(1) <= 3);
  double r1 = 0, r2 = 0;
  for (int i = 0; i < n; i++) {
    double s = fmem[i];
    for (int j = 0; j < n; j++) s += H(i, j) * x0.z(j);
    for (int k = 0; k < q; k++) s += A(k, i) * x0.v(k);
    r1 += s * s;
  }
  for (int k = 0; k < q; k++) { const double mn = std::fmin(x0.y(k), x0.v(k)); r2 += mn * mn; }
  EXPECT_NEAR(std::sqrt(r1) + std::sqrt(r2), 0, 1e-6);
}

static void InfeasibleAndUnboundedQP() {
  {
    FBstabDense::ProblemData data(2, 0, 5);
    data.H = {1, 0, 0, 0};
    data.f = {1, -1};
    data.A = {1, 1, 1, 0, 0, 1, -1, 0, 0, -1};
    data.b = {0, 3, 3, -1, -1};
    FBstabDense::Variable x0(2, 0, 5);
    FBstabDense solver(2, 0, 5);
    solver.UpdateOptions(DenseOpts());
    EXPECT_TRUE(solver.Solve(data, &x0).eflag == ExitFlag::PRIMAL_INFEASIBLE);
  }
  {
    FBstabDense::ProblemData data(2, 0, 4);
    data.H = {1, 0, 0, 0};
    data.f = {1, -1};
    data.A = {0, 0, 1, 0, -1, 0, 0, -1};
    data.b = {0, 3, -1, -1};
    FBstabDense::Variable x0(2, 0, 4);
    FBstabDense solver(2, 0, 4);
    solver.UpdateOptions(DenseOpts());
    EXPECT_TRUE(solver.Solve(data, &x0).eflag == ExitFlag::DUAL_INFEASIBLE);
  }
}

// The double-integrator and servo-motor problems of
// fbstab/test/ocp_generator.cc:253-371 (CopyOverHorizon zeroes E(0), :403-408).
struct Ocp {
  FBstabMpc::ProblemData data;
  int N, nx, nu, nc;
  void Fill(const MatrixXd& Q, const MatrixXd& R, const MatrixXd& S, const VectorXd& q, const VectorXd& r,
            const MatrixXd& A, const MatrixXd& B, const VectorXd& c, const MatrixXd& E, const MatrixXd& L,
            const VectorXd& d, const VectorXd& x0, int N_) {
    N = N_; nx = Q.rows(); nu = R.rows(); nc = E.rows();
    data.Q = MatrixSequence(N + 1, nx, nx); data.R = MatrixSequence(N + 1, nu, nu);
    data.S = MatrixSequence(N + 1, nu, nx); data.q = MatrixSequence(N + 1, nx);
    data.r = MatrixSequence(N + 1, nu); data.A = MatrixSequence(N, nx, nx);
    data.B = MatrixSequence(N, nx, nu); data.c = MatrixSequence(N, nx);
    data.E = MatrixSequence(N + 1, nc, nx); data.L = MatrixSequence(N + 1, nc, nu);
    data.d = MatrixSequence(N + 1, nc); data.x0 = x0;
    MatrixXd E0(nc, nx);
    struct Col { const VectorXd& v; int rows() const { return v.size(); } int cols() const { return 1; }
                 double operator()(int i, int) const { return v(i); } };
    for (int i = 0; i < N + 1; i++) {
      data.Q(i) = Q; data.R(i) = R; data.S(i) = S; data.q(i) = Col{q}; data.r(i) = Col{r};
      if (i == 0) data.E(i) = E0; else data.E(i) = E;
      data.L(i) = L; data.d(i) = Col{d};
    }
    for (int i = 0; i < N; i++) { data.A(i) = A; data.B(i) = B; data.c(i) = Col{c}; }
  }
  void DoubleIntegrator(int N_) {
    MatrixXd Q(2, 2), R(1, 1), S(1, 2), A(2, 2), B(2, 1), E(6, 2), L(6, 1);
    VectorXd q(2), r(1), c(2), d(6), x0(2);
    Q = {2, 0, 0, 1}; S = {1, 0}; R = {3}; q = {-2, 0}; r = {0};
    A = {1, 1, 0, 1}; B = {0, 1}; c = {0, 0};
    E = {-1, 0, 0, -1, 1, 0, 0, 1, 0, 0, 0, 0}; L = {0, 0, 0, 0, -1, 1}; d = {0, 0, -2, -2, -1, -1};
    x0 = {0, 0};
    Fill(Q, R, S, q, r, A, B, c, E, L, d, x0, N_);
  }
  void ServoMotor(int N_) {
    const double kt = 10.0, bl = 25.0, Jm = 0.5, bm = 0.1, ktheta = 1280.2, RR = 20.0, rho = 20.0;
    const double Jl = 20 * Jm, umax = 220.0, ymax = 78.5358, ts = 0.05;
    MatrixXd Ac(4, 4), A(4, 4), B(4, 1), C(2, 4), Q(4, 4), R(1, 1), S(1, 4), E(4, 4), L(4, 1);
    Ac = {0, 1, 0, 0, -ktheta / Jl, -bl / Jl, ktheta / (rho * Jl), 0, 0, 0, 0, 1,
          ktheta / (rho * Jm), 0, -ktheta / (rho * rho * Jm), -(bm + kt * kt / RR) / Jm};
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) A(i, j) = (i == j ? 1.0 : 0.0) + ts * Ac(i, j);
    B = {0, 0, 0, ts * kt / (RR * Jm)};
    C = {1, 0, 0, 0, ktheta, 0, -ktheta / rho, 0};
    Q(0, 0) = 1000; R(0, 0) = 1e-4;
    const double pi = 3.1415926535897;
    VectorXd q(4), r(1), c(4), d(4), x0(4);
    q = {-1000 * 30 * pi / 180, 0, 0, 0}; r = {-0.0};
    for (int j = 0; j < 4; j++) { E(0, j) = C(1, j); E(1, j) = -C(1, j); }
    L = {0, 0, 1, -1}; d = {-ymax, -ymax, -umax, -umax};
    Fill(Q, R, S, q, r, A, B, c, E, L, d, x0, N_);
  }
};

static FBstabMpc::Options MpcOpts() {
  FBstabMpc::Options o = FBstabMpc::DefaultOptions();
  o.abs_tol = 1e-8;
  o.display_level = Display::OFF;
  return o;
}

static void DoubleIntegrator() {
  Ocp ocp;
  ocp.DoubleIntegrator(2);
  FBstabMpc::Variable x(ocp.N, ocp.nx, ocp.nu, ocp.nc);
  FBstabMpc solver(ocp.N, ocp.nx, ocp.nu, ocp.nc);
  solver.UpdateOptions(MpcOpts());
  SolverOut out = solver.Solve(ocp.data, &x);
  EXPECT_TRUE(out.eflag == ExitFlag::SUCCESS);
  EXPECT_TRUE(out.residual <= 1e-6);
  // MATLAB quadprog solution, fbstab_mpc_unit_tests.cc:37-47
  const double zopt[9] = {-5.31028204670497e-14, 5.02854354118183e-13, 0.311688311338095,
                          5.35637944798588e-13, 0.311688311339015, -0.0779220779990502,
                          0.311688311339667, 0.233766233340057, -0.103896103779874};
  const double lopt[6] = {-5.24675324688535, -4.49350649223710, -3.55844155822323,
                          -0.935064934014372, -1.48051948022526, 0.233766233996585};
  const double vopt[18] = {1.06213597221667e-13, -1.41190425869539e-21, 0, 0, 0, 0,
                           -1.50393600622818e-21, -8.75144622575045e-10, 0, 0, 0, 0,
                           -8.75144611157041e-10, -6.56358459377444e-10, 0, 0, 0, 0};
  for (int i = 0; i < 9; i++) EXPECT_NEAR(x.z(i), zopt[i], 1e-8);
  for (int i = 0; i < 6; i++) EXPECT_NEAR(x.l(i), lopt[i], 1e-8);
  for (int i = 0; i < 18; i++) EXPECT_NEAR(x.v(i), vopt[i], 1e-8);
}

static void LongHorizonRefAndServo() {
  {
    Ocp ocp;
    ocp.DoubleIntegrator(20);
    FBstabMpc::ProblemDataRef ref(&ocp.data.Q, &ocp.data.R, &ocp.data.S, &ocp.data.q, &ocp.data.r,
                                  &ocp.data.A, &ocp.data.B, &ocp.data.c, &ocp.data.E, &ocp.data.L,
                                  &ocp.data.d, &ocp.data.x0);
    FBstabMpc::Variable x(ocp.N, ocp.nx, ocp.nu, ocp.nc);
    FBstabMpc solver(ocp.N, ocp.nx, ocp.nu, ocp.nc);
    solver.UpdateOptions(MpcOpts());
    SolverOut out = solver.Solve(ref, &x);
    EXPECT_TRUE(out.eflag == ExitFlag::SUCCESS);
    EXPECT_TRUE(out.residual <= 1e-6);
    EXPECT_TRUE(out.newton_iters == 9 && out.prox_iters == 4);  // oracle / reference loop
  }
  {
    Ocp ocp;
    ocp.ServoMotor(25);
    FBstabMpc::Variable x(ocp.N, ocp.nx, ocp.nu, ocp.nc);
    FBstabMpc solver(ocp.N, ocp.nx, ocp.nu, ocp.nc);
    solver.UpdateOptions(MpcOpts());
    SolverOut out = solver.Solve(ocp.data, &x);
    EXPECT_TRUE(out.eflag == ExitFlag::SUCCESS);
    EXPECT_TRUE(out.residual <= 1e-6);
    EXPECT_TRUE(out.newton_iters == 27 && out.prox_iters == 4);
  }
}


// Stand-ins for Eigen::Map<Eigen::VectorXd> / Eigen::Map<Eigen::MatrixXd> (Eigen is
// not in this image): the accessors a user's Map offers, nothing of the facade's.
struct MockVecMap {
  MockVecMap(double* p, long n) : p_(p), n_(n) {}
  double* data() const { return p_; }
  long size() const { return n_; }
  double* p_;
  long n_;
};
struct MockMatMap {
  MockMatMap(double* p, long r, long c) : p_(p), r_(r), c_(c) {}
  double* data() const { return p_; }
  long rows() const { return r_; }
  long cols() const { return c_; }
  long size() const { return r_ * c_; }
  double* p_;
  long r_, c_;
};

// The reference's Ref constructors take Eigen::Map arguments (fbstab_dense.h:69-74,
// :97-100 by pointer; fbstab_mpc.h:139-150 by value): any type with the same
// accessors binds, and the 4-vector forms of the MPC constructors exist
// (fbstab_mpc.h:130, :168).
static void MapTypedRefsAndVector4() {
  {  // DegenerateQP once more, through Map-like user types
    const int n = 2, m = 0, q = 5;
    double zm[2] = {0, 0}, lm[1] = {0}, vm[5] = {0, 0, 0, 0, 0}, ym[5] = {0, 0, 0, 0, 0};
    MockVecMap z(zm, n), l(lm, m), v(vm, q), y(ym, q);
    FBstabDense::VariableRef x0(&z, &l, &v, &y);
    double Hmem[4] = {1, 0, 0, 0}, fmem[2] = {1, 0}, Gmem[1] = {0}, hmem[1] = {0};
    double Amem[10] = {0, 1, 0, -1, 0, 0, 0, 1, 0, -1}, bmem[5] = {0, 3, 3, -1, -1};
    MockMatMap H(Hmem, n, n), G(Gmem, m, n), A(Amem, q, n);
    MockVecMap f(fmem, n), h(hmem, m), b(bmem, q);
    FBstabDense::ProblemDataRef data(&H, &f, &G, &h, &A, &b);
    FBstabDense solver(n, m, q);
    solver.UpdateOptions(DenseOpts());
    SolverOut out = solver.Solve(data, &x0);
    EXPECT_TRUE(out.eflag == ExitFlag::SUCCESS);
    EXPECT_NEAR(zm[0], 1, 1e-8);
    EXPECT_TRUE(zm[1] >= 1 && zm[1] <= 3);
  }
  {  // DoubleIntegrator(2) with VariableRef over Map-like views and sizes as a 4-vector
    Ocp ocp;
    ocp.DoubleIntegrator(2);
    const Vector4d s(ocp.N, ocp.nx, ocp.nu, ocp.nc);
    FBstabMpc::Variable sized(s);
    EXPECT_TRUE(sized.z.size() == 9 && sized.l.size() == 6 && sized.v.size() == 18 && sized.y.size() == 18);
    std::vector<double> zm(9, 0.0), lm(6, 0.0), vm(18, 0.0), ym(18, 0.0);
    FBstabMpc::VariableRef x(MockVecMap(zm.data(), 9), MockVecMap(lm.data(), 6), MockVecMap(vm.data(), 18),
                             MockVecMap(ym.data(), 18));
    FBstabMpc solver(s);
    solver.UpdateOptions(MpcOpts());
    FBstabMpc::ProblemDataRef ref(&ocp.data.Q, &ocp.data.R, &ocp.data.S, &ocp.data.q, &ocp.data.r, &ocp.data.A,
                                  &ocp.data.B, &ocp.data.c, &ocp.data.E, &ocp.data.L, &ocp.data.d, &ocp.data.x0);
    SolverOut out = solver.Solve(ref, &x);
    EXPECT_TRUE(out.eflag == ExitFlag::SUCCESS);
    EXPECT_NEAR(zm[2], 0.311688311338095, 1e-8);
    EXPECT_NEAR(lm[0], -5.24675324688535, 1e-8);
  }
}

// A stream of the user's own (tools/output_stream.h:15-37) collecting the display.
class StringOutput : public OutputStream<StringOutput> {
 public:
  explicit StringOutput(std::string* s) : s_(s) {}

 protected:
  void PrintImplementation(const char* message) const { s_->append(message); }
  friend class OutputStream<StringOutput>;

 private:
  std::string* s_;
};

// ProblemDataRef / VariableRef over DEVICE memory: Solve passes the pointers on
// as they are (FBstabMpc::SetMemory, FBSTAB_HIP_DEVICE_POINTERS | OUT_ON_HOST).
struct DeviceArray {
  explicit DeviceArray(const std::vector<double>& h) : n(h.size()) {
    if (hipMalloc(reinterpret_cast<void**>(&p), sizeof(double) * (n ? n : 1)) != hipSuccess) p = nullptr;
    if (p && n) (void)hipMemcpy(p, h.data(), sizeof(double) * n, hipMemcpyHostToDevice);
  }
  DeviceArray(const double* h, size_t n_) : DeviceArray(std::vector<double>(h, h + n_)) {}
  ~DeviceArray() { if (p) (void)hipFree(p); }
  void upload(const std::vector<double>& h) {
    if (p && n) (void)hipMemcpy(p, h.data(), sizeof(double) * n, hipMemcpyHostToDevice);
  }
  std::vector<double> host() const {
    std::vector<double> h(n);
    if (n) (void)hipMemcpy(h.data(), p, sizeof(double) * n, hipMemcpyDeviceToHost);
    return h;
  }
  double* p = nullptr;
  size_t n;
};

static void DeviceResidentRefs() {
  Ocp ocp;
  ocp.DoubleIntegrator(20);
  const FBstabMpc::ProblemData& d = ocp.data;
  DeviceArray Q(d.Q.data(), d.Q.size()), R(d.R.data(), d.R.size()), S(d.S.data(), d.S.size()),
      q(d.q.data(), d.q.size()), r(d.r.data(), d.r.size()), A(d.A.data(), d.A.size()), B(d.B.data(), d.B.size()),
      c(d.c.data(), d.c.size()), E(d.E.data(), d.E.size()), L(d.L.data(), d.L.size()), dd(d.d.data(), d.d.size()),
      x0(d.x0.data(), d.x0.size());
  const int N = ocp.N, nx = ocp.nx, nu = ocp.nu, nc = ocp.nc;
  FBstabMpc::ProblemDataRef ref;
  ref.Q = MapMatrixSequence(Q.p, N + 1, nx, nx); ref.R = MapMatrixSequence(R.p, N + 1, nu, nu);
  ref.S = MapMatrixSequence(S.p, N + 1, nu, nx); ref.q = MapMatrixSequence(q.p, N + 1, nx, 1);
  ref.r = MapMatrixSequence(r.p, N + 1, nu, 1); ref.A = MapMatrixSequence(A.p, N, nx, nx);
  ref.B = MapMatrixSequence(B.p, N, nx, nu); ref.c = MapMatrixSequence(c.p, N, nx, 1);
  ref.E = MapMatrixSequence(E.p, N + 1, nc, nx); ref.L = MapMatrixSequence(L.p, N + 1, nc, nu);
  ref.d = MapMatrixSequence(dd.p, N + 1, nc, 1);
  ref.SetX0(MockVecMap(x0.p, nx));
  const int nz = (N + 1) * (nx + nu), nl = (N + 1) * nx, nv = (N + 1) * nc;
  DeviceArray z(std::vector<double>(nz, 0.0)), l(std::vector<double>(nl, 0.0)), v(std::vector<double>(nv, 0.0)),
      y(std::vector<double>(nv, 0.0));
  FBstabMpc::VariableRef x(MockVecMap(z.p, nz), MockVecMap(l.p, nl), MockVecMap(v.p, nv), MockVecMap(y.p, nv));
  FBstabMpc solver(N, nx, nu, nc);
  solver.UpdateOptions(MpcOpts());
  solver.SetMemory(FBstabMpc::Memory::DEVICE);
  SolverOut out = solver.Solve(ref, &x);
  EXPECT_TRUE(out.eflag == ExitFlag::SUCCESS);
  EXPECT_TRUE(out.residual <= 1e-6);
  EXPECT_TRUE(out.newton_iters == 9 && out.prox_iters == 4);  // as from host memory (LongHorizonRefAndServo)
  // the solution is in the caller's device buffers; the host path gives the same numbers
  FBstabMpc::Variable xh(N, nx, nu, nc);
  FBstabMpc host_solver(N, nx, nu, nc);
  host_solver.UpdateOptions(MpcOpts());
  host_solver.Solve(ocp.data, &xh);
  const std::vector<double> zd = z.host(), vd = v.host();
  for (int i = 0; i < nz; i++) EXPECT_TRUE(zd[i] == xh.z(i));
  for (int i = 0; i < nv; i++) EXPECT_TRUE(vd[i] == xh.v(i));
  // Display::FINAL, the reference's default level, runs the same kernel on the device
  // buffers (summary block from fbstab_hip_mpc_solve_batch_final); the iteration
  // displays need host memory
  FBstabMpc::Options o = MpcOpts();
  o.display_level = Display::FINAL;
  solver.UpdateOptions(o);
  z.upload(std::vector<double>(nz, 0.0)); l.upload(std::vector<double>(nl, 0.0)); v.upload(std::vector<double>(nv, 0.0));
  std::string text;
  StringOutput os(&text);
  SolverOut outf = solver.Solve(ref, &x, os);
  EXPECT_TRUE(outf.eflag == ExitFlag::SUCCESS && outf.newton_iters == 9 && outf.prox_iters == 4);
  EXPECT_TRUE(text.find("Exit code: Success") != std::string::npos && text.find("|rz|") != std::string::npos);
  const std::vector<double> zf = z.host();
  for (int i = 0; i < nz; i++) EXPECT_TRUE(zf[i] == xh.z(i));
  o.display_level = Display::ITER;
  solver.UpdateOptions(o);
  EXPECT_THROW(solver.Solve(ref, &x));
}

// FBstabMpcSharded with a group of ONE device (what a test box has): the shard's
// solution is gathered into separate root buffers and equals the host-memory solve.
static void ShardedOnOneDevice() {
  Ocp ocp;
  ocp.DoubleIntegrator(20);
  const FBstabMpc::ProblemData& d = ocp.data;
  DeviceArray Q(d.Q.data(), d.Q.size()), R(d.R.data(), d.R.size()), S(d.S.data(), d.S.size()),
      q(d.q.data(), d.q.size()), r(d.r.data(), d.r.size()), A(d.A.data(), d.A.size()), B(d.B.data(), d.B.size()),
      c(d.c.data(), d.c.size()), E(d.E.data(), d.E.size()), L(d.L.data(), d.L.size()), dd(d.d.data(), d.d.size()),
      x0(d.x0.data(), d.x0.size());
  const int N = ocp.N, nx = ocp.nx, nu = ocp.nu, nc = ocp.nc;
  const int nz = (N + 1) * (nx + nu), nl = (N + 1) * nx, nv = (N + 1) * nc;
  DeviceArray z(std::vector<double>(nz, 0.0)), l(std::vector<double>(nl, 0.0)), v(std::vector<double>(nv, 0.0)),
      y(std::vector<double>(nv, 0.0));
  DeviceArray rz(std::vector<double>(nz, 9.0)), rl(std::vector<double>(nl, 9.0)), rv(std::vector<double>(nv, 9.0)),
      ry(std::vector<double>(nv, 9.0));
  DeviceArray outs(std::vector<double>(5, 0.0)), routs(std::vector<double>(5, 0.0));  // 40-byte records
  fbstab_mpc_batch_t b;
  const DeviceArray* arr[FBSTAB_MPC_NSEQ] = {&Q, &R, &S, &q, &r, &A, &B, &c, &E, &L, &dd, &x0};
  for (int i = 0; i < FBSTAB_MPC_NSEQ; i++) { b.base[i] = arr[i]->p; b.stride[i] = static_cast<long long>(arr[i]->n); }
  fbstab_var_batch_t xv, rxv;
  double* xp[4] = {z.p, l.p, v.p, y.p};
  double* rp[4] = {rz.p, rl.p, rv.p, ry.p};
  const long long len[4] = {nz, nl, nv, nv};
  for (int i = 0; i < 4; i++) { xv.base[i] = xp[i]; xv.stride[i] = len[i]; rxv.base[i] = rp[i]; rxv.stride[i] = len[i]; }
  FBstabMpcSharded solver(N, nx, nu, nc, std::vector<int>(1, 0), 1);
  EXPECT_TRUE(solver.devices() == 1);
  solver.UpdateOptions(MpcOpts());
  solver.Solve(std::vector<int>(1, 1), std::vector<fbstab_mpc_batch_t>(1, b), std::vector<fbstab_var_batch_t>(1, xv),
               std::vector<fbstab_solver_out_t*>(1, reinterpret_cast<fbstab_solver_out_t*>(outs.p)), 0, rxv,
               reinterpret_cast<fbstab_solver_out_t*>(routs.p));
  FBstabMpc::Variable xh(N, nx, nu, nc);
  FBstabMpc host_solver(N, nx, nu, nc);
  host_solver.UpdateOptions(MpcOpts());
  SolverOut oh = host_solver.Solve(ocp.data, &xh);
  const std::vector<double> zr = rz.host(), vr = rv.host(), orec = routs.host();
  for (int i = 0; i < nz; i++) EXPECT_TRUE(zr[i] == xh.z(i));
  for (int i = 0; i < nv; i++) EXPECT_TRUE(vr[i] == xh.v(i));
  fbstab_solver_out_t o;
  std::memcpy(&o, orec.data(), sizeof(o));
  EXPECT_TRUE(o.eflag == FBSTAB_SUCCESS && o.newton_iters == oh.newton_iters && o.prox_iters == oh.prox_iters);
  EXPECT_THROW(solver.Solve(std::vector<int>(2, 1), std::vector<fbstab_mpc_batch_t>(1, b),
                            std::vector<fbstab_var_batch_t>(1, xv),
                            std::vector<fbstab_solver_out_t*>(1, reinterpret_cast<fbstab_solver_out_t*>(outs.p)), 0, rxv,
                            reinterpret_cast<fbstab_solver_out_t*>(routs.p)));
}

static void ErrorBehaviour() {
  EXPECT_THROW(FBstabMpc(0, 2, 1, 6));     // fbstab_mpc.cc:62-65
  EXPECT_THROW(FBstabDense(2, -1, 2));     // fbstab_dense.cc:19-23
  EXPECT_THROW(MatrixSequence(-1, 2, 2));  // matrix_sequence.h:31-33
  EXPECT_THROW(MapMatrixSequence(nullptr, 1, 2, 2));
  Ocp ocp;
  ocp.DoubleIntegrator(2);
  FBstabMpc solver(3, ocp.nx, ocp.nu, ocp.nc);  // horizon mismatch
  FBstabMpc::Variable x(3, ocp.nx, ocp.nu, ocp.nc);
  EXPECT_THROW(solver.Solve(ocp.data, &x));     // fbstab_mpc.h:229-236
  FBstabMpc solver2(2, ocp.nx, ocp.nu, ocp.nc);
  EXPECT_THROW(solver2.Solve(ocp.data, &x));    // fbstab_mpc.h:237-241 (guess size)
  FBstabMpc::Options o = FBstabMpc::DefaultOptions();
  EXPECT_TRUE(o.sigma_max == 1e-6 && o.beta == 0.75 && o.max_newton_iters == 200);  // impl:33-59
  AlgorithmParameters raw;
  EXPECT_TRUE(raw.beta == 0.7 && raw.max_newton_iters == 500);  // header initialisers differ
}

// `facade_tests display`: FeasibleQP and DoubleIntegrator(2) at Display::FINAL (the
// reference's default level), ITER and ITER_DETAILED (default options), text between
// markers; tests/test_facade.py compares it with what the reference prints
// (tests/golden/reference_display.json).
static int DisplayMode() {
  const Display levels[3] = {Display::FINAL, Display::ITER, Display::ITER_DETAILED};
  for (int k = 0; k < 3; k++) {
    {
      FBstabDense::Variable x0(2, 0, 2);
      FBstabDense::ProblemData data(2, 0, 2);
      data.H = {3, 1, 1, 1};
      data.f = {10, 5};
      data.A = {-1, 0, 0, 1};
      data.b = {0, 0};
      FBstabDense solver(2, 0, 2);
      FBstabDense::Options o = FBstabDense::DefaultOptions();
      o.display_level = levels[k];
      solver.UpdateOptions(o);
      std::string text;
      StringOutput os(&text);
      SolverOut out = solver.Solve(data, &x0, os);
      printf("===BEGIN dense FeasibleQP %d===\n%s===END===\n", static_cast<int>(levels[k]), text.c_str());
      EXPECT_TRUE(out.eflag == ExitFlag::SUCCESS);
      EXPECT_NEAR(x0.z(1), -5.0, 1e-5);
    }
    {
      Ocp ocp;
      ocp.DoubleIntegrator(2);
      FBstabMpc::Variable x(ocp.N, ocp.nx, ocp.nu, ocp.nc);
      FBstabMpc solver(ocp.N, ocp.nx, ocp.nu, ocp.nc);
      FBstabMpc::Options o = FBstabMpc::DefaultOptions();
      o.display_level = levels[k];
      solver.UpdateOptions(o);
      std::string text;
      StringOutput os(&text);
      SolverOut out = solver.Solve(ocp.data, &x, os);
      printf("===BEGIN mpc DoubleIntegrator %d===\n%s===END===\n", static_cast<int>(levels[k]), text.c_str());
      EXPECT_TRUE(out.eflag == ExitFlag::SUCCESS);
    }
  }
  {
    // an infeasibility exit at the default level: the reference's summary shows the residual
    // of x(k), not of the certificate it returns (impl:204-212) - the facade's second solve
    FBstabDense::ProblemData data(2, 0, 5);
    data.H = {1, 0, 0, 0};
    data.f = {1, -1};
    data.A = {1, 1, 1, 0, 0, 1, -1, 0, 0, -1};
    data.b = {0, 3, 3, -1, -1};
    FBstabDense::Variable x0(2, 0, 5);
    FBstabDense solver(2, 0, 5);  // default options: Display::FINAL
    std::string text;
    StringOutput os(&text);
    SolverOut out = solver.Solve(data, &x0, os);
    printf("===BEGIN dense InfeasibleQP 1===\n%s===END===\n", text.c_str());
    EXPECT_TRUE(out.eflag == ExitFlag::PRIMAL_INFEASIBLE);
  }
  return g_fail ? 1 : 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && std::string(argv[1]) == "display") return DisplayMode();
  FeasibleQP();
  FeasibleQPwithEQ();
  FactorisationOrders();
  DegenerateQP();
  InfeasibleAndUnboundedQP();
  DoubleIntegrator();
  LongHorizonRefAndServo();
  MapTypedRefsAndVector4();
  DeviceResidentRefs();
  ShardedOnOneDevice();
  ErrorBehaviour();
  printf(g_fail ? "%d FAILED\n" : "ALL FACADE TESTS PASSED\n", g_fail);
  return g_fail ? 1 : 0;
}

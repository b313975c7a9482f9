// Compile-only (g++ -fsyntax-only, tests/test_facade.py): include/fbstab/ with
// <Eigen/Dense> on the include path - here tests/cpp/eigen_mock, a stand-in with Eigen's
// types (the image has no Eigen) - so that the branch a reference user's build takes,
// `__has_include(<Eigen/Dense>)` in include/fbstab/dense_types.h, meets a compiler: the
// facade's ProblemData / Variable hold Eigen::MatrixXd / Eigen::VectorXd themselves, the
// Ref types bind Eigen::Map arguments exactly as the reference's signatures take them
// (fbstab/fbstab_dense.h:55-107, fbstab/fbstab_mpc.h:67-150, their tests
// fbstab_dense_unit_tests.cc:121-177, fbstab_mpc_unit_tests.cc:62-82).
#include <type_traits>

#include "fbstab/fbstab_dense.h"
#include "fbstab/fbstab_mpc.h"

#ifndef FBSTAB_FACADE_HAS_EIGEN
#error "the Eigen branch of include/fbstab/dense_types.h was not taken"
#endif
static_assert(std::is_same<fbstab::VectorXd, Eigen::VectorXd>::value, "facade vectors are Eigen's");
static_assert(std::is_same<fbstab::MatrixXd, Eigen::MatrixXd>::value, "facade matrices are Eigen's");
static_assert(std::is_same<decltype(fbstab::FBstabDense::ProblemData::H), Eigen::MatrixXd>::value, "ProblemData::H");
static_assert(std::is_same<decltype(fbstab::FBstabMpc::Variable::z), Eigen::VectorXd>::value, "Variable::z");

using namespace fbstab;

// fbstab_dense_unit_tests.cc:28-61 (owning Eigen types) and :121-177 (Eigen::Map views)
SolverOut DenseOwning() {
  FBstabDense::ProblemData data(2, 0, 2);
  data.H(0, 0) = 3; data.H(0, 1) = 1; data.H(1, 0) = 1; data.H(1, 1) = 1;
  data.f(0) = 10; data.f(1) = 5;
  data.A(0, 0) = -1; data.A(1, 1) = 1;
  data.b.setZero();
  FBstabDense::Variable x(2, 0, 2);
  FBstabDense solver(2, 0, 2);
  FBstabDense::Options o = FBstabDense::DefaultOptions();
  o.abs_tol = 1e-8;
  o.display_level = Display::OFF;
  solver.UpdateOptions(o);
  return solver.Solve(data, &x);
}
SolverOut DenseMaps(double* Hm, double* fm, double* Gm, double* hm, double* Am, double* bm, double* zm, double* lm,
                    double* vm, double* ym) {
  const int n = 2, m = 0, q = 5;
  Eigen::Map<Eigen::VectorXd> z(zm, n), l(lm, m), v(vm, q), y(ym, q);
  FBstabDense::VariableRef x0(&z, &l, &v, &y);
  x0.fill(0.0);
  Eigen::Map<Eigen::MatrixXd> H(Hm, n, n), G(Gm, m, n), A(Am, q, n);
  Eigen::Map<Eigen::VectorXd> f(fm, n), h(hm, m), b(bm, q);
  FBstabDense::ProblemDataRef data(&H, &f, &G, &h, &A, &b);
  FBstabDense solver(n, m, q);
  return solver.Solve(data, &x0);  // default options: Display::FINAL
}

// fbstab_mpc_unit_tests.cc:15-82: owning data, Vector4d sizes, Map-typed VariableRef and x0
SolverOut Mpc(const FBstabMpc::ProblemData& owned, double* zm, double* lm, double* vm, double* ym) {
  const Eigen::Vector4d s(2, 2, 1, 6);
  FBstabMpc::Variable x(s);
  FBstabMpc solver(s);
  solver.UpdateOptions(FBstabMpc::ReliableOptions());
  SolverOut a = solver.Solve(owned, &x);
  FBstabMpc::ProblemDataRef ref(&owned.Q, &owned.R, &owned.S, &owned.q, &owned.r, &owned.A, &owned.B, &owned.c,
                                &owned.E, &owned.L, &owned.d, &owned.x0);
  ref.SetX0(Eigen::Map<const Eigen::VectorXd>(owned.x0.data(), owned.x0.size()));
  FBstabMpc::VariableRef xr(Eigen::Map<Eigen::VectorXd>(zm, 9), Eigen::Map<Eigen::VectorXd>(lm, 6),
                            Eigen::Map<Eigen::VectorXd>(vm, 18), Eigen::Map<Eigen::VectorXd>(ym, 18));
  xr.fill(0.0);
  SolverOut b = solver.Solve(ref, &xr);
  return a.newton_iters > b.newton_iters ? a : b;
}

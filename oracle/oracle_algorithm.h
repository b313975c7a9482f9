// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_linalg.h header).
//
// Restatement of FBstabAlgorithm<>::Solve / ::SolveProximalSubproblem
// (fbstab/fbstab_algorithm-impl.h:113-304) over the oracle components.  The
// control flow, the order of residual evaluations and every quirk listed in
// SURVEY.md appendix C (stale Eo, merit FIFO cleared per subproblem, the
// t = beta^max_ls step, residual left stale on infeasibility exits, ...) are
// kept.  oracle/_ref/ holds the same components driven by the reference's OWN
// fbstab_algorithm.h template; tests assert the two agree bit for bit.
#pragma once

#include <algorithm>
#include <array>
#include <stdexcept>

#include "../include/fbstab_types.h"
#include "oracle_components.h"

namespace fbo {

// tools/utilities.h:19-28
inline double saturate(double x, double a, double b) {
  if (a > b)
    throw std::runtime_error(
        "In tools::saturate: upper bound must be larger than the lower bound");
  return std::max(std::min(x, b), a);
}

template <class LinearSolver>
class Algorithm {
 public:
  Algorithm(FullVariable* x1, FullVariable* x2, FullVariable* x3,
            FullVariable* x4, FullResidual* r1, FullResidual* r2,
            LinearSolver* lin_sol, FullFeasibility* fcheck)
      : xk_(x1), xi_(x2), xp_(x3), dx_(x4), rk_(r1), ri_(r2),
        linear_solver_(lin_sol), feasibility_(fcheck) {
    fbstab_options_default(&opts_);
  }

  // fbstab_algorithm-impl.h:307-332
  void UpdateParameters(const fbstab_options_t* o) {
    opts_ = *o;
    fbstab_options_validate(&opts_);
  }

  // Where the reference prints a display line (PrintIterLine, PrintDetailed*,
  // PrintFinal; fbstab_algorithm-impl.h:411-541) this restatement appends one
  // record of 8 doubles [kind, i0, i1, v0..v4] to *sink, whatever the display
  // level (fbstab_trace_record_t in include/fbstab_types.h names the fields).
  void SetTraceSink(std::vector<double>* sink) { trace_ = sink; }

  // fbstab_algorithm-impl.h:113-224.  z0,l0,v0 in/out; y0 out.
  template <class ProblemData>
  fbstab_solver_out_t Solve(const ProblemData& qp, Vec* z0, Vec* l0, Vec* v0,
                            Vec* y0) {
    rk_->SetAlpha(opts_.alpha);
    ri_->SetAlpha(opts_.alpha);
    linear_solver_->SetAlpha(opts_.alpha);
    xk_->LinkData(&qp);
    xi_->LinkData(&qp);
    dx_->LinkData(&qp);
    xp_->LinkData(&qp);
    rk_->LinkData(&qp);
    ri_->LinkData(&qp);
    feasibility_->LinkData(&qp);
    linear_solver_->LinkData(&qp);

    const double sigma = opts_.sigma0;
    combo_tol_ = opts_.abs_tol + opts_.rel_tol * (1.0 + qp.ForcingNorm());

    xk_->z() = *z0;
    xk_->l() = *l0;
    xk_->v() = *v0;
    xk_->InitializeConstraintMargin();
    xi_->Copy(*xk_);
    dx_->Fill(1.0);

    rk_->PenalizedNaturalResidual(*xk_);
    ri_->Fill(0.0);
    const double E0 = rk_->Norm();
    double Ek = E0;
    double inner_tol = saturate(E0, opts_.inner_tol_min, opts_.inner_tol_max);
    newton_iters_ = 0;
    prox_iters_ = 0;

    for (int k = 0; k < opts_.max_prox_iters; k++) {
      rk_->PenalizedNaturalResidual(*xk_);
      Ek = rk_->Norm();
      if (Ek <= combo_tol_ || dx_->Norm() <= opts_.stall_tol) {
        EmitIterLine(inner_tol);  // impl:165
        fbstab_solver_out_t out = Output(FBSTAB_SUCCESS, E0);
        Write(*xk_, z0, l0, v0, y0);
        return out;
      }
      Emit(FBSTAB_TRACE_DETAILED_HEADER, prox_iters_, newton_iters_, rk_->Norm());  // impl:171
      EmitIterLine(inner_tol);                                                     // impl:172
      inner_tol = saturate(inner_tol * opts_.delta, opts_.inner_tol_min, Ek);
      xi_->Copy(*xk_);
      const double Eo = SolveProximalSubproblem(xi_, xk_, inner_tol, sigma, Ek);
      if (newton_iters_ >= opts_.max_newton_iters) {
        if (Eo < Ek) {
          Write(*xi_, z0, l0, v0, y0);
          rk_->PenalizedNaturalResidual(*xi_);
        } else {
          Write(*xk_, z0, l0, v0, y0);
          rk_->PenalizedNaturalResidual(*xk_);
        }
        return Output(FBSTAB_MAXITERATIONS, E0);
      }
      dx_->Copy(*xi_);
      dx_->axpy(-1.0, *xk_);
      if (opts_.check_feasibility) {
        const int eflag = CheckForInfeasibility(*dx_);
        if (eflag != FBSTAB_SUCCESS) {
          fbstab_solver_out_t out = Output(eflag, E0);
          Write(*dx_, z0, l0, v0, y0);
          return out;
        }
      }
      xk_->Copy(*xi_);
      prox_iters_++;
    }
    fbstab_solver_out_t out = Output(FBSTAB_MAXITERATIONS, E0);
    Write(*xk_, z0, l0, v0, y0);
    return out;
  }

 private:
  // fbstab_algorithm-impl.h:229-304
  double SolveProximalSubproblem(FullVariable* x, FullVariable* xbar,
                                 double tol, double sigma,
                                 double current_outer_residual) {
    merit_buffer_.fill(0.0);
    double Eo = 0;
    double t = 1.0;
    for (int i = 0; i < opts_.max_inner_iters; i++) {
      ri_->InnerResidual(*x, *xbar, sigma);
      const double Ei = ri_->Norm();
      rk_->PenalizedNaturalResidual(*x);
      Eo = rk_->Norm();
      // impl:250-257
      Emit(FBSTAB_TRACE_DETAILED_LINE, i, 0, t, ri_->z_norm(), ri_->l_norm(), ri_->v_norm());
      if ((Ei <= tol && Eo < current_outer_residual) ||
          (Ei <= opts_.inner_tol_min)) {
        Emit(FBSTAB_TRACE_DETAILED_FOOTER, 0, 0, ri_->Norm(), tol);
        break;
      }
      if (newton_iters_ >= opts_.max_newton_iters) break;
      if (!linear_solver_->Initialize(*x, *xbar, sigma))
        throw std::runtime_error(
            "In FBstabAlgorithm::Solve: LinearSolver::Initialize failed.");
      ri_->Negate();
      if (!linear_solver_->Solve(*ri_, dx_))
        throw std::runtime_error(
            "In FBstabAlgorithm::Solve: LinearSolver::Solve failed.");
      newton_iters_++;
      const double current_merit = ri_->Merit();
      InsertMerit(current_merit);
      const double m0 = opts_.nonmonotone_linesearch
                            ? *std::max_element(merit_buffer_.begin(),
                                                merit_buffer_.end())
                            : current_merit;
      t = 1.0;
      for (int j = 0; j < opts_.max_linesearch_iters; j++) {
        xp_->Copy(*x);
        xp_->axpy(t, *dx_);
        ri_->InnerResidual(*xp_, *xbar, sigma);
        const double mp = ri_->Merit();
        if (mp <= m0 - 2.0 * t * opts_.eta * current_merit)
          break;
        else
          t *= opts_.beta;
#ifdef FBO_LINESEARCH_OBSERVER  // studies only (tools/cpp/ldlt_order_observer.h); never in liboracle.so
        FBO_LINESEARCH_OBSERVER(0);
#endif
      }
#ifdef FBO_LINESEARCH_OBSERVER
      FBO_LINESEARCH_OBSERVER(1);
#endif
      x->axpy(t, *dx_);
    }
    x->ProjectDuals();
    return Eo;
  }

  // fbstab_algorithm-impl.h:385-400
  int CheckForInfeasibility(const FullVariable& x) {
    const FullFeasibility::FeasibilityStatus f =
        feasibility_->CheckFeasibility(x, opts_.infeas_tol);
    if (f == FullFeasibility::FeasibilityStatus::FEASIBLE) return FBSTAB_SUCCESS;
    if (f == FullFeasibility::FeasibilityStatus::PRIMAL_INFEASIBLE)
      return FBSTAB_PRIMAL_INFEASIBLE;
    if (f == FullFeasibility::FeasibilityStatus::DUAL_INFEASIBLE)
      return FBSTAB_DUAL_INFEASIBLE;
    return FBSTAB_PRIMAL_DUAL_INFEASIBLE;
  }
  // fbstab_algorithm-impl.h:402-409
  void InsertMerit(double x) {
    for (size_t i = merit_buffer_.size() - 1; i > 0; i--)
      merit_buffer_[i] = merit_buffer_[i - 1];
    merit_buffer_[0] = x;
  }
  // fbstab_algorithm-impl.h:362-383 (timing filled by the caller)
  fbstab_solver_out_t Output(int eflag, double E0) const {
    // PrintFinal (impl:381, :493-541)
    Emit(FBSTAB_TRACE_FINAL, eflag, 0, rk_->z_norm(), rk_->l_norm(), rk_->v_norm(), combo_tol_);
    fbstab_solver_out_t o;
    o.eflag = eflag;
    o.pad_ = 0;
    o.residual = rk_->Norm();
    o.newton_iters = newton_iters_;
    o.prox_iters = prox_iters_;
    o.solve_time = -1.0;
    o.initial_residual = E0;
    return o;
  }
  void Emit(int kind, int i0, int i1, double v0 = 0, double v1 = 0, double v2 = 0, double v3 = 0,
            double v4 = 0) const {
    if (!trace_) return;
    const double r[8] = {double(kind), double(i0), double(i1), v0, v1, v2, v3, v4};
    trace_->insert(trace_->end(), r, r + 8);
  }
  // PrintIterLine (impl:411-426): ri is whatever the inner loop last left there
  void EmitIterLine(double inner_tol) const {
    Emit(FBSTAB_TRACE_ITER_LINE, prox_iters_, newton_iters_, rk_->z_norm(), rk_->l_norm(),
         rk_->v_norm(), ri_->Norm(), inner_tol);
  }
  static void Write(const FullVariable& x, Vec* z, Vec* l, Vec* v, Vec* y) {
    *z = x.z();
    *l = x.l();
    *v = x.v();
    *y = x.y();
  }

  std::vector<double>* trace_ = nullptr;
  double combo_tol_ = 0.0;
  int newton_iters_ = 0;
  int prox_iters_ = 0;
  FullVariable *xk_, *xi_, *xp_, *dx_;
  FullResidual *rk_, *ri_;
  LinearSolver* linear_solver_;
  FullFeasibility* feasibility_;
  fbstab_options_t opts_;
  std::array<double, 5> merit_buffer_ = {{0.0, 0.0, 0.0, 0.0, 0.0}};
};

}  // namespace fbo

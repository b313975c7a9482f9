// ExitFlag, SolverOut, Display and AlgorithmParameters with the reference's
// names, members and defaults (fbstab/fbstab_algorithm.h:17-82,
// fbstab_algorithm-impl.h:7-74), bridging to the plain-C types of the C-ABI.
#pragma once

#include <cstdio>
#include <stdexcept>
#include <vector>
#include <string>

#include "../fbstab_types.h"

namespace fbstab {

enum class ExitFlag {
  SUCCESS = 0,
  DIVERGENCE = 1,
  MAXITERATIONS = 2,
  PRIMAL_INFEASIBLE = 3,
  DUAL_INFEASIBLE = 4,
  PRIMAL_DUAL_INFEASIBLE = 5
};

struct SolverOut {
  ExitFlag eflag = ExitFlag::MAXITERATIONS;
  double residual = 0.0;
  int newton_iters = 0;
  int prox_iters = 0;
  double solve_time = 0.0;  // seconds
  double initial_residual = 0.0;
};

enum class Display { OFF = 0, FINAL = 1, ITER = 2, ITER_DETAILED = 3 };

// In-class initialisers are the reference header's (fbstab_algorithm.h:49-74);
// the effective defaults are DefaultParameters() (impl:33-59), which every
// solver constructor applies, exactly as in the reference.
struct AlgorithmParameters {
  double sigma0 = 1e-8;
  double sigma_max = 1e-8;
  double sigma_min = 1e-10;
  double alpha = 0.95;
  double beta = 0.7;
  double eta = 1e-8;
  double delta = 1.0 / 5.0;
  double gamma = 1.0 / 10.0;
  double abs_tol = 1e-6;
  double rel_tol = 1e-12;
  double stall_tol = 1e-10;
  double infeas_tol = 1e-8;
  double inner_tol_max = 1e-1;
  double inner_tol_min = 1e-12;
  int max_newton_iters = 500;
  int max_prox_iters = 100;
  int max_inner_iters = 100;
  int max_linesearch_iters = 20;
  bool check_feasibility = true;
  bool nonmonotone_linesearch = true;
  Display display_level = Display::FINAL;

  fbstab_options_t ToC() const {
    fbstab_options_t o;
    o.sigma0 = sigma0; o.sigma_max = sigma_max; o.sigma_min = sigma_min; o.alpha = alpha;
    o.beta = beta; o.eta = eta; o.delta = delta; o.gamma = gamma; o.abs_tol = abs_tol;
    o.rel_tol = rel_tol; o.stall_tol = stall_tol; o.infeas_tol = infeas_tol;
    o.inner_tol_max = inner_tol_max; o.inner_tol_min = inner_tol_min;
    o.max_newton_iters = max_newton_iters; o.max_prox_iters = max_prox_iters;
    o.max_inner_iters = max_inner_iters; o.max_linesearch_iters = max_linesearch_iters;
    o.check_feasibility = check_feasibility ? 1 : 0;
    o.nonmonotone_linesearch = nonmonotone_linesearch ? 1 : 0;
    o.display_level = static_cast<int>(display_level);
    o.reserved = 0;
    return o;
  }
  void FromC(const fbstab_options_t& o) {
    sigma0 = o.sigma0; sigma_max = o.sigma_max; sigma_min = o.sigma_min; alpha = o.alpha;
    beta = o.beta; eta = o.eta; delta = o.delta; gamma = o.gamma; abs_tol = o.abs_tol;
    rel_tol = o.rel_tol; stall_tol = o.stall_tol; infeas_tol = o.infeas_tol;
    inner_tol_max = o.inner_tol_max; inner_tol_min = o.inner_tol_min;
    max_newton_iters = o.max_newton_iters; max_prox_iters = o.max_prox_iters;
    max_inner_iters = o.max_inner_iters; max_linesearch_iters = o.max_linesearch_iters;
    check_feasibility = o.check_feasibility != 0;
    nonmonotone_linesearch = o.nonmonotone_linesearch != 0;
    display_level = static_cast<Display>(o.display_level);
  }
  void ValidateOptions() { fbstab_options_t o = ToC(); fbstab_options_validate(&o); FromC(o); }
  void DefaultParameters() { fbstab_options_t o; fbstab_options_default(&o); FromC(o); }
  void ReliableParameters() { fbstab_options_t o; fbstab_options_reliable(&o); FromC(o); }
};

namespace detail {

// SolverOut of the C-ABI -> reference SolverOut.  Two outcomes that the
// reference signals by throwing out of Solve are re-thrown here:
// FBSTAB_DIVERGENCE (factorisation failure, impl:263-274) and
// FBSTAB_SATURATE_ERROR (tools::saturate, utilities.h:19-28).
inline SolverOut FromC(const fbstab_solver_out_t& c) {
  if (c.eflag == FBSTAB_DIVERGENCE)
    throw std::runtime_error("In FBstabAlgorithm::Solve: LinearSolver::Initialize failed.");
  if (c.eflag == FBSTAB_SATURATE_ERROR)
    throw std::runtime_error(
        "In tools::saturate: upper bound must be larger than the lower bound");
  SolverOut o;
  o.eflag = static_cast<ExitFlag>(c.eflag);
  o.residual = c.residual;
  o.newton_iters = c.newton_iters;
  o.prox_iters = c.prox_iters;
  o.solve_time = c.solve_time;
  o.initial_residual = c.initial_residual;
  return o;
}

// The display of the reference (fbstab_algorithm-impl.h:411-541).
//
// Every level above OFF prints the component norms |rz| |rl| |rv| of the residual,
// which SolverOut does not carry.  Display::FINAL - the reference's DEFAULT level, its
// summary block only (impl:493-541) - runs the batch kernels like Display::OFF and
// takes the three norms from fbstab_hip_*_solve_batch_final; ITER and ITER_DETAILED run
// fbstab_hip_*_solve_traced and PrintTrace() formats the records it returns, line for
// line as the reference prints them.
inline const char* ExitMessage(ExitFlag e) {
  switch (e) {
    case ExitFlag::SUCCESS: return " Success\n";
    case ExitFlag::DIVERGENCE: return " Divergence\n";
    case ExitFlag::MAXITERATIONS: return " Iteration limit exceeded\n";
    case ExitFlag::PRIMAL_INFEASIBLE: return " Primal Infeasibility\n";
    case ExitFlag::DUAL_INFEASIBLE: return " Dual Infeasibility\n";
    case ExitFlag::PRIMAL_DUAL_INFEASIBLE: return " Primal-Dual Infeasibility\n";
  }
  return " Undefined\n";
}

// Head of the summary block (impl:497-531).
template <class OutStream>
void PrintSummaryHead(const SolverOut& s, const AlgorithmParameters& p, const OutStream& os) {
  char buff[100];
  os.Print("\nOptimization completed!  Exit code:");
  os.Print(ExitMessage(s.eflag));
  snprintf(buff, 100, "Time elapsed: %f ms (-1.0 indicates timing disabled)\n", 1000.0 * s.solve_time);
  os.Print(buff);
  snprintf(buff, 100, "Proximal iterations: %d out of %d\n", s.prox_iters, p.max_prox_iters);
  os.Print(buff);
  snprintf(buff, 100, "Newton iterations: %d out of %d\n", s.newton_iters, p.max_newton_iters);
  os.Print(buff);
}

// Whether the residual blocks the reference prints in its summary belong to the point
// the solve returned: SUCCESS (rk_ evaluated at x(k), impl:162-170) and the Newton
// iteration limit (rk_ re-evaluated at the returned point, impl:188-199).  The
// infeasibility exits return the certificate while rk_ still belongs to x(k)
// (impl:204-212), and at the proximal iteration limit rk_ is one iteration old
// (impl:219-223).
inline bool FinalNormsAtReturnedPoint(const fbstab_solver_out_t& c, const AlgorithmParameters& p) {
  return c.eflag == FBSTAB_SUCCESS ||
         (c.eflag == FBSTAB_MAXITERATIONS && c.newton_iters >= p.max_newton_iters);
}

// Number of records one solve can produce: two per proximal iteration, one per
// inner-loop pass (at most one more pass than Newton steps per subproblem), one
// footer per subproblem, the summary.
inline int TraceCapacity(const AlgorithmParameters& p) {
  return 8 + 4 * (p.max_prox_iters + 1) + p.max_newton_iters + p.max_prox_iters * 2;
}

template <class OutStream>
void PrintTrace(const fbstab_trace_record_t* rec, int n, const SolverOut& s,
                const AlgorithmParameters& p, const OutStream& os) {
  char buff[100];
  const bool iter = p.display_level == Display::ITER;
  const bool detailed = p.display_level == Display::ITER_DETAILED;
  if (iter) {  // PrintIterHeader, impl:429-441
    snprintf(buff, 100, "%12s  %12s  %12s  %12s  %12s  %12s  %12s\n", "prox iter", "newton iters", "|rz|",
             "|rl|", "|rv|", "Inner res", "Inner tol");
    os.Print(buff);
  }
  for (int k = 0; k < n; k++) {
    const fbstab_trace_record_t& r = rec[k];
    const int i0 = static_cast<int>(r.i0), i1 = static_cast<int>(r.i1);
    switch (static_cast<int>(r.kind)) {
      case FBSTAB_TRACE_ITER_LINE:  // impl:411-426
        if (!iter) break;
        snprintf(buff, 100, "%12d  %12d  %12.4e  %12.4e  %12.4e  %12.4e  %12.4e\n", i0, i1, r.v[0], r.v[1],
                 r.v[2], r.v[3], r.v[4]);
        os.Print(buff);
        break;
      case FBSTAB_TRACE_DETAILED_HEADER:  // impl:443-458
        if (!detailed) break;
        snprintf(buff, 100, "Begin Prox Iter: %d, Total Newton Iters: %d, Residual: %6.4e\n", i0, i1, r.v[0]);
        os.Print(buff);
        snprintf(buff, 100, "%10s  %10s  %10s  %10s  %10s\n", "Iter", "Step Size", "|rz|", "|rl|", "|rv|");
        os.Print(buff);
        break;
      case FBSTAB_TRACE_DETAILED_LINE:  // impl:460-471
        if (!detailed) break;
        snprintf(buff, 100, "%10d  %10e  %10e  %10e  %10e\n", i0, r.v[0], r.v[1], r.v[2], r.v[3]);
        os.Print(buff);
        break;
      case FBSTAB_TRACE_DETAILED_FOOTER:  // impl:473-488
        if (!detailed) break;
        snprintf(buff, 100, "Exiting inner loop. Inner residual: %6.4e, Inner tolerance: %6.4e\n", r.v[0],
                 r.v[1]);
        os.Print(buff);
        break;
      case FBSTAB_TRACE_FINAL:  // impl:490-541
        PrintSummaryHead(s, p, os);
        snprintf(buff, 100, "%10s  %10s  %10s  %10s\n", "|rz|", "|rl|", "|rv|", "Tolerance");
        os.Print(buff);
        snprintf(buff, 100, "%10.4e  %10.4e  %10.4e  %10.4e\n", r.v[0], r.v[1], r.v[2], r.v[3]);
        os.Print(buff);
        os.Print("\n");
        break;
      default:
        break;
    }
  }
}

// The summary block (impl:493-541) from the four numbers of fbstab_hip_*_solve_batch_final.
template <class OutStream>
void PrintFinalBlock(const double (&norms)[4], const SolverOut& s, const AlgorithmParameters& p,
                     const OutStream& os) {
  fbstab_trace_record_t r;
  r.kind = FBSTAB_TRACE_FINAL;
  r.i0 = static_cast<double>(static_cast<int>(s.eflag));
  r.i1 = 0.0;
  for (int k = 0; k < 4; k++) r.v[k] = norms[k];
  r.v[4] = 0.0;
  AlgorithmParameters q = p;
  q.display_level = Display::FINAL;
  PrintTrace(&r, 1, s, q, os);
}

}  // namespace detail

// OutputStream / StandardOutput (tools/output_stream.h:15-37).
template <class T>
class OutputStream {
 public:
  void Print(const char* message) const { static_cast<const T*>(this)->PrintImplementation(message); }
};
class StandardOutput : public OutputStream<StandardOutput> {
 protected:
  void PrintImplementation(const char* message) const { printf("%s", message); }
  friend class OutputStream<StandardOutput>;
};

}  // namespace fbstab

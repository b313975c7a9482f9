/*
 * Plain-C types shared by the C-ABI of the MI355X batched FBstab solver
 * (include/fbstab_hip.h), the C++ facade (include/fbstab/) and the test
 * oracle.  Every type mirrors one declared in the reference's
 * fbstab/fbstab_algorithm.h; the citation next to each says which.
 */
#ifndef FBSTAB_TYPES_H_
#define FBSTAB_TYPES_H_

#ifdef __cplusplus
extern "C" {
#endif

/* ExitFlag, fbstab/fbstab_algorithm.h:17-24.  Values are identical. */
enum fbstab_exit_flag {
  FBSTAB_SUCCESS = 0,
  FBSTAB_DIVERGENCE = 1,      /* never produced by the reference; used here for a
                                 per-QP factorisation failure inside a batch,
                                 where the reference would throw
                                 (fbstab_algorithm-impl.h:263-274) */
  FBSTAB_MAXITERATIONS = 2,
  FBSTAB_PRIMAL_INFEASIBLE = 3,
  FBSTAB_DUAL_INFEASIBLE = 4,
  FBSTAB_PRIMAL_DUAL_INFEASIBLE = 5,
  /* Batch-only: tools::saturate(lo > hi) would have thrown inside the prox
     loop (tools/utilities.h:19-28 via fbstab_algorithm-impl.h:179-180). */
  FBSTAB_SATURATE_ERROR = 6
};

/* Display, fbstab/fbstab_algorithm.h:40-45. */
enum fbstab_display {
  FBSTAB_DISPLAY_OFF = 0,
  FBSTAB_DISPLAY_FINAL = 1,
  FBSTAB_DISPLAY_ITER = 2,
  FBSTAB_DISPLAY_ITER_DETAILED = 3
};

/* One line of the reference's per-iteration display, as data.  The reference
 * formats these inside the solver loop (PrintIterLine, PrintDetailedHeader,
 * PrintDetailedLine, PrintDetailedFooter, PrintFinal;
 * fbstab_algorithm-impl.h:411-541); here the solver records the numbers in the
 * order the reference would print them and the caller formats the ones its
 * display level shows (include/fbstab/fbstab_algorithm.h: detail::PrintTrace).
 *
 *   kind              i0           i1            v[0..4]
 *   ITER_LINE         prox_iters   newton_iters  |rz| |rl| |rv| of the outer residual,
 *                                                inner residual norm, inner tolerance
 *   DETAILED_HEADER   prox_iters   newton_iters  outer residual norm
 *   DETAILED_LINE     inner iter   -             step size, |rz| |rl| |rv| of the inner residual
 *   DETAILED_FOOTER   -            -             inner residual norm, inner tolerance
 *   FINAL             exit flag    -             |rz| |rl| |rv| of the outer residual, tolerance
 */
enum fbstab_trace_kind {
  FBSTAB_TRACE_ITER_LINE = 1,
  FBSTAB_TRACE_DETAILED_HEADER = 2,
  FBSTAB_TRACE_DETAILED_LINE = 3,
  FBSTAB_TRACE_DETAILED_FOOTER = 4,
  FBSTAB_TRACE_FINAL = 5
};
typedef struct fbstab_trace_record_t {
  double kind; /* enum fbstab_trace_kind (all fields double: one device store format) */
  double i0;
  double i1;
  double v[5];
} fbstab_trace_record_t;

/* SolverOut, fbstab/fbstab_algorithm.h:30-37.  Same member order, hence the
 * same 40-byte layout as the reference struct on LP64. */
typedef struct fbstab_solver_out_t {
  int eflag;               /* enum fbstab_exit_flag */
  int pad_;                /* the padding the reference struct has here */
  double residual;
  int newton_iters;
  int prox_iters;
  double solve_time;       /* seconds; batch calls report the batch wall time */
  double initial_residual;
} fbstab_solver_out_t;

/* AlgorithmParameters, fbstab/fbstab_algorithm.h:48-82.  Field meaning and
 * clamping (fbstab_options_validate) follow fbstab_algorithm-impl.h:7-31. */
typedef struct fbstab_options_t {
  double sigma0;
  double sigma_max;
  double sigma_min;
  double alpha;
  double beta;
  double eta;
  double delta;
  double gamma;
  double abs_tol;
  double rel_tol;
  double stall_tol;
  double infeas_tol;
  double inner_tol_max;
  double inner_tol_min;
  int max_newton_iters;
  int max_prox_iters;
  int max_inner_iters;
  int max_linesearch_iters;
  int check_feasibility;       /* bool */
  int nonmonotone_linesearch;  /* bool */
  int display_level;           /* enum fbstab_display */
  int reserved;                /* MUST BE 0 unless iterative refinement is wanted (a caller that fills the struct by
                                * hand: zero it first or start from fbstab_options_default).  k in 1..60: the MPC
                                * kernels refine a Newton step whose linear residual exceeds 2^(1 - k) of the
                                * tolerance in play - fbstab_hip.h, fbstab_hip_mpc_refined_steps.  Anything outside
                                * 0..60 is reset to 0 (off) by fbstab_options_validate. */
} fbstab_options_t;

/* AlgorithmParameters::DefaultParameters, fbstab_algorithm-impl.h:33-59. */
static inline void fbstab_options_default(fbstab_options_t* o) {
  o->sigma0 = 1e-8;
  o->sigma_max = 1e-6;
  o->sigma_min = 1e-12;
  o->alpha = 0.95;
  o->beta = 0.75;
  o->eta = 1e-8;
  o->delta = 0.2;
  o->gamma = 0.1;
  o->abs_tol = 1e-6;
  o->rel_tol = 1e-12;
  o->stall_tol = 1e-10;
  o->infeas_tol = 1e-8;
  o->inner_tol_max = 1e-2;
  o->inner_tol_min = 1e-12;
  o->max_newton_iters = 200;
  o->max_prox_iters = 30;
  o->max_inner_iters = 50;
  o->max_linesearch_iters = 20;
  o->check_feasibility = 1;
  o->nonmonotone_linesearch = 1;
  o->display_level = FBSTAB_DISPLAY_FINAL;
  o->reserved = 0;
}

/* AlgorithmParameters::ReliableParameters, fbstab_algorithm-impl.h:61-74. */
static inline void fbstab_options_reliable(fbstab_options_t* o) {
  fbstab_options_default(o);
  o->sigma0 = 1e-4;
  o->sigma_max = 1e-2;
  o->sigma_min = 1e-10;
  o->beta = 0.9;
  o->abs_tol = 1e-4;
  o->rel_tol = 1e-6;
  o->max_linesearch_iters = 40;
  o->max_newton_iters = 500;
  o->max_prox_iters = 100;
  o->nonmonotone_linesearch = 0;
}

static inline double fbstab_sat_(double x, double lo, double hi) {
  double t = x < hi ? x : hi;
  return t > lo ? t : lo;
}
static inline double fbstab_max_(double a, double b) { return a > b ? a : b; }
static inline int fbstab_imax_(int a, int b) { return a > b ? a : b; }

/* AlgorithmParameters::ValidateOptions, fbstab_algorithm-impl.h:7-31.
 * Every saturate() here has constant lo <= hi except the sigma0 one, whose
 * bounds are already clamped into [1e-13,1e-8] x [1e-6,1e2], so none can throw. */
static inline void fbstab_options_validate(fbstab_options_t* o) {
  o->sigma0 = fbstab_max_(o->sigma0, 1e-10);
  o->sigma_max = fbstab_sat_(o->sigma_max, 1e-6, 1e2);
  o->sigma_min = fbstab_sat_(o->sigma_min, 1e-13, 1e-8);
  o->sigma0 = fbstab_sat_(o->sigma0, o->sigma_min, o->sigma_max);
  o->alpha = fbstab_sat_(o->alpha, 0.001, 0.999);
  o->beta = fbstab_sat_(o->beta, 0.1, 0.99);
  o->eta = fbstab_sat_(o->eta, 1e-12, 0.499);
  o->delta = fbstab_sat_(o->delta, 0.0001, 0.99);
  o->gamma = fbstab_sat_(o->gamma, 0.001, 0.9);
  o->abs_tol = fbstab_max_(o->abs_tol, 1e-14);
  o->rel_tol = fbstab_max_(o->rel_tol, 0.0);
  o->stall_tol = fbstab_max_(o->stall_tol, 1e-14);
  o->infeas_tol = fbstab_max_(o->infeas_tol, 1e-14);
  o->inner_tol_max = fbstab_sat_(o->inner_tol_max, 1e-8, 1e2);
  o->inner_tol_min = fbstab_sat_(o->inner_tol_min, 1e-14, 1e-2);
  o->max_newton_iters = fbstab_imax_(o->max_newton_iters, 1);
  o->max_prox_iters = fbstab_imax_(o->max_prox_iters, 1);
  o->max_inner_iters = fbstab_imax_(o->max_inner_iters, 1);
  o->max_linesearch_iters = fbstab_imax_(o->max_linesearch_iters, 1);
  /* (not one of the reference's options: an uninitialised value must not switch refinement on) */
  if (o->reserved < 0 || o->reserved > 60) o->reserved = 0;
}

#ifdef __cplusplus
}
#endif

#endif /* FBSTAB_TYPES_H_ */

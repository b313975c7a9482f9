// The FBstab outer proximal-point loop and inner semismooth Newton loop,
// executed by ONE workgroup for ONE QP, entirely on the device
// (reference: fbstab/fbstab_algorithm-impl.h:113-224 `Solve` and :229-304
// `SolveProximalSubproblem`; semantics and quirks per SURVEY.md appendix A/C).
//
// What is restructured relative to the reference (same mathematics, different
// evaluation order, so results agree to rounding, not bitwise):
//   * the natural residual (rz, rl) is evaluated from the problem data once
//     per proximal iteration; inside the Newton loop it is carried forward
//     with the exact affine update  r(x + t dx) = r(x) + t*W,
//     W = (H dz + G'dl + A'dv, -G dz), which the problem policy produces
//     together with the Newton step.  A line-search trial therefore costs one
//     pass over the iterate vectors and no pass over the problem data
//     (the reference re-evaluates 4 GEMVs per trial, full_residual.cc:49-74);
//   * the trial point xp is never materialised (impl:283-297 only needs its
//     merit);
//   * the inner residual and the penalised natural residual share (rz, rl)
//     (full_residual.cc:52-66 vs :79-91 differ only by the sigma terms).
//
// A Problem policy `P` (fb_mpc.h, fb_dense.h) owns the iterate vectors and
// provides: forcing_norm, load_guess, residual, newton_step, feasibility,
// bvec, and the write_* functions.  All threads of the workgroup call every
// function; scalar results are workgroup-uniform.
#pragma once

#include <type_traits>

#include "fb_common.h"

// Step lengths evaluated per line-search pass (build knob).
#ifndef FB_LS_KT
#define FB_LS_KT 4
#endif

namespace fbk {

// TRACE: the kernel instance behind fbstab_hip_*_solve_traced.  It records the
// numbers of every display line of the reference (PrintIterLine,
// PrintDetailed*, PrintFinal; impl:411-541) into `tr` as fbstab_trace_record_t,
// in the order the reference prints them.  tr[0] counts records, tr[1] is the
// capacity, records start at tr + 8.  The component norms need extra passes over
// the vectors, so only the flat-vector policies (fb_mpc.h, fb_dense.h) have a
// traced instance; the batch kernels are compiled without any of it.
template <bool TRACE>
struct TraceState {};
template <>
struct TraceState<true> {
  double* tr = nullptr;
  // |rz|, |rl|, |rv| of the reference's rk_ and ri_ objects and ri_->Norm() as the
  // reference would hold them at this point (line-search trials overwrite ri_, impl:289)
  mutable double rkc[3] = {0.0, 0.0, 0.0}, ric[3] = {0.0, 0.0, 0.0}, ri_norm = 0.0;
  mutable double step_t = 1.0, combo = 0.0;
};
// Kernel argument carrying the trace buffer (empty for the untraced instances).
template <bool TRACE>
struct TraceArg {
  FB_DEV double* get() const { return nullptr; }
};
template <>
struct TraceArg<true> {
  double* p;
  FB_DEV double* get() const { return p; }
};

// policies whose Newton step can be refined after the fact (P::kCanRefine: linear_residual2(), refine_step())
template <class P, class = void>
struct can_refine_of : std::false_type {};
template <class P>
struct can_refine_of<P, typename std::enable_if<P::kCanRefine>::type> : std::true_type {};

// policies that pack a QP as a cooperative pass (P::kCoopLoad, P::load_guess_coop)
template <class P, class = void>
struct coop_load_of : std::false_type {};
template <class P>
struct coop_load_of<P, typename std::enable_if<P::kCoopLoad>::type> : std::true_type {};

template <class P, class C, bool TRACE = false>
struct Solver : TraceState<TRACE> {
  P& p;
  const C& c;
  const fbstab_options_t& o;
  // (solve()) the stopping tolerance abs_tol + rel_tol (1 + |(f, h, b)|) of the QP in hand and the number
  // of its Newton steps that were refined (wants_refinement)
  mutable double combo_tol_ = 0.0;
  mutable int refined_ = 0;

  FB_DEV Solver(P& p_, const C& c_, const fbstab_options_t& o_, double* tr_ = nullptr)
      : p(p_), c(c_), o(o_) {
    static_assert(!TRACE || (!P::kOwnVectorOps && !P::kFusedTrial), "no traced instance of this policy");
    if constexpr (TRACE) this->tr = tr_;
  }

  FB_DEV void emit(int kind, int i0, int i1, double v0 = 0.0, double v1 = 0.0, double v2 = 0.0,
                   double v3 = 0.0, double v4 = 0.0) const {
    if constexpr (TRACE) {
      if (c.tid == 0) {
        double* tr = this->tr;
        const int n = (int)tr[0];
        if (n < (int)tr[1]) {
          double* r = tr + 8 + 8 * (long)n;
          r[0] = kind; r[1] = i0; r[2] = i1;
          r[3] = v0; r[4] = v1; r[5] = v2; r[6] = v3; r[7] = v4;
        }
        tr[0] = n + 1;
      }
    }
  }
  // Component norms of the penalised natural residual at x (rk_, full_residual.cc:99-109).
  FB_DEV void trace_rk() const {
    if constexpr (TRACE) {
      double s[3] = {0.0, 0.0, 0.0};
      for (int i = c.tid; i < p.nz; i += C::nt) s[0] += p.rz[i] * p.rz[i];
      for (int i = c.tid; i < p.nl; i += C::nt) s[1] += p.rl[i] * p.rl[i];
      for (int i = c.tid; i < p.nv; i += C::nt) {
        const double r = pnr(p.y[i], p.v[i], o.alpha);
        s[2] += r * r;
      }
      c.sum(s);
      for (int k = 0; k < 3; k++) this->rkc[k] = sqrt(s[k]);
    }
  }
  // Component norms of the inner residual at x (ri_, full_residual.cc:49-74).
  FB_DEV void trace_ri(double sigma) const {
    if constexpr (TRACE) {
      double s[3] = {0.0, 0.0, 0.0};
      for (int i = c.tid; i < p.nz; i += C::nt) {
        const double r = p.rz[i] + sigma * (p.z[i] - p.zb[i]);
        s[0] += r * r;
      }
      for (int i = c.tid; i < p.nl; i += C::nt) {
        const double r = p.rl[i] + sigma * (p.l[i] - p.lb[i]);
        s[1] += r * r;
      }
      for (int i = c.tid; i < p.nv; i += C::nt) {
        const double r = pfb(p.y[i] + sigma * (p.v[i] - p.vb[i]), p.v[i], o.alpha);
        s[2] += r * r;
      }
      c.sum(s);
      for (int k = 0; k < 3; k++) this->ric[k] = sqrt(s[k]);
    }
  }
  FB_DEV void emit_iter_line(int prox, int newton, double inner_tol) const {  // impl:411-426
    if constexpr (TRACE)
      emit(FBSTAB_TRACE_ITER_LINE, prox, newton, this->rkc[0], this->rkc[1], this->rkc[2], this->ri_norm,
           inner_tol);
  }

  // sqrt(sum rz^2 + sum rl^2 + sum pnr(y,v)^2): norm of the penalised natural
  // residual at the current x (full_residual.cc:99-109, :40-42).
  FB_DEV double pnr_norm() const {
    if constexpr (P::kOwnVectorOps) {
      return p.pnr_norm(c, o.alpha);
    } else {
      double s[1] = {0.0};
      for (int i = c.tid; i < p.nz; i += C::nt) s[0] += p.rz[i] * p.rz[i];
      for (int i = c.tid; i < p.nl; i += C::nt) s[0] += p.rl[i] * p.rl[i];
      for (int i = c.tid; i < p.nv; i += C::nt) {
        const double r = pnr(p.y[i], p.v[i], o.alpha);
        s[0] += r * r;
      }
      c.sum(s);
      return sqrt(s[0]);
    }
  }

  // (Ei, Eo) at x + t*dx for the proximal subproblem centred at xbar
  // (full_residual.cc:49-74 and :99-109).  t == 0 never touches dx/W.
  FB_DEV void norms_at(double t, double sigma, bool want_outer, double* Ei, double* Eo) const {
    FB_WAVE_TIMER(17);
    if constexpr (P::kOwnVectorOps) {
      double e[1], f[1];
      p.template norms_at_multi<1>(c, t, 1.0, sigma, o.alpha, e, f);
      *Ei = e[0];
      *Eo = f[0];
    } else {
    double s[2] = {0.0, 0.0};
    for (int i = c.tid; i < p.nz; i += C::nt) {
      double r = p.rz[i], zi = p.z[i];
      if (t != 0.0) {
        r += t * p.wz[i];
        zi += t * p.dz[i];
      }
      const double ri = r + sigma * (zi - p.zb[i]);
      s[0] += ri * ri;
      s[1] += r * r;
    }
    for (int i = c.tid; i < p.nl; i += C::nt) {
      double r = p.rl[i], li = p.l[i];
      if (t != 0.0) {
        r += t * p.wl[i];
        li += t * p.dl[i];
      }
      const double ri = r + sigma * (li - p.lb[i]);
      s[0] += ri * ri;
      s[1] += r * r;
    }
    for (int i = c.tid; i < p.nv; i += C::nt) {
      double vi = p.v[i], yi = p.y[i];
      if (t != 0.0) {
        vi += t * p.dv[i];
        yi -= t * p.adz[i];
      }
      const double ys = yi + sigma * (vi - p.vb[i]);
      const double r = pfb(ys, vi, o.alpha);
      s[0] += r * r;
      if (want_outer) {
        const double q = pnr(yi, vi, o.alpha);
        s[1] += q * q;
      }
    }
    c.sum(s);
    *Ei = sqrt(s[0]);
    *Eo = sqrt(s[1]);
    }
  }

  // The norms of norms_at() for K step lengths t0*beta^k in ONE pass over the
  // iterate vectors (a backtracking line search usually needs several trials,
  // impl:283-297; the pass is bound by the loads, not by the arithmetic).
  template <int K>
  FB_DEV void norms_at_multi(double t0, double beta, double sigma, double (&Ei)[K],
                             double (&Eo)[K]) const {
    if constexpr (P::kOwnVectorOps) {
      p.template norms_at_multi<K>(c, t0, beta, sigma, o.alpha, Ei, Eo);
    } else {
    double tt[K];
    tt[0] = t0;
#pragma unroll
    for (int k = 1; k < K; k++) tt[k] = tt[k - 1] * beta;
    double s[2 * K];
#pragma unroll
    for (int k = 0; k < 2 * K; k++) s[k] = 0.0;
    for (int i = c.tid; i < p.nz; i += C::nt) {
      const double r0 = p.rz[i], w = p.wz[i], z0 = p.z[i], d = p.dz[i], zb = p.zb[i];
#pragma unroll
      for (int k = 0; k < K; k++) {
        const double r = fma(tt[k], w, r0);
        const double ri = r + sigma * (fma(tt[k], d, z0) - zb);
        s[k] = fma(ri, ri, s[k]);
        s[K + k] = fma(r, r, s[K + k]);
      }
    }
    for (int i = c.tid; i < p.nl; i += C::nt) {
      const double r0 = p.rl[i], w = p.wl[i], l0 = p.l[i], d = p.dl[i], lb = p.lb[i];
#pragma unroll
      for (int k = 0; k < K; k++) {
        const double r = fma(tt[k], w, r0);
        const double ri = r + sigma * (fma(tt[k], d, l0) - lb);
        s[k] = fma(ri, ri, s[k]);
        s[K + k] = fma(r, r, s[K + k]);
      }
    }
    for (int i = c.tid; i < p.nv; i += C::nt) {
      const double v0 = p.v[i], dv = p.dv[i], y0 = p.y[i], ad = p.adz[i], vb = p.vb[i];
#pragma unroll
      for (int k = 0; k < K; k++) {
        const double vi = fma(tt[k], dv, v0);
        const double yi = fma(-tt[k], ad, y0);
        const double ys = yi + sigma * (vi - vb);
        const double r = pfb(ys, vi, o.alpha);
        const double q = pnr(yi, vi, o.alpha);
        s[k] = fma(r, r, s[k]);
        s[K + k] = fma(q, q, s[K + k]);
      }
    }
    c.sum(s);
#pragma unroll
    for (int k = 0; k < K; k++) {
      Ei[k] = sqrt(s[k]);
      Eo[k] = sqrt(s[K + k]);
    }
    }
  }

  // x <- x + t*dx and the matching residual update (impl:298,
  // full_variable.cc:55-65: y += t*(dy - b) with dy = b - A dz).
  FB_DEV void accept(double t) const {
    FB_WAVE_TIMER(18);
    for (int i = c.tid; i < p.nz; i += C::nt) {
      p.z[i] += t * p.dz[i];
      p.rz[i] += t * p.wz[i];
    }
    for (int i = c.tid; i < p.nl; i += C::nt) {
      p.l[i] += t * p.dl[i];
      p.rl[i] += t * p.wl[i];
    }
    for (int i = c.tid; i < p.nv; i += C::nt) {
      p.v[i] += t * p.dv[i];
      p.y[i] -= t * p.adz[i];
    }
    c.sync();
  }

  // SolveProximalSubproblem (impl:229-304).  Returns Eo; *fail is set when
  // the linear solver could not factor (the reference throws, impl:263-274).
  FB_DEV double subproblem(double tol, double sigma, double Ek, double Ei0, int* newton_iters,
                           double* rk_last, bool* fail) const {
    if constexpr (P::kFusedTrial) {
      return subproblem_fused(tol, sigma, Ek, Ei0, newton_iters, rk_last, fail);
    } else {
      double merit[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
      double Eo = 0.0;
      FB_STAMP_DECL;
      if constexpr (TRACE) this->step_t = 1.0;  // impl:236
      for (int i = 0; i < o.max_inner_iters; i++) {
        double Ei;
        norms_at(0.0, sigma, true, &Ei, &Eo);
        *rk_last = Eo;
        if constexpr (TRACE) {  // impl:250-257
          trace_ri(sigma);
          trace_rk();
          this->ri_norm = Ei;
          emit(FBSTAB_TRACE_DETAILED_LINE, i, 0, this->step_t, this->ric[0], this->ric[1], this->ric[2]);
          if ((Ei <= tol && Eo < Ek) || (Ei <= o.inner_tol_min))
            emit(FBSTAB_TRACE_DETAILED_FOOTER, 0, 0, Ei, tol);
        }
        if ((Ei <= tol && Eo < Ek) || (Ei <= o.inner_tol_min)) break;
        if (*newton_iters >= o.max_newton_iters) break;
        FB_STAMP_LAP(16);
        if (!p.newton_step(c, sigma, o.alpha)) {
          *fail = true;
          return Eo;
        }
        if constexpr (can_refine_of<P>::value) {
          // (the leftover is a pass over z and l plus a reduction: not evaluated with the option off)
          if (o.reserved > 0 && wants_refinement(p.linear_residual2(c, sigma), tol, combo_tol_)) {
            p.refine_step(c, sigma);
            refined_++;
          }
        }
        (*newton_iters)++;
        FB_STAMP_LAP(17);
        const double cm = 0.5 * Ei * Ei;
        merit[4] = merit[3];
        merit[3] = merit[2];
        merit[2] = merit[1];
        merit[1] = merit[0];
        merit[0] = cm;
        double m0 = cm;
        if (o.nonmonotone_linesearch) {
          for (int k = 1; k < 5; k++) m0 = merit[k] > m0 ? merit[k] : m0;
        }
        double t = 1.0;
        for (int j = 0; j < o.max_linesearch_iters; j++) {
          double Et, unused;
          norms_at(t, sigma, false, &Et, &unused);
          if constexpr (TRACE) this->ri_norm = Et;
          const double mp = 0.5 * Et * Et;
          if (mp <= m0 - 2.0 * t * o.eta * cm) break;
          t *= o.beta;
        }
        if constexpr (TRACE) this->step_t = t;
        accept(t);
        FB_STAMP_LAP(18);
      }
      // ProjectDuals (impl:301, full_variable.cc:75)
      for (int i = c.tid; i < p.nv; i += C::nt) p.v[i] = fmax0(p.v[i]);
      c.sync();
      return Eo;
    }
  }

  // The same loop for policies whose newton_step also returns the residual
  // norms of the full step (the first line-search trial) and applies an
  // accepted step lazily (P::pend_t / P::flush): in the common case, t = 1
  // accepted, a Newton iteration costs no pass over the iterate vectors beyond
  // the two stage sweeps of newton_step, and the loop-top norms of the next
  // iteration are the accepted trial's norms.
  FB_DEV double subproblem_fused(double tol, double sigma, double Ek, double Ei0, int* newton_iters,
                                 double* rk_last, bool* fail) const {
    double merit[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    double Ei, Eo;
    FB_STAMP_DECL;
    if constexpr (P::kOwnVectorOps) {
      // open_prox() evaluated both norms at x = xbar together with the residual
      Ei = Ei0;
      Eo = Ek;
    } else {
      norms_at(0.0, sigma, true, &Ei, &Eo);
    }
    double Eo_top = Eo;
    for (int i = 0; i < o.max_inner_iters; i++) {
      Eo_top = Eo;
      *rk_last = Eo;
      if ((Ei <= tol && Eo < Ek) || (Ei <= o.inner_tol_min)) break;
      if (*newton_iters >= o.max_newton_iters) break;
      FB_STAMP_LAP(16);
      double ti2, to2, lin2;
      if (!p.newton_step(c, sigma, o.alpha, &ti2, &to2, &lin2)) {
        *fail = true;
        return Eo;
      }
      if (wants_refinement(lin2, tol, combo_tol_)) {
        p.refine_step(c, sigma, o.alpha, &ti2, &to2, &lin2);
        refined_++;
      }
      (*newton_iters)++;
      FB_STAMP_LAP(17);
      FB_STAMP_COUNT(31);
      const double cm = 0.5 * Ei * Ei;
      merit[4] = merit[3];
      merit[3] = merit[2];
      merit[2] = merit[1];
      merit[1] = merit[0];
      merit[0] = cm;
      double m0 = cm;
      if (o.nonmonotone_linesearch) {
        for (int k = 1; k < 5; k++) m0 = merit[k] > m0 ? merit[k] : m0;
      }
      // impl:283-297: trial j = 0 at t = 1 came with the Newton step; later
      // trials are evaluated four step lengths per pass.
      double t = 1.0;
      double Et = sqrt(ti2), Eot = sqrt(to2);
      bool known = true;  // (Et, Eot) belong to the current t
      constexpr int KT = FB_LS_KT;
      double Em[KT], Eom[KT];
      int have = 0, used = 0;  // batch of trial norms for t, t*beta, ...
      for (int j = 0; j < o.max_linesearch_iters; j++) {
        if (j > 0) {
          if (used == have) {
            FB_STAMP_COUNT(30);
            norms_at_multi<KT>(t, o.beta, sigma, Em, Eom);
            have = KT;
            used = 0;
          }
          Et = Em[0];
          Eot = Eom[0];
#pragma unroll
          for (int k = 1; k < KT; k++) {
            if (used == k) { Et = Em[k]; Eot = Eom[k]; }
          }
          used++;
          known = true;
        }
        const double mp = 0.5 * Et * Et;
        if (mp <= m0 - 2.0 * t * o.eta * cm) break;
        t *= o.beta;
        known = false;
      }
      if (!known) {
        // t = beta^max_ls is applied untested (impl:283-298); its norms are
        // needed as the next loop-top values
        if (used < have) {
          Et = Em[0];
          Eot = Eom[0];
#pragma unroll
          for (int k = 1; k < KT; k++) {
            if (used == k) { Et = Em[k]; Eot = Eom[k]; }
          }
        } else {
          norms_at(t, sigma, true, &Et, &Eot);
        }
      }
      p.pend_t = t;
      Ei = Et;
      Eo = Eot;
      FB_STAMP_LAP(18);
    }
    if constexpr (!P::kOwnVectorOps) {  // (own policies: close_subproblem() does this)
      p.flush(c);
      for (int i = c.tid; i < p.nv; i += C::nt) p.v[i] = fmax0(p.v[i]);
      c.sync();
    }
    return Eo_top;
  }

  // ---- iterative refinement of a Newton step: an OPTION, off by default -------------------------------
  // (policies with refine_step(): fb_mpc_r16.h, fb_mpc.h).  The z and l blocks of the inner residual are
  // affine in x, so their share of the first trial's norm, sqrt(lin2), is what the LINEAR solve left of
  // the Newton system.  With fbstab_options_t::reserved = k > 0 a step whose leftover exceeds
  // 2^(1 - k) of the tolerance the solver's next tests compare with - Ei <= inner_tol (impl:247),
  // Ek <= abs_tol + rel_tol (1 + |(f, h, b)|) (impl:158) - is refined once with the same factors, which
  // takes it to eps |V| |dx| in every block row: a solve MORE accurate than the reference's.
  // It is not the default because parity is measured against the reference, whose own linear solve has an
  // error: on the reference's servo-motor problem the second proximal iteration ends 11 % under the
  // tolerance, the oracle's leftover (5e-6 at that step, like the kernels') takes it over and it runs a
  // third; with the rule at k = 1 the kernels stop after two (tests/test_hostsim.py).  What closes the
  // deviation round 4 found is structural instead: the kernels substitute with the Cholesky factors where
  // that deviation came from (fb_row16.h subst_rows, fb_mpc.h solve_lower), as the reference does.
  FB_DEV bool wants_refinement(double lin2, double inner_tol, double combo_tol) const {
    if (o.reserved <= 0) return false;
    double tau = inner_tol < combo_tol ? inner_tol : combo_tol;
    for (int k = 1; k < o.reserved && k < 60; k++) tau *= 0.5;
    return lin2 > tau * tau;
  }

  // The whole solve as ONE flat loop over "Newton iterations of this row", for
  // policies that host several QPs per wavefront (fb_mpc_r16.h).  Each trip of the
  // outer loop first lets every row run the bookkeeping between two Newton
  // steps as a small state machine - subproblem exit tests, the proximal-level
  // passes (impl:158-216), finishing a QP and fetching the next - until the row
  // stands in front of a Newton step (or the queue is empty), and then all rows
  // take their Newton step and line search together.  Rows therefore never wait
  // for each other at the end of a subproblem or of a QP (nested loops reconverge
  // at their exits), and a row loses no Newton slot to its bookkeeping.
  // Same statements, same order per QP as solve()/subproblem_fused().
  // `qu.fetch(p)` binds the policy to the next QP and returns its index or -1.
  //
  template <class Queue>
  FB_DEV void solve_stream(Queue& qu, fbstab_solver_out_t* out_base) const {
    // kPause: (queues that ask for it, qu.align_rows()) a row that has finished a QP
    // waits here until no row of the wavefront stands before a Newton step, and the
    // waiting rows then fetch together: their passes over a fresh QP run side by side
    // again instead of one row at a time.
    // kWantOpen / kWantClose (policies with P::kCoopProx): the row asks for its open_prox /
    // close_subproblem pass and yields; the wavefront runs the pass for it with ALL its rows
    // (below) and the row goes on at kAfterOpen / kAfterClose.
    enum { kFetch = 0, kProxTop = 1, kInnerTop = 2, kEpilogue = 3, kNewton = 4, kDone = 5, kPause = 6,
           kWantOpen = 7, kWantClose = 8, kAfterOpen = 9, kAfterClose = 10, kWantLoad = 11, kAfterLoad = 12 };
    int phase = kFetch;
    fbstab_solver_out_t* out = out_base;
    const double sigma = o.sigma0;
    double combo_tol = 0.0, Ek = 0.0, E0 = 0.0, rk_last = 0.0, inner_tol = 0.0, dx_norm = 0.0;
    double Ei = 0.0, Eo = 0.0, Eo_top = 0.0, Ei0 = 0.0;
    // (kCoop) squared norms of the z and l blocks of the inner and of the natural residual at x: the line
    // search's A of P::trial_zl() - set by open_prox (x = xbar), advanced with every accepted step
    [[maybe_unused]] double Azi = 0.0, Azo = 0.0;
    double merit[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    int newton = 0, prox = 0, k = 0, inner_i = 0;
    [[maybe_unused]] bool fetch_now = true;  // (align_rows) the wavefront has just released its waiting rows
    // line-search trial passes run by all rows of the wavefront for one of them at a time
    constexpr bool kCoop = P::kCoopTrials && !Queue::kCanAlignRows;
    constexpr bool kCoopP = kCoop && P::kCoopProx;
    constexpr bool kCoopL = kCoopP && coop_load_of<P>::value;  // load_guess as a cooperative pass too
    [[maybe_unused]] bool fresh = false;   // (kCoopP) the open_prox asked for is the first of its QP
    [[maybe_unused]] int feas_c = kFeasible;  // (kCoopP) verdict of the close pass served last
    for (;;) {
      while (phase != kNewton && phase != kDone && phase != kPause && phase != kWantOpen && phase != kWantClose &&
             phase != kWantLoad) {
        if (phase == kFetch) {
          if constexpr (Queue::kCanAlignRows) {
            if (qu.align_rows() && !fetch_now) {
              phase = kPause;
              continue;
            }
            fetch_now = false;
          }
          const int q = qu.fetch(p);
          if (q < 0) {
            phase = kDone;
            continue;
          }
          out = out_base + q;
          combo_tol = o.abs_tol + o.rel_tol * (1.0 + p.forcing_norm(c));
          if constexpr (kCoopL) {
            phase = kWantLoad;
            continue;
          }
          p.load_guess(c);
          p.choose_costate_form(sigma);
          dx_norm = sqrt((double)p.num_primal_dual());
          if constexpr (kCoopP) {
            fresh = true;
            phase = kWantOpen;
            continue;
          }
          open_prox(sigma, &Ek, &Ei0);
          E0 = Ek;
          rk_last = Ek;
          newton = 0;
          prox = 0;
          k = 0;
          if (o.inner_tol_min > o.inner_tol_max) {
            p.write_x(c);
            finish(out, FBSTAB_SATURATE_ERROR, rk_last, newton, prox, E0);
            continue;  // phase stays kFetch
          }
          inner_tol = sat(E0, o.inner_tol_min, o.inner_tol_max);
          phase = kProxTop;
        } else if (phase == kAfterLoad) {
          // (kCoopL) the wavefront has packed this row's QP
          p.choose_costate_form(sigma);
          dx_norm = sqrt((double)p.num_primal_dual());
          fresh = true;
          phase = kWantOpen;
        } else if (phase == kAfterOpen) {
          // (kCoopP) the wavefront has run this row's open_prox pass: Ek, Ei0 are in
          if (fresh) {
            fresh = false;
            E0 = Ek;
            rk_last = Ek;
            newton = 0;
            prox = 0;
            k = 0;
            if (o.inner_tol_min > o.inner_tol_max) {
              p.write_x(c);
              finish(out, FBSTAB_SATURATE_ERROR, rk_last, newton, prox, E0);
              phase = kFetch;
              continue;
            }
            inner_tol = sat(E0, o.inner_tol_min, o.inner_tol_max);
          }
          phase = kProxTop;
        } else if (phase == kProxTop) {
          if (k >= o.max_prox_iters) {  // impl:219-223
            p.write_x(c);
            finish(out, FBSTAB_MAXITERATIONS, rk_last, newton, prox, E0);
            phase = kFetch;
            continue;
          }
          rk_last = Ek;
          if (Ek <= combo_tol || dx_norm <= o.stall_tol) {
            p.write_x(c);
            finish(out, FBSTAB_SUCCESS, rk_last, newton, prox, E0);
            phase = kFetch;
            continue;
          }
          if (o.inner_tol_min > Ek) {
            p.write_x(c);
            finish(out, FBSTAB_SATURATE_ERROR, rk_last, newton, prox, E0);
            phase = kFetch;
            continue;
          }
          inner_tol = sat(inner_tol * o.delta, o.inner_tol_min, Ek);
          // SolveProximalSubproblem prologue (impl:233-243)
          for (int m = 0; m < 5; m++) merit[m] = 0.0;
          if constexpr (P::kOwnVectorOps) {
            Ei = Ei0;
            Eo = Ek;
          } else {
            norms_at(0.0, sigma, true, &Ei, &Eo);
          }
          Eo_top = Eo;
          inner_i = 0;
          phase = kInnerTop;
        } else if (phase == kInnerTop) {
          // top of one inner iteration (impl:237-260)
          bool leave = inner_i >= o.max_inner_iters;
          if (!leave) {
            Eo_top = Eo;
            rk_last = Eo;
            leave = ((Ei <= inner_tol && Eo < Ek) || (Ei <= o.inner_tol_min)) ||
                    (newton >= o.max_newton_iters);
          }
          phase = leave ? kEpilogue : kNewton;
        } else {
          // kEpilogue: subproblem epilogue (impl:301-303) and the rest of the
          // proximal iteration (impl:186-216)
          int feas = kFeasible;
          if constexpr (kCoopP) {
            if (phase == kEpilogue) {
              phase = kWantClose;
              continue;
            }
            feas = feas_c;  // kAfterClose
          } else if constexpr (P::kOwnVectorOps) {
            feas = close_subproblem(&dx_norm);
          } else {
            p.flush(c);
            for (int i = c.tid; i < p.nv; i += C::nt) p.v[i] = fmax0(p.v[i]);
            c.sync();
          }
          const double Eo_ret = Eo_top;
          if (newton >= o.max_newton_iters) {
            if (Eo_ret < Ek) {
              p.residual(c);
              rk_last = pnr_norm();
              p.write_x(c);
            } else {
              rk_last = Ek;
              p.write_xbar(c);
            }
            finish(out, FBSTAB_MAXITERATIONS, rk_last, newton, prox, E0);
            phase = kFetch;
            continue;
          }
          if constexpr (!P::kOwnVectorOps) feas = close_subproblem(&dx_norm);
          if (feas != kFeasible) {
            const int eflag = feas == kPrimalInfeasible ? FBSTAB_PRIMAL_INFEASIBLE
                              : feas == kDualInfeasible ? FBSTAB_DUAL_INFEASIBLE
                                                        : FBSTAB_PRIMAL_DUAL_INFEASIBLE;
            p.write_certificate(c);
            finish(out, eflag, rk_last, newton, prox, E0);
            phase = kFetch;
            continue;
          }
          prox++;
          k++;
          if constexpr (kCoopP) {
            phase = kWantOpen;
            continue;
          }
          open_prox(sigma, &Ek, &Ei0);
          phase = kProxTop;
        }
      }
      if constexpr (kCoopP) {
        // ---- the proximal-level passes the rows have asked for, each run by ALL rows of the
        // wavefront for one of them (P::close_subproblem_coop / open_prox_coop); the rows that
        // stand before a Newton step wait for them - with their lanes at work
        unsigned long long wc = __ballot(phase == kWantClose), wo = __ballot(phase == kWantOpen);
        unsigned long long wl = 0ull;
        if constexpr (kCoopL) wl = __ballot(phase == kWantLoad);
        if ((wc | wo | wl) != 0ull) {
          constexpr unsigned long long kRowMask = (C::nt == 64) ? ~0ull : ((1ull << C::nt) - 1ull);
          if constexpr (kCoopL) {
            while (wl != 0ull) {
              const int owner = __builtin_ctzll(wl);
              p.load_guess_coop(owner);
              if (((threadIdx.x ^ owner) & 63 & ~(C::nt - 1)) == 0) phase = kAfterLoad;
              wl &= ~(kRowMask << (owner & ~(C::nt - 1)));
            }
          }
          while (wc != 0ull) {
            const int owner = __builtin_ctzll(wc);
            double dxn;
            const int f = p.close_subproblem_coop(owner, o.infeas_tol, o.check_feasibility != 0, &dxn);
            if (((threadIdx.x ^ owner) & 63 & ~(C::nt - 1)) == 0) {
              feas_c = f;
              dx_norm = dxn;
              phase = kAfterClose;
            }
            wc &= ~(kRowMask << (owner & ~(C::nt - 1)));
          }
          while (wo != 0ull) {
            const int owner = __builtin_ctzll(wo);
            double ek, ei0, nat2;
            p.open_prox_coop(owner, o.alpha, &ek, &ei0, &nat2);
            if (((threadIdx.x ^ owner) & 63 & ~(C::nt - 1)) == 0) {
              Ek = ek;
              Ei0 = ei0;
              Azi = Azo = nat2;
              phase = kAfterOpen;
            }
            wo &= ~(kRowMask << (owner & ~(C::nt - 1)));
          }
          continue;
        }
      }
      if constexpr (kCoop) {
        // ---- one Newton step, then the line search with COOPERATIVE trial passes: the
        // rows stay in the loop until the last one of the wavefront is done, and every
        // trial pass of a row is run by all rows (P::norms_at_multi_coop).  Same
        // statements per QP as below; a row's scalars wait in its save area while the
        // sweeps run, the rows without a step included (nothing of theirs may stay in
        // registers across a Newton step: the sweeps have none to spare).
        if (__ballot(phase != kDone) == 0ull) break;
        {
          auto sv = qu.save_area();
          sv[0] = combo_tol; sv[1] = Ek; sv[2] = E0; sv[3] = rk_last; sv[4] = inner_tol; sv[5] = dx_norm;
          sv[6] = Eo_top; sv[7] = Ei0; sv[8] = merit[0]; sv[9] = merit[1]; sv[10] = merit[2]; sv[11] = merit[3];
          sv[12] = Eo;
          sv[13] = __hiloint2double(newton, prox);
          sv[14] = __hiloint2double(k, inner_i);
          sv[15] = __longlong_as_double((long long)(out - out_base));
          sv[16] = Ei;
          sv[17] = __hiloint2double(phase, 0);
          sv[18] = Azi;
          sv[19] = Azo;
        }
        bool searching = false, taken = false;
        double t = 1.0, cm = 0.0, m0 = 0.0, Et = 0.0, Eot = 0.0;
        int fails = 0;  // failed sufficient-decrease tests so far
        // the step's sums of the z and l blocks (P::StepOut): S = the blocks' share of the t = 1 norms,
        // B = sum b^2; with A (sv[18], sv[19]) they give the blocks' share at every other step length
        double Szi = 0.0, Szo = 0.0, Bzi = 0.0, Bzo = 0.0;
        if (phase == kNewton) {
          double ti2, to2, lin2;
          typename P::StepOut so;
          const bool stepped = p.newton_step(c, sigma, o.alpha, &ti2, &to2, &lin2, &so);
          auto sv = qu.save_area();
          if (stepped && wants_refinement(lin2, sv[4], sv[0])) {
            qu.count_refinement();
            p.refine_step(c, sigma, o.alpha, &ti2, &to2, &lin2, &so);
          }
          Szi = so.lin2; Szo = so.zo2; Bzi = so.bi2; Bzo = so.bo2;
          if (!stepped) {
            out = out_base + __double_as_longlong(sv[15]);
            p.flush(c);
            p.write_x(c);
            finish(out, FBSTAB_DIVERGENCE, sv[3], __double2hiint(sv[13]), __double2loint(sv[13]), sv[2]);
            sv[17] = __hiloint2double(kFetch, 0);
          } else {
            taken = true;
            cm = 0.5 * sv[16] * sv[16];
            m0 = cm;
            if (o.nonmonotone_linesearch) {
              for (int m = 8; m < 12; m++) m0 = sv[m] > m0 ? sv[m] : m0;
            }
            Et = sqrt(ti2);
            Eot = sqrt(to2);
            // trial 0 (t = 1) came with the Newton step (impl:283-297)
            if (o.max_linesearch_iters > 0 && !(0.5 * Et * Et <= m0 - 2.0 * t * o.eta * cm)) {
              t *= o.beta;
              fails = 1;
              searching = true;
            }
          }
        }
        constexpr int KT = FB_LS_KT;
        for (;;) {
          unsigned long long need = __ballot(searching);
          if (need == 0ull) break;
          double Em[KT], Eom[KT];
#pragma unroll
          for (int m = 0; m < KT; m++) Em[m] = Eom[m] = 0.0;
          while (need != 0ull) {
            const int owner = __builtin_ctzll(need);
            double Ec[KT], Eoc[KT];  // the constraint blocks' share, squared (the pass reads nothing else)
            p.template trial_v_coop<KT>(owner, t, o.beta, sigma, o.alpha, Ec, Eoc);
            const bool mine = ((threadIdx.x ^ owner) & 63 & ~(C::nt - 1)) == 0;
#pragma unroll
            for (int m = 0; m < KT; m++) {
              Em[m] = mine ? Ec[m] : Em[m];
              Eom[m] = mine ? Eoc[m] : Eom[m];
            }
            need &= ~(((C::nt == 64) ? ~0ull : ((1ull << C::nt) - 1ull)) << (owner & ~(C::nt - 1)));
          }
          {  // ... plus the z and l blocks' share at t beta^m (P::trial_zl), and the root
            auto sv = qu.save_area();
            double tm = t;
#pragma unroll
            for (int m = 0; m < KT; m++) {
              Em[m] = sqrt(Em[m] + P::trial_zl(sv[18], Szi, Bzi, tm));
              Eom[m] = sqrt(Eom[m] + P::trial_zl(sv[19], Szo, Bzo, tm));
              tm *= o.beta;
            }
          }
#pragma unroll
          for (int m = 0; m < KT; m++) {
            if (searching) {
              Et = Em[m];
              Eot = Eom[m];
              // (the trial after the last allowed test is taken as it is, impl:283-297)
              if (fails >= o.max_linesearch_iters || 0.5 * Et * Et <= m0 - 2.0 * t * o.eta * cm) {
                searching = false;
              } else {
                t *= o.beta;
                fails++;
              }
            }
          }
        }
        {
          auto sv = qu.save_area();
          combo_tol = sv[0]; Ek = sv[1]; E0 = sv[2]; rk_last = sv[3]; inner_tol = sv[4]; dx_norm = sv[5];
          Eo_top = sv[6]; Ei0 = sv[7];
          merit[0] = sv[8]; merit[1] = sv[9]; merit[2] = sv[10]; merit[3] = sv[11];  // ([4] is only ever written)
          Eo = sv[12];
          newton = __double2hiint(sv[13]);
          prox = __double2loint(sv[13]);
          k = __double2hiint(sv[14]);
          inner_i = __double2loint(sv[14]);
          out = out_base + __double_as_longlong(sv[15]);
          Ei = sv[16];
          phase = __double2hiint(sv[17]);
          Azi = sv[18];
          Azo = sv[19];
          if (taken) {
            // merit FIFO (impl:276-278), the step and its norms
            merit[4] = sv[11]; merit[3] = sv[10]; merit[2] = sv[9]; merit[1] = sv[8]; merit[0] = cm;
            // the z and l blocks' norms at the accepted point (t = 1: the sums themselves)
            Azi = t == 1.0 ? Szi : P::trial_zl(Azi, Szi, Bzi, t);
            Azo = t == 1.0 ? Szo : P::trial_zl(Azo, Szo, Bzo, t);
            p.pend_t = t;
            Ei = Et;
            Eo = Eot;
            newton++;
            inner_i++;
            phase = kInnerTop;
          }
        }
        continue;
      }
      if constexpr (Queue::kCanAlignRows) {
        // (every row is here: none leaves the loop before all are done)
        if (__ballot(phase == kNewton) == 0ull) {
          if (__ballot(phase == kPause) == 0ull) break;
          if (phase == kPause) {
            phase = kFetch;
            fetch_now = true;
          }
          continue;
        }
      } else {
        if (phase == kDone) break;
      }
      // ---- one Newton step and its line search (impl:262-298), all rows together
      if (!Queue::kCanAlignRows || phase == kNewton) {
      // The loop's scalars are not needed until the line search is over: they wait
      // in LDS meanwhile (the sweeps have no registers to spare; left to the
      // compiler they are spilled to scratch memory inside the passes).
      {
        auto sv = qu.save_area();
        sv[0] = combo_tol; sv[1] = Ek; sv[2] = E0; sv[3] = rk_last; sv[4] = inner_tol; sv[5] = dx_norm;
        sv[6] = Eo_top; sv[7] = Ei0; sv[8] = merit[0]; sv[9] = merit[1]; sv[10] = merit[2]; sv[11] = merit[3];
        sv[12] = Eo;
        sv[13] = __hiloint2double(newton, prox);
        sv[14] = __hiloint2double(k, inner_i);
        sv[15] = __longlong_as_double((long long)(out - out_base));
        sv[16] = Ei;
      }
      double ti2, to2, lin2;
      const bool stepped = p.newton_step(c, sigma, o.alpha, &ti2, &to2, &lin2);
      if (stepped) {
        auto sv = qu.save_area();
        if (wants_refinement(lin2, sv[4], sv[0])) {
          qu.count_refinement();
          p.refine_step(c, sigma, o.alpha, &ti2, &to2, &lin2);
        }
      }
      if (!stepped) {
        auto sv = qu.save_area();
        out = out_base + __double_as_longlong(sv[15]);
        p.flush(c);
        p.write_x(c);
        finish(out, FBSTAB_DIVERGENCE, sv[3], __double2hiint(sv[13]), __double2loint(sv[13]), sv[2]);
        phase = kFetch;
      } else {
      // merit FIFO (impl:276-278): the four older entries are in the save area
      double cm, m0;
      {
        auto sv = qu.save_area();
        cm = 0.5 * sv[16] * sv[16];
        m0 = cm;
        if (o.nonmonotone_linesearch) {
          for (int m = 8; m < 12; m++) m0 = sv[m] > m0 ? sv[m] : m0;
        }
      }
      double t = 1.0;
      double Et = sqrt(ti2), Eot = sqrt(to2);
      bool known = true;
      constexpr int KT = FB_LS_KT;
      double Em[KT], Eom[KT];
      int have = 0, used = 0;
      for (int j = 0; j < o.max_linesearch_iters; j++) {
        if (j > 0) {
          if (used == have) {
            norms_at_multi<KT>(t, o.beta, sigma, Em, Eom);
            have = KT;
            used = 0;
          }
          Et = Em[0];
          Eot = Eom[0];
#pragma unroll
          for (int m = 1; m < KT; m++) {
            if (used == m) { Et = Em[m]; Eot = Eom[m]; }
          }
          used++;
          known = true;
        }
        const double mp = 0.5 * Et * Et;
        if (mp <= m0 - 2.0 * t * o.eta * cm) break;
        t *= o.beta;
        known = false;
      }
      if (!known) {
        if (used < have) {
          Et = Em[0];
          Eot = Eom[0];
#pragma unroll
          for (int m = 1; m < KT; m++) {
            if (used == m) { Et = Em[m]; Eot = Eom[m]; }
          }
        } else {
          norms_at(t, sigma, true, &Et, &Eot);
        }
      }
#if defined(FB_CLOCKSTAMP)
      {  // histogram of backtracking depth (row level), diagnostic builds only
        int depth = 0;
        for (double tq = t; tq < 0.999; tq /= o.beta) depth++;
        if (c.tid == 0) atomicAdd(&g_stamps[8 + (depth < 7 ? depth : 7)], 1ull);
      }
#endif
      p.pend_t = t;
      Ei = Et;
      Eo = Eot;
      {
        auto sv = qu.save_area();
        combo_tol = sv[0]; Ek = sv[1]; E0 = sv[2]; rk_last = sv[3]; inner_tol = sv[4]; dx_norm = sv[5];
        Eo_top = sv[6]; Ei0 = sv[7];
        merit[4] = sv[11]; merit[3] = sv[10]; merit[2] = sv[9]; merit[1] = sv[8]; merit[0] = cm;
        newton = __double2hiint(sv[13]) + 1;
        prox = __double2loint(sv[13]);
        k = __double2hiint(sv[14]);
        inner_i = __double2loint(sv[14]) + 1;
        out = out_base + __double_as_longlong(sv[15]);
      }
      phase = kInnerTop;
      }  // step taken
      }
    }
  }


  // FBstabAlgorithm::Solve (impl:113-224).
  FB_DEV void solve(fbstab_solver_out_t* out) const {
    const double sigma = o.sigma0;
    const double combo_tol = o.abs_tol + o.rel_tol * (1.0 + p.forcing_norm(c));
    combo_tol_ = combo_tol;
    p.load_guess(c);  // xk <- (z0,l0,v0), y = b - A z (impl:140, :334-347)
    double dx_norm = sqrt((double)p.num_primal_dual());  // dx.Fill(1) (impl:142)
    double Ek, Ei0;
    open_prox(sigma, &Ek, &Ei0);
    if constexpr (TRACE) {
      this->combo = combo_tol;
      trace_rk();
    }
    const double E0 = Ek;
    double rk_last = Ek;
    int newton = 0, prox = 0;
    int eflag = FBSTAB_MAXITERATIONS;
    double inner_tol = 0.0;
    if (o.inner_tol_min > o.inner_tol_max) {
      eflag = FBSTAB_SATURATE_ERROR;  // tools::saturate would throw (impl:150-151)
      p.write_x(c);
      finish(out, eflag, rk_last, newton, prox, E0);
      return;
    }
    inner_tol = sat(E0, o.inner_tol_min, o.inner_tol_max);

    bool done = false;
    for (int k = 0; k < o.max_prox_iters && !done; k++) {
      rk_last = Ek;
      [[maybe_unused]] double rk_top[3];
      if constexpr (TRACE) {
        for (int j = 0; j < 3; j++) rk_top[j] = this->rkc[j];
      }
      if (Ek <= combo_tol || dx_norm <= o.stall_tol) {
        emit_iter_line(prox, newton, inner_tol);  // impl:165
        eflag = FBSTAB_SUCCESS;
        p.write_x(c);
        done = true;
        break;
      }
      if constexpr (TRACE) emit(FBSTAB_TRACE_DETAILED_HEADER, prox, newton, Ek);  // impl:171
      emit_iter_line(prox, newton, inner_tol);                                    // impl:172
      if (o.inner_tol_min > Ek) {  // tools::saturate(lo > hi) throws (impl:179-180)
        eflag = FBSTAB_SATURATE_ERROR;
        p.write_x(c);
        done = true;
        break;
      }
      inner_tol = sat(inner_tol * o.delta, o.inner_tol_min, Ek);
      bool fail = false;
      const double Eo = subproblem(inner_tol, sigma, Ek, Ei0, &newton, &rk_last, &fail);
      if (fail) {
        eflag = FBSTAB_DIVERGENCE;
        if constexpr (P::kOwnVectorOps) p.flush(c);
        p.write_x(c);
        done = true;
        break;
      }
      int feas = kFeasible;
      if constexpr (P::kOwnVectorOps) feas = close_subproblem(&dx_norm);
      if (newton >= o.max_newton_iters) {  // impl:188-199
        eflag = FBSTAB_MAXITERATIONS;
        if (Eo < Ek) {
          p.residual(c);
          rk_last = pnr_norm();
          trace_rk();
          p.write_x(c);
        } else {
          rk_last = Ek;
          if constexpr (TRACE) {
            for (int j = 0; j < 3; j++) this->rkc[j] = rk_top[j];
          }
          p.write_xbar(c);
        }
        done = true;
        break;
      }
      // dx <- x(k+1) - x(k) (impl:202-203; its norm excludes y,
      // full_variable.cc:77-83) and the infeasibility certificates
      if constexpr (!P::kOwnVectorOps) feas = close_subproblem(&dx_norm);
      if (feas != kFeasible) {
        eflag = feas == kPrimalInfeasible ? FBSTAB_PRIMAL_INFEASIBLE
                : feas == kDualInfeasible ? FBSTAB_DUAL_INFEASIBLE
                                          : FBSTAB_PRIMAL_DUAL_INFEASIBLE;
        p.write_certificate(c);  // x <- dx (impl:205-210); residual stays stale
        done = true;
        break;
      }
      prox++;
      // xbar <- x and the residual at the projected x(k+1): serves the next
      // loop-top test (impl:162-163) and the first inner iteration (impl:239-243).
      open_prox(sigma, &Ek, &Ei0);
      trace_rk();
    }
    if (!done) {
      // Timeout exit (impl:219-223): residual is whatever rk last held.
      eflag = FBSTAB_MAXITERATIONS;
      p.write_x(c);
    }
    finish(out, eflag, rk_last, newton, prox, E0);
  }

 private:
  // xbar <- x, natural residual at x, Ek = its norm (impl:140-146, :212-216).
  // Policies with their own vector passes do all of it in one sweep and also
  // return the inner residual norm at x = xbar (impl:239-243).
  FB_DEV void open_prox(double sigma, double* Ek, double* Ei0) const {
    if constexpr (P::kOwnVectorOps) {
      p.open_prox(c, sigma, o.alpha, Ek, Ei0);
    } else {
      copy_x_to_xbar();
      p.residual(c);
      *Ek = pnr_norm();
      *Ei0 = 0.0;  // unused: subproblem() evaluates it
    }
  }
  // dx <- x - xbar, its norm, and the feasibility verdict (impl:202-210).  Own
  // policies also apply the pending step and project the duals here (impl:301).
  FB_DEV int close_subproblem(double* dx_norm) const {
    if constexpr (P::kOwnVectorOps) {
      return p.close_subproblem(c, o.infeas_tol, o.check_feasibility != 0, dx_norm);
    } else {
      *dx_norm = dx_from_xbar();
      return o.check_feasibility ? p.feasibility(c, o.infeas_tol) : kFeasible;
    }
  }
  // dx <- x - xbar (z, l, v blocks); returns its norm.
  FB_DEV double dx_from_xbar() const {
    if constexpr (P::kOwnVectorOps) {
      return p.dx_from_xbar(c);
    } else {
      double s[1] = {0.0};
      for (int i = c.tid; i < p.nz; i += C::nt) {
        const double d = p.z[i] - p.zb[i];
        p.dz[i] = d;
        s[0] += d * d;
      }
      for (int i = c.tid; i < p.nl; i += C::nt) {
        const double d = p.l[i] - p.lb[i];
        p.dl[i] = d;
        s[0] += d * d;
      }
      for (int i = c.tid; i < p.nv; i += C::nt) {
        const double d = p.v[i] - p.vb[i];
        p.dv[i] = d;
        s[0] += d * d;
      }
      c.sum(s);
      c.sync();
      return sqrt(s[0]);
    }
  }
  FB_DEV void copy_x_to_xbar() const {
    if constexpr (P::kOwnVectorOps) {
      p.copy_x_to_xbar(c);
    } else {
      for (int i = c.tid; i < p.nz; i += C::nt) p.zb[i] = p.z[i];
      for (int i = c.tid; i < p.nl; i += C::nt) p.lb[i] = p.l[i];
      for (int i = c.tid; i < p.nv; i += C::nt) {
        p.vb[i] = p.v[i];
        p.yb[i] = p.y[i];
      }
      c.sync();
    }
  }
  FB_DEV void finish(fbstab_solver_out_t* out, int eflag, double residual, int newton,
                     int prox, double E0) const {
    if constexpr (TRACE)  // impl:381, :493-541
      emit(FBSTAB_TRACE_FINAL, eflag, 0, this->rkc[0], this->rkc[1], this->rkc[2], this->combo);
    if (c.tid == 0) {
      out->eflag = eflag;
      out->pad_ = 0;
      out->residual = residual;
      out->newton_iters = newton;
      out->prox_iters = prox;
      out->solve_time = -1.0;  // filled by the host with the batch wall time
      out->initial_residual = E0;
    }
  }
};

}  // namespace fbk

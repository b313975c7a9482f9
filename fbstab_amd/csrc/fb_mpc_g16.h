// Register-resident Riccati Newton step for MPC QPs whose stage width
// nx + nu fits one 16-lane DPP row: FOUR QPs per 64-wide wavefront, lane r of a
// row owning row r of every stage matrix.  Compile-time (NX, NU, NC).
//
// The stage recursion of RiccatiLinearSolver (riccati_linear_solver.cc:125-206)
// is used in its equivalent block form.  With
//     K_i  = [Qbar + inv(Pi_i)  Sbar'; Sbar  Rbar]     (NS x NS, NS = nx+nu)
//     Lc   = chol(K_i) = [M 0; SM SG]      (the reference's M, SM, SG are its blocks)
//     W    = [A B] inv(Lc)' = [AM  -P]     (the reference's AM and P)
//     Pi_{i+1} = sigma I + W W'            (= sigma I + P P' + AM AM', :177-183)
// the vector recursions (:212-327) become, with g_i = [-h_i; ru_i]:
//     t_i = inv(Lc) g_i = [-tx; tu],  theta_{i+1} = r2_{i+1} - W t_i,
//     h_{i+1} = inv(Pi_{i+1}) theta_{i+1} - rx_{i+1},
//     [dx_i; du_i] = inv(Lc)' (t_i - W' dl_{i+1}),   dl_i = -inv(Pi_i)(theta_i + dx_i).
// One 16-step Cholesky per stage therefore replaces the reference's chol(M),
// two right-solves and chol(SG); L(i+1) = chol(Pi_{i+1}) is the second chain.
//
// Cross-lane traffic: the sequential chains (Cholesky, triangular inverse)
// use DPP row broadcasts (v_mov_b32_dpp row_newbcast), no LDS round trip; the
// bulk products (C'Gamma C, W W', T'T) read the other rows as LDS broadcasts.
// Everything else of the policy (residual, feasibility, I/O) is the generic
// MpcProblem code run by the 16 lanes of the row.
#pragma once

#include <type_traits>
#include <utility>

#include "fb_mpc.h"

namespace fbk {

#if !defined(FB_HOSTSIM)

// ---- compile-time loops ------------------------------------------------------
template <int B, class F, int... I>
FB_DEV void sfor_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, B + I>{}), ...);
}
template <int B, int E, class F>
FB_DEV void sfor(F&& f) {
  if constexpr (E > B) sfor_impl<B>(f, std::make_integer_sequence<int, E - B>{});
}

// ---- DPP helpers on one 16-lane row -------------------------------------------
template <int CTRL>
FB_DEV double dpp_mov(double x) {
  // 64-bit DPP move: for row_newbcast gfx950 has a single v_mov_b64_dpp; other
  // controls are split into two 32-bit moves by the compiler.  old = 0 with
  // bound_ctrl, so the destination is not tied to a copy of the source.
  return __builtin_amdgcn_update_dpp(0.0, x, CTRL, 0xf, 0xf, true);
}
// value of lane J of this lane's 16-lane row
template <int J>
FB_DEV double bc(double x) { return dpp_mov<0x150 + J>(x); }
template <int J>
FB_DEV int bci(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x150 + J, 0xf, 0xf, true); }

struct OpSum16 { static FB_DEV double apply(double a, double b) { return a + b; } };
struct OpMax16 { static FB_DEV double apply(double a, double b) { return a > b ? a : b; } };

// All-lanes reduction over the row; rotate-by-half-period keeps every lane's
// result bitwise identical (each step pairs lanes that hold equal values).
template <class Op>
FB_DEV double row_reduce(double x) {
  x = Op::apply(x, dpp_mov<0x128>(x));  // row_ror:8
  x = Op::apply(x, dpp_mov<0x124>(x));  // row_ror:4
  x = Op::apply(x, dpp_mov<0x122>(x));  // row_ror:2
  x = Op::apply(x, dpp_mov<0x121>(x));  // row_ror:1
  return x;
}

// Thread context of one 16-lane row (a "virtual workgroup" of 16 threads).
struct Ctx16 {
  int tid;  // lane within the row
  static constexpr int nt = 16;
  FB_DEV void sync() const {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
  }
  template <int K>
  FB_DEV void sum(double (&v)[K]) const {
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = row_reduce<OpSum16>(v[k]);
  }
  template <int K>
  FB_DEV void max(double (&v)[K]) const {
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = row_reduce<OpMax16>(v[k]);
  }
};

// 1/sqrt(d) to full double precision: v_rsq_f64 seed + two Newton steps.
FB_DEV double rsqrt_full(double d) {
  double r = __builtin_amdgcn_rsq(d);
  double e = fma(-d * r, r, 1.0);
  r = fma(r * 0.5, e, r);
  e = fma(-d * r, r, 1.0);
  r = fma(r * 0.5, e, r);
  return r;
}

// In-place Cholesky of an N x N SPD matrix held one row per lane (a[c] = A[r][c],
// lower triangle meaningful).  On return a[k] = L[r][k] for k < r and the
// diagonal slot a[r] holds 1/L[r][r] (only the reciprocal is ever needed).
// Returns false (row-uniform) on a non-positive pivot.
template <int N>
FB_DEV bool chol_rows(double (&a)[N], int r, double diag_add) {
  bool ok = true;
  sfor<0, N>([&](auto J) {
    constexpr int j = decltype(J)::value;
    // the caller's "+ diag_add * I" is applied here, at pivot time: no per-lane
    // (r == j) selects, which the compiler would otherwise hoist and keep live
    const double d = bc<j>(a[j]) + diag_add;
    ok = ok && (d > 0.0);
    const double q = rsqrt_full(d);
    const double lj = a[j] * q;  // L[r][j] for r > j
    a[j] = (r == j) ? q : lj;
    sfor<j + 1, N>([&](auto Cc) {
      constexpr int c = decltype(Cc)::value;
      a[c] = fma(-lj, bc<c>(lj), a[c]);
    });
  });
  return ok;
}

// Column r of inv(L) for the row-held factor of chol_rows (a[k] = L[r][k],
// a[r] = 1/L[r][r]).
template <int N>
FB_DEV void tri_inv_cols(const double (&a)[N], double (&x)[N], int r) {
  sfor<0, N>([&](auto RR) {
    constexpr int rr = decltype(RR)::value;
    double s = (r == rr) ? 1.0 : 0.0;
    sfor<0, rr>([&](auto Kk) {
      constexpr int k = decltype(Kk)::value;
      s = fma(-bc<rr>(a[k]), x[k], s);
    });
    x[rr] = s * bc<rr>(a[rr]);
  });
}

template <int NX, int NU, int NC>
struct MpcProblemG16 : MpcProblem<Ctx16> {
  typedef Ctx16 C;
  static constexpr bool kFusedTrial = true;
  static constexpr int NS = NX + NU;
  static constexpr int KS = (NC + 15) / 16;  // constraint slots per lane
  static_assert(NS <= 16, "stage width must fit one DPP row");
  // LDS regions of the hot path (they alias the generic tile, idle meanwhile).
  // Row strides are odd numbers of doubles and the four rows' regions are
  // offset by 16 doubles mod 32, so that the 32 lanes of a half-wave (two
  // rows) hit 32 distinct 8-byte bank pairs both for "own row, strided" and
  // for "all lanes of a row read one address" accesses.
  static constexpr int CS = NC | 1;           // stride of Cl / Gc rows
  static constexpr int TS = 17;               // stride of the 16x16 scratch
  static constexpr int kCl = 0;               // C = [E L] as [col][k]
  static constexpr int kGc = 16 * CS;         // Gamma C, same shape
  static constexpr int kTr = 32 * CS;         // 16x16 transpose / W / T buffer
  static constexpr int kLdsDoubles = 32 * CS + 16 * TS;
  // factor record offsets (doubles)
  static constexpr int fXc = 0, fW = 256, fPinv = 512, fT = 512 + 16 * NX, fTh = fT + 16;
  static constexpr int kRecord = fTh + 16;

  // Step length of an accepted but not yet applied Newton step (0 = none).
  // The forward sweep of the next newton_step applies it stage by stage; every
  // other consumer of x calls flush() first.
  double pend_t = 0.0;

  // x <- x + t dx, (rz, rl) <- (rz, rl) + t W (impl:298, full_variable.cc:55-65).
  FB_DEV void flush(const C& c) {
    const double t = pend_t;
    if (t != 0.0) {
      for (int i = c.tid; i < nz; i += C::nt) { z[i] += t * dz[i]; rz[i] += t * wz[i]; }
      for (int i = c.tid; i < nl; i += C::nt) { l[i] += t * dl[i]; rl[i] += t * wl[i]; }
      for (int i = c.tid; i < nv; i += C::nt) { v[i] += t * dv[i]; y[i] -= t * adz[i]; }
      c.sync();
    }
    pend_t = 0.0;
  }

  // One Newton step.  Forward sweep: applies the pending step, factors and
  // runs the forward substitution.  Backward sweep: back substitution fused
  // with dv, A dz, the residual increment W and the squared norms of the inner
  // and penalised natural residuals at the FULL step x + dx (the first
  // line-search trial, fbstab_algorithm-impl.h:283-290), returned in
  // *trial_inner2 / *trial_outer2.
  FB_DEV bool newton_step(const C& c, double sigma, double alpha, double* trial_inner2,
                          double* trial_outer2) {
    // Locals only below: lambdas must not capture `this`, or the policy object
    // is forced into (scratch) memory and every pointer load becomes a flat op.
    const int N = this->lay.N;
    const long f_stride = this->lay.f_stride;
    const int r = c.tid;
    const MpcData D = this->D;
    lds_ptr Cl = this->lds + kCl;
    lds_ptr Tr = this->lds + kTr;
    lds_ptr Gc = this->lds + kGc;
    double* const fac_ = this->fac;
    double* const z_ = this->z; double* const l_ = this->l; double* const v_ = this->v;
    double* const y_ = this->y; double* const zb_ = this->zb; double* const lb_ = this->lb;
    double* const vb_ = this->vb; double* const dz_ = this->dz; double* const dl_ = this->dl;
    double* const dv_ = this->dv; double* const adz_ = this->adz; double* const rz_ = this->rz;
    double* const rl_ = this->rl; double* const wz_ = this->wz; double* const wl_ = this->wl;
    double* const gam_ = this->gam; double* const rvm_ = this->rvm;
    const bool rx = r < NX;          // lane owns a state row
    const bool rs_ = r < NS;         // lane owns a row of the stage block
    const int ru = r - NX;
    const double tp = pend_t;        // pending step length
    pend_t = 0.0;

    double Pinv[NX];  // row r of inv(Pi_i); Pi_0 = sigma I (riccati_linear_solver.cc:127)
    sfor<0, NX>([&](auto Cc) { Pinv[decltype(Cc)::value] = (rx && r == decltype(Cc)::value) ? 1.0 / sigma : 0.0; });
    double thp = 0.0;
    bool ok = true;

    FB_STAMP_DECL;
    // Everything a forward stage reads from memory, as one register bundle, so
    // that the loads of stage i+1 can be issued while stage i still runs its
    // second Cholesky chain (software pipelining by hand: the compiler does not
    // move loads across the loop back-edge).  No lane-dependent branches:
    // out-of-range lanes read a clamped, valid address.
    struct FwdIn {
      double sv[KS], sy[KS], svb[KS], sdv[KS], sadz[KS];
      double zz, rzz, zbz, dzz, wzz, ll, rll, lbl, dll, wll;
      double Cc[NC];
      double K[NS];
    };
    auto load_fwd = [&](int i, FwdIn& in) {
      sfor<0, KS>([&](auto S_) {
        constexpr int s = decltype(S_)::value;
        const int k = r + 16 * s;
        const long g = (long)i * NC + (k < NC ? k : NC - 1);
        in.sv[s] = v_[g];
        in.sy[s] = y_[g];
        in.svb[s] = vb_[g];
        // (0 * stale data could be NaN: no pending step => exact zeros)
        in.sdv[s] = tp != 0.0 ? dv_[g] : 0.0;
        in.sadz[s] = tp != 0.0 ? adz_[g] : 0.0;
      });
      const long gz = (long)i * NS + (rs_ ? r : 0);
      const long gl = (long)i * NX + (rx ? r : 0);
      in.zz = z_[gz];
      in.rzz = rz_[gz];
      in.zbz = zb_[gz];
      in.dzz = tp != 0.0 ? dz_[gz] : 0.0;
      in.wzz = tp != 0.0 ? wz_[gz] : 0.0;
      in.ll = l_[gl];
      in.rll = rl_[gl];
      in.lbl = lb_[gl];
      in.dll = tp != 0.0 ? dl_[gl] : 0.0;
      in.wll = tp != 0.0 ? wl_[gl] : 0.0;
      // column r of C = [E L]
      {
        const double* src = rx ? D.E + ((long)i * NX + r) * NC : D.L + ((long)i * NU + (rs_ ? ru : 0)) * NC;
        sfor<0, NC>([&](auto Kk) { in.Cc[decltype(Kk)::value] = src[decltype(Kk)::value]; });
      }
      // Row r of [Q S'; S R].  Two lane classes, each with compile-time strides
      // so that every element is an immediate offset from one base pointer
      // (per-lane strides make the compiler hoist one address per element).
      if (rx) {
        const double* q = D.Q + (long)i * NX * NX + r;
        const double* st = D.S + (long)i * NU * NX + (long)r * NU;
        sfor<0, NX>([&](auto Cc) { in.K[decltype(Cc)::value] = q[decltype(Cc)::value * NX]; });
        sfor<NX, NS>([&](auto Cc) { in.K[decltype(Cc)::value] = st[decltype(Cc)::value - NX]; });
      } else {
        const double* sr = D.S + (long)i * NU * NX + (rs_ ? ru : 0);
        const double* rr = D.R + (long)i * NU * NU + (rs_ ? ru : 0);
        sfor<0, NX>([&](auto Cc) { in.K[decltype(Cc)::value] = sr[decltype(Cc)::value * NU]; });
        sfor<NX, NS>([&](auto Cc) { in.K[decltype(Cc)::value] = rr[(decltype(Cc)::value - NX) * NU]; });
      }
      if constexpr (NS < 16) {
        if (!rs_) {
          sfor<0, NS>([&](auto Cc) { in.K[decltype(Cc)::value] = 0.0; });
          sfor<0, NC>([&](auto Kk) { in.Cc[decltype(Kk)::value] = 0.0; });
        }
      }
    };
    FwdIn cur;
    load_fwd(0, cur);
    // ===================== forward sweep ===================================
    for (int i = 0; i <= N; i++) {
      double* F = fac_ + (long)i * f_stride;
      // Lane id made opaque per iteration: (ro == j) selects are then recomputed
      // where used instead of being hoisted out of the loop as 16+ live masks.
      int ro = r;
      asm volatile("" : "+v"(ro));
      const long gz = (long)i * NS + (rs_ ? r : 0);
      const long gl = (long)i * NX + (rx ? r : 0);
      double Cc_[NC];
      sfor<0, NC>([&](auto Kk) { Cc_[decltype(Kk)::value] = cur.Cc[decltype(Kk)::value]; });
      double K[NS];
      sfor<0, NS>([&](auto Cc) { K[decltype(Cc)::value] = cur.K[decltype(Cc)::value]; });
      // ---- pending step (tp = 0: no-op), PFB gradient (riccati_linear_solver.cc:91-99)
      double Gam[KS], Rvm[KS];
      sfor<0, KS>([&](auto S_) {
        constexpr int s = decltype(S_)::value;
        const int k = r + 16 * s;
        const double vk = fma(tp, cur.sdv[s], cur.sv[s]);
        const double yk = fma(-tp, cur.sadz[s], cur.sy[s]);
        const double ys = yk + sigma * (vk - cur.svb[s]);
        double ph, g0, g1;
        pfb_all(ys, vk, alpha, &ph, &g0, &g1);
        const double imu = rcp_fast(g1 + sigma * g0);
        const double G_ = g0 * imu;
        const double rm = -ph * imu;
        if (k < NC) {
          const long g = (long)i * NC + k;
          v_[g] = vk;
          y_[g] = yk;
          gam_[g] = G_;
          rvm_[g] = rm;
        }
        Gam[s] = k < NC ? G_ : 0.0;
        Rvm[s] = k < NC ? rm : 0.0;
      });
      // pending step on (z, rz), (l, rl); eliminated right-hand side (:222-225)
      const double zz = fma(tp, cur.dzz, cur.zz);
      const double rzz = fma(tp, cur.wzz, cur.rzz);
      const double ll = fma(tp, cur.dll, cur.ll);
      const double rll = fma(tp, cur.wll, cur.rll);
      if (rs_) {
        z_[gz] = zz;
        rz_[gz] = rzz;
      }
      if (rx) {
        l_[gl] = ll;
        rl_[gl] = rll;
      }
      double r1 = rs_ ? -(rzz + sigma * (zz - cur.zbz)) : 0.0;
      const double r2 = rx ? rll + sigma * (ll - cur.lbl) : 0.0;
      __builtin_amdgcn_sched_barrier(0);
      FB_STAMP_LAP(0);
      // C to LDS (other rows read it as broadcasts)
      c.sync();
      sfor<0, NC>([&](auto Kk) { Cl[r * CS + decltype(Kk)::value] = Cc_[decltype(Kk)::value]; });
      // K row: H + sigma I + inv(Pi) block + C' Gamma C (:101-123, :142-145).
      // Gamma_k C[k][r] goes to LDS too so that k can stay a rolled loop (a
      // fully unrolled 16x20 product drives the register allocator to spill).
      sfor<0, NC>([&](auto Kk) {
        constexpr int k = decltype(Kk)::value;
        const double gk = bc<(k & 15)>(Gam[k >> 4]);
        const double rk = bc<(k & 15)>(Rvm[k >> 4]);
        r1 = fma(-Cc_[k], rk, r1);
        Gc[r * CS + k] = gk * Cc_[k];
      });
      sfor<0, NX>([&](auto Cc) { K[decltype(Cc)::value] += Pinv[decltype(Cc)::value]; });
      c.sync();
#pragma unroll 2
      for (int k = 0; k < NC; k++) {
        const double gck = Gc[r * CS + k];
        sfor<0, NS>([&](auto Cc) { K[decltype(Cc)::value] = fma(gck, Cl[decltype(Cc)::value * CS + k], K[decltype(Cc)::value]); });
      }
      __builtin_amdgcn_sched_barrier(0);
      FB_STAMP_LAP(1);
      // theta(i), h(i) = inv(Pi) theta - rx, g = [-h; ru] (:231-236, :252-261)
      const double th = thp + r2;
      double gv = r1;
      {
        double hsum = 0.0;
        sfor<0, NX>([&](auto Cc) { hsum = fma(Pinv[decltype(Cc)::value], bc<decltype(Cc)::value>(th), hsum); });
        if (rx) gv = r1 - hsum;
      }
      sfor<0, NX>([&](auto Cc) { F[fPinv + decltype(Cc)::value * 16 + r] = Pinv[decltype(Cc)::value]; });
      F[fTh + r] = th;
      __builtin_amdgcn_sched_barrier(0);
      FB_STAMP_LAP(2);
      // [A B] row r for W, requested now so that it arrives behind the chains
      double AB[NS];
      sfor<0, NS>([&](auto Cc) { AB[decltype(Cc)::value] = 0.0; });
      if (rx && i < N) {
        const double* pa = D.A + (long)i * NX * NX + r;
        const double* pb = D.B + (long)i * NX * NU + r;
        sfor<0, NX>([&](auto Cc) { AB[decltype(Cc)::value] = pa[decltype(Cc)::value * NX]; });
        sfor<NX, NS>([&](auto Cc) { AB[decltype(Cc)::value] = pb[(decltype(Cc)::value - NX) * NX]; });
      }
      // ---- Lc = chol(K), columns of inv(Lc)
      ok = chol_rows<NS>(K, ro, sigma) && ok;
      if (!ok) return false;
      double XC[NS];
      FB_STAMP_LAP(3);
      __builtin_amdgcn_sched_barrier(0);
      tri_inv_cols<NS>(K, XC, ro);
      FB_STAMP_LAP(4);
      __builtin_amdgcn_sched_barrier(0);
      sfor<0, NS>([&](auto RR) { F[fXc + decltype(RR)::value * 16 + r] = XC[decltype(RR)::value]; });
      // rows of inv(Lc) through an LDS transpose
      double XR[NS];
      c.sync();
      sfor<0, NS>([&](auto RR) { Tr[decltype(RR)::value * TS + r] = XC[decltype(RR)::value]; });
      c.sync();
      sfor<0, NS>([&](auto Cc) { XR[decltype(Cc)::value] = Tr[r * TS + decltype(Cc)::value]; });
      __builtin_amdgcn_sched_barrier(0);
      // t = inv(Lc) g
      double tvec = 0.0;
      sfor<0, NS>([&](auto Cc) { tvec = fma(XR[decltype(Cc)::value], bc<decltype(Cc)::value>(gv), tvec); });
      F[fT + r] = tvec;
      FB_STAMP_LAP(5);
      if (i < N) {
        __builtin_amdgcn_sched_barrier(0);
        // ---- W = [A B] inv(Lc)'  (AM and -P of :149-175)
        double W[NS];
        sfor<0, NS>([&](auto Cc) {
          constexpr int cc = decltype(Cc)::value;
          double s = 0.0;
          sfor<0, cc + 1>([&](auto Kk) { s = fma(AB[decltype(Kk)::value], bc<cc>(XR[decltype(Kk)::value]), s); });
          W[cc] = s;
        });
        sfor<0, NS>([&](auto Cc) { F[fW + decltype(Cc)::value * 16 + r] = W[decltype(Cc)::value]; });
        // next stage's inputs: in flight during the second chain below
        load_fwd(i + 1, cur);
        __builtin_amdgcn_sched_barrier(0);
        FB_STAMP_LAP(6);
        // theta(i+1) partial = -W t
        thp = 0.0;
        sfor<0, NS>([&](auto Cc) { thp = fma(-W[decltype(Cc)::value], bc<decltype(Cc)::value>(tvec), thp); });
        // ---- Pi(i+1) = sigma I + W W' ; L = chol ; inv(Pi) = T'T, T = inv(L)
        double Pn[NX];
        sfor<0, NX>([&](auto Cc) {
          constexpr int cc = decltype(Cc)::value;
          double acc = 0.0;
          sfor<0, NS>([&](auto Kk) { acc = fma(W[decltype(Kk)::value], bc<cc>(W[decltype(Kk)::value]), acc); });
          Pn[cc] = rx ? acc : 0.0;
          __builtin_amdgcn_sched_barrier(0);
        });
        __builtin_amdgcn_sched_barrier(0);
        FB_STAMP_LAP(7);
        ok = chol_rows<NX>(Pn, ro, sigma) && ok;
        if (!ok) return false;
        __builtin_amdgcn_sched_barrier(0);
        double T[NX];
        tri_inv_cols<NX>(Pn, T, ro);
        __builtin_amdgcn_sched_barrier(0);
        // inv(Pi)[r][cc] = sum_k T[k][r] T[k][cc], T[k][cc] = lane cc's T[k] (zero for k < cc)
        sfor<0, NX>([&](auto Cc) {
          constexpr int cc = decltype(Cc)::value;
          double acc = 0.0;
          sfor<cc, NX>([&](auto Kk) { acc = fma(T[decltype(Kk)::value], bc<cc>(T[decltype(Kk)::value]), acc); });
          Pinv[cc] = rx ? acc : 0.0;
          __builtin_amdgcn_sched_barrier(0);
        });
        FB_STAMP_LAP(8);
      }
    }

    // ============ backward sweep (:267-341), fused with dv, A dz, W and the
    // residual norms of the first line-search trial ==========================
    double lp = 0.0;    // dl(i+1), lanes < NX
    double dzn = 0.0;   // dx(i+1), lanes < NX
    double s_in = 0.0, s_out = 0.0;
    for (int i = N; i >= 0; i--) {
      const double* F = fac_ + (long)i * f_stride;
      // ---- everything this stage reads from memory, up front
      double XC[NS];
      sfor<0, NS>([&](auto RR) { XC[decltype(RR)::value] = F[fXc + decltype(RR)::value * 16 + r]; });
      sfor<0, NX>([&](auto Cc) { Pinv[decltype(Cc)::value] = F[fPinv + decltype(Cc)::value * 16 + r]; });
      double s = F[fT + r];
      const double th = F[fTh + r];
      double Cc_[NC];
      {
        const double* src = rx ? D.E + ((long)i * NX + r) * NC : D.L + ((long)i * NU + (rs_ ? ru : 0)) * NC;
        sfor<0, NC>([&](auto Kk) { Cc_[decltype(Kk)::value] = rs_ ? src[decltype(Kk)::value] : 0.0; });
      }
      if (i < N) {
        double Wc[NX];
        sfor<0, NX>([&](auto RR) { Wc[decltype(RR)::value] = F[fW + r * 16 + decltype(RR)::value]; });
        sfor<0, NX>([&](auto RR) { s = fma(-Wc[decltype(RR)::value], bc<decltype(RR)::value>(lp), s); });
      }
      __builtin_amdgcn_sched_barrier(0);
      // [dx; du] = inv(Lc)' s
      double dzu = 0.0;
      sfor<0, NS>([&](auto RR) { dzu = fma(XC[decltype(RR)::value], bc<decltype(RR)::value>(s), dzu); });
      // dl = -inv(Pi)(theta + dx)
      const double tx = th + dzu;
      double dli = 0.0;
      sfor<0, NX>([&](auto Cc) { dli = fma(-Pinv[decltype(Cc)::value], bc<decltype(Cc)::value>(tx), dli); });
      if (!rx) dli = 0.0;
      if (rs_) dz_[(long)i * NS + r] = dzu;
      if (rx) dl_[(long)i * NX + r] = dli;
      FB_STAMP_LAP(9);
      // ---- A dz and dv (:329-341) through the LDS copy of C
      c.sync();
      sfor<0, NC>([&](auto Kk) { Cl[r * CS + decltype(Kk)::value] = Cc_[decltype(Kk)::value]; });
      c.sync();
      double dvs[KS];
      sfor<0, KS>([&](auto S_) {
        constexpr int sl = decltype(S_)::value;
        const int k = r + 16 * sl;
        const int kk = k < NC ? k : 0;
        double a = 0.0;
        sfor<0, NS>([&](auto Cc) { a = fma(Cl[decltype(Cc)::value * CS + kk], bc<decltype(Cc)::value>(dzu), a); });
        double d = 0.0;
        if (k < NC) {
          const long g = (long)i * NC + k;
          d = rvm_[g] + gam_[g] * a;
          adz_[g] = a;
          dv_[g] = d;
          // first line-search trial, v block (full_residual.cc:68-71, :99-106)
          const double vi = v_[g] + d;
          const double yi = y_[g] - a;
          const double ys = yi + sigma * (vi - vb_[g]);
          const double ph = pfb(ys, vi, alpha);
          const double pn = pnr(yi, vi, alpha);
          s_in = fma(ph, ph, s_in);
          s_out = fma(pn, pn, s_out);
        }
        dvs[sl] = d;
      });
      // ---- wz = H dz + G'dl + A'dv
      double w = 0.0;
      {
        double Hr[NS];
        if (rx) {
          const double* q = D.Q + (long)i * NX * NX + r;
          const double* st = D.S + (long)i * NU * NX + (long)r * NU;
          sfor<0, NX>([&](auto Cc) { Hr[decltype(Cc)::value] = q[decltype(Cc)::value * NX]; });
          sfor<NX, NS>([&](auto Cc) { Hr[decltype(Cc)::value] = st[decltype(Cc)::value - NX]; });
        } else if (rs_) {
          const double* sr = D.S + (long)i * NU * NX + ru;
          const double* rr = D.R + (long)i * NU * NU + ru;
          sfor<0, NX>([&](auto Cc) { Hr[decltype(Cc)::value] = sr[decltype(Cc)::value * NU]; });
          sfor<NX, NS>([&](auto Cc) { Hr[decltype(Cc)::value] = rr[(decltype(Cc)::value - NX) * NU]; });
        } else {
          sfor<0, NS>([&](auto Cc) { Hr[decltype(Cc)::value] = 0.0; });
        }
        sfor<0, NS>([&](auto Cc) { w = fma(Hr[decltype(Cc)::value], bc<decltype(Cc)::value>(dzu), w); });
      }
      sfor<0, NC>([&](auto Kk) { constexpr int k = decltype(Kk)::value; w = fma(Cc_[k], bc<(k & 15)>(dvs[k >> 4]), w); });
      w -= dli;  // zero on the input rows
      if (i < N) {
        // column r of [A B] dot dl(i+1); row r of [A B] dot dz(i)
        if (rs_) {
          const double* pc = rx ? D.A + (long)i * NX * NX + (long)r * NX : D.B + (long)i * NX * NU + (long)ru * NX;
          double Ac[NX];
          sfor<0, NX>([&](auto RR) { Ac[decltype(RR)::value] = pc[decltype(RR)::value]; });
          sfor<0, NX>([&](auto RR) { w = fma(Ac[decltype(RR)::value], bc<decltype(RR)::value>(lp), w); });
        }
        double abz = 0.0;
        {
          double AB[NS];
          sfor<0, NS>([&](auto Cc) { AB[decltype(Cc)::value] = 0.0; });
          if (rx) {
            const double* pa = D.A + (long)i * NX * NX + r;
            const double* pb = D.B + (long)i * NX * NU + r;
            sfor<0, NX>([&](auto Cc) { AB[decltype(Cc)::value] = pa[decltype(Cc)::value * NX]; });
            sfor<NX, NS>([&](auto Cc) { AB[decltype(Cc)::value] = pb[(decltype(Cc)::value - NX) * NX]; });
          }
          sfor<0, NS>([&](auto Cc) { abz = fma(AB[decltype(Cc)::value], bc<decltype(Cc)::value>(dzu), abz); });
        }
        if (rx) {
          // l block i+1: wl = -(A dx + B du - dx(i+1)); trial norms (full_residual.cc:60-66)
          const long g = (long)(i + 1) * NX + r;
          const double wlv = -(abz - dzn);
          wl_[g] = wlv;
          const double lr = rl_[g] + wlv;
          const double li = l_[g] + lp;
          const double ri = lr + sigma * (li - lb_[g]);
          s_in = fma(ri, ri, s_in);
          s_out = fma(lr, lr, s_out);
        }
      }
      if (rs_) {
        const long g = (long)i * NS + r;
        wz_[g] = w;
        const double zr = rz_[g] + w;
        const double zi = z_[g] + dzu;
        const double ri = zr + sigma * (zi - zb_[g]);
        s_in = fma(ri, ri, s_in);
        s_out = fma(zr, zr, s_out);
      }
      if (i == 0 && rx) {
        // l block 0: -(G dz)_0 = dx(0)
        wl_[r] = dzu;
        const double lr = rl_[r] + dzu;
        const double li = l_[r] + dli;
        const double ri = lr + sigma * (li - lb_[r]);
        s_in = fma(ri, ri, s_in);
        s_out = fma(lr, lr, s_out);
      }
      lp = dli;
      dzn = rx ? dzu : 0.0;
      FB_STAMP_LAP(10);
    }
    *trial_inner2 = row_reduce<OpSum16>(s_in);
    *trial_outer2 = row_reduce<OpSum16>(s_out);
    return true;
  }
};

#endif  // !FB_HOSTSIM

}  // namespace fbk

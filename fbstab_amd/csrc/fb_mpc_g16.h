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
  // Number of the wavefront's four rows for which the (row-uniform) predicate
  // holds; wavefront-uniform.
  static FB_DEV int rows_where(bool pred) {
    return __builtin_popcountll(__ballot(pred) & 0x0001000100010001ull);
  }
};

// Pins the emitted instruction order at this point.  The 64-bit DPP move has a
// result latency of ~17 cycles while a wave can issue one FP64/DPP instruction
// every ~6.5 (tools/probes/bcast_probe.hip): a broadcast must be issued a few
// instructions ahead of the FMA that consumes it, or the pair costs 2.4x its
// issue slots.  The compiler's scheduler places them back to back, so the hot
// products below spell the order out and fence it.
#define FB_SB() __builtin_amdgcn_sched_barrier(0)

#ifndef FB_BC_AHEAD
#define FB_BC_AHEAD 4
#endif
constexpr int kBcAhead = FB_BC_AHEAD;  // broadcasts in flight ahead of their consumers

// Runs consume(I, mov(I)) for I in [0, CNT) with the mov of I + kBcAhead issued
// before the consumer of I.
template <int CNT, class Mov, class Use>
FB_DEV void bc_pipeline(Mov&& mov, Use&& use) {
  if constexpr (CNT > 0) {
    double t[CNT];
    constexpr int P = CNT < kBcAhead ? CNT : kBcAhead;
    sfor<0, P>([&](auto I) {
      t[decltype(I)::value] = mov(I);
      FB_SB();
    });
    sfor<0, CNT>([&](auto I) {
      constexpr int i = decltype(I)::value;
      if constexpr (i + P < CNT) {
        t[i + P] = mov(std::integral_constant<int, i + P>{});
        FB_SB();
      }
      use(I, t[i]);
      FB_SB();
    });
  }
}

// 1/sqrt(d) to full double precision: v_rsq_f64 seed (~2^-24 relative) and one
// third-order step r(1 + e/2 + 3e^2/8), e = 1 - d r^2 (error ~e^3): four
// dependent levels instead of the six of two Newton steps.  Split into stages
// so that a caller can put independent work between the dependent levels.
struct RsqrtChain {
  double d, r, e, p, re, q;
  template <int S>
  FB_DEV void stage() {
    if constexpr (S == 0) r = __builtin_amdgcn_rsq(d);
    if constexpr (S == 1) e = -d * r;
    if constexpr (S == 2) e = fma(e, r, 1.0);
    if constexpr (S == 3) { p = fma(0.375, e, 0.5); re = r * e; }
    if constexpr (S == 4) q = fma(re, p, r);
  }
  static constexpr int kStages = 5;
};
FB_DEV double rsqrt_full(double d) {
  RsqrtChain c;
  c.d = d;
  sfor<0, RsqrtChain::kStages>([&](auto S) { c.template stage<decltype(S)::value>(); });
  return c.q;
}

// In-place Cholesky of an N x N SPD matrix held one row per lane (a[c] = A[r][c],
// lower triangle meaningful).  On return a[k] = L[r][k] for k < r and the
// diagonal slot a[r] holds 1/L[r][r] (only the reciprocal is ever needed).
// Returns false (row-uniform) on a non-positive pivot.
//
// Pivot j + 1's dependent chain (broadcast, rsqrt levels, scaling) is issued one
// level at a time between the rank-1 update instructions of pivot j, which do
// not depend on it: the chain's latency is covered instead of exposed.
template <int N>
FB_DEV bool chol_rows(double (&a)[N], int r, double diag_add) {
  bool ok = true;
  RsqrtChain ch;
  double lj, nlj;
  // chain levels of pivot J: 0 = pivot broadcast, 1..5 = rsqrt, 6 = scale column J
  constexpr int kLevels = RsqrtChain::kStages + 2;
  auto level = [&](auto J, auto S) {
    constexpr int j = decltype(J)::value;
    constexpr int lv = decltype(S)::value;
    if constexpr (lv == 0) {
      // the caller's "+ diag_add * I" is applied here, at pivot time: no per-lane
      // (r == j) selects, which the compiler would otherwise hoist and keep live
      ch.d = bc<j>(a[j]) + diag_add;
      ok = ok && (ch.d > 0.0);
    } else if constexpr (lv <= RsqrtChain::kStages) {
      ch.template stage<lv - 1>();
    } else {
      lj = a[j] * ch.q;  // L[r][j] for r > j
      nlj = -lj;
      a[j] = (r == j) ? ch.q : lj;
    }
    FB_SB();
  };
  sfor<0, kLevels>([&](auto S) { level(std::integral_constant<int, 0>{}, S); });
  sfor<0, N>([&](auto J) {
    constexpr int j = decltype(J)::value;
    constexpr int cnt = N - j - 1;
    const double ljj = lj, nljj = nlj;  // this pivot's column (lj is rewritten by level 6)
    // column j + 1 first: the next pivot's chain hangs on it
    bc_pipeline<cnt>(
        [&](auto I) { return bc<j + 1 + decltype(I)::value>(ljj); },
        [&](auto I, double t) {
          constexpr int i = decltype(I)::value;
          a[j + 1 + i] = fma(nljj, t, a[j + 1 + i]);
          if constexpr (i < kLevels) {
            FB_SB();
            level(std::integral_constant<int, j + 1>{}, I);
          }
        });
    if constexpr (j + 1 < N) {
      sfor<(cnt < kLevels ? cnt : kLevels), kLevels>(
          [&](auto S) { level(std::integral_constant<int, j + 1>{}, S); });
    }
  });
  return ok;
}

// Column r of inv(L) for the row-held factor of chol_rows (a[k] = L[r][k],
// a[r] = 1/L[r][r]).  Column-oriented: once x[k] is known it is folded into every
// later row's sum, so only one FMA and one multiply per row sit on the chain.
template <int N>
FB_DEV void tri_inv_cols(const double (&a)[N], double (&x)[N], int r) {
  sfor<0, N>([&](auto RR) { x[decltype(RR)::value] = (r == decltype(RR)::value) ? 1.0 : 0.0; });
  double dg = bc<0>(a[0]);  // 1 / L[k][k], fetched one column ahead
  sfor<0, N>([&](auto K) {
    constexpr int k = decltype(K)::value;
    if constexpr (k == 0) x[0] *= dg;
    const double nx = -x[k];
    if constexpr (k + 1 < N) dg = bc<k + 1>(a[k + 1]);
    FB_SB();
    bc_pipeline<N - k - 1>(
        [&](auto I) { return bc<k + 1 + decltype(I)::value>(a[k]); },
        [&](auto I, double t) {
          constexpr int i = decltype(I)::value;
          x[k + 1 + i] = fma(t, nx, x[k + 1 + i]);
          if constexpr (i == 0) {
            FB_SB();
            x[k + 1] *= dg;  // final: rows < k + 1 are all folded in
          }
        });
  });
}

// acc = sum_c m[c] * (lane c's v), c in [B, E): four partial sums so that the
// FMAs do not form one dependent chain.
template <int B, int E, int N>
FB_DEV double bc_dot(const double (&m)[N], double v, double init = 0.0) {
  double p[4] = {init, 0.0, 0.0, 0.0};
  bc_pipeline<E - B>([&](auto I) { return bc<B + decltype(I)::value>(v); },
                     [&](auto I, double t) {
                       constexpr int i = decltype(I)::value;
                       p[i & 3] = fma(m[B + i], t, p[i & 3]);
                     });
  return (p[0] + p[1]) + (p[2] + p[3]);
}

// out[c] = lane c's v, c in [0, N)
template <int N>
FB_DEV void bc_all(double v, double (&out)[N]) {
  sfor<0, N>([&](auto Cc) { out[decltype(Cc)::value] = bc<decltype(Cc)::value>(v); });
  FB_SB();
}
// sum_c m[c] * b[c], c in [0, N), four partial sums
template <int N, int NM, int NB>
FB_DEV double dot4(const double (&m)[NM], const double (&b)[NB], double init = 0.0) {
  double p[4] = {init, 0.0, 0.0, 0.0};
  sfor<0, N>([&](auto Cc) {
    constexpr int c = decltype(Cc)::value;
    p[c & 3] = fma(m[c], b[c], p[c & 3]);
  });
  return (p[0] + p[1]) + (p[2] + p[3]);
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
  // Lane-major copy of the stage matrices ("pack", built once per QP by
  // load_guess): slot pairs interleaved so that one 16-byte load per lane
  // fetches two slots and every load instruction of a row reads 256 contiguous
  // bytes.  Element (slot s, lane r) of stage i lives at
  //   pack + i * kPackStride + (s >> 1) * 32 + r * 2 + (s & 1).
  // Lanes and stages that have no such row/column hold zeros, so the sweeps load
  // without predicates.  (Reading E/L columns straight from the MatrixSequence
  // image costs 64 cache lines per load instruction.)
  static constexpr int pK = 0;                  // row r of [Q S'; S R], 16 slots
  static constexpr int pC = 16;                 // column r of C = [E L], NC slots
  static constexpr int pABr = pC + ((NC + 1) & ~1);  // row r of [A B], 16 slots
  static constexpr int pABc = pABr + 16;        // column r of [A B], NX slots
  static constexpr int kPackSlots = (pABc + NX + 1) & ~1;
  static constexpr int kPackStride = 16 * kPackSlots;
  typedef double dbl2 __attribute__((ext_vector_type(2)));
  // slots [S0, S0 + CNT) of the pack record `pk` (already offset by lane) into out[0..CNT)
  template <int S0, int CNT, int NOUT>
  static FB_DEV void pack_load(const double* pk, double (&out)[NOUT]) {
    static_assert((S0 & 1) == 0, "slot ranges start on a pair");
    sfor<0, (CNT + 1) / 2>([&](auto P_) {
      constexpr int pr = decltype(P_)::value;
      const dbl2 t = *reinterpret_cast<const dbl2*>(pk + (S0 / 2 + pr) * 32);
      out[2 * pr] = t[0];
      if constexpr (2 * pr + 1 < CNT) out[2 * pr + 1] = t[1];
    });
  }
  double* pack = nullptr;

  // x <- caller's guess, y = b - A z (generic), then the lane-major data copy.
  FB_DEV void load_guess(const C& c) {
    MpcProblem<C>::load_guess(c);
    const int r = c.tid;
    const MpcData D = this->D;
    const int N = this->lay.N;
    const bool rx = r < NX, rs_ = r < NS;
    const int ru = r - NX;
    pack = this->ws + this->lay.v_pack;
    for (int i = 0; i <= N; i++) {
      double* pk = pack + (long)i * kPackStride + r * 2;
      auto put = [&](int slot, double val) { pk[(slot >> 1) * 32 + (slot & 1)] = val; };
      for (int cc = 0; cc < 16; cc++) {
        double kv = 0.0;
        if (rx && cc < NX) kv = D.Q[(long)i * NX * NX + r + cc * NX];
        else if (rx && cc < NS) kv = D.S[(long)i * NU * NX + (long)r * NU + (cc - NX)];
        else if (rs_ && cc < NX) kv = D.S[(long)i * NU * NX + ru + cc * NU];
        else if (rs_ && cc < NS) kv = D.R[(long)i * NU * NU + ru + (cc - NX) * NU];
        put(pK + cc, kv);
        double ab = 0.0;
        if (rx && i < N && cc < NX) ab = D.A[(long)i * NX * NX + r + cc * NX];
        else if (rx && i < N && cc < NS) ab = D.B[(long)i * NX * NU + r + (cc - NX) * NX];
        put(pABr + cc, ab);
      }
      for (int k = 0; k < ((NC + 1) & ~1); k++) {
        double cv = 0.0;
        if (k < NC && rx) cv = D.E[((long)i * NX + r) * NC + k];
        else if (k < NC && rs_) cv = D.L[((long)i * NU + ru) * NC + k];
        put(pC + k, cv);
      }
      for (int j = 0; j < kPackSlots - pABc; j++) {
        double av = 0.0;
        if (j < NX && i < N && rx) av = D.A[(long)i * NX * NX + (long)r * NX + j];
        else if (j < NX && i < N && rs_) av = D.B[(long)i * NX * NU + (long)ru * NX + j];
        put(pABc + j, av);
      }
    }
    c.sync();
  }

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
    const double* const pack_ = this->pack + r * 2;
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
      // column r of C = [E L] and row r of [Q S'; S R] from the lane-major copy
      const double* pk = pack_ + (long)i * kPackStride;
      pack_load<pC, NC>(pk, in.Cc);
      pack_load<pK, NS>(pk, in.K);
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
      bc_pipeline<NC>([&](auto I) { return bc<(decltype(I)::value & 15)>(Gam[decltype(I)::value >> 4]); },
                      [&](auto I, double gk) { Gc[r * CS + decltype(I)::value] = gk * Cc_[decltype(I)::value]; });
      {
        double p[4] = {r1, 0.0, 0.0, 0.0};
        bc_pipeline<NC>([&](auto I) { return bc<(decltype(I)::value & 15)>(Rvm[decltype(I)::value >> 4]); },
                        [&](auto I, double rk) {
                          constexpr int k = decltype(I)::value;
                          p[k & 3] = fma(-Cc_[k], rk, p[k & 3]);
                        });
        r1 = (p[0] + p[1]) + (p[2] + p[3]);
      }
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
        const double hsum = bc_dot<0, NX>(Pinv, th);
        if (rx) gv = r1 - hsum;
      }
      sfor<0, NX>([&](auto Cc) { F[fPinv + decltype(Cc)::value * 16 + r] = Pinv[decltype(Cc)::value]; });
      F[fTh + r] = th;
      __builtin_amdgcn_sched_barrier(0);
      FB_STAMP_LAP(2);
      // [A B] row r for W, requested now so that it arrives behind the chains
      double AB[NS];
      pack_load<pABr, NS>(pack_ + (long)i * kPackStride, AB);
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
      const double tvec = bc_dot<0, NS>(XR, gv);
      F[fT + r] = tvec;
      FB_STAMP_LAP(5);
      if (i < N) {
        __builtin_amdgcn_sched_barrier(0);
        // ---- W = [A B] inv(Lc)'  (AM and -P of :149-175)
        double W[NS];
        sfor<0, NS>([&](auto Cc) { W[decltype(Cc)::value] = 0.0; });
        sfor<0, NS>([&](auto Kk) {
          constexpr int k = decltype(Kk)::value;
          // X[cc][k] = lane cc's XR[k], nonzero for cc >= k
          bc_pipeline<NS - k>([&](auto I) { return bc<k + decltype(I)::value>(XR[k]); },
                              [&](auto I, double t) { W[k + decltype(I)::value] = fma(AB[k], t, W[k + decltype(I)::value]); });
        });
        sfor<0, NS>([&](auto Cc) { F[fW + decltype(Cc)::value * 16 + r] = W[decltype(Cc)::value]; });
        // next stage's inputs: in flight during the second chain below
        load_fwd(i + 1, cur);
        __builtin_amdgcn_sched_barrier(0);
        FB_STAMP_LAP(6);
        // theta(i+1) partial = -W t
        thp = -bc_dot<0, NS>(W, tvec);
        // ---- Pi(i+1) = sigma I + W W' ; L = chol ; inv(Pi) = T'T, T = inv(L)
        double Pn[NX];
        sfor<0, NX>([&](auto Cc) { Pn[decltype(Cc)::value] = 0.0; });
        sfor<0, NS>([&](auto Kk) {
          constexpr int k = decltype(Kk)::value;
          bc_pipeline<NX>([&](auto I) { return bc<decltype(I)::value>(W[k]); },
                          [&](auto I, double t) { Pn[decltype(I)::value] = fma(W[k], t, Pn[decltype(I)::value]); });
        });
        sfor<0, NX>([&](auto Cc) { Pn[decltype(Cc)::value] = rx ? Pn[decltype(Cc)::value] : 0.0; });
        __builtin_amdgcn_sched_barrier(0);
        FB_STAMP_LAP(7);
        ok = chol_rows<NX>(Pn, ro, sigma) && ok;
        if (!ok) return false;
        __builtin_amdgcn_sched_barrier(0);
        double T[NX];
        tri_inv_cols<NX>(Pn, T, ro);
        __builtin_amdgcn_sched_barrier(0);
        // inv(Pi)[r][cc] = sum_k T[k][r] T[k][cc], T[k][cc] = lane cc's T[k] (zero for k < cc)
        sfor<0, NX>([&](auto Cc) { Pinv[decltype(Cc)::value] = 0.0; });
        sfor<0, NX>([&](auto Kk) {
          constexpr int k = decltype(Kk)::value;
          // T[k][cc] = lane cc's T[k], nonzero for cc <= k
          bc_pipeline<k + 1>([&](auto I) { return bc<decltype(I)::value>(T[k]); },
                             [&](auto I, double t) { Pinv[decltype(I)::value] = fma(T[k], t, Pinv[decltype(I)::value]); });
        });
        sfor<0, NX>([&](auto Cc) { Pinv[decltype(Cc)::value] = rx ? Pinv[decltype(Cc)::value] : 0.0; });
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
      const double* pk = pack_ + (long)i * kPackStride;
      double Cc_[NC];
      pack_load<pC, NC>(pk, Cc_);
      double lpb[NX];  // dl(i+1), every lane
      bc_all<NX>(lp, lpb);
      if (i < N) {
        double Wc[NX];
        sfor<0, NX>([&](auto RR) { Wc[decltype(RR)::value] = F[fW + r * 16 + decltype(RR)::value]; });
        s -= dot4<NX>(Wc, lpb);
      }
      __builtin_amdgcn_sched_barrier(0);
      // [dx; du] = inv(Lc)' s
      const double dzu = bc_dot<0, NS>(XC, s);
      // dl = -inv(Pi)(theta + dx)
      const double tx = th + dzu;
      double dli = -bc_dot<0, NX>(Pinv, tx);
      if (!rx) dli = 0.0;
      double dzb[NS];  // [dx; du](i), every lane
      bc_all<NS>(dzu, dzb);
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
        double clk[NS];
        sfor<0, NS>([&](auto Cc) { clk[decltype(Cc)::value] = Cl[decltype(Cc)::value * CS + kk]; });
        const double a = dot4<NS>(clk, dzb);
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
        pack_load<pK, NS>(pk, Hr);
        w = dot4<NS>(Hr, dzb);
      }
      {
        double p[4] = {w, 0.0, 0.0, 0.0};
        bc_pipeline<NC>([&](auto I) { return bc<(decltype(I)::value & 15)>(dvs[decltype(I)::value >> 4]); },
                        [&](auto I, double t) {
                          constexpr int k = decltype(I)::value;
                          p[k & 3] = fma(Cc_[k], t, p[k & 3]);
                        });
        w = (p[0] + p[1]) + (p[2] + p[3]);
      }
      w -= dli;  // zero on the input rows
      if (i < N) {
        // column r of [A B] dot dl(i+1); row r of [A B] dot dz(i)
        {
          double Ac[NX];
          pack_load<pABc, NX>(pk, Ac);
          w += dot4<NX>(Ac, lpb);  // zero rows where there is no column
        }
        double abz = 0.0;
        {
          double AB[NS];
          pack_load<pABr, NS>(pk, AB);
          abz = dot4<NS>(AB, dzb);
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

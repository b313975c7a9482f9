// DPP building blocks for solvers that keep one QP per 16-lane row of a wavefront
// (four QPs per 64-wide wavefront, lane r of a row owning row r of every stage
// matrix): row broadcasts and reductions, the software-pipelined broadcast-FMA
// product, Cholesky and triangular inverse of a row-held matrix.
//
// The stage recursion of RiccatiLinearSolver (riccati_linear_solver.cc:125-206)
// is used by the record kernel (fb_mpc_r16.h) in its equivalent block form.  With
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
#pragma once

#include <type_traits>
#include <utility>

#include "fb_common.h"

namespace fbk {


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

// ---- QPs spanning R 16-lane rows of the wavefront --------------------------------
// R = 1: one QP per DPP row (stage width nx + nu <= 16).  R = 2: an even/odd row
// pair per QP (width <= 32).  DPP reaches the lanes of a row only; the other row of
// the pair comes through v_permlane16_swap_b32 (gfx950), which for (x, x) returns
// the even rows' values replicated into both rows of each pair and the odd rows'
// likewise (tools/probes/permlane_probe.hip prints the map).  A value that is
// broadcast from many lanes is "spread" once - two swaps for a double - and every
// broadcast from it is then the same single v_mov_b64_dpp as for R = 1.
template <int R>
struct Spread;
template <>
struct Spread<1> {
  double v;
};
template <>
struct Spread<2> {
  double lo, hi;  // the even row's and the odd row's values, each present in both rows
};
template <int R>
FB_DEV Spread<R> spread(double x) {
  if constexpr (R == 1) {
    return Spread<1>{x};
  } else {
    const unsigned xl = __double2loint(x), xh = __double2hiint(x);
    const auto a = __builtin_amdgcn_permlane16_swap(xl, xl, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(xh, xh, false, false);
    return Spread<2>{__hiloint2double(b[0], a[0]), __hiloint2double(b[1], a[1])};
  }
}
// value of lane J of this lane's QP (16 R lanes)
template <int R, int J>
FB_DEV double bcs(const Spread<R>& s) {
  if constexpr (R == 1) return bc<J>(s.v);
  else if constexpr (J < 16) return bc<J>(s.lo);
  else return bc<J - 16>(s.hi);
}
template <int R, int J>
FB_DEV double bcr(double x) { return bcs<R, J>(spread<R>(x)); }
template <int R, int J>
FB_DEV int bcri(int x) {
  if constexpr (R == 1) {
    return bci<J>(x);
  } else {
    const auto a = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);
    return J < 16 ? bci<(J & 15)>((int)a[0]) : bci<(J & 15)>((int)a[1]);
  }
}
// all-lanes reduction over the QP: every lane gets the same bits
template <int R, class Op>
FB_DEV double qp_reduce(double x) {
  x = row_reduce<Op>(x);
  if constexpr (R == 2) {
    const Spread<2> s = spread<2>(x);
    x = Op::apply(s.lo, s.hi);
  }
  return x;
}

// Thread context of one QP's lanes (a "virtual workgroup" of 16 R threads).
template <int R>
struct CtxRow {
  int tid;  // lane within the QP
  static constexpr int nt = 16 * R;
  static constexpr int rows = R;
  FB_DEV void sync() const {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
  }
  template <int K>
  FB_DEV void sum(double (&v)[K]) const {
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = qp_reduce<R, OpSum16>(v[k]);
  }
  template <int K>
  FB_DEV void max(double (&v)[K]) const {
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = qp_reduce<R, OpMax16>(v[k]);
  }
};
typedef CtxRow<1> Ctx16;

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

// ---- the broadcast fused into the FMA (round 4) -------------------------------------------
// v_fmac_f64_dpp acc, src row_newbcast:J, mult  =  acc += (lane J of this row's src) * mult: the
// pair v_mov_b64_dpp + v_fma_f64 of the streams above as ONE instruction, bitwise the same
// result.  In the forward stage's arithmetic core (tools/probes/halfrow_probe.hip) it is
// 18 % faster than the hand-pipelined pairs - no temporaries, no 17-cycle move-to-use distance
// to schedule round - where the isolated instruction had measured 10 % (round 3).  The 64-bit DPP
// forms exist for row_newbcast alone: a row PAIR (R = 2) broadcasts from the two halves of a
// spread value, each an ordinary register.  Inline assembly is outside the compiler's hazard
// recognizer, so the two rules a DPP operand brings are kept by hand:
//   * two wait states between the VALU instruction that writes `src` and the first reader.  All
//     members of a group read the same src, so only the group's FIRST instruction can be too
//     close; a scheduling barrier behind every member pins the group's order (a token operand
//     would too - and makes the compiler put an s_nop between any two members: it treats a
//     register an asm statement defines as a possible partial write).  Whether the first
//     instruction carries an s_nop 1: FB_FMAC_GUARD_NOP and GUARD below;
//   * five wait states behind a VALU write of EXEC: the only such writes are the v_cmpx of the
//     hand-written LDS image blocks, which end with s_nop 4.
// tools/check_dpp_hazards.py verifies both - and the trans-forwarding rule - on the disassembly of
// every built record object (part of `make`).
#ifndef FB_FMAC_DPP
#define FB_FMAC_DPP 1
#endif
#ifndef FB_FMAC_DPP_R2
#define FB_FMAC_DPP_R2 1  // row pairs too: the spread halves are ordinary registers, each its own group
#endif
// (NEG: acc -= ...: the source modifier of the multiplier, no instruction of its own)
// FB_FMAC_GUARD_NOP = 1: the first instruction of every group carries the s_nop 1 itself - safe by
// construction, 155 s_nop per forward stage, 6 % of the arithmetic core's time
// (tools/probes/halfrow_probe.hip -DPROBE_NO_NOP).  0 (the product build): no s_nop - in the code as it
// is compiled the producer of a group's source is never among the two instructions before the group
// (the pivot's scaling is followed by the negation and the diagonal select, a product's operands come
// from an earlier phase), and tools/check_dpp_hazards.py PROVES that on the disassembly of the built
// objects, every path into every fused instruction; `make` runs it and fails on a finding.
#ifndef FB_FMAC_GUARD_NOP
#define FB_FMAC_GUARD_NOP 0
#endif
// GUARD: the groups that keep their s_nop in every build - the source of a broadcast dot product
// is as a rule the value just computed (theta + r, t - inv(Lc) u, ...), and a spread source comes
// straight out of its v_permlane16_swap.
template <int J, bool FIRST, bool NEG = false, bool GUARD = false>
FB_DEV void fmac_bc(double& acc, double src, double mult) {
  constexpr bool NOP = FIRST && (GUARD || FB_FMAC_GUARD_NOP != 0);
  if constexpr (NOP && NEG)
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
        : "+v"(acc) : "v"(src), "v"(mult), "n"(J));
  else if constexpr (NOP)
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
        : "+v"(acc) : "v"(src), "v"(mult), "n"(J));
  else if constexpr (NEG)
    asm("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mult), "n"(J));
  else
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mult), "n"(J));
  FB_SB();
}
template <int R>
constexpr bool kFmacDpp = FB_FMAC_DPP != 0 && (R == 1 || FB_FMAC_DPP_R2 != 0);
// The same for a spread source: lane J of the QP is lane J of the value itself (R = 1), or lane
// J & 15 of the even row's copy (J < 16) / of the odd row's (J >= 16), which spread() left in
// both rows of the pair.  J0: the lane the group starts at - the first reader of each copy is
// the one that waits (FIRST_OK = false: a second accumulator fed by the same lane, never first).
template <int R, int J, int J0, bool NEG = false, bool FIRST_OK = true, bool GUARD = false>
FB_DEV void fmac_bcs(double& acc, const Spread<R>& s, double mult) {
  if constexpr (R == 1) {
    fmac_bc<J, FIRST_OK && J == J0, NEG, GUARD>(acc, s.v, mult);
  } else {
    constexpr bool first = FIRST_OK && (J == J0 || (J == 16 && J0 < 16));
    if constexpr (J < 16) fmac_bc<J, first, NEG, true>(acc, s.lo, mult);
    else fmac_bc<J - 16, first, NEG, true>(acc, s.hi, mult);
  }
}
#ifndef FB_FMAC_DPP_DOT
#define FB_FMAC_DPP_DOT 1   // the broadcast dot products (bc_dot) too
#endif
#ifndef FB_FMAC_DPP_SOLVE
#define FB_FMAC_DPP_SOLVE 1 // tri_inv_cols_solve, where one broadcast feeds two FMAs (measured: +1.3 % on top)
#endif

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
template <int N, int R = 1>
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
      // (the pivot's broadcast fused as diag_add + 1.0 * pivot - only v_fmac_f64 has a DPP form among
      // the f64 instructions - changed nothing: 580 k against 583 k, gpurun of round 4)
      ch.d = bcr<R, j>(a[j]) + diag_add;
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
    const double nljj = nlj;  // this pivot's column (lj is rewritten by level 6)
    if constexpr (kFmacDpp<R>) {
      const Spread<R> src = spread<R>(lj);
      sfor<0, cnt>([&](auto I) {
        constexpr int i = decltype(I)::value;
        fmac_bcs<R, j + 1 + i, j + 1>(a[j + 1 + i], src, nljj);
        if constexpr (i < kLevels) level(std::integral_constant<int, j + 1>{}, I);
      });
    } else {
    const Spread<R> ljs = spread<R>(lj);
    // column j + 1 first: the next pivot's chain hangs on it
    bc_pipeline<cnt>(
        [&](auto I) { return bcs<R, j + 1 + decltype(I)::value>(ljs); },
        [&](auto I, double t) {
          constexpr int i = decltype(I)::value;
          a[j + 1 + i] = fma(nljj, t, a[j + 1 + i]);
          if constexpr (i < kLevels) {
            FB_SB();
            level(std::integral_constant<int, j + 1>{}, I);
          }
        });
    }
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
template <int N, int R = 1>
FB_DEV void tri_inv_cols(const double (&a)[N], double (&x)[N], int r) {
  sfor<0, N>([&](auto RR) { x[decltype(RR)::value] = (r == decltype(RR)::value) ? 1.0 : 0.0; });
  double dg = bcr<R, 0>(a[0]);  // 1 / L[k][k], fetched one column ahead
  sfor<0, N>([&](auto K) {
    constexpr int k = decltype(K)::value;
    if constexpr (k == 0) x[0] *= dg;
    const double nx = -x[k];
    if constexpr (k + 1 < N) dg = bcr<R, k + 1>(a[k + 1]);
    if constexpr (kFmacDpp<R>) {
      const Spread<R> aks = spread<R>(a[k]);
      FB_SB();
      sfor<0, N - k - 1>([&](auto I) {
        constexpr int i = decltype(I)::value;
        fmac_bcs<R, k + 1 + i, k + 1>(x[k + 1 + i], aks, nx);
        if constexpr (i == 0) {
          x[k + 1] *= dg;  // final: rows < k + 1 are all folded in
          FB_SB();
        }
      });
    } else {
    const Spread<R> aks = spread<R>(a[k]);
    FB_SB();
    bc_pipeline<N - k - 1>(
        [&](auto I) { return bcs<R, k + 1 + decltype(I)::value>(aks); },
        [&](auto I, double t) {
          constexpr int i = decltype(I)::value;
          x[k + 1 + i] = fma(t, nx, x[k + 1 + i]);
          if constexpr (i == 0) {
            FB_SB();
            x[k + 1] *= dg;  // final: rows < k + 1 are all folded in
          }
        });
    }
  });
}

// tri_inv_cols fused with the right-solve W Lc' = B for a second row-held matrix:
// on entry w[c] = B[r][c], on return w[c] = (B inv(Lc)')[r][c] (forward
// substitution along the row, W[r][m] = (B[r][m] - sum_{k<m} W[r][k] L[m][k]) / L[m][m]).
// Both recurrences consume the same broadcasts L[m][k], so the product
// B inv(Lc)' costs one more FMA per broadcast instead of a broadcast-FMA pair of
// its own per element.
template <int N, int R = 1>
FB_DEV void tri_inv_cols_solve(const double (&a)[N], double (&x)[N], double (&w)[N], int r) {
  sfor<0, N>([&](auto RR) { x[decltype(RR)::value] = (r == decltype(RR)::value) ? 1.0 : 0.0; });
  double dg = bcr<R, 0>(a[0]);  // 1 / L[k][k], fetched one column ahead
  sfor<0, N>([&](auto K) {
    constexpr int k = decltype(K)::value;
    if constexpr (k == 0) {
      x[0] *= dg;
      w[0] *= dg;
    }
    const double nx = -x[k], nw = -w[k];
    if constexpr (k + 1 < N) dg = bcr<R, k + 1>(a[k + 1]);
    if constexpr (kFmacDpp<R> && FB_FMAC_DPP_SOLVE != 0) {
      const Spread<R> aks = spread<R>(a[k]);
      FB_SB();
      sfor<0, N - k - 1>([&](auto I) {
        constexpr int i = decltype(I)::value;
        fmac_bcs<R, k + 1 + i, k + 1>(x[k + 1 + i], aks, nx);
        fmac_bcs<R, k + 1 + i, k + 1, false, false>(w[k + 1 + i], aks, nw);
        if constexpr (i == 0) {
          x[k + 1] *= dg;  // final: rows < k + 1 are all folded in
          w[k + 1] *= dg;
          FB_SB();
        }
      });
      return;
    }
    const Spread<R> aks = spread<R>(a[k]);
    FB_SB();
    bc_pipeline<N - k - 1>(
        [&](auto I) { return bcs<R, k + 1 + decltype(I)::value>(aks); },
        [&](auto I, double t) {
          constexpr int i = decltype(I)::value;
          x[k + 1 + i] = fma(t, nx, x[k + 1 + i]);
          w[k + 1 + i] = fma(t, nw, w[k + 1 + i]);
          if constexpr (i == 0) {
            FB_SB();
            x[k + 1] *= dg;  // final: rows < k + 1 are all folded in
            w[k + 1] *= dg;
          }
        });
  });
}

// ---- chol_rows and tri_inv_cols[_solve] as ONE pass (round 6) ----------------------------------------
// Step k of the column-oriented inverse needs column k of L and the reciprocal of pivot k + 1 - exactly
// what pivot k of the factorisation has just produced.  Run together, pivot j's three streams
//     a[m] -= L[m][j] L[r][j],   x[m] -= L[m][j] x[j],   w[m] -= L[m][j] w[j]        (m = j + 1 .. N - 1)
// read the SAME broadcast source (lane m's L[m][j]), the next pivot's reciprocal square root is in every
// lane when x[j + 1], w[j + 1] want it, and what the two-pass form spends on handing the factor over goes:
// the diagonal select that parks 1 / L[j][j] in lane j's a[j] (three instructions per pivot), the
// broadcast that fetches it back (one), and - with the negation as the multiplier's source modifier - the
// copies -lj, -x[k], -w[k] (two each).  Same operations on the same values in the same order per
// accumulator: bitwise the results of chol_rows + tri_inv_cols_solve.  On return a[k] = L[r][k] for k < r
// (the diagonal slot is NOT the reciprocal here: nothing reads it), x = column r of inv(L), w = row r of
// B inv(L)'.  The next pivot's dependent chain is spread over three times as many independent
// instructions as in chol_rows.
#ifndef FB_CHOL_FUSED
#define FB_CHOL_FUSED 1
#endif
// WITH_X / WITH_W: which of the two riders of the factorisation run (the inverse's column, the right-solve).
// PARK: lane j's a[j] receives 1 / L[j][j] as chol_rows leaves it - for callers that go on to SUBSTITUTE with
// the rows of the factor (the row-pair instances: subst_rows reads the reciprocal out of the row).
template <int N, int R, bool WITH_W, bool WITH_X = true, bool PARK = false>
FB_DEV bool chol_inv_fused_impl(double (&a)[N], double (&x)[WITH_X ? N : 1], double (&w)[WITH_W ? N : 1], int r, double diag_add) {
  static_assert(kFmacDpp<R>, "the fused pass is written for the fused broadcast-FMA");
  static_assert(WITH_X || WITH_W, "nothing rides along: that is chol_rows");
  bool ok = true;
  RsqrtChain ch;
  double lj = 0.0;
  if constexpr (WITH_X) sfor<0, N>([&](auto RR) { x[decltype(RR)::value] = (r == decltype(RR)::value) ? 1.0 : 0.0; });
  constexpr int kLevels = RsqrtChain::kStages + 2;
  auto level = [&](auto J, auto S) {
    constexpr int j = decltype(J)::value;
    constexpr int lv = decltype(S)::value;
    if constexpr (lv == 0) {
      ch.d = bcr<R, j>(a[j]) + diag_add;  // (chol_rows: the caller's "+ diag_add I" at pivot time)
      ok = ok && (ch.d > 0.0);
    } else if constexpr (lv <= RsqrtChain::kStages) {
      ch.template stage<lv - 1>();
    } else {
      lj = a[j] * ch.q;  // L[r][j] for r > j: the streams' broadcast source - written first, two
      FB_SB();           // instructions ahead of its first reader at the least (the multiplies below)
      if constexpr (WITH_X) x[j] *= ch.q;  // final: rows < j are all folded in
      if constexpr (WITH_W) w[j] *= ch.q;
      if constexpr (PARK) a[j] = (r == j) ? ch.q : lj;
    }
    FB_SB();
  };
  // (where the source's multiply stands directly in front of its first reader: with both riders two multiplies
  // follow it; with one, one more wait state; PARK on one row per QP - the select that follows is the compiler's
  // to place - two.  Row pairs guard every group themselves, fmac_bcs.)
  constexpr bool kShortTail = !(WITH_W && WITH_X) && !PARK;
  constexpr bool kParkTail = PARK && R == 1;
  sfor<0, kLevels>([&](auto S) { level(std::integral_constant<int, 0>{}, S); });
  if constexpr (kShortTail) asm volatile("s_nop 0");
  if constexpr (kParkTail) asm volatile("s_nop 1");
  sfor<0, N>([&](auto J) {
    constexpr int j = decltype(J)::value;
    constexpr int cnt = N - j - 1;
    const double ljj = lj;  // this pivot's values (level 6 of the next pivot rewrites lj)
    double xj = 0.0, wj = 0.0;
    if constexpr (WITH_X) xj = x[j];
    if constexpr (WITH_W) wj = w[j];
    const Spread<R> src = spread<R>(ljj);
    sfor<0, cnt>([&](auto I) {
      constexpr int i = decltype(I)::value;
      fmac_bcs<R, j + 1 + i, j + 1, true>(a[j + 1 + i], src, ljj);
      if constexpr (WITH_X) fmac_bcs<R, j + 1 + i, j + 1, true, false>(x[j + 1 + i], src, xj);
      if constexpr (WITH_W) fmac_bcs<R, j + 1 + i, j + 1, true, false>(w[j + 1 + i], src, wj);
      if constexpr (i < kLevels) level(std::integral_constant<int, j + 1>{}, I);
    });
    if constexpr (j + 1 < N) {
      sfor<(cnt < kLevels ? cnt : kLevels), kLevels>(
          [&](auto S) { level(std::integral_constant<int, j + 1>{}, S); });
      // (the source's multiply directly in front of its first reader: one more wait state)
      if constexpr (cnt <= kLevels && kShortTail) asm volatile("s_nop 0");
      if constexpr (cnt <= kLevels && kParkTail) asm volatile("s_nop 1");
    }
  });
  return ok;
}
template <int N, int R = 1>
FB_DEV bool chol_inv_cols_solve(double (&a)[N], double (&x)[N], double (&w)[N], int r, double diag_add) {
  return chol_inv_fused_impl<N, R, true>(a, x, w, r, diag_add);
}
template <int N, int R = 1>
FB_DEV bool chol_inv_cols(double (&a)[N], double (&x)[N], int r, double diag_add) {
  double none[1] = {0.0};
  return chol_inv_fused_impl<N, R, false>(a, x, none, r, diag_add);
}
// chol_rows + tri_solve_right as one pass: a <- the factor as chol_rows leaves it (reciprocal diagonal parked),
// w <- W inv(Lc)' (the row-pair instances, which substitute with the factor's rows afterwards)
template <int N, int R = 1>
FB_DEV bool chol_solve_right(double (&a)[N], double (&w)[N], int r, double diag_add) {
  double none[1] = {0.0};
  return chol_inv_fused_impl<N, R, true, false, true>(a, none, w, r, diag_add);
}

// ---- substitution with the row-held factor (the reference's solveInPlace; round 5) ----------------------
// The sweeps of the one-row instances multiply with an explicitly inverted Lc (tri_inv_cols): every
// product is an independent stream of broadcast-FMAs, but the result is forward stable only - the
// residual of Lc x = b grows with cond(Lc), where a substitution's stays at eps |Lc| |x| whatever the
// conditioning (what RiccatiLinearSolver::Solve relies on, riccati_linear_solver.cc:234-325).  On the
// 16-wide stages that difference stays two orders under the tolerance; on wide stages whose Pi_i keeps
// eigenvalues of order sigma (nx > N nu) it reached 4.6e-6 and cost a QP an iteration.  The
// row-pair instances therefore substitute: a chain of N dependent steps, each one multiply (the solved
// entry), one broadcast-FMA into the entries still open.
//
// W <- W inv(Lc)' alone (tri_inv_cols_solve without the inverse): w[c] = B[r][c] on entry.
template <int N, int R = 1>
FB_DEV void tri_solve_right(const double (&a)[N], double (&w)[N], int r) {
  (void)r;
  double dg = bcr<R, 0>(a[0]);  // 1 / L[k][k], fetched one column ahead
  sfor<0, N>([&](auto K) {
    constexpr int k = decltype(K)::value;
    if constexpr (k == 0) w[0] *= dg;
    const double nw = -w[k];
    if constexpr (k + 1 < N) dg = bcr<R, k + 1>(a[k + 1]);
    const Spread<R> aks = spread<R>(a[k]);
    FB_SB();
    sfor<0, N - k - 1>([&](auto I) {
      constexpr int i = decltype(I)::value;
      if constexpr (kFmacDpp<R>) {
        fmac_bcs<R, k + 1 + i, k + 1>(w[k + 1 + i], aks, nw);
      } else {
        w[k + 1 + i] = fma(bcs<R, k + 1 + i>(aks), nw, w[k + 1 + i]);
      }
      if constexpr (i == 0) {
        w[k + 1] *= dg;  // final: columns < k + 1 are all folded in
        FB_SB();
      }
    });
  });
}
// x = inv(L) b for the factor of chol_rows held by ROWS (lane r: a[k] = L[r][k] for k < r, a[r] = 1 / L[r][r];
// entries beyond the diagonal are ignored), b and x one entry per lane.  Column-oriented: once x_k is
// final (lane k) it is folded into every later lane's sum.
template <int N, int R = 1>
FB_DEV double subst_rows(const double (&a)[N], double b, int r) {
  double dinv = 0.0;
  sfor<0, N>([&](auto J) { dinv = (r == decltype(J)::value) ? a[decltype(J)::value] : dinv; });
  double acc = b;
  sfor<0, N - 1>([&](auto K) {
    constexpr int k = decltype(K)::value;
    const double xk = acc * dinv;                 // lane k: x_k
    const double m = (k < r) ? a[k] : 0.0;        // L[r][k] on the lanes below row k
    if constexpr (kFmacDpp<R>) {
      fmac_bcs<R, k, k, true, true, true>(acc, spread<R>(xk), m);
    } else {
      acc = fma(-m, bcr<R, k>(xk), acc);
    }
  });
  return acc * dinv;
}
// x = inv(L)' b for the factor held by COLUMNS (lane r: a[j] = L[j][r] for j > r, a[r] = 1 / L[r][r];
// entries before the diagonal are ignored): the same chain from the last row upwards.
template <int N, int R = 1>
FB_DEV double subst_cols_t(const double (&a)[N], double b, int r) {
  double dinv = 0.0;
  sfor<0, N>([&](auto J) { dinv = (r == decltype(J)::value) ? a[decltype(J)::value] : dinv; });
  double acc = b;
  sfor<0, N - 1>([&](auto KK) {
    constexpr int k = N - 1 - decltype(KK)::value;
    const double xk = acc * dinv;                 // lane k: x_k
    const double m = (k > r) ? a[k] : 0.0;        // L[k][r] on the lanes above row k
    if constexpr (kFmacDpp<R>) {
      fmac_bcs<R, k, k, true, true, true>(acc, spread<R>(xk), m);
    } else {
      acc = fma(-m, bcr<R, k>(xk), acc);
    }
  });
  return acc * dinv;
}

// acc = sum_c m[c] * (lane c's v), c in [B, E): four partial sums so that the
// FMAs do not form one dependent chain.
template <int B, int E, int R = 1, int N>
FB_DEV double bc_dot(const double (&m)[N], double v, double init = 0.0) {
  double p[4] = {init, 0.0, 0.0, 0.0};
  if constexpr (kFmacDpp<R> && FB_FMAC_DPP_DOT != 0) {
    const Spread<R> vs = spread<R>(v);
    sfor<0, E - B>([&](auto I) {
      constexpr int i = decltype(I)::value;
      fmac_bcs<R, B + i, B, false, true, true>(p[i & 3], vs, m[B + i]);
    });
    return (p[0] + p[1]) + (p[2] + p[3]);
  }
  const Spread<R> vs = spread<R>(v);
  bc_pipeline<E - B>([&](auto I) { return bcs<R, B + decltype(I)::value>(vs); },
                     [&](auto I, double t) {
                       constexpr int i = decltype(I)::value;
                       p[i & 3] = fma(m[B + i], t, p[i & 3]);
                     });
  return (p[0] + p[1]) + (p[2] + p[3]);
}

// p[k & 3] += (NEG: -=) C[k] * (lane k % (16 R) of src[k / (16 R)]), k in [0, NC): a column of the
// constraint matrix against a vector that lives one entry per lane (v, dv, rv / mu, ...), four
// partial sums as in bc_dot.
template <int NC, int R, bool NEG = false, int NSRC>
FB_DEV void bc_cols_dot(const double (&C)[NC], const double (&src)[NSRC], double (&p)[4]) {
  constexpr int LPQ = 16 * R;
  if constexpr (kFmacDpp<R> && FB_FMAC_DPP_DOT != 0) {
    sfor<0, (NC + LPQ - 1) / LPQ>([&](auto S_) {
      constexpr int sl = decltype(S_)::value;
      const Spread<R> ss = spread<R>(src[sl]);
      sfor<LPQ * sl, (LPQ * (sl + 1) < NC ? LPQ * (sl + 1) : NC)>([&](auto Kk) {
        constexpr int k = decltype(Kk)::value;
        fmac_bcs<R, k % LPQ, 0, NEG, true, true>(p[k & 3], ss, C[k]);
      });
    });
  } else {
    bc_pipeline<NC>([&](auto I) { return bcr<R, (decltype(I)::value % LPQ)>(src[decltype(I)::value / LPQ]); },
                    [&](auto I, double t) {
                      constexpr int k = decltype(I)::value;
                      p[k & 3] = fma(NEG ? -C[k] : C[k], t, p[k & 3]);
                    });
  }
}

// out[c] = lane c's v, c in [0, N)
template <int N, int R = 1>
FB_DEV void bc_all(double v, double (&out)[N]) {
  const Spread<R> vs = spread<R>(v);
  sfor<0, N>([&](auto Cc) { out[decltype(Cc)::value] = bcs<R, decltype(Cc)::value>(vs); });
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


}  // namespace fbk

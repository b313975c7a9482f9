// Shared device-side building blocks of the MI355X batched FBstab kernels:
// the per-workgroup thread context (barrier + workgroup reductions), the
// penalised Fischer-Burmeister function and its generalised gradient.
//
// The kernels are written against the small `Ctx` abstraction below instead of
// raw threadIdx/__syncthreads, with the thread count as a template parameter: the
// flat-vector solver logic is therefore also well-formed for a workgroup of ONE
// thread, which is what tests/hostsim compiles with g++ against its own stand-in
// for <hip/hip_runtime.h> (tests/hostsim/shim) so that the CPU test-suite can step
// through the kernels' arithmetic where no GPU exists.  That build is a debugging
// aid owned by tests/ and everything host-specific lives there; the product
// library contains the gfx950 code only and has no CPU execution path.
#pragma once

#include <math.h>
#include <stdint.h>

#include "../../include/fbstab_types.h"

#include <hip/hip_runtime.h>
#define FB_DEV __device__ __forceinline__
// Explicit LDS address space: every access through an lds_ptr is a ds_* op.
#define FB_LDS __attribute__((address_space(3)))

namespace fbk {

typedef FB_LDS double* lds_ptr;

// In-kernel phase timing for diagnostic builds (-DFB_STAMP): per-phase shader
// cycles summed over all waves into g_stamps (read with
// fbstab_hip_debug_stamps).  Production builds compile the macro away.
#if defined(FB_STAMP)
extern __device__ unsigned long long g_stamps[32];
struct StampClock {
  unsigned long long t0;
  __device__ __forceinline__ void start() { t0 = now(); }
  static __device__ __forceinline__ unsigned long long now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
  }
  __device__ __forceinline__ void lap(int k) {
    const unsigned long long t = now();
    if ((threadIdx.x & 63) == 0) atomicAdd(&g_stamps[k], t - t0);
    t0 = now();
  }
};
#define FB_STAMP_DECL StampClock fb_clk_; fb_clk_.start()
#define FB_STAMP_LAP(k) fb_clk_.lap(k)
#define FB_STAMP_COUNT(k) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_stamps[k], 1ull); } while (0)
#elif defined(FB_PHASE_MARKERS)
// tools/isa_ledger.py: the phase boundaries as comments in the assembly (no instruction;
// the comment stays where the boundary is because the asm statement is volatile)
#define FB_STAMP_DECL
#define FB_STAMP_LAP(k) asm volatile("; FBPHASE " #k)
#define FB_STAMP_COUNT(k)
#else
#define FB_STAMP_DECL
#define FB_STAMP_LAP(k)
#define FB_STAMP_COUNT(k)
#endif

#if defined(FB_PHASE_MARKERS)
#define FB_PHASE(name) asm volatile("; FBPHASE " #name)
#else
#define FB_PHASE(name)
#endif

// Wave-level event counters of the light diagnostic build (-DFB_CLOCKSTAMP).
#if defined(FB_STAMP) || defined(FB_CLOCKSTAMP)
#if !defined(FB_STAMP)
extern __device__ unsigned long long g_stamps[32];
#endif
#define FB_WAVE_COUNT(k) do { if ((threadIdx.x & 63) == __builtin_ctzll(__builtin_amdgcn_read_exec())) atomicAdd(&g_stamps[k], 1ull); } while (0)
// Shader cycles the wavefront spends inside a scope (first active lane reports).
struct WaveTimer {
  int k;
  long long t0;
  __device__ __forceinline__ explicit WaveTimer(int k_) : k(k_), t0(__builtin_readcyclecounter()) {}
  __device__ __forceinline__ ~WaveTimer() {
    const long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == __builtin_ctzll(__builtin_amdgcn_read_exec()))
      atomicAdd(&g_stamps[k], (unsigned long long)(t1 - t0));
  }
};
#define FB_WAVE_TIMER(k) WaveTimer fb_wave_timer_##k(k)
struct WaveLap {
  long long t0;
  __device__ __forceinline__ WaveLap() : t0(__builtin_readcyclecounter()) {}
  __device__ __forceinline__ void lap(int k) {
    const long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == __builtin_ctzll(__builtin_amdgcn_read_exec()))
      atomicAdd(&g_stamps[k], (unsigned long long)(t1 - t0));
    t0 = __builtin_readcyclecounter();
  }
};
#define FB_WAVE_LAP_DECL WaveLap fb_wave_lap_
#define FB_WAVE_LAP(k) fb_wave_lap_.lap(k)
#else
#define FB_WAVE_LAP_DECL
#define FB_WAVE_LAP(k)
#define FB_WAVE_COUNT(k)
#define FB_WAVE_TIMER(k)
#endif

// Maximum number of values reduced together by one block_reduce call.
constexpr int kMaxReduce = 12;

struct OpSum {
  static FB_DEV double apply(double a, double b) { return a + b; }
};
struct OpMax {
  static FB_DEV double apply(double a, double b) { return a > b ? a : b; }
};

// Per-thread view of the workgroup.  NT threads cooperate on one QP.
template <int NT>
struct Ctx {
  int tid;
  lds_ptr red;  // LDS scratch, kMaxReduce * (NT/64) doubles (unused if NT<=64)

  static constexpr int nt = NT;

  // Workgroup barrier.  A single-wavefront workgroup only needs its LDS
  // traffic ordered (lanes run in lockstep and the wave's global-memory
  // operations stay in program order), so it waits on lgkmcnt alone and leaves
  // global loads/stores in flight; __syncthreads() would drain vmcnt too.
  FB_DEV void sync() const {
#if !defined(FB_NO_WAVE_SYNC)
    if (NT <= 64) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
      return;
    }
#endif
    __syncthreads();
  }

  // In-place reduction of K values over the NT threads; every thread gets the
  // result (bitwise identical in all threads, so branches on it are uniform).
  template <class Op, int K>
  FB_DEV void reduce(double (&v)[K]) const {
    static_assert(K <= kMaxReduce, "too many values");
    constexpr int kLanes = NT < 64 ? NT : 64;  // lanes of the (first) wavefront that hold a value
#pragma unroll
    for (int m = kLanes / 2; m >= 1; m >>= 1) {
#pragma unroll
      for (int k = 0; k < K; k++) v[k] = Op::apply(v[k], __shfl_xor(v[k], m, 64));
    }
    if (NT > 64) {
      const int wave = tid >> 6;
      const int nw = NT / 64;
      sync();  // previous users of `red` are done
      if ((tid & 63) == 0) {
#pragma unroll
        for (int k = 0; k < K; k++) red[k * nw + wave] = v[k];
      }
      sync();
#pragma unroll
      for (int k = 0; k < K; k++) {
        double s = red[k * nw];
        for (int w = 1; w < nw; w++) s = Op::apply(s, red[k * nw + w]);
        v[k] = s;
      }
    }
  }
  // argmax with ties resolved to the SMALLEST index (Eigen's maxCoeff(&index)
  // returns the first maximum): one pass over (value, index) pairs.
  // (value, index) of the better of two candidates: larger value, then smaller index
  static FB_DEV void better(double& v, int& ix, double ov, int oi) {
    const bool take = (ov > v) || (ov == v && oi < ix);
    v = take ? ov : v;
    ix = take ? oi : ix;
  }
  template <int CTRL>
  static FB_DEV void better_dpp(double& v, int& ix) {
    const double ov = __builtin_amdgcn_update_dpp(0.0, v, CTRL, 0xf, 0xf, true);
    const int oi = __builtin_amdgcn_update_dpp(0, ix, CTRL, 0xf, 0xf, true);
    better(v, ix, ov, oi);
  }
  // Wave-wide argmax without LDS shuffles: rotations inside the 16-lane rows
  // (DPP), then the four row results through v_readlane.  Every lane returns
  // the same pair.
  static FB_DEV void wave_argmax_first(double& v, int& ix) {
    better_dpp<0x128>(v, ix);  // row_ror:8
    better_dpp<0x124>(v, ix);  // row_ror:4
    better_dpp<0x122>(v, ix);  // row_ror:2
    better_dpp<0x121>(v, ix);  // row_ror:1
    double bv = v;
    int bi = ix;
#pragma unroll
    for (int row = 0; row < 4; row++) {
      const double ov = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 16 * row),
                                         __builtin_amdgcn_readlane(__double2loint(v), 16 * row));
      const int oi = __builtin_amdgcn_readlane(ix, 16 * row);
      if (row == 0) { bv = ov; bi = oi; } else better(bv, bi, ov, oi);
    }
    v = bv;
    ix = bi;
  }
  FB_DEV void argmax_first(double* val, int* idx) const {
    double v = *val;
    int ix = *idx;
    wave_argmax_first(v, ix);
    if (NT > 64) {
      const int wave = tid >> 6;
      const int nw = NT / 64;
      sync();
      if ((tid & 63) == 0) {
        red[wave] = v;
        red[nw + wave] = (double)ix;
      }
      sync();
      v = red[0];
      ix = (int)red[nw];
      for (int w = 1; w < nw; w++) {
        const double ov = red[w];
        const int oi = (int)red[nw + w];
        const bool take = (ov > v) || (ov == v && oi < ix);
        v = take ? ov : v;
        ix = take ? oi : ix;
      }
    }
    *val = v;
    *idx = ix;
  }
  template <int K>
  FB_DEV void sum(double (&v)[K]) const { reduce<OpSum, K>(v); }
  template <int K>
  FB_DEV void max(double (&v)[K]) const { reduce<OpMax, K>(v); }
};

FB_DEV double fmax0(double a) { return a > 0.0 ? a : 0.0; }

// tools::saturate for lo <= hi (tools/utilities.h:19-28); callers test lo > hi first.
FB_DEV double sat(double x, double lo, double hi) {
  const double t = x < hi ? x : hi;
  return t > lo ? t : lo;
}

// sqrt(x) for x >= 0 inside the Fischer-Burmeister function, where half of the
// passes' instructions used to be the IEEE sqrt sequence (two residual steps,
// range scaling and special-case selects around v_rsq_f64): hardware seed, one
// coupled Goldschmidt step and ONE residual correction - within an ulp - and no
// scaling (a^2 + b^2 has underflowed or overflowed long before the scaling would
// matter).  Zero, infinity and NaN come back as they are (sqrt(0) = 0 exactly; an
// overflowed or NaN iterate then propagates as it does through the reference's sqrt and
// ends where the reference's solve ends - in a failed factorisation, impl:263-267).
// PRECONDITION x >= 0 (or NaN): every caller passes a sum of squares.  A negative argument
// is NOT turned into NaN - it comes back as it is, like the other inputs outside the
// +normal / +denormal classes; a caller that can produce one must test for it itself.
FB_DEV double fsqrt(double x) {
  const double r = __builtin_amdgcn_rsq(x);
  double g = x * r, h = 0.5 * r;
  const double d = fma(-h, g, 0.5);
  g = fma(g, d, g);
  h = fma(h, d, h);
  const double e = fma(-g, g, x);
  g = fma(e, h, g);
  return __builtin_amdgcn_class(x, 0x180 /* +denormal | +normal */) ? g : x;
}

// phi(a,b) = alpha (a + b - sqrt(a^2+b^2)) + (1-alpha) max(0,a) max(0,b)
// (reference: full_residual.cc:115-118).
FB_DEV double pfb(double a, double b, double alpha) {
  const double fb = a + b - fsqrt(a * a + b * b);
  return alpha * fb + (1.0 - alpha) * fmax0(a) * fmax0(b);
}

// alpha*min(y,v) + (1-alpha) max(0,y) max(0,v): the v-block of the penalised
// natural residual (reference: full_residual.cc:93-105).
FB_DEV double pnr(double y, double v, double alpha) {
  const double m = y < v ? y : v;
  return alpha * m + (1.0 - alpha) * fmax0(y) * fmax0(v);
}

// Generalised gradient of phi (reference: riccati_linear_solver.cc:346-365 and
// dense_cholesky_solver.cc:129-148, zero_tolerance_ = 1e-13).
FB_DEV void pfb_gradient(double a, double b, double alpha, double* g0, double* g1) {
  const double r = fsqrt(a * a + b * b);
  const double d = 0.70710678118654752440;  // 1/sqrt(2)
  if (r < 1e-13) {
    *g0 = alpha * (1.0 - d);
    *g1 = alpha * (1.0 - d);
  } else if (a > 0.0 && b > 0.0) {
    *g0 = alpha * (1.0 - a / r) + (1.0 - alpha) * b;
    *g1 = alpha * (1.0 - b / r) + (1.0 - alpha) * a;
  } else {
    *g0 = alpha * (1.0 - a / r);
    *g1 = alpha * (1.0 - b / r);
  }
}

// 1/x to (almost) full double precision without the IEEE division sequence:
// hardware reciprocal seed + two Newton steps.  Used where the reference
// divides but parity is by tolerance (never on a value that feeds a branch
// directly).
FB_DEV double rcp_fast(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

// phi(a,b) and its generalised gradient from ONE sqrt and ONE reciprocal
// (pfb + pfb_gradient share r = sqrt(a^2+b^2)); same formulas as above.
FB_DEV void pfb_all(double a, double b, double alpha, double* phi, double* g0, double* g1) {
  const double r = fsqrt(a * a + b * b);
  const double pa = fmax0(a), pb = fmax0(b);
  *phi = alpha * (a + b - r) + (1.0 - alpha) * pa * pb;
  if (r < 1e-13) {
    const double d = 0.70710678118654752440;
    *g0 = alpha * (1.0 - d);
    *g1 = alpha * (1.0 - d);
  } else {
    const double ir = rcp_fast(r);
    const bool both = a > 0.0 && b > 0.0;
    *g0 = alpha * (1.0 - a * ir) + (both ? (1.0 - alpha) * b : 0.0);
    *g1 = alpha * (1.0 - b * ir) + (both ? (1.0 - alpha) * a : 0.0);
  }
}

// Result of the infeasibility test (reference: full_feasibility.h enum).
enum Feasibility { kFeasible = 0, kPrimalInfeasible = 1, kDualInfeasible = 2, kBothInfeasible = 3 };

}  // namespace fbk

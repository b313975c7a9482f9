// TEST INFRASTRUCTURE ONLY - the host twin of the device environment, for tests/hostsim.
//
// tests/hostsim compiles the flat-vector solver logic of fbstab_amd/csrc (fb_common.h, fb_algorithm.h,
// fb_mpc.h, fb_dense.h) as ordinary single-threaded C++ with g++ so that the CPU test-suite can step
// through the kernels' arithmetic where no GPU exists.  Those headers include <hip/hip_runtime.h>; the
// host build puts THIS directory first on its include path, and everything that differs between the
// device and a one-thread host lives here and nowhere else: the product headers carry no host branch.
//
// The model is one workgroup of ONE thread (Ctx<1>): lane exchanges return the caller's own value,
// barriers and fences are nothing, a ballot is the caller's predicate.  The code paths that need a
// real wavefront (DPP pivot searches, one-wavefront LDL', MFMA assembly, the record kernels' row state
// machine) are templates on the thread count and are never instantiated for Ctx<1>; the stubs below
// only have to let them parse.  The product library is built by hipcc against the real header and has
// no CPU execution path.
#pragma once

#include <math.h>
#include <stdint.h>
#include <string.h>

#define FB_HOST_TWIN 1

// ---- function and variable qualifiers ------------------------------------------------------------------
#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)
// clang's short-vector attribute, for the parser only (the code that uses it needs a wavefront and is not
// instantiated here): every such vector in the solver headers is one of doubles
#define ext_vector_type(n) vector_size((n) * sizeof(double))

// ---- the one thread ------------------------------------------------------------------------------------
struct fb_host_dim3 {
  unsigned x, y, z;
};
static const fb_host_dim3 threadIdx = {0, 0, 0}, blockIdx = {0, 0, 0}, blockDim = {1, 1, 1};

// ---- barriers, fences, scheduling hints: nothing to order ------------------------------------------------
#define __syncthreads() ((void)0)
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_wave_barrier() ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_read_exec() (1ull)

// ---- lane exchanges of a one-lane wavefront ----------------------------------------------------------------
template <class T>
inline T fb_host_self(T v) {
  return v;
}
#define __shfl_xor(v, m, w) fb_host_self(v)
#define __builtin_amdgcn_readlane(v, lane) fb_host_self(v)
#define __builtin_amdgcn_readfirstlane(v) fb_host_self(v)
#define __builtin_amdgcn_update_dpp(old, v, ctrl, row_mask, bank_mask, bound_ctrl) fb_host_self(v)
inline unsigned long long __ballot(bool p) { return p ? 1ull : 0ull; }

template <class T, class U>
inline T atomicAdd(T* p, U v) {
  const T old = *p;
  *p = old + (T)v;
  return old;
}

// ---- bit casts ---------------------------------------------------------------------------------------------
inline double __hiloint2double(int hi, int lo) {
  const uint64_t u = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
  double d;
  memcpy(&d, &u, 8);
  return d;
}
inline int __double2hiint(double d) {
  uint64_t u;
  memcpy(&u, &d, 8);
  return (int)(uint32_t)(u >> 32);
}
inline int __double2loint(double d) {
  uint64_t u;
  memcpy(&u, &d, 8);
  return (int)(uint32_t)u;
}
inline double __longlong_as_double(long long x) {
  double d;
  memcpy(&d, &x, 8);
  return d;
}
inline long long __double_as_longlong(double d) {
  long long x;
  memcpy(&x, &d, 8);
  return x;
}

// ---- transcendental seeds: the host has the exact operations, and hands them out as the "seed" - the
// refinement steps the device code runs on top of a seed leave an exact value where it is (to an ulp) ------
inline double fb_host_rsq(double x) { return 1.0 / sqrt(x); }
inline double fb_host_rcp(double x) { return 1.0 / x; }
#define __builtin_amdgcn_rsq(x) fb_host_rsq(x)
#define __builtin_amdgcn_rcp(x) fb_host_rcp(x)
// v_cmp_class_f64: only the classes the solver asks about (0x180 = +denormal | +normal)
#define __builtin_amdgcn_class(x, mask) fb_host_class(x, mask)
inline bool fb_host_class(double x, int mask) {
  bool r = false;
  if (mask & 0x100) r = r || (x >= 2.2250738585072014e-308 && x <= 1.7976931348623157e308);  // +normal
  if (mask & 0x080) r = r || (x > 0.0 && x < 2.2250738585072014e-308);                       // +denormal
  return r;
}

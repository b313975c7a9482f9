// The record kernels' entry point and queue (fb_mpc_r16.h holds the numerics), shared by
// the translation units of the library: every instance <NX, NU, NC, R> is compiled in a
// file of its own (rec_*.hip, a minute or two each, in parallel under `make -j`) and
// hands fbstab_hip.hip a RecordInstance with the addresses of its six kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/fbstab_hip.h"
#include "fb_algorithm.h"
#include "fb_mpc_r16.h"

#if defined(FB_STAMP) || defined(FB_CLOCKSTAMP)
#define FB_ANY_STAMP 1
#endif

// One compiled instance of the record kernel family: its entry points (batch,
// FBSTAB_HIP_KEEP_MATRICES, Newton-step probe) and its footprint.
struct RecordInstance {
  const char* name;
  int nx, nu, nc;        // largest problem it runs (smaller ones zero-padded)
  int qps_per_wg;
  int (*lds_bytes)(int N);
  long long (*ws_doubles)(int N);
  const void *solve, *solve_keep, *probe;  // kernels of the padded instance
  const void *solve_exact, *solve_keep_exact, *probe_exact;  // problem == instance shape
};

namespace {

using namespace fbk;

// Diagnostic probe (tests only): one Newton step at (x, xbar, sigma) instead of
// a solve.  dbg holds [zb, lb, vb] on input and receives
// [dz, dl, dv, adz, wz, wl, rz, rl, ok].
template <class P, class C>
__device__ __forceinline__ void newton_probe(P& p, const C& ctx, const fbstab_options_t& opts, double* dbg) {
  p.load_guess(ctx);
  if constexpr (P::kOwnVectorOps) {
    p.choose_costate_form(opts.sigma0);
    p.probe_set_xbar(ctx, dbg);
    p.residual(ctx);
    double a, b, lin2;
    const bool ok = p.newton_step(ctx, opts.sigma0, opts.alpha, &a, &b, &lin2);
    // the solver's own rule for refining a step (Solver::wants_refinement: an option, off by default), no
    // inner tolerance in play
    Solver<P, C> rule(p, ctx, opts);
    if (ok && rule.wants_refinement(lin2, opts.abs_tol, opts.abs_tol))
      p.refine_step(ctx, opts.sigma0, opts.alpha, &a, &b, &lin2);
    ctx.sync();
    p.probe_dump(ctx, dbg, ok);
    return;
  } else {
  const int nz = p.nz, nl = p.nl, nv = p.nv;
  for (int i = ctx.tid; i < nz; i += C::nt) p.zb[i] = dbg[i];
  for (int i = ctx.tid; i < nl; i += C::nt) p.lb[i] = dbg[nz + i];
  for (int i = ctx.tid; i < nv; i += C::nt) p.vb[i] = dbg[nz + nl + i];
  ctx.sync();
  p.residual(ctx);
  bool ok;
  if constexpr (P::kFusedTrial) {
    double a, b;
    ok = p.newton_step(ctx, opts.sigma0, opts.alpha, &a, &b);
  } else {
    ok = p.newton_step(ctx, opts.sigma0, opts.alpha);
    if constexpr (can_refine_of<P>::value) {  // the solver's own rule (Solver::wants_refinement)
      Solver<P, C> rule(p, ctx, opts);
      if (ok && opts.reserved > 0 && rule.wants_refinement(p.linear_residual2(ctx, opts.sigma0), opts.abs_tol, opts.abs_tol))
        p.refine_step(ctx, opts.sigma0);
    }
  }
  ctx.sync();
  double* o = dbg;
  for (int i = ctx.tid; i < nz; i += C::nt) o[i] = p.dz[i];
  o += nz;
  for (int i = ctx.tid; i < nl; i += C::nt) o[i] = p.dl[i];
  o += nl;
  for (int i = ctx.tid; i < nv; i += C::nt) o[i] = p.dv[i];
  o += nv;
  for (int i = ctx.tid; i < nv; i += C::nt) o[i] = p.adz[i];
  o += nv;
  for (int i = ctx.tid; i < nz; i += C::nt) o[i] = p.wz[i];
  o += nz;
  for (int i = ctx.tid; i < nl; i += C::nt) o[i] = p.wl[i];
  o += nl;
  for (int i = ctx.tid; i < nz; i += C::nt) o[i] = p.rz[i];
  o += nz;
  for (int i = ctx.tid; i < nl; i += C::nt) o[i] = p.rl[i];
  o += nl;
  if (ctx.tid == 0) o[0] = ok ? 1.0 : 0.0;
  }
}

// Record-based 16-lane kernel (fb_mpc_r16.h): four QPs per wavefront, rows pull
// QP indices from the shared counter.  scratch: rows * ws_doubles(N).
// One queue object per 16-lane row; every function is called by the whole row and
// returns row-uniform values.  Nothing in here waits for another wavefront.
constexpr size_t kQueueBytes = 8 * sizeof(int);

// The receding-horizon sweep as ONE launch of a KEEP instance (fbstab_hip_mpc_receding_sweep):
// a row then solves ITS trajectory `steps` times, advancing the plant in between, and
// never waits for another trajectory - a batch launch per step lasts as long as its
// slowest QP, and a trajectory that runs to the iteration limit before it is retired
// holds up the other 4095 for a hundred solves' worth of time.  Lives in device memory;
// the kernel gets the pointer through its (otherwise unused) probe argument.
struct SweepArgs {
  const double* A;  // simulation model x+ = A x + B u0 (ocp_generator.h:31-38), column-major
  const double* B;
  long long sA, sB;     // doubles between trajectories (0: one plant for all)
  double* x0;           // the batch's initial states, advanced in place
  long long sx0;
  double* u_log;        // NULL or [steps][batch][nu]
  unsigned long long* stats;  // [steps][4]
  int steps, retire;
  int nx, nu, nz, nl, nv;
};

// Closed-loop step of trajectory q after its solve number `step`, by the lanes of its
// row (t = lane within the row, lpq = lanes per row): retirement, statistics, u0 and
// x0 <- A x0 + B u0 - what fbstab_receding_plant_kernel does for a whole batch between
// two launches.  Returns the updated `retired` flag.  A real call: inlined into the
// solver loop its temporaries cost the sweeps 60 spilled registers.
__device__ __noinline__ bool receding_plant_step(const SweepArgs* sweep, const VarBatchPtrs* x,
                                                 const fbstab_solver_out_t* out, int batch, int q, int step, int t,
                                                 int lpq, bool gone) {
  const SweepArgs& a = *sweep;
  // the solve's own stores (solution, SolverOut) are read back by other lanes
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const int eflag = out[q].eflag, newton = out[q].newton_iters;
  double* z = x->base[0] + q * x->stride[0];
  if (a.retire && !gone && eflag != FBSTAB_SUCCESS) {
    gone = true;
    double* l = x->base[1] + q * x->stride[1];
    double* v = x->base[2] + q * x->stride[2];
    for (int i = t; i < a.nz; i += lpq) z[i] = 0.0;
    for (int i = t; i < a.nl; i += lpq) l[i] = 0.0;
    for (int i = t; i < a.nv; i += lpq) v[i] = 0.0;
  }
  if (t == 0) {
    unsigned long long* st = a.stats + 4 * (long long)step;
    atomicAdd(&st[0], (unsigned long long)newton);
    atomicAdd(&st[1], (unsigned long long)(eflag == FBSTAB_SUCCESS ? 1 : 0));
    atomicAdd(&st[2], (unsigned long long)(gone ? 1 : 0));
    atomicMax(&st[3], (unsigned long long)newton);
  }
  if (a.u_log && t < a.nu) a.u_log[((long long)step * batch + q) * a.nu + t] = gone ? 0.0 : z[a.nx + t];
  double* xs = a.x0 + q * a.sx0;
  const double* Aq = a.A + q * a.sA;
  const double* Bq = a.B + q * a.sB;
  double acc = 0.0;  // (nx <= lanes of the row: one entry per lane)
  if (t < a.nx) {
    for (int c = 0; c < a.nx; c++) acc = fma(Aq[t + c * a.nx], xs[c], acc);
    for (int j = 0; j < a.nu; j++) acc = fma(Bq[t + j * a.nx], gone ? 0.0 : z[a.nx + j], acc);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // every lane has read x0
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  if (t < a.nx) xs[t] = gone ? 0.0 : acc;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return gone;
}

template <class P, bool KEEP>
struct R16Queue {
  // Only launch-uniform values live in here (SGPRs): the sweeps have no registers
  // to spare (a handful of VGPRs held across the Newton step turned 2 spilled
  // registers into 44).
  const MpcBatchPtrs* data;
  const VarBatchPtrs* x;
  int* ctl;  // ctl[0]: next QP index
  double* scratch;
  int batch, N;
  bool reuse;
  bool taken = false;  // (KEEP) this row has had its one QP
  // (KEEP, sweep) solves of this row's trajectory finished so far; bit 30: retired
  int swept = 0;
  const SweepArgs* sweep = nullptr;
  fbstab_solver_out_t* out = nullptr;

  static __device__ __forceinline__ int tid() { return threadIdx.x & (P::LPQ - 1); }
  static __device__ __forceinline__ int row() { return threadIdx.x / P::LPQ; }  // QP slot of the wavefront
  static __device__ __forceinline__ int home() { return blockIdx.x * P::kQpPerWave + row(); }
  static __device__ __forceinline__ lds_ptr lds() {
    extern __shared__ __attribute__((aligned(16))) double smem_[];
    return (lds_ptr)smem_ + P::kPackArea + row() * P::kLdsPerRow;
  }
  // this row's image in the wavefront's matrix-copy area (in front of the rows' own regions)
  static __device__ __forceinline__ lds_ptr pack_lds() {
    extern __shared__ __attribute__((aligned(16))) double smem_[];
    return (lds_ptr)smem_ + row() * P::kPackQp;
  }
  // Twenty spare doubles of the row's LDS region: the solver loop parks its scalars
  // there while a Newton step and its line search run (Solver::solve_stream).
  static __device__ __forceinline__ lds_ptr save_area() { return lds() + P::kLdsDoubles + 4; }
  // the row's table of matrix-copy offsets, behind the four row regions
  __device__ __forceinline__ typename P::lds_iptr lpo() const {
    extern __shared__ __attribute__((aligned(16))) double smem_[];
    return (typename P::lds_iptr)((lds_ptr)smem_ + P::kPackArea + P::kQpPerWave * P::kLdsPerRow) + row() * P::lpo_ints(N);
  }
  __device__ __forceinline__ double* slot_ptr(long slot) const { return scratch + slot * P::ws_doubles(N); }
  // ctl[1]: Newton steps of this launch that were refined (fbstab_hip_mpc_refined_steps)
  __device__ __forceinline__ void count_refinement() const {
    if (tid() == 0) atomicAdd(&ctl[1], 1);
  }

  // Sweep: the rows of a wavefront start every step of their trajectories together
  // (Solver::solve_stream, kPause) - warm-started steps are mostly passes over the
  // records, which four rows out of step would run one after the other.
  static constexpr bool kCanAlignRows = KEEP;  // (the batch instances compile the plain loop)
  __device__ __forceinline__ bool align_rows() const { return sweep != nullptr; }
  // Binds the policy to the next QP of the queue, in this row's own slot.
  __device__ __forceinline__ int fetch(P& pp) {
    int q = 0;
    if constexpr (KEEP) {
      q = home();
      if (sweep) {
        if (q >= batch) return -1;
        const int done = swept & 0xffff;
        bool gone = (swept & (1 << 30)) != 0;
        if (done > 0) gone = receding_plant_step(sweep, x, out, batch, q, done - 1, tid(), P::LPQ, gone);
        if (done >= sweep->steps) return -1;
        swept = (done + 1) | (gone ? (1 << 30) : 0);
        pp.bind(slot_ptr(home()), lds(), pack_lds(), lpo(), data, x, q, N, tid());
        pp.reuse = reuse || done > 0;
        return q;
      }
      if (taken) return -1;
      taken = true;
    } else {
      // a batch that does not outnumber the launch's wavefronts: one QP per wavefront (fbstab_hip.hip sizes
      // the grid to the batch then); the other rows only lend their lanes to the cooperative passes
      if (batch <= (int)gridDim.x && row() != 0) return -1;
      if (tid() == 0) q = atomicAdd(&ctl[0], 1);
      q = bcri<P::LPQ / 16, 0>(q);
    }
    if (q >= batch) return -1;
    pp.bind(slot_ptr(home()), lds(), pack_lds(), lpo(), data, x, q, N, tid());
    if constexpr (KEEP) pp.reuse = reuse;
    return q;
  }
};

// KEEP (FBSTAB_HIP_KEEP_MATRICES): QP q is solved in slot q, so that the slot's
// matrix copies survive from call to call; `reuse` says they are valid already.
// FB_R16_REG_CAP: the register budget of the kernel, arch VGPRs + AGPRs, in units of TWO registers (the
// compiler doubles "amdgpu-num-vgpr" on the unified file of gfx90a and later, and gives a function without
// MFMA all 256 arch VGPRs first): 248 = 496 registers.  One wavefront holds its SIMD for the whole launch;
// at 496 of the SIMD's 512 registers sixteen stay free and the small kernels a caller queues between two
// solves (the fill kernel behind hipMemsetAsync / torch's zero_()) find room beside it - at 504 they wait for a
// whole launch to end, and eight launches in flight run one after the other: 630 k -> 250 k QP/s
// (LABNOTES R6.3; tools/check_vgpr_budget.py is the build's gate on the result).
#ifndef FB_R16_REG_CAP
#define FB_R16_REG_CAP 248
#endif
#if FB_R16_REG_CAP > 0
#define FB_R16_REG_ATTR __attribute__((amdgpu_num_vgpr(FB_R16_REG_CAP)))
#else
#define FB_R16_REG_ATTR
#endif
template <int NX, int NU, int NC, bool DBG, bool EXACT, bool KEEP = false, int R = 1>
__global__ __launch_bounds__(64, 1) FB_R16_REG_ATTR void fbstab_mpc_r16_kernel(
    MpcBatchPtrs data, VarBatchPtrs x, fbstab_solver_out_t* out, fbstab_options_t opts, double* scratch,
    int* counter, int batch, int N, int reuse, double* dbg) {
  typedef MpcR16<NX, NU, NC, EXACT, KEEP, R> P;
  extern __shared__ __attribute__((aligned(16))) double smem[];
#if defined(FB_ANY_STAMP)
  const long long clk0 = __builtin_readcyclecounter(), rt0 = wall_clock64();
#endif
  const int lane = threadIdx.x;
#if defined(FB_SCRATCH_PAD)
  // (diagnostic knob: FB_SCRATCH_PAD more bytes of private memory per lane, nothing else changed - the
  // runtime's handling of a dispatch depends on its scratch size, LABNOTES R6.3)
  volatile char scratch_pad_[FB_SCRATCH_PAD];
  scratch_pad_[threadIdx.x % FB_SCRATCH_PAD] = 1;
#endif
  typename P::C ctx;
  ctx.tid = lane & (P::LPQ - 1);
  P p;
  R16Queue<P, KEEP> qu;
  qu.data = &data;
  qu.x = &x;
  qu.ctl = counter;
  qu.scratch = scratch;
  qu.batch = batch;
  qu.N = N;
  qu.reuse = reuse != 0;
  if constexpr (KEEP && !DBG) {
    qu.sweep = reinterpret_cast<const SweepArgs*>(dbg);
    qu.out = out;
  }
#if !defined(FB_R16_NO_BIND_IDLE)  // (the switch exists to show what the rows' unbound policy objects did: DESIGN.md section 7)
  p.bind_idle(qu.lds(), qu.pack_lds(), qu.lpo(), N);
#endif
  if constexpr (DBG) {
    if (qu.fetch(p) >= 0) newton_probe(p, ctx, opts, dbg);
  } else {
    Solver<P, typename P::C> solver(p, ctx, opts);
    solver.solve_stream(qu, out);
  }
#if defined(FB_ANY_STAMP)
  // shader clock actually delivered to this wavefront: s_memtime vs the 100 MHz counter
  if (threadIdx.x == 0) {
    atomicAdd(&g_stamps[28], (unsigned long long)(__builtin_readcyclecounter() - clk0));
    atomicAdd(&g_stamps[29], (unsigned long long)(wall_clock64() - rt0));
  }
#endif
}

template <int NX, int NU, int NC, int R>
long long r16_ws_doubles(int N) { return fbk::MpcR16<NX, NU, NC, true, false, R>::ws_doubles(N); }
template <int NX, int NU, int NC, int R>
int r16_lds_bytes(int N) {
  typedef fbk::MpcR16<NX, NU, NC, true, false, R> P;
  return (P::kPackArea + P::kQpPerWave * P::kLdsPerRow) * (int)sizeof(double) + P::kQpPerWave * P::lpo_ints(N) * (int)sizeof(int);
}
// R: 16-lane rows of the wavefront per QP (1: four QPs per wavefront, stage width
// <= 16; 2: two QPs per wavefront, stage width <= 32)
template <int NX, int NU, int NC, int R = 1>
RecordInstance r16_instance(const char* name) {
  RecordInstance r;
  r.name = name;
  r.nx = NX; r.nu = NU; r.nc = NC;
  r.qps_per_wg = 4 / R;
  r.lds_bytes = r16_lds_bytes<NX, NU, NC, R>;
  r.ws_doubles = r16_ws_doubles<NX, NU, NC, R>;
  r.solve = reinterpret_cast<const void*>(fbstab_mpc_r16_kernel<NX, NU, NC, false, false, false, R>);
  r.solve_keep = reinterpret_cast<const void*>(fbstab_mpc_r16_kernel<NX, NU, NC, false, false, true, R>);
  r.probe = reinterpret_cast<const void*>(fbstab_mpc_r16_kernel<NX, NU, NC, true, false, false, R>);
  r.solve_exact = reinterpret_cast<const void*>(fbstab_mpc_r16_kernel<NX, NU, NC, false, true, false, R>);
  r.solve_keep_exact = reinterpret_cast<const void*>(fbstab_mpc_r16_kernel<NX, NU, NC, false, true, true, R>);
  r.probe_exact = reinterpret_cast<const void*>(fbstab_mpc_r16_kernel<NX, NU, NC, true, true, false, R>);
  return r;
}

}  // namespace

// Defines the factory of one instance (one per rec_*.hip); fbstab_hip.hip lists them.
#define FB_RECORD_INSTANCE(NX, NU, NC, R, NAME)                                              \
  __attribute__((visibility("hidden"))) RecordInstance fbstab_record_instance_##NX##_##NU##_##NC##_##R() { \
    return r16_instance<NX, NU, NC, R>(NAME);                                                \
  }
#define FB_RECORD_INSTANCE_DECL(NX, NU, NC, R) \
  __attribute__((visibility("hidden"))) RecordInstance fbstab_record_instance_##NX##_##NU##_##NC##_##R();

// libfbstab_hip.so: gfx950 kernels and the C-ABI of include/fbstab_hip.h.
//
// Execution model: a persistent grid of workgroups, one QP per workgroup at a
// time.  Workgroups pull QP indices from a device-side counter (iteration
// counts differ per QP by 5-10x, so a static assignment would leave CUs idle at
// the tail).  The whole FBstab solve of a QP (proximal loop, Newton loop,
// factorisation, line search) runs inside the kernel; the host only launches.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/fbstab_hip.h"
#include "fb_algorithm.h"
#include "fb_dense.h"
#include "fb_dense_wave.h"
#include "fb_final_norms.h"
#include "fb_mpc.h"
#include "fb_mpc_r16.h"
#include "fb_record_kernel.h"

#if defined(FB_ANY_STAMP)
namespace fbk { __device__ unsigned long long g_stamps[32]; }
#endif

namespace {

using namespace fbk;

constexpr int kMpcThreads = 64;     // one wavefront per MPC QP
constexpr int kDenseThreads = 256;  // four wavefronts per dense QP
constexpr int kLdsLimitBytes = 160 * 1024;
// word of the queue block (kQueueBytes, zeroed before every launch) in which the one-wavefront
// dense kernel counts the Newton steps it handed to the pivoted factorisation
constexpr int kDenseFallbackSlot = 4;

struct MpcBatchArgs {
  const double* base[FBSTAB_MPC_NSEQ];
  long long stride[FBSTAB_MPC_NSEQ];
};
struct DenseBatchArgs {
  const double* base[FBSTAB_DENSE_NARR];
  long long stride[FBSTAB_DENSE_NARR];
};
struct VarBatchArgs {
  double* base[4];
  long long stride[4];
};

// Next QP index for this workgroup (workgroup-uniform).
template <int NT>
__device__ __forceinline__ int next_qp(int* counter, lds_ptr slot) {
  if (NT <= 64) {
    int q = 0;
    if (threadIdx.x == 0) q = atomicAdd(counter, 1);
    return __shfl(q, 0, 64);
  } else {
    __syncthreads();
    if (threadIdx.x == 0) *((FB_LDS int*)slot) = atomicAdd(counter, 1);
    __syncthreads();
    return *((FB_LDS int*)slot);
  }
}

__device__ __forceinline__ MpcData mpc_data_of(const MpcBatchArgs& data, long q) {
  MpcData D;
  D.Q = data.base[FBSTAB_MPC_Q] + q * data.stride[FBSTAB_MPC_Q];
  D.R = data.base[FBSTAB_MPC_R] + q * data.stride[FBSTAB_MPC_R];
  D.S = data.base[FBSTAB_MPC_S] + q * data.stride[FBSTAB_MPC_S];
  D.q = data.base[FBSTAB_MPC_q] + q * data.stride[FBSTAB_MPC_q];
  D.r = data.base[FBSTAB_MPC_r] + q * data.stride[FBSTAB_MPC_r];
  D.A = data.base[FBSTAB_MPC_A] + q * data.stride[FBSTAB_MPC_A];
  D.B = data.base[FBSTAB_MPC_B] + q * data.stride[FBSTAB_MPC_B];
  D.c = data.base[FBSTAB_MPC_c] + q * data.stride[FBSTAB_MPC_c];
  D.E = data.base[FBSTAB_MPC_E] + q * data.stride[FBSTAB_MPC_E];
  D.L = data.base[FBSTAB_MPC_L] + q * data.stride[FBSTAB_MPC_L];
  D.d = data.base[FBSTAB_MPC_d] + q * data.stride[FBSTAB_MPC_d];
  D.x0 = data.base[FBSTAB_MPC_x0] + q * data.stride[FBSTAB_MPC_x0];
  return D;
}

#ifndef FB_MPC_MIN_WAVES
#define FB_MPC_MIN_WAVES 1
#endif
// DBG: the Newton-step probe; TRACE: `dbg` is the trace buffer of
// fbstab_hip_mpc_solve_traced (see Solver in fb_algorithm.h); WG: the stage tile and the
// work matrices in global scratch (MpcLayout::wglobal: shapes beyond the LDS).
template <int NT, bool DBG, bool TRACE = false, bool WG = false>
__global__ __launch_bounds__(NT, FB_MPC_MIN_WAVES) void fbstab_mpc_kernel(MpcLayout lay, MpcBatchArgs data,
                                                        VarBatchArgs x,
                                                        fbstab_solver_out_t* out,
                                                        fbstab_options_t opts, double* scratch,
                                                        int* counter, int batch, double* dbg) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  lds_ptr lds = (lds_ptr)smem;
  typedef Ctx<NT> C;
  C ctx;
  ctx.tid = threadIdx.x;
  ctx.red = WG ? lds : lds + lay.w_red;
  double* ws = scratch + (long)blockIdx.x * lay.ws_doubles;
  for (;;) {
    const int q = next_qp<NT>(counter, WG ? lds + kMaxReduce * ((NT + 63) / 64) : lds + lay.w_out);
    if (q >= batch) break;
    MpcProblem<C, WG> p;
    typename MpcProblem<C, WG>::mptr mb;
    if constexpr (WG) mb = ws + lay.v_carve;
    else mb = lds;
    p.bind(lay, mpc_data_of(data, q), x.base[0] + q * x.stride[0], x.base[1] + q * x.stride[1],
           x.base[2] + q * x.stride[2], x.base[3] + q * x.stride[3], mb, ws);
    if constexpr (DBG) {
      newton_probe(p, ctx, opts, dbg);
    } else {
      Solver<MpcProblem<C, WG>, C, TRACE> solver(p, ctx, opts, dbg);
      solver.solve(out + q);
      if (solver.refined_ > 0 && ctx.tid == 0) atomicAdd(counter + 1, solver.refined_);  // fbstab_hip_mpc_refined_steps
    }
    ctx.sync();
  }
}


// Closed-loop step of the receding-horizon sweep (fbstab_hip_mpc_receding_sweep):
// one thread per trajectory.  u0 = first input of the solution just computed,
// x0 <- A x0 + B u0 (the SimulationInputs of the reference's generator,
// fbstab/test/ocp_generator.h:31-38); (z, l, v) stay where they are and are the
// next step's initial guess, unshifted (the reference has no shift logic;
// fbstab_algorithm-impl.h:140 starts from whatever the caller's Variable holds).
// With `retire`, a trajectory whose solve did not end in SUCCESS is parked at the
// origin for the rest of the sweep (x0 = 0, guess 0), as a controller's fallback
// would: its QP is then solved by the zero vector in one proximal iteration.
// stats[step] = {sum of Newton iterations, solves that ended in SUCCESS,
// trajectories retired so far, largest Newton count}.
__global__ void fbstab_receding_plant_kernel(int batch, int nx, int nu, int nz, int nl, int nv, const double* A,
                                             long long sA, const double* B, long long sB, double* x0, long long sx0,
                                             VarBatchArgs x, const fbstab_solver_out_t* out, int* retired,
                                             int retire, double* u_log, unsigned long long* stats, double* xtmp) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = q < batch;
  fbstab_solver_out_t o;
  o.eflag = FBSTAB_SUCCESS;
  o.newton_iters = 0;
  if (live) o = out[q];
  double* z = x.base[0] + (live ? q : 0) * x.stride[0];
  double* xs = x0 + (live ? q : 0) * sx0;
  bool gone = live && retired[q] != 0;
  if (live && retire && !gone && o.eflag != FBSTAB_SUCCESS) {
    gone = true;
    retired[q] = 1;
    for (int i = 0; i < nz; i++) z[i] = 0.0;
    double* l = x.base[1] + q * x.stride[1];
    double* v = x.base[2] + q * x.stride[2];
    for (int i = 0; i < nl; i++) l[i] = 0.0;
    for (int i = 0; i < nv; i++) v[i] = 0.0;
  }
  {
    // statistics: one set of atomics per wavefront (thousands of same-address atomics
    // a step would cost more than the solve)
    int ns = live ? o.newton_iters : 0, ok = (live && o.eflag == FBSTAB_SUCCESS) ? 1 : 0, gn = gone ? 1 : 0, mx = ns;
    for (int m = 32; m >= 1; m >>= 1) {
      ns += __shfl_xor(ns, m, 64);
      ok += __shfl_xor(ok, m, 64);
      gn += __shfl_xor(gn, m, 64);
      const int om = __shfl_xor(mx, m, 64);
      mx = om > mx ? om : mx;
    }
    if ((threadIdx.x & 63) == 0) {
      atomicAdd(&stats[0], (unsigned long long)ns);
      atomicAdd(&stats[1], (unsigned long long)ok);
      atomicAdd(&stats[2], (unsigned long long)gn);
      atomicMax(&stats[3], (unsigned long long)mx);
    }
  }
  if (!live) return;
  // (u0 = z[nx .. nx + nu) is read where it is used; the new state is collected in
  // registers up to 64 states, in the caller's per-trajectory buffer beyond)
  if (u_log)
    for (int j = 0; j < nu; j++) u_log[(long long)q * nu + j] = gone ? 0.0 : z[nx + j];
  const double* Aq = A + q * sA;
  const double* Bq = B + q * sB;
  double xloc[64];
  double* xn = nx <= 64 ? xloc : xtmp + (long long)q * nx;
  for (int r = 0; r < nx; r++) {
    double acc = 0.0;
    for (int c = 0; c < nx; c++) acc = fma(Aq[r + c * nx], xs[c], acc);
    for (int j = 0; j < nu; j++) acc = fma(Bq[r + j * nx], gone ? 0.0 : z[nx + j], acc);
    xn[r] = gone ? 0.0 : acc;
  }
  for (int r = 0; r < nx; r++) xs[r] = xn[r];
}

// KGLOBAL: K in a per-workgroup global scratch (fb_dense.h); the argument is
// empty for the LDS instance, like the trace buffer for the untraced ones.
template <bool KGLOBAL>
struct KScratchArg {
  __device__ double* get() const { return nullptr; }
};
template <>
struct KScratchArg<true> {
  double* p;
  __device__ double* get() const { return p; }
};

// VGLOBAL (with KGLOBAL): the iterate vectors too live in the workgroup's global scratch
// (DenseLayout::v_global: nv beyond what the LDS holds).
template <int NT, bool TRACE = false, bool KGLOBAL = false, bool VGLOBAL = false>
__global__ __launch_bounds__(NT, (NT > 64 ? 2 : 1)) void fbstab_dense_kernel(DenseLayout lay, DenseBatchArgs data,
                                                          VarBatchArgs x,
                                                          fbstab_solver_out_t* out,
                                                          fbstab_options_t opts, int* counter,
                                                          int batch, TraceArg<TRACE> trace,
                                                          KScratchArg<KGLOBAL> kscratch) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  FB_WAVE_TIMER(28);  // total wave cycles (diagnostic builds)
  lds_ptr lds = (lds_ptr)smem;
  typedef Ctx<NT> C;
  C ctx;
  ctx.tid = threadIdx.x;
  ctx.red = lds + lay.o_red;
  for (;;) {
    const int q = next_qp<NT>(counter, lds + lay.o_slot);
    if (q >= batch) break;
    DenseData D;
    D.H = data.base[FBSTAB_DENSE_H] + q * data.stride[FBSTAB_DENSE_H];
    D.f = data.base[FBSTAB_DENSE_f] + q * data.stride[FBSTAB_DENSE_f];
    D.G = data.base[FBSTAB_DENSE_G] + q * data.stride[FBSTAB_DENSE_G];
    D.h = data.base[FBSTAB_DENSE_h] + q * data.stride[FBSTAB_DENSE_h];
    D.A = data.base[FBSTAB_DENSE_A] + q * data.stride[FBSTAB_DENSE_A];
    D.b = data.base[FBSTAB_DENSE_b] + q * data.stride[FBSTAB_DENSE_b];
    DenseProblem<C, KGLOBAL, VGLOBAL> p;
    double* ks = nullptr;
    if constexpr (KGLOBAL) ks = kscratch.get() + (long)blockIdx.x * (lay.k_doubles + lay.v_doubles);
    p.bind(lay, D, x.base[0] + q * x.stride[0], x.base[1] + q * x.stride[1],
           x.base[2] + q * x.stride[2], x.base[3] + q * x.stride[3], lds, ks);
    Solver<DenseProblem<C, KGLOBAL, VGLOBAL>, C, TRACE> solver(p, ctx, opts, trace.get());
    solver.solve(out + q);
    ctx.sync();
  }
}

// Diagnostic probe for the dense path (tests only): one Newton step at (x, xbar,
// sigma0) for QP 0 of the batch arrays, see newton_probe.
template <int NT>
__global__ __launch_bounds__(NT, (NT > 64 ? 2 : 1)) void fbstab_dense_probe_kernel(DenseLayout lay, DenseBatchArgs data,
                                                                                VarBatchArgs x,
                                                                                fbstab_options_t opts, double* dbg) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  lds_ptr lds = (lds_ptr)smem;
  typedef Ctx<NT> C;
  C ctx;
  ctx.tid = threadIdx.x;
  ctx.red = lds + lay.o_red;
  DenseData D;
  D.H = data.base[FBSTAB_DENSE_H];
  D.f = data.base[FBSTAB_DENSE_f];
  D.G = data.base[FBSTAB_DENSE_G];
  D.h = data.base[FBSTAB_DENSE_h];
  D.A = data.base[FBSTAB_DENSE_A];
  D.b = data.base[FBSTAB_DENSE_b];
  DenseProblem<C, false> p;
  p.bind(lay, D, x.base[0], x.base[1], x.base[2], x.base[3], lds, nullptr);
  newton_probe(p, ctx, opts, dbg);
}

// One wavefront per dense QP for every phase (fb_dense_wave.h; nz + nl <= 64): the KKT
// matrix in registers, two wavefronts per SIMD.  scratch: one region of
// lay.ws_doubles per workgroup (A' and the multipliers).  DBG: the Newton-step probe.
#ifndef FB_DW_MIN_WAVES
#define FB_DW_MIN_WAVES 2
#endif
template <bool DBG>
__global__ __launch_bounds__(64, FB_DW_MIN_WAVES) void fbstab_dense_wave_kernel(DenseWaveLayout lay, DenseBatchArgs data,
                                                                   VarBatchArgs x, fbstab_solver_out_t* out,
                                                                   fbstab_options_t opts, int* counter, int batch,
                                                                   double* scratch, double* dbg) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  FB_WAVE_TIMER(28);  // total wave cycles (diagnostic builds)
  lds_ptr lds = (lds_ptr)smem;
  typedef DenseWave::C C;
  C ctx;
  ctx.tid = threadIdx.x;
  ctx.red = lds;  // (unused: one wavefront reduces in registers)
  double* ws = scratch + (long)blockIdx.x * lay.ws_doubles;
  for (;;) {
    const int q = DBG ? (int)blockIdx.x : next_qp<64>(counter, lds);
    if (q >= batch) break;
    DenseData D;
    D.H = data.base[FBSTAB_DENSE_H] + q * data.stride[FBSTAB_DENSE_H];
    D.f = data.base[FBSTAB_DENSE_f] + q * data.stride[FBSTAB_DENSE_f];
    D.G = data.base[FBSTAB_DENSE_G] + q * data.stride[FBSTAB_DENSE_G];
    D.h = data.base[FBSTAB_DENSE_h] + q * data.stride[FBSTAB_DENSE_h];
    D.A = data.base[FBSTAB_DENSE_A] + q * data.stride[FBSTAB_DENSE_A];
    D.b = data.base[FBSTAB_DENSE_b] + q * data.stride[FBSTAB_DENSE_b];
    DenseWave p;
    p.bind(lay, D, x.base[0] + q * x.stride[0], x.base[1] + q * x.stride[1], x.base[2] + q * x.stride[2],
           x.base[3] + q * x.stride[3], lds, ws, counter + kDenseFallbackSlot);
    if constexpr (DBG) {
      newton_probe(p, ctx, opts, dbg);
      break;
    } else {
      Solver<DenseWave, C> solver(p, ctx, opts);
      solver.solve(out + q);
      ctx.sync();
    }
  }
}

// ---------------------------------------------------------------------------
thread_local std::string g_error;

int fail(int code, const std::string& msg) {
  g_error = msg;
  return code;
}

#define HIP_TRY(expr)                                                                  \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess)                                                              \
      return fail(FBSTAB_HIP_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

// State shared by both solver kinds.
struct SolverBase {
  int device = 0;
  int max_batch = 0;
  int threads = 0;
  int lds_bytes = 0;
  int workgroups = 0;
  long long scratch_bytes = 0;
  fbstab_options_t opts;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timed = false;
  int* counter = nullptr;
  double* scratch = nullptr;
  // staging for host-pointer calls
  std::vector<double*> d_arr;  // problem arrays
  std::vector<long long> arr_len;
  double* d_var[4] = {nullptr, nullptr, nullptr, nullptr};
  long long var_len[4] = {0, 0, 0, 0};
  fbstab_solver_out_t* d_out = nullptr;
  fbstab_solver_out_t* d_out_only = nullptr;  // FBSTAB_HIP_OUT_ON_HOST: device side of the records

  int ensure_out() {
    if (!d_out_only) HIP_TRY(hipMalloc(&d_out_only, sizeof(fbstab_solver_out_t) * (size_t)max_batch));
    return FBSTAB_HIP_OK;
  }
  double* d_norms = nullptr;  // fbstab_hip_*_solve_batch_final: device side of the norms returned to the host
  int ensure_norms() {
    if (!d_norms) HIP_TRY(hipMalloc(&d_norms, sizeof(double) * 4 * (size_t)max_batch));
    return FBSTAB_HIP_OK;
  }
  // workspace of the traced flat-vector MPC solve (kept from call to call)
  double* trace_ws = nullptr;

  int release() {
    (void)hipSetDevice(device);
    for (double* p : d_arr)
      if (p) (void)hipFree(p);
    for (int i = 0; i < 4; i++)
      if (d_var[i]) (void)hipFree(d_var[i]);
    if (d_out) (void)hipFree(d_out);
    if (d_out_only) (void)hipFree(d_out_only);
    if (d_norms) (void)hipFree(d_norms);
    if (trace_ws) (void)hipFree(trace_ws);
    if (scratch) (void)hipFree(scratch);
    if (counter) (void)hipFree(counter);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (stream) (void)hipStreamDestroy(stream);
    return FBSTAB_HIP_OK;
  }

  int common_init(int dev, int maxb) {
    device = dev;
    max_batch = maxb;
    fbstab_options_default(&opts);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
      return fail(FBSTAB_HIP_ERR_DEVICE, "no HIP device available (this library has no CPU path)");
    if (dev < 0 || dev >= ndev) return fail(FBSTAB_HIP_ERR_ARGUMENT, "bad device index");
    HIP_TRY(hipSetDevice(dev));
    // A blocking stream: ordered against the device's null stream in both
    // directions, so a caller that prepared its device arrays on the null stream
    // (and passes stream = NULL) needs no extra synchronisation.
    HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamDefault));
    HIP_TRY(hipEventCreate(&ev0));
    HIP_TRY(hipEventCreate(&ev1));
    HIP_TRY(hipMalloc(&counter, kQueueBytes));  // queue counter
    return FBSTAB_HIP_OK;
  }

  // Lazily allocated staging for host-pointer calls.
  int ensure_staging() {
    if (d_out) return FBSTAB_HIP_OK;
    d_arr.assign(arr_len.size(), nullptr);
    for (size_t i = 0; i < arr_len.size(); i++)
      HIP_TRY(hipMalloc(&d_arr[i], sizeof(double) * (size_t)(arr_len[i] > 0 ? arr_len[i] : 1) * max_batch));
    for (int i = 0; i < 4; i++)
      HIP_TRY(hipMalloc(&d_var[i], sizeof(double) * (size_t)(var_len[i] > 0 ? var_len[i] : 1) * max_batch));
    HIP_TRY(hipMalloc(&d_out, sizeof(fbstab_solver_out_t) * (size_t)max_batch));
    return FBSTAB_HIP_OK;
  }

  // Host batch array -> packed device array (stride = len); stride 0 is kept.
  int upload(const double* host, long long stride, long long len, int batch, double* dev,
             long long* dev_stride, hipStream_t s) {
    if (len == 0) {
      *dev_stride = 0;
      return FBSTAB_HIP_OK;
    }
    if (stride == 0) {
      HIP_TRY(hipMemcpyAsync(dev, host, sizeof(double) * len, hipMemcpyHostToDevice, s));
      *dev_stride = 0;
    } else if (stride == len) {
      HIP_TRY(hipMemcpyAsync(dev, host, sizeof(double) * len * batch, hipMemcpyHostToDevice, s));
      *dev_stride = len;
    } else {
      HIP_TRY(hipMemcpy2DAsync(dev, sizeof(double) * len, host, sizeof(double) * stride,
                               sizeof(double) * len, batch, hipMemcpyHostToDevice, s));
      *dev_stride = len;
    }
    return FBSTAB_HIP_OK;
  }
  int download(double* host, long long stride, long long len, int batch, const double* dev,
               hipStream_t s) {
    if (len == 0) return FBSTAB_HIP_OK;
    if (stride == len) {
      HIP_TRY(hipMemcpyAsync(host, dev, sizeof(double) * len * batch, hipMemcpyDeviceToHost, s));
    } else {
      HIP_TRY(hipMemcpy2DAsync(host, sizeof(double) * stride, dev, sizeof(double) * len,
                               sizeof(double) * len, batch, hipMemcpyDeviceToHost, s));
    }
    return FBSTAB_HIP_OK;
  }

  double last_kernel_ms() {
    if (!timed) return -1.0;
    (void)hipSetDevice(device);
    float ms = -1.f;
    if (hipEventSynchronize(ev1) != hipSuccess) return -1.0;
    if (hipEventElapsedTime(&ms, ev0, ev1) != hipSuccess) return -1.0;
    return (double)ms;
  }
};

// Device allocation released at scope exit.
struct DevBuf {
  void* p = nullptr;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { if (p) (void)hipFree(p); }
};

// Device side of fbstab_hip_*_solve_traced: 8 header doubles (record count,
// capacity) followed by the records (see Solver::emit in fb_algorithm.h).
struct TraceBuf {
  DevBuf buf;
  double* dev() const { return static_cast<double*>(buf.p); }
  int open(int device, const fbstab_trace_record_t* trace, int capacity, const int* count) {
    static_assert(sizeof(fbstab_trace_record_t) == 8 * sizeof(double), "record = 8 doubles");
    if (!trace || !count || capacity < 1)
      return fail(FBSTAB_HIP_ERR_ARGUMENT, "trace buffer, its capacity and the count pointer are required");
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMalloc(&buf.p, sizeof(double) * 8 * ((size_t)capacity + 1)));
    const double hdr[8] = {0.0, (double)capacity, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    HIP_TRY(hipMemcpy(buf.p, hdr, sizeof(hdr), hipMemcpyHostToDevice));
    return FBSTAB_HIP_OK;
  }
  // the solve has synchronised its stream by now (host-pointer call)
  int close(fbstab_trace_record_t* trace, int capacity, int* count) {
    double hdr[8];
    HIP_TRY(hipMemcpy(hdr, buf.p, sizeof(hdr), hipMemcpyDeviceToHost));
    *count = (int)hdr[0];
    const int n = *count < capacity ? *count : capacity;
    if (n > 0)
      HIP_TRY(hipMemcpy(trace, dev() + 8, sizeof(fbstab_trace_record_t) * (size_t)n, hipMemcpyDeviceToHost));
    return FBSTAB_HIP_OK;
  }
};

int check_common(const void* handle, int batch, const void* data, const fbstab_var_batch_t* x,
                 const void* out, int max_batch) {
  if (!handle) return fail(FBSTAB_HIP_ERR_ARGUMENT, "null solver handle");
  if (!data || !x || !out) return fail(FBSTAB_HIP_ERR_ARGUMENT, "null argument");
  if (batch < 0 || batch > max_batch)
    return fail(FBSTAB_HIP_ERR_ARGUMENT, "batch exceeds the max_batch the handle was created with");
  return FBSTAB_HIP_OK;
}

}  // namespace

struct fbstab_mpc_solver : SolverBase {
  fbk::MpcLayout lay;
  const RecordInstance* rec = nullptr;  // record kernel (fb_mpc_r16.h) serving this shape, or the flat-vector kernel
  bool exact = false;     // the problem has exactly the instance's shape
  int kept_batch = -1;    // batch size of the last FBSTAB_HIP_KEEP_MATRICES call whose copies are still in the slots
  int qps_per_wg = 1;
};

namespace {
// The record-kernel instances compiled into the library (one translation unit each),
// smallest first: a shape runs on the first one it fits (zero-padded unless it is that
// instance's own).
}  // namespace
FB_RECORD_INSTANCE_DECL(12, 4, 20, 1)
// the same stage width with up to two constraint rows per stage variable (two-sided
// bounds on all of x and u written as 32 rows); a third of its registers' worth of
// constraint slots more than the instance above, so that one stays the first choice
FB_RECORD_INSTANCE_DECL(12, 4, 32, 1)
// two 16-lane rows per QP: 16 < nx + nu <= 23 (the reference's copolymerization
// reactor, nx = 18, nu = 5, nc = 10: fbstab/test/ocp_generator.cc:73-174)
FB_RECORD_INSTANCE_DECL(18, 5, 10, 2)
// any stage width up to 32 (nx <= 24, nu <= 8) with up to 16 or 32 constraint rows: a row
// of a 32-wide stage matrix per lane is more than the register file holds (660-850
// spilled registers), three to four times the flat-vector kernel all the same
FB_RECORD_INSTANCE_DECL(24, 8, 16, 2)
FB_RECORD_INSTANCE_DECL(24, 8, 32, 2)
#if defined(FB_SINGLE_TU)
// (diagnostic builds with device-side counters: one translation unit, one g_stamps)
#include "rec_12_4_20.hip"
#include "rec_12_4_32.hip"
#include "rec_18_5_10.hip"
#include "rec_24_8_16.hip"
#include "rec_24_8_32.hip"
#endif
namespace {
const RecordInstance* record_instances(int* count) {
  static const RecordInstance table[] = {
      fbstab_record_instance_12_4_20_1(),
      fbstab_record_instance_12_4_32_1(),
      fbstab_record_instance_18_5_10_2(),
      fbstab_record_instance_24_8_16_2(),
      fbstab_record_instance_24_8_32_2(),
  };
  *count = (int)(sizeof(table) / sizeof(table[0]));
  return table;
}
const RecordInstance* record_instance_for(int nx, int nu, int nc) {
  int n = 0;
  const RecordInstance* t = record_instances(&n);
  for (int i = 0; i < n; i++)
    if (nx <= t[i].nx && nu <= t[i].nu && nc <= t[i].nc) return &t[i];
  return nullptr;
}

// Launches one kernel of a record instance (same argument list for all of them).
int launch_record(fbstab_mpc_solver* h, const void* kern, int grid, hipStream_t s, const MpcBatchArgs& a,
                  const VarBatchArgs& v, fbstab_solver_out_t* out, int batch, double* dbg, bool reuse) {
  MpcBatchPtrs d;
  VarBatchPtrs x;
  for (int i = 0; i < FBSTAB_MPC_NSEQ; i++) { d.base[i] = a.base[i]; d.stride[i] = a.stride[i]; }
  for (int i = 0; i < 4; i++) { x.base[i] = v.base[i]; x.stride[i] = v.stride[i]; }
  d.nx = h->lay.nx;
  d.nu = h->lay.nu;
  d.nc = h->lay.nc;
  int N = h->lay.N, ru = reuse ? 1 : 0;
  void* args[] = {&d, &x, &out, &h->opts, &h->scratch, &h->counter, &batch, &N, &ru, &dbg};
  HIP_TRY(hipLaunchKernel(kern, dim3(grid), dim3(64), args, (size_t)h->lds_bytes, s));
  return FBSTAB_HIP_OK;
}
}  // namespace
struct fbstab_dense_solver : SolverBase {
  fbk::DenseLayout lay;
  // one wavefront per QP with the KKT matrix in registers (fb_dense_wave.h): the
  // kernel batches of nz + nl <= 64 run on; `lay` then only serves the traced solve
  bool wave = false;
  fbk::DenseWaveLayout wlay;
};

extern "C" {

const char* fbstab_hip_last_error(void) { return g_error.c_str(); }

int fbstab_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// ---------------------------------------------------------------------------
int fbstab_hip_mpc_create(int N, int nx, int nu, int nc, int max_batch, int device,
                          fbstab_mpc_handle_t* handle) {
  return fbstab_hip_mpc_create_in_flight(N, nx, nu, nc, max_batch, device, 1, handle);
}

int fbstab_hip_mpc_create_in_flight(int N, int nx, int nu, int nc, int max_batch, int device, int handles_in_flight,
                                    fbstab_mpc_handle_t* handle) {
  if (!handle) return fail(FBSTAB_HIP_ERR_ARGUMENT, "null handle pointer");
  if (handles_in_flight < 1 || handles_in_flight > 64) return fail(FBSTAB_HIP_ERR_ARGUMENT, "handles_in_flight must be in 1..64");
  *handle = nullptr;
  // fbstab_mpc.cc:62-65
  if (N < 1 || nx < 1 || nu < 1 || nc < 1)
    return fail(FBSTAB_HIP_ERR_ARGUMENT, "In FBstabMpc::FBstabMpc: problem sizes must be positive.");
  if (max_batch < 1) return fail(FBSTAB_HIP_ERR_ARGUMENT, "max_batch must be positive");
  fbstab_mpc_solver* s = new (std::nothrow) fbstab_mpc_solver();
  if (!s) return fail(FBSTAB_HIP_ERR_DEVICE, "out of host memory");
  s->threads = kMpcThreads;
  s->lay.init(N, nx, nu, nc, s->threads);
  s->lds_bytes = s->lay.launch_lds_doubles * (int)sizeof(double);
  // FBSTAB_HIP_GENERIC=1 forces the flat-vector kernel (comparisons, tests)
  const char* force_generic = getenv("FBSTAB_HIP_GENERIC");
  if (!(force_generic && atoi(force_generic) > 0)) s->rec = record_instance_for(nx, nu, nc);
  if (s->rec) {
    s->exact = nx == s->rec->nx && nu == s->rec->nu && nc == s->rec->nc;
    s->qps_per_wg = s->rec->qps_per_wg;
    s->lds_bytes = s->rec->lds_bytes(N);
  }
  {
    // developer knob: extra (unused) LDS per workgroup, to lower the number of
    // resident wavefronts in occupancy experiments
    const char* pad = getenv("FBSTAB_HIP_LDS_PAD_BYTES");
    if (pad && atoi(pad) > 0) s->lds_bytes += atoi(pad) & ~15;
  }
  if (s->lds_bytes > kLdsLimitBytes) {  // (not reached: the flat-vector layout moves to global scratch first)
    delete s;
    return fail(FBSTAB_HIP_ERR_UNSUPPORTED, "stage vectors do not fit the 160 KiB LDS budget");
  }
  int rc = s->common_init(device, max_batch);
  if (rc != FBSTAB_HIP_OK) { s->release(); delete s; return rc; }
  // every kernel this handle can launch gets the LDS attribute; the occupancy
  // query runs on the one batches are launched with
  std::vector<const void*> kerns;
  if (s->rec) {
    const RecordInstance& r = *s->rec;
    if (s->exact) kerns = {r.solve_exact, r.solve_keep_exact, r.probe_exact};
    else kerns = {r.solve, r.solve_keep, r.probe};
  } else {
    if (s->lay.wglobal)
      kerns = {reinterpret_cast<const void*>(fbstab_mpc_kernel<kMpcThreads, false, false, true>),
               reinterpret_cast<const void*>(fbstab_mpc_kernel<kMpcThreads, true, false, true>)};
    else
      kerns = {reinterpret_cast<const void*>(fbstab_mpc_kernel<kMpcThreads, false>),
               reinterpret_cast<const void*>(fbstab_mpc_kernel<kMpcThreads, true>)};
  }
  hipError_t e = hipSuccess;
  for (const void* k : kerns)
    if (e == hipSuccess) e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, s->lds_bytes);
  int per_cu = 0, cus = 0;
  if (e == hipSuccess)
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kerns[0], s->threads, s->lds_bytes);
  hipDeviceProp_t prop;
  if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) {
    s->release(); delete s;
    return fail(FBSTAB_HIP_ERR_DEVICE, std::string("occupancy query: ") + hipGetErrorString(e));
  }
  cus = prop.multiProcessorCount;
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 8) per_cu = 8;
  // a handle that shares the device with others takes its share of the resident workgroups (and of the
  // scratch memory that goes with them), at least one per CU: the launches of the other handles fill the rest
  per_cu = (per_cu + handles_in_flight - 1) / handles_in_flight;
  const char* env = getenv("FBSTAB_HIP_WGS_PER_CU");
  if (env && atoi(env) > 0) per_cu = atoi(env);
  s->workgroups = cus * per_cu;
  {
    // (record kernels: a batch that does not outnumber the grid is SPREAD, one QP per wavefront - below -, so a
    // handle for a small max_batch keeps one workgroup per QP rather than one per four)
    const int need = s->rec ? max_batch : (max_batch + s->qps_per_wg - 1) / s->qps_per_wg;
    if (s->workgroups > need) s->workgroups = need;
  }
  const long long ws_doubles = s->rec ? s->rec->ws_doubles(N) : (long long)s->lay.ws_doubles;
  s->scratch_bytes = ws_doubles * sizeof(double) * s->workgroups * s->qps_per_wg;
  e = hipMalloc(&s->scratch, (size_t)s->scratch_bytes);
  if (e != hipSuccess) {
    s->release(); delete s;
    return fail(FBSTAB_HIP_ERR_DEVICE, std::string("scratch allocation: ") + hipGetErrorString(e));
  }
  const fbk::MpcLayout& L = s->lay;
  s->arr_len = {(long long)(N + 1) * nx * nx, (long long)(N + 1) * nu * nu,
                (long long)(N + 1) * nu * nx, (long long)(N + 1) * nx, (long long)(N + 1) * nu,
                (long long)N * nx * nx, (long long)N * nx * nu, (long long)N * nx,
                (long long)(N + 1) * nc * nx, (long long)(N + 1) * nc * nu,
                (long long)(N + 1) * nc, (long long)nx};
  s->var_len[0] = L.nz; s->var_len[1] = L.nl; s->var_len[2] = L.nv; s->var_len[3] = L.nv;
  *handle = s;
  return FBSTAB_HIP_OK;
}

int fbstab_hip_mpc_destroy(fbstab_mpc_handle_t h) {
  if (!h) return FBSTAB_HIP_OK;
  h->release();
  delete h;
  return FBSTAB_HIP_OK;
}

int fbstab_hip_mpc_set_options(fbstab_mpc_handle_t h, const fbstab_options_t* o) {
  if (!h || !o) return fail(FBSTAB_HIP_ERR_ARGUMENT, "null argument");
  h->opts = *o;
  fbstab_options_validate(&h->opts);  // UpdateParameters -> ValidateOptions
  return FBSTAB_HIP_OK;
}
int fbstab_hip_mpc_get_options(fbstab_mpc_handle_t h, fbstab_options_t* o) {
  if (!h || !o) return fail(FBSTAB_HIP_ERR_ARGUMENT, "null argument");
  *o = h->opts;
  return FBSTAB_HIP_OK;
}

// solve_batch; with d_trace != nullptr the ONE QP of the call runs on the traced
// instance of the flat-vector kernel instead (fbstab_hip_mpc_solve_traced).
// norms != nullptr: the component norms of the summary block follow the solve
// (fbstab_hip_mpc_solve_batch_final); they live where `out` lives.
static int mpc_solve_impl(fbstab_mpc_handle_t h, int batch, const fbstab_mpc_batch_t* data,
                          const fbstab_var_batch_t* x, fbstab_solver_out_t* out, int flags,
                          void* stream, double* d_trace, double* norms = nullptr) {
  int rc = check_common(h, batch, data, x, out, h ? h->max_batch : 0);
  if (rc != FBSTAB_HIP_OK) return rc;
  for (int i = 0; i < FBSTAB_MPC_NSEQ; i++)
    if (!data->base[i]) return fail(FBSTAB_HIP_ERR_ARGUMENT, "null problem data pointer");
  for (int i = 0; i < 4; i++)
    if (!x->base[i] && h->var_len[i] > 0) return fail(FBSTAB_HIP_ERR_ARGUMENT, "null variable pointer");
  if (batch == 0) return FBSTAB_HIP_OK;
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  const bool dev_ptrs = (flags & FBSTAB_HIP_DEVICE_POINTERS) != 0;
  const auto t0 = std::chrono::high_resolution_clock::now();
  MpcBatchArgs a;
  VarBatchArgs v;
  fbstab_solver_out_t* d_out = out;
  const bool out_host = dev_ptrs && (flags & FBSTAB_HIP_OUT_ON_HOST);
  if (dev_ptrs) {
    for (int i = 0; i < FBSTAB_MPC_NSEQ; i++) { a.base[i] = data->base[i]; a.stride[i] = data->stride[i]; }
    for (int i = 0; i < 4; i++) { v.base[i] = x->base[i]; v.stride[i] = x->stride[i]; }
    if (out_host) {
      rc = h->ensure_out();
      if (rc != FBSTAB_HIP_OK) return rc;
      d_out = h->d_out_only;
    }
  } else {
    rc = h->ensure_staging();
    if (rc != FBSTAB_HIP_OK) return rc;
    for (int i = 0; i < FBSTAB_MPC_NSEQ; i++) {
      if (data->stride[i] != 0 && data->stride[i] < h->arr_len[i] && batch > 1)
        return fail(FBSTAB_HIP_ERR_ARGUMENT, "problem data stride smaller than the array length");
      rc = h->upload(data->base[i], data->stride[i], h->arr_len[i], batch, h->d_arr[i], &a.stride[i], s);
      if (rc != FBSTAB_HIP_OK) return rc;
      a.base[i] = h->d_arr[i];
    }
    for (int i = 0; i < 4; i++) {
      if (x->stride[i] < h->var_len[i] && batch > 1)
        return fail(FBSTAB_HIP_ERR_ARGUMENT, "variable stride smaller than the vector length");
      if (i < 3) {
        long long st;
        rc = h->upload(x->base[i], x->stride[i] ? x->stride[i] : h->var_len[i], h->var_len[i], batch,
                       h->d_var[i], &st, s);
        if (rc != FBSTAB_HIP_OK) return rc;
      }
      v.base[i] = h->d_var[i];
      v.stride[i] = h->var_len[i];
    }
    d_out = h->d_out;
  }
  HIP_TRY(hipMemsetAsync(h->counter, 0, kQueueBytes, s));
  int grid = (batch + h->qps_per_wg - 1) / h->qps_per_wg;
  // Record kernels, a batch of no more QPs than the handle has workgroups: one QP per WAVEFRONT (row 0 of
  // each; fb_record_kernel.h, R16Queue::fetch) instead of four - the rows of a wavefront share its program
  // counter and its cooperative passes, so four QPs on one wavefront finish with the slowest of them and
  // queue for each other's passes, while the chip has SIMDs to spare (round 6: batch 16, 7.9 -> ms below).
  if (h->rec && batch <= h->workgroups) grid = batch;
  if (grid > h->workgroups) grid = h->workgroups;
  HIP_TRY(hipEventRecord(h->ev0, s));
  if (d_trace) {
    const int lds = h->lay.launch_lds_doubles * (int)sizeof(double);
    const void* kern = h->lay.wglobal
                           ? reinterpret_cast<const void*>(fbstab_mpc_kernel<kMpcThreads, false, true, true>)
                           : reinterpret_cast<const void*>(fbstab_mpc_kernel<kMpcThreads, false, true, false>);
    HIP_TRY(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    if (!h->trace_ws) HIP_TRY(hipMalloc(&h->trace_ws, sizeof(double) * (size_t)h->lay.ws_doubles));
    int one = 1;
    void* args[] = {&h->lay, &a, &v, &d_out, &h->opts, &h->trace_ws, &h->counter, &one, &d_trace};
    HIP_TRY(hipLaunchKernel(kern, dim3(1), dim3(kMpcThreads), args, (size_t)lds, s));
  } else if (h->rec) {
    // FBSTAB_HIP_KEEP_MATRICES: one QP per slot, slot = QP index
    const bool keep = (flags & FBSTAB_HIP_KEEP_MATRICES) && dev_ptrs && batch <= h->workgroups * h->qps_per_wg;
    const bool reuse = keep && h->kept_batch == batch;
    const RecordInstance& r = *h->rec;
    const void* kern = keep ? (h->exact ? r.solve_keep_exact : r.solve_keep) : (h->exact ? r.solve_exact : r.solve);
    rc = launch_record(h, kern, keep ? (batch + h->qps_per_wg - 1) / h->qps_per_wg : grid, s, a, v, d_out, batch,
                       nullptr, reuse);
    if (rc != FBSTAB_HIP_OK) return rc;
    h->kept_batch = keep ? batch : -1;
  } else {
    const void* kern = h->lay.wglobal
                           ? reinterpret_cast<const void*>(fbstab_mpc_kernel<kMpcThreads, false, false, true>)
                           : reinterpret_cast<const void*>(fbstab_mpc_kernel<kMpcThreads, false, false, false>);
    double* no_dbg = nullptr;
    void* args[] = {&h->lay, &a, &v, &d_out, &h->opts, &h->scratch, &h->counter, &batch, &no_dbg};
    HIP_TRY(hipLaunchKernel(kern, dim3(grid), dim3(h->threads), args, (size_t)h->lds_bytes, s));
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(h->ev1, s));
  h->timed = true;
  const bool norms_host = norms && (!dev_ptrs || out_host);
  if (norms) {
    double* dn = norms;
    if (norms_host) {
      rc = h->ensure_norms();
      if (rc != FBSTAB_HIP_OK) return rc;
      dn = h->d_norms;
    }
    MpcNormArgs na;
    for (int i = 0; i < FBSTAB_MPC_NSEQ; i++) { na.base[i] = a.base[i]; na.stride[i] = a.stride[i]; }
    for (int i = 0; i < 4; i++) { na.x[i] = v.base[i]; na.xstride[i] = v.stride[i]; }
    na.N = h->lay.N; na.nx = h->lay.nx; na.nu = h->lay.nu; na.nc = h->lay.nc;
    hipLaunchKernelGGL(fbstab_mpc_final_norms_kernel, dim3(batch), dim3(64), 0, s, na, h->opts, dn, batch);
    HIP_TRY(hipGetLastError());
    if (norms_host)
      HIP_TRY(hipMemcpyAsync(norms, dn, sizeof(double) * 4 * (size_t)batch, hipMemcpyDeviceToHost, s));
  }
  if (!dev_ptrs) {
    for (int i = 0; i < 4; i++) {
      rc = h->download(x->base[i], x->stride[i] ? x->stride[i] : h->var_len[i], h->var_len[i], batch,
                       h->d_var[i], s);
      if (rc != FBSTAB_HIP_OK) return rc;
    }
    HIP_TRY(hipMemcpyAsync(out, h->d_out, sizeof(fbstab_solver_out_t) * batch, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const double dt = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
    for (int i = 0; i < batch; i++) out[i].solve_time = dt;
  } else if (out_host) {
    HIP_TRY(hipMemcpyAsync(out, d_out, sizeof(fbstab_solver_out_t) * batch, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const double dt = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
    for (int i = 0; i < batch; i++) out[i].solve_time = dt;
  } else if (!(flags & FBSTAB_HIP_ASYNC)) {
    HIP_TRY(hipStreamSynchronize(s));
  }
  return FBSTAB_HIP_OK;
}

int fbstab_hip_mpc_solve_batch_final(fbstab_mpc_handle_t h, int batch, const fbstab_mpc_batch_t* data,
                                     const fbstab_var_batch_t* x, fbstab_solver_out_t* out, double* norms,
                                     int flags, void* stream) {
  if (!norms) return fail(FBSTAB_HIP_ERR_ARGUMENT, "null norms pointer");
  return mpc_solve_impl(h, batch, data, x, out, flags, stream, nullptr, norms);
}

int fbstab_hip_mpc_solve_batch(fbstab_mpc_handle_t h, int batch, const fbstab_mpc_batch_t* data,
                               const fbstab_var_batch_t* x, fbstab_solver_out_t* out, int flags,
                               void* stream) {
  return mpc_solve_impl(h, batch, data, x, out, flags, stream, nullptr);
}

int fbstab_hip_mpc_solve_traced(fbstab_mpc_handle_t h, const fbstab_mpc_batch_t* data,
                                const fbstab_var_batch_t* x, fbstab_solver_out_t* out,
                                fbstab_trace_record_t* trace, int capacity, int* count) {
  if (!h) return fail(FBSTAB_HIP_ERR_ARGUMENT, "null solver handle");
  TraceBuf tb;
  int rc = tb.open(h->device, trace, capacity, count);
  if (rc != FBSTAB_HIP_OK) return rc;
  rc = mpc_solve_impl(h, 1, data, x, out, FBSTAB_HIP_HOST_POINTERS, nullptr, tb.dev());
  if (rc != FBSTAB_HIP_OK) return rc;
  return tb.close(trace, capacity, count);
}

// BASELINE configs[4]: `steps` closed-loop steps without a host round trip in
// between - solve, plant step, solve, ... queued on one stream.
int fbstab_hip_mpc_receding_sweep(fbstab_mpc_handle_t h, int batch, const fbstab_mpc_batch_t* data,
                                  const fbstab_var_batch_t* x, fbstab_solver_out_t* out,
                                  const fbstab_receding_plant_t* plant, int steps, int retire,
                                  double* u_log, unsigned long long* stats, float* kernel_ms, void* stream) {
  int rc = check_common(h, batch, data, x, out, h ? h->max_batch : 0);
  if (rc != FBSTAB_HIP_OK) return rc;
  if (!plant || !plant->A || !plant->B || steps < 0)
    return fail(FBSTAB_HIP_ERR_ARGUMENT, "plant matrices and a non-negative step count are required");
  for (int i = 0; i < FBSTAB_MPC_NSEQ; i++)
    if (!data->base[i]) return fail(FBSTAB_HIP_ERR_ARGUMENT, "null problem data pointer");
  for (int i = 0; i < 4; i++)
    if (!x->base[i]) return fail(FBSTAB_HIP_ERR_ARGUMENT, "null variable pointer");
  // x0 is advanced in place, one state per trajectory: a shared x0 would be written by all of them
  if (batch > 1 && data->stride[FBSTAB_MPC_x0] < h->lay.nx)
    return fail(FBSTAB_HIP_ERR_ARGUMENT, "receding sweep: every trajectory needs its own x0 (stride >= nx)");
  for (int i = 0; i < 4; i++)
    if (batch > 1 && x->stride[i] < h->var_len[i])
      return fail(FBSTAB_HIP_ERR_ARGUMENT, "variable stride smaller than the vector length");
  if (batch == 0 || steps == 0) return FBSTAB_HIP_OK;
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  DevBuf d_ret, d_stats;
  HIP_TRY(hipMalloc(&d_ret.p, sizeof(int) * (size_t)batch));
  HIP_TRY(hipMalloc(&d_stats.p, sizeof(unsigned long long) * 4 * (size_t)steps));
  HIP_TRY(hipMemsetAsync(d_ret.p, 0, sizeof(int) * (size_t)batch, s));
  HIP_TRY(hipMemsetAsync(d_stats.p, 0, sizeof(unsigned long long) * 4 * (size_t)steps, s));
  std::vector<hipEvent_t> ev;
  struct EvGuard {
    std::vector<hipEvent_t>& e;
    ~EvGuard() { for (hipEvent_t x : e) (void)hipEventDestroy(x); }
  } guard{ev};
  if (kernel_ms) {
    ev.resize(2 * (size_t)steps, nullptr);
    for (auto& e : ev) HIP_TRY(hipEventCreate(&e));
  }
  VarBatchArgs v;
  for (int i = 0; i < 4; i++) { v.base[i] = x->base[i]; v.stride[i] = x->stride[i]; }
  const fbk::MpcLayout& L = h->lay;
  double* x0 = const_cast<double*>(data->base[FBSTAB_MPC_x0]);
  // Record kernels: the whole sweep is ONE launch of the KEEP instance, every row
  // looping over its own trajectory (SweepArgs; FBSTAB_HIP_SWEEP_PER_STEP=1 keeps the
  // launch per step below, which the flat-vector kernel always uses).
  const char* per_step = getenv("FBSTAB_HIP_SWEEP_PER_STEP");
  if (h->rec && batch <= h->workgroups * h->qps_per_wg && steps < 0xffff && !(per_step && atoi(per_step) != 0)) {
    for (int i = 0; i < FBSTAB_MPC_NSEQ; i++)
      if (!data->base[i]) return fail(FBSTAB_HIP_ERR_ARGUMENT, "null problem data pointer");
    SweepArgs sa;
    sa.A = plant->A; sa.B = plant->B; sa.sA = plant->stride_A; sa.sB = plant->stride_B;
    sa.x0 = x0; sa.sx0 = data->stride[FBSTAB_MPC_x0];
    sa.u_log = u_log; sa.stats = static_cast<unsigned long long*>(d_stats.p);
    sa.steps = steps; sa.retire = retire;
    sa.nx = L.nx; sa.nu = L.nu; sa.nz = L.nz; sa.nl = L.nl; sa.nv = L.nv;
    DevBuf d_sa;
    HIP_TRY(hipMalloc(&d_sa.p, sizeof(SweepArgs)));
    HIP_TRY(hipMemcpyAsync(d_sa.p, &sa, sizeof(SweepArgs), hipMemcpyHostToDevice, s));
    MpcBatchArgs a;
    for (int i = 0; i < FBSTAB_MPC_NSEQ; i++) { a.base[i] = data->base[i]; a.stride[i] = data->stride[i]; }
    HIP_TRY(hipMemsetAsync(h->counter, 0, kQueueBytes, s));
    HIP_TRY(hipEventRecord(h->ev0, s));
    const RecordInstance& r = *h->rec;
    rc = launch_record(h, h->exact ? r.solve_keep_exact : r.solve_keep, (batch + h->qps_per_wg - 1) / h->qps_per_wg, s,
                       a, v, out, batch, static_cast<double*>(d_sa.p), false);
    if (rc != FBSTAB_HIP_OK) return rc;
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(h->ev1, s));
    h->timed = true;
    h->kept_batch = batch;
    if (stats)
      HIP_TRY(hipMemcpyAsync(stats, d_stats.p, sizeof(unsigned long long) * 4 * (size_t)steps, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (kernel_ms) {  // one launch: every step is charged its share
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
      for (int k = 0; k < steps; k++) kernel_ms[k] = ms / (float)steps;
    }
    return FBSTAB_HIP_OK;
  }
  const int flags = FBSTAB_HIP_DEVICE_POINTERS | FBSTAB_HIP_ASYNC | FBSTAB_HIP_KEEP_MATRICES;
  h->kept_batch = -1;  // the first step builds the matrix copies
  DevBuf d_xtmp;  // the plant step's new states, for more than 64 of them per trajectory
  if (L.nx > 64) HIP_TRY(hipMalloc(&d_xtmp.p, sizeof(double) * (size_t)batch * L.nx));
  for (int k = 0; k < steps; k++) {
    if (kernel_ms) HIP_TRY(hipEventRecord(ev[2 * k], s));
    rc = mpc_solve_impl(h, batch, data, x, out, flags, s, nullptr);
    if (rc != FBSTAB_HIP_OK) return rc;
    if (kernel_ms) HIP_TRY(hipEventRecord(ev[2 * k + 1], s));
    hipLaunchKernelGGL(fbstab_receding_plant_kernel, dim3((batch + 63) / 64), dim3(64), 0, s, batch, L.nx, L.nu,
                       L.nz, L.nl, L.nv, plant->A, plant->stride_A, plant->B, plant->stride_B, x0,
                       data->stride[FBSTAB_MPC_x0], v, out, static_cast<int*>(d_ret.p), retire,
                       u_log ? u_log + (long long)k * batch * L.nu : nullptr,
                       static_cast<unsigned long long*>(d_stats.p) + 4 * k, static_cast<double*>(d_xtmp.p));
  }
  HIP_TRY(hipGetLastError());
  if (stats)
    HIP_TRY(hipMemcpyAsync(stats, d_stats.p, sizeof(unsigned long long) * 4 * (size_t)steps, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (kernel_ms)
    for (int k = 0; k < steps; k++) HIP_TRY(hipEventElapsedTime(&kernel_ms[k], ev[2 * k], ev[2 * k + 1]));
  return FBSTAB_HIP_OK;
}

// Diagnostics for the tests: one Newton step of the device path at (x, xbar,
// sigma0) for ONE QP given by host pointers.  io holds [zb, lb, vb] on input
// and [dz, dl, dv, adz, wz, wl, rz, rl, ok] on output
// (2*nz + 2*nl + 2*nv + nz + nl + 1 doubles).
int fbstab_hip_mpc_debug_newton(fbstab_mpc_handle_t h, const fbstab_mpc_batch_t* data,
                                const fbstab_var_batch_t* x, double* io) {
  if (!h || !data || !x || !io) return fail(FBSTAB_HIP_ERR_ARGUMENT, "null argument");
  HIP_TRY(hipSetDevice(h->device));
  int rc = h->ensure_staging();
  if (rc != FBSTAB_HIP_OK) return rc;
  hipStream_t s = h->stream;
  MpcBatchArgs a;
  VarBatchArgs v;
  for (int i = 0; i < FBSTAB_MPC_NSEQ; i++) {
    rc = h->upload(data->base[i], h->arr_len[i], h->arr_len[i], 1, h->d_arr[i], &a.stride[i], s);
    if (rc != FBSTAB_HIP_OK) return rc;
    a.base[i] = h->d_arr[i];
  }
  for (int i = 0; i < 4; i++) {
    long long st;
    if (i < 3) {
      rc = h->upload(x->base[i], h->var_len[i], h->var_len[i], 1, h->d_var[i], &st, s);
      if (rc != FBSTAB_HIP_OK) return rc;
    }
    v.base[i] = h->d_var[i];
    v.stride[i] = h->var_len[i];
  }
  const fbk::MpcLayout& L = h->lay;
  const size_t n_io = (size_t)(3 * L.nz + 3 * L.nl + 2 * L.nv + 1);
  DevBuf d_io_buf;
  HIP_TRY(hipMalloc(&d_io_buf.p, n_io * sizeof(double)));
  double* d_io = static_cast<double*>(d_io_buf.p);
  HIP_TRY(hipMemcpyAsync(d_io, io, sizeof(double) * (L.nz + L.nl + L.nv), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemsetAsync(h->counter, 0, kQueueBytes, s));
  h->kept_batch = -1;  // the probe runs in slot 0 and overwrites its matrix copies
  if (h->rec) {
    rc = launch_record(h, h->exact ? h->rec->probe_exact : h->rec->probe, 1, s, a, v, h->d_out, 1, d_io, false);
    if (rc != FBSTAB_HIP_OK) return rc;
  } else {
    const void* kern = h->lay.wglobal
                           ? reinterpret_cast<const void*>(fbstab_mpc_kernel<kMpcThreads, true, false, true>)
                           : reinterpret_cast<const void*>(fbstab_mpc_kernel<kMpcThreads, true, false, false>);
    int one = 1;
    void* args[] = {&h->lay, &a, &v, &h->d_out, &h->opts, &h->scratch, &h->counter, &one, &d_io};
    HIP_TRY(hipLaunchKernel(kern, dim3(1), dim3(h->threads), args, (size_t)h->lds_bytes, s));
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(io, d_io, sizeof(double) * n_io, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return FBSTAB_HIP_OK;
}

// Diagnostic builds (-DFB_STAMP): per-phase shader cycles summed over waves;
// zeros otherwise.  reset != 0 clears the counters after reading.
int fbstab_hip_debug_stamps(unsigned long long* out32, int reset) {
  if (!out32) return fail(FBSTAB_HIP_ERR_ARGUMENT, "null argument");
  memset(out32, 0, 32 * sizeof(unsigned long long));
#if defined(FB_ANY_STAMP)
  HIP_TRY(hipMemcpyFromSymbol(out32, HIP_SYMBOL(fbk::g_stamps), 32 * sizeof(unsigned long long)));
  if (reset) {
    unsigned long long z[32] = {0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(fbk::g_stamps), z, sizeof(z)));
  }
#else
  (void)reset;
#endif
  return FBSTAB_HIP_OK;
}

double fbstab_hip_mpc_last_kernel_ms(fbstab_mpc_handle_t h) { return h ? h->last_kernel_ms() : -1.0; }

int fbstab_hip_mpc_query(fbstab_mpc_handle_t h, long long* scratch_bytes, int* lds_bytes,
                         int* workgroups, int* threads) {
  if (!h) return fail(FBSTAB_HIP_ERR_ARGUMENT, "null solver handle");
  if (scratch_bytes) *scratch_bytes = h->scratch_bytes;
  if (lds_bytes) *lds_bytes = h->lds_bytes;
  if (workgroups) *workgroups = h->workgroups;
  if (threads) *threads = h->threads;
  return FBSTAB_HIP_OK;
}

// Name of the kernel batches of this handle run on (diagnostics, tests).
const char* fbstab_hip_mpc_kernel_name(fbstab_mpc_handle_t h) {
  if (!h) return "";
  return h->rec ? h->rec->name : "fbstab_mpc_kernel<64>";
}

int fbstab_hip_mpc_refined_steps(fbstab_mpc_handle_t h, long long* steps) {
  if (!h) return fail(FBSTAB_HIP_ERR_ARGUMENT, "null solver handle");
  if (!steps) return fail(FBSTAB_HIP_ERR_ARGUMENT, "null output pointer");
  *steps = -1;
  if (!h->timed) return FBSTAB_HIP_OK;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipEventSynchronize(h->ev1));
  int n = 0;  // word 1 of the queue block (zeroed before every launch): R16Queue::count_refinement, fb_mpc.h
  HIP_TRY(hipMemcpy(&n, h->counter + 1, sizeof(n), hipMemcpyDeviceToHost));
  *steps = n;
  return FBSTAB_HIP_OK;
}

// ---------------------------------------------------------------------------
int fbstab_hip_dense_create(int nz, int nl, int nv, int max_batch, int device,
                            fbstab_dense_handle_t* handle) {
  if (!handle) return fail(FBSTAB_HIP_ERR_ARGUMENT, "null handle pointer");
  *handle = nullptr;
  // fbstab_dense.cc:19-23
  if (nz < 1 || nv < 1 || nl < 0)
    return fail(FBSTAB_HIP_ERR_ARGUMENT, "In FBstabDense::FBstabDense: nz and nv must be positive, nl nonnegative.");
  if (max_batch < 1) return fail(FBSTAB_HIP_ERR_ARGUMENT, "max_batch must be positive");
  fbstab_dense_solver* s = new (std::nothrow) fbstab_dense_solver();
  if (!s) return fail(FBSTAB_HIP_ERR_DEVICE, "out of host memory");
  // nz + nl <= 64: one wavefront per QP for every phase, the KKT matrix in registers,
  // up to eight QPs per CU (fb_dense_wave.h).  Otherwise four wavefronts per QP with
  // K in LDS or, beyond nz + nl ~ 140, in global scratch (fb_dense.h).
  // FBSTAB_HIP_DENSE_THREADS=256 forces the four-wavefront kernel (comparisons);
  // =64 its one-wavefront instance with K in LDS.
  s->threads = kDenseThreads;
  s->wlay.init(nz, nl, nv);
  s->wave = s->wlay.fits();
  {
    const char* th = getenv("FBSTAB_HIP_DENSE_THREADS");
    if (th && atoi(th) == 256) s->wave = false;
    if (th && atoi(th) == 64 && nz + nl <= 64) { s->threads = 64; s->wave = false; }
    // Initial value of what fbstab_hip_dense_set_factorisation sets per handle (developer
    // switches; the default is the reference's order): FBSTAB_HIP_DENSE_ORDER=pivoted | auto |
    // natural, FBSTAB_HIP_DENSE_SPREAD_BITS, FBSTAB_HIP_DENSE_ACT_BITS, FBSTAB_HIP_DENSE_STICKY.
    const char* od = getenv("FBSTAB_HIP_DENSE_ORDER");
    if (od && !strcmp(od, "auto")) s->wlay.order = FBSTAB_HIP_DENSE_ORDER_AUTO;
    if (od && !strcmp(od, "natural")) s->wlay.order = FBSTAB_HIP_DENSE_ORDER_NATURAL;
    if (od && !strcmp(od, "pivoted")) s->wlay.order = FBSTAB_HIP_DENSE_ORDER_PIVOTED;
    const char* sb = getenv("FBSTAB_HIP_DENSE_SPREAD_BITS");
    if (sb && atoi(sb) > 0) s->wlay.spread_bits = atoi(sb);
    const char* sk = getenv("FBSTAB_HIP_DENSE_STICKY");
    if (sk) s->wlay.sticky = atoi(sk) != 0 ? 1 : 0;
    const char* ab = getenv("FBSTAB_HIP_DENSE_ACT_BITS");
    if (ab && atoi(ab) >= 0 && atoi(ab) < 1000) s->wlay.act_bits = atoi(ab);
  }
  s->lay.init(nz, nl, nv, s->threads);
  if (s->threads == 64 && (s->lay.k_global || !s->lay.a_lds)) {  // does not fit that way
    s->threads = kDenseThreads;
    s->lay.init(nz, nl, nv, s->threads);
  }
  s->lds_bytes = s->lay.lds_doubles * (int)sizeof(double);
  if (s->wave) {
    s->threads = 64;
    s->lay.init(nz, nl, nv, kDenseThreads);  // (the traced solve's layout)
    s->lds_bytes = s->wlay.lds_doubles * (int)sizeof(double);
  }
  if (s->lds_bytes > kLdsLimitBytes) {  // (not reached: the vectors move to global scratch first)
    delete s;
    return fail(FBSTAB_HIP_ERR_UNSUPPORTED, "the iterate vectors do not fit the 160 KiB LDS budget");
  }
  int rc = s->common_init(device, max_batch);
  if (rc != FBSTAB_HIP_OK) { s->release(); delete s; return rc; }
  const void* kern = s->wave ? reinterpret_cast<const void*>(fbstab_dense_wave_kernel<false>)
                     : s->threads == 64 ? reinterpret_cast<const void*>(fbstab_dense_kernel<64>)
                     : s->lay.v_global
                         ? reinterpret_cast<const void*>(fbstab_dense_kernel<kDenseThreads, false, true, true>)
                     : s->lay.k_global
                         ? reinterpret_cast<const void*>(fbstab_dense_kernel<kDenseThreads, false, true>)
                         : reinterpret_cast<const void*>(fbstab_dense_kernel<kDenseThreads>);
  hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, s->lds_bytes);
  if (e == hipSuccess && s->wave)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(fbstab_dense_wave_kernel<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, s->lds_bytes);
  int per_cu = 0;
  if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, s->threads, s->lds_bytes);
  hipDeviceProp_t prop;
  if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) {
    s->release(); delete s;
    return fail(FBSTAB_HIP_ERR_DEVICE, std::string("occupancy query: ") + hipGetErrorString(e));
  }
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 8) per_cu = 8;
  const char* env = getenv("FBSTAB_HIP_WGS_PER_CU");
  if (env && atoi(env) > 0) per_cu = atoi(env);
  s->workgroups = prop.multiProcessorCount * per_cu;
  if (s->workgroups > max_batch) s->workgroups = max_batch;
  s->scratch_bytes = 0;
  if (s->wave)  // A' and the multipliers of every resident workgroup (fb_dense_wave.h)
    s->scratch_bytes = (long long)sizeof(double) * s->wlay.ws_doubles * s->workgroups;
  else if (s->lay.k_global)  // K (and, v_global, the iterate vectors) of every resident workgroup (fb_dense.h)
    s->scratch_bytes = (long long)sizeof(double) * (s->lay.k_doubles + s->lay.v_doubles) * s->workgroups;
  if (s->scratch_bytes > 0) {
    e = hipMalloc(&s->scratch, (size_t)s->scratch_bytes);
    // (fb_dense_wave.h relies on the multiplier rows past nz + nl being zero)
    if (e == hipSuccess && s->wave) e = hipMemset(s->scratch, 0, (size_t)s->scratch_bytes);
    if (e != hipSuccess) {
      s->release(); delete s;
      return fail(FBSTAB_HIP_ERR_DEVICE, std::string("scratch allocation: ") + hipGetErrorString(e));
    }
  }
  s->arr_len = {(long long)nz * nz, (long long)nz, (long long)nl * nz, (long long)nl,
                (long long)nv * nz, (long long)nv};
  s->var_len[0] = nz; s->var_len[1] = nl; s->var_len[2] = nv; s->var_len[3] = nv;
  *handle = s;
  return FBSTAB_HIP_OK;
}

int fbstab_hip_dense_destroy(fbstab_dense_handle_t h) {
  if (!h) return FBSTAB_HIP_OK;
  h->release();
  delete h;
  return FBSTAB_HIP_OK;
}

int fbstab_hip_dense_set_options(fbstab_dense_handle_t h, const fbstab_options_t* o) {
  if (!h || !o) return fail(FBSTAB_HIP_ERR_ARGUMENT, "null argument");
  h->opts = *o;
  fbstab_options_validate(&h->opts);
  return FBSTAB_HIP_OK;
}
int fbstab_hip_dense_get_options(fbstab_dense_handle_t h, fbstab_options_t* o) {
  if (!h || !o) return fail(FBSTAB_HIP_ERR_ARGUMENT, "null argument");
  *o = h->opts;
  return FBSTAB_HIP_OK;
}

static int dense_solve_impl(fbstab_dense_handle_t h, int batch, const fbstab_dense_batch_t* data,
                            const fbstab_var_batch_t* x, fbstab_solver_out_t* out, int flags,
                            void* stream, double* d_trace, double* norms = nullptr) {
  int rc = check_common(h, batch, data, x, out, h ? h->max_batch : 0);
  if (rc != FBSTAB_HIP_OK) return rc;
  for (int i = 0; i < FBSTAB_DENSE_NARR; i++)
    if (!data->base[i] && h->arr_len[i] > 0)
      return fail(FBSTAB_HIP_ERR_ARGUMENT, "null problem data pointer");
  for (int i = 0; i < 4; i++)
    if (!x->base[i] && h->var_len[i] > 0) return fail(FBSTAB_HIP_ERR_ARGUMENT, "null variable pointer");
  if (batch == 0) return FBSTAB_HIP_OK;
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  const bool dev_ptrs = (flags & FBSTAB_HIP_DEVICE_POINTERS) != 0;
  const auto t0 = std::chrono::high_resolution_clock::now();
  DenseBatchArgs a;
  VarBatchArgs v;
  fbstab_solver_out_t* d_out = out;
  const bool out_host = dev_ptrs && (flags & FBSTAB_HIP_OUT_ON_HOST);
  if (dev_ptrs) {
    for (int i = 0; i < FBSTAB_DENSE_NARR; i++) { a.base[i] = data->base[i]; a.stride[i] = data->stride[i]; }
    for (int i = 0; i < 4; i++) { v.base[i] = x->base[i]; v.stride[i] = x->stride[i]; }
    if (out_host) {
      rc = h->ensure_out();
      if (rc != FBSTAB_HIP_OK) return rc;
      d_out = h->d_out_only;
    }
  } else {
    rc = h->ensure_staging();
    if (rc != FBSTAB_HIP_OK) return rc;
    for (int i = 0; i < FBSTAB_DENSE_NARR; i++) {
      if (data->stride[i] != 0 && data->stride[i] < h->arr_len[i] && batch > 1)
        return fail(FBSTAB_HIP_ERR_ARGUMENT, "problem data stride smaller than the array length");
      rc = h->upload(data->base[i], data->stride[i], h->arr_len[i], batch, h->d_arr[i], &a.stride[i], s);
      if (rc != FBSTAB_HIP_OK) return rc;
      a.base[i] = h->d_arr[i];
    }
    for (int i = 0; i < 4; i++) {
      if (x->stride[i] < h->var_len[i] && batch > 1)
        return fail(FBSTAB_HIP_ERR_ARGUMENT, "variable stride smaller than the vector length");
      if (i < 3) {
        long long st;
        rc = h->upload(x->base[i], x->stride[i] ? x->stride[i] : h->var_len[i], h->var_len[i], batch,
                       h->d_var[i], &st, s);
        if (rc != FBSTAB_HIP_OK) return rc;
      }
      v.base[i] = h->d_var[i];
      v.stride[i] = h->var_len[i];
    }
    d_out = h->d_out;
  }
  HIP_TRY(hipMemsetAsync(h->counter, 0, kQueueBytes, s));  // (the queue word and the kDenseFallbackSlot counters)
  int grid = h->workgroups < batch ? h->workgroups : batch;
  HIP_TRY(hipEventRecord(h->ev0, s));
  if (d_trace && h->lay.v_global) {
    auto kern = fbstab_dense_kernel<kDenseThreads, true, true, true>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_bytes));
    hipLaunchKernelGGL(kern, dim3(1), dim3(h->threads), h->lds_bytes, s, h->lay, a, v, d_out, h->opts,
                       h->counter, 1, TraceArg<true>{d_trace}, KScratchArg<true>{h->scratch});
  } else if (d_trace && h->lay.k_global) {
    auto kern = fbstab_dense_kernel<kDenseThreads, true, true>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_bytes));
    hipLaunchKernelGGL(kern, dim3(1), dim3(h->threads), h->lds_bytes, s, h->lay, a, v, d_out, h->opts,
                       h->counter, 1, TraceArg<true>{d_trace}, KScratchArg<true>{h->scratch});
  } else if (d_trace) {
    // the traced instance is the four-wavefront kernel with a layout of its own
    fbk::DenseLayout tl;
    tl.init(h->lay.nz, h->lay.nl, h->lay.nv, kDenseThreads);
    const int tlds = tl.lds_doubles * (int)sizeof(double);
    if (tl.k_global || tlds > kLdsLimitBytes)
      return fail(FBSTAB_HIP_ERR_UNSUPPORTED, "traced dense solve: layout does not fit");
    auto kern = fbstab_dense_kernel<kDenseThreads, true>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, tlds));
    hipLaunchKernelGGL(kern, dim3(1), dim3(kDenseThreads), tlds, s, tl, a, v, d_out, h->opts,
                       h->counter, 1, TraceArg<true>{d_trace}, KScratchArg<false>());
  } else if (h->wave) {
    hipLaunchKernelGGL(fbstab_dense_wave_kernel<false>, dim3(grid), dim3(64), h->lds_bytes, s, h->wlay, a, v, d_out,
                       h->opts, h->counter, batch, h->scratch, (double*)nullptr);
  } else if (h->threads == 64) {
    hipLaunchKernelGGL(fbstab_dense_kernel<64>, dim3(grid), dim3(64), h->lds_bytes, s, h->lay, a, v, d_out,
                       h->opts, h->counter, batch, TraceArg<false>(), KScratchArg<false>());
  } else if (h->lay.v_global) {
    hipLaunchKernelGGL((fbstab_dense_kernel<kDenseThreads, false, true, true>), dim3(grid), dim3(h->threads),
                       h->lds_bytes, s, h->lay, a, v, d_out, h->opts, h->counter, batch, TraceArg<false>(),
                       KScratchArg<true>{h->scratch});
  } else if (h->lay.k_global) {
    hipLaunchKernelGGL((fbstab_dense_kernel<kDenseThreads, false, true>), dim3(grid), dim3(h->threads),
                       h->lds_bytes, s, h->lay, a, v, d_out, h->opts, h->counter, batch, TraceArg<false>(),
                       KScratchArg<true>{h->scratch});
  } else {
    hipLaunchKernelGGL(fbstab_dense_kernel<kDenseThreads>, dim3(grid), dim3(h->threads), h->lds_bytes, s,
                       h->lay, a, v, d_out, h->opts, h->counter, batch, TraceArg<false>(),
                       KScratchArg<false>());
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(h->ev1, s));
  h->timed = true;
  const bool norms_host = norms && (!dev_ptrs || out_host);
  if (norms) {
    double* dn = norms;
    if (norms_host) {
      rc = h->ensure_norms();
      if (rc != FBSTAB_HIP_OK) return rc;
      dn = h->d_norms;
    }
    DenseNormArgs na;
    for (int i = 0; i < FBSTAB_DENSE_NARR; i++) { na.base[i] = a.base[i]; na.stride[i] = a.stride[i]; }
    for (int i = 0; i < 4; i++) { na.x[i] = v.base[i]; na.xstride[i] = v.stride[i]; }
    na.nz = h->lay.nz; na.nl = h->lay.nl; na.nv = h->lay.nv;
    hipLaunchKernelGGL(fbstab_dense_final_norms_kernel, dim3(batch), dim3(64), 0, s, na, h->opts, dn, batch);
    HIP_TRY(hipGetLastError());
    if (norms_host)
      HIP_TRY(hipMemcpyAsync(norms, dn, sizeof(double) * 4 * (size_t)batch, hipMemcpyDeviceToHost, s));
  }
  if (!dev_ptrs) {
    for (int i = 0; i < 4; i++) {
      rc = h->download(x->base[i], x->stride[i] ? x->stride[i] : h->var_len[i], h->var_len[i], batch,
                       h->d_var[i], s);
      if (rc != FBSTAB_HIP_OK) return rc;
    }
    HIP_TRY(hipMemcpyAsync(out, h->d_out, sizeof(fbstab_solver_out_t) * batch, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const double dt = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
    for (int i = 0; i < batch; i++) out[i].solve_time = dt;
  } else if (out_host) {
    HIP_TRY(hipMemcpyAsync(out, d_out, sizeof(fbstab_solver_out_t) * batch, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const double dt = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
    for (int i = 0; i < batch; i++) out[i].solve_time = dt;
  } else if (!(flags & FBSTAB_HIP_ASYNC)) {
    HIP_TRY(hipStreamSynchronize(s));
  }
  return FBSTAB_HIP_OK;
}

int fbstab_hip_dense_solve_batch_final(fbstab_dense_handle_t h, int batch, const fbstab_dense_batch_t* data,
                                       const fbstab_var_batch_t* x, fbstab_solver_out_t* out, double* norms,
                                       int flags, void* stream) {
  if (!norms) return fail(FBSTAB_HIP_ERR_ARGUMENT, "null norms pointer");
  return dense_solve_impl(h, batch, data, x, out, flags, stream, nullptr, norms);
}

int fbstab_hip_dense_solve_batch(fbstab_dense_handle_t h, int batch,
                                 const fbstab_dense_batch_t* data, const fbstab_var_batch_t* x,
                                 fbstab_solver_out_t* out, int flags, void* stream) {
  return dense_solve_impl(h, batch, data, x, out, flags, stream, nullptr);
}

int fbstab_hip_dense_solve_traced(fbstab_dense_handle_t h, const fbstab_dense_batch_t* data,
                                  const fbstab_var_batch_t* x, fbstab_solver_out_t* out,
                                  fbstab_trace_record_t* trace, int capacity, int* count) {
  if (!h) return fail(FBSTAB_HIP_ERR_ARGUMENT, "null solver handle");
  TraceBuf tb;
  int rc = tb.open(h->device, trace, capacity, count);
  if (rc != FBSTAB_HIP_OK) return rc;
  rc = dense_solve_impl(h, 1, data, x, out, FBSTAB_HIP_HOST_POINTERS, nullptr, tb.dev());
  if (rc != FBSTAB_HIP_OK) return rc;
  return tb.close(trace, capacity, count);
}

// Diagnostics for the tests: one Newton step of the dense device path
// (DenseCholeskySolver::Initialize + Solve, dense_cholesky_solver.cc:32-127) at
// (x, xbar, sigma0) for ONE QP given by host pointers; io as in
// fbstab_hip_mpc_debug_newton.  K in LDS only (nz + nl up to ~140).
int fbstab_hip_dense_debug_newton(fbstab_dense_handle_t h, const fbstab_dense_batch_t* data,
                                  const fbstab_var_batch_t* x, double* io) {
  if (!h || !data || !x || !io) return fail(FBSTAB_HIP_ERR_ARGUMENT, "null argument");
  if (!h->wave && h->lay.k_global) return fail(FBSTAB_HIP_ERR_UNSUPPORTED, "dense probe: K must fit the LDS");
  HIP_TRY(hipSetDevice(h->device));
  int rc = h->ensure_staging();
  if (rc != FBSTAB_HIP_OK) return rc;
  hipStream_t s = h->stream;
  DenseBatchArgs a;
  VarBatchArgs v;
  for (int i = 0; i < FBSTAB_DENSE_NARR; i++) {
    rc = h->upload(data->base[i], h->arr_len[i], h->arr_len[i], 1, h->d_arr[i], &a.stride[i], s);
    if (rc != FBSTAB_HIP_OK) return rc;
    a.base[i] = h->d_arr[i];
  }
  for (int i = 0; i < 4; i++) {
    long long st;
    if (i < 3) {
      rc = h->upload(x->base[i], h->var_len[i], h->var_len[i], 1, h->d_var[i], &st, s);
      if (rc != FBSTAB_HIP_OK) return rc;
    }
    v.base[i] = h->d_var[i];
    v.stride[i] = h->var_len[i];
  }
  const fbk::DenseLayout& L = h->lay;
  const size_t n_io = (size_t)(3 * L.nz + 3 * L.nl + 2 * L.nv + 1);
  DevBuf d_io_buf;
  HIP_TRY(hipMalloc(&d_io_buf.p, n_io * sizeof(double)));
  double* d_io = static_cast<double*>(d_io_buf.p);
  HIP_TRY(hipMemcpyAsync(d_io, io, sizeof(double) * (L.nz + L.nl + L.nv), hipMemcpyHostToDevice, s));
  if (h->wave) {
    hipLaunchKernelGGL(fbstab_dense_wave_kernel<true>, dim3(1), dim3(64), h->lds_bytes, s, h->wlay, a, v,
                       (fbstab_solver_out_t*)nullptr, h->opts, h->counter, 1, h->scratch, d_io);
  } else if (h->threads == 64) {
    auto kern = fbstab_dense_probe_kernel<64>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                h->lds_bytes));
    hipLaunchKernelGGL(kern, dim3(1), dim3(64), h->lds_bytes, s, h->lay, a, v, h->opts, d_io);
  } else {
    auto kern = fbstab_dense_probe_kernel<kDenseThreads>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                h->lds_bytes));
    hipLaunchKernelGGL(kern, dim3(1), dim3(kDenseThreads), h->lds_bytes, s, h->lay, a, v, h->opts, d_io);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(io, d_io, sizeof(double) * n_io, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return FBSTAB_HIP_OK;
}

double fbstab_hip_dense_last_kernel_ms(fbstab_dense_handle_t h) { return h ? h->last_kernel_ms() : -1.0; }

int fbstab_hip_dense_set_factorisation(fbstab_dense_handle_t h, int order, int spread_bits) {
  if (!h) return fail(FBSTAB_HIP_ERR_ARGUMENT, "null solver handle");
  if (order != FBSTAB_HIP_DENSE_ORDER_AUTO && order != FBSTAB_HIP_DENSE_ORDER_PIVOTED &&
      order != FBSTAB_HIP_DENSE_ORDER_NATURAL)
    return fail(FBSTAB_HIP_ERR_ARGUMENT, "fbstab_hip_dense_set_factorisation: unknown elimination order");
  if (spread_bits < 0 || spread_bits > 2046)
    return fail(FBSTAB_HIP_ERR_ARGUMENT, "fbstab_hip_dense_set_factorisation: spread_bits out of range");
  h->wlay.order = order;
  if (spread_bits > 0) h->wlay.spread_bits = spread_bits;
  return FBSTAB_HIP_OK;
}

int fbstab_hip_dense_get_factorisation(fbstab_dense_handle_t h, int* order, int* spread_bits,
                                       long long* pivoted_steps) {
  if (!h) return fail(FBSTAB_HIP_ERR_ARGUMENT, "null solver handle");
  // (handles that run the four-wavefront kernels always pivot)
  if (order) *order = h->wave ? h->wlay.order : FBSTAB_HIP_DENSE_ORDER_PIVOTED;
  if (spread_bits) *spread_bits = h->wlay.spread_bits;
  if (pivoted_steps) {
    *pivoted_steps = -1;
    if (h->wave && h->timed) {
      HIP_TRY(hipSetDevice(h->device));
      HIP_TRY(hipEventSynchronize(h->ev1));
      int n[2] = {0, 0};  // natural-order attempts handed on; steps of QPs that stayed pivoted after one
      HIP_TRY(hipMemcpy(n, h->counter + kDenseFallbackSlot, sizeof(n), hipMemcpyDeviceToHost));
      *pivoted_steps = (long long)n[0] + n[1];
    }
  }
  return FBSTAB_HIP_OK;
}

int fbstab_hip_dense_query(fbstab_dense_handle_t h, long long* scratch_bytes, int* lds_bytes,
                           int* workgroups, int* threads) {
  if (!h) return fail(FBSTAB_HIP_ERR_ARGUMENT, "null solver handle");
  if (scratch_bytes) *scratch_bytes = h->scratch_bytes;
  if (lds_bytes) *lds_bytes = h->lds_bytes;
  if (workgroups) *workgroups = h->workgroups;
  if (threads) *threads = h->threads;
  return FBSTAB_HIP_OK;
}

}  // extern "C"

#include "fb_shard.h"

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
#include "fb_mpc.h"
#include "fb_mpc_g16.h"
#include "fb_mpc_r16.h"

#if defined(FB_STAMP) || defined(FB_CLOCKSTAMP)
namespace fbk { __device__ unsigned long long g_stamps[32]; }
#define FB_ANY_STAMP 1
#endif

namespace {

using namespace fbk;

constexpr int kMpcThreads = 64;     // one wavefront per MPC QP
constexpr int kDenseThreads = 256;  // four wavefronts per dense QP
constexpr int kLdsLimitBytes = 160 * 1024;

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

// Diagnostic probe (tests only): one Newton step at (x, xbar, sigma) instead of
// a solve.  dbg holds [zb, lb, vb] on input and receives
// [dz, dl, dv, adz, wz, wl, rz, rl, ok].
template <class P, class C>
__device__ __forceinline__ void newton_probe(P& p, const C& ctx, const fbstab_options_t& opts, double* dbg) {
  p.load_guess(ctx);
  if constexpr (P::kOwnVectorOps) {
    p.probe_set_xbar(ctx, dbg);
    p.residual(ctx);
    double a, b;
    const bool ok = p.newton_step(ctx, opts.sigma0, opts.alpha, &a, &b);
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
// fbstab_hip_mpc_solve_traced (see Solver in fb_algorithm.h).
template <int NT, bool DBG, bool TRACE = false>
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
  ctx.red = lds + lay.w_red;
  double* ws = scratch + (long)blockIdx.x * lay.ws_doubles;
  for (;;) {
    const int q = next_qp<NT>(counter, lds + lay.w_out);
    if (q >= batch) break;
    MpcProblem<C> p;
    p.bind(lay, mpc_data_of(data, q), x.base[0] + q * x.stride[0], x.base[1] + q * x.stride[1],
           x.base[2] + q * x.stride[2], x.base[3] + q * x.stride[3], lds, ws);
    if constexpr (DBG) {
      newton_probe(p, ctx, opts, dbg);
    } else {
      Solver<MpcProblem<C>, C, TRACE> solver(p, ctx, opts, dbg);
      solver.solve(out + q);
    }
    ctx.sync();
  }
}


#ifndef FB_G16_MIN_WAVES
#define FB_G16_MIN_WAVES 1
#endif
// Four QPs per wavefront, one per 16-lane DPP row (fb_mpc_g16.h).  Rows run
// the solver loop independently (SIMT divergence between rows) and pull QP
// indices from the shared counter.
template <int NX, int NU, int NC, bool DBG>
__global__ __launch_bounds__(64, FB_G16_MIN_WAVES) void fbstab_mpc_g16_kernel(
    MpcLayout lay, MpcBatchArgs data, VarBatchArgs x, fbstab_solver_out_t* out,
    fbstab_options_t opts, double* scratch, int* counter, int batch, int lds_per_row, double* dbg) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x, row = lane >> 4;
  lds_ptr lds = (lds_ptr)smem + row * lds_per_row;
  Ctx16 ctx;
  ctx.tid = lane & 15;
  double* ws = scratch + ((long)blockIdx.x * 4 + row) * lay.ws_doubles;
  MpcProblemG16<NX, NU, NC> p;
  // Binds the policy to the next QP of the shared queue; -1 when it is empty.
  auto next = [&](MpcProblemG16<NX, NU, NC>& pp) -> int {
    int q = 0;
    if (ctx.tid == 0) q = atomicAdd(counter, 1);
    q = bci<0>(q);
    if (q >= batch) return -1;
    pp.bind(lay, mpc_data_of(data, q), x.base[0] + q * x.stride[0], x.base[1] + q * x.stride[1],
            x.base[2] + q * x.stride[2], x.base[3] + q * x.stride[3], lds, ws);
    pp.pend_t = 0.0;
    return q;
  };
  if constexpr (DBG) {
    if (next(p) >= 0) newton_probe(p, ctx, opts, dbg);
  } else {
    Solver<MpcProblemG16<NX, NU, NC>, Ctx16> solver(p, ctx, opts);
    for (;;) {
      const int q = next(p);
      if (q < 0) break;
      solver.solve(out + q);
    }
  }
}

// Build knobs (measured on the BASELINE workload):
//   FB_R16_MIN_WAVES  1: 512 registers per wave, loads prefetched a stage ahead
//                     2: two waves per SIMD, loads at the point of use
//   FB_R16_NESTED     the four rows of a wavefront run solve() in step instead of
//                     the per-row state machine of solve_stream (rows never wait
//                     for each other; +11 % QP/s with two batches in flight)
#ifndef FB_R16_MIN_WAVES
#define FB_R16_MIN_WAVES 1
#endif
// Record-based 16-lane kernel (fb_mpc_r16.h): four QPs per wavefront, rows pull
// QP indices from the shared counter.  scratch: rows * ws_doubles(N).
// Work distribution of the record kernel: the shared queue of QP indices and the
// board on which solves in progress change wavefronts (Solver::solve_stream
// explains why).  One instance per 16-lane row; every function is called by
// the whole row and returns row-uniform values.  Nothing in here waits for
// another wavefront.
//
// Device memory (zeroed by the host before each launch):
//   ctl[0] next QP index     ctl[1] board entries reserved
//   ctl[3] every board entry below this index is closed
//   board[i]: 0 reserved, not written yet
//             OPEN    = busy << 61 | (workgroup + 1) << 48 | (slot + 1) << 24 | (q + 1): an
//                       invitation - the row keeps solving while it stands; busy = busy
//                       rows of its wavefront when it was posted
//             CLAIMED = OPEN | 1 << 63: a row of another wavefront will continue this
//                       solve as soon as its owner has parked it
//             DEAD    = ~0: withdrawn (the solve ended first)
// and in the slot header of every QP the hand-over word (P::park_flag): 0 running,
// 1 parked and ready, 2 finished before the claim was noticed.
// Invitations are accepted by the rows of wavefronts that have run out of work
// (Solver::solve_stream), and a solve moves at most once.  Coherence: the two rows
// may sit in different XCDs, whose L2s are not coherent; the owner writes its L2
// back (agent-scope release fence) before it sets the hand-over word, the new
// owner invalidates (acquire fence) after it has read it, and the words
// themselves are agent-scope atomics.
constexpr int kQueueCtlInts = 8;
constexpr int kBoardEntries = 16384;
constexpr size_t kBoardBytes = kQueueCtlInts * sizeof(int) + kBoardEntries * sizeof(unsigned long long);
constexpr unsigned long long kBoardClaimed = 1ull << 63, kBoardDead = ~0ull;
// The board operations are rare: as real calls they stay out of the register
// allocation of the solver loop (build knob FB_R16_MIG_INLINE for comparison).
#if defined(FB_R16_MIG_INLINE)
#define FB_COLD __device__ __forceinline__
#else
#define FB_COLD __device__ __attribute__((noinline))
#endif

// Tail compaction by relaunch (PHASE >= 0; the default for batches).  A batch runs
// as three launches of this kernel on its stream.  In the first two, a wavefront
// that is down to FB_COMPACT_MAX_BUSY (2) busy rows after the queue has run dry parks the
// solves it still hosts (P::park - the rest of a solve lives in its slot already),
// appends them to a list and leaves; the next launch starts from that list with
// four parked solves per wavefront (P::resume) and continues each with the very
// Newton step it was about to take.  Launches are ordered by the stream, so no
// wavefront ever waits for another and nothing needs fences or atomics beyond the
// two counters.  Why: once the queue is empty the rows of a wavefront finish at
// different times while the wavefront holds its SIMD until the last one does -
// 17 % of all row slots idle at batch 8192; a simulation over the measured iteration
// counts gives 13 % less wavefront time for three launches.
//   blk[0] next index into the input (QP queue or parked list)   blk[1] solves parked
//   blk[8 + 2 i], blk[9 + 2 i]  slot and QP index of parked solve i
constexpr int kCompactCap = 8192;  // >= 4 x resident wavefronts
constexpr int kCompactBlkInts = 8 + 2 * kCompactCap;
#if defined(FB_R16_MIGRATE) || defined(FB_R16_NO_COMPACT)
constexpr bool kCompactBatches = false;
#else
constexpr bool kCompactBatches = true;
#endif
// the counter buffer of a handle: the board of the migration experiment or the three blocks
constexpr size_t kQueueBytes = kBoardBytes > 3 * kCompactBlkInts * sizeof(int) ? kBoardBytes : 3 * kCompactBlkInts * sizeof(int);

template <class P, bool KEEP, int PHASE = -1>
struct R16Queue {
  static constexpr bool kCompact = PHASE >= 0;
  static constexpr bool kResume = PHASE >= 1;
  static constexpr bool kMayPark = PHASE == 0 || PHASE == 1;
  int* in_blk = nullptr;   // (kResume) the list the previous launch wrote
  int* out_blk = nullptr;  // (kMayPark) the list this launch writes
  // Only launch-uniform values live in here (SGPRs).  What a row remembers between
  // trips sits in four spare words of its LDS region - the sweeps have no
  // registers to spare for it (a handful of VGPRs held across the Newton step
  // turned 2 spilled registers into 44):
  //   [0] cursor: board entries below it are closed for good
  //   [1] index of this row's standing invitation, -1 none
  //   [2] slot of the solve this row has claimed and waits for, -1 none   [3] its QP index
  const MpcBatchPtrs* data;
  const VarBatchPtrs* x;
  int* ctl;
  unsigned long long* board;
  double* scratch;
  int batch, N;
  bool reuse;
  bool taken = false;  // (KEEP) this row has had its one QP

  static __device__ __forceinline__ int tid() { return threadIdx.x & 15; }
  static __device__ __forceinline__ int row() { return threadIdx.x >> 4; }
  static __device__ __forceinline__ int home() { return blockIdx.x * 4 + row(); }
  static __device__ __forceinline__ int wg() { return blockIdx.x; }
  static __device__ __forceinline__ lds_ptr lds() {
    extern __shared__ __attribute__((aligned(16))) double smem_[];
    return (lds_ptr)smem_ + row() * P::kLdsPerRow;
  }
  static __device__ __forceinline__ FB_LDS int* mem() { return (FB_LDS int*)(lds() + P::kLdsDoubles); }
  // Twenty more spare doubles of the row's LDS region: the solver loop parks its
  // scalars there while a Newton step and its line search run (Solver::solve_stream).
  static __device__ __forceinline__ lds_ptr save_area() { return lds() + P::kLdsDoubles + 4; }
  static __device__ __forceinline__ void init() {
    FB_LDS int* m = mem();
    m[0] = 0;
    m[1] = -1;
    m[2] = -1;
    m[3] = -1;
  }

  // Reads and writes of the words other wavefronts change go through read-modify-
  // write atomics, which are performed at the memory side.  An agent-scope atomic
  // LOAD is not enough on this part: it may be served from this XCD's L2, which
  // another XCD's writes do not update (measured: a polled word stayed stale
  // for the rest of the kernel).
  static __device__ __forceinline__ int load(int* p) { return atomicAdd(p, 0); }
  static __device__ __forceinline__ unsigned long long load(unsigned long long* p) { return atomicAdd(p, 0ull); }
  static __device__ __forceinline__ double load_flag(double* p) {
    return __longlong_as_double((long long)atomicAdd(reinterpret_cast<unsigned long long*>(p), 0ull));
  }
  static __device__ __forceinline__ void store_flag(double* p, double v) {
    atomicExch(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v));
  }
  __device__ __forceinline__ double* slot_ptr(long slot) const { return scratch + slot * P::ws_doubles(N); }

  // (kResume) binds the policy to the next parked solve; st receives its scalars.
  template <int NS>
  __device__ __forceinline__ int fetch_parked(P& pp, double (&st)[NS]) {
    int i = 0;
    if (tid() == 0) i = atomicAdd(&ctl[0], 1);
    i = bci<0>(i);
    if (i >= in_blk[1]) return -1;
    const int slot = in_blk[8 + 2 * i], q = in_blk[9 + 2 * i];
    pp.resume(tid(), slot_ptr(slot), lds(), data, x, q, N, st);
    return q;
  }
  // (kMayPark) appends the solve pp has parked in its slot to this launch's list.
  __device__ __forceinline__ void park_out(const P& pp) {
    if (tid() == 0) {
      const int i = atomicAdd(&out_blk[1], 1);
      out_blk[8 + 2 * i] = (int)((reinterpret_cast<const double*>(pp.poff) - scratch) / P::ws_doubles(N));
      out_blk[9 + 2 * i] = (int)pp.q;
    }
  }

  // Binds the policy to the next QP of the queue, in this row's own slot.
  __device__ __forceinline__ int fetch(P& pp) {
    int q = 0;
    if constexpr (KEEP) {
      q = home();
      if (taken) return -1;
      taken = true;
    } else {
      if (tid() == 0) q = atomicAdd(&ctl[0], 1);
      q = bci<0>(q);
    }
    if (q >= batch) return -1;
    pp.bind(slot_ptr(home()), lds(), data, x, q, N, tid());
    if constexpr (KEEP) pp.reuse = reuse;
    if constexpr (P::kMigrate) {
      if (tid() == 0)
        store_flag(P::park_flag(slot_ptr(home()), N), 0.0);
    }
    return q;
  }

  // ---- the row that owns a solve ---------------------------------------------------
  __device__ __forceinline__ bool invited() const { return mem()[1] >= 0; }
  // Posts an invitation for the solve pp is bound to (no-op when the board is full).
  FB_COLD void invite(const P& pp, int busy) {
    // (the reset of this slot's hand-over word by fetch / take_over must have landed
    // before anybody can see the invitation)
    __threadfence();
    int i = -1;
    if (tid() == 0 && pp.q + 1 < (1 << 24)) {
      i = atomicAdd(&ctl[1], 1);
      if (i < kBoardEntries) {
        const long slot = (reinterpret_cast<const double*>(pp.poff) - scratch) / P::ws_doubles(N);
        const unsigned long long v = ((unsigned long long)busy << 61) | ((unsigned long long)(wg() + 1) << 48) | ((unsigned long long)(slot + 1) << 24) |
                                     (unsigned long long)(pp.q + 1);
        // (the hand-over word of this slot was reset by an atomic before: it is out already)
        atomicExch(&board[i], v);
      } else {
        i = -1;
      }
    }
    mem()[1] = bci<0>(i);
  }
  // Has somebody accepted the standing invitation?
  __device__ __forceinline__ bool claimed(const P& pp) const {
    int c = 0;
    if (tid() == 0) {
      const unsigned long long e = load(&board[mem()[1]]);
      c = (e & kBoardClaimed) ? 1 : 0;
    }
    return bci<0>(c) != 0;
  }
  // The solve is parked in its slot (P::park): let the claiming row have it.
  FB_COLD void hand_over(const P& pp) {
    __threadfence();
    if (tid() == 0) store_flag(P::park_flag(reinterpret_cast<double*>(pp.poff), N), 1.0);
    mem()[1] = -1;
  }
  // The solve has ended on this row: withdraw the invitation, or tell the row
  // that accepted it in the meantime.
  FB_COLD void retire(const P& pp) {
    const int invite_idx = mem()[1];
    if (invite_idx < 0) return;
    if (tid() == 0) {
      unsigned long long v = load(&board[invite_idx]);
      if (!(v & kBoardClaimed)) v = atomicCAS(&board[invite_idx], v, kBoardDead);
      if (v & kBoardClaimed) store_flag(P::park_flag(reinterpret_cast<double*>(pp.poff), N), 2.0);
    }
    mem()[1] = -1;
  }

  // ---- an idle row ------------------------------------------------------------------
  __device__ __forceinline__ bool waiting() const { return mem()[2] >= 0; }
  // Accepts one open invitation, if there is one (called by the rows of a wavefront
  // that has run out of work).
  FB_COLD void claim() {
    int n = 0, head = 0;
    int cursor = mem()[0];
    if (tid() == 0) {
      n = load(&ctl[1]);
      if (cursor == 0) head = load(&ctl[3]);  // (a row that has not looked yet starts at the shared hint)
    }
    n = bci<0>(n);
    head = bci<0>(head);
    if (n > kBoardEntries) n = kBoardEntries;
    const bool from_hint = cursor == 0;
    if (cursor < head) cursor = head;
    else head = cursor;
    // The row's cursor passes closed entries for good; an entry that is reserved but
    // not written yet holds it back.  The shared hint ctl[3] lets rows that have
    // not looked yet skip the closed prefix.
    bool all_closed = from_hint;  // every entry in [hint, sc) is closed
    for (int sc = cursor; sc < n;) {
      const int idx = sc + tid();  // sixteen entries at a time, one per lane
      unsigned long long v = 0ull;
      if (idx < n) v = load(&board[idx]);
      const bool closed = idx >= n || (v & kBoardClaimed) != 0ull;  // CLAIMED or DEAD
      const bool mine = !closed && v != 0ull;  // OPEN
      const int first = 16 - (int)row_reduce<OpMax16>(mine ? (double)(16 - tid()) : 0.0);
      if (first < 16) {
        int won = 0;
        if (tid() == first) won = atomicCAS(&board[idx], v, v | kBoardClaimed) == v ? 1 : 0;
        won = __shfl(won, first, 16);
        if (won) {
          const int lo = __shfl((int)(v & 0xffffffffull), first, 16);
          const int hi = __shfl((int)(v >> 32), first, 16);
          const unsigned long long e = ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
          mem()[2] = (int)((e >> 24) & 0xffffffull) - 1;
          mem()[3] = (int)(e & 0xffffffull) - 1;
          mem()[0] = sc;  // (this window is looked at again next time)
          return;
        }
        continue;  // another row was quicker: look at the window again
      }
      const int first_unwritten =
          16 - (int)row_reduce<OpMax16>((idx < n && v == 0ull) ? (double)(16 - tid()) : 0.0);
      if (first_unwritten < 16) {
        cursor = sc + first_unwritten;
        break;
      }
      const int nclosed = (int)row_reduce<OpSum16>(closed ? 1.0 : 0.0);
      all_closed = all_closed && nclosed == 16 && sc == head;
      sc += 16;
      cursor = sc < n ? sc : n;
      if (all_closed) {
        head = cursor;
        if (tid() == 0) atomicMax(&ctl[3], cursor);
      }
    }
    mem()[0] = cursor;
  }
  // (never expected) the wavefront waited in vain: ctl[4] tells the host
  FB_COLD void give_up() {
    if (tid() == 0 && waiting()) {
      atomicAdd(&ctl[4], 1);
    }
  }
  // Is the claimed solve ready?  1: pp is bound to its slot and st holds the parked
  // scalars, *q its index; 2: it ended on its old row; 0: not yet.
  template <int NS>
  FB_COLD int take_over(P& pp, double (&st)[NS], int* q) {
    double* ws = slot_ptr(mem()[2]);
    const int wait_q = mem()[3];
    int f = 0;
    if (tid() == 0)
      f = (int)load_flag(P::park_flag(ws, N));
    f = bci<0>(f);
    if (f == 0) return 0;
    if (f == 1) {
      __threadfence();
      pp.resume(tid(), ws, lds(), data, x, wait_q, N, st);
      if (tid() == 0) store_flag(P::park_flag(ws, N), 0.0);
      *q = wait_q;
    }
    mem()[2] = -1;
    return f;
  }
};

// KEEP (FBSTAB_HIP_KEEP_MATRICES): QP q is solved in slot q, so that the slot's
// matrix copies survive from call to call; `reuse` says they are valid already.
template <int NX, int NU, int NC, bool DBG, bool EXACT, bool KEEP = false, int PHASE = -1>
__global__ __launch_bounds__(64, FB_R16_MIN_WAVES) void fbstab_mpc_r16_kernel(
    MpcBatchPtrs data, VarBatchPtrs x, fbstab_solver_out_t* out, fbstab_options_t opts, double* scratch,
    int* counter, int batch, int N, int reuse, double* dbg) {
  typedef MpcR16<NX, NU, NC, EXACT, KEEP> P;
  extern __shared__ __attribute__((aligned(16))) double smem[];
#if defined(FB_ANY_STAMP)
  const long long clk0 = __builtin_readcyclecounter(), rt0 = wall_clock64();
#endif
  const int lane = threadIdx.x;
  Ctx16 ctx;
  ctx.tid = lane & 15;
  P p;
  R16Queue<P, KEEP, PHASE> qu;
  qu.data = &data;
  qu.x = &x;
  if constexpr (PHASE >= 0) {
    // one block of the counter buffer per launch of the batch
    qu.ctl = counter + PHASE * kCompactBlkInts;
    qu.out_blk = qu.ctl;
    if constexpr (PHASE >= 1) qu.in_blk = counter + reuse * kCompactBlkInts;  // (`reuse`: block the list is in)
  } else {
    qu.ctl = counter;
  }
  qu.board = reinterpret_cast<unsigned long long*>(counter + kQueueCtlInts);
  qu.scratch = scratch;
  qu.batch = batch;
  qu.N = N;
  qu.reuse = reuse != 0;
  qu.init();
  if constexpr (DBG) {
    if (qu.fetch(p) >= 0) newton_probe(p, ctx, opts, dbg);
  } else {
    Solver<P, Ctx16> solver(p, ctx, opts);
#ifdef FB_R16_NESTED
    for (;;) {
      const int q = qu.fetch(p);
      if (q < 0) break;
      solver.solve(out + q);
    }
#else
    solver.solve_stream(qu, out);
#endif
  }
#if defined(FB_ANY_STAMP)
  // shader clock actually delivered to this wavefront: s_memtime vs the 100 MHz counter
  if (threadIdx.x == 0) {
    atomicAdd(&g_stamps[28], (unsigned long long)(__builtin_readcyclecounter() - clk0));
    atomicAdd(&g_stamps[29], (unsigned long long)(wall_clock64() - rt0));
  }
#endif
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

template <int NT, bool TRACE = false, bool KGLOBAL = false>
__global__ __launch_bounds__(NT) void fbstab_dense_kernel(DenseLayout lay, DenseBatchArgs data,
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
    const int q = next_qp<NT>(counter, lds + lay.o_rhs);
    if (q >= batch) break;
    DenseData D;
    D.H = data.base[FBSTAB_DENSE_H] + q * data.stride[FBSTAB_DENSE_H];
    D.f = data.base[FBSTAB_DENSE_f] + q * data.stride[FBSTAB_DENSE_f];
    D.G = data.base[FBSTAB_DENSE_G] + q * data.stride[FBSTAB_DENSE_G];
    D.h = data.base[FBSTAB_DENSE_h] + q * data.stride[FBSTAB_DENSE_h];
    D.A = data.base[FBSTAB_DENSE_A] + q * data.stride[FBSTAB_DENSE_A];
    D.b = data.base[FBSTAB_DENSE_b] + q * data.stride[FBSTAB_DENSE_b];
    DenseProblem<C, KGLOBAL> p;
    double* ks = nullptr;
    if constexpr (KGLOBAL) ks = kscratch.get() + (long)blockIdx.x * lay.k_doubles;
    p.bind(lay, D, x.base[0] + q * x.stride[0], x.base[1] + q * x.stride[1],
           x.base[2] + q * x.stride[2], x.base[3] + q * x.stride[3], lds, ks);
    Solver<DenseProblem<C, KGLOBAL>, C, TRACE> solver(p, ctx, opts, trace.get());
    solver.solve(out + q);
    ctx.sync();
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

  int release() {
    (void)hipSetDevice(device);
    for (double* p : d_arr)
      if (p) (void)hipFree(p);
    for (int i = 0; i < 4; i++)
      if (d_var[i]) (void)hipFree(d_var[i]);
    if (d_out) (void)hipFree(d_out);
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
    HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&ev0));
    HIP_TRY(hipEventCreate(&ev1));
    HIP_TRY(hipMalloc(&counter, kQueueBytes));  // queue counter (+ the record kernel's board)
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
  bool g16 = false;       // 16-lane register kernel (four QPs per wavefront)
  bool r16 = false;       // record-based 16-lane kernel (fb_mpc_r16.h), the default for its shapes
  int kept_batch = -1;    // batch size of the last FBSTAB_HIP_KEEP_MATRICES call whose copies are still in the slots
  int compact_phases = 1; // launches per batch of the record kernel (tail compaction, R16Queue)
  int lds_per_row = 0;
  int qps_per_wg = 1;
};

namespace {
// The specialised shapes compiled into the library: the flat-layout register
// kernel needs the exact shape, the record kernel runs anything that fits its
// (12, 4, 20) instance zero-padded.
bool g16_shape(int nx, int nu, int nc) { return nx == 12 && nu == 4 && nc == 20; }
bool r16_fits(int nx, int nu, int nc) { return nx <= 12 && nu <= 4 && nc <= 20; }

template <class... A>
void launch_mpc(fbstab_mpc_solver* h, int grid, hipStream_t s, A... args) {
  if (h->r16) {
    // unreachable: the record kernel takes a different argument list (launch_r16)
  } else if (h->g16) {
    hipLaunchKernelGGL((fbstab_mpc_g16_kernel<12, 4, 20, false>), dim3(grid), dim3(64), h->lds_bytes, s,
                       args..., h->lds_per_row, (double*)nullptr);
  } else {
    hipLaunchKernelGGL((fbstab_mpc_kernel<kMpcThreads, false>), dim3(grid), dim3(h->threads), h->lds_bytes, s,
                       args..., (double*)nullptr);
  }
}
template <bool DBG>
void launch_r16(fbstab_mpc_solver* h, int grid, hipStream_t s, const MpcBatchArgs& a, const VarBatchArgs& v,
                fbstab_solver_out_t* out, int batch, double* dbg, bool keep = false, bool reuse = false) {
  MpcBatchPtrs d;
  VarBatchPtrs x;
  for (int i = 0; i < FBSTAB_MPC_NSEQ; i++) { d.base[i] = a.base[i]; d.stride[i] = a.stride[i]; }
  for (int i = 0; i < 4; i++) { x.base[i] = v.base[i]; x.stride[i] = v.stride[i]; }
  d.nx = h->lay.nx;
  d.nu = h->lay.nu;
  d.nc = h->lay.nc;
  const bool exact = g16_shape(h->lay.nx, h->lay.nu, h->lay.nc);
  const int ru = reuse ? 1 : 0;
  if constexpr (!DBG) {
    if (keep) {
      if (exact)
        hipLaunchKernelGGL((fbstab_mpc_r16_kernel<12, 4, 20, false, true, true>), dim3(grid), dim3(64), h->lds_bytes,
                           s, d, x, out, h->opts, h->scratch, h->counter, batch, h->lay.N, ru, dbg);
      else
        hipLaunchKernelGGL((fbstab_mpc_r16_kernel<12, 4, 20, false, false, true>), dim3(grid), dim3(64), h->lds_bytes,
                           s, d, x, out, h->opts, h->scratch, h->counter, batch, h->lay.N, ru, dbg);
      return;
    }
  }
  if constexpr (!DBG && kCompactBatches) {
    // tail compaction (R16Queue): up to three launches, each at most two parked solves
    // per wavefront of the one before, four per wavefront of its own
    if (h->compact_phases >= 2) {
      const int g1 = (grid + 1) / 2, g2 = (g1 + 1) / 2;
#define FB_LAUNCH_PHASE(EX, PH, G, IN)                                                                             \
  hipLaunchKernelGGL((fbstab_mpc_r16_kernel<12, 4, 20, false, EX, false, PH>), dim3(G), dim3(64), h->lds_bytes, s, d, \
                     x, out, h->opts, h->scratch, h->counter, batch, h->lay.N, IN, dbg)
      if (exact) {
        FB_LAUNCH_PHASE(true, 0, grid, 0);
        if (h->compact_phases >= 3) {
          FB_LAUNCH_PHASE(true, 1, g1, 0);
          FB_LAUNCH_PHASE(true, 2, g2, 1);
        } else {
          FB_LAUNCH_PHASE(true, 2, g1, 0);
        }
      } else {
        FB_LAUNCH_PHASE(false, 0, grid, 0);
        if (h->compact_phases >= 3) {
          FB_LAUNCH_PHASE(false, 1, g1, 0);
          FB_LAUNCH_PHASE(false, 2, g2, 1);
        } else {
          FB_LAUNCH_PHASE(false, 2, g1, 0);
        }
      }
#undef FB_LAUNCH_PHASE
      return;
    }
  }
  if (exact)
    hipLaunchKernelGGL((fbstab_mpc_r16_kernel<12, 4, 20, DBG, true>), dim3(grid), dim3(64), h->lds_bytes, s, d, x, out,
                       h->opts, h->scratch, h->counter, batch, h->lay.N, 0, dbg);
  else
    hipLaunchKernelGGL((fbstab_mpc_r16_kernel<12, 4, 20, DBG, false>), dim3(grid), dim3(64), h->lds_bytes, s, d, x,
                       out, h->opts, h->scratch, h->counter, batch, h->lay.N, 0, dbg);
}
}  // namespace
struct fbstab_dense_solver : SolverBase {
  fbk::DenseLayout lay;
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
  if (!handle) return fail(FBSTAB_HIP_ERR_ARGUMENT, "null handle pointer");
  *handle = nullptr;
  // fbstab_mpc.cc:62-65
  if (N < 1 || nx < 1 || nu < 1 || nc < 1)
    return fail(FBSTAB_HIP_ERR_ARGUMENT, "In FBstabMpc::FBstabMpc: problem sizes must be positive.");
  if (max_batch < 1) return fail(FBSTAB_HIP_ERR_ARGUMENT, "max_batch must be positive");
  fbstab_mpc_solver* s = new (std::nothrow) fbstab_mpc_solver();
  if (!s) return fail(FBSTAB_HIP_ERR_DEVICE, "out of host memory");
  s->threads = kMpcThreads;
  s->lay.init(N, nx, nu, nc, s->threads);
  s->lds_bytes = s->lay.lds_doubles * (int)sizeof(double);
  const char* force_generic = getenv("FBSTAB_HIP_GENERIC");
  s->g16 = g16_shape(nx, nu, nc) && !(force_generic && atoi(force_generic) > 0);
  // FBSTAB_HIP_MPC_KERNEL=g16 selects the previous register kernel (comparisons)
  const char* which = getenv("FBSTAB_HIP_MPC_KERNEL");
  s->r16 = r16_fits(nx, nu, nc) && !(force_generic && atoi(force_generic) > 0) &&
           !(which && strcmp(which, "g16") == 0);
  typedef fbk::MpcR16<12, 4, 20> R16;
  if (s->r16) {
    s->g16 = false;
    s->lds_per_row = R16::kLdsPerRow;
    s->qps_per_wg = 4;
    s->lds_bytes = 4 * R16::kLdsPerRow * (int)sizeof(double);
  }
  if (s->g16) {
    int d = s->lay.w_sb;  // tile + stage slices of the generic passes
    if (d < fbk::MpcProblemG16<12, 4, 20>::kLdsDoubles) d = fbk::MpcProblemG16<12, 4, 20>::kLdsDoubles;
    // region stride == 16 doubles mod 32 (bank placement, see fb_mpc_g16.h)
    s->lds_per_row = ((d + 31) & ~31) + 16;
    s->qps_per_wg = 4;
    s->lds_bytes = 4 * s->lds_per_row * (int)sizeof(double);
  }
  if (nx > s->threads || s->lds_bytes > kLdsLimitBytes) {
    delete s;
    return fail(FBSTAB_HIP_ERR_UNSUPPORTED, "stage matrices do not fit the 160 KiB LDS budget");
  }
  int rc = s->common_init(device, max_batch);
  if (rc != FBSTAB_HIP_OK) { s->release(); delete s; return rc; }
  const bool exact = g16_shape(nx, nu, nc);
  const void* r16_first =
      kCompactBatches ? (exact ? reinterpret_cast<const void*>(fbstab_mpc_r16_kernel<12, 4, 20, false, true, false, 0>)
                               : reinterpret_cast<const void*>(fbstab_mpc_r16_kernel<12, 4, 20, false, false, false, 0>))
                      : (exact ? reinterpret_cast<const void*>(fbstab_mpc_r16_kernel<12, 4, 20, false, true>)
                               : reinterpret_cast<const void*>(fbstab_mpc_r16_kernel<12, 4, 20, false, false>));
  const void* kern = s->r16   ? r16_first
                     : s->g16 ? reinterpret_cast<const void*>(fbstab_mpc_g16_kernel<12, 4, 20, false>)
                              : reinterpret_cast<const void*>(fbstab_mpc_kernel<kMpcThreads, false>);
  const void* kern_dbg = s->r16   ? (exact ? reinterpret_cast<const void*>(fbstab_mpc_r16_kernel<12, 4, 20, true, true>)
                                           : reinterpret_cast<const void*>(fbstab_mpc_r16_kernel<12, 4, 20, true, false>))
                         : s->g16 ? reinterpret_cast<const void*>(fbstab_mpc_g16_kernel<12, 4, 20, true>)
                                  : reinterpret_cast<const void*>(fbstab_mpc_kernel<kMpcThreads, true>);
  hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, s->lds_bytes);
  if (e == hipSuccess && s->r16) {
    const void* kk = exact ? reinterpret_cast<const void*>(fbstab_mpc_r16_kernel<12, 4, 20, false, true, true>)
                           : reinterpret_cast<const void*>(fbstab_mpc_r16_kernel<12, 4, 20, false, false, true>);
    e = hipFuncSetAttribute(kk, hipFuncAttributeMaxDynamicSharedMemorySize, s->lds_bytes);
    if (kCompactBatches) {
      const void* k1 = exact ? reinterpret_cast<const void*>(fbstab_mpc_r16_kernel<12, 4, 20, false, true, false, 1>)
                             : reinterpret_cast<const void*>(fbstab_mpc_r16_kernel<12, 4, 20, false, false, false, 1>);
      const void* k2 = exact ? reinterpret_cast<const void*>(fbstab_mpc_r16_kernel<12, 4, 20, false, true, false, 2>)
                             : reinterpret_cast<const void*>(fbstab_mpc_r16_kernel<12, 4, 20, false, false, false, 2>);
      if (e == hipSuccess) e = hipFuncSetAttribute(k1, hipFuncAttributeMaxDynamicSharedMemorySize, s->lds_bytes);
      if (e == hipSuccess) e = hipFuncSetAttribute(k2, hipFuncAttributeMaxDynamicSharedMemorySize, s->lds_bytes);
    }
  }
  if (e == hipSuccess) e = hipFuncSetAttribute(kern_dbg, hipFuncAttributeMaxDynamicSharedMemorySize, s->lds_bytes);
  int per_cu = 0, cus = 0;
  if (e == hipSuccess)
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, s->threads, s->lds_bytes);
  hipDeviceProp_t prop;
  if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) {
    s->release(); delete s;
    return fail(FBSTAB_HIP_ERR_DEVICE, std::string("occupancy query: ") + hipGetErrorString(e));
  }
  cus = prop.multiProcessorCount;
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 8) per_cu = 8;
  const char* env = getenv("FBSTAB_HIP_WGS_PER_CU");
  if (env && atoi(env) > 0) per_cu = atoi(env);
  {
    const char* cp = getenv("FBSTAB_HIP_COMPACT_PHASES");
    if (cp && atoi(cp) >= 1 && atoi(cp) <= 3) s->compact_phases = atoi(cp);
  }
  s->workgroups = cus * per_cu;
  if (s->r16 && 2 * s->workgroups > kCompactCap) s->workgroups = kCompactCap / 2;  // (list capacity, R16Queue)
  {
    const int need = (max_batch + s->qps_per_wg - 1) / s->qps_per_wg;
    if (s->workgroups > need) s->workgroups = need;
  }
  const long long ws_doubles = s->r16 ? (long long)R16::ws_doubles(N) : (long long)s->lay.ws_doubles;
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
static int mpc_solve_impl(fbstab_mpc_handle_t h, int batch, const fbstab_mpc_batch_t* data,
                          const fbstab_var_batch_t* x, fbstab_solver_out_t* out, int flags,
                          void* stream, double* d_trace) {
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
  if (dev_ptrs) {
    for (int i = 0; i < FBSTAB_MPC_NSEQ; i++) { a.base[i] = data->base[i]; a.stride[i] = data->stride[i]; }
    for (int i = 0; i < 4; i++) { v.base[i] = x->base[i]; v.stride[i] = x->stride[i]; }
  } else {
    rc = h->ensure_staging();
    if (rc != FBSTAB_HIP_OK) return rc;
    for (int i = 0; i < FBSTAB_MPC_NSEQ; i++) {
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
  HIP_TRY(hipMemsetAsync(h->counter, 0, h->r16 ? kQueueBytes : sizeof(int), s));
  int grid = (batch + h->qps_per_wg - 1) / h->qps_per_wg;
  if (grid > h->workgroups) grid = h->workgroups;
  HIP_TRY(hipEventRecord(h->ev0, s));
  DevBuf tmp_ws;
  if (d_trace) {
    const int lds = h->lay.lds_doubles * (int)sizeof(double);
    if (h->lay.nx > kMpcThreads || lds > kLdsLimitBytes)
      return fail(FBSTAB_HIP_ERR_UNSUPPORTED, "stage matrices do not fit the 160 KiB LDS budget");
    auto kern = fbstab_mpc_kernel<kMpcThreads, false, true>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIP_TRY(hipMalloc(&tmp_ws.p, sizeof(double) * (size_t)h->lay.ws_doubles));
    hipLaunchKernelGGL(kern, dim3(1), dim3(kMpcThreads), lds, s, h->lay, a, v, d_out, h->opts,
                       static_cast<double*>(tmp_ws.p), h->counter, 1, d_trace);
  } else if (h->r16) {
    // FBSTAB_HIP_KEEP_MATRICES: one QP per slot, slot = QP index
    const bool keep = (flags & FBSTAB_HIP_KEEP_MATRICES) && dev_ptrs && batch <= h->workgroups * h->qps_per_wg;
    const bool reuse = keep && h->kept_batch == batch;
    launch_r16<false>(h, keep ? (batch + 3) / 4 : grid, s, a, v, d_out, batch, nullptr, keep, reuse);
    h->kept_batch = keep ? batch : -1;
  } else {
    launch_mpc(h, grid, s, h->lay, a, v, d_out, h->opts, h->scratch, h->counter, batch);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(h->ev1, s));
  h->timed = true;
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
  } else if (!(flags & FBSTAB_HIP_ASYNC)) {
    HIP_TRY(hipStreamSynchronize(s));
  }
  return FBSTAB_HIP_OK;
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
  double* d_io = nullptr;
  HIP_TRY(hipMalloc(&d_io, n_io * sizeof(double)));
  HIP_TRY(hipMemcpyAsync(d_io, io, sizeof(double) * (L.nz + L.nl + L.nv), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemsetAsync(h->counter, 0, h->r16 ? kQueueBytes : sizeof(int), s));
  if (h->r16) {
    launch_r16<true>(h, 1, s, a, v, h->d_out, 1, d_io);
  } else if (h->g16) {
    hipLaunchKernelGGL((fbstab_mpc_g16_kernel<12, 4, 20, true>), dim3(1), dim3(64), h->lds_bytes, s, h->lay, a, v,
                       h->d_out, h->opts, h->scratch, h->counter, 1, h->lds_per_row, d_io);
  } else {
    hipLaunchKernelGGL((fbstab_mpc_kernel<kMpcThreads, true>), dim3(1), dim3(h->threads), h->lds_bytes, s, h->lay, a,
                       v, h->d_out, h->opts, h->scratch, h->counter, 1, d_io);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(io, d_io, sizeof(double) * n_io, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipFree(d_io));
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
  s->threads = kDenseThreads;
  s->lay.init(nz, nl, nv, s->threads);
  s->lds_bytes = s->lay.lds_doubles * (int)sizeof(double);
  if (s->lds_bytes > kLdsLimitBytes) {
    delete s;
    return fail(FBSTAB_HIP_ERR_UNSUPPORTED, "the iterate vectors do not fit the 160 KiB LDS budget");
  }
  int rc = s->common_init(device, max_batch);
  if (rc != FBSTAB_HIP_OK) { s->release(); delete s; return rc; }
  const void* kern = s->lay.k_global
                         ? reinterpret_cast<const void*>(fbstab_dense_kernel<kDenseThreads, false, true>)
                         : reinterpret_cast<const void*>(fbstab_dense_kernel<kDenseThreads>);
  hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, s->lds_bytes);
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
  if (s->lay.k_global) {  // K of every resident workgroup (fb_dense.h)
    s->scratch_bytes = (long long)sizeof(double) * s->lay.k_doubles * s->workgroups;
    e = hipMalloc(&s->scratch, (size_t)s->scratch_bytes);
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
                            void* stream, double* d_trace) {
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
  if (dev_ptrs) {
    for (int i = 0; i < FBSTAB_DENSE_NARR; i++) { a.base[i] = data->base[i]; a.stride[i] = data->stride[i]; }
    for (int i = 0; i < 4; i++) { v.base[i] = x->base[i]; v.stride[i] = x->stride[i]; }
  } else {
    rc = h->ensure_staging();
    if (rc != FBSTAB_HIP_OK) return rc;
    for (int i = 0; i < FBSTAB_DENSE_NARR; i++) {
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
  HIP_TRY(hipMemsetAsync(h->counter, 0, sizeof(int), s));
  int grid = h->workgroups < batch ? h->workgroups : batch;
  HIP_TRY(hipEventRecord(h->ev0, s));
  if (d_trace && h->lay.k_global) {
    auto kern = fbstab_dense_kernel<kDenseThreads, true, true>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_bytes));
    hipLaunchKernelGGL(kern, dim3(1), dim3(h->threads), h->lds_bytes, s, h->lay, a, v, d_out, h->opts,
                       h->counter, 1, TraceArg<true>{d_trace}, KScratchArg<true>{h->scratch});
  } else if (d_trace) {
    auto kern = fbstab_dense_kernel<kDenseThreads, true>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_bytes));
    hipLaunchKernelGGL(kern, dim3(1), dim3(h->threads), h->lds_bytes, s, h->lay, a, v, d_out, h->opts,
                       h->counter, 1, TraceArg<true>{d_trace}, KScratchArg<false>());
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
  } else if (!(flags & FBSTAB_HIP_ASYNC)) {
    HIP_TRY(hipStreamSynchronize(s));
  }
  return FBSTAB_HIP_OK;
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

double fbstab_hip_dense_last_kernel_ms(fbstab_dense_handle_t h) { return h ? h->last_kernel_ms() : -1.0; }

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

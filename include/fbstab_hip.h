/*
 * C-ABI of the MI355X-native batched FBstab solver (libfbstab_hip.so).
 *
 * This is the drop-in boundary for the reference's inner-loop path: one call
 * solves a whole batch of QPs with the FBstab algorithm running entirely on
 * the GPU (one QP per workgroup).  Each entry point names the reference
 * interface it stands in for (paths relative to dliaomcp/fbstab):
 *
 *   fbstab_hip_mpc_create / _destroy    FBstabMpc::FBstabMpc(N,nx,nu,nc) / dtor
 *                                       fbstab/fbstab_mpc.h:165, fbstab_mpc.cc:61-89
 *   fbstab_hip_mpc_set_options          FBstabMpc::UpdateOptions, fbstab_mpc.h:202,
 *                                       fbstab_mpc.cc:96-100 (-> UpdateParameters +
 *                                       ValidateOptions, fbstab_algorithm-impl.h:307-332)
 *   fbstab_hip_mpc_solve_batch          FBstabMpc::Solve(qp, &x), fbstab_mpc.h:181-195
 *                                       (batch == 1 with host pointers is exactly one
 *                                       reference Solve call)
 *   fbstab_hip_mpc_solve_batch_final    the same Solve at Display::FINAL, the reference's default
 *                                       level: the batch kernels, then the |rz| |rl| |rv| and
 *                                       tolerance of the summary block PrintFinal prints
 *                                       (fbstab_algorithm-impl.h:493-541)
 *   fbstab_hip_mpc_solve_traced         the same Solve with Display::ITER / ITER_DETAILED:
 *                                       the numbers of PrintIterLine, PrintDetailedHeader/
 *                                       Line/Footer and PrintFinal
 *                                       (fbstab_algorithm-impl.h:411-541) as records
 *   fbstab_hip_dense_create / _destroy  FBstabDense::FBstabDense(nz,nl,nv),
 *                                       fbstab/fbstab_dense.h:122, fbstab_dense.cc:18-42
 *   fbstab_hip_dense_set_options        FBstabDense::UpdateOptions, fbstab_dense.h:158
 *   fbstab_hip_dense_solve_batch        FBstabDense::Solve(qp, &x), fbstab_dense.h:136-149
 *   fbstab_hip_dense_solve_traced       as fbstab_hip_mpc_solve_traced
 *
 * Data layout is the reference's: every MPC sequence is a MatrixSequence image
 * data[k*nr*nc + j*nr + i] (tools/matrix_sequence.h:81-83), dense matrices are
 * column-major (Eigen::MatrixXd).  A batch is described by one base pointer
 * and one stride (in doubles) per array: QP b lives at base + b*stride, so both
 * "array of structures" (all sequences of a QP contiguous) and "structure of
 * arrays" placements work, and a stride of 0 shares an array across the batch.
 *
 * Error behaviour: no exception crosses this boundary.  Functions return
 * FBSTAB_HIP_OK or an error code and fbstab_hip_last_error() describes the
 * failure (the C++ facade in include/fbstab/ turns these into the
 * std::runtime_error the reference throws).  Per-QP outcomes are reported in
 * fbstab_solver_out_t::eflag; a per-QP factorisation failure, where the
 * reference would throw out of Solve (fbstab_algorithm-impl.h:263-274), is
 * reported as FBSTAB_DIVERGENCE for that QP only.
 *
 * There is no CPU execution path in this library: without a usable HIP device
 * every create call fails with FBSTAB_HIP_ERR_DEVICE.
 */
#ifndef FBSTAB_HIP_H_
#define FBSTAB_HIP_H_

#include "fbstab_types.h"

#ifdef __cplusplus
extern "C" {
#endif

enum fbstab_hip_status {
  FBSTAB_HIP_OK = 0,
  FBSTAB_HIP_ERR_ARGUMENT = 1,    /* null pointer, non-positive size, batch > max_batch */
  FBSTAB_HIP_ERR_DEVICE = 2,      /* HIP runtime error / no device */
  FBSTAB_HIP_ERR_UNSUPPORTED = 3  /* problem does not fit the on-chip budget */
};

/* Where the caller's arrays live, and whether the call may return before the
 * GPU has finished (device memory only; the caller then syncs the stream). */
enum fbstab_hip_flags {
  FBSTAB_HIP_HOST_POINTERS = 0,
  FBSTAB_HIP_DEVICE_POINTERS = 1,
  FBSTAB_HIP_ASYNC = 2,
  /* Receding-horizon hint (MPC, device pointers): the matrix sequences
   * Q, R, S, A, B, E, L of every QP of this call are the ones of the previous
   * call on this handle that carried the flag (same batch size); only
   * q, r, c, d, x0 and the initial guess may have changed.  The library then
   * keeps its per-QP copies of the matrices between calls instead of
   * rebuilding them.  The first flagged call, and a flagged call after an
   * unflagged one, build them.  Ignored where it cannot be honoured (batch
   * larger than the resident QP slots, kernels without such copies); results
   * are the same with or without it. */
  FBSTAB_HIP_KEEP_MATRICES = 4,
  /* With FBSTAB_HIP_DEVICE_POINTERS: `out` is a HOST array all the same.  The call
   * then waits for the solve and copies the SolverOut records back (the zero-copy
   * single-QP path of the C++ facade: ProblemDataRef / VariableRef over device
   * memory, fbstab_mpc.h:90-150, with the SolverOut returned by value). */
  FBSTAB_HIP_OUT_ON_HOST = 8
};

/* Index of each MPC sequence in fbstab_mpc_batch_t (FBstabMpc::ProblemData
 * member order, fbstab/fbstab_mpc.h:67-81). */
enum fbstab_mpc_seq {
  FBSTAB_MPC_Q = 0, FBSTAB_MPC_R, FBSTAB_MPC_S, FBSTAB_MPC_q, FBSTAB_MPC_r,
  FBSTAB_MPC_A, FBSTAB_MPC_B, FBSTAB_MPC_c, FBSTAB_MPC_E, FBSTAB_MPC_L,
  FBSTAB_MPC_d, FBSTAB_MPC_x0, FBSTAB_MPC_NSEQ
};

/* Index of each dense array (FBstabDense::ProblemData, fbstab_dense.h:55-64). */
enum fbstab_dense_arr {
  FBSTAB_DENSE_H = 0, FBSTAB_DENSE_f, FBSTAB_DENSE_G, FBSTAB_DENSE_h,
  FBSTAB_DENSE_A, FBSTAB_DENSE_b, FBSTAB_DENSE_NARR
};

typedef struct fbstab_mpc_batch_t {
  const double* base[FBSTAB_MPC_NSEQ];
  long long stride[FBSTAB_MPC_NSEQ]; /* doubles between consecutive QPs */
} fbstab_mpc_batch_t;

typedef struct fbstab_dense_batch_t {
  const double* base[FBSTAB_DENSE_NARR];
  long long stride[FBSTAB_DENSE_NARR];
} fbstab_dense_batch_t;

/* Initial guess in (z, l, v), solution out (z, l, v, y); y is ignored on input
 * (FBstabMpc::Variable, fbstab_mpc.h:126-136; fbstab_algorithm-impl.h:334-347). */
typedef struct fbstab_var_batch_t {
  double* base[4];       /* z, l, v, y */
  long long stride[4];
} fbstab_var_batch_t;

typedef struct fbstab_mpc_solver* fbstab_mpc_handle_t;
typedef struct fbstab_dense_solver* fbstab_dense_handle_t;

const char* fbstab_hip_last_error(void);
int fbstab_hip_device_count(void);

/* ---- MPC ---------------------------------------------------------------- */
int fbstab_hip_mpc_create(int N, int nx, int nu, int nc, int max_batch, int device,
                          fbstab_mpc_handle_t* handle);
/* The same for a caller that keeps `handles_in_flight` handles busy on this device at the same time - one
 * batch each, every handle on a stream of its own (the reference has no counterpart: FBstabMpc is a
 * single-threaded CPU object, fbstab/fbstab_mpc.h:56-60).  A batch launch is a persistent grid that pulls QPs
 * from a queue; alone on the device it wants every resident wavefront slot (four workgroups per CU on the
 * BASELINE shape), but with several launches in flight the others fill the device and a launch does better
 * with its share: each row of a wavefront then gets more QPs of the batch and the tail of the launch - rows
 * that have run out of QPs while their wavefront's last one finishes - shrinks (rows busy 0.95 instead of 0.82
 * per Newton step at eight in flight; +2.7 % throughput, and a quarter of the scratch memory: 0.44 instead
 * of 1.75 GB per handle).  handles_in_flight = 1 is fbstab_hip_mpc_create.
 * What the share costs a handle that is then used ALONE: it keeps its fraction of the grid - one eighth of
 * the workgroups at handles_in_flight = 8 (never fewer than one per CU), so a lone launch fills an eighth of
 * the chip - and fbstab_hip_mpc_receding_sweep, whose one-launch form needs batch <= workgroups x QPs per
 * workgroup, falls back to one launch per step at a batch that much smaller (same results, bitwise; slower).
 * fbstab_hip_mpc_query reports the handle's `workgroups` and `scratch_bytes` as created, so a caller can see
 * the share it got.  Values outside 1..64 are refused (FBSTAB_HIP_ERR_ARGUMENT).
 * The share is a hint measured on the BASELINE shape.  Shapes of the widest record instance (stage width up
 * to 32 with up to 16 constraint rows) did better as a stream of batches with
 * every handle keeping the whole grid (handles_in_flight = 1 on each of eight handles: 28 k against 24 k
 * QPs/sec on (30,20,6,16)) - measure both on a new shape. */
int fbstab_hip_mpc_create_in_flight(int N, int nx, int nu, int nc, int max_batch, int device,
                                    int handles_in_flight, fbstab_mpc_handle_t* handle);
int fbstab_hip_mpc_destroy(fbstab_mpc_handle_t handle);
int fbstab_hip_mpc_set_options(fbstab_mpc_handle_t handle, const fbstab_options_t* options);
int fbstab_hip_mpc_get_options(fbstab_mpc_handle_t handle, fbstab_options_t* options);
/* stream: a hipStream_t, or NULL for the handle's own stream, which is a blocking
 * stream (ordered against the device's null stream both ways).  With
 * FBSTAB_HIP_DEVICE_POINTERS the caller's arrays must be ready on the stream the
 * call runs on: work queued on another non-blocking stream needs an event. */
int fbstab_hip_mpc_solve_batch(fbstab_mpc_handle_t handle, int batch,
                               const fbstab_mpc_batch_t* data, const fbstab_var_batch_t* x,
                               fbstab_solver_out_t* out, int flags, void* stream);
/* fbstab_hip_mpc_solve_batch followed, on the same stream, by the numbers of the
 * summary block the reference prints at Display::FINAL (PrintFinal,
 * fbstab_algorithm-impl.h:493-541): norms[4 q .. 4 q + 3] = {|rz|, |rl|, |rv|} of the
 * penalised natural residual (full_residual.cc:99-109) at the point QP q returned, and
 * the stopping tolerance abs_tol + rel_tol (1 + ||(f, h, b)||) (impl:137).  `norms` lives
 * where `out` lives (host for host-pointer calls and with FBSTAB_HIP_OUT_ON_HOST, device
 * otherwise).  Same kernels, same iteration counts as solve_batch.  For SUCCESS and the
 * Newton-iteration limit these are the numbers the reference prints (its rk_ is evaluated
 * at the returned point, impl:162-170, :188-199); for infeasibility exits (impl:204-212:
 * the certificate is returned, rk_ belongs to x(k)) and the proximal-iteration limit
 * (impl:219-223: rk_ is one iteration old) the reference prints a residual of a point the
 * solve does not return - callers that need that text use solve_traced. */
int fbstab_hip_mpc_solve_batch_final(fbstab_mpc_handle_t handle, int batch,
                                     const fbstab_mpc_batch_t* data, const fbstab_var_batch_t* x,
                                     fbstab_solver_out_t* out, double* norms, int flags, void* stream);
/* ONE QP given by host pointers, solved synchronously, with the per-iteration
 * display of the reference returned as data: every line the reference's
 * Display::ITER and ITER_DETAILED levels would print during this solve
 * (fbstab_algorithm-impl.h:155-172, :250-257, :381) becomes one
 * fbstab_trace_record_t, in the reference's print order; the caller formats
 * the kinds its display level shows (the C++ facade does, PrintTrace in
 * include/fbstab/fbstab_algorithm.h).  At most `capacity` records are stored;
 * *count receives the number produced.  Runs on the flat-vector kernel (one
 * QP per wavefront) whatever kernel the handle uses for batches, so iteration
 * counts can differ from a solve_batch call within the tolerance the parity
 * tests state. */
int fbstab_hip_mpc_solve_traced(fbstab_mpc_handle_t handle, const fbstab_mpc_batch_t* data,
                                const fbstab_var_batch_t* x, fbstab_solver_out_t* out,
                                fbstab_trace_record_t* trace, int capacity, int* count);
/* Warm-started receding-horizon sweep on the device (BASELINE configs[4]): `steps`
 * closed-loop steps of `batch` independent trajectories queued on one stream with
 * no host round trip in between.  Step k solves the batch as
 * fbstab_hip_mpc_solve_batch would (FBstabMpc::Solve, fbstab_mpc.h:181-195, with
 * the previous step's (z, l, v) as the initial guess, unshifted - what a caller of
 * the reference gets by passing the same Variable again, fbstab_algorithm-impl.h:140)
 * and then advances every trajectory's initial state with the simulation model the
 * reference's generator hands out (OcpGenerator::SimulationInputs,
 * fbstab/test/ocp_generator.h:31-38):  x0 <- A x0 + B u0,  u0 = the first input of
 * the solution.  All pointers are DEVICE pointers; data->base[FBSTAB_MPC_x0] is
 * updated in place (it must be writable), x holds the last step's solution on
 * return, out its SolverOut records.  The matrix sequences must not change during
 * the sweep (FBSTAB_HIP_KEEP_MATRICES semantics).
 *   retire != 0: a trajectory whose solve does not end in SUCCESS is parked at the
 *                origin for the rest of the sweep (x0 = 0, zero guess, u0 = 0).
 *   u_log:   NULL or device array [steps][batch][nu] receiving every u0.
 *   stats:   NULL or HOST array [steps][4]: sum of Newton iterations, solves ended in
 *            SUCCESS, trajectories retired so far, largest Newton count of the step.
 *   kernel_ms: NULL or HOST array [steps]: device time of each step's solve.
 * Shapes served by a record kernel run the whole sweep as ONE launch in which every
 * trajectory advances at its own pace (no trajectory waits for the slowest solve of a
 * step; same results per trajectory and per step); kernel_ms[k] is then the launch
 * time / steps for every k.  Other shapes (or FBSTAB_HIP_SWEEP_PER_STEP=1) queue one
 * solve launch and one plant launch per step.
 * Synchronous: returns when the sweep has finished. */
typedef struct fbstab_receding_plant_t {
  const double* A;      /* nx x nx, column-major */
  const double* B;      /* nx x nu, column-major */
  long long stride_A;   /* doubles between trajectories; 0 = one plant for all */
  long long stride_B;
} fbstab_receding_plant_t;
int fbstab_hip_mpc_receding_sweep(fbstab_mpc_handle_t handle, int batch, const fbstab_mpc_batch_t* data,
                                  const fbstab_var_batch_t* x, fbstab_solver_out_t* out,
                                  const fbstab_receding_plant_t* plant, int steps, int retire,
                                  double* u_log, unsigned long long* stats, float* kernel_ms,
                                  void* stream);

/* Device time of the solver kernel in the most recent solve_batch call on this
 * handle, measured with HIP events on the stream it ran on (ms; < 0 if none). */
double fbstab_hip_mpc_last_kernel_ms(fbstab_mpc_handle_t handle);
/* Bytes of device scratch and of LDS per workgroup the handle uses, and the
 * number of resident workgroups it launches (for DESIGN.md / diagnostics).  (Record kernels, round 6: a batch
 * of no more QPs than `workgroups` is spread one QP per wavefront - a handle created for a small max_batch
 * therefore keeps one workgroup, and four QP slots of scratch, per QP - LABNOTES R6.6.) */
int fbstab_hip_mpc_query(fbstab_mpc_handle_t handle, long long* scratch_bytes,
                         int* lds_bytes, int* workgroups, int* threads);

/* Name of the kernel instance this handle's batches run on, e.g.
 * "fbstab_mpc_r16_kernel<12,4,20>" (record kernel, four QPs per wavefront) or
 * "fbstab_mpc_kernel<64>" (any shape, one QP per wavefront). */
const char* fbstab_hip_mpc_kernel_name(fbstab_mpc_handle_t handle);

/* Iterative refinement of the Newton steps - an option, OFF by default (fbstab_options_t::reserved = 0).
 * The kernels measure with every Newton step what the linear solve left of the Newton system (the z and l
 * share of the first line-search trial's norm).  With reserved = k > 0 a step whose leftover exceeds
 * 2^(1 - k) of the tolerance the solver's next tests compare against is solved once more for its
 * residual with the same factors and corrected (eps |V| |dx| in every block row afterwards): a linear
 * solve more accurate than RiccatiLinearSolver::Solve's (fbstab/components/riccati_linear_solver.cc:
 * 212-344), hence not the default - iteration counts then part from the reference's where the reference's
 * own rounding decides a stopping test.  This call returns the number of Newton steps of the most recent
 * solve_batch / receding_sweep call that were refined (waits for that launch; -1 if there was none;
 * 0 with the option off). */
int fbstab_hip_mpc_refined_steps(fbstab_mpc_handle_t handle, long long* steps);

/* Diagnostics used by the parity tests: runs ONE Newton step of the device path
 * (LinearSolver::Initialize + Solve of the reference, abstract_components.h:291-338)
 * for a single QP given by host pointers.  io: [zbar, lbar, vbar] in,
 * [dz, dl, dv, A*dz, W_z, W_l, r_z, r_l, ok] out. */
int fbstab_hip_mpc_debug_newton(fbstab_mpc_handle_t handle, const fbstab_mpc_batch_t* data,
                                const fbstab_var_batch_t* x, double* io);

/* Diagnostic builds only (-DFB_STAMP): in-kernel per-phase cycle counters. */
int fbstab_hip_debug_stamps(unsigned long long* out32, int reset);

/* ---- several GPUs of one node, one process --------------------------------
 * The path shards embarrassingly: QP q of a batch depends on nothing but its own data
 * (FBstabMpc::Solve is a pure function of qp and the guess, fbstab/fbstab_mpc.h:181-195).
 * A shard group names the devices; handles[d] is a solver created on devices[d]; shard d
 * is the contiguous block of counts[d] QPs behind the shards 0 .. d-1, its arrays
 * (data[d], x[d], out[d]) resident on device d.  The shards run side by side without
 * any exchange, then the solutions (z, l, v, y) and SolverOut records of all shards go to
 * root_x / root_out on devices[root] in ONE RCCL operation (grouped ncclSend / ncclRecv
 * over xGMI; the root's own shard is a device copy).  The solution arrays are packed
 * (stride = length) or column slices of one record per QP, with the same layout on the
 * root.  Synchronous.  No counterpart in the reference, which is single-threaded. */
typedef struct fbstab_shard_group* fbstab_shard_group_t;
int fbstab_hip_shard_group_create(int ndev, const int* devices, fbstab_shard_group_t* group);
int fbstab_hip_shard_group_destroy(fbstab_shard_group_t group);
/* collectives issued so far and send/recv pairs inside them (tests, diagnostics) */
int fbstab_hip_shard_group_stats(fbstab_shard_group_t group, long long* gathers, long long* rccl_ops);
int fbstab_hip_mpc_solve_batch_sharded(fbstab_shard_group_t group, const fbstab_mpc_handle_t* handles,
                                       const int* counts, const fbstab_mpc_batch_t* data,
                                       const fbstab_var_batch_t* x, fbstab_solver_out_t* const* out,
                                       int root, const fbstab_var_batch_t* root_x,
                                       fbstab_solver_out_t* root_out);
/* BASELINE configs[4] sharded by trajectory: fbstab_hip_mpc_receding_sweep on every
 * device at once, then ONE collective that brings the applied inputs to the root:
 * shard d's [steps][counts[d]][nu] log at root_u_log + steps * nu * (counts[0] + ... +
 * counts[d-1]).  stats (host, may be NULL): per step, summed over the shards. */
int fbstab_hip_mpc_receding_sweep_sharded(fbstab_shard_group_t group, const fbstab_mpc_handle_t* handles,
                                          const int* counts, const fbstab_mpc_batch_t* data,
                                          const fbstab_var_batch_t* x, fbstab_solver_out_t* const* out,
                                          const fbstab_receding_plant_t* plants, int steps, int retire,
                                          double* const* u_log, int root, double* root_u_log,
                                          unsigned long long* stats);

/* ---- dense -------------------------------------------------------------- */
/* Environment read by fbstab_hip_dense_create (developer / comparison switches):
 *   FBSTAB_HIP_DENSE_ORDER=pivoted|auto|natural   initial value of fbstab_hip_dense_set_factorisation's `order`
 *   FBSTAB_HIP_DENSE_SPREAD_BITS, _ACT_BITS       the two thresholds of the AUTO order
 *   FBSTAB_HIP_DENSE_THREADS=256 the four-wavefront kernel (always pivoted) for every shape */
int fbstab_hip_dense_create(int nz, int nl, int nv, int max_batch, int device,
                            fbstab_dense_handle_t* handle);
int fbstab_hip_dense_destroy(fbstab_dense_handle_t handle);
/* Elimination order of the LDL' factorisation of the Newton system's KKT matrix
 * (DenseCholeskySolver::Initialize, dense_cholesky_solver.cc:70-79: Eigen::LDLT, symmetric
 * pivoting on the largest remaining |diagonal|).
 *   FBSTAB_HIP_DENSE_ORDER_PIVOTED (default)  Eigen's rule at every Newton step: the
 *       reference's order of rounding errors.  Exit flags, proximal and Newton counts equal
 *       the CPU restatement's on every QP of every test and of three fuzz families built to
 *       be degenerate (tools/fuzz_dense.py, 3017 QPs; profiles/r04_a_dense_order_choice.txt).
 *   FBSTAB_HIP_DENSE_ORDER_NATURAL  handles with nz + nl <= 64 (one wavefront per QP, the
 *       matrix in registers) eliminate in the natural order instead: K is quasi-definite,
 *       every order factors it, and a compile-time order is 1.6 x faster per launch on
 *       BASELINE configs[1].  The systems are the same, the rounding is not: a step's error
 *       in the directions where K is small grows with the spread of the pivots, and about
 *       one per cent of the QPs of the degenerate fuzz families then take a different
 *       number of proximal or Newton iterations (same solutions to the tolerance; none of
 *       the 4096 QPs of configs[1] differs).  A zero, denormal, infinite or NaN pivot still
 *       goes to the pivoted path, whose verdict is Eigen's.
 *   FBSTAB_HIP_DENSE_ORDER_AUTO  natural order until a QP shows itself ill-conditioned, then
 *       Eigen's rule for the rest of that QP's solve: at the first Newton step whose
 *       natural-order pivots span more than `spread_bits` binary orders of magnitude
 *       (max |d_k| / min |d_k| >= 2^spread_bits; that step is factored again), or at whose
 *       iterate at least nz - nl inequality rows carry a barrier weight above
 *       2^-act_bits / sigma (more active rows than free variables: a degenerate vertex).
 *       At the defaults (32, 20) configs[1] runs at the natural order's speed and the
 *       degenerate families differ on 0.4 % of their QPs; spread_bits = 30 costs configs[1]
 *       12 % and leaves 0.13 %.  No threshold that keeps the speed closes the gap - the
 *       sensitive steps have the pivot spread of ordinary ones - which is why the default
 *       is the reference's order and this one is the caller's choice.
 * spread_bits: 1..2046, or 0 to keep the current value.  Handles on the four-wavefront
 * kernels (nz + nl > 64) always pivot; the call is accepted and has no effect there. */
enum fbstab_hip_dense_order {
  FBSTAB_HIP_DENSE_ORDER_AUTO = 0,
  FBSTAB_HIP_DENSE_ORDER_PIVOTED = 1,
  FBSTAB_HIP_DENSE_ORDER_NATURAL = 2
};
int fbstab_hip_dense_set_factorisation(fbstab_dense_handle_t handle, int order, int spread_bits);
/* The settings in force and (pivoted_steps, may be NULL; waits for the handle's last launch)
 * the number of Newton steps of the most recent solve_batch call that AUTO handed to the
 * pivoted factorisation; -1 where that does not apply. */
int fbstab_hip_dense_get_factorisation(fbstab_dense_handle_t handle, int* order, int* spread_bits,
                                       long long* pivoted_steps);
int fbstab_hip_dense_set_options(fbstab_dense_handle_t handle, const fbstab_options_t* options);
int fbstab_hip_dense_get_options(fbstab_dense_handle_t handle, fbstab_options_t* options);
int fbstab_hip_dense_solve_batch(fbstab_dense_handle_t handle, int batch,
                                 const fbstab_dense_batch_t* data,
                                 const fbstab_var_batch_t* x, fbstab_solver_out_t* out,
                                 int flags, void* stream);
/* As fbstab_hip_mpc_solve_batch_final (FBstabDense::Solve at Display::FINAL). */
int fbstab_hip_dense_solve_batch_final(fbstab_dense_handle_t handle, int batch,
                                       const fbstab_dense_batch_t* data, const fbstab_var_batch_t* x,
                                       fbstab_solver_out_t* out, double* norms, int flags, void* stream);
int fbstab_hip_dense_solve_traced(fbstab_dense_handle_t handle, const fbstab_dense_batch_t* data,
                                  const fbstab_var_batch_t* x, fbstab_solver_out_t* out,
                                  fbstab_trace_record_t* trace, int capacity, int* count);
/* As fbstab_hip_mpc_solve_batch_sharded (FBstabDense::Solve, fbstab/fbstab_dense.h:136-149). */
int fbstab_hip_dense_solve_batch_sharded(fbstab_shard_group_t group, const fbstab_dense_handle_t* handles,
                                         const int* counts, const fbstab_dense_batch_t* data,
                                         const fbstab_var_batch_t* x, fbstab_solver_out_t* const* out,
                                         int root, const fbstab_var_batch_t* root_x,
                                         fbstab_solver_out_t* root_out);
/* As fbstab_hip_mpc_debug_newton, for the dense path (DenseCholeskySolver::Initialize +
 * Solve, dense_cholesky_solver.cc:32-127).  io: [zbar, lbar, vbar] in,
 * [dz, dl, dv, A*dz, W_z, W_l, r_z, r_l, ok] out. */
int fbstab_hip_dense_debug_newton(fbstab_dense_handle_t handle, const fbstab_dense_batch_t* data,
                                  const fbstab_var_batch_t* x, double* io);
double fbstab_hip_dense_last_kernel_ms(fbstab_dense_handle_t handle);
int fbstab_hip_dense_query(fbstab_dense_handle_t handle, long long* scratch_bytes,
                           int* lds_bytes, int* workgroups, int* threads);

#ifdef __cplusplus
}
#endif

#endif /* FBSTAB_HIP_H_ */

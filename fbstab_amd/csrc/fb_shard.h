// Multi-GPU entry points of the C-ABI (host code; included by fbstab_hip.hip behind the
// solver structs): ONE process drives the GPUs of a node, one solver handle per device.
// The path shards embarrassingly (BASELINE configs[3] and [4]): shard d is a
// contiguous block of QPs (trajectories) resident on device d, nothing is exchanged while
// the shards run, and the results go to one root device in ONE RCCL operation per batch -
// grouped ncclSend / ncclRecv over xGMI, every peer using its direct link to the root.
// The reference has no counterpart (a single-threaded CPU library); the calls keep the
// argument meaning of fbstab_hip_mpc_solve_batch / _receding_sweep per shard
// (FBstabMpc::Solve, fbstab/fbstab_mpc.h:181-195).
//
// RCCL is bound with dlopen at the first gather: the library carries no link-time
// dependency on it - nor a build-time one: the handful of types and prototypes used are
// declared below, so that a ROCm install without the RCCL headers still builds the library
// - and a process that has RCCL loaded already (PyTorch brings its own copy) keeps exactly
// one: the loaded copy is looked up first (RTLD_NOLOAD).
//
// State of verification: groups of ONE physical device have run on hardware (one shard;
// two shards on the same device through the test-only switch below, which exercises the
// offsets `first`, the per-shard pieces and the log offsets of the sweep; the RCCL leg as a
// self-send).  ncclCommInitAll over two or more devices and sends between devices have NOT
// run: no multi-GPU node was available to any round of this build.
#pragma once

#include <dlfcn.h>

#include <thread>

namespace {

// The part of <rccl/rccl.h> this file uses (RCCL 2.x ABI: opaque communicator, int-sized enums).
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;
typedef int ncclDataType_t;
constexpr ncclResult_t ncclSuccess = 0;
constexpr ncclDataType_t ncclChar = 0;  // ncclInt8

struct RcclApi {
  void* so = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  int version = 0;  // NCCL_VERSION_CODE of the bound library (major * 10000 + minor * 100 + patch since 2.9)
  int load() {
    if (so) return FBSTAB_HIP_OK;
    const char* const names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* name : names) {  // a copy the process has loaded already
      so = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
      if (so) break;
    }
    for (const char* name : names) {
      if (so) break;
      so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    }
    if (!so) return fail(FBSTAB_HIP_ERR_DEVICE, std::string("librccl.so not found: ") + dlerror());
    bool ok = true;
    auto sym = [&](const char* n) { void* p = dlsym(so, n); ok = ok && p; return p; };
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
    Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
    Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
    GetVersion = reinterpret_cast<decltype(GetVersion)>(sym("ncclGetVersion"));
    if (!ok) {
      so = nullptr;
      return fail(FBSTAB_HIP_ERR_DEVICE, "librccl.so lacks ncclSend / ncclRecv / ncclCommInitAll / ncclGetVersion");
    }
    // The prototypes above are written out by hand for the 2.x ABI (ncclSend / ncclRecv exist since 2.7):
    // a library that reports anything else is refused rather than called with a guessed signature.
    if (GetVersion(&version) != ncclSuccess || version < 2700 || version >= 30000) {
      so = nullptr;
      return fail(FBSTAB_HIP_ERR_DEVICE, "librccl.so reports version code " + std::to_string(version) +
                                             ": outside the 2.x ABI (>= 2.7) fb_shard.h declares");
    }
    return FBSTAB_HIP_OK;
  }
};

}  // namespace

struct fbstab_shard_group {
  std::vector<int> devices;
  bool repeated = false;  // (test-only) a physical device carries more than one shard
  std::vector<ncclComm_t> comms;  // created by the first gather that needs them
  RcclApi rccl;
  long long gathers = 0, rccl_ops = 0;  // (diagnostics: collectives issued, send/recv pairs in them)
};

namespace {

#define RCCL_TRY(g, expr)                                                                         \
  do {                                                                                            \
    ncclResult_t r_ = (expr);                                                                     \
    if (r_ != ncclSuccess)                                                                        \
      return fail(FBSTAB_HIP_ERR_DEVICE, std::string("RCCL: ") + (g)->rccl.GetErrorString(r_));   \
  } while (0)

struct GatherPiece {
  const void* src;  // on the shard's device
  void* dst;        // on the root device
  size_t bytes;
};

// The pieces of the solution arrays (z, l, v, y) of one shard: `count` QPs from x (on the
// shard's device) to QPs [first, first + count) of root_x.  Packed arrays travel one by
// one; arrays that are column slices of ONE record per QP (a common layout: the solution
// and whatever the caller keeps behind it side by side) travel as whole records.
int solution_pieces(const long long var_len[4], const fbstab_var_batch_t& x, const fbstab_var_batch_t& rx,
                    long long first, int count, std::vector<GatherPiece>* out) {
  bool packed = true, record = true;
  for (int i = 0; i < 4; i++) {
    if (var_len[i] == 0) continue;
    packed = packed && x.stride[i] == var_len[i] && rx.stride[i] == var_len[i];
    record = record && x.stride[i] == x.stride[0] && rx.stride[i] == x.stride[0];
  }
  for (int i = 0; i + 1 < 4 && record; i++)
    record = x.base[i + 1] == x.base[i] + var_len[i] && rx.base[i + 1] == rx.base[i] + var_len[i];
  if (packed) {
    for (int i = 0; i < 4; i++)
      if (var_len[i] > 0)
        out->push_back({x.base[i], rx.base[i] + first * var_len[i], sizeof(double) * (size_t)var_len[i] * count});
    return FBSTAB_HIP_OK;
  }
  if (record) {
    // From z of the shard's first QP to the end of y of its last: whatever the caller keeps
    // between y of one record and z of the next travels along and lands in the same columns
    // of the root's records (the caller's own layout on both sides); nothing before the
    // first z or behind the last y is touched, wherever z sits inside the record.
    const long long span = var_len[0] + var_len[1] + var_len[2] + var_len[3];
    if (count > 0)
      out->push_back({x.base[0], rx.base[0] + first * rx.stride[0],
                      sizeof(double) * (size_t)(x.stride[0] * (count - 1) + span)});
    return FBSTAB_HIP_OK;
  }
  return fail(FBSTAB_HIP_ERR_UNSUPPORTED,
              "sharded gather: the solution arrays must be packed (stride = length) or slices of one record per QP, "
              "with the same layout on the root");
}

// ONE collective: every piece of every shard to the root, fused between ncclGroupStart and
// ncclGroupEnd; sends ride on the shard's stream (behind its solve), receives on the
// root's.  A shard that lives on the root's own physical device is a device-to-device copy
// on the shard's stream (FBSTAB_HIP_SHARD_SELF_SEND=1 sends the root's shard through RCCL
// as well: the one-GPU rehearsal of the path; ignored for groups with a repeated device,
// over which RCCL cannot build communicators).  An error inside the group still closes it.
int shard_gather(fbstab_shard_group* g, int root, const std::vector<std::vector<GatherPiece>>& pieces,
                 const std::vector<hipStream_t>& streams) {
  const int ndev = (int)g->devices.size();
  const char* self_env = getenv("FBSTAB_HIP_SHARD_SELF_SEND");
  const bool self_send = self_env && atoi(self_env) != 0 && !g->repeated;
  auto local = [&](int d) { return g->devices[d] == g->devices[root] && !self_send; };
  bool need_rccl = false;
  for (int d = 0; d < ndev; d++) need_rccl = need_rccl || !local(d);
  if (need_rccl && g->repeated)
    return fail(FBSTAB_HIP_ERR_UNSUPPORTED, "a group with a repeated device (test-only) must live on ONE device");
  if (need_rccl && g->comms.empty()) {
    int rc = g->rccl.load();
    if (rc != FBSTAB_HIP_OK) return rc;
    g->comms.assign(ndev, nullptr);
    ncclResult_t r = g->rccl.CommInitAll(g->comms.data(), ndev, g->devices.data());
    if (r != ncclSuccess) {
      g->comms.clear();
      return fail(FBSTAB_HIP_ERR_DEVICE, std::string("RCCL: ncclCommInitAll: ") + g->rccl.GetErrorString(r));
    }
  }
  g->gathers++;
  if (need_rccl) RCCL_TRY(g, g->rccl.GroupStart());
  auto queue = [&]() -> int {
    for (int d = 0; d < ndev; d++) {
      for (const GatherPiece& p : pieces[d]) {
        if (p.bytes == 0) continue;
        if (local(d)) {
          HIP_TRY(hipSetDevice(g->devices[d]));
          HIP_TRY(hipMemcpyAsync(p.dst, p.src, p.bytes, hipMemcpyDeviceToDevice, streams[d]));
        } else {
          RCCL_TRY(g, g->rccl.Send(p.src, p.bytes, ncclChar, root, g->comms[d], streams[d]));
          RCCL_TRY(g, g->rccl.Recv(p.dst, p.bytes, ncclChar, d, g->comms[root], streams[root]));
          g->rccl_ops++;
        }
      }
    }
    return FBSTAB_HIP_OK;
  };
  const int rc = queue();
  const std::string msg = rc != FBSTAB_HIP_OK ? std::string(fbstab_hip_last_error()) : std::string();
  if (need_rccl) {
    const ncclResult_t r = g->rccl.GroupEnd();  // (always: an open group would swallow the thread's next RCCL calls)
    if (rc == FBSTAB_HIP_OK && r != ncclSuccess)
      return fail(FBSTAB_HIP_ERR_DEVICE, std::string("RCCL: ncclGroupEnd: ") + g->rccl.GetErrorString(r));
  }
  return rc != FBSTAB_HIP_OK ? fail(rc, msg) : FBSTAB_HIP_OK;
}

int sync_all(fbstab_shard_group* g, const std::vector<hipStream_t>& streams) {
  for (size_t d = 0; d < g->devices.size(); d++) {
    HIP_TRY(hipSetDevice(g->devices[d]));
    HIP_TRY(hipStreamSynchronize(streams[d]));
  }
  return FBSTAB_HIP_OK;
}
// A failure after the first shard was queued: nothing may be left in flight when the call
// returns (the caller is free to release its buffers); the first error is what is reported.
int drain_and_fail(fbstab_shard_group* g, const std::vector<hipStream_t>& streams, int rc) {
  const std::string msg = fbstab_hip_last_error();
  (void)sync_all(g, streams);
  return fail(rc, msg);
}

template <class Handle>
int check_shards(fbstab_shard_group* g, Handle* const* handles, const int* counts, int root, long long* total) {
  if (!g || !handles || !counts) return fail(FBSTAB_HIP_ERR_ARGUMENT, "null argument");
  const int ndev = (int)g->devices.size();
  if (root < 0 || root >= ndev) return fail(FBSTAB_HIP_ERR_ARGUMENT, "root is an index into the group's devices");
  *total = 0;
  for (int d = 0; d < ndev; d++) {
    if (!handles[d]) return fail(FBSTAB_HIP_ERR_ARGUMENT, "null solver handle");
    if (handles[d]->device != g->devices[d])
      return fail(FBSTAB_HIP_ERR_ARGUMENT, "handles[d] must live on the group's device d");
    if (counts[d] < 0 || counts[d] > handles[d]->max_batch)
      return fail(FBSTAB_HIP_ERR_ARGUMENT, "shard exceeds the max_batch its handle was created with");
    for (int e = 0; e < d; e++)
      if (handles[e] == handles[d]) return fail(FBSTAB_HIP_ERR_ARGUMENT, "a solver handle serves one shard");
    for (int i = 0; i < 4; i++)
      if (handles[d]->var_len[i] != handles[0]->var_len[i])
        return fail(FBSTAB_HIP_ERR_ARGUMENT, "the shards' handles must have one problem size");
    *total += counts[d];
  }
  return FBSTAB_HIP_OK;
}

}  // namespace

extern "C" {

int fbstab_hip_shard_group_create(int ndev, const int* devices, fbstab_shard_group_t* group) {
  if (!group) return fail(FBSTAB_HIP_ERR_ARGUMENT, "null group pointer");
  *group = nullptr;
  if (ndev < 1 || !devices) return fail(FBSTAB_HIP_ERR_ARGUMENT, "at least one device");
  int have = 0;
  if (hipGetDeviceCount(&have) != hipSuccess || have <= 0)
    return fail(FBSTAB_HIP_ERR_DEVICE, "no HIP device available (this library has no CPU path)");
  // FBSTAB_HIP_SHARD_ALLOW_REPEATED_DEVICE=1 (tests): several shards on ONE physical device,
  // each with its own handle and stream, gathered by device copies - the index arithmetic of
  // the sharded entries (offsets of the shards on the root, per-shard pieces, log offsets)
  // on a box with a single GPU.
  const char* rep_env = getenv("FBSTAB_HIP_SHARD_ALLOW_REPEATED_DEVICE");
  const bool allow_rep = rep_env && atoi(rep_env) != 0;
  bool repeated = false;
  for (int d = 0; d < ndev; d++) {
    if (devices[d] < 0 || devices[d] >= have) return fail(FBSTAB_HIP_ERR_ARGUMENT, "bad device index");
    for (int e = 0; e < d; e++)
      if (devices[e] == devices[d]) {
        if (!allow_rep) return fail(FBSTAB_HIP_ERR_ARGUMENT, "a device appears twice in the group");
        repeated = true;
      }
  }
  fbstab_shard_group* g = new (std::nothrow) fbstab_shard_group();
  if (!g) return fail(FBSTAB_HIP_ERR_DEVICE, "out of host memory");
  g->devices.assign(devices, devices + ndev);
  g->repeated = repeated;
  *group = g;
  return FBSTAB_HIP_OK;
}

int fbstab_hip_shard_group_destroy(fbstab_shard_group_t g) {
  if (!g) return FBSTAB_HIP_OK;
  for (ncclComm_t c : g->comms)
    if (c) (void)g->rccl.CommDestroy(c);
  delete g;
  return FBSTAB_HIP_OK;
}

int fbstab_hip_shard_group_stats(fbstab_shard_group_t g, long long* gathers, long long* rccl_ops) {
  if (!g) return fail(FBSTAB_HIP_ERR_ARGUMENT, "null group");
  if (gathers) *gathers = g->gathers;
  if (rccl_ops) *rccl_ops = g->rccl_ops;
  return FBSTAB_HIP_OK;
}

int fbstab_hip_mpc_solve_batch_sharded(fbstab_shard_group_t g, const fbstab_mpc_handle_t* handles, const int* counts,
                                       const fbstab_mpc_batch_t* data, const fbstab_var_batch_t* x,
                                       fbstab_solver_out_t* const* out, int root, const fbstab_var_batch_t* root_x,
                                       fbstab_solver_out_t* root_out) {
  long long total = 0;
  int rc = check_shards(g, handles, counts, root, &total);
  if (rc != FBSTAB_HIP_OK) return rc;
  if (!data || !x || !out || !root_x || !root_out) return fail(FBSTAB_HIP_ERR_ARGUMENT, "null argument");
  const int ndev = (int)g->devices.size();
  std::vector<hipStream_t> streams(ndev);
  std::vector<std::vector<GatherPiece>> pieces(ndev);
  long long first = 0;
  // every shard's arguments are checked before the first shard is queued (an empty shard
  // needs no arrays and is skipped)
  for (int d = 0; d < ndev; d++) {
    streams[d] = handles[d]->stream;
    if (counts[d] == 0) continue;
    if (!out[d]) return fail(FBSTAB_HIP_ERR_ARGUMENT, "null SolverOut pointer of a non-empty shard");
    for (int i = 0; i < FBSTAB_MPC_NSEQ; i++)
      if (!data[d].base[i] && handles[d]->arr_len[i] > 0)
        return fail(FBSTAB_HIP_ERR_ARGUMENT, "null problem data pointer of a non-empty shard");
    for (int i = 0; i < 4; i++)
      if (!x[d].base[i] && handles[d]->var_len[i] > 0)
        return fail(FBSTAB_HIP_ERR_ARGUMENT, "null variable pointer of a non-empty shard");
  }
  for (int i = 0; i < 4; i++)
    if (!root_x->base[i] && handles[0]->var_len[i] > 0 && total > 0)
      return fail(FBSTAB_HIP_ERR_ARGUMENT, "null variable pointer on the root");
  for (int d = 0; d < ndev; d++) {
    if (counts[d] > 0) {
      rc = solution_pieces(handles[d]->var_len, x[d], *root_x, first, counts[d], &pieces[d]);
      if (rc != FBSTAB_HIP_OK) return rc;
      pieces[d].push_back({out[d], root_out + first, sizeof(fbstab_solver_out_t) * (size_t)counts[d]});
    }
    first += counts[d];
  }
  // every shard is queued on its device (nothing waits for another device), then the gather
  for (int d = 0; d < ndev; d++) {
    if (counts[d] == 0) continue;
    rc = fbstab_hip_mpc_solve_batch(handles[d], counts[d], &data[d], &x[d], out[d],
                                    FBSTAB_HIP_DEVICE_POINTERS | FBSTAB_HIP_ASYNC, nullptr);
    if (rc != FBSTAB_HIP_OK) return drain_and_fail(g, streams, rc);
  }
  rc = shard_gather(g, root, pieces, streams);
  if (rc != FBSTAB_HIP_OK) return drain_and_fail(g, streams, rc);
  return sync_all(g, streams);
}

int fbstab_hip_dense_solve_batch_sharded(fbstab_shard_group_t g, const fbstab_dense_handle_t* handles,
                                         const int* counts, const fbstab_dense_batch_t* data,
                                         const fbstab_var_batch_t* x, fbstab_solver_out_t* const* out, int root,
                                         const fbstab_var_batch_t* root_x, fbstab_solver_out_t* root_out) {
  long long total = 0;
  int rc = check_shards(g, handles, counts, root, &total);
  if (rc != FBSTAB_HIP_OK) return rc;
  if (!data || !x || !out || !root_x || !root_out) return fail(FBSTAB_HIP_ERR_ARGUMENT, "null argument");
  const int ndev = (int)g->devices.size();
  std::vector<hipStream_t> streams(ndev);
  std::vector<std::vector<GatherPiece>> pieces(ndev);
  long long first = 0;
  // every shard's arguments are checked before the first shard is queued (an empty shard
  // needs no arrays and is skipped)
  for (int d = 0; d < ndev; d++) {
    streams[d] = handles[d]->stream;
    if (counts[d] == 0) continue;
    if (!out[d]) return fail(FBSTAB_HIP_ERR_ARGUMENT, "null SolverOut pointer of a non-empty shard");
    for (int i = 0; i < FBSTAB_DENSE_NARR; i++)
      if (!data[d].base[i] && handles[d]->arr_len[i] > 0)
        return fail(FBSTAB_HIP_ERR_ARGUMENT, "null problem data pointer of a non-empty shard");
    for (int i = 0; i < 4; i++)
      if (!x[d].base[i] && handles[d]->var_len[i] > 0)
        return fail(FBSTAB_HIP_ERR_ARGUMENT, "null variable pointer of a non-empty shard");
  }
  for (int i = 0; i < 4; i++)
    if (!root_x->base[i] && handles[0]->var_len[i] > 0 && total > 0)
      return fail(FBSTAB_HIP_ERR_ARGUMENT, "null variable pointer on the root");
  for (int d = 0; d < ndev; d++) {
    if (counts[d] > 0) {
      rc = solution_pieces(handles[d]->var_len, x[d], *root_x, first, counts[d], &pieces[d]);
      if (rc != FBSTAB_HIP_OK) return rc;
      pieces[d].push_back({out[d], root_out + first, sizeof(fbstab_solver_out_t) * (size_t)counts[d]});
    }
    first += counts[d];
  }
  for (int d = 0; d < ndev; d++) {
    if (counts[d] == 0) continue;
    rc = fbstab_hip_dense_solve_batch(handles[d], counts[d], &data[d], &x[d], out[d],
                                      FBSTAB_HIP_DEVICE_POINTERS | FBSTAB_HIP_ASYNC, nullptr);
    if (rc != FBSTAB_HIP_OK) return drain_and_fail(g, streams, rc);
  }
  rc = shard_gather(g, root, pieces, streams);
  if (rc != FBSTAB_HIP_OK) return drain_and_fail(g, streams, rc);
  return sync_all(g, streams);
}

// BASELINE configs[4] over the GPUs of a node: every device sweeps its own trajectories
// in one launch (fbstab_hip_mpc_receding_sweep; a host thread per device, the sweeps run
// side by side and never exchange anything), then the applied inputs go to the root in
// one collective: root_u_log holds the shards' logs one after the other, shard d's
// [steps][counts[d]][nu] block at root_u_log + steps * nu * (counts[0] + ... + counts[d-1]).
int fbstab_hip_mpc_receding_sweep_sharded(fbstab_shard_group_t g, const fbstab_mpc_handle_t* handles,
                                          const int* counts, const fbstab_mpc_batch_t* data,
                                          const fbstab_var_batch_t* x, fbstab_solver_out_t* const* out,
                                          const fbstab_receding_plant_t* plants, int steps, int retire,
                                          double* const* u_log, int root, double* root_u_log,
                                          unsigned long long* stats) {
  long long total = 0;
  int rc = check_shards(g, handles, counts, root, &total);
  if (rc != FBSTAB_HIP_OK) return rc;
  if (!data || !x || !out || !plants || !u_log || !root_u_log || steps < 0)
    return fail(FBSTAB_HIP_ERR_ARGUMENT, "null argument");
  const int ndev = (int)g->devices.size();
  const int nu = handles[0]->lay.nu;
  std::vector<int> rcs(ndev, FBSTAB_HIP_OK);
  std::vector<std::string> errs(ndev);
  std::vector<std::vector<unsigned long long>> st(ndev, std::vector<unsigned long long>(4 * (size_t)steps, 0ull));
  {
    std::vector<std::thread> th;
    for (int d = 0; d < ndev; d++)
      th.emplace_back([&, d]() {
        if (counts[d] == 0) return;  // (an empty shard: nothing to sweep, nothing to gather)
        rcs[d] = fbstab_hip_mpc_receding_sweep(handles[d], counts[d], &data[d], &x[d], out[d], &plants[d], steps,
                                               retire, u_log[d], st[d].data(), nullptr, nullptr);
        if (rcs[d] != FBSTAB_HIP_OK) errs[d] = fbstab_hip_last_error();  // (the message is per thread)
      });
    for (std::thread& t : th) t.join();
  }
  for (int d = 0; d < ndev; d++)
    if (rcs[d] != FBSTAB_HIP_OK) return fail(rcs[d], errs[d]);
  if (stats)  // per step: sums over the shards, the largest Newton count their maximum
    for (int k = 0; k < steps; k++) {
      unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0;
      for (int d = 0; d < ndev; d++) {
        s0 += st[d][4 * k]; s1 += st[d][4 * k + 1]; s2 += st[d][4 * k + 2];
        s3 = st[d][4 * k + 3] > s3 ? st[d][4 * k + 3] : s3;
      }
      stats[4 * k] = s0; stats[4 * k + 1] = s1; stats[4 * k + 2] = s2; stats[4 * k + 3] = s3;
    }
  std::vector<hipStream_t> streams(ndev);
  std::vector<std::vector<GatherPiece>> pieces(ndev);
  long long first = 0;
  for (int d = 0; d < ndev; d++) {
    streams[d] = handles[d]->stream;
    const size_t n = (size_t)steps * counts[d] * nu;
    pieces[d].push_back({u_log[d], root_u_log + (size_t)steps * nu * first, sizeof(double) * n});
    first += counts[d];
  }
  rc = shard_gather(g, root, pieces, streams);
  if (rc != FBSTAB_HIP_OK) return drain_and_fail(g, streams, rc);
  return sync_all(g, streams);
}

}  // extern "C"

"""ctypes binding of libfbstab_hip.so (the C-ABI in include/fbstab_hip.h).

This is plumbing for the Python tests and ``bench.py``: it passes raw pointers
(numpy host arrays or torch CUDA tensors) to the C entry points and adds no
computation of its own.  There is no fallback: if the shared library is missing
or no HIP device is usable, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FBSTAB_HIP_LIB", os.path.join(_HERE, "libfbstab_hip.so"))

MPC_SEQ = ("Q", "R", "S", "q", "r", "A", "B", "c", "E", "L", "d", "x0")
DENSE_ARR = ("H", "f", "G", "h", "A", "b")

HOST_POINTERS = 0
DEVICE_POINTERS = 1
ASYNC = 2
KEEP_MATRICES = 4

EXIT_FLAGS = {0: "SUCCESS", 1: "DIVERGENCE", 2: "MAXITERATIONS", 3: "PRIMAL_INFEASIBLE",
              4: "DUAL_INFEASIBLE", 5: "PRIMAL_DUAL_INFEASIBLE", 6: "SATURATE_ERROR"}


class Options(C.Structure):
    """fbstab_options_t (include/fbstab_types.h) == AlgorithmParameters
    (fbstab/fbstab_algorithm.h:48-82)."""
    _fields_ = [(n, C.c_double) for n in (
        "sigma0", "sigma_max", "sigma_min", "alpha", "beta", "eta", "delta",
        "gamma", "abs_tol", "rel_tol", "stall_tol", "infeas_tol",
        "inner_tol_max", "inner_tol_min")] + [(n, C.c_int) for n in (
            "max_newton_iters", "max_prox_iters", "max_inner_iters",
            "max_linesearch_iters", "check_feasibility",
            "nonmonotone_linesearch", "display_level", "reserved")]


def DefaultOptions(**kw) -> Options:
    """FBstabMpc::DefaultOptions / FBstabDense::DefaultOptions
    (fbstab_algorithm-impl.h:33-59); display is FINAL there, the batch path
    never prints so the field is carried but unused."""
    o = Options(1e-8, 1e-6, 1e-12, 0.95, 0.75, 1e-8, 0.2, 0.1, 1e-6, 1e-12, 1e-10,
                1e-8, 1e-2, 1e-12, 200, 30, 50, 20, 1, 1, 1, 0)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def ReliableOptions(**kw) -> Options:
    """ReliableOptions (fbstab_algorithm-impl.h:61-74)."""
    o = DefaultOptions(sigma0=1e-4, sigma_max=1e-2, sigma_min=1e-10, beta=0.9,
                       abs_tol=1e-4, rel_tol=1e-6, max_linesearch_iters=40,
                       max_newton_iters=500, max_prox_iters=100,
                       nonmonotone_linesearch=0)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


OUT_DTYPE = np.dtype([("eflag", np.int32), ("pad_", np.int32),
                      ("residual", np.float64), ("newton_iters", np.int32),
                      ("prox_iters", np.int32), ("solve_time", np.float64),
                      ("initial_residual", np.float64)])
assert OUT_DTYPE.itemsize == 40


class _MpcBatch(C.Structure):
    _fields_ = [("base", C.c_void_p * 12), ("stride", C.c_longlong * 12)]


class _DenseBatch(C.Structure):
    _fields_ = [("base", C.c_void_p * 6), ("stride", C.c_longlong * 6)]


class _VarBatch(C.Structure):
    _fields_ = [("base", C.c_void_p * 4), ("stride", C.c_longlong * 4)]


class _Plant(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("stride_A", C.c_longlong), ("stride_B", C.c_longlong)]


_libs: Dict[str, C.CDLL] = {}
_current = LIB_PATH  # the library new solver objects bind to (see `library`)


def current_library_path() -> str:
    """Path of the library new solver objects bind to (what bench.py hashes as "this build")."""
    return _current


class library:
    """``with hip_api.library(path):`` - solver objects created inside bind to ANOTHER build of the
    same sources at ``path`` (the tests compare a pattern-initialised build with the product
    library, tests/helpers.py: VARIANT_LIBS); both libraries can be in use in one process."""

    def __init__(self, path: str):
        self.path = path

    def __enter__(self):
        global _current
        self.prev, _current = _current, self.path
        return load_library()

    def __exit__(self, *exc):
        global _current
        _current = self.prev
        return False


def load_library() -> C.CDLL:
    """Load libfbstab_hip.so; raises (never falls back) when it is absent."""
    LIB_PATH = _current
    if LIB_PATH in _libs:
        return _libs[LIB_PATH]
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C fbstab_amd/csrc` "
            "(__graft_entry__.build()); fbstab_amd has no CPU fallback")
    # torch (device memory / streams / torch.distributed plumbing) bundles its
    # own HIP runtime: import it first so this process ends up with ONE
    # libamdhip64 (loading ours first makes torch see "No HIP GPUs").
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    lib.fbstab_hip_last_error.restype = C.c_char_p
    for kind in ("mpc", "dense"):
        getattr(lib, f"fbstab_hip_{kind}_last_kernel_ms").restype = C.c_double
        getattr(lib, f"fbstab_hip_{kind}_last_kernel_ms").argtypes = [C.c_void_p]
        getattr(lib, f"fbstab_hip_{kind}_destroy").argtypes = [C.c_void_p]
        getattr(lib, f"fbstab_hip_{kind}_set_options").argtypes = [C.c_void_p, C.c_void_p]
        getattr(lib, f"fbstab_hip_{kind}_get_options").argtypes = [C.c_void_p, C.c_void_p]
        getattr(lib, f"fbstab_hip_{kind}_solve_batch").argtypes = [
            C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        getattr(lib, f"fbstab_hip_{kind}_solve_batch_final").argtypes = [
            C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        getattr(lib, f"fbstab_hip_{kind}_query").argtypes = [
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        getattr(lib, f"fbstab_hip_{kind}_solve_traced").argtypes = [
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.fbstab_hip_mpc_kernel_name.restype = C.c_char_p
    lib.fbstab_hip_mpc_kernel_name.argtypes = [C.c_void_p]
    if hasattr(lib, "fbstab_hip_mpc_refined_steps"):  # (absent from a round-4 build loaded for an A/B: FBSTAB_HIP_LIB)
        lib.fbstab_hip_mpc_refined_steps.argtypes = [C.c_void_p, C.c_void_p]
    lib.fbstab_hip_mpc_receding_sweep.argtypes = [
        C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.fbstab_hip_shard_group_create.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    lib.fbstab_hip_shard_group_destroy.argtypes = [C.c_void_p]
    lib.fbstab_hip_shard_group_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    for name in ("fbstab_hip_mpc_solve_batch_sharded", "fbstab_hip_dense_solve_batch_sharded"):
        getattr(lib, name).argtypes = [C.c_void_p] * 6 + [C.c_int, C.c_void_p, C.c_void_p]
    lib.fbstab_hip_mpc_receding_sweep_sharded.argtypes = [C.c_void_p] * 7 + [C.c_int, C.c_int, C.c_void_p, C.c_int,
                                                          C.c_void_p, C.c_void_p]
    lib.fbstab_hip_mpc_create.argtypes = [C.c_int] * 6 + [C.c_void_p]
    if hasattr(lib, "fbstab_hip_mpc_create_in_flight"):  # (absent from a build of an earlier round loaded for an A/B)
        lib.fbstab_hip_mpc_create_in_flight.argtypes = [C.c_int] * 7 + [C.c_void_p]
    lib.fbstab_hip_dense_create.argtypes = [C.c_int] * 5 + [C.c_void_p]
    lib.fbstab_hip_dense_set_factorisation.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.fbstab_hip_dense_get_factorisation.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    _libs[LIB_PATH] = lib
    return lib


EXPORTED_SYMBOLS = (
    "fbstab_hip_last_error", "fbstab_hip_device_count",
    "fbstab_hip_mpc_create", "fbstab_hip_mpc_destroy", "fbstab_hip_mpc_set_options",
    "fbstab_hip_mpc_get_options", "fbstab_hip_mpc_solve_batch", "fbstab_hip_mpc_solve_batch_final",
    "fbstab_hip_mpc_solve_traced", "fbstab_hip_mpc_receding_sweep",
    "fbstab_hip_mpc_last_kernel_ms", "fbstab_hip_mpc_query", "fbstab_hip_mpc_kernel_name",
    "fbstab_hip_mpc_refined_steps", "fbstab_hip_mpc_create_in_flight",
    "fbstab_hip_mpc_debug_newton", "fbstab_hip_debug_stamps",
    "fbstab_hip_dense_create", "fbstab_hip_dense_destroy", "fbstab_hip_dense_set_options",
    "fbstab_hip_dense_get_options", "fbstab_hip_dense_solve_batch", "fbstab_hip_dense_solve_batch_final",
    "fbstab_hip_dense_solve_traced",
    "fbstab_hip_dense_debug_newton", "fbstab_hip_dense_last_kernel_ms", "fbstab_hip_dense_query",
    "fbstab_hip_dense_set_factorisation", "fbstab_hip_dense_get_factorisation",
    "fbstab_hip_shard_group_create", "fbstab_hip_shard_group_destroy", "fbstab_hip_shard_group_stats",
    "fbstab_hip_mpc_solve_batch_sharded", "fbstab_hip_mpc_receding_sweep_sharded",
    "fbstab_hip_dense_solve_batch_sharded")


class FBstabHipError(RuntimeError):
    """Mirrors the std::runtime_error the reference throws (fbstab_mpc.cc:62-65,
    fbstab_mpc.h:229-242, ...)."""


def _check(lib, rc):
    if rc != 0:
        raise FBstabHipError(f"[{rc}] {lib.fbstab_hip_last_error().decode()}")


def _is_torch(a) -> bool:
    return type(a).__module__.startswith("torch")


def _ptr_stride(a, length: int):
    """(pointer, batch stride in doubles, is_device) of a (batch, length) array."""
    if _is_torch(a):
        import torch
        assert a.dtype == torch.float64 and a.dim() == 2 and a.shape[1] == length, \
            (a.shape, length)
        assert a.stride(1) == 1 or length <= 1
        return a.data_ptr(), (a.stride(0) if a.shape[0] > 1 else length), a.is_cuda
    assert isinstance(a, np.ndarray) and a.dtype == np.float64 and a.ndim == 2 \
        and a.shape[1] == length, (getattr(a, "shape", None), length)
    assert a.strides[1] == 8 or length <= 1
    return a.ctypes.data, (a.strides[0] // 8 if a.shape[0] > 1 else length), False


class _SolverBase:
    _kind = ""

    def __init__(self):
        self._h = C.c_void_p()
        self._lib = load_library()

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            getattr(self._lib, f"fbstab_hip_{self._kind}_destroy")(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def UpdateOptions(self, options: Options):
        _check(self._lib, getattr(self._lib, f"fbstab_hip_{self._kind}_set_options")(
            self._h, C.byref(options)))

    def CurrentOptions(self) -> Options:
        o = Options()
        _check(self._lib, getattr(self._lib, f"fbstab_hip_{self._kind}_get_options")(
            self._h, C.byref(o)))
        return o

    def last_kernel_ms(self) -> float:
        return float(getattr(self._lib, f"fbstab_hip_{self._kind}_last_kernel_ms")(self._h))

    def query(self) -> Dict[str, int]:
        sb, lds, wg, th = C.c_longlong(), C.c_int(), C.c_int(), C.c_int()
        _check(self._lib, getattr(self._lib, f"fbstab_hip_{self._kind}_query")(
            self._h, C.byref(sb), C.byref(lds), C.byref(wg), C.byref(th)))
        return dict(scratch_bytes=sb.value, lds_bytes=lds.value, workgroups=wg.value,
                    threads=th.value)

    def _solve(self, batch_struct, names: Sequence[str], lens: Sequence[int], arrays,
               var_lens, z, l, v, y, out, stream, async_, keep_matrices=False, norms=False):
        dev_flags = []
        B = None
        for i, (k, n) in enumerate(zip(names, lens)):
            a = arrays[k]
            if n == 0:
                batch_struct.base[i] = None
                batch_struct.stride[i] = 0
                continue
            p, s, d = _ptr_stride(a, n)
            batch_struct.base[i] = p
            batch_struct.stride[i] = s
            dev_flags.append(d)
            B = a.shape[0] if B is None else B
        vb = _VarBatch()
        for i, (a, n) in enumerate(zip((z, l, v, y), var_lens)):
            if n == 0:
                vb.base[i] = None
                vb.stride[i] = 0
                continue
            p, s, d = _ptr_stride(a, n)
            assert a.shape[0] == B
            vb.base[i] = p
            vb.stride[i] = s
            dev_flags.append(d)
        on_dev = all(dev_flags)
        assert on_dev or not any(dev_flags), "mix of host and device arrays"
        if on_dev:
            import torch
            if out is None:
                out = torch.zeros((B, 40), dtype=torch.uint8, device=z.device)
            out_ptr = out.data_ptr()
            if not stream:
                # the arrays were produced on torch's current stream: run there (0 is
                # the null stream, which the handle's own blocking stream is ordered with)
                stream = torch.cuda.current_stream(z.device).cuda_stream
            flags = DEVICE_POINTERS | (ASYNC if async_ else 0) | (KEEP_MATRICES if keep_matrices else 0)
        else:
            if out is None:
                out = np.zeros(B, dtype=OUT_DTYPE)
            out_ptr = out.ctypes.data
            flags = HOST_POINTERS
        if norms:
            # fbstab_hip_*_solve_batch_final: (B, 4) |rz| |rl| |rv| tolerance, living where out lives
            if on_dev:
                nrm = torch.zeros((B, 4), dtype=torch.float64, device=z.device)
                nptr = nrm.data_ptr()
            else:
                nrm = np.zeros((B, 4))
                nptr = nrm.ctypes.data
            rc = getattr(self._lib, f"fbstab_hip_{self._kind}_solve_batch_final")(
                self._h, B, C.byref(batch_struct), C.byref(vb), out_ptr, C.c_void_p(nptr), flags,
                C.c_void_p(stream) if stream else None)
            _check(self._lib, rc)
            return out, nrm
        rc = getattr(self._lib, f"fbstab_hip_{self._kind}_solve_batch")(
            self._h, B, C.byref(batch_struct), C.byref(vb), out_ptr, flags,
            C.c_void_p(stream) if stream else None)
        _check(self._lib, rc)
        return out


def _solve_traced(self, batch_struct, names, lens, arrays, var_lens, z, l, v, y, capacity):
    """fbstab_hip_*_solve_traced for ONE QP in (1, n) numpy arrays.  Returns
    ``(out, records)``; records is ``(n, 8)``: kind, i0, i1, v0..v4
    (fbstab_trace_record_t, include/fbstab_types.h)."""
    for i, (k, n) in enumerate(zip(names, lens)):
        if n == 0:
            batch_struct.base[i], batch_struct.stride[i] = None, 0
            continue
        p, st, d = _ptr_stride(arrays[k], n)
        assert not d and arrays[k].shape[0] == 1
        batch_struct.base[i], batch_struct.stride[i] = p, st
    vb = _VarBatch()
    for i, (a, n) in enumerate(zip((z, l, v, y), var_lens)):
        if n == 0:
            vb.base[i], vb.stride[i] = None, 0
            continue
        p, st, d = _ptr_stride(a, n)
        assert not d and a.shape[0] == 1
        vb.base[i], vb.stride[i] = p, st
    out = np.zeros(1, dtype=OUT_DTYPE)
    rec = np.zeros((capacity, 8))
    count = C.c_int(0)
    rc = getattr(self._lib, f"fbstab_hip_{self._kind}_solve_traced")(
        self._h, C.byref(batch_struct), C.byref(vb), out.ctypes.data, rec.ctypes.data, capacity,
        C.byref(count))
    _check(self._lib, rc)
    return out, rec[:min(count.value, capacity)].copy()


def out_to_numpy(out) -> np.ndarray:
    """SolverOut records (numpy structured array) from a solve's ``out``."""
    if _is_torch(out):
        return np.frombuffer(out.cpu().numpy().tobytes(), dtype=OUT_DTYPE).copy()
    return out


class FBstabMpcBatch(_SolverBase):
    """Batched counterpart of ``fbstab::FBstabMpc`` (fbstab/fbstab_mpc.h:56-243):
    ``FBstabMpcBatch(N, nx, nu, nc, max_batch)`` allocates the device workspace
    (the reference constructor allocates the CPU workspace, fbstab_mpc.cc:61-89)
    and ``Solve`` solves a batch in the reference data layout."""
    _kind = "mpc"

    def __init__(self, N: int, nx: int, nu: int, nc: int, max_batch: int = 1,
                 device: int = 0, handles_in_flight: int = 1):
        """handles_in_flight: how many such solvers the caller keeps busy on the device at the same time
        (fbstab_hip_mpc_create_in_flight: each then takes its share of the resident workgroups)."""
        super().__init__()
        if hasattr(self._lib, "fbstab_hip_mpc_create_in_flight"):
            _check(self._lib, self._lib.fbstab_hip_mpc_create_in_flight(
                N, nx, nu, nc, max_batch, device, handles_in_flight, C.byref(self._h)))
        else:  # (a build of an earlier round, loaded for an A/B)
            _check(self._lib, self._lib.fbstab_hip_mpc_create(N, nx, nu, nc, max_batch, device, C.byref(self._h)))
        self.N, self.nx, self.nu, self.nc = N, nx, nu, nc
        self.nz, self.nl, self.nv = (N + 1) * (nx + nu), (N + 1) * nx, (N + 1) * nc
        self.seq_len = [(N + 1) * nx * nx, (N + 1) * nu * nu, (N + 1) * nu * nx,
                        (N + 1) * nx, (N + 1) * nu, N * nx * nx, N * nx * nu, N * nx,
                        (N + 1) * nc * nx, (N + 1) * nc * nu, (N + 1) * nc, nx]

    def kernel_name(self) -> str:
        return self._lib.fbstab_hip_mpc_kernel_name(self._h).decode()

    def refined_steps(self) -> int:
        """Newton steps of the last call that were refined (fbstab_hip_mpc_refined_steps)."""
        n = C.c_longlong(-1)
        _check(self._lib, self._lib.fbstab_hip_mpc_refined_steps(self._h, C.byref(n)))
        return int(n.value)

    def Solve(self, data: Dict[str, object], z, l, v, y, out=None, stream: int = 0,
              async_: bool = False, keep_matrices: bool = False):
        """``data``: dict of the 12 sequences, each ``(batch, len)`` float64
        (all numpy, or all torch CUDA tensors); ``z,l,v``: initial guess,
        overwritten with the solution, ``y`` overwritten (fbstab_mpc.h:181-195).
        ``keep_matrices``: FBSTAB_HIP_KEEP_MATRICES (receding horizon: only
        q, r, c, d, x0 and the guess changed since the previous flagged call)."""
        return self._solve(_MpcBatch(), MPC_SEQ, self.seq_len, data,
                           (self.nz, self.nl, self.nv, self.nv), z, l, v, y, out,
                           stream, async_, keep_matrices)

    def SolveFinal(self, data, z, l, v, y, out=None, stream: int = 0, async_: bool = False):
        """Solve followed by the numbers of the reference's Display::FINAL summary
        (fbstab_hip_mpc_solve_batch_final): returns ``(out, norms)``, norms ``(batch, 4)``
        = |rz|, |rl|, |rv|, tolerance at the returned point."""
        return self._solve(_MpcBatch(), MPC_SEQ, self.seq_len, data,
                           (self.nz, self.nl, self.nv, self.nv), z, l, v, y, out, stream, async_, norms=True)

    def SolveTraced(self, data, z, l, v, y, capacity: int = 4096):
        """One QP (``(1, n)`` numpy arrays) with the reference's per-iteration
        display returned as records (fbstab_hip_mpc_solve_traced)."""
        return _solve_traced(self, _MpcBatch(), MPC_SEQ, self.seq_len, data,
                             (self.nz, self.nl, self.nv, self.nv), z, l, v, y, capacity)

    def RecedingSweep(self, data, z, l, v, y, A, B, steps: int, retire: bool = True,
                      log_inputs: bool = False, stream: int = 0):
        """fbstab_hip_mpc_receding_sweep: ``steps`` warm-started closed-loop steps on
        the device (torch CUDA tensors; ``data["x0"]`` is advanced in place, ``z, l,
        v, y`` hold the last solution).  ``A``/``B``: the simulation model as
        ``(nx, nx)``/``(nx, nu)`` numpy arrays shared by all trajectories.  Returns
        ``dict(out, stats, kernel_ms[, u])`` with ``stats`` a structured array per
        step (newton_sum, success, retired_total, newton_max)."""
        import torch
        b = _MpcBatch()
        B_ = None
        for i, (k, n) in enumerate(zip(MPC_SEQ, self.seq_len)):
            p, st, d = _ptr_stride(data[k], n)
            assert d, "device tensors only"
            b.base[i], b.stride[i] = p, st
            B_ = data[k].shape[0] if B_ is None else B_
        vb = _VarBatch()
        for i, (a, n) in enumerate(zip((z, l, v, y), (self.nz, self.nl, self.nv, self.nv))):
            p, st, d = _ptr_stride(a, n)
            assert d and a.shape[0] == B_
            vb.base[i], vb.stride[i] = p, st
        dev = z.device
        Ad = torch.from_numpy(np.asfortranarray(A).T.copy().reshape(-1)).to(dev)   # column-major image
        Bd = torch.from_numpy(np.asfortranarray(B).T.copy().reshape(-1)).to(dev)
        plant = _Plant(Ad.data_ptr(), Bd.data_ptr(), 0, 0)
        out = torch.zeros((B_, 40), dtype=torch.uint8, device=dev)
        stats = np.zeros((steps, 4), dtype=np.uint64)
        kms = np.zeros(steps, dtype=np.float32)
        u = torch.zeros((steps, B_, self.nu), dtype=torch.float64, device=dev) if log_inputs else None
        if not stream:
            stream = torch.cuda.current_stream(dev).cuda_stream
        _check(self._lib, self._lib.fbstab_hip_mpc_receding_sweep(
            self._h, B_, C.byref(b), C.byref(vb), out.data_ptr(), C.byref(plant), steps, 1 if retire else 0,
            u.data_ptr() if u is not None else None, stats.ctypes.data, kms.ctypes.data,
            C.c_void_p(stream) if stream else None))
        st = np.zeros(steps, dtype=[("newton_sum", np.int64), ("success", np.int64),
                                    ("retired_total", np.int64), ("newton_max", np.int64)])
        for j, n in enumerate(st.dtype.names):
            st[n] = stats[:, j].astype(np.int64)
        r = dict(out=out, stats=st, kernel_ms=kms)
        if u is not None:
            r["u"] = u
        return r


    def debug_newton(self, data, z, l, v, zb, lb, vb):
        """Tests only: one Newton step of the device path at (x, xbar,
        sigma0, alpha of the current options) for ONE QP (numpy arrays).
        Returns dict(dz, dl, dv, adz, wz, wl, rz, rl, ok)."""
        lib = self._lib
        lib.fbstab_hip_mpc_debug_newton.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        b = _MpcBatch()
        keep = []
        for i, k in enumerate(MPC_SEQ):
            a = np.ascontiguousarray(np.asarray(data[k], dtype=np.float64).reshape(-1))
            keep.append(a)
            b.base[i] = a.ctypes.data
            b.stride[i] = a.size
        vb_ = _VarBatch()
        xs = [np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1)) for a in (z, l, v)]
        xs.append(np.zeros(self.nv))
        for i, a in enumerate(xs):
            vb_.base[i] = a.ctypes.data
            vb_.stride[i] = a.size
        nz, nl, nv = self.nz, self.nl, self.nv
        io = np.zeros(3 * nz + 3 * nl + 2 * nv + 1)
        io[:nz + nl + nv] = np.concatenate([np.ravel(zb), np.ravel(lb), np.ravel(vb)])
        _check(lib, lib.fbstab_hip_mpc_debug_newton(self._h, C.byref(b), C.byref(vb_), io.ctypes.data))
        names = ("dz", "dl", "dv", "adz", "wz", "wl", "rz", "rl")
        sizes = (nz, nl, nv, nv, nz, nl, nz, nl)
        out, o = {}, 0
        for n, s in zip(names, sizes):
            out[n] = io[o:o + s].copy()
            o += s
        out["ok"] = bool(io[o] > 0.5)
        return out


class FBstabDenseBatch(_SolverBase):
    """Batched counterpart of ``fbstab::FBstabDense`` (fbstab/fbstab_dense.h:50-194)."""
    _kind = "dense"

    def __init__(self, nz: int, nl: int, nv: int, max_batch: int = 1, device: int = 0):
        super().__init__()
        _check(self._lib, self._lib.fbstab_hip_dense_create(
            nz, nl, nv, max_batch, device, C.byref(self._h)))
        self.nz, self.nl, self.nv = nz, nl, nv
        self.arr_len = [nz * nz, nz, nl * nz, nl, nv * nz, nv]

    def Solve(self, data: Dict[str, object], z, l, v, y, out=None, stream: int = 0,
              async_: bool = False):
        return self._solve(_DenseBatch(), DENSE_ARR, self.arr_len, data,
                           (self.nz, self.nl, self.nv, self.nv), z, l, v, y, out,
                           stream, async_)

    ORDER_AUTO, ORDER_PIVOTED, ORDER_NATURAL = 0, 1, 2

    def SetFactorisation(self, order: int = 0, spread_bits: int = 0):
        """fbstab_hip_dense_set_factorisation: elimination order of the LDL' of the KKT
        matrix (dense_cholesky_solver.cc:70-79); ``spread_bits`` 0 keeps the current value."""
        _check(self._lib, self._lib.fbstab_hip_dense_set_factorisation(self._h, order, spread_bits))

    def Factorisation(self) -> Dict[str, int]:
        """Settings in force and the number of Newton steps of the last batch that went to
        the pivoted factorisation behind a natural-order attempt (-1: does not apply)."""
        o, b, n = C.c_int(), C.c_int(), C.c_longlong()
        _check(self._lib, self._lib.fbstab_hip_dense_get_factorisation(
            self._h, C.byref(o), C.byref(b), C.byref(n)))
        return dict(order=o.value, spread_bits=b.value, pivoted_steps=n.value)

    def SolveFinal(self, data, z, l, v, y, out=None, stream: int = 0, async_: bool = False):
        """As FBstabMpcBatch.SolveFinal (fbstab_hip_dense_solve_batch_final)."""
        return self._solve(_DenseBatch(), DENSE_ARR, self.arr_len, data,
                           (self.nz, self.nl, self.nv, self.nv), z, l, v, y, out, stream, async_, norms=True)

    def debug_newton(self, data, z, l, v, zb, lb, vb):
        """Tests only: one Newton step of the dense device path at (x, xbar, sigma0,
        alpha of the current options) for ONE QP (numpy arrays).  Returns
        dict(dz, dl, dv, adz, wz, wl, rz, rl, ok)."""
        lib = self._lib
        lib.fbstab_hip_dense_debug_newton.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        b = _DenseBatch()
        keep = []
        for i, k in enumerate(DENSE_ARR):
            a = np.ascontiguousarray(np.asarray(data[k], dtype=np.float64).reshape(-1))
            if a.size == 0:
                a = np.zeros(1)
            keep.append(a)
            b.base[i] = a.ctypes.data
            b.stride[i] = a.size
        vb_ = _VarBatch()
        xs = [np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1)) for a in (z, l, v)]
        xs.append(np.zeros(self.nv))
        for i, a in enumerate(xs):
            if a.size == 0:
                a = xs[i] = np.zeros(1)
            vb_.base[i] = a.ctypes.data
            vb_.stride[i] = a.size
        nz, nl, nv = self.nz, self.nl, self.nv
        io = np.zeros(3 * nz + 3 * nl + 2 * nv + 1)
        io[:nz + nl + nv] = np.concatenate([np.ravel(zb), np.ravel(lb), np.ravel(vb)])
        _check(lib, lib.fbstab_hip_dense_debug_newton(self._h, C.byref(b), C.byref(vb_), io.ctypes.data))
        names = ("dz", "dl", "dv", "adz", "wz", "wl", "rz", "rl")
        sizes = (nz, nl, nv, nv, nz, nl, nz, nl)
        out, o = {}, 0
        for n, sz in zip(names, sizes):
            out[n] = io[o:o + sz].copy()
            o += sz
        out["ok"] = bool(io[o] > 0.5)
        return out

    def SolveTraced(self, data, z, l, v, y, capacity: int = 4096):
        """One QP (``(1, n)`` numpy arrays) with the reference's per-iteration
        display returned as records (fbstab_hip_dense_solve_traced)."""
        return _solve_traced(self, _DenseBatch(), DENSE_ARR, self.arr_len, data,
                             (self.nz, self.nl, self.nv, self.nv), z, l, v, y, capacity)


class ShardGroup:
    """fbstab_hip_shard_group_*: the GPUs of one node driven by ONE process; shard d of a
    batch lives on ``devices[d]`` and is solved by ``solvers[d]`` (a solver created on
    that device); the results are gathered to ``devices[root]`` in one RCCL operation
    (fbstab_hip_*_solve_batch_sharded).  Arrays are torch CUDA tensors."""

    def __init__(self, devices: Sequence[int]):
        self._lib = load_library()
        self._g = C.c_void_p()
        self.devices = list(devices)
        arr = (C.c_int * len(self.devices))(*self.devices)
        _check(self._lib, self._lib.fbstab_hip_shard_group_create(len(self.devices), arr, C.byref(self._g)))

    def close(self):
        if getattr(self, "_g", None) and self._g.value:
            self._lib.fbstab_hip_shard_group_destroy(self._g)
            self._g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stats(self) -> Dict[str, int]:
        a, b = C.c_longlong(), C.c_longlong()
        _check(self._lib, self._lib.fbstab_hip_shard_group_stats(self._g, C.byref(a), C.byref(b)))
        return dict(gathers=a.value, rccl_ops=b.value)

    def _shards(self, solvers, names, lens, data, xs, outs):
        n = len(self.devices)
        assert len(solvers) == len(data) == len(xs) == len(outs) == n
        kind = solvers[0]._kind
        bs = ((_MpcBatch if kind == "mpc" else _DenseBatch) * n)()
        vs = (_VarBatch * n)()
        counts = (C.c_int * n)()
        hs = (C.c_void_p * n)(*[s._h.value for s in solvers])
        op = (C.c_void_p * n)()
        var_lens = (solvers[0].nz, solvers[0].nl, solvers[0].nv, solvers[0].nv)
        for d in range(n):
            B = None
            for i, (k, ln) in enumerate(zip(names, lens)):
                if ln == 0:
                    bs[d].base[i], bs[d].stride[i] = None, 0
                    continue
                p, st, dev = _ptr_stride(data[d][k], ln)
                assert dev
                bs[d].base[i], bs[d].stride[i] = p, st
                B = data[d][k].shape[0] if B is None else B
            _fill_var(vs[d], xs[d], var_lens)
            counts[d] = B
            op[d] = outs[d].data_ptr()
        return kind, bs, vs, counts, hs, op, var_lens

    def Solve(self, solvers, data, xs, outs, root: int, root_x, root_out):
        """``data[d]``: dict of shard d's arrays, ``xs[d] = (z, l, v, y)``, ``outs[d]``: ``(B_d, 40)``
        uint8, all on ``devices[d]``; ``root_x = (z, l, v, y)`` and ``root_out`` hold the whole batch
        on ``devices[root]``."""
        names, lens = (MPC_SEQ, solvers[0].seq_len) if solvers[0]._kind == "mpc" else (DENSE_ARR, solvers[0].arr_len)
        kind, bs, vs, counts, hs, op, var_lens = self._shards(solvers, names, lens, data, xs, outs)
        rv = _VarBatch()
        _fill_var(rv, root_x, var_lens)
        _check(self._lib, getattr(self._lib, f"fbstab_hip_{kind}_solve_batch_sharded")(
            self._g, hs, counts, bs, vs, op, root, C.byref(rv), C.c_void_p(root_out.data_ptr())))

    def RecedingSweep(self, solvers, data, xs, outs, A, B, steps: int, retire: bool, u_logs, root: int, root_u_log):
        """fbstab_hip_mpc_receding_sweep_sharded; ``A``/``B``: one plant (numpy) for all trajectories;
        ``u_logs[d]``: ``(steps, B_d, nu)`` on devices[d]; ``root_u_log``: the shards' logs one after the
        other on devices[root].  Returns the per-step statistics summed over the shards."""
        import torch
        kind, bs, vs, counts, hs, op, _ = self._shards(solvers, MPC_SEQ, solvers[0].seq_len, data, xs, outs)
        n = len(self.devices)
        plants = (_Plant * n)()
        keep = []
        for d in range(n):
            dev = xs[d][0].device
            Ad = torch.from_numpy(np.asfortranarray(np.asarray(A, dtype=np.float64)).T.copy().reshape(-1)).to(dev)
            Bd = torch.from_numpy(np.asfortranarray(np.asarray(B, dtype=np.float64)).T.copy().reshape(-1)).to(dev)
            keep += [Ad, Bd]
            plants[d] = _Plant(Ad.data_ptr(), Bd.data_ptr(), 0, 0)
        ul = (C.c_void_p * n)(*[u.data_ptr() for u in u_logs])
        stats = np.zeros((steps, 4), dtype=np.uint64)
        _check(self._lib, self._lib.fbstab_hip_mpc_receding_sweep_sharded(
            self._g, hs, counts, bs, vs, op, plants, steps, 1 if retire else 0, ul, root,
            C.c_void_p(root_u_log.data_ptr()), stats.ctypes.data))
        return stats


def _fill_var(vb, x, var_lens):
    for i, (a, n) in enumerate(zip(x, var_lens)):
        if n == 0:
            vb.base[i], vb.stride[i] = None, 0
            continue
        p, st, dev = _ptr_stride(a, n)
        assert dev
        vb.base[i], vb.stride[i] = p, st

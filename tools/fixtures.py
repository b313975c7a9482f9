"""Problem fixtures and synthetic workloads (pure data generation, numpy only).

Two groups:

* ``OcpGenerator`` mirrors the reference's test fixture generator
  ``fbstab/test/ocp_generator.{h,cc}`` (the four canned LTI optimal control
  problems its end-to-end tests solve), producing the 11 stage sequences in the
  reference ``MatrixSequence`` layout ``data[k*nr*nc + j*nr + i]``
  (``tools/matrix_sequence.h:81-83``).
* ``synthetic_mpc_batch`` / ``synthetic_dense_batch`` are the BASELINE.json
  workloads: a counter-based generator
  ``splitmix64`` -> uniform double, keyed by (seed, global instance id,
  element), so that sharding a batch over ranks never changes the data.

Nothing here imports the oracle or the HIP library; the module is test and
benchmark infrastructure, not part of the product package.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List

import numpy as np

_MPC_SEQ = ("Q", "R", "S", "q", "r", "A", "B", "c", "E", "L", "d", "x0")
_DENSE_SEQ = ("H", "f", "G", "h", "A", "b")


def _colmajor(mats: List[np.ndarray]) -> np.ndarray:
    """Sequence of 2-D matrices -> flat reference layout (stage slowest,
    column-major inside a matrix)."""
    return np.ascontiguousarray(
        np.stack([np.asarray(m, dtype=np.float64).reshape(m.shape[0], -1).T
                  for m in mats])).reshape(-1)


@dataclass
class MpcProblem:
    """One or more MPC QPs in the reference layout.

    Every array has shape ``(batch, len)`` with ``len`` the flat length of the
    sequence for one QP (fbstab_mpc.h:67-81): Q (N+1)*nx*nx, R (N+1)*nu*nu,
    S (N+1)*nu*nx, q (N+1)*nx, r (N+1)*nu, A N*nx*nx, B N*nx*nu, c N*nx,
    E (N+1)*nc*nx, L (N+1)*nc*nu, d (N+1)*nc, x0 nx.
    """
    N: int
    nx: int
    nu: int
    nc: int
    arrays: Dict[str, np.ndarray] = field(default_factory=dict)

    @property
    def batch(self) -> int:
        return self.arrays["x0"].shape[0]

    @property
    def nz(self) -> int:
        return (self.N + 1) * (self.nx + self.nu)

    @property
    def nl(self) -> int:
        return (self.N + 1) * self.nx

    @property
    def nv(self) -> int:
        return (self.N + 1) * self.nc

    def sizes(self):
        return (self.N, self.nx, self.nu, self.nc)

    def seq_lengths(self) -> Dict[str, int]:
        N, nx, nu, nc = self.sizes()
        return dict(Q=(N + 1) * nx * nx, R=(N + 1) * nu * nu, S=(N + 1) * nu * nx,
                    q=(N + 1) * nx, r=(N + 1) * nu, A=N * nx * nx, B=N * nx * nu,
                    c=N * nx, E=(N + 1) * nc * nx, L=(N + 1) * nc * nu,
                    d=(N + 1) * nc, x0=nx)

    def doubles_per_qp(self) -> int:
        return sum(self.seq_lengths().values())


@dataclass
class DenseProblem:
    """One or more dense QPs: H nz*nz, G nl*nz, A nv*nz column-major
    (fbstab_dense.h:55-64); arrays are ``(batch, len)``."""
    nz: int
    nl: int
    nv: int
    arrays: Dict[str, np.ndarray] = field(default_factory=dict)
    solution: Dict[str, np.ndarray] = field(default_factory=dict)

    @property
    def batch(self) -> int:
        return self.arrays["f"].shape[0]

    def doubles_per_qp(self) -> int:
        return (self.nz * self.nz + self.nl * self.nz + self.nv * self.nz +
                self.nz + self.nl + self.nv)


def dense_problem(H, f, G, h, A, b) -> DenseProblem:
    """Single dense QP from 2-D numpy matrices (row-major literals, like the
    Eigen ``<<`` initialisers in fbstab_dense_unit_tests.cc)."""
    H = np.atleast_2d(np.asarray(H, dtype=np.float64))
    A = np.atleast_2d(np.asarray(A, dtype=np.float64))
    f = np.asarray(f, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    nz, nv = f.size, b.size
    h = np.asarray(h, dtype=np.float64).reshape(-1)
    nl = h.size
    G = np.asarray(G, dtype=np.float64).reshape(nl, nz)
    p = DenseProblem(nz, nl, nv)
    p.arrays = dict(H=H.T.reshape(1, -1).copy(), f=f.reshape(1, -1).copy(),
                    G=G.T.reshape(1, -1).copy(), h=h.reshape(1, -1).copy(),
                    A=A.T.reshape(1, -1).copy(), b=b.reshape(1, -1).copy())
    return p


# ---------------------------------------------------------------------------
# fbstab/test/ocp_generator.cc
class OcpGenerator:
    """Mirror of ``fbstab::test::OcpGenerator`` (ocp_generator.h:21-201)."""

    def __init__(self):
        self._populated = False

    # ocp_generator.cc:373-420: repeat one stage over the horizon, E(0) = 0.
    def CopyOverHorizon(self, Q, R, S, q, r, A, B, c, E, L, d, x0, N):
        Q, R, S, A, B, E, L = [np.atleast_2d(np.asarray(m, dtype=np.float64))
                               for m in (Q, R, S, A, B, E, L)]
        q, r, c, d, x0 = [np.asarray(m, dtype=np.float64).reshape(-1)
                          for m in (q, r, c, d, x0)]
        self.N_ = N
        self.nx_ = Q.shape[0]
        self.nu_ = R.shape[0]
        self.nc_ = E.shape[0]
        L = L.reshape(self.nc_, self.nu_)
        B = B.reshape(self.nx_, self.nu_)
        S = S.reshape(self.nu_, self.nx_)
        E0 = np.zeros_like(E)
        col = lambda v: v.reshape(-1, 1)
        self._seq = dict(
            Q=_colmajor([Q] * (N + 1)), R=_colmajor([R] * (N + 1)),
            S=_colmajor([S] * (N + 1)), q=_colmajor([col(q)] * (N + 1)),
            r=_colmajor([col(r)] * (N + 1)), A=_colmajor([A] * N),
            B=_colmajor([B] * N), c=_colmajor([col(c)] * N),
            E=_colmajor([E0] + [E] * N), L=_colmajor([L] * (N + 1)),
            d=_colmajor([col(d)] * (N + 1)), x0=x0.copy())
        self._sim = dict(A=A.copy(), B=B.copy())
        self._populated = True

    def GetFBstabInput(self) -> MpcProblem:
        if not self._populated:
            raise RuntimeError("In OcpGenerator::GetFBstabInput: Call a problem "
                               "creator method first.")
        p = MpcProblem(self.N_, self.nx_, self.nu_, self.nc_)
        p.arrays = {k: v.reshape(1, -1).copy() for k, v in self._seq.items()}
        return p

    def GetSimulationInputs(self):
        if not self._populated:
            raise RuntimeError("In OcpGenerator::GetSimulationInputs: Call a "
                               "problem creator method first.")
        return dict(x0=self._seq["x0"].copy(), A=self._sim["A"], B=self._sim["B"],
                    C=self._sim["C"], D=self._sim["D"], T=self._sim["T"])

    def ProblemSize(self):
        return (self.N_, self.nx_, self.nu_, self.nc_)

    def nz(self):
        return (self.nx_ + self.nu_) * (self.N_ + 1)

    def nl(self):
        return self.nx_ * (self.N_ + 1)

    def nv(self):
        return self.nc_ * (self.N_ + 1)

    # ocp_generator.cc:329-371
    def DoubleIntegrator(self, N=10):
        if N <= 0:
            raise RuntimeError("In OcpGenerator::DoubleIntegrator: N <= 0.")
        Q = [[2, 0], [0, 1]]
        S = [[1, 0]]
        R = [[3]]
        q = [-2, 0]
        r = [0]
        A = [[1, 1], [0, 1]]
        B = [[0], [1]]
        c = [0, 0]
        E = [[-1, 0], [0, -1], [1, 0], [0, 1], [0, 0], [0, 0]]
        L = [[0], [0], [0], [0], [-1], [1]]
        d = [0, 0, -2, -2, -1, -1]
        x0 = [0, 0]
        self.CopyOverHorizon(Q, R, S, q, r, A, B, c, E, L, d, x0, N)
        self._sim.update(C=np.eye(2), D=np.zeros((2, 1)), T=40)

    # ocp_generator.cc:253-325
    def ServoMotor(self, N=20):
        if N <= 0:
            raise RuntimeError("In OcpGenerator::ServoMotor: N <= 0.")
        kt, bl, Jm, bm, ktheta, RR, rho = 10.0, 25.0, 0.5, 0.1, 1280.2, 20.0, 20.0
        Jl = 20 * Jm
        umax, ymax = 220.0, 78.5358
        A = np.array([[0, 1, 0, 0],
                      [-ktheta / Jl, -bl / Jl, ktheta / (rho * Jl), 0],
                      [0, 0, 0, 1],
                      [ktheta / (rho * Jm), 0, -ktheta / (rho * rho * Jm),
                       -(bm + kt * kt / RR) / Jm]])
        B = np.array([[0], [0], [0], [kt / (RR * Jm)]])
        C = np.array([[1, 0, 0, 0], [ktheta, 0, -ktheta / rho, 0]])
        ts = 0.05
        A = np.eye(4) + ts * A
        B = ts * B
        c = np.zeros(4)
        x0 = np.zeros(4)
        Q = np.zeros((4, 4))
        R = np.zeros((1, 1))
        S = np.zeros((1, 4))
        Q[0, 0] = 1000
        R[0, 0] = 1e-4
        pi = 3.1415926535897
        xtrg = np.array([30 * pi / 180, 0, 0, 0])
        utrg = np.array([0.0])
        q = -Q @ xtrg
        r = -R @ utrg
        E = np.vstack([C[1], -C[1], np.zeros((2, 4))])
        L = np.array([[0], [0], [1], [-1]], dtype=np.float64)
        d = np.array([-ymax, -ymax, -umax, -umax])
        self.CopyOverHorizon(Q, R, S, q, r, A, B, c, E, L, d, x0, N)
        self._sim.update(C=C, D=np.zeros((2, 1)), T=40)

    # ocp_generator.cc:175-252
    def SpacecraftRelativeMotion(self, N=40):
        if N <= 0:
            raise RuntimeError("In OcpGenerator::SpacecraftRelativeMotion: N <= 0.")
        mu, Re, alt = 398600.4418, 6371.0, 650.0
        n = math.sqrt(mu / math.pow(Re + alt, 3))
        A21 = np.array([[2 * n * n, 0, 0], [0, 0, 0], [0, 0, -n * n]])
        A22 = np.array([[0, 2 * n, 0], [-2 * n, 0, 0], [0, 0, 0]])
        A = np.block([[np.zeros((3, 3)), np.eye(3)], [A21, A22]])
        B = np.vstack([np.zeros((3, 3)), np.eye(3)])
        C = np.eye(6)
        ts = 30.0
        A = np.eye(6) + ts * A
        B = ts * B
        B = A @ B
        c = np.zeros(6)
        x0 = np.array([-2.8, -0.01, -1, 0, 0, 0])
        Q = np.diag([1, 1, 1, 1e-3, 1e-3, 1e-3]).astype(np.float64)
        R = np.eye(3)
        S = np.zeros((3, 6))
        q = np.zeros(6)
        r = np.zeros(3)
        umax = 1e-3
        vmax = 1e-3
        E = np.vstack([np.zeros((6, 6)),
                       np.hstack([np.zeros((3, 3)), np.eye(3)]),
                       np.hstack([np.zeros((3, 3)), -np.eye(3)])])
        L = np.vstack([np.eye(3), -np.eye(3), np.zeros((6, 3))])
        d = np.concatenate([-umax * np.ones(6), -vmax * np.ones(6)])
        self.CopyOverHorizon(Q, R, S, q, r, A, B, c, E, L, d, x0, N)
        self._sim.update(C=C, D=np.zeros((6, 3)), T=100)

    # ocp_generator.cc:73-174
    def CopolymerizationReactor(self, N=70):
        if N <= 0:
            raise RuntimeError("In OcpGenerator::CopolymerizationReactor: N <= 0.")
        A = np.zeros((18, 18))
        i = [1, 2, 3, 4, 5, 6, 7, 8, 7, 8, 9, 10, 11, 12, 13, 12, 13, 14, 15, 16,
             15, 16, 17, 18, 17, 18]
        j = [1, 2, 3, 4, 5, 6, 7, 7, 8, 8, 9, 10, 11, 12, 12, 13, 13, 14, 15, 15,
             16, 16, 17, 17, 18, 18]
        v = [0.55531, 0.81264, 0.82131, 0.30408, 0.71811, 0.72276, 0.97319,
             0.12353, -0.16471, 0.98966, 0.70834, 0.69048, 0.83152, -0.016569,
             0.07277, -0.040608, 0.17835, 0.53526, -0.015422, 0.04805, -0.093847,
             0.2924, -0.22577, 0.43126, -0.38505, 0.2517]
        for a, b_, val in zip(i, j, v):
            A[a - 1, b_ - 1] = val
        B = np.zeros((18, 5))
        i = list(range(1, 19))
        j = [1, 1, 1, 2, 2, 2, 3, 3, 3, 3, 4, 5, 5, 5, 5, 5, 5, 5]
        v = [0.18899, 0.22577, 0.11347, 0.14614, 0.21282, 0.21347, 0.24707,
             0.015512, 0.21145, 0.41785, 0.11415, 0.14554, 2.9448, 0.1859,
             0.04805, 0.36229, 0.21563, 0.41905]
        for a, b_, val in zip(i, j, v):
            B[a - 1, b_ - 1] = val
        C = np.zeros((4, 18))
        C[0] = [0.8, 0, 0, 1, 0, 0, 0.0416666666666667, 0.333333333333333, 0, 0,
                0, 25.9553571428571, 1.80245535714286, 0, 0, 0, 0, 0]
        C[1] = [0, -0.340248962655602, 0, 0, 0.874172185430464, 0, 0, 0,
                -0.413793103448276, 0, 0, 0, 0, -0.930000000000000, 0, 0, 0, 0]
        C[2] = [0, 0, 0.47244, 0, 0, 0.63636, 0, 0, 0, -0.52593, -0.2952, 0, 0,
                0, 0, -9.1992, 0, 0]
        C[3] = [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1.6757, 1.8214]
        c = np.zeros(18)
        x0 = np.array([0.2 * math.sin(k + 1) for k in range(18)])
        Q = C.T @ C
        R = 0.1 * np.eye(5)
        S = np.zeros((5, 18))
        q = np.zeros(18)
        r = np.zeros(5)
        umax = 5.0 / 100.0
        E = np.zeros((10, 18))
        L = np.vstack([np.eye(5), -np.eye(5)])
        d = -umax * np.ones(10)
        self.CopyOverHorizon(Q, R, S, q, r, A, B, c, E, L, d, x0, N)
        self._sim.update(C=C, D=np.zeros((4, 5)), T=200)


# ---------------------------------------------------------------------------
# Counter-based RNG (SURVEY.md section 8d "RNG / seeds").
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x: np.ndarray) -> np.ndarray:
    """One splitmix64 output step applied element-wise to uint64 ``x``."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(seed: int, instance: np.ndarray, n_elements: int,
              stream: int = 0) -> np.ndarray:
    """``(len(instance), n_elements)`` uniforms in [0,1):
    ``u = splitmix64(splitmix64(seed ^ stream<<56 ^ instance) + element)``,
    value ``(u >> 11) * 2**-53``."""
    inst = np.asarray(instance, dtype=np.uint64).reshape(-1, 1)
    key = splitmix64(np.uint64(seed) ^ (np.uint64(stream) << np.uint64(56)) ^ inst)
    el = np.arange(n_elements, dtype=np.uint64).reshape(1, -1)
    with np.errstate(over="ignore"):
        u = splitmix64(key + el)
    return (u >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)


MASTER_SEED = 12345


def quadrotor_model(ts: float = 0.1):
    """Hover-linearised quadrotor-like LTI model, forward Euler
    (12 states: position, velocity, roll/pitch/yaw, body rates; 4 inputs:
    thrust deviation and three torques)."""
    g, m = 9.81, 1.0
    J = np.array([0.5, 0.5, 1.0])
    Ac = np.zeros((12, 12))
    Ac[0:3, 3:6] = np.eye(3)
    Ac[3, 7] = g       # vx' =  g * pitch
    Ac[4, 6] = -g      # vy' = -g * roll
    Ac[6:9, 9:12] = np.eye(3)
    Bc = np.zeros((12, 4))
    Bc[5, 0] = 1.0 / m
    Bc[9, 1] = 1.0 / J[0]
    Bc[10, 2] = 1.0 / J[1]
    Bc[11, 3] = 1.0 / J[2]
    return np.eye(12) + ts * Ac, ts * Bc


def synthetic_mpc_batch(batch: int, first_id: int = 0, seed: int = MASTER_SEED,
                        N: int = 30) -> MpcProblem:
    """BASELINE.json config 3/4 workload: ``batch`` MPC QPs of shape
    N=30, nx=12, nu=4, nc=20 with global instance ids
    ``first_id .. first_id+batch-1``.  All 11 sequences are materialised per
    QP (the API is time-varying; nothing is shared across the batch)."""
    nx, nu, nc = 12, 4, 20
    A, B = quadrotor_model()
    Q = np.diag([10, 10, 10, 1, 1, 1, 5, 5, 5, .1, .1, .1]).astype(np.float64)
    R = 0.1 * np.eye(nu)
    S = np.zeros((nu, nx))
    umax = np.array([4.0, 2.0, 2.0, 1.0])
    E = np.zeros((nc, nx))
    L = np.zeros((nc, nu))
    d = np.zeros(nc)
    L[0:4] = np.eye(4)
    L[4:8] = -np.eye(4)
    d[0:4] = -umax
    d[4:8] = -umax
    E[8:11, 6:9] = np.eye(3)
    E[11:14, 6:9] = -np.eye(3)
    d[8:14] = -0.35
    E[14:17, 3:6] = np.eye(3)
    E[17:20, 3:6] = -np.eye(3)
    d[14:20] = -2.0
    gen = OcpGenerator()
    gen.CopyOverHorizon(Q, R, S, np.zeros(nx), np.zeros(nu), A, B, np.zeros(nx),
                        E, L, d, np.zeros(nx), N)
    one = gen.GetFBstabInput()
    ids = np.arange(first_id, first_id + batch, dtype=np.uint64)
    u = uniform01(seed, ids, nx, stream=1)
    x0 = np.zeros((batch, nx))
    x0[:, 0:3] = (2.0 * u[:, 0:3] - 1.0) * 3.0
    x0[:, 3:6] = (2.0 * u[:, 3:6] - 1.0) * 1.0
    x0[:, 6:9] = (2.0 * u[:, 6:9] - 1.0) * 0.2
    p = MpcProblem(N, nx, nu, nc)
    for k in _MPC_SEQ:
        if k == "x0":
            p.arrays[k] = x0
        else:
            p.arrays[k] = np.ascontiguousarray(
                np.broadcast_to(one.arrays[k], (batch, one.arrays[k].shape[1])))
    return p


def boxed_mpc_batch(batch: int, first_id: int = 0, seed: int = MASTER_SEED, N: int = 30) -> MpcProblem:
    """The plant and initial states of ``synthetic_mpc_batch`` with two-sided bounds on
    ALL sixteen stage variables, written as 32 single-entry rows per stage
    (E x + L u + d <= 0): the shape of the ``<12,4,32>`` record instance."""
    p = synthetic_mpc_batch(batch, first_id=first_id, seed=seed, N=N)
    nx, nu = p.nx, p.nu
    nc = 2 * (nx + nu)
    xmax = np.array([6.0, 6.0, 6.0, 2.0, 2.0, 2.0, 0.35, 0.35, 0.35, 3.0, 3.0, 3.0])
    umax = np.array([4.0, 2.0, 2.0, 1.0])
    E = np.zeros((nc, nx)); L = np.zeros((nc, nu)); d = np.zeros(nc)
    E[0:nx] = np.eye(nx); E[nx:2 * nx] = -np.eye(nx)
    d[0:nx] = -xmax; d[nx:2 * nx] = -xmax
    L[2 * nx:2 * nx + nu] = np.eye(nu); L[2 * nx + nu:] = -np.eye(nu)
    d[2 * nx:2 * nx + nu] = -umax; d[2 * nx + nu:] = -umax
    q = MpcProblem(N, nx, nu, nc)
    q.arrays = dict(p.arrays)
    rep = lambda m: np.ascontiguousarray(np.broadcast_to(np.tile(m.T.reshape(-1), N + 1), (batch, (N + 1) * m.size)))
    q.arrays["E"], q.arrays["L"] = rep(E), rep(L)      # column-major images
    q.arrays["d"] = np.ascontiguousarray(np.broadcast_to(np.tile(d, N + 1), (batch, (N + 1) * nc)))
    return q


def synthetic_mpc_ltv_batch(batch: int, first_id: int = 0, seed: int = MASTER_SEED,
                            N: int = 30) -> MpcProblem:
    """The same plant and initial states as ``synthetic_mpc_batch`` posed as a
    time-VARYING problem with DENSE constraint rows, which is what the API
    allows (fbstab_mpc.h:67-81; SURVEY.md 8d assumes no sharing between stages):
    every stage has its own Q, R, S, A, B (smooth drift along the horizon, a
    small state-input cross term) and its own E, L, whose rows are mixtures of
    two or three of the box rows (polytope still containing the origin strictly,
    so the problems stay feasible).  No two stages share matrices and no
    constraint row has a single nonzero: the record kernel can neither share
    matrix copies between stages nor take its bound-constraint path."""
    base = synthetic_mpc_batch(1, first_id=0, seed=seed, N=N)
    nx, nu, nc = base.nx, base.nu, base.nc
    a = {k: v[0].copy() for k, v in base.arrays.items()}
    Q = a["Q"].reshape(N + 1, nx, nx)
    R = a["R"].reshape(N + 1, nu, nu)
    S = a["S"].reshape(N + 1, nx, nu)      # column-major (nu x nx): [col(x), row(u)]
    A = a["A"].reshape(N, nx, nx)          # [col, row]
    B = a["B"].reshape(N, nu, nx)          # [col(u), row(x)]
    E = a["E"].reshape(N + 1, nx, nc)      # [col(x), row(k)]
    L = a["L"].reshape(N + 1, nu, nc)      # [col(u), row(k)]
    mix = np.eye(nc)
    for k in range(nc):
        mix[k, (k + 3) % nc] += 0.30
        mix[k, (k + 7) % nc] += 0.15
    for i in range(N + 1):
        Q[i] *= 1.0 + 0.02 * i
        R[i] *= 1.0 + 0.01 * i
        S[i, 3:6, 0:3] = 0.01 * (1 + i % 4) * np.eye(3)
        rowmix = mix * (1.0 + 0.02 * (i % 5))
        E[i] = E[i] @ rowmix.T
        L[i] = L[i] @ rowmix.T
        if i < N:
            A[i] += 1e-3 * i * np.diag(np.linspace(-1.0, 1.0, nx))
            B[i] *= 1.0 + 0.005 * i
    d = a["d"].reshape(N + 1, nc)
    for i in range(N + 1):
        d[i] = (mix * (1.0 + 0.02 * (i % 5))) @ d[i]
    p = synthetic_mpc_batch(batch, first_id=first_id, seed=seed, N=N)
    for k in _MPC_SEQ:
        if k != "x0":
            p.arrays[k] = np.ascontiguousarray(np.broadcast_to(a[k], (batch, a[k].shape[0])))
    return p


def random_ltv_mpc(rng, batch, N, nx, nu, nc, dyn_noise=0.15):
    """Random time-varying MPC QPs: a positive definite stage Hessian [Q S';S R],
    dynamics near the identity, dense constraint rows with a strictly feasible
    trajectory by construction, small linear terms."""
    ns = nx + nu
    a = {}
    Q = np.zeros((batch, N + 1, nx * nx)); R = np.zeros((batch, N + 1, nu * nu)); S = np.zeros((batch, N + 1, nu * nx))
    for b in range(batch):
        for i in range(N + 1):
            M = rng.standard_normal((ns, ns))
            Hs = M.T @ M / ns + 0.5 * np.eye(ns)
            Q[b, i] = Hs[:nx, :nx].T.reshape(-1)       # column-major images
            R[b, i] = Hs[nx:, nx:].T.reshape(-1)
            S[b, i] = Hs[nx:, :nx].T.reshape(-1)       # S is nu x nx
    a["Q"], a["R"], a["S"] = Q.reshape(batch, -1), R.reshape(batch, -1), S.reshape(batch, -1)
    a["q"] = 0.1 * rng.standard_normal((batch, (N + 1) * nx))
    a["r"] = 0.1 * rng.standard_normal((batch, (N + 1) * nu))
    A = np.eye(nx)[None, None] + dyn_noise * rng.standard_normal((batch, N, nx, nx))
    a["A"] = np.transpose(A, (0, 1, 3, 2)).reshape(batch, -1)
    a["B"] = (0.5 * rng.standard_normal((batch, N, nu, nx))).reshape(batch, -1)   # (nx x nu) column-major
    a["c"] = 0.05 * rng.standard_normal((batch, N * nx))
    a["E"] = (rng.standard_normal((batch, N + 1, nx, nc)) * (rng.random((batch, N + 1, nx, nc)) < 0.4)).reshape(batch, -1)
    a["L"] = (rng.standard_normal((batch, N + 1, nu, nc)) * (rng.random((batch, N + 1, nu, nc)) < 0.6)).reshape(batch, -1)
    a["x0"] = 0.5 * rng.standard_normal((batch, nx))
    # feasible by construction: d is set from a simulated trajectory with small
    # random inputs, with a strictly positive slack
    Bm = np.transpose(a["B"].reshape(batch, N, nu, nx), (0, 1, 3, 2))   # (batch, N, nx, nu)
    Em = np.transpose(a["E"].reshape(batch, N + 1, nx, nc), (0, 1, 3, 2))
    Lm = np.transpose(a["L"].reshape(batch, N + 1, nu, nc), (0, 1, 3, 2))
    cm = a["c"].reshape(batch, N, nx)
    d = np.zeros((batch, N + 1, nc))
    for b in range(batch):
        x = a["x0"][b].copy()
        for i in range(N + 1):
            u = 0.2 * rng.standard_normal(nu)
            d[b, i] = -(Em[b, i] @ x + Lm[b, i] @ u) - (0.2 + 0.8 * rng.random(nc))
            if i < N:
                x = A[b, i] @ x + Bm[b, i] @ u + cm[b, i]
    a["d"] = d.reshape(batch, -1)
    p = MpcProblem(N, nx, nu, nc)
    p.arrays = {k: np.ascontiguousarray(v) for k, v in a.items()}
    return p


def random_ltv_mpc_sparse_rows(rng, batch, N, nx, nu, nc, dyn_noise=0.15):
    """random_ltv_mpc with SPARSE constraint rows: two or three entries of magnitude 0.3 .. 0.8 per row on
    stage variables drawn at random (at most four rows per variable) - rows for which the record kernels
    take the row form of the costate step without being bounds (choose_costate_form: nzmax x cmax2 <= 8),
    like the bench line's time-varying workload."""
    return random_ltv_mpc_bounds(rng, batch, N, nx, nu, nc, dyn_noise, entries=(2, 3))


def random_ltv_mpc_bounds(rng, batch, N, nx, nu, nc, dyn_noise=0.15, entries=None):
    """random_ltv_mpc with BOUND constraints: every constraint row has one entry, +1 or -1, on a stage
    variable drawn at random (at most four rows per variable) - the constraints the record kernels' row
    form of the costate step serves (fb_mpc_r16.h: choose_costate_form) - strictly feasible by construction."""
    p = random_ltv_mpc(rng, batch, N, nx, nu, nc, dyn_noise)
    ns = nx + nu
    E = np.zeros((batch, N + 1, nx, nc)); L = np.zeros((batch, N + 1, nu, nc))
    for b in range(batch):
        for i in range(N + 1):
            load = np.zeros(ns, dtype=int)
            for k in range(nc):
                for _e in range(1 if entries is None else int(rng.integers(entries[0], entries[1] + 1))):
                    free = np.flatnonzero(load < 4)
                    c = int(free[rng.integers(0, len(free))]) if len(free) else int(rng.integers(0, ns))
                    load[c] += 1
                    sgn = 1.0 if rng.random() < 0.5 else -1.0
                    if entries is not None: sgn *= 0.3 + 0.5 * rng.random()
                    if c < nx: E[b, i, c, k] = sgn
                    else: L[b, i, c - nx, k] = sgn
    p.arrays["E"] = np.ascontiguousarray(E.reshape(batch, -1))
    p.arrays["L"] = np.ascontiguousarray(L.reshape(batch, -1))
    A = np.transpose(p.arrays["A"].reshape(batch, N, nx, nx), (0, 1, 3, 2))
    Bm = np.transpose(p.arrays["B"].reshape(batch, N, nu, nx), (0, 1, 3, 2))
    Em = np.transpose(E, (0, 1, 3, 2)); Lm = np.transpose(L, (0, 1, 3, 2))
    cm = p.arrays["c"].reshape(batch, N, nx)
    d = np.zeros((batch, N + 1, nc))
    for b in range(batch):
        x = p.arrays["x0"][b].copy()
        for i in range(N + 1):
            u = 0.2 * rng.standard_normal(nu)
            d[b, i] = -(Em[b, i] @ x + Lm[b, i] @ u) - (0.05 + 0.5 * rng.random(nc))
            if i < N:
                x = A[b, i] @ x + Bm[b, i] @ u + cm[b, i]
    p.arrays["d"] = np.ascontiguousarray(d.reshape(batch, -1))
    return p


def synthetic_dense_batch(batch: int, nz: int, nl: int, nv: int,
                          first_id: int = 0,
                          seed: int = MASTER_SEED) -> DenseProblem:
    """BASELINE.json config 1/2 workload: strictly convex
    random dense QPs with a known optimal primal-dual point (stored in
    ``.solution``)."""
    ids = np.arange(first_id, first_id + batch, dtype=np.uint64)
    sym = lambda n, s: 2.0 * uniform01(seed, ids, n, stream=s) - 1.0
    M = sym(nz * nz, 2).reshape(batch, nz, nz)       # M[b, col, row]
    H = np.einsum("bik,bjk->bij", M, M) / nz + 0.1 * np.eye(nz)
    G = sym(nl * nz, 3).reshape(batch, nz, nl)       # column-major: [col,row]
    A = sym(nv * nz, 4).reshape(batch, nz, nv)
    zs = sym(nz, 5)
    ls = sym(nl, 6)
    s = uniform01(seed, ids, nv, stream=7)
    act = uniform01(seed, ids, nv, stream=8) < 0.25
    s = np.where(act, 0.0, s)
    vs = np.where(act, uniform01(seed, ids, nv, stream=9), 0.0)
    Gz = np.einsum("bcr,bc->br", G, zs)
    Az = np.einsum("bcr,bc->br", A, zs)
    h = Gz
    b = Az + s
    f = -(np.einsum("bij,bj->bi", H, zs) + np.einsum("bcr,br->bc", G, ls) +
          np.einsum("bcr,br->bc", A, vs))
    p = DenseProblem(nz, nl, nv)
    # H is symmetric so its column-major image equals the row-major one.
    p.arrays = dict(H=np.ascontiguousarray(H.reshape(batch, -1)), f=f,
                    G=np.ascontiguousarray(G.reshape(batch, -1)), h=h,
                    A=np.ascontiguousarray(A.reshape(batch, -1)), b=b)
    p.solution = dict(z=zs, l=ls, v=vs)
    return p

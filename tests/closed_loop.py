"""Warm-started receding-horizon sweep (BASELINE.json config 5): many closed-loop
trajectories, each step solving the same MPC QP with a new initial state
``x0 <- A x0 + B u0*`` and the previous solution ``(z, l, v)`` as the initial
guess, UNSHIFTED (the reference has no shift logic; its OcpGenerator only
exposes the simulation matrices, fbstab/test/ocp_generator.h:31-38).

``solve(x0, z, l, v) -> (z, l, v, y, out)`` is any batched solver: the HIP
library (arrays may stay on the device between steps) or, in the tests, the
oracle.  Arrays are ``(trajectories, n)`` numpy or torch, used through the
operations both support.
"""
from __future__ import annotations

from typing import Callable, Dict, List


def closed_loop(solve: Callable, x0, z, l, v, A, B, nx: int, nu: int, steps: int) -> List[Dict]:
    """Runs ``steps`` MPC steps.  ``A``/``B`` are (nx,nx)/(nx,nu) arrays of the
    same kind as ``x0``.  Returns per-step records (copies of x0, u0, out)."""
    log = []
    for k in range(steps):
        z, l, v, y, out = solve(x0, z, l, v)
        u0 = z[:, nx:nx + nu]
        log.append(dict(x0=x0.clone() if hasattr(x0, "clone") else x0.copy(),
                        u0=u0.clone() if hasattr(u0, "clone") else u0.copy(), out=out))
        x0 = x0 @ A.T + u0 @ B.T
    return log

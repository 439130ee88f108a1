"""Problem setup for the batched optimal-control QPs (host side, NumPy).

The reference (README.md:1-2) names only the subject -- ADMM for astrodynamics
problems -- and ships no problem-setup code, so the generators here are this
repository's own specification.  They produce plain arrays that
are handed unchanged to the HIP solver and to the CPU oracle; neither path
generates its own inputs.

Array conventions: "QP-major" C-order arrays, shape (batch, L) etc., which is
MATLAB's column-major L x batch seen from C.
"""
from __future__ import annotations

import dataclasses
from typing import Optional

import numpy as np

MU_EARTH = 3.986004418e14      # m^3 / s^2
A_REF = 6_778_137.0            # m, circular reference orbit radius (400 km altitude)
SEED0 = 20231004               # fixed base seed of the synthetic workloads


@dataclasses.dataclass
class Problem:
    """One batch of QPs sharing dynamics, weights and box; differing in x0 (and q).

    minimise   1/2 sum_k [u_k' R u_k + x_{k+1}' Q_{k+1} x_{k+1}] + q' w
    subject to x_{k+1} = A_k x_k + B_k u_k,  x_0 given,  lo <= w <= hi
    with w = (u_0, x_1, ..., u_{N-1}, x_N), block k = (u_k, x_{k+1}).
    """
    N: int
    A: np.ndarray                 # (n, n) LTI, (N, n, n) LTV, or (batch, N, n, n) per-instance LTV
    B: np.ndarray                 # (n, m), (N, n, m) or (batch, N, n, m)
    Q: np.ndarray                 # (n, n)
    R: np.ndarray                 # (m, m)
    QN: np.ndarray                # (n, n)
    x0: np.ndarray                # (batch, n)
    lo: np.ndarray                # (m + n,), (N, m + n), or (batch, N, m + n) per instance; -inf allowed
    hi: np.ndarray                # likewise; +inf allowed
    q: Optional[np.ndarray] = None  # (batch, L) or None
    # thrust-magnitude bound ||u_k||_2 <= unorm (scalar, or (N,) when the box is per stage); None = off.
    # Where finite, the box of the control rows must be (-inf, inf).
    unorm: Optional[np.ndarray] = None
    name: str = ""

    @property
    def n(self) -> int:
        return int(self.B.shape[-2])

    @property
    def m(self) -> int:
        return int(self.B.shape[-1])

    @property
    def nb(self) -> int:
        return self.n + self.m

    @property
    def L(self) -> int:
        return self.N * self.nb

    @property
    def batch(self) -> int:
        return int(self.x0.shape[0])

    @property
    def time_varying(self) -> bool:
        return self.A.ndim >= 3

    @property
    def per_instance(self) -> bool:
        """Every QP has its own dynamics (admm_problem.time_varying = 2)."""
        return self.A.ndim == 4

    @property
    def per_instance_bounds(self) -> bool:
        """Every QP has its own per-stage box (admm_problem.stage_bounds = 2)."""
        return self.lo.ndim == 3

    def slice(self, start: int, stop: int) -> "Problem":
        """Contiguous sub-batch [start, stop): the sharding unit."""
        rep = dict(x0=np.ascontiguousarray(self.x0[start:stop]),
                   q=None if self.q is None else np.ascontiguousarray(self.q[start:stop]))
        if self.per_instance:
            rep.update(A=np.ascontiguousarray(self.A[start:stop]), B=np.ascontiguousarray(self.B[start:stop]))
        if self.per_instance_bounds:
            rep.update(lo=np.ascontiguousarray(self.lo[start:stop]), hi=np.ascontiguousarray(self.hi[start:stop]))
        return dataclasses.replace(self, **rep)

    def validate(self) -> None:
        n, m, N = self.n, self.m, self.N
        if N < 1 or n < 1 or m < 1:
            raise ValueError("N, n, m must be positive")
        if self.A.shape[-2:] != (n, n) or self.B.shape[-2:] != (n, m):
            raise ValueError("A/B shape mismatch")
        if self.A.ndim == 3 and self.A.shape[0] != N:
            raise ValueError("time-varying A must have N stages")
        if self.B.ndim == 3 and self.B.shape[0] != N:
            raise ValueError("time-varying B must have N stages")
        if self.A.ndim != self.B.ndim:
            raise ValueError("A and B must both be LTI, both LTV or both per-instance")
        if self.A.ndim == 4 and (self.A.shape[:2] != (self.x0.shape[0], N) or self.B.shape[:2] != (self.x0.shape[0], N)):
            raise ValueError("per-instance A / B must be (batch, N, n, n) / (batch, N, n, m)")
        if self.Q.shape != (n, n) or self.QN.shape != (n, n) or self.R.shape != (m, m):
            raise ValueError("weight shape mismatch")
        if self.x0.ndim != 2 or self.x0.shape[1] != n:
            raise ValueError("x0 must be (batch, n)")
        for b in (self.lo, self.hi):
            if b.shape not in ((n + m,), (N, n + m), (self.x0.shape[0], N, n + m)):
                raise ValueError("bounds must be (n+m,), (N, n+m) or (batch, N, n+m)")
        if self.lo.shape != self.hi.shape:
            raise ValueError("lo and hi must have the same shape")
        if self.lo.ndim == 3 and not self.per_instance:
            raise ValueError("per-instance bounds need per-instance dynamics")
        if np.any(np.isnan(self.lo)) or np.any(np.isnan(self.hi)) or np.any(self.lo > self.hi):
            raise ValueError("bounds must satisfy lo <= hi and contain no NaN")
        if self.q is not None and self.q.shape != (self.batch, self.L):
            raise ValueError("q must be (batch, L)")
        if self.unorm is not None:
            un = np.asarray(self.unorm, np.float64)
            if un.ndim > 1 or (un.ndim == 1 and (self.lo.ndim < 2 or un.shape != (N,))):
                raise ValueError("unorm must be a scalar, or (N,) together with per-stage bounds")
            if np.any(np.isnan(un)) or np.any(un <= 0):
                raise ValueError("unorm must be positive (inf = off)")
            # (stages, m) control-row bounds of every QP: (1 | N, m) shared, (batch, N, m) per instance
            lo_u = (self.lo if self.lo.ndim == 3 else np.atleast_2d(self.lo)[None])[..., :m]
            hi_u = (self.hi if self.hi.ndim == 3 else np.atleast_2d(self.hi)[None])[..., :m]
            fin = np.isfinite(np.broadcast_to(un, (lo_u.shape[1],)))
            if np.any(np.isfinite(lo_u[:, fin])) or np.any(np.isfinite(hi_u[:, fin])):
                raise ValueError("control rows must be unbounded (-inf, inf) where unorm is finite")
        for a in (self.A, self.B, self.Q, self.R, self.QN, self.x0):
            # (stacks of hundreds of MB -- per-instance dynamics -- are left to the library's own threaded check at admm_setup /
            #  admm_update_problem, which Solver reports as the same ValueError: NumPy's pass over 7 GB costs 0.4 s per call)
            if a.nbytes <= (256 << 20) and not np.all(np.isfinite(a)):
                raise ValueError("non-finite problem data")


def double_integrator(N: int = 50, batch: int = 1, seed0: int = SEED0,
                      dt: float = 0.2, u_max: float = 1.0) -> Problem:
    """BASELINE.json configs[0]: 2-state LQR QP, box input constraint |u| <= u_max.
    m = 1 (one force input); instance 0 starts at (4, 0), further instances are
    drawn from default_rng(seed0 + instance)."""
    A = np.array([[1.0, dt], [0.0, 1.0]])
    B = np.array([[0.5 * dt * dt], [dt]])
    Q = np.diag([1.0, 0.1])
    R = np.array([[0.1]])
    QN = np.diag([10.0, 1.0])
    x0 = np.empty((batch, 2))
    for i in range(batch):
        if i == 0:
            x0[i] = (4.0, 0.0)
        else:
            rng = np.random.default_rng(seed0 + i)
            x0[i] = rng.uniform([-5.0, -1.0], [5.0, 1.0])
    inf = np.inf
    lo = np.array([-u_max, -inf, -inf])
    hi = np.array([u_max, inf, inf])
    return Problem(N=N, A=A, B=B, Q=Q, R=R, QN=QN, x0=x0, lo=lo, hi=hi,
                   name=f"double_integrator_N{N}_b{batch}")


def cw_matrices(dt: float):
    """Zero-order-hold discretisation of the Clohessy-Wiltshire equations in
    nondimensional form (time unit 1/mean-motion, so the mean motion is 1):

        x'' = 3 x + 2 y' + u_x,   y'' = -2 x' + u_y,   z'' = -z + u_z

    State (x, y, z, vx, vy, vz).  Closed form of expm([[Ac, Bc],[0,0]] dt)."""
    s, c = np.sin(dt), np.cos(dt)
    A = np.array([
        [4 - 3 * c,       0, 0,  s,            2 * (1 - c),      0],
        [6 * (s - dt),    1, 0, -2 * (1 - c),  4 * s - 3 * dt,   0],
        [0,               0, c,  0,            0,                s],
        [3 * s,           0, 0,  c,            2 * s,            0],
        [-6 * (1 - c),    0, 0, -2 * s,        4 * c - 3,        0],
        [0,               0, -s, 0,            0,                c],
    ])
    # B = int_0^dt Phi(tau) d tau  @ [0; I]
    B = np.array([
        [1 - c,                 2 * (dt - s),                 0],
        [-2 * (dt - s),         4 * (1 - c) - 1.5 * dt * dt,  0],
        [0,                     0,                            1 - c],
        [s,                     2 * (1 - c),                  0],
        [-2 * (1 - c),          4 * s - 3 * dt,               0],
        [0,                     0,                            s],
    ])
    return A, B


def mean_motion(mu: float = MU_EARTH, a: float = A_REF) -> float:
    """rad/s of the circular reference orbit; defines the time unit of cw_matrices."""
    return float(np.sqrt(mu / a ** 3))


def cw_rendezvous(N: int = 1000, batch: int = 1, seed0: int = SEED0,
                  u_max: float = 0.2, thrust_norm: bool = False) -> Problem:
    """BASELINE.json configs[1..3]: orbit-transfer (rendezvous) QP, n = 6, m = 3.

    Horizon = one orbital period, dt = 2 pi / N in units of 1/mean-motion; length
    unit 1 km, so velocities are km * mean-motion and thrust accelerations are
    km * mean-motion^2: all variables are O(1).  Input box |u_i| <= u_max, states
    unbounded, q = 0.  x0 of instance i ~ U(box) from default_rng(seed0 + i).
    thrust_norm=True replaces the input box by the thrust-magnitude bound ||u_k||_2 <= u_max
    (the natural constraint of a single gimballed thruster)."""
    dt = 2.0 * np.pi / N
    A, B = cw_matrices(dt)
    Q = np.diag([1.0, 1.0, 1.0, 0.1, 0.1, 0.1]) * dt
    R = np.eye(3) * dt
    QN = np.diag([50.0, 50.0, 50.0, 20.0, 20.0, 20.0])
    box = np.array([0.3, 2.0, 1.0, 0.1, 0.1, 0.1])
    x0 = np.empty((batch, 6))
    for i in range(batch):
        rng = np.random.default_rng(seed0 + i)
        x0[i] = rng.uniform(-box, box)
    inf = np.inf
    if thrust_norm:
        return Problem(N=N, A=A, B=B, Q=Q, R=R, QN=QN, x0=x0, lo=np.full(9, -inf), hi=np.full(9, inf),
                       unorm=np.float64(u_max), name=f"cw_rendezvous_soc_N{N}_b{batch}")
    lo = np.array([-u_max] * 3 + [-inf] * 6)
    hi = np.array([u_max] * 3 + [inf] * 6)
    return Problem(N=N, A=A, B=B, Q=Q, R=R, QN=QN, x0=x0, lo=lo, hi=hi,
                   name=f"cw_rendezvous_N{N}_b{batch}")


def cw_formation(N: int = 1000, batch: int = 1, seed0: int = SEED0, u_max: float = 0.2) -> Problem:
    """BASELINE.json configs[4] shape (n = 12, m = 6): two spacecraft in Clohessy-Wiltshire
    relative motion about the same reference orbit, each with its own thrust box, coupled through
    the cost: the stage and terminal weights penalise each craft's state AND their separation, so
    Q and QN are full 12 x 12 matrices (block [[Q1+Qr, -Qr], [-Qr, Q2+Qr]])."""
    dt = 2.0 * np.pi / N
    A1, B1 = cw_matrices(dt)
    A = np.block([[A1, np.zeros((6, 6))], [np.zeros((6, 6)), A1]])
    B = np.block([[B1, np.zeros((6, 3))], [np.zeros((6, 3)), B1]])
    q1 = np.diag([1.0, 1.0, 1.0, 0.1, 0.1, 0.1]) * dt
    qr = np.diag([2.0, 2.0, 2.0, 0.2, 0.2, 0.2]) * dt
    Q = np.block([[q1 + qr, -qr], [-qr, q1 + qr]])
    t1 = np.diag([50.0, 50.0, 50.0, 20.0, 20.0, 20.0])
    tr = np.diag([25.0, 25.0, 25.0, 10.0, 10.0, 10.0])
    QN = np.block([[t1 + tr, -tr], [-tr, t1 + tr]])
    R = np.eye(6) * dt
    box = np.array([0.3, 2.0, 1.0, 0.1, 0.1, 0.1] * 2)
    x0 = np.empty((batch, 12))
    for i in range(batch):
        rng = np.random.default_rng(seed0 + i)
        x0[i] = rng.uniform(-box, box)
    inf = np.inf
    lo = np.array([-u_max] * 6 + [-inf] * 12)
    hi = np.array([u_max] * 6 + [inf] * 12)
    return Problem(N=N, A=A, B=B, Q=Q, R=R, QN=QN, x0=x0, lo=lo, hi=hi,
                   name=f"cw_formation_N{N}_b{batch}")


def random_ltv(N: int, n: int, m: int, batch: int, seed: int = SEED0,
               with_q: bool = True, state_bounds: bool = True, thrust_norm: bool = False) -> Problem:
    """Random stable-ish time-varying problem with full weights, per-stage
    bounds and a linear term: exercises every code path in the parity tests."""
    rng = np.random.default_rng(seed)
    A = np.eye(n)[None] + 0.15 * rng.standard_normal((N, n, n)) / np.sqrt(n)
    B = 0.5 * rng.standard_normal((N, n, m))
    def spd(k, lo_eig):
        M = rng.standard_normal((k, k))
        return M @ M.T / k + lo_eig * np.eye(k)
    Q, R, QN = spd(n, 0.1), spd(m, 0.05), spd(n, 1.0)
    x0 = rng.uniform(-1.0, 1.0, (batch, n))
    lo = np.empty((N, n + m))
    hi = np.empty((N, n + m))
    lo[:, :m] = -rng.uniform(0.1, 0.6, (N, m))
    hi[:, :m] = rng.uniform(0.1, 0.6, (N, m))
    if state_bounds:
        lo[:, m:] = -rng.uniform(0.8, 3.0, (N, n))
        hi[:, m:] = rng.uniform(0.8, 3.0, (N, n))
        lo[::3, m] = -np.inf
        hi[1::4, m + n - 1] = np.inf
    else:
        lo[:, m:] = -np.inf
        hi[:, m:] = np.inf
    q = 0.1 * rng.standard_normal((batch, N * (n + m))) if with_q else None
    unorm = None
    if thrust_norm:      # per-stage thrust-magnitude bounds on most stages, box on the rest
        unorm = rng.uniform(0.15, 0.7, N)
        unorm[::4] = np.inf
        soc = np.isfinite(unorm)
        lo[soc, :m] = -np.inf
        hi[soc, :m] = np.inf
    return Problem(N=N, A=A, B=B, Q=Q, R=R, QN=QN, x0=x0, lo=lo, hi=hi, q=q, unorm=unorm,
                   name=f"random_ltv_N{N}_n{n}_m{m}_b{batch}")


def random_instances(N: int, n: int, m: int, batch: int, seed: int = SEED0, with_q: bool = True,
                     instance_bounds: bool = True, thrust_norm: bool = False) -> Problem:
    """Per-instance time-varying dynamics (admm_problem.time_varying = 2): every QP is its own
    perturbation of a common random LTV plant, with its own per-stage box (instance_bounds) and linear term --
    the QP class a batched successive-convexification loop produces."""
    base = random_ltv(N, n, m, batch, seed, with_q=with_q, thrust_norm=thrust_norm)
    rng = np.random.default_rng(seed + 7)
    A = base.A[None] + 0.05 * rng.standard_normal((batch, N, n, n)) / np.sqrt(n)
    B = base.B[None] + 0.05 * rng.standard_normal((batch, N, n, m))
    lo, hi = base.lo, base.hi
    if instance_bounds:
        wid = rng.uniform(0.7, 1.3, (batch, N, n + m))
        lo = base.lo[None] * wid
        hi = base.hi[None] * rng.uniform(0.7, 1.3, (batch, N, n + m))
    return dataclasses.replace(base, A=A, B=B, lo=lo, hi=hi, name=f"random_instances_N{N}_n{n}_m{m}_b{batch}")


def cw_rendezvous_instances(N: int = 200, batch: int = 64, seed0: int = SEED0, u_max: float = 0.2,
                            spread: float = 0.05) -> Problem:
    """cw_rendezvous with PER-INSTANCE dynamics: QP i flies about its own reference orbit (mean motion scaled by
    1 + spread * U(-1, 1), so its Clohessy-Wiltshire matrices are those of a different time step) and has its own
    input box -- the shape of a Monte-Carlo / batched successive-convexification workload."""
    base = cw_rendezvous(N=N, batch=batch, seed0=seed0, u_max=u_max)
    dt = 2.0 * np.pi / N
    A = np.empty((batch, N, 6, 6))
    B = np.empty((batch, N, 6, 3))
    lo = np.empty((batch, N, 9))
    hi = np.empty((batch, N, 9))
    for i in range(batch):
        rng = np.random.default_rng(seed0 + 100003 * (i + 1))
        Ai, Bi = cw_matrices(dt * (1.0 + spread * rng.uniform(-1.0, 1.0)))
        A[i], B[i] = Ai[None], Bi[None]
        um = u_max * (1.0 + spread * rng.uniform(-1.0, 1.0))
        lo[i] = np.array([-um] * 3 + [-np.inf] * 6)[None]
        hi[i] = np.array([um] * 3 + [np.inf] * 6)[None]
    return dataclasses.replace(base, A=A, B=B, lo=lo, hi=hi, name=f"cw_rendezvous_instances_N{N}_b{batch}")


def cw_formation_instances(N: int = 200, batch: int = 64, seed0: int = SEED0, u_max: float = 0.2,
                           spread: float = 0.05) -> Problem:
    """cw_formation (n = 12, m = 6) with PER-INSTANCE dynamics: in QP i each of the two craft flies about its own reference
    orbit (mean motion scaled by 1 + spread * U(-1, 1)) and has its own input box -- the wide-shape twin of
    cw_rendezvous_instances."""
    base = cw_formation(N=N, batch=batch, seed0=seed0, u_max=u_max)
    dt = 2.0 * np.pi / N
    A = np.zeros((batch, N, 12, 12))
    B = np.zeros((batch, N, 12, 6))
    lo = np.empty((batch, N, 18))
    hi = np.empty((batch, N, 18))
    for i in range(batch):
        rng = np.random.default_rng(seed0 + 100003 * (i + 1))
        um = np.empty(6)
        for c in range(2):
            Ai, Bi = cw_matrices(dt * (1.0 + spread * rng.uniform(-1.0, 1.0)))
            A[i, :, 6 * c:6 * c + 6, 6 * c:6 * c + 6] = Ai[None]
            B[i, :, 6 * c:6 * c + 6, 3 * c:3 * c + 3] = Bi[None]
            um[3 * c:3 * c + 3] = u_max * (1.0 + spread * rng.uniform(-1.0, 1.0))
        lo[i] = np.concatenate([-um, [-np.inf] * 12])[None]
        hi[i] = np.concatenate([um, [np.inf] * 12])[None]
    return dataclasses.replace(base, A=A, B=B, lo=lo, hi=hi, name=f"cw_formation_instances_N{N}_b{batch}")

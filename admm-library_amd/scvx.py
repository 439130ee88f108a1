"""Successive-convexification outer loop over the batched ADMM solver.

The caller that generates the hot path's QPs in practice: a NONLINEAR trajectory problem is solved
by repeatedly linearising the dynamics about the current trajectory, solving the resulting
box-constrained optimal-control QP (libadmm_hip.so) for a correction, and re-propagating the
nonlinear dynamics.  Host-side NumPy; the QP solve is the only device work.  No reference
counterpart exists (README.md:1-2 only); the scheme is the standard trust-region successive
linearisation (Mao, Szmuk, Acikmese 2016) without virtual control: the reference trajectory is
always re-propagated through the nonlinear dynamics, so the linearised model has no defect term.

Model shipped here: exact nonlinear relative motion about a circular reference orbit in the
rotating (LVLH) frame, in the units of problems.cw_matrices (time 1/mean-motion, length 1 km) --
its linearisation at the origin is the Clohessy-Wiltshire system of the benchmark workload:

    x'' =  2 y' + x + rc - rc^3 (rc + x) / rd^3 + u_x
    y'' = -2 x' + y      - rc^3 y        / rd^3 + u_y          rd = |(rc + x, y, z)|,  rc = a / 1 km
    z'' =                - rc^3 z        / rd^3 + u_z

The QP of one outer iteration, in the correction (du, dx) about the reference (ub, xb):

    minimise   1/2 sum_k [(ub_k + du_k)' R (ub_k + du_k) + (xb_{k+1} + dx_{k+1})' Q_{k+1} (xb_{k+1} + dx_{k+1})]
    subject to dx_{k+1} = A_k dx_k + B_k du_k,  dx_0 = 0,
               max(u_lo - ub_k, -tr_u) <= du_k <= min(u_hi - ub_k, tr_u),   |dx| <= tr_x

i.e. the hot path's QP with time-varying (A_k, B_k), per-stage bounds and the linear term
q = (R ub_k, Q_{k+1} xb_{k+1}).  scvx() drives ONE trajectory (batch-shared dynamics, batch = 1);
scvx_batch() drives MANY at once -- a Monte-Carlo set of initial conditions, say: every trajectory
has its own linearisation, box and linear term, i.e. one QP batch with per-instance dynamics
(admm_problem.time_varying = 2, stage_bounds = 2) per outer iteration.
"""
from __future__ import annotations

import dataclasses
from typing import Callable, List, Optional, Tuple

import numpy as np

from .problems import A_REF, Problem

RC_KM = A_REF / 1000.0        # reference orbit radius in the length unit of cw_matrices (1 km)


def relative_motion_rhs(s: np.ndarray, u: np.ndarray, rc: float = RC_KM) -> np.ndarray:
    """Time derivative of s = (x, y, z, vx, vy, vz) (..., 6) under thrust acceleration u (..., 3)."""
    x, y, z, vx, vy, vz = (s[..., i] for i in range(6))
    rd3 = ((rc + x) ** 2 + y ** 2 + z ** 2) ** 1.5
    k = rc ** 3 / rd3
    ax = 2.0 * vy + x + rc - k * (rc + x) + u[..., 0]
    ay = -2.0 * vx + y - k * y + u[..., 1]
    az = -k * z + u[..., 2]
    return np.stack([vx, vy, vz, ax, ay, az], axis=-1)


def rk4_step(s: np.ndarray, u: np.ndarray, dt: float, substeps: int = 4, rc: float = RC_KM) -> np.ndarray:
    """Zero-order-hold propagation of one stage (classical RK4, `substeps` sub-intervals)."""
    h = dt / substeps
    for _ in range(substeps):
        k1 = relative_motion_rhs(s, u, rc)
        k2 = relative_motion_rhs(s + 0.5 * h * k1, u, rc)
        k3 = relative_motion_rhs(s + 0.5 * h * k2, u, rc)
        k4 = relative_motion_rhs(s + h * k3, u, rc)
        s = s + (h / 6.0) * (k1 + 2.0 * k2 + 2.0 * k3 + k4)
    return s


def rollout(x0: np.ndarray, u: np.ndarray, dt: float, step=rk4_step) -> np.ndarray:
    """States x_1..x_N (N, n) of the nonlinear dynamics from x0 under controls u (N, m); with a leading
    batch axis on both ((B, n), (B, N, m)) the B trajectories are propagated together -> (B, N, n)."""
    u = np.asarray(u, np.float64)
    s = np.asarray(x0, np.float64)
    N = u.shape[-2]
    xs = np.empty(u.shape[:-1] + (s.shape[-1],))
    for k in range(N):
        s = step(s, u[..., k, :], dt)
        xs[..., k, :] = s
    return xs


def linearise(xprev: np.ndarray, u: np.ndarray, dt: float, step=rk4_step, eps: float = 1e-6
              ) -> Tuple[np.ndarray, np.ndarray]:
    """A_k = dF/dx, B_k = dF/du of the stage map F at (xprev_k, u_k), k = 0..N-1, by central
    differences (vectorised over stages and perturbation directions)."""
    lead, n = xprev.shape[:-1], xprev.shape[-1]          # (N,) or (B, N)
    m = u.shape[-1]
    A = np.empty(lead + (n, n))
    B = np.empty(lead + (n, m))
    for j in range(n):
        d = np.zeros(n); d[j] = eps
        A[..., :, j] = (step(xprev + d, u, dt) - step(xprev - d, u, dt)) / (2.0 * eps)
    for j in range(m):
        d = np.zeros(m); d[j] = eps
        B[..., :, j] = (step(xprev, u + d, dt) - step(xprev, u - d, dt)) / (2.0 * eps)
    return A, B


def linearise_device(xprev: np.ndarray, u: np.ndarray, dt: float, device: str = "cuda:0", eps: float = 1e-6,
                     substeps: int = 4, rc: float = RC_KM) -> Tuple[np.ndarray, np.ndarray]:
    """linearise() of the shipped model (rk4_step of relative_motion_rhs) with the n + m central differences evaluated as ONE
    batch of torch tensors on `device`: the same formulas in the same order, fp64.  The NumPy version makes 2 (n + m) x 16
    passes over (B, N, 6) arrays -- 10 s per outer iteration of a 4096-trajectory batch, most of the wall time of
    examples/scvx_batch_rendezvous.py; this is host plumbing, not the solver."""
    import torch
    xp = torch.as_tensor(np.ascontiguousarray(xprev), dtype=torch.float64, device=device)
    up = torch.as_tensor(np.ascontiguousarray(u), dtype=torch.float64, device=device)
    n, m = xp.shape[-1], up.shape[-1]

    def rhs(s, uu):
        x, y, z, vx, vy, vz = (s[..., i] for i in range(6))
        rd3 = ((rc + x) ** 2 + y ** 2 + z ** 2) ** 1.5
        k = rc ** 3 / rd3
        ax = 2.0 * vy + x + rc - k * (rc + x) + uu[..., 0]
        ay = -2.0 * vx + y - k * y + uu[..., 1]
        az = -k * z + uu[..., 2]
        return torch.stack([vx, vy, vz, ax, ay, az], dim=-1)

    def step(s, uu):
        h = dt / substeps
        for _ in range(substeps):
            k1 = rhs(s, uu)
            k2 = rhs(s + 0.5 * h * k1, uu)
            k3 = rhs(s + 0.5 * h * k2, uu)
            k4 = rhs(s + h * k3, uu)
            s = s + (h / 6.0) * (k1 + 2.0 * k2 + 2.0 * k3 + k4)
        return s

    ex = eps * torch.eye(n, dtype=torch.float64, device=device).reshape((n,) + (1,) * (xp.dim() - 1) + (n,))
    eu = eps * torch.eye(m, dtype=torch.float64, device=device).reshape((m,) + (1,) * (up.dim() - 1) + (m,))
    A = (step(xp[None] + ex, up[None]) - step(xp[None] - ex, up[None])) / (2.0 * eps)        # (n, ..., n): [j, ..., i] = dF_i / dx_j
    B = (step(xp[None].expand((m,) + xp.shape), up[None] + eu) - step(xp[None].expand((m,) + xp.shape), up[None] - eu)) / (2.0 * eps)
    return torch.movedim(A, 0, -1).contiguous().cpu().numpy(), torch.movedim(B, 0, -1).contiguous().cpu().numpy()


@dataclasses.dataclass
class ScvxResult:
    u: np.ndarray                 # (N, m) controls
    x: np.ndarray                 # (N, n) states x_1..x_N of the NONLINEAR dynamics under u
    cost: float                   # nonlinear cost of (x, u)
    outer_iterations: int
    accepted: int
    converged: bool
    history: List[dict]           # per outer iteration: cost, predicted / actual decrease, ratio, trust radius, |du|, ADMM iterations


def trajectory_cost(x: np.ndarray, u: np.ndarray, Q, R, QN):
    """Nonlinear cost of one trajectory (float), or of B trajectories with a leading batch axis ((B,) array)."""
    c = 0.5 * np.einsum("...ki,ij,...kj->...", u, R, u)
    c = c + 0.5 * np.einsum("...ki,ij,...kj->...", x[..., :-1, :], Q, x[..., :-1, :])
    c = c + 0.5 * np.einsum("...i,ij,...j->...", x[..., -1, :], QN, x[..., -1, :])
    return float(c) if np.ndim(c) == 0 else c


def correction_qp(xb: np.ndarray, ub: np.ndarray, x0: np.ndarray, dt: float, Q, R, QN, u_lo, u_hi,
                  tr_u: float, tr_x: float, step=rk4_step) -> Problem:
    """The QP of one outer iteration (module docstring) as a hot-path Problem (batch = 1)."""
    N, n = xb.shape
    m = ub.shape[1]
    xprev = np.vstack([x0[None], xb[:-1]])
    A, B = linearise(xprev, ub, dt, step)
    q = np.empty((N, m + n))
    q[:, :m] = ub @ R.T
    q[:-1, m:] = xb[:-1] @ Q.T
    q[-1, m:] = QN @ xb[-1]
    lo = np.empty((N, m + n))
    hi = np.empty((N, m + n))
    lo[:, :m] = np.maximum(u_lo - ub, -tr_u)
    hi[:, :m] = np.minimum(u_hi - ub, tr_u)
    lo[:, m:] = -tr_x
    hi[:, m:] = tr_x
    return Problem(N=N, A=A, B=B, Q=np.asarray(Q, np.float64), R=np.asarray(R, np.float64),
                   QN=np.asarray(QN, np.float64), x0=np.zeros((1, n)), lo=lo, hi=hi, q=q.reshape(1, -1),
                   name=f"scvx_correction_N{N}")


def gpu_qp_solver(**options) -> Callable[[Problem], Tuple[np.ndarray, int]]:
    """QP solver for scvx(): the HIP solver (raises without a device: no CPU fallback).  One handle
    serves every outer iteration: admm_update_problem refactors in place (same N, n, m, batch), each
    solve starts cold (z = y = 0), like a fresh handle."""
    from .solver import Options, Solver
    state = {"solver": None, "shape": None}

    def solve(p: Problem):
        shape = (p.N, p.n, p.m, p.batch, p.q is not None, p.unorm is not None)
        if state["solver"] is None or state["shape"] != shape:
            if state["solver"] is not None:
                state["solver"].close()
            state["solver"], state["shape"] = Solver(p, Options(**options)), shape
        else:
            state["solver"].update_problem(p)
        s = state["solver"]
        zero = np.zeros((p.batch, p.L))
        info = s.solve(z0=zero, y0=zero)
        _, z, _ = s.get(w=False)
        return z, int(info.iters_run)      # z: the feasible (projected) iterate
    return solve


def scvx(x0: np.ndarray, N: int, dt: float, Q, R, QN, u_lo, u_hi,
         qp_solver: Optional[Callable[[Problem], Tuple[np.ndarray, int]]] = None,
         u_init: Optional[np.ndarray] = None, tr_u: float = 0.1, tr_x: float = 20.0,
         max_outer: int = 20, tol: float = 1e-6, rho_reject: float = 0.1, rho_expand: float = 0.7,
         step=rk4_step, qp_options: Optional[dict] = None) -> ScvxResult:
    """Trust-region successive convexification of

        minimise 1/2 sum_k [u_k' R u_k + x_{k+1}' Q_{k+1} x_{k+1}]   s.t.  x_{k+1} = F(x_k, u_k),  u_lo <= u_k <= u_hi.

    qp_solver(problem) -> (z of shape (1, L), ADMM iterations); default = the HIP solver.
    A step is accepted when actual / predicted cost decrease >= rho_reject (trust radii halve
    otherwise, double above rho_expand); stops when the accepted correction is below `tol`."""
    x0 = np.asarray(x0, np.float64)
    Q, R, QN = (np.asarray(a, np.float64) for a in (Q, R, QN))
    n, m = Q.shape[0], R.shape[0]
    u_lo = np.broadcast_to(np.asarray(u_lo, np.float64), (m,))
    u_hi = np.broadcast_to(np.asarray(u_hi, np.float64), (m,))
    if qp_solver is None:
        qp_solver = gpu_qp_solver(**(qp_options or dict(rho=0.5, eps_abs=1e-8, eps_rel=1e-8, max_iter=20000,
                                                        check_interval=25)))
    ub = np.zeros((N, m)) if u_init is None else np.clip(np.asarray(u_init, np.float64), u_lo, u_hi)
    xb = rollout(x0, ub, dt, step)
    J = trajectory_cost(xb, ub, Q, R, QN)
    hist: List[dict] = []
    accepted, converged = 0, False
    for it in range(1, max_outer + 1):
        p = correction_qp(xb, ub, x0, dt, Q, R, QN, u_lo, u_hi, tr_u, tr_x, step)
        z, admm_iters = qp_solver(p)
        d = np.asarray(z, np.float64).reshape(N, m + n)
        du, dx = d[:, :m], d[:, m:]
        # cost the QP's model predicts for (xb + dx, ub + du)
        J_lin = trajectory_cost(xb + dx, ub + du, Q, R, QN)
        u_new = np.clip(ub + du, u_lo, u_hi)
        x_new = rollout(x0, u_new, dt, step)
        J_new = trajectory_cost(x_new, u_new, Q, R, QN)
        predicted, actual = J - J_lin, J - J_new
        ratio = actual / predicted if predicted > 0 else -np.inf
        step_norm = float(np.abs(du).max())
        rec = dict(iteration=it, cost=J, cost_candidate=J_new, predicted=predicted, actual=actual, ratio=ratio,
                   tr_u=tr_u, tr_x=tr_x, du_max=step_norm, admm_iterations=admm_iters, accepted=False)
        if predicted <= tol * max(1.0, abs(J)):          # the model sees nothing left to gain
            hist.append(rec)
            converged = True
            break
        if ratio >= rho_reject:
            ub, xb, J = u_new, x_new, J_new
            accepted += 1
            rec["accepted"] = True
            if ratio >= rho_expand:
                tr_u, tr_x = 2.0 * tr_u, 2.0 * tr_x
        else:
            tr_u, tr_x = 0.5 * tr_u, 0.5 * tr_x
        hist.append(rec)
        if rec["accepted"] and step_norm <= tol:
            converged = True
            break
    return ScvxResult(u=ub, x=xb, cost=J, outer_iterations=len(hist), accepted=accepted, converged=converged,
                      history=hist)


def correction_qp_batch(xb: np.ndarray, ub: np.ndarray, x0: np.ndarray, dt: float, Q, R, QN, u_lo, u_hi,
                        tr_u: np.ndarray, tr_x: np.ndarray, step=rk4_step, device: Optional[str] = None) -> Problem:
    """The correction QPs of B trajectories as ONE batch with per-instance dynamics, box and linear term
    (xb (B, N, n), ub (B, N, m), x0 (B, n), trust radii (B,))."""
    Bn, N, n = xb.shape
    m = ub.shape[-1]
    xprev = np.concatenate([x0[:, None, :], xb[:, :-1, :]], axis=1)
    A, B = linearise_device(xprev, ub, dt, device) if (device is not None and step is rk4_step) else linearise(xprev, ub, dt, step)
    q = np.empty((Bn, N, m + n))
    q[..., :m] = ub @ R.T
    q[:, :-1, m:] = xb[:, :-1, :] @ Q.T
    q[:, -1, m:] = xb[:, -1, :] @ QN.T
    lo = np.empty((Bn, N, m + n))
    hi = np.empty((Bn, N, m + n))
    lo[..., :m] = np.maximum(u_lo - ub, -tr_u[:, None, None])
    hi[..., :m] = np.minimum(u_hi - ub, tr_u[:, None, None])
    lo[..., m:] = -tr_x[:, None, None]
    hi[..., m:] = tr_x[:, None, None]
    return Problem(N=N, A=A, B=B, Q=np.asarray(Q, np.float64), R=np.asarray(R, np.float64),
                   QN=np.asarray(QN, np.float64), x0=np.zeros((Bn, n)), lo=lo, hi=hi, q=q.reshape(Bn, -1),
                   name=f"scvx_correction_batch{Bn}_N{N}")


def scvx_batch(x0: np.ndarray, N: int, dt: float, Q, R, QN, u_lo, u_hi,
               qp_solver: Optional[Callable[[Problem], Tuple[np.ndarray, int]]] = None,
               tr_u: float = 0.1, tr_x: float = 20.0, max_outer: int = 20, tol: float = 1e-6,
               rho_reject: float = 0.1, rho_expand: float = 0.7, step=rk4_step,
               qp_options: Optional[dict] = None, linearise_on: Optional[str] = None) -> List[ScvxResult]:
    """scvx() for B initial conditions x0 (B, n) at once: the same trust-region loop per trajectory (own trust radii,
    own accept / reject decisions, own stop), but ONE batched QP solve per outer iteration -- per-instance dynamics,
    bounds and linear term (correction_qp_batch).  A trajectory that has stopped keeps its place in the batch with a
    zero-width box (its correction is then exactly zero) until the last one stops.
    linearise_on = "cuda:0": the central differences of the shipped model run as torch tensors on that device (linearise_device)."""
    x0 = np.atleast_2d(np.asarray(x0, np.float64))
    Bn = x0.shape[0]
    Q, R, QN = (np.asarray(a, np.float64) for a in (Q, R, QN))
    n, m = Q.shape[0], R.shape[0]
    u_lo = np.broadcast_to(np.asarray(u_lo, np.float64), (m,))
    u_hi = np.broadcast_to(np.asarray(u_hi, np.float64), (m,))
    if qp_solver is None:
        qp_solver = gpu_qp_solver(**(qp_options or dict(rho=0.5, eps_abs=1e-8, eps_rel=1e-8, max_iter=20000,
                                                        check_interval=25)))
    ub = np.zeros((Bn, N, m))
    xb = rollout(x0, ub, dt, step)
    J = trajectory_cost(xb, ub, Q, R, QN)
    tru = np.full(Bn, float(tr_u))
    trx = np.full(Bn, float(tr_x))
    active = np.ones(Bn, bool)
    converged = np.zeros(Bn, bool)
    accepted = np.zeros(Bn, int)
    hist: List[List[dict]] = [[] for _ in range(Bn)]
    for it in range(1, max_outer + 1):
        if not active.any():
            break
        p = correction_qp_batch(xb, ub, x0, dt, Q, R, QN, u_lo, u_hi, np.where(active, tru, 0.0), np.where(active, trx, 0.0), step,
                                device=linearise_on)
        z, admm_iters = qp_solver(p)
        d = np.asarray(z, np.float64).reshape(Bn, N, m + n)
        du, dx = d[..., :m], d[..., m:]
        J_lin = trajectory_cost(xb + dx, ub + du, Q, R, QN)
        u_new = np.clip(ub + du, u_lo, u_hi)
        x_new = rollout(x0, u_new, dt, step)
        J_new = trajectory_cost(x_new, u_new, Q, R, QN)
        predicted, actual = J - J_lin, J - J_new
        for b in np.flatnonzero(active):
            ratio = actual[b] / predicted[b] if predicted[b] > 0 else -np.inf
            step_norm = float(np.abs(du[b]).max())
            rec = dict(iteration=it, cost=float(J[b]), cost_candidate=float(J_new[b]), predicted=float(predicted[b]),
                       actual=float(actual[b]), ratio=float(ratio), tr_u=float(tru[b]), tr_x=float(trx[b]), du_max=step_norm,
                       admm_iterations=admm_iters, accepted=False)
            if predicted[b] <= tol * max(1.0, abs(J[b])):
                hist[b].append(rec)
                converged[b], active[b] = True, False
                continue
            if ratio >= rho_reject:
                ub[b], xb[b], J[b] = u_new[b], x_new[b], J_new[b]
                accepted[b] += 1
                rec["accepted"] = True
                if ratio >= rho_expand:
                    tru[b], trx[b] = 2.0 * tru[b], 2.0 * trx[b]
            else:
                tru[b], trx[b] = 0.5 * tru[b], 0.5 * trx[b]
            hist[b].append(rec)
            if rec["accepted"] and step_norm <= tol:
                converged[b], active[b] = True, False
    return [ScvxResult(u=ub[b], x=xb[b], cost=float(J[b]), outer_iterations=len(hist[b]), accepted=int(accepted[b]),
                       converged=bool(converged[b]), history=hist[b]) for b in range(Bn)]

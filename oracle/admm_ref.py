"""NumPy fp64 restatement of the batched optimal-control ADMM iteration.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product
path: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py`` may import it, and only as the checker.

PARITY UNPINNED.  The reference mount (/root/reference) holds README.md:1-2
("Implementation of Alternating Direction Method of Multipliers for
astrodynamics problems") and a LICENSE and nothing else: no MATLAB source, no
fixtures, no golden vectors.  There is therefore no reference file:line this
restatement can follow beyond README.md:1-2, which fixes only the subject.
The algorithm below is this repository's own specification,
standard scaled-form ADMM (Boyd et al. 2011, §3.1/§3.3) on the optimal-control
splitting of O'Donoghue, Stathopoulos & Boyd 2013.  It is pinned instead by
solver-independent checks in tests/test_oracle.py (dense KKT solve, bounded
least squares via SciPy, KKT optimality certificate).

Conventions
--------------------------
Stacked variable per QP:  w = (u_0, x_1, u_1, x_2, ..., u_{N-1}, x_N),
block k = (u_k, x_{k+1}), nb = m + n rows per block, L = N * nb.
Arrays are "QP-major": w has shape (batch, L) in C order, i.e. each QP's
vector is contiguous -- this is MATLAB's column-major L x batch.

minimise   1/2 sum_k [u_k' R u_k + x_{k+1}' Q_{k+1} x_{k+1}] + q' w
subject to x_{k+1} = A_k x_k + B_k u_k,  x_0 given,   lo <= w <= hi
(Q_{k+1} = Q for k+1 < N and QN for k+1 = N).
"""
from __future__ import annotations

import dataclasses
import numpy as np


@dataclasses.dataclass
class Factor:
    """Per-stage quantities of the (P + rho I)-regularised Riccati recursion."""
    K: np.ndarray      # (N, m, n) feedback gains
    Sinv: np.ndarray   # (N, m, m) inverse of R + rho I + B' P_{k+1} B
    A: np.ndarray      # (N, n, n)
    B: np.ndarray      # (N, n, m)
    P: np.ndarray      # (N + 1, n, n) cost-to-go Hessians (P[0] unused)


def expand_dynamics(A, B, N):
    """Accept LTI (n,n)/(n,m) or LTV (N,n,n)/(N,n,m) and return the LTV form."""
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    if A.ndim == 2:
        A = np.broadcast_to(A, (N,) + A.shape)
    if B.ndim == 2:
        B = np.broadcast_to(B, (N,) + B.shape)
    assert A.shape[0] == N and B.shape[0] == N
    return np.ascontiguousarray(A), np.ascontiguousarray(B)


def expand_bounds(lo, hi, N, nb):
    """Accept (nb,) or (N, nb) block bounds and return (L,) vectors."""
    lo = np.asarray(lo, dtype=np.float64)
    hi = np.asarray(hi, dtype=np.float64)
    if lo.ndim == 1:
        lo = np.tile(lo, N)
    if hi.ndim == 1:
        hi = np.tile(hi, N)
    return lo.reshape(N * nb), hi.reshape(N * nb)


def factor(A, B, Q, R, QN, rho, N) -> Factor:
    """Backward Riccati sweep on the x-update's KKT system
    [P + rho I, G'; G, 0].  One-off, host side."""
    A, B = expand_dynamics(A, B, N)
    n, m = B.shape[1], B.shape[2]
    Qr = np.asarray(Q, np.float64) + rho * np.eye(n)
    Rr = np.asarray(R, np.float64) + rho * np.eye(m)
    P = np.zeros((N + 1, n, n))
    K = np.zeros((N, m, n))
    Sinv = np.zeros((N, m, m))
    P[N] = np.asarray(QN, np.float64) + rho * np.eye(n)
    for k in range(N - 1, -1, -1):
        PB = P[k + 1] @ B[k]
        S = Rr + B[k].T @ PB
        S = 0.5 * (S + S.T)
        Sinv[k] = np.linalg.inv(S)
        Sinv[k] = 0.5 * (Sinv[k] + Sinv[k].T)
        K[k] = Sinv[k] @ (PB.T @ A[k])
        Pk = Qr + A[k].T @ P[k + 1] @ A[k] - K[k].T @ S @ K[k]
        P[k] = 0.5 * (Pk + Pk.T)
    return Factor(K=K, Sinv=Sinv, A=A, B=B, P=P)


def x_update(f: Factor, g, x0):
    """Solve  min 1/2 w'(P+rho I)w + g'w  s.t. dynamics, for a batch.

    g: (batch, L) linear term (= q - rho (z - y));  x0: (batch, n).
    Returns w (batch, L).  Backward substitution for the cost-to-go linear
    terms, then forward rollout:

        t = 0
        for k = N-1..0:  p = g^x_{k+1} + t;  h = B_k' p + g^u_k
                         d_k = Sinv_k h;     t = A_k' p - K_k' h
        x = x0
        for k = 0..N-1:  u = -K_k x - d_k;   x = A_k x + B_k u
    """
    N, m, n = f.K.shape
    nb = n + m
    batch = g.shape[0]
    gb = g.reshape(batch, N, nb)
    d = np.empty((N, batch, m))
    t = np.zeros((batch, n))
    for k in range(N - 1, -1, -1):
        p = gb[:, k, m:] + t
        h = p @ f.B[k] + gb[:, k, :m]
        d[k] = h @ f.Sinv[k].T
        t = p @ f.A[k] - h @ f.K[k]
    w = np.empty((batch, N, nb))
    x = np.array(x0, dtype=np.float64, copy=True)
    for k in range(N):
        u = -(x @ f.K[k].T) - d[k]
        x = x @ f.A[k].T + u @ f.B[k].T
        w[:, k, :m] = u
        w[:, k, m:] = x
    return w.reshape(batch, N * nb)


def expand_unorm(unorm, N):
    """None / scalar / (N,) -> (N,) with inf = off."""
    if unorm is None:
        return np.full(N, np.inf)
    return np.broadcast_to(np.asarray(unorm, np.float64), (N,)).copy()


def project(v, lo, hi, unorm=None, m=0):
    """Projection onto the constraint set: box on every row, except that
    the control rows of a stage with a finite thrust bound are scaled onto the ball
    ||u_k||_2 <= unorm_k.  v: (batch, L);  unorm: (N,) or None."""
    zn = np.minimum(np.maximum(v, lo), hi)
    if unorm is not None and np.isfinite(unorm).any():
        N = unorm.shape[0]
        nb = v.shape[1] // N
        vb = v.reshape(v.shape[0], N, nb)
        zb = zn.reshape(v.shape[0], N, nb)
        nrm = np.sqrt(np.sum(vb[:, :, :m] ** 2, axis=2))
        scale = np.where(nrm > unorm[None, :], unorm[None, :] / np.where(nrm > 0, nrm, 1.0), 1.0)
        soc = np.isfinite(unorm)
        zb[:, soc, :m] = vb[:, soc, :m] * scale[:, soc, None]
    return zn


def z_update(w, z, y, lo, hi, alpha=1.0, unorm=None, m=0):
    """Fused z-update + dual ascent.  Returns z+, y+."""
    wh = alpha * w + (1.0 - alpha) * z if alpha != 1.0 else w
    v = wh + y
    zn = project(v, lo, hi, unorm, m)
    yn = v - zn
    return zn, yn


def residuals(w, z_old, z_new, y_new, rho):
    """Per-QP residual norms: r, s, |w|, |z+|, rho |y+|."""
    r = np.sqrt(np.sum((w - z_new) ** 2, axis=1))
    s = rho * np.sqrt(np.sum((z_new - z_old) ** 2, axis=1))
    nw = np.sqrt(np.sum(w ** 2, axis=1))
    nz = np.sqrt(np.sum(z_new ** 2, axis=1))
    ny = rho * np.sqrt(np.sum(y_new ** 2, axis=1))
    return r, s, nw, nz, ny


def converged(r, s, nw, nz, ny, L, eps_abs, eps_rel):
    e_pri = np.sqrt(L) * eps_abs + eps_rel * np.maximum(nw, nz)
    e_dua = np.sqrt(L) * eps_abs + eps_rel * ny
    return (r <= e_pri) & (s <= e_dua)


@dataclasses.dataclass
class Result:
    w: np.ndarray
    z: np.ndarray
    y: np.ndarray
    iters_run: int
    iters: np.ndarray     # per-QP first checked iteration meeting the rule (max_iter if never)
    status: np.ndarray    # 1 converged, 0 not
    r: np.ndarray
    s: np.ndarray
    history: list
    rho: float = 0.0
    rho_updates: int = 0


def solve(A, B, Q, R, QN, x0, lo, hi, N, q=None, rho=1.0, alpha=1.0,
          eps_abs=1e-6, eps_rel=1e-6, max_iter=1000, check_interval=10,
          z0=None, y0=None, record=None, stop=True,
          adapt_interval=0, adapt_max=16, adapt_mu=10.0, adapt_tau=2.0, unorm=None) -> Result:
    """Run the batch loop.  All QPs iterate together until every one has met
    the stopping rule at a checked iteration (iterations that are multiples of
    ``check_interval``, and ``max_iter``), or ``max_iter`` is reached.
    ``record`` is an optional collection of iteration numbers whose (w, z, y)
    are kept in ``history``.  ``stop=False`` runs exactly max_iter iterations.
    ``adapt_interval > 0`` enables batch-level residual balancing of rho.
    """
    x0 = np.atleast_2d(np.asarray(x0, np.float64))
    batch = x0.shape[0]
    R_weight = R
    f = factor(A, B, Q, R_weight, QN, rho, N)
    n, m = f.B.shape[1], f.B.shape[2]
    nb = n + m
    L = N * nb
    lo, hi = expand_bounds(lo, hi, N, nb)
    un = None if unorm is None else expand_unorm(unorm, N)
    z = np.zeros((batch, L)) if z0 is None else np.array(z0, np.float64).reshape(batch, L)
    y = np.zeros((batch, L)) if y0 is None else np.array(y0, np.float64).reshape(batch, L)
    qq = None if q is None else np.asarray(q, np.float64).reshape(batch, L)
    w = np.zeros((batch, L))
    iters = np.full(batch, max_iter, np.int32)
    status = np.zeros(batch, np.int32)
    r = np.full(batch, np.inf)
    s = np.full(batch, np.inf)
    history = []
    it = 0
    n_updates = 0
    for it in range(1, max_iter + 1):
        g = -rho * (z - y)
        if qq is not None:
            g = g + qq
        w = x_update(f, g, x0)
        zn, yn = z_update(w, z, y, lo, hi, alpha, un, m)
        check = (it % check_interval == 0) or it == max_iter
        if check:
            r, s, nw, nz, ny = residuals(w, z, zn, yn, rho)
            ok = converged(r, s, nw, nz, ny, L, eps_abs, eps_rel)
            newly = ok & (status == 0)
            iters[newly] = it
            status[newly] = 1
        z, y = zn, yn
        if record is not None and it in record:
            history.append((it, w.copy(), z.copy(), y.copy()))
        if stop and check and status.all():
            break
        if (check and adapt_interval > 0 and it % adapt_interval == 0 and n_updates < adapt_max
                and it < max_iter):
            Rsum = Ssum = 0.0
            for b in range(batch):              # same summation order as the C oracle and the HIP host code
                if not status[b]:
                    Rsum += r[b] * r[b]
                    Ssum += s[b] * s[b]
            rho_new = rho
            if Rsum > adapt_mu ** 2 * Ssum:
                rho_new = rho * adapt_tau
            elif Ssum > adapt_mu ** 2 * Rsum:
                rho_new = rho / adapt_tau
            if rho_new != rho:
                y = y * (rho / rho_new)
                f = factor(A, B, Q, R_weight, QN, rho_new, N)
                rho = rho_new
                n_updates += 1
    return Result(w=w, z=z, y=y, iters_run=it, iters=iters, status=status,
                  r=r, s=s, history=history, rho=rho, rho_updates=n_updates)


# ---------------------------------------------------------------------------
# Solver-independent helpers used by the tests (T1-T3 of SURVEY.md §4).
# ---------------------------------------------------------------------------

def dense_qp(A, B, Q, R, QN, x0, N, q=None):
    """Dense (P, q, G, b) of ONE QP in the stacked ordering; small N only."""
    A, B = expand_dynamics(A, B, N)
    n, m = B.shape[1], B.shape[2]
    nb = n + m
    L = N * nb
    P = np.zeros((L, L))
    G = np.zeros((N * n, L))
    b = np.zeros(N * n)
    for k in range(N):
        o = k * nb
        P[o:o + m, o:o + m] = R
        P[o + m:o + nb, o + m:o + nb] = QN if k == N - 1 else Q
        # x_{k+1} - A_k x_k - B_k u_k = 0
        G[k * n:(k + 1) * n, o + m:o + nb] = np.eye(n)
        G[k * n:(k + 1) * n, o:o + m] = -B[k]
        if k == 0:
            b[:n] = A[0] @ np.asarray(x0, np.float64)
        else:
            G[k * n:(k + 1) * n, o - n:o] = -A[k]
    qv = np.zeros(L) if q is None else np.asarray(q, np.float64).reshape(L)
    return P, qv, G, b


def kkt_x_update_dense(A, B, Q, R, QN, x0, N, g, rho):
    """x-update by a dense solve of [P+rho I, G'; G, 0][w; nu] = [-g; b]."""
    P, _, G, b = dense_qp(A, B, Q, R, QN, x0, N)
    L = P.shape[0]
    nc = G.shape[0]
    Kmat = np.block([[P + rho * np.eye(L), G.T], [G, np.zeros((nc, nc))]])
    rhs = np.concatenate([-np.asarray(g, np.float64).reshape(L), b])
    sol = np.linalg.solve(Kmat, rhs)
    return sol[:L]

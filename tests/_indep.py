"""Oracle-free checks at BASELINE's sizes.

Nothing here imports oracle/ or the product: only NumPy / SciPy and the problem arrays.  The routines are
written from the QP's definition, not from the Riccati form either solver uses:

  banded_x_update      the x-update  argmin 1/2 w'(P + rho I)w + g'w  s.t. Gw = b  as ONE banded LU solve
                       (scipy.linalg.solve_banded) of the KKT system in a stage-interleaved ordering;
  kkt_certificate_batch  the optimality conditions of the original QP at (z, lambda = rho y) for every QP of a
                       batch at once: dynamics defect, box feasibility, stationarity with the least-squares
                       multiplier of the dynamics (banded Cholesky of G G'), complementarity;
  condensed_bvls       configs[0] condensed to a bounded least-squares problem in u for scipy.optimize.lsq_linear.

Batch-shared dynamics (A: (n, n) or (N, n, n)); kkt_certificate_instances applies the certificate QP by QP to per-instance
dynamics (A: (batch, N, n, n), own box per QP).  Box constraints (no thrust-magnitude bound).
"""
import numpy as np
import scipy.sparse as sp
from scipy.linalg import cholesky_banded, cho_solve_banded, solve_banded
from scipy.optimize import lsq_linear


def stage_dynamics(p):
    A = np.broadcast_to(p.A, (p.N,) + p.A.shape[-2:])
    B = np.broadcast_to(p.B, (p.N,) + p.B.shape[-2:])
    return A, B


def stage_bounds(p):
    return np.broadcast_to(p.lo, (p.N, p.nb)), np.broadcast_to(p.hi, (p.N, p.nb))


def dynamics_matrix(p):
    """Sparse G (N n x L) with G w = b(x0): row block k is  x_{k+1} - A_k x_k - B_k u_k  (x_0 moved to the right)."""
    N, n, m, nb = p.N, p.n, p.m, p.nb
    A, B = stage_dynamics(p)
    rows, cols, vals = [], [], []
    ii, jn = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    iu, ju = np.meshgrid(np.arange(n), np.arange(m), indexing="ij")
    for k in range(N):
        r0, c0 = k * n, k * nb
        rows.append(r0 + np.arange(n)); cols.append(c0 + m + np.arange(n)); vals.append(np.ones(n))
        rows.append((r0 + iu).ravel()); cols.append((c0 + ju).ravel()); vals.append(-B[k].ravel())
        if k > 0:
            rows.append((r0 + ii).ravel()); cols.append((c0 - nb + m + jn).ravel()); vals.append(-A[k].ravel())
    G = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(N * n, N * nb))
    return G


def dynamics_rhs(p):
    """b of G w = b for every QP: (batch, N n), non-zero in the first block only (A_0 x0)."""
    A, _ = stage_dynamics(p)
    b = np.zeros((p.batch, p.N * p.n))
    b[:, :p.n] = p.x0 @ A[0].T
    return b


def hessian_diag_blocks(p):
    """(N, nb, nb) diagonal blocks of P = blkdiag(R, Q, R, Q, ..., R, QN)."""
    N, n, m, nb = p.N, p.n, p.m, p.nb
    blk = np.zeros((N, nb, nb))
    blk[:, :m, :m] = p.R
    blk[:, m:, m:] = p.Q
    blk[N - 1, m:, m:] = p.QN
    return blk


def _to_banded(M):
    """scipy banded storage (l, u, ab) of a sparse square matrix."""
    M = M.tocoo()
    l = int((M.row - M.col).max())
    u = int((M.col - M.row).max())
    ab = np.zeros((l + u + 1, M.shape[0]))
    np.add.at(ab, (u + M.row - M.col, M.col), M.data)
    return l, u, ab


def banded_x_update(p, g, rho):
    """w+ of every QP (batch, L) from ONE banded LU of the KKT matrix [P + rho I, G'; G, 0].

    Ordering per stage k: (u_k, nu_{k+1}, x_{k+1}) -- the multiplier of stage k's dynamics row sits between the
    variables it couples, so the matrix is banded with half-width <= m + 2n + n."""
    N, n, m, nb = p.N, p.n, p.m, p.nb
    L, nc = N * nb, N * n
    G = dynamics_matrix(p)
    Pd = hessian_diag_blocks(p)
    Pr = sp.block_diag([Pd[k] for k in range(N)], format="csr") + rho * sp.identity(L, format="csr")
    K = sp.bmat([[Pr, G.T], [G, None]], format="csr")
    # permutation: new index of (u_k) = k (nb + n) + [0, m); nu_{k+1} = ... + m + [0, n); x_{k+1} = ... + m + n + [0, n)
    st = nb + n
    perm = np.empty(L + nc, np.int64)
    k = np.arange(N)
    for j in range(m):
        perm[k * nb + j] = k * st + j
    for j in range(n):
        perm[k * nb + m + j] = k * st + m + n + j
        perm[L + k * n + j] = k * st + m + j
    inv = np.empty_like(perm)
    inv[perm] = np.arange(L + nc)
    Kp = K[inv][:, inv]
    l, u, ab = _to_banded(Kp)
    assert max(l, u) <= 2 * st, (l, u)
    rhs = np.concatenate([-np.asarray(g, np.float64), dynamics_rhs(p)], axis=1)      # (batch, L + nc)
    sol = solve_banded((l, u), ab, rhs[:, inv].T, check_finite=True)                # columns = QPs
    full = sol.T[:, perm]
    w = full[:, :L]
    resid = (K @ full.T).T - rhs                                                    # backward error of the solve itself
    return w, float(np.abs(resid).max())


def kkt_certificate_batch(p, z, y, rho):
    """Optimality conditions of the ORIGINAL QP at z with box multiplier lambda = rho y, per QP (arrays of length batch):
       feas_dyn  max |x_{k+1} - A_k x_k - B_k u_k|            (stage recursion, written out -- not through G)
       feas_box  max violation of lo <= z <= hi
       stat      max |P z + q + lambda + G' nu|  with nu the least-squares multiplier  (G G') nu = -G (P z + q + lambda)
       comp      complementarity: |lambda| where z is strictly inside, the wrong-signed part of lambda on a bound
    """
    N, n, m, nb, Bt = p.N, p.n, p.m, p.nb, p.batch
    A, Bm = stage_dynamics(p)
    lo, hi = stage_bounds(p)
    Z = z.reshape(Bt, N, nb)
    lam = rho * y.reshape(Bt, N, nb)
    u, x = Z[:, :, :m], Z[:, :, m:]
    xprev = np.concatenate([p.x0[:, None, :], x[:, :-1]], axis=1)
    defect = x - np.einsum("kij,bkj->bki", A, xprev) - np.einsum("kij,bkj->bki", Bm, u)
    feas_dyn = np.abs(defect).reshape(Bt, -1).max(axis=1)
    feas_box = np.maximum(np.maximum(lo - Z, 0.0), np.maximum(Z - hi, 0.0)).reshape(Bt, -1).max(axis=1)
    grad = np.einsum("kij,bkj->bki", hessian_diag_blocks(p), Z) + lam
    if p.q is not None:
        grad = grad + p.q.reshape(Bt, N, nb)
    grad = grad.reshape(Bt, N * nb)
    G = dynamics_matrix(p)
    GGt = (G @ G.T).tocoo()
    up = GGt.row <= GGt.col
    ub = int((GGt.col - GGt.row).max())
    ab = np.zeros((ub + 1, N * n))
    ab[ub + GGt.row[up] - GGt.col[up], GGt.col[up]] = GGt.data[up]
    c = cholesky_banded(ab, lower=False)
    nu = cho_solve_banded((c, False), -(G @ grad.T))                 # (N n, batch)
    stat = np.abs(grad + (G.T @ nu).T).max(axis=1)
    lamf = lam.reshape(Bt, -1)
    Zf = Z.reshape(Bt, -1)
    lof = np.broadcast_to(lo.reshape(-1), Zf.shape)
    hif = np.broadcast_to(hi.reshape(-1), Zf.shape)
    at_lo = np.isclose(Zf, lof, rtol=0, atol=1e-9)
    at_hi = np.isclose(Zf, hif, rtol=0, atol=1e-9)
    viol = np.where(at_lo, np.maximum(lamf, 0.0), np.where(at_hi, np.maximum(-lamf, 0.0), np.abs(lamf)))
    comp = viol.max(axis=1)
    return feas_dyn, feas_box, stat, comp, int((at_lo | at_hi).sum())


def kkt_certificate_instances(p, z, y, rho):
    """kkt_certificate_batch for PER-INSTANCE dynamics / bounds (admm_problem.time_varying = 2): every QP with its own A_k, B_k
    (and box), one QP at a time.  rho: scalar or one value per QP (the per-QP adaptive rule)."""
    import dataclasses
    rho = np.broadcast_to(np.asarray(rho, dtype=float), (p.batch,))
    per_box = np.ndim(p.lo) == 3
    out = [[], [], [], []]
    n_active = 0
    for b in range(p.batch):
        one = dataclasses.replace(p, A=p.A[b], B=p.B[b], x0=p.x0[b:b + 1], q=None if p.q is None else p.q[b:b + 1],
                                  lo=p.lo[b] if per_box else p.lo, hi=p.hi[b] if per_box else p.hi)
        r = kkt_certificate_batch(one, z[b:b + 1], y[b:b + 1], float(rho[b]))
        for acc, v in zip(out, r[:4]):
            acc.append(float(v[0]))
        n_active += r[4]
    return tuple(np.asarray(a) for a in out) + (n_active,)


def condense(p):
    """x = Sx x0 + Su u over the horizon (x_1..x_N stacked)."""
    A, B = stage_dynamics(p)
    n, m, N = p.n, p.m, p.N
    Sx = np.zeros((N * n, n))
    Su = np.zeros((N * n, N * m))
    Phi = np.eye(n)
    for k in range(N):
        if k > 0:
            Su[k * n:(k + 1) * n] = A[k] @ Su[(k - 1) * n:k * n]
        Su[k * n:(k + 1) * n, k * m:(k + 1) * m] = B[k]
        Phi = A[k] @ Phi
        Sx[k * n:(k + 1) * n] = Phi
    return Sx, Su


def condensed_bvls(p, b):
    """u* (N m) of QP b of a control-box-only problem without q by SciPy's bounded least squares."""
    n, m, N = p.n, p.m, p.N
    lo, hi = stage_bounds(p)
    assert not np.isfinite(lo[:, m:]).any() and not np.isfinite(hi[:, m:]).any() and p.q is None
    Sx, Su = condense(p)
    Qbar = np.kron(np.eye(N), p.Q)
    Qbar[-n:, -n:] = p.QN
    H = np.kron(np.eye(N), p.R) + Su.T @ Qbar @ Su
    f = Su.T @ Qbar @ Sx @ p.x0[b]
    Cu = np.linalg.cholesky(H).T
    c = -np.linalg.solve(Cu.T, f)
    sol = lsq_linear(Cu, c, bounds=(lo[:, :m].reshape(-1), hi[:, :m].reshape(-1)), method="bvls", tol=1e-14, max_iter=4000)
    return sol.x

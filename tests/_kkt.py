"""Solver-independent optimality check of one QP (T3 of SURVEY.md §4), shared by the oracle tests and the
GPU tests of the reduced-precision mode."""
import numpy as np

import admm_ref as ar


def kkt_certificate(p, b, z, y, rho):
    """T3: separately written optimality check of one QP at (z, lambda = rho y)."""
    P, q, G, bvec = ar.dense_qp(p.A, p.B, p.Q, p.R, p.QN, p.x0[b], p.N, None if p.q is None else p.q[b])
    lo, hi = ar.expand_bounds(p.lo, p.hi, p.N, p.nb)
    lam = rho * y
    feas_dyn = np.abs(G @ z - bvec).max()
    feas_box = max(np.maximum(lo - z, 0).max(), np.maximum(z - hi, 0).max())
    grad = P @ z + q + lam
    nu, *_ = np.linalg.lstsq(G.T, -grad, rcond=None)
    stat = np.abs(grad + G.T @ nu).max()
    at_lo = np.isclose(z, lo, atol=1e-9)
    at_hi = np.isclose(z, hi, atol=1e-9)
    interior = ~(at_lo | at_hi)
    comp = 0.0
    if interior.any():
        comp = max(comp, np.abs(lam[interior]).max())
    if at_lo.any():
        comp = max(comp, np.maximum(lam[at_lo], 0).max())     # lambda <= 0 at a lower bound
    if at_hi.any():
        comp = max(comp, np.maximum(-lam[at_hi], 0).max())    # lambda >= 0 at an upper bound
    return feas_dyn, feas_box, stat, comp

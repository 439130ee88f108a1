"""Shared scenario of the successive-convexification tests: a 150 km along-track, 30 km cross-track
rendezvous over one orbit, where the Clohessy-Wiltshire model alone misses by tens of km."""
import numpy as np

import oracle_c

N = 80
DT = 2 * np.pi / N
Q = np.diag([1, 1, 1, .1, .1, .1]) * DT * 1e-3
R = np.eye(3) * DT * 0.05
QN = np.diag([50., 50, 50, 20, 20, 20])
X0 = np.array([10.0, 150.0, 30.0, 0.0, -15.0, 0.0])
U_MAX = 3.0
QP = dict(rho=0.5, eps_abs=1e-8, eps_rel=1e-8, max_iter=20000, check_interval=25)
SCVX = dict(tr_u=1.0, tr_x=100.0, max_outer=25, tol=1e-7)


def oracle_qp_solver(**kw):
    """The CPU oracle as scvx()'s QP solver (test infrastructure)."""
    def solve(p):
        r = oracle_c.solve(p, **kw)
        return r["z"], r["iters_run"]
    return solve

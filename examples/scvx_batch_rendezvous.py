#!/usr/bin/env python3
"""A Monte-Carlo set of nonlinear rendezvous problems by BATCHED successive convexification (needs an MI355X).

B chasers start from scattered relative states; every outer iteration linearises the exact relative-motion dynamics about
each chaser's own trajectory and solves ALL correction QPs as one batch with per-instance dynamics, box and linear term
(admm_problem.time_varying = 2: device factorisation, per-QP segments in time).

    python examples/scvx_batch_rendezvous.py [B=64] [N=200]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import admm_library_amd as pkg                        # noqa: E402
from admm_library_amd import scvx as sc               # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dt = 2 * np.pi / N
Q = np.diag([1, 1, 1, .1, .1, .1]) * dt * 1e-3
R = np.eye(3) * dt * 0.05
QN = np.diag([50., 50, 50, 20, 20, 20])
rng = np.random.default_rng(11)
x0 = np.array([10.0, 150.0, 30.0, 0.0, -15.0, 0.0]) * (1.0 + 0.05 * rng.standard_normal((B, 6)))

solves = []


def timed_solver():
    # per-QP residual balancing on the device; SCVX_ADAPT=0 for fixed rho
    every = int(os.environ.get("SCVX_ADAPT", "100"))
    adapt = dict(adapt_interval=every, adapt_mu=5.0) if every > 0 else {}
    inner = sc.gpu_qp_solver(rho=0.5, eps_abs=1e-8, eps_rel=1e-8, max_iter=20000, check_interval=25, **adapt)

    def solve(p):
        t = time.perf_counter()
        z, iters = inner(p)
        solves.append((time.perf_counter() - t, iters))
        return z, iters
    return solve


t0 = time.perf_counter()
res = sc.scvx_batch(x0, N, dt, Q, R, QN, -3.0, 3.0, qp_solver=timed_solver(), tr_u=1.0, tr_x=100.0, max_outer=25, tol=1e-7,
                    linearise_on=None if os.environ.get("SCVX_HOST_LINEARISE") else "cuda:0")
wall = time.perf_counter() - t0
outer = max(r.outer_iterations for r in res)
print(f"{B} trajectories, N = {N}: {sum(r.converged for r in res)} converged in at most {outer} outer iterations, {wall:.2f} s wall")
print(f"  QP batches: {len(solves)} solves, {sum(s[1] for s in solves)} ADMM batch-iterations, "
      f"{sum(s[0] for s in solves):.2f} s in the solver calls (upload + device refactor + iterations + read-out)")
print(f"  first / last QP batch: {solves[0][1]} iterations in {solves[0][0] * 1e3:.0f} ms, {solves[-1][1]} in {solves[-1][0] * 1e3:.0f} ms")
err = np.array([np.linalg.norm(r.x[-1, :3]) for r in res])
print(f"  terminal position error [km]: median {np.median(err):.3f}, max {err.max():.3f}; cost median {np.median([r.cost for r in res]):.3f}")

#!/usr/bin/env python3
"""Solve a batch of Clohessy-Wiltshire rendezvous QPs on the GPU and print a summary.

    python examples/solve_cw_rendezvous.py [batch] [horizon]

Needs an MI355X and the built library (python -c "import __graft_entry__ as g; g.build()")."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import admm_library_amd as pkg   # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
problem = pkg.cw_rendezvous(N=N, batch=batch)
options = pkg.Options(rho=0.05, eps_abs=1e-6, eps_rel=1e-6, max_iter=4000, check_interval=10,
                      adapt_interval=50)                       # batch-level adaptive rho
w, z, y, info = pkg.admm_solve(problem, options)
u = z.reshape(batch, N, 9)[:, :, :3]
xN = z.reshape(batch, N, 9)[:, -1, 3:]
print(f"{batch} QPs, horizon {N}: {info.iters_run} iterations in {info.solve_ms:.1f} ms, "
      f"{info.n_converged}/{batch} converged, rho {options.rho} -> {info.rho} ({info.rho_updates} updates)")
print(f"per-QP iterations: median {np.median(info.iters):.0f}, max {info.iters.max()}")
print(f"controls on their bound: {100 * (np.abs(np.abs(u) - 0.2) < 1e-9).mean():.1f} %,  "
      f"terminal |state| max {np.abs(xN).max():.3e}")

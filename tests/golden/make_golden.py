#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/.

The reference ships no golden vectors (README.md:1-2 + LICENSE only, SURVEY.md
§0/§8c), so these are produced by THIS repository's NumPy oracle
(oracle/admm_ref.py) -- "parity unpinned" with respect to the reference; what
they pin is that the C oracle and the HIP path keep producing the iterates the
T1-T3-validated NumPy restatement produced when the fixture was cut.

Each .npz holds the problem inputs explicitly (so the tests do not depend on
the generator functions staying unchanged), the options, and (w, z, y) after
the listed iteration counts, plus the result of a full solve.
golden_config1.mat is the same data as golden_config1.npz for MATLAB users.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
from scipy.io import savemat

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import admm_library_amd as pkg   # noqa: E402
import admm_ref as ar            # noqa: E402


def cut(name, p, rho, alpha, iters, solve_kw):
    d = dict(N=p.N, A=p.A, B=p.B, Q=p.Q, R=p.R, QN=p.QN, x0=p.x0, lo=p.lo, hi=p.hi,
             rho=rho, alpha=alpha, iters=np.array(iters, np.int32))
    if p.q is not None:
        d["q"] = p.q
    if p.unorm is not None:
        d["unorm"] = np.asarray(p.unorm, np.float64)
    res = ar.solve(p.A, p.B, p.Q, p.R, p.QN, p.x0, p.lo, p.hi, p.N, q=p.q, rho=rho, alpha=alpha,
                   max_iter=max(iters), stop=False, record=set(iters), unorm=p.unorm)
    for it, w, z, y in res.history:
        d[f"w_{it}"], d[f"z_{it}"], d[f"y_{it}"] = w, z, y
    full = ar.solve(p.A, p.B, p.Q, p.R, p.QN, p.x0, p.lo, p.hi, p.N, q=p.q, rho=rho, alpha=alpha, unorm=p.unorm,
                    **solve_kw)
    d.update(solve_eps_abs=solve_kw["eps_abs"], solve_eps_rel=solve_kw["eps_rel"],
             solve_max_iter=solve_kw["max_iter"], solve_check_interval=solve_kw["check_interval"],
             solve_adapt_interval=solve_kw.get("adapt_interval", 0), solve_rho_final=full.rho,
             solve_iters_run=full.iters_run, solve_iters=full.iters, solve_status=full.status,
             solve_z=full.z, solve_y=full.y, solve_w=full.w)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    return d


def main():
    kw = dict(eps_abs=1e-7, eps_rel=1e-7, max_iter=3000, check_interval=10)
    d1 = cut("golden_config1", pkg.double_integrator(N=50, batch=3), 1.0, 1.0, [1, 2, 10, 100], kw)
    savemat(os.path.join(HERE, "golden_config1.mat"), {k: v for k, v in d1.items()}, do_compression=True)
    cut("golden_cw_small", pkg.cw_rendezvous(N=60, batch=4), 0.05, 1.0, [1, 2, 10, 60],
        dict(eps_abs=1e-6, eps_rel=1e-6, max_iter=2000, check_interval=10))
    cut("golden_ltv_relaxed", pkg.random_ltv(N=16, n=4, m=2, batch=3, seed=77), 0.4, 1.5, [1, 5, 30],
        dict(eps_abs=1e-6, eps_rel=1e-6, max_iter=2000, check_interval=5))
    # thrust-magnitude bound + adaptive rho
    cut("golden_cw_soc_adaptive", pkg.cw_rendezvous(N=60, batch=4, thrust_norm=True), 0.05, 1.0, [1, 2, 10, 60],
        dict(eps_abs=1e-6, eps_rel=1e-6, max_iter=3000, check_interval=10, adapt_interval=20))


if __name__ == "__main__":
    main()

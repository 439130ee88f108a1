#!/usr/bin/env python3
"""Wall time of admm_solve with the adaptive-rho rule, and of one rho change (host refactor + upload), on the GPU box.

    python tools/solve_time.py [--workload cw_rendezvous|cw_formation] [--precision fp64|mixed|fp64_mfma] [--batch 4096]

Run it with ADMM_NO_SPECULATE=1 for the synchronous refactors, and with ADMM_FACTOR_THREADS=1/4/16 for the thread
scaling of the host factorisation (both are read once per process).  ADMM_SPEC_DEBUG=1 prints how many rho changes
were served by a background factor."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import admm_library_amd as pkg  # noqa: E402

PM = {"fp64": 0, "mixed": 1, "fp64_mfma": 2}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cw_rendezvous")
    ap.add_argument("--precision", default="fp64")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--horizon", type=int, default=1000)
    ap.add_argument("--alpha", type=float, default=1.6)
    ap.add_argument("--repeats", type=int, default=3)
    a = ap.parse_args()
    p = getattr(pkg, a.workload)(N=a.horizon, batch=a.batch)
    tag = (f"{a.workload} {a.precision} batch {a.batch} threads {os.environ.get('ADMM_FACTOR_THREADS', 'auto')} "
           f"speculate {'off' if os.environ.get('ADMM_NO_SPECULATE') else 'on'}")
    # one rho change with no adaptive options: factorise + upload, nothing kept or speculated
    with pkg.Solver(p, pkg.Options(rho=0.05, precision_mode=PM[a.precision])) as s:
        s.iterate(4)
        ts = []
        for r in (0.1, 0.05, 0.2, 0.05, 0.1):
            s.sync()
            t = time.perf_counter()
            s.set_rho(r)
            s.sync()
            ts.append((time.perf_counter() - t) * 1e3)
            s.iterate(2)
    print(f"{tag}: set_rho {np.median(ts):.2f} ms (median of {len(ts)}: {' '.join('%.1f' % t for t in ts)})")
    opts = dict(rho=0.05, eps_abs=1e-6, eps_rel=1e-6, max_iter=4000, check_interval=10, adapt_interval=50, adapt_mu=10.0,
                adapt_tau=2.0, alpha=a.alpha, precision_mode=PM[a.precision])
    with pkg.Solver(p, pkg.Options(**opts)) as s:
        for rep in range(a.repeats):
            s.set_rho(0.05)
            zero = np.zeros((p.batch, p.L))
            info = s.solve(z0=zero, y0=zero)
            print(f"{tag}: solve {info.solve_ms:.1f} ms, {info.iters_run} iterations, {info.rho_updates} rho changes, "
                  f"{info.n_converged} converged, rho {info.rho}", flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Exploration / A-B tool for the MFMA forms: iterate errors vs the C oracle and iteration rates
of the three precision modes.   python tools/mfma_check.py [--full]"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import admm_library_amd as pkg
import oracle_c as oc

MODES = {"fp64": 0, "mixed": 1, "fp64_mfma": 2}
cases = [("cw_formation N=120 b=66", lambda: pkg.cw_formation(N=120, batch=66), 0.05),
         ("cw_rendezvous N=200 b=70", lambda: pkg.cw_rendezvous(N=200, batch=70), 0.05),
         ("random_ltv (10,4) N=30 b=5", lambda: pkg.random_ltv(N=30, n=10, m=4, batch=5, seed=3, with_q=False), 0.3)]
for name, make, rho in cases:
    p = make()
    for mode, pm in MODES.items():
        errs = []
        try:
            with pkg.Solver(p, pkg.Options(rho=rho, precision_mode=pm)) as s:
                done = 0
                for upto in (1, 2, 3, 10, 40):
                    s.iterate(upto - done); done = upto
                    w, z, y = s.get()
                    ref = oc.solve(p, rho=rho, max_iter=upto, stop=False)
                    errs.append(max(np.abs(a - ref[k]).max() / max(1, np.abs(ref[k]).max()) for a, k in ((w, "w"), (z, "z"), (y, "y"))))
            print(f"{name:32s} {mode:10s} err@1,2,3,10,40: " + " ".join(f"{e:.2e}" for e in errs), flush=True)
        except pkg.AdmmError as e:
            print(f"{name:32s} {mode:10s} {e}", flush=True)
# solve to eps
p = pkg.cw_formation(N=120, batch=66)
kw = dict(rho=0.05, eps_abs=1e-6, eps_rel=1e-6, max_iter=4000, check_interval=10)
ref = oc.solve(p, **kw)
for mode, pm in MODES.items():
    with pkg.Solver(p, pkg.Options(precision_mode=pm, **kw)) as s:
        info = s.solve(); w, z, y = s.get()
    print(f"solve {mode:10s} iters {info.iters_run} (oracle {ref['iters_run']}) mixed_iters {info.mixed_iters} conv {info.n_converged} "
          f"|z-zref| {np.abs(z - ref['z']).max():.2e} max_r {info.max_r:.2e} max_s {info.max_s:.2e}", flush=True)
if "--full" in sys.argv:
    for wl, make in (("cw_formation", pkg.cw_formation), ("cw_rendezvous", pkg.cw_rendezvous)):
        p = make(N=1000, batch=4096)
        for mode, pm in MODES.items():
            with pkg.Solver(p, pkg.Options(rho=0.05, precision_mode=pm)) as s:
                s.run(200, residual_every=1)
                t_end = time.perf_counter() + 0.5
                while time.perf_counter() < t_end:
                    s.run(100, residual_every=1)
                t0 = time.perf_counter(); s.run(300, residual_every=1); dt = time.perf_counter() - t0
                t0 = time.perf_counter(); s.run(300, residual_every=0); dt0 = time.perf_counter() - t0
            print(f"{wl:14s} {mode:10s} {300 / dt:8.1f} it/s ({dt / 300 * 1e6:.1f} us) with residuals; {300 / dt0:8.1f} it/s without", flush=True)

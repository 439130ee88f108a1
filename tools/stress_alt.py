"""Stress of the iteration paths on a GPU box (STRESS_FLAGS=1: random option flags too; default: the
alternating path only): seeded random problems over the
compiled (n, m) set, with / without q, box or thrust-magnitude bound, many rho -- more iterations and far
more draws than the unit tests -- against the C oracle.  Prints the error distribution."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import admm_library_amd as pkg
import oracle_c as oc

dims = [(1, 1), (2, 1), (2, 2), (3, 1), (3, 2), (4, 1), (4, 3), (5, 2), (6, 1), (6, 3), (6, 4), (7, 3), (8, 2), (8, 4),
        (9, 3), (10, 2), (10, 4), (12, 3), (12, 4), (12, 6)]
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 300
worst, alt_on, errs = (0.0, None), 0, []
for trial in range(trials):
    n, m = dims[rng.integers(len(dims))]
    N = int(rng.integers(1, 160))
    batch = int(rng.choice([511, 1025, 2049, 4096, 4160]) if os.environ.get("STRESS_BIG") else rng.choice([1, 3, 64, 65, 130, 300]))
    segs = int(rng.choice([0, 0, 1, 2, 3, 5, 8, 13]))
    alpha = float(rng.choice([1.0, 1.0, 1.5]))
    with_q = bool(rng.integers(2))
    soc = bool(rng.integers(4) == 0)
    rho = float(rng.choice([0.02, 0.1, 0.5, 2.0, 8.0]))
    iters = int(rng.choice([3, 8, 17, 40]))
    flags = int(rng.choice([0, 0, 0, 8, 2, 16, 4, 6, 24])) if os.environ.get("STRESS_FLAGS") else 0
    if rng.integers(3) == 0:
        p = pkg.cw_rendezvous(N=max(N, 8) * 4, batch=batch, seed0=500 + trial, thrust_norm=soc)
    else:
        p = pkg.random_ltv(N=N, n=n, m=m, batch=batch, seed=3000 + trial, with_q=with_q, state_bounds=bool(rng.integers(2)),
                           thrust_norm=soc)
    ref = oc.solve(p, rho=rho, alpha=alpha, max_iter=iters, check_interval=1, eps_abs=0, eps_rel=0, stop=False)
    with pkg.Solver(p, pkg.Options(rho=rho, alpha=alpha, segments=segs, flags=flags)) as s:
        on = s.path()["alternating"]
        s.set_state(z=np.zeros((p.batch, p.L)), y=np.zeros((p.batch, p.L)))
        s.run(iters, residual_every=int(rng.choice([0, 1, 4])))
        w, z, y = s.get()
    e = max(np.abs(a - b).max() / max(1.0, np.abs(b).max()) for a, b in ((w, ref["w"]), (z, ref["z"]), (y, ref["y"])))
    alt_on += on
    errs.append(e)
    if e > worst[0]:
        worst = (e, dict(trial=trial, name=p.name, n=p.n, m=p.m, N=p.N, batch=batch, segs=segs, alpha=alpha, rho=rho, q=p.q is not None,
                         soc=soc, iters=iters, alt=on, flags=flags))
    if e > 1e-10:
        print("FAIL", e, worst[1], flush=True)
errs = np.array(errs)
print(f"{trials} trials, alternating path on in {alt_on}; relative error max {errs.max():.2e}, 99th pct {np.percentile(errs, 99):.2e}, "
      f"median {np.median(errs):.2e}; > 1e-11: {(errs > 1e-11).sum()}, > 1e-10: {(errs > 1e-10).sum()}")
print("worst:", worst)

"""Stress of admm_solve on a GPU box: seeded random problems, stopping rule, per-QP first-converged
iteration, adaptive rho, over-relaxation, warm start -- against the C oracle's solve."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import admm_library_amd as pkg
import oracle_c as oc

dims = [(1, 1), (2, 1), (2, 2), (3, 2), (4, 1), (4, 3), (6, 3), (6, 4), (8, 4), (12, 6)]
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0
worst = 0.0
for trial in range(trials):
    n, m = dims[rng.integers(len(dims))]
    N = int(rng.integers(2, 80))
    batch = int(rng.choice([1, 3, 64, 70, 130]))
    kw = dict(rho=float(rng.choice([0.05, 0.3, 1.0])), alpha=float(rng.choice([1.0, 1.6])), eps_abs=1e-5, eps_rel=1e-5,
              max_iter=int(rng.choice([60, 300, 1000])), check_interval=int(rng.choice([1, 5, 10, 25])))
    if rng.integers(2):
        kw.update(adapt_interval=kw["check_interval"] * int(rng.choice([1, 2, 4])), adapt_mu=float(rng.choice([5.0, 10.0])))
    flags = int(rng.choice([0, 0, 8, 2]))
    if rng.integers(3) == 0:
        p = pkg.cw_rendezvous(N=max(N, 8) * 2, batch=batch, seed0=900 + trial, thrust_norm=bool(rng.integers(3) == 0))
    else:
        p = pkg.random_ltv(N=N, n=n, m=m, batch=batch, seed=7000 + trial, with_q=bool(rng.integers(2)))
    ref = oc.solve(p, **kw)
    with pkg.Solver(p, pkg.Options(flags=flags, **kw)) as s:
        info = s.solve()
        w, z, y = s.get()
    e = max(np.abs(a - b).max() / max(1.0, np.abs(b).max()) for a, b in ((w, ref["w"]), (z, ref["z"]), (y, ref["y"])))
    same_run = info.iters_run == ref["iters_run"]
    dit = np.abs(info.iters.astype(int) - ref["iters"].astype(int)).max()
    ok = same_run and dit <= kw["check_interval"] and e <= 1e-9 and info.rho == ref["rho"]
    worst = max(worst, e if same_run else 0.0)
    if not ok:
        bad += 1
        print("MISMATCH", dict(trial=trial, name=p.name, kw=kw, flags=flags, gpu_run=info.iters_run, ref_run=ref["iters_run"], dit=int(dit), err=e,
                                rho=(info.rho, ref["rho"])), flush=True)
print(f"{trials} solves: {bad} mismatches; worst iterate error where the runs have equal length {worst:.2e}")

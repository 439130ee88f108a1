"""Stress of the per-instance path on a GPU box: seeded random per-instance problems over the compiled (n, m) pairs, segment
counts (automatic, 1 .. 32, more than N), q on / off, shared or per-instance box, alpha, plain iterations with residuals at
random intervals and full solves with the per-QP adaptive rule -- against the C oracle applied QP by QP.
    python tools/stress_pinst.py [seed=3] [trials=120]        (STRESS_WIDE=1: the wide shapes only)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import admm_library_amd as pkg
import oracle_c as oc

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 120
dims = [(6, 3), (2, 1), (4, 2), (3, 2), (1, 1), (2, 2), (4, 1), (6, 1), (6, 2), (6, 4)]
wide = [(12, 6), (8, 4), (12, 3), (9, 3)]          # rows-over-lanes kernels only (csrc/admm_pinst_wide.hpp)
only_wide = os.environ.get("STRESS_WIDE") == "1"
worst, bad = 0.0, 0
for trial in range(trials):
    n, m = wide[rng.integers(len(wide))] if (only_wide or rng.integers(3) == 0) else dims[rng.integers(len(dims))]
    N = int(rng.integers(1, 90))
    batch = int(rng.choice([1, 3, 64, 65, 130]))
    segs = int(rng.choice([0, 0, 1, 2, 3, 7, 32]))
    alpha = float(rng.choice([1.0, 1.0, 1.6]))
    rho = float(rng.choice([0.05, 0.3, 1.0]))
    soc = bool(rng.integers(3) == 0)                 # thrust-magnitude bound on most stages
    p = pkg.random_instances(N=N, n=n, m=m, batch=batch, seed=1000 + trial, with_q=bool(rng.integers(2)),
                             instance_bounds=bool(rng.integers(2)), thrust_norm=soc)
    form = str(rng.choice(["auto", "lane_per_qp", "rows"]))          # csrc/admm_pinst.hpp / admm_pinst_rows.hpp, forced either way
    os.environ.pop("ADMM_PI_LANE_PER_QP", None)
    os.environ.pop("ADMM_PI_ROWS", None)
    if form == "lane_per_qp":
        os.environ["ADMM_PI_LANE_PER_QP"] = "1"
    elif form == "rows":
        os.environ["ADMM_PI_ROWS"] = "1"
    desc = dict(trial=trial, n=n, m=m, N=N, batch=batch, segs=segs, alpha=alpha, rho=rho, form=form, soc=soc)
    try:
        if rng.integers(3) == 0:                      # a solve with the per-QP adaptive rule
            ci = int(rng.choice([1, 5, 10]))
            kw = dict(rho=rho, alpha=alpha, eps_abs=1e-6, eps_rel=1e-6, max_iter=int(rng.choice([100, 400])), check_interval=ci,
                      adapt_interval=ci * int(rng.choice([1, 2, 4])), adapt_mu=float(rng.choice([1.5, 5.0])), adapt_max=6)
            ref = oc.solve(p, **kw)
            with pkg.Solver(p, pkg.Options(segments=segs, **kw)) as s:
                info = s.solve()
                got = s.get()
                rho_q = s.rho_per_qp()
            ok = (int(info.iters_run) == ref["iters_run"] and np.array_equal(rho_q, np.broadcast_to(ref["rho"], rho_q.shape))
                  and np.array_equal(info.iters, ref["iters"]))
            desc["solve"] = True
        else:
            iters, every = int(rng.integers(1, 60)), int(rng.choice([0, 1, 3, 10]))
            ref = oc.solve(p, rho=rho, alpha=alpha, max_iter=iters, stop=False)
            with pkg.Solver(p, pkg.Options(rho=rho, alpha=alpha, segments=segs)) as s:
                s.run(iters, residual_every=every)
                got = s.get()
            ok = True
        err = max(np.abs(a - ref[k]).max() / max(1.0, np.abs(ref[k]).max()) for a, k in zip(got, ("w", "z", "y")))
    except pkg.AdmmError as e:
        print("ERROR", desc, e, flush=True)
        bad += 1
        continue
    worst = max(worst, err)
    if not ok or not (err <= 1e-10):
        bad += 1
        print("MISMATCH", desc, err, ok, flush=True)
print(f"{trials} trials: {bad} bad; worst relative iterate error {worst:.2e}")

"""Quick parity of a development variant (ADMM_HIP_LIB) on the shapes a -DADMM_DEV_DIMS build holds: iterates vs the C oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np
import admm_library_amd as pkg
import oracle_c as oc
worst = 0.0
cases = [("cw 6,3 b300", pkg.cw_rendezvous(N=200, batch=300), 0.05, {}),
         ("cw 6,3 soc", pkg.cw_rendezvous(N=64, batch=70, thrust_norm=True), 0.05, {}),
         ("ltv 6,3 q bounds", pkg.random_ltv(N=70, n=6, m=3, batch=67, seed=77), 0.3, {}),
         ("ltv 6,3 q xfree", pkg.random_ltv(N=70, n=6, m=3, batch=67, seed=77, state_bounds=False), 0.3, {}),
         ("ltv 8,4", pkg.random_ltv(N=45, n=8, m=4, batch=130, seed=79, with_q=False, state_bounds=False), 0.3, {}),
         ("ltv 2,1", pkg.random_ltv(N=33, n=2, m=1, batch=9, seed=5), 0.3, {}),
         ("cw 12,6 one-lane", pkg.cw_formation(N=120, batch=130), 0.05, {"flags": 32}),
         ("cw 6,3 alpha", pkg.cw_rendezvous(N=200, batch=130), 0.05, {"alpha": 1.6})]
for name, p, rho, kw in cases:
    alpha = kw.get("alpha", 1.0)
    with pkg.Solver(p, pkg.Options(rho=rho, **kw)) as s:
        done, errs = 0, []
        for upto in (1, 2, 3, 6, 13, 40):
            s.run(upto - done, residual_every=5); done = upto
            ref = oc.solve(p, rho=rho, alpha=alpha, max_iter=upto, stop=False)
            errs.append(max(np.abs(a - ref[k]).max() / max(1.0, np.abs(ref[k]).max()) for a, k in zip(s.get(), "wzy")))
        print(f"{name:20s} path {s.path()['kernel_family']:14s} alt {s.path()['alternating']}  max err {max(errs):.2e}", flush=True)
        worst = max(worst, max(errs))
print("worst", worst)
sys.exit(0 if worst <= 1e-10 else 1)

"""GPU check of the alternating-direction iteration against the CPU oracle and the plain path."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import admm_library_amd as pkg
from admm_library_amd import _abi
import oracle_c

p = pkg.cw_rendezvous(N=200, batch=70)
for K in (1, 2, 3, 4, 5, 8, 11):
    ref = oracle_c.solve(p, rho=0.05, max_iter=K, check_interval=10, stop=False)
    for flags in (0, _abi.FLAG_GRAPH, _abi.FLAG_NO_ALTERNATE):
        with pkg.Solver(p, pkg.Options(rho=0.05, segments=4, flags=flags)) as s:
            s.iterate(K)
            w, z, y = s.get()
        err = max(np.abs(w - ref["w"]).max(), np.abs(z - ref["z"]).max(), np.abs(y - ref["y"]).max())
        print(f"K={K} flags={flags} err={err:.2e}", flush=True)
# several calls (state carried across calls), relaxed
ref = oracle_c.solve(p, rho=0.05, alpha=1.6, max_iter=23, check_interval=10, stop=False)
with pkg.Solver(p, pkg.Options(rho=0.05, alpha=1.6, segments=5)) as s:
    for k in (3, 4, 1, 6, 2, 7):
        s.run(k, 1)
    w, z, y = s.get()
    r = s.residuals()
print("multi-call relaxed err %.2e" % max(np.abs(w - ref["w"]).max(), np.abs(z - ref["z"]).max(), np.abs(y - ref["y"]).max()))
# full size timing
p = pkg.cw_rendezvous(N=1000, batch=4096)
for flags in (_abi.FLAG_NO_ALTERNATE, 0):
    with pkg.Solver(p, pkg.Options(rho=0.05, flags=flags)) as s:
        s.run(20, 1); s.sync()
        t0 = time.perf_counter(); s.run(400, 1); s.sync(); dt = time.perf_counter() - t0
        print(f"flags={flags}: {400 / dt:.0f} it/s ({dt / 400 * 1e6:.1f} us/it)", flush=True)

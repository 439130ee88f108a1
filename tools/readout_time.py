import sys, time
sys.path.insert(0, '.')
import admm_library_amd as pkg
p = pkg.cw_rendezvous(N=1000, batch=4096)
with pkg.Solver(p, pkg.Options(rho=0.05)) as s:
    s.run(200, 1)
    pr = s.profile(50, residuals=False, fused=False)
    print("xf_ms %.1f us" % (pr["xf_ms"] * 1e3), {k: round(v * 1e3, 1) for k, v in pr.items()})

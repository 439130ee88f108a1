"""Plain fused path (ADMM_FLAG_NO_ALTERNATE: xb + xscan + xfz) and read-out (xf) timing at full size, for variant A/Bs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import admm_library_amd as pkg
p = pkg.cw_rendezvous(N=int(os.environ.get('ALT_N', 1000)), batch=int(os.environ.get('ALT_BATCH', 4096)))
with pkg.Solver(p, pkg.Options(rho=0.05, flags=8)) as s:
    s.run(50, 1); s.sync()
    t0 = time.perf_counter(); s.run(200, 1); s.sync(); dt = time.perf_counter() - t0
    pr = s.profile(50, residuals=True, fused=True)
    pu = s.profile(50, residuals=True, fused=False)
    print(f"plain path: {200 / dt:.0f} it/s ({dt / 200 * 1e6:.1f} us/it)  " + " ".join(f"{k}={v * 1e3:.1f}us" for k, v in pr.items()) + f"  | unfused xf={pu['xf_ms'] * 1e3:.1f}us zdual={pu['zdual_ms'] * 1e3:.1f}us", flush=True)

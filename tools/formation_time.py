"""configs[4] shape (n = 12, m = 6) on the library's default family: rate with residuals every iteration / every 10th + per-kernel times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import admm_library_amd as pkg
p = pkg.cw_formation(N=1000, batch=int(os.environ.get("ALT_BATCH", 4096)))
with pkg.Solver(p, pkg.Options(rho=0.05)) as s:
    for every in (1, 10):
        s.run(50, every); s.sync()
        t0 = time.perf_counter(); s.run(300, every); s.sync(); dt = time.perf_counter() - t0
        print(f"{s.path()['kernel_family']} residuals every {every}: {300 / dt:.0f} it/s ({dt / 300 * 1e6:.1f} us/it)", flush=True)
    for res in (True, False):
        pr = s.profile(50, residuals=res, alternating=True)
        print("   resid=%d: " % res + " ".join(f"{k}={v * 1e3:.1f}us" for k, v in pr.items()), flush=True)

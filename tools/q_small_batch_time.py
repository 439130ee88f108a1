import os, sys, time
sys.path.insert(0, os.environ.get("R", "/root/repo"))
import admm_library_amd as pkg
for N in (200, 1000):
    p = pkg.random_ltv(N=N, n=6, m=3, batch=1, seed=5)
    for flags in (0, 32):
        with pkg.Solver(p, pkg.Options(rho=0.3, flags=flags)) as s:
            t_end = time.perf_counter() + 0.3
            while time.perf_counter() < t_end:
                s.run(200, 10)
            t0 = time.perf_counter(); s.run(2000, 10); s.sync(); dt = time.perf_counter() - t0
            print(f"N={N} batch 1 with q, flags={flags} S={s.geometry()['segments']}: {dt / 2000 * 1e6:.2f} us/iteration", flush=True)

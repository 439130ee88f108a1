"""Fused-kernel time against batch size and segment count: does a working set that
fits the 256 MB infinity cache run faster per QP?  (It does not.)  Run on a GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import admm_library_amd as pkg
for b, S in ((4096, 16), (2048, 32), (2048, 16), (1536, 42), (1024, 64), (1024, 32), (1024, 16), (512, 64)):
    p = pkg.cw_rendezvous(N=1000, batch=b)
    with pkg.Solver(p, pkg.Options(rho=0.05, segments=S)) as s:
        s.run(20, 1); s.sync()
        t0 = time.perf_counter(); s.run(200, 1); s.sync(); dt = time.perf_counter() - t0
        pr = s.profile(50, residuals=True, alternating=True)
        ws = 16.0 * 9000 * b / 1e6
        print(f"batch={b} S={S} working set {ws:.0f} MB: {dt / 200 * 1e6:.1f} us/it  " + " ".join(f"{k}={v * 1e3:.1f}" for k, v in pr.items())
              + f"  | xfze per 4096 QPs: {pr['xfze_ms'] * 1e3 * 4096 / b:.1f} us, xbze: {pr['xbze_ms'] * 1e3 * 4096 / b:.1f} us", flush=True)

"""Timing of the default iteration at full size (for rocprofv3 runs and variant sweeps)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import admm_library_amd as pkg
flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
resid = int(sys.argv[2]) if len(sys.argv) > 2 else 1
segs = int(sys.argv[3]) if len(sys.argv) > 3 else 0
p = pkg.cw_rendezvous(N=int(os.environ.get('ALT_N', 1000)), batch=int(os.environ.get('ALT_BATCH', 4096)))
with pkg.Solver(p, pkg.Options(rho=0.05, flags=flags, segments=segs)) as s:
    s.run(20, resid); s.sync()
    t0 = time.perf_counter(); s.run(200, resid); s.sync(); dt = time.perf_counter() - t0
    print(f"flags={flags} resid={resid} S={s.geometry()['segments']}: {200 / dt:.0f} it/s ({dt / 200 * 1e6:.1f} us/it)", flush=True)
    if not flags & 8:
        for res in (True, False):
            pr = s.profile(50, residuals=res, alternating=True)
            print("   alt profile resid=%d: " % res + " ".join(f"{k}={v * 1e3:.1f}us" for k, v in pr.items()), flush=True)

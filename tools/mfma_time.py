"""Iteration rate + per-kernel HIP-event times of one precision mode at full size (variant sweeps, rocprofv3 runs).
   python tools/mfma_time.py <precision_mode 0|1|2> [workload=cw_formation] [resid=1]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import admm_library_amd as pkg
pm = int(sys.argv[1]) if len(sys.argv) > 1 else 2
wl = sys.argv[2] if len(sys.argv) > 2 else "cw_formation"
resid = int(sys.argv[3]) if len(sys.argv) > 3 else 1
p = getattr(pkg, wl)(N=int(os.environ.get('ALT_N', 1000)), batch=int(os.environ.get('ALT_BATCH', 4096)))
with pkg.Solver(p, pkg.Options(rho=0.05, precision_mode=pm)) as s:
    t_end = time.perf_counter() + 0.5
    while time.perf_counter() < t_end:
        s.run(100, resid)
    t0 = time.perf_counter(); s.run(300, resid); s.sync(); dt = time.perf_counter() - t0
    print(f"{wl} precision_mode={pm} resid={resid} S={s.geometry()['segments']}: {300 / dt:.0f} it/s ({dt / 300 * 1e6:.1f} us/it)", flush=True)
    if True:
        pr = s.profile(100, residuals=bool(resid), alternating=True)
        print("   per-launch events: " + " ".join(f"{k}={v * 1e3:.1f}us" for k, v in pr.items()), flush=True)
    else:
        pr = s.profile(100, residuals=bool(resid), fused=True)
        print("   per-launch events: " + " ".join(f"{k}={v * 1e3:.1f}us" for k, v in pr.items()), flush=True)

"""One-lane (DPP operand) kernels against the fp64 MFMA form: iteration rate by shape and batch, residuals every iteration and
every 10th -- the measurement behind the ADMM_PRECISION_FP64 selection rule (include/admm_hip.h).
   python tools/family_time.py            (dev variants hold (6,3), (12,6), (8,4), (2,1) only)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import admm_library_amd as pkg
from admm_library_amd import _abi

def rate(p, resid, **kw):
    with pkg.Solver(p, pkg.Options(rho=0.05, **kw)) as s:
        fam = s.path()["kernel_family"]
        t_end = time.perf_counter() + 0.3
        while time.perf_counter() < t_end:
            s.run(100, resid)
        t0 = time.perf_counter(); s.run(300, resid); s.sync(); dt = time.perf_counter() - t0
    return dt / 300 * 1e6, fam

cases = [("cw_rendezvous", 1000, b) for b in (1, 16, 64, 128, 256, 1024, 4096)] + [("cw_rendezvous", 200, b) for b in (1, 64)] + \
        [("cw_formation", 1000, b) for b in (1, 64, 256, 1024, 4096)]
for wl, N, b in cases:
    p = getattr(pkg, wl)(N=N, batch=b)
    row = []
    for name, kw in (("one_lane", dict(flags=_abi.FLAG_NO_MFMA)), ("fp64_mfma", dict(precision_mode=_abi.PRECISION_FP64_MFMA))):
        for resid in (1, 10):
            us, fam = rate(p, resid, **kw)
            row.append(f"{name} r{resid}: {us:7.1f} us")
    print(f"{wl:14s} N={N:5d} batch={b:5d}  " + " | ".join(row), flush=True)

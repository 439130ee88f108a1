#!/usr/bin/env python3
"""Per-instance sweeps at every wide shape: per-launch times of a 4096-QP batch with per-instance boxes and their fraction of the
HBM roofline (bytes as in bench.py's cw_formation_perinstance line).   python tools/pinst_wide_time.py [N=500] [batch=4096]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import admm_library_amd as pkg  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
for n, m in ((12, 6), (8, 4), (12, 3), (9, 3), (6, 3)):
    p = pkg.random_instances(N=N, n=n, m=m, batch=batch, seed=5, with_q=False)
    with pkg.Solver(p, pkg.Options(rho=0.3)) as s:
        geo = s.geometry()
        s.run(20, residual_every=1)
        pr = s.profile(30, residuals=True, fused=True)
    nb = n + m
    seg = n * m if geo["segments"] > 1 else 0
    b_xb = (n * n + 2 * n * m + m * m + seg) * 8 + (nb * 3 + m) * 8          # A, B, K, S^-1 (+ Omega); v, lo, hi read, d written
    b_xfz = (n * n + 2 * n * m + seg) * 8 + (m + nb * 3 + nb) * 8            # K, A, B (+ Psi); d, v, lo, hi read, v+ written
    tot = (b_xb + b_xfz) * N * geo["pitch"]
    ms = pr["xb_ms"] + pr["xfz_ms"]
    print(f"(n, m) = ({n}, {m}) batch {batch} N {N} segments {geo['segments']}: pxb {pr['xb_ms']:.3f} ms, pxfz {pr['xfz_ms']:.3f} ms, "
          f"scan {pr['xscan_ms']:.3f} ms; {tot / 1e9:.2f} GB per iteration -> {tot / ms / 1e9:.2f} TB/s = {tot / ms / 8e9:.2f} of 8 TB/s", flush=True)

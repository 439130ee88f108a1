#!/usr/bin/env python3
"""Per-instance dynamics: iteration time against the batch size.  One lane per QP sweeps the whole horizon,
so an iteration takes N x (latency of one stage) until the batch is large enough to saturate HBM.
    python tools/pinst_time.py [N=1000] [batches, default "1024 4096 16384 32768"]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import admm_library_amd as pkg  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
batches = [int(b) for b in (sys.argv[2].split() if len(sys.argv) > 2 else "1024 4096 16384 32768".split())]
for batch in batches:
    t = time.perf_counter()
    p = pkg.cw_rendezvous_instances(N=N, batch=batch)
    tb = time.perf_counter() - t
    nb = p.n + p.m
    with pkg.Solver(p, pkg.Options(rho=0.05, check_interval=1)) as s:
        s.run(5, residual_every=1)
        t_end = time.perf_counter() + 0.5
        while time.perf_counter() < t_end:
            s.run(5, residual_every=1)
        t0 = time.perf_counter()
        s.run(20, residual_every=1)
        s.sync()
        dt = (time.perf_counter() - t0) / 20
        pr = s.profile(10, residuals=True, fused=True)
        geo = s.geometry()
        elems = p.L * geo["pitch"]
        # operands (A, B, K, Si | K, A, B) + state + per-instance box (+ Omega_k / Psi_k with segments), DESIGN.md §4.10
        ops = (2 * p.n * p.n + 4 * p.n * p.m + p.m * p.m + (2 * p.n * p.m if geo["segments"] > 1 else 0)) * 8.0 / nb
        bytes_per_elem = ops + 29.33 + 32.0
    print("   profile: " + " ".join(f"{k}={v * 1e3:.1f}us" for k, v in pr.items()))
    print(f"batch {batch} N {N} segments {geo['segments']}: {dt * 1e3:.3f} ms/iteration ({batch / dt / 1e6:.2f} M QP-iterations/s), xb {pr['xb_ms']:.3f} ms "
          f"xfz {pr['xfz_ms']:.3f} ms, {bytes_per_elem * elems / dt / 1e12:.2f} TB/s algorithmic (problem built in {tb:.0f} s)", flush=True)
    del p

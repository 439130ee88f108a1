#!/usr/bin/env python3
"""Per-instance dynamics: cost of a device refactor (admm_set_rho = trial factorisation + commit over every QP) and of
admm_update_problem (upload + refactor).    python tools/pinst_refactor_time.py [N=1000] [batches="64 1024 4096"] [n=6|12]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import admm_library_amd as pkg  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
make = pkg.cw_formation_instances if (len(sys.argv) > 3 and sys.argv[3] == "12") else pkg.cw_rendezvous_instances
for batch in [int(b) for b in (sys.argv[2].split() if len(sys.argv) > 2 else "64 1024 4096".split())]:
    p = make(N=N, batch=batch)
    t = time.perf_counter()
    with pkg.Solver(p, pkg.Options(rho=0.05)) as s:
        s.sync()
        t_setup = (time.perf_counter() - t) * 1e3
        s.iterate(3)
        ts = []
        for r in (0.1, 0.05, 0.2, 0.05):
            s.sync()
            t = time.perf_counter()
            s.set_rho(r)
            s.sync()
            ts.append((time.perf_counter() - t) * 1e3)
        t = time.perf_counter()
        s.update_problem(p)
        s.sync()
        tu = (time.perf_counter() - t) * 1e3
        print(f"n {p.n} batch {batch} N {N} segments {s.geometry()['segments']}: setup {t_setup:.0f} ms, set_rho {min(ts):.2f} ms (of {' '.join('%.2f' % x for x in ts)}), "
              f"update_problem {tu:.1f} ms (admm_update_problem itself: {s.last_update_ms:.1f} ms; the rest is host marshalling)", flush=True)

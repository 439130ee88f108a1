#!/usr/bin/env python3
"""Stress at BASELINE's horizon: random option / form combinations on batches of 1024-4096 QPs of N = 1000 stages, shared and
per-instance dynamics, compared with the C oracle on 12 QPs spread over the batch (the other stress tools and the parity tests use
N < 100; round 3's set-up race needed GBs of arrays).     python tools/stress_scale.py [seed=5] [trials=24]"""
import dataclasses
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import admm_library_amd as pkg
import oracle_c as oc
from admm_library_amd import _abi

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 24
worst, bad, n_mixed = 0.0, 0, 0
for trial in range(trials):
    kind = str(rng.choice(["cw6", "cw12", "pinst6", "pinst12"]))
    batch = int(rng.choice([1024, 2048, 4096])) if kind != "pinst12" else int(rng.choice([512, 1024]))
    soc = bool(rng.integers(3) == 0) and kind in ("cw6", "pinst6", "pinst12")
    alpha = float(rng.choice([1.0, 1.6]))
    segs = int(rng.choice([0, 0, 4, 16])) if kind.startswith("pinst") else int(rng.choice([0, 0, 8, 32]))
    flags, pm = 0, 0
    if kind.startswith("cw"):
        family = str(rng.choice(["default", "one_lane", "plain", "graph", "unfused", "fp64_mfma", "mixed"]))
        if soc and family in ("fp64_mfma", "mixed"):
            family = "default"                      # (the MFMA family has no thrust-magnitude forms)
        flags = {"one_lane": _abi.FLAG_NO_MFMA, "plain": _abi.FLAG_NO_ALTERNATE, "graph": _abi.FLAG_GRAPH, "unfused": _abi.FLAG_UNFUSED}.get(family, 0)
        pm = {"fp64_mfma": _abi.PRECISION_FP64_MFMA, "mixed": _abi.PRECISION_MIXED}.get(family, 0)
    if kind == "cw6":
        p = pkg.cw_rendezvous(N=1000, batch=batch, seed0=1000 + trial, thrust_norm=soc)
    elif kind == "cw12":
        p = pkg.cw_formation(N=1000, batch=batch, seed0=1000 + trial)
    else:
        p = (pkg.cw_rendezvous_instances if kind == "pinst6" else pkg.cw_formation_instances)(N=1000, batch=batch, seed0=1000 + trial)
        if soc:
            lo, hi = p.lo.copy(), p.hi.copy()
            lo[..., :p.m], hi[..., :p.m] = -np.inf, np.inf
            p = dataclasses.replace(p, lo=lo, hi=hi, unorm=0.25)
    idx = np.linspace(0, batch - 1, 12).astype(int)
    kw = dict(x0=p.x0[idx])
    if p.per_instance:
        kw.update(A=p.A[idx], B=p.B[idx], lo=p.lo[idx], hi=p.hi[idx])
    sub = dataclasses.replace(p, **kw)
    desc = dict(trial=trial, kind=kind, batch=batch, soc=soc, alpha=alpha, segs=segs, flags=flags, precision=pm)
    try:
        ref = oc.solve(sub, rho=0.05, alpha=alpha, max_iter=13, stop=False)
        with pkg.Solver(p, pkg.Options(rho=0.05, alpha=alpha, segments=segs, flags=flags, precision_mode=pm)) as s:
            s.run(9, residual_every=4)
            s.iterate(4)
            got = s.get()
    except pkg.AdmmError as e:
        print("ERROR", desc, e, flush=True)
        bad += 1
        continue
    err = max(np.abs(a[idx] - ref[k]).max() / max(1.0, np.abs(ref[k]).max()) for a, k in zip(got, ("w", "z", "y")))
    tol = 5e-5 if pm == _abi.PRECISION_MIXED else 1e-10          # (the mixed mode at full size: its stated 5e-5, tests/test_gpu_mfma.py)
    if pm == _abi.PRECISION_MIXED:
        n_mixed += 1
        worst_mixed = max(globals().get("worst_mixed", 0.0), err)
    else:
        worst = max(worst, err)
    if not err <= tol:
        bad += 1
        print("MISMATCH", desc, err, flush=True)
    del p, sub, got
print(f"{trials} trials at N = 1000: {bad} bad; worst relative iterate error on the sampled QPs {worst:.2e} (fp64 paths), "
      f"{globals().get('worst_mixed', 0.0):.2e} (mixed mode, {n_mixed} trials)")

"""Default-path parity at BASELINE.json's own sizes (VERDICT r01, next #1a).

What ships is the default path: automatic segment count, the fp64-MFMA scan (split-K for small
batches) and the alternating-direction fused kernels.  These tests run exactly
that -- default Options apart from rho -- on configs[1] (ONE N=1000 n=6 m=3 orbit-transfer QP) and on
a 64-QP slice of configs[2], against the C oracle's plain sequential Riccati sweep, through many
iterations and through a full admm_solve.

Tolerance: 1e-10 absolute on fp64 iterates (BASELINE.json north_star), scaled by max(1, |ref|_inf).
PARITY UNPINNED: the oracle is the build's own CPU restatement.
"""
import numpy as np
import pytest

import admm_library_amd as pkg
import oracle_c as oc

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _close(got, ref, what):
    err = np.abs(got - ref).max()
    assert err <= TOL * max(1.0, np.abs(ref).max()), (what, err)
    return err


@pytest.mark.parametrize("flags", [0, 32], ids=["default_mfma_form", "one_lane_kernels"])
def test_config1_iterates_default_path(gpu, flags):
    """configs[1]: cw_rendezvous(N=1000, batch=1), segments=0 (auto -> 64 one-wave segments, split-K
    scan), default flags: iterates vs the oracle after 1, 2, 10, 40, 200 iterations.  The default for a batch
    this small is the fp64 MFMA form of the fused kernels; ADMM_FLAG_NO_MFMA = the one-lane kernels."""
    p = pkg.cw_rendezvous(N=1000, batch=1)
    with pkg.Solver(p, pkg.Options(rho=0.05, flags=flags)) as s:
        geo = s.geometry()
        assert geo["segments"] > 1                       # the parallel-in-time form really runs
        done = 0
        for upto in (1, 2, 10, 40, 200):
            s.iterate(upto - done)
            done = upto
            w, z, y = s.get()
            ref = oc.solve(p, rho=0.05, max_iter=upto, stop=False)
            for name, a in (("w", w), ("z", z), ("y", y)):
                _close(a, ref[name], (upto, name))


@pytest.mark.parametrize("flags", [0, 32], ids=["default_mfma_form", "one_lane_kernels"])
def test_config1_solve_default_path(gpu, flags):
    """configs[1]: a full admm_solve on the default path -- same iteration count, per-QP first-converged
    iteration, residuals and solution as the oracle."""
    p = pkg.cw_rendezvous(N=1000, batch=1)
    kw = dict(rho=0.05, eps_abs=1e-6, eps_rel=1e-6, max_iter=4000, check_interval=10)
    ref = oc.solve(p, **kw)
    with pkg.Solver(p, pkg.Options(flags=flags, **kw)) as s:
        info = s.solve()
        w, z, y = s.get()
    assert int(info.iters_run) == int(ref["iters_run"])
    np.testing.assert_array_equal(info.iters, ref["iters"])
    np.testing.assert_array_equal(info.status, ref["status"])
    for name, a in (("w", w), ("z", z), ("y", y)):
        _close(a, ref[name], name)
    assert abs(info.r[0] - ref["r"][0]) <= 1e-10 and abs(info.s[0] - ref["s"][0]) <= 1e-10


def test_config1_solve_adaptive_default_path(gpu):
    """configs[1] with the adaptive-rho rule: rho trajectory, counts and solution as the oracle."""
    p = pkg.cw_rendezvous(N=1000, batch=1)
    kw = dict(rho=0.05, eps_abs=1e-6, eps_rel=1e-6, max_iter=4000, check_interval=10,
              adapt_interval=50, adapt_mu=10.0, adapt_tau=2.0)
    ref = oc.solve(p, **kw)
    with pkg.Solver(p, pkg.Options(**kw)) as s:
        info = s.solve()
        w, z, y = s.get()
    assert int(info.iters_run) == int(ref["iters_run"])
    assert float(info.rho) == float(ref["rho"]) and int(info.rho_updates) == int(ref["rho_updates"])
    for name, a in (("w", w), ("z", z), ("y", y)):
        _close(a, ref[name], name)


@pytest.mark.parametrize("iters", [200, 400])
def test_config2_slice_many_iterations(gpu, iters):
    """A 64-QP slice of configs[2] (N=1000, n=6, m=3) on the default path for >= 200 iterations vs the oracle
    (the oracle needs ~3 s for it).  The slice is solved as its own batch: pitch 64, auto segments."""
    p = pkg.cw_rendezvous(N=1000, batch=4096).slice(1000, 1064)
    with pkg.Solver(p, pkg.Options(rho=0.05)) as s:
        s.iterate(iters)
        w, z, y = s.get()
    ref = oc.solve(p, rho=0.05, max_iter=iters, stop=False)
    for name, a in (("w", w), ("z", z), ("y", y)):
        _close(a, ref[name], (iters, name))


def test_config2_full_batch_slice_many_iterations(gpu):
    """configs[2] itself (batch 4096, S = 16 segments of 62.5 stages, one workgroup per CU) for 200 iterations
    with residuals every 10th: a 64-QP slice of the result vs the oracle run on those 64 QPs alone (QPs are
    independent, so the slice of the batched solve must equal the solve of the slice)."""
    full = pkg.cw_rendezvous(N=1000, batch=4096)
    lo = 2048 - 32
    with pkg.Solver(full, pkg.Options(rho=0.05)) as s:
        assert s.geometry()["segments"] == 16
        s.run(200, residual_every=10)
        w, z, y = s.get()
        r, sd, nw, nz, ny = s.residuals()
    sub = full.slice(lo, lo + 64)
    ref = oc.solve(sub, rho=0.05, max_iter=200, check_interval=10, stop=False)
    for name, a in (("w", w), ("z", z), ("y", y)):
        _close(a[lo:lo + 64], ref[name], name)
    assert np.abs(r[lo:lo + 64] - ref["r"]).max() <= 1e-10
    assert np.abs(sd[lo:lo + 64] - ref["s"]).max() <= 1e-10


@pytest.mark.parametrize("batch", [1, 2, 3, 4])
def test_matrix_vector_scan_of_tiny_batches(gpu, batch, monkeypatch):
    """Batches of up to 4 QPs run the segment scan as a matrix-vector product per column (xscan_gemv_kernel, DESIGN.md
    §4.9) instead of the 16-column MFMA panel: same iterates as with ADMM_NO_GEMV_SCAN within the parity tolerance, both within 1e-10 of
    the oracle, through residual and non-residual iterations (the finalise role rides in either scan launch)."""
    p = pkg.cw_rendezvous(N=600, batch=batch)
    ref = oc.solve(p, rho=0.05, max_iter=63, check_interval=7, stop=False)
    got = []
    for gemv in (True, False):
        if gemv:
            monkeypatch.delenv("ADMM_NO_GEMV_SCAN", raising=False)
        else:
            monkeypatch.setenv("ADMM_NO_GEMV_SCAN", "1")
        with pkg.Solver(p, pkg.Options(rho=0.05)) as s:
            assert s.geometry()["segments"] > 8
            s.run(63, residual_every=7)
            got.append(s.get() + s.residuals()[:2])
    for a, b in zip(*got):        # (rounding of the scan differs; the early-stage gains of the forward-elimination form amplify it, §4.8)
        assert np.abs(a - b).max() <= TOL * max(1.0, np.abs(b).max()), np.abs(a - b).max()
    for name, a in zip(("w", "z", "y"), got[0][:3]):
        _close(a, ref[name], (batch, name))
    assert np.abs(got[0][3] - ref["r"]).max() <= 1e-10 and np.abs(got[0][4] - ref["s"]).max() <= 1e-10

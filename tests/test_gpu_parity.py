"""GPU parity tests: the HIP path, called through the C ABI (ctypes), against
the CPU oracle on the same seeded inputs.

Tolerance: 1e-10 absolute on fp64 iterates (BASELINE.json north_star), with
variables scaled to O(1).  The oracle is the build's own CPU restatement
(PARITY UNPINNED: the reference ships no code or fixtures), and it
uses the plain sequential Riccati sweep, not the segmented form of the kernels.
"""
import numpy as np
import pytest

import admm_library_amd as pkg
import admm_ref as ar
import oracle_c as oc

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _g(p, z, y, rho):
    g = -rho * (z - y)
    return g if p.q is None else g + p.q


CASES = [
    # (factory, rho, segments)
    (lambda: pkg.double_integrator(N=50, batch=1), 1.0, 0),
    (lambda: pkg.double_integrator(N=50, batch=130), 1.0, 7),
    (lambda: pkg.cw_rendezvous(N=200, batch=3), 0.05, 0),
    (lambda: pkg.cw_rendezvous(N=1000, batch=65), 0.05, 0),
    (lambda: pkg.cw_rendezvous(N=1000, batch=1), 0.05, 1),
    (lambda: pkg.random_ltv(N=37, n=4, m=2, batch=70, seed=2), 0.3, 5),
    (lambda: pkg.random_ltv(N=16, n=6, m=6, batch=5, seed=4), 0.5, 3),
    (lambda: pkg.random_ltv(N=24, n=12, m=6, batch=3, seed=6), 0.4, 4),
    (lambda: pkg.random_ltv(N=9, n=2, m=2, batch=2, seed=8, with_q=False), 0.4, 9),
    (lambda: pkg.random_ltv(N=30, n=8, m=4, batch=64, seed=10), 0.2, 0),
    (lambda: pkg.random_ltv(N=11, n=3, m=1, batch=4, seed=12), 0.2, 2),
    (lambda: pkg.random_ltv(N=11, n=4, m=1, batch=4, seed=13), 0.2, 2),
    (lambda: pkg.random_ltv(N=11, n=4, m=4, batch=4, seed=14), 0.2, 2),
    (lambda: pkg.random_ltv(N=11, n=6, m=2, batch=4, seed=15), 0.2, 2),
    (lambda: pkg.random_ltv(N=11, n=12, m=3, batch=4, seed=16), 0.2, 2),
    # segments longer than one LDS record chunk (several refills per segment)
    (lambda: pkg.random_ltv(N=40, n=12, m=6, batch=3, seed=17), 0.4, 2),
    (lambda: pkg.cw_rendezvous(N=200, batch=70), 0.05, 2),
    (lambda: pkg.cw_formation(N=120, batch=66), 0.05, 0),       # configs[4] shape, full Q / QN
    # the wider compiled set (admm_dims_g*.hip)
    (lambda: pkg.random_ltv(N=14, n=1, m=1, batch=5, seed=51), 0.3, 2),
    (lambda: pkg.random_ltv(N=14, n=5, m=3, batch=5, seed=52), 0.3, 2),
    (lambda: pkg.random_ltv(N=14, n=7, m=3, batch=5, seed=53), 0.3, 2),
    (lambda: pkg.random_ltv(N=14, n=9, m=3, batch=5, seed=54), 0.3, 2),
    (lambda: pkg.random_ltv(N=14, n=10, m=4, batch=5, seed=55), 0.3, 2),
    # degenerate horizons
    (lambda: pkg.random_ltv(N=1, n=4, m=2, batch=3, seed=18), 0.3, 0),
    (lambda: pkg.random_ltv(N=2, n=6, m=3, batch=2, seed=19), 0.3, 0),
    (lambda: pkg.random_ltv(N=3, n=2, m=1, batch=129, seed=20), 0.3, 3),
]


@pytest.mark.parametrize("flags", [0, 4], ids=["scan_mfma", "scan_chain"])
@pytest.mark.parametrize("idx", range(len(CASES)))
def test_x_update_kernels(gpu, idx, flags):
    """T5: xb + xscan + xf on random (z, y) vs the oracle's sequential sweep, with the
    segment scan as the fp64-MFMA GEMM (default) and as the sequential chain."""
    make, rho, segs = CASES[idx]
    p = make()
    rng = np.random.default_rng(100 + idx)
    z = rng.standard_normal((p.batch, p.L))
    y = rng.standard_normal((p.batch, p.L))
    with pkg.Solver(p, pkg.Options(rho=rho, segments=segs, flags=flags)) as s:
        s.set_state(z=z, y=y)
        s.step_x()
        w, z2, y2 = s.get()
    np.testing.assert_array_equal(z2, z)     # layout round trip is exact
    np.testing.assert_array_equal(y2, y)
    f = ar.factor(p.A, p.B, p.Q, p.R, p.QN, rho, p.N)
    w_ref = ar.x_update(f, _g(p, z, y, rho), p.x0)
    scale = max(1.0, np.abs(w_ref).max())
    assert np.abs(w - w_ref).max() <= TOL * scale


@pytest.mark.parametrize("alpha", [1.0, 1.7])
@pytest.mark.parametrize("resid", [False, True])
def test_zdual_kernel(gpu, alpha, resid):
    """T5: fused z/dual/residual kernel, incl. elements exactly on bounds and
    +-inf bounds; z+, y+ must be BIT-exact (no reassociation is involved)."""
    p = pkg.random_ltv(N=41, n=4, m=2, batch=67, seed=21)
    rho = 0.35
    rng = np.random.default_rng(5)
    lo, hi = ar.expand_bounds(p.lo, p.hi, p.N, p.nb)
    w = rng.standard_normal((p.batch, p.L))
    z = rng.standard_normal((p.batch, p.L))
    y = 0.5 * rng.standard_normal((p.batch, p.L))
    fin = np.isfinite(hi)
    y[:, ::5] = 0.0
    w[:, ::5] = np.where(fin, hi, 0.25)[None, ::5]        # exactly on the upper bound
    with pkg.Solver(p, pkg.Options(rho=rho, alpha=alpha, zrows=8)) as s:
        s.set_state(w=w, z=z, y=y)
        s.step_z(residuals=resid)
        w2, zn, yn = s.get()
        if resid:
            r, sd, nw, nz, ny = s.residuals()
    zr, yr = ar.z_update(w, z, y, lo, hi, alpha)
    np.testing.assert_array_equal(w2, w)
    if alpha == 1.0:
        np.testing.assert_array_equal(zn, zr)
        np.testing.assert_array_equal(yn, yr)
    else:   # fma(alpha, w, (1-alpha) z) vs two roundings
        assert np.abs(zn - zr).max() < 1e-14 and np.abs(yn - yr).max() < 1e-14
    if resid:
        rr = ar.residuals(w, z, zr, yr, rho)
        for got, ref in zip((r, sd, nw, nz, ny), rr):
            assert np.abs(got - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("flags", [0, 32, 8, 2], ids=["default", "one_lane", "fused_plain", "unfused"])
@pytest.mark.parametrize("idx", [0, 1, 3, 5, 7, 9, 15, 16, 17, 18, 20])
def test_iterate_parity(gpu, idx, flags):
    """T4: iterates of the full loop vs the C oracle after 1, 2, 10, 40 iterations, on the
    default path (the alternating-direction kernels where compiled, tests/test_gpu_alternating.py,
    else xb, xscan, xfz), on the plain fused path (ADMM_FLAG_NO_ALTERNATE) and on the
    ADMM_FLAG_UNFUSED path (xb, xscan, xf, zdual).  w is re-materialised by admm_get on the fused paths."""
    make, rho, segs = CASES[idx]
    p = make()
    with pkg.Solver(p, pkg.Options(rho=rho, segments=segs, flags=flags)) as s:
        done = 0
        for upto in (1, 2, 10, 40):
            s.iterate(upto - done)
            done = upto
            w, z, y = s.get()
            ref = oc.solve(p, rho=rho, max_iter=upto, stop=False)
            for a, b in ((w, ref["w"]), (z, ref["z"]), (y, ref["y"])):
                assert np.abs(a - b).max() <= TOL * max(1.0, np.abs(b).max()), (upto,)


def test_fused_relaxed_with_residuals(gpu):
    """Fused path with over-relaxation and residuals every iteration (admm_run)."""
    p = pkg.random_ltv(N=33, n=6, m=3, batch=69, seed=31)
    rho, alpha = 0.3, 1.6
    with pkg.Solver(p, pkg.Options(rho=rho, alpha=alpha, segments=4)) as s:
        s.run(12, residual_every=1)
        w, z, y = s.get()
        r, sd, nw, nz, ny = s.residuals()
    ref = oc.solve(p, rho=rho, alpha=alpha, max_iter=12, check_interval=1, eps_abs=0, eps_rel=0, stop=False)
    for a, b in ((w, ref["w"]), (z, ref["z"]), (y, ref["y"])):
        assert np.abs(a - b).max() <= TOL * max(1.0, np.abs(b).max())
    assert np.abs(r - ref["r"]).max() <= 1e-10 and np.abs(sd - ref["s"]).max() <= 1e-10


def test_graph_and_direct_launch_agree(gpu):
    p = pkg.cw_rendezvous(N=100, batch=66)
    outs = []
    for flags in (0, 16):          # direct launches (default) / ADMM_FLAG_GRAPH
        with pkg.Solver(p, pkg.Options(rho=0.05, flags=flags)) as s:
            s.iterate(17)
            outs.append(s.get())
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("flags", [0, 2], ids=["fused", "unfused"])
@pytest.mark.parametrize("alpha", [1.0, 1.5])
def test_solve_matches_oracle(gpu, alpha, flags):
    """Full admm_solve: stopping rule, per-QP iteration counts, warm start."""
    p = pkg.double_integrator(N=50, batch=37)
    kw = dict(rho=1.0, alpha=alpha, eps_abs=1e-7, eps_rel=1e-7, max_iter=3000, check_interval=10)
    ref = oc.solve(p, **kw)
    with pkg.Solver(p, pkg.Options(flags=flags, **kw)) as s:
        info = s.solve()
        w, z, y = s.get()
        assert info.iters_run == ref["iters_run"]
        assert info.n_converged == int(ref["status"].sum()) == p.batch
        # a stop decision may flip by one check near the threshold (reduction order)
        assert (np.abs(info.iters - ref["iters"]) <= kw["check_interval"]).all()
        assert (info.iters == ref["iters"]).mean() > 0.9
        for a, b in ((w, ref["w"]), (z, ref["z"]), (y, ref["y"])):
            assert np.abs(a - b).max() <= TOL * max(1.0, np.abs(b).max())
        # warm start from the solution converges at the first check
        info2 = s.solve(z0=z, y0=y)
        assert info2.iters_run == kw["check_interval"]


def test_solve_max_iter_and_unconverged(gpu):
    p = pkg.cw_rendezvous(N=150, batch=5)
    kw = dict(rho=0.05, eps_abs=1e-12, eps_rel=1e-12, max_iter=23, check_interval=10)
    ref = oc.solve(p, **kw)
    with pkg.Solver(p, pkg.Options(**kw)) as s:
        info = s.solve()
        assert info.iters_run == 23 == ref["iters_run"]
        assert info.n_converged == 0
        np.testing.assert_array_equal(info.status, ref["status"])
        np.testing.assert_array_equal(info.iters, np.full(5, 23, np.int32))
        assert np.abs(info.r - ref["r"]).max() < 1e-10 and np.abs(info.s - ref["s"]).max() < 1e-10


@pytest.mark.parametrize("flags", [0, 2], ids=["fused", "unfused"])
@pytest.mark.parametrize("kw", [dict(adapt_interval=20), dict(adapt_interval=20, adapt_mu=5.0),
                                dict(adapt_interval=50, adapt_tau=3.0, adapt_max=2)])
def test_adaptive_rho_matches_oracle(gpu, kw, flags):
    """batch-level residual balancing.  Same rho trajectory, same iteration
    counts, same iterates as the oracle (the decision sums run in the same order on both sides)."""
    p = pkg.cw_rendezvous(N=200, batch=12)
    base = dict(rho=0.05, eps_abs=1e-6, eps_rel=1e-6, max_iter=3000, check_interval=10)
    ref = oc.solve(p, **base, **kw)
    assert ref["rho_updates"] >= 2
    with pkg.Solver(p, pkg.Options(flags=flags, **base, **kw)) as s:
        info = s.solve()
        w, z, y = s.get()
    assert info.iters_run == ref["iters_run"] and info.rho_updates == ref["rho_updates"]
    assert info.rho == ref["rho"]
    assert (np.abs(info.iters - ref["iters"]) <= base["check_interval"]).all()
    for a, b in ((w, ref["w"]), (z, ref["z"]), (y, ref["y"])):
        assert np.abs(a - b).max() <= TOL * max(1.0, np.abs(b).max())


def test_set_rho_keeps_the_multiplier(gpu):
    """admm_set_rho mid-run = refactor + y *= rho_old / rho_new (lambda = rho y unchanged)."""
    p = pkg.random_ltv(N=30, n=6, m=3, batch=9, seed=41)
    r1, r2 = 0.3, 0.9
    a = oc.solve(p, rho=r1, max_iter=15, stop=False)
    b = oc.solve(p, rho=r2, max_iter=11, stop=False, z0=a["z"], y0=a["y"] * (r1 / r2))
    with pkg.Solver(p, pkg.Options(rho=r1)) as s:
        s.iterate(15)
        s.set_rho(r2)
        _, z_mid, y_mid = s.get(False, True, True)
        s.iterate(11)
        w, z, y = s.get()
    assert np.abs(z_mid - a["z"]).max() <= TOL and np.abs(y_mid - a["y"] * (r1 / r2)).max() <= TOL
    # w read AFTER the rho change is still the w of the last x-update (it is re-materialised from
    # d / t_in / x_in, which belong to the old records)
    with pkg.Solver(p, pkg.Options(rho=r1)) as s2:
        s2.iterate(15)
        s2.set_rho(r2)
        w_after, _, _ = s2.get(True, False, False)
    assert np.abs(w_after - a["w"]).max() <= TOL * max(1.0, np.abs(a["w"]).max())
    for got, ref in ((w, b["w"]), (z, b["z"]), (y, b["y"])):
        assert np.abs(got - ref).max() <= TOL * max(1.0, np.abs(ref).max())


SOC_CASES = [
    (lambda: pkg.cw_rendezvous(N=150, batch=70, thrust_norm=True), 0.05, 1.0),
    (lambda: pkg.random_ltv(N=23, n=6, m=3, batch=5, seed=3, thrust_norm=True), 0.4, 1.0),
    (lambda: pkg.random_ltv(N=31, n=4, m=2, batch=66, seed=4, thrust_norm=True), 0.3, 1.5),
    (lambda: pkg.random_ltv(N=12, n=12, m=6, batch=3, seed=5, thrust_norm=True), 0.5, 1.0),
    (lambda: pkg.random_ltv(N=9, n=3, m=1, batch=4, seed=6, thrust_norm=True), 0.5, 1.2),   # m = 1: |u| <= ub
]


@pytest.mark.parametrize("flags", [0, 2], ids=["fused", "unfused"])
@pytest.mark.parametrize("idx", range(len(SOC_CASES)))
def test_thrust_magnitude_constraint(gpu, idx, flags):
    """||u_k||_2 <= unorm_k replaces the control box (projection = radial scaling).
    Iterates, residuals and the full solve against the oracle, fused (register-resident blocks) and
    unfused (block-structured standalone kernels) paths."""
    make, rho, alpha = SOC_CASES[idx]
    p = make()
    with pkg.Solver(p, pkg.Options(rho=rho, alpha=alpha, flags=flags)) as s:
        done = 0
        for upto in (1, 2, 9, 30):
            s.run(upto - done, residual_every=3)
            done = upto
            w, z, y = s.get()
            ref = oc.solve(p, rho=rho, alpha=alpha, max_iter=upto, check_interval=3, eps_abs=0, eps_rel=0, stop=False)
            for a, b in ((w, ref["w"]), (z, ref["z"]), (y, ref["y"])):
                assert np.abs(a - b).max() <= TOL * max(1.0, np.abs(b).max()), (upto,)
        r, sd, *_ = s.residuals()
        assert np.abs(r - ref["r"]).max() <= 1e-10 and np.abs(sd - ref["s"]).max() <= 1e-10
    un = ar.expand_unorm(p.unorm, p.N)
    nr = np.linalg.norm(z.reshape(p.batch, p.N, p.nb)[:, :, :p.m], axis=2)
    assert (nr <= un[None] * (1 + 1e-14)).all()
    kw = dict(rho=rho, alpha=alpha, eps_abs=1e-6, eps_rel=1e-6, max_iter=1500, check_interval=10)
    ref = oc.solve(p, **kw)
    with pkg.Solver(p, pkg.Options(flags=flags, **kw)) as s:
        info = s.solve()
        _, z, _ = s.get(False, True, False)
    assert info.iters_run == ref["iters_run"] and info.n_converged == int(ref["status"].sum())
    assert np.abs(z - ref["z"]).max() <= TOL * max(1.0, np.abs(ref["z"]).max())


def test_update_instances(gpu):
    """New x0 on an existing handle = fresh setup with that x0."""
    p = pkg.cw_rendezvous(N=80, batch=10)
    p2 = pkg.cw_rendezvous(N=80, batch=10, seed0=999)
    with pkg.Solver(p, pkg.Options(rho=0.05)) as s:
        s.iterate(5)
        s.update_instances(x0=p2.x0)
        s.set_state(z=np.zeros((10, p.L)), y=np.zeros((10, p.L)))
        s.iterate(12)
        _, z, _ = s.get()
    ref = oc.solve(p2, rho=0.05, max_iter=12, stop=False)
    assert np.abs(z - ref["z"]).max() <= TOL


def test_properties_at_full_size(gpu):
    """BASELINE.json configs[2] shape (N=1000, n=6, m=3, batch=4096): size-independent
    properties instead of a full oracle run.
      - dynamics feasibility of w: x_{k+1} = A x_k + B u_k to rounding;
      - z inside the box, y complementary (y_i != 0 only where z_i sits on a bound);
      - z/dual step is idempotent on its own fixed point structure: v = w + y = z+ + y+;
      - a 64-QP slice matches the oracle run on that slice alone (QPs are independent)."""
    p = pkg.cw_rendezvous(N=1000, batch=4096)
    rho = 0.05
    with pkg.Solver(p, pkg.Options(rho=rho)) as s:
        s.iterate(8)
        w0, z0, y0 = s.get()
        s.step_x()
        w, _, _ = s.get(True, False, False)
        s.step_z(residuals=True)
        _, z, y = s.get(False, True, True)
        r, sd, nw, nz, ny = s.residuals()
    A, B = p.A, p.B
    wb = w.reshape(p.batch, p.N, 9)
    xprev = np.concatenate([p.x0[:, None, :], wb[:, :-1, 3:]], axis=1)
    dyn = wb[:, :, 3:] - (xprev @ A.T + wb[:, :, :3] @ B.T)
    assert np.abs(dyn).max() < 1e-12 * max(1.0, np.abs(wb).max())
    lo, hi = ar.expand_bounds(p.lo, p.hi, p.N, p.nb)
    assert (z >= lo).all() and (z <= hi).all()
    on_bound = (z == lo) | (z == hi)
    assert (y[~on_bound] == 0).all()
    assert np.abs((z + y) - (w + y0)).max() < 1e-14 * max(1.0, np.abs(w).max())   # z+ + y+ = w + y
    assert np.abs(r - np.sqrt(((w - z) ** 2).sum(1))).max() < 1e-10
    assert np.abs(sd - rho * np.sqrt(((z - z0) ** 2).sum(1))).max() < 1e-10
    sl = slice(2048 - 32, 2048 + 32)
    ref = oc.solve(p.slice(sl.start, sl.stop), rho=rho, max_iter=9, stop=False)
    assert np.abs(z[sl] - ref["z"]).max() <= TOL and np.abs(y[sl] - ref["y"]).max() <= TOL
    assert np.abs(w[sl] - ref["w"]).max() <= TOL * max(1.0, np.abs(ref["w"]).max())


def test_errors_are_loud(gpu):
    p = pkg.random_ltv(N=5, n=11, m=5, batch=2)
    with pytest.raises(pkg.AdmmError) as e:
        pkg.Solver(p)
    assert e.value.code == 2            # ADMM_ERR_UNSUPPORTED: (11, 5) not compiled
    p = pkg.double_integrator(N=10)
    with pytest.raises(pkg.AdmmError) as e:
        pkg.Solver(p, pkg.Options(rho=-1.0))
    assert e.value.code == 1


def test_randomised_configurations(gpu):
    """40 seeded random configurations over the compiled (n, m) set: horizon, batch (pad columns),
    segment count (incl. S = 1 and S = N), prefetch-ring tails, LDS record chunking, q on/off, box
    per stage or shared, over-relaxation, fused / unfused / chain-scan paths, residual cadence --
    8 iterations each against the C oracle."""
    dims = [(1, 1), (2, 1), (2, 2), (3, 2), (4, 1), (4, 3), (5, 2), (6, 1), (6, 3), (6, 4), (7, 3), (8, 2),
            (8, 4), (9, 3), (10, 2), (12, 4), (12, 6)]
    rng = np.random.default_rng(2024)
    for trial in range(40):
        n, m = dims[rng.integers(len(dims))]
        N = int(rng.integers(1, 90))
        batch = int(rng.choice([1, 2, 63, 64, 65, 100, 129, 257]))
        segs = int(rng.choice([0, 1, 2, 3, 5, 8, N]))
        alpha = float(rng.choice([1.0, 1.0, 1.4]))
        flags = int(rng.choice([0, 0, 2, 4, 6]))
        with_q = bool(rng.integers(2))
        p = pkg.random_ltv(N=N, n=n, m=m, batch=batch, seed=1000 + trial, with_q=with_q,
                           state_bounds=bool(rng.integers(2)))
        if rng.integers(2):                       # one box for every stage
            p.lo, p.hi = p.lo[0].copy(), p.hi[0].copy()
        rho = float(rng.choice([0.1, 0.5, 2.0]))
        every = int(rng.choice([0, 1, 3]))
        ref = oc.solve(p, rho=rho, alpha=alpha, max_iter=8, check_interval=1, eps_abs=0, eps_rel=0, stop=False)
        with pkg.Solver(p, pkg.Options(rho=rho, alpha=alpha, segments=segs, flags=flags)) as s:
            s.run(8, residual_every=every)
            w, z, y = s.get()
        ctx = dict(trial=trial, n=n, m=m, N=N, batch=batch, segs=segs, alpha=alpha, flags=flags, q=with_q)
        for a, b in ((w, ref["w"]), (z, ref["z"]), (y, ref["y"])):
            assert np.abs(a - b).max() <= TOL * max(1.0, np.abs(b).max()), ctx


@pytest.mark.parametrize("flags", [0, 32, 8, 2], ids=["default", "one_lane", "fused_plain", "unfused"])
@pytest.mark.parametrize("shape", [(148, 1, 1, 300), (300, 2, 2, 64), (200, 3, 1, 5)])
def test_tiny_blocks_with_segments_longer_than_one_record_chunk(gpu, shape, flags):
    """Regression (found by tools/stress_alt.py): for tiny (n, m) the per-stage records are so small
    that the LDS chunk hits its 128-stage cap, which was not a multiple of the 3-deep prefetch ring --
    segments longer than 128 stages then read misaligned ring slots (errors of 1e-5)."""
    N, n, m, batch = shape
    p = pkg.random_ltv(N=N, n=n, m=m, batch=batch, seed=3238, with_q=False, state_bounds=False)
    with pkg.Solver(p, pkg.Options(rho=0.1, segments=1, flags=flags)) as s:
        s.iterate(5)
        w, z, y = s.get()
    ref = oc.solve(p, rho=0.1, max_iter=5, stop=False)
    for a, b in ((w, ref["w"]), (z, ref["z"]), (y, ref["y"])):
        assert np.abs(a - b).max() <= TOL * max(1.0, np.abs(b).max())


def test_plain_c_program_over_the_abi(gpu, tmp_path):
    """examples/c_abi_demo.c: the boundary really is a C ABI -- a C11 program compiled with gcc,
    linked against libadmm_hip.so only, sets up, solves and reads back a batch."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "c_abi_demo")
    libdir = os.path.join(root, "admm-library_amd")
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
                    os.path.join(root, "examples", "c_abi_demo.c"), "-L", libdir, "-ladmm_hip", "-lm", "-o", exe], check=True)
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "converged 5/5" in out.stdout


@pytest.mark.parametrize("batch", [65, 130, 200])
def test_clamped_lanes_never_store(gpu, batch):
    """pitch not a multiple of the 256-column workgroup: the lanes past the pitch are clamped
    onto the last column for loads and must not store (v is updated in place, so a duplicate of the
    last column running ahead would corrupt it).  120 iterations of a long horizon, last columns
    included, against the oracle."""
    p = pkg.cw_rendezvous(N=300, batch=batch)
    ref = oc.solve(p, rho=0.05, max_iter=120, stop=False)
    with pkg.Solver(p, pkg.Options(rho=0.05, segments=4)) as s:      # long segments: many stages for a wave to run ahead
        s.iterate(120)
        w, z, y = s.get()
    for a, b in ((w, ref["w"]), (z, ref["z"]), (y, ref["y"])):
        assert np.abs(a - b).max() <= TOL * max(1.0, np.abs(b).max())
    assert np.abs(z[-1] - ref["z"][-1]).max() <= TOL          # the column the clamped lanes alias


@pytest.mark.parametrize("tau", [2.0, 3.0], ids=["tau2", "tau3"])
def test_background_refactor_is_the_synchronous_one(gpu, tau, monkeypatch):
    """The adaptive rule's rho changes are served by factorisations run ahead on host threads (and by the kept factor of
    the rho just left): same iterates, BIT for bit, same rho trajectory and counts as with ADMM_NO_SPECULATE (refactor
    on demand) -- tau = 3 makes (rho tau) / tau differ from rho in the last bit, so a kept factor must not be reused."""
    p = pkg.cw_rendezvous(N=200, batch=12)
    kw = dict(rho=0.05, eps_abs=1e-6, eps_rel=1e-6, max_iter=3000, check_interval=10, adapt_interval=20, adapt_tau=tau)
    out = []
    for spec in (True, False):
        if spec:
            monkeypatch.delenv("ADMM_NO_SPECULATE", raising=False)
        else:
            monkeypatch.setenv("ADMM_NO_SPECULATE", "1")
        with pkg.Solver(p, pkg.Options(**kw)) as s:
            info = s.solve()
            zero = np.zeros((p.batch, p.L))
            info2 = s.solve(z0=zero, y0=zero)          # second solve on the handle: starts from the rho the first ended at
            out.append((info.iters_run, info.rho_updates, info.rho, info2.iters_run, info2.rho_updates, info2.rho) + s.get())
    assert out[0][1] >= 2
    assert out[0][:6] == out[1][:6]
    for a, b in zip(out[0][6:], out[1][6:]):
        np.testing.assert_array_equal(a, b)
    ref = oc.solve(p, **kw)
    assert out[0][0] == ref["iters_run"] and out[0][1] == ref["rho_updates"] and out[0][2] == ref["rho"]


def test_background_refactor_survives_problem_updates_and_early_release(gpu):
    """A handle whose background factorisations are still running is (a) given new problem data -- they are joined and
    dropped, the solve of the new problem equals the oracle's -- and (b) released right after admm_solve_begin."""
    p1 = pkg.cw_formation(N=400, batch=3)
    p2 = pkg.cw_formation(N=400, batch=3, u_max=0.15)
    kw = dict(rho=0.05, eps_abs=1e-6, eps_rel=1e-6, max_iter=600, check_interval=10, adapt_interval=20)
    ref = oc.solve(p2, **kw)
    with pkg.Solver(p1, pkg.Options(**kw)) as s:
        s.solve_begin()                      # starts the two candidate factorisations
        s.update_problem(p2)                 # ... which read the problem copy this call replaces
        zero = np.zeros((p2.batch, p2.L))
        info = s.solve(z0=zero, y0=zero)
        w, z, y = s.get()
    assert info.iters_run == ref["iters_run"] and info.rho == ref["rho"] and info.rho_updates == ref["rho_updates"]
    for a, b in ((w, ref["w"]), (z, ref["z"]), (y, ref["y"])):
        assert np.abs(a - b).max() <= TOL * max(1.0, np.abs(b).max())
    s = pkg.Solver(p1, pkg.Options(**kw))
    s.solve_begin()
    s.close()


def test_residual_history(gpu):
    """ADMM_FLAG_HISTORY (SURVEY.md §5 "optional residual history buffer"): one record per stopping test of admm_solve --
    iteration, converged count, the batch maxima of r and s, rho in force -- checked against the oracle run to that iteration."""
    p = pkg.cw_rendezvous(N=150, batch=9)
    kw = dict(rho=0.05, eps_abs=1e-6, eps_rel=1e-6, max_iter=1200, check_interval=10, adapt_interval=50)
    with pkg.Solver(p, pkg.Options(flags=64, **kw)) as s:           # ADMM_FLAG_HISTORY
        info = s.solve()
        h = s.history()
        n = len(h["iteration"])
        assert n == -(-int(info.iters_run) // 10) and n >= 5
        np.testing.assert_array_equal(h["iteration"][:-1], 10 * np.arange(1, n))
        assert int(h["iteration"][-1]) == int(info.iters_run) and int(h["n_converged"][-1]) == int(info.n_converged)
        assert (np.diff(h["n_converged"]) >= 0).all()
        assert abs(h["max_r"][-1] - info.max_r) <= 1e-15 and abs(h["max_s"][-1] - info.max_s) <= 1e-15
        assert float(h["rho"][0]) == 0.05 and float(info.rho) in set(h["rho"].tolist()) | {float(info.rho)}
        for k in (0, 3, n - 2):
            it = int(h["iteration"][k])
            ref = oc.solve(p, **{**kw, "max_iter": it}, stop=False)
            assert abs(h["max_r"][k] - ref["r"].max()) <= 1e-10 and abs(h["max_s"][k] - ref["s"].max()) <= 1e-10
            assert int(h["n_converged"][k]) == int(ref["status"].sum())
        s.solve()                                   # a second solve starts its own history
        assert int(s.history()["iteration"][0]) == 10
    with pkg.Solver(p, pkg.Options(**kw)) as s:       # off by default: nothing recorded
        s.solve()
        assert len(s.history()["iteration"]) == 0

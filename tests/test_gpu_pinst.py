"""Per-instance dynamics (admm_problem.time_varying = 2, stage_bounds = 2): every QP has its own
A_k, B_k and box; the KKT factor is computed on the device.  HIP path through the C ABI against the C oracle applied
QP by QP (oracle_c._solve_per_instance).  Tolerance 1e-10 on fp64 iterates.  PARITY UNPINNED."""
import dataclasses

import numpy as np
import pytest

import admm_library_amd as pkg
import oracle_c as oc
from admm_library_amd import _abi

pytestmark = pytest.mark.gpu
TOL = 1e-10

CASES = [
    dict(N=25, n=6, m=3, batch=70, seed=3),                                   # the SCvx model's shape, two waves
    dict(N=25, n=6, m=3, batch=9, seed=4, instance_bounds=False),             # box shared by the batch
    dict(N=40, n=2, m=1, batch=5, seed=5, with_q=False),
    dict(N=12, n=4, m=2, batch=130, seed=6),
    dict(N=7, n=3, m=2, batch=3, seed=7),
    dict(N=1, n=6, m=3, batch=2, seed=8),
    dict(N=30, n=6, m=4, batch=67, seed=9),                                   # the shapes of csrc/admm_pinst_g1.hip
    dict(N=33, n=6, m=2, batch=5, seed=10),
    dict(N=20, n=6, m=1, batch=65, seed=11, with_q=False),
    dict(N=50, n=4, m=1, batch=3, seed=12),
    dict(N=17, n=2, m=2, batch=130, seed=13, instance_bounds=False),
    dict(N=60, n=1, m=1, batch=64, seed=14),
]


def _close(got, ref):
    return all(np.abs(a - ref[k]).max() <= TOL * max(1.0, np.abs(ref[k]).max()) for a, k in zip(got, ("w", "z", "y")))


@pytest.mark.parametrize("form", ["auto", "lane_per_qp", "rows_over_lanes"])
@pytest.mark.parametrize("segments", [0, 1, 5], ids=["auto_segments", "one_segment", "five_segments"])
@pytest.mark.parametrize("alpha", [1.0, 1.5])
@pytest.mark.parametrize("idx", range(len(CASES)))
def test_iterates_match_the_oracle(gpu, idx, alpha, segments, form, monkeypatch):
    """segments: 0 = the library's choice (N / 8 at these sizes), 1 = one lane sweeps the whole horizon, 5 = segments in
    time with per-QP transfer matrices computed on the device (pseg_kernel, pscan_kernel; capped at N).
    form: the sweeps with one lane per QP (csrc/admm_pinst.hpp) or with a QP's rows spread over lanes
    (csrc/admm_pinst_rows.hpp; the library's choice for batches of up to 64 QPs), forced either way."""
    monkeypatch.delenv("ADMM_PI_LANE_PER_QP", raising=False)
    monkeypatch.delenv("ADMM_PI_ROWS", raising=False)
    if form == "lane_per_qp":
        monkeypatch.setenv("ADMM_PI_LANE_PER_QP", "1")
    elif form == "rows_over_lanes":
        monkeypatch.setenv("ADMM_PI_ROWS", "1")
    p = pkg.random_instances(**CASES[idx])
    with pkg.Solver(p, pkg.Options(rho=0.3, alpha=alpha, segments=segments)) as s:
        want = {0: max(1, min(64, p.N // 8)), 1: 1, 5: min(5, p.N)}[segments]
        assert s.geometry()["segments"] == want
        done = 0
        for upto in (1, 2, 3, 10, 40):
            s.run(upto - done, residual_every=2)
            done = upto
            ref = oc.solve(p, rho=0.3, alpha=alpha, max_iter=upto, stop=False)
            assert _close(s.get(), ref), upto


def test_factor_kernel_matches_the_oracle_factor(gpu):
    """One x-update from a random (z, y): w = the oracle's Riccati solve of every QP with ITS dynamics."""
    import admm_ref as ar
    p = pkg.random_instances(N=18, n=6, m=3, batch=11, seed=12)
    rng = np.random.default_rng(1)
    z, y = rng.standard_normal((p.batch, p.L)), rng.standard_normal((p.batch, p.L))
    with pkg.Solver(p, pkg.Options(rho=0.7)) as s:
        s.set_state(z=z, y=y)
        s.step_x()
        w = s.get()[0]
    for b in range(p.batch):
        f = ar.factor(p.A[b], p.B[b], p.Q, p.R, p.QN, 0.7, p.N)
        g = -0.7 * (z[b] - y[b]) + p.q[b]
        w_ref = ar.x_update(f, g[None], p.x0[b:b + 1])[0]
        assert np.abs(w[b] - w_ref).max() <= 1e-11 * max(1.0, np.abs(w_ref).max())


def test_solve_set_rho_update_problem(gpu):
    p = pkg.random_instances(N=20, n=6, m=3, batch=66, seed=21)
    kw = dict(rho=0.3, eps_abs=1e-7, eps_rel=1e-7, max_iter=3000, check_interval=10)
    ref = oc.solve(p, **kw)
    with pkg.Solver(p, pkg.Options(**kw)) as s:
        info = s.solve()
        assert int(info.iters_run) == ref["iters_run"]
        np.testing.assert_array_equal(info.status, ref["status"])
        assert (np.abs(info.iters - ref["iters"]) <= 10).all()
        assert _close(s.get(), ref)
        # a rho change refactors every QP on the device; the scaled dual is rescaled
        s.set_rho(0.9)
        s.iterate(7)
        got = s.get()
        z0, y0 = ref["z"], ref["y"] * (0.3 / 0.9)
        ref2 = oc.solve(p, rho=0.9, max_iter=7, stop=False, z0=z0, y0=y0)
        assert _close(got, ref2)
        # new dynamics / bounds / x0 / q on the same handle = a fresh handle
        p2 = pkg.random_instances(N=20, n=6, m=3, batch=66, seed=22)
        s.update_problem(p2)
        s.set_state(z=np.zeros((66, p2.L)), y=np.zeros((66, p2.L)))
        s.iterate(12)
        assert _close(s.get(), oc.solve(p2, rho=0.9, max_iter=12, stop=False))


def test_validation_and_unsupported(gpu):
    p = pkg.random_instances(N=10, n=6, m=3, batch=4, seed=1)
    inv = {v: k for k, v in _abi.STATUS_NAMES.items()}
    bad = dataclasses.replace(p, A=p.A.copy())
    bad.A[2, 3, 1, 1] = np.nan
    with pytest.raises(ValueError):
        pkg.Solver(bad, pkg.Options(rho=0.3))
    with pytest.raises(pkg.AdmmError) as e:
        pkg.Solver(pkg.random_instances(N=10, n=5, m=3, batch=4, seed=1), pkg.Options(rho=0.3))       # no kernel for (5, 3)
    assert e.value.code == inv["ADMM_ERR_UNSUPPORTED"]
    with pytest.raises(pkg.AdmmError) as e:
        pkg.Solver(p, pkg.Options(rho=0.3, precision_mode=_abi.PRECISION_MIXED))
    assert e.value.code == inv["ADMM_ERR_UNSUPPORTED"]
    # an indefinite R makes S_k indefinite for rho small: the device factorisation reports it
    ind = dataclasses.replace(p, R=-0.5 * np.eye(3))
    with pytest.raises(pkg.AdmmError) as e:
        pkg.Solver(ind, pkg.Options(rho=0.01))
    assert e.value.code == inv["ADMM_ERR_NUMERIC"]


@pytest.mark.parametrize("case", [dict(N=30, n=6, m=3, batch=70, seed=31), dict(N=25, n=4, m=2, batch=9, seed=32, instance_bounds=False)],
                         ids=["6_3_batch70", "4_2_shared_box"])
@pytest.mark.parametrize("alpha", [1.0, 1.6])
def test_adaptive_rho_is_per_qp(gpu, case, alpha):
    """With per-instance dynamics the residual-balancing rule runs QP by QP on the device (padapt_kernel: each QP's own
    rho, dual rescale and refactor).  Against the one-QP oracle with the same options applied QP by QP: every QP's rho
    trajectory end point and change count, its first-converged iteration, and the iterates."""
    p = pkg.random_instances(**case)
    kw = dict(rho=0.3, alpha=alpha, eps_abs=1e-7, eps_rel=1e-7, max_iter=2000, check_interval=10, adapt_interval=20,
              adapt_mu=1.5, adapt_tau=2.0, adapt_max=8)
    ref = oc.solve(p, **kw)
    assert len(set(ref["rho"].tolist())) >= 3 and ref["rho_updates"].max() >= 3       # the QPs really end at different rho
    with pkg.Solver(p, pkg.Options(**kw)) as s:
        info = s.solve()
        rho = s.rho_per_qp()
        got = s.get()
        np.testing.assert_array_equal(rho, ref["rho"])
        assert int(info.iters_run) == ref["iters_run"]
        assert int(info.rho_updates) == int(ref["rho_updates"].sum()) and float(info.rho) == float(ref["rho"].max())
        np.testing.assert_array_equal(info.status, ref["status"])
        np.testing.assert_array_equal(info.iters, ref["iters"])
        assert _close(got, ref)
        # admm_set_rho puts every QP back on one rho (each dual rescaled by its own rho_b / rho_new)
        s.set_rho(0.5)
        np.testing.assert_array_equal(s.rho_per_qp(), np.full(p.batch, 0.5))
        s.iterate(5)
        y0 = ref["y"] * (ref["rho"] / 0.5)[:, None]
        ref2 = oc.solve(p, rho=0.5, alpha=alpha, max_iter=5, stop=False, z0=ref["z"], y0=y0)
        assert _close(s.get(), ref2)


def test_adaptive_rho_per_qp_through_the_pieces_api(gpu):
    """admm_solve_begin / _step / _adapt / _end (the sharded solve's loop): the per-QP rule ignores the batch sums R, S."""
    p = pkg.random_instances(N=20, n=6, m=3, batch=12, seed=33)
    kw = dict(rho=0.3, eps_abs=1e-7, eps_rel=1e-7, max_iter=1500, check_interval=10, adapt_interval=20, adapt_mu=1.5, adapt_max=8)
    ref = oc.solve(p, **kw)
    with pkg.Solver(p, pkg.Options(**kw)) as s:
        s.solve_begin()
        while True:
            it, nconv, _, _ = s.solve_step()
            if nconv == p.batch or it >= kw["max_iter"]:
                break
            s.solve_adapt(float("nan"), float("nan"))
        info = s.solve_end()
        np.testing.assert_array_equal(s.rho_per_qp(), ref["rho"])
        assert int(info.iters_run) == ref["iters_run"]
        assert _close(s.get(), ref)


@pytest.mark.parametrize("case", [dict(N=30, n=6, m=3, batch=70, seed=61), dict(N=24, n=6, m=3, batch=9, seed=62, instance_bounds=False),
                                  dict(N=20, n=4, m=2, batch=5, seed=63, with_q=False), dict(N=16, n=6, m=4, batch=65, seed=64)],
                         ids=["6_3_per_qp_box", "6_3_shared_box", "4_2", "6_4"])
@pytest.mark.parametrize("alpha", [1.0, 1.6])
@pytest.mark.parametrize("segments", [0, 1])
def test_thrust_magnitude_bound_with_per_instance_dynamics(gpu, case, alpha, segments):
    """||u_k||_2 <= ub_k on per-instance problems: per-stage bounds on most stages (shared by the batch),
    the box on the rest, every QP with its own dynamics -- iterates and residuals against the one-QP oracle, a solve, the
    read-out of (z, y) from v, and a rho change."""
    p = pkg.random_instances(thrust_norm=True, **case)
    assert p.unorm is not None and np.isfinite(p.unorm).any() and np.isinf(p.unorm).any()
    with pkg.Solver(p, pkg.Options(rho=0.3, alpha=alpha, segments=segments)) as s:
        done = 0
        for upto in (1, 2, 9, 30):
            s.run(upto - done, residual_every=3)
            done = upto
            ref = oc.solve(p, rho=0.3, alpha=alpha, max_iter=upto, check_interval=3, stop=False)
            assert _close(s.get(), ref), upto
        r, sd = s.residuals()[:2]
        assert np.abs(r - ref["r"]).max() <= 1e-10 and np.abs(sd - ref["s"]).max() <= 1e-10
        # the thrust really is on its bound somewhere, and never beyond it
        z = s.get(False, True, False)[1].reshape(p.batch, p.N, p.n + p.m)
        nrm = np.linalg.norm(z[:, :, :p.m], axis=2)
        fin = np.isfinite(p.unorm)
        assert (nrm[:, fin] <= p.unorm[fin] * (1 + 1e-12)).all() and (nrm[:, fin] >= p.unorm[fin] * (1 - 1e-12)).any()
        s.set_rho(0.9)
        s.iterate(4)
        ref2 = oc.solve(p, rho=0.9, alpha=alpha, max_iter=4, stop=False, z0=ref["z"], y0=ref["y"] * (0.3 / 0.9))
        assert _close(s.get(), ref2)
    kw = dict(rho=0.3, alpha=alpha, eps_abs=1e-7, eps_rel=1e-7, max_iter=1500, check_interval=10, adapt_interval=20, adapt_mu=2.0)
    ref = oc.solve(p, **kw)
    with pkg.Solver(p, pkg.Options(segments=segments, **kw)) as s:
        info = s.solve()
        assert int(info.iters_run) == ref["iters_run"]
        np.testing.assert_array_equal(s.rho_per_qp(), ref["rho"])
        assert _close(s.get(), ref)


# ---- wide shapes ((8, 4), (12, 6), ...): every kernel with a QP's rows spread over the lanes (csrc/admm_pinst_wide.hpp) ----
WIDE_CASES = [
    dict(N=25, n=12, m=6, batch=70, seed=41),                                 # configs[4]'s shape, more than 64 QPs: 4-wave workgroups
    dict(N=24, n=12, m=6, batch=9, seed=42, instance_bounds=False),
    dict(N=16, n=12, m=6, batch=3, seed=43, with_q=False),
    dict(N=1, n=12, m=6, batch=2, seed=44),
    dict(N=30, n=8, m=4, batch=67, seed=45),
    dict(N=21, n=8, m=4, batch=5, seed=46, instance_bounds=False),
    dict(N=20, n=12, m=3, batch=65, seed=47),
    dict(N=33, n=9, m=3, batch=12, seed=48, with_q=False),
    # thrust-magnitude bound at the wide shapes (the norm over a QP's control rows is read across the lanes)
    dict(N=26, n=12, m=6, batch=70, seed=49, thrust_norm=True),
    dict(N=24, n=12, m=6, batch=9, seed=50, thrust_norm=True, instance_bounds=False, with_q=False),
    dict(N=22, n=8, m=4, batch=66, seed=51, thrust_norm=True),
    dict(N=21, n=9, m=3, batch=5, seed=52, thrust_norm=True),
]


@pytest.mark.parametrize("segments", [0, 1, 5], ids=["auto_segments", "one_segment", "five_segments"])
@pytest.mark.parametrize("alpha", [1.0, 1.5])
@pytest.mark.parametrize("idx", range(len(WIDE_CASES)))
def test_wide_shapes_match_the_oracle(gpu, idx, alpha, segments, monkeypatch):
    """Device factorisation (pfactor_rows_kernel), per-QP transfer matrices (pseg_rows_kernel), scan and both sweeps at the
    shapes one lane cannot hold: iterates against the C oracle applied QP by QP."""
    monkeypatch.delenv("ADMM_PI_LANE_PER_QP", raising=False)
    monkeypatch.delenv("ADMM_PI_ROWS", raising=False)
    p = pkg.random_instances(**WIDE_CASES[idx])
    with pkg.Solver(p, pkg.Options(rho=0.3, alpha=alpha, segments=segments)) as s:
        want = {0: max(1, min(64, p.N // 8)), 1: 1, 5: min(5, p.N)}[segments]
        assert s.geometry()["segments"] == want
        done = 0
        for upto in (1, 2, 3, 10, 40):
            s.run(upto - done, residual_every=2)
            done = upto
            ref = oc.solve(p, rho=0.3, alpha=alpha, max_iter=upto, stop=False)
            assert _close(s.get(), ref), upto


@pytest.mark.parametrize("segments", [1, 4])
@pytest.mark.parametrize("batch", [9, 70])
def test_rows_factorisation_agrees_with_the_one_lane_factorisation(gpu, batch, segments, monkeypatch):
    """The wide shapes' factor / transfer-matrix kernels instantiated at (6, 3), where the one-lane kernels exist: the same
    recursion in the same summation order -- the iterates through either factor agree to rounding (observed <= 1e-15 absolute
    after 25 iterations; the two compilations differ by an ulp in rare entries), far inside the tolerance against the oracle."""
    p = pkg.random_instances(N=26, n=6, m=3, batch=batch, seed=51)
    out = {}
    for twin in (False, True):
        monkeypatch.delenv("ADMM_PI_ROWS_FACTOR", raising=False)
        if twin:
            monkeypatch.setenv("ADMM_PI_ROWS_FACTOR", "1")
        with pkg.Solver(p, pkg.Options(rho=0.3, segments=segments)) as s:
            s.run(25, residual_every=5)
            out[twin] = s.get() + (s.residuals(),)
            s.set_rho(0.8)                       # a refactorisation (on trial first), then more iterations
            s.run(10, residual_every=5)
            out[twin] += s.get()
    for a, b in zip(out[False], out[True]):
        a, b = np.asarray(a), np.asarray(b)
        assert np.abs(a - b).max() <= 1e-13 * max(1.0, np.abs(a).max())


def test_wide_shape_solve_set_rho_update_problem_and_per_qp_rho(gpu):
    p = pkg.random_instances(N=20, n=12, m=6, batch=66, seed=53)
    kw = dict(rho=0.3, eps_abs=1e-7, eps_rel=1e-7, max_iter=3000, check_interval=10)
    ref = oc.solve(p, **kw)
    with pkg.Solver(p, pkg.Options(**kw)) as s:
        assert s.path()["per_instance"]
        info = s.solve()
        assert int(info.iters_run) == ref["iters_run"]
        np.testing.assert_array_equal(info.status, ref["status"])
        assert _close(s.get(), ref)
        s.set_rho(0.9)
        s.iterate(7)
        ref2 = oc.solve(p, rho=0.9, max_iter=7, stop=False, z0=ref["z"], y0=ref["y"] * (0.3 / 0.9))
        assert _close(s.get(), ref2)
        p2 = pkg.random_instances(N=20, n=12, m=6, batch=66, seed=54)
        s.update_problem(p2)
        s.set_state(z=np.zeros((66, p2.L)), y=np.zeros((66, p2.L)))
        s.iterate(12)
        assert _close(s.get(), oc.solve(p2, rho=0.9, max_iter=12, stop=False))
    # the per-QP adaptive rule (masked trial refactorisation of the QPs whose rho has moved)
    kw = dict(rho=0.3, eps_abs=1e-7, eps_rel=1e-7, max_iter=2000, check_interval=10, adapt_interval=20, adapt_mu=1.5, adapt_tau=2.0,
              adapt_max=8)
    ref = oc.solve(p, **kw)
    assert ref["rho_updates"].max() >= 1
    with pkg.Solver(p, pkg.Options(**kw)) as s:
        info = s.solve()
        np.testing.assert_array_equal(s.rho_per_qp(), ref["rho"])
        assert int(info.iters_run) == ref["iters_run"]
        np.testing.assert_array_equal(info.status, ref["status"])
        np.testing.assert_array_equal(info.iters, ref["iters"])
        assert _close(s.get(), ref)


def test_wide_shape_failures_are_reported(gpu):
    inv = {v: k for k, v in _abi.STATUS_NAMES.items()}
    p = pkg.random_instances(N=10, n=12, m=6, batch=4, seed=55)
    ind = dataclasses.replace(p, R=-0.5 * np.eye(6))
    with pytest.raises(pkg.AdmmError) as e:
        pkg.Solver(ind, pkg.Options(rho=0.01))
    assert e.value.code == inv["ADMM_ERR_NUMERIC"]


@pytest.mark.parametrize("shape", [(6, 3), (12, 6), (8, 4)], ids=["n6_batch_minor", "n12_tiled", "n8_tiled"])
def test_row_major_blocks_are_the_column_major_blocks(gpu, shape, monkeypatch):
    """ADMM_FLAG_ROW_MAJOR (ABI v8; what the Python wrapper uses for per-instance dynamics): A, B handed over in NumPy's own order and
    transposed on the device on the way into the library's layout -- bit for bit the handle that received host-transposed, column-major
    blocks (ADMM_PY_COLMAJOR=1), at set-up and through admm_update_problem."""
    n, m = shape
    p = pkg.random_instances(N=17, n=n, m=m, batch=70, seed=71)
    p2 = pkg.random_instances(N=17, n=n, m=m, batch=70, seed=72)
    out = {}
    for colmajor in (False, True):
        monkeypatch.delenv("ADMM_PY_COLMAJOR", raising=False)
        if colmajor:
            monkeypatch.setenv("ADMM_PY_COLMAJOR", "1")
        with pkg.Solver(p, pkg.Options(rho=0.3)) as s:
            assert s._row_major == (not colmajor)
            s.run(6, residual_every=2)
            a = s.get()
            s.update_problem(p2)
            s.run(5, residual_every=1)
            out[colmajor] = a + s.get()
    for x, y in zip(out[False], out[True]):
        np.testing.assert_array_equal(x, y)
    assert _close(out[False][:3], oc.solve(p, rho=0.3, max_iter=6, stop=False))
    # the flag is for per-instance dynamics only
    shared = pkg.random_ltv(N=10, n=6, m=3, batch=4, seed=1)
    with pytest.raises(pkg.AdmmError) as e:
        pkg.Solver(shared, pkg.Options(rho=0.3, flags=_abi.FLAG_ROW_MAJOR))
    assert e.value.code == {v: k for k, v in _abi.STATUS_NAMES.items()}["ADMM_ERR_UNSUPPORTED"]


def test_large_wide_handles_one_after_another(gpu):
    """Regression (round 3): set-up zero-filled every device array asynchronously on the handle's non-blocking stream and then copied
    the weights / shared box with synchronous null-stream copies, which do not wait for it -- with GBs of fills queued (n = 12, 1024 x
    1000: the second large handle of a process, whose allocations are fast) a copy landed first and was zeroed afterwards: Q = R = 0 and
    every QP wrong from the first iteration.  Three handles in a row, different segment counts, against the oracle on 32 of the QPs."""
    import dataclasses
    p = pkg.cw_formation_instances(N=1000, batch=1024)
    idx = np.concatenate([np.arange(0, 16), np.arange(p.batch - 16, p.batch)])
    sub = dataclasses.replace(p, A=p.A[idx], B=p.B[idx], x0=p.x0[idx], lo=p.lo[idx], hi=p.hi[idx])
    ref = oc.solve(sub, rho=0.05, max_iter=6, stop=False)
    for segments in (0, 1, 2):
        with pkg.Solver(p, pkg.Options(rho=0.05, segments=segments)) as s:
            s.run(6, residual_every=2)
            w, z, y = s.get()
        assert _close((w[idx], z[idx], y[idx]), ref), segments

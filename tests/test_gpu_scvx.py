"""Successive-convexification outer loop with the HIP solver as its QP solver, against the same
loop driven by the CPU oracle (PARITY UNPINNED: no reference code; tests/test_scvx.py holds the
solver-independent checks).  The QPs have time-varying dynamics, per-stage bounds and a linear
term, batch 1."""
import numpy as np
import pytest

from admm_library_amd import scvx as sc

import _scvx_case as case

pytestmark = pytest.mark.gpu


def test_scvx_on_the_gpu_matches_the_oracle_driven_loop(gpu):
    ref = sc.scvx(case.X0, case.N, case.DT, case.Q, case.R, case.QN, -case.U_MAX, case.U_MAX,
                  qp_solver=case.oracle_qp_solver(**case.QP), **case.SCVX)
    res = sc.scvx(case.X0, case.N, case.DT, case.Q, case.R, case.QN, -case.U_MAX, case.U_MAX,
                  qp_options=case.QP, **case.SCVX)              # default QP solver = libadmm_hip.so
    assert res.converged and res.outer_iterations == ref.outer_iterations and res.accepted == ref.accepted
    # the stop decision may flip by a check near the threshold (reduction order): +- a few check intervals
    for a, b in zip(res.history, ref.history):
        assert abs(a["admm_iterations"] - b["admm_iterations"]) <= 3 * case.QP["check_interval"], (a, b)
    # each QP is solved to eps = 1e-8 on its residuals, not to a solution accuracy, and the cost is flat
    # along some thrust directions: the two loops agree to the QP tolerance, the costs much closer
    assert np.abs(res.u - ref.u).max() <= 1e-2
    assert np.abs(res.x - ref.x).max() <= 5e-2
    assert abs(res.cost - ref.cost) <= 1e-6 * abs(ref.cost)
    np.testing.assert_allclose(res.x, sc.rollout(case.X0, res.u, case.DT), atol=1e-12)


def test_correction_qp_iterates_match_the_oracle(gpu):
    """Parity proper on this problem class (time-varying A_k, B_k, per-stage bounds, linear term,
    batch 1): fixed iteration counts, iterates within 1e-10 of the C oracle."""
    import admm_library_amd as pkg
    import oracle_c as oc
    ub = np.zeros((case.N, 3))
    xb = sc.rollout(case.X0, ub, case.DT)
    p = sc.correction_qp(xb, ub, case.X0, case.DT, case.Q, case.R, case.QN, -case.U_MAX, case.U_MAX, 1.0, 100.0)
    with pkg.Solver(p, pkg.Options(rho=0.5)) as s:
        done = 0
        for upto in (1, 10, 150):
            s.iterate(upto - done)
            done = upto
            w, z, y = s.get()
            ref = oc.solve(p, rho=0.5, max_iter=upto, stop=False)
            for a, b in ((w, ref["w"]), (z, ref["z"]), (y, ref["y"])):
                assert np.abs(a - b).max() <= 1e-10 * max(1.0, np.abs(b).max()), upto


def test_batched_scvx_on_the_gpu(gpu):
    """the outer loop batched over 64 initial conditions -- ONE per-instance QP batch per outer
    iteration on the HIP solver (device-side factorisation of 64 linearisations).  Three of the trajectories are
    checked against the single-trajectory loop driven by the CPU oracle; all must converge and fly the nonlinear plant."""
    rng = np.random.default_rng(11)
    B = 64
    x0s = case.X0[None] * (1.0 + 0.05 * rng.standard_normal((B, 6)))
    res = sc.scvx_batch(x0s, case.N, case.DT, case.Q, case.R, case.QN, -case.U_MAX, case.U_MAX, qp_options=case.QP, **case.SCVX)
    assert len(res) == B and all(r.converged for r in res)
    for r, x0 in zip(res, x0s):
        np.testing.assert_allclose(r.x, sc.rollout(x0, r.u, case.DT), atol=1e-12)
        assert (np.abs(r.u) <= case.U_MAX + 1e-12).all()
    qp = case.oracle_qp_solver(**case.QP)
    for b in (0, 31, 63):
        one = sc.scvx(x0s[b], case.N, case.DT, case.Q, case.R, case.QN, -case.U_MAX, case.U_MAX, qp_solver=qp, **case.SCVX)
        assert res[b].outer_iterations == one.outer_iterations and res[b].accepted == one.accepted
        assert abs(res[b].cost - one.cost) <= 1e-6 * abs(one.cost)
        assert np.abs(res[b].u - one.u).max() <= 1e-2


def test_batched_correction_qp_iterates_match_the_oracle(gpu):
    """Parity proper on the batched loop's QP class: per-instance time-varying dynamics, per-instance per-stage bounds,
    linear term, batch 64: fixed iteration counts, iterates within 1e-10 of the C oracle (applied QP by QP)."""
    import admm_library_amd as pkg
    import oracle_c as oc
    rng = np.random.default_rng(12)
    B = 64
    x0s = case.X0[None] * (1.0 + 0.15 * rng.standard_normal((B, 6)))
    ub = np.zeros((B, case.N, 3))
    xb = sc.rollout(x0s, ub, case.DT)
    p = sc.correction_qp_batch(xb, ub, x0s, case.DT, case.Q, case.R, case.QN, -case.U_MAX, case.U_MAX,
                               np.full(B, 1.0), np.full(B, 100.0))
    with pkg.Solver(p, pkg.Options(rho=0.5)) as s:
        done = 0
        for upto in (1, 10, 150):
            s.iterate(upto - done)
            done = upto
            got = s.get()
            ref = oc.solve(p, rho=0.5, max_iter=upto, stop=False)
            for a, k in zip(got, ("w", "z", "y")):
                assert np.abs(a - ref[k]).max() <= 1e-10 * max(1.0, np.abs(ref[k]).max()), upto

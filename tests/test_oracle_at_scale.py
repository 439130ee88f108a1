"""The oracle pinned WHERE IT IS USED: N = 1000, n = 6 and n = 12 (VERDICT r02, next #1d).

tests/test_oracle.py pins the NumPy oracle at N <= 60 with dense algebra.  Every GPU parity test at BASELINE's sizes
then trusts the C oracle at N = 1000; these tests check it there against routines that share nothing with it
(tests/_indep.py: one banded LU of the KKT system; the QP's optimality conditions written from its definition).
PARITY UNPINNED all the same: the reference holds no source or fixture."""
import numpy as np
import pytest

import admm_library_amd as pkg
import admm_ref as ar
import oracle_c as oc
import _indep as ind

CASES = {
    "configs1_n6": (lambda: pkg.cw_rendezvous(N=1000, batch=4), 0.05),
    "configs4_n12": (lambda: pkg.cw_formation(N=1000, batch=3), 0.05),
    "ltv_q_stage_bounds": (lambda: pkg.random_ltv(N=400, n=6, m=3, batch=3, seed=4), 0.3),
}


@pytest.mark.parametrize("case", CASES)
def test_oracle_x_update_vs_banded_kkt_at_full_horizon(case):
    """One x-update of both oracles from a random (z, y) equals the banded LU solve of the KKT system: <= 1e-11."""
    make, rho = CASES[case]
    p = make()
    rng = np.random.default_rng(5)
    z0, y0 = rng.standard_normal((2, p.batch, p.L))
    g = -rho * (z0 - y0) + (0 if p.q is None else p.q)
    w, backward_err = ind.banded_x_update(p, g, rho)
    assert backward_err < 1e-12
    c = oc.solve(p, rho=rho, max_iter=1, stop=False, z0=z0, y0=y0)["w"]
    assert np.abs(c - w).max() <= 1e-11 * max(1.0, np.abs(w).max())
    f = ar.factor(p.A, p.B, p.Q, p.R, p.QN, rho, p.N)
    npw = ar.x_update(f, g, p.x0)
    assert np.abs(npw - w).max() <= 1e-11 * max(1.0, np.abs(w).max())


@pytest.mark.parametrize("case", CASES)
def test_oracle_converged_solution_is_optimal_at_full_horizon(case):
    """A C-oracle solve to eps = 1e-8 satisfies the QP's optimality conditions (stated bounds below), and the box binds."""
    make, rho = CASES[case]
    p = make()
    res = oc.solve(p, rho=rho, eps_abs=1e-8, eps_rel=1e-8, max_iter=20000, check_interval=10, adapt_interval=50, alpha=1.6)
    assert res["status"].all()
    feas_dyn, feas_box, stat, comp, n_active = ind.kkt_certificate_batch(p, res["z"], res["y"], res["rho"])
    assert feas_dyn.max() < 1e-6 and feas_box.max() == 0.0 and stat.max() < 1e-6 and comp.max() < 1e-6
    assert n_active >= 50


def test_certificate_rejects_a_wrong_answer():
    """The checker is not vacuous: the solution of a DIFFERENT x0, a shifted multiplier and a clipped-away control each fail."""
    p = pkg.cw_rendezvous(N=200, batch=2)
    res = oc.solve(p, rho=0.05, eps_abs=1e-9, eps_rel=1e-9, max_iter=20000, adapt_interval=50, alpha=1.6)
    ok = ind.kkt_certificate_batch(p, res["z"], res["y"], res["rho"])
    assert max(ok[0].max(), ok[2].max(), ok[3].max()) < 1e-6
    swapped = ind.kkt_certificate_batch(p, res["z"][::-1].copy(), res["y"][::-1].copy(), res["rho"])
    assert swapped[0].max() > 1e-3                                   # dynamics defect: wrong x0
    bad_y = res["y"] + 1e-3
    assert ind.kkt_certificate_batch(p, res["z"], bad_y, res["rho"])[3].max() > 1e-5
    z2 = res["z"].copy()
    z2.reshape(2, p.N, p.nb)[:, :, :p.m] *= 0.5
    assert ind.kkt_certificate_batch(p, z2, res["y"], res["rho"])[2].max() > 1e-4


def test_oracle_with_per_instance_dynamics_is_optimal_qp_by_qp():
    """Per-instance dynamics (n = 12, m = 6; every QP its own A_k, B_k and box): the C oracle's converged (z, rho y) satisfies each
    QP's own optimality conditions -- the checker tests/test_gpu_independent.py applies to the HIP path's wide-shape kernels."""
    p = pkg.cw_formation_instances(N=40, batch=3)
    r = oc.solve(p, rho=0.05, alpha=1.6, eps_abs=1e-9, eps_rel=1e-9, max_iter=30000, check_interval=10)
    assert r["status"].all()
    feas_dyn, feas_box, stat, comp, n_active = ind.kkt_certificate_instances(p, r["z"], r["y"], 0.05)
    assert feas_box.max() == 0.0 and feas_dyn.max() < 1e-7 and stat.max() < 1e-7 and comp.max() < 1e-7
    assert n_active > 50
    bad = r["z"].copy()
    bad[1, 7] += 1e-3                                  # a perturbed state entry of QP 1 breaks ITS dynamics, nobody else's
    fd = ind.kkt_certificate_instances(p, bad, r["y"], 0.05)[0]
    assert fd[1] > 1e-4 and fd[0] < 1e-7 and fd[2] < 1e-7

"""ORACLE-FREE checks of the HIP path at BASELINE's sizes.

Every other -m gpu test compares the HIP path with the build's CPU oracle.  These compare it with routines that share
no code with either (tests/_indep.py -- NumPy / SciPy only, written from the QP's definition):
  (a) configs[0] solved on the GPU vs scipy.optimize.lsq_linear (BVLS) on the condensed problem;
  (b) the QP's optimality conditions over ALL 4096 QPs of a converged GPU solve of configs[2] and of configs[4];
  (c) one GPU x-update at N = 1000, n = 6 / n = 12, full batch, vs ONE banded LU (scipy.linalg.solve_banded) of the
      KKT system; and four whole default-path iterations (start form, xfze, xbze: DESIGN.md §4.8) vs an ADMM loop whose
      x-update is that banded solve (512 QPs spread over the 4096 the GPU runs).
Nothing in this file imports oracle/.  PARITY UNPINNED regardless: these pin the HIP path to the QP, not to a
reference implementation, because none exists."""
import numpy as np
import pytest

import admm_library_amd as pkg
from admm_library_amd import _abi
import _indep as ind

pytestmark = pytest.mark.gpu


def test_config0_gpu_solve_vs_scipy_bvls(gpu):
    """configs[0]: N = 50 double integrator, |u| <= 1 -- the GPU's converged controls equal SciPy's bounded least squares."""
    p = pkg.double_integrator(N=50, batch=3)
    with pkg.Solver(p, pkg.Options(rho=1.0, eps_abs=1e-10, eps_rel=1e-10, max_iter=20000, check_interval=10)) as s:
        info = s.solve()
        _, z, _ = s.get()
    assert info.status.all()
    n_active = 0
    for b in range(p.batch):
        u_ref = ind.condensed_bvls(p, b)
        u_gpu = z[b].reshape(p.N, p.nb)[:, :p.m].reshape(-1)
        assert np.abs(u_gpu - u_ref).max() < 1e-6
        n_active += int((np.abs(np.abs(u_ref) - 1.0) < 1e-9).sum())
    assert n_active >= 10


FULL = {
    "configs2_n6": (lambda: pkg.cw_rendezvous(N=1000, batch=4096), _abi.PRECISION_FP64),
    "configs4_n12_one_lane": (lambda: pkg.cw_formation(N=1000, batch=4096), _abi.PRECISION_FP64),
    "configs4_n12_fp64_mfma": (lambda: pkg.cw_formation(N=1000, batch=4096), _abi.PRECISION_FP64_MFMA),
    "configs4_n12_mixed": (lambda: pkg.cw_formation(N=1000, batch=4096), _abi.PRECISION_MIXED),
}


@pytest.mark.parametrize("case", FULL)
def test_optimality_of_every_qp_of_a_converged_full_batch(gpu, case):
    """All 4096 QPs of configs[2] / configs[4] (the latter on the default one-lane kernels, on the fp64-MFMA form and in the
    mixed mode with fp64 refinement), solved to eps = 1e-8 with the adaptive rule: every QP converged, and every QP's (z, rho y) satisfies
    dynamics defect < 1e-6, box violation == 0, stationarity < 1e-6, complementarity < 1e-6 (variables are O(1))."""
    make, mode = FULL[case]
    p = make()
    opt = pkg.Options(rho=0.05, alpha=1.6, eps_abs=1e-8, eps_rel=1e-8, max_iter=20000, check_interval=10,
                      adapt_interval=50, precision_mode=mode)
    with pkg.Solver(p, opt) as s:
        info = s.solve()
        _, z, y = s.get()
    assert int(info.n_converged) == p.batch and info.status.all()
    feas_dyn, feas_box, stat, comp, n_active = ind.kkt_certificate_batch(p, z, y, float(info.rho))
    worst = dict(feas_dyn=feas_dyn.max(), feas_box=feas_box.max(), stat=stat.max(), comp=comp.max())
    print(case, "iterations", info.iters_run, "rho", info.rho, worst, "active", n_active)
    assert feas_box.max() == 0.0, worst
    assert feas_dyn.max() < 1e-6 and stat.max() < 1e-6 and comp.max() < 1e-6, worst
    assert n_active > 100 * p.batch // 10                                          # the thrust box binds throughout the batch


@pytest.mark.parametrize("case", ["rendezvous_n6", "formation_n12"])
def test_optimality_of_every_qp_with_per_instance_dynamics(gpu, case):
    """Per-instance dynamics and boxes (time_varying = 2, stage_bounds = 2) -- n = 6 on the one-lane kernels,
    n = 12 on the rows-over-lanes kernels of the wide shapes (device factorisation, tiled operands staged through LDS): 256 QPs of
    N = 200 stages solved to eps = 1e-8 with the per-QP adaptive rule; every QP's (z, rho_b y) satisfies ITS OWN QP's optimality
    conditions (its A_k, B_k, its box) to the bounds of the batch-shared test."""
    make = pkg.cw_rendezvous_instances if case == "rendezvous_n6" else pkg.cw_formation_instances
    p = make(N=200, batch=256)
    opt = pkg.Options(rho=0.05, alpha=1.6, eps_abs=1e-8, eps_rel=1e-8, max_iter=20000, check_interval=10, adapt_interval=50)
    with pkg.Solver(p, opt) as s:
        assert s.path()["per_instance"]
        info = s.solve()
        _, z, y = s.get()
        rho = s.rho_per_qp()
    assert int(info.n_converged) == p.batch and info.status.all()
    feas_dyn, feas_box, stat, comp, n_active = ind.kkt_certificate_instances(p, z, y, rho)
    worst = dict(feas_dyn=feas_dyn.max(), feas_box=feas_box.max(), stat=stat.max(), comp=comp.max())
    print(case, "iterations", info.iters_run, "rho", rho.min(), rho.max(), worst, "active", n_active)
    assert feas_box.max() == 0.0, worst
    assert feas_dyn.max() < 1e-6 and stat.max() < 1e-6 and comp.max() < 1e-6, worst
    assert n_active > 10 * p.batch


@pytest.mark.parametrize("case", ["rendezvous_n6_4096x1000", "formation_n12_1024x1000"])
def test_optimality_with_per_instance_dynamics_at_full_horizon(gpu, case):
    """The same certificate at BASELINE's horizon: 4096 QPs of N = 1000 at n = 6 (1.8 GB of per-instance A, B; one-lane kernels, 8
    segments per QP) and 1024 QPs at n = 12 (1.8 GB; rows-over-lanes kernels, tiled operands, 4 segments) solved to eps = 1e-8 with the
    per-QP adaptive rule; 96 QPs spread over the batch are checked against THEIR OWN optimality conditions on the host."""
    import dataclasses
    if case.startswith("rendezvous"):
        p = pkg.cw_rendezvous_instances(N=1000, batch=4096)
    else:
        p = pkg.cw_formation_instances(N=1000, batch=1024)
    opt = pkg.Options(rho=0.05, alpha=1.6, eps_abs=1e-8, eps_rel=1e-8, max_iter=30000, check_interval=10, adapt_interval=50)
    with pkg.Solver(p, opt) as s:
        info = s.solve()
        _, z, y = s.get()
        rho = s.rho_per_qp()
        segs = s.geometry()["segments"]
    assert int(info.n_converged) == p.batch and info.status.all()
    idx = np.linspace(0, p.batch - 1, 96).astype(int)
    sub = dataclasses.replace(p, A=p.A[idx], B=p.B[idx], x0=p.x0[idx], lo=p.lo[idx], hi=p.hi[idx], q=None if p.q is None else p.q[idx])
    feas_dyn, feas_box, stat, comp, n_active = ind.kkt_certificate_instances(sub, z[idx], y[idx], rho[idx])
    worst = dict(feas_dyn=feas_dyn.max(), feas_box=feas_box.max(), stat=stat.max(), comp=comp.max())
    print(case, "iterations", info.iters_run, "segments", segs, "rho", rho.min(), rho.max(), worst, "active", n_active)
    assert feas_box.max() == 0.0, worst
    assert feas_dyn.max() < 1e-6 and stat.max() < 1e-6 and comp.max() < 1e-6, worst
    assert n_active > 100 * len(idx)


XCASES = {
    "n6": (lambda: pkg.cw_rendezvous(N=1000, batch=4096), _abi.PRECISION_FP64),
    "n12": (lambda: pkg.cw_formation(N=1000, batch=4096), _abi.PRECISION_FP64),
    "n12_fp64_mfma": (lambda: pkg.cw_formation(N=1000, batch=4096), _abi.PRECISION_FP64_MFMA),
}


def _state(p, seed):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((p.batch, p.L)), 0.3 * rng.standard_normal((p.batch, p.L))


@pytest.mark.parametrize("case", XCASES)
def test_x_update_vs_banded_kkt_at_full_size(gpu, case):
    """w = x-update(z, y) of the whole 4096-QP batch (segmented sweeps + MFMA scan) vs one banded LU solve with 4096 right-hand sides."""
    make, mode = XCASES[case]
    p = make()
    rho = 0.05
    z0, y0 = _state(p, 11)
    with pkg.Solver(p, pkg.Options(rho=rho, precision_mode=mode)) as s:
        assert s.geometry()["segments"] > 1
        s.set_state(z=z0, y=y0)
        s.step_x()
        w, _, _ = s.get()
    ref, backward_err = ind.banded_x_update(p, -rho * (z0 - y0), rho)
    assert backward_err < 1e-12
    err = np.abs(w - ref).max()
    print(case, "x-update vs banded KKT:", err, "|w|", np.abs(ref).max())
    assert err <= 1e-10 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("case", XCASES)
def test_default_path_iterations_vs_banded_admm(gpu, case):
    """Four iterations of the default path from a random (z, y) -- a start form, then the fused alternating kernels in both
    directions (xfze / xbze; n12_fp64_mfma: their MFMA forms) -- vs the textbook loop of DESIGN.md §2 with the banded
    KKT solve as x-update.  The GPU runs all 4096 QPs; the host loop (the test's cost: 4 banded solves per QP) checks every 8th of
    them -- 512 QPs spread over the whole batch, every wave and panel position included: 1e-10 on w, z, y."""
    import dataclasses
    make, mode = XCASES[case]
    p = make()
    rho, iters = 0.05, 4
    z, y = _state(p, 12)
    with pkg.Solver(p, pkg.Options(rho=rho, precision_mode=mode)) as s:
        s.set_state(z=z, y=y)
        s.iterate(iters)
        wg, zg, yg = s.get()
    idx = np.arange(0, p.batch, 8) + (np.arange(p.batch // 8) % 8)          # 0, 9, 18, ..., 63, 64, 73, ...: every residue mod 8
    sub = dataclasses.replace(p, x0=p.x0[idx], q=None if p.q is None else p.q[idx])
    z, y = z[idx], y[idx]
    lo, hi = (np.tile(b, p.N) for b in (p.lo, p.hi))
    for _ in range(iters):
        w, _ = ind.banded_x_update(sub, -rho * (z - y), rho)
        v = w + y
        z = np.minimum(np.maximum(v, lo), hi)
        y = v - z
    for name, a, r in (("w", wg[idx], w), ("z", zg[idx], z), ("y", yg[idx], y)):
        err = np.abs(a - r).max()
        print(case, name, err)
        assert err <= 1e-10 * max(1.0, np.abs(r).max()), (name, err)

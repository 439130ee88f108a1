"""Paths that every other test exercises on SMALL problems, at BASELINE's horizon and large batches, against the oracle on a sample of
the QPs.  Round 3: a set-up race (asynchronous zero-fills overtaking synchronous uploads) needed GBs of device arrays to fire and was
seen by none of ~600 small-problem parity tests; the timed benchmark checks no values.  PARITY UNPINNED."""
import dataclasses

import numpy as np
import pytest

import admm_library_amd as pkg
import oracle_c as oc

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _sample(p, idx):
    kw = dict(x0=p.x0[idx], q=None if p.q is None else p.q[idx])
    if p.per_instance:
        kw.update(A=p.A[idx], B=p.B[idx])
        if p.lo.ndim == 3:
            kw.update(lo=p.lo[idx], hi=p.hi[idx])
    return dataclasses.replace(p, **kw)


def _close(got, ref, idx):
    return all(np.abs(a[idx] - ref[k]).max() <= TOL * max(1.0, np.abs(ref[k]).max()) for a, k in zip(got, ("w", "z", "y")))


@pytest.mark.parametrize("case", ["n12_1024", "n6_4096"])
def test_per_instance_update_set_rho_and_state_round_trip_at_scale(gpu, case):
    """admm_update_problem (trial factorisation into scratch copies of A, B, K, S^-1, pointer-swap commit), admm_set_rho (trial + commit)
    and a set_state / get round trip on handles with GBs of per-instance data."""
    make, batch = (pkg.cw_formation_instances, 1024) if case == "n12_1024" else (pkg.cw_rendezvous_instances, 4096)
    p = make(N=1000, batch=batch)
    p2 = make(N=1000, batch=batch, seed0=pkg.SEED0 + 999, spread=0.08)
    idx = np.linspace(0, batch - 1, 12).astype(int)
    rng = np.random.default_rng(3)
    z0, y0 = 0.1 * rng.standard_normal((batch, p.L)), 0.1 * rng.standard_normal((batch, p.L))
    with pkg.Solver(p, pkg.Options(rho=0.05, alpha=1.6)) as s:
        s.set_state(z=z0, y=y0)
        _, zb, yb = s.get()
        np.testing.assert_array_equal(zb, z0)
        np.testing.assert_array_equal(yb, y0)
        s.run(5, residual_every=2)
        ref = oc.solve(_sample(p, idx), rho=0.05, alpha=1.6, max_iter=5, stop=False, z0=z0[idx], y0=y0[idx])
        assert _close(s.get(), ref, idx)
        s.set_rho(0.2)                                                   # every QP refactored on the device, dual rescaled
        s.run(4, residual_every=1)
        ref2 = oc.solve(_sample(p, idx), rho=0.2, alpha=1.6, max_iter=4, stop=False, z0=ref["z"], y0=ref["y"] * (0.05 / 0.2))
        assert _close(s.get(), ref2, idx)
        s.update_problem(p2)                                             # new dynamics, boxes and x0 for every QP
        s.set_state(z=np.zeros((batch, p.L)), y=np.zeros((batch, p.L)))
        s.run(6, residual_every=3)
        assert _close(s.get(), oc.solve(_sample(p2, idx), rho=0.2, alpha=1.6, max_iter=6, stop=False), idx)


def test_per_instance_thrust_magnitude_bound_at_scale(gpu):
    """The thrust-magnitude forms of the one-lane per-instance kernels (n = 6) on 2048 QPs of N = 1000."""
    p = pkg.cw_rendezvous_instances(N=1000, batch=2048)
    lo, hi = p.lo.copy(), p.hi.copy()
    lo[..., :3], hi[..., :3] = -np.inf, np.inf
    p = dataclasses.replace(p, lo=lo, hi=hi, unorm=0.25)
    idx = np.linspace(0, p.batch - 1, 8).astype(int)
    with pkg.Solver(p, pkg.Options(rho=0.05, alpha=1.6)) as s:
        s.run(8, residual_every=4)
        assert _close(s.get(), oc.solve(_sample(p, idx), rho=0.05, alpha=1.6, max_iter=8, stop=False), idx)


def test_shared_dynamics_update_problem_at_scale(gpu):
    """configs[2]'s handle given new shared dynamics, box and initial states (host refactor, record upload) and then a new rho."""
    p = pkg.cw_rendezvous(N=1000, batch=4096)
    p2 = pkg.cw_rendezvous(N=1000, batch=4096, seed0=pkg.SEED0 + 5, u_max=0.15)
    idx = np.linspace(0, p.batch - 1, 16).astype(int)
    with pkg.Solver(p, pkg.Options(rho=0.05)) as s:
        s.run(7, residual_every=3)
        assert _close(s.get(), oc.solve(_sample(p, idx), rho=0.05, max_iter=7, stop=False), idx)
        s.update_problem(p2)
        s.set_state(z=np.zeros((p.batch, p.L)), y=np.zeros((p.batch, p.L)))
        s.run(6, residual_every=2)
        ref = oc.solve(_sample(p2, idx), rho=0.05, max_iter=6, stop=False)
        assert _close(s.get(), ref, idx)
        s.set_rho(0.1)
        s.run(5, residual_every=1)
        assert _close(s.get(), oc.solve(_sample(p2, idx), rho=0.1, max_iter=5, stop=False, z0=ref["z"], y0=ref["y"] * 0.5), idx)

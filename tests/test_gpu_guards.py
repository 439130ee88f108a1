"""Guards on a live handle: non-finite warm starts are refused, and a refactor (admm_set_rho,
the adaptive rule, admm_update_problem) re-checks the conditioning bound that admm_setup applied when it
chose the segment count."""
import numpy as np
import pytest

import admm_library_amd as pkg
import oracle_c as oc
from admm_library_amd import _abi

pytestmark = pytest.mark.gpu


def _code(name):
    return {v: k for k, v in _abi.STATUS_NAMES.items()}[name]


@pytest.mark.parametrize("which", ["w", "z", "y"])
@pytest.mark.parametrize("bad", [np.nan, np.inf, -np.inf])
def test_set_state_rejects_non_finite(gpu, which, bad):
    p = pkg.cw_rendezvous(N=40, batch=5)
    with pkg.Solver(p, pkg.Options(rho=0.05)) as s:
        s.iterate(3)
        w0, z0, y0 = s.get()
        a = np.zeros((p.batch, p.L))
        a[3, 17] = bad
        with pytest.raises(pkg.AdmmError) as e:
            s.set_state(**{which: a})
        assert e.value.code == _code("ADMM_ERR_INVALID")
        if which != "w":                           # the warm start of admm_solve goes through the same check
            with pytest.raises(pkg.AdmmError):
                s.solve(**{which + "0": a})
        w1, z1, y1 = s.get()                       # the handle's state is untouched
        for a0, a1 in ((w0, w1), (z0, z1), (y0, y1)):
            np.testing.assert_array_equal(a0, a1)


def _rho_sensitive_plant(batch=3):
    """Unstable plant, no state cost: the closed loop does not depend on rho, but the segment transfer matrix
    Xi ~ 1 / rho, so max |W| = 2.48 at rho = 0.1 and 2.5e3 at rho = 1e-4 (tests/test_host.py has the host check)."""
    n, m, N = 2, 1, 64
    A = np.array([[1.3, 1.0], [0.0, 1.3]])
    B = np.array([[0.0], [1.0]])
    rng = np.random.default_rng(7)
    return pkg.Problem(N=N, A=A, B=B, Q=np.zeros((n, n)), R=1e-9 * np.eye(m), QN=np.zeros((n, n)),
                       x0=rng.standard_normal((batch, n)), lo=np.r_[-0.5, -np.inf, -np.inf], hi=np.r_[0.5, np.inf, np.inf])


def test_set_rho_rechecks_the_conditioning_bound(gpu):
    p = _rho_sensitive_plant()
    with pkg.Solver(p, pkg.Options(rho=0.1)) as s:
        assert s.geometry()["segments"] > 1
        s.iterate(5)
        with pytest.raises(pkg.AdmmError) as e:
            s.set_rho(1e-4)                         # max |W| would be 2.5e3 with the handle's frozen segment count
        assert e.value.code == _code("ADMM_ERR_NUMERIC") and "segment" in str(e.value)
        s.iterate(5)                                # the handle is unchanged and keeps iterating at rho = 0.1
        w, z, y = s.get()
    ref = oc.solve(p, rho=0.1, max_iter=10, stop=False)
    assert np.abs(z - ref["z"]).max() <= 1e-10 and np.abs(w - ref["w"]).max() <= 1e-10
    # a fixed segment count is the caller's responsibility: no guard, the change goes through
    with pkg.Solver(p, pkg.Options(rho=0.1, segments=4)) as s:
        s.set_rho(1e-4)
    # set up at the small rho, the automatic choice backs off to fewer segments by itself
    with pkg.Solver(p, pkg.Options(rho=1e-4)) as s:
        assert s.geometry()["segments"] == 1


def test_adaptive_rule_stops_adapting_when_a_change_is_refused(gpu):
    """The adaptive rule wants rho lower than the conditioning bound allows: the solve keeps the last admissible
    rho (and stops adapting) instead of failing or running an ill-conditioned factor."""
    p = _rho_sensitive_plant()
    opt = pkg.Options(rho=0.1, max_iter=400, check_interval=10, adapt_interval=10, adapt_mu=1.0001, adapt_tau=10.0,
                      eps_abs=1e-12, eps_rel=1e-12)
    with pkg.Solver(p, opt) as s:
        info = s.solve()
        assert np.isfinite(s.get()[0]).all()
    assert info.rho >= 5e-3          # 0.1 -> 0.01 is admissible (max |W| = 24.8); 1e-3 (248) is refused


def test_per_instance_segments_keep_the_conditioning_bound(gpu):
    """Per-instance dynamics with segments in time (per-QP transfer matrices from pseg_kernel): the automatic segment count
    falls back to one segment when some QP's transfer matrices exceed the bound at setup, a rho change that would break it
    is refused with the handle unchanged, and a segment count fixed by the caller is not guarded."""
    import dataclasses
    base = _rho_sensitive_plant(batch=5)
    A = np.broadcast_to(base.A, (5, base.N, 2, 2)).copy()
    A[:, :, 0, 1] += 0.01 * np.arange(5)[:, None]                  # five different plants
    B = np.broadcast_to(base.B, (5, base.N, 2, 1)).copy()
    p = dataclasses.replace(base, A=A, B=B)
    assert p.per_instance
    with pkg.Solver(p, pkg.Options(rho=1e-4)) as s:
        assert s.geometry()["segments"] == 1                        # backed off at setup
    with pkg.Solver(p, pkg.Options(rho=0.1)) as s:
        assert s.geometry()["segments"] == 8
        s.iterate(5)
        with pytest.raises(pkg.AdmmError) as e:
            s.set_rho(1e-4)
        assert e.value.code == _code("ADMM_ERR_NUMERIC") and "segment" in str(e.value)
        np.testing.assert_array_equal(s.rho_per_qp(), np.full(5, 0.1))
        s.iterate(5)
        w, z, y = s.get()
    ref = oc.solve(p, rho=0.1, max_iter=10, stop=False)
    assert np.abs(z - ref["z"]).max() <= 1e-10 and np.abs(w - ref["w"]).max() <= 1e-10
    with pkg.Solver(p, pkg.Options(rho=0.1, segments=4)) as s:
        s.set_rho(1e-4)


# ---- the kernel path and its margin are reported, never silent (ABI v6: admm_get_path, admm_last_warning)

def test_path_of_the_headline_handle(gpu):
    p = pkg.cw_rendezvous(N=1000, batch=256)
    import warnings
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        s = pkg.Solver(p, pkg.Options(rho=0.05))
    with s:
        path, geo = s.path(), s.geometry()
    assert not [w for w in rec if issubclass(w.category, RuntimeWarning)] and s.last_warning == ""
    assert path["alternating"] and path["alt_requested"] and path["kernel_family"] == "one_lane_fp64" and path["xfree"]
    assert path["scan_form"] == "mfma_gemm" and path["segments"] == geo["segments"] and path["auto_segments"]
    assert 0.0 <= path["alt_check"] <= path["alt_gate"] == 5e-12          # the measured margin of the forward-elimination form
    assert 0.0 < path["scan_growth"] <= 100.0


@pytest.mark.parametrize("make,rho,segments", [
    (lambda: pkg.cw_rendezvous(N=1000, batch=8), 32.0, 0),                      # CW at N = 1000 with rho = 32
    (lambda: pkg.random_ltv(N=40, n=6, m=1, batch=2, seed=5), 0.05, 4),         # a random n = 6, m = 1 system at rho = 0.05
], ids=["cw_rho32", "random_n6_m1"])
def test_forward_elimination_gate_failure_is_reported(gpu, make, rho, segments):
    """The two known problems whose forward-elimination form misses its host check: the handle runs the plain fused path,
    SAYS so (warning + admm_get_path), and its iterates are the oracle's."""
    p = make()
    with pytest.warns(RuntimeWarning, match="forward-elimination form failed its host check"):
        s = pkg.Solver(p, pkg.Options(rho=rho, segments=segments))
    with s:
        path = s.path()
        assert path["alt_requested"] and not path["alternating"]
        assert path["alt_check"] > path["alt_gate"]
        assert "plain fused path" in s.last_warning and "plain fused path" in pkg.last_warning()
        with pytest.raises(pkg.AdmmError):                       # admm_profile of the alternating pair: unsupported on this handle
            s.profile(2, alternating=True)
        s.iterate(12)
        w, z, y = s.get()
    ref = oc.solve(p, rho=rho, max_iter=12, stop=False)
    for a, k in ((w, "w"), (z, "z"), (y, "y")):
        assert np.abs(a - ref[k]).max() <= 1e-10 * max(1.0, np.abs(ref[k]).max())


def test_losing_the_alternating_form_on_a_rho_change_is_reported(gpu):
    """admm_set_rho to a rho whose forward form fails: the call succeeds, the handle falls back, the warning says so; a later
    setup at a good rho reports nothing."""
    p = pkg.cw_rendezvous(N=1000, batch=8)
    with pkg.Solver(p, pkg.Options(rho=0.05, segments=16)) as s:
        assert s.path()["alternating"] and s.last_warning == ""
        s.iterate(4)
        with pytest.warns(RuntimeWarning, match="refactor: the forward-elimination form failed"):
            s.set_rho(32.0)
        assert not s.path()["alternating"] and s.path()["alt_check"] > 5e-12
        s.iterate(4)
        w, z, y = s.get()
    ref = oc.solve(p, rho=0.05, max_iter=4, stop=False)
    ref2 = oc.solve(p, rho=32.0, max_iter=4, stop=False, z0=ref["z"], y0=ref["y"] * (0.05 / 32.0))
    assert np.abs(z - ref2["z"]).max() <= 1e-10 and np.abs(w - ref2["w"]).max() <= 1e-10


def test_flags_that_exclude_alternation_are_not_warnings(gpu):
    p = pkg.cw_rendezvous(N=100, batch=8)
    with pkg.Solver(p, pkg.Options(rho=0.05, flags=_abi.FLAG_NO_ALTERNATE)) as s:
        path = s.path()
        assert not path["alt_requested"] and not path["alternating"] and s.last_warning == ""


# ---- per-instance dynamics: a refused change leaves the handle untouched; the per-QP adaptive rule keeps the conditioning bound
# (admm_update_problem overwrote the device data before it knew the factor was acceptable; the masked refactor of the
# per-QP rule ignored the bound)

def _sensitive_instances(batch=5):
    import dataclasses
    base = _rho_sensitive_plant(batch=batch)
    A = np.broadcast_to(base.A, (batch, base.N, 2, 2)).copy()
    A[:, :, 0, 1] += 0.01 * np.arange(batch)[:, None]
    B = np.broadcast_to(base.B, (batch, base.N, 2, 1)).copy()
    return dataclasses.replace(base, A=A, B=B)


def test_per_instance_update_problem_refused_leaves_the_handle_unchanged(gpu):
    """admm_update_problem with dynamics whose segment transfer matrices break the conditioning bound (a faster-growing plant
    with hardly any control authority, at the handle's rho and 8 segments): ADMM_ERR_NUMERIC, and the handle still holds the OLD dynamics, box, x0 and
    factors -- its iterates continue exactly as if the call had not happened (vs the oracle on the old problem)."""
    import dataclasses
    p = _sensitive_instances()
    # a faster-growing plant with hardly any authority: the uncontrolled segment transfer matrix A^8 has entries of 135-210
    # (NumPy emulation of pseg_kernel), beyond the bound of 100
    worse = dataclasses.replace(p, A=p.A + 0.3 * np.eye(2), B=p.B * 1e-3, x0=p.x0 + 1.0, lo=p.lo * 0.5, hi=p.hi * 0.5)
    with pkg.Solver(p, pkg.Options(rho=0.1)) as s:
        assert s.geometry()["segments"] == 8
        s.iterate(4)
        with pytest.raises(pkg.AdmmError) as e:
            s.update_problem(worse)
        assert e.value.code == _code("ADMM_ERR_NUMERIC") and "refused" in str(e.value)
        s.iterate(6)
        w, z, y = s.get()
        # an acceptable update still goes through afterwards (the trial buffers are reused)
        ok = dataclasses.replace(p, x0=p.x0 * 0.5)
        s.update_problem(ok)
        s.iterate(3)
        w2, z2, y2 = s.get()
    ref = oc.solve(p, rho=0.1, max_iter=10, stop=False)
    for a, k in ((w, "w"), (z, "z"), (y, "y")):
        assert np.abs(a - ref[k]).max() <= 1e-10, k
    ref2 = oc.solve(ok, rho=0.1, max_iter=3, stop=False, z0=ref["z"], y0=ref["y"])
    assert np.abs(z2 - ref2["z"]).max() <= 1e-10 and np.abs(w2 - ref2["w"]).max() <= 1e-10


def test_per_qp_adaptive_rule_keeps_the_conditioning_bound(gpu):
    """The per-QP adaptive rule on the rho-sensitive plants (adapt_mu ~ 1: every test wants a change; tau = 10): a QP's change
    to a rho whose transfer matrices exceed the bound is refused -- it keeps the last admissible rho and stops adapting, the
    call reports it -- instead of iterating on ill-conditioned segments.  With one segment there is no bound and the rule
    runs on (same options), and the two solves' QPs end on different rho."""
    p = _sensitive_instances()
    kw = dict(rho=0.1, max_iter=300, check_interval=10, adapt_interval=10, adapt_mu=1.0001, adapt_tau=10.0, eps_abs=1e-12, eps_rel=1e-12)
    with pytest.warns(RuntimeWarning, match="adaptive rho: the change of"):
        s = pkg.Solver(p, pkg.Options(**kw))
        with s:
            assert s.geometry()["segments"] == 8
            info = s.solve()
            rho = s.rho_per_qp()
            w, z, y = s.get()
    assert np.isfinite(w).all() and (rho >= 5e-3).all()            # 0.1 -> 0.01 admissible, 1e-3 refused (as with shared dynamics)
    with pkg.Solver(p, pkg.Options(segments=1, **kw)) as s1:
        s1.solve()
        rho1 = s1.rho_per_qp()
    assert (rho1 < 5e-3).any()                                      # unguarded, the rule does go lower
    # the guarded solve's iterates are those of an exact x-update: dynamics feasibility of w to rounding
    wb = w.reshape(p.batch, p.N, p.nb)
    x = np.concatenate([p.x0[:, None, :], wb[:, :, p.m:]], axis=1)
    defect = x[:, 1:] - np.einsum("bkij,bkj->bki", p.A, x[:, :-1]) - np.einsum("bkij,bkj->bki", p.B, wb[:, :, :p.m])
    assert np.abs(defect).max() <= 1e-9 * max(1.0, np.abs(x).max())

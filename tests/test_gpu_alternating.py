"""GPU tests of the alternating-direction iteration (xfze_kernel / xbze_kernel):
consecutive iterations solve the x-update in opposite elimination orders so that each
substitution sweep is fused with the next elimination sweep.  It is the default iteration where
the forward form passes its host check (practically always); ADMM_FLAG_NO_ALTERNATE (8) selects the
plain xb + xfz kernels.  Checked against the C oracle (PARITY UNPINNED, see test_gpu_parity.py)
and against the plain path, tolerance 1e-10 on O(1) iterates."""
import numpy as np
import pytest

import admm_library_amd as pkg
from admm_library_amd import _abi
import oracle_c as oc

pytestmark = pytest.mark.gpu
TOL = 1e-10
NO_ALT = _abi.FLAG_NO_ALTERNATE

ALT_CASES = [
    # (factory, rho, segments)
    (lambda: pkg.cw_rendezvous(N=200, batch=70), 0.05, 4),           # clamped lanes (pitch 128 < 256)
    (lambda: pkg.cw_rendezvous(N=1000, batch=65), 0.05, 0),
    (lambda: pkg.cw_rendezvous(N=1000, batch=3), 0.8, 1),            # one segment
    (lambda: pkg.double_integrator(N=50, batch=130), 1.0, 7),
    (lambda: pkg.random_ltv(N=37, n=4, m=2, batch=70, seed=2, with_q=False), 0.3, 5),
    (lambda: pkg.random_ltv(N=9, n=2, m=2, batch=2, seed=8, with_q=False), 0.4, 9),     # one stage per segment
    (lambda: pkg.random_ltv(N=24, n=3, m=1, batch=5, seed=4, with_q=False), 0.2, 4),    # m < n: singular C_k early
    (lambda: pkg.random_ltv(N=40, n=6, m=3, batch=300, seed=5, with_q=False), 0.3, 3),  # two column blocks
    (lambda: pkg.random_ltv(N=1, n=4, m=2, batch=3, seed=18, with_q=False), 0.3, 0),
    (lambda: pkg.random_ltv(N=2, n=6, m=3, batch=2, seed=19, with_q=False), 0.3, 0),
    (lambda: pkg.random_ltv(N=90, n=5, m=3, batch=10, seed=21, with_q=False), 0.3, 2),  # several LDS refills per segment
    # larger blocks (other prefetch depth / register budget rules, admm_kernels_alt.hpp)
    (lambda: pkg.random_ltv(N=30, n=8, m=4, batch=64, seed=10, with_q=False), 0.2, 3),
    (lambda: pkg.random_ltv(N=20, n=9, m=3, batch=5, seed=54, with_q=False), 0.3, 2),
    (lambda: pkg.random_ltv(N=20, n=10, m=4, batch=5, seed=55, with_q=False), 0.3, 2),
    (lambda: pkg.random_ltv(N=40, n=12, m=3, batch=4, seed=16, with_q=False), 0.2, 3),
    (lambda: pkg.random_ltv(N=40, n=12, m=6, batch=3, seed=17, with_q=False), 0.4, 2),
    (lambda: pkg.cw_formation(N=120, batch=66), 0.05, 0),
    # with a linear term q (per-stage bounds, full weights): the HASQ kernel forms
    (lambda: pkg.random_ltv(N=37, n=6, m=3, batch=70, seed=2), 0.3, 5),
    (lambda: pkg.random_ltv(N=30, n=4, m=2, batch=130, seed=22), 0.3, 4),
    (lambda: pkg.random_ltv(N=20, n=10, m=4, batch=5, seed=55), 0.3, 2),
    (lambda: pkg.random_ltv(N=24, n=12, m=6, batch=3, seed=6), 0.4, 4),
    # thrust-magnitude (second-order-cone) bound on the control rows: the SOC kernel forms
    (lambda: pkg.cw_rendezvous(N=200, batch=70, thrust_norm=True), 0.05, 4),
    (lambda: pkg.random_ltv(N=37, n=6, m=3, batch=9, seed=31, thrust_norm=True), 0.3, 5),
    (lambda: pkg.random_ltv(N=20, n=4, m=2, batch=5, seed=32, thrust_norm=True, with_q=False), 0.3, 3),
]


def _close(a, b):
    return np.abs(a - b).max() <= TOL * max(1.0, np.abs(b).max())


@pytest.mark.parametrize("idx", range(len(ALT_CASES)))
def test_alternating_iterates_match_oracle(gpu, idx):
    """Every iteration count 1..9 and 16 (both parities, from a fresh handle each time, so the
    schedule start / end rules of next_form are all exercised), direct launches and graph replay."""
    make, rho, segs = ALT_CASES[idx]
    p = make()
    for K in (1, 2, 3, 4, 5, 6, 7, 8, 9, 16):
        ref = oc.solve(p, rho=rho, max_iter=K, stop=False)
        # FLAG_NO_MFMA: the one-lane kernels this file is about (for small batches of q-free problems of an
        # MFMA-compiled shape the default is the fp64 MFMA form, tests/test_gpu_mfma.py)
        for flags in (0, _abi.FLAG_NO_MFMA, _abi.FLAG_NO_MFMA | _abi.FLAG_GRAPH):
            with pkg.Solver(p, pkg.Options(rho=rho, segments=segs, flags=flags)) as s:
                s.iterate(K)
                w, z, y = s.get()
            assert _close(w, ref["w"]) and _close(z, ref["z"]) and _close(y, ref["y"]), (K, flags)


def test_alternating_is_enabled_and_optional(gpu):
    """The alternating kernels run by default for the headline shape and can be switched off;
    a linear term or a thrust-magnitude bound selects the HASQ / SOC kernel forms."""
    p = pkg.cw_rendezvous(N=100, batch=66)
    with pkg.Solver(p, pkg.Options(rho=0.05)) as s:
        pr = s.profile(2, residuals=True, alternating=True)
        assert pr["xfze_ms"] > 0 and pr["xbze_ms"] > 0
    with pkg.Solver(p, pkg.Options(rho=0.05, flags=NO_ALT)) as s:
        with pytest.raises(pkg.AdmmError):
            s.profile(2, alternating=True)
    pq = pkg.random_ltv(N=20, n=4, m=2, batch=5, seed=3)          # a linear term q rides along (HASQ kernel forms)
    assert pq.q is not None
    with pkg.Solver(pq, pkg.Options(rho=0.3)) as s:
        assert s.profile(2, alternating=True)["xbze_ms"] > 0
    ps = pkg.cw_rendezvous(N=60, batch=5, thrust_norm=True)       # thrust-magnitude bound: SOC kernel forms
    with pkg.Solver(ps, pkg.Options(rho=0.05)) as s:
        assert s.profile(2, alternating=True)["xfze_ms"] > 0


def test_falls_back_to_the_plain_kernels_when_the_forward_form_is_unavailable(gpu):
    """Singular A_k: admm_setup leaves the alternating kernels off; the iteration is the plain one."""
    p = pkg.random_ltv(N=12, n=3, m=2, batch=5, seed=77, with_q=False)
    A = np.array(p.A)
    A[5] = np.diag([1.0, 0.0, 0.5])
    p = pkg.Problem(N=p.N, A=A, B=p.B, Q=p.Q, R=p.R, QN=p.QN, x0=p.x0, lo=p.lo, hi=p.hi)
    with pkg.Solver(p, pkg.Options(rho=0.3, segments=3)) as s:
        with pytest.raises(pkg.AdmmError):
            s.profile(1, alternating=True)
        s.iterate(9)
        w, z, y = s.get()
    ref = oc.solve(p, rho=0.3, max_iter=9, stop=False)
    assert _close(w, ref["w"]) and _close(z, ref["z"]) and _close(y, ref["y"])


@pytest.mark.parametrize("alpha", [1.0, 1.6])
def test_call_patterns_and_residuals(gpu, alpha):
    """State carried across calls of every length and parity, read-outs in between (w is
    re-materialised by admm_get after a forward-form iteration), residuals of every iteration."""
    p = pkg.random_ltv(N=33, n=6, m=3, batch=69, seed=31, with_q=False)
    rho = 0.3
    done = 0
    with pkg.Solver(p, pkg.Options(rho=rho, alpha=alpha, segments=4)) as s:
        for k, every in ((3, 1), (4, 1), (1, 1), (6, 3), (2, 1), (7, 7), (1, 0), (5, 1)):
            s.run(k, residual_every=every)
            done += k
            ref = oc.solve(p, rho=rho, alpha=alpha, max_iter=done, check_interval=1, eps_abs=0, eps_rel=0, stop=False)
            w, z, y = s.get()
            assert _close(w, ref["w"]) and _close(z, ref["z"]) and _close(y, ref["y"]), done
            if every and k % every == 0:          # the last iteration of the call evaluated residuals
                r, sd, nw, nz, ny = s.residuals()
                assert np.abs(r - ref["r"]).max() <= 1e-10 and np.abs(sd - ref["s"]).max() <= 1e-10, done


def test_alternating_agrees_with_plain_path(gpu):
    """Same problem, same calls, alternating vs ADMM_FLAG_NO_ALTERNATE: iterates agree to
    rounding (different elimination orders of the same KKT system), residuals likewise."""
    p = pkg.cw_rendezvous(N=300, batch=130)
    outs = []
    for flags in (0, NO_ALT):
        with pkg.Solver(p, pkg.Options(rho=0.05, flags=flags)) as s:
            s.run(41, residual_every=1)
            outs.append(s.get() + tuple(s.residuals()))
    for a, b in zip(*outs):
        assert np.abs(a - b).max() <= 1e-11 * max(1.0, np.abs(b).max())


def test_state_changes_between_alternating_iterations(gpu):
    """admm_set_rho, admm_update_instances and admm_set_state drop whatever the last fused
    elimination left behind; the iteration restarts cleanly from the state."""
    p = pkg.cw_rendezvous(N=120, batch=20)
    p2 = pkg.cw_rendezvous(N=120, batch=20, seed0=4242)
    res = []
    for flags in (0, NO_ALT):
        with pkg.Solver(p, pkg.Options(rho=0.05, flags=flags)) as s:
            s.iterate(5)
            s.set_rho(0.2)
            s.iterate(4)
            s.update_instances(x0=p2.x0)
            s.iterate(6)
            w, z, y = s.get()
            s.set_state(z=z * 0.5, y=y)
            s.iterate(3)
            s.step_x()                       # plain x-update in between
            s.iterate(2)
            res.append(s.get())
    for a, b in zip(*res):
        assert np.abs(a - b).max() <= 1e-11 * max(1.0, np.abs(b).max())


def test_update_problem_equals_a_fresh_handle(gpu):
    """admm_update_problem: new dynamics, weights, box, x0 and q on an existing handle (what a
    successive-convexification caller does between outer iterations) = admm_setup on the new data."""
    p1 = pkg.random_ltv(N=33, n=6, m=3, batch=20, seed=41)
    p2 = pkg.random_ltv(N=33, n=6, m=3, batch=20, seed=42)
    zero = np.zeros((p2.batch, p2.L))
    with pkg.Solver(p1, pkg.Options(rho=0.3, segments=4)) as s:
        s.run(7, residual_every=1)
        s.update_problem(p2)
        s.set_state(z=zero, y=zero)
        s.run(12, residual_every=1)
        got = s.get() + tuple(s.residuals())
        info = s.solve(z0=zero, y0=zero)             # and a full solve on the updated handle
        with pytest.raises(pkg.AdmmError):           # shape changes are refused, the handle stays usable
            s.update_problem(pkg.random_ltv(N=34, n=6, m=3, batch=20, seed=42))
        with pytest.raises(pkg.AdmmError):
            s.update_problem(pkg.random_ltv(N=33, n=6, m=3, batch=20, seed=42, with_q=False))
        s.iterate(2)
    with pkg.Solver(p2, pkg.Options(rho=0.3, segments=4)) as s:
        s.run(12, residual_every=1)
        want = s.get() + tuple(s.residuals())
        info2 = s.solve(z0=zero, y0=zero)
    for a, b in zip(got, want):
        assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(b).max())
    ref = oc.solve(p2, rho=0.3, max_iter=12, check_interval=1, eps_abs=0, eps_rel=0, stop=False)
    assert _close(got[0], ref["w"]) and _close(got[1], ref["z"]) and _close(got[2], ref["y"])
    assert info.iters_run == info2.iters_run and np.array_equal(info.iters, info2.iters)


@pytest.mark.parametrize("kw", [dict(), dict(alpha=1.5), dict(check_interval=7), dict(adapt_interval=20),
                                dict(max_iter=33, check_interval=10)])
def test_solve_with_alternating_iterations(gpu, kw):
    """admm_solve (stopping rule, per-QP first-converged iteration, adaptive rho) on the
    alternating path vs the C oracle."""
    p = pkg.cw_rendezvous(N=80, batch=40)
    args = dict(rho=0.05, eps_abs=1e-5, eps_rel=1e-5, max_iter=600, check_interval=10)
    args.update(kw)
    with pkg.Solver(p, pkg.Options(**args)) as s:
        info = s.solve()
        w, z, y = s.get()
    ref = oc.solve(p, **args)
    assert info.iters_run == ref["iters_run"]
    np.testing.assert_array_equal(info.status, ref["status"])
    np.testing.assert_array_equal(info.iters, ref["iters"])
    assert _close(w, ref["w"]) and _close(z, ref["z"]) and _close(y, ref["y"])
    assert info.rho == ref["rho"] and info.rho_updates == ref["rho_updates"]


def test_alternating_properties_at_full_size(gpu):
    """configs[2] shape: after 40 alternating iterations the forward-form w satisfies the dynamics,
    z lies in the box, and the state agrees with the plain path."""
    p = pkg.cw_rendezvous(N=1000, batch=4096)
    outs = []
    for flags in (0, NO_ALT):
        with pkg.Solver(p, pkg.Options(rho=0.05, flags=flags)) as s:
            s.run(40, residual_every=10)
            outs.append(s.get())
    (w, z, y), (w2, z2, y2) = outs
    for a, b in ((w, w2), (z, z2), (y, y2)):
        assert np.abs(a - b).max() <= 1e-10 * max(1.0, np.abs(b).max())
    wb = w.reshape(p.batch, p.N, p.nb)
    x = np.concatenate([p.x0[:, None, :], wb[:, :, p.m:]], axis=1)
    u = wb[:, :, :p.m]
    Ad, Bd = np.asarray(p.A).reshape(p.n, p.n), np.asarray(p.B).reshape(p.n, p.m)
    defect = x[:, 1:] - (x[:, :-1] @ Ad.T + u @ Bd.T)
    assert np.abs(defect).max() <= 1e-10 * max(1.0, np.abs(x).max())


@pytest.mark.parametrize("make,opts", [(lambda: pkg.cw_rendezvous(N=200, batch=300), {}), (lambda: pkg.cw_formation(N=120, batch=300), {}),
                                       (lambda: pkg.cw_formation(N=120, batch=300), {"precision_mode": 2}),
                                       (lambda: pkg.cw_rendezvous(N=120, batch=40), {}),
                                       (lambda: pkg.cw_rendezvous(N=64, batch=5, thrust_norm=True), {}),
                                       (lambda: pkg.random_ltv(N=70, n=6, m=3, batch=67, seed=77, state_bounds=False), {}),
                                       (lambda: pkg.random_ltv(N=45, n=4, m=2, batch=9, seed=78, state_bounds=False, thrust_norm=True), {}),
                                       (lambda: pkg.random_ltv(N=45, n=8, m=4, batch=130, seed=79, with_q=False, state_bounds=False), {})],
                         ids=["one_lane_6_3", "one_lane_12_6", "mfma_12_6", "mfma_6_3_small_batch", "thrust_magnitude", "ltv_q_stage_bounds",
                              "ltv_q_stage_thrust_bounds", "ltv_8_4"])
def test_skipping_v_of_unbounded_state_rows_is_exact(gpu, make, opts, monkeypatch):
    """XFREE kernel forms: where every state row is unbounded at every stage, the iterations that evaluate
    no residuals do not read v of those rows (y = 0 identically there) -- nor write it while the next iteration is of the same kind.  Same iterates, BIT for bit, as with the skip
    disabled (ADMM_NO_SKIPV), through a mix of residual and non-residual iterations and both kernel families."""
    p = make()
    out = []
    for skip in (True, False):
        if skip:
            monkeypatch.delenv("ADMM_NO_SKIPV", raising=False)
        else:
            monkeypatch.setenv("ADMM_NO_SKIPV", "1")
        with pkg.Solver(p, pkg.Options(rho=0.05, **opts)) as s:
            s.run(23, residual_every=5)
            s.iterate(4)
            for k in (1, 2, 3, 7, 10):          # call lengths of both parities: the no-store form (XFREE = 2) needs its successor
                s.iterate(k)                    # to be a fused alternating kernel of the same call, else the rows are written
            s.run(11, residual_every=4)
            out.append(s.get() + s.residuals())
    for a, b in zip(*out):
        np.testing.assert_array_equal(a, b)
    ref = oc.solve(p, rho=0.05, max_iter=27 + 23 + 11, stop=False)
    assert _close(out[0][0], ref["w"]) and _close(out[0][1], ref["z"]) and _close(out[0][2], ref["y"])

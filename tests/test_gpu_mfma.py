"""MFMA forms of the fused kernels (BASELINE.json configs[4]), through the C ABI
(options.precision_mode), against the C oracle.

ADMM_PRECISION_FP64_MFMA is the same fp64 iteration as the one-lane kernels with the stage operators applied by
v_mfma_f64_16x16x4_f64: tolerance 1e-10 on the iterates, like every fp64 path.
ADMM_PRECISION_MIXED runs the Riccati form's two products in fp32: it cannot meet 1e-10.  Its stated
tolerances: iterates within 1e-5 (relative to max(1, |ref|_inf)) of the fp64 oracle's at equal iteration counts
(measured 2e-7 .. 1.1e-6; 5e-5 at N = 1000, measured 1.4e-5); a solve refined in fp64 meets the SAME stopping rule as the fp64 path, in a comparable number
of iterations (within 25 %), with a KKT certificate as good as the fp64 path's at the same eps.
PARITY UNPINNED: the oracle is the build's own CPU restatement."""
import numpy as np
import pytest

import admm_library_amd as pkg
import oracle_c as oc
from admm_library_amd import _abi
from _kkt import kkt_certificate

pytestmark = pytest.mark.gpu
FP64, MIXED, FP64_MFMA = _abi.PRECISION_FP64, _abi.PRECISION_MIXED, _abi.PRECISION_FP64_MFMA

CASES = [
    (lambda: pkg.cw_formation(N=120, batch=66), 0.05, 0),                                   # configs[4]'s shape
    (lambda: pkg.cw_formation(N=64, batch=300), 0.05, 5),                                   # several column blocks, odd segments
    (lambda: pkg.cw_rendezvous(N=200, batch=70), 0.05, 0),
    (lambda: pkg.random_ltv(N=30, n=10, m=4, batch=5, seed=3, with_q=False), 0.3, 3),
    (lambda: pkg.random_ltv(N=9, n=12, m=6, batch=17, seed=5, with_q=False), 0.4, 2),        # per-stage bounds incl. +-inf
    (lambda: pkg.random_ltv(N=3, n=6, m=3, batch=2, seed=6, with_q=False), 0.4, 0),          # tiny horizon
]


def _err(got, ref):
    return max(np.abs(a - ref[k]).max() / max(1.0, np.abs(ref[k]).max()) for a, k in zip(got, ("w", "z", "y")))


@pytest.mark.parametrize("alpha", [1.0, 1.6])
@pytest.mark.parametrize("idx", range(len(CASES)))
def test_fp64_mfma_iterates_match_the_oracle(gpu, idx, alpha):
    make, rho, segs = CASES[idx]
    p = make()
    with pkg.Solver(p, pkg.Options(rho=rho, alpha=alpha, segments=segs, precision_mode=FP64_MFMA)) as s:
        done = 0
        for upto in (1, 2, 3, 4, 5, 10, 41):
            s.run(upto - done, residual_every=3)
            done = upto
            ref = oc.solve(p, rho=rho, alpha=alpha, max_iter=upto, stop=False)
            assert _err(s.get(), ref) <= 1e-10, upto


@pytest.mark.parametrize("idx", range(len(CASES)))
def test_mixed_iterates_within_the_stated_tolerance(gpu, idx):
    make, rho, segs = CASES[idx]
    p = make()
    worst = 0.0
    with pkg.Solver(p, pkg.Options(rho=rho, segments=segs, precision_mode=MIXED)) as s:
        done = 0
        for upto in (1, 2, 5, 10, 40):
            s.iterate(upto - done)
            done = upto
            ref = oc.solve(p, rho=rho, max_iter=upto, stop=False)
            worst = max(worst, _err(s.get(), ref))
    assert worst <= 1e-5
    assert worst > 1e-12 or p.N < 4        # it really is the reduced-precision path


def test_residuals_of_the_mfma_forms(gpu):
    """Residual partials are reduced across the four lane groups of a column in the MFMA kernels."""
    p = pkg.cw_formation(N=120, batch=66)
    ref = oc.solve(p, rho=0.05, max_iter=20, check_interval=1, stop=False)
    for mode, tol in ((FP64_MFMA, 1e-10), (MIXED, 1e-5)):
        with pkg.Solver(p, pkg.Options(rho=0.05, precision_mode=mode)) as s:
            s.run(20, residual_every=1)
            r, sd, nw, nz, ny = s.residuals()
        assert np.abs(r - ref["r"]).max() <= tol * max(1, ref["r"].max())
        assert np.abs(sd - ref["s"]).max() <= tol * max(1, ref["s"].max())


@pytest.mark.parametrize("make,kw", [
    (lambda: pkg.cw_rendezvous(N=200, batch=70), dict(rho=0.05, adapt_interval=50, alpha=1.6)),
    (lambda: pkg.cw_formation(N=120, batch=66), dict(rho=0.05, adapt_interval=50, alpha=1.6)),
], ids=["cw_rendezvous_adaptive", "cw_formation_adaptive"])
def test_solves_iters_to_eps_and_kkt(gpu, make, kw):
    """the reduced-precision mode is judged on iterations-to-eps and the final KKT residual against
    the fp64 path.  FP64_MFMA must reproduce the fp64 solve; MIXED (fp32 phase + fp64 refinement) must meet the same
    stopping rule, in a comparable number of iterations, with a certificate as good."""
    p = make()
    base = dict(eps_abs=1e-6, eps_rel=1e-6, max_iter=6000, check_interval=10, **kw)
    ref = oc.solve(p, **base)
    assert ref["status"].all()
    out = {}
    for mode in (FP64, FP64_MFMA, MIXED):
        with pkg.Solver(p, pkg.Options(precision_mode=mode, **base)) as s:
            info = s.solve()
            w, z, y = s.get()
        out[mode] = (info, z, y)
        assert info.n_converged == p.batch
    i64, z64, y64 = out[FP64]
    im, zm, ym = out[FP64_MFMA]
    assert im.iters_run == i64.iters_run == ref["iters_run"] and im.rho == i64.rho and im.mixed_iters == 0
    np.testing.assert_array_equal(im.iters, i64.iters)
    assert np.abs(zm - ref["z"]).max() <= 1e-10
    ix, zx, yx = out[MIXED]
    assert 0 < ix.mixed_iters <= ix.iters_run
    assert ix.iters_run <= 1.25 * i64.iters_run + 20
    assert np.abs(zx - ref["z"]).max() <= 5e-4            # two solutions stopped at eps = 1e-6 (the fp64 GPU path vs oracle: 1e-12)
    rho_f = float(ix.rho)
    for b in (0, p.batch // 2, p.batch - 1):
        c64 = kkt_certificate(p, b, z64[b], y64[b], float(i64.rho))
        cx = kkt_certificate(p, b, zx[b], yx[b], rho_f)
        for a, bb in zip(cx, c64):
            assert a <= max(3 * bb, 2e-5), (cx, c64)


def test_full_size_fp64_mfma_against_the_one_lane_kernels(gpu):
    """configs[4] at BASELINE's full size (N = 1000, n = 12, m = 6, batch 4096): the MFMA form's iterates against the
    fp64 one-lane kernels' after 40 iterations (both are fp64; 1e-10), dynamics feasibility of w, box feasibility of
    z, and a 16-QP slice against the oracle; MIXED against the same within its stated tolerance."""
    p = pkg.cw_formation(N=1000, batch=4096)
    got = {}
    for mode in (FP64, FP64_MFMA, MIXED):
        with pkg.Solver(p, pkg.Options(rho=0.05, precision_mode=mode)) as s:
            assert s.geometry()["segments"] == 16
            s.run(40, residual_every=10)
            got[mode] = s.get()
    for a, b in zip(got[FP64_MFMA], got[FP64]):
        assert np.abs(a - b).max() <= 1e-10 * max(1.0, np.abs(b).max())
    for a, b in zip(got[MIXED], got[FP64]):           # 62-stage segments: the fp32 chains are 8x longer than in the small cases
        assert np.abs(a - b).max() <= 5e-5 * max(1.0, np.abs(b).max())          # measured 1.4e-5
    w, z, y = got[FP64_MFMA]
    n, m, nb = p.n, p.m, p.nb
    W = w.reshape(p.batch, p.N, nb)
    x_prev = np.concatenate([p.x0[:, None, :], W[:, :-1, m:]], axis=1)
    defect = W[:, :, m:] - (x_prev @ p.A.T + W[:, :, :m] @ p.B.T)
    assert np.abs(defect).max() <= 1e-11 * max(1.0, np.abs(W).max())
    assert (z.reshape(p.batch, p.N, nb) <= p.hi + 1e-15).all() and (z.reshape(p.batch, p.N, nb) >= p.lo - 1e-15).all()
    lo = 1000
    ref = oc.solve(p.slice(lo, lo + 16), rho=0.05, max_iter=40, stop=False)
    assert _err([a[lo:lo + 16] for a in got[FP64_MFMA]], ref) <= 1e-10


def test_unsupported_combinations_fail_loudly(gpu):
    unsup = {v: k for k, v in _abi.STATUS_NAMES.items()}["ADMM_ERR_UNSUPPORTED"]
    for p, opt in ((pkg.random_ltv(N=10, n=12, m=6, batch=3, seed=1, with_q=True), dict(precision_mode=MIXED)),      # q
                   (pkg.cw_rendezvous(N=40, batch=3, thrust_norm=True), dict(precision_mode=FP64_MFMA)),            # ball
                   (pkg.random_ltv(N=10, n=8, m=3, batch=3, seed=1, with_q=False), dict(precision_mode=MIXED)),     # not compiled
                   (pkg.cw_rendezvous(N=40, batch=3), dict(precision_mode=FP64_MFMA, flags=_abi.FLAG_UNFUSED))):
        with pytest.raises(pkg.AdmmError) as e:
            pkg.Solver(p, pkg.Options(rho=0.1, **opt))
        assert e.value.code == unsup
    with pytest.raises(pkg.AdmmError):
        pkg.Solver(pkg.cw_rendezvous(N=40, batch=3), pkg.Options(rho=0.1, precision_mode=7))


@pytest.mark.parametrize("make", [lambda: pkg.random_ltv(N=40, n=6, m=3, batch=1, seed=71), lambda: pkg.random_ltv(N=33, n=6, m=3, batch=70, seed=72),
                                  lambda: pkg.random_ltv(N=25, n=6, m=3, batch=17, seed=73, state_bounds=False)],
                         ids=["one_qp", "two_waves", "state_rows_unbounded"])
@pytest.mark.parametrize("alpha", [1.0, 1.6])
def test_fp64_mfma_with_a_linear_term(gpu, make, alpha):
    """The HASQ forms of the fp64 MFMA kernels ((6, 3), batches of up to 128 QPs: the default there, and what the
    single-trajectory successive-convexification QPs run): iterates, residuals and a solve against the oracle, and against the
    one-lane kernels (ADMM_FLAG_NO_MFMA)."""
    p = make()
    assert p.q is not None
    with pkg.Solver(p, pkg.Options(rho=0.3, alpha=alpha, precision_mode=FP64_MFMA)) as s, \
            pkg.Solver(p, pkg.Options(rho=0.3, alpha=alpha, flags=_abi.FLAG_NO_MFMA)) as s1:
        done = 0
        for upto in (1, 2, 3, 8, 9, 30):
            s.run(upto - done, residual_every=4)
            s1.run(upto - done, residual_every=4)
            done = upto
            ref = oc.solve(p, rho=0.3, alpha=alpha, max_iter=upto, check_interval=4, stop=False)
            assert _err(s.get(), ref) <= 1e-10 and _err(s1.get(), ref) <= 1e-10, upto
    with pkg.Solver(p, pkg.Options(rho=0.3, alpha=alpha, precision_mode=FP64_MFMA)) as s:
        s.run(28, residual_every=4)
        ref = oc.solve(p, rho=0.3, alpha=alpha, max_iter=28, check_interval=4, stop=False)
        r, sd = s.residuals()[:2]
        assert np.abs(r - ref["r"]).max() <= 1e-10 and np.abs(sd - ref["s"]).max() <= 1e-10
    kw = dict(rho=0.3, alpha=alpha, eps_abs=1e-7, eps_rel=1e-7, max_iter=2000, check_interval=10, adapt_interval=50)
    ref = oc.solve(p, **kw)
    with pkg.Solver(p, pkg.Options(**kw)) as s:                      # the default picks the MFMA form for these batches
        info = s.solve()
        assert int(info.iters_run) == ref["iters_run"] and float(info.rho) == ref["rho"]
        assert _err(s.get(), ref) <= 1e-10
    # not available: mixed precision with q, q at n = 12, q with a batch beyond 128
    unsup = {v: k for k, v in _abi.STATUS_NAMES.items()}["ADMM_ERR_UNSUPPORTED"]
    for bad, opt in ((p, dict(precision_mode=MIXED)), (pkg.random_ltv(N=10, n=6, m=3, batch=200, seed=1), dict(precision_mode=FP64_MFMA))):
        with pytest.raises(pkg.AdmmError) as e:
            pkg.Solver(bad, pkg.Options(rho=0.3, **opt))
        assert e.value.code == unsup

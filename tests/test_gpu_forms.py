"""Form sweep at the headline shapes (VERDICT r02 hygiene; tools/form_coverage.py): the fused kernels are templates over
RESID x RELAX x HASQ x SOC x XFREE (and the MFMA family over the precision modes) -- profiles/r03c_gpu_tests_kernel_stats.csv showed
that the suite launched 187 of the 288 forms compiled for (6, 3) and 112 of 194 for (12, 6); the combinations it missed were
over-relaxation or a linear term together with the thrust-magnitude bound, and most of those at (12, 6).  Every combination of
{q, thrust-magnitude bound, alpha != 1, state rows unbounded} runs here through a mix of residual / non-residual iterations and
call lengths of both parities, on the one-lane kernels, on the library's default family and (where compiled) on the fp64 MFMA
form, against the C oracle within 1e-10.  PARITY UNPINNED."""
import itertools

import numpy as np
import pytest

import admm_library_amd as pkg
import oracle_c as oc
from admm_library_amd import _abi

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _close(got, ref, tol=TOL):
    return all(np.abs(a - ref[k]).max() <= tol * max(1.0, np.abs(ref[k]).max()) for a, k in zip(got, ("w", "z", "y")))


COMBOS = list(itertools.product([False, True], repeat=4))      # (q, thrust bound, relaxed, state rows unbounded)


@pytest.mark.parametrize("shape", [(6, 3), (12, 6)], ids=["n6", "n12"])
@pytest.mark.parametrize("combo", COMBOS, ids=["".join(c for c, on in zip("qsrx", t) if on) or "plain" for t in COMBOS])
def test_every_form_combination_matches_the_oracle(gpu, shape, combo):
    with_q, soc, relaxed, xfree = combo
    n, m = shape
    alpha = 1.6 if relaxed else 1.0
    for batch in (70, 9, 300):                     # above / below the batch sizes where the MFMA family is the default; two panels per wave
        p = pkg.random_ltv(N=26, n=n, m=m, batch=batch, seed=100 + 7 * n + batch, with_q=with_q, state_bounds=not xfree,
                           thrust_norm=soc)
        ref = None
        ran = 0
        rng = np.random.default_rng(batch)
        z0, y0 = 0.1 * rng.standard_normal((batch, p.L)), 0.1 * rng.standard_normal((batch, p.L))
        # (mixed fp32 / fp64: its stated tolerance, tests/test_gpu_mfma.py)
        for opts, tol in ((dict(flags=_abi.FLAG_NO_MFMA), TOL), (dict(), TOL), (dict(precision_mode=_abi.PRECISION_FP64_MFMA), TOL),
                          (dict(precision_mode=_abi.PRECISION_MIXED), 1e-5),
                          (dict(flags=_abi.FLAG_NO_MFMA | _abi.FLAG_NO_ALTERNATE), TOL),                 # the plain path's kernels
                          (dict(flags=_abi.FLAG_NO_ALTERNATE, precision_mode=_abi.PRECISION_MIXED), 1e-5)):
            if batch == 300 and "precision_mode" not in opts:
                continue
            try:
                s = pkg.Solver(p, pkg.Options(rho=0.3, alpha=alpha, **opts))
            except pkg.AdmmError as e:             # a family without this form (e.g. the MFMA kernels and the thrust bound)
                assert e.code == {v: k for k, v in _abi.STATUS_NAMES.items()}["ADMM_ERR_UNSUPPORTED"], e
                continue
            with s:
                s.set_state(z=z0, y=y0)            # a caller's (z, y): the first sweep runs in the (z, y)-form kernels,
                s.run(1, residual_every=1 if batch == 9 else 0)       # with or without residuals
                s.run(8, residual_every=4)         # residual and non-residual forms, XFREE 0 / 1 / 2 ...
                s.iterate(5)
                s.iterate(2)
                s.run(6, residual_every=1)
                s.run(10, residual_every=3)        # ... and the other parity: a forward / a backward kernel before a residual one
                s.run(7, residual_every=2)
                got = s.get()
            if ref is None:
                ref = oc.solve(p, rho=0.3, alpha=alpha, max_iter=39, stop=False, z0=z0, y0=y0)
            assert _close(got, ref, tol), (batch, opts)
            ran += 1
        assert ran >= 2 or batch == 300            # (the MFMA family has no thrust-magnitude forms: nothing runs there at batch 300)


PCOMBOS = list(itertools.product([False, True], repeat=3))     # (q, relaxed, per-instance box)


@pytest.mark.parametrize("shape", [(6, 3), (12, 6)], ids=["n6", "n12"])
@pytest.mark.parametrize("combo", PCOMBOS, ids=["".join(c for c, on in zip("qrb", t) if on) or "plain" for t in PCOMBOS])
def test_every_per_instance_form_combination_matches_the_oracle(gpu, shape, combo, monkeypatch):
    """The per-instance sweeps (templates over HASQ x VFORM / VIN x RESID x RELAX x per-instance box x segments x thrust bound, in the
    one-lane and the rows-over-lanes family): from a caller-supplied (z, y) -- the (z, y)-form kernels, with and without residuals --
    on into the v-form, one segment and several, both families where both exist."""
    with_q, relaxed, pbox = combo
    n, m = shape
    alpha = 1.6 if relaxed else 1.0
    rng = np.random.default_rng(5)
    for batch, soc in ((70, False), (9, False), (70, True)):
        p = pkg.random_instances(N=26, n=n, m=m, batch=batch, seed=300 + n + batch, with_q=with_q, instance_bounds=pbox, thrust_norm=soc)
        z0, y0 = 0.1 * rng.standard_normal((batch, p.L)), 0.1 * rng.standard_normal((batch, p.L))
        for segments in (1, 4):
            for form in ("auto", "lane_per_qp", "rows_over_lanes"):
                monkeypatch.delenv("ADMM_PI_LANE_PER_QP", raising=False)
                monkeypatch.delenv("ADMM_PI_ROWS", raising=False)
                if form == "lane_per_qp":
                    monkeypatch.setenv("ADMM_PI_LANE_PER_QP", "1")
                elif form == "rows_over_lanes":
                    monkeypatch.setenv("ADMM_PI_ROWS", "1")
                for first_with_residuals in (True, False):
                    with pkg.Solver(p, pkg.Options(rho=0.3, alpha=alpha, segments=segments)) as s:
                        s.set_state(z=z0, y=y0)
                        s.run(1, residual_every=1 if first_with_residuals else 0)     # the (z, y)-form kernels
                        s.run(6, residual_every=3)
                        s.iterate(3)
                        got = s.get()
                    ref = oc.solve(p, rho=0.3, alpha=alpha, max_iter=10, stop=False, z0=z0, y0=y0)
                    assert _close(got, ref), (batch, soc, segments, form, first_with_residuals)

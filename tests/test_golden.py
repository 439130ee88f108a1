"""Golden fixtures (tests/golden/, cut by make_golden.py from the NumPy oracle;
the reference itself ships none -- parity unpinned).

CPU: both oracles reproduce the stored iterates.  GPU: the HIP path does."""
import numpy as np
import pytest
from scipy.io import loadmat

import admm_library_amd as pkg
import admm_ref as ar
import oracle_c as oc
from _golden import NAMES, GOLDEN_DIR, load


@pytest.mark.parametrize("name", NAMES)
def test_oracles_reproduce_golden(name):
    p, d = load(name)
    rho, alpha = float(d["rho"]), float(d["alpha"])
    for it in d["iters"]:
        it = int(it)
        c = oc.solve(p, rho=rho, alpha=alpha, max_iter=it, stop=False)
        n = ar.solve(p.A, p.B, p.Q, p.R, p.QN, p.x0, p.lo, p.hi, p.N, q=p.q, rho=rho, alpha=alpha,
                     max_iter=it, stop=False, unorm=p.unorm)
        for k in "wzy":
            ref = d[f"{k}_{it}"]
            assert np.abs(c[k] - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
            assert np.abs(getattr(n, k) - ref).max() <= 1e-13 * max(1.0, np.abs(ref).max())
    kw = dict(eps_abs=float(d["solve_eps_abs"]), eps_rel=float(d["solve_eps_rel"]),
              max_iter=int(d["solve_max_iter"]), check_interval=int(d["solve_check_interval"]),
              adapt_interval=int(d["solve_adapt_interval"]))
    c = oc.solve(p, rho=rho, alpha=alpha, **kw)
    assert c["iters_run"] == int(d["solve_iters_run"]) and c["rho"] == float(d["solve_rho_final"])
    np.testing.assert_array_equal(c["iters"], d["solve_iters"])
    np.testing.assert_array_equal(c["status"], d["solve_status"])
    assert np.abs(c["z"] - d["solve_z"]).max() <= 1e-11


def test_mat_fixture_matches_npz():
    """The .mat copy (for MATLAB users) holds the same numbers as the .npz."""
    _, d = load("golden_config1")
    m = loadmat(GOLDEN_DIR + "/golden_config1.mat")
    for k in ("A", "B", "x0", "z_100", "y_100", "w_10", "solve_z"):
        np.testing.assert_array_equal(np.asarray(m[k]).reshape(d[k].shape), d[k])


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [0, 2], ids=["fused", "unfused"])
@pytest.mark.parametrize("name", NAMES)
def test_hip_reproduces_golden(gpu, name, flags):
    p, d = load(name)
    rho, alpha = float(d["rho"]), float(d["alpha"])
    with pkg.Solver(p, pkg.Options(rho=rho, alpha=alpha, flags=flags)) as s:
        done = 0
        for it in d["iters"]:
            it = int(it)
            s.iterate(it - done)
            done = it
            got = dict(zip("wzy", s.get()))
            for k in "wzy":
                ref = d[f"{k}_{it}"]
                assert np.abs(got[k] - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max()), (k, it)
    kw = dict(eps_abs=float(d["solve_eps_abs"]), eps_rel=float(d["solve_eps_rel"]),
              max_iter=int(d["solve_max_iter"]), check_interval=int(d["solve_check_interval"]),
              adapt_interval=int(d["solve_adapt_interval"]))
    with pkg.Solver(p, pkg.Options(rho=rho, alpha=alpha, flags=flags, **kw)) as s:
        info = s.solve()
        _, z, y = s.get(False, True, True)
    assert info.iters_run == int(d["solve_iters_run"]) and info.rho == float(d["solve_rho_final"])
    assert (np.abs(info.iters - d["solve_iters"]) <= kw["check_interval"]).all()
    np.testing.assert_array_equal(info.status, d["solve_status"])
    assert np.abs(z - d["solve_z"]).max() <= 1e-10

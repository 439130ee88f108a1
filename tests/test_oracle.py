"""CPU tests that pin the oracle (tiers T1-T3).

The reference ships no source, fixture or golden vector (README.md:1-2 only),
so "parity" for this project is PARITY UNPINNED: the oracle is the build's own
CPU restatement.  These tests therefore check it against things that do not
depend on the build's ADMM code at all: a dense KKT solve, SciPy's bounded
least squares, and a KKT optimality certificate evaluated by a separate routine.
"""
import numpy as np
import pytest
from scipy.optimize import lsq_linear

import admm_library_amd as pkg
import admm_ref as ar
import oracle_c as oc


def _solve_np(p, **kw):
    return ar.solve(p.A, p.B, p.Q, p.R, p.QN, p.x0, p.lo, p.hi, p.N, q=p.q, **kw)


def test_x_update_matches_dense_kkt():
    """The Riccati sweep equals a dense solve of [P+rho I, G'; G, 0]."""
    p = pkg.random_ltv(N=9, n=4, m=2, batch=3, seed=3)
    rho = 0.7
    f = ar.factor(p.A, p.B, p.Q, p.R, p.QN, rho, p.N)
    g = np.random.default_rng(1).standard_normal((p.batch, p.L))
    w = ar.x_update(f, g, p.x0)
    for b in range(p.batch):
        wd = ar.kkt_x_update_dense(p.A, p.B, p.Q, p.R, p.QN, p.x0[b], p.N, g[b], rho)
        assert np.abs(w[b] - wd).max() < 1e-11


def test_T1_unconstrained_equals_lqr():
    """T1: with bounds at +-inf the ADMM fixed point is the finite-horizon LQR
    solution, obtained here from a dense KKT solve of the original QP."""
    p = pkg.double_integrator(N=30, batch=2)
    p.lo = np.full(3, -np.inf)
    p.hi = np.full(3, np.inf)
    res = _solve_np(p, rho=1.0, eps_abs=1e-11, eps_rel=1e-11, max_iter=4000, check_interval=5)
    assert res.status.all()
    for b in range(p.batch):
        P, q, G, bvec = ar.dense_qp(p.A, p.B, p.Q, p.R, p.QN, p.x0[b], p.N)
        L, nc = P.shape[0], G.shape[0]
        sol = np.linalg.solve(np.block([[P, G.T], [G, np.zeros((nc, nc))]]), np.concatenate([-q, bvec]))
        assert np.abs(res.z[b] - sol[:L]).max() < 1e-8
        assert np.abs(res.w[b] - sol[:L]).max() < 1e-8


def _condense(p, b):
    """x = Sx x0 + Su u for one instance (LTI or LTV), stacked x_1..x_N."""
    A, B = ar.expand_dynamics(p.A, p.B, p.N)
    n, m, N = p.n, p.m, p.N
    Sx = np.zeros((N * n, n))
    Su = np.zeros((N * n, N * m))
    Phi = np.eye(n)
    for k in range(N):
        if k > 0:
            Su[k * n:(k + 1) * n] = A[k] @ Su[(k - 1) * n:k * n]
        Su[k * n:(k + 1) * n, k * m:(k + 1) * m] = B[k]
        Phi = A[k] @ Phi
        Sx[k * n:(k + 1) * n] = Phi
    return Sx, Su


def test_T2_config1_vs_scipy_bvls():
    """T2: BASELINE.json configs[0] (N=50 double integrator, |u|<=1): condense the
    states out and solve the bounded least-squares problem in u with SciPy."""
    p = pkg.double_integrator(N=50, batch=3)
    res = _solve_np(p, rho=1.0, eps_abs=1e-10, eps_rel=1e-10, max_iter=20000, check_interval=10)
    assert res.status.all()
    n, m, N = p.n, p.m, p.N
    Qbar = np.kron(np.eye(N), p.Q)
    Qbar[-n:, -n:] = p.QN
    Rbar = np.kron(np.eye(N), p.R)
    n_active = 0
    for b in range(p.batch):
        Sx, Su = _condense(p, b)
        H = Rbar + Su.T @ Qbar @ Su
        fvec = Su.T @ Qbar @ Sx @ p.x0[b]
        Cu = np.linalg.cholesky(H).T                 # H = Cu' Cu
        c = -np.linalg.solve(Cu.T, fvec)
        sol = lsq_linear(Cu, c, bounds=(-1.0, 1.0), method="bvls", tol=1e-14, max_iter=2000)
        u_admm = res.z[b].reshape(N, n + m)[:, :m].reshape(-1)
        assert np.abs(u_admm - sol.x).max() < 1e-6
        n_active += int((np.abs(np.abs(sol.x) - 1.0) < 1e-9).sum())
    assert n_active >= 10                                           # the box really binds


def _kkt_certificate(p, b, z, y, rho):
    """T3: separately written optimality check of one QP at (z, lambda = rho y)."""
    P, q, G, bvec = ar.dense_qp(p.A, p.B, p.Q, p.R, p.QN, p.x0[b], p.N, None if p.q is None else p.q[b])
    lo, hi = ar.expand_bounds(p.lo, p.hi, p.N, p.nb)
    lam = rho * y
    feas_dyn = np.abs(G @ z - bvec).max()
    feas_box = max(np.maximum(lo - z, 0).max(), np.maximum(z - hi, 0).max())
    grad = P @ z + q + lam
    nu, *_ = np.linalg.lstsq(G.T, -grad, rcond=None)
    stat = np.abs(grad + G.T @ nu).max()
    at_lo = np.isclose(z, lo, atol=1e-9)
    at_hi = np.isclose(z, hi, atol=1e-9)
    interior = ~(at_lo | at_hi)
    comp = 0.0
    if interior.any():
        comp = max(comp, np.abs(lam[interior]).max())
    if at_lo.any():
        comp = max(comp, np.maximum(lam[at_lo], 0).max())     # lambda <= 0 at a lower bound
    if at_hi.any():
        comp = max(comp, np.maximum(-lam[at_hi], 0).max())    # lambda >= 0 at an upper bound
    return feas_dyn, feas_box, stat, comp


@pytest.mark.parametrize("make,rho", [
    (lambda: pkg.double_integrator(N=50, batch=2), 1.0),
    (lambda: pkg.cw_rendezvous(N=60, batch=2, u_max=0.5), 0.3),
    (lambda: pkg.random_ltv(N=12, n=4, m=2, batch=2, seed=11), 0.5),
])
def test_T3_kkt_certificate(make, rho):
    p = make()
    res = _solve_np(p, rho=rho, eps_abs=1e-10, eps_rel=1e-10, max_iter=60000, check_interval=20)
    assert res.status.all(), res.iters
    for b in range(p.batch):
        feas_dyn, feas_box, stat, comp = _kkt_certificate(p, b, res.z[b], res.y[b], rho)
        assert feas_dyn < 1e-6 and feas_box < 1e-12 and stat < 1e-6 and comp < 1e-6


@pytest.mark.parametrize("alpha,with_q", [(1.0, True), (1.6, True), (1.0, False)])
def test_c_oracle_matches_numpy(alpha, with_q):
    """The C/OpenMP restatement and the NumPy one are written separately; they
    must agree on every iterate and on the per-QP iteration counts."""
    p = pkg.random_ltv(N=40, n=4, m=2, batch=6, seed=5, with_q=with_q)
    kw = dict(rho=0.3, alpha=alpha, max_iter=150, check_interval=7, eps_abs=1e-5, eps_rel=1e-5)
    a = _solve_np(p, **kw)
    c = oc.solve(p, **kw)
    assert a.iters_run == c["iters_run"]
    np.testing.assert_array_equal(a.iters, c["iters"])
    np.testing.assert_array_equal(a.status, c["status"])
    for k in "wzy":
        assert np.abs(getattr(a, k) - c[k]).max() < 1e-12


def test_c_oracle_cw_and_threads():
    """Same result for 1 and several OpenMP threads (QPs are independent)."""
    p = pkg.cw_rendezvous(N=120, batch=9)
    a = oc.solve(p, rho=0.05, max_iter=40, stop=False, nthreads=1)
    b = oc.solve(p, rho=0.05, max_iter=40, stop=False, nthreads=4)
    for k in "wzy":
        np.testing.assert_array_equal(a[k], b[k])
    n = _solve_np(p, rho=0.05, max_iter=40, stop=False)
    assert np.abs(n.z - a["z"]).max() < 1e-12


def test_oracle_factor_agreement():
    p = pkg.random_ltv(N=25, n=6, m=3, batch=1, seed=9)
    f = ar.factor(p.A, p.B, p.Q, p.R, p.QN, 0.2, p.N)
    K, Si = oc.factor(p, 0.2)
    assert np.abs(f.K - K).max() < 1e-12 and np.abs(f.Sinv - Si).max() < 1e-12


def test_warm_start_and_elements_on_bounds():
    """z0/y0 honoured; elements exactly on a bound stay there (clip is idempotent)."""
    p = pkg.double_integrator(N=20, batch=2)
    cold = oc.solve(p, rho=1.0, max_iter=30, stop=False)
    warm = oc.solve(p, rho=1.0, max_iter=10, stop=False, z0=cold["z"], y0=cold["y"])
    cont = oc.solve(p, rho=1.0, max_iter=40, stop=False)
    assert np.abs(warm["z"] - cont["z"]).max() < 1e-13
    lo, hi = ar.expand_bounds(p.lo, p.hi, p.N, p.nb)
    w = np.tile(np.where(np.isfinite(hi), hi, 0.3), (2, 1))
    zn, yn = ar.z_update(w, np.zeros_like(w), np.zeros_like(w), lo, hi)
    np.testing.assert_array_equal(zn, w)
    np.testing.assert_array_equal(yn, np.zeros_like(w))


def test_c_oracle_under_sanitizers(tmp_path):
    """the CPU oracle built with -fsanitize=address,undefined runs a small solve
    cleanly (GPU sanitizers are not available on the pool; this covers the checker itself)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "oracle"), "-s", "liboracle_asan.so"], check=True)
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True, check=True).stdout.strip()
    code = f"""
import sys, ctypes as C, numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {os.path.join(root, 'oracle')!r})
import admm_library_amd as pkg
from admm_library_amd import _abi
lib = C.CDLL({os.path.join(root, 'oracle', 'liboracle_asan.so')!r})
p = pkg.random_ltv(N=13, n=4, m=2, batch=5, seed=3)
cp, keep = _abi.marshal_problem(p)
co = _abi.make_options(rho=0.3, max_iter=60, check_interval=4, adapt_interval=8, adapt_mu=2.0)
z = np.zeros((5, p.L)); y = np.zeros((5, p.L)); w = np.zeros((5, p.L))
it = np.zeros(5, np.int32); st = np.zeros(5, np.int32); r = np.zeros(5); s = np.zeros(5); run = C.c_int32()
lib.oracle_solve.argtypes = [C.POINTER(_abi.CProblem), C.POINTER(_abi.COptions), C.c_int32] + [_abi.c_double_p] * 3 + [_abi.c_int32_p] * 2 + [_abi.c_double_p] * 2 + [_abi.c_int32_p, C.c_int32, _abi.c_double_p, _abi.c_int32_p]
rho = C.c_double(); upd = C.c_int32()
rc = lib.oracle_solve(C.byref(cp), C.byref(co), 1, _abi.dptr(z), _abi.dptr(y), _abi.dptr(w), _abi.iptr(it), _abi.iptr(st), _abi.dptr(r), _abi.dptr(s), C.byref(run), 2, C.byref(rho), C.byref(upd))
assert rc == 0 and np.isfinite(z).all()
print("ok", run.value)
"""
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stderr[-2000:]

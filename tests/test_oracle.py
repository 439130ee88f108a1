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
    return ar.solve(p.A, p.B, p.Q, p.R, p.QN, p.x0, p.lo, p.hi, p.N, q=p.q, unorm=p.unorm, **kw)


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


from _kkt import kkt_certificate as _kkt_certificate      # shared with tests/test_gpu_mfma.py


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


# ---------------------------------------------------------------------------
# Thrust-magnitude (second-order-cone) constraint, DESIGN.md §2.7
# ---------------------------------------------------------------------------

def test_soc_projection_is_the_euclidean_projection():
    rng = np.random.default_rng(0)
    N, n, m = 7, 3, 3
    un = np.array([0.5, np.inf, 0.2, 1.0, np.inf, 0.7, 0.3])
    lo = np.tile(np.r_[np.full(m, -np.inf), -np.ones(n)], N)
    hi = np.tile(np.r_[np.full(m, np.inf), np.ones(n)], N)
    v = rng.standard_normal((50, N * (n + m))) * 1.5
    z = ar.project(v, lo, hi, un, m)
    zb, vb = z.reshape(50, N, n + m), v.reshape(50, N, n + m)
    nr = np.linalg.norm(zb[:, :, :m], axis=2)
    assert (nr <= un[None] * (1 + 1e-15)).all()                       # feasible
    assert np.allclose(ar.project(z, lo, hi, un, m), z, atol=1e-15)  # idempotent
    # optimality of the projection: <v - z, c - z> <= 0 for feasible c (variational inequality)
    c = ar.project(rng.standard_normal(v.shape), lo, hi, un, m)
    assert ((v - z) * (c - z)).sum(1).max() <= 1e-12
    # stages without a bound keep the box (here: open) on the control rows
    np.testing.assert_array_equal(zb[:, 1, :m], vb[:, 1, :m])


def test_soc_qp_vs_scipy_slsqp():
    """T2 for the ball constraint: a tiny instance solved by SciPy's SLSQP on the condensed problem."""
    from scipy.optimize import minimize
    rng = np.random.default_rng(5)
    N, n, m = 6, 2, 2
    A = np.array([[1.0, 0.3], [0.0, 1.0]])
    B = np.array([[0.05, 0.0], [0.3, 0.2]])
    inf = np.inf
    p = pkg.Problem(N=N, A=A, B=B, Q=np.eye(n), R=0.1 * np.eye(m), QN=5 * np.eye(n), x0=np.array([[3.0, -1.0], [-2.0, 2.0]]),
                    lo=np.full(n + m, -inf), hi=np.full(n + m, inf), unorm=np.float64(0.8))
    res = _solve_np(p, rho=1.0, eps_abs=1e-10, eps_rel=1e-10, max_iter=50000, check_interval=10)
    assert res.status.all()
    Qbar = np.kron(np.eye(N), p.Q); Qbar[-n:, -n:] = p.QN
    Rbar = np.kron(np.eye(N), p.R)
    n_active = 0
    for b in range(p.batch):
        Sx, Su = _condense(p, b)
        H = Rbar + Su.T @ Qbar @ Su
        f = Su.T @ Qbar @ Sx @ p.x0[b]
        cons = [{"type": "ineq", "fun": (lambda u, k=k: 0.8 ** 2 - u[k * m:(k + 1) * m] @ u[k * m:(k + 1) * m]),
                 "jac": (lambda u, k=k: np.r_[np.zeros(k * m), -2 * u[k * m:(k + 1) * m], np.zeros((N - k - 1) * m)])}
                for k in range(N)]
        sol = minimize(lambda u: 0.5 * u @ H @ u + f @ u, np.zeros(N * m), jac=lambda u: H @ u + f, constraints=cons,
                       method="SLSQP", options={"ftol": 1e-12, "maxiter": 1000})
        u_admm = res.z[b].reshape(N, n + m)[:, :m].reshape(-1)
        obj = lambda u: 0.5 * u @ H @ u + f @ u
        # SLSQP may stop on its line search right at the optimum; what matters is that it found the same point
        assert sol.success or abs(obj(sol.x) - obj(u_admm)) < 1e-9, sol.message
        assert np.abs(u_admm - sol.x).max() < 1e-4
        assert obj(u_admm) <= obj(sol.x) + 1e-8            # ADMM's point is at least as good (and feasible)
        n_active += int((np.abs(np.linalg.norm(u_admm.reshape(N, m), axis=1) - 0.8) < 1e-7).sum())
    assert n_active >= 2


def test_soc_kkt_certificate():
    """T3 for the ball: lambda_u = rho y_u lies in the normal cone of the ball at z_u."""
    p = pkg.cw_rendezvous(N=50, batch=3, u_max=0.3, thrust_norm=True)
    rho = 0.3
    res = _solve_np(p, rho=rho, eps_abs=1e-10, eps_rel=1e-10, max_iter=60000, check_interval=20)
    assert res.status.all()
    for b in range(p.batch):
        P, q, G, bvec = ar.dense_qp(p.A, p.B, p.Q, p.R, p.QN, p.x0[b], p.N)
        z, lam = res.z[b], rho * res.y[b]
        assert np.abs(G @ z - bvec).max() < 1e-6
        nu, *_ = np.linalg.lstsq(G.T, -(P @ z + q + lam), rcond=None)
        assert np.abs(P @ z + q + lam + G.T @ nu).max() < 1e-6
        zu = z.reshape(p.N, 9)[:, :3]; lu = lam.reshape(p.N, 9)[:, :3]; lx = lam.reshape(p.N, 9)[:, 3:]
        nr = np.linalg.norm(zu, axis=1)
        assert (nr <= 0.3 + 1e-12).all() and np.abs(lx).max() < 1e-9
        inside = nr < 0.3 - 1e-9
        assert np.abs(lu[inside]).max(initial=0.0) < 1e-7                 # no multiplier inside the ball
        on = ~inside
        assert on.sum() >= 3
        kappa = (lu[on] * zu[on]).sum(1) / (0.3 ** 2)
        assert (kappa >= -1e-9).all() and np.abs(lu[on] - kappa[:, None] * zu[on]).max() < 1e-7   # lambda = kappa z, kappa >= 0


def test_soc_c_oracle_matches_numpy():
    for p, rho in ((pkg.cw_rendezvous(N=60, batch=5, thrust_norm=True), 0.05),
                   (pkg.random_ltv(N=23, n=6, m=3, batch=5, seed=3, thrust_norm=True), 0.4)):
        kw = dict(rho=rho, alpha=1.3, max_iter=200, check_interval=10)
        c = oc.solve(p, **kw)
        a = _solve_np(p, **kw)
        assert c["iters_run"] == a.iters_run
        np.testing.assert_array_equal(c["iters"], a.iters)
        for k in "wzy":
            assert np.abs(getattr(a, k) - c[k]).max() < 1e-12

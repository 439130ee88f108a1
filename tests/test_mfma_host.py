"""Host packing of the MFMA form (csrc/admm_mfma_layout.hpp): the per-stage fragment records
admm_setup uploads for ADMM_PRECISION_FP64_MFMA / _MIXED, pushed through a NumPy emulation of
v_mfma_{f64,f32}_16x16x4 (operand and accumulator lane maps of cdna_hip_programming.md §3) with the slot
addressing of the kernels, against the plain block formulas of the one-lane kernels (records of admm_host_factor).
CPU only: catches index / sign / folding errors before a GPU is involved."""
import numpy as np
import pytest

import admm_library_amd as pkg
from admm_library_amd.solver import host_factor, host_factor_mfma


def _layout(n, m):
    xt, nr = 1 if m > 4 else 0, (n + 3) // 4
    return dict(xt=xt, nr=nr, urows=max(0, m - 4), ks_sub=2 * nr + 1 + xt, ot_sub=1, ks_ef=2 * nr + 1 + xt, ks_eb=nr + 1 + xt)


def _urows(table, regs, L):
    """Rows 4.. of u as the kernels form them: every lane multiplies its own slot of each k-step register by the
    UROW coefficient [row - 4][ks][g]; the four lane groups of a column are then added."""
    t = table.reshape(L["urows"], L["ks_sub"], 4)
    return np.array([sum((t[j, ks][:, None] * regs[ks]).sum(0) for ks in range(L["ks_sub"])) for j in range(L["urows"])])


def _blocks(rec, offs):
    return {k: rec[o:o + r * c].reshape(r, c) for k, (o, r, c) in offs.items()}


def _even(v):
    return (v + 1) & ~1


def _rec_f(n, m):
    o = {}; p = 0
    for k, (r, c) in (("PSI", (m, n)), ("K", (m, n)), ("A", (n, n)), ("B", (n, m))):
        o[k] = (p, r, c); p += _even(r * c)
    return o


def _rec_b(n, m):
    o = {}; p = 0
    for k, (r, c) in (("AT", (n, n)), ("BT", (m, n)), ("SI", (m, m)), ("KT", (n, m)), ("OM", (n, m))):
        o[k] = (p, r, c); p += _even(r * c)
    o["LO"] = (p, 1, n + m); p += _even(n + m)
    o["HI"] = (p, 1, n + m)
    return o


def _rec_fe(n, m):
    o = {}; p = 0
    for k, (r, c) in (("PSI", (m, n)), ("K", (m, n)), ("A", (n, n)), ("B", (n, m)), ("FM", (n, n)), ("GA", (n, m)),
                      ("PI", (n, n)), ("DK", (m, n)), ("DG", (m, m)), ("OB", (n, m))):
        o[k] = (p, r, c); p += _even(r * c)
    return o


def _rec_be(n, m):
    o = {}; p = 0
    for k, (r, c) in (("PSB", (m, n)), ("KB", (m, n)), ("AI", (n, n)), ("AIB", (n, m))):
        o[k] = (p, r, c); p += _even(r * c)
    return o


def _to_slots(vec_n=None, vec_m=None, n=0, m=0):
    """Registers of a block vector: regs 0..2 = the n-vector (row 4 r + g), reg 3 = rows 0..3 of the m-vector,
    reg 4 = rows 4..7.  Each register is (4 lane groups, 16 QPs)."""
    regs = np.zeros((5, 4, 16))
    for r in range(3):
        for g in range(4):
            if vec_n is not None and 4 * r + g < n:
                regs[r, g] = vec_n[4 * r + g]
    for g in range(4):
        if vec_m is not None and g < m:
            regs[3, g] = vec_m[g]
        if vec_m is not None and 4 + g < m:
            regs[4, g] = vec_m[4 + g]
    return regs


def _mfma(frags, b_regs, ot_n, f32, acc=None):
    """frags (ks, ot, 64), b_regs list of (4, 16) registers (one per k-step) -> out tiles (ot, 4 regs, 4 g, 16)."""
    dt = np.float32 if f32 else np.float64
    out = np.zeros((ot_n, 16, 16), dt) if acc is None else acc.astype(dt)
    for ks, b in enumerate(b_regs):
        for ot in range(ot_n):
            a = frags[ks, ot].astype(dt).reshape(4, 16).T          # lane l = i + 16 kk  ->  A[i][kk]
            out[ot] += a @ b.astype(dt)                             # B[kk][j] = register value of lane group kk, column j
    tiles = np.zeros((ot_n, 4, 4, 16))
    for ot in range(ot_n):
        for r in range(4):
            for g in range(4):
                tiles[ot, r, g] = out[ot, 4 * g + r if f32 else g + 4 * r]
    return tiles


@pytest.mark.parametrize("mode", [2, 1], ids=["fp64", "mixed"])
@pytest.mark.parametrize("n,m", [(12, 6), (6, 3), (10, 4), (9, 8), (3, 1)])
def test_mfma_records_reproduce_the_stage_operators(built, n, m, mode):
    # element size per product (admm_mfma_layout.hpp): mixed = SUB_F and ELIM_B in fp32, ELIM_F and SUB_B in fp64
    es = {"sub_f": 4 if mode == 1 else 8, "elim_f": 8, "sub_b": 8, "elim_b": 4 if mode == 1 else 8}
    tol_of = lambda e: 2e-5 if e == 4 else 1e-12
    dt_of = lambda e: np.float32 if e == 4 else np.float64
    p = pkg.random_ltv(N=6, n=n, m=m, batch=2, seed=100 + n + m, with_q=False)
    rho, S = 0.4, 2
    hf = host_factor(p, rho, S)
    recMF, recMB, alt_ok = host_factor_mfma(p, rho, S, mode)
    assert alt_ok and hf["alt_ok"]
    L = _layout(n, m)
    rng = np.random.default_rng(5)
    for k in (0, 3, 5):
        F = _blocks(hf["recF"][k], _rec_f(n, m)); B_ = _blocks(hf["recB"][k], _rec_b(n, m))
        FE = _blocks(hf["recFE"][k], _rec_fe(n, m)); BE = _blocks(hf["recBE"][k], _rec_be(n, m))
        x, t, mu, gx, p_ = (rng.standard_normal((n, 16)) for _ in range(5))
        d, gu = (rng.standard_normal((m, 16)) for _ in range(2))
        nf_sub = L["ks_sub"] * L["ot_sub"]
        # ---------------- forward record ----------------
        o_el = nf_sub * 64 * es["sub_f"]
        o_lh = o_el + 2 * L["ks_ef"] * 64 * es["elim_f"]
        sub = recMF[k][:o_el].view(dt_of(es["sub_f"])).reshape(L["ks_sub"], L["ot_sub"], 64)
        eli = recMF[k][o_el:o_lh].view(dt_of(es["elim_f"])).reshape(L["ks_ef"], 2, 64)
        f32, tol = es["sub_f"] == 4, tol_of(es["sub_f"])
        X, T, D = _to_slots(x, None, n, m), _to_slots(t, None, n, m), _to_slots(None, d, n, m)
        NR = L["nr"]
        regs = list(X[:NR]) + list(T[:NR]) + [D[3]] + ([D[4]] if L["xt"] else [])
        o = _mfma(sub, regs, L["ot_sub"], f32)
        u_ref = -(F["K"] @ x + F["PSI"] @ t + d)
        x_ref = F["A"] @ x + F["B"] @ u_ref
        urow = recMF[k][o_lh + 320:].view(np.float64)
        got_x = np.array([o[0, r, g] for r in range(3) for g in range(4)])[:n]
        got_u = np.concatenate([np.array([o[0, 3, g] for g in range(4)])[:min(m, 4)]] + ([_urows(urow, regs, L)] if L["urows"] else []))
        assert np.abs(got_x - x_ref).max() <= tol * max(1, np.abs(x_ref).max())
        assert np.abs(got_u - u_ref).max() <= tol * max(1, np.abs(u_ref).max())
        if n < 12:                                               # padding slots of the n-vector come out as exact zeros
            assert all(np.all(o[0, r, g] == 0) for r in range(3) for g in range(4) if 4 * r + g >= n)
        M, G = _to_slots(mu, None, n, m), _to_slots(gx, gu, n, m)
        regs = list(M[:NR]) + list(G[:NR]) + [G[3]] + ([G[4]] if L["xt"] else [])
        f32, tol = es["elim_f"] == 4, tol_of(es["elim_f"])
        eps0 = rng.standard_normal((n, 16))
        acc = np.zeros((2, 16, 16))
        E = _to_slots(eps0, None, n, m)
        for r in range(3):
            for g in range(4):
                acc[1, 4 * g + r if f32 else g + 4 * r] = E[r, g]
        o = _mfma(eli, regs, 2, f32, acc)
        db_ref = FE["DK"] @ mu + FE["DG"] @ gu
        mu_ref = FE["FM"] @ mu + FE["GA"] @ gu + FE["PI"] @ gx
        eps_ref = eps0 + FE["OB"] @ db_ref
        got_mu = np.array([o[0, r, g] for r in range(3) for g in range(4)])[:n]
        got_eps = np.array([o[1, r, g] for r in range(3) for g in range(4)])[:n]
        got_db = np.array([o[0, 3, g] for g in range(4)] + [o[1, 3, g] for g in range(4)])[:m]
        for got, ref in ((got_mu, mu_ref), (got_eps, eps_ref), (got_db, db_ref)):
            assert np.abs(got - ref).max() <= tol * max(1, np.abs(ref).max(), np.abs(FE["DK"]).max())
        # lo / hi in slot order
        lohi = recMF[k][o_lh:o_lh + 320].view(np.float64)
        lo_b, hi_b = B_["LO"][0], B_["HI"][0]
        for r in range(5):
            for g in range(4):
                row = (m + 4 * r + g if 4 * r + g < n else -1) if r < 3 else (g if r == 3 else 4 + g)
                if r >= 3 and row >= m:
                    row = -1
                assert lohi[r * 4 + g] == (lo_b[row] if row >= 0 else -np.inf)
                assert lohi[20 + r * 4 + g] == (hi_b[row] if row >= 0 else np.inf)
        # ---------------- backward record ----------------
        o_el = nf_sub * 64 * es["sub_b"]
        o_lh = o_el + 2 * L["ks_eb"] * 64 * es["elim_b"]
        sub = recMB[k][:o_el].view(dt_of(es["sub_b"])).reshape(L["ks_sub"], L["ot_sub"], 64)
        eli = recMB[k][o_el:o_lh].view(dt_of(es["elim_b"])).reshape(L["ks_eb"], 2, 64)
        f32, tol = es["sub_b"] == 4, tol_of(es["sub_b"])
        regs = list(X[:NR]) + list(T[:NR]) + [D[3]] + ([D[4]] if L["xt"] else [])     # x_{k+1}, m_in, db
        o = _mfma(sub, regs, L["ot_sub"], f32)
        u_ref = -(BE["KB"] @ x + BE["PSB"] @ t + d)
        xk_ref = BE["AI"] @ x + BE["AIB"] @ u_ref
        urow = recMB[k][o_lh + 320:].view(np.float64)
        got_x = np.array([o[0, r, g] for r in range(3) for g in range(4)])[:n]
        got_u = np.concatenate([np.array([o[0, 3, g] for g in range(4)])[:min(m, 4)]] + ([_urows(urow, regs, L)] if L["urows"] else []))
        scale = max(1, np.abs(BE["KB"]).max())
        assert np.abs(got_x - xk_ref).max() <= tol * scale * max(1, np.abs(xk_ref).max())
        assert np.abs(got_u - u_ref).max() <= tol * scale * max(1, np.abs(u_ref).max())
        f32, tol = es["elim_b"] == 4, tol_of(es["elim_b"])
        Pp, Gu = _to_slots(p_, None, n, m), _to_slots(None, gu, n, m)
        regs = list(Pp[:NR]) + [Gu[3]] + ([Gu[4]] if L["xt"] else [])
        o = _mfma(eli, regs, 2, f32)
        h = B_["BT"] @ p_ + gu
        d_ref = B_["SI"] @ h
        t_ref = B_["AT"] @ p_ - B_["KT"] @ h
        e_ref = B_["OM"] @ d_ref
        got_t = np.array([o[0, r, g] for r in range(3) for g in range(4)])[:n]
        got_e = np.array([o[1, r, g] for r in range(3) for g in range(4)])[:n]
        got_d = np.array([o[0, 3, g] for g in range(4)] + [o[1, 3, g] for g in range(4)])[:m]
        for got, ref in ((got_t, t_ref), (got_e, e_ref), (got_d, d_ref)):
            assert np.abs(got - ref).max() <= tol * max(1, np.abs(ref).max())


def test_mfma_record_sizes_and_unsupported_dims(built):
    import ctypes as C
    lib = pkg.load_library()
    f, b = C.c_int32(), C.c_int32()
    assert lib.admm_mfma_record_bytes(12, 6, 2, C.byref(f), C.byref(b)) == 0      # fp64: every fragment 512 B
    assert (f.value, b.value) == ((8 + 16) * 512 + 320 + 2 * 8 * 4 * 8, (8 + 10) * 512 + 320 + 2 * 8 * 4 * 8)
    assert lib.admm_mfma_record_bytes(6, 3, 1, C.byref(f), C.byref(b)) == 0       # mixed: SUB_F, ELIM_B fragments 256 B
    assert (f.value, b.value) == (5 * 256 + 10 * 512 + 320, 5 * 512 + 6 * 256 + 320)          # n = 6: two registers per n-vector
    assert lib.admm_mfma_record_bytes(13, 3, 2, C.byref(f), C.byref(b)) != 0      # n > 12
    assert lib.admm_mfma_record_bytes(6, 9, 2, C.byref(f), C.byref(b)) != 0       # m > 8
    assert lib.admm_mfma_record_bytes(6, 3, 0, C.byref(f), C.byref(b)) != 0

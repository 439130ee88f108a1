"""NumPy emulation of the HIP kernels' segmented (parallel-in-time) x-update,
driven by the packed factor records libadmm_hip.so builds on the host
(admm_host_factor).  Test helper: lets the CPU-only suite check the host
factorisation and the segment algebra of DESIGN.md §4.2 against the oracle's
plain sequential sweep without a GPU.  Mirrors xb_kernel / xscan_kernel /
xf_kernel of csrc/admm_kernels.hpp statement by statement."""
import numpy as np


def _even(v):
    return (v + 1) & ~1


def unpack(rec, n, m):
    """Split the packed records (layout: csrc/admm_layout.hpp -- every block starts at an
    even offset, zero padded)."""
    N = rec["recB"].shape[0]
    rb, rf, rs = rec["recB"], rec["recF"], rec["recS"]
    nb = n + m

    def take(a, o, rows, cols):
        blk = a[:, o:o + rows * cols].reshape(N, rows, cols) if cols > 1 or rows > 1 else a[:, o:o + 1]
        pad = a[:, o + rows * cols:o + _even(rows * cols)]
        assert not pad.any()
        return blk, o + _even(rows * cols)

    o = 0
    AT, o = take(rb, o, n, n)
    BT, o = take(rb, o, m, n)
    SI, o = take(rb, o, m, m)
    KT, o = take(rb, o, n, m)
    OM, o = take(rb, o, n, m)
    LO = rb[:, o:o + nb]; o += _even(nb)
    HI = rb[:, o:o + nb]; o += _even(nb)
    UB = rb[:, o]; o += 2
    assert o == rb.shape[1]
    o = 0
    PSI, o = take(rf, o, m, n)
    K, o = take(rf, o, m, n)
    A, o = take(rf, o, n, n)
    B, o = take(rf, o, n, m)
    assert np.array_equal(rf[:, o:o + nb], LO) and np.array_equal(rf[:, o + _even(nb):o + _even(nb) + nb], HI)
    o += 2 * _even(nb)
    assert np.array_equal(rf[:, o], UB)
    o += 2
    assert o == rf.shape[1]
    S = rs.shape[0]
    PHI = rs[:, :n * n].reshape(S, n, n)
    XI = rs[:, n * n:2 * n * n].reshape(S, n, n)
    TH = rs[:, 2 * n * n:].reshape(S, n, n)
    AT, BT, SI, KT, OM, PSI, K, A, B = [np.asarray(x).reshape(N, r, c) for x, (r, c) in
                                        zip((AT, BT, SI, KT, OM, PSI, K, A, B),
                                            ((n, n), (m, n), (m, m), (n, m), (n, m), (m, n), (m, n), (n, n), (n, m)))]
    return dict(AT=AT, BT=BT, SI=SI, KT=KT, OM=OM, PSI=PSI, K=K, A=A, B=B, PHI=PHI, XI=XI, TH=TH, LO=LO, HI=HI, UB=UB)


def x_update_segmented(rec, n, m, g, x0, scan="chain", return_parts=False):
    """g: (batch, L) linear term, x0: (batch, n).  Returns w (batch, L).
    scan = "chain": xscan_kernel's sequential recurrences;  "gemm": xscan_mfma_kernel's
    single dense product with the host-built matrix rec["scanW"]."""
    u = unpack(rec, n, m)
    seg = rec["seg_start"]
    S = len(seg) - 1
    N = rec["recB"].shape[0]
    nb = n + m
    batch = g.shape[0]
    gb = g.reshape(batch, N, nb)
    d0 = np.zeros((N, batch, m))
    tseg = np.zeros((S, batch, n))
    eseg = np.zeros((S, batch, n))
    # xb_kernel: every segment independently, tail 0 on entry
    for s in range(S):
        t = np.zeros((batch, n))
        e = np.zeros((batch, n))
        for k in range(seg[s + 1] - 1, seg[s] - 1, -1):
            p = gb[:, k, m:] + t
            h = gb[:, k, :m] + p @ u["BT"][k].T
            d = h @ u["SI"][k].T
            d0[k] = d
            t = p @ u["AT"][k].T - h @ u["KT"][k].T
            e = e + d @ u["OM"][k].T
        tseg[s], eseg[s] = t, e
    if scan == "gemm":
        Wm, Mt = rec["scanW"], rec["scanMt"]
        cin = np.zeros((Wm.shape[1], batch))
        cin[:S * n] = tseg.transpose(0, 2, 1).reshape(S * n, batch)
        cin[S * n:S * n + n] = np.asarray(x0, np.float64).T
        cin[S * n + n:2 * S * n + n] = eseg.transpose(0, 2, 1).reshape(S * n, batch)
        out = Wm @ cin
        tin_g = out[:S * n].reshape(S, n, batch).transpose(0, 2, 1)
        xin_g = out[Mt:Mt + S * n].reshape(S, n, batch).transpose(0, 2, 1)
    # xscan_kernel
    tin = np.zeros((S, batch, n))
    t = np.zeros((batch, n))
    for s in range(S - 1, -1, -1):
        tin[s] = t
        if s == 0:
            break
        t = tseg[s] + t @ u["PHI"][s].T
    xin = np.zeros((S, batch, n))
    x = np.array(x0, dtype=np.float64)
    for s in range(S):
        xin[s] = x
        if s == S - 1:
            break
        x = eseg[s] + tin[s] @ u["XI"][s].T + x @ u["TH"][s].T
    if scan == "gemm":
        chain = (tin, xin)
        tin, xin = tin_g, xin_g
    # xf_kernel
    w = np.zeros((batch, N, nb))
    for s in range(S):
        x = xin[s]
        for k in range(seg[s], seg[s + 1]):
            d = d0[k] + tin[s] @ u["PSI"][k].T
            uu = -(d + x @ u["K"][k].T)
            x = x @ u["A"][k].T + uu @ u["B"][k].T
            w[:, k, :m] = uu
            w[:, k, m:] = x
    if return_parts:
        return w.reshape(batch, N * nb), dict(tin=tin, xin=xin, chain=chain if scan == "gemm" else None)
    return w.reshape(batch, N * nb)


# ---------------------------------------------------------------------------
# Alternating-direction iteration: the forward-elimination /
# backward-substitution form of the x-update, from the host records recFE /
# recBE and the scan matrix scanWB.  Mirrors the elimination half of
# xfze_kernel, xscan_mfma_kernel and the substitution half of xbze_kernel.
# ---------------------------------------------------------------------------
def _blocks(a, spec):
    N = a.shape[0]
    out, o = {}, 0
    for name, rows, cols in spec:
        out[name] = a[:, o:o + rows * cols].reshape(N, rows, cols)
        assert not a[:, o + rows * cols:o + _even(rows * cols)].any()
        o += _even(rows * cols)
    return out, o


def unpack_alt(rec, n, m):
    nb = n + m
    fe, o = _blocks(rec["recFE"], (("PSI", m, n), ("K", m, n), ("A", n, n), ("B", n, m), ("FM", n, n), ("GA", n, m),
                                   ("PI", n, n), ("DK", m, n), ("DG", m, m), ("OB", n, m)))
    assert o + 2 * _even(nb) + 2 == rec["recFE"].shape[1]
    be, o = _blocks(rec["recBE"], (("PSB", m, n), ("KB", m, n), ("AI", n, n), ("AIB", n, m), ("AT", n, n),
                                   ("BT", m, n), ("SI", m, m), ("KT", n, m), ("OM", n, m)))
    assert o + 2 * _even(nb) + 2 == rec["recBE"].shape[1]
    u = unpack(rec, n, m)
    for k in ("PSI", "K", "A", "B"):
        assert np.array_equal(fe[k], u[k])
    for k in ("AT", "BT", "SI", "KT", "OM"):
        assert np.array_equal(be[k], u[k])
    fe.update(be)
    return fe


def x_update_alt(rec, n, m, g, x0):
    """The x-update by forward elimination (segment-local), one dense scan product and backward
    substitution (feedback law + backward rollout).  g: (batch, L), x0: (batch, n).  Returns w (batch, L)."""
    a = unpack_alt(rec, n, m)
    seg = rec["seg_start"]
    S = len(seg) - 1
    N = rec["recFE"].shape[0]
    nb = n + m
    batch = g.shape[0]
    gb = g.reshape(batch, N, nb)
    db = np.zeros((N, batch, m)); mseg = np.zeros((S, batch, n)); eseg = np.zeros((S, batch, n))
    for s in range(S):                                   # elimination half of xfze_kernel
        mu = np.zeros((batch, n)); ee = np.zeros((batch, n))
        for k in range(seg[s], seg[s + 1]):
            gu, gx = gb[:, k, :m], gb[:, k, m:]
            d = mu @ a["DK"][k].T + gu @ a["DG"][k].T
            db[k] = d
            ee = ee + d @ a["OB"][k].T
            mu = mu @ a["FM"][k].T + gu @ a["GA"][k].T + gx @ a["PI"][k].T
        mseg[s], eseg[s] = mu, ee
    Wm, Mt = rec["scanWB"], rec["scanMt"]                # the scan (same kernel, other matrix)
    cin = np.zeros((Wm.shape[1], batch))
    cin[:S * n] = mseg.transpose(0, 2, 1).reshape(S * n, batch)
    cin[S * n:S * n + n] = np.asarray(x0, np.float64).T
    cin[S * n + n:2 * S * n + n] = eseg.transpose(0, 2, 1).reshape(S * n, batch)
    out = Wm @ cin
    min_ = out[:S * n].reshape(S, n, batch).transpose(0, 2, 1)
    xend = out[Mt:Mt + S * n].reshape(S, n, batch).transpose(0, 2, 1)
    w = np.zeros((batch, N, nb))
    for s in range(S):                                   # substitution half of xbze_kernel
        x = xend[s].copy()
        for k in range(seg[s + 1] - 1, seg[s] - 1, -1):
            d = db[k] + min_[s] @ a["PSB"][k].T
            u = -(x @ a["KB"][k].T) - d
            w[:, k, :m] = u
            w[:, k, m:] = x
            x = x @ a["AI"][k].T + u @ a["AIB"][k].T
    return w.reshape(batch, N * nb)

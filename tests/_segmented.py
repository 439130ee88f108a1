"""NumPy emulation of the HIP kernels' segmented (parallel-in-time) x-update,
driven by the packed factor records libadmm_hip.so builds on the host
(admm_host_factor).  Test helper: lets the CPU-only suite check the host
factorisation and the segment algebra of DESIGN.md §4.2 against the oracle's
plain sequential sweep without a GPU.  Mirrors xb_kernel / xscan_kernel /
xf_kernel of csrc/admm_kernels.hpp statement by statement."""
import numpy as np


def unpack(rec, n, m):
    N = rec["recB"].shape[0]
    rb, rf, rs = rec["recB"], rec["recF"], rec["recS"]
    o = 0
    AT = rb[:, o:o + n * n].reshape(N, n, n); o += n * n
    BT = rb[:, o:o + m * n].reshape(N, m, n); o += m * n
    SI = rb[:, o:o + m * m].reshape(N, m, m); o += m * m
    KT = rb[:, o:o + n * m].reshape(N, n, m); o += n * m
    OM = rb[:, o:o + n * m].reshape(N, n, m); o += n * m
    LO, HI = rb[:, o:o + n + m], rb[:, o + n + m:o + 2 * (n + m)]; o += 2 * (n + m)
    assert o == rb.shape[1]
    o = 0
    PSI = rf[:, o:o + m * n].reshape(N, m, n); o += m * n
    K = rf[:, o:o + m * n].reshape(N, m, n); o += m * n
    A = rf[:, o:o + n * n].reshape(N, n, n); o += n * n
    B = rf[:, o:o + n * m].reshape(N, n, m); o += n * m
    assert np.array_equal(rf[:, o:o + n + m], LO) and np.array_equal(rf[:, o + n + m:o + 2 * (n + m)], HI)
    o += 2 * (n + m)
    assert o == rf.shape[1]
    S = rs.shape[0]
    PHI = rs[:, :n * n].reshape(S, n, n)
    XI = rs[:, n * n:2 * n * n].reshape(S, n, n)
    TH = rs[:, 2 * n * n:].reshape(S, n, n)
    return dict(AT=AT, BT=BT, SI=SI, KT=KT, OM=OM, PSI=PSI, K=K, A=A, B=B, PHI=PHI, XI=XI, TH=TH, LO=LO, HI=HI)


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

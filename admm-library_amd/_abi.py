"""ctypes mirror of include/admm_hip.h (structs + marshalling of a Problem).

Kept free of any library loading so that both the product binding
(`solver.py` -> libadmm_hip.so) and the test-side oracle wrapper can share one
definition of the ABI structs.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .problems import Problem

ABI_VERSION = 8

ADMM_OK = 0
STATUS_NAMES = {0: "ADMM_OK", 1: "ADMM_ERR_INVALID", 2: "ADMM_ERR_UNSUPPORTED",
                3: "ADMM_ERR_NO_DEVICE", 4: "ADMM_ERR_HIP", 5: "ADMM_ERR_NUMERIC",
                6: "ADMM_ERR_ALLOC"}

FLAG_NONE = 0
FLAG_NO_GRAPH = 1
FLAG_UNFUSED = 2
FLAG_SCAN_CHAIN = 4
FLAG_NO_ALTERNATE = 8
FLAG_GRAPH = 16
FLAG_NO_MFMA = 32
FLAG_HISTORY = 64
FLAG_ROW_MAJOR = 128      # per-instance dynamics: A, B blocks row-major (NumPy order), transposed on the device (ABI v8)

PRECISION_FP64 = 0
PRECISION_MIXED = 1
PRECISION_FP64_MFMA = 2

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)


class CProblem(C.Structure):
    _fields_ = [("N", C.c_int32), ("n", C.c_int32), ("m", C.c_int32), ("batch", C.c_int32),
                ("time_varying", C.c_int32), ("stage_bounds", C.c_int32),
                ("A", c_double_p), ("B", c_double_p), ("Q", c_double_p), ("R", c_double_p),
                ("QN", c_double_p), ("x0", c_double_p), ("lo", c_double_p), ("hi", c_double_p),
                ("q", c_double_p), ("unorm", c_double_p)]


class COptions(C.Structure):
    _fields_ = [("rho", C.c_double), ("alpha", C.c_double), ("eps_abs", C.c_double),
                ("eps_rel", C.c_double), ("max_iter", C.c_int32), ("check_interval", C.c_int32),
                ("segments", C.c_int32), ("device", C.c_int32), ("zrows", C.c_int32),
                ("flags", C.c_int32), ("adapt_interval", C.c_int32), ("adapt_max", C.c_int32),
                ("adapt_mu", C.c_double), ("adapt_tau", C.c_double),
                ("precision_mode", C.c_int32), ("reserved", C.c_int32)]


class CInfo(C.Structure):
    _fields_ = [("iters_run", C.c_int32), ("n_converged", C.c_int32),
                ("max_r", C.c_double), ("max_s", C.c_double), ("solve_ms", C.c_double),
                ("rho", C.c_double), ("rho_updates", C.c_int32), ("mixed_iters", C.c_int32)]


class CPathInfo(C.Structure):
    _fields_ = [("alternating", C.c_int32), ("alt_requested", C.c_int32), ("mfma", C.c_int32), ("xfree", C.c_int32),
                ("segments", C.c_int32), ("auto_segments", C.c_int32), ("scan_form", C.c_int32), ("per_instance", C.c_int32),
                ("alt_check", C.c_double), ("alt_gate", C.c_double), ("scan_growth", C.c_double)]


EXCHANGE_ALLGATHER = 0
# int (*admm_exchange_fn)(void* ctx, void* hip_stream, int32_t op, double* buf, int64_t count)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64)

SCAN_FORMS = {0: "mfma_gemm", 1: "matrix_vector", 2: "sequential_chain", 3: "per_qp"}
KERNEL_FAMILIES = {0: "one_lane_fp64", 1: "mfma_mixed", 2: "mfma_fp64"}


def dptr(a):
    """double* of a C-contiguous float64 array (None -> NULL)."""
    if a is None:
        return c_double_p()
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_double_p)


def iptr(a):
    if a is None:
        return c_int32_p()
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_int32_p)


def _colmajor(mat: np.ndarray) -> np.ndarray:
    """(..., r, c) row-major matrices -> buffer holding each one column-major.  Large stacks (per-instance dynamics: 1.2 GB of
    A at 4096 x 1000 stages) are transposed by torch's threaded copy where torch is importable -- NumPy's strided copy is one
    thread and was most of what an admm_update_problem call cost from Python once the C side took 60 ms."""
    a = np.asarray(mat, np.float64)
    if a.nbytes >= (16 << 20) and a.ndim >= 3:
        try:
            import torch
            return torch.from_numpy(np.ascontiguousarray(a)).transpose(-1, -2).contiguous().numpy()
        except ImportError:
            pass
    return np.ascontiguousarray(np.swapaxes(a, -1, -2))


def marshal_problem(p: Problem, row_major: bool = False):
    """Build a CProblem over freshly laid-out arrays.  Returns (cproblem, keepalive).  row_major (a handle set up with
    FLAG_ROW_MAJOR; per-instance dynamics): A and B are handed over as they are -- NumPy's own order -- instead of being
    transposed block by block on the host (7 GB, ~2 s, at n = 12, 4096 x 1000)."""
    p.validate()
    mat = (lambda m: np.ascontiguousarray(m, np.float64)) if (row_major and p.per_instance) else _colmajor
    keep = {
        "A": mat(p.A), "B": mat(p.B), "Q": _colmajor(p.Q), "R": _colmajor(p.R),
        "QN": _colmajor(p.QN),
        "x0": np.ascontiguousarray(p.x0, np.float64),
        "lo": np.ascontiguousarray(p.lo, np.float64),
        "hi": np.ascontiguousarray(p.hi, np.float64),
        "q": None if p.q is None else np.ascontiguousarray(p.q, np.float64),
        "unorm": None if p.unorm is None else np.array(
            np.broadcast_to(np.asarray(p.unorm, np.float64), (p.N,) if p.lo.ndim >= 2 else (1,)), np.float64),
    }
    cp = CProblem(N=p.N, n=p.n, m=p.m, batch=p.batch,
                  time_varying=2 if p.per_instance else int(p.time_varying), stage_bounds=p.lo.ndim - 1,
                  A=dptr(keep["A"]), B=dptr(keep["B"]), Q=dptr(keep["Q"]), R=dptr(keep["R"]),
                  QN=dptr(keep["QN"]), x0=dptr(keep["x0"]), lo=dptr(keep["lo"]),
                  hi=dptr(keep["hi"]), q=dptr(keep["q"]), unorm=dptr(keep["unorm"]))
    return cp, keep


def make_options(rho=0.1, alpha=1.0, eps_abs=1e-6, eps_rel=1e-6, max_iter=4000,
                 check_interval=10, segments=0, device=-1, zrows=0, flags=0,
                 adapt_interval=0, adapt_max=16, adapt_mu=10.0, adapt_tau=2.0, precision_mode=0) -> COptions:
    return COptions(rho=rho, alpha=alpha, eps_abs=eps_abs, eps_rel=eps_rel,
                    max_iter=max_iter, check_interval=check_interval, segments=segments,
                    device=device, zrows=zrows, flags=flags, adapt_interval=adapt_interval,
                    adapt_max=adapt_max, adapt_mu=adapt_mu, adapt_tau=adapt_tau,
                    precision_mode=precision_mode, reserved=0)

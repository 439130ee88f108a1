"""ctypes binding of libadmm_hip.so and the host-side solver front end.

Mirrors the entry-point surface of include/admm_hip.h one to one
(`admm_setup` / `admm_solve` / ... keep their C names as methods or module
functions).  The reference defines no such surface (README.md:1-2 only), so
the names are this repository's own; a MATLAB caller gets the same surface
through matlab/admm_mex.cpp.

No CPU fallback: if the shared library is missing this module raises at load
time, and without a GPU `admm_setup` fails with ADMM_ERR_NO_DEVICE.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import os
from typing import Optional

import numpy as np

from . import _abi
from ._abi import CInfo, COptions, CProblem, c_double_p, c_int32_p, dptr, iptr
from .problems import Problem

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libadmm_hip.so"
_lib = None


class AdmmError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{_abi.STATUS_NAMES.get(code, code)}: {msg}")
        self.code = code


def library_path() -> str:
    """In-tree libadmm_hip.so; ADMM_HIP_LIB overrides it (A/B runs of kernel variants)."""
    return os.environ.get("ADMM_HIP_LIB") or os.path.join(_PKG_DIR, _LIB_NAME)


# name -> (restype, argtypes); every symbol include/admm_hip.h declares.
_SIGNATURES = {
    "admm_default_options": (None, [C.POINTER(COptions)]),
    "admm_setup": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(CProblem), C.POINTER(COptions)]),
    "admm_update_instances": (C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    "admm_set_state": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_double_p]),
    "admm_set_rho": (C.c_int, [C.c_void_p, C.c_double]),
    "admm_solve": (C.c_int, [C.c_void_p, c_double_p, c_double_p, C.POINTER(CInfo)]),
    "admm_solve_begin": (C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    "admm_solve_step": (C.c_int, [C.c_void_p, c_int32_p, c_int32_p, c_double_p, c_double_p]),
    "admm_solve_adapt": (C.c_int, [C.c_void_p, C.c_double, C.c_double, c_int32_p]),
    "admm_solve_end": (C.c_int, [C.c_void_p, C.POINTER(CInfo)]),
    "admm_iterate": (C.c_int, [C.c_void_p, C.c_int32]),
    "admm_run": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "admm_sync": (C.c_int, [C.c_void_p]),
    "admm_step_x": (C.c_int, [C.c_void_p]),
    "admm_step_z": (C.c_int, [C.c_void_p, C.c_int32]),
    "admm_get_residuals": (C.c_int, [C.c_void_p] + [c_double_p] * 5),
    "admm_get": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_double_p]),
    "admm_get_info": (C.c_int, [C.c_void_p, c_int32_p, c_int32_p, c_double_p, c_double_p]),
    "admm_profile": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, c_double_p]),
    "admm_get_geometry": (C.c_int, [C.c_void_p] + [c_int32_p] * 4),
    "admm_get_rho": (C.c_int, [C.c_void_p, c_double_p]),
    "admm_get_history": (C.c_int, [C.c_void_p, C.c_int32, c_int32_p, c_int32_p, c_int32_p, c_double_p, c_double_p, c_double_p]),
    "admm_get_path": (C.c_int, [C.c_void_p, C.POINTER(_abi.CPathInfo)]),
    "admm_setup_timeshard": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(CProblem), C.POINTER(COptions), C.c_int32, C.c_int32,
                                      _abi.EXCHANGE_FN, C.c_void_p]),
    "admm_get_window": (C.c_int, [C.c_void_p] + [c_int32_p] * 5),
    "admm_free": (None, [C.c_void_p]),
    "admm_last_error": (C.c_char_p, []),
    "admm_last_warning": (C.c_char_p, []),
    "admm_abi_version": (C.c_int, []),
    "admm_device_count": (C.c_int, []),
    "admm_record_sizes": (C.c_int, [C.c_int32, C.c_int32, c_int32_p, c_int32_p, c_int32_p]),
    "admm_host_scan_matrix": (C.c_int, [C.POINTER(CProblem), C.c_double, C.c_int32, c_double_p, c_int32_p,
                                        c_int32_p, c_int32_p]),
    "admm_host_factor": (C.c_int, [C.POINTER(CProblem), C.c_double, C.c_int32, c_double_p, c_double_p,
                                   c_double_p, c_double_p, c_double_p, c_int32_p]),
    "admm_host_scan_matrices_timeshard": (C.c_int, [C.POINTER(CProblem), C.c_double, C.c_int32, C.c_int32, c_double_p, c_double_p,
                                                   c_int32_p]),
    "admm_update_problem": (C.c_int, [C.c_void_p, C.POINTER(CProblem)]),
    "admm_record_sizes_alt": (C.c_int, [C.c_int32, C.c_int32, c_int32_p, c_int32_p]),
    "admm_host_factor_alt": (C.c_int, [C.POINTER(CProblem), C.c_double, C.c_int32, c_double_p, c_double_p,
                                       c_double_p, c_int32_p]),
    "admm_mfma_record_bytes": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, c_int32_p, c_int32_p]),
    "admm_host_factor_mfma": (C.c_int, [C.POINTER(CProblem), C.c_double, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                        c_int32_p]),
}


def load_library(path: Optional[str] = None):
    """Load libadmm_hip.so (built in-tree by __graft_entry__.build()).  Raises if absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or library_path()
    if not os.path.exists(p):
        raise FileNotFoundError(
            f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = C.CDLL(p)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.admm_abi_version() != _abi.ABI_VERSION:
        raise RuntimeError("libadmm_hip.so ABI version mismatch")
    if path is None:
        _lib = lib
    return lib


def _check(lib, rc: int):
    if rc != 0:
        msg = lib.admm_last_error().decode()
        if "non-finite entry in A, B" in msg or "non-finite entry in x0" in msg:    # what Problem.validate raises for small arrays (it leaves the large ones to the library)
            raise ValueError("non-finite problem data: " + msg)
        raise AdmmError(rc, msg)


def last_warning() -> str:
    """admm_last_warning(): what the last setup / set_rho / update_problem / solve call on this thread changed about the
    kernels a handle runs ('' = nothing) -- e.g. the forward-elimination gate of the default path failing."""
    return load_library().admm_last_warning().decode()


def device_count() -> int:
    return int(load_library().admm_device_count())


@dataclasses.dataclass
class Options:
    rho: float = 0.1
    alpha: float = 1.0
    eps_abs: float = 1e-6
    eps_rel: float = 1e-6
    max_iter: int = 4000
    check_interval: int = 10
    segments: int = 0
    device: int = -1
    zrows: int = 0
    flags: int = 0
    adapt_interval: int = 0
    adapt_max: int = 16
    adapt_mu: float = 10.0
    adapt_tau: float = 2.0
    precision_mode: int = 0       # _abi.PRECISION_FP64 / _MIXED / _FP64_MFMA

    def to_c(self) -> COptions:
        return _abi.make_options(**dataclasses.asdict(self))


@dataclasses.dataclass
class SolveInfo:
    iters_run: int
    n_converged: int
    max_r: float
    max_s: float
    solve_ms: float
    rho: float
    rho_updates: int
    mixed_iters: int
    iters: np.ndarray
    status: np.ndarray
    r: np.ndarray
    s: np.ndarray


class Solver:
    """One handle = one batch of QPs on one GPU."""

    def __init__(self, problem: Problem, options: Optional[Options] = None, timeshard=None):
        """timeshard = (rank, nranks, exchange): a TIME-SHARDED handle (admm_setup_timeshard; sharding.TimeShardedSolver builds
        the exchange function over torch.distributed) -- `exchange` is an _abi.EXCHANGE_FN the caller keeps alive."""
        self._lib = load_library()
        self._h = C.c_void_p()
        self.problem = problem
        self.options = options or Options()
        co = self.options.to_c()
        # per-instance dynamics: NumPy's row-major blocks go to the library as they are (ADMM_FLAG_ROW_MAJOR, transposed on the
        # device); ADMM_PY_COLMAJOR=1 keeps the host transposition (the column-major path of the ABI)
        self._row_major = bool(problem.per_instance and not os.environ.get("ADMM_PY_COLMAJOR"))
        if self._row_major:
            co.flags |= _abi.FLAG_ROW_MAJOR
        cp, keep = _abi.marshal_problem(problem, self._row_major)
        if timeshard is None:
            _check(self._lib, self._lib.admm_setup(C.byref(self._h), C.byref(cp), C.byref(co)))
        else:
            rank, nranks, fn = timeshard
            self._exchange = fn                     # the C side calls it for as long as the handle lives
            _check(self._lib, self._lib.admm_setup_timeshard(C.byref(self._h), C.byref(cp), C.byref(co), int(rank), int(nranks),
                                                            fn if fn is not None else _abi.EXCHANGE_FN(0), None))
        del keep
        self.batch, self.L = problem.batch, problem.L
        self._warn()

    def _warn(self):
        """Surface admm_last_warning() (the library never changes the kernel path silently)."""
        msg = self._lib.admm_last_warning().decode()
        self.last_warning = msg
        if msg:
            import warnings
            warnings.warn(msg, RuntimeWarning, stacklevel=3)

    # -- lifetime ---------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.admm_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- helpers ----------------------------------------------------------
    def _vec(self, a, rows=None):
        if a is None:
            return None
        a = np.ascontiguousarray(a, np.float64)
        if a.shape != (self.batch, rows or self.L):
            raise ValueError(f"expected shape {(self.batch, rows or self.L)}, got {a.shape}")
        return a

    # -- C ABI, one method per entry point -------------------------------
    def update_instances(self, x0=None, q=None):
        x0 = self._vec(x0, self.problem.n)
        q = self._vec(q)
        _check(self._lib, self._lib.admm_update_instances(self._h, dptr(x0), dptr(q)))

    def update_problem(self, problem: Problem):
        """New shared data (dynamics, weights, box, x0, q) on this handle; same N, n, m, batch."""
        cp, keep = _abi.marshal_problem(problem, self._row_major)        # (validates)
        import time
        t0 = time.perf_counter()
        _check(self._lib, self._lib.admm_update_problem(self._h, C.byref(cp)))
        self.last_update_ms = (time.perf_counter() - t0) * 1e3      # the C call alone (validation, upload, refactor)
        del keep
        self.problem = problem
        self._warn()

    def set_rho(self, rho: float):
        _check(self._lib, self._lib.admm_set_rho(self._h, float(rho)))
        self._warn()

    def set_state(self, w=None, z=None, y=None):
        w, z, y = self._vec(w), self._vec(z), self._vec(y)
        _check(self._lib, self._lib.admm_set_state(self._h, dptr(w), dptr(z), dptr(y)))

    def _info(self, ci: CInfo) -> SolveInfo:
        iters = np.empty(self.batch, np.int32)
        status = np.empty(self.batch, np.int32)
        r = np.empty(self.batch)
        s = np.empty(self.batch)
        _check(self._lib, self._lib.admm_get_info(self._h, iptr(iters), iptr(status), dptr(r), dptr(s)))
        return SolveInfo(ci.iters_run, ci.n_converged, ci.max_r, ci.max_s, ci.solve_ms, ci.rho, ci.rho_updates,
                         ci.mixed_iters, iters, status, r, s)

    def solve(self, z0=None, y0=None) -> SolveInfo:
        z0, y0 = self._vec(z0), self._vec(y0)
        ci = CInfo()
        _check(self._lib, self._lib.admm_solve(self._h, dptr(z0), dptr(y0), C.byref(ci)))
        self._warn()
        return self._info(ci)

    # admm_solve in pieces (global stop / adaptive-rho decisions of a sharded solve)
    def solve_begin(self, z0=None, y0=None):
        z0, y0 = self._vec(z0), self._vec(y0)
        _check(self._lib, self._lib.admm_solve_begin(self._h, dptr(z0), dptr(y0)))

    def solve_step(self, sums: bool = False):
        """Run up to and including the next checked iteration.
        Returns (iterations so far, converged QPs of this handle, R, S)."""
        it, nc = C.c_int32(), C.c_int32()
        R, S = C.c_double(), C.c_double()
        _check(self._lib, self._lib.admm_solve_step(self._h, C.byref(it), C.byref(nc),
                                                    C.byref(R) if sums else None, C.byref(S) if sums else None))
        return it.value, nc.value, R.value, S.value

    def solve_adapt(self, R: float, S: float) -> bool:
        ch = C.c_int32()
        _check(self._lib, self._lib.admm_solve_adapt(self._h, float(R), float(S), C.byref(ch)))
        return bool(ch.value)

    def solve_end(self) -> SolveInfo:
        ci = CInfo()
        _check(self._lib, self._lib.admm_solve_end(self._h, C.byref(ci)))
        self._warn()
        return self._info(ci)

    def iterate(self, iters: int, sync: bool = True):
        _check(self._lib, self._lib.admm_iterate(self._h, int(iters)))
        if sync:
            self.sync()

    def run(self, iters: int, residual_every: int = 0, sync: bool = True):
        _check(self._lib, self._lib.admm_run(self._h, int(iters), int(residual_every)))
        if sync:
            self.sync()

    def sync(self):
        _check(self._lib, self._lib.admm_sync(self._h))

    def step_x(self):
        _check(self._lib, self._lib.admm_step_x(self._h))

    def step_z(self, residuals: bool = False):
        _check(self._lib, self._lib.admm_step_z(self._h, int(bool(residuals))))

    def residuals(self):
        out = [np.empty(self.batch) for _ in range(5)]
        _check(self._lib, self._lib.admm_get_residuals(self._h, *[dptr(o) for o in out]))
        return tuple(out)

    def get(self, w=True, z=True, y=True):
        outs = [np.empty((self.batch, self.L)) if f else None for f in (w, z, y)]
        _check(self._lib, self._lib.admm_get(self._h, *[dptr(o) for o in outs]))
        return tuple(outs)

    def profile(self, iters: int, residuals: bool = True, fused: bool = True, alternating: bool = False,
                back_to_back: bool = False):
        """Per-kernel HIP-event timings (ms).  alternating: `iters` PAIRS of the alternating-direction
        iteration (forward form, backward form) instead of the plain kernels; with
        back_to_back each of the pair's kernels is launched `iters` times in a row between two events
        (admm_profile mode 3: no event-record bubble inside the averages)."""
        ms = np.zeros(6)
        mode = (3 if back_to_back else 2) if alternating else int(bool(fused))
        _check(self._lib, self._lib.admm_profile(self._h, int(iters), int(bool(residuals)), mode, dptr(ms)))
        if alternating:
            return {"xscan_ms": ms[0], "xfze_ms": ms[1], "finalize_xscan_ms": ms[2], "xbze_ms": ms[3],
                    "finalize_ms": ms[4], "pair_ms": ms[5]}
        if fused:
            return {"xb_ms": ms[0], "xscan_ms": ms[1], "xfz_ms": ms[2], "finalize_ms": ms[4], "iter_ms": ms[5]}
        return {"xb_ms": ms[0], "xscan_ms": ms[1], "xf_ms": ms[2], "zdual_ms": ms[3], "finalize_ms": ms[4],
                "iter_ms": ms[5]}

    def history(self) -> dict:
        """Residual history of the last solve (Options(flags=FLAG_HISTORY)): one entry per stopping test --
        iteration, n_converged, max_r, max_s (maxima over the batch), rho."""
        n = C.c_int32()
        _check(self._lib, self._lib.admm_get_history(self._h, 0, C.byref(n), None, None, None, None, None))
        it, nc = np.empty(n.value, np.int32), np.empty(n.value, np.int32)
        r, s, rho = np.empty(n.value), np.empty(n.value), np.empty(n.value)
        _check(self._lib, self._lib.admm_get_history(self._h, n.value, C.byref(n), iptr(it), iptr(nc), dptr(r), dptr(s), dptr(rho)))
        return {"iteration": it, "n_converged": nc, "max_r": r, "max_s": s, "rho": rho}

    def rho_per_qp(self) -> np.ndarray:
        """rho of every QP: the handle's rho for batch-shared dynamics; with per-instance dynamics each QP's own
        (the adaptive rule runs per QP there)."""
        out = np.empty(self.problem.batch)
        _check(self._lib, self._lib.admm_get_rho(self._h, dptr(out)))
        return out

    def geometry(self):
        v = [C.c_int32() for _ in range(4)]
        _check(self._lib, self._lib.admm_get_geometry(self._h, *[C.byref(x) for x in v]))
        return {"pitch": v[0].value, "segments": v[1].value, "zrows": v[2].value, "zchunks": v[3].value}

    def window(self) -> dict:
        """admm_get_window: the stages / segments this handle iterates (the whole horizon unless it is a time shard)."""
        v = [C.c_int32() for _ in range(5)]
        _check(self._lib, self._lib.admm_get_window(self._h, *[C.byref(x) for x in v]))
        return {"stage_lo": v[0].value, "stage_hi": v[1].value, "seg_lo": v[2].value, "segs_local": v[3].value, "segs_total": v[4].value}

    def path(self) -> dict:
        """admm_get_path: which kernels this handle runs and the measured margin of the default path."""
        pi = _abi.CPathInfo()
        _check(self._lib, self._lib.admm_get_path(self._h, C.byref(pi)))
        return {"alternating": bool(pi.alternating), "alt_requested": bool(pi.alt_requested),
                "kernel_family": _abi.KERNEL_FAMILIES[pi.mfma], "xfree": bool(pi.xfree), "segments": pi.segments,
                "auto_segments": bool(pi.auto_segments), "scan_form": _abi.SCAN_FORMS[pi.scan_form],
                "per_instance": bool(pi.per_instance), "alt_check": pi.alt_check, "alt_gate": pi.alt_gate,
                "scan_growth": pi.scan_growth}


def admm_setup(problem: Problem, options: Optional[Options] = None) -> Solver:
    """admm_setup of the C ABI: validate, factor the KKT system, upload."""
    return Solver(problem, options)


def admm_solve(problem_or_solver, options: Optional[Options] = None, z0=None, y0=None):
    """One-call front end: setup (if given a Problem) + solve + read-out.
    Returns (w, z, y, info)."""
    own = not isinstance(problem_or_solver, Solver)
    s = Solver(problem_or_solver, options) if own else problem_or_solver
    try:
        info = s.solve(z0, y0)
        w, z, y = s.get()
    finally:
        if own:
            s.close()
    return w, z, y, info


def host_factor(problem: Problem, rho: float, segments: int):
    """Host-only: the packed factor records exactly as admm_setup uploads them
    (no GPU needed).  Used by the CPU tests of the segment algebra."""
    lib = load_library()
    cp, keep = _abi.marshal_problem(problem)
    N, n, m = problem.N, problem.n, problem.m
    S = max(1, min(segments, N))
    rb, rf, rs = C.c_int32(), C.c_int32(), C.c_int32()
    _check(lib, lib.admm_record_sizes(n, m, C.byref(rb), C.byref(rf), C.byref(rs)))
    K = np.empty((N, m, n)); Sinv = np.empty((N, m, m))
    recB = np.empty((N, rb.value)); recF = np.empty((N, rf.value)); recS = np.empty((S, rs.value))
    seg = np.empty(S + 1, np.int32)
    _check(lib, lib.admm_host_factor(C.byref(cp), float(rho), S, dptr(K), dptr(Sinv), dptr(recB),
                                     dptr(recF), dptr(recS), iptr(seg)))
    M, Mt, Kd = C.c_int32(), C.c_int32(), C.c_int32()
    _check(lib, lib.admm_host_scan_matrix(C.byref(cp), float(rho), S, None, C.byref(M), C.byref(Mt), C.byref(Kd)))
    W = np.empty((M.value, Kd.value))
    _check(lib, lib.admm_host_scan_matrix(C.byref(cp), float(rho), S, dptr(W), None, None, None))
    out = {"K": K, "Sinv": Sinv, "recB": recB, "recF": recF, "recS": recS, "seg_start": seg,
           "scanW": W, "scanMt": Mt.value, "alt_ok": False}
    # records of the alternating-direction iteration
    rfe, rbe, ok = C.c_int32(), C.c_int32(), C.c_int32()
    _check(lib, lib.admm_record_sizes_alt(n, m, C.byref(rfe), C.byref(rbe)))
    recFE = np.empty((N, rfe.value)); recBE = np.empty((N, rbe.value)); WB = np.empty_like(W)
    _check(lib, lib.admm_host_factor_alt(C.byref(cp), float(rho), S, dptr(recFE), dptr(recBE), dptr(WB),
                                         C.byref(ok)))
    if ok.value:
        out.update(alt_ok=True, recFE=recFE, recBE=recBE, scanWB=WB)
    del keep
    return out


def host_factor_mfma(problem: Problem, rho: float, segments: int, mode: int):
    """Host-only: the MFMA fragment records exactly as admm_setup uploads them, as
    (N, bytes) uint8 arrays (forward, backward), plus whether the forward-elimination products are present."""
    lib = load_library()
    cp, keep = _abi.marshal_problem(problem)
    fb, bb, ok = C.c_int32(), C.c_int32(), C.c_int32()
    _check(lib, lib.admm_mfma_record_bytes(problem.n, problem.m, int(mode), C.byref(fb), C.byref(bb)))
    recMF = np.zeros((problem.N, fb.value), np.uint8)
    recMB = np.zeros((problem.N, bb.value), np.uint8)
    _check(lib, lib.admm_host_factor_mfma(C.byref(cp), float(rho), int(segments), int(mode),
                                          recMF.ctypes.data_as(C.c_void_p), recMB.ctypes.data_as(C.c_void_p), C.byref(ok)))
    del keep
    return recMF, recMB, bool(ok.value)

"""Time-sharded handles (ABI v7): the segments of ONE horizon on different ranks, coupled exactly by an all-gather of
the segment summaries before each scan -- BASELINE.json's "shooting segments".  The iterates must be those of one handle holding
every segment (and of the oracle).

Three levels: one rank (no exchange); two ranks emulated by two THREADS of one process with a hand-written exchange (barrier +
device-to-device copies: the protocol itself, no torch.distributed); two PROCESSES over torch.distributed (gloo on the one-GPU box;
the same TimeShardedSolver runs over RCCL with one GPU per rank)."""
import os
import socket
import sys
import threading

import numpy as np
import pytest
import torch

import admm_library_amd as pkg
import oracle_c as oc
from admm_library_amd import _abi
from admm_library_amd.solver import Solver

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
TOL = 1e-10
UPTO = (1, 2, 3, 4, 8, 32)


def _err(got, ref):
    return max(np.abs(a - ref[k]).max() / max(1.0, np.abs(ref[k]).max()) for a, k in zip(got, "wzy"))


def test_one_rank_time_shard_is_the_ordinary_handle(gpu):
    p = pkg.cw_rendezvous(N=240, batch=70)
    with Solver(p, pkg.Options(rho=0.05), timeshard=(0, 1, None)) as s, pkg.Solver(p, pkg.Options(rho=0.05)) as s0:
        assert s.window() == {"stage_lo": 0, "stage_hi": 240, "seg_lo": 0, "segs_local": s.geometry()["segments"], "segs_total": s.geometry()["segments"]}
        s.run(25, residual_every=5)
        s0.run(25, residual_every=5)
        for a, b in zip(s.get() + s.residuals(), s0.get() + s0.residuals()):
            np.testing.assert_array_equal(a, b)


class _ThreadExchange:
    """Two 'ranks' in one process: each handle's exchange publishes its buffer, waits for the peer, copies the peer's slice."""

    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.calls = 0

    def make(self, rank):
        def fn(ctx, stream, op, buf, count):
            try:
                ext = torch.cuda.ExternalStream(stream, device="cuda:0")
                ext.synchronize()                                     # my slice is complete
                self.slots[rank] = (buf, count)
                self.barrier.wait(timeout=60)
                full = pkg.device_tensor(buf, count * self.world, "cuda:0")
                with torch.cuda.stream(ext):
                    for r in range(self.world):
                        if r != rank:
                            pb, pc = self.slots[r]
                            assert pc == count
                            peer = pkg.device_tensor(pb, count * self.world, "cuda:0")
                            full[r * count:(r + 1) * count].copy_(peer[r * count:(r + 1) * count])
                ext.synchronize()
                self.barrier.wait(timeout=60)                         # nobody overwrites a buffer a peer is still reading
                if rank == 0:
                    self.calls += 1
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                self.barrier.abort()
                return 1
        return _abi.EXCHANGE_FN(fn)


CASES = {
    "cw_6_3_alternating": (lambda: pkg.cw_rendezvous(N=400, batch=130), dict(rho=0.05), 0),
    "cw_6_3_plain_path": (lambda: pkg.cw_rendezvous(N=400, batch=130), dict(rho=0.05, flags=_abi.FLAG_NO_ALTERNATE), 0),
    "ltv_q_bounds_alpha": (lambda: pkg.random_ltv(N=90, n=6, m=3, batch=67, seed=77), dict(rho=0.3, alpha=1.6, segments=6), 0),
    "formation_12_6_small_batch_mfma": (lambda: pkg.cw_formation(N=160, batch=40), dict(rho=0.05, segments=8), 0),
    "thrust_magnitude": (lambda: pkg.cw_rendezvous(N=96, batch=20, thrust_norm=True), dict(rho=0.05, segments=4), 0),
    # BASELINE's horizon (round 3: every other case here is small, and a set-up race that needed large arrays went unseen elsewhere)
    "cw_6_3_full_horizon_1024": (lambda: pkg.cw_rendezvous(N=1000, batch=1024), dict(rho=0.05), 0),
    "formation_12_6_full_horizon_512": (lambda: pkg.cw_formation(N=1000, batch=512), dict(rho=0.05), 0),
}


@pytest.mark.parametrize("case", CASES)
def test_two_ranks_in_one_process_match_the_oracle(gpu, case):
    """Two time shards of one batch (threads, hand-written exchange): after 1, 2, 3, 4, 8 and 32 iterations with residuals every 4th,
    the assembled iterates and the residuals equal the oracle's, and both ranks hold the same residuals."""
    make, kw, _ = CASES[case]
    p = make()
    world = 2
    ex = _ThreadExchange(world)
    fns = [ex.make(r) for r in range(world)]
    solvers = [Solver(p, pkg.Options(**kw), timeshard=(r, world, fns[r])) for r in range(world)]
    wins = [s.window() for s in solvers]
    assert wins[0]["stage_lo"] == 0 and wins[0]["stage_hi"] == wins[1]["stage_lo"] and wins[1]["stage_hi"] == p.N
    assert wins[0]["segs_local"] == wins[1]["segs_local"] and wins[0]["segs_total"] == 2 * wins[0]["segs_local"]
    out, errs = {}, []

    def work(r):
        try:
            s, done = solvers[r], 0
            for upto in UPTO:                        # residual evaluations land on the multiples of 4 (the oracle's check_interval)
                every = 0 if upto < 4 else (1 if upto == 4 else 4)
                s.run(upto - done, residual_every=every)
                done = upto
                out[(r, upto)] = s.get() + (s.residuals()[:2] if upto >= 4 else ())
        except Exception as e:                       # noqa: BLE001
            errs.append(e)
            ex.barrier.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    for s in solvers:
        s.close()
    assert not errs, errs
    assert ex.calls > 32                             # at least one exchange per iteration really happened
    nb = p.nb
    alpha = kw.get("alpha", 1.0)
    for upto in UPTO:
        ref = oc.solve(p, rho=kw["rho"], alpha=alpha, max_iter=upto, check_interval=4, stop=False)
        full = []
        for i in range(3):
            a = np.empty((p.batch, p.L))
            for r in range(world):
                lo, hi = wins[r]["stage_lo"] * nb, wins[r]["stage_hi"] * nb
                a[:, lo:hi] = out[(r, upto)][i][:, lo:hi]
            full.append(a)
        assert _err(full, ref) <= TOL, (upto, _err(full, ref))
        if upto >= 4:
            for r in range(world):
                assert np.abs(out[(r, upto)][3] - ref["r"]).max() <= TOL and np.abs(out[(r, upto)][4] - ref["s"]).max() <= TOL
            np.testing.assert_array_equal(out[(0, upto)][3], out[(1, upto)][3])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import admm_library_amd as pkg2
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = pkg2.cw_rendezvous(N=300, batch=37)
    opt = pkg2.Options(rho=0.05, alpha=1.6, eps_abs=1e-6, eps_rel=1e-6, max_iter=3000, check_interval=10, adapt_interval=50, device=0)
    with pkg2.TimeShardedSolver(p, opt, device="cuda:0") as ts:
        info = ts.solve()
        w, z, y = ts.get()
        calls = ts.exchange_stats["calls"]
    dist.barrier()
    np.savez(os.path.join(out_dir, f"ts{rank}.npz"), w=w, z=z, y=y, iters=info.iters, iters_run=info.iters_run, rho=info.rho,
             calls=calls, stage_lo=ts.window["stage_lo"], stage_hi=ts.window["stage_hi"])
    dist.destroy_process_group()


def test_two_process_time_sharded_solve_equals_the_oracle(gpu, tmp_path):
    """admm_solve (adaptive rho, over-relaxation) of one batch time-sharded over two processes (torch.distributed, gloo): same
    iteration count, rho trajectory end, per-QP counts and solution as the unsharded oracle, identical on both ranks."""
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    g = [np.load(tmp_path / f"ts{r}.npz") for r in range(2)]
    p = pkg.cw_rendezvous(N=300, batch=37)
    ref = oc.solve(p, rho=0.05, alpha=1.6, eps_abs=1e-6, eps_rel=1e-6, max_iter=3000, check_interval=10, adapt_interval=50)
    assert g[0]["stage_lo"] == 0 and g[0]["stage_hi"] == g[1]["stage_lo"] and g[1]["stage_hi"] == 300
    for r in range(2):
        assert int(g[r]["iters_run"]) == ref["iters_run"] and float(g[r]["rho"]) == ref["rho"]
        assert (np.abs(g[r]["iters"] - ref["iters"]) <= 10).all()
        for k in "wzy":
            assert np.abs(g[r][k] - ref[k]).max() <= TOL * max(1.0, np.abs(ref[k]).max()), (r, k)
        assert int(g[r]["calls"]) >= ref["iters_run"]                # one all-gather of the segment summaries per iteration at least
    np.testing.assert_array_equal(g[0]["z"], g[1]["z"])

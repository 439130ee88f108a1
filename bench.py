#!/usr/bin/env python3
"""bench.py -- ADMM iterations/sec on the BASELINE.json workload.

    python bench.py --gpus N --steps K --warmup W

Workload (config.workload): BASELINE.json configs[2] -- a batch of 4096
independent N=1000, n=6, m=3 Clohessy-Wiltshire rendezvous QPs per GPU, fp64,
synthetic inputs from admm_library_amd.cw_rendezvous (seeded), resident in HBM
before the timed region.  For N > 1 GPUs the global batch is 4096 * N QPs,
sharded contiguously (weak scaling, no data-path collective: QPs are
independent).

A "step" = one batch-iteration over the rank's shard on the default path, the
alternating-direction iteration: segment scan (xscan), then ONE
fused kernel -- the x-update's substitution sweep, z-update / dual ascent /
residual partials, and the next x-update's elimination sweep (xfze forward on
even iterations, xbze backward on odd ones) -- then the residual finalise
kernel, i.e. every iteration evaluates the residuals on the device
(check_interval = 1), the most expensive honest form of the iteration.
`value` = QP-iterations/s summed over all ranks.  (`plain_path` re-times the
same steps with ADMM_FLAG_NO_ALTERNATE: xb + xscan + xfz + finalise.)

Extra objects on the JSON line:
  roofline     -- the dominant kernels, the pair xfze<RESID> / xbze<RESID>: each
                  moves 16 + 16 m/(n+m) = 21.33 B per stacked element (d, v read
                  and v+, db written, resp. db, v read and v+, d written; fp64,
                  state in v-form) x L x pitch per launch;
                  achieved = the pair's bytes / the pair's average launch
                  durations, measured with HIP events on the library's own stream
                  (admm_profile).  `per_kernel` holds each kernel's own figures.
  roofline_zdual_standalone -- the standalone fused z/dual/residual kernel of
                  the ADMM_FLAG_UNFUSED path (40 B per element),
                  measured the same way in the same run.
  cpu_baseline -- the C/OpenMP CPU oracle (kind "port": the reference ships no
                  code) on a bounded sample of the same workload, rank 0, N=1.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec
BYTES_PER_ELEM_ZDUAL = 40       # w, y, z read; z+, y+ written; fp64
BYTES_PER_ELEM_ZPLAIN = 32      # non-residual form: w, y read; z+, y+ written


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096, help="QPs per GPU")
    ap.add_argument("--horizon", type=int, default=1000)
    ap.add_argument("--workload", choices=["cw_rendezvous", "cw_formation", "cw_rendezvous_soc", "cw_perinstance", "cw_formation_perinstance"], default="cw_rendezvous",
                    help="cw_rendezvous = configs[1..3] (n=6, m=3, the metric's workload); cw_formation = configs[4]'s "
                         "shape (n=12, m=6) in fp64 -- a side measurement, never the reported metric's config")
    ap.add_argument("--precision", choices=["fp64", "fp64_one_lane", "mixed", "fp64_mfma"], default="fp64",
                    help="options.precision_mode of the timed solver (DESIGN.md §4.9): fp64 = the default (the library picks the "
                         "one-lane or the fp64 MFMA kernels: one-lane at the metric's n=6 batch=4096); fp64_one_lane = "
                         "ADMM_FLAG_NO_MFMA; fp64_mfma / mixed = the MFMA forms (configs[4]; compiled for n=12 m=6, n=6 m=3, n=10 m=4)")
    ap.add_argument("--segments", type=int, default=0)
    ap.add_argument("--zrows", type=int, default=0)
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed blocks of --steps steps each; the median block is reported (all blocks are listed)")
    ap.add_argument("--warm-seconds", type=float, default=0.6,
                    help="after the --warmup steps, keep iterating until at least this much wall time has been spent "
                         "warming up, so that the timed region runs at steady clocks however small --warmup is")
    ap.add_argument("--min-timed-seconds", type=float, default=1.0,
                    help="the timed region (all blocks of the headline leg together) spans at least this long: the number of "
                         "blocks is raised beyond --repeats as needed, each block stays EXACTLY --steps steps")
    ap.add_argument("--profile-launches", type=int, default=200,
                    help="launches of each kernel averaged by the HIP-event per-kernel timing (admm_profile)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs4", action="store_true", help="skip the compact configs[4] side object of the default line")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    return ap.parse_args()


def newest_profile(suffix, workload_tag=""):
    """Newest committed summary profiles/r<NN><x>[_<workload_tag>]<suffix> (tags sort by round, then letter)."""
    import glob
    import re
    pat = re.compile(r"^r(\d+)([a-z]*)" + (f"_{workload_tag}" if workload_tag else "") + re.escape(suffix) + "$")
    hits = [(int(m.group(1)), m.group(2), f) for f in glob.glob(os.path.join(ROOT, "profiles", "*" + suffix))
            for m in [pat.match(os.path.basename(f))] if m]
    return max(hits)[2] if hits else None


def pmc_rows(path, base, lead_args):
    """Rows of a profiles/*.csv whose kernel is admm::<base><lead_args..., ...>: matched on the kernel's BASE NAME and its
    LEADING template arguments, so a template parameter appended later does not silently break the match (r02: the names
    gained a trailing XFREE argument and the lookup walked back to an older file)."""
    import csv
    want = f"admm::{base}<" + ", ".join(lead_args)
    out = []
    for row in csv.DictReader(open(path)):
        k = row["kernel"].replace("void ", "")
        if k.startswith(want) and k[len(want):len(want) + 1] in (",", ">"):
            out.append(row)
    return out


def pmc_traffic(base, lead_args, workload_tag=""):
    """HBM bytes per launch of a kernel from the NEWEST committed PMC summary under profiles/ (rocprofv3 --pmc FETCH_SIZE and
    --pmc WRITE_SIZE in separate passes of this same command, FETCH_SIZE doubled per the gfx950 correction of
    MI355X_MICROARCH.md §HBM; profiles/summarize.py).  bench.py cannot collect counters on itself, so it reports the stored
    measurement.  RAISES if the newest summary does not hold the kernel: a stale file must never be cited in its place."""
    path = newest_profile("_hbm_traffic.csv", workload_tag)
    if path is None:
        return None, None
    rows = [r for r in pmc_rows(path, base, lead_args) if float(r["hbm_MB_per_launch"]) > 0]
    if not rows:
        raise RuntimeError(f"bench.py: {os.path.relpath(path, ROOT)} (the newest PMC summary) has no row for admm::{base}<"
                           f"{', '.join(lead_args)}, ...>: re-run tools/gpu_profile.sh and commit the summary")
    row = max(rows, key=lambda r: int(r["launches_fetch_pass"]))          # the form the timed region launches most
    return float(row["hbm_MB_per_launch"]) * 1e6, f"{os.path.relpath(path, ROOT)}: {row['kernel']}"


def pmc_mfma(base, lead_args, workload_tag=""):
    """Measured matrix-pipe counters of a kernel from the newest profiles/*_mfma.csv (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES
    GRBM_GUI_ACTIVE and --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_F32, separate passes; summarize.py)."""
    path = newest_profile("_mfma.csv", workload_tag)
    if path is None:
        return None
    rows = pmc_rows(path, base, lead_args)
    if not rows:
        return None
    row = max(rows, key=lambda r: int(r["launches"]))
    out = {k: (float(v) if k not in ("kernel",) else v) for k, v in row.items()}
    out["source"] = os.path.relpath(path, ROOT)
    return out


def cpu_baseline(N, target_s, workload="cw_rendezvous"):
    """Time the CPU oracle (C/OpenMP restatement) on a bounded sample of the same workload: oracle/cpu_baseline.py as a child
    process (its OpenMP runtime starts with the thread count this box really offers -- affinity and cgroup quota -- and
    bound threads, whatever this process has loaded)."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--workload", workload,
                        "--horizon", str(N), "--seconds", str(target_s)], capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        raise RuntimeError("bench.py: oracle/cpu_baseline.py failed:\n" + r.stderr[-2000:])
    return json.loads(r.stdout.strip().splitlines()[-1])


def mfma_accounting(n_, m_, precision, pitch, stages, fwd_ms, bwd_ms, pmc_tag=None):
    """MFMA accounting of the fused pair xfzem / xbzem (configs[4]): instructions per 16-QP tile and stage from
    the layout (csrc/admm_mfma_layout.hpp), padding counted as waste: "useful" = the multiply-adds of the one-lane kernels'
    operator list, "issued" = 16 x 16 x 4 per MFMA.  Beside the host count, the MEASURED matrix-pipe counters of the same
    kernels from the newest profiles/*_mfma.csv (SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE, SQ_INSTS_VALU_MFMA_MOPS_*)."""
    mode = 1 if precision == "mixed" else 2
    xt = 1 if m_ > 4 else 0
    nm = {"fwd": {"sub": 7 + xt, "elim": 2 * (7 + xt)}, "bwd": {"sub": 7 + xt, "elim": 2 * (4 + xt)}}
    es = {"fwd": {"sub": 4 if mode == 1 else 8, "elim": 8}, "bwd": {"sub": 8, "elim": 4 if mode == 1 else 8}}
    alg = {"fwd": {"sub": 3 * m_ * n_ + n_ * n_, "elim": 3 * m_ * n_ + m_ * m_ + 2 * n_ * n_},
           "bwd": {"sub": 3 * m_ * n_ + n_ * n_, "elim": 3 * m_ * n_ + m_ * m_ + n_ * n_}}
    PEAK = {4: 157.3, 8: 78.6}                    # dense TFLOP/s: v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64 (MI355X_MICROARCH.md)
    CYC = {4: 32, 8: 64}                          # issue cycles per MFMA and SIMD
    tiles, qps = pitch // 16, pitch
    nt = "1" if pitch <= 128 else "2"
    ids = {"xfzem": ("xfzem_kernel", [str(n_), str(m_), nt, "float" if mode == 1 else "double", "double", "true", "false", "true"]),
           "xbzem": ("xbzem_kernel", [str(n_), str(m_), nt, "double", "float" if mode == 1 else "double", "true", "false", "true"])}
    per = {}
    for kname, d, ms in (("xfzem", "fwd", fwd_ms), ("xbzem", "bwd", bwd_ms)):
        useful = sum(2.0 * alg[d][p] for p in ("sub", "elim")) * qps * stages
        issued = sum(nm[d][p] for p in ("sub", "elim")) * 2048.0 * tiles * stages
        t_peak = sum(2.0 * alg[d][p] * qps * stages / (PEAK[es[d][p]] * 1e12) for p in ("sub", "elim"))
        pipe_cyc = sum(nm[d][p] * CYC[es[d][p]] for p in ("sub", "elim")) * tiles * stages
        per[kname] = {"avg_launch_ms": ms, "mfma_per_tile_stage": nm[d], "element_bytes": es[d],
                      "useful_flop": useful, "issued_flop": issued, "useful_over_issued": useful / issued,
                      "useful_TFLOPs": useful / (ms * 1e-3) / 1e12, "issued_TFLOPs": issued / (ms * 1e-3) / 1e12,
                      "peak_TFLOPs_for_this_mix": useful / t_peak / 1e12,
                      "matrix_pipe_busy_at_2p4GHz": pipe_cyc / (1024 * 2.4e9 * ms * 1e-3),
                      "measured_counters": None if pmc_tag is None else pmc_mfma(ids[kname][0], ids[kname][1], pmc_tag)}
    tot_ms = fwd_ms + bwd_ms
    useful = per["xfzem"]["useful_flop"] + per["xbzem"]["useful_flop"]
    t_peak = sum(per[k]["useful_flop"] / (per[k]["peak_TFLOPs_for_this_mix"] * 1e12) for k in per)
    ach = useful / (tot_ms * 1e-3) / 1e12
    meas = [per[k]["measured_counters"] for k in per]
    return {"kernel": f"xfzem_kernel / xbzem_kernel <{n_},{m_}> ({precision}): the alternating pair with the stage "
                      "operators as v_mfma_*_16x16x4 chains over 16-QP panels",
            "bound": "mfma", "achieved": ach, "peak": useful / t_peak / 1e12, "unit": "TFLOP/s",
            "frac": ach / (useful / t_peak / 1e12), "traffic": None,
            "mfma_util_measured": (None if any(x is None for x in meas) else
                                   sum(x["MfmaUtil_pct"] * x["launches"] for x in meas) / sum(x["launches"] for x in meas) / 100.0),
            "note": ("achieved = USEFUL flops (the one-lane kernels' operator list, padding and folded blocks "
                     "counted as waste) / measured kernel time; peak = dense MFMA peak of the element types, weighted "
                     "by each product's useful flops.  The kernels are HBM-bound (roofline_hbm): the matrix pipe is "
                     "busy for the fraction per_kernel.*.matrix_pipe_busy_at_2p4GHz of the launch by the host's count of issued "
                     "MFMAs x their issue cycles at an ASSUMED 2.4 GHz, and for mfma_util_measured = SQ_VALU_MFMA_BUSY_CYCLES / "
                     "(GRBM_GUI_ACTIVE x SIMDs) by the hardware counters of the stored rocprofv3 pass (per_kernel.*.measured_counters)."),
            "per_kernel": per}


def launch_ranks(a):
    """`python bench.py --gpus N` without an external launcher: spawn the N ranks ourselves, as fresh child
    processes created BEFORE anything in this process touches the GPU (never an exec of a process that has),
    one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment exactly as
    torch.distributed.run would set them.  Rank 0's JSON line is this process's stdout."""
    import socket
    import subprocess
    import torch
    backend = os.environ.get("ADMM_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()           # counting devices does not initialise the GPU on this image
    if backend == "nccl" and ndev < a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but only {ndev} GPU(s) visible: refusing to report a "
                         f"{a.gpus}-GPU figure from fewer devices")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    raise SystemExit(rc)


def bench_perinstance(a, pkg, np, world, rank, dev_index, lo_i, hi_i, gbatch, barrier, max_over_ranks):
    """SIDE MEASUREMENT: per-instance dynamics (admm_problem.time_varying = 2, stage_bounds = 2).  A different
    roofline from the metric's: every QP streams its own factor operands, ~200 B per stacked element and iteration."""
    make = pkg.cw_formation_instances if a.workload == "cw_formation_perinstance" else pkg.cw_rendezvous_instances
    full = make(N=a.horizon, batch=hi_i - lo_i, seed0=pkg.SEED0 + lo_i)
    n_, m_, nb = full.n, full.m, full.nb
    with pkg.Solver(full, pkg.Options(rho=0.05, check_interval=1, segments=a.segments, device=dev_index)) as sv:
        geo = sv.geometry()
        sv.run(max(a.warmup, 3), residual_every=1)
        t_end = time.perf_counter() + a.warm_seconds
        while time.perf_counter() < t_end:
            sv.run(10, residual_every=1, sync=True)
        blocks = []
        for _ in range(a.repeats):
            barrier()
            t0 = time.perf_counter()
            sv.run(a.steps, residual_every=1, sync=True)
            barrier()
            blocks.append(max_over_ranks(time.perf_counter() - t0))
        prof = sv.profile(min(a.profile_launches, 50), residuals=True, fused=True)
        # post-timing value check (rank 0): eight iterations from zero on this very handle against the CPU oracle on four QPs spread
        # over the batch -- a batch this size is checked nowhere else (round 3: a set-up race zeroed the weights of large handles
        # and nothing noticed, the timings being what they should)
        check = None
        if rank == 0:
            import dataclasses
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle_c
            idx = np.linspace(0, full.batch - 1, 4).astype(int)
            sub = dataclasses.replace(full, A=full.A[idx], B=full.B[idx], x0=full.x0[idx], lo=full.lo[idx], hi=full.hi[idx],
                                      q=None if full.q is None else full.q[idx])
            ref = oracle_c.solve(sub, rho=0.05, max_iter=8, stop=False, nthreads=4)
            sv.set_state(z=np.zeros((full.batch, full.L)), y=np.zeros((full.batch, full.L)))
            sv.run(8, residual_every=4)
            got = sv.get()
            err = max(float(np.abs(g[idx] - ref[k]).max()) for g, k in zip(got, ("w", "z", "y")))
            check = {"qps": [int(i) for i in idx], "iterations": 8, "max_abs_difference_w_z_y": err}
            if not err <= 1e-9:
                raise SystemExit(f"bench.py: the HIP path disagrees with the oracle on the sampled QPs: {check}")
    dt = float(np.median(blocks))
    elems = full.L * geo["pitch"]
    seg_ops = n_ * m_ * 8.0 / nb if geo["segments"] > 1 else 0.0        # Omega_k (backward) / Psi_k (forward) with segments in time
    ops_b = (n_ * n_ + n_ * m_ + m_ * n_ + m_ * m_) * 8.0 / nb + seg_ops   # A, B, K, Si per stage -> per stacked element
    ops_f = (m_ * n_ + n_ * n_ + n_ * m_) * 8.0 / nb + seg_ops              # K, A, B
    b_xb = 8.0 + 16.0 + ops_b + 8.0 * m_ / nb                            # v, lo, hi read; operands; d written
    b_xfz = 8.0 * m_ / nb + 8.0 + 16.0 + ops_f + 8.0                     # d, v, lo, hi read; operands; v+ written
    ms = prof["xb_ms"] + prof["xfz_ms"]
    ach = (b_xb + b_xfz) * elems / (ms * 1e-3) / 1e9
    if rank == 0:
        print(json.dumps({
            "metric": "ADMM iterations/sec (fp64), per-instance dynamics (SIDE MEASUREMENT, not BASELINE's metric)",
            "value": gbatch * a.steps / dt, "unit": "QP-iterations/s", "batch_iterations_per_s": a.steps / dt,
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "repeats": a.repeats, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"SIDE MEASUREMENT ({a.workload}): batch of {a.batch} N={a.horizon} n={n_} m={m_} QPs per GPU, "
                                   f"every QP with its own dynamics and box (time_varying = 2, stage_bounds = 2), residuals every iteration",
                       "N": a.horizon, "n": n_, "m": m_, "batch_per_gpu": a.batch, "global_batch": gbatch, **geo},
            "roofline": {"kernel": ("pxb_rows_kernel + pxfz_rows_kernel (a QP's rows over the lanes of a wave; operands per QP from HBM)"
                                    if n_ >= 8 else "pxb_kernel + pxfz_kernel (one lane per QP; operands per QP from HBM)"),
                         "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": None, "bytes_per_element": b_xb + b_xfz, "bytes_per_launch_pair": (b_xb + b_xfz) * elems,
                         "avg_launch_ms": {"pxb": prof["xb_ms"], "pxfz": prof["xfz_ms"]},
                         "note": "one lane sweeps one segment of one QP (segments in time with per-QP transfer matrices, DESIGN.md "
                                 "§4.10); xscan = the per-QP segment scan", "xscan_ms": prof.get("xscan_ms")},
            "oracle_sample": check}))


def main():
    a = parse()
    if a.gpus < 1 or a.steps < 1 or a.repeats < 1:
        raise SystemExit("bench.py: --gpus, --steps, --repeats must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        launch_ranks(a)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: one rank per GPU, launch with "
                         f"--nproc-per-node {a.gpus} (or leave WORLD_SIZE unset and bench.py spawns the ranks itself)")
    import numpy as np
    import torch
    import admm_library_amd as pkg

    import __graft_entry__ as ge
    if not os.path.exists(pkg.library_path()):
        ge.build()
    pkg.load_library()

    # The CPU baseline runs FIRST (rank 0, N = 1 only), as a child process started before this process makes any HIP call:
    # ~15 s of host work between the GPU legs would leave the GPU at idle clocks for whatever is timed next (r01: the
    # driver's 3.3 ms timed region ran on a 0.8 %-busy GPU).
    cpu_base = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu_base = cpu_baseline(a.horizon, a.cpu_seconds, a.workload)
    if pkg.device_count() < 1:
        raise SystemExit("bench.py: no HIP device visible; the solver has no CPU fallback")

    # One process per GPU.  ADMM_BENCH_BACKEND=gloo (+ several ranks sharing one GPU) exists only to
    # rehearse the N > 1 code path on a 1-GPU box; the driver's runs use RCCL ("nccl").
    backend = os.environ.get("ADMM_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and ndev < world:
        raise SystemExit(f"bench.py: {world} ranks but only {ndev} GPU(s) visible (backend nccl needs one GPU per rank)")
    dev_index = local_rank % max(ndev, 1) if backend != "nccl" else local_rank
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    red_device = f"cuda:{dev_index}" if backend == "nccl" else "cpu"

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=red_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # global problem = batch * world QPs; this rank's contiguous shard
    gbatch = a.batch * world
    lo_i, hi_i = pkg.shard_bounds(gbatch, world, rank)
    if a.workload in ("cw_perinstance", "cw_formation_perinstance"):
        return bench_perinstance(a, pkg, np, world, rank, dev_index, lo_i, hi_i, gbatch, barrier, max_over_ranks)
    if a.workload == "cw_rendezvous_soc":      # side measurement: thrust-magnitude bound instead of the input box
        full = pkg.cw_rendezvous(N=a.horizon, batch=hi_i - lo_i, seed0=pkg.SEED0 + lo_i, thrust_norm=True)
    else:
        make = pkg.cw_rendezvous if a.workload == "cw_rendezvous" else pkg.cw_formation
        full = make(N=a.horizon, batch=hi_i - lo_i, seed0=pkg.SEED0 + lo_i)
    PM = {"fp64": 0, "fp64_one_lane": 0, "mixed": 1, "fp64_mfma": 2}
    PFLAGS = {"fp64": 0, "fp64_one_lane": 32, "mixed": 0, "fp64_mfma": 0}          # 32 = ADMM_FLAG_NO_MFMA
    opt = pkg.Options(rho=0.05, check_interval=1, segments=a.segments, zrows=a.zrows, device=dev_index,
                      precision_mode=PM[a.precision], flags=PFLAGS[a.precision])
    solver = pkg.Solver(full, opt)
    geo = solver.geometry()
    path = solver.path()            # which kernels the handle runs + the margin of the default path (admm_get_path, ABI v6)

    def warm(sv, seconds, every=1):
        """steady clocks: iterate until `seconds` of wall time have gone by (the state simply keeps converging)"""
        t_end = time.perf_counter() + seconds
        while time.perf_counter() < t_end:
            sv.run(200, residual_every=every, sync=True)

    def timed_blocks(sv, every, min_seconds=0.0):
        """Blocks of EXACTLY --steps steps, each bracketed by barrier + synchronize on both sides and reduced with MAX over
        the ranks: at least --repeats of them, and as many more as it takes for the blocks to add up to `min_seconds` (decided
        from the MAX-reduced times, so every rank runs the same number).  Returns the per-block seconds."""
        out = []
        while len(out) < a.repeats or (sum(out) < min_seconds and len(out) < 100000):
            barrier()
            t0 = time.perf_counter()
            sv.run(a.steps, residual_every=every, sync=True)
            barrier()
            out.append(max_over_ranks(time.perf_counter() - t0))
        return out

    solver.run(a.warmup, residual_every=1)          # the W untimed warmup steps ...
    warm(solver, a.warm_seconds)                    # ... and then by time, whatever W was
    t_region = time.perf_counter()
    blocks = timed_blocks(solver, 1, a.min_timed_seconds)
    t_region = time.perf_counter() - t_region
    dt = float(np.median(blocks))

    ms_per_step = dt / a.steps * 1e3
    value = gbatch * a.steps / dt

    # per-kernel timing on the library's stream (HIP events)
    npf = max(1, min(a.profile_launches, 4096))     # launches of each kernel, regardless of --steps
    try:
        # the timed path: one HIP-event pair around every launch (the fused kernels' intervals agree with rocprofv3's
        # kernel durations within 1 %, profiles/r02a_*; the event-record bubbles fall into the short scan intervals).
        # prof_alt_b2b is the cross-check without events between launches (admm_profile mode 3).
        prof_alt = solver.profile(npf, residuals=True, alternating=True)
        prof_alt_bracketed = solver.profile(npf, residuals=True, alternating=True, back_to_back=True)
        prof_alt_plain = solver.profile(npf, residuals=False, alternating=True)
    except pkg.AdmmError:                          # no alternating kernels for this shape: the plain kernels are timed
        prof_alt = prof_alt_plain = prof_alt_bracketed = None
    prof = solver.profile(npf, residuals=True, fused=True)           # plain fused path (the timed path without alternation)
    prof_plain = solver.profile(npf, residuals=False, fused=True)
    prof_unf = solver.profile(npf, residuals=True, fused=False)      # standalone z/dual kernel
    L = full.L
    elems = L * geo["pitch"]
    n_, m_ = full.n, full.m
    uses_mfma = path["kernel_family"] != "one_lane_fp64"      # reported by the library, not re-derived here
    b_xfz = 8.0 * m_ / (n_ + m_) + 16.0          # d read + v read + v+ written
    xfz_ms = prof["xfz_ms"]
    # the stored PMC runs (profiles/r<NN><x>[_formation_mixed]_hbm_traffic.csv) are of these two workloads only
    pmc_tag = {(4096, 1000, "cw_rendezvous", "fp64"): "", (4096, 1000, "cw_formation", "mixed"): "formation_mixed"}.get(
        (a.batch, a.horizon, a.workload, a.precision))
    pmc_note = " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH_SIZE x2)"
    tf = {True: "true", False: "false"}

    def kernel_roofline(name, desc, bytes_per_elem, ms, pmc_id):
        ach = bytes_per_elem * elems / (ms * 1e-3) / 1e9
        r = {"kernel": f"{name} ({desc})", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": ach / HBM_PEAK_GBS, "traffic": None, "bytes_per_launch": bytes_per_elem * elems,
             "bytes_per_element": bytes_per_elem, "avg_launch_ms": ms}
        if pmc_tag is not None:
            tr, src = pmc_traffic(pmc_id[0], pmc_id[1], pmc_tag)
            r["traffic"] = tr
            if src:
                r["traffic_source"] = src + pmc_note
        return r

    roofline_xfz = kernel_roofline(f"xfz_kernel<{n_},{m_},RESID=true,RELAX=false,VIN=true>",
                                   "plain path: forward rollout fused with z-update + dual ascent + residual partials, "
                                   "state in v-form", b_xfz, xfz_ms, ("xfz_kernel", [str(n_), str(m_), "true", "false", "true"]))
    if prof_alt is not None:
        b_alt = 16.0 + 16.0 * m_ / (n_ + m_) + (8.0 if full.q is not None else 0.0)   # v, v+, d and db rows (+ q)
        kf = "xfzem_kernel" if uses_mfma else "xfze_kernel"
        kb = "xbzem_kernel" if uses_mfma else "xbze_kernel"
        if uses_mfma:       # <NX, NU, NT, TS, TE, RESID, RELAX, ELIM / SUBST, ...>: csrc/admm_mfma.hpp
            nt = "1" if geo["pitch"] <= 128 else "2"
            mixed = a.precision == "mixed"
            id_f = (kf, [str(n_), str(m_), nt, "float" if mixed else "double", "double", "true", "false", "true"])
            id_b = (kb, [str(n_), str(m_), nt, "double", "float" if mixed else "double", "true", "false", "true"])
        else:               # <n, m, RESID, RELAX, HASQ, SOC, ...>: csrc/admm_kernels_alt.hpp
            lead = [str(n_), str(m_), "true", "false", tf[full.q is not None], tf[full.unorm is not None]]
            id_f, id_b = (kf, lead), (kb, lead)
        rf = kernel_roofline(f"{kf}<{n_},{m_},RESID=true,RELAX=false,HASQ=false,SOC=false>",
                             "forward rollout + z-update + dual ascent + residual partials + forward elimination of v+",
                             b_alt, prof_alt["xfze_ms"], id_f)
        rb = kernel_roofline(f"{kb}<{n_},{m_},RESID=true,RELAX=false,HASQ=false,SOC=false>",
                             "backward rollout + z-update + dual ascent + residual partials + backward elimination of v+",
                             b_alt, prof_alt["xbze_ms"], id_b)
        pair_ms = prof_alt["xfze_ms"] + prof_alt["xbze_ms"]
        ach = 2 * b_alt * elems / (pair_ms * 1e-3) / 1e9
        roofline = {"kernel": f"xfze_kernel / xbze_kernel <{n_},{m_},RESID=true,RELAX=false> (the alternating pair: one of them "
                              "per iteration; substitution sweep + z-update + dual ascent + residual partials + next "
                              "elimination sweep, state in v-form)",
                    "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "traffic": (None if rf["traffic"] is None or rb["traffic"] is None else 0.5 * (rf["traffic"] + rb["traffic"])),
                    "bytes_per_launch": b_alt * elems, "bytes_per_element": b_alt, "avg_launch_ms": 0.5 * pair_ms,
                    "per_kernel": {"xfze": rf, "xbze": rb}}
        if "traffic_source" in rf:
            roofline["traffic_source"] = rf["traffic_source"] + "; mean of the two kernels"
    else:
        roofline = roofline_xfz
    roofline_mfma = None
    if uses_mfma and prof_alt is not None:
        roofline_mfma = mfma_accounting(n_, m_, "mixed" if path["kernel_family"] == "mfma_mixed" else "fp64_mfma", geo["pitch"], full.N,
                                        prof_alt["xfze_ms"], prof_alt["xbze_ms"], pmc_tag)
    zs_ms = prof_unf["zdual_ms"]
    zs = BYTES_PER_ELEM_ZDUAL * elems / (zs_ms * 1e-3) / 1e9
    standalone = {"kernel": "zdual_kernel<RESID=true> (standalone fused z-update + dual + residual, ADMM_FLAG_UNFUSED path)",
                  "bound": "hbm", "achieved": zs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": zs / HBM_PEAK_GBS,
                  "bytes_per_launch": BYTES_PER_ELEM_ZDUAL * elems, "bytes_per_element": BYTES_PER_ELEM_ZDUAL,
                  "avg_launch_ms": zs_ms}
    b_xb = 8.0 + 8.0 * m_ / (n_ + m_)             # v read + d written
    xb_gbs = b_xb * elems / (prof["xb_ms"] * 1e-3) / 1e9
    # algorithmic HBM bytes per element per iteration of the timed path
    b_iter = (16.0 + 16.0 * m_ / (n_ + m_) + (8.0 if full.q is not None else 0.0)) if prof_alt is not None else b_xb + b_xfz

    # the same steps on the plain fused path (xb + xscan + xfz + finalise), for the A/B in DESIGN.md §4.8
    plain_path = None
    if prof_alt is not None:
        from admm_library_amd import _abi
        with pkg.Solver(full, pkg.Options(rho=0.05, check_interval=1, segments=a.segments, zrows=a.zrows, device=dev_index,
                                          flags=_abi.FLAG_NO_ALTERNATE)) as sp:
            sp.run(a.warmup, residual_every=1)
            warm(sp, 0.5 * a.warm_seconds)
            dtp = float(np.median(timed_blocks(sp, 1)))
        plain_path = {"batch_iterations_per_s": a.steps / dtp, "ms_per_step": dtp / a.steps * 1e3,
                      "iteration_bytes_per_element": b_xb + b_xfz, "roofline_xfz": roofline_xfz}

    # iterations that evaluate no residuals: where every state row is unbounded at every stage (all the CW workloads) the
    # kernels do not read v of the state rows (XFREE forms): v of the m input rows + v+ of all rows + d + db
    xfree = (prof_alt is not None and not full.per_instance
             and bool(np.all(np.asarray(full.lo)[..., m_:] == -np.inf) and np.all(np.asarray(full.hi)[..., m_:] == np.inf)))
    b_iter_plain = (8.0 + 24.0 * m_ / (n_ + m_) + (8.0 if full.q is not None else 0.0)) if xfree else b_iter
    # ... and do not write it either while the next iteration is of the same kind (XFREE = 2 forms: 8 of 10 at check_interval 10)
    b_iter_nostore = (32.0 * m_ / (n_ + m_) + (8.0 if full.q is not None else 0.0)) if xfree else b_iter
    # mixed mode a solver would normally run: residuals every 10th iteration
    warm(solver, 0.25 * a.warm_seconds, every=10)
    dt10 = float(np.median(timed_blocks(solver, 10, 0.5 * a.min_timed_seconds)))

    # configs[4]: the three precision modes side by side on this workload (rate with residuals every iteration)
    precision_modes = None
    if a.workload == "cw_formation" or a.precision != "fp64":
        precision_modes = {}
        for name, pm in PM.items():
            try:
                with pkg.Solver(full, pkg.Options(rho=0.05, check_interval=1, segments=a.segments, zrows=a.zrows,
                                                  device=dev_index, precision_mode=pm, flags=PFLAGS[name])) as sp:
                    warm(sp, 0.4 * a.warm_seconds)
                    dtm = float(np.median(timed_blocks(sp, 1)))
                precision_modes[name] = {"batch_iterations_per_s": a.steps / dtm, "ms_per_step": dtm / a.steps * 1e3}
            except pkg.AdmmError as e:
                precision_modes[name] = {"error": str(e)}

    # configs[4] beside the headline, so that its numbers are driver-observed (VERDICT r02 next #2e): the n = 12, m = 6 formation
    # workload at the same batch and horizon, fp64-MFMA (exact) and mixed, ~1.5 s each: rate with residuals every iteration, HBM
    # roofline of the fused pair (21.33 B/element), host-counted and measured matrix-pipe utilisation.
    configs4 = None
    if a.workload == "cw_rendezvous" and rank == 0 and world == 1 and not a.no_configs4:
        configs4 = {"workload": f"configs[4]: batch of {a.batch} N={a.horizon} n=12 m=6 Clohessy-Wiltshire formation QPs, residuals every "
                                "iteration: fp64 (the library's choice: one-lane kernels with DPP-distributed operators), and the x-update "
                                "as MFMA batched GEMM in fp64 (fp64_mfma) and in mixed fp32/fp64 (mixed)"}
        form = pkg.cw_formation(N=a.horizon, batch=a.batch)
        for name in ("fp64", "fp64_mfma", "mixed"):          # fp64 = the library's choice at this shape and batch (the one-lane kernels)
            with pkg.Solver(form, pkg.Options(rho=0.05, check_interval=1, device=dev_index, precision_mode=PM[name])) as sp:
                g4, p4 = sp.geometry(), sp.path()
                warm(sp, 0.3)
                b4 = timed_blocks(sp, 1, 0.3)
                b4_10 = timed_blocks(sp, 10, 0.2)
                pr = sp.profile(min(npf, 50), residuals=True, alternating=True)
            d4 = float(np.median(b4))
            e4 = form.L * g4["pitch"]
            pair_ms = pr["xfze_ms"] + pr["xbze_ms"]
            hbm = 2 * (16.0 + 16.0 * 6 / 18) * e4 / (pair_ms * 1e-3) / 1e9
            configs4[name] = {"batch_iterations_per_s": a.steps / d4, "ms_per_step": d4 / a.steps * 1e3, "timed_blocks": len(b4),
                              "batch_iterations_per_s_check_interval_10": a.steps / float(np.median(b4_10)),
                              "kernel_family": p4["kernel_family"], "alternating": p4["alternating"], "alt_check": p4["alt_check"],
                              "segments": g4["segments"],
                              "forward_kernel_ms": pr["xfze_ms"], "backward_kernel_ms": pr["xbze_ms"], "xscan_ms": pr["xscan_ms"],
                              "hbm": {"achieved": hbm, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm / HBM_PEAK_GBS,
                                      "bytes_per_element": 16.0 + 16.0 * 6 / 18}}
            if p4["kernel_family"] != "one_lane_fp64":
                acc = mfma_accounting(12, 6, name, g4["pitch"], form.N, pr["xfze_ms"], pr["xbze_ms"],
                                      "formation_mixed" if (name, a.batch, a.horizon) == ("mixed", 4096, 1000) else None)
                configs4[name]["mfma"] = {
                    "useful_TFLOPs": acc["achieved"], "peak_TFLOPs_for_this_mix": acc["peak"], "frac": acc["frac"],
                    "useful_over_issued": {k: v["useful_over_issued"] for k, v in acc["per_kernel"].items()},
                    "matrix_pipe_busy_at_2p4GHz": {k: v["matrix_pipe_busy_at_2p4GHz"] for k, v in acc["per_kernel"].items()},
                    "mfma_util_measured": acc["mfma_util_measured"],
                    "measured_counters_source": next((v["measured_counters"]["source"] for v in acc["per_kernel"].values()
                                                      if v["measured_counters"]), None)}
        del form

    # iterations-to-epsilon (second half of BASELINE.json's metric; "vs MATLAB" -> vs the CPU oracle):
    # a full admm_solve of this rank's shard to eps_abs = eps_rel = 1e-6, stop test every 10 iterations,
    # once with fixed rho and once with the batch-level adaptive rule.
    solver.close()
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_c
        from cpu_baseline import usable_cpus
        oracle_threads = usable_cpus()[0]      # never more OpenMP threads than this box's CPU share
    else:
        oracle_c = None

    host_boundary = {}

    def to_eps(whole_job=False, **adapt):
        base = dict(rho=0.05, eps_abs=1e-6, eps_rel=1e-6, max_iter=4000, check_interval=10)
        t0 = time.perf_counter()
        with pkg.Solver(full, pkg.Options(segments=a.segments, zrows=a.zrows, device=dev_index, **base, **adapt)) as sv:
            sv.sync()
            t1 = time.perf_counter()
            info = sv.solve()
            t2 = time.perf_counter()
            if whole_job:
                # The C ABI hands over HOST buffers (admm_setup in, admm_get out): one whole job through it, transfers and the host
                # factorisation included -- reported beside `value`, never as `value` (which is timed with everything resident in HBM).
                res = sv.get()
                t3 = time.perf_counter()
                qp_it = float(np.sum(info.iters))          # iterations each QP needed (the batch runs until its last QP converges)
                host_boundary.update({
                    "what": "admm_setup (host arrays in: host factorisation + H2D) + admm_solve to eps = 1e-6 + admm_get (w, z, y to "
                            "host: %.0f MB D2H), wall clock from the Python wrapper" % (3 * res[0].nbytes / 1e6),
                    "setup_ms": (t1 - t0) * 1e3, "solve_ms": (t2 - t1) * 1e3, "get_ms": (t3 - t2) * 1e3,
                    "batch_iterations_run": int(info.iters_run),
                    "QP_iterations_per_s_resident": full.batch * int(info.iters_run) / (t2 - t1),
                    "QP_iterations_per_s_host_buffers_included": full.batch * int(info.iters_run) / (t3 - t0),
                    "per_qp_iterations_total": qp_it, "options": {**base, **adapt}})
                del res
            if adapt.get("precision_mode", 0) != 0:
                z_mode = sv.get(False, True, False)[1]
        out = {**base, **adapt, "batch_iterations_run": int(info.iters_run), "converged": int(info.n_converged),
               "mixed_iters": int(info.mixed_iters), "max_r": float(info.max_r), "max_s": float(info.max_s),
               "batch": int(full.batch), "per_qp_median": float(np.median(info.iters)),
               "per_qp_max": int(info.iters.max()), "solve_ms": float(info.solve_ms),
               "rho_final": float(info.rho), "rho_updates": int(info.rho_updates)}
        if adapt.get("precision_mode", 0) != 0:
            # final point of the reduced-precision / MFMA solve against the fp64 one-lane solve of the same options
            with pkg.Solver(full, pkg.Options(segments=a.segments, zrows=a.zrows, device=dev_index, **base,
                                              **{**adapt, "precision_mode": 0})) as sv:
                i64 = sv.solve()
                z64 = sv.get(False, True, False)[1]
            out["vs_fp64_path"] = {"fp64_batch_iterations_run": int(i64.iters_run), "fp64_converged": int(i64.n_converged),
                                   "max_abs_z_difference": float(np.abs(z_mode - z64).max()),
                                   "fp64_max_r": float(i64.max_r), "fp64_max_s": float(i64.max_s)}
            del z64
        if oracle_c is not None and adapt.get("precision_mode", 0) == 0 and "flags" not in adapt:
            ns = min(64, full.batch)     # the rule is batch-level, so the sample is solved as its own batch on both sides
            sub = full.slice(0, ns)
            ref = oracle_c.solve(sub, nthreads=min(oracle_threads, ns), **base, **adapt)
            with pkg.Solver(sub, pkg.Options(device=dev_index, **base, **adapt)) as sv:
                gi = sv.solve()
            out["oracle_sample"] = {"qps": ns, "per_qp_iters_equal": int((ref["iters"] == gi.iters).sum()),
                                    "oracle_iterations_run": int(ref["iters_run"]), "gpu_iterations_run": int(gi.iters_run),
                                    "oracle_rho_final": float(ref["rho"]), "gpu_rho_final": float(gi.rho)}
        return out

    iters_to_eps = to_eps()
    iters_to_eps_adaptive = to_eps(adapt_interval=50, adapt_mu=10.0, adapt_tau=2.0)
    iters_to_eps_relaxed = to_eps(whole_job=True, adapt_interval=50, adapt_mu=10.0, adapt_tau=2.0, alpha=1.6)   # + over-relaxation
    iters_to_eps_modes = None
    if precision_modes is not None:         # the reduced-precision mode is judged on iterations-to-eps vs the fp64 path
        iters_to_eps_modes = {name: to_eps(adapt_interval=50, adapt_mu=10.0, adapt_tau=2.0, alpha=1.6, precision_mode=pm)
                              for name, pm in PM.items() if pm != 0 and "error" not in precision_modes[name]}
        iters_to_eps_modes["fp64_one_lane"] = to_eps(adapt_interval=50, adapt_mu=10.0, adapt_tau=2.0, alpha=1.6, flags=32)
    solver = None

    if rank == 0:
        out = {
            "metric": "ADMM iterations/sec (fp64) at N=1000 n=6 batch=4096",
            "value": value, "unit": "QP-iterations/s",
            "batch_iterations_per_s": a.steps / dt,
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
            "repeats": len(blocks), "timed_region_s": t_region, "timed_steps_total": len(blocks) * a.steps,
            "ms_per_step_blocks": {"min": min(blocks) / a.steps * 1e3, "median": dt / a.steps * 1e3, "max": max(blocks) / a.steps * 1e3,
                                   "first": [b / a.steps * 1e3 for b in blocks[:5]]},
            "timing": (f"median of {len(blocks)} blocks of exactly {a.steps} steps (>= {a.repeats}, as many as it takes to time "
                       f">= {a.min_timed_seconds} s), each bracketed by barrier + synchronize and reduced with MAX over ranks; "
                       f"warm-up = {a.warmup} steps + {a.warm_seconds} s"),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64" if a.precision != "mixed" else "f64 state + f32/f64 x-update (mixed)", "data": "synthetic",
            "config": {"workload": (f"configs[2]: batch of {a.batch} independent N={a.horizon} n=6 m=3 "
                                    f"Clohessy-Wiltshire QPs per GPU, residuals every iteration")
                       if a.workload == "cw_rendezvous" else
                       (f"configs[4]: x-update as MFMA batched GEMM ({a.precision}), batch of {a.batch} N={a.horizon} n={n_} m={m_} "
                        f"Clohessy-Wiltshire formation QPs per GPU, residuals every iteration" if a.workload == "cw_formation" else
                        f"SIDE MEASUREMENT ({a.workload}, {a.precision}): batch of {a.batch} N={a.horizon} n={n_} m={m_} "
                        f"Clohessy-Wiltshire QPs per GPU, residuals every iteration"),
                       "N": a.horizon, "n": n_, "m": m_, "batch_per_gpu": a.batch, "global_batch": gbatch,
                       "rho": 0.05, "sharding": f"batch/{world}, no collective in the iteration",
                       **geo, "path": path},
            "roofline": roofline if roofline_mfma is None else roofline_mfma,
            "roofline_hbm": None if roofline_mfma is None else roofline,
            "precision": a.precision,
            "configs4": configs4,
            "precision_modes": precision_modes,
            "iters_to_eps_precision_modes": iters_to_eps_modes,
            "roofline_zdual_standalone": standalone,
            "plain_path": plain_path,
            "kernels_ms": {"alternating_resid": None if prof_alt is None else {k: round(v, 5) for k, v in prof_alt.items()},
                           "alternating_resid_back_to_back": None if prof_alt_bracketed is None else
                           {k: round(v, 5) for k, v in prof_alt_bracketed.items()},
                           "measurement": (f"HIP events on the library's stream, {npf} launches of each kernel; alternating_*: one "
                                           "event pair per launch (the intervals of the two fused kernels match rocprofv3's kernel "
                                           "durations; the event-record bubbles fall into the scan intervals); *_back_to_back: "
                                           "no events between launches -- consecutive (xfze, xbze) pairs, and each scan form in a "
                                           "row (admm_profile mode 3): xfze_ms = xbze_ms = the pair's mean"),
                           "alternating_plain": None if prof_alt_plain is None else {k: round(v, 5) for k, v in prof_alt_plain.items()},
                           "fused_resid": {k: round(v, 5) for k, v in prof.items()},
                           "fused_plain": {k: round(v, 5) for k, v in prof_plain.items()},
                           "unfused_resid": {k: round(v, 5) for k, v in prof_unf.items()},
                           "xb_GBs": xb_gbs, "xb_bytes_per_element": b_xb,
                           "iteration_bytes_per_element": b_iter,
                           "iteration_GBs": b_iter * elems / (ms_per_step * 1e-3) / 1e9},
            "iters_to_eps": iters_to_eps,
            "iters_to_eps_adaptive_rho": iters_to_eps_adaptive,
            "iters_to_eps_adaptive_rho_alpha_1p6": iters_to_eps_relaxed,
            "host_boundary": host_boundary,
            "check_interval_10": {"batch_iterations_per_s": a.steps / dt10,
                                  "QP_iterations_per_s": gbatch * a.steps / dt10,
                                  "state_rows_unbounded_v_not_read": xfree,
                                  "bytes_per_element_without_residuals": b_iter_plain,
                                  "bytes_per_element_neither_read_nor_written": b_iter_nostore,
                                  "iteration_GBs": (0.8 * b_iter_nostore + 0.1 * b_iter_plain + 0.1 * b_iter) * elems / (dt10 / a.steps) / 1e9},
        }
        if cpu_base is not None:
            out["cpu_baseline"] = cpu_base
        print(json.dumps(out))
    if solver is not None:
        solver.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

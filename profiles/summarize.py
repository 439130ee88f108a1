#!/usr/bin/env python3
"""Condense rocprofv3 output directories (gpurun_out/...) into the small
summaries kept under profiles/.

    python profiles/summarize.py <tag> <stats_dir> [<fetch_dir> <write_dir> [<mfma_busy_dir> <mfma_ops_dir>]]

Writes profiles/<tag>_kernel_stats.csv (copy of rocprofv3 --kernel-trace --stats)
and profiles/<tag>_hbm_traffic.csv (per-kernel FETCH_SIZE / WRITE_SIZE averages
from the two separate --pmc passes, with the gfx950 correction of
MI355X_MICROARCH.md §HBM: FETCH_SIZE counts half the bytes of wide coalesced
reads, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE * 1024 is exact), and -- with the two MFMA passes
(--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE MfmaUtil  and  --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_F32)
-- profiles/<tag>_mfma.csv: per kernel the matrix pipe's busy cycles, the derived MfmaUtil (busy cycles summed over the
SIMDs / (GRBM_GUI_ACTIVE x number of SIMDs)) and the MFMA flops the hardware counted (MOPS x 512)."""
import collections
import csv
import glob
import os
import shutil
import sys

HERE = os.environ.get("PROFILES_OUT") or os.path.dirname(os.path.abspath(__file__))   # PROFILES_OUT: summarise on the GPU box into gpurun_out/


def pmc(dirname, counter):
    f = max(glob.glob(os.path.join(dirname, "*", "*_counter_collection.csv")), key=os.path.getmtime)   # newest run
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v)) for k, v in agg.items()}


def main():
    tag, stats = sys.argv[1], sys.argv[2]
    src = max(glob.glob(os.path.join(stats, "*", "*_kernel_stats.csv")), key=os.path.getmtime)   # newest run
    shutil.copy(src, os.path.join(HERE, f"{tag}_kernel_stats.csv"))
    if len(sys.argv) >= 5:
        fe, wr = pmc(sys.argv[3], "FETCH_SIZE"), pmc(sys.argv[4], "WRITE_SIZE")
        with open(os.path.join(HERE, f"{tag}_hbm_traffic.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "launches_fetch_pass", "FETCH_SIZE_avg_KB", "read_MB_corrected(2x)",
                        "launches_write_pass", "WRITE_SIZE_avg_KB", "write_MB", "hbm_MB_per_launch"])
            for k in sorted(set(fe) | set(wr)):
                nf, f_kb = fe.get(k, (0, 0.0))
                nw, w_kb = wr.get(k, (0, 0.0))
                rd, wrb = 2 * f_kb * 1024 / 1e6, w_kb * 1024 / 1e6
                w.writerow([k, nf, f"{f_kb:.1f}", f"{rd:.2f}", nw, f"{w_kb:.1f}", f"{wrb:.2f}", f"{rd + wrb:.2f}"])
    if len(sys.argv) >= 7:
        mfma(tag, sys.argv[5], sys.argv[6])


def mfma(tag, busy_dir, ops_dir):
    busy, gui, util = pmc(busy_dir, "SQ_VALU_MFMA_BUSY_CYCLES"), pmc(busy_dir, "GRBM_GUI_ACTIVE"), pmc(busy_dir, "MfmaUtil")
    f64, f32 = pmc(ops_dir, "SQ_INSTS_VALU_MFMA_MOPS_F64"), pmc(ops_dir, "SQ_INSTS_VALU_MFMA_MOPS_F32")
    with open(os.path.join(HERE, f"{tag}_mfma.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches", "SQ_VALU_MFMA_BUSY_CYCLES_avg", "GRBM_GUI_ACTIVE_avg", "MfmaUtil_pct",
                    "MFMA_MOPS_F64_avg", "MFMA_MOPS_F32_avg", "mfma_GFLOP_per_launch"])
        for k in sorted(set(busy) | set(f64)):
            nb, b = busy.get(k, (0, 0.0))
            _, g = gui.get(k, (0, 0.0))
            _, u = util.get(k, (0, float("nan")))
            _, o64 = f64.get(k, (0, 0.0))
            _, o32 = f32.get(k, (0, 0.0))
            if b == 0 and o64 == 0 and o32 == 0:
                continue                                     # kernels without matrix instructions
            w.writerow([k, nb, f"{b:.0f}", f"{g:.0f}", f"{u:.3f}", f"{o64:.0f}", f"{o32:.0f}", f"{(o64 + o32) * 512 / 1e9:.3f}"])


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Condense rocprofv3 output directories (gpurun_out/...) into the small
summaries kept under profiles/.

    python profiles/summarize.py <tag> <stats_dir> [<fetch_dir> <write_dir>]

Writes profiles/<tag>_kernel_stats.csv (copy of rocprofv3 --kernel-trace --stats)
and profiles/<tag>_hbm_traffic.csv (per-kernel FETCH_SIZE / WRITE_SIZE averages
from the two separate --pmc passes, with the gfx950 correction of
MI355X_MICROARCH.md §HBM: FETCH_SIZE counts half the bytes of wide coalesced
reads, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE * 1024 is exact)."""
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


if __name__ == "__main__":
    main()

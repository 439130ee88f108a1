#!/usr/bin/env python3
"""Which compiled kernel forms does the GPU test suite launch?  (hygiene: audit the instantiations.)

  on the GPU box:   cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 -m pytest <repo>/tests -m gpu -q
                    -> the per-process *_kernel_stats.csv files, merged into profiles/<tag>_gpu_tests_kernel_stats.csv (Name, Calls)
  here (no GPU):    python tools/form_coverage.py profiles/r03c_gpu_tests_kernel_stats.csv
compares the launched names with every kernel the build instantiated (build/obj/*.resource_usage.txt, kept by build())."""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def norm(name):
    return name.split("(")[0].replace("void admm::", "").replace("admm::", "")


launched = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    launched[norm(r["Name"])] += int(r["Calls"])
inst = set()
for f in sorted(glob.glob(os.path.join(ROOT, "build", "obj", "*.resource_usage.txt"))):
    inst.update(norm(r["name"]) for r in ge.parse_resource_usage(open(f).read()))
never = sorted(k for k in inst if k not in launched)
print(f"{len(inst)} kernels instantiated, {len(inst) - len(never)} launched by the suite, {len(never)} never")
tot, miss = collections.Counter(k.split("<")[0] for k in inst), collections.Counter(k.split("<")[0] for k in never)
for fam, n in tot.most_common():
    print(f"  {fam}: {n - miss.get(fam, 0)} of {n}")
for shape in ("<6, 3,", "<12, 6,"):
    sub = [k for k in inst if shape in k]
    print(f"shape {shape.strip('<,')}: {sum(k in launched for k in sub)} of {len(sub)} forms launched")
if "-v" in sys.argv:
    print("\n".join(never))

#!/usr/bin/env python3
"""Per-kernel register / scratch report of the (n, m)-templated kernels, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks, and the build-time spill gate.

    python tools/resource_usage.py [--group g1] [--filter '<6, 3'] [--check] [-D...]

--check exits non-zero if any kernel instantiated for the headline shapes (n, m) = (6, 3) and (12, 6)
spills VGPRs or uses scratch (the SOC forms of xfze<6,3,...> used to).
The same gate runs inside every build: __graft_entry__.build_library() compiles the admm_dims_g*.hip units
with the remarks on and raises if a headline-shape kernel uses scratch (build/obj/*.resource_usage.txt keeps the
report; tests/test_host.py::test_no_spills_at_headline_shapes reads it).
"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

HEADLINE = ("<6, 3,", "<12, 6,")


def report(group: str, extra=()):
    import __graft_entry__ as ge
    src = os.path.join(ge.CSRC, f"admm_dims_{group}.hip")
    cmd = ([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + ge.HIPCC_FLAGS + list(extra) +
           ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.devnull])
    err = subprocess.run(cmd, cwd=ge.CSRC, capture_output=True, text=True, check=True).stderr
    return ge.parse_resource_usage(err)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--group", action="append", help="g0..g3 (default: the groups holding (6,3) and (12,6))")
    ap.add_argument("--filter", default="")
    ap.add_argument("--check", action="store_true")
    a, extra = ap.parse_known_args()
    groups = a.group or ["g1", "g3"]
    bad = 0
    for grp in groups:
        for r in report(grp, extra):
            if a.filter and a.filter not in r["name"]:
                continue
            headline = any(h in r["name"] for h in HEADLINE)
            spilled = r["scratch"] > 0 or r["vgpr_spill"] > 0
            if a.check and not (headline and spilled):
                continue
            if headline and spilled:
                bad += 1
            print(f'{r["name"]:78s} vgpr {r["vgpr"]:3d} agpr {r["agpr"]:3d} sgpr {r["sgpr"]:3d} scratch {r["scratch"]:4d} '
                  f'spill {r["vgpr_spill"]:3d} occ {r["occupancy"]} lds {r["lds"]}')
    if a.check:
        if bad:
            print(f"{bad} headline-shape kernel(s) spill", file=sys.stderr)
            sys.exit(1)
        print("no spills at the headline shapes (6,3), (12,6)")


if __name__ == "__main__":
    main()

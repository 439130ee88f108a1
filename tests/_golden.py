"""Load a golden fixture (tests/golden/*.npz) as (Problem, dict)."""
import os

import numpy as np

import admm_library_amd as pkg

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ["golden_config1", "golden_cw_small", "golden_ltv_relaxed", "golden_cw_soc_adaptive"]


def load(name):
    d = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False))
    p = pkg.Problem(N=int(d["N"]), A=d["A"], B=d["B"], Q=d["Q"], R=d["R"], QN=d["QN"], x0=d["x0"],
                    lo=d["lo"], hi=d["hi"], q=d.get("q"), unorm=d.get("unorm"), name=name)
    return p, d

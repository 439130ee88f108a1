"""admm-library_amd: MI355X-native batched ADMM for box-constrained
optimal-control QPs (astrodynamics trajectory optimisation).

Host side of the path (Python stands in for the MATLAB host the north star
names: no MATLAB/Octave/mex exists in the build image).  The
compute path is libadmm_hip.so (hand-written HIP for gfx950) behind the C ABI
of include/admm_hip.h; this package holds the problem setup, the ctypes binding
and the multi-GPU sharding helpers.  There is no CPU fallback: without the
built library or without a GPU the solver raises.
"""
from .problems import (Problem, double_integrator, cw_rendezvous, cw_formation, cw_matrices,
                       random_ltv, random_instances, cw_rendezvous_instances, cw_formation_instances, mean_motion, SEED0)
from .solver import (AdmmError, Options, Solver, admm_setup, admm_solve,
                     library_path, load_library, device_count, last_warning)
from .sharding import (shard_bounds, shard_problem, gather_batch, global_residual_max, solve_sharded, TimeShardedSolver,
                       make_exchange, device_tensor)

__all__ = ["TimeShardedSolver", "make_exchange", "device_tensor", "last_warning", "Problem", "double_integrator", "cw_rendezvous", "cw_formation", "cw_matrices", "random_ltv", "random_instances", "cw_rendezvous_instances", "cw_formation_instances",
           "mean_motion", "SEED0", "AdmmError", "Options", "Solver", "admm_setup",
           "admm_solve", "library_path", "load_library", "device_count",
           "shard_bounds", "shard_problem", "gather_batch", "global_residual_max", "solve_sharded"]

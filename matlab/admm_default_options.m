function opts = admm_default_options()
%ADMM_DEFAULT_OPTIONS  Options struct of the MI355X ADMM solver (own specification).
%   Fields mirror admm_options in include/admm_hip.h.  NEVER RUN: no MATLAB exists
%   in the build pipeline (INTEGRATION.md).
%   flags (bit or): 2 = unfused kernels, 4 = sequential scan, 8 = no alternating-direction
%   iteration (plain xb + xfz kernels), 16 = hipGraph replay, 32 = never the MFMA kernel form.  0 = the defaults (fastest).
%   precision_mode: 0 = fp64 (default), 1 = mixed fp32/fp64 x-update with fp64 refinement, 2 = fp64 MFMA form always.
opts = struct('rho', 0.1, 'alpha', 1.0, 'eps_abs', 1e-6, 'eps_rel', 1e-6, ...
              'max_iter', 4000, 'check_interval', 10, 'segments', 0, 'device', -1, 'flags', 0, ...
              'adapt_interval', 0, 'adapt_max', 16, 'adapt_mu', 10, 'adapt_tau', 2, 'precision_mode', 0);
end

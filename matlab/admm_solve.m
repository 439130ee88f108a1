function [w, z, y, info] = admm_solve(problem_or_handle, opts, z0, y0)
%ADMM_SOLVE  Solve a batch of box-constrained optimal-control QPs with ADMM on the GPU.
%   [w, z, y, info] = admm_solve(problem, opts)         one-shot: setup + solve + free
%   [w, z, y, info] = admm_solve(h, [], z0, y0)         on an existing handle, warm start optional
%   w, z, y are L x batch; info has iters_run, n_converged, max_r, max_s, iters, status, r, s.
if nargin < 2, opts = []; end
if nargin < 3, z0 = []; end
if nargin < 4, y0 = []; end
own = isstruct(problem_or_handle);
if own, h = admm_setup(problem_or_handle, opts); else, h = problem_or_handle; end
try
    info = admm_mex('solve', h, z0, y0);
    [w, z, y] = admm_mex('get', h);
catch err
    if own, admm_mex('free', h); end
    rethrow(err);
end
if own, admm_mex('free', h); end
end

function admm_update(h, problem)
%ADMM_UPDATE  New shared problem data on an existing handle (same N, n, m, batch): dynamics,
%   weights, box, x0 and q are replaced and the KKT system is refactored in place
%   (admm_update_problem of include/admm_hip.h) -- e.g. between the outer iterations of a
%   successive-convexification loop.  NEVER RUN: no MATLAB exists in the build pipeline.
admm_mex('update', h, problem);
end

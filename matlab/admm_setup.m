function h = admm_setup(problem, opts)
%ADMM_SETUP  Factor the KKT system on the host and upload a batch of QPs to the GPU.
%   problem: struct with fields N, A (n x n [x N]), B (n x m [x N]), Q, R, QN,
%            x0 (n x batch), lo, hi ((m+n) x 1 or x N; +-inf allowed), q (L x batch, optional)
%            stacked variable w = (u_0, x_1, u_1, ..., u_{N-1}, x_N).
%   Returns an opaque uint64 handle; release it with admm_free(h).
%   Thin wrapper over admm_mex('setup', ...) -> admm_setup() of libadmm_hip.so.
if nargin < 2 || isempty(opts), opts = admm_default_options(); end
h = admm_mex('setup', problem, opts);
end

function admm_free(h)
%ADMM_FREE  Release a handle returned by admm_setup.
admm_mex('free', h);
end

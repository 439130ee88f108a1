% Clohessy-Wiltshire rendezvous batch, the BASELINE.json configs[2] shape (own spec, DESIGN.md §3).
N = 1000; batch = 4096; dt = 2*pi/N; s = sin(dt); c = cos(dt);
A = [4-3*c 0 0 s 2*(1-c) 0; 6*(s-dt) 1 0 -2*(1-c) 4*s-3*dt 0; 0 0 c 0 0 s; ...
     3*s 0 0 c 2*s 0; -6*(1-c) 0 0 -2*s 4*c-3 0; 0 0 -s 0 0 c];
B = [1-c 2*(dt-s) 0; -2*(dt-s) 4*(1-c)-1.5*dt^2 0; 0 0 1-c; s 2*(1-c) 0; -2*(1-c) 4*s-3*dt 0; 0 0 s];
p = struct('N', N, 'A', A, 'B', B, 'Q', dt*diag([1 1 1 .1 .1 .1]), 'R', dt*eye(3), ...
           'QN', diag([50 50 50 20 20 20]), 'x0', (2*rand(6, batch)-1) .* [0.3; 2; 1; .1; .1; .1], ...
           'lo', [-0.2*ones(3,1); -inf(6,1)], 'hi', [0.2*ones(3,1); inf(6,1)]);
o = admm_default_options(); o.rho = 0.05;
[w, z, y, info] = admm_solve(p, o);
fprintf('%d iterations, %d / %d converged\n', info.iters_run, info.n_converged, batch);

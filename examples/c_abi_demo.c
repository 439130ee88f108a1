/* c_abi_demo.c -- libadmm_hip.so from plain C: the C ABI of include/admm_hip.h with no Python,
 * no C++ and no torch anywhere.  Solves a small batch of double-integrator QPs (BASELINE.json
 * configs[0] shape) and checks the result is feasible.
 *
 *   gcc -std=c11 -Iinclude examples/c_abi_demo.c -Ladmm-library_amd -ladmm_hip -lm -o c_abi_demo
 *   LD_LIBRARY_PATH=admm-library_amd ./c_abi_demo
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "admm_hip.h"

int main(void) {
  enum { N = 50, n = 2, m = 1, batch = 5, nb = n + m, L = N * nb };
  const double dt = 0.2;
  /* column-major */
  const double A[4] = {1.0, 0.0, dt, 1.0};
  const double B[2] = {0.5 * dt * dt, dt};
  const double Q[4] = {1.0, 0.0, 0.0, 0.1}, R[1] = {0.1}, QN[4] = {10.0, 0.0, 0.0, 1.0};
  const double lo[3] = {-1.0, -INFINITY, -INFINITY}, hi[3] = {1.0, INFINITY, INFINITY};
  double x0[n * batch];
  for (int b = 0; b < batch; ++b) { x0[b * n] = 4.0 - 1.5 * b; x0[b * n + 1] = 0.2 * b; }

  admm_problem p = {.N = N, .n = n, .m = m, .batch = batch, .time_varying = 0, .stage_bounds = 0,
                    .A = A, .B = B, .Q = Q, .R = R, .QN = QN, .x0 = x0, .lo = lo, .hi = hi, .q = NULL};
  admm_options o;
  admm_default_options(&o);
  o.rho = 1.0;
  o.eps_abs = o.eps_rel = 1e-8;

  admm_handle* h = NULL;
  if (admm_setup(&h, &p, &o)) { fprintf(stderr, "admm_setup: %s\n", admm_last_error()); return 2; }
  admm_info info;
  if (admm_solve(h, NULL, NULL, &info)) { fprintf(stderr, "admm_solve: %s\n", admm_last_error()); return 2; }
  double* w = malloc(sizeof(double) * L * batch);
  double* z = malloc(sizeof(double) * L * batch);
  if (admm_get(h, w, z, NULL)) { fprintf(stderr, "admm_get: %s\n", admm_last_error()); return 2; }
  admm_free(h);

  /* z obeys the box, w obeys the dynamics, and they agree at convergence */
  double worst_box = 0, worst_dyn = 0, worst_gap = 0;
  for (int b = 0; b < batch; ++b) {
    double x[2] = {x0[b * n], x0[b * n + 1]};
    for (int k = 0; k < N; ++k) {
      const double* wk = w + (size_t)b * L + k * nb;
      const double* zk = z + (size_t)b * L + k * nb;
      const double u = wk[0];
      const double xn0 = A[0] * x[0] + A[2] * x[1] + B[0] * u, xn1 = A[1] * x[0] + A[3] * x[1] + B[1] * u;
      worst_dyn = fmax(worst_dyn, fmax(fabs(wk[1] - xn0), fabs(wk[2] - xn1)));
      worst_box = fmax(worst_box, fmax(zk[0] - 1.0, -1.0 - zk[0]));
      for (int r = 0; r < nb; ++r) worst_gap = fmax(worst_gap, fabs(wk[r] - zk[r]));
      x[0] = wk[1]; x[1] = wk[2];
    }
  }
  printf("iterations %d, converged %d/%d, max r %.2e, |w-z| %.2e, dynamics defect %.2e, box violation %.2e\n",
         info.iters_run, info.n_converged, batch, info.max_r, worst_gap, worst_dyn, worst_box);
  free(w); free(z);
  return (info.n_converged == batch && worst_box <= 0.0 && worst_dyn < 1e-12 && worst_gap < 1e-5) ? 0 : 1;
}

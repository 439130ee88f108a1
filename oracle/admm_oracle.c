/* admm_oracle.c -- CPU fp64 restatement of the batched optimal-control ADMM
 * iteration, in plain C with OpenMP over the batch.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product
 * path: only tests/, __graft_entry__.smoke() and the cpu_baseline leg of
 * bench.py may load this library, and only as the checker / the timed CPU
 * baseline.  libadmm_hip.so never links or calls it.
 *
 * PARITY UNPINNED.  /root/reference holds README.md:1-2 and a LICENSE only
 * (SURVEY.md §0): there is no reference source line, fixture or golden vector
 * for this path, so no function here can cite a reference file:line beyond
 * README.md:1-2 (which fixes the subject: ADMM for astrodynamics problems).
 * The arithmetic follows DESIGN.md §2 (this repository's own specification)
 * and is pinned by the solver-independent checks of tests/test_oracle.py and
 * by agreement with the NumPy restatement oracle/admm_ref.py.
 *
 * Layout: the same C ABI structs as include/admm_hip.h (column-major matrices,
 * per-QP vectors contiguous).  The x-update is the plain sequential sweep --
 * deliberately NOT the segmented parallel-in-time form the HIP kernels use --
 * so that agreement between the two is evidence, not self-agreement.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/admm_hip.h"

/* Factor: per stage K (m x n), Sinv (m x m), row-major. */
typedef struct {
  int N, n, m;
  double *K, *Sinv;
} ofactor;

static void matmul(const double* a, const double* b, double* c, int p, int q, int r) {
  /* c(p x r) = a(p x q) b(q x r), row-major */
  for (int i = 0; i < p; ++i)
    for (int j = 0; j < r; ++j) {
      double s = 0.0;
      for (int k = 0; k < q; ++k) s += a[i * q + k] * b[k * r + j];
      c[i * r + j] = s;
    }
}

static void transpose(const double* a, double* at, int p, int q) {
  for (int i = 0; i < p; ++i)
    for (int j = 0; j < q; ++j) at[j * p + i] = a[i * q + j];
}

/* In-place Gauss-Jordan inverse of a small SPD matrix with partial pivoting. */
static int invert(double* a, double* inv, int k) {
  for (int i = 0; i < k; ++i)
    for (int j = 0; j < k; ++j) inv[i * k + j] = (i == j) ? 1.0 : 0.0;
  for (int c = 0; c < k; ++c) {
    int piv = c;
    for (int r = c + 1; r < k; ++r)
      if (fabs(a[r * k + c]) > fabs(a[piv * k + c])) piv = r;
    if (fabs(a[piv * k + c]) < 1e-300) return 1;
    if (piv != c)
      for (int j = 0; j < k; ++j) {
        double t = a[c * k + j]; a[c * k + j] = a[piv * k + j]; a[piv * k + j] = t;
        t = inv[c * k + j]; inv[c * k + j] = inv[piv * k + j]; inv[piv * k + j] = t;
      }
    double d = 1.0 / a[c * k + c];
    for (int j = 0; j < k; ++j) { a[c * k + j] *= d; inv[c * k + j] *= d; }
    for (int r = 0; r < k; ++r) {
      if (r == c) continue;
      double f = a[r * k + c];
      if (f == 0.0) continue;
      for (int j = 0; j < k; ++j) { a[r * k + j] -= f * a[c * k + j]; inv[r * k + j] -= f * inv[c * k + j]; }
    }
  }
  return 0;
}

/* Stage matrices in row-major from the column-major ABI arrays. */
static void stage_AB(const admm_problem* p, int k, double* A, double* B) {
  int n = p->n, m = p->m;
  const double* Ac = p->A + (p->time_varying ? (size_t)k * n * n : 0);
  const double* Bc = p->B + (p->time_varying ? (size_t)k * n * m : 0);
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < n; ++j) A[i * n + j] = Ac[j * n + i];
    for (int j = 0; j < m; ++j) B[i * m + j] = Bc[j * n + i];
  }
}

/* Backward Riccati sweep on [P + rho I, G'; G, 0]. */
static int factorise(const admm_problem* p, double rho, ofactor* f) {
  int N = p->N, n = p->n, m = p->m;
  f->N = N; f->n = n; f->m = m;
  f->K = (double*)malloc(sizeof(double) * (size_t)N * m * n);
  f->Sinv = (double*)malloc(sizeof(double) * (size_t)N * m * m);
  double* P = (double*)malloc(sizeof(double) * n * n);
  double* A = (double*)malloc(sizeof(double) * n * n);
  double* B = (double*)malloc(sizeof(double) * n * m);
  double* Bt = (double*)malloc(sizeof(double) * n * m);
  double* At = (double*)malloc(sizeof(double) * n * n);
  double* PB = (double*)malloc(sizeof(double) * n * m);
  double* PA = (double*)malloc(sizeof(double) * n * n);
  double* S = (double*)malloc(sizeof(double) * m * m);
  double* BtPA = (double*)malloc(sizeof(double) * m * n);
  double* AtPA = (double*)malloc(sizeof(double) * n * n);
  double* SK = (double*)malloc(sizeof(double) * m * n);
  double* Kt = (double*)malloc(sizeof(double) * m * n);
  double* KtSK = (double*)malloc(sizeof(double) * n * n);
  int rc = 0;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) P[i * n + j] = p->QN[j * n + i] + (i == j ? rho : 0.0);
  for (int k = N - 1; k >= 0 && rc == 0; --k) {
    double* K = f->K + (size_t)k * m * n;
    double* Sinv = f->Sinv + (size_t)k * m * m;
    stage_AB(p, k, A, B);
    transpose(B, Bt, n, m);
    transpose(A, At, n, n);
    matmul(P, B, PB, n, n, m);
    matmul(Bt, PB, S, m, n, m);
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < m; ++j) S[i * m + j] += p->R[j * m + i] + (i == j ? rho : 0.0);
    for (int i = 0; i < m; ++i)
      for (int j = i + 1; j < m; ++j) { double a = 0.5 * (S[i * m + j] + S[j * m + i]); S[i * m + j] = S[j * m + i] = a; }
    {
      double* Sc = (double*)malloc(sizeof(double) * m * m);
      memcpy(Sc, S, sizeof(double) * m * m);
      rc = invert(Sc, Sinv, m);
      free(Sc);
      if (rc) break;
      for (int i = 0; i < m; ++i) {
        if (!(S[i * m + i] > 0.0)) rc = 1;
        for (int j = i + 1; j < m; ++j) { double a = 0.5 * (Sinv[i * m + j] + Sinv[j * m + i]); Sinv[i * m + j] = Sinv[j * m + i] = a; }
      }
    }
    matmul(P, A, PA, n, n, n);
    matmul(Bt, PA, BtPA, m, n, n);
    matmul(Sinv, BtPA, K, m, m, n);
    matmul(At, PA, AtPA, n, n, n);
    matmul(S, K, SK, m, m, n);
    transpose(K, Kt, m, n);
    matmul(Kt, SK, KtSK, n, m, n);
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j)
        P[i * n + j] = p->Q[j * n + i] + (i == j ? rho : 0.0) + AtPA[i * n + j] - KtSK[i * n + j];
    for (int i = 0; i < n; ++i)
      for (int j = i + 1; j < n; ++j) { double a = 0.5 * (P[i * n + j] + P[j * n + i]); P[i * n + j] = P[j * n + i] = a; }
  }
  free(P); free(A); free(B); free(Bt); free(At); free(PB); free(PA); free(S);
  free(BtPA); free(AtPA); free(SK); free(Kt); free(KtSK);
  return rc;
}

/* One QP, one x-update: g = q - rho (z - y) built on the fly; d is N*m scratch. */
static void x_update_one(const admm_problem* p, const ofactor* f, const double* AB,
                         double rho, const double* z, const double* y, const double* q,
                         const double* x0, double* w, double* d) {
  const int N = p->N, n = p->n, m = p->m, nb = n + m;
  double t[64], pv[64], h[64], x[64], u[64];
  for (int i = 0; i < n; ++i) t[i] = 0.0;
  for (int k = N - 1; k >= 0; --k) {
    const double* A = AB + (size_t)(p->time_varying ? k : 0) * (n * n + n * m);
    const double* B = A + n * n;
    const double* K = f->K + (size_t)k * m * n;
    const double* Sinv = f->Sinv + (size_t)k * m * m;
    const size_t o = (size_t)k * nb;
    for (int i = 0; i < n; ++i) {
      double gx = -rho * (z[o + m + i] - y[o + m + i]);
      if (q) gx += q[o + m + i];
      pv[i] = gx + t[i];
    }
    for (int j = 0; j < m; ++j) {
      double gu = -rho * (z[o + j] - y[o + j]);
      if (q) gu += q[o + j];
      double s = gu;
      for (int i = 0; i < n; ++i) s += B[i * m + j] * pv[i];
      h[j] = s;
    }
    for (int j = 0; j < m; ++j) {
      double s = 0.0;
      for (int l = 0; l < m; ++l) s += Sinv[j * m + l] * h[l];
      d[(size_t)k * m + j] = s;
    }
    for (int i = 0; i < n; ++i) {
      double s = 0.0;
      for (int l = 0; l < n; ++l) s += A[l * n + i] * pv[l];
      for (int j = 0; j < m; ++j) s -= K[j * n + i] * h[j];
      t[i] = s;
    }
  }
  for (int i = 0; i < n; ++i) x[i] = x0[i];
  for (int k = 0; k < N; ++k) {
    const double* A = AB + (size_t)(p->time_varying ? k : 0) * (n * n + n * m);
    const double* B = A + n * n;
    const double* K = f->K + (size_t)k * m * n;
    const size_t o = (size_t)k * nb;
    for (int j = 0; j < m; ++j) {
      double s = -d[(size_t)k * m + j];
      for (int i = 0; i < n; ++i) s -= K[j * n + i] * x[i];
      u[j] = s;
    }
    for (int i = 0; i < n; ++i) {
      double s = 0.0;
      for (int l = 0; l < n; ++l) s += A[i * n + l] * x[l];
      for (int j = 0; j < m; ++j) s += B[i * m + j] * u[j];
      pv[i] = s;
    }
    for (int j = 0; j < m; ++j) w[o + j] = u[j];
    for (int i = 0; i < n; ++i) { x[i] = pv[i]; w[o + m + i] = pv[i]; }
  }
}

/* Run the batch loop.  z, y (L*batch) are in/out, w is out.
 * iters/status/r/s are per-QP outputs (may be NULL).  If stop == 0 exactly
 * max_iter iterations run.  Returns 0, or 1 on bad input / failed factor. */
int oracle_solve(const admm_problem* p, const admm_options* o, int32_t stop,
                 double* z, double* y, double* w,
                 int32_t* iters, int32_t* status, double* r_out, double* s_out,
                 int32_t* iters_run, int32_t nthreads, double* rho_out, int32_t* rho_updates_out) {
  const int N = p->N, n = p->n, m = p->m, nb = n + m, batch = p->batch;
  if (n > 64 || m > 64 || N < 1 || batch < 1) return 1;
  const size_t L = (size_t)N * nb;
  double rho = o->rho;
  const double alpha = o->alpha;
  int n_updates = 0;
  double* rr = (double*)calloc(batch, sizeof(double));   /* last checked r, s of every QP */
  double* ss = (double*)calloc(batch, sizeof(double));
  ofactor f;
  if (factorise(p, rho, &f)) return 1;
  /* row-major copies of the stage dynamics */
  const int nst = p->time_varying ? N : 1;
  double* AB = (double*)malloc(sizeof(double) * (size_t)nst * (n * n + n * m));
  for (int k = 0; k < nst; ++k) stage_AB(p, k, AB + (size_t)k * (n * n + n * m), AB + (size_t)k * (n * n + n * m) + n * n);
  int32_t* st = (int32_t*)calloc(batch, sizeof(int32_t));
  int32_t* itv = (int32_t*)malloc(sizeof(int32_t) * batch);
  for (int b = 0; b < batch; ++b) itv[b] = o->max_iter;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
  int nthr = 1;
#ifdef _OPENMP
  nthr = omp_get_max_threads();
#endif
  double* dscr = (double*)malloc(sizeof(double) * (size_t)nthr * N * m);   /* x-update scratch (d), one slab per thread */
  const double sqrtL = sqrt((double)L);
  int it = 0;
  for (it = 1; it <= o->max_iter; ++it) {
    const int check = (it % o->check_interval == 0) || it == o->max_iter;
    int nconv = 0;
#pragma omp parallel for schedule(static) reduction(+ : nconv)
    for (int b = 0; b < batch; ++b) {
      double* zb = z + (size_t)b * L;
      double* yb = y + (size_t)b * L;
      double* wb = w + (size_t)b * L;
#ifdef _OPENMP
      double* d = dscr + (size_t)omp_get_thread_num() * N * m;   /* per-thread scratch, allocated once per solve */
#else
      double* d = dscr;
#endif
      x_update_one(p, &f, AB, rho, zb, yb, p->q ? p->q + (size_t)b * L : NULL,
                   p->x0 + (size_t)b * n, wb, d);
      double a_r = 0, a_s = 0, a_w = 0, a_z = 0, a_y = 0;
      for (int blk = 0; blk < N; ++blk) {       /* block by block, rows in order: the element order e = blk * nb + row */
        const size_t e0 = (size_t)blk * nb;
        const double* lob = p->lo + (p->stage_bounds ? e0 : 0);
        const double* hib = p->hi + (p->stage_bounds ? e0 : 0);
        /* block start: ||v_u|| of this block decides the scaling of its control rows */
        const double ub = p->unorm ? p->unorm[p->stage_bounds ? blk : 0] : INFINITY;
        const int soc = isfinite(ub);
        double cscale = 1.0;      /* thrust-magnitude projection factor of the current block */
        if (soc) {
          double ss = 0.0;
          for (int j = 0; j < m; ++j) {
            const double wj = wb[e0 + j], zj = zb[e0 + j];
            const double whj = (alpha == 1.0) ? wj : alpha * wj + (1.0 - alpha) * zj;
            const double vj = whj + yb[e0 + j];
            ss = fma(vj, vj, ss);
          }
          const double nrm = sqrt(ss);
          if (nrm > ub) cscale = ub / nrm;
        }
        for (int row = 0; row < nb; ++row) {
          const size_t e = e0 + row;
          const double lo = lob[row], hi = hib[row];
          const double wv = wb[e], zo = zb[e];
          const double wh = (alpha == 1.0) ? wv : alpha * wv + (1.0 - alpha) * zo;
          const double v = wh + yb[e];
          const double zn = (soc && row < m) ? ((cscale == 1.0) ? v : v * cscale) : fmin(fmax(v, lo), hi);
          const double yn = v - zn;
          zb[e] = zn; yb[e] = yn;
          if (check) {
            const double dr = wv - zn, ds = zn - zo;
            a_r += dr * dr; a_s += ds * ds; a_w += wv * wv; a_z += zn * zn; a_y += yn * yn;
          }
        }
      }
      if (check) {
        const double r = sqrt(a_r), s = rho * sqrt(a_s);
        const double e_pri = sqrtL * o->eps_abs + o->eps_rel * fmax(sqrt(a_w), sqrt(a_z));
        const double e_dua = sqrtL * o->eps_abs + o->eps_rel * rho * sqrt(a_y);
        if (r_out) r_out[b] = r;
        if (s_out) s_out[b] = s;
        rr[b] = r;
        ss[b] = s;
        if (r <= e_pri && s <= e_dua && !st[b]) { st[b] = 1; itv[b] = it; }
        nconv += st[b];
      }
    }
    if (stop && check && nconv == batch) break;
    /* adaptive rho: batch-level residual balancing over the unconverged QPs */
    if (check && o->adapt_interval > 0 && it % o->adapt_interval == 0 && n_updates < o->adapt_max &&
        it < o->max_iter) {
      double R = 0.0, S = 0.0;
      for (int b = 0; b < batch; ++b)
        if (!st[b]) { R += rr[b] * rr[b]; S += ss[b] * ss[b]; }
      const double mu2 = o->adapt_mu * o->adapt_mu;
      double rho_new = rho;
      if (R > mu2 * S) rho_new = rho * o->adapt_tau;
      else if (S > mu2 * R) rho_new = rho / o->adapt_tau;
      if (rho_new != rho) {
        const double c = rho / rho_new;
        for (size_t e = 0; e < L * (size_t)batch; ++e) y[e] *= c;
        free(f.K); free(f.Sinv);
        if (factorise(p, rho_new, &f)) { free(rr); free(ss); free(st); free(itv); free(AB); free(dscr); return 1; }
        rho = rho_new;
        ++n_updates;
      }
    }
  }
  if (it > o->max_iter) it = o->max_iter;
  if (iters_run) *iters_run = it;
  if (rho_out) *rho_out = rho;
  if (rho_updates_out) *rho_updates_out = n_updates;
  free(rr); free(ss);
  if (iters) memcpy(iters, itv, sizeof(int32_t) * batch);
  if (status) memcpy(status, st, sizeof(int32_t) * batch);
  free(st); free(itv); free(AB); free(dscr); free(f.K); free(f.Sinv);
  return 0;
}

/* Expose the factor for tests: K (N*m*n) and Sinv (N*m*m), row-major. */
int oracle_factor(const admm_problem* p, double rho, double* K, double* Sinv) {
  ofactor f;
  if (factorise(p, rho, &f)) return 1;
  memcpy(K, f.K, sizeof(double) * (size_t)p->N * p->m * p->n);
  memcpy(Sinv, f.Sinv, sizeof(double) * (size_t)p->N * p->m * p->m);
  free(f.K); free(f.Sinv);
  return 0;
}

int oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

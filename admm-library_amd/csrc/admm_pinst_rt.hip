// admm_pinst_rt.hip -- solver runtime: per-instance dynamics -- device factorisation, trial factorisation, set-up (admm_runtime.hpp; kernels in admm_pinst.hpp)
#include "admm_runtime.hpp"

namespace admm {
namespace rt {

// ---- per-instance dynamics ----
// Riccati factorisation of every QP on the device; ADMM_ERR_NUMERIC if some S_k is not positive definite.
// `only_marked`: refactor the QPs marked in todo_d (per-QP adaptive rule); rho comes from rho_d either way.
int pinst_factor(admm_handle* h, bool only_marked) {
  HIP_TRY(hipMemsetAsync(h->pfail, 0, sizeof(int), h->stream));
  admm::PLaunch l = plaunch_of(h);
  l.todo = only_marked ? h->todo_d : nullptr;
  if (!admm::launch_pinst(l, admm::PKernel::FACTOR, false)) return fail(ADMM_ERR_UNSUPPORTED, "no per-instance kernel for this (n, m)");
  HIP_TRY(hipGetLastError());
  int bad = 0, grown = 0;
  if (h->S > 1) {                                  // transfer matrices of the segments, from the new factor
    HIP_TRY(hipMemsetAsync(h->pgrow, 0, sizeof(int), h->stream));
    admm::launch_pinst(l, admm::PKernel::SEGMENTS, false);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(&grown, h->pgrow, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  }
  HIP_TRY(hipMemcpyAsync(&bad, h->pfail, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (bad) return fail(ADMM_ERR_NUMERIC, "R + rho I + B'PB is not positive definite for some QP");
  // the conditioning bound of admm_setup, per QP; the per-QP adaptive rule (only_marked) refactors without it -- a
  // refused change of one QP could not be undone without the factor it has just overwritten
  if (grown && !only_marked && h->auto_segments)   // (a segment count the caller fixed is the caller's responsibility, as with shared dynamics)
    return fail(ADMM_ERR_NUMERIC, "the segment transfer matrices of some QP grow beyond the conditioning bound (max entry > 100) "
                                  "with " + std::to_string(h->S) + " segments; use options.segments = 1");
  return ADMM_OK;
}

// every QP's rho := rho
int pinst_fill_rho(admm_handle* h, double rho) {
  std::vector<double> r(h->pitch, rho);
  HIP_TRY(hipMemcpyAsync(h->rho_d, r.data(), sizeof(double) * r.size(), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return ADMM_OK;
}

// shared weights as row-major device arrays + A, B per instance, into the given buffers (the handle's, or the trial set)
int pinst_upload_dynamics(admm_handle* h, const admm_problem* p, double* Ad, double* Bd, double* Qd, double* Rd, double* QNd) {
  const int n = h->n, m = h->m;
  std::vector<double> Q((size_t)n * n), R((size_t)m * m), QN((size_t)n * n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) { Q[(size_t)i * n + j] = 0.5 * (p->Q[(size_t)j * n + i] + p->Q[(size_t)i * n + j]); QN[(size_t)i * n + j] = 0.5 * (p->QN[(size_t)j * n + i] + p->QN[(size_t)i * n + j]); }
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < m; ++j) R[(size_t)i * m + j] = 0.5 * (p->R[(size_t)j * m + i] + p->R[(size_t)i * m + j]);
  // (on the handle's stream, like everything else that touches its arrays; the vectors live until the synchronisation below)
  HIP_TRY(hipMemcpyAsync(Qd, Q.data(), sizeof(double) * Q.size(), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(Rd, R.data(), sizeof(double) * R.size(), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(QNd, QN.data(), sizeof(double) * QN.size(), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  int rc;
  const bool rowmaj = (h->opt.flags & ADMM_FLAG_ROW_MAJOR) != 0;        // the caller's blocks are row-major: transposed on the device
  if (h->pi_tiled) {
    if ((rc = upload_tiled(h, p->A, Ad, n * n, rowmaj ? n : 0, n))) return rc;
    if ((rc = upload_tiled(h, p->B, Bd, n * m, rowmaj ? n : 0, m))) return rc;
    return ADMM_OK;
  }
  if ((rc = upload_transposed(h, p->A, Ad, h->N * n * n, rowmaj ? n : 0, n))) return rc;
  if ((rc = upload_transposed(h, p->B, Bd, h->N * n * m, rowmaj ? n : 0, m))) return rc;
  return ADMM_OK;
}

// the box (per instance or shared) and the thrust-magnitude bounds
int pinst_upload_bounds(admm_handle* h, const admm_problem* p) {
  int rc;
  if (h->pbounds) {
    if ((rc = upload_transposed(h, p->lo, h->lod, h->L))) return rc;
    if ((rc = upload_transposed(h, p->hi, h->hid, h->L))) return rc;
    if (h->lodT) {                                  // the sweeps of the wide shapes read the box as tiles, staged with the operands
      if ((rc = upload_tiled(h, p->lo, h->lodT, h->nb))) return rc;
      if ((rc = upload_tiled(h, p->hi, h->hidT, h->nb))) return rc;
    }
    std::vector<double> ub(h->N, INFINITY);         // thrust-magnitude bound per stage (shared by the batch)
    if (p->unorm)
      for (int k = 0; k < h->N; ++k) ub[k] = p->unorm[k];
    HIP_TRY(hipMemcpyAsync(h->ub, ub.data(), sizeof(double) * h->N, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
  } else if ((rc = upload_bounds(h, p))) {
    return rc;
  }
  return ADMM_OK;
}

int pinst_upload(admm_handle* h, const admm_problem* p) {
  int rc;
  if ((rc = pinst_upload_dynamics(h, p, h->Ad, h->Bd, h->Qd, h->Rd, h->QNd))) return rc;
  return pinst_upload_bounds(h, p);
}

// Trial buffers (see admm_handle): K / S always, A / B / weights when the problem data change.
int pinst_alloc_trial(admm_handle* h, bool dynamics) {
  const size_t P = h->pitch;
  const int n = h->n, m = h->m, N = h->N;
  int rc;
  if (!h->Kd2 && (rc = dalloc(&h->Kd2, (size_t)N * m * n * P))) return rc;
  if (!h->Sd2 && (rc = dalloc(&h->Sd2, (size_t)N * m * m * P))) return rc;
  if (!h->rho2_d && (rc = dalloc(&h->rho2_d, P))) return rc;
  if (!h->qflag_d && (rc = dalloc(&h->qflag_d, P))) return rc;
  if (!h->nveto_d && (rc = dalloc(&h->nveto_d, (size_t)1))) return rc;
  if (dynamics) {
    if (!h->Ad2 && (rc = dalloc(&h->Ad2, (size_t)N * n * n * P))) return rc;
    if (!h->Bd2 && (rc = dalloc(&h->Bd2, (size_t)N * n * m * P))) return rc;
    if (!h->Qd2 && (rc = dalloc(&h->Qd2, (size_t)n * n))) return rc;
    if (!h->Rd2 && (rc = dalloc(&h->Rd2, (size_t)m * m))) return rc;
    if (!h->QNd2 && (rc = dalloc(&h->QNd2, (size_t)n * n))) return rc;
  }
  return ADMM_OK;
}

// TRIAL factorisation: the Riccati recursion of the QPs marked in `todo` (NULL = all) with the given dynamics, weights and rho
// into the trial K / S, then -- with segments -- the transfer matrices WITHOUT storing them.  Leaves one verdict per QP in
// qflag_d (bit 0: some S_k not positive definite, bit 1: a transfer matrix beyond the conditioning bound) and returns how many
// real QPs carry each bit.  Nothing the iteration reads is written.
int pinst_try(admm_handle* h, const double* Ad, const double* Bd, const double* Qd, const double* Rd, const double* QNd,
              const double* rhov, const int* todo, int* n_not_pd, int* n_grown) {
  admm::PLaunch l = plaunch_of(h);
  l.Ad = Ad; l.Bd = Bd; l.Q = Qd; l.R = Rd; l.QN = QNd; l.rhov = rhov; l.todo = todo;
  l.Kd = h->Kd2; l.Sd = h->Sd2; l.qflag = h->qflag_d;
  l.Omd = nullptr; l.Psd = nullptr; l.Segd = nullptr;
  HIP_TRY(hipMemsetAsync(h->qflag_d, 0, sizeof(int) * (size_t)h->pitch, h->stream));
  HIP_TRY(hipMemsetAsync(h->pfail, 0, sizeof(int), h->stream));
  HIP_TRY(hipMemsetAsync(h->pgrow, 0, sizeof(int), h->stream));
  if (!admm::launch_pinst(l, admm::PKernel::FACTOR, false)) return fail(ADMM_ERR_UNSUPPORTED, "no per-instance kernel for this (n, m)");
  HIP_TRY(hipGetLastError());
  if (h->S > 1) {
    admm::launch_pinst(l, admm::PKernel::SEGMENTS, false);
    HIP_TRY(hipGetLastError());
  }
  std::vector<int> q(h->pitch);
  HIP_TRY(hipMemcpyAsync(q.data(), h->qflag_d, sizeof(int) * q.size(), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  *n_not_pd = *n_grown = 0;
  for (int b = 0; b < h->batch; ++b) { *n_not_pd += q[b] & 1; *n_grown += (q[b] >> 1) & 1; }
  return ADMM_OK;
}

// Segment transfer matrices of the factor in force (after a committed change; cannot fail: the trial run has checked them).
int pinst_segments(admm_handle* h) {
  if (h->S <= 1) return ADMM_OK;
  HIP_TRY(hipMemsetAsync(h->pgrow, 0, sizeof(int), h->stream));
  admm::PLaunch l = plaunch_of(h);
  admm::launch_pinst(l, admm::PKernel::SEGMENTS, false);
  HIP_TRY(hipGetLastError());
  return ADMM_OK;
}

int setup_pinst(admm_handle* h, const admm_problem* p) {
  const admm_options& o = h->opt;
  if (o.precision_mode != ADMM_PRECISION_FP64) return fail(ADMM_ERR_UNSUPPORTED, "precision_mode: the MFMA forms need batch-shared dynamics");
  if (o.flags & (ADMM_FLAG_UNFUSED | ADMM_FLAG_GRAPH))
    return fail(ADMM_ERR_UNSUPPORTED, "ADMM_FLAG_UNFUSED / ADMM_FLAG_GRAPH are not available with per-instance dynamics");
  {
    admm::PLaunch lq{};
    lq.n = p->n; lq.m = p->m;
    if (!admm::launch_pinst(lq, admm::PKernel::XB, true))
      return fail(ADMM_ERR_UNSUPPORTED, "(n, m) = (" + std::to_string(p->n) + ", " + std::to_string(p->m) +
                                            ") has no per-instance kernel; compiled: " + admm::dims_pinst());
  }
  h->pinst = true;
  h->pbounds = p->stage_bounds == 2;
  h->pi_tiled = admm::pinst_rows_only(p->n, p->m);
  // wide shapes ((8, 4), (12, 6), ...; csrc/admm_pinst_wide.hpp): every kernel in its rows-over-lanes form, whatever the batch
  const bool wide = admm::pinst_rows_only(p->n, p->m);
  if (wide) {        // the staged sweeps hold two stages of operand tiles per wave in LDS: more than the 64 KB default limit
    hipDeviceProp_t prop;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    const size_t need = admm::pinst_wide_lds_bytes(p->n, p->m);
    if (need > (size_t)prop.maxSharedMemoryPerMultiProcessor)
      return fail(ADMM_ERR_UNSUPPORTED, "per-instance dynamics at (n, m) = (" + std::to_string(p->n) + ", " + std::to_string(p->m) + ") need " +
                                            std::to_string(need) + " bytes of LDS per workgroup; this device offers " +
                                            std::to_string((size_t)prop.maxSharedMemoryPerMultiProcessor));
  }
  // Segments in time (csrc/admm_pinst.hpp): one lane sweeps one segment of one QP, so an iteration takes N / S dependent
  // stage round trips instead of N.  Automatic count: enough (64-QP wave, segment) pairs for one wave per SIMD, segments
  // of at least 8 stages, at most 64 (32 from 512 QPs: the scan is S sequential steps per QP); large batches fill the chip
  // alone (S = 1).
  h->auto_segments = o.segments == 0;
  {
    int S = o.segments;
    if (S == 0 && wide) {
      // a wave serves QPW QPs: enough (wave, segment) pairs for one wave per SIMD, segments of at least 8 stages
      const int qpw = p->n <= 2 ? 32 : (p->n <= 4 ? 16 : (p->n <= 8 ? 8 : 4));
      const int waves = h->pitch / qpw;
      S = (4 * h->num_cus + waves - 1) / waves;
      if (S > h->N / 8) S = h->N / 8;
      if (S > (waves <= 64 ? 64 : 32)) S = waves <= 64 ? 64 : 32;
      if (std::getenv("ADMM_PI_NO_SEGMENTS")) S = 1;
    } else if (S == 0) {
      const int waves = h->pitch / 64;
      // (measured, N = 1000, n = 6: 64 QPs 2.26 -> 0.27 ms per iteration, 4096 QPs 3.04 -> 1.60 ms; from 8192 QPs the batch
      //  alone reaches the HBM roofline and the segments' extra operands -- Omega_k, Psi_k: +16 % bytes -- only cost)
      // (round 3, 4096 QPs = 64 waves: the sweeps are flat from 4 segments on and the per-QP scan costs 3 + 1.8 S us -- N = 200:
      //  3090 batch-iterations/s with 16 segments, 3376 with 8; N = 1000: 674 / 682.  1024 QPs still want their 32.)
      S = waves <= 64 ? ((waves >= 16 ? 2 : 4) * h->num_cus) / std::max(1, waves) : 1;
      if (S > h->N / 8) S = h->N / 8;
      if (S > (waves <= 4 ? 64 : 32)) S = waves <= 4 ? 64 : 32;     // (from 512 QPs the sweeps stop gaining, the scan keeps growing)
      if (std::getenv("ADMM_PI_NO_SEGMENTS")) S = 1;
    }
    if (S > h->N) S = h->N;
    if (S < 1) S = 1;
    if (S > 64) return fail(ADMM_ERR_INVALID, "options.segments: at most 64 with per-instance dynamics");
    h->S = S;
    // few QPs: a lane per (QP, row) instead of a lane per QP (ADMM_PI_LANE_PER_QP=1 / ADMM_PI_ROWS=1 force either form)
    // (measured, N = 1000: 64 QPs 29 -> 23 us per sweep, N = 200: 15 -> 9 us; from 128 QPs the 8-QP waves' 64-byte accesses lose:
    //  256 QPs 44 -> 86 us)
    h->pi_rows = h->pitch <= 64;
    if (h->has_soc) h->pi_rows = false;          // (narrow shapes: the thrust-magnitude forms exist for the one-lane kernels)
    if (std::getenv("ADMM_PI_LANE_PER_QP")) h->pi_rows = false;
    if (std::getenv("ADMM_PI_ROWS") && !h->has_soc) h->pi_rows = true;
    if (wide) h->pi_rows = true;
    h->pi_rows_factor = std::getenv("ADMM_PI_ROWS_FACTOR") != nullptr;
  }
  h->alt = h->alt_allowed = false;
  h->time_varying = 2;
  h->stage_bounds = p->stage_bounds;
  {  // z-kernel chunking of the shared-bounds read-out kernel (as in admm_setup)
    const int col_groups = (h->pitch / 2 + Z_THREADS - 1) / Z_THREADS;
    int chunks = std::max(1, (h->num_cus + col_groups - 1) / col_groups);
    int zr = ((h->L + chunks - 1) / chunks + 3) / 4 * 4;
    if (zr < 4) zr = 4;
    if (h->has_soc) zr = ((zr + h->nb - 1) / h->nb) * h->nb;   // block-structured kernels: whole blocks per chunk
    h->zrows = zr;
    h->zchunks = (h->L + zr - 1) / zr;
  }
  const size_t P = h->pitch, L = h->L;
  const int n = h->n, m = h->m, N = h->N;
  HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  int rc;
#define PD(ptr, cnt) do { if ((rc = dalloc(&(ptr), (size_t)(cnt)))) return rc; HIP_TRY(hipMemsetAsync((ptr), 0, sizeof(*(ptr)) * (size_t)(cnt), h->stream)); } while (0)
  PD(h->w, L * P); PD(h->z, L * P); PD(h->y, L * P); PD(h->v, L * P);
  if (h->has_q) PD(h->q, L * P);
  PD(h->dbuf, (size_t)N * m * P);
  {  // scan_in = x0 | tseg | eseg,  scan_out = t_in | x_in   ([S][n][pitch] each; only x0 with one segment)
    const size_t Sn = (size_t)h->S * n;
    PD(h->scan_in, (size_t)(n + 2 * Sn) * P);
    PD(h->scan_out, 2 * Sn * P);
    h->x0 = h->scan_in;
    h->tseg = h->scan_in + (size_t)n * P;
    h->eseg = h->tseg + Sn * P;
    h->tin = h->scan_out;
    h->xin = h->scan_out + Sn * P;
    std::vector<int32_t> ss(h->S + 1);
    for (int sgm = 0; sgm <= h->S; ++sgm) ss[sgm] = (int32_t)(((int64_t)sgm * N) / h->S);
    PD(h->seg_start, (size_t)h->S + 1);
    HIP_TRY(hipMemcpyAsync(h->seg_start, ss.data(), sizeof(int32_t) * ss.size(), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    PD(h->pgrow, 1);
    if (h->S > 1) {
      PD(h->Omd, (size_t)N * n * m * P);
      PD(h->Psd, (size_t)N * m * n * P);
      PD(h->Segd, (size_t)h->S * 3 * n * n * P);
    }
  }
  PD(h->part, (size_t)std::max(h->zchunks, h->S) * 5 * P);
  PD(h->resid, 5 * P);
  PD(h->lo, L); PD(h->hi, L); PD(h->ub, (size_t)N);
  PD(h->Ad, (size_t)N * n * n * P); PD(h->Bd, (size_t)N * n * m * P);
  PD(h->Kd, (size_t)N * m * n * P); PD(h->Sd, (size_t)N * m * m * P);
  if (h->pbounds) { PD(h->lod, L * P); PD(h->hid, L * P); }
  if (h->pbounds && h->pi_tiled) { PD(h->lodT, L * P); PD(h->hidT, L * P); }
  PD(h->Qd, (size_t)n * n); PD(h->Rd, (size_t)m * m); PD(h->QNd, (size_t)n * n);
  PD(h->pfail, 1); PD(h->status, P); PD(h->iters, P); PD(h->nconv, 1);
  PD(h->rho_d, P); PD(h->cscale_d, P); PD(h->nupd_d, P); PD(h->todo_d, P); PD(h->nchanged_d, 1);
  h->stage_rows = std::max(L, (size_t)N * n * n);
  if ((rc = dalloc(&h->stage, h->stage_rows * (size_t)h->batch))) return rc;
#undef PD
  // The zero-fills above are asynchronous on the handle's NON-BLOCKING stream, and the small uploads below (weights, shared box, thrust
  // bounds) are synchronous copies on the null stream, which does not wait for it: with GBs of fills queued (a wide shape's operand
  // arrays) a copy could land first and be zeroed afterwards -- Q = R = 0, a factor of nonsense, every QP wrong from the first iteration
  // (found by the full-horizon optimality certificate of round 3; the second large handle of a process, whose allocations are fast).
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipHostMalloc((void**)&h->h_nconv, sizeof(int), hipHostMallocDefault));
  if ((rc = pinst_upload(h, p))) return rc;
  if ((rc = upload_transposed(h, p->x0, h->x0, n))) return rc;
  if (h->has_q && (rc = upload_transposed(h, p->q, h->q, (int)L))) return rc;
  if ((rc = pinst_fill_rho(h, o.rho))) return rc;
  rc = pinst_factor(h);
  if (rc == ADMM_ERR_NUMERIC && h->S > 1 && h->auto_segments) {      // conditioning bound hit: sweep the whole horizon per lane
    h->S = 1;
    rc = pinst_factor(h);
  }
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(h->stream));
  return ADMM_OK;
}


}  // namespace rt
}  // namespace admm

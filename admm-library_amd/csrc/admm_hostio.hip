// admm_hostio.hip -- solver runtime: host <-> device transfers, validation, uploads of a factor, handle lifetime (admm_runtime.hpp)
#include "admm_runtime.hpp"

namespace admm {
namespace rt {


// Host -> device copy of a caller's (pageable) array.  Large ones go through two pinned bounce buffers: host threads fill one
// while the DMA engine drains the other -- hipMemcpy from pageable memory alone ran at a few GB/s and made
// admm_update_problem of 4096 x 1000 per-instance stages a 0.7 s call (round 2).
constexpr size_t PIN_BYTES = (size_t)32 << 20;
int upload_h2d(admm_handle* h, void* dst, const void* src, size_t bytes) {
  if (bytes < 2 * PIN_BYTES) {
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
    return ADMM_OK;
  }
  for (int i = 0; i < 2; ++i) {
    if (!h->pin[i]) HIP_TRY(hipHostMalloc((void**)&h->pin[i], PIN_BYTES, hipHostMallocDefault));
    if (!h->pin_ev[i]) HIP_TRY(hipEventCreateWithFlags(&h->pin_ev[i], hipEventDisableTiming));
  }
  int slot = 0;
  for (size_t off = 0; off < bytes; off += PIN_BYTES, slot ^= 1) {
    const size_t len = std::min(PIN_BYTES, bytes - off);
    HIP_TRY(hipEventSynchronize(h->pin_ev[slot]));            // the copy that last read this buffer is done (no-op if never recorded)
    unsigned char* pb = h->pin[slot];
    const unsigned char* sb = static_cast<const unsigned char*>(src) + off;
    host_parallel(len, 1, [pb, sb](size_t b, size_t e) { std::memcpy(pb + b, sb + b, e - b); });
    HIP_TRY(hipMemcpyAsync(static_cast<unsigned char*>(dst) + off, pb, len, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipEventRecord(h->pin_ev[slot], h->stream));
  }
  return ADMM_OK;
}

// Device -> host copy into a caller's (pageable) array, the mirror of upload_h2d: the DMA engine fills one pinned buffer while host
// threads empty the other into the caller's memory (admm_get of three 295 MB vectors: 57 ms through pageable copies).
int download_d2h(admm_handle* h, void* dst, const void* src, size_t bytes) {
  if (bytes < 2 * PIN_BYTES) {
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return ADMM_OK;
  }
  for (int i = 0; i < 2; ++i) {
    if (!h->pin[i]) HIP_TRY(hipHostMalloc((void**)&h->pin[i], PIN_BYTES, hipHostMallocDefault));
    if (!h->pin_ev[i]) HIP_TRY(hipEventCreateWithFlags(&h->pin_ev[i], hipEventDisableTiming));
  }
  HIP_TRY(hipEventSynchronize(h->pin_ev[0]));                 // (an upload may have been the buffers' last user)
  HIP_TRY(hipEventSynchronize(h->pin_ev[1]));
  auto drain = [&](int slot, size_t off, size_t len) -> int {
    HIP_TRY(hipEventSynchronize(h->pin_ev[slot]));            // the DMA into this buffer is done
    const unsigned char* pb = h->pin[slot];
    unsigned char* db = static_cast<unsigned char*>(dst) + off;
    host_parallel(len, 1, [pb, db](size_t b, size_t e) { std::memcpy(db + b, pb + b, e - b); });
    return ADMM_OK;
  };
  int slot = 0, rc;
  size_t prev_off = 0, prev_len = 0;
  for (size_t off = 0; off < bytes; off += PIN_BYTES, slot ^= 1) {
    const size_t len = std::min(PIN_BYTES, bytes - off);
    HIP_TRY(hipMemcpyAsync(h->pin[slot], static_cast<const unsigned char*>(src) + off, len, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipEventRecord(h->pin_ev[slot], h->stream));
    if (prev_len && (rc = drain(slot ^ 1, prev_off, prev_len))) return rc;     // the previous chunk, while this one is in flight
    prev_off = off;
    prev_len = len;
  }
  return drain(slot ^ 1, prev_off, prev_len);
}

// QP-major host array (batch x rows) -> batch-minor device array (rows x pitch)
int upload_transposed(admm_handle* h, const double* src, double* dst, int rows, int nr, int nc) {
  if (rows == h->L && windowed(h)) {
    // a state-sized array of a handle that holds a stage window only: rows [r0, r0 + Lw) of every QP's vector (a strided 2-D
    // copy out of the caller's L x batch array), transposed into the window (dst is the biased pointer)
    const size_t r0 = win_row0(h), Lw = win_rows(h);
    HIP_TRY(hipMemcpy2DAsync(h->stage, Lw * sizeof(double), src + r0, (size_t)h->L * sizeof(double), Lw * sizeof(double), h->batch,
                             hipMemcpyHostToDevice, h->stream));
    dim3 gridw(((int)Lw + admm::T_TILE - 1) / admm::T_TILE, (h->pitch + admm::T_TILE - 1) / admm::T_TILE), blockw(admm::T_TILE * 8);
    hipLaunchKernelGGL(admm::to_batch_minor_kernel, gridw, blockw, 0, h->stream, h->stage, dst + win_bias(h), h->batch, (int)Lw, h->pitch, 0, 0);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    return ADMM_OK;
  }
  if ((size_t)rows > h->stage_rows) return fail(ADMM_ERR_INVALID, "internal: staging buffer too small");
  int rc_up;
  if ((rc_up = upload_h2d(h, h->stage, src, sizeof(double) * (size_t)rows * h->batch))) return rc_up;
  dim3 grid((rows + admm::T_TILE - 1) / admm::T_TILE, (h->pitch + admm::T_TILE - 1) / admm::T_TILE), block(admm::T_TILE * 8);
  hipLaunchKernelGGL(admm::to_batch_minor_kernel, grid, block, 0, h->stream, h->stage, dst, h->batch, rows, h->pitch, nr, nc);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  return ADMM_OK;
}

// QP-major host operand (batch x N x E) -> the wide shapes' tiled device layout
int upload_tiled(admm_handle* h, const double* src, double* dst, int E, int nr, int nc) {
  const size_t rows = (size_t)h->N * E;
  if (rows > h->stage_rows) return fail(ADMM_ERR_INVALID, "internal: staging buffer too small");
  int rc_up;
  if ((rc_up = upload_h2d(h, h->stage, src, sizeof(double) * rows * h->batch))) return rc_up;
  admm::launch_to_tiled(h->stream, h->stage, dst, h->batch, h->N, E, h->n, h->pitch, nr, nc);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  return ADMM_OK;
}

int download_transposed(admm_handle* h, const double* src, double* dst, int rows) {
  if (rows == h->L && windowed(h)) {          // the window's rows only; the caller's other rows are left alone
    const size_t r0 = win_row0(h), Lw = win_rows(h);
    dim3 gridw(((int)Lw + admm::T_TILE - 1) / admm::T_TILE, (h->pitch + admm::T_TILE - 1) / admm::T_TILE), blockw(admm::T_TILE * 8);
    hipLaunchKernelGGL(admm::from_batch_minor_kernel, gridw, blockw, 0, h->stream, src + win_bias(h), h->stage, h->batch, (int)Lw, h->pitch);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy2DAsync(dst + r0, (size_t)h->L * sizeof(double), h->stage, Lw * sizeof(double), Lw * sizeof(double), h->batch,
                             hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return ADMM_OK;
  }
  dim3 grid((rows + admm::T_TILE - 1) / admm::T_TILE, (h->pitch + admm::T_TILE - 1) / admm::T_TILE), block(admm::T_TILE * 8);
  hipLaunchKernelGGL(admm::from_batch_minor_kernel, grid, block, 0, h->stream, src, h->stage, h->batch, rows, h->pitch);
  HIP_TRY(hipGetLastError());
  return download_d2h(h, dst, h->stage, sizeof(double) * (size_t)rows * h->batch);
}

bool finite_all(const double* a, size_t cnt) {          // (threaded from ~1 M entries: 2.7 GB of problem data at 4096 x 1000 stages)
  std::atomic<bool> ok{true};
  host_parallel(cnt, sizeof(double), [a, &ok](size_t b, size_t e) {
    // |x| < inf  <=>  finite; the exponent test on the bit pattern vectorises (isfinite in a loop with an early exit does not)
    uint64_t bad = 0;
    for (size_t i = b; i < e; ++i) {
      uint64_t u;
      std::memcpy(&u, a + i, sizeof u);
      bad |= ((u >> 52) & 0x7ff) == 0x7ff;
    }
    if (bad) ok.store(false, std::memory_order_relaxed);
  });
  return ok.load();
}

int validate_options(const admm_options* o) {
  if (!(o->rho > 0.0) || !std::isfinite(o->rho)) return fail(ADMM_ERR_INVALID, "rho must be positive and finite");
  if (!(o->alpha > 0.0 && o->alpha < 2.0)) return fail(ADMM_ERR_INVALID, "alpha must lie in (0, 2)");
  if (!(o->eps_abs >= 0.0) || !(o->eps_rel >= 0.0)) return fail(ADMM_ERR_INVALID, "eps_abs / eps_rel must be >= 0");
  if (o->max_iter < 1) return fail(ADMM_ERR_INVALID, "max_iter must be >= 1");
  if (o->check_interval < 1) return fail(ADMM_ERR_INVALID, "check_interval must be >= 1");
  if (o->segments < 0 || o->zrows < 0) return fail(ADMM_ERR_INVALID, "segments / zrows must be >= 0");
  if (o->adapt_interval < 0 || o->adapt_max < 0) return fail(ADMM_ERR_INVALID, "adapt_interval / adapt_max must be >= 0");
  if (o->precision_mode < 0 || o->precision_mode > 2) return fail(ADMM_ERR_INVALID, "precision_mode must be ADMM_PRECISION_FP64, _MIXED or _FP64_MFMA");
  if (o->reserved != 0) return fail(ADMM_ERR_INVALID, "options.reserved must be 0");
  if (o->adapt_interval > 0) {
    if (o->adapt_interval % o->check_interval != 0)
      return fail(ADMM_ERR_INVALID, "adapt_interval must be a multiple of check_interval");
    if (!(o->adapt_mu > 1.0) || !(o->adapt_tau > 1.0) || !std::isfinite(o->adapt_mu) || !std::isfinite(o->adapt_tau))
      return fail(ADMM_ERR_INVALID, "adapt_mu and adapt_tau must be finite and > 1");
  }
  return ADMM_OK;
}

int validate_problem(const admm_problem* p) {
  if (p->N < 1 || p->n < 1 || p->m < 1 || p->batch < 1) return fail(ADMM_ERR_INVALID, "N, n, m, batch must be positive");
  if (!p->A || !p->B || !p->Q || !p->R || !p->QN || !p->x0 || !p->lo || !p->hi)
    return fail(ADMM_ERR_INVALID, "A, B, Q, R, QN, x0, lo, hi must be non-NULL");
  const int nb = p->n + p->m;
  const size_t L = (size_t)p->N * nb;
  if (L * (size_t)p->batch > ((size_t)1 << 40)) return fail(ADMM_ERR_INVALID, "problem too large");
  if (L > (size_t)0x7fffffff) return fail(ADMM_ERR_INVALID, "L = N (n + m) exceeds 2^31 - 1");
  if (p->time_varying < 0 || p->time_varying > 2 || p->stage_bounds < 0 || p->stage_bounds > 2)
    return fail(ADMM_ERR_INVALID, "time_varying / stage_bounds must be 0, 1 or 2");
  if (p->stage_bounds == 2 && p->time_varying != 2)
    return fail(ADMM_ERR_INVALID, "per-instance bounds (stage_bounds = 2) need per-instance dynamics (time_varying = 2)");
  if (p->time_varying == 2) {
    if (!p->Q || !p->R || !p->QN) return fail(ADMM_ERR_INVALID, "Q, R, QN must be non-NULL");
    if (!finite_all(p->A, (size_t)p->n * p->n * p->N * p->batch) || !finite_all(p->B, (size_t)p->n * p->m * p->N * p->batch) ||
        !finite_all(p->Q, (size_t)p->n * p->n) || !finite_all(p->R, (size_t)p->m * p->m) || !finite_all(p->QN, (size_t)p->n * p->n))
      return fail(ADMM_ERR_INVALID, "non-finite entry in A, B, Q, R or QN");
  }
  const size_t nbnd = (size_t)nb * (p->stage_bounds ? p->N : 1) * (p->stage_bounds == 2 ? p->batch : 1);
  {
    std::atomic<size_t> first_bad{SIZE_MAX};         // smallest offending index (threads take disjoint ranges)
    const double *lo = p->lo, *hi = p->hi;
    host_parallel(nbnd, 2 * sizeof(double), [lo, hi, &first_bad](size_t b, size_t e) {
      for (size_t i = b; i < e; ++i)
        if (!(lo[i] <= hi[i]) || lo[i] == INFINITY || hi[i] == -INFINITY) {      // (!(<=) also catches NaN)
          size_t cur = first_bad.load();
          while (i < cur && !first_bad.compare_exchange_weak(cur, i)) {}
          return;
        }
    });
    const size_t i = first_bad.load();
    if (i != SIZE_MAX) {
      if (std::isnan(p->lo[i]) || std::isnan(p->hi[i])) return fail(ADMM_ERR_INVALID, "NaN in bounds");
      if (p->lo[i] > p->hi[i]) return fail(ADMM_ERR_INVALID, "lo > hi at bound index " + std::to_string(i));
      return fail(ADMM_ERR_INVALID, "lo = +inf or hi = -inf");
    }
  }
  if (p->unorm) {
    const int cnt = p->stage_bounds ? p->N : 1;
    for (int k = 0; k < cnt; ++k) {
      const double ub = p->unorm[k];
      if (std::isnan(ub) || !(ub > 0.0)) return fail(ADMM_ERR_INVALID, "unorm entries must be positive (inf = off)");
      if (std::isfinite(ub))
        for (int b = 0; b < (p->stage_bounds == 2 ? p->batch : 1); ++b)       // (per-instance box: every QP's)
          for (int j = 0; j < p->m; ++j) {
            const size_t o = ((size_t)b * cnt + k) * nb + j;
            if (std::isfinite(p->lo[o]) || std::isfinite(p->hi[o]))
              return fail(ADMM_ERR_INVALID, "control rows must be unbounded (-inf, inf) where unorm is finite");
          }
    }
  }
  if (!finite_all(p->x0, (size_t)p->n * p->batch)) return fail(ADMM_ERR_INVALID, "non-finite entry in x0");
  if (p->q && !finite_all(p->q, L * p->batch)) return fail(ADMM_ERR_INVALID, "non-finite entry in q");
  return ADMM_OK;
}


// MIXED precision: fp32 = the mixed MFMA kernels; otherwise (the fp64 refinement phase of
// admm_solve) the all-fp64 MFMA kernels on their own records.  Both forms share every other device array and the
// alternation schedule; what an alternating iteration left pending is dropped at the switch (the next iteration
// starts with a backward sweep).
void set_mixed_form(admm_handle* h, bool fp32) {
  if (h->opt.precision_mode != ADMM_PRECISION_MIXED) return;
  h->mfma_on = fp32;
  h->mfma_refine = !fp32;
  h->alt_state = admm_handle::ALT_NONE;
}

// admm_last_warning(): the forward-elimination form of a factor failed its host check, so the handle runs (or falls back
// to) the plain fused path.  `when` names the call.
void warn_alt_gate(const admm::Factor& f, double rho, const char* when) {
  char buf[512];
  if (f.alt_check >= 0.0)
    std::snprintf(buf, sizeof buf, "%s: the forward-elimination form failed its host check at rho = %g (relative mismatch %.3g > %.1g): "
                  "the handle runs the plain fused path (xb + xfz kernels, no alternation: ~8 B per stacked element and iteration more)",
                  when, rho, f.alt_check, ALT_GATE);
  else
    std::snprintf(buf, sizeof buf, "%s: the forward-elimination form could not be built at rho = %g (a singular A_k or filter covariance): "
                  "the handle runs the plain fused path (xb + xfz kernels, no alternation)", when, rho);
  g_warn = buf;
}

// Conditioning guard of the parallel-in-time form (see admm_setup): largest entry of the dense scan matrices.
double scan_growth(const admm::Factor& f) {
  double g = 0.0;
  for (double v : f.scanW) g = std::max(g, std::fabs(v));
  return g;
}

void destroy_graph(admm_handle* h) {
  for (int v = 0; v < 16; ++v) {
    if (h->graph_exec[v]) { (void)hipGraphExecDestroy(h->graph_exec[v]); h->graph_exec[v] = nullptr; }
    if (h->graph[v]) { (void)hipGraphDestroy(h->graph[v]); h->graph[v] = nullptr; }
  }
}

void release(admm_handle* h) {
  if (!h) return;
  h->spec.clear();               // joins the background factorisations (they read the handle's problem copy)
  h->spec_stale.clear();
  (void)hipSetDevice(h->device);
  destroy_graph(h);
  // the big per-stage arrays are held as pointers biased by the stage window (admm_runtime.hpp): undo that before freeing
  for (double** b : {&h->w, &h->z, &h->y, &h->v, &h->q})
    if (*b) *b += win_bias(h);
  for (double** b : {&h->dbuf, &h->mvec})
    if (*b) *b += win_bias_m(h);
  double** bufs[] = {&h->w, &h->z, &h->y, &h->v, &h->q, &h->dbuf, &h->scan_in, &h->scan_out, &h->scanWp,
                     &h->part, &h->resid, &h->lo, &h->hi, &h->ub, &h->recB, &h->recF, &h->recS, &h->stage,
                     &h->recFE, &h->recBE, &h->mvec, &h->scanWpB};
  for (auto b : bufs)
    if (*b) { (void)hipFree(*b); *b = nullptr; }
  {
    double** pb[] = {&h->Ad, &h->Bd, &h->Kd, &h->Sd, &h->lod, &h->hid, &h->lodT, &h->hidT, &h->Qd, &h->Rd, &h->QNd,
                     &h->Ad2, &h->Bd2, &h->Kd2, &h->Sd2, &h->Qd2, &h->Rd2, &h->QNd2, &h->rho2_d};
    if (h->qflag_d) { (void)hipFree(h->qflag_d); h->qflag_d = nullptr; }
    if (h->nveto_d) { (void)hipFree(h->nveto_d); h->nveto_d = nullptr; }
    for (auto b : pb)
      if (*b) { (void)hipFree(*b); *b = nullptr; }
    if (h->pfail) { (void)hipFree(h->pfail); h->pfail = nullptr; }
    if (h->rho_d) { (void)hipFree(h->rho_d); h->rho_d = nullptr; }
    double** sb[] = {&h->Omd, &h->Psd, &h->Segd};
    for (auto b : sb)
      if (*b) { (void)hipFree(*b); *b = nullptr; }
    if (h->pgrow) { (void)hipFree(h->pgrow); h->pgrow = nullptr; }
    if (h->cscale_d) { (void)hipFree(h->cscale_d); h->cscale_d = nullptr; }
    int** ib[] = {&h->nupd_d, &h->todo_d, &h->nchanged_d};
    for (auto b : ib)
      if (*b) { (void)hipFree(*b); *b = nullptr; }
  }
  if (h->scanWd) { (void)hipFree(h->scanWd); h->scanWd = nullptr; }
  if (h->scanWBd) { (void)hipFree(h->scanWBd); h->scanWBd = nullptr; }
  if (h->scan_rows) { (void)hipFree(h->scan_rows); h->scan_rows = nullptr; }
  if (h->scan_rowsB) { (void)hipFree(h->scan_rowsB); h->scan_rowsB = nullptr; }
  if (h->recMF) { (void)hipFree(h->recMF); h->recMF = nullptr; }
  if (h->recMB) { (void)hipFree(h->recMB); h->recMB = nullptr; }
  if (h->recMF64) { (void)hipFree(h->recMF64); h->recMF64 = nullptr; }
  if (h->recMB64) { (void)hipFree(h->recMB64); h->recMB64 = nullptr; }
  int** ibufs[] = {&h->seg_start, &h->status, &h->iters, &h->nconv, &h->scan_range, &h->scan_rangeB, &h->status1, &h->iters1};
  for (auto b : ibufs)
    if (*b) { (void)hipFree(*b); *b = nullptr; }
  if (h->h_nconv) { (void)hipHostFree(h->h_nconv); h->h_nconv = nullptr; }
  for (int i = 0; i < 2; ++i) {
    if (h->pin[i]) { (void)hipHostFree(h->pin[i]); h->pin[i] = nullptr; }
    if (h->pin_ev[i]) { (void)hipEventDestroy(h->pin_ev[i]); h->pin_ev[i] = nullptr; }
  }
  if (h->stream) { (void)hipStreamDestroy(h->stream); h->stream = nullptr; }
  delete h;
}


// device copies of everything in h->fac (records, scan matrices; the alternating set if enabled)
// dense scan matrix + each row's non-zero column range [begin, end) for xscan_gemv_kernel
int upload_scan_dense(const std::vector<double>& W, int M, int K, double* Wd, int* rows_d) {
  std::vector<int32_t> rr((size_t)2 * M);
  for (int r = 0; r < M; ++r) {
    int kb = K, ke = 0;
    const double* row = &W[(size_t)r * K];
    for (int k = 0; k < K; ++k)
      if (row[k] != 0.0) { if (k < kb) kb = k; ke = k + 1; }
    if (ke < kb) { kb = 0; ke = 0; }
    rr[2 * r] = kb; rr[2 * r + 1] = ke;
  }
  HIP_TRY(hipMemcpy(Wd, W.data(), sizeof(double) * W.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(rows_d, rr.data(), sizeof(int32_t) * rr.size(), hipMemcpyHostToDevice));
  return ADMM_OK;
}

int upload_factor(admm_handle* h) {
  HIP_TRY(hipMemcpy(h->recB, h->fac.recB.data(), sizeof(double) * h->fac.recB.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->recF, h->fac.recF.data(), sizeof(double) * h->fac.recF.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->recS, h->fac.recS.data(), sizeof(double) * h->fac.recS.size(), hipMemcpyHostToDevice));
  if (!h->scan_gemv) HIP_TRY(hipMemcpy(h->scanWp, h->fac.scanWp.data(), sizeof(double) * h->fac.scanWp.size(), hipMemcpyHostToDevice));
  if (!h->scan_gemv) HIP_TRY(hipMemcpy(h->scan_range, h->fac.scanRange.data(), sizeof(int32_t) * h->fac.scanRange.size(), hipMemcpyHostToDevice));
  int rc;
  if (h->scan_gemv && (rc = upload_scan_dense(h->fac.scanW, h->fac.scanM, h->fac.scanK, h->scanWd, h->scan_rows))) return rc;
  h->alt_state = admm_handle::ALT_NONE;
  if (!h->fac.alt_ok) {                                             // the forward-elimination form did not survive the refactor
    if (h->alt_allowed) warn_alt_gate(h->fac, h->fac.rho, "refactor");
    h->alt = false; h->alt_allowed = false;
  }
  if (h->mfma_mode) {
    HIP_TRY(hipMemcpy(h->recMF, h->fac.recMF.data(), h->fac.recMF.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->recMB, h->fac.recMB.data(), h->fac.recMB.size(), hipMemcpyHostToDevice));
    if (h->recMF64) {
      HIP_TRY(hipMemcpy(h->recMF64, h->fac.recMF64.data(), h->fac.recMF64.size(), hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(h->recMB64, h->fac.recMB64.data(), h->fac.recMB64.size(), hipMemcpyHostToDevice));
    }
  }
  if (h->alt_allowed) {
    HIP_TRY(hipMemcpy(h->recFE, h->fac.recFE.data(), sizeof(double) * h->fac.recFE.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->recBE, h->fac.recBE.data(), sizeof(double) * h->fac.recBE.size(), hipMemcpyHostToDevice));
    if (!h->scan_gemv) HIP_TRY(hipMemcpy(h->scanWpB, h->fac.scanWpB.data(), sizeof(double) * h->fac.scanWpB.size(), hipMemcpyHostToDevice));
    if (!h->scan_gemv) HIP_TRY(hipMemcpy(h->scan_rangeB, h->fac.scanRangeB.data(), sizeof(int32_t) * h->fac.scanRangeB.size(), hipMemcpyHostToDevice));
    if (h->scan_gemv && (rc = upload_scan_dense(h->fac.scanWB, h->fac.scanM, h->fac.scanK, h->scanWBd, h->scan_rowsB))) return rc;
  }
  return ADMM_OK;
}

// bounds expanded to one entry per stacked row (standalone z kernels), thrust-magnitude bound per stage
int upload_bounds(admm_handle* h, const admm_problem* p) {
  const size_t L = h->L;
  std::vector<double> lo(L), hi(L);
  for (size_t e = 0; e < L; ++e) {
    const size_t blk = e / h->nb, row = e % h->nb;
    lo[e] = p->lo[(p->stage_bounds ? blk * h->nb : 0) + row];
    hi[e] = p->hi[(p->stage_bounds ? blk * h->nb : 0) + row];
  }
  // (on the handle's stream -- it is non-blocking, a null-stream copy would not be ordered with what is queued on it)
  HIP_TRY(hipMemcpyAsync(h->lo, lo.data(), sizeof(double) * L, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->hi, hi.data(), sizeof(double) * L, hipMemcpyHostToDevice, h->stream));
  std::vector<double> ub(h->N, INFINITY);
  if (p->unorm)
    for (int k = 0; k < h->N; ++k) ub[k] = p->unorm[p->stage_bounds ? k : 0];
  HIP_TRY(hipMemcpyAsync(h->ub, ub.data(), sizeof(double) * h->N, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return ADMM_OK;
}

// host copy of the shared problem data (the caller's pointers are never kept): admm_set_rho refactors from it
void keep_shared(admm_handle* h, const admm_problem* p) {
  const size_t nst = p->time_varying ? (size_t)p->N : 1, nbd = (size_t)h->nb * (p->stage_bounds ? p->N : 1);
  h->pA.assign(p->A, p->A + nst * p->n * p->n);
  h->pB.assign(p->B, p->B + nst * p->n * p->m);
  h->pQ.assign(p->Q, p->Q + (size_t)p->n * p->n);
  h->pR.assign(p->R, p->R + (size_t)p->m * p->m);
  h->pQN.assign(p->QN, p->QN + (size_t)p->n * p->n);
  h->plo.assign(p->lo, p->lo + nbd);
  h->phi.assign(p->hi, p->hi + nbd);
  h->pun.clear();
  if (p->unorm) h->pun.assign(p->unorm, p->unorm + (p->stage_bounds ? p->N : 1));
  h->time_varying = p->time_varying;
  h->stage_bounds = p->stage_bounds;
  // every STATE row unbounded at EVERY stage: its dual is identically zero, which lets the non-residual kernel forms
  // skip reading its v (XFREE, xfze_kernel)
  bool open = std::getenv("ADMM_NO_SKIPV") == nullptr;
  for (size_t k = 0; open && k < (p->stage_bounds ? (size_t)p->N : 1); ++k)
    for (int r = h->m; open && r < h->nb; ++r)
      open = p->lo[k * h->nb + r] == -INFINITY && p->hi[k * h->nb + r] == INFINITY;
  h->xfree = open;
}

bool problem_has_soc(const admm_problem* p) {
  bool soc = false;
  if (p->unorm)
    for (int k = 0; k < (p->stage_bounds ? p->N : 1); ++k) soc = soc || std::isfinite(p->unorm[k]);
  return soc;
}



}  // namespace rt
}  // namespace admm

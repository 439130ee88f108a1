// admm_profile.hip -- admm_profile: per-kernel HIP-event timing on the handle's stream (admm_runtime.hpp)
#include "admm_runtime.hpp"


using namespace admm::rt;

extern "C" {

int admm_profile(admm_handle* h, int32_t iters, int32_t residuals, int32_t fused_path, double ms[6]) {
  if (!h || !ms) return fail(ADMM_ERR_INVALID, "NULL argument");
  if (iters < 1 || iters > 4096) return fail(ADMM_ERR_INVALID, "iters must lie in [1, 4096]");
  if (h->pinst && fused_path != 1) return fail(ADMM_ERR_UNSUPPORTED, "admm_profile: per-instance dynamics run the plain fused path only (fused_path = 1)");
  if (h->ts_n) return fail(ADMM_ERR_UNSUPPORTED, "admm_profile is not available on a time-sharded handle");
  h->xfree_mode = 1;
  HIP_TRY(hipSetDevice(h->device));
  constexpr int NE = 6;     // events per iteration
  if ((fused_path == 2 || fused_path == 3) && !h->alt) return fail(ADMM_ERR_UNSUPPORTED, "the alternating-direction kernels are not enabled for this handle");
  if (fused_path == 3) {
    // Back-to-back mode: a cross-check of mode 2 that records NO event between launches.  The two fused kernels are
    // launched as `iters` consecutive (xfze, xbze) pairs with no scan in between -- same bytes, same instructions and
    // the same sweep alternation (each kernel starts on the rows the previous one has just written, which is worth
    // ~9 %: re-running ONE of them in a row measured 157 us against 136 us) -- and each scan form `iters` times in a
    // row.  Without the scans the numbers are not ADMM iterates, so the state v is parked in the w buffer, restored
    // afterwards, and one plain iteration makes the handle consistent again.
    int rc3 = ADMM_OK;
    const bool res3 = residuals != 0;
    if (!h->v_valid) {
      if ((rc3 = flush_finalize(h))) return rc3;
      if ((rc3 = enqueue_iteration(h, false, false))) return rc3;
      after_iterations(h, 1);
    }
    if ((rc3 = flush_finalize(h))) return rc3;
    const size_t bytes = sizeof(double) * (size_t)h->L * h->pitch;
    HIP_TRY(hipMemcpyAsync(h->w, h->v, bytes, hipMemcpyDeviceToDevice, h->stream));
    hipEvent_t e3[8];
    for (auto& e : e3) HIP_TRY(hipEventCreate(&e));
    h->alt_state = admm_handle::ALT_NONE;
    rc3 = launch_xb(h, true);
    if (!rc3) rc3 = launch_xscan_mfma(h, false, false);
    if (!rc3) rc3 = launch_x(h, admm::XKernel::XFZE, false, res3);     // valid db / m_in / x_end for the first xbze
    if (!rc3) rc3 = launch_xscan_mfma(h, true, false);
    if (!rc3) rc3 = launch_x(h, admm::XKernel::XBZE, false, res3);
    HIP_TRY(hipEventRecord(e3[0], h->stream));
    for (int it = 0; it < iters && !rc3; ++it) {
      rc3 = launch_x(h, admm::XKernel::XFZE, false, res3);
      if (!rc3) rc3 = launch_x(h, admm::XKernel::XBZE, false, res3);
    }
    HIP_TRY(hipEventRecord(e3[1], h->stream));
    HIP_TRY(hipEventRecord(e3[2], h->stream));
    HIP_TRY(hipEventRecord(e3[3], h->stream));
    HIP_TRY(hipEventRecord(e3[4], h->stream));
    for (int it = 0; it < iters && !rc3; ++it) rc3 = launch_xscan_mfma(h, false, res3);
    HIP_TRY(hipEventRecord(e3[5], h->stream));
    HIP_TRY(hipEventRecord(e3[6], h->stream));
    for (int it = 0; it < iters && !rc3; ++it) rc3 = launch_xscan_mfma(h, true, res3);
    HIP_TRY(hipEventRecord(e3[7], h->stream));
    HIP_TRY(hipMemcpyAsync(h->v, h->w, bytes, hipMemcpyDeviceToDevice, h->stream));
    h->alt_state = admm_handle::ALT_NONE;
    h->v_valid = true; h->zy_valid = false;
    if (!rc3) rc3 = enqueue_iteration(h, res3, true);
    if (!rc3 && res3) rc3 = launch_finalize(h, 0, chunks_of_iteration(h));
    if (!rc3) after_iterations(h, 1);
    if (res3) h->resid_valid = true;
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int v = 0; v < 6; ++v) ms[v] = 0.0;
    if (!rc3) {
      float tp = 0.f, ta = 0.f, tb = 0.f;
      HIP_TRY(hipEventElapsedTime(&tp, e3[0], e3[1]));          // iters (xfze, xbze) pairs
      HIP_TRY(hipEventElapsedTime(&ta, e3[4], e3[5]));          // iters scans (W)
      HIP_TRY(hipEventElapsedTime(&tb, e3[6], e3[7]));          // iters scans (WB)
      ms[0] = ta / iters;
      ms[1] = ms[3] = 0.5 * tp / iters;                          // mean of the two fused kernels
      ms[2] = tb / iters;
      ms[5] = ms[0] + ms[1] + ms[2] + ms[3];
    }
    for (auto& e : e3) (void)hipEventDestroy(e);
    return rc3;
  }
  std::vector<hipEvent_t> ev((size_t)iters * NE);
  for (auto& e : ev) HIP_TRY(hipEventCreate(&e));
  int rc = ADMM_OK;
  const bool res = residuals != 0;
  h->alt_state = admm_handle::ALT_NONE;       // the plain kernels are profiled; they overwrite the scan operands
  if (fused_path == 2) {
    // `iters` PAIRS of alternating iterations (forward form, backward form); ms[] = scan, xfze,
    // scan, xbze, 0, whole pair (with residuals each scan launch also finalises the iteration before
    // it).  2 iters + 1 (+1) iterations are applied.
    if (!h->v_valid) {
      rc = enqueue_iteration(h, false, false);
      if (!rc) after_iterations(h, 1);
    }
    if (!rc) rc = launch_xb(h, true);
    // without residuals these are the kernels a run spends its time in: each is followed by another fused alternating kernel
    // of the same kind (the closing IT_FWD below writes everything), so the XFREE = 2 forms apply (enqueue_one)
    if (!res && h->opt.alpha == 1.0 && std::getenv("ADMM_NO_SKIPV_STORE") == nullptr) h->xfree_mode = 2;
    for (int it = 0; it < iters && !rc; ++it) {
      hipEvent_t* e = &ev[(size_t)it * NE];
      HIP_TRY(hipEventRecord(e[0], h->stream));
      rc = launch_xscan_mfma(h, false, res && it > 0);
      HIP_TRY(hipEventRecord(e[1], h->stream));
      if (!rc) rc = launch_x(h, admm::XKernel::XFZE, false, res);
      HIP_TRY(hipEventRecord(e[2], h->stream));
      if (!rc) rc = launch_xscan_mfma(h, true, res);
      HIP_TRY(hipEventRecord(e[3], h->stream));
      if (!rc) rc = launch_x(h, admm::XKernel::XBZE, false, res);
      HIP_TRY(hipEventRecord(e[4], h->stream));
      HIP_TRY(hipEventRecord(e[5], h->stream));
    }
    h->xfree_mode = 1;
    if (!rc) rc = enqueue_form(h, IT_FWD, res, res);      // never stop after the backward form
    if (!rc && res) rc = launch_finalize(h, 0, h->S);
    if (!rc) after_form(h, IT_FWD);
  }
  for (int it = 0; it < iters && !rc && fused_path != 2; ++it) {
    hipEvent_t* e = &ev[(size_t)it * NE];
    const bool use_v = fused_path && h->v_valid;
    if (!fused_path && (rc = ensure_zy(h))) break;
    HIP_TRY(hipEventRecord(e[0], h->stream));
    rc = launch_xb(h, use_v);
    HIP_TRY(hipEventRecord(e[1], h->stream));
    if (!rc) rc = launch_xscan(h);
    HIP_TRY(hipEventRecord(e[2], h->stream));
    if (!rc) rc = fused_path ? launch_xfz(h, res, use_v) : launch_xf(h);
    HIP_TRY(hipEventRecord(e[3], h->stream));
    if (!rc && !fused_path) rc = launch_z(h, res);
    if (!rc) {
      if (fused_path) { h->v_valid = true; h->zy_valid = false; h->w_stale = true; }
      else            { h->zy_valid = true; h->v_valid = false; h->w_stale = false; }
    }
    HIP_TRY(hipEventRecord(e[4], h->stream));
    if (!rc && res) rc = launch_finalize(h, 0, fused_path ? h->S : h->zchunks);
    HIP_TRY(hipEventRecord(e[5], h->stream));
  }
  if (res) h->resid_valid = true;
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (int v = 0; v < 6; ++v) ms[v] = 0.0;
  if (!rc) {
    for (int it = 0; it < iters; ++it) {
      hipEvent_t* e = &ev[(size_t)it * NE];
      for (int v = 0; v < 5; ++v) {
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, e[v], e[v + 1]));
        ms[v] += t;
      }
      float t = 0.f;
      HIP_TRY(hipEventElapsedTime(&t, e[0], e[5]));
      ms[5] += t;
    }
    for (int v = 0; v < 6; ++v) ms[v] /= iters;
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  return rc;
}

int admm_get_history(admm_handle* h, int32_t capacity, int32_t* count, int32_t* iteration, int32_t* n_converged,
                     double* max_r, double* max_s, double* rho) {
  if (!h || !count) return fail(ADMM_ERR_INVALID, "NULL argument");
  if (capacity < 0) return fail(ADMM_ERR_INVALID, "capacity must be >= 0");
  *count = (int32_t)h->history.size();
  const size_t n = std::min((size_t)capacity, h->history.size());
  for (size_t i = 0; i < n; ++i) {
    const admm_handle::HistoryEntry& e = h->history[i];
    if (iteration) iteration[i] = e.it;
    if (n_converged) n_converged[i] = e.nconv;
    if (max_r) max_r[i] = e.max_r;
    if (max_s) max_s[i] = e.max_s;
    if (rho) rho[i] = e.rho;
  }
  return ADMM_OK;
}


}  // extern "C"

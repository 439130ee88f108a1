// admm_rho_update.hip -- solver runtime: rho changes and problem updates -- background candidate factors of the adaptive rule, admm_set_rho, admm_update_problem (admm_runtime.hpp)
#include "admm_runtime.hpp"

namespace admm {
namespace rt {

admm_problem shared_problem(const admm_handle* h) {
  admm_problem p{};
  p.N = h->N; p.n = h->n; p.m = h->m; p.batch = h->batch;
  p.time_varying = h->time_varying; p.stage_bounds = h->stage_bounds;
  p.A = h->pA.data(); p.B = h->pB.data(); p.Q = h->pQ.data(); p.R = h->pR.data(); p.QN = h->pQN.data();
  p.lo = h->plo.data(); p.hi = h->phi.data();
  p.unorm = h->pun.empty() ? nullptr : h->pun.data();
  return p;
}

// ---- background refactors for the adaptive-rho rule ----
// The rule can only move rho to rho * tau or rho / tau, and a host factorisation costs as much as tens of iterations
// (n = 12: as much as 100+).  While admm_solve iterates, both candidates are factorised on background host threads from
// the handle's own copy of the problem; when the rule fires, set_rho_internal finds the factor ready (or waits for the
// rest of it).  The factor of the rho being left is kept as a candidate too.  Same code, same inputs: the factor -- and
// so every iterate -- is the one a synchronous refactor would produce.  ADMM_NO_SPECULATE=1 turns this off.
bool spec_enabled(const admm_handle* h) {
  const bool off = std::getenv("ADMM_NO_SPECULATE") != nullptr;     // read per call: tests switch it within a process
  // (only inside a solve: a caller sweeping admm_set_rho on an adaptive handle outside one must not accumulate a full Factor
  //  copy per call -- ADVICE r02)
  return !off && h->solve_active && !h->pinst && h->opt.adapt_interval > 0 && h->rho_updates < h->opt.adapt_max;
}

void spec_reap(admm_handle* h, bool all) {          // drop stale entries whose thread has finished (all: join them)
  if (all) { h->spec_stale.clear(); return; }
  // a finished thread is recognised by rc having been published; unfinished ones stay until the next reap
  for (size_t i = 0; i < h->spec_stale.size();)
    if (!h->spec_stale[i]->th.joinable() || __atomic_load_n(&h->spec_stale[i]->rc, __ATOMIC_ACQUIRE) != INT32_MIN)
      h->spec_stale.erase(h->spec_stale.begin() + i);
    else ++i;
}

std::unique_ptr<SpecFactor> spec_take(admm_handle* h, double rho) {
  for (size_t i = 0; i < h->spec.size(); ++i)
    if (h->spec[i]->rho == rho) {
      std::unique_ptr<SpecFactor> sp = std::move(h->spec[i]);
      h->spec.erase(h->spec.begin() + i);
      if (sp->th.joinable()) sp->th.join();
      return sp;
    }
  return nullptr;
}

void spec_start(admm_handle* h) {
  spec_reap(h, false);
  if (!spec_enabled(h)) {
    for (auto& s : h->spec) h->spec_stale.push_back(std::move(s));
    h->spec.clear();
    return;
  }
  const double cand[2] = {h->opt.rho * h->opt.adapt_tau, h->opt.rho / h->opt.adapt_tau};   // as admm_solve_adapt forms them
  for (size_t i = 0; i < h->spec.size();)
    if (h->spec[i]->rho != cand[0] && h->spec[i]->rho != cand[1]) {
      h->spec_stale.push_back(std::move(h->spec[i]));
      h->spec.erase(h->spec.begin() + i);
    } else {
      ++i;
    }
  for (double rho : cand) {
    if (!(rho > 0.0) || !std::isfinite(rho)) continue;
    bool have = false;
    for (auto& s : h->spec) have = have || s->rho == rho;
    if (have) continue;
    std::unique_ptr<SpecFactor> sp(new SpecFactor);
    sp->rho = rho;
    sp->rc = INT32_MIN;                                     // "not finished" (read only after join, or by spec_reap)
    SpecFactor* s = sp.get();
    const admm_handle* hc = h;
    try {
    s->th = std::thread([hc, s] {
      admm::set_factor_thread_cap(8);                       // two of these run beside the thread that launches kernels
      const admm_problem p = shared_problem(hc);
      int rc;
      try {
        admm::Factor f;
        std::string err;
        rc = admm::factorise(p, s->rho, hc->S, f, err, hc->mfma_mode, !hc->scan_gemv, hc->ts_n);
        s->f = std::move(f);
        s->err = std::move(err);
      } catch (...) {
        rc = ADMM_ERR_ALLOC;
        s->err = "background factorisation ran out of memory";
      }
      __atomic_store_n(&s->rc, rc, __ATOMIC_RELEASE);
    });
    } catch (...) {                                         // no thread to be had: this candidate is factorised on demand
      continue;
    }
    h->spec.push_back(std::move(sp));
  }
}

// Refactor for a new rho, re-upload the records, rescale the scaled dual.  The state is
// switched to (z, y) form so that y *= rho_old / rho_new is applied to the very numbers the
// iteration produced (bit-identical to the oracle); the next iteration reads z, y directly.
int set_rho_internal(admm_handle* h, double rho_new) {
  if (!(rho_new > 0.0) || !std::isfinite(rho_new)) return fail(ADMM_ERR_INVALID, "rho must be positive and finite");
  if (rho_new == h->opt.rho && !h->pinst) return ADMM_OK;     // (per-instance: the QPs' own rho may have moved away from it)
  if (h->pinst) {
    // every QP's rho := rho_new (the per-QP adaptive rule may have moved them apart): y_b *= rho_b / rho_new.
    // TRIAL first: the factor of rho_new goes into the scratch K / S and the conditioning bound is evaluated without storing
    // anything; only a change that passes is committed (pointer swap), so a refused one leaves the handle untouched.
    int rc;
    if ((rc = pinst_alloc_trial(h, false))) return rc;
    const size_t P = h->pitch;
    {
      std::vector<double> cand(P, rho_new);
      HIP_TRY(hipMemcpyAsync(h->rho2_d, cand.data(), sizeof(double) * P, hipMemcpyHostToDevice, h->stream));
      HIP_TRY(hipStreamSynchronize(h->stream));
    }
    int not_pd = 0, grown = 0;
    if ((rc = pinst_try(h, h->Ad, h->Bd, h->Qd, h->Rd, h->QNd, h->rho2_d, nullptr, &not_pd, &grown))) return rc;
    if (not_pd) return fail(ADMM_ERR_NUMERIC, "rho change refused: R + rho I + B'PB is not positive definite for " + std::to_string(not_pd) + " QP(s)");
    if (grown && h->auto_segments)
      return fail(ADMM_ERR_NUMERIC, "rho change refused: with this rho the segment transfer matrices of " + std::to_string(grown) +
                                        " QP(s) grow beyond the conditioning bound (max entry > 100) with the handle's " +
                                        std::to_string(h->S) + " segments; use options.segments = 1");
    if ((rc = ensure_w(h))) return rc;          // w of the last x-update must be rebuilt with the OLD factor (still in place)
    if ((rc = ensure_zy(h))) return rc;
    std::vector<double> old(P), c(P);
    HIP_TRY(hipMemcpyAsync(old.data(), h->rho_d, sizeof(double) * P, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    std::vector<int> all(P, 1);
    for (size_t b = 0; b < P; ++b) c[b] = old[b] / rho_new;
    HIP_TRY(hipMemcpyAsync(h->cscale_d, c.data(), sizeof(double) * P, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->todo_d, all.data(), sizeof(int) * P, hipMemcpyHostToDevice, h->stream));
    admm::launch_padapt_scale(h->stream, h->y, h->cscale_d, h->todo_d, h->L, h->pitch);
    HIP_TRY(hipGetLastError());
    h->zy_valid = true;
    h->v_valid = false;
    if ((rc = pinst_fill_rho(h, rho_new))) return rc;       // (synchronises: the kernels of the old factor are done)
    std::swap(h->Kd, h->Kd2);                               // commit
    std::swap(h->Sd, h->Sd2);
    if ((rc = pinst_segments(h))) return rc;
    h->opt.rho = rho_new;
    return ADMM_OK;
  }
  const admm_problem p = shared_problem(h);
  admm::Factor f;
  std::string err;
  int rc;
  static const bool debug = std::getenv("ADMM_SPEC_DEBUG") != nullptr;
  auto t0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!debug) return;
    const auto t1 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[admm] set_rho %-22s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  };
  std::unique_ptr<SpecFactor> sp = spec_take(h, rho_new);
  if (sp && sp->rc != ADMM_ERR_ALLOC) {                           // factorised in the background (or kept from before)
    rc = sp->rc;
    err = std::move(sp->err);
    f = std::move(sp->f);
    ++h->spec_hits;
  } else {
    rc = admm::factorise(p, rho_new, h->S, f, err, h->mfma_mode, !h->scan_gemv, h->ts_n);
    ++h->spec_misses;
  }
  lap("factor (take / compute)");
  if (rc) return fail(rc, err);
  if (f.recB.size() != h->fac.recB.size() || f.scanWp.size() != h->fac.scanWp.size())
    return fail(ADMM_ERR_NUMERIC, "internal: record sizes changed on refactor");
  // the segment count is frozen on a live handle, so the conditioning guard of admm_setup can only refuse here
  if (h->auto_segments && h->S > 1 && scan_growth(f) > SCAN_GROWTH_MAX)
    return fail(ADMM_ERR_NUMERIC, "rho change refused: with this rho the segment transfer matrices of the handle's " +
                                      std::to_string(h->S) + " segments grow beyond the conditioning bound (max |W| > 100); "
                                      "set the handle up with this rho (fewer segments are chosen then) or give options.segments");
  if ((rc = ensure_w(h))) return rc;          // w of the last x-update must be rebuilt with the OLD records
  if ((rc = ensure_zy(h))) return rc;
  {
    const double c = h->opt.rho / rho_new;
    const size_t count2 = win_rows(h) * h->pitch / 2;       // pitch is even; the handle's stage window
    hipLaunchKernelGGL(admm::scale_kernel, dim3(2048), dim3(256), 0, h->stream, h->y + win_bias(h), c, count2);
    HIP_TRY(hipGetLastError());
  }
  h->zy_valid = true;
  h->v_valid = false;
  HIP_TRY(hipStreamSynchronize(h->stream));                // kernels of the old rho are done before the records change
  lap("state to (z, y), sync");
  if (spec_enabled(h)) {                                   // the rule may come back to the rho it leaves: keep that factor
    bool have = false;                                     // (once: spec_take returns the first entry of a rho)
    for (auto& sp2 : h->spec) have = have || sp2->rho == h->opt.rho;
    if (!have) {
      std::unique_ptr<SpecFactor> old(new SpecFactor);
      old->rho = h->opt.rho;
      old->f = std::move(h->fac);
      h->spec.push_back(std::move(old));
    }
  }
  h->fac = std::move(f);
  lap("keep / move factor");
  if ((rc = upload_factor(h))) return rc;
  lap("upload");
  h->opt.rho = rho_new;
  destroy_graph(h);                                        // rho is a captured kernel argument
  return ADMM_OK;
}


}  // namespace rt
}  // namespace admm

using namespace admm::rt;

extern "C" {

int admm_set_rho(admm_handle* h, double rho) {
  if (!h) return fail(ADMM_ERR_INVALID, "NULL handle");
  g_warn.clear();
  HIP_TRY(hipSetDevice(h->device));
  return set_rho_internal(h, rho);
}

int admm_update_problem(admm_handle* h, const admm_problem* p) {
  if (!h || !p) return fail(ADMM_ERR_INVALID, "NULL argument");
  g_warn.clear();
  HIP_TRY(hipSetDevice(h->device));
  int rc;
  if ((rc = validate_problem(p))) return rc;
  if (p->N != h->N || p->n != h->n || p->m != h->m || p->batch != h->batch)
    return fail(ADMM_ERR_INVALID, "admm_update_problem: N, n, m, batch must equal those of admm_setup");
  if ((p->q != nullptr) != h->has_q)
    return fail(ADMM_ERR_INVALID, "admm_update_problem: q must be given iff the handle was set up with one");
  if (problem_has_soc(p) != h->has_soc)
    return fail(ADMM_ERR_INVALID, "admm_update_problem: a thrust-magnitude bound cannot be added to or removed from a handle");
  if ((p->time_varying == 2) != h->pinst || (h->pinst && (p->stage_bounds == 2) != h->pbounds))
    return fail(ADMM_ERR_INVALID, "admm_update_problem: per-instance dynamics / bounds cannot be added to or removed from a handle");
  if (h->pinst) {
    // TRIAL first: the new dynamics and weights go into the scratch buffers and are factorised there, every QP
    // with the rho it has; only if every factor exists and meets the conditioning bound is anything of the handle replaced
    // (pointer swaps) -- "on failure the handle is unchanged" holds for this mode as for shared dynamics.
    if ((rc = pinst_alloc_trial(h, true))) return rc;
    if ((rc = pinst_upload_dynamics(h, p, h->Ad2, h->Bd2, h->Qd2, h->Rd2, h->QNd2))) return rc;
    int not_pd = 0, grown = 0;
    if ((rc = pinst_try(h, h->Ad2, h->Bd2, h->Qd2, h->Rd2, h->QNd2, h->rho_d, nullptr, &not_pd, &grown))) return rc;
    if (not_pd) return fail(ADMM_ERR_NUMERIC, "problem update refused: R + rho I + B'PB is not positive definite for " + std::to_string(not_pd) + " QP(s)");
    if (grown && h->auto_segments)
      return fail(ADMM_ERR_NUMERIC, "problem update refused: the new dynamics make the segment transfer matrices of " + std::to_string(grown) +
                                        " QP(s) grow beyond the conditioning bound (max entry > 100) with the handle's " +
                                        std::to_string(h->S) + " segments; set up a new handle or use options.segments = 1");
    if ((rc = ensure_w(h))) return rc;           // w of the last x-update belongs to the old problem data (still in place)
    if ((rc = ensure_zy(h))) return rc;
    h->zy_valid = true;
    h->v_valid = false;
    HIP_TRY(hipStreamSynchronize(h->stream));
    std::swap(h->Ad, h->Ad2); std::swap(h->Bd, h->Bd2); std::swap(h->Kd, h->Kd2); std::swap(h->Sd, h->Sd2);      // commit
    std::swap(h->Qd, h->Qd2); std::swap(h->Rd, h->Rd2); std::swap(h->QNd, h->QNd2);
    h->stage_bounds = p->stage_bounds;
    if ((rc = pinst_upload_bounds(h, p))) return rc;
    if ((rc = upload_transposed(h, p->x0, h->x0, h->n))) return rc;
    if (h->has_q && (rc = upload_transposed(h, p->q, h->q, h->L))) return rc;
    return pinst_segments(h);                    // transfer matrices of the new factor
  }
  admm::Factor f;
  std::string err;
  if ((rc = admm::factorise(*p, h->opt.rho, h->S, f, err, h->mfma_mode, !h->scan_gemv, h->ts_n))) return fail(rc, err);
  if (f.recB.size() != h->fac.recB.size() || f.scanWp.size() != h->fac.scanWp.size())
    return fail(ADMM_ERR_NUMERIC, "internal: record sizes changed on refactor");
  if (h->auto_segments && h->S > 1 && scan_growth(f) > SCAN_GROWTH_MAX)
    return fail(ADMM_ERR_NUMERIC, "problem update refused: the new dynamics make the segment transfer matrices of the handle's " +
                                      std::to_string(h->S) + " segments grow beyond the conditioning bound (max |W| > 100); "
                                      "set up a new handle (fewer segments are chosen then) or give options.segments");
  if ((rc = ensure_w(h))) return rc;           // w of the last x-update belongs to the old problem data
  if ((rc = ensure_zy(h))) return rc;          // the state is kept as the (z, y) pair it was under the old box
  h->zy_valid = true;
  h->v_valid = false;
  HIP_TRY(hipStreamSynchronize(h->stream));    // kernels of the old records are done before they change
  h->spec.clear();                             // background factorisations read the problem copy that changes now
  h->spec_stale.clear();
  keep_shared(h, p);
  h->fac = std::move(f);
  if ((rc = upload_factor(h))) return rc;
  if ((rc = upload_bounds(h, p))) return rc;
  if ((rc = upload_transposed(h, p->x0, h->x0, h->n))) return rc;
  if (h->has_q && (rc = upload_transposed(h, p->q, h->q, h->L))) return rc;
  destroy_graph(h);
  return ADMM_OK;
}


}  // extern "C"

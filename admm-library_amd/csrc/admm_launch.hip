// admm_launch.hip -- solver runtime: kernel launches, the time-shard exchange, iteration forms and their schedule, graph capture (admm_runtime.hpp)
#include "admm_runtime.hpp"

namespace admm {
namespace rt {

thread_local std::string g_err;
thread_local std::string g_warn;



// The (n, m)-templated kernels live in admm_dims_g*.hip (compiled in parallel); this file only
// fills the launch descriptor and asks each group in turn.
admm::XLaunch xlaunch_of(const admm_handle* h) {
  admm::XLaunch l{};
  l.stream = h->stream;
  l.n = h->n; l.m = h->m; l.S = h->S; l.pitch = h->pitch; l.batch = h->batch; l.xfree = h->xfree ? h->xfree_mode : 0;
  l.has_q = h->has_q;
  l.has_soc = h->has_soc;
  l.rho = h->opt.rho; l.alpha = h->opt.alpha;
  l.z = h->z; l.y = h->y; l.q = h->q; l.v = h->v; l.w = h->w;
  l.recB = h->recB; l.recF = h->recF; l.recS = h->recS; l.seg_start = h->seg_start;
  l.recFE = h->recFE; l.recBE = h->recBE; l.mvec = h->mvec;
  l.dbuf = h->dbuf; l.tseg = h->tseg; l.eseg = h->eseg; l.tin = h->tin; l.xin = h->xin; l.part = h->part;
  l.x0 = h->x0;
  if (h->ts_n) {        // time shard: the kernels see this rank's segments only (per-segment arrays start at its first one)
    const size_t o = (size_t)h->ts_s0 * h->n * h->pitch;
    l.S = h->ts_sl;
    l.seg_start = h->seg_start + h->ts_s0;
    if (h->ts_n == 1) { l.tseg = h->tseg + o; l.eseg = h->eseg + o; }     // (several ranks: h->tseg / h->eseg are this rank's block already)
    l.tin = h->tin + o; l.xin = h->xin + o;
    l.part = h->part + (size_t)h->ts_s0 * 5 * h->pitch;
  }
  const bool chain = (h->opt.flags & ADMM_FLAG_SCAN_CHAIN) != 0;     // the chain scan writes slab 0 only
  l.nsplit = chain ? 1 : h->scan_split;
  l.split_stride = (size_t)h->fac.scanM * h->pitch;
  l.recMF = h->recMF; l.recMB = h->recMB;
  l.mfma_mode = h->mfma_on ? h->mfma_mode : 0;
  if (h->mfma_refine && h->recMF64) { l.recMF = h->recMF64; l.recMB = h->recMB64; l.mfma_mode = 2; }
  return l;
}

admm::PLaunch plaunch_of(const admm_handle* h) {
  admm::PLaunch l{};
  l.stream = h->stream;
  l.n = h->n; l.m = h->m; l.N = h->N; l.pitch = h->pitch; l.batch = h->batch;
  l.has_q = h->has_q; l.pbounds = h->pbounds;
  l.alpha = h->opt.alpha;
  l.rhov = h->rho_d; l.todo = nullptr;
  l.S = h->S; l.seg_start = h->seg_start; l.Omd = h->Omd; l.Psd = h->Psd; l.Segd = h->Segd;
  l.tseg = h->tseg; l.eseg = h->eseg; l.tin = h->tin; l.xin = h->xin; l.grow = h->pgrow;
  l.rows = h->pi_rows;
  l.rows_factor = h->pi_rows_factor;
  l.has_soc = h->has_soc; l.ub = h->ub;
  l.Ad = h->Ad; l.Bd = h->Bd; l.Q = h->Qd; l.R = h->Rd; l.QN = h->QNd; l.Kd = h->Kd; l.Sd = h->Sd; l.fail = h->pfail;
  l.qflag = nullptr;
  l.lo = h->pbounds ? h->lod : h->lo; l.hi = h->pbounds ? h->hid : h->hi;
  l.loT = h->lodT; l.hiT = h->hidT;
  l.z = h->z; l.y = h->y; l.q = h->q; l.x0 = h->x0; l.v = h->v; l.w = h->w; l.dbuf = h->dbuf; l.part = h->part;
  return l;
}

int launch_p(admm_handle* h, admm::PKernel k, bool vform, bool resid) {
  admm::PLaunch l = plaunch_of(h);
  l.vform = vform; l.resid = resid;
  if (!admm::launch_pinst(l, k, false)) return fail(ADMM_ERR_UNSUPPORTED, "no per-instance kernel for this (n, m)");
  return ADMM_OK;
}

bool dispatch_x(const admm::XLaunch& l, admm::XKernel k, bool a, bool b, bool query_only) {
  return admm::launch_group0(l, k, a, b, query_only) || admm::launch_group1(l, k, a, b, query_only) ||
         admm::launch_group2(l, k, a, b, query_only) || admm::launch_group3(l, k, a, b, query_only);
}

bool dims_supported(int n, int m) {
  admm::XLaunch l{};
  l.n = n; l.m = m;
  return dispatch_x(l, admm::XKernel::XB, false, false, /*query_only=*/true);
}

std::string supported_list() {
  return std::string(admm::dims_group0()) + admm::dims_group1() + admm::dims_group2() + admm::dims_group3();
}

int launch_x(admm_handle* h, admm::XKernel k, bool a, bool b) {
  if (h->pinst) {
    switch (k) {
      case admm::XKernel::XB: return launch_p(h, admm::PKernel::XB, a, false);
      case admm::XKernel::XF: return launch_p(h, admm::PKernel::XF, false, false);
      case admm::XKernel::XFZ: return launch_p(h, admm::PKernel::XFZ, a, b);
      default: return fail(ADMM_ERR_UNSUPPORTED, "kernel form not available with per-instance dynamics");
    }
  }
  const admm::XLaunch l = xlaunch_of(h);
  // MFMA form: the alternating pair (fp64 records) or the plain path's v-form kernels (fp32 records); every other
  // kernel form -- (z, y)-input first iterations, read-out -- stays on the one-lane fp64 kernels (same arrays)
  if (l.mfma_mode) {
    const bool vform_ok = (k == admm::XKernel::XFZE || k == admm::XKernel::XBZE) || a;
    if (vform_ok && admm::launch_mfma(l, k, b, false)) return ADMM_OK;
  }
  if (!dispatch_x(l, k, a, b, false)) return fail(ADMM_ERR_UNSUPPORTED, "no x-update kernel for this (n, m)");
  return ADMM_OK;
}


static_assert(admm::SCAN_KALIGN == admm::SCAN_U, "host range alignment must match the kernel's batch");

admm::FinArgs fin_args(const admm_handle* h, int it, int nchunks) {
  admm::FinArgs fa{};
  fa.part = h->part; fa.resid = h->resid; fa.status = h->status; fa.iters = h->iters; fa.nconv = h->nconv;
  fa.rho = h->opt.rho; fa.eps_abs = h->opt.eps_abs; fa.eps_rel = h->opt.eps_rel; fa.sqrtL = std::sqrt((double)h->L);
  fa.nchunks = nchunks; fa.batch = h->batch; fa.it = it;
  fa.rhov = h->pinst ? h->rho_d : nullptr;
  if (h->mixed_phase1 && it > 0) {              // fp32 phase of a MIXED solve: raised tolerances, scratch status
    fa.eps_abs = std::max(fa.eps_abs, 1e-4);
    fa.eps_rel = std::max(fa.eps_rel, 1e-4);
    fa.status = h->status1;
    fa.iters = h->iters1;
  }
  return fa;
}

// Time-sharded handles: complete the per-segment arrays the next launch reads -- every rank has written the rows of its own
// segments -- with an all-gather through the caller's transport (include/admm_hip.h).
int ts_allgather(admm_handle* h, double* base, size_t count_per_rank) {
  if (!h->ts_n || h->ts_n == 1) return ADMM_OK;
  const int rc = h->ts_fn(h->ts_ctx, (void*)h->stream, ADMM_EXCHANGE_ALLGATHER, base, (int64_t)count_per_rank);
  if (rc) return fail(ADMM_ERR_HIP, "time-sharded handle: the exchange callback failed (" + std::to_string(rc) + ")");
  return ADMM_OK;
}
int ts_exchange_summaries(admm_handle* h) {       // before a segment scan: tseg | eseg (or mseg | ebseg: same slots) of every segment
  // ONE all-gather: the scan's input rows are laid out rank by rank (admm_factor.hpp), each rank's block = 2 n rows per segment
  return ts_allgather(h, h->scan_in, (size_t)2 * h->ts_sl * h->n * h->pitch);
}
int ts_exchange_partials(admm_handle* h) {        // before a finalise: the residual partial sums of every segment
  return ts_allgather(h, h->part, (size_t)h->ts_sl * 5 * h->pitch);
}

// forward_form: the scan of the forward-elimination form (matrix WB).
// with_finalize: one extra row of workgroups finalises the previous iteration's residual partials
// (S chunks, it = 0: no stopping rule -- checked iterations of admm_solve finalise standalone).
int launch_xscan_mfma(admm_handle* h, bool forward_form, bool with_finalize) {
  if (h->ts_n) {
    int rc;
    if ((rc = ts_exchange_summaries(h))) return rc;
    if (with_finalize && (rc = ts_exchange_partials(h))) return rc;
  }
  if (h->scan_gemv) {                           // a handful of QPs: matrix-vector form (admm_kernels.hpp)
    const int M = h->fac.scanM;
    dim3 grid((M + 3) / 4, with_finalize ? 2 : 1), block(256);
    const double* W = forward_form ? h->scanWBd : h->scanWd;
    const int* rows = forward_form ? h->scan_rowsB : h->scan_rows;
    const admm::FinArgs fa = fin_args(h, 0, h->S);
#define GEMV(NC) hipLaunchKernelGGL((admm::xscan_gemv_kernel<NC>), grid, block, 0, h->stream, W, rows, h->scan_in, h->scan_out, \
                                    M, h->fac.scanK, h->pitch, fa)
    if (h->batch == 1) GEMV(1);
    else if (h->batch == 2) GEMV(2);
    else GEMV(4);
#undef GEMV
    return ADMM_OK;
  }
  const int mtiles = h->fac.scanM / 16, ngroups = mtiles / admm::SCAN_MT;
  dim3 grid(h->pitch / 64, ngroups + (with_finalize ? 1 : 0), h->scan_split), block(256);
  hipLaunchKernelGGL((admm::xscan_mfma_kernel<admm::SCAN_MT>), grid, block, 0, h->stream,
                     forward_form ? h->scanWpB : h->scanWp, h->scan_in, h->scan_out,
                     forward_form ? h->scan_rangeB : h->scan_range, mtiles, ngroups, h->pitch, h->scan_split,
                     (size_t)h->fac.scanM * h->pitch, fin_args(h, 0, h->S));
  return ADMM_OK;
}

int launch_xscan(admm_handle* h) {
  if (h->pinst) {                               // per-QP segment scan (nothing to couple with one segment)
    if (h->S > 1 && !admm::launch_pinst(plaunch_of(h), admm::PKernel::SCAN, false))
      return fail(ADMM_ERR_UNSUPPORTED, "no per-instance kernel for this (n, m)");
    return ADMM_OK;
  }
  if (!(h->opt.flags & ADMM_FLAG_SCAN_CHAIN)) return launch_xscan_mfma(h);
  return launch_x(h, admm::XKernel::XSCAN_CHAIN, false, false);
}



// z = clip(v), y = v - z into the z / y arrays (read-out and mode switches)
int ensure_zy(admm_handle* h) {
  if (h->zy_valid) return ADMM_OK;
  if (h->pbounds) {
    if (h->has_soc) admm::launch_pv_to_zy_soc(h->stream, h->v, h->z, h->y, h->lod, h->hid, h->ub, h->N, h->nb, h->m, h->pitch);
    else admm::launch_pv_to_zy(h->stream, h->v, h->z, h->y, h->lod, h->hid, (size_t)h->L * h->pitch);
    h->zy_valid = true;
    return ADMM_OK;
  }
  // (over the handle's stage window: the whole horizon unless it is a time shard; rows are relative to the window's first)
  const size_t b = win_bias(h), r0 = win_row0(h);
  const int Lw = (int)win_rows(h);
  dim3 grid((h->pitch / 2 + Z_THREADS - 1) / Z_THREADS, (Lw + h->zrows - 1) / h->zrows), block(Z_THREADS);
  if (h->has_soc)
    hipLaunchKernelGGL(admm::v_to_zy_soc_kernel, grid, block, 0, h->stream, (const double*)(h->v + b), h->z + b, h->y + b,
                       h->lo + r0, h->hi + r0, h->ub + h->wk0, Lw, h->zrows, h->pitch, h->nb, h->m);
  else
    hipLaunchKernelGGL(admm::v_to_zy_kernel, grid, block, 0, h->stream, (const double*)(h->v + b), h->z + b, h->y + b,
                       h->lo + r0, h->hi + r0, Lw, h->zrows, h->pitch);
  h->zy_valid = true;
  return ADMM_OK;
}

int launch_z(admm_handle* h, bool resid) {
  dim3 grid((h->pitch / 2 + Z_THREADS - 1) / Z_THREADS, h->zchunks), block(Z_THREADS);
  const bool relax = h->opt.alpha != 1.0;
#define ZL(RS, RX)                                                                                       \
  do {                                                                                                   \
    if (h->has_soc)                                                                                      \
      hipLaunchKernelGGL((admm::zdual_soc_kernel<RS, RX>), grid, block, 0, h->stream, h->w, h->z, h->y,  \
                         h->lo, h->hi, h->ub, h->part, h->opt.alpha, h->L, h->zrows, h->pitch, h->nb, h->m); \
    else                                                                                                 \
      hipLaunchKernelGGL((admm::zdual_kernel<RS, RX>), grid, block, 0, h->stream, h->w, h->z, h->y,      \
                         h->lo, h->hi, h->part, h->opt.alpha, h->L, h->zrows, h->pitch);                 \
  } while (0)
  if (resid) {
    if (relax) ZL(true, true); else ZL(true, false);
  } else {
    if (relax) ZL(false, true); else ZL(false, false);
  }
#undef ZL
  return ADMM_OK;
}

// nchunks = zchunks after the standalone z kernel, S after the fused xfz kernel
int launch_finalize(admm_handle* h, int it, int nchunks) {
  if (h->ts_n) {
    int rc;
    if ((rc = ts_exchange_partials(h))) return rc;
  }
  dim3 grid(h->pitch / admm::FIN_COLS), block(admm::FIN_COLS * admm::FIN_GROUPS);
  hipLaunchKernelGGL(admm::resid_finalize_kernel, grid, block, 0, h->stream, fin_args(h, it, nchunks), h->pitch);
  return ADMM_OK;
}

// the deferred finalise of the last alternating iteration, when no scan launch will carry it
int flush_finalize(admm_handle* h, int it) {
  if (!h->fin_pending) return ADMM_OK;
  h->fin_pending = false;
  return launch_finalize(h, it, h->S);
}




// The form of the next iteration when `remaining` iterations (this one included) are still to be
// enqueued before control returns to the caller.  A call must never return after IT_BWD (w of that
// iteration cannot be rebuilt), so a backward iteration is started only if an even number remains.
IterForm next_form(const admm_handle* h, int remaining) {
  if (!h->alt || !h->v_valid) return IT_PLAIN;
  if (h->alt_state == admm_handle::ALT_BWD) return IT_FWD;
  if (h->alt_state == admm_handle::ALT_FWD && remaining % 2 == 0) return IT_BWD;
  return remaining >= 2 ? IT_FWD_START : IT_PLAIN;
}

// fin_prev: the previous iteration evaluated residuals and left their finalise to this scan launch
int enqueue_form(admm_handle* h, IterForm f, bool resid, bool fin_prev) {
  int rc;
  if (f == IT_FWD_START && (rc = launch_xb(h, true))) return rc;
  if ((rc = launch_xscan_mfma(h, f == IT_BWD, fin_prev))) return rc;
  return launch_x(h, f == IT_BWD ? admm::XKernel::XBZE : admm::XKernel::XFZE, false, resid);
}

void after_form(admm_handle* h, IterForm f) {
  h->v_valid = true; h->zy_valid = false; h->w_stale = true;
  h->alt_state = f == IT_PLAIN ? admm_handle::ALT_NONE : (f == IT_BWD ? admm_handle::ALT_BWD : admm_handle::ALT_FWD);
}

// One full iteration on the stream: x-update + z/dual (+ residual partials).
// Fused path: state in = h->v if use_v else h->z / h->y; state out = h->v.
// Unfused path: state in and out = h->z / h->y (caller has made them valid).
// Pure enqueue: the caller updates v_valid / zy_valid / w_stale (graph capture replays this).
int enqueue_iteration(admm_handle* h, bool resid, bool use_v) {
  int rc;
  if (fused(h)) {
    if ((rc = launch_xb(h, use_v))) return rc;
    if ((rc = launch_xscan(h))) return rc;
    return launch_xfz(h, resid, use_v);
  }
  if ((rc = launch_xb(h, false))) return rc;
  if ((rc = launch_xscan(h))) return rc;
  if ((rc = launch_xf(h))) return rc;
  return launch_z(h, resid);
}

// bookkeeping after `count` enqueued iterations
void after_iterations(admm_handle* h, int count) {
  if (count <= 0) return;
  h->alt_state = admm_handle::ALT_NONE;
  if (fused(h)) { h->v_valid = true; h->zy_valid = false; h->w_stale = true; }
  else          { h->zy_valid = true; h->v_valid = false; h->w_stale = false; }
}


// w of the last x-update, if the fused path skipped storing it
int ensure_w(admm_handle* h) {
  if (!h->w_stale) return ADMM_OK;
  int rc = launch_xf(h);
  if (rc) return rc;
  h->w_stale = false;
  return ADMM_OK;
}

int step_x(admm_handle* h) {
  int rc;
  h->alt_state = admm_handle::ALT_NONE;     // xb and the scan overwrite what a fused elimination left
  if ((rc = launch_xb(h, h->v_valid))) return rc;
  if ((rc = launch_xscan(h))) return rc;
  if ((rc = launch_xf(h))) return rc;
  return ADMM_OK;
}


int capture_iterations(admm_handle* h) {
  destroy_graph(h);
  for (int v = 0; v < (h->alt ? 16 : 2); ++v) {
    if (v == 2 || v == 3) continue;                              // (unused slots: form 0 is the plain iteration)
    HIP_TRY(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    const bool res = v < 2 ? (v & 1) != 0 : (v & 2) != 0;
    int rc = v < 2 ? enqueue_iteration(h, res, /*use_v=*/true)   // steady state of the fused path
                   : enqueue_form(h, (IterForm)(v >> 2), res, (v & 1) != 0);
    if (!rc && res && v < 2) rc = launch_finalize(h, 0, chunks_of_iteration(h));
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(h->stream, &g);
    if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
    if (e != hipSuccess) return fail(ADMM_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    h->graph[v] = g;
    HIP_TRY(hipGraphInstantiate(&h->graph_exec[v], h->graph[v], nullptr, nullptr, 0));
  }
  return ADMM_OK;
}


// Enqueue iteration number `it` (1-based within the caller's loop).  The graphs hold the
// steady-state kernel forms; an iteration whose input state is not in that form (first
// fused iteration after setup / set_state / an unfused step) is launched directly.
// `remaining` = iterations (this one included) the caller still enqueues before it returns:
// it selects the iteration form (next_form).  it_number > 0: a checked iteration of admm_solve
// (residuals + finalise with that iteration number, launched directly).
// next_plain: another iteration follows in this call and evaluates no residuals (admm_run / admm_solve_step know).
int enqueue_one(admm_handle* h, bool resid, bool use_graph, int remaining, int it_number, bool next_plain) {
  int rc;
  h->xfree_mode = 1;                       // (never inherited: an error path of admm_profile could have left it set)
  const bool steady = fused(h) ? h->v_valid : h->zy_valid;
  if (!fused(h) && !h->zy_valid && (rc = ensure_zy(h))) return rc;
  const IterForm form = next_form(h, remaining);
  if (form != IT_PLAIN) {
    const bool fin_prev = h->fin_pending;
    if (use_graph) {
      HIP_TRY(hipGraphLaunch(h->graph_exec[4 * (int)form + (resid ? 2 : 0) + (fin_prev ? 1 : 0)], h->stream));
    } else {
      // XFREE = 2: where the state rows are unbounded everywhere, an iteration without residuals or
      // relaxation neither reads their v nor -- if its successor in this call is of the same kind AND a fused alternating
      // kernel (not a start form: xb_kernel reads all of v) -- writes it.  The last iteration of a call always writes.
      const bool next_alternates = form == IT_BWD || (remaining - 1 >= 1 && (remaining - 1) % 2 == 0);
      static const bool no_skip_store = std::getenv("ADMM_NO_SKIPV_STORE") != nullptr;
      h->xfree_mode = (next_plain && !resid && next_alternates && h->opt.alpha == 1.0 && !no_skip_store) ? 2 : 1;
      rc = enqueue_form(h, form, resid, fin_prev);
      h->xfree_mode = 1;
      if (rc) return rc;
    }
    after_form(h, form);
    h->fin_pending = resid;                 // carried by the next scan launch, or flushed by the caller
    if (resid) h->resid_valid = true;
    if (it_number > 0) return flush_finalize(h, it_number);
    return ADMM_OK;
  }
  if ((rc = flush_finalize(h))) return rc;
  if (it_number > 0) {
    if ((rc = enqueue_iteration(h, true, fused(h) && h->v_valid))) return rc;
    if ((rc = launch_finalize(h, it_number, chunks_of_iteration(h)))) return rc;
  } else if (use_graph && steady) {
    HIP_TRY(hipGraphLaunch(h->graph_exec[resid ? 1 : 0], h->stream));
  } else {
    rc = enqueue_iteration(h, resid, fused(h) && h->v_valid);
    if (!rc && resid) rc = launch_finalize(h, 0, chunks_of_iteration(h));
    if (rc) return rc;
  }
  after_iterations(h, 1);
  if (resid) h->resid_valid = true;
  return ADMM_OK;
}


}  // namespace rt
}  // namespace admm

// admm_api.hip -- C ABI of libadmm_hip.so (include/admm_hip.h) over the HIP
// kernels of admm_kernels.hpp.  Solver runtime: device buffer ownership,
// iteration driver, stopping logic.  No CPU fallback: without a HIP device
// every compute entry point returns ADMM_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/admm_hip.h"
#include "admm_dispatch.hpp"
#include "admm_factor.hpp"
#include "admm_kernels.hpp"

namespace {

thread_local std::string g_err;
thread_local std::string g_warn;     // admm_last_warning(): a call succeeded but changed the kernels a handle runs

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return fail(ADMM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_) + " (" +     \
                                    __FILE__ + ":" + std::to_string(__LINE__) + ")");          \
  } while (0)

}  // namespace

// A factorisation for a rho the adaptive rule may ask for next, computed on a background thread while the GPU iterates
// (or the factor of the previous rho, kept).  See spec_start().
struct SpecFactor {
  double rho = 0.0;
  admm::Factor f;
  std::string err;
  int rc = 0;
  std::thread th;
  ~SpecFactor() { if (th.joinable()) th.join(); }
};

struct admm_handle {
  int N = 0, n = 0, m = 0, nb = 0, batch = 0, pitch = 0, L = 0;
  int S = 0, zrows = 0, zchunks = 0;
  int scan_split = 1;            // split-K factor of the MFMA scan (small batches)
  int device = 0;
  int num_cus = 256;             // hipDeviceProp_t::multiProcessorCount of the handle's device
  bool xfree = false;            // every state row is unbounded at every stage (XFREE kernel forms, see xfze_kernel)
  int xfree_mode = 1;            // 2 while enqueue_one launches an iteration whose successor will not read those rows' v
  bool auto_segments = false;    // the segment count was chosen by admm_setup (and is guarded by scan_growth)
  bool has_q = false;
  bool has_soc = false;          // some stage has a finite thrust-magnitude bound
  admm_options opt{};
  admm::Factor fac;
  // host copy of the shared problem data (the caller's pointers are never kept): admm_set_rho refactors from it
  std::vector<double> pA, pB, pQ, pR, pQN, plo, phi, pun;
  int time_varying = 0, stage_bounds = 0;
  int rho_updates = 0;
  std::vector<std::unique_ptr<SpecFactor>> spec;        // candidates of the adaptive rule (rho tau, rho / tau)
  std::vector<std::unique_ptr<SpecFactor>> spec_stale;  // no longer candidates; their threads are joined lazily
  int spec_hits = 0, spec_misses = 0;
  // time-sharded handle (admm_setup_timeshard): this rank runs segments [ts_s0, ts_s0 + ts_sl) of the S the horizon is cut into
  int ts_n = 0, ts_rank = 0, ts_s0 = 0, ts_sl = 0;      // ts_n = 0: an ordinary handle
  admm_exchange_fn ts_fn = nullptr;
  void* ts_ctx = nullptr;
  bool solve_active = false;     // between admm_solve_begin and admm_solve_end: only then are candidate factors kept / started
  // ADMM_FLAG_HISTORY: one record per stopping test of the last admm_solve
  struct HistoryEntry { int32_t it, nconv; double max_r, max_s, rho; };
  std::vector<HistoryEntry> history;
  int solve_it = 0, solve_nconv = 0;     // admm_solve_begin / _step / _end state
  std::chrono::steady_clock::time_point solve_t0;
  hipStream_t stream = nullptr;
  // batch-minor state and work buffers
  double *w = nullptr, *z = nullptr, *y = nullptr, *v = nullptr, *q = nullptr, *x0 = nullptr;
  double *dbuf = nullptr, *tseg = nullptr, *eseg = nullptr, *tin = nullptr, *xin = nullptr;
  double *part = nullptr, *resid = nullptr, *lo = nullptr, *hi = nullptr, *ub = nullptr;
  double *recB = nullptr, *recF = nullptr, *recS = nullptr;
  double *scan_in = nullptr, *scan_out = nullptr, *scanWp = nullptr;   // tseg|x0|eseg and t_in|x_in live inside these
  int* scan_range = nullptr;
  // batches of up to SCAN_GEMV_MAXCOLS QPs: the scan as a matrix-vector product per column (xscan_gemv_kernel) on the
  // dense row-major matrices, with each row's non-zero column range
  bool scan_gemv = false;
  double *scanWd = nullptr, *scanWBd = nullptr;
  int *scan_rows = nullptr, *scan_rowsB = nullptr;
  // alternating-direction iteration
  double *recFE = nullptr, *recBE = nullptr, *mvec = nullptr, *scanWpB = nullptr;
  int* scan_rangeB = nullptr;
  // MFMA form of the fused kernels: fragment records, mode (0 = not in use, 1 mixed, 2 fp64), and whether
  // launch_x currently routes to it (the fp64 refinement phase of a MIXED solve turns it off)
  unsigned char *recMF = nullptr, *recMB = nullptr;
  unsigned char *recMF64 = nullptr, *recMB64 = nullptr;    // MIXED only: all-fp64 records of the refinement phase
  int mfma_mode = 0;
  bool mfma_on = false;
  bool mfma_refine = false;      // MIXED, refinement phase: the fp64 MFMA kernels on recMF64 / recMB64
  bool alt_allowed = false;      // alternation permitted by the options / compiled kernels (before the precision mode)
  bool alt_requested = false;    // ... whether or not the forward-elimination form passed its host check (admm_get_path)
  // MIXED solve: phase 1 (fp32) checks the stopping rule with raised tolerances on scratch status arrays
  bool mixed_phase1 = false;
  int mixed_iters = 0;
  int *status1 = nullptr, *iters1 = nullptr;
  // per-instance dynamics (csrc/admm_pinst.hpp): device-side factor, operands per QP in HBM
  bool pinst = false, pbounds = false;
  double *Ad = nullptr, *Bd = nullptr, *Kd = nullptr, *Sd = nullptr, *lod = nullptr, *hid = nullptr;
  double *Qd = nullptr, *Rd = nullptr, *QNd = nullptr;
  int* pfail = nullptr;
  // TRIAL buffers of the per-instance path (allocated on first use): a change of rho or of the problem data is factorised
  // into these first and only then committed by swapping pointers, so that a refused change leaves the handle untouched
  double *Ad2 = nullptr, *Bd2 = nullptr, *Kd2 = nullptr, *Sd2 = nullptr, *Qd2 = nullptr, *Rd2 = nullptr, *QNd2 = nullptr;
  double *rho2_d = nullptr;      // [pitch] candidate rho (admm_set_rho) / the rho being left (per-QP adaptive rule)
  int *qflag_d = nullptr, *nveto_d = nullptr;      // [pitch] per-QP verdict of a trial factorisation; [1] refused changes
  // segments in time of the per-instance path (S > 1; csrc/admm_pinst.hpp, pseg_kernel): per-QP transfer matrices
  double *Omd = nullptr, *Psd = nullptr, *Segd = nullptr;
  int* pgrow = nullptr;
  bool pi_rows = false;          // small batches: sweeps with a QP's rows spread over lanes (csrc/admm_pinst_rows.hpp)
  // per-QP rho (every QP of a per-instance problem has its own factor, so the adaptive rule runs QP by QP on the device)
  double *rho_d = nullptr, *cscale_d = nullptr;     // [pitch]
  int *nupd_d = nullptr, *todo_d = nullptr, *nchanged_d = nullptr;
  size_t stage_rows = 0;         // rows the staging buffer holds (L, or N n^2 for the per-instance upload of A)
  bool alt = false;              // the alternating kernels exist for this problem and are enabled
  // what the last kernel left behind for the next x-update:
  //   ALT_NONE  nothing (the next iteration starts with xb_kernel)
  //   ALT_FWD   xfze ran: db rows | mseg | ebseg -> next: scan (WB) + xbze
  //   ALT_BWD   xbze ran: dbuf | tseg | eseg    -> next: scan (W)  + xfze   (w of that iteration cannot be
  //             re-materialised, so no API call ever returns in this state)
  enum { ALT_NONE = 0, ALT_FWD = 1, ALT_BWD = 2 };
  int alt_state = ALT_NONE;
  int *seg_start = nullptr, *status = nullptr, *iters = nullptr, *nconv = nullptr;
  double* stage = nullptr;      // QP-major staging buffer, L * batch
  int* h_nconv = nullptr;       // pinned
  // pinned bounce buffers of large host-to-device uploads (allocated on first use; upload_h2d)
  unsigned char* pin[2] = {nullptr, nullptr};
  hipEvent_t pin_ev[2] = {nullptr, nullptr};
  int iters_run = 0;
  bool resid_valid = false;
  // A residual-evaluating alternating iteration leaves its finalise to the NEXT scan launch (finalise
  // role of xscan_mfma_kernel); flush_finalize() runs it standalone when no scan follows.
  bool fin_pending = false;
  bool w_stale = false;         // fused iterations do not store w; admm_get re-materialises it
  // State form: the fused path keeps v = z + y only; z, y are rebuilt on demand.
  bool v_valid = false;         // h->v holds the current state
  bool zy_valid = true;         // h->z, h->y hold the current state
  // captured iterations, replayed by admm_run / admm_solve:
  //   [r]: plain iteration, r = 1 with residuals + finalise (it = 0);
  //   [4 t + 2 r + p] (t = IT_FWD_START .. IT_BWD): alternating forms, p = 1 if the scan launch also
  //   finalises the previous iteration's residuals
  hipGraph_t graph[16] = {};
  hipGraphExec_t graph_exec[16] = {};
};

namespace {

using admm::Z_THREADS;

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
  l.has_soc = h->has_soc; l.ub = h->ub;
  l.Ad = h->Ad; l.Bd = h->Bd; l.Q = h->Qd; l.R = h->Rd; l.QN = h->QNd; l.Kd = h->Kd; l.Sd = h->Sd; l.fail = h->pfail;
  l.qflag = nullptr;
  l.lo = h->pbounds ? h->lod : h->lo; l.hi = h->pbounds ? h->hid : h->hi;
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

// vform: read the state from h->v (z = clip(v), y = v - z rebuilt in registers)
int launch_xb(admm_handle* h, bool vform) { return launch_x(h, admm::XKernel::XB, vform, false); }

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
int launch_xscan_mfma(admm_handle* h, bool forward_form = false, bool with_finalize = false) {
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

int launch_xf(admm_handle* h) { return launch_x(h, admm::XKernel::XF, false, false); }

// fused forward rollout + z/dual (+ residual partials per segment); writes v+ into h->v.
// vin: previous state read from h->v, otherwise from h->z / h->y.
int launch_xfz(admm_handle* h, bool resid, bool vin) { return launch_x(h, admm::XKernel::XFZ, vin, resid); }

// z = clip(v), y = v - z into the z / y arrays (read-out and mode switches)
int ensure_zy(admm_handle* h) {
  if (h->zy_valid) return ADMM_OK;
  if (h->pbounds) {
    if (h->has_soc) admm::launch_pv_to_zy_soc(h->stream, h->v, h->z, h->y, h->lod, h->hid, h->ub, h->N, h->nb, h->m, h->pitch);
    else admm::launch_pv_to_zy(h->stream, h->v, h->z, h->y, h->lod, h->hid, (size_t)h->L * h->pitch);
    h->zy_valid = true;
    return ADMM_OK;
  }
  dim3 grid((h->pitch / 2 + Z_THREADS - 1) / Z_THREADS, h->zchunks), block(Z_THREADS);
  if (h->has_soc)
    hipLaunchKernelGGL(admm::v_to_zy_soc_kernel, grid, block, 0, h->stream, (const double*)h->v, h->z, h->y,
                       h->lo, h->hi, h->ub, h->L, h->zrows, h->pitch, h->nb, h->m);
  else
    hipLaunchKernelGGL(admm::v_to_zy_kernel, grid, block, 0, h->stream, (const double*)h->v, h->z, h->y,
                       h->lo, h->hi, h->L, h->zrows, h->pitch);
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
int flush_finalize(admm_handle* h, int it = 0) {
  if (!h->fin_pending) return ADMM_OK;
  h->fin_pending = false;
  return launch_finalize(h, it, h->S);
}

bool fused(const admm_handle* h) { return !(h->opt.flags & ADMM_FLAG_UNFUSED); }

// Iteration forms.  IT_PLAIN is always available; the others need h->alt and
// the state in v-form.
enum IterForm {
  IT_PLAIN = 0,     // xb + scan + xfz                    leaves ALT_NONE
  IT_FWD_START = 1, // xb + scan + xfze                   leaves ALT_FWD
  IT_FWD = 2,       // scan + xfze        (needs ALT_BWD) leaves ALT_FWD
  IT_BWD = 3        // scan (WB) + xbze   (needs ALT_FWD) leaves ALT_BWD
};

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

int chunks_of_iteration(const admm_handle* h) { return fused(h) ? h->S : h->zchunks; }

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

// Host threads for the O(problem size) host loops of the API (finiteness checks, copies into pinned memory): at most 16, never
// more than the work is worth (one per 4 MB).  fn(begin, end) over a partition of [0, count); results are combined by the caller.
template <class F>
void host_parallel(size_t count, size_t bytes_per_item, F&& fn) {
  size_t nt = std::min<size_t>(std::min<size_t>(16, std::max(1u, std::thread::hardware_concurrency())),
                               count * bytes_per_item / ((size_t)4 << 20));
  if (nt <= 1) { fn((size_t)0, count); return; }
  std::vector<std::thread> th;
  size_t started = 1;
  try {
    for (size_t t = 1; t < nt; ++t) {
      th.emplace_back([&fn, t, nt, count] { fn(count * t / nt, count * (t + 1) / nt); });
      started = t + 1;
    }
  } catch (...) {                                // no more threads to be had: the remaining slices run here
  }
  fn((size_t)0, count / nt);
  for (size_t t = started; t < nt; ++t) fn(count * t / nt, count * (t + 1) / nt);
  for (auto& x : th) x.join();
}

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

// QP-major host array (batch x rows) -> batch-minor device array (rows x pitch)
int upload_transposed(admm_handle* h, const double* src, double* dst, int rows) {
  if ((size_t)rows > h->stage_rows) return fail(ADMM_ERR_INVALID, "internal: staging buffer too small");
  int rc_up;
  if ((rc_up = upload_h2d(h, h->stage, src, sizeof(double) * (size_t)rows * h->batch))) return rc_up;
  dim3 grid((rows + admm::T_TILE - 1) / admm::T_TILE, (h->pitch + admm::T_TILE - 1) / admm::T_TILE), block(admm::T_TILE * 8);
  hipLaunchKernelGGL(admm::to_batch_minor_kernel, grid, block, 0, h->stream, h->stage, dst, h->batch, rows, h->pitch);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  return ADMM_OK;
}

int download_transposed(admm_handle* h, const double* src, double* dst, int rows) {
  dim3 grid((rows + admm::T_TILE - 1) / admm::T_TILE, (h->pitch + admm::T_TILE - 1) / admm::T_TILE), block(admm::T_TILE * 8);
  hipLaunchKernelGGL(admm::from_batch_minor_kernel, grid, block, 0, h->stream, src, h->stage, h->batch, rows, h->pitch);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(dst, h->stage, sizeof(double) * (size_t)rows * h->batch, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return ADMM_OK;
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
constexpr double ALT_GATE = 5e-12;      // the bound build_alternating applies (csrc/admm_factor.cpp)
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
constexpr double SCAN_GROWTH_MAX = 100.0;
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
  double** bufs[] = {&h->w, &h->z, &h->y, &h->v, &h->q, &h->dbuf, &h->scan_in, &h->scan_out, &h->scanWp,
                     &h->part, &h->resid, &h->lo, &h->hi, &h->ub, &h->recB, &h->recF, &h->recS, &h->stage,
                     &h->recFE, &h->recBE, &h->mvec, &h->scanWpB};
  for (auto b : bufs)
    if (*b) { (void)hipFree(*b); *b = nullptr; }
  {
    double** pb[] = {&h->Ad, &h->Bd, &h->Kd, &h->Sd, &h->lod, &h->hid, &h->Qd, &h->Rd, &h->QNd,
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

template <typename T>
int dalloc(T** p, size_t count) {
  hipError_t e = hipMalloc((void**)p, sizeof(T) * (count ? count : 1));
  if (e != hipSuccess) return fail(ADMM_ERR_ALLOC, std::string("hipMalloc failed: ") + hipGetErrorString(e));
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
  HIP_TRY(hipMemcpy(h->lo, lo.data(), sizeof(double) * L, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->hi, hi.data(), sizeof(double) * L, hipMemcpyHostToDevice));
  std::vector<double> ub(h->N, INFINITY);
  if (p->unorm)
    for (int k = 0; k < h->N; ++k) ub[k] = p->unorm[p->stage_bounds ? k : 0];
  HIP_TRY(hipMemcpy(h->ub, ub.data(), sizeof(double) * h->N, hipMemcpyHostToDevice));
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


// ---- per-instance dynamics ----
// Riccati factorisation of every QP on the device; ADMM_ERR_NUMERIC if some S_k is not positive definite.
// `only_marked`: refactor the QPs marked in todo_d (per-QP adaptive rule); rho comes from rho_d either way.
int pinst_factor(admm_handle* h, bool only_marked = false) {
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
  HIP_TRY(hipMemcpy(Qd, Q.data(), sizeof(double) * Q.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(Rd, R.data(), sizeof(double) * R.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(QNd, QN.data(), sizeof(double) * QN.size(), hipMemcpyHostToDevice));
  int rc;
  if ((rc = upload_transposed(h, p->A, Ad, h->N * n * n))) return rc;
  if ((rc = upload_transposed(h, p->B, Bd, h->N * n * m))) return rc;
  return ADMM_OK;
}

// the box (per instance or shared) and the thrust-magnitude bounds
int pinst_upload_bounds(admm_handle* h, const admm_problem* p) {
  int rc;
  if (h->pbounds) {
    if ((rc = upload_transposed(h, p->lo, h->lod, h->L))) return rc;
    if ((rc = upload_transposed(h, p->hi, h->hid, h->L))) return rc;
    std::vector<double> ub(h->N, INFINITY);         // thrust-magnitude bound per stage (shared by the batch)
    if (p->unorm)
      for (int k = 0; k < h->N; ++k) ub[k] = p->unorm[k];
    HIP_TRY(hipMemcpy(h->ub, ub.data(), sizeof(double) * h->N, hipMemcpyHostToDevice));
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
  // Segments in time (csrc/admm_pinst.hpp): one lane sweeps one segment of one QP, so an iteration takes N / S dependent
  // stage round trips instead of N.  Automatic count: enough (64-QP wave, segment) pairs for one wave per SIMD, segments
  // of at least 8 stages, at most 64 (32 from 512 QPs: the scan is S sequential steps per QP); large batches fill the chip
  // alone (S = 1).
  h->auto_segments = o.segments == 0;
  {
    int S = o.segments;
    if (S == 0) {
      const int waves = h->pitch / 64;
      // (measured, N = 1000, n = 6: 64 QPs 2.26 -> 0.27 ms per iteration, 4096 QPs 3.04 -> 1.60 ms; from 8192 QPs the batch
      //  alone reaches the HBM roofline and the segments' extra operands -- Omega_k, Psi_k: +16 % bytes -- only cost)
      S = waves <= 64 ? (4 * h->num_cus) / std::max(1, waves) : 1;
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
    if (h->has_soc) h->pi_rows = false;          // the thrust-magnitude forms exist for the one-lane kernels only
    if (std::getenv("ADMM_PI_LANE_PER_QP")) h->pi_rows = false;
    if (std::getenv("ADMM_PI_ROWS") && !h->has_soc) h->pi_rows = true;
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
  PD(h->Qd, (size_t)n * n); PD(h->Rd, (size_t)m * m); PD(h->QNd, (size_t)n * n);
  PD(h->pfail, 1); PD(h->status, P); PD(h->iters, P); PD(h->nconv, 1);
  PD(h->rho_d, P); PD(h->cscale_d, P); PD(h->nupd_d, P); PD(h->todo_d, P); PD(h->nchanged_d, 1);
  h->stage_rows = std::max(L, (size_t)N * n * n);
  if ((rc = dalloc(&h->stage, h->stage_rows * (size_t)h->batch))) return rc;
#undef PD
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

}  // namespace

extern "C" {

void admm_default_options(admm_options* o) {
  if (!o) return;
  o->rho = 0.1;
  o->alpha = 1.0;
  o->eps_abs = 1e-6;
  o->eps_rel = 1e-6;
  o->max_iter = 4000;
  o->check_interval = 10;
  o->segments = 0;
  o->device = -1;
  o->zrows = 0;
  o->flags = ADMM_FLAG_NONE;
  o->adapt_interval = 0;
  o->adapt_max = 16;
  o->adapt_mu = 10.0;
  o->adapt_tau = 2.0;
  o->precision_mode = ADMM_PRECISION_FP64;
  o->reserved = 0;
}

const char* admm_last_error(void) { return g_err.c_str(); }
const char* admm_last_warning(void) { return g_warn.c_str(); }
int admm_abi_version(void) { return ADMM_HIP_ABI_VERSION; }

int admm_device_count(void) {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) return 0;
  return c;
}

int admm_record_sizes(int32_t n, int32_t m, int32_t* rb, int32_t* rf, int32_t* rs) {
  if (n < 1 || m < 1) return fail(ADMM_ERR_INVALID, "n, m must be positive");
  if (rb) *rb = admm::rec_b_size(n, m);
  if (rf) *rf = admm::rec_f_size(n, m);
  if (rs) *rs = admm::rec_s_size(n);
  return ADMM_OK;
}

int admm_host_factor(const admm_problem* p, double rho, int32_t segments, double* K, double* Sinv,
                     double* recB, double* recF, double* recS, int32_t* seg_start) {
  if (!p) return fail(ADMM_ERR_INVALID, "NULL problem");
  admm::Factor f;
  std::string err;
  int rc = admm::factorise(*p, rho, segments, f, err);
  if (rc) return fail(rc, err);
  if (K) std::memcpy(K, f.K.data(), sizeof(double) * f.K.size());
  if (Sinv) std::memcpy(Sinv, f.Sinv.data(), sizeof(double) * f.Sinv.size());
  if (recB) std::memcpy(recB, f.recB.data(), sizeof(double) * f.recB.size());
  if (recF) std::memcpy(recF, f.recF.data(), sizeof(double) * f.recF.size());
  if (recS) std::memcpy(recS, f.recS.data(), sizeof(double) * f.recS.size());
  if (seg_start) std::memcpy(seg_start, f.seg_start.data(), sizeof(int32_t) * f.seg_start.size());
  return ADMM_OK;
}

int admm_host_scan_matrix(const admm_problem* p, double rho, int32_t segments, double* W, int32_t* M,
                          int32_t* Mt, int32_t* K) {
  if (!p) return fail(ADMM_ERR_INVALID, "NULL problem");
  admm::Factor f;
  std::string err;
  int rc = admm::factorise(*p, rho, segments, f, err);
  if (rc) return fail(rc, err);
  if (M) *M = f.scanM;
  if (Mt) *Mt = f.scanMt;
  if (K) *K = f.scanK;
  if (W) std::memcpy(W, f.scanW.data(), sizeof(double) * f.scanW.size());
  return ADMM_OK;
}

int admm_host_scan_matrices_timeshard(const admm_problem* p, double rho, int32_t segments, int32_t nranks, double* W, double* WB,
                                      int32_t* ok) {
  if (!p || nranks < 1 || segments < 1 || segments % nranks != 0) return fail(ADMM_ERR_INVALID, "need segments >= 1, a multiple of nranks >= 1");
  admm::Factor f;
  std::string err;
  int rc = admm::factorise(*p, rho, segments, f, err, 0, true, nranks);
  if (rc) return fail(rc, err);
  if (W) std::memcpy(W, f.scanW.data(), sizeof(double) * f.scanW.size());
  if (ok) *ok = f.alt_ok ? 1 : 0;
  if (WB && f.alt_ok) std::memcpy(WB, f.scanWB.data(), sizeof(double) * f.scanWB.size());
  return ADMM_OK;
}

int admm_record_sizes_alt(int32_t n, int32_t m, int32_t* rfe, int32_t* rbe) {
  if (n < 1 || m < 1) return fail(ADMM_ERR_INVALID, "n, m must be positive");
  if (rfe) *rfe = admm::rec_fe_size(n, m);
  if (rbe) *rbe = admm::rec_be_size(n, m);
  return ADMM_OK;
}

int admm_host_factor_alt(const admm_problem* p, double rho, int32_t segments, double* recFE, double* recBE,
                         double* WB, int32_t* ok) {
  if (!p) return fail(ADMM_ERR_INVALID, "NULL problem");
  admm::Factor f;
  std::string err;
  int rc = admm::factorise(*p, rho, segments, f, err);
  if (rc) return fail(rc, err);
  if (ok) *ok = f.alt_ok ? 1 : 0;
  if (!f.alt_ok) return ADMM_OK;
  if (recFE) std::memcpy(recFE, f.recFE.data(), sizeof(double) * f.recFE.size());
  if (recBE) std::memcpy(recBE, f.recBE.data(), sizeof(double) * f.recBE.size());
  if (WB) std::memcpy(WB, f.scanWB.data(), sizeof(double) * f.scanWB.size());
  return ADMM_OK;
}

int admm_mfma_record_bytes(int32_t n, int32_t m, int32_t mode, int32_t* fwd, int32_t* bwd) {
  if (!admm::mfma_dims(n, m)) return fail(ADMM_ERR_UNSUPPORTED, "the MFMA form needs 1 <= n <= 12 and 1 <= m <= 8");
  if (mode != 1 && mode != 2) return fail(ADMM_ERR_INVALID, "mode must be ADMM_PRECISION_MIXED or ADMM_PRECISION_FP64_MFMA");
  if (fwd) *fwd = admm::mfma_rec_bytes_fwd(n, m, mode);
  if (bwd) *bwd = admm::mfma_rec_bytes_bwd(n, m, mode);
  return ADMM_OK;
}

int admm_host_factor_mfma(const admm_problem* p, double rho, int32_t segments, int32_t mode, void* recMF,
                          void* recMB, int32_t* alt_ok) {
  if (!p) return fail(ADMM_ERR_INVALID, "NULL problem");
  if (mode != 1 && mode != 2) return fail(ADMM_ERR_INVALID, "mode must be ADMM_PRECISION_MIXED or ADMM_PRECISION_FP64_MFMA");
  admm::Factor f;
  std::string err;
  int rc = admm::factorise(*p, rho, segments, f, err, mode);
  if (rc) return fail(rc, err);
  if (recMF) std::memcpy(recMF, f.recMF.data(), f.recMF.size());
  if (recMB) std::memcpy(recMB, f.recMB.data(), f.recMB.size());
  if (alt_ok) *alt_ok = f.alt_ok ? 1 : 0;
  return ADMM_OK;
}

static int setup_common(admm_handle** out, const admm_problem* p, const admm_options* o_in, int ts_rank, int ts_n,
                        admm_exchange_fn ts_fn, void* ts_ctx) {
  if (!out || !p) return fail(ADMM_ERR_INVALID, "NULL argument");
  *out = nullptr;
  g_warn.clear();
  admm_options o;
  if (o_in) o = *o_in; else admm_default_options(&o);
  int rc;
  if ((rc = validate_options(&o))) return rc;
  if ((rc = validate_problem(p))) return rc;
  if (ts_n) {
    if (ts_n < 1 || ts_rank < 0 || ts_rank >= ts_n) return fail(ADMM_ERR_INVALID, "admm_setup_timeshard: need 0 <= rank < nranks");
    if (ts_n > 1 && !ts_fn) return fail(ADMM_ERR_INVALID, "admm_setup_timeshard: an exchange function is needed with more than one rank");
    if (p->time_varying == 2) return fail(ADMM_ERR_UNSUPPORTED, "time-sharded handles need batch-shared dynamics");
    if (o.flags & (ADMM_FLAG_UNFUSED | ADMM_FLAG_GRAPH | ADMM_FLAG_SCAN_CHAIN))
      return fail(ADMM_ERR_UNSUPPORTED, "time-sharded handles run the fused paths with direct launches (no ADMM_FLAG_UNFUSED / _GRAPH / _SCAN_CHAIN)");
    if (o.segments % ts_n != 0) return fail(ADMM_ERR_INVALID, "admm_setup_timeshard: options.segments (the total) must be a multiple of nranks");
    if (p->N < ts_n) return fail(ADMM_ERR_INVALID, "admm_setup_timeshard: fewer stages than ranks");
  }
  if (!dims_supported(p->n, p->m))
    return fail(ADMM_ERR_UNSUPPORTED, "(n, m) = (" + std::to_string(p->n) + ", " + std::to_string(p->m) +
                                          ") has no compiled kernel; supported: " + supported_list());
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return fail(ADMM_ERR_NO_DEVICE, "no HIP device visible: libadmm_hip has no CPU fallback");
  int dev = o.device;
  if (dev < 0) HIP_TRY(hipGetDevice(&dev));
  if (dev >= ndev) return fail(ADMM_ERR_INVALID, "device ordinal out of range");
  HIP_TRY(hipSetDevice(dev));

  admm_handle* h = new admm_handle();
  h->device = dev;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) h->num_cus = prop.multiProcessorCount;
  }
  h->opt = o;
  h->ts_n = ts_n; h->ts_rank = ts_rank; h->ts_fn = ts_fn; h->ts_ctx = ts_ctx;
  h->N = p->N; h->n = p->n; h->m = p->m; h->nb = p->n + p->m; h->batch = p->batch;
  h->L = p->N * h->nb;
  h->pitch = ((p->batch + 63) / 64) * 64;
  h->has_q = p->q != nullptr;
  h->has_soc = problem_has_soc(p);
  if (p->time_varying == 2) {                    // per-instance dynamics: its own set-up (device factorisation)
    rc = setup_pinst(h, p);
    if (rc) { std::string keep = g_err; release(h); g_err = keep; return rc; }
    *out = h;
    return ADMM_OK;
  }

  // x-update segments: ONE workgroup (256 columns x one segment) per CU -- the grid
  // ceil(pitch / 256) x S should fill the 256 CUs once and not spill into a ragged second round
  // (measured, DESIGN.md §4.6: pitch 4160 with S = 16 is 272 workgroups and runs 1.45x slower per
  // QP than pitch 4096; S = 15 = 255 workgroups restores the rate).  The x kernels run as fast at
  // one wave per SIMD as at two, and the scan's work grows with S^2, so fewer, longer segments win.
  {
    int S = o.segments;
    if (S == 0) {
      const int col_blocks = (h->pitch + admm::XB_THREADS - 1) / admm::XB_THREADS;
      S = h->num_cus / col_blocks;
      // (a lone wave per segment is latency-bound at ~2.2 us per stage: with up to 64 QPs shorter segments
      //  pay -- N = 200, batch 1: 24.9 -> 19.9 us per iteration at S = 50 instead of 25; beyond 64 segments
      //  the host-side factorisation -- 80 ms at S = 125 -- costs more than a solve saves)
      const int per_seg = h->pitch <= 64 ? 4 : 8;
      const int max_by_len = h->N >= 2 * per_seg ? h->N / per_seg : 1;
      if (S > max_by_len) S = max_by_len;
      if (S > 64) S = 64;
      // a segment of the state must stay below the 2 GiB a buffer descriptor can span (with margin)
      const double total = (double)h->N * h->nb * h->pitch * 8.0;
      const int min_by_span = (int)(total / 1.9e9) + 1;
      if (S < min_by_span) S = min_by_span;
    }
    if (ts_n) {               // time shards: the rule above gives the segments of ONE rank's grid; the horizon gets nranks times as many
      if (o.segments == 0) {
        const int max_total = h->N >= 16 ? h->N / 8 : h->N;
        S = std::max(1, std::min(S, max_total / ts_n)) * ts_n;
      }
      if (S > h->N) S = (h->N / ts_n) * ts_n;
    }
    if (S > h->N) S = h->N;
    if (S < 1) S = 1;
    h->S = S;
  }
  // z-kernel chunking: ONE workgroup per CU (256 workgroups), rows per chunk a multiple of 4.
  // Measured on configs[2]: 2000 workgroups 265 us, 504 -> 255 us, 256 -> 245 us,
  // 200 -> 244 us, 360 -> 280 us (a ragged second round), 128 -> 301 us: few long-running
  // workgroups, each streaming consecutive rows, and a grid that fills the CUs exactly once.
  {
    int zr = o.zrows;
    if (zr == 0) {
      const int col_groups = (h->pitch / 2 + Z_THREADS - 1) / Z_THREADS;
      int chunks = (h->num_cus + col_groups - 1) / col_groups;
      if (chunks < 1) chunks = 1;
      zr = (h->L + chunks - 1) / chunks;
      zr = ((zr + 3) / 4) * 4;
      if (zr < 4) zr = 4;
    }
    if (h->has_soc) zr = ((zr + h->nb - 1) / h->nb) * h->nb;   // block-structured kernels: whole blocks per chunk
    h->zrows = zr;
    h->zchunks = (h->L + zr - 1) / zr;
  }

  // precision mode: the MFMA forms exist for a few (n, m), without q / thrust-magnitude bound
  {
    admm::XLaunch lq{};
    lq.n = p->n; lq.m = p->m;
    lq.mfma_mode = o.precision_mode == ADMM_PRECISION_MIXED ? 1 : 2;
    const bool compiled = admm::mfma_dims(p->n, p->m) && admm::launch_mfma(lq, admm::XKernel::XFZE, false, true);
    std::string why;
    if (!compiled) why = "(n, m) has no MFMA instantiation; compiled: " + std::string(admm::dims_mfma());
    else if (h->has_q && !(p->n == 6 && p->m == 3 && h->pitch <= 128 && o.precision_mode != ADMM_PRECISION_MIXED))
      why = "a linear term q is supported by the fp64 MFMA forms of (6, 3) for batches of up to 128 QPs only";
    else if (h->has_soc) why = "a thrust-magnitude bound is not supported by the MFMA forms";
    else if (o.flags & (ADMM_FLAG_UNFUSED | ADMM_FLAG_SCAN_CHAIN)) why = "ADMM_FLAG_UNFUSED / ADMM_FLAG_SCAN_CHAIN exclude the MFMA forms";
    if (o.precision_mode != ADMM_PRECISION_FP64) {
      if (!why.empty()) { release(h); return fail(ADMM_ERR_UNSUPPORTED, "precision_mode " + std::to_string(o.precision_mode) + ": " + why); }
      h->mfma_mode = lq.mfma_mode;
    } else if (why.empty() && !(o.flags & (ADMM_FLAG_NO_MFMA | ADMM_FLAG_NO_ALTERNATE)) &&
               (h->pitch <= 64 || (p->n >= 9 && h->pitch <= 128))) {
      // FP64: the fp64 MFMA form where it is the faster one (ADMM_FLAG_NO_MFMA).  Measured, round 3 (tools/family_time.py;
      // one-lane kernels with the operators distributed over the lanes, dpp_matvec_acc): the MFMA form wins for the
      // smallest batches only -- one wave per segment running ~15 MFMAs per stage instead of a few hundred dependent FMAs --
      // (6, 3): 22 vs 33 us per iteration for one QP, 27 vs 35 at 64 QPs, level at 128, 59 vs 41 at 256, 187 vs 151 at 4096;
      // (12, 6): 34 vs 72 us for one QP, 42 vs 76 at 64, 103 vs 95 at 256, 343 vs 315 at 4096.
      h->mfma_mode = 2;
    }
  }
  // batches of a few QPs run the scan as a matrix-vector product (xscan_gemv_kernel) and need no MFMA-packed scan matrices
  h->scan_gemv = h->batch <= admm::SCAN_GEMV_MAXCOLS && !(o.flags & ADMM_FLAG_SCAN_CHAIN) &&
                 std::getenv("ADMM_NO_GEMV_SCAN") == nullptr;
  std::string err;
  rc = admm::factorise(*p, o.rho, h->S, h->fac, err, h->mfma_mode, !h->scan_gemv, ts_n);
  if (rc) { release(h); return fail(rc, err); }
  // Conditioning guard of the parallel-in-time form: the segment coupling is exact in exact
  // arithmetic, but its transfer matrices are products of closed-loop matrices, and for a barely
  // stabilised plant (tiny rho, no state cost, unstable A) their entries grow with the number of
  // segments and amplify rounding (measured: max|W| 1e3 -> 1e-5 relative error in w, against
  // 1e-14 for the workloads of DESIGN.md §3 where max|W| is O(1)).  With an automatic segment
  // count, fall back to fewer, longer segments until the growth is benign.
  h->auto_segments = o.segments == 0;
  if (o.segments == 0 && ts_n && h->fac.S > ts_n && scan_growth(h->fac) > SCAN_GROWTH_MAX) {
    release(h);            // (every rank must arrive at the same count: no silent back-off here)
    return fail(ADMM_ERR_NUMERIC, "admm_setup_timeshard: the segment transfer matrices grow beyond the conditioning bound with the "
                                  "automatic segment count; give options.segments (a multiple of nranks)");
  }
  if (o.segments == 0 && !ts_n) {
    while (h->fac.S > 1 && scan_growth(h->fac) > SCAN_GROWTH_MAX) {
      const int S2 = std::max(1, h->fac.S / 2);
      rc = admm::factorise(*p, o.rho, S2, h->fac, err, h->mfma_mode, !h->scan_gemv);
      if (rc) { release(h); return fail(rc, err); }
    }
  }
  h->S = h->fac.S;
  if (ts_n) {
    if (h->S % ts_n != 0) { release(h); return fail(ADMM_ERR_INVALID, "admm_setup_timeshard: the segment count " + std::to_string(h->S) + " is not a multiple of nranks"); }
    h->ts_sl = h->S / ts_n;
    h->ts_s0 = ts_rank * h->ts_sl;
  }
  keep_shared(h, p);     // host copy of the shared problem data, for admm_set_rho / the adaptive rule
  {  // the x kernels address one segment of an array through a 32-bit buffer descriptor
    int longest = 0;
    for (int s = 0; s < h->S; ++s) longest = std::max(longest, h->fac.seg_start[s + 1] - h->fac.seg_start[s]);
    if ((double)longest * h->nb * h->pitch * 8.0 >= 2147483648.0) {
      release(h);
      return fail(ADMM_ERR_UNSUPPORTED, "one segment of the state exceeds 2 GiB: use more segments or a smaller batch per GPU");
    }
  }

#define TRY_RELEASE(expr) do { int rc_ = (expr); if (rc_) { std::string keep = g_err; release(h); g_err = keep; return rc_; } } while (0)
#define HIP_TRY_RELEASE(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { release(h); return fail(ADMM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } } while (0)

  HIP_TRY_RELEASE(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  const size_t P = h->pitch, L = h->L;
  TRY_RELEASE(dalloc(&h->w, L * P));
  TRY_RELEASE(dalloc(&h->z, L * P));
  TRY_RELEASE(dalloc(&h->y, L * P));
  TRY_RELEASE(dalloc(&h->v, L * P));
  if (h->has_q) TRY_RELEASE(dalloc(&h->q, L * P));
  TRY_RELEASE(dalloc(&h->dbuf, (size_t)h->N * h->m * P));
  {  // scan operands: in = tseg | x0 | eseg | pad,  out = t_in | pad | x_in | pad  (admm_factor.hpp)
    const size_t Sn = (size_t)h->S * h->n;
    TRY_RELEASE(dalloc(&h->scan_in, (size_t)h->fac.scanK * P));
    {  // split-K of the scan when the grid would be small: aim at >= 256 workgroups, <= 8 slices
      const int wgs = (h->pitch / 64) * (h->fac.scanM / 16 / admm::SCAN_MT);
      int sp = 1;
      while (sp < 8 && wgs * sp < h->num_cus) sp *= 2;
      const int ksteps = h->fac.scanK / 4;
      while (sp > 1 && ksteps / sp < 2 * admm::SCAN_U) sp /= 2;     // at least two batches per slice
      if (const char* e = std::getenv("ADMM_SCAN_SPLIT")) {           // tuning override (1, 2, 4, 8)
        const int v = std::atoi(e);
        if (v == 1 || v == 2 || v == 4 || v == 8) sp = v;
      }
      if (h->scan_gemv) sp = 1;                        // the matrix-vector form writes whole sums
      h->scan_split = sp;
    }
    TRY_RELEASE(dalloc(&h->scan_out, (size_t)h->scan_split * h->fac.scanM * P));
    HIP_TRY_RELEASE(hipMemsetAsync(h->scan_in, 0, sizeof(double) * (size_t)h->fac.scanK * P, h->stream));
    HIP_TRY_RELEASE(hipMemsetAsync(h->scan_out, 0, sizeof(double) * (size_t)h->scan_split * h->fac.scanM * P, h->stream));
    h->tseg = h->scan_in;
    h->x0 = h->scan_in + Sn * P;
    h->eseg = h->scan_in + (Sn + h->n) * P;
    if (h->fac.ts_ranks > 1) {     // time shards: rank-by-rank input rows (admm_factor.hpp); tseg / eseg = THIS rank's block
      const size_t blk = (size_t)2 * h->ts_sl * h->n;
      h->tseg = h->scan_in + (size_t)h->ts_rank * blk * P;
      h->eseg = h->tseg + (size_t)h->ts_sl * h->n * P;
      h->x0 = h->scan_in + 2 * Sn * P;
    }
    h->tin = h->scan_out;
    h->xin = h->scan_out + (size_t)h->fac.scanMt * P;
    TRY_RELEASE(dalloc(&h->scanWp, h->fac.scanWp.size()));
    TRY_RELEASE(dalloc(&h->scan_range, h->fac.scanRange.size()));
    if (h->scan_gemv) {
      TRY_RELEASE(dalloc(&h->scanWd, h->fac.scanW.size()));
      TRY_RELEASE(dalloc(&h->scanWBd, h->fac.scanW.size()));
      TRY_RELEASE(dalloc(&h->scan_rows, (size_t)2 * h->fac.scanM));
      TRY_RELEASE(dalloc(&h->scan_rowsB, (size_t)2 * h->fac.scanM));
    }
  }
  // alternating-direction iteration: compiled for this (n, m), buildable for this problem, not disabled
  h->alt_allowed = h->fac.alt_ok && fused(h) &&
                   !(h->opt.flags & (ADMM_FLAG_SCAN_CHAIN | ADMM_FLAG_NO_ALTERNATE)) &&
                   dispatch_x(xlaunch_of(h), admm::XKernel::XFZE, false, false, /*query_only=*/true);
  h->alt = h->alt_allowed;
  h->alt_requested = fused(h) && !(h->opt.flags & (ADMM_FLAG_SCAN_CHAIN | ADMM_FLAG_NO_ALTERNATE)) &&
                     dispatch_x(xlaunch_of(h), admm::XKernel::XFZE, false, false, /*query_only=*/true);
  if (h->alt_requested && !h->fac.alt_ok) warn_alt_gate(h->fac, o.rho, "admm_setup");
  h->mfma_on = h->mfma_mode != 0 && (o.precision_mode == ADMM_PRECISION_MIXED || h->alt);
  if (h->mfma_mode) {
    // (+ 1 KiB: the LDS-DMA copy of a chunk moves whole KiB pieces, admm_mfma.hpp)
    HIP_TRY_RELEASE(hipMalloc((void**)&h->recMF, h->fac.recMF.size() + 1024));
    HIP_TRY_RELEASE(hipMalloc((void**)&h->recMB, h->fac.recMB.size() + 1024));
    HIP_TRY_RELEASE(hipMemset(h->recMF, 0, h->fac.recMF.size() + 1024));
    HIP_TRY_RELEASE(hipMemset(h->recMB, 0, h->fac.recMB.size() + 1024));
    HIP_TRY_RELEASE(hipMemcpy(h->recMF, h->fac.recMF.data(), h->fac.recMF.size(), hipMemcpyHostToDevice));
    HIP_TRY_RELEASE(hipMemcpy(h->recMB, h->fac.recMB.data(), h->fac.recMB.size(), hipMemcpyHostToDevice));
    if (!h->fac.recMF64.empty()) {
      HIP_TRY_RELEASE(hipMalloc((void**)&h->recMF64, h->fac.recMF64.size() + 1024));
      HIP_TRY_RELEASE(hipMalloc((void**)&h->recMB64, h->fac.recMB64.size() + 1024));
      HIP_TRY_RELEASE(hipMemset(h->recMF64, 0, h->fac.recMF64.size() + 1024));
      HIP_TRY_RELEASE(hipMemset(h->recMB64, 0, h->fac.recMB64.size() + 1024));
      HIP_TRY_RELEASE(hipMemcpy(h->recMF64, h->fac.recMF64.data(), h->fac.recMF64.size(), hipMemcpyHostToDevice));
      HIP_TRY_RELEASE(hipMemcpy(h->recMB64, h->fac.recMB64.data(), h->fac.recMB64.size(), hipMemcpyHostToDevice));
    }
    TRY_RELEASE(dalloc(&h->status1, (size_t)h->pitch));
    TRY_RELEASE(dalloc(&h->iters1, (size_t)h->pitch));
  }
  const bool need_alt_buffers = h->alt_allowed;
  if (need_alt_buffers) {
    TRY_RELEASE(dalloc(&h->recFE, h->fac.recFE.size()));
    TRY_RELEASE(dalloc(&h->recBE, h->fac.recBE.size()));
    TRY_RELEASE(dalloc(&h->scanWpB, h->fac.scanWpB.size()));
    TRY_RELEASE(dalloc(&h->scan_rangeB, h->fac.scanRangeB.size()));
    TRY_RELEASE(dalloc(&h->mvec, (size_t)h->N * h->m * P));     // db rows of the forward elimination
    HIP_TRY_RELEASE(hipMemsetAsync(h->mvec, 0, sizeof(double) * (size_t)h->N * h->m * P, h->stream));
    HIP_TRY_RELEASE(hipMemcpy(h->recFE, h->fac.recFE.data(), sizeof(double) * h->fac.recFE.size(), hipMemcpyHostToDevice));
    HIP_TRY_RELEASE(hipMemcpy(h->recBE, h->fac.recBE.data(), sizeof(double) * h->fac.recBE.size(), hipMemcpyHostToDevice));
    if (!h->scan_gemv) HIP_TRY_RELEASE(hipMemcpy(h->scanWpB, h->fac.scanWpB.data(), sizeof(double) * h->fac.scanWpB.size(), hipMemcpyHostToDevice));
    if (!h->scan_gemv) HIP_TRY_RELEASE(hipMemcpy(h->scan_rangeB, h->fac.scanRangeB.data(), sizeof(int32_t) * h->fac.scanRangeB.size(), hipMemcpyHostToDevice));
  }
  const size_t part_chunks = (size_t)(h->zchunks > h->S ? h->zchunks : h->S);
  TRY_RELEASE(dalloc(&h->part, part_chunks * 5 * P));
  TRY_RELEASE(dalloc(&h->resid, 5 * P));
  TRY_RELEASE(dalloc(&h->lo, L));
  TRY_RELEASE(dalloc(&h->hi, L));
  TRY_RELEASE(dalloc(&h->ub, (size_t)h->N));
  TRY_RELEASE(dalloc(&h->recB, h->fac.recB.size()));
  TRY_RELEASE(dalloc(&h->recF, h->fac.recF.size()));
  TRY_RELEASE(dalloc(&h->recS, h->fac.recS.size()));
  TRY_RELEASE(dalloc(&h->seg_start, h->fac.seg_start.size()));
  TRY_RELEASE(dalloc(&h->status, P));
  TRY_RELEASE(dalloc(&h->iters, P));
  TRY_RELEASE(dalloc(&h->nconv, 1));
  h->stage_rows = L;
  TRY_RELEASE(dalloc(&h->stage, L * (size_t)h->batch));
  HIP_TRY_RELEASE(hipHostMalloc((void**)&h->h_nconv, sizeof(int), hipHostMallocDefault));

  HIP_TRY_RELEASE(hipMemsetAsync(h->w, 0, sizeof(double) * L * P, h->stream));
  HIP_TRY_RELEASE(hipMemsetAsync(h->z, 0, sizeof(double) * L * P, h->stream));
  HIP_TRY_RELEASE(hipMemsetAsync(h->y, 0, sizeof(double) * L * P, h->stream));
  HIP_TRY_RELEASE(hipMemsetAsync(h->v, 0, sizeof(double) * L * P, h->stream));
  HIP_TRY_RELEASE(hipMemsetAsync(h->dbuf, 0, sizeof(double) * (size_t)h->N * h->m * P, h->stream));
  HIP_TRY_RELEASE(hipMemsetAsync(h->part, 0, sizeof(double) * part_chunks * 5 * P, h->stream));
  HIP_TRY_RELEASE(hipMemsetAsync(h->resid, 0, sizeof(double) * 5 * P, h->stream));
  HIP_TRY_RELEASE(hipMemsetAsync(h->status, 0, sizeof(int) * P, h->stream));
  HIP_TRY_RELEASE(hipMemsetAsync(h->iters, 0, sizeof(int) * P, h->stream));

  TRY_RELEASE(upload_bounds(h, p));
  HIP_TRY_RELEASE(hipMemcpy(h->recB, h->fac.recB.data(), sizeof(double) * h->fac.recB.size(), hipMemcpyHostToDevice));
  HIP_TRY_RELEASE(hipMemcpy(h->recF, h->fac.recF.data(), sizeof(double) * h->fac.recF.size(), hipMemcpyHostToDevice));
  HIP_TRY_RELEASE(hipMemcpy(h->recS, h->fac.recS.data(), sizeof(double) * h->fac.recS.size(), hipMemcpyHostToDevice));
  if (!h->scan_gemv) HIP_TRY_RELEASE(hipMemcpy(h->scanWp, h->fac.scanWp.data(), sizeof(double) * h->fac.scanWp.size(), hipMemcpyHostToDevice));
  if (!h->scan_gemv) HIP_TRY_RELEASE(hipMemcpy(h->scan_range, h->fac.scanRange.data(), sizeof(int32_t) * h->fac.scanRange.size(), hipMemcpyHostToDevice));
  HIP_TRY_RELEASE(hipMemcpy(h->seg_start, h->fac.seg_start.data(), sizeof(int32_t) * h->fac.seg_start.size(), hipMemcpyHostToDevice));
  if (h->scan_gemv) {
    TRY_RELEASE(upload_scan_dense(h->fac.scanW, h->fac.scanM, h->fac.scanK, h->scanWd, h->scan_rows));
    if (h->alt_allowed) TRY_RELEASE(upload_scan_dense(h->fac.scanWB, h->fac.scanM, h->fac.scanK, h->scanWBd, h->scan_rowsB));
  }
  TRY_RELEASE(upload_transposed(h, p->x0, h->x0, h->n));
  if (h->has_q) TRY_RELEASE(upload_transposed(h, p->q, h->q, h->L));
  HIP_TRY_RELEASE(hipStreamSynchronize(h->stream));
#undef TRY_RELEASE
#undef HIP_TRY_RELEASE
  *out = h;
  return ADMM_OK;
}

int admm_setup(admm_handle** out, const admm_problem* p, const admm_options* o) {
  return setup_common(out, p, o, 0, 0, nullptr, nullptr);
}

int admm_setup_timeshard(admm_handle** out, const admm_problem* p, const admm_options* o, int32_t rank, int32_t nranks,
                         admm_exchange_fn exchange, void* ctx) {
  if (nranks < 1) return fail(ADMM_ERR_INVALID, "admm_setup_timeshard: nranks must be >= 1");
  return setup_common(out, p, o, rank, nranks, exchange, ctx);
}

int admm_get_window(admm_handle* h, int32_t* stage_lo, int32_t* stage_hi, int32_t* seg_lo, int32_t* segs_local, int32_t* segs_total) {
  if (!h) return fail(ADMM_ERR_INVALID, "NULL handle");
  const bool ts = h->ts_n > 0 && !h->pinst;
  const int s0 = ts ? h->ts_s0 : 0, sl = ts ? h->ts_sl : h->S;
  if (stage_lo) *stage_lo = h->pinst ? 0 : h->fac.seg_start[s0];
  if (stage_hi) *stage_hi = h->pinst ? h->N : h->fac.seg_start[s0 + sl];
  if (seg_lo) *seg_lo = s0;
  if (segs_local) *segs_local = sl;
  if (segs_total) *segs_total = h->S;
  return ADMM_OK;
}

int admm_update_instances(admm_handle* h, const double* x0, const double* q) {
  if (!h) return fail(ADMM_ERR_INVALID, "NULL handle");
  HIP_TRY(hipSetDevice(h->device));
  int rc;
  if ((x0 || q) && (rc = ensure_w(h))) return rc;   // w of the last x-update belongs to the old instance data
  if (x0 || q) h->alt_state = admm_handle::ALT_NONE;
  if (x0) {
    if (!finite_all(x0, (size_t)h->n * h->batch)) return fail(ADMM_ERR_INVALID, "non-finite entry in x0");
    if ((rc = upload_transposed(h, x0, h->x0, h->n))) return rc;
  }
  if (q) {
    if (!h->has_q) return fail(ADMM_ERR_INVALID, "handle was set up without q; cannot add one later");
    if (!finite_all(q, (size_t)h->L * h->batch)) return fail(ADMM_ERR_INVALID, "non-finite entry in q");
    if ((rc = upload_transposed(h, q, h->q, h->L))) return rc;
  }
  return ADMM_OK;
}

static admm_problem shared_problem(const admm_handle* h) {
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
static bool spec_enabled(const admm_handle* h) {
  const bool off = std::getenv("ADMM_NO_SPECULATE") != nullptr;     // read per call: tests switch it within a process
  // (only inside a solve: a caller sweeping admm_set_rho on an adaptive handle outside one must not accumulate a full Factor
  //  copy per call -- ADVICE r02)
  return !off && h->solve_active && !h->pinst && h->opt.adapt_interval > 0 && h->rho_updates < h->opt.adapt_max;
}

static void spec_reap(admm_handle* h, bool all) {          // drop stale entries whose thread has finished (all: join them)
  if (all) { h->spec_stale.clear(); return; }
  // a finished thread is recognised by rc having been published; unfinished ones stay until the next reap
  for (size_t i = 0; i < h->spec_stale.size();)
    if (!h->spec_stale[i]->th.joinable() || __atomic_load_n(&h->spec_stale[i]->rc, __ATOMIC_ACQUIRE) != INT32_MIN)
      h->spec_stale.erase(h->spec_stale.begin() + i);
    else ++i;
}

static std::unique_ptr<SpecFactor> spec_take(admm_handle* h, double rho) {
  for (size_t i = 0; i < h->spec.size(); ++i)
    if (h->spec[i]->rho == rho) {
      std::unique_ptr<SpecFactor> sp = std::move(h->spec[i]);
      h->spec.erase(h->spec.begin() + i);
      if (sp->th.joinable()) sp->th.join();
      return sp;
    }
  return nullptr;
}

static void spec_start(admm_handle* h) {
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
static int set_rho_internal(admm_handle* h, double rho_new) {
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
    const size_t count2 = (size_t)h->L * h->pitch / 2;      // pitch is even
    hipLaunchKernelGGL(admm::scale_kernel, dim3(2048), dim3(256), 0, h->stream, h->y, c, count2);
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

int admm_set_state(admm_handle* h, const double* w, const double* z, const double* y) {
  if (!h) return fail(ADMM_ERR_INVALID, "NULL handle");
  HIP_TRY(hipSetDevice(h->device));
  int rc;
  // the device code is compiled with -fno-honor-nans on the premise that nothing non-finite gets in
  const size_t cnt = (size_t)h->L * h->batch;
  if ((w && !finite_all(w, cnt)) || (z && !finite_all(z, cnt)) || (y && !finite_all(y, cnt)))
    return fail(ADMM_ERR_INVALID, "non-finite entry in w, z or y");
  if (w) {
    if ((rc = upload_transposed(h, w, h->w, h->L))) return rc;
    h->w_stale = false;
  }
  if (z || y) {
    if ((rc = ensure_zy(h))) return rc;        // keep the one that is not overwritten
    if (z && (rc = upload_transposed(h, z, h->z, h->L))) return rc;
    if (y && (rc = upload_transposed(h, y, h->y, h->L))) return rc;
    h->zy_valid = true;
    h->v_valid = false;                         // an arbitrary (z, y) pair need not be of the form (clip(v), v - clip(v))
    h->alt_state = admm_handle::ALT_NONE;
  }
  return ADMM_OK;
}

int admm_step_x(admm_handle* h) {
  if (!h) return fail(ADMM_ERR_INVALID, "NULL handle");
  HIP_TRY(hipSetDevice(h->device));
  int rc = step_x(h);
  if (rc) return rc;
  h->w_stale = false;
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  return ADMM_OK;
}

int admm_step_z(admm_handle* h, int32_t residuals) {
  if (!h) return fail(ADMM_ERR_INVALID, "NULL handle");
  if (h->pbounds) return fail(ADMM_ERR_UNSUPPORTED, "admm_step_z: the standalone z kernel has no per-instance bounds form");
  if (h->ts_n) return fail(ADMM_ERR_UNSUPPORTED, "admm_step_z is not available on a time-sharded handle (the standalone z kernel's partial sums are not per segment)");
  HIP_TRY(hipSetDevice(h->device));
  int rc = ensure_w(h);
  if (!rc) rc = ensure_zy(h);
  if (!rc) rc = launch_z(h, residuals != 0);
  if (rc) return rc;
  h->v_valid = false;
  h->alt_state = admm_handle::ALT_NONE;
  if (residuals) {
    launch_finalize(h, 0, h->zchunks);
    h->resid_valid = true;
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  return ADMM_OK;
}

// Enqueue iteration number `it` (1-based within the caller's loop).  The graphs hold the
// steady-state kernel forms; an iteration whose input state is not in that form (first
// fused iteration after setup / set_state / an unfused step) is launched directly.
// `remaining` = iterations (this one included) the caller still enqueues before it returns:
// it selects the iteration form (next_form).  it_number > 0: a checked iteration of admm_solve
// (residuals + finalise with that iteration number, launched directly).
// next_plain: another iteration follows in this call and evaluates no residuals (admm_run / admm_solve_step know).
static int enqueue_one(admm_handle* h, bool resid, bool use_graph, int remaining, int it_number = 0, bool next_plain = false) {
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

int admm_run(admm_handle* h, int32_t iters, int32_t residual_every) {
  if (!h) return fail(ADMM_ERR_INVALID, "NULL handle");
  if (iters < 0 || residual_every < 0) return fail(ADMM_ERR_INVALID, "iters / residual_every must be >= 0");
  HIP_TRY(hipSetDevice(h->device));
  const bool use_graph = (h->opt.flags & ADMM_FLAG_GRAPH) != 0;
  if (use_graph && !h->graph_exec[0]) {
    int rc = capture_iterations(h);
    if (rc) return rc;
  }
  for (int it = 1; it <= iters; ++it) {
    const bool resid = residual_every > 0 && (it % residual_every == 0);
    const bool next_plain = it < iters && !(residual_every > 0 && ((it + 1) % residual_every == 0));
    int rc = enqueue_one(h, resid, use_graph, iters - it + 1, 0, next_plain);
    if (rc) return rc;
  }
  int rcf = flush_finalize(h);
  if (rcf) return rcf;
  HIP_TRY(hipGetLastError());
  return ADMM_OK;
}

int admm_iterate(admm_handle* h, int32_t iters) { return admm_run(h, iters, 0); }

int admm_sync(admm_handle* h) {
  if (!h) return fail(ADMM_ERR_INVALID, "NULL handle");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return ADMM_OK;
}

int admm_solve_begin(admm_handle* h, const double* z0, const double* y0) {
  g_warn.clear();
  if (!h) return fail(ADMM_ERR_INVALID, "NULL handle");
  HIP_TRY(hipSetDevice(h->device));
  h->solve_t0 = std::chrono::steady_clock::now();
  int rc;
  if ((rc = admm_set_state(h, nullptr, z0, y0))) return rc;
  const size_t P = h->pitch;
  HIP_TRY(hipMemsetAsync(h->status, 0, sizeof(int) * P, h->stream));
  {
    std::vector<int> init(P, h->opt.max_iter);
    HIP_TRY(hipMemcpyAsync(h->iters, init.data(), sizeof(int) * P, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
  }
  const bool use_graph = (h->opt.flags & ADMM_FLAG_GRAPH) != 0;
  if (use_graph && !h->graph_exec[0] && (rc = capture_iterations(h))) return rc;
  h->rho_updates = 0;
  h->history.clear();
  if (h->pinst) HIP_TRY(hipMemsetAsync(h->nupd_d, 0, sizeof(int) * P, h->stream));
  h->spec_hits = h->spec_misses = 0;
  h->solve_active = true;
  spec_start(h);                               // adaptive rule: factorise rho tau and rho / tau while the GPU iterates
  h->solve_it = 0;
  h->solve_nconv = 0;
  h->mixed_iters = 0;
  h->mixed_phase1 = h->opt.precision_mode == ADMM_PRECISION_MIXED;
  if (h->mixed_phase1) {
    if ((rc = flush_finalize(h))) return rc;
    set_mixed_form(h, true);
    HIP_TRY(hipMemsetAsync(h->status1, 0, sizeof(int) * P, h->stream));
  }
  return ADMM_OK;
}

int admm_solve_step(admm_handle* h, int32_t* iters_done, int32_t* n_converged, double* R_out, double* S_out) {
  if (!h) return fail(ADMM_ERR_INVALID, "NULL handle");
  if (h->solve_it >= h->opt.max_iter) return fail(ADMM_ERR_INVALID, "admm_solve_step: max_iter already reached");
  HIP_TRY(hipSetDevice(h->device));
  const bool use_graph = (h->opt.flags & ADMM_FLAG_GRAPH) != 0;
  if (use_graph && !h->graph_exec[0]) { int rc0 = capture_iterations(h); if (rc0) return rc0; }
  const int ci = h->opt.check_interval;
  const size_t P = h->pitch;
  int rc;
  for (;;) {
    const int it = ++h->solve_it;
    const bool check = (it % ci == 0) || it == h->opt.max_iter;
    if (!check) {
      const int next_check = std::min(((it / ci) + 1) * ci, h->opt.max_iter);
      if ((rc = enqueue_one(h, false, use_graph, next_check - it + 1, 0, /*next_plain=*/it + 1 < next_check))) return rc;
      continue;
    }
    // checked iteration: launched directly so that the finalise kernel gets the iteration number
    HIP_TRY(hipMemsetAsync(h->nconv, 0, sizeof(int), h->stream));
    if ((rc = enqueue_one(h, true, use_graph, 1, it))) return rc;
    HIP_TRY(hipMemcpyAsync(h->h_nconv, h->nconv, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->solve_nconv = *h->h_nconv;
    h->resid_valid = true;
    if (h->opt.flags & ADMM_FLAG_HISTORY) {
      std::vector<double> rs(2 * P);
      HIP_TRY(hipMemcpy(rs.data(), h->resid, sizeof(double) * 2 * P, hipMemcpyDeviceToHost));
      admm_handle::HistoryEntry e{it, h->solve_nconv, 0.0, 0.0, h->opt.rho};
      for (int b = 0; b < h->batch; ++b) { e.max_r = std::max(e.max_r, rs[b]); e.max_s = std::max(e.max_s, rs[P + b]); }
      if (h->pinst) {
        std::vector<double> rq(P);
        HIP_TRY(hipMemcpy(rq.data(), h->rho_d, sizeof(double) * P, hipMemcpyDeviceToHost));
        e.rho = *std::max_element(rq.begin(), rq.begin() + h->batch);
      }
      if (h->mixed_phase1) e.nconv = 0;            // (the fp32 phase counts against raised tolerances: nothing has converged)
      h->history.push_back(e);
    }
    if (h->mixed_phase1) {
      // fp32 phase: the count is of the RAISED tolerances.  Once every QP meets them (or the budget is spent) the
      // solve continues with the fp64 kernels, which check the rule as given; nothing has converged so far.
      h->mixed_iters = h->solve_it;
      if (h->solve_nconv >= h->batch) {
        h->mixed_phase1 = false;
        set_mixed_form(h, false);
      }
      h->solve_nconv = 0;
    }
    break;
  }
  if (iters_done) *iters_done = h->solve_it;
  if (n_converged) *n_converged = h->solve_nconv;
  if (R_out || S_out) {
    // sums of r^2 and s^2 over the QPs that have not converged, in batch order (adaptive-rho rule)
    std::vector<double> rs(2 * P);
    std::vector<int> st(P);
    HIP_TRY(hipMemcpy(rs.data(), h->resid, sizeof(double) * 2 * P, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(st.data(), h->status, sizeof(int) * P, hipMemcpyDeviceToHost));
    double R = 0.0, S = 0.0;
    for (int b = 0; b < h->batch; ++b)
      if (!st[b]) { R += rs[b] * rs[b]; S += rs[P + b] * rs[P + b]; }
    if (R_out) *R_out = R;
    if (S_out) *S_out = S;
  }
  return ADMM_OK;
}

int admm_solve_adapt(admm_handle* h, double R, double S, int32_t* changed) {
  if (!h) return fail(ADMM_ERR_INVALID, "NULL handle");
  HIP_TRY(hipSetDevice(h->device));
  if (changed) *changed = 0;
  const int it = h->solve_it;
  if (!(h->opt.adapt_interval > 0 && it % h->opt.adapt_interval == 0 && (h->pinst || h->rho_updates < h->opt.adapt_max) &&
        it < h->opt.max_iter))
    return ADMM_OK;                      // (per-instance: adapt_max is counted per QP, on the device)
  const double mu2 = h->opt.adapt_mu * h->opt.adapt_mu;
  if (h->pinst) {
    // Per-instance dynamics: the rule runs QP by QP on the device (csrc/admm_pinst.hpp, padapt_kernel); R and S -- the
    // batch sums of the shared-factor rule -- are not used, so a sharded solve needs no exchange for it.
    int rc;
    HIP_TRY(hipMemsetAsync(h->nchanged_d, 0, sizeof(int), h->stream));
    if ((rc = pinst_alloc_trial(h, false))) return rc;
    admm::launch_padapt(h->stream, h->resid, h->status, h->rho_d, h->nupd_d, h->todo_d, h->cscale_d, h->nchanged_d, mu2,
                        h->opt.adapt_tau, h->opt.adapt_max, h->pitch, h->batch, h->rho2_d);
    HIP_TRY(hipGetLastError());
    int nchanged = 0;
    HIP_TRY(hipMemcpyAsync(&nchanged, h->nchanged_d, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (!nchanged) return ADMM_OK;
    if (h->S > 1 && h->auto_segments) {
      // the candidates are factorised on trial (scratch K / S, transfer matrices not stored); a QP whose new rho
      // breaks the conditioning bound of its segments -- or has no factor -- keeps its rho and stops adapting, before its dual
      // is rescaled or its factor touched: what the batch-level rule does for shared dynamics.
      int not_pd = 0, grown = 0;
      if ((rc = pinst_try(h, h->Ad, h->Bd, h->Qd, h->Rd, h->QNd, h->rho_d, h->todo_d, &not_pd, &grown))) return rc;
      if (not_pd || grown) {
        int nveto = 0;
        HIP_TRY(hipMemsetAsync(h->nveto_d, 0, sizeof(int), h->stream));
        admm::launch_padapt_veto(h->stream, h->qflag_d, h->rho2_d, h->rho_d, h->nupd_d, h->todo_d, h->cscale_d, h->nchanged_d,
                                 h->nveto_d, h->opt.adapt_max, h->pitch);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(&nveto, h->nveto_d, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipMemcpyAsync(&nchanged, h->nchanged_d, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        g_warn = "adaptive rho: the change of " + std::to_string(nveto) + " QP(s) was refused (segment transfer matrices beyond the "
                 "conditioning bound, or no factor, with the new rho); those QPs keep their rho and stop adapting for this solve";
        if (!nchanged) return ADMM_OK;
      }
    }
    if ((rc = ensure_w(h))) return rc;            // w of the last x-update is rebuilt with the OLD factors (rho is not used)
    if ((rc = ensure_zy(h))) return rc;
    admm::launch_padapt_scale(h->stream, h->y, h->cscale_d, h->todo_d, h->L, h->pitch);
    HIP_TRY(hipGetLastError());
    h->zy_valid = true;
    h->v_valid = false;
    if ((rc = pinst_factor(h, /*only_marked=*/true))) return rc;
    h->rho_updates += nchanged;
    if (changed) *changed = 1;
    return ADMM_OK;
  }
  double rho_new = h->opt.rho;
  if (R > mu2 * S) rho_new = h->opt.rho * h->opt.adapt_tau;
  else if (S > mu2 * R) rho_new = h->opt.rho / h->opt.adapt_tau;
  if (rho_new != h->opt.rho) {
    int rc = set_rho_internal(h, rho_new);
    if (rc == ADMM_ERR_NUMERIC) {          // refused (conditioning guard / factorisation): keep rho, stop adapting
      g_warn = "adaptive rho: the change to rho = " + std::to_string(rho_new) + " was refused (" + g_err + "); rho stays at " +
               std::to_string(h->opt.rho) + " and the rule stops adapting for this solve";
      h->rho_updates = h->opt.adapt_max;
      spec_start(h);
      return ADMM_OK;
    }
    if (rc) return rc;
    ++h->rho_updates;
    spec_start(h);                         // the candidates of the new rho
    if (changed) *changed = 1;
  }
  return ADMM_OK;
}

int admm_solve_end(admm_handle* h, admm_info* info) {
  if (!h) return fail(ADMM_ERR_INVALID, "NULL handle");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipGetLastError());
  if (std::getenv("ADMM_SPEC_DEBUG"))
    std::fprintf(stderr, "[admm] rho changes served by a background / kept factor: %d, factorised on demand: %d\n", h->spec_hits, h->spec_misses);
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->iters_run = h->solve_it;
  h->solve_active = false;                      // candidates of the adaptive rule are not kept between solves (their threads
  for (auto& sp2 : h->spec) h->spec_stale.push_back(std::move(sp2));      // are joined lazily, or by admm_free / admm_update_problem)
  h->spec.clear();
  spec_reap(h, false);
  if (h->opt.precision_mode == ADMM_PRECISION_MIXED) {
    if (h->mixed_phase1) h->mixed_iters = h->solve_it;
    h->mixed_phase1 = false;
    set_mixed_form(h, true);                 // admm_run / admm_iterate after a solve run the fp32 form again
  }
  const size_t P = h->pitch;
  if (info) {
    info->iters_run = h->solve_it;
    info->n_converged = h->solve_nconv;
    std::vector<double> rs(2 * P);
    HIP_TRY(hipMemcpy(rs.data(), h->resid, sizeof(double) * 2 * P, hipMemcpyDeviceToHost));
    double mr = 0, ms = 0;
    for (int b = 0; b < h->batch; ++b) {
      if (rs[b] > mr) mr = rs[b];
      if (rs[P + b] > ms) ms = rs[P + b];
    }
    info->max_r = mr;
    info->max_s = ms;
    info->solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h->solve_t0).count();
    info->rho = h->opt.rho;
    if (h->pinst) {                       // per-QP rho: the largest in force (admm_get_rho returns all of them)
      std::vector<double> r(h->pitch);
      HIP_TRY(hipMemcpy(r.data(), h->rho_d, sizeof(double) * r.size(), hipMemcpyDeviceToHost));
      info->rho = *std::max_element(r.begin(), r.begin() + h->batch);
    }
    info->rho_updates = h->rho_updates;
    info->mixed_iters = h->opt.precision_mode == ADMM_PRECISION_MIXED ? h->mixed_iters : 0;
  }
  return ADMM_OK;
}

int admm_solve(admm_handle* h, const double* z0, const double* y0, admm_info* info) {
  int rc = admm_solve_begin(h, z0, y0);
  if (rc) return rc;
  const bool adaptive = h->opt.adapt_interval > 0;
  for (;;) {
    int32_t it = 0, nconv = 0;
    double R = 0.0, S = 0.0;
    if ((rc = admm_solve_step(h, &it, &nconv, adaptive ? &R : nullptr, adaptive ? &S : nullptr))) return rc;
    if (nconv >= h->batch || it >= h->opt.max_iter) break;
    if (adaptive && (rc = admm_solve_adapt(h, R, S, nullptr))) return rc;
  }
  return admm_solve_end(h, info);
}

int admm_get_residuals(admm_handle* h, double* r, double* s, double* nw, double* nz, double* ny) {
  if (!h) return fail(ADMM_ERR_INVALID, "NULL handle");
  if (!h->resid_valid) return fail(ADMM_ERR_INVALID, "no residual-evaluating z step has run yet");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(h->stream));
  double* dst[5] = {r, s, nw, nz, ny};
  for (int v = 0; v < 5; ++v)
    if (dst[v]) HIP_TRY(hipMemcpy(dst[v], h->resid + (size_t)v * h->pitch, sizeof(double) * h->batch, hipMemcpyDeviceToHost));
  return ADMM_OK;
}

int admm_get(admm_handle* h, double* w, double* z, double* y) {
  if (!h) return fail(ADMM_ERR_INVALID, "NULL handle");
  HIP_TRY(hipSetDevice(h->device));
  int rc;
  if (w && (rc = ensure_w(h))) return rc;
  if ((z || y) && (rc = ensure_zy(h))) return rc;
  if (w && (rc = download_transposed(h, h->w, w, h->L))) return rc;
  if (z && (rc = download_transposed(h, h->z, z, h->L))) return rc;
  if (y && (rc = download_transposed(h, h->y, y, h->L))) return rc;
  return ADMM_OK;
}

int admm_get_info(admm_handle* h, int32_t* iters, int32_t* status, double* r, double* s) {
  if (!h) return fail(ADMM_ERR_INVALID, "NULL handle");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (iters) HIP_TRY(hipMemcpy(iters, h->iters, sizeof(int32_t) * h->batch, hipMemcpyDeviceToHost));
  if (status) HIP_TRY(hipMemcpy(status, h->status, sizeof(int32_t) * h->batch, hipMemcpyDeviceToHost));
  if (r) HIP_TRY(hipMemcpy(r, h->resid, sizeof(double) * h->batch, hipMemcpyDeviceToHost));
  if (s) HIP_TRY(hipMemcpy(s, h->resid + h->pitch, sizeof(double) * h->batch, hipMemcpyDeviceToHost));
  return ADMM_OK;
}

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

int admm_get_rho(admm_handle* h, double* rho) {
  if (!h || !rho) return fail(ADMM_ERR_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(h->device));
  if (!h->pinst) {
    for (int b = 0; b < h->batch; ++b) rho[b] = h->opt.rho;
    return ADMM_OK;
  }
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(rho, h->rho_d, sizeof(double) * h->batch, hipMemcpyDeviceToHost));
  return ADMM_OK;
}

int admm_get_geometry(admm_handle* h, int32_t* pitch, int32_t* segs, int32_t* zrows, int32_t* zchunks) {
  if (!h) return fail(ADMM_ERR_INVALID, "NULL handle");
  if (pitch) *pitch = h->pitch;
  if (segs) *segs = h->S;
  if (zrows) *zrows = h->zrows;
  if (zchunks) *zchunks = h->zchunks;
  return ADMM_OK;
}

int admm_get_path(admm_handle* h, admm_path_info* info) {
  if (!h || !info) return fail(ADMM_ERR_INVALID, "NULL argument");
  std::memset(info, 0, sizeof *info);
  info->alternating = h->alt ? 1 : 0;
  info->alt_requested = h->alt_requested ? 1 : 0;
  info->mfma = h->mfma_mode != 0 && (h->opt.precision_mode == ADMM_PRECISION_MIXED || h->alt) ? h->mfma_mode : 0;
  info->xfree = h->xfree ? 1 : 0;
  info->segments = h->S;
  info->auto_segments = h->auto_segments ? 1 : 0;
  info->scan_form = h->pinst ? 3 : (h->opt.flags & ADMM_FLAG_SCAN_CHAIN) ? 2 : h->scan_gemv ? 1 : 0;
  info->per_instance = h->pinst ? 1 : 0;
  info->alt_check = h->pinst ? -1.0 : h->fac.alt_check;
  info->alt_gate = ALT_GATE;
  info->scan_growth = h->pinst ? 0.0 : scan_growth(h->fac);
  return ADMM_OK;
}

void admm_free(admm_handle* h) { release(h); }

}  // extern "C"

// admm_runtime.hpp -- internals shared by the translation units of the solver runtime (round 3: csrc/admm_api.hip used to hold
// all of it in 2.4 k lines):
//   admm_api.hip      the C ABI entry points (setup, state, iteration, solve, read-out)
//   admm_launch.hip   kernel launches, the time-shard exchange, iteration forms and their schedule, graph capture
//   admm_hostio.hip   host <-> device transfers, validation, uploads of a factor, handle lifetime
//   admm_rho_update.hip rho changes and problem updates: background candidate factors, admm_set_rho, admm_update_problem
//   admm_pinst_rt.hip per-instance dynamics: device factorisation, trial factorisation, set-up
//   admm_profile.hip  admm_profile (per-kernel HIP-event timing)
// Everything here is namespace admm::rt; nothing in this header is part of the ABI.
#pragma once

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
  // Stage window [wk0, wk1) of the big per-stage arrays (w, z, y, v, q: (n + m) rows per stage; dbuf, mvec: m rows per stage):
  // the whole horizon for an ordinary handle; a time shard of several ranks ALLOCATES only its own stages.  The pointers
  // h->w ... are biased by the window's first row (allocation - wk0 * rows_per_stage * pitch), so the kernels, which address
  // rows by their global stage index, run unchanged; only the window's rows are ever touched (admm::rt::win_* below).
  int wk0 = 0, wk1 = 0;
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
  double *lodT = nullptr, *hidT = nullptr;   // wide shapes: a second, TILED copy of the per-instance box for the sweeps (csrc/admm_pinst.hpp, Operand)
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
  bool pi_tiled = false;         // wide shapes: operand arrays A, B, K, S^-1, Omega, Psi in the tiled layout (csrc/admm_pinst.hpp, Operand)
  bool pi_rows_factor = false;   // (6, 3), ADMM_PI_ROWS_FACTOR=1: factorisation / transfer matrices through the wide shapes' kernels
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


namespace admm {
namespace rt {

using admm::Z_THREADS;

extern thread_local std::string g_err;       // admm_last_error()
extern thread_local std::string g_warn;      // admm_last_warning(): a call succeeded but changed the kernels a handle runs

inline int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return ::admm::rt::fail(ADMM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_) + " (" +     \
                                    __FILE__ + ":" + std::to_string(__LINE__) + ")");          \
  } while (0)

constexpr double ALT_GATE = 5e-12;      // the bound build_alternating applies (csrc/admm_factor.cpp)
constexpr double SCAN_GROWTH_MAX = 100.0;

// Iteration forms.  IT_PLAIN is always available; the others need h->alt and
// the state in v-form.
enum IterForm {
  IT_PLAIN = 0,     // xb + scan + xfz                    leaves ALT_NONE
  IT_FWD_START = 1, // xb + scan + xfze                   leaves ALT_FWD
  IT_FWD = 2,       // scan + xfze        (needs ALT_BWD) leaves ALT_FWD
  IT_BWD = 3        // scan (WB) + xbze   (needs ALT_FWD) leaves ALT_BWD
};

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

template <typename T>
int dalloc(T** p, size_t count) {
  hipError_t e = hipMalloc((void**)p, sizeof(T) * (count ? count : 1));
  if (e != hipSuccess) return fail(ADMM_ERR_ALLOC, std::string("hipMalloc failed: ") + hipGetErrorString(e));
  return ADMM_OK;
}


// ---- launches, exchange and iteration forms (admm_launch.hip)
admm::XLaunch xlaunch_of(const admm_handle* h);
admm::PLaunch plaunch_of(const admm_handle* h);
int launch_p(admm_handle* h, admm::PKernel k, bool vform, bool resid);
bool dispatch_x(const admm::XLaunch& l, admm::XKernel k, bool a, bool b, bool query_only);
bool dims_supported(int n, int m);
std::string supported_list();
int launch_x(admm_handle* h, admm::XKernel k, bool a, bool b);
admm::FinArgs fin_args(const admm_handle* h, int it, int nchunks);
int ts_allgather(admm_handle* h, double* base, size_t count_per_rank);
int ts_exchange_summaries(admm_handle* h);
int ts_exchange_partials(admm_handle* h);
int launch_xscan_mfma(admm_handle* h, bool forward_form = false, bool with_finalize = false);
int launch_xscan(admm_handle* h);
int ensure_zy(admm_handle* h);
int launch_z(admm_handle* h, bool resid);
int launch_finalize(admm_handle* h, int it, int nchunks);
int flush_finalize(admm_handle* h, int it = 0);
IterForm next_form(const admm_handle* h, int remaining);
int enqueue_form(admm_handle* h, IterForm f, bool resid, bool fin_prev);
void after_form(admm_handle* h, IterForm f);
int enqueue_iteration(admm_handle* h, bool resid, bool use_v);
void after_iterations(admm_handle* h, int count);
int ensure_w(admm_handle* h);
int step_x(admm_handle* h);
int capture_iterations(admm_handle* h);
int enqueue_one(admm_handle* h, bool resid, bool use_graph, int remaining, int it_number = 0, bool next_plain = false);

// ---- host <-> device transfers, validation, uploads of the factor, handle lifetime (admm_hostio.hip)
int upload_h2d(admm_handle* h, void* dst, const void* src, size_t bytes);
int download_d2h(admm_handle* h, void* dst, const void* src, size_t bytes);   // ... and back; returns with the copy complete
int upload_tiled(admm_handle* h, const double* src, double* dst, int E, int nr = 0, int nc = 0);   // per-instance operand (batch x N x E) -> tiled layout; nr x nc: row-major source blocks
int upload_transposed(admm_handle* h, const double* src, double* dst, int rows, int nr = 0, int nc = 0);   // nr x nc: the rows are stacks of row-major blocks
int download_transposed(admm_handle* h, const double* src, double* dst, int rows);
bool finite_all(const double* a, size_t cnt);
int validate_options(const admm_options* o);
int validate_problem(const admm_problem* p);
void set_mixed_form(admm_handle* h, bool fp32);
void warn_alt_gate(const admm::Factor& f, double rho, const char* when);
double scan_growth(const admm::Factor& f);
void destroy_graph(admm_handle* h);
void release(admm_handle* h);
int upload_scan_dense(const std::vector<double>& W, int M, int K, double* Wd, int* rows_d);
int upload_factor(admm_handle* h);
int upload_bounds(admm_handle* h, const admm_problem* p);
void keep_shared(admm_handle* h, const admm_problem* p);
bool problem_has_soc(const admm_problem* p);

// ---- per-instance dynamics (admm_pinst_rt.hip)
int pinst_factor(admm_handle* h, bool only_marked = false);
int pinst_fill_rho(admm_handle* h, double rho);
int pinst_upload_dynamics(admm_handle* h, const admm_problem* p, double* Ad, double* Bd, double* Qd, double* Rd, double* QNd);
int pinst_upload_bounds(admm_handle* h, const admm_problem* p);
int pinst_upload(admm_handle* h, const admm_problem* p);
int pinst_alloc_trial(admm_handle* h, bool dynamics);
int pinst_try(admm_handle* h, const double* Ad, const double* Bd, const double* Qd, const double* Rd, const double* QNd,
              const double* rhov, const int* todo, int* n_not_pd, int* n_grown);
int pinst_segments(admm_handle* h);
int setup_pinst(admm_handle* h, const admm_problem* p);

// ---- refactors: background candidates of the adaptive rule, rho changes (admm_rho_update.hip)
admm_problem shared_problem(const admm_handle* h);
bool spec_enabled(const admm_handle* h);
void spec_reap(admm_handle* h, bool all);
std::unique_ptr<SpecFactor> spec_take(admm_handle* h, double rho);
void spec_start(admm_handle* h);
int set_rho_internal(admm_handle* h, double rho_new);

// ---- the stage window of the big per-stage arrays (see admm_handle::wk0)
inline size_t win_row0(const admm_handle* h) { return (size_t)h->wk0 * h->nb; }                       // first stacked row held
inline size_t win_rows(const admm_handle* h) { return (size_t)(h->wk1 - h->wk0) * h->nb; }            // stacked rows held
inline size_t win_bias(const admm_handle* h) { return win_row0(h) * (size_t)h->pitch; }               // elements the state pointers are biased by
inline size_t win_bias_m(const admm_handle* h) { return (size_t)h->wk0 * h->m * (size_t)h->pitch; }   // ... dbuf / mvec
inline bool windowed(const admm_handle* h) { return h->wk0 != 0 || h->wk1 != h->N; }

// ---- one-line helpers
// vform: read the state from h->v (z = clip(v), y = v - z rebuilt in registers)
inline int launch_xb(admm_handle* h, bool vform) { return launch_x(h, admm::XKernel::XB, vform, false); }
inline int launch_xf(admm_handle* h) { return launch_x(h, admm::XKernel::XF, false, false); }
// fused forward rollout + z/dual (+ residual partials per segment); writes v+ into h->v.
// vin: previous state read from h->v, otherwise from h->z / h->y.
inline int launch_xfz(admm_handle* h, bool resid, bool vin) { return launch_x(h, admm::XKernel::XFZ, vin, resid); }
inline bool fused(const admm_handle* h) { return !(h->opt.flags & ADMM_FLAG_UNFUSED); }
inline int chunks_of_iteration(const admm_handle* h) { return fused(h) ? h->S : h->zchunks; }

}  // namespace rt
}  // namespace admm

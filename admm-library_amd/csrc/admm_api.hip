// admm_api.hip -- C ABI of libadmm_hip.so (include/admm_hip.h): set-up, state, iteration, solve, read-out.  The runtime behind it is split over
// admm_launch.hip, admm_hostio.hip, admm_rho_update.hip, admm_pinst_rt.hip, admm_profile.hip (admm_runtime.hpp).  No CPU fallback: without a HIP device
// every compute entry point returns ADMM_ERR_NO_DEVICE.
#include "admm_runtime.hpp"


using namespace admm::rt;

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
  h->wk0 = 0; h->wk1 = p->N;           // stage window of the big arrays: the whole horizon unless this becomes a time shard (below)
  h->pitch = ((p->batch + 63) / 64) * 64;
  h->has_q = p->q != nullptr;
  h->has_soc = problem_has_soc(p);
  if ((h->opt.flags & ADMM_FLAG_ROW_MAJOR) && p->time_varying != 2) {
    release(h);
    return fail(ADMM_ERR_UNSUPPORTED, "ADMM_FLAG_ROW_MAJOR: per-instance dynamics only (time_varying = 2)");
  }
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
  // the stage window of the big arrays: a time shard of several ranks holds its own stages only (admm_runtime.hpp)
  h->wk0 = 0; h->wk1 = h->N;
  if (ts_n > 1 && std::getenv("ADMM_TS_FULL_ARRAYS") == nullptr) { h->wk0 = h->fac.seg_start[h->ts_s0]; h->wk1 = h->fac.seg_start[h->ts_s0 + h->ts_sl]; }
  const size_t Lw = win_rows(h), Nmw = (size_t)(h->wk1 - h->wk0) * h->m;
  // (dalloc into the field, then bias it: release() undoes the bias before hipFree)
#define DALLOC_WIN(field, count, bias) do { TRY_RELEASE(dalloc(&(field), (count))); (field) -= (bias); } while (0)
  DALLOC_WIN(h->w, Lw * P, win_bias(h));
  DALLOC_WIN(h->z, Lw * P, win_bias(h));
  DALLOC_WIN(h->y, Lw * P, win_bias(h));
  DALLOC_WIN(h->v, Lw * P, win_bias(h));
  if (h->has_q) DALLOC_WIN(h->q, Lw * P, win_bias(h));
  DALLOC_WIN(h->dbuf, Nmw * P, win_bias_m(h));
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
    DALLOC_WIN(h->mvec, Nmw * P, win_bias_m(h));               // db rows of the forward elimination
    HIP_TRY_RELEASE(hipMemsetAsync(h->mvec + win_bias_m(h), 0, sizeof(double) * Nmw * P, h->stream));
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
  h->stage_rows = Lw;
  TRY_RELEASE(dalloc(&h->stage, Lw * (size_t)h->batch));
  HIP_TRY_RELEASE(hipHostMalloc((void**)&h->h_nconv, sizeof(int), hipHostMallocDefault));

  HIP_TRY_RELEASE(hipMemsetAsync(h->w + win_bias(h), 0, sizeof(double) * Lw * P, h->stream));
  HIP_TRY_RELEASE(hipMemsetAsync(h->z + win_bias(h), 0, sizeof(double) * Lw * P, h->stream));
  HIP_TRY_RELEASE(hipMemsetAsync(h->y + win_bias(h), 0, sizeof(double) * Lw * P, h->stream));
  HIP_TRY_RELEASE(hipMemsetAsync(h->v + win_bias(h), 0, sizeof(double) * Lw * P, h->stream));
  HIP_TRY_RELEASE(hipMemsetAsync(h->dbuf + win_bias_m(h), 0, sizeof(double) * Nmw * P, h->stream));
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
#undef DALLOC_WIN
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

/* admm_hip.h -- C ABI of libadmm_hip.so: batched ADMM for box-constrained
 * optimal-control QPs on MI355X (gfx950).
 *
 * Reference interface replaced: NONE EXISTS.  The reference snapshot is
 * /root/reference/README.md:1-2 ("Implementation of Alternating Direction
 * Method of Multipliers for astrodynamics problems") plus a LICENSE; it defines
 * no MATLAB entry point, MEX gateway or FFI for this path.
 * Every entry point below is therefore this repository's own specification of
 * the "problem-setup / solver entry-point surface" BASELINE.json's north_star
 * asks for; INTEGRATION.md shows the MEX and ctypes bindings over it.
 *
 * Conventions
 *   - Plain C, no C++ or torch types cross this boundary.
 *   - All real arrays are fp64.  Matrices are column-major (MATLAB-native).
 *   - Per-QP vectors are "L x batch column-major": QP b occupies
 *     [b*L, (b+1)*L).  The stacked variable of one QP is
 *       w = (u_0, x_1, u_1, x_2, ..., u_{N-1}, x_N),  L = N*(n+m),
 *     block k = (u_k, x_{k+1}).
 *   - Every function returns an admm_status (0 = ok); admm_last_error() gives
 *     a thread-local message for the last non-zero return.  No exceptions, no
 *     exit().
 *   - The library never keeps caller pointers after a call returns.
 *   - A handle is bound to one GPU and is not thread-safe; distinct handles are.
 */
#ifndef ADMM_HIP_H
#define ADMM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADMM_HIP_ABI_VERSION 8

typedef enum admm_status {
  ADMM_OK = 0,
  ADMM_ERR_INVALID = 1,      /* bad argument / problem data (NaN, lo > hi, sizes) */
  ADMM_ERR_UNSUPPORTED = 2,  /* (n, m) outside the compiled kernel set, etc. */
  ADMM_ERR_NO_DEVICE = 3,    /* no HIP device: the product path has no CPU fallback */
  ADMM_ERR_HIP = 4,          /* a HIP runtime call failed */
  ADMM_ERR_NUMERIC = 5,      /* factorisation failed (S_k not SPD) */
  ADMM_ERR_ALLOC = 6
} admm_status;

/* minimise  1/2 sum_k [u_k' R u_k + x_{k+1}' Q_{k+1} x_{k+1}] + q' w
 * s.t.      x_{k+1} = A_k x_k + B_k u_k,  x_0 given,   lo <= w <= hi   [and/or ||u_k||_2 <= unorm_k]
 * Dynamics, weights and the box are shared by the whole batch; x0 and q are
 * per instance.  (time_varying = 2 / stage_bounds = 2: dynamics / box per instance as well; weights stay shared.) */
typedef struct admm_problem {
  int32_t N;              /* horizon (stages) */
  int32_t n;              /* state dimension */
  int32_t m;              /* control dimension */
  int32_t batch;          /* number of independent QPs */
  int32_t time_varying;   /* 0: A is n*n, B is n*m;  1: A is n*n*N, B is n*m*N;
                             2: PER-INSTANCE dynamics, A is n*n*N*batch, B is n*m*N*batch (QP b's stages are contiguous):
                                the KKT system of every QP is factored on the device and the sweeps stream their stage
                                operators from HBM per QP (the (n, m) pairs of csrc/admm_pinst.hip and
                                admm_pinst_g1.hip, n <= 6; precision_mode FP64 only) */
  int32_t stage_bounds;   /* 0: lo/hi are (m+n);     1: lo/hi are (m+n)*N;   2 (with time_varying = 2 only): per
                             instance, lo/hi are (m+n)*N*batch */
  const double* A;
  const double* B;
  const double* Q;        /* n*n symmetric PSD */
  const double* R;        /* m*m symmetric PSD (R + rho I must be PD) */
  const double* QN;       /* n*n symmetric PSD, terminal weight */
  const double* x0;       /* n*batch */
  const double* lo;       /* block order (u then x); -inf allowed */
  const double* hi;       /* +inf allowed */
  const double* q;        /* L*batch linear cost, or NULL for q = 0 */
  /* Optional thrust-magnitude (second-order-cone) constraint ||u_k||_2 <= unorm_k on the control
   * rows, replacing their box: NULL = off; otherwise 1 entry (stage_bounds = 0) or N entries
   * (stage_bounds = 1), +inf = no constraint at that stage.  Where it is finite the box of the
   * control rows must be (-inf, +inf); the state rows keep their box. */
  const double* unorm;
} admm_problem;

typedef struct admm_options {
  double rho;             /* > 0 */
  double alpha;           /* over-relaxation in (0, 2); 1 = none */
  double eps_abs;
  double eps_rel;
  int32_t max_iter;
  int32_t check_interval; /* residuals + stop test every this many iterations */
  int32_t segments;       /* parallel-in-time segments of the x-update; 0 = auto (per-instance dynamics: at most 64) */
  int32_t device;         /* HIP device ordinal; -1 = current device */
  int32_t zrows;          /* rows per workgroup chunk in the z/dual kernel; 0 = auto */
  int32_t flags;          /* ADMM_FLAG_* */
  /* Adaptive rho by residual balancing (Boyd et al. 2011 §3.4.1), batch-level because the KKT
   * factor is shared by the batch.  At a checked iteration it with it % adapt_interval == 0, with
   * R = sum r_b^2 and S = sum s_b^2 over the QPs that have not converged yet:
   *   R > adapt_mu^2 S  ->  rho *= adapt_tau;     S > adapt_mu^2 R  ->  rho /= adapt_tau
   * then the scaled dual is rescaled (y *= rho_old / rho_new), the KKT system is refactored on
   * the host and the records re-uploaded.  adapt_interval = 0 disables (default).
   * With per-instance dynamics (time_varying = 2) every QP has its own factor, and the rule runs PER QP, on the
   * device (ABI v5): QP b compares R = r_b^2 with S = s_b^2, its rho, dual rescale and refactor are its own, adapt_max
   * counts its changes; admm_info.rho is then the largest rho in force, admm_info.rho_updates the number of changes
   * summed over the batch, and admm_get_rho returns every QP's rho. */
  int32_t adapt_interval; /* iterations between adaptation tests; 0 = fixed rho; must be a multiple of check_interval */
  int32_t adapt_max;      /* at most this many rho changes per admm_solve */
  double adapt_mu;        /* > 1 */
  double adapt_tau;       /* > 1 */
  /* Arithmetic of the x-update (ABI v4; BASELINE.json configs[4]).  ADMM_PRECISION_*:
   *   FP64       fp64 throughout; the library picks the kernel form: one lane per QP with fp64 vector FMAs (every
   *              compiled (n, m), q, thrust-magnitude bound), or FP64_MFMA's kernels where those are faster -- the
   *              smallest batches (ADMM_FLAG_NO_MFMA).
   *   FP64_MFMA  the fused stage operators as chains of v_mfma_f64_16x16x4_f64 over 16-QP panels, always: the same
   *              fp64 iteration (iterates equal to the one-lane kernels' to rounding).
   *   MIXED      as FP64_MFMA, but the two products of the Riccati form (forward rollout, backward elimination:
   *              operators O(1)) run in fp32 on v_mfma_f32_16x16x4_f32, at half the matrix-pipe cycles; the two of
   *              the forward-elimination form (gains up to 2.5e4) stay fp64, as do the state v, the z-update, the
   *              dual and the residuals.  Iterates then carry ~1e-6 relative error.  admm_solve refines in fp64: it
   *              iterates in this form until the stopping rule holds with eps_abs, eps_rel raised to at least 1e-4,
   *              then continues with the FP64_MFMA kernels until it holds as given (admm_info.mixed_iters = length
   *              of the first phase).  admm_run / admm_iterate always run the mixed form.
   * The two MFMA forms exist for the (n, m) pairs of csrc/admm_mfma.hip, without a thrust-magnitude bound and without q
   * (except FP64 / FP64_MFMA at (6, 3) with a batch of up to 128 QPs, where q is supported): ADMM_ERR_UNSUPPORTED otherwise. */
  int32_t precision_mode;
  int32_t reserved;       /* must be 0 */
} admm_options;

#define ADMM_PRECISION_FP64 0
#define ADMM_PRECISION_MIXED 1
#define ADMM_PRECISION_FP64_MFMA 2

#define ADMM_FLAG_NONE 0
#define ADMM_FLAG_NO_GRAPH 1   /* accepted, no effect: direct launches are the default (see ADMM_FLAG_GRAPH) */
#define ADMM_FLAG_GRAPH 16     /* replay captured hipGraphs (one per iteration form) instead of launching
                                  the 2-4 kernels of an iteration directly.  Off by default: on ROCm 7.2 /
                                  MI355X a graph launch per iteration costs ~5 us MORE than the direct
                                  launches it replaces, at every problem size measured */
#define ADMM_FLAG_SCAN_CHAIN 4 /* segment scan as the sequential per-QP chain (xscan_kernel) instead of
                                  the fp64-MFMA GEMM form (xscan_mfma_kernel) */
#define ADMM_FLAG_UNFUSED 2    /* iterate with separate forward-rollout and z/dual kernels (w stored
                                  every iteration) instead of the fused xfz kernel */

#define ADMM_FLAG_NO_MFMA 32    /* ADMM_PRECISION_FP64 only: always the one-lane-per-QP kernels.  By default FP64 takes the
                                  fp64 MFMA form of the fused kernels (same iteration, iterates equal to rounding;
                                  DESIGN.md §4.9) where it is compiled for (n, m), the problem has no q and no
                                  thrust-magnitude bound, and it is the faster of the two -- measured (tools/family_time.py):
                                  batches of at most 64 QPs, or at most 128 with n >= 9 (one wave per segment runs a chain
                                  of ~15 MFMAs per stage instead of a few hundred dependent vector FMAs).  Larger batches
                                  run the one-lane kernels, whose stage operators are distributed over the lanes of a row
                                  and applied with v_fmac_f64_dpp (round 3): faster than the MFMA form at every shape
                                  from 256 QPs on, n = 12 included */

#define ADMM_FLAG_HISTORY 64     /* admm_solve records, at every stopping test, (iteration, converged QPs, max primal residual, max dual
                                   residual, rho in force) for admm_get_history; costs one read-back of the per-QP residuals per test */
#define ADMM_FLAG_ROW_MAJOR 128   /* per-instance dynamics only (time_varying = 2; ABI v8): every n x n block of A and n x m block of B is
                                   ROW-major (C / NumPy order) instead of column-major.  The caller's arrays are then uploaded as they
                                   are and the blocks are transposed on the device on the way into the library's layout -- a NumPy or C
                                   caller otherwise transposes 7 GB on the host per admm_update_problem at n = 12, 4096 x 1000.  Fixed at
                                   admm_setup: admm_update_problem of the handle reads A, B the same way.  Q, R, QN are symmetric;
                                   ADMM_ERR_UNSUPPORTED with batch-shared dynamics (their small A, B are cheap to transpose). */
#define ADMM_FLAG_NO_ALTERNATE 8 /* always eliminate backward / substitute forward (xb + xfz kernels); by
                                  default (unless the forward form fails its host check for the problem),
                                  consecutive iterations alternate the elimination direction so that each
                                  substitution sweep is fused with the next elimination sweep */

typedef struct admm_info {
  int32_t iters_run;      /* batch iterations executed by the last admm_solve */
  int32_t n_converged;    /* QPs that met the stopping rule */
  double  max_r;          /* max over the batch of the last checked primal residual */
  double  max_s;          /* ... dual residual */
  double  solve_ms;       /* wall time of the last admm_solve (host clock) */
  double  rho;            /* rho in force at the end of the solve */
  int32_t rho_updates;    /* rho changes made by the adaptive rule during the solve */
  int32_t mixed_iters;    /* ADMM_PRECISION_MIXED: iterations run in fp32 before the fp64 refinement phase (else 0) */
} admm_info;

typedef struct admm_handle admm_handle;

/* Defaults: rho 0.1, alpha 1, eps 1e-6/1e-6, max_iter 4000, check_interval 10, fixed rho
 * (adapt_interval 0, adapt_max 16, adapt_mu 10, adapt_tau 2). */
void admm_default_options(admm_options* o);

/* Validate, factor the x-update's KKT system on the host (fp64 Riccati sweep),
 * allocate device memory, upload.  State starts at z = y = w = 0. */
int admm_setup(admm_handle** out, const admm_problem* p, const admm_options* o);

/* Replace the per-instance data (x0: n*batch, q: L*batch or NULL = keep) of an
 * existing handle without refactoring. */
int admm_update_instances(admm_handle* h, const double* x0, const double* q);

/* Change rho on an existing handle: refactors the KKT system on the host, re-uploads the
 * records and rescales the scaled dual (y *= rho_old / rho_new) so that the unscaled multiplier
 * rho y is unchanged.  If the handle's segment count was chosen by admm_setup (options.segments = 0), the
 * conditioning bound applied there (largest entry of the segment-scan matrix <= 100) is re-checked on the new
 * factor: a rho that breaks it is refused with ADMM_ERR_NUMERIC and the handle is unchanged (the segment count
 * of a live handle is frozen).  The adaptive rule of admm_solve treats such a refusal as "keep rho and stop
 * adapting". */
int admm_set_rho(admm_handle* h, double rho);

/* New shared problem data on an existing handle: dynamics, weights, box, thrust-magnitude bounds, x0 and
 * q of *p replace those of admm_setup (same N, n, m, batch; q present iff it was; a thrust-magnitude
 * bound present iff one was).  The KKT system is refactored with the current rho and segment count and
 * everything re-uploaded; device buffers, the state (kept as the (z, y) pair) and per-QP results are left
 * alone.  This is what a successive-convexification caller does between outer iterations instead of
 * admm_free + admm_setup.  The conditioning bound of admm_set_rho applies here too.  On failure the handle is
 * unchanged. */
int admm_update_problem(admm_handle* h, const admm_problem* p);

/* Warm start / test hook: overwrite device state.  Any pointer may be NULL
 * (left unchanged).  Each is L*batch and must be finite (ADMM_ERR_INVALID otherwise, nothing uploaded). */
int admm_set_state(admm_handle* h, const double* w, const double* z, const double* y);

/* Run until every QP met the rule at a checked iteration, or max_iter.
 * z0/y0 (L*batch) may be NULL: continue from the handle's current state. */
int admm_solve(admm_handle* h, const double* z0, const double* y0, admm_info* info);

/* admm_solve in pieces, for callers that need a decision between checks -- in particular a
 * multi-GPU solve, where each rank owns a shard of the batch and the stop decision (and the
 * adaptive-rho sums) must be global:
 *   admm_solve_begin(h, z0, y0);
 *   loop: admm_solve_step(h, &it, &nconv, &R, &S);       // runs up to and including the next check
 *         [all-reduce  batch - nconv  (SUM),  R, S (SUM)  over the ranks]
 *         stop if nothing is left unconverged or it >= max_iter;
 *         admm_solve_adapt(h, R_global, S_global, &changed);   // no-op unless an adaptation is due
 *   admm_solve_end(h, &info);
 * admm_solve is exactly this loop with local values.  R, S = sums of r^2, s^2 over the QPs that
 * have not converged (NULL to skip the read-back). */
int admm_solve_begin(admm_handle* h, const double* z0, const double* y0);
int admm_solve_step(admm_handle* h, int32_t* iters_done, int32_t* n_converged, double* R, double* S);
int admm_solve_adapt(admm_handle* h, double R, double S, int32_t* changed);
int admm_solve_end(admm_handle* h, admm_info* info);

/* Run exactly `iters` iterations, no residuals, no stop test (benchmark and
 * iterate-parity path).  Asynchronous on the handle's stream; admm_sync waits. */
int admm_iterate(admm_handle* h, int32_t iters);
/* As admm_iterate, but every `residual_every`-th iteration (0 = never) runs the
 * residual-evaluating form of the fused z kernel plus the residual finalise
 * kernel on the device; still no host synchronisation and no early exit.
 * residual_every = 1 is the benchmark's "step": x-update + fused
 * z/dual/residual-partials + finalise, every iteration. */
int admm_run(admm_handle* h, int32_t iters, int32_t residual_every);
int admm_sync(admm_handle* h);

/* Single steps, for kernel-level parity tests. */
int admm_step_x(admm_handle* h);                    /* w <- x-update(z, y) */
int admm_step_z(admm_handle* h, int32_t residuals); /* (z, y) <- z/dual update; residual partials if != 0 */

/* Residual norms of the last residual-evaluating z-step: each array is `batch`
 * long and may be NULL.  r = |w - z+|, s = rho |z+ - z|, nw = |w|, nz = |z+|,
 * ny = rho |y+|. */
int admm_get_residuals(admm_handle* h, double* r, double* s, double* nw, double* nz, double* ny);

/* Copy out the state; any pointer may be NULL.  Each is L*batch. */
int admm_get(admm_handle* h, double* w, double* z, double* y);

/* Per-QP results of the last admm_solve: first checked iteration at which the
 * rule held (max_iter if never), status (1 converged / 0 not), last r and s. */
int admm_get_info(admm_handle* h, int32_t* iters, int32_t* status, double* r, double* s);

/* Timing hook for bench.py: runs `iters` iterations with HIP events recorded
 * on the handle's stream around each kernel launch and returns average
 * durations in ms.  fused_path == 1 (the plain fused path, ADMM_FLAG_NO_ALTERNATE):
 *   ms = {xb, xscan, xfz, 0, finalise, whole iteration}
 * fused_path == 0 (ADMM_FLAG_UNFUSED path):
 *   ms = {xb, xscan, xf, zdual, finalise, whole iteration}
 * fused_path == 2 (the default alternating-direction path; ADMM_ERR_UNSUPPORTED if the
 * handle does not run it): `iters` PAIRS of iterations,
 *   ms = {xscan, xfze, xscan (+ finalise role), xbze, 0, whole pair}
 * fused_path == 3 (the same path, BACK-TO-BACK cross-check: no event between launches): `iters` consecutive
 * (xfze, xbze) pairs without the scans, then each scan form `iters` times in a row,
 *   ms = {xscan (W), mean of xfze and xbze, xscan (WB) (+ finalise role), the same mean, 0, sum of the four}
 * (not ADMM iterates: the state is parked, restored, and ONE plain iteration is applied to leave the handle consistent).
 * xb = x-update backward sweep, xscan = segment scan, xf = forward rollout,
 * zdual = standalone fused z/dual/residual kernel, xfz = forward rollout fused
 * with z/dual/residual.  `residuals` selects the residual-evaluating kernel
 * forms (+ the finalise kernel). */
int admm_profile(admm_handle* h, int32_t iters, int32_t residuals, int32_t fused_path, double ms[6]);

/* Residual history of the last admm_solve (ADMM_FLAG_HISTORY; ABI v5): one entry per stopping test, oldest first.  *count =
 * number of entries recorded; at most `capacity` of them are copied into each non-NULL array.  max_r / max_s are maxima over
 * the real QPs of the batch (converged or not) of the residuals of that iteration; rho is the one in force during it (the
 * largest one with per-instance dynamics). */
int admm_get_history(admm_handle* h, int32_t capacity, int32_t* count, int32_t* iteration, int32_t* n_converged,
                     double* max_r, double* max_s, double* rho);

/* rho of every QP (batch entries): the handle's rho for batch-shared dynamics, each QP's own with per-instance
 * dynamics (where the adaptive rule moves them apart; admm_set_rho sets them all).  ABI v5. */
int admm_get_rho(admm_handle* h, double* rho);

/* Geometry chosen at setup, for roofline accounting: pitch = padded batch,
 * segs = x-update segments, zrows = rows per z-kernel chunk, zchunks. */
int admm_get_geometry(admm_handle* h, int32_t* pitch, int32_t* segs, int32_t* zrows, int32_t* zchunks);

/* Which kernels the handle runs, and the measured margin of the default path (ABI v6).  The default iteration alternates
 * the elimination direction; its forward-elimination factor carries large early-stage gains, so admm_setup
 * (and every refactor: admm_set_rho, the adaptive rule, admm_update_problem) verifies it on the host -- one x-update with a
 * random linear term through both forms -- and falls back to the plain fused path (29.33 instead of 21.33 B per stacked
 * element and iteration at n = 6, m = 3) when the relative mismatch alt_check exceeds alt_gate.  That fallback is reported
 * here and through admm_last_warning(), never silently. */
typedef struct admm_path_info {
  int32_t alternating;     /* 1: the alternating-direction fused kernels run; 0: the plain fused (or unfused) path */
  int32_t alt_requested;   /* 1: options / compiled kernels allow alternation for this problem class (0: ADMM_FLAG_NO_ALTERNATE,
                              _UNFUSED, _SCAN_CHAIN, per-instance dynamics, (n, m) without the fused kernels) */
  int32_t mfma;            /* kernel family of the fused sweeps: 0 one lane per QP (fp64 VALU), 1 MIXED, 2 fp64 MFMA */
  int32_t xfree;           /* 1: every state row is unbounded at every stage: iterations without residuals neither read nor
                              write v of those rows (XFREE forms) */
  int32_t segments;        /* parallel-in-time segments of the x-update */
  int32_t auto_segments;   /* 1: chosen by admm_setup (and guarded by the conditioning bound on every refactor) */
  int32_t scan_form;       /* segment scan: 0 fp64-MFMA GEMM, 1 matrix-vector (batches <= 4), 2 sequential chain, 3 per-QP (per-instance) */
  int32_t per_instance;    /* 1: time_varying = 2 (device factor, operands per QP from HBM) */
  double  alt_check;       /* relative mismatch of the forward-elimination form on the host verification vector; -1: not run
                              (alternation not requested, or the form could not be built: a singular A_k or covariance) */
  double  alt_gate;        /* the bound alt_check must meet (5e-12) */
  double  scan_growth;     /* largest |entry| of the segment-scan matrices (conditioning bound: <= 100 with automatic segments) */
} admm_path_info;
int admm_get_path(admm_handle* h, admm_path_info* info);

/* Thread-local diagnostic of the last admm_setup / admm_set_rho / admm_update_problem / admm_solve* call on this thread that
 * SUCCEEDED but changed the kernels a handle runs (the forward-elimination gate above; the adaptive rule refused a rho); ""
 * if there was nothing to report.  Cleared at the start of each of those calls. */
const char* admm_last_warning(void);

/* TIME-SHARDED handles: one batch of QPs whose horizon is cut into segments that live on DIFFERENT ranks (ABI v7).
 * The parallel-in-time x-update couples its segments EXACTLY through 2 n numbers per segment and QP (the segment summaries the
 * elimination sweeps leave, and the boundary values the scan returns) -- so the "shooting segments" of one horizon can sit on
 * different GPUs and exchange just those: every rank factorises the whole horizon on the host, runs the fused kernels of its own
 * S / nranks consecutive segments, and before each segment scan the summaries are completed by an ALL-GATHER over the ranks (the
 * only data-path collective of this library; the residual partial sums of the segments travel the same way before each stopping
 * test, so every rank takes the same decisions).  The iterates are those of one handle holding every segment (up to the rounding
 * of the scan product).
 *
 * The transport is the caller's: `exchange` is called from inside admm_run / admm_iterate / admm_solve* at those points, with
 *   op = ADMM_EXCHANGE_ALLGATHER: `buf` holds nranks slices of `count` doubles in rank order, this rank's slice (at
 *        buf + rank * count) is complete; when the function's work on `hip_stream` is done every slice must be.
 * buf is DEVICE memory.  The function either enqueues the collective on hip_stream (RCCL: nothing waits on the host) or
 * synchronises the stream and moves the data itself (any other transport); it returns 0 on success (anything else fails the
 * calling entry point with ADMM_ERR_HIP).
 *
 * admm_setup_timeshard takes the GLOBAL problem on every rank.  options.segments = total segment count (0 = automatic), a
 * multiple of nranks.  Shared dynamics only, fused paths only (no ADMM_FLAG_UNFUSED / _GRAPH / _SCAN_CHAIN).  A rank ALLOCATES
 * the state (w, z, y, v, q and the feed-forward rows) of its own stages only: admm_set_state / admm_solve(z0, y0) /
 * admm_update_instances(q) take the usual L x batch arrays and use the rows [stage_lo, stage_hi) * (n + m) of every QP,
 * admm_get writes those rows of the caller's arrays and leaves the others alone; the host side assembles the windows
 * (admm_get_window). */
#define ADMM_EXCHANGE_ALLGATHER 0
typedef int (*admm_exchange_fn)(void* ctx, void* hip_stream, int32_t op, double* buf, int64_t count);
int admm_setup_timeshard(admm_handle** out, const admm_problem* p, const admm_options* o, int32_t rank, int32_t nranks,
                         admm_exchange_fn exchange, void* ctx);
/* The rank's window: stages [stage_lo, stage_hi), segments [seg_lo, seg_lo + segs_local) of segs_total.  An ordinary handle
 * reports the whole horizon and rank 0 of 1. */
int admm_get_window(admm_handle* h, int32_t* stage_lo, int32_t* stage_hi, int32_t* seg_lo, int32_t* segs_local, int32_t* segs_total);

void admm_free(admm_handle* h);

/* Host-only helpers (no GPU needed): the fp64 pre-factorisation exactly as
 * admm_setup uploads it, exposed so that the host logic can be tested on a
 * CPU-only machine.  admm_record_sizes gives the per-stage / per-segment record
 * lengths in doubles; admm_host_factor fills K (N*m*n), Sinv (N*m*m), both
 * row-major, the packed backward / forward / segment records (N*rb, N*rf,
 * segments*rs doubles; layouts in csrc/admm_factor.hpp) and seg_start
 * (segments + 1).  Any output pointer may be NULL. */
int admm_record_sizes(int32_t n, int32_t m, int32_t* rb, int32_t* rf, int32_t* rs);
int admm_host_factor(const admm_problem* p, double rho, int32_t segments, double* K, double* Sinv,
                     double* recB, double* recF, double* recS, int32_t* seg_start);
/* The dense segment-scan matrix W (row-major M x K) of the MFMA scan kernel:
 *   [t_in(0..S-1) | pad to Mt | x_in(0..S-1) | pad] = W [tseg(0..S-1) | x0 | eseg(0..S-1) | pad]
 * Call with W = NULL to query M, Mt, K first. */
int admm_host_scan_matrix(const admm_problem* p, double rho, int32_t segments, double* W, int32_t* M,
                          int32_t* Mt, int32_t* K);

/* The same two dense scan matrices (W, and WB if *ok) with their INPUT columns in the rank-by-rank layout of time-sharded handles
 * (admm_setup_timeshard): [rank 0: tseg of its segments | eseg of them][rank 1: ...] ... | x0 | pad.  Shapes as above. */
int admm_host_scan_matrices_timeshard(const admm_problem* p, double rho, int32_t segments, int32_t nranks, double* W, double* WB,
                                      int32_t* ok);

/* Alternating-direction iteration: per-stage records of the two fused kernels
 * (rfe / rbe doubles per stage; layouts in csrc/admm_layout.hpp) and the dense scan matrix WB of
 * the forward-elimination form (same M x K shape and row / column layout as W above:
 *   [m_in(0..S-1) | pad to Mt | lam_in(0..S-1) | pad] = WB [mseg(0..S-1) | x0 | epsseg(0..S-1) | pad]).
 * *ok = 0 if that form could not be built for this problem (outputs are then untouched).
 * Any output pointer may be NULL. */
int admm_record_sizes_alt(int32_t n, int32_t m, int32_t* rfe, int32_t* rbe);
int admm_host_factor_alt(const admm_problem* p, double rho, int32_t segments, double* recFE, double* recBE,
                         double* WB, int32_t* ok);

/* MFMA form (csrc/admm_mfma_layout.hpp): bytes per stage of the forward / backward fragment
 * records of mode ADMM_PRECISION_MIXED or ADMM_PRECISION_FP64_MFMA, and the records themselves (N * bytes each;
 * either pointer may be NULL).  *alt_ok = 0: the ELIM_F / SUB_B fragments are zero (no forward-elimination form). */
int admm_mfma_record_bytes(int32_t n, int32_t m, int32_t mode, int32_t* fwd, int32_t* bwd);
int admm_host_factor_mfma(const admm_problem* p, double rho, int32_t segments, int32_t mode, void* recMF,
                          void* recMB, int32_t* alt_ok);

const char* admm_last_error(void);
int admm_abi_version(void);
/* Number of HIP devices visible (0 if none / no driver). */
int admm_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* ADMM_HIP_H */

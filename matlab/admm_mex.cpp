// admm_mex.cpp -- MEX gateway: marshals mxArray <-> the C ABI of libadmm_hip.so
// (include/admm_hip.h).  It contains no arithmetic.
//
// STATUS: written against the documented MEX C API (mex.h / matrix.h), compile-
// checked only against a minimal declaration shim (tests/mex_shim/mex.h); it has
// NEVER been linked against libmx/libmex or executed: no MATLAB, Octave, mex or
// mex.h exists in the build image or on the GPU box.  The
// reference defines no MEX interface to mirror (README.md:1-2 only), so the
// command set below is this repository's own.
//
// Build (on a machine with MATLAB + ROCm):
//   mex -I../include admm_mex.cpp -L../admm-library_amd -ladmm_hip
//
// Usage from MATLAB (see admm_setup.m / admm_solve.m / admm_get.m / admm_free.m):
//   h    = admm_mex('setup', problem, options)   % uint64 scalar handle
//   info = admm_mex('solve', h, z0, y0)          % z0, y0 optional / []
//   [w, z, y] = admm_mex('get', h)
//   admm_mex('iterate', h, iters)
//   p    = admm_mex('path', h)                   % kernels in use, margin of the default path (admm_get_path)
//   admm_mex('free', h)
//
// MATLAB arrays are column-major doubles, exactly the ABI's layout: A (n x n or
// n x n x N), B (n x m [x N]), x0 (n x batch), q (L x batch), lo/hi ((m+n) x 1 or
// (m+n) x N) are passed by pointer, no copies.
#include <cstdint>
#include <cstring>
#include <vector>

#include "mex.h"

#include "admm_hip.h"

namespace {

void fail(const char* id, const char* msg) { mexErrMsgIdAndTxt(id, "%s", msg); }

void check(int rc) {
  if (rc != ADMM_OK) mexErrMsgIdAndTxt("admm:library", "libadmm_hip error %d: %s", rc, admm_last_error());
  // a call that succeeded but changed the kernels the handle runs says so (ABI v6): pass it on as a MATLAB warning
  const char* w = admm_last_warning();
  if (w && w[0]) mexWarnMsgIdAndTxt("admm:path", "%s", w);
}

const mxArray* field(const mxArray* s, const char* name, bool required) {
  const mxArray* f = mxGetField(s, 0, name);
  if (!f && required) mexErrMsgIdAndTxt("admm:input", "problem field '%s' is missing", name);
  return f;
}

const double* dbl(const mxArray* a, const char* name) {
  if (!a || mxIsEmpty(a)) return nullptr;
  if (!mxIsDouble(a) || mxIsComplex(a) || mxIsSparse(a))
    mexErrMsgIdAndTxt("admm:input", "'%s' must be a full real double array", name);
  return mxGetPr(a);
}

double scalar_or(const mxArray* s, const char* name, double dflt) {
  const mxArray* f = s ? mxGetField(s, 0, name) : nullptr;
  return (f && !mxIsEmpty(f)) ? mxGetScalar(f) : dflt;
}

admm_handle* handle_of(const mxArray* a) {
  if (!a || !mxIsUint64(a) || mxGetNumberOfElements(a) != 1) fail("admm:input", "handle must be a uint64 scalar");
  admm_handle* h = reinterpret_cast<admm_handle*>(static_cast<uintptr_t>(*static_cast<const uint64_t*>(mxGetData(a))));
  if (!h) fail("admm:input", "handle is null (already freed?)");
  return h;
}

// sizes remembered per handle so that 'get' can size its outputs.  Unbounded (a vector, freed slots are
// reused); every handle still alive when MATLAB clears the MEX file or exits is released by at_exit().
struct Dims { admm_handle* h; int L, batch; };
std::vector<Dims> g_dims;
bool g_at_exit_registered = false;

void at_exit() {
  for (Dims& d : g_dims)
    if (d.h) { admm_free(d.h); d.h = nullptr; }
  g_dims.clear();
}

void remember(admm_handle* h, int L, int batch) {
  if (!g_at_exit_registered) { mexAtExit(at_exit); g_at_exit_registered = true; }
  for (Dims& d : g_dims) if (!d.h) { d = {h, L, batch}; return; }
  g_dims.push_back({h, L, batch});
}
Dims* lookup(admm_handle* h) {
  for (Dims& d : g_dims) if (d.h == h) return &d;
  fail("admm:input", "unknown handle");
  return nullptr;
}

// MATLAB problem struct -> admm_problem (pointers into the struct's arrays; valid while P is)
void parse_problem(const mxArray* P, admm_problem& p) {
  const mxArray *A = field(P, "A", true), *B = field(P, "B", true), *x0 = field(P, "x0", true);
  const mxArray *lo = field(P, "lo", true), *hi = field(P, "hi", true), *q = field(P, "q", false);
  std::memset(&p, 0, sizeof p);
  p.N = static_cast<int32_t>(mxGetScalar(field(P, "N", true)));
  p.n = static_cast<int32_t>(mxGetM(B));
  const mwSize* bd = mxGetDimensions(B);
  p.m = static_cast<int32_t>(bd[1]);
  p.batch = static_cast<int32_t>(mxGetN(x0));
  // per_instance (logical field, optional): A is n x n x N x batch, B n x m x N x batch (time_varying = 2); with
  // per_instance_bounds lo / hi are (m+n) x N x batch (stage_bounds = 2).  Explicit flags, because MATLAB drops
  // trailing singleton dimensions (an n x n x N x 1 array IS 3-D).
  const mxArray* pi = field(P, "per_instance", false);
  const mxArray* pib = field(P, "per_instance_bounds", false);
  const bool per_inst = pi && !mxIsEmpty(pi) && mxGetScalar(pi) != 0.0;
  const bool per_inst_b = per_inst && pib && !mxIsEmpty(pib) && mxGetScalar(pib) != 0.0;
  p.time_varying = per_inst ? 2 : (mxGetNumberOfDimensions(A) == 3 ? 1 : 0);
  p.stage_bounds = per_inst_b ? 2 : (mxGetN(lo) > 1 ? 1 : 0);
  if (static_cast<int32_t>(mxGetM(x0)) != p.n) fail("admm:input", "x0 must be n x batch");
  if (static_cast<int32_t>(mxGetM(lo)) != p.n + p.m || static_cast<int32_t>(mxGetM(hi)) != p.n + p.m)
    fail("admm:input", "lo/hi must have m+n rows (u block, then x block)");
  if (p.stage_bounds == 1 && static_cast<int32_t>(mxGetN(lo)) != p.N) fail("admm:input", "per-stage bounds need N columns");
  if (p.stage_bounds == 2 && mxGetNumberOfElements(lo) != static_cast<size_t>(p.n + p.m) * p.N * p.batch)
    fail("admm:input", "per-instance bounds must be (m+n) x N x batch");
  if (p.time_varying == 1 && static_cast<int32_t>(mxGetDimensions(A)[2]) != p.N) fail("admm:input", "time-varying A needs N pages");
  if (p.time_varying == 2 && (mxGetNumberOfElements(A) != static_cast<size_t>(p.n) * p.n * p.N * p.batch ||
                              mxGetNumberOfElements(B) != static_cast<size_t>(p.n) * p.m * p.N * p.batch))
    fail("admm:input", "per-instance A / B must be n x n x N x batch / n x m x N x batch");
  p.A = dbl(A, "A"); p.B = dbl(B, "B");
  p.Q = dbl(field(P, "Q", true), "Q"); p.R = dbl(field(P, "R", true), "R"); p.QN = dbl(field(P, "QN", true), "QN");
  p.x0 = dbl(x0, "x0"); p.lo = dbl(lo, "lo"); p.hi = dbl(hi, "hi");
  p.q = dbl(q, "q");
  const mxArray* un = field(P, "unorm", false);     // thrust-magnitude bound, optional: 1 or N entries
  p.unorm = dbl(un, "unorm");
  if (p.unorm && static_cast<int32_t>(mxGetNumberOfElements(un)) != (p.stage_bounds ? p.N : 1))
    fail("admm:input", "unorm must have 1 entry, or N entries together with per-stage bounds");
  const int L = p.N * (p.n + p.m);
  if (p.q && (static_cast<int>(mxGetM(q)) != L || static_cast<int32_t>(mxGetN(q)) != p.batch)) fail("admm:input", "q must be L x batch");
}

}  // namespace

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  if (nrhs < 1 || !mxIsChar(prhs[0])) fail("admm:input", "first argument must be a command string");
  char cmd[32];
  mxGetString(prhs[0], cmd, sizeof cmd);

  if (!std::strcmp(cmd, "setup")) {
    if (nrhs < 2 || !mxIsStruct(prhs[1])) fail("admm:input", "setup needs a problem struct");
    const mxArray* P = prhs[1];
    const mxArray* O = (nrhs > 2 && mxIsStruct(prhs[2])) ? prhs[2] : nullptr;
    admm_problem p;
    parse_problem(P, p);
    const int L = p.N * (p.n + p.m);
    admm_options o;
    admm_default_options(&o);
    o.rho = scalar_or(O, "rho", o.rho);
    o.alpha = scalar_or(O, "alpha", o.alpha);
    o.eps_abs = scalar_or(O, "eps_abs", o.eps_abs);
    o.eps_rel = scalar_or(O, "eps_rel", o.eps_rel);
    o.max_iter = static_cast<int32_t>(scalar_or(O, "max_iter", o.max_iter));
    o.check_interval = static_cast<int32_t>(scalar_or(O, "check_interval", o.check_interval));
    o.segments = static_cast<int32_t>(scalar_or(O, "segments", 0));
    o.device = static_cast<int32_t>(scalar_or(O, "device", -1));
    o.flags = static_cast<int32_t>(scalar_or(O, "flags", 0));
    o.adapt_interval = static_cast<int32_t>(scalar_or(O, "adapt_interval", o.adapt_interval));
    o.adapt_max = static_cast<int32_t>(scalar_or(O, "adapt_max", o.adapt_max));
    o.adapt_mu = scalar_or(O, "adapt_mu", o.adapt_mu);
    o.adapt_tau = scalar_or(O, "adapt_tau", o.adapt_tau);
    o.precision_mode = static_cast<int32_t>(scalar_or(O, "precision_mode", o.precision_mode));   // ADMM_PRECISION_*
    admm_handle* h = nullptr;
    check(admm_setup(&h, &p, &o));
    remember(h, L, p.batch);
    plhs[0] = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL);
    *static_cast<uint64_t*>(mxGetData(plhs[0])) = static_cast<uint64_t>(reinterpret_cast<uintptr_t>(h));
    return;
  }

  if (nrhs < 2) fail("admm:input", "this command needs a handle");
  admm_handle* h = handle_of(prhs[1]);
  Dims* d = lookup(h);

  if (!std::strcmp(cmd, "solve")) {
    const double* z0 = nrhs > 2 ? dbl(prhs[2], "z0") : nullptr;
    const double* y0 = nrhs > 3 ? dbl(prhs[3], "y0") : nullptr;
    if (z0 && mxGetNumberOfElements(prhs[2]) != static_cast<size_t>(d->L) * d->batch) fail("admm:input", "z0 must be L x batch");
    if (y0 && mxGetNumberOfElements(prhs[3]) != static_cast<size_t>(d->L) * d->batch) fail("admm:input", "y0 must be L x batch");
    admm_info info;
    check(admm_solve(h, z0, y0, &info));
    const char* names[] = {"iters_run", "n_converged", "max_r", "max_s", "solve_ms", "iters", "status", "r", "s",
                           "rho", "rho_updates", "mixed_iters", "rho_per_qp"};
    plhs[0] = mxCreateStructMatrix(1, 1, 13, names);
    {
      mxArray* rq = mxCreateDoubleMatrix(d->batch, 1, mxREAL);      // per-instance dynamics: each QP's own rho (ABI v5)
      check(admm_get_rho(h, mxGetPr(rq)));
      mxSetField(plhs[0], 0, "rho_per_qp", rq);
    }
    mxSetField(plhs[0], 0, "mixed_iters", mxCreateDoubleScalar(info.mixed_iters));
    mxSetField(plhs[0], 0, "rho", mxCreateDoubleScalar(info.rho));
    mxSetField(plhs[0], 0, "rho_updates", mxCreateDoubleScalar(info.rho_updates));
    mxSetField(plhs[0], 0, "iters_run", mxCreateDoubleScalar(info.iters_run));
    mxSetField(plhs[0], 0, "n_converged", mxCreateDoubleScalar(info.n_converged));
    mxSetField(plhs[0], 0, "max_r", mxCreateDoubleScalar(info.max_r));
    mxSetField(plhs[0], 0, "max_s", mxCreateDoubleScalar(info.max_s));
    mxSetField(plhs[0], 0, "solve_ms", mxCreateDoubleScalar(info.solve_ms));
    mxArray* it = mxCreateNumericMatrix(d->batch, 1, mxINT32_CLASS, mxREAL);
    mxArray* st = mxCreateNumericMatrix(d->batch, 1, mxINT32_CLASS, mxREAL);
    mxArray* r = mxCreateDoubleMatrix(d->batch, 1, mxREAL);
    mxArray* s = mxCreateDoubleMatrix(d->batch, 1, mxREAL);
    check(admm_get_info(h, static_cast<int32_t*>(mxGetData(it)), static_cast<int32_t*>(mxGetData(st)), mxGetPr(r), mxGetPr(s)));
    mxSetField(plhs[0], 0, "iters", it);
    mxSetField(plhs[0], 0, "status", st);
    mxSetField(plhs[0], 0, "r", r);
    mxSetField(plhs[0], 0, "s", s);
  } else if (!std::strcmp(cmd, "get")) {
    double* out[3] = {nullptr, nullptr, nullptr};
    for (int i = 0; i < 3 && i < (nlhs > 0 ? nlhs : 1); ++i) {
      plhs[i] = mxCreateDoubleMatrix(d->L, d->batch, mxREAL);
      out[i] = mxGetPr(plhs[i]);
    }
    check(admm_get(h, out[0], out[1], out[2]));
  } else if (!std::strcmp(cmd, "history")) {      // admm_mex('history', h): the records of ADMM_FLAG_HISTORY (o.flags = 64)
    int32_t n = 0;
    check(admm_get_history(h, 0, &n, nullptr, nullptr, nullptr, nullptr, nullptr));
    const char* names[] = {"iteration", "n_converged", "max_r", "max_s", "rho"};
    plhs[0] = mxCreateStructMatrix(1, 1, 5, names);
    mxArray* it = mxCreateNumericMatrix(n, 1, mxINT32_CLASS, mxREAL);
    mxArray* nc = mxCreateNumericMatrix(n, 1, mxINT32_CLASS, mxREAL);
    mxArray* r = mxCreateDoubleMatrix(n, 1, mxREAL);
    mxArray* s = mxCreateDoubleMatrix(n, 1, mxREAL);
    mxArray* rho = mxCreateDoubleMatrix(n, 1, mxREAL);
    check(admm_get_history(h, n, &n, static_cast<int32_t*>(mxGetData(it)), static_cast<int32_t*>(mxGetData(nc)), mxGetPr(r),
                           mxGetPr(s), mxGetPr(rho)));
    mxSetField(plhs[0], 0, "iteration", it);
    mxSetField(plhs[0], 0, "n_converged", nc);
    mxSetField(plhs[0], 0, "max_r", r);
    mxSetField(plhs[0], 0, "max_s", s);
    mxSetField(plhs[0], 0, "rho", rho);
  } else if (!std::strcmp(cmd, "path")) {         // admm_mex('path', h): which kernels the handle runs + the default path's margin
    admm_path_info pi;
    check(admm_get_path(h, &pi));
    const char* names[] = {"alternating", "alt_requested", "mfma", "xfree", "segments", "auto_segments", "scan_form",
                           "per_instance", "alt_check", "alt_gate", "scan_growth"};
    plhs[0] = mxCreateStructMatrix(1, 1, 11, names);
    const double vals[] = {(double)pi.alternating, (double)pi.alt_requested, (double)pi.mfma, (double)pi.xfree, (double)pi.segments,
                           (double)pi.auto_segments, (double)pi.scan_form, (double)pi.per_instance, pi.alt_check, pi.alt_gate,
                           pi.scan_growth};
    for (int i = 0; i < 11; ++i) mxSetField(plhs[0], 0, names[i], mxCreateDoubleScalar(vals[i]));
  } else if (!std::strcmp(cmd, "iterate")) {
    if (nrhs < 3) fail("admm:input", "iterate needs an iteration count");
    check(admm_iterate(h, static_cast<int32_t>(mxGetScalar(prhs[2]))));
    check(admm_sync(h));
  } else if (!std::strcmp(cmd, "update")) {       // admm_mex('update', h, problem): new shared data, same shape
    if (nrhs < 3 || !mxIsStruct(prhs[2])) fail("admm:input", "update needs a problem struct");
    admm_problem p;
    parse_problem(prhs[2], p);
    check(admm_update_problem(h, &p));
  } else if (!std::strcmp(cmd, "free")) {
    d->h = nullptr;
    admm_free(h);
  } else {
    mexErrMsgIdAndTxt("admm:input", "unknown command '%s'", cmd);
  }
}

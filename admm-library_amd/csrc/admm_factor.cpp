// admm_factor.cpp -- see admm_factor.hpp.  Plain C++ (no HIP): runs on the host
// once per admm_setup (and again only if rho changes).
#include "admm_factor.hpp"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <exception>
#include <mutex>
#include <thread>

namespace admm {
namespace {

using Mat = std::vector<double>;  // row-major

// The stage recursions (Riccati sweep, filter covariance) are sequential, but most of the work per rho -- the gains and
// folded operators of each stage, the segment transfer matrices, the MFMA fragment packing -- is independent per stage or
// per segment: those loops run on a few host threads (ADMM_FACTOR_THREADS, default min(cores, 16)).  Every stage is
// computed by exactly one thread with the same arithmetic, so the result does not depend on the thread count.
thread_local int g_thread_cap = 0;      // set_factor_thread_cap(): limit for factorisations started from this thread
int factor_threads() {
  static const int n = [] {
    const char* e = std::getenv("ADMM_FACTOR_THREADS");
    int v = e ? std::atoi(e) : (int)std::thread::hardware_concurrency();
    return v < 1 ? 1 : (v > 16 ? 16 : v);
  }();
  return (g_thread_cap > 0 && g_thread_cap < n) ? g_thread_cap : n;
}
#ifdef ADMM_FACTOR_TIMING          // phase timing on stderr (build with -DADMM_FACTOR_TIMING; tools only)
}  // namespace
}  // namespace admm
#include <chrono>
#include <cstdio>
namespace admm {
namespace {
struct PhaseTimer {
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void lap(const char* what) {
    const auto n = std::chrono::steady_clock::now();
    std::fprintf(stderr, "  [factor] %-28s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
    t = n;
  }
};
#else
struct PhaseTimer { void lap(const char*) {} };
#endif

template <class F>
void parallel_for(int count, F&& fn) {          // fn(begin, end) over a partition of [0, count)
  const int nt = std::min(factor_threads(), count);
  if (nt <= 1) { fn(0, count); return; }
  std::vector<std::thread> th;
  th.reserve(nt - 1);
  // An exception in a worker (std::bad_alloc from a Mat: large N, n = 12, 16 threads x two background factorisations) must
  // not reach std::terminate: the first one is kept, every worker is joined, then it is rethrown on the calling thread --
  // factorise() and the background threads of admm_api.hip turn it into ADMM_ERR_ALLOC.
  std::exception_ptr first;
  std::mutex first_mu;
  auto guarded = [&fn, &first, &first_mu](int b, int e) {
    try {
      fn(b, e);
    } catch (...) {
      std::lock_guard<std::mutex> g(first_mu);
      if (!first) first = std::current_exception();
    }
  };
  int started = 1;                               // slices [0, started) have an owner (slice 0 = this thread)
  try {
    for (int t = 1; t < nt; ++t) {
      th.emplace_back([&guarded, t, nt, count] { guarded((int)((int64_t)count * t / nt), (int)((int64_t)count * (t + 1) / nt)); });
      started = t + 1;
    }
  } catch (...) {                                // no more threads to be had: the remaining slices run here
  }
  guarded(0, count / nt);
  for (int t = started; t < nt; ++t) guarded((int)((int64_t)count * t / nt), (int)((int64_t)count * (t + 1) / nt));
  for (auto& x : th) x.join();
  if (first) std::rethrow_exception(first);
}

// c (p x r) = a (p x q) b (q x r).  Every output element is summed over k in ascending order from 0.0, whatever the
// instantiation.  The inner dimension is a template parameter for the sizes that occur (n, m <= 16): the output row then
// lives in registers across the k loop, which is worth 4-5x on the ~50 000 small products of one factorisation.
template <int R>
void mul_rows(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ c, int p, int q) {
  for (int i = 0; i < p; ++i) {
    double acc[R];
    for (int j = 0; j < R; ++j) acc[j] = 0.0;
    for (int k = 0; k < q; ++k) {
      const double aik = a[(size_t)i * q + k];
      for (int j = 0; j < R; ++j) acc[j] += aik * b[(size_t)k * R + j];
    }
    for (int j = 0; j < R; ++j) c[(size_t)i * R + j] = acc[j];
  }
}
Mat mul(const Mat& a, const Mat& b, int p, int q, int r) {
  Mat c((size_t)p * r, 0.0);
  switch (r) {
#define ADMM_MUL_CASE(R) case R: mul_rows<R>(a.data(), b.data(), c.data(), p, q); return c;
    ADMM_MUL_CASE(1) ADMM_MUL_CASE(2) ADMM_MUL_CASE(3) ADMM_MUL_CASE(4) ADMM_MUL_CASE(5) ADMM_MUL_CASE(6) ADMM_MUL_CASE(7) ADMM_MUL_CASE(8)
    ADMM_MUL_CASE(9) ADMM_MUL_CASE(10) ADMM_MUL_CASE(11) ADMM_MUL_CASE(12) ADMM_MUL_CASE(13) ADMM_MUL_CASE(14) ADMM_MUL_CASE(15) ADMM_MUL_CASE(16)
#undef ADMM_MUL_CASE
    default: break;
  }
  for (int i = 0; i < p; ++i)
    for (int k = 0; k < q; ++k) {
      const double aik = a[(size_t)i * q + k];
      for (int j = 0; j < r; ++j) c[(size_t)i * r + j] += aik * b[(size_t)k * r + j];
    }
  return c;
}

Mat tr(const Mat& a, int p, int q) {
  Mat t((size_t)p * q);
  for (int i = 0; i < p; ++i)
    for (int j = 0; j < q; ++j) t[(size_t)j * p + i] = a[(size_t)i * q + j];
  return t;
}

Mat eye(int n) {
  Mat e((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) e[(size_t)i * n + i] = 1.0;
  return e;
}

void symmetrise(Mat& a, int n) {
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j) {
      const double v = 0.5 * (a[(size_t)i * n + j] + a[(size_t)j * n + i]);
      a[(size_t)i * n + j] = a[(size_t)j * n + i] = v;
    }
}

// SPD inverse through Cholesky; false if not positive definite.
bool spd_inverse(const Mat& s, int k, Mat& inv) {
  Mat l((size_t)k * k, 0.0);
  for (int i = 0; i < k; ++i)
    for (int j = 0; j <= i; ++j) {
      double v = s[(size_t)i * k + j];
      for (int t = 0; t < j; ++t) v -= l[(size_t)i * k + t] * l[(size_t)j * k + t];
      if (i == j) {
        if (!(v > 0.0) || !std::isfinite(v)) return false;
        l[(size_t)i * k + i] = std::sqrt(v);
      } else {
        l[(size_t)i * k + j] = v / l[(size_t)j * k + j];
      }
    }
  // li = L^{-1}
  Mat li((size_t)k * k, 0.0);
  for (int c = 0; c < k; ++c)
    for (int i = c; i < k; ++i) {
      double v = (i == c) ? 1.0 : 0.0;
      for (int t = c; t < i; ++t) v -= l[(size_t)i * k + t] * li[(size_t)t * k + c];
      li[(size_t)i * k + c] = v / l[(size_t)i * k + i];
    }
  inv.assign((size_t)k * k, 0.0);
  for (int i = 0; i < k; ++i)
    for (int j = 0; j < k; ++j) {
      double v = 0.0;
      for (int t = (i > j ? i : j); t < k; ++t) v += li[(size_t)t * k + i] * li[(size_t)t * k + j];
      inv[(size_t)i * k + j] = v;
    }
  return true;
}

// column-major ABI matrix (r x c) -> row-major
Mat from_colmajor(const double* a, int r, int c) {
  Mat o((size_t)r * c);
  for (int i = 0; i < r; ++i)
    for (int j = 0; j < c; ++j) o[(size_t)i * c + j] = a[(size_t)j * r + i];
  return o;
}

bool all_finite(const double* a, size_t cnt) {
  for (size_t i = 0; i < cnt; ++i)
    if (!std::isfinite(a[i])) return false;
  return true;
}


// W (row-major M x K) -> MFMA A-fragment order, plus each M-group's non-zero k-step range.
// Columns of a dense scan matrix (M x K, row-major) from the ordinary input layout  tseg(0..S-1) | x0 | eseg(0..S-1) | pad
// to the rank-by-rank layout of time-sharded handles (admm_factor.hpp).
void timeshard_columns(std::vector<double>& W, int M, int K, int S, int n, int ranks) {
  if (ranks <= 1) return;
  const int sl = S / ranks, Sn = S * n;
  std::vector<int> newcol(K);
  for (int c = 0; c < K; ++c) newcol[c] = c;
  for (int s = 0; s < S; ++s)
    for (int i = 0; i < n; ++i) {
      const int base = (s / sl) * 2 * sl * n + (s % sl) * n + i;
      newcol[s * n + i] = base;                          // tseg(s)
      newcol[Sn + n + s * n + i] = base + sl * n;        // eseg(s)
    }
  for (int i = 0; i < n; ++i) newcol[Sn + i] = 2 * Sn + i;  // x0
  std::vector<double> row(K);
  for (int r = 0; r < M; ++r) {
    double* w = &W[(size_t)r * K];
    for (int c = 0; c < K; ++c) row[newcol[c]] = w[c];
    for (int c = 0; c < K; ++c) w[c] = row[c];
  }
}

void pack_scan(const std::vector<double>& Wm, int M, int K, std::vector<double>& Wp, std::vector<int32_t>& range) {
  auto round_up = [](int v, int q) { return ((v + q - 1) / q) * q; };
  auto W = [&](int r, int c) -> double { return Wm[(size_t)r * K + c]; };
  const int mtiles = M / 16, ksteps = K / 4, groups = mtiles / SCAN_MT;
  Wp.assign((size_t)ksteps * mtiles * 64, 0.0);
  for (int ks = 0; ks < ksteps; ++ks)
    for (int mt = 0; mt < mtiles; ++mt)
      for (int lane = 0; lane < 64; ++lane)
        Wp[((size_t)ks * mtiles + mt) * 64 + lane] = W(16 * mt + (lane & 15), 4 * ks + (lane >> 4));
  range.assign((size_t)2 * groups, 0);
  for (int g = 0; g < groups; ++g) {
    int kb = ksteps, ke = 0;
    for (int r = g * SCAN_MT * 16; r < (g + 1) * SCAN_MT * 16; ++r)
      for (int c = 0; c < K; ++c)
        if (W(r, c) != 0.0) {
          if (c / 4 < kb) kb = c / 4;
          if (c / 4 + 1 > ke) ke = c / 4 + 1;
        }
    if (ke < kb) { kb = 0; ke = 0; }
    kb = (kb / SCAN_KALIGN) * SCAN_KALIGN;            // W is zero outside the true range, so
    ke = round_up(ke, SCAN_KALIGN);                   // widening it to the batch size is harmless
    range[2 * g] = kb;
    range[2 * g + 1] = ke;
  }
}

Mat add(const Mat& a, const Mat& b) {
  Mat c(a);
  for (size_t i = 0; i < c.size(); ++i) c[i] += b[i];
  return c;
}
Mat sub(const Mat& a, const Mat& b) {
  Mat c(a);
  for (size_t i = 0; i < c.size(); ++i) c[i] -= b[i];
  return c;
}
Mat neg(const Mat& a) {
  Mat c(a);
  for (auto& v : c) v = -v;
  return c;
}
void put(double* dst, const Mat& a) {
  for (size_t i = 0; i < a.size(); ++i) dst[i] = a[i];
}

// Row-block recurrences on a dense scan matrix W (row-major, K columns, blocks of n rows): the chains the
// scan solves are linear, so each row block of W is the chain step applied to earlier row blocks,
//     W[dst] = M1 W[src1] (+ M2 W[src2]) (+ I in the n columns starting at unit_col)
// -- O(S^2 n^3) for the whole matrix instead of summing products of transfer matrices per entry.
// src < 0 = absent; dst must differ from src1 and src2.
void rowblock_step(std::vector<double>& W, int K, int n, int dst, const Mat* M1, int src1, const Mat* M2, int src2,
                   int unit_col) {
  double* d = &W[(size_t)dst * K];
  for (int i = 0; i < n; ++i)
    for (int c = 0; c < K; ++c) d[(size_t)i * K + c] = 0.0;
  auto acc = [&](const Mat& Mx, int src) {
    const double* sp = &W[(size_t)src * K];
    for (int i = 0; i < n; ++i)
      for (int l = 0; l < n; ++l) {
        const double a = Mx[(size_t)i * n + l];
        if (a == 0.0) continue;
        const double* srow = sp + (size_t)l * K;
        double* drow = d + (size_t)i * K;
        for (int c = 0; c < K; ++c) drow[c] += a * srow[c];
      }
  };
  if (M1 && src1 >= 0) acc(*M1, src1);
  if (M2 && src2 >= 0) acc(*M2, src2);
  if (unit_col >= 0)
    for (int i = 0; i < n; ++i) d[(size_t)i * K + unit_col + i] += 1.0;
}

// general n x n inverse (Gauss-Jordan, partial pivoting); false if a pivot is tiny relative to the matrix
bool general_inverse(const Mat& a, int n, Mat& inv) {
  Mat w(a);
  inv = eye(n);
  double scale = 0.0;
  for (double v : a) scale = std::max(scale, std::fabs(v));
  if (!(scale > 0.0) || !std::isfinite(scale)) return false;
  for (int c = 0; c < n; ++c) {
    int piv = c;
    for (int r = c + 1; r < n; ++r)
      if (std::fabs(w[(size_t)r * n + c]) > std::fabs(w[(size_t)piv * n + c])) piv = r;
    const double pv = w[(size_t)piv * n + c];
    if (!(std::fabs(pv) > 1e-10 * scale)) return false;
    if (piv != c)
      for (int j = 0; j < n; ++j) {
        std::swap(w[(size_t)piv * n + j], w[(size_t)c * n + j]);
        std::swap(inv[(size_t)piv * n + j], inv[(size_t)c * n + j]);
      }
    for (int j = 0; j < n; ++j) { w[(size_t)c * n + j] /= pv; inv[(size_t)c * n + j] /= pv; }
    for (int r = 0; r < n; ++r) {
      if (r == c) continue;
      const double fct = w[(size_t)r * n + c];
      if (fct == 0.0) continue;
      for (int j = 0; j < n; ++j) {
        w[(size_t)r * n + j] -= fct * w[(size_t)c * n + j];
        inv[(size_t)r * n + j] -= fct * inv[(size_t)c * n + j];
      }
    }
  }
  return true;
}

// Moore-Penrose inverse of a symmetric positive semidefinite matrix (cyclic Jacobi eigen-decomposition;
// eigenvalues below tol * largest count as zero).
Mat pinv_psd(const Mat& a_in, int n, double tol) {
  Mat a(a_in), v = eye(n);
  symmetrise(a, n);
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) (i == j ? diag : off) += a[(size_t)i * n + j] * a[(size_t)i * n + j];
    if (off <= 1e-32 * diag || off == 0.0) break;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = a[(size_t)p * n + q];
        if (apq == 0.0) continue;
        const double theta = (a[(size_t)q * n + q] - a[(size_t)p * n + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
        for (int k = 0; k < n; ++k) {      // columns p, q of a
          const double akp = a[(size_t)k * n + p], akq = a[(size_t)k * n + q];
          a[(size_t)k * n + p] = c * akp - sn * akq;
          a[(size_t)k * n + q] = sn * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {      // rows p, q of a
          const double apk = a[(size_t)p * n + k], aqk = a[(size_t)q * n + k];
          a[(size_t)p * n + k] = c * apk - sn * aqk;
          a[(size_t)q * n + k] = sn * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {      // eigenvectors
          const double vkp = v[(size_t)k * n + p], vkq = v[(size_t)k * n + q];
          v[(size_t)k * n + p] = c * vkp - sn * vkq;
          v[(size_t)k * n + q] = sn * vkp + c * vkq;
        }
      }
  }
  double wmax = 0.0;
  for (int i = 0; i < n; ++i) wmax = std::max(wmax, a[(size_t)i * n + i]);
  Mat out((size_t)n * n, 0.0);
  for (int e = 0; e < n; ++e) {
    const double w = a[(size_t)e * n + e];
    if (!(w > tol * wmax) || !(w > 0.0)) continue;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) out[(size_t)i * n + j] += v[(size_t)i * n + e] * v[(size_t)j * n + e] / w;
  }
  return out;
}

// Forward-elimination form of the x-update -- the time-reversed mirror of the Riccati form -- and its
// segment algebra.  With Rr = R + rho I, Qr_k = Q (QN at k = N) + rho I and the
// filter covariance C_0 = 0 (x_0 is known):
//     Pm      = A_k C_k A_k' + B_k Rr^{-1} B_k'                     (prior covariance of x_{k+1})
//     G_{k+1} = Pm (Pm + Qr_{k+1}^{-1})^{-1},       C_{k+1} = Pm - G_{k+1} Pm
//     Kb_k    = -Rr^{-1} B_k' Pm^+                                  (backward feedback gain, m x n)
// elimination (forward, mu_0 = x_0):
//     db_k     = DK_k mu_k + DG_k g^u_k           -> stored, m rows per stage   (DK = -Kb A, DG = Rr^{-1} (I + B' Kb'))
//     mu_{k+1} = F_k mu_k + Gam_k g^u_k + Pi_k g^x_{k+1}
// substitution (backward, x_N = mu_N):
//     u_k = -Kb_k x_{k+1} - db_k,      x_k = A_k^{-1} (x_{k+1} - B_k u_k)
// i.e. a feedback law and a rollout, as in the Riccati form but running backward in time; the map
// x_{k+1} -> x_k is the (stable) smoother gain wherever Pm has full rank.  Pm is rank deficient for
// the first ceil(n/m) stages (pseudo-inverse: x_{k+1} - m^-_{k+1} lies in its range), and the gains
// there are large, so the form is verified below against the Riccati form on random data before it
// is enabled.  Needs every A_k invertible (true for any discretised ODE).
// Leaves f.alt_ok false if a pivot fails, a transfer matrix overflows or the verification misses.
void build_alternating(Factor& f, const std::vector<Mat>& A, const std::vector<Mat>& B, const Mat& Q,
                       const Mat& R, const Mat& QN, double rho, bool pack_scan_mfma) {
  const int N = f.N, n = f.n, m = f.m, S = f.S;
  f.alt_ok = false;
  f.alt_check = -1.0;
  f.rho = rho;
  f.RFE = rec_fe_size(n, m);
  f.RBE = rec_be_size(n, m);
  f.recFE.assign((size_t)N * f.RFE, 0.0);
  f.recBE.assign((size_t)N * f.RBE, 0.0);
  const RecFELayout lfe = rec_fe_layout(n, m);
  const RecBELayout lbe = rec_be_layout(n, m);
  const RecBLayout lb = rec_b_layout(n, m);
  const RecFLayout lf = rec_f_layout(n, m);
  const Mat I = eye(n);
  Mat Rr = R;
  for (int i = 0; i < m; ++i) Rr[(size_t)i * m + i] += rho;
  symmetrise(Rr, m);
  Mat Ri;
  if (!spd_inverse(Rr, m, Ri)) return;
  symmetrise(Ri, m);
  Mat Qr = Q, QNr = QN;
  for (int i = 0; i < n; ++i) { Qr[(size_t)i * n + i] += rho; QNr[(size_t)i * n + i] += rho; }
  symmetrise(Qr, n);
  symmetrise(QNr, n);
  Mat Qi, QNi;
  if (!spd_inverse(Qr, n, Qi) || !spd_inverse(QNr, n, QNi)) return;

  std::vector<Mat> Fm(N), Gam(N), Pi(N), DK(N), DG(N), Kb(N), Ai(N), AiB(N), Jb(N);
  PhaseTimer timer;
  // (1) the covariance recursion, sequential: Pm_k, G_{k+1}
  std::vector<Mat> Pmk(N), Gk(N);
  {
    Mat C((size_t)n * n, 0.0);
    Mat BRB;                                                  // B Rr^{-1} B' (the same at every stage unless time varying)
    for (int k = 0; k < N; ++k) {
      const Mat At = tr(A[k], n, n);
      if (k == 0 || B[k] != B[k - 1]) {
        const Mat Bt = tr(B[k], n, m);
        BRB = mul(mul(B[k], Ri, n, m, m), Bt, n, m, n);
      }
      Mat Pm = add(mul(mul(A[k], C, n, n, n), At, n, n, n), BRB);
      symmetrise(Pm, n);
      const Mat& Qin = (k + 1 == N) ? QNi : Qi;
      Mat D = add(Pm, Qin);
      symmetrise(D, n);
      Mat Di;
      if (!spd_inverse(D, n, Di)) return;
      Gk[k] = mul(Pm, Di, n, n, n);
      C = sub(Pm, mul(Gk[k], Pm, n, n, n));
      symmetrise(C, n);
      Pmk[k] = std::move(Pm);
    }
  }
  timer.lap("  alt: covariance recursion");
  // (2) gains, folded operators and records of each stage, independent given (Pm_k, G_{k+1})
  std::atomic<bool> bad{false};
  parallel_for(N, [&](int k_begin, int k_end) {
  for (int k = k_begin; k < k_end && !bad; ++k) {
    const Mat Bt = tr(B[k], n, m);
    const Mat BRi = mul(B[k], Ri, n, m, m);
    const Mat& Pm = Pmk[k];
    const Mat& G = Gk[k];
    const Mat& Qin = (k + 1 == N) ? QNi : Qi;
    const Mat ImG = sub(I, G);
    Kb[k] = neg(mul(mul(Ri, Bt, m, m, n), pinv_psd(Pm, n, 1e-13), m, n, n));
    if (k > k_begin && A[k] == A[k - 1]) Ai[k] = Ai[k - 1];
    else if (!general_inverse(A[k], n, Ai[k])) { bad = true; return; }
    AiB[k] = mul(Ai[k], B[k], n, n, m);
    Jb[k] = mul(Ai[k], add(I, mul(B[k], Kb[k], n, m, n)), n, n, n);       // x_k = Jb x_{k+1} + AiB db_k
    Fm[k] = mul(ImG, A[k], n, n, n);
    Gam[k] = neg(mul(ImG, BRi, n, n, m));
    Pi[k] = neg(mul(G, Qin, n, n, n));
    DK[k] = neg(mul(Kb[k], A[k], m, n, n));
    DG[k] = add(Ri, mul(Kb[k], BRi, m, n, m));                            // Rr^{-1} - Kb (-B Rr^{-1})
    // stage-local blocks; the rollout / elimination blocks are copied from the plain records
    double* rfe = &f.recFE[(size_t)k * f.RFE];
    double* rbe = &f.recBE[(size_t)k * f.RBE];
    const double* rf = &f.recF[(size_t)k * f.RF];
    const double* rb = &f.recB[(size_t)k * f.RB];
    for (int i = 0; i < even_up(m * n); ++i) { rfe[lfe.PSI + i] = rf[lf.PSI + i]; rfe[lfe.K + i] = rf[lf.K + i]; }
    for (int i = 0; i < even_up(n * n); ++i) rfe[lfe.A + i] = rf[lf.A + i];
    for (int i = 0; i < even_up(n * m); ++i) rfe[lfe.B + i] = rf[lf.B + i];
    put(rfe + lfe.FM, Fm[k]);
    put(rfe + lfe.GA, Gam[k]);
    put(rfe + lfe.PI, Pi[k]);
    put(rfe + lfe.DK, DK[k]);
    put(rfe + lfe.DG, DG[k]);
    put(rbe + lbe.KB, Kb[k]);
    put(rbe + lbe.AI, Ai[k]);
    put(rbe + lbe.AIB, neg(AiB[k]));                                      // x_k = AI x_{k+1} + AIB u_k
    for (int i = 0; i < even_up(n * n); ++i) rbe[lbe.AT + i] = rb[lb.AT + i];
    for (int i = 0; i < even_up(m * n); ++i) rbe[lbe.BT + i] = rb[lb.BT + i];
    for (int i = 0; i < even_up(m * m); ++i) rbe[lbe.SI + i] = rb[lb.SI + i];
    for (int i = 0; i < even_up(n * m); ++i) { rbe[lbe.KT + i] = rb[lb.KT + i]; rbe[lbe.OM + i] = rb[lb.OM + i]; }
    for (int i = 0; i < even_up(n + m); ++i) {
      rfe[lfe.LO + i] = rf[lf.LO + i]; rfe[lfe.HI + i] = rf[lf.HI + i];
      rbe[lbe.LO + i] = rb[lb.LO + i]; rbe[lbe.HI + i] = rb[lb.HI + i];
    }
    rfe[lfe.UB] = rf[lf.UB];
    rbe[lbe.UB] = rb[lb.UB];
  }
  });
  if (bad) return;
  timer.lap("  alt: stage operators");

  // ---- segment algebra (a = segment start, b = end) ----
  //   mu_k  = mu0_k + (F_{k-1} ... F_a) m_in              =>  db_k = db0_k + Psb_k m_in,  Psb_k = DK_k F_{k-1} ... F_a
  //   x_a   = (Jb_a ... Jb_{b-1}) x_b + sum_k Omb_k db_k,      Omb_k = (Jb_a ... Jb_{k-1}) A_k^{-1} B_k
  //   m_out = mseg + Phs m_in;    x_end(s-1) = ebseg(s) + Thb(s) x_end(s) + Xib(s) m_in(s),   ebseg = sum_k Omb_k db0_k
  std::vector<Mat> Phs(S), Thb(S), Xib(S);
  parallel_for(S, [&](int s_begin, int s_end) {
  for (int s = s_begin; s < s_end; ++s) {
    const int a = f.seg_start[s], b = f.seg_start[s + 1];
    Mat P = I, Jp = I, Xi((size_t)n * n, 0.0);
    for (int k = a; k < b; ++k) {
      const Mat Psb = mul(DK[k], P, m, n, n);
      const Mat Omb = mul(Jp, AiB[k], n, n, m);
      put(&f.recBE[(size_t)k * f.RBE + lbe.PSB], Psb);
      put(&f.recFE[(size_t)k * f.RFE + lfe.OB], Omb);
      Xi = add(Xi, mul(Omb, Psb, n, m, n));
      P = mul(Fm[k], P, n, n, n);
      Jp = mul(Jp, Jb[k], n, n, n);
    }
    Phs[s] = P;
    Thb[s] = Jp;
    Xib[s] = Xi;
  }
  });
  for (int k = 0; k < N; ++k) {                       // (the box blocks may hold +-inf: matrices only)
    for (int i = 0; i < lfe.LO; ++i) if (!std::isfinite(f.recFE[(size_t)k * f.RFE + i])) return;
    for (int i = 0; i < lbe.LO; ++i) if (!std::isfinite(f.recBE[(size_t)k * f.RBE + i])) return;
  }

  timer.lap("  alt: segment algebra");
  // ---- scan matrix, same shape and layout as scanW ----
  //   in rows:  mseg(0..S-1) | x0 | ebseg(0..S-1)          out rows:  m_in(0..S-1) | x_end(0..S-1)
  const int Sn = S * n, Mt = f.scanMt, M = f.scanM, K = f.scanK;
  f.scanWB.assign((size_t)M * K, 0.0);
  const int c_m = 0, c_x0 = Sn, c_e = Sn + n, r_m = 0, r_x = Mt;
  // The chains applied to row blocks of WB (rowblock_step):
  //   m_in(0)    = x0;       m_in(s+1)  = mseg(s) + Phs_s m_in(s)
  //   x_end(S-1) = m_in(S);  x_end(s-1) = ebseg(s) + Thb_s x_end(s) + Xib_s m_in(s)
  rowblock_step(f.scanWB, K, n, r_m, nullptr, -1, nullptr, -1, c_x0);
  for (int s = 0; s + 1 < S; ++s)
    rowblock_step(f.scanWB, K, n, r_m + (s + 1) * n, &Phs[s], r_m + s * n, nullptr, -1, c_m + s * n);
  rowblock_step(f.scanWB, K, n, r_x + (S - 1) * n, &Phs[S - 1], r_m + (S - 1) * n, nullptr, -1, c_m + (S - 1) * n);
  for (int s = S - 1; s >= 1; --s)
    rowblock_step(f.scanWB, K, n, r_x + (s - 1) * n, &Thb[s], r_x + s * n, &Xib[s], r_m + s * n, c_e + s * n);
  for (double v : f.scanWB) if (!std::isfinite(v)) return;

  timer.lap("  alt: scan matrix");
  // ---- verification against the Riccati form (one QP, random linear term, the scan as the dense product) ----
  {
    const int nb = n + m;
    uint64_t lcg = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; return ((double)(lcg >> 11) / 9007199254740992.0) * 2.0 - 1.0; };
    std::vector<double> g((size_t)N * nb), x0(n), wa((size_t)N * nb), wb((size_t)N * nb);
    for (auto& v : g) v = rnd();
    for (auto& v : x0) v = rnd();
    auto matvec = [](const double* Mx, const double* x, int r, int c, double* y, double sgn, bool accumulate) {
      for (int i = 0; i < r; ++i) {
        double a = accumulate ? y[i] : 0.0;
        for (int j = 0; j < c; ++j) a += sgn * Mx[(size_t)i * c + j] * x[j];
        y[i] = a;
      }
    };
    {  // Riccati form, sequential: records hold AT, BT, SI, KT (backward) and K, A, B (forward)
      std::vector<double> t(n, 0.0), p(n), h(m), d((size_t)N * m), x(x0), xn(n), u(m), tn(n);
      for (int k = N - 1; k >= 0; --k) {
        const double* rb = &f.recB[(size_t)k * f.RB];
        for (int i = 0; i < n; ++i) p[i] = g[(size_t)k * nb + m + i] + t[i];
        for (int j = 0; j < m; ++j) h[j] = g[(size_t)k * nb + j];
        matvec(rb + lb.BT, p.data(), m, n, h.data(), 1.0, true);
        matvec(rb + lb.SI, h.data(), m, m, &d[(size_t)k * m], 1.0, false);
        matvec(rb + lb.AT, p.data(), n, n, tn.data(), 1.0, false);
        matvec(rb + lb.KT, h.data(), n, m, tn.data(), -1.0, true);
        t = tn;
      }
      for (int k = 0; k < N; ++k) {
        const double* rf = &f.recF[(size_t)k * f.RF];
        for (int j = 0; j < m; ++j) u[j] = -d[(size_t)k * m + j];
        matvec(rf + lf.K, x.data(), m, n, u.data(), -1.0, true);
        matvec(rf + lf.A, x.data(), n, n, xn.data(), 1.0, false);
        matvec(rf + lf.B, u.data(), n, m, xn.data(), 1.0, true);
        x = xn;
        for (int j = 0; j < m; ++j) wa[(size_t)k * nb + j] = u[j];
        for (int i = 0; i < n; ++i) wa[(size_t)k * nb + m + i] = x[i];
      }
    }
    {  // forward-elimination form through the segment records and WB
      std::vector<double> db((size_t)N * m), in(K, 0.0), out(M, 0.0), mu(n), mn(n), eb(n), dd(m);
      for (int s = 0; s < S; ++s) {
        std::fill(mu.begin(), mu.end(), 0.0);
        std::fill(eb.begin(), eb.end(), 0.0);
        for (int k = f.seg_start[s]; k < f.seg_start[s + 1]; ++k) {
          const double* rfe = &f.recFE[(size_t)k * f.RFE];
          const double* gu = &g[(size_t)k * nb];
          const double* gx = gu + m;
          matvec(rfe + lfe.DK, mu.data(), m, n, dd.data(), 1.0, false);
          matvec(rfe + lfe.DG, gu, m, m, dd.data(), 1.0, true);
          for (int j = 0; j < m; ++j) db[(size_t)k * m + j] = dd[j];
          matvec(rfe + lfe.OB, dd.data(), n, m, eb.data(), 1.0, true);
          matvec(rfe + lfe.FM, mu.data(), n, n, mn.data(), 1.0, false);
          matvec(rfe + lfe.GA, gu, n, m, mn.data(), 1.0, true);
          matvec(rfe + lfe.PI, gx, n, n, mn.data(), 1.0, true);
          mu = mn;
        }
        for (int i = 0; i < n; ++i) { in[c_m + s * n + i] = mu[i]; in[c_e + s * n + i] = eb[i]; }
      }
      for (int i = 0; i < n; ++i) in[c_x0 + i] = x0[i];
      matvec(f.scanWB.data(), in.data(), M, K, out.data(), 1.0, false);
      std::vector<double> x(n), xk(n), u(m), d(m);
      for (int s = 0; s < S; ++s) {
        const double* mi = &out[r_m + s * n];
        for (int i = 0; i < n; ++i) x[i] = out[r_x + s * n + i];
        for (int k = f.seg_start[s + 1] - 1; k >= f.seg_start[s]; --k) {
          const double* rbe = &f.recBE[(size_t)k * f.RBE];
          for (int j = 0; j < m; ++j) d[j] = db[(size_t)k * m + j];
          matvec(rbe + lbe.PSB, mi, m, n, d.data(), 1.0, true);
          for (int j = 0; j < m; ++j) u[j] = -d[j];
          matvec(rbe + lbe.KB, x.data(), m, n, u.data(), -1.0, true);
          for (int j = 0; j < m; ++j) wb[(size_t)k * nb + j] = u[j];
          for (int i = 0; i < n; ++i) wb[(size_t)k * nb + m + i] = x[i];
          matvec(rbe + lbe.AI, x.data(), n, n, xk.data(), 1.0, false);
          matvec(rbe + lbe.AIB, u.data(), n, m, xk.data(), 1.0, true);
          x = xk;
        }
      }
    }
    double err = 0.0, scale = 1.0;
    for (size_t i = 0; i < wa.size(); ++i) {
      if (!std::isfinite(wb[i])) return;
      err = std::max(err, std::fabs(wa[i] - wb[i]));
      scale = std::max(scale, std::fabs(wa[i]));
    }
    f.alt_check = err / scale;
    // 5e-12 on one x-update: iterates then stay within ~3e-11 of the Riccati-form iterates (900 random problems,
    // tools/stress_alt.py: max 2.5e-11), inside the 1e-10 parity tolerance with margin
    if (!(f.alt_check <= 5e-12)) return;
  }
  timer.lap("  alt: verification");
  f.scanWpB.clear();
  f.scanRangeB.clear();
  timeshard_columns(f.scanWB, M, K, S, n, f.ts_ranks);      // (after the verification above, which uses the ordinary layout)
  if (pack_scan_mfma) pack_scan(f.scanWB, M, K, f.scanWpB, f.scanRangeB);
  f.alt_ok = true;
  timer.lap("  alt: pack scan");
}


// ---- MFMA form of the fused stage operators (admm_mfma_layout.hpp) ----
// One product: M (R x C, row-major) -> ks * ot fragments of 64 values.  out_row(ot, r, g) / in_col(ks, g) give the
// matrix row / column a slot stands for (-1 = padding).  The map from fragment element to matrix element is the same
// at every stage, so it is built once (offsets into M, -1 = zero) and each stage is a gather.
template <class OutRow, class InCol>
std::vector<int32_t> product_map(int C, int ks_n, int ot_n, bool f32, OutRow out_row, InCol in_col) {
  std::vector<int32_t> map;
  map.reserve((size_t)ks_n * ot_n * 64);
  for (int ks = 0; ks < ks_n; ++ks)
    for (int ot = 0; ot < ot_n; ++ot)
      for (int lane = 0; lane < 64; ++lane) {
        const int i = lane & 15, kk = lane >> 4;
        const int g = f32 ? (i >> 2) : (i & 3), r = f32 ? (i & 3) : (i >> 2);   // slot (r, g) of hardware row i
        const int row = out_row(ot, r, g), col = in_col(ks, kk);
        map.push_back((row >= 0 && col >= 0) ? row * C + col : -1);
      }
  return map;
}

unsigned char* write_product(unsigned char* dst, const std::vector<int32_t>& map, const Mat& M, int elem) {
  if (elem == 8) {
    double* o = reinterpret_cast<double*>(dst);
    for (size_t i = 0; i < map.size(); ++i) o[i] = map[i] < 0 ? 0.0 : M[map[i]];
  } else {
    float* o = reinterpret_cast<float*>(dst);
    for (size_t i = 0; i < map.size(); ++i) o[i] = map[i] < 0 ? 0.0f : (float)M[map[i]];
  }
  return dst + map.size() * elem;
}

Mat block(const double* rec, int off, int r, int c) { return Mat(rec + off, rec + off + (size_t)r * c); }

void put_block(Mat& M, int C, int r0, int c0, const Mat& B, int r, int c, double sgn) {
  for (int i = 0; i < r; ++i)
    for (int j = 0; j < c; ++j) M[(size_t)(r0 + i) * C + c0 + j] = sgn * B[(size_t)i * c + j];
}

// One pass over the stages folds the operators and writes the records of `mode`; for mode 1 (mixed) the all-fp64
// records the refinement phase needs (recMF64 / recMB64) are written from the same folded matrices.
void build_mfma(Factor& f, int mode) {
  const int N = f.N, n = f.n, m = f.m;
  const MfmaLayout L = mfma_layout(n, m);
  struct Out {
    int mode, RMF, RMB;
    std::vector<unsigned char>*recMF, *recMB;
    std::vector<int32_t> sub_f, elim_f, sub_b, elim_b;
  };
  std::vector<Out> outs;
  outs.push_back(Out{mode, 0, 0, &f.recMF, &f.recMB, {}, {}, {}, {}});
  if (mode == 1) outs.push_back(Out{2, 0, 0, &f.recMF64, &f.recMB64, {}, {}, {}, {}});
  for (Out& o : outs) {
    o.RMF = mfma_rec_bytes_fwd(n, m, o.mode);
    o.RMB = mfma_rec_bytes_bwd(n, m, o.mode);
    o.recMF->assign((size_t)N * o.RMF, 0);
    o.recMB->assign((size_t)N * o.RMB, 0);
  }
  f.RMF = outs[0].RMF; f.RMB = outs[0].RMB;
  if (mode == 1) { f.RMF64 = outs[1].RMF; f.RMB64 = outs[1].RMB; }
  const RecBLayout lb = rec_b_layout(n, m);
  const RecFLayout lf = rec_f_layout(n, m);
  const RecFELayout lfe = rec_fe_layout(n, m);
  const RecBELayout lbe = rec_be_layout(n, m);
  // slot -> row maps shared by the products
  const int NR = L.nr;
  auto nvec = [&](int r, int g) { return (r < NR && 4 * r + g < n) ? 4 * r + g : -1; };    // slots of an n-vector
  auto sub_out = [&](int, int r, int g) -> int {             // rows: [x (n) ; u (m)]; one tile: x + u rows 0..3
    return r < 3 ? nvec(r, g) : (g < m ? n + g : -1);
  };
  auto sub_in = [&](int ks, int kk) -> int {                 // cols: [x (n) ; second n-vector ; m-vector]
    if (ks < NR) return nvec(ks, kk);
    if (ks < 2 * NR) return nvec(ks - NR, kk) < 0 ? -1 : n + nvec(ks - NR, kk);
    if (ks == 2 * NR) return kk < m ? 2 * n + kk : -1;
    return 4 + kk < m ? 2 * n + 4 + kk : -1;
  };
  // rows 4.. of u: coefficient of the lane's own slot, [row - 4][ks][g] (admm_mfma_layout.hpp)
  auto urow_table = [&](const Mat& Ms, int C, double* out) {
    for (int j = 0; j < L.urows; ++j)
      for (int ks = 0; ks < L.ks_sub; ++ks)
        for (int g = 0; g < 4; ++g) {
          const int col = sub_in(ks, g);
          out[(j * L.ks_sub + ks) * 4 + g] = col >= 0 ? Ms[(size_t)(n + 4 + j) * C + col] : 0.0;
        }
  };
  auto elim_out = [&](int ot, int r, int g) -> int {         // rows: [first n-vector ; second n-vector ; m-vector]
    if (r < 3) return nvec(r, g) < 0 ? -1 : ot * n + nvec(r, g);
    const int j = 4 * ot + g;
    return j < m ? 2 * n + j : -1;
  };
  auto elimb_in = [&](int ks, int kk) -> int {               // cols: [p (n) ; g^u (m)]
    if (ks < NR) return nvec(ks, kk);
    const int j = 4 * (ks - NR) + kk;
    return j < m ? n + j : -1;
  };
  const Mat Im = eye(m);
  const int C3 = 2 * n + m, C2 = n + m;
  for (Out& o : outs) {
    o.sub_f = product_map(C3, L.ks_sub, 1, mfma_es_sub_f(o.mode) == 4, sub_out, sub_in);
    o.elim_f = product_map(C3, L.ks_elim_f, 2, mfma_es_elim_f(o.mode) == 4, elim_out, sub_in);
    o.sub_b = product_map(C3, L.ks_sub, 1, mfma_es_sub_b(o.mode) == 4, sub_out, sub_in);
    o.elim_b = product_map(C2, L.ks_elim_b, 2, mfma_es_elim_b(o.mode) == 4, elim_out, elimb_in);
  }
  parallel_for(N, [&](int k_begin, int k_end) {
  for (int k = k_begin; k < k_end; ++k) {
    const double* rf = &f.recF[(size_t)k * f.RF];
    const double* rb = &f.recB[(size_t)k * f.RB];
    const Mat K = block(rf, lf.K, m, n), Psi = block(rf, lf.PSI, m, n), A = block(rf, lf.A, n, n), B = block(rf, lf.B, n, m);
    const Mat AT = block(rb, lb.AT, n, n), BT = block(rb, lb.BT, m, n), SI = block(rb, lb.SI, m, m),
              KT = block(rb, lb.KT, n, m), OM = block(rb, lb.OM, n, m);
    // lo / hi in slot order (slot index r * 4 + g, r = 0..4)
    double lohi[40];
    for (int r = 0; r < 5; ++r)
      for (int g = 0; g < 4; ++g) {
        int row = -1;                                        // row of block k (u rows first)
        if (r < 3) row = nvec(r, g) < 0 ? -1 : m + nvec(r, g);
        else if (r == 3) row = g < m ? g : -1;
        else row = 4 + g < m ? 4 + g : -1;
        lohi[r * 4 + g] = row < 0 ? -INFINITY : rb[lb.LO + row];
        lohi[20 + r * 4 + g] = row < 0 ? INFINITY : rb[lb.HI + row];
      }
    {  // ---------------- forward record: SUB_F, ELIM_F ----------------
      Mat Ms((size_t)(n + m) * C3, 0.0);
      const Mat BK = mul(B, K, n, m, n), BPsi = mul(B, Psi, n, m, n);
      put_block(Ms, C3, 0, 0, sub(A, BK), n, n, 1.0);        // x+ = (A - B K) x - B Psi t - B d
      put_block(Ms, C3, 0, n, BPsi, n, n, -1.0);
      put_block(Ms, C3, 0, 2 * n, B, n, m, -1.0);
      put_block(Ms, C3, n, 0, K, m, n, -1.0);                // u  = -K x - Psi t - d
      put_block(Ms, C3, n, n, Psi, m, n, -1.0);
      put_block(Ms, C3, n, 2 * n, Im, m, m, -1.0);
      std::vector<double> urow((size_t)L.urows * L.ks_sub * 4 + 1, 0.0);
      urow_table(Ms, C3, urow.data());
      Mat Me((size_t)(2 * n + m) * C3, 0.0);                 // rows [mu+ ; deps ; db], cols [mu ; g^x ; g^u]
      if (f.alt_ok) {
        const double* rfe = &f.recFE[(size_t)k * f.RFE];
        const Mat FM = block(rfe, lfe.FM, n, n), GA = block(rfe, lfe.GA, n, m), PI = block(rfe, lfe.PI, n, n),
                  DK = block(rfe, lfe.DK, m, n), DG = block(rfe, lfe.DG, m, m), OB = block(rfe, lfe.OB, n, m);
        put_block(Me, C3, 0, 0, FM, n, n, 1.0);
        put_block(Me, C3, 0, n, PI, n, n, 1.0);
        put_block(Me, C3, 0, 2 * n, GA, n, m, 1.0);
        put_block(Me, C3, n, 0, mul(OB, DK, n, m, n), n, n, 1.0);
        put_block(Me, C3, n, 2 * n, mul(OB, DG, n, m, m), n, m, 1.0);
        put_block(Me, C3, 2 * n, 0, DK, m, n, 1.0);
        put_block(Me, C3, 2 * n, 2 * n, DG, m, m, 1.0);
      }
      for (const Out& w : outs) {
        unsigned char* o = write_product(&(*w.recMF)[(size_t)k * w.RMF], w.sub_f, Ms, mfma_es_sub_f(w.mode));
        o = write_product(o, w.elim_f, Me, mfma_es_elim_f(w.mode));
        std::memcpy(o, lohi, 40 * 8);
        std::memcpy(o + 40 * 8, urow.data(), (size_t)L.urows * L.ks_sub * 4 * 8);
      }
    }
    {  // ---------------- backward record: SUB_B, ELIM_B ----------------
      Mat Ms((size_t)(n + m) * C3, 0.0);                     // rows [x_k ; u], cols [x_{k+1} ; m_in ; db]
      if (f.alt_ok) {
        const double* rbe = &f.recBE[(size_t)k * f.RBE];
        const Mat PSB = block(rbe, lbe.PSB, m, n), KB = block(rbe, lbe.KB, m, n), AI = block(rbe, lbe.AI, n, n),
                  AIB = block(rbe, lbe.AIB, n, m);           // AIB = -A^-1 B:  x_k = AI x_{k+1} + AIB u
        put_block(Ms, C3, 0, 0, sub(AI, mul(AIB, KB, n, m, n)), n, n, 1.0);     // u = -KB x - PSB m_in - db
        put_block(Ms, C3, 0, n, mul(AIB, PSB, n, m, n), n, n, -1.0);
        put_block(Ms, C3, 0, 2 * n, AIB, n, m, -1.0);
        put_block(Ms, C3, n, 0, KB, m, n, -1.0);
        put_block(Ms, C3, n, n, PSB, m, n, -1.0);
        put_block(Ms, C3, n, 2 * n, Im, m, m, -1.0);
      }
      std::vector<double> urow((size_t)L.urows * L.ks_sub * 4 + 1, 0.0);
      urow_table(Ms, C3, urow.data());
      Mat Me((size_t)(2 * n + m) * C2, 0.0);                 // rows [t+ ; de ; d0], cols [p ; g^u]
      const Mat SIBT = mul(SI, BT, m, m, n);
      put_block(Me, C2, 0, 0, sub(AT, mul(KT, BT, n, m, n)), n, n, 1.0);        // t+ = A' p - K' (B' p + g^u)
      put_block(Me, C2, 0, n, KT, n, m, -1.0);
      put_block(Me, C2, n, 0, mul(OM, SIBT, n, m, n), n, n, 1.0);               // de = Om d0
      put_block(Me, C2, n, n, mul(OM, SI, n, m, m), n, m, 1.0);
      put_block(Me, C2, 2 * n, 0, SIBT, m, n, 1.0);                             // d0 = Si (B' p + g^u)
      put_block(Me, C2, 2 * n, n, SI, m, m, 1.0);
      for (const Out& w : outs) {
        unsigned char* o = write_product(&(*w.recMB)[(size_t)k * w.RMB], w.sub_b, Ms, mfma_es_sub_b(w.mode));
        o = write_product(o, w.elim_b, Me, mfma_es_elim_b(w.mode));
        std::memcpy(o, lohi, 40 * 8);
        std::memcpy(o + 40 * 8, urow.data(), (size_t)L.urows * L.ks_sub * 4 * 8);
      }
    }
  }
  });
}

}  // namespace

void set_factor_thread_cap(int cap) { g_thread_cap = cap; }

static int factorise_impl(const admm_problem& p, double rho, int segments, Factor& f, std::string& err, int mfma_mode, bool pack_scan_mfma, int ts_ranks) {
  if (mfma_mode < 0 || mfma_mode > 2) { err = "mfma_mode must be 0, 1 or 2"; return ADMM_ERR_INVALID; }
  if (mfma_mode != 0 && !mfma_dims(p.n, p.m)) { err = "the MFMA form needs n <= 12 and m <= 8"; return ADMM_ERR_UNSUPPORTED; }
  const int N = p.N, n = p.n, m = p.m;
  if (N < 1 || n < 1 || m < 1) { err = "N, n, m must be positive"; return ADMM_ERR_INVALID; }
  if (!(rho > 0.0) || !std::isfinite(rho)) { err = "rho must be positive and finite"; return ADMM_ERR_INVALID; }
  if (!p.A || !p.B || !p.Q || !p.R || !p.QN || !p.lo || !p.hi) { err = "A, B, Q, R, QN, lo, hi must be non-NULL"; return ADMM_ERR_INVALID; }
  const int nst = p.time_varying ? N : 1;
  if (!all_finite(p.A, (size_t)nst * n * n) || !all_finite(p.B, (size_t)nst * n * m) ||
      !all_finite(p.Q, (size_t)n * n) || !all_finite(p.R, (size_t)m * m) || !all_finite(p.QN, (size_t)n * n)) {
    err = "non-finite entry in A, B, Q, R or QN";
    return ADMM_ERR_INVALID;
  }
  int S = segments;
  if (S < 1) S = 1;
  if (S > N) S = N;

  f.N = N; f.n = n; f.m = m; f.S = S; f.rho = rho;
  f.ts_ranks = (ts_ranks > 1 && S % ts_ranks == 0) ? ts_ranks : 0;
  f.RB = rec_b_size(n, m); f.RF = rec_f_size(n, m); f.RS = rec_s_size(n);
  f.seg_start.resize(S + 1);
  for (int s = 0; s <= S; ++s) f.seg_start[s] = (int32_t)(((int64_t)s * N) / S);
  f.recB.assign((size_t)N * f.RB, 0.0);
  f.recF.assign((size_t)N * f.RF, 0.0);
  f.recS.assign((size_t)S * f.RS, 0.0);
  f.K.assign((size_t)N * m * n, 0.0);
  f.Sinv.assign((size_t)N * m * m, 0.0);

  const Mat Q = from_colmajor(p.Q, n, n), R = from_colmajor(p.R, m, m), QN = from_colmajor(p.QN, n, n);
  std::vector<Mat> A(N), B(N), Acl(N);

  PhaseTimer timer;
  // ---- Riccati sweep ----
  Mat P = QN;
  for (int i = 0; i < n; ++i) P[(size_t)i * n + i] += rho;
  symmetrise(P, n);
  for (int k = N - 1; k >= 0; --k) {
    A[k] = from_colmajor(p.A + (p.time_varying ? (size_t)k * n * n : 0), n, n);
    B[k] = from_colmajor(p.B + (p.time_varying ? (size_t)k * n * m : 0), n, m);
    const Mat Bt = tr(B[k], n, m), At = tr(A[k], n, n);
    const Mat PB = mul(P, B[k], n, n, m);
    Mat S_ = mul(Bt, PB, m, n, m);
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < m; ++j) S_[(size_t)i * m + j] += R[(size_t)i * m + j] + (i == j ? rho : 0.0);
    symmetrise(S_, m);
    Mat Si;
    if (!spd_inverse(S_, m, Si)) {
      err = "R + rho I + B'PB is not positive definite at stage " + std::to_string(k);
      return ADMM_ERR_NUMERIC;
    }
    symmetrise(Si, m);
    const Mat PA = mul(P, A[k], n, n, n);
    const Mat BtPA = mul(Bt, PA, m, n, n);
    const Mat K = mul(Si, BtPA, m, m, n);
    const Mat AtPA = mul(At, PA, n, n, n);
    const Mat SK = mul(S_, K, m, m, n);
    const Mat KtSK = mul(tr(K, m, n), SK, n, m, n);
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j)
        P[(size_t)i * n + j] = Q[(size_t)i * n + j] + (i == j ? rho : 0.0) + AtPA[(size_t)i * n + j] - KtSK[(size_t)i * n + j];
    symmetrise(P, n);
    for (size_t i = 0; i < P.size(); ++i)
      if (!std::isfinite(P[i])) { err = "Riccati recursion diverged at stage " + std::to_string(k); return ADMM_ERR_NUMERIC; }
    // closed loop
    Acl[k] = A[k];
    const Mat BK = mul(B[k], K, n, m, n);
    for (size_t i = 0; i < BK.size(); ++i) Acl[k][i] -= BK[i];

    for (size_t i = 0; i < K.size(); ++i) f.K[(size_t)k * m * n + i] = K[i];
    for (size_t i = 0; i < Si.size(); ++i) f.Sinv[(size_t)k * m * m + i] = Si[i];

    // stage-local parts of the records (offsets: admm_layout.hpp)
    const RecBLayout lb = rec_b_layout(n, m);
    const RecFLayout lf = rec_f_layout(n, m);
    double* rb = &f.recB[(size_t)k * f.RB];
    const Mat KT = tr(K, m, n);
    for (size_t i = 0; i < At.size(); ++i) rb[lb.AT + i] = At[i];
    for (size_t i = 0; i < Bt.size(); ++i) rb[lb.BT + i] = Bt[i];
    for (size_t i = 0; i < Si.size(); ++i) rb[lb.SI + i] = Si[i];
    for (size_t i = 0; i < KT.size(); ++i) rb[lb.KT + i] = KT[i];
    // Om filled below
    double* rf = &f.recF[(size_t)k * f.RF];
    // Psi filled below
    for (size_t i = 0; i < K.size(); ++i) rf[lf.K + i] = K[i];
    for (size_t i = 0; i < A[k].size(); ++i) rf[lf.A + i] = A[k][i];
    for (size_t i = 0; i < B[k].size(); ++i) rf[lf.B + i] = B[k][i];
    // the box of block k rides at the tail of both records
    const int nb = n + m;
    const double* blo = p.lo + (p.stage_bounds ? (size_t)k * nb : 0);
    const double* bhi = p.hi + (p.stage_bounds ? (size_t)k * nb : 0);
    for (int r = 0; r < nb; ++r) {
      rb[lb.LO + r] = blo[r];
      rb[lb.HI + r] = bhi[r];
      rf[lf.LO + r] = blo[r];
      rf[lf.HI + r] = bhi[r];
    }
    const double ub = p.unorm ? p.unorm[p.stage_bounds ? k : 0] : INFINITY;   // thrust-magnitude bound
    rb[lb.UB] = ub;
    rf[lf.UB] = ub;
  }

  timer.lap("riccati sweep");
  // ---- segment algebra ----
  std::atomic<bool> seg_overflow{false};
  parallel_for(S, [&](int s_begin, int s_end) {
  for (int s = s_begin; s < s_end; ++s) {
    const int a = f.seg_start[s], b = f.seg_start[s + 1];
    Mat Lam = eye(n);             // Lambda_k = Acl_{b-1} ... Acl_{k+1}
    Mat Xi((size_t)n * n, 0.0);
    for (int k = b - 1; k >= a; --k) {
      const Mat Si(f.Sinv.begin() + (size_t)k * m * m, f.Sinv.begin() + (size_t)(k + 1) * m * m);
      const Mat Bt = tr(B[k], n, m);
      Mat Om = mul(Lam, B[k], n, n, m);
      for (auto& v : Om) v = -v;
      const Mat Phi = tr(Lam, n, n);
      const Mat Psi = mul(Si, mul(Bt, Phi, m, n, n), m, m, n);
      const Mat OP = mul(Om, Psi, n, m, n);
      for (size_t i = 0; i < Xi.size(); ++i) Xi[i] += OP[i];
      double* rb = &f.recB[(size_t)k * f.RB + rec_b_layout(n, m).OM];
      for (size_t i = 0; i < Om.size(); ++i) rb[i] = Om[i];
      double* rf = &f.recF[(size_t)k * f.RF + rec_f_layout(n, m).PSI];
      for (size_t i = 0; i < Psi.size(); ++i) rf[i] = Psi[i];
      Lam = mul(Lam, Acl[k], n, n, n);
    }
    // Lam is now Acl_{b-1} ... Acl_a
    double* rs = &f.recS[(size_t)s * f.RS];
    const Mat Phis = tr(Lam, n, n);
    for (size_t i = 0; i < Phis.size(); ++i) rs[i] = Phis[i];
    for (size_t i = 0; i < Xi.size(); ++i) rs[(size_t)n * n + i] = Xi[i];
    for (size_t i = 0; i < Lam.size(); ++i) rs[(size_t)2 * n * n + i] = Lam[i];
    for (int i = 0; i < f.RS; ++i)
      if (!std::isfinite(rs[i])) seg_overflow = true;
  }
  });
  if (seg_overflow) { err = "segment transfer matrices overflowed; use more segments"; return ADMM_ERR_NUMERIC; }

  timer.lap("segment algebra");
  // ---- scan matrix ----
  {
    const int Sn = S * n;
    auto round_up = [](int v, int q) { return ((v + q - 1) / q) * q; };
    const int Mt = round_up(Sn, 16 * SCAN_MT);          // rows of the t_in half (padded)
    const int M = 2 * Mt;
    const int K = round_up(2 * Sn + n, 4 * SCAN_KALIGN);
    f.scanM = M; f.scanMt = Mt; f.scanK = K;
    f.scanW.assign((size_t)M * K, 0.0);
    auto blk = [&](const double* rs, int which) { return Mat(rs + (size_t)which * n * n, rs + (size_t)(which + 1) * n * n); };
    std::vector<Mat> Phi(S), Xi(S), Th(S);
    for (int s = 0; s < S; ++s) {
      const double* rs = &f.recS[(size_t)s * f.RS];
      Phi[s] = blk(rs, 0); Xi[s] = blk(rs, 1); Th[s] = blk(rs, 2);
    }
    const int c_t = 0, c_x0 = Sn, c_e = Sn + n;           // column offsets of tseg | x0 | eseg
    const int r_t = 0, r_x = Mt;                          // row offsets of t_in | x_in
    // The chains of xscan_kernel applied to row blocks of W (rowblock_step):
    //   t_in(S-1) = 0;   t_in(s-1) = tseg(s) + Phi_s t_in(s)
    //   x_in(0)   = x0;  x_in(s+1) = eseg(s) + Xi_s t_in(s) + Th_s x_in(s)
    for (int s = S - 1; s >= 1; --s)
      rowblock_step(f.scanW, K, n, r_t + (s - 1) * n, &Phi[s], r_t + s * n, nullptr, -1, c_t + s * n);
    rowblock_step(f.scanW, K, n, r_x, nullptr, -1, nullptr, -1, c_x0);
    for (int s = 0; s + 1 < S; ++s)
      rowblock_step(f.scanW, K, n, r_x + (s + 1) * n, &Th[s], r_x + s * n, &Xi[s], r_t + s * n, c_e + s * n);
    for (double v : f.scanW)
      if (!std::isfinite(v)) { err = "scan matrix overflowed; use more segments"; return ADMM_ERR_NUMERIC; }
    f.scanWp.clear();
    f.scanRange.clear();
    timeshard_columns(f.scanW, M, K, S, n, f.ts_ranks);
    if (pack_scan_mfma) pack_scan(f.scanW, M, K, f.scanWp, f.scanRange);
  }

  timer.lap("scan matrix");
  build_alternating(f, A, B, Q, R, QN, rho, pack_scan_mfma);
  timer.lap("alternating form");
  f.mfma_mode = mfma_mode;
  f.recMF.clear();
  f.recMB.clear();
  f.recMF64.clear();
  f.recMB64.clear();
  if (mfma_mode) build_mfma(f, mfma_mode);
  timer.lap("mfma records");
  return ADMM_OK;
}

// No exception leaves the factorisation (C ABI above it; worker threads below it -- parallel_for rethrows theirs here).
int factorise(const admm_problem& p, double rho, int segments, Factor& f, std::string& err, int mfma_mode, bool pack_scan_mfma, int ts_ranks) {
  try {
    return factorise_impl(p, rho, segments, f, err, mfma_mode, pack_scan_mfma, ts_ranks);
  } catch (const std::bad_alloc&) {
    err = "out of host memory in the KKT factorisation";
    return ADMM_ERR_ALLOC;
  } catch (const std::exception& e) {
    err = std::string("KKT factorisation: ") + e.what();
    return ADMM_ERR_ALLOC;
  }
}

}  // namespace admm

// admm_factor.cpp -- see admm_factor.hpp.  Plain C++ (no HIP): runs on the host
// once per admm_setup (and again only if rho changes).
#include "admm_factor.hpp"

#include <cmath>

namespace admm {
namespace {

using Mat = std::vector<double>;  // row-major

Mat mul(const Mat& a, const Mat& b, int p, int q, int r) {
  Mat c((size_t)p * r, 0.0);
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

}  // namespace

int factorise(const admm_problem& p, double rho, int segments, Factor& f, std::string& err) {
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

  f.N = N; f.n = n; f.m = m; f.S = S;
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

  // ---- segment algebra ----
  for (int s = 0; s < S; ++s) {
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
      if (!std::isfinite(rs[i])) { err = "segment transfer matrices overflowed; use more segments"; return ADMM_ERR_NUMERIC; }
  }

  // ---- scan matrix ----
  {
    const int Sn = S * n;
    auto round_up = [](int v, int q) { return ((v + q - 1) / q) * q; };
    const int Mt = round_up(Sn, 16 * SCAN_MT);          // rows of the t_in half (padded)
    const int M = 2 * Mt;
    const int K = round_up(2 * Sn + n, 4 * SCAN_KALIGN);
    f.scanM = M; f.scanMt = Mt; f.scanK = K;
    f.scanW.assign((size_t)M * K, 0.0);
    auto W = [&](int r, int c) -> double& { return f.scanW[(size_t)r * K + c]; };
    auto blk = [&](const double* rs, int which) { return Mat(rs + (size_t)which * n * n, rs + (size_t)(which + 1) * n * n); };
    std::vector<Mat> Phi(S), Xi(S), Th(S);
    for (int s = 0; s < S; ++s) {
      const double* rs = &f.recS[(size_t)s * f.RS];
      Phi[s] = blk(rs, 0); Xi[s] = blk(rs, 1); Th[s] = blk(rs, 2);
    }
    const int c_t = 0, c_x0 = Sn, c_e = Sn + n;           // column offsets of tseg | x0 | eseg
    const int r_t = 0, r_x = Mt;                          // row offsets of t_in | x_in
    // t_in(s) = sum_{s' > s} [Phi_{s+1} ... Phi_{s'-1}] tseg(s')
    std::vector<std::vector<Mat>> MtB(S, std::vector<Mat>(S));   // MtB[s][s'] (s' > s)
    for (int s = 0; s < S; ++s) {
      Mat P = eye(n);
      for (int sp = s + 1; sp < S; ++sp) {
        MtB[s][sp] = P;
        for (int i = 0; i < n; ++i)
          for (int j = 0; j < n; ++j) W(r_t + s * n + i, c_t + sp * n + j) = P[(size_t)i * n + j];
        P = mul(P, Phi[sp], n, n, n);
      }
    }
    // x_in(s) = [Th_{s-1}...Th_0] x0 + sum_{s' < s} [Th_{s-1}...Th_{s'+1}] (eseg(s') + Xi_{s'} t_in(s'))
    for (int s = 0; s < S; ++s) {
      Mat P = eye(n);
      for (int sp = s - 1; sp >= 0; --sp) {
        for (int i = 0; i < n; ++i)
          for (int j = 0; j < n; ++j) W(r_x + s * n + i, c_e + sp * n + j) = P[(size_t)i * n + j];
        const Mat PX = mul(P, Xi[sp], n, n, n);            // U(s, s') Xi_{s'}
        for (int spp = sp + 1; spp < S; ++spp) {           // ... times Mt(s', s'')
          const Mat G = mul(PX, MtB[sp][spp], n, n, n);
          for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) W(r_x + s * n + i, c_t + spp * n + j) += G[(size_t)i * n + j];
        }
        P = mul(P, Th[sp], n, n, n);
      }
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) W(r_x + s * n + i, c_x0 + j) = P[(size_t)i * n + j];
    }
    for (double v : f.scanW)
      if (!std::isfinite(v)) { err = "scan matrix overflowed; use more segments"; return ADMM_ERR_NUMERIC; }
    // pack in MFMA A-fragment order and find each M-group's non-zero k-step range
    const int mtiles = M / 16, ksteps = K / 4, groups = mtiles / SCAN_MT;
    f.scanWp.assign((size_t)ksteps * mtiles * 64, 0.0);
    for (int ks = 0; ks < ksteps; ++ks)
      for (int mt = 0; mt < mtiles; ++mt)
        for (int lane = 0; lane < 64; ++lane)
          f.scanWp[((size_t)ks * mtiles + mt) * 64 + lane] = W(16 * mt + (lane & 15), 4 * ks + (lane >> 4));
    f.scanRange.assign((size_t)2 * groups, 0);
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
      f.scanRange[2 * g] = kb;
      f.scanRange[2 * g + 1] = ke;
    }
  }
  return ADMM_OK;
}

}  // namespace admm

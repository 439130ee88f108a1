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


// W (row-major M x K) -> MFMA A-fragment order, plus each M-group's non-zero k-step range.
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
void put_block(std::vector<double>& W, int K, int r0, int c0, const Mat& a, int n, bool accumulate) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double& w = W[(size_t)(r0 + i) * K + c0 + j];
      w = accumulate ? w + a[(size_t)i * n + j] : a[(size_t)i * n + j];
    }
}

// Forward-elimination (information-filter) form of the x-update and its segment algebra
// (DESIGN.md §4.8).  With Rr = R + rho I, Qr_k = Q (QN at k = N) + rho I and C_0 = 0:
//     Pm      = A_k C_k A_k' + B_k Rr^{-1} B_k'
//     G_{k+1} = Pm (Pm + Qr_{k+1}^{-1})^{-1},       C_{k+1} = Pm - G_{k+1} Pm
// elimination (forward):   m_{k+1} = F_k m_k + Gam_k g^u_k + Pi_k g^x_{k+1},   m_0 = x_0
// substitution (backward, costate form; lam = 0 after the last block):
//     x_{k+1} = m_{k+1} + C_{k+1} lam,   nu = lam - Qr_{k+1} x_{k+1} - g^x_{k+1},
//     u_k = Rr^{-1} (B_k' nu - g^u_k),   lam <- A_k' nu
// Only inverses of positive definite matrices occur (C_k itself is singular for small k).
// Leaves f.alt_ok false if a pivot fails or a transfer matrix overflows.
void build_alternating(Factor& f, const std::vector<Mat>& A, const std::vector<Mat>& B, const Mat& Q,
                       const Mat& R, const Mat& QN, double rho) {
  const int N = f.N, n = f.n, m = f.m, S = f.S;
  f.alt_ok = false;
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

  std::vector<Mat> Fm(N), Gam(N), Pi(N), E(N), Cn(N), Qk(N);
  Mat C((size_t)n * n, 0.0);
  for (int k = 0; k < N; ++k) {
    const Mat At = tr(A[k], n, n), Bt = tr(B[k], n, m);
    const Mat BRi = mul(B[k], Ri, n, m, m);
    Mat Pm = add(mul(mul(A[k], C, n, n, n), At, n, n, n), mul(BRi, Bt, n, m, n));
    symmetrise(Pm, n);
    const Mat& Qn = (k + 1 == N) ? QNr : Qr;
    const Mat& Qin = (k + 1 == N) ? QNi : Qi;
    Mat D = add(Pm, Qin);
    symmetrise(D, n);
    Mat Di;
    if (!spd_inverse(D, n, Di)) return;
    const Mat G = mul(Pm, Di, n, n, n);
    const Mat ImG = sub(I, G);
    C = sub(Pm, mul(G, Pm, n, n, n));
    symmetrise(C, n);
    Fm[k] = mul(ImG, A[k], n, n, n);
    Gam[k] = neg(mul(ImG, BRi, n, n, m));
    Pi[k] = neg(mul(G, Qin, n, n, n));
    E[k] = mul(At, sub(I, mul(Qn, C, n, n, n)), n, n, n);
    Cn[k] = C;
    Qk[k] = Qn;
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
    put(rbe + lbe.CM, C);
    put(rbe + lbe.QM, Qn);
    put(rbe + lbe.RB, mul(Ri, Bt, m, m, n));
    put(rbe + lbe.RI, Ri);
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

  // ---- segment algebra ----
  //   Phf_k = F_k ... F_a,   Omb_k = (E_a ... E_{k-1}) A_k'
  //   eps(s) = - sum_k Omb_k (Qr_{k+1} m0_{k+1} + g^x_{k+1}) = sum_k (YU_k g^u_k + YX_k g^x_{k+1})
  //   lam_out(s) = eps(s) + Es(s) lam_in(s) + Xib(s) m_in(s),   m_out(s) = mseg(s) + Phs(s) m_in(s)
  std::vector<Mat> Phs(S), Es(S), Xib(S);
  for (int s = 0; s < S; ++s) {
    const int a = f.seg_start[s], b = f.seg_start[s + 1];
    std::vector<Mat> Phf(b - a), Omb(b - a);
    Mat P = I, Ep = I;
    for (int k = a; k < b; ++k) {
      P = mul(Fm[k], P, n, n, n);
      Phf[k - a] = P;
      Omb[k - a] = mul(Ep, tr(A[k], n, n), n, n, n);
      Ep = mul(Ep, E[k], n, n, n);
      put(&f.recBE[(size_t)k * f.RBE + lbe.PHF], P);
    }
    Phs[s] = P;
    Es[s] = Ep;
    Mat Z((size_t)n * n, 0.0), Xi((size_t)n * n, 0.0);
    for (int j = b - 1; j >= a; --j) {
      const Mat OQ = mul(Omb[j - a], Qk[j], n, n, n);
      Z = (j + 1 < b) ? add(OQ, mul(Z, Fm[j + 1], n, n, n)) : OQ;
      put(&f.recFE[(size_t)j * f.RFE + lfe.YU], neg(mul(Z, Gam[j], n, n, m)));
      put(&f.recFE[(size_t)j * f.RFE + lfe.YX], sub(neg(mul(Z, Pi[j], n, n, n)), Omb[j - a]));
      Xi = sub(Xi, mul(OQ, Phf[j - a], n, n, n));
    }
    Xib[s] = Xi;
  }
  for (int k = 0; k < N; ++k) {                       // (the box blocks may hold +-inf: matrices only)
    for (int i = 0; i < lfe.LO; ++i) if (!std::isfinite(f.recFE[(size_t)k * f.RFE + i])) return;
    for (int i = 0; i < lbe.LO; ++i) if (!std::isfinite(f.recBE[(size_t)k * f.RBE + i])) return;
  }

  // ---- scan matrix, same shape and layout as scanW ----
  const int Sn = S * n, Mt = f.scanMt, M = f.scanM, K = f.scanK;
  f.scanWB.assign((size_t)M * K, 0.0);
  const int c_m = 0, c_x0 = Sn, c_e = Sn + n, r_m = 0, r_l = Mt;
  // m_in(s) = [Phs_{s-1} ... Phs_0] x0 + sum_{s' < s} [Phs_{s-1} ... Phs_{s'+1}] mseg(s')
  std::vector<std::vector<Mat>> Mm(S, std::vector<Mat>(S));   // Mm[s][s'] (s' < s)
  std::vector<Mat> Mx0(S);
  for (int s = 0; s < S; ++s) {
    Mat P = I;
    for (int sp = s - 1; sp >= 0; --sp) {
      Mm[s][sp] = P;
      put_block(f.scanWB, K, r_m + s * n, c_m + sp * n, P, n, false);
      P = mul(P, Phs[sp], n, n, n);
    }
    Mx0[s] = P;
    put_block(f.scanWB, K, r_m + s * n, c_x0, P, n, false);
  }
  // lam_in(s) = sum_{s' > s} [Es_{s+1} ... Es_{s'-1}] (eps(s') + Xib_{s'} m_in(s'))
  for (int s = 0; s < S; ++s) {
    Mat P = I;
    for (int sp = s + 1; sp < S; ++sp) {
      put_block(f.scanWB, K, r_l + s * n, c_e + sp * n, P, n, false);
      const Mat PX = mul(P, Xib[sp], n, n, n);
      put_block(f.scanWB, K, r_l + s * n, c_x0, mul(PX, Mx0[sp], n, n, n), n, true);
      for (int spp = 0; spp < sp; ++spp)
        put_block(f.scanWB, K, r_l + s * n, c_m + spp * n, mul(PX, Mm[sp][spp], n, n, n), n, true);
      P = mul(P, Es[sp], n, n, n);
    }
  }
  for (double v : f.scanWB) if (!std::isfinite(v)) return;
  pack_scan(f.scanWB, M, K, f.scanWpB, f.scanRangeB);
  f.alt_ok = true;
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
    pack_scan(f.scanW, M, K, f.scanWp, f.scanRange);
  }

  build_alternating(f, A, B, Q, R, QN, rho);
  return ADMM_OK;
}

}  // namespace admm

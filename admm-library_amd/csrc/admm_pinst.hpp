// admm_pinst.hpp -- per-instance dynamics (admm_problem.time_varying = 2): every QP has its own
// A_k, B_k (and, with stage_bounds = 2, its own per-stage box) -- the QP class a batched successive-convexification
// loop produces.  No reference counterpart exists (README.md:1-2 only).
//
// Nothing is shared by the batch any more, so nothing is wave-uniform: the KKT factor is per QP and is computed ON THE
// DEVICE (pfactor_kernel: one lane runs the Riccati recursion of its QP), and the sweeps read their stage operators
// from HBM per lane instead of from LDS broadcasts.  All per-instance arrays are batch-minor like the state:
//     X[(k * E + e) * pitch + col]          element e of stage k of QP col
// so every wave-level access is still one contiguous 512-B row segment.
//   Ad [k][n*n] A_k and Bd [k][n*m] B_k, COLUMN-major per stage (the ABI's own order: the upload is one transposition
//   of the caller's array)      Kd [k][m*n] row-major      Sd [k][m*m] = (R + rho I + B'PB)^-1
//   lod / hid [k][m+n]  (stage_bounds = 2; otherwise the shared expanded arrays lo / hi [L])
// (Wide shapes -- (8, 4), (12, 6), ...: one lane cannot hold a QP; admm_pinst_rows.hpp / admm_pinst_wide.hpp run every kernel with a QP's rows
//  spread over the lanes of a wave, and keep these operand arrays TILED: see Operand / StageTile below.)
// Iteration = the plain path: pxb_kernel  backward sweep (d rows -> dbuf)      pxfz_kernel  forward rollout + z-update + dual
// + residuals; one lane sweeps one SEGMENT of one QP's horizon (pseg_kernel / pscan_kernel below: per-QP transfer matrices,
// computed on the device), the whole horizon when the batch alone fills the chip.
// Algorithmic HBM bytes per stacked element and iteration (n = 6, m = 3, fp64, v-form):
//     factor operands  A 36 + B 18 + K 18 + S 9 = 81 doubles (backward), K + A + B = 72 (forward)  -> 153 x 8 / 9 = 136 B
//     state            v 8 + d 2.67 (pxb), d 2.67 + v 8 + v+ 8 (pxfz)                              ->  29.33 B
//     per-instance box lo, hi read by both kernels                                                  ->  32 B
// i.e. ~200 B/element against 21-29 B for batch-shared dynamics: this path is bound by FACTOR traffic (SURVEY.md §7
// "factor traffic can dominate"), a different roofline from the headline's.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>

namespace admm {

constexpr int PI_THREADS = 64;      // one wave per workgroup: small batches still spread over the CUs

// ---------------------------------------------------------------------------
// Riccati factorisation of every QP, one lane per QP, stages N-1 .. 0:
//     S = R + rho I + B'PB,  Si = S^-1,  K = Si B'PA,  P <- Q + rho I + A'PA - K'SK      (P_N = QN + rho I)
// Q, R, QN: shared, row-major.  *fail is set if some S is not positive definite / not finite.
// ---------------------------------------------------------------------------
template <int NX, int NU>
__global__ __launch_bounds__(PI_THREADS) void pfactor_kernel(
    const double* __restrict__ Ad, const double* __restrict__ Bd, const double* __restrict__ Qm,
    const double* __restrict__ Rm, const double* __restrict__ QNm, const double* __restrict__ rhov,
    const int* __restrict__ todo, double* __restrict__ Kd, double* __restrict__ Sd, int* __restrict__ fail, int N, int pitch,
    int batch, int* __restrict__ qflag) {
  // qflag (may be NULL): per-QP verdict of a TRIAL factorisation into scratch K / S (admm_api.hip, pinst_try): bit 0 = some
  // S_k of this QP is not positive definite
  const int col = blockIdx.x * PI_THREADS + threadIdx.x;
  if (col >= pitch) return;
  if (todo && !todo[col]) return;           // per-QP adaptive rho: only the QPs whose rho has just changed are refactored
  const double rho = rhov[col];             // every QP carries its own rho (all equal unless the adaptive rule is on)
  const size_t P_ = (size_t)pitch;
  const bool real = col < batch;            // pad columns hold zero dynamics: factor them like the rest (finite), never flag them
  double P[NX][NX];
#pragma unroll
  for (int i = 0; i < NX; ++i)
#pragma unroll
    for (int j = 0; j < NX; ++j) P[i][j] = QNm[i * NX + j] + (i == j ? rho : 0.0);
  bool bad = false;
  for (int k = N - 1; k >= 0; --k) {
    double A[NX][NX], B[NX][NU];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
#pragma unroll
      for (int j = 0; j < NX; ++j) A[i][j] = Ad[((size_t)k * NX * NX + j * NX + i) * P_ + col];
#pragma unroll
      for (int j = 0; j < NU; ++j) B[i][j] = Bd[((size_t)k * NX * NU + j * NX + i) * P_ + col];
    }
    double S[NU][NU], Si[NU][NU];
    {
      double PB[NX][NU];
#pragma unroll
      for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int j = 0; j < NU; ++j) {
          double a = 0.0;
#pragma unroll
          for (int l = 0; l < NX; ++l) a = fma(P[i][l], B[l][j], a);
          PB[i][j] = a;
        }
#pragma unroll
      for (int a_ = 0; a_ < NU; ++a_)
#pragma unroll
        for (int b_ = 0; b_ < NU; ++b_) {
          double a = Rm[a_ * NU + b_] + (a_ == b_ ? rho : 0.0);
#pragma unroll
          for (int i = 0; i < NX; ++i) a = fma(B[i][a_], PB[i][b_], a);
          S[a_][b_] = a;
        }
    }
#pragma unroll
    for (int a_ = 0; a_ < NU; ++a_)
#pragma unroll
      for (int b_ = a_ + 1; b_ < NU; ++b_) { const double v = 0.5 * (S[a_][b_] + S[b_][a_]); S[a_][b_] = S[b_][a_] = v; }
    {  // Gauss-Jordan inverse of the SPD m x m matrix (no pivoting needed; a non-positive pivot flags failure)
      double W[NU][NU];
#pragma unroll
      for (int a_ = 0; a_ < NU; ++a_)
#pragma unroll
        for (int b_ = 0; b_ < NU; ++b_) { W[a_][b_] = S[a_][b_]; Si[a_][b_] = (a_ == b_) ? 1.0 : 0.0; }
#pragma unroll
      for (int c = 0; c < NU; ++c) {
        const double pv = W[c][c];
        if (!(pv > 0.0) || !(pv < INFINITY)) bad = true;
        const double ip = 1.0 / pv;
#pragma unroll
        for (int j = 0; j < NU; ++j) { W[c][j] *= ip; Si[c][j] *= ip; }
#pragma unroll
        for (int r = 0; r < NU; ++r) {
          if (r == c) continue;
          const double f = W[r][c];
#pragma unroll
          for (int j = 0; j < NU; ++j) { W[r][j] = fma(-f, W[c][j], W[r][j]); Si[r][j] = fma(-f, Si[c][j], Si[r][j]); }
        }
      }
#pragma unroll
      for (int a_ = 0; a_ < NU; ++a_)
#pragma unroll
        for (int b_ = a_ + 1; b_ < NU; ++b_) { const double v = 0.5 * (Si[a_][b_] + Si[b_][a_]); Si[a_][b_] = Si[b_][a_] = v; }
    }
    double PA[NX][NX];
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
      for (int l = 0; l < NX; ++l) {
        double a = 0.0;
#pragma unroll
        for (int r = 0; r < NX; ++r) a = fma(P[i][r], A[r][l], a);
        PA[i][l] = a;
      }
    double K[NU][NX];
    {
      double BtPA[NU][NX];
#pragma unroll
      for (int j = 0; j < NU; ++j)
#pragma unroll
        for (int l = 0; l < NX; ++l) {
          double a = 0.0;
#pragma unroll
          for (int i = 0; i < NX; ++i) a = fma(B[i][j], PA[i][l], a);
          BtPA[j][l] = a;
        }
#pragma unroll
      for (int j = 0; j < NU; ++j)
#pragma unroll
        for (int l = 0; l < NX; ++l) {
          double a = 0.0;
#pragma unroll
          for (int t = 0; t < NU; ++t) a = fma(Si[j][t], BtPA[t][l], a);
          K[j][l] = a;
        }
    }
    // P <- Q + rho I + A'PA - K' S K
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
      for (int l = 0; l < NX; ++l) {
        double a = Qm[i * NX + l] + (i == l ? rho : 0.0);
#pragma unroll
        for (int r = 0; r < NX; ++r) a = fma(A[r][i], PA[r][l], a);
        P[i][l] = a;
      }
    {
      double SK[NU][NX];
#pragma unroll
      for (int j = 0; j < NU; ++j)
#pragma unroll
        for (int l = 0; l < NX; ++l) {
          double a = 0.0;
#pragma unroll
          for (int t = 0; t < NU; ++t) a = fma(S[j][t], K[t][l], a);
          SK[j][l] = a;
        }
#pragma unroll
      for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int l = 0; l < NX; ++l) {
          double a = P[i][l];
#pragma unroll
          for (int j = 0; j < NU; ++j) a = fma(-K[j][i], SK[j][l], a);
          P[i][l] = a;
        }
    }
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
      for (int l = i + 1; l < NX; ++l) { const double v = 0.5 * (P[i][l] + P[l][i]); P[i][l] = P[l][i] = v; }
#pragma unroll
    for (int j = 0; j < NU; ++j) {
#pragma unroll
      for (int l = 0; l < NX; ++l) Kd[((size_t)k * NU * NX + j * NX + l) * P_ + col] = K[j][l];
#pragma unroll
      for (int t = 0; t < NU; ++t) Sd[((size_t)k * NU * NU + j * NU + t) * P_ + col] = Si[j][t];
    }
  }
  if (bad && real) {
    atomicOr(fail, 1);
    if (qflag) atomicOr(&qflag[col], 1);
  }
}

// Thrust-magnitude bound ||u_k||_2 <= ub_k on the control rows of a block: the factor c with z_u = c v_u.
// sqrt and the division are the correctly rounded fp64 forms, accumulated in row order with fma -- the oracle's operations.
// ub = +inf (no bound at this stage) gives exactly 1.
template <int NU, int NB>
__device__ __forceinline__ double pi_soc_scale(const double (&vblk)[NB], double ub) {
  double ss = 0.0;
#pragma unroll
  for (int j = 0; j < NU; ++j) ss = fma(vblk[j], vblk[j], ss);
  const double nrm = sqrt(ss);
  return nrm > ub ? ub / nrm : 1.0;
}

// ---------------------------------------------------------------------------
// Stage operands of one QP, fetched as ONE batch of independent loads (the first version read each operand where it
// was used: ~100 dependent round trips per stage, 18-30 us per stage; batched: one round trip).  The sweeps keep TWO of
// these in registers and ping-pong: the loads of stage k -+ 1 are in flight while stage k is computed.
// ---------------------------------------------------------------------------
template <int NX, int NU, bool BACKWARD, bool HASQ, bool TWO /* second state array (z, y form) */, bool PB, bool BOUNDS, bool DIN,
          bool SEG = false /* segment operand: Omega_k [n][m] (backward) / Psi_k [m][n] (forward) */>
struct PiStage {
  static constexpr int NB = NX + NU;
  double A[NX * NX], B[NX * NU], K[NU * NX];
  double S[BACKWARD ? NU * NU : 1];
  double G[SEG ? NX * NU : 1];
  double s0[NB], s1[TWO ? NB : 1], q[HASQ ? NB : 1], lo[BOUNDS ? NB : 1], hi[BOUNDS ? NB : 1], d[DIN ? NU : 1];
  __device__ __forceinline__ void load(int k, const double* Ad, const double* Bd, const double* Kd, const double* Sd,
                                       const double* st0, const double* st1, const double* qd, const double* lod,
                                       const double* hid, const double* dd, size_t P_, int col, const double* segd = nullptr) {
    if (SEG) {
#pragma unroll
      for (int e = 0; e < NX * NU; ++e) G[e] = segd[((size_t)k * NX * NU + e) * P_ + col];
    }
#pragma unroll
    for (int e = 0; e < NX * NX; ++e) A[e] = Ad[((size_t)k * NX * NX + e) * P_ + col];
#pragma unroll
    for (int e = 0; e < NX * NU; ++e) B[e] = Bd[((size_t)k * NX * NU + e) * P_ + col];
#pragma unroll
    for (int e = 0; e < NU * NX; ++e) K[e] = Kd[((size_t)k * NU * NX + e) * P_ + col];
    if (BACKWARD) {
#pragma unroll
      for (int e = 0; e < NU * NU; ++e) S[e] = Sd[((size_t)k * NU * NU + e) * P_ + col];
    }
#pragma unroll
    for (int r = 0; r < NB; ++r) {
      const size_t o = ((size_t)k * NB + r) * P_ + col;
      s0[r] = st0[o];
      if (TWO) s1[r] = st1[o];
      if (HASQ) q[r] = qd[o];
      if (BOUNDS) {
        if (PB) { lo[r] = lod[o]; hi[r] = hid[o]; }
        else    { lo[r] = lod[(size_t)k * NB + r]; hi[r] = hid[(size_t)k * NB + r]; }
      }
    }
    if (DIN) {
#pragma unroll
      for (int j = 0; j < NU; ++j) d[j] = dd[((size_t)k * NU + j) * P_ + col];
    }
  }
};

// ---------------------------------------------------------------------------
// Backward sweep (as xb_kernel, one segment = the whole horizon):
//     g = q - rho (z - y);  p = g^x + t;  h = B'p + g^u;  d_k = Si h -> dbuf;  t = A'p - K'h
// A, B are column-major per stage: A[i][l] = A[l * NX + i], B[i][j] = B[j * NX + i]; K row-major [j][i].
// ---------------------------------------------------------------------------
// SEG (segments in time, see pseg_kernel): blockIdx.y = segment s, stages seg_start[s] .. seg_start[s+1]-1, zero tail;
// the segment also accumulates e = sum_k Omega_k d0_k and leaves (t, e) in tseg / eseg [s][n][pitch] for pscan_kernel.
template <int NX, int NU, bool HASQ, bool VFORM, bool PB, bool SEG = false, bool SOC = false>
__global__ __launch_bounds__(PI_THREADS) void pxb_kernel(
    const double* __restrict__ z, const double* __restrict__ y, const double* __restrict__ q,
    const double* __restrict__ Ad, const double* __restrict__ Bd, const double* __restrict__ Kd,
    const double* __restrict__ Sd, const double* __restrict__ lo, const double* __restrict__ hi,
    double* __restrict__ dbuf, const double* __restrict__ rhov, int N, int pitch,
    const double* __restrict__ Omd = nullptr, const int* __restrict__ seg_start = nullptr, double* __restrict__ tseg = nullptr,
    double* __restrict__ eseg = nullptr, const double* __restrict__ ubd = nullptr) {
  constexpr int NB = NX + NU;
  static_assert(!SOC || VFORM, "the (z, y) form carries the projected z: only the v-form projects");
  typedef PiStage<NX, NU, true, HASQ, !VFORM, PB, VFORM, false, SEG> Stage;
  const int col = blockIdx.x * PI_THREADS + threadIdx.x;
  if (col >= pitch) return;
  const double rho = rhov[col];
  const size_t P_ = (size_t)pitch;
  const int sg = SEG ? (int)blockIdx.y : 0;
  const int ka = SEG ? seg_start[sg] : 0, kb = SEG ? seg_start[sg + 1] : N;
  double t[NX], es[SEG ? NX : 1];
#pragma unroll
  for (int i = 0; i < NX; ++i) t[i] = 0.0;
#pragma unroll
  for (int i = 0; i < (SEG ? NX : 1); ++i) es[i] = 0.0;
  auto body = [&](const Stage& s, int k) {
    double g[NB];
    double cs = 1.0;                               // thrust-magnitude bound of this stage (branch-free: +inf -> 1)
    if constexpr (SOC) cs = pi_soc_scale<NU, NB>(s.s0, ubd[k]);
#pragma unroll
    for (int r = 0; r < NB; ++r) {
      double zz = s.s0[r], yy;
      if (VFORM) {
        zz = fmin(fmax((SOC && r < NU) ? s.s0[r] * cs : s.s0[r], s.lo[r]), s.hi[r]);
        yy = s.s0[r] - zz;
      } else {
        yy = s.s1[r];
      }
      g[r] = -rho * (zz - yy);
      if (HASQ) g[r] += s.q[r];
    }
    double p[NX], h[NU];
#pragma unroll
    for (int i = 0; i < NX; ++i) p[i] = g[NU + i] + t[i];
#pragma unroll
    for (int j = 0; j < NU; ++j) {
      double a = g[j];
#pragma unroll
      for (int i = 0; i < NX; ++i) a = fma(s.B[j * NX + i], p[i], a);
      h[j] = a;
    }
    double dv[NU];
#pragma unroll
    for (int j = 0; j < NU; ++j) {
      double a = 0.0;
#pragma unroll
      for (int l = 0; l < NU; ++l) a = fma(s.S[j * NU + l], h[l], a);
      dv[j] = a;
      dbuf[((size_t)k * NU + j) * P_ + col] = a;
    }
    if (SEG) {
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        double a = es[i];
#pragma unroll
        for (int j = 0; j < NU; ++j) a = fma(s.G[i * NU + j], dv[j], a);
        es[i] = a;
      }
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      double a = 0.0;
#pragma unroll
      for (int l = 0; l < NX; ++l) a = fma(s.A[i * NX + l], p[l], a);          // A'p: A[l][i] = A[i * NX + l]
#pragma unroll
      for (int j = 0; j < NU; ++j) a = fma(-s.K[j * NX + i], h[j], a);
      t[i] = a;
    }
  };
  const double* st0 = z;
  Stage sa, sb;
  sa.load(kb - 1, Ad, Bd, Kd, Sd, st0, y, q, lo, hi, nullptr, P_, col, Omd);
  for (int k = kb - 1; k >= ka; k -= 2) {
    sb.load(k - 1 >= ka ? k - 1 : ka, Ad, Bd, Kd, Sd, st0, y, q, lo, hi, nullptr, P_, col, Omd);
    body(sa, k);
    if (k - 1 < ka) break;
    sa.load(k - 2 >= ka ? k - 2 : ka, Ad, Bd, Kd, Sd, st0, y, q, lo, hi, nullptr, P_, col, Omd);
    body(sb, k - 1);
  }
  if (SEG) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      tseg[((size_t)sg * NX + i) * P_ + col] = t[i];
      eseg[((size_t)sg * NX + i) * P_ + col] = es[i];
    }
  }
}

// ---------------------------------------------------------------------------
// Forward rollout (+ z-update, dual ascent, residual partials when ZUP; + w stored when STOREW):
//     u = -K x - d;  x <- A x + B u;   ZUP: v+ = wh + y_old -> v, partials -> part[5][pitch]
// ZUP = false, STOREW = true is the read-out kernel (w of the last x-update).
// ---------------------------------------------------------------------------
// SEG: blockIdx.y = segment s; the rollout starts from x_in(s) with d_k = d0_k + Psi_k t_in(s) (xin / tin [s][n][pitch] from
// pscan_kernel); residual partials go to part[s][5][pitch] and are summed over the segments by the finalise kernel.
template <int NX, int NU, bool ZUP, bool RESID, bool RELAX, bool VIN, bool PB, bool STOREW, bool SEG = false, bool SOC = false>
__global__ __launch_bounds__(PI_THREADS) void pxfz_kernel(
    const double* __restrict__ dbuf, const double* __restrict__ x0, const double* __restrict__ Ad,
    const double* __restrict__ Bd, const double* __restrict__ Kd, const double* __restrict__ lo,
    const double* __restrict__ hi, const double* __restrict__ zin, const double* __restrict__ yin,
    double* __restrict__ v, double* __restrict__ w, double* __restrict__ part, double alpha, int N, int pitch,
    const double* __restrict__ Psd = nullptr, const int* __restrict__ seg_start = nullptr, const double* __restrict__ tin = nullptr,
    const double* __restrict__ xin = nullptr, const double* __restrict__ ubd = nullptr) {
  constexpr int NB = NX + NU;
  constexpr bool NEEDZ = RESID || RELAX;
  static_assert(!SOC || ZUP, "the read-out form projects nothing");
  // state operands: VIN: s0 = v;  else s0 = y and (NEEDZ) s1 = z
  typedef PiStage<NX, NU, false, false, ZUP && !VIN && NEEDZ, PB, ZUP, true, SEG> Stage;
  const int col = blockIdx.x * PI_THREADS + threadIdx.x;
  if (col >= pitch) return;
  const size_t P_ = (size_t)pitch;
  const int sg = SEG ? (int)blockIdx.y : 0;
  const int ka = SEG ? seg_start[sg] : 0, kb = SEG ? seg_start[sg + 1] : N;
  double x[NX], ti[SEG ? NX : 1];
#pragma unroll
  for (int i = 0; i < NX; ++i) x[i] = SEG ? xin[((size_t)sg * NX + i) * P_ + col] : x0[(size_t)i * P_ + col];
#pragma unroll
  for (int i = 0; i < (SEG ? NX : 1); ++i) ti[i] = SEG ? tin[((size_t)sg * NX + i) * P_ + col] : 0.0;
  double a_r = 0, a_s = 0, a_w = 0, a_z = 0, a_y = 0;
  auto body = [&](const Stage& s, int k) {
    double wv[NB];
#pragma unroll
    for (int j = 0; j < NU; ++j) {
      double a = s.d[j];
      if (SEG) {
#pragma unroll
        for (int i = 0; i < NX; ++i) a = fma(s.G[j * NX + i], ti[i], a);
      }
#pragma unroll
      for (int i = 0; i < NX; ++i) a = fma(s.K[j * NX + i], x[i], a);
      wv[j] = -a;
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      double a = 0.0;
#pragma unroll
      for (int l = 0; l < NX; ++l) a = fma(s.A[l * NX + i], x[l], a);
#pragma unroll
      for (int j = 0; j < NU; ++j) a = fma(s.B[j * NX + i], wv[j], a);
      wv[NU + i] = a;
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = wv[NU + i];
    // thrust-magnitude bound (as in xfz_kernel): the factors of the state before and after this z-update, branch-free
    double cs_old = 1.0, cs_new = 1.0;
    if constexpr (SOC) {
      const double ub = ubd[k];
      if (VIN) cs_old = pi_soc_scale<NU, NB>(s.s0, ub);
      if (RESID) {                                  // z+ needs ||v+_u||: the control rows of v+ first
        double vnew[NB];
#pragma unroll
        for (int r = 0; r < NB; ++r) {
          if (r < NU) {
            double zo, yo;
            if (VIN) { zo = fmin(fmax(s.s0[r] * cs_old, s.lo[r]), s.hi[r]); yo = s.s0[r] - zo; }
            else { yo = s.s0[r]; zo = NEEDZ ? s.s1[NEEDZ ? r : 0] : 0.0; }
            double wh = wv[r];
            if (RELAX) wh = fma(alpha, wv[r], (1.0 - alpha) * zo);
            vnew[r] = wh + yo;
          } else {
            vnew[r] = 0.0;
          }
        }
        cs_new = pi_soc_scale<NU, NB>(vnew, ub);
      }
    }
#pragma unroll
    for (int r = 0; r < NB; ++r) {
      const size_t o = ((size_t)k * NB + r) * P_ + col;
      const bool ball = SOC && r < NU;
      if (STOREW) w[o] = wv[r];
      if (ZUP) {
        double zo, yo;
        if (VIN) {
          zo = fmin(fmax(ball ? s.s0[r] * cs_old : s.s0[r], s.lo[r]), s.hi[r]);
          yo = s.s0[r] - zo;
        } else {
          yo = s.s0[r];
          zo = NEEDZ ? s.s1[NEEDZ ? r : 0] : 0.0;
        }
        double wh = wv[r];
        if (RELAX) wh = fma(alpha, wv[r], (1.0 - alpha) * zo);
        const double vn = wh + yo;
        v[o] = vn;
        if (RESID) {
          const double zn = fmin(fmax(ball ? vn * cs_new : vn, s.lo[r]), s.hi[r]);
          const double yn = vn - zn;
          const double dr = wv[r] - zn, ds = zn - zo;
          a_r = fma(dr, dr, a_r);
          a_s = fma(ds, ds, a_s);
          a_w = fma(wv[r], wv[r], a_w);
          a_z = fma(zn, zn, a_z);
          a_y = fma(yn, yn, a_y);
        }
      }
    }
  };
  const double* st0 = ZUP ? (VIN ? v : yin) : dbuf;      // (!ZUP: the state operands are unused; any valid array)
  Stage sa, sb;
  sa.load(ka, Ad, Bd, Kd, nullptr, st0, zin, nullptr, lo, hi, dbuf, P_, col, Psd);
  for (int k = ka; k < kb; k += 2) {
    sb.load(k + 1 < kb ? k + 1 : kb - 1, Ad, Bd, Kd, nullptr, st0, zin, nullptr, lo, hi, dbuf, P_, col, Psd);
    body(sa, k);
    if (k + 1 >= kb) break;
    sa.load(k + 2 < kb ? k + 2 : kb - 1, Ad, Bd, Kd, nullptr, st0, zin, nullptr, lo, hi, dbuf, P_, col, Psd);
    body(sb, k + 1);
  }
  if (ZUP && RESID) {
    double* pp = part + (size_t)sg * 5 * P_;
    pp[0 * P_ + col] = a_r;
    pp[1 * P_ + col] = a_s;
    pp[2 * P_ + col] = a_w;
    pp[3 * P_ + col] = a_z;
    pp[4 * P_ + col] = a_y;
  }
}

// ---------------------------------------------------------------------------
// Segments in time for per-instance dynamics (the segment algebra of DESIGN.md §4.2 with PER-QP transfer matrices, computed
// on the device).  One lane = one (QP, segment): stages b-1 .. a of the segment, Lam = Acl_{b-1} ... Acl_{k+1} (I at k = b-1),
// Acl = A - B K:
//     Omega_k = -Lam B_k  [n][m] -> Omd      Psi_k = Si_k B_k' Lam' = -Si_k Omega_k'  [m][n] -> Psd      Xi += Omega_k Psi_k
// and per segment Phi = Lam_a', Xi, Th = Lam_a (Lam_a = Acl_{b-1} ... Acl_a) -> Segd [s][3][n*n], row-major.
// *grow is set if an entry of a transfer matrix exceeds 100 in magnitude (the conditioning bound of admm_setup).
// ---------------------------------------------------------------------------
template <int NX, int NU>
__global__ __launch_bounds__(PI_THREADS) void pseg_kernel(
    const double* __restrict__ Ad, const double* __restrict__ Bd, const double* __restrict__ Kd, const double* __restrict__ Sd,
    const int* __restrict__ seg_start, const int* __restrict__ todo, double* __restrict__ Omd, double* __restrict__ Psd,
    double* __restrict__ Segd, int* __restrict__ grow, int pitch, int batch, int* __restrict__ qflag) {
  // Omd == NULL: TRIAL run -- nothing is stored, only the conditioning verdict is formed (*grow, and per QP bit 1 of qflag)
  const bool store = Omd != nullptr;
  const int col = blockIdx.x * PI_THREADS + threadIdx.x;
  if (col >= pitch) return;
  if (todo && !todo[col]) return;
  const size_t P_ = (size_t)pitch;
  const int sg = blockIdx.y, ka = seg_start[sg], kb = seg_start[sg + 1];
  double Lam[NX][NX], Xi[NX][NX];
#pragma unroll
  for (int i = 0; i < NX; ++i)
#pragma unroll
    for (int l = 0; l < NX; ++l) { Lam[i][l] = (i == l) ? 1.0 : 0.0; Xi[i][l] = 0.0; }
  for (int k = kb - 1; k >= ka; --k) {
    double A[NX][NX], B[NX][NU], K[NU][NX], Si[NU][NU];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
#pragma unroll
      for (int l = 0; l < NX; ++l) A[i][l] = Ad[((size_t)k * NX * NX + l * NX + i) * P_ + col];
#pragma unroll
      for (int j = 0; j < NU; ++j) B[i][j] = Bd[((size_t)k * NX * NU + j * NX + i) * P_ + col];
    }
#pragma unroll
    for (int j = 0; j < NU; ++j) {
#pragma unroll
      for (int l = 0; l < NX; ++l) K[j][l] = Kd[((size_t)k * NU * NX + j * NX + l) * P_ + col];
#pragma unroll
      for (int t = 0; t < NU; ++t) Si[j][t] = Sd[((size_t)k * NU * NU + j * NU + t) * P_ + col];
    }
    double Om[NX][NU], Ps[NU][NX];
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
      for (int j = 0; j < NU; ++j) {
        double a = 0.0;
#pragma unroll
        for (int l = 0; l < NX; ++l) a = fma(Lam[i][l], B[l][j], a);
        Om[i][j] = -a;
      }
#pragma unroll
    for (int j = 0; j < NU; ++j)
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        double a = 0.0;
#pragma unroll
        for (int t = 0; t < NU; ++t) a = fma(Si[j][t], Om[i][t], a);
        Ps[j][i] = -a;
      }
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
      for (int l = 0; l < NX; ++l) {
        double a = Xi[i][l];
#pragma unroll
        for (int j = 0; j < NU; ++j) a = fma(Om[i][j], Ps[j][l], a);
        Xi[i][l] = a;
      }
    if (store) {
#pragma unroll
      for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int j = 0; j < NU; ++j) {
          Omd[((size_t)k * NX * NU + i * NU + j) * P_ + col] = Om[i][j];
          Psd[((size_t)k * NU * NX + j * NX + i) * P_ + col] = Ps[j][i];
        }
    }
    // Lam <- Lam (A - B K)
    double Acl[NX][NX];
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
      for (int l = 0; l < NX; ++l) {
        double a = A[i][l];
#pragma unroll
        for (int j = 0; j < NU; ++j) a = fma(-B[i][j], K[j][l], a);
        Acl[i][l] = a;
      }
    double Ln[NX][NX];
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
      for (int l = 0; l < NX; ++l) {
        double a = 0.0;
#pragma unroll
        for (int r = 0; r < NX; ++r) a = fma(Lam[i][r], Acl[r][l], a);
        Ln[i][l] = a;
      }
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
      for (int l = 0; l < NX; ++l) Lam[i][l] = Ln[i][l];
  }
  double big = 0.0;
  double* sd = Segd + (size_t)sg * 3 * NX * NX * P_;
#pragma unroll
  for (int i = 0; i < NX; ++i)
#pragma unroll
    for (int l = 0; l < NX; ++l) {
      if (store) {
        sd[((size_t)0 * NX * NX + i * NX + l) * P_ + col] = Lam[l][i];     // Phi = Lam'
        sd[((size_t)1 * NX * NX + i * NX + l) * P_ + col] = Xi[i][l];
        sd[((size_t)2 * NX * NX + i * NX + l) * P_ + col] = Lam[i][l];     // Th
      }
      big = fmax(big, fmax(fabs(Lam[i][l]), fabs(Xi[i][l])));
    }
  if (col < batch && !(big <= 100.0)) {
    atomicOr(grow, 1);
    if (qflag) atomicOr(&qflag[col], 2);
  }
}

// Segment scan, per QP (both chains are S sequential n x n mat-vecs with the QP's own matrices):
//     t_in(S-1) = 0;  t_in(s-1) = tseg(s) + Phi_s t_in(s)          x_in(0) = x0;  x_in(s+1) = c(s) + Th_s x_in(s),  c(s) = eseg(s) + Xi_s t_in(s)
// One WAVE serves QPW QPs x n rows: lane (i, c) forms row i of every mat-vec of QP c, so it loads n doubles per step (not
// n^2; the operands of the next PSCAN_D steps wait in registers) and the vector never leaves the wave: element l of QP c's
// vector is the running value of lane (l, c), fetched with a cross-lane read -- no LDS, no barrier.  (History: one lane per QP
// doing whole mat-vecs, 2.5 us per step; one wave per row with the vector in LDS and a barrier per step, 0.7 us per step
// whatever was taken out of it -- the scan of 32 segments cost as much as a whole sweep.)  c(s) does not depend on the x chain:
// it is formed in the t chain, at the step whose input is t_in(s), and overwrites eseg(s).
constexpr int PSCAN_D = 8;        // steps whose operands are prefetched as one group ...
#ifndef ADMM_PSCAN_WIDE_D
#define ADMM_PSCAN_WIDE_D 2
#endif
template <int NX> struct PscanDepth { static constexpr int D = NX > 8 ? ADMM_PSCAN_WIDE_D : PSCAN_D; };   // ... (wide shapes: 2 n doubles per lane and step, a group of 8 steps is more loads than a wave keeps in flight -- measured at n = 12, 64 QPs, 64 segments: 118 us with groups of 8, 102 with 4, 93 with 2)
template <int NX> struct PscanShape { static constexpr int QPW = NX <= 2 ? 32 : (NX <= 4 ? 16 : (NX <= 8 ? 8 : 4)); };

// ---- operand layout of the rows-over-lanes kernels ----
// Batch-minor (every other array of the library): element e of stage k of QP col at [(k * E + e) * pitch + col].  A wave of the
// rows-over-lanes kernels serves QPW QPs, so each of its accesses takes 8 * QPW bytes out of every 128-byte line, and the line is
// complete only if the waves of the neighbouring QPs read it while it is cached: measured at (12, 6), 4096 x 1000, 1.7-1.8 x the
// algorithmic HBM bytes.  TILED (the operand arrays A, B, K, S^-1, Omega, Psi of the wide shapes; nothing else reads them):
//     [stage k][QP group g = col / QPW][element e][c = col % QPW]
// -- the E x QPW doubles a wave needs of a stage are ONE contiguous chunk (4.6 KB for A at n = 12), whole lines, its own.
template <int NX, bool TILED>
struct Operand {
  static constexpr int QPW = PscanShape<NX>::QPW;
  const double* p;       // element 0 of stage 0, this lane's QP
  size_t kstride;        // doubles per stage: E * pitch in both layouts
  size_t pitch_;
  __device__ __forceinline__ Operand(const double* arr, int E, size_t P_, int col) {
    p = TILED ? arr + (size_t)(col / QPW) * E * QPW + col % QPW : arr + col;
    kstride = P_ * E;
    pitch_ = P_;
  }
  __device__ __forceinline__ const double* stage(int k) const { return p + (size_t)k * kstride; }
  __device__ __forceinline__ size_t el(int e) const { return TILED ? (size_t)e * QPW : (size_t)e * pitch_; }
};

// ---- operand tiles staged through LDS (the sweeps of the wide shapes) ----
// A wave keeps at most 63 vector loads in flight, and a rows-over-lanes load carries 8 * QPW bytes for each of <= n lane rows (384 B
// at (12, 6), 192 B for the control-row operands): one stage per wave in flight, ~16 MB over the chip -- measured 0.53 of the HBM
// roofline at 4096 x 1000 whatever the prefetch depth.  Staged: all 64 lanes copy the (contiguous) tile of stage k + 2 with 16-byte
// loads -- 1 KB per instruction, every byte used -- into registers, the registers go to LDS one stage later, and the lane rows read
// THEIR operands from LDS when the stage is computed.  E = elements of the operand, M = length of its minor run (the index the lanes
// of a read differ in jumps by M elements: one element of padding per run spreads those reads over the LDS banks).
typedef double pair_t __attribute__((ext_vector_type(2)));      // 16 bytes: one global_load_dwordx4 / ds_write_b128

template <int E, int M, int QPW>
struct StageTile {
  static_assert(E % M == 0 && QPW % 2 == 0, "whole runs; a 16-byte pair stays inside one element row");
  static constexpr int MP = M + 1;
  static constexpr int LWORDS = (E / M) * MP * QPW;        // doubles of the padded LDS copy
  static constexpr int GPAIRS = E * QPW / 2;               // 16-byte pairs of the tile in memory
  static constexpr int NLD = (GPAIRS + 63) / 64;           // pairs per lane
  __device__ static __forceinline__ int pair_of(int lane, int n) {      // (the last, partial round re-reads the tile's last pair)
    const int q = lane + 64 * n;
    return (GPAIRS % 64 == 0 || n + 1 < NLD) ? q : (q < GPAIRS ? q : GPAIRS - 1);
  }
  __device__ static __forceinline__ void gload(pair_t (&r)[NLD], const double* tile, int lane) {
    const pair_t* t2 = reinterpret_cast<const pair_t*>(tile);
#pragma unroll
    for (int n = 0; n < NLD; ++n) r[n] = t2[pair_of(lane, n)];
  }
  __device__ static __forceinline__ void lstore(const pair_t (&r)[NLD], double* lds, int lane) {
#pragma unroll
    for (int n = 0; n < NLD; ++n) {
      const int w = 2 * pair_of(lane, n), e = w / QPW, c = w % QPW;
      *reinterpret_cast<pair_t*>(&lds[((e / M) * MP + e % M) * QPW + c]) = r[n];
    }
  }
  // element e (= run * M + minor) of QP c
  __device__ static __forceinline__ int at(int run, int minor, int c) { return (run * MP + minor) * QPW + c; }
};

// ---- lane geometry shared by the rows-over-lanes kernels (admm_pinst_rows.hpp, admm_pinst_wide.hpp) ----
constexpr int PROWS_BLOCK = 256;      // up to 4 waves per workgroup: the 16 / QPW waves that share a 128-byte line of every array

// lane geometry of the rows-over-lanes kernels: wave w of block b serves the QPW QPs from (b * waves + w) * QPW
template <int NX>
struct RowsLane {
  static constexpr int QPW = PscanShape<NX>::QPW;
  int c, ir, col;
  __device__ __forceinline__ RowsLane() {
    const int lane = threadIdx.x % PI_THREADS, wave = threadIdx.x / PI_THREADS, wpb = blockDim.x / PI_THREADS;
    c = lane % QPW;
    ir = lane / QPW;
    col = (blockIdx.x * wpb + wave) * QPW + c;
  }
  // the value `v` holds on the lane that owns row l of this lane's QP
  __device__ __forceinline__ double across(double v, int l) const { return __shfl(v, l * QPW + c, PI_THREADS); }
};


template <int NX>
__global__ __launch_bounds__(PI_THREADS) void pscan_kernel(
    const double* __restrict__ Segd, const double* __restrict__ tseg, double* __restrict__ eseg,
    const double* __restrict__ x0, double* __restrict__ tin, double* __restrict__ xin, int S, int pitch) {
  // (eseg and tin are each read and written here, through these pointers only; no two of the arrays overlap)
  constexpr int D = PscanDepth<NX>::D, QPW = PscanShape<NX>::QPW;
  static_assert(QPW * NX <= PI_THREADS, "rows x QPs of a wave");
  const int lane = threadIdx.x;
  const int c = lane % QPW, ir = lane / QPW;
  const bool live = ir < NX;                          // (lanes beyond n x QPW shadow row 0: loads and arithmetic only)
  const int i = live ? ir : 0;
  const int col = blockIdx.x * QPW + c;               // pitch is a multiple of 64, hence of QPW
  const size_t P_ = (size_t)pitch;
  const double* rowp = Segd + (size_t)i * NX * P_ + col;     // row i of block (sg, which): + ((sg * 3 + which) * NX * NX + l) * P_
  auto element = [&](double mine, int l) { return __shfl(mine, l * QPW + c, PI_THREADS); };   // element l of this QP's vector
  // Both chains run as groups of steps with straight-line code inside a group: a step beyond the chain's end keeps the
  // lane's value and stores nothing (its operand loads are clamped to valid addresses).
  double mine = 0.0;
  // ---- t chain: step u = 0 .. S-1 handles segment sg = S-1-u: the wave holds t_in(sg); it produces t_in(sg-1) (sg >= 1)
  //      and c(sg) = eseg(sg) + Xi_sg t_in(sg) from the same cross-lane reads ----
  if (live) tin[((size_t)(S - 1) * NX + i) * P_ + col] = 0.0;
  {
    constexpr int DT = D / 2;                         // two rows of operands per step: half the steps per group
    struct OpsT { double M[DT][NX], X[DT][NX], ts[DT], es[DT]; };
    const int nst = S, ngt = (nst + DT - 1) / DT;
    auto load = [&](OpsT& o, int u0) {
#pragma unroll
      for (int u = 0; u < DT; ++u) {
        const int sg = (u0 + u < nst) ? S - 1 - (u0 + u) : 0;
#pragma unroll
        for (int l = 0; l < NX; ++l) {
          o.M[u][l] = rowp[(((size_t)sg * 3 + 0) * NX * NX + l) * P_];
          o.X[u][l] = rowp[(((size_t)sg * 3 + 1) * NX * NX + l) * P_];
        }
        o.ts[u] = tseg[((size_t)sg * NX + i) * P_ + col];
        o.es[u] = eseg[((size_t)sg * NX + i) * P_ + col];
      }
    };
    auto steps = [&](const OpsT& o, int u0) {
#pragma unroll
      for (int u = 0; u < DT; ++u) {
        const bool valid = u0 + u < nst;
        const int sg = S - 1 - (u0 + u);
        double a = o.ts[u], cc = o.es[u];
#pragma unroll
        for (int l = 0; l < NX; ++l) {
          const double v = element(mine, l);
          a = fma(o.M[u][l], v, a);
          cc = fma(o.X[u][l], v, cc);
        }
        const bool chain = valid && sg >= 1;           // (sg = 0: only c(0) is left to form)
        mine = chain ? a : mine;
        if (chain && live) tin[((size_t)(sg - 1) * NX + i) * P_ + col] = a;
        if (valid && live) eseg[((size_t)sg * NX + i) * P_ + col] = cc;
      }
    };
    OpsT A, B;
    load(A, 0);
    for (int g = 0; g < ngt; g += 2) {
      if (g + 1 < ngt) load(B, (g + 1) * DT);
      steps(A, g * DT);
      if (g + 1 >= ngt) break;
      if (g + 2 < ngt) load(A, (g + 2) * DT);
      steps(B, (g + 1) * DT);
    }
  }
  // ---- x chain: step sg = 0 .. S-2 produces x_in(sg+1) (c(sg): this lane's own store above) ----
  {
    struct Ops { double M[D][NX], c[D]; };
    const int nsteps = S - 1, ngroups = (nsteps + D - 1) / D;
    mine = x0[(size_t)i * P_ + col];
    if (live) xin[(size_t)i * P_ + col] = mine;
    auto load = [&](Ops& o, int u0) {
#pragma unroll
      for (int u = 0; u < D; ++u) {
        const int sg = (u0 + u < nsteps) ? u0 + u : 0;
#pragma unroll
        for (int l = 0; l < NX; ++l) o.M[u][l] = rowp[(((size_t)sg * 3 + 2) * NX * NX + l) * P_];
        o.c[u] = eseg[((size_t)sg * NX + i) * P_ + col];
      }
    };
    auto steps = [&](const Ops& o, int u0) {
#pragma unroll
      for (int u = 0; u < D; ++u) {
        const bool valid = u0 + u < nsteps;
        const int sg = u0 + u;
        double a = o.c[u];
#pragma unroll
        for (int l = 0; l < NX; ++l) a = fma(o.M[u][l], element(mine, l), a);
        mine = valid ? a : mine;
        if (valid && live) xin[((size_t)(sg + 1) * NX + i) * P_ + col] = a;
      }
    };
    Ops A, B;
    load(A, 0);
    for (int g = 0; g < ngroups; g += 2) {
      if (g + 1 < ngroups) load(B, (g + 1) * D);
      steps(A, g * D);
      if (g + 1 >= ngroups) break;
      if (g + 2 < ngroups) load(A, (g + 2) * D);
      steps(B, (g + 1) * D);
    }
  }
}

// QP-major staged array src[b][rows] (rows = N * E) -> TILED operand array dst[k][g][e][c] (Operand<., true>); columns beyond the
// batch are zero.  Thread t of block (k, g-chunk) moves element (e, c) = (t / QPW, t % QPW): contiguous stores, QPW interleaved
// contiguous loads.
// nr > 0: the source blocks are ROW-major nr x nc (ADMM_FLAG_ROW_MAJOR): element e = j * nr + i of the (column-major) tile is read from
// i * nc + j.
static __global__ __launch_bounds__(256) void to_tiled_kernel(const double* __restrict__ src, double* __restrict__ dst, int batch,
                                                               int N, int E, int qpw, int pitch, int nr, int nc) {
  const int G = pitch / qpw;
  const size_t tile = (size_t)E * qpw;
  for (size_t kg = blockIdx.x; kg < (size_t)N * G; kg += gridDim.x) {
    const int k = (int)(kg / G), g = (int)(kg % G);
    for (int t = threadIdx.x; t < (int)tile; t += 256) {
      const int e = t / qpw, c = t % qpw, b = g * qpw + c;
      const int es = nr > 0 ? (e % nr) * nc + e / nr : e;
      dst[kg * tile + t] = b < batch ? src[((size_t)b * N + k) * E + es] : 0.0;
    }
  }
}

// v -> (z, y) with per-instance bounds (read-out, mode switches)
// ---- per-QP adaptive rho: residual balancing QP by QP, entirely on the device ----
// With per-instance dynamics nothing is shared, so the rule of DESIGN.md §2.6 applies to each QP alone -- exactly what
// the one-QP oracle does: at a checked iteration that is a multiple of adapt_interval, a QP that has not converged and
// has changed rho fewer than adapt_max times compares R = r^2 with S = s^2:  R > mu^2 S -> rho tau;  S > mu^2 R -> rho / tau.
// The kernel records the QPs that change (todo), the factor rho_old / rho_new their scaled dual is multiplied by, and
// how many changed; padapt_scale_kernel rescales y, the masked pfactor_kernel refactors those QPs.
static __global__ __launch_bounds__(256) void padapt_kernel(
    const double* __restrict__ resid, const int* __restrict__ status, double* __restrict__ rhov, int* __restrict__ nupd,
    int* __restrict__ todo, double* __restrict__ cscale, int* __restrict__ nchanged, double mu2, double tau, int adapt_max,
    int pitch, int batch, double* __restrict__ rho_prev) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= pitch) return;
  int change = 0;
  double c = 1.0;
  rho_prev[col] = rhov[col];                 // (padapt_veto_kernel puts it back if the new rho is refused)
  if (col < batch && !status[col] && nupd[col] < adapt_max) {
    const double r = resid[col], s_ = resid[(size_t)pitch + col];
    const double R = r * r, S = s_ * s_;
    const double rho = rhov[col];
    double rho_new = rho;
    if (R > mu2 * S) rho_new = rho * tau;
    else if (S > mu2 * R) rho_new = rho / tau;
    if (rho_new != rho) {
      change = 1;
      c = rho / rho_new;
      rhov[col] = rho_new;
      nupd[col] += 1;
    }
  }
  todo[col] = change;
  cscale[col] = c;
  if (change) atomicAdd(nchanged, 1);
}

// The rule's change of a QP is REFUSED when the trial factorisation with its new rho fails or its segment transfer matrices
// break the conditioning bound (qflag != 0; admm_api.hip, admm_solve_adapt): that QP keeps its rho and factor and stops
// adapting -- what the batch-level rule does for shared dynamics.  Runs before anything of the change
// (dual rescale, refactor) has been applied.
static __global__ __launch_bounds__(256) void padapt_veto_kernel(const int* __restrict__ qflag, const double* __restrict__ rho_prev,
                                                                 double* __restrict__ rhov, int* __restrict__ nupd, int* __restrict__ todo,
                                                                 double* __restrict__ cscale, int* __restrict__ nchanged,
                                                                 int* __restrict__ nveto, int adapt_max, int pitch) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= pitch || !todo[col] || !qflag[col]) return;
  rhov[col] = rho_prev[col];
  nupd[col] = adapt_max;
  todo[col] = 0;
  cscale[col] = 1.0;
  atomicSub(nchanged, 1);
  atomicAdd(nveto, 1);
}

// y[row][col] *= cscale[col] for the columns marked in todo (cscale = 1 elsewhere: those columns are not touched, so
// their y keeps its bits)
static __global__ __launch_bounds__(256) void padapt_scale_kernel(double* __restrict__ y, const double* __restrict__ cscale,
                                                                  const int* __restrict__ todo, int rows, int pitch) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= pitch || !todo[col]) return;
  const double c = cscale[col];
  for (int r = blockIdx.y; r < rows; r += gridDim.y) y[(size_t)r * pitch + col] *= c;
}

// ... with the thrust-magnitude bound: one thread per (stage, QP) walks its block (the norm needs the m control rows together)
static __global__ __launch_bounds__(256) void pv_to_zy_soc_kernel(const double* __restrict__ v, double* __restrict__ z,
                                                                  double* __restrict__ y, const double* __restrict__ lo,
                                                                  const double* __restrict__ hi, const double* __restrict__ ub,
                                                                  int N, int nb, int m, int pitch) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= pitch) return;
  for (int k = blockIdx.y; k < N; k += gridDim.y) {
    const size_t base = (size_t)k * nb * pitch + col;
    double ss = 0.0;
    for (int j = 0; j < m; ++j) { const double vu = v[base + (size_t)j * pitch]; ss = fma(vu, vu, ss); }
    const double nrm = sqrt(ss), ubk = ub[k];
    const double cs = nrm > ubk ? ubk / nrm : 1.0;
    for (int r = 0; r < nb; ++r) {
      const size_t o = base + (size_t)r * pitch;
      const double vv = v[o];
      const double zz = fmin(fmax(r < m ? vv * cs : vv, lo[o]), hi[o]);
      z[o] = zz;
      y[o] = vv - zz;
    }
  }
}

static __global__ __launch_bounds__(256) void pv_to_zy_kernel(const double* __restrict__ v, double* __restrict__ z,
                                                              double* __restrict__ y, const double* __restrict__ lo,
                                                              const double* __restrict__ hi, size_t count) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
    const double vv = v[i];
    const double zz = fmin(fmax(vv, lo[i]), hi[i]);
    z[i] = zz;
    y[i] = vv - zz;
  }
}

}  // namespace admm

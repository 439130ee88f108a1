// admm_pinst_wide.hpp -- per-instance dynamics, wide shapes ((8, 4), (12, 6), ...): the device factorisation and the segment
// transfer matrices with the ROWS of a QP's matrices spread over the lanes of a wave.  No reference
// counterpart exists (README.md:1-2 only).
//
// pfactor_kernel / pseg_kernel (admm_pinst.hpp) keep P, A, PA, ... of one QP in the registers of ONE lane: 2 n^2 + ... doubles,
// > 512 registers from n = 8.  Here lane (i, c) of a wave owns ROW i of every matrix of QP c (the lane layout of
// pscan_kernel and of the rows-over-lanes sweeps): a matrix product C = A B becomes, on lane i,
//     C_i[l] = sum_r A_i[r] * (B_r[l] as held by lane r)
// -- the second factor is read across the lanes, one register at a time; a lane holds O(n) doubles per matrix.  Where a product
// contracts over the ROW index of its first factor (B'F, A'G, K'(SK)) the lane loads / forms the COLUMN it needs instead: A and B
// are read from memory both ways, and K (born distributed by rows) is transposed across the lanes by `column_of`.
// Every sum is accumulated in the order of the one-lane kernels (same recursion, same symmetrisations): at (6, 3), where both
// exist, the iterates through either factorisation agree to a few units in the last place after tens of iterations
// (tests/test_gpu_pinst.py: <= 1e-15 absolute; the two compilations differ by an ulp in rare entries).
// Needs m <= n and n * QPW <= 64.
#pragma once

#include "admm_pinst.hpp"

namespace admm {

// Scheduling fences between the phases of a stage / the steps of a phase: OFF.  They were put in while the kernel ran out of
// registers; the causes turned out to be elsewhere (see pfactor_rows_kernel: the branch at the end of the stage, the transpositions
// through LDS), and with those fixed the fences only serialise the cross-lane reads (a refactor of 4096 x 1000 QPs at (12, 6): 21.8 ms
// with them, 16.7 ms without).  -DADMM_WIDE_FENCES brings them back for experiments.
#ifdef ADMM_WIDE_FENCES
#define ADMM_PHASE() __builtin_amdgcn_sched_barrier(0)
#define ADMM_STEP() __builtin_amdgcn_sched_barrier(0)
#else
#define ADMM_PHASE()
#define ADMM_STEP()
#endif

// This lane's column `mine` of a matrix distributed by rows: out[r] = (reg[mine] as held by row-lane r), r < ROWS -- a transposition
// across the lanes.  The register index differs from lane to lane, so a cross-lane read cannot fetch it (it names ONE register for
// the whole wave; fetching every column and keeping one costs ROWS x COLS reads and as many selects, which the compiler moves to
// the end of the stage with all the fetched values waiting in registers): the rows go through LDS instead.  `board` is this QP's
// patch of the wave's LDS, [64 / QPW rows][LD]; every lane writes the row named by its lane row (the shadow lanes their own, never
// read), then reads its column.  A wave's LDS operations execute in order: no barrier, only compiler fences.
template <int ROWS, int COLS, int LD>
__device__ __forceinline__ void column_of(double (&out)[ROWS], const double (&reg)[COLS], int mine, double* board, int lane_row) {
  static_assert(COLS <= LD, "board row");
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int cc = 0; cc < COLS; ++cc) board[lane_row * LD + cc] = reg[cc];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int r = 0; r < ROWS; ++r) out[r] = board[r * LD + mine];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------
// Riccati factorisation (the recursion and the summation order of pfactor_kernel), rows over lanes.
// ---------------------------------------------------------------------------
template <int NX, int NU, bool TILED>
__global__ __launch_bounds__(PROWS_BLOCK) void pfactor_rows_kernel(
    const double* __restrict__ Ad, const double* __restrict__ Bd, const double* __restrict__ Qm,
    const double* __restrict__ Rm, const double* __restrict__ QNm, const double* __restrict__ rhov,
    const int* __restrict__ todo, double* __restrict__ Kd, double* __restrict__ Sd, int* __restrict__ fail, int N, int pitch,
    int batch, int* __restrict__ qflag) {
  static_assert(NU <= NX && PscanShape<NX>::QPW * NX <= PI_THREADS, "rows over lanes: m <= n, n x QPW lanes");
  const RowsLane<NX> ln;
  const int col = ln.col;
  if (col >= pitch) return;                     // (a whole wave: pitch is a multiple of QPW)
  if (todo && !todo[col]) return;               // (a whole QP: its rows share col; the cross-lane reads stay inside a QP)
  const bool live_x = ln.ir < NX, live_u = ln.ir < NU;
  const int i = live_x ? ln.ir : 0, j = live_u ? ln.ir : 0;      // (other lanes shadow row 0: loads and arithmetic only)
  const size_t P_ = (size_t)pitch;
  const double rho = rhov[col];
  auto X = [&](double v, int l) { return ln.across(v, l); };
  constexpr int QPW = PscanShape<NX>::QPW, LROWS = PI_THREADS / QPW, LD = NX;
  __shared__ double boards[PROWS_BLOCK / PI_THREADS][QPW][LROWS * LD];
  double* board = boards[threadIdx.x / PI_THREADS][ln.c];
  // Stage operands: row i / column i of A_k, row i / column j of B_k.  ONE set of registers: each group is re-loaded with the
  // next stage's values right after its last use in this stage (a rotating prefetch -- two whole sets would not fit at (12, 6)).
  double Ai[NX], Ati[NX], Bi[NU], Btj[NX];
  const Operand<NX, TILED> opA(Ad, NX * NX, P_, col), opB(Bd, NX * NU, P_, col), opK(Kd, NU * NX, P_, col), opS(Sd, NU * NU, P_, col);
  auto load_Ai = [&](int k) {
    const double* Ak = opA.stage(k);
#pragma unroll
    for (int l = 0; l < NX; ++l) Ai[l] = Ak[opA.el(l * NX + i)];            // A[i][l]
  };
  auto load_Bi = [&](int k) {
    const double* Bk = opB.stage(k);
#pragma unroll
    for (int t = 0; t < NU; ++t) Bi[t] = Bk[opB.el(t * NX + i)];            // B[i][t]
  };
  auto load_cols = [&](int k) {
    const double *Ak = opA.stage(k), *Bk = opB.stage(k);
#pragma unroll
    for (int l = 0; l < NX; ++l) {
      Ati[l] = Ak[opA.el(i * NX + l)];                                      // A[l][i]
      Btj[l] = Bk[opB.el(j * NX + l)];                                      // B[l][j]
    }
  };
  double Pi[NX], Qi[NX], Rj[NU];
#pragma unroll
  for (int l = 0; l < NX; ++l) {
    Pi[l] = QNm[i * NX + l] + (i == l ? rho : 0.0);
    Qi[l] = Qm[i * NX + l] + (i == l ? rho : 0.0);
  }
#pragma unroll
  for (int t = 0; t < NU; ++t) Rj[t] = Rm[j * NU + t] + (j == t ? rho : 0.0);
  bool bad = false;
  load_Bi(N - 1);
  load_cols(N - 1);
  load_Ai(N - 1);
  for (int k = N - 1; k >= 0; --k) {
    const int kn = k > 0 ? k - 1 : 0;            // (the last stage re-reads itself)
    // F = P B  (row i)
    double F[NU];
#pragma unroll
    for (int t = 0; t < NU; ++t) F[t] = 0.0;
#pragma unroll
    for (int l = 0; l < NX; ++l) {
#pragma unroll
      for (int t = 0; t < NU; ++t) F[t] = fma(Pi[l], X(Bi[t], l), F[t]);
      ADMM_STEP();
    }
    load_Bi(kn);
    ADMM_PHASE();
    // S = R + rho I + B'F  (row j), symmetrised
    double S[NU];
#pragma unroll
    for (int t = 0; t < NU; ++t) S[t] = Rj[t];
#pragma unroll
    for (int r = 0; r < NX; ++r) {
#pragma unroll
      for (int t = 0; t < NU; ++t) S[t] = fma(Btj[r], X(F[t], r), S[t]);
      ADMM_STEP();
    }
    {
      double St[NU];
      column_of<NU, NU, LD>(St, S, j, board, ln.ir);
#pragma unroll
      for (int t = 0; t < NU; ++t) S[t] = (t == j) ? S[t] : 0.5 * (S[t] + St[t]);
    }
    ADMM_PHASE();
    // Si = S^-1: Gauss-Jordan on rows (no pivoting; a non-positive pivot flags failure), symmetrised
    double W[NU], Si[NU];
#pragma unroll
    for (int t = 0; t < NU; ++t) { W[t] = S[t]; Si[t] = (t == j) ? 1.0 : 0.0; }
#pragma unroll
    for (int cc = 0; cc < NU; ++cc) {
      const double pv = X(W[cc], cc);
      bad = bad | !(pv > 0.0) | !(pv < INFINITY);
      const double ip = 1.0 / pv;
      const double f = W[cc];
#pragma unroll
      for (int t = 0; t < NU; ++t) {
        const double pw = X(W[t], cc) * ip, ps = X(Si[t], cc) * ip;      // row cc, scaled
        W[t] = (j == cc) ? pw : fma(-f, pw, W[t]);
        Si[t] = (j == cc) ? ps : fma(-f, ps, Si[t]);
      }
      ADMM_STEP();
    }
    {
      double Sit[NU];
      column_of<NU, NU, LD>(Sit, Si, j, board, ln.ir);
#pragma unroll
      for (int t = 0; t < NU; ++t) Si[t] = (t == j) ? Si[t] : 0.5 * (Si[t] + Sit[t]);
    }
    ADMM_PHASE();
    // G = P A  (row i)
    double G[NX];
#pragma unroll
    for (int l = 0; l < NX; ++l) G[l] = 0.0;
#pragma unroll
    for (int r = 0; r < NX; ++r) {
#pragma unroll
      for (int l = 0; l < NX; ++l) G[l] = fma(Pi[r], X(Ai[l], r), G[l]);
      ADMM_STEP();
    }
    load_Ai(kn);
    ADMM_PHASE();
    // H = B'G  (row j)   and   P+ = Q + rho I + A'G  (row i): both read the rows of G
    double H[NX], Pn[NX];
#pragma unroll
    for (int l = 0; l < NX; ++l) { H[l] = 0.0; Pn[l] = Qi[l]; }
#pragma unroll
    for (int r = 0; r < NX; ++r) {
#pragma unroll
      for (int l = 0; l < NX; ++l) {
        const double g = X(G[l], r);
        H[l] = fma(Btj[r], g, H[l]);
        Pn[l] = fma(Ati[r], g, Pn[l]);
      }
      ADMM_STEP();
    }
    load_cols(kn);
    ADMM_PHASE();
    // K = Si H  (row j)
    double K[NX];
#pragma unroll
    for (int l = 0; l < NX; ++l) K[l] = 0.0;
#pragma unroll
    for (int t = 0; t < NU; ++t) {
#pragma unroll
      for (int l = 0; l < NX; ++l) K[l] = fma(Si[t], X(H[l], t), K[l]);
      ADMM_STEP();
    }
    ADMM_PHASE();
    // SK = S K  (row j);   P+ -= K'(SK): lane i needs column i of K
    double SK[NX];
#pragma unroll
    for (int l = 0; l < NX; ++l) SK[l] = 0.0;
#pragma unroll
    for (int t = 0; t < NU; ++t) {
#pragma unroll
      for (int l = 0; l < NX; ++l) SK[l] = fma(S[t], X(K[l], t), SK[l]);
      ADMM_STEP();
    }
    double Kt[NU];
    column_of<NU, NX, LD>(Kt, K, i, board, ln.ir);
#pragma unroll
    for (int t = 0; t < NU; ++t) {
#pragma unroll
      for (int l = 0; l < NX; ++l) Pn[l] = fma(-Kt[t], X(SK[l], t), Pn[l]);
      ADMM_STEP();
    }
    {
      double Pt[NX];
      column_of<NX, NX, LD>(Pt, Pn, i, board, ln.ir);
#pragma unroll
      for (int l = 0; l < NX; ++l) Pi[l] = (l == i) ? Pn[l] : 0.5 * (Pn[l] + Pt[l]);
    }
    ADMM_PHASE();
    // (the only branch of the stage, kept LAST: the compiler sinks arithmetic below a branch it is not needed before, and the
    //  cross-lane reads feeding it would wait in registers -- 288 of them when this block sat after K)
    if (live_u) {
      double* Kk = const_cast<double*>(opK.stage(k));
      double* Sk = const_cast<double*>(opS.stage(k));
#pragma unroll
      for (int l = 0; l < NX; ++l) Kk[opK.el(j * NX + l)] = K[l];           // K[j][l]
#pragma unroll
      for (int t = 0; t < NU; ++t) Sk[opS.el(j * NU + t)] = Si[t];          // Si[j][t]
    }
    ADMM_PHASE();
  }
  if (bad && col < batch && ln.ir == 0) {
    atomicOr(fail, 1);
    if (qflag) atomicOr(&qflag[col], 1);
  }
}

// ---------------------------------------------------------------------------
// Segment transfer matrices (the recursion and the summation order of pseg_kernel), rows over lanes; blockIdx.y = segment.
// ---------------------------------------------------------------------------
template <int NX, int NU, bool TILED>
__global__ __launch_bounds__(PROWS_BLOCK) void pseg_rows_kernel(
    const double* __restrict__ Ad, const double* __restrict__ Bd, const double* __restrict__ Kd, const double* __restrict__ Sd,
    const int* __restrict__ seg_start, const int* __restrict__ todo, double* __restrict__ Omd, double* __restrict__ Psd,
    double* __restrict__ Segd, int* __restrict__ grow, int pitch, int batch, int* __restrict__ qflag) {
  static_assert(NU <= NX && PscanShape<NX>::QPW * NX <= PI_THREADS, "rows over lanes: m <= n, n x QPW lanes");
  const bool store = Omd != nullptr;             // NULL: trial run -- only the conditioning verdict is formed
  const RowsLane<NX> ln;
  const int col = ln.col;
  if (col >= pitch) return;
  if (todo && !todo[col]) return;
  const bool live_x = ln.ir < NX;
  const int i = live_x ? ln.ir : 0, j = ln.ir < NU ? ln.ir : 0;
  const size_t P_ = (size_t)pitch;
  const int sg = blockIdx.y, ka = seg_start[sg], kb = seg_start[sg + 1];
  auto X = [&](double v, int l) { return ln.across(v, l); };
  struct Ops { double Ai[NX], Bi[NU], Kj[NX], Sij[NU]; };      // rows i of A_k, B_k; rows j of K_k, S_k^-1
  const Operand<NX, TILED> opA(Ad, NX * NX, P_, col), opB(Bd, NX * NU, P_, col), opK(Kd, NU * NX, P_, col), opS(Sd, NU * NU, P_, col),
      opO(store ? Omd : Ad, NX * NU, P_, col), opP(store ? Psd : Ad, NU * NX, P_, col);
  auto load = [&](Ops& o, int k) {
    const double *Ak = opA.stage(k), *Bk = opB.stage(k), *Kk = opK.stage(k), *Sk = opS.stage(k);
#pragma unroll
    for (int l = 0; l < NX; ++l) {
      o.Ai[l] = Ak[opA.el(l * NX + i)];
      o.Kj[l] = Kk[opK.el(j * NX + l)];
    }
#pragma unroll
    for (int t = 0; t < NU; ++t) {
      o.Bi[t] = Bk[opB.el(t * NX + i)];
      o.Sij[t] = Sk[opS.el(j * NU + t)];
    }
  };
  double Lam[NX], Xi[NX];
#pragma unroll
  for (int l = 0; l < NX; ++l) { Lam[l] = (i == l) ? 1.0 : 0.0; Xi[l] = 0.0; }
  auto stage = [&](const Ops& o, int k) {
    // Omega = -Lam B  (row i)
    double Om[NU];
#pragma unroll
    for (int t = 0; t < NU; ++t) Om[t] = 0.0;
#pragma unroll
    for (int l = 0; l < NX; ++l)
#pragma unroll
      for (int t = 0; t < NU; ++t) Om[t] = fma(Lam[l], X(o.Bi[t], l), Om[t]);
#pragma unroll
    for (int t = 0; t < NU; ++t) Om[t] = -Om[t];
    // Psi = -Si Omega'  (COLUMN i: Psi[t][i] = -sum_u Si[t][u] Omega[i][u], Si symmetric)
    double Ps[NU];
#pragma unroll
    for (int t = 0; t < NU; ++t) Ps[t] = 0.0;
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int t = 0; t < NU; ++t) Ps[t] = fma(X(o.Sij[t], u), Om[u], Ps[t]);
#pragma unroll
    for (int t = 0; t < NU; ++t) Ps[t] = -Ps[t];
    // Xi += Omega Psi  (row i)
#pragma unroll
    for (int t = 0; t < NU; ++t)
#pragma unroll
      for (int l = 0; l < NX; ++l) Xi[l] = fma(Om[t], X(Ps[t], l), Xi[l]);
    if (store && live_x) {
      double* Ok = const_cast<double*>(opO.stage(k));
      double* Pk = const_cast<double*>(opP.stage(k));
#pragma unroll
      for (int t = 0; t < NU; ++t) {
        Ok[opO.el(i * NU + t)] = Om[t];
        Pk[opP.el(t * NX + i)] = Ps[t];
      }
    }
    // Lam <- Lam (A - B K)
    double Acl[NX];
#pragma unroll
    for (int l = 0; l < NX; ++l) Acl[l] = o.Ai[l];
#pragma unroll
    for (int t = 0; t < NU; ++t)
#pragma unroll
      for (int l = 0; l < NX; ++l) Acl[l] = fma(-o.Bi[t], X(o.Kj[l], t), Acl[l]);
    double Ln[NX];
#pragma unroll
    for (int l = 0; l < NX; ++l) Ln[l] = 0.0;
#pragma unroll
    for (int r = 0; r < NX; ++r)
#pragma unroll
      for (int l = 0; l < NX; ++l) Ln[l] = fma(Lam[r], X(Acl[l], r), Ln[l]);
#pragma unroll
    for (int l = 0; l < NX; ++l) Lam[l] = Ln[l];
  };
  Ops oa, ob;
  load(oa, kb - 1);
  for (int k = kb - 1; k >= ka; k -= 2) {
    load(ob, k - 1 >= ka ? k - 1 : ka);
    stage(oa, k);
    if (k - 1 < ka) break;
    load(oa, k - 2 >= ka ? k - 2 : ka);
    stage(ob, k - 1);
  }
  bool over = false;                         // (a NaN entry breaks the bound too)
  double* sd = Segd + (size_t)sg * 3 * NX * NX * P_;
#pragma unroll
  for (int l = 0; l < NX; ++l) {
    if (store && live_x) {
      sd[((size_t)0 * NX * NX + l * NX + i) * P_ + col] = Lam[l];     // Phi = Lam':  Phi[l][i] = Lam[i][l]
      sd[((size_t)1 * NX * NX + i * NX + l) * P_ + col] = Xi[l];
      sd[((size_t)2 * NX * NX + i * NX + l) * P_ + col] = Lam[l];     // Th
    }
    over = over || !(fabs(Lam[l]) <= 100.0) || !(fabs(Xi[l]) <= 100.0);
  }
  double rows_over = 0.0;
#pragma unroll
  for (int l = 0; l < NX; ++l) rows_over += X(over ? 1.0 : 0.0, l);
  if (col < batch && ln.ir == 0 && rows_over > 0.0) {
    atomicOr(grow, 1);
    if (qflag) atomicOr(&qflag[col], 2);
  }
}

}  // namespace admm

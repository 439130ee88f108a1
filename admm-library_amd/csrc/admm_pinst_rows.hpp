// admm_pinst_rows.hpp -- per-instance dynamics, small batches: the sweeps of admm_pinst.hpp with the ROWS of a QP's stage
// operators spread over the lanes of a wave.  No reference counterpart exists (README.md:1-2 only).
//
// With one lane per QP (pxb_kernel / pxfz_kernel) a lane fetches ~100 operands per stage and can keep two stages in flight:
// a stage costs one memory round trip, ~1.6 us, however few QPs there are.  Here a wave serves QPW QPs x n rows, as in
// pscan_kernel: lane (i, c) owns row i of QP c -- state row x_i, row / column i of A_k, K_k, Omega_k, and, for i < m, control
// row u_i with column / row i of B_k, S_k^-1, K_k, Psi_k.  It fetches ~30 operands per stage, so several stages wait in
// registers, and the short vectors a mat-vec needs (p, h, d; x, u) are read across the lanes -- no LDS, no barrier.
// Every dot product is accumulated in the same order as in the one-lane kernels, so the iterates are the same bit for bit;
// the residual partial sums are formed per row and then added over a QP's rows in a fixed order (they differ from the
// one-lane kernels' by rounding).
// Needs m <= n and n * QPW <= 64 (true for every compiled shape).
//
// Wide shapes ((8, 4), (12, 6), ...; TILED): these kernels at ANY batch.  The matrix operands live in the tiled layout and are
// staged through LDS (admm_pinst.hpp: Operand, StageTile): all 64 lanes copy the contiguous tiles of the stage after next with
// 16-byte loads, the tiles go to LDS one stage later, the lane rows read their operands from LDS.  Measured at (12, 6), 4096 x 1000
// (one sweep): batch-minor operands 4.9 ms (1.7-1.8 x the algorithmic HBM bytes), tiled 2.9 ms (1.0 x), tiled + staged 2.6-2.7 ms,
// + the waves of a block kept in step (one barrier per stage: they share the lines of the batch-minor state / bound rows) 2.25 ms;
// without the state / bound loads (32-byte pieces of batch-minor rows) 1.9 ms, arithmetic + LDS alone 1.0 ms.
#pragma once

#include "admm_pinst.hpp"

namespace admm {

#ifndef ADMM_PROWS_D
#define ADMM_PROWS_D 2
#endif
// stages per prefetch group (two groups alternate: up to 2 * D - 1 stages in flight); one from n = 8: a stage is ~50 operands there
template <int NX> struct ProwsDepth { static constexpr int D = NX >= 8 ? 1 : ADMM_PROWS_D; };

// ---------------------------------------------------------------------------
// Backward sweep, rows over lanes (the arithmetic of pxb_kernel):
//     g = q - rho (z - y);  p = g^x + t;  h = B'p + g^u;  d_k = Si h -> dbuf;  t = A'p - K'h;   SEG: e += Omega_k d_k
// ---------------------------------------------------------------------------
template <int NX, int NU, bool HASQ, bool VFORM, bool PB, bool SEG, bool TILED = false, bool SOC = false>
__global__ __launch_bounds__(PROWS_BLOCK) void pxb_rows_kernel(
    const double* __restrict__ z, const double* __restrict__ y, const double* __restrict__ q,
    const double* __restrict__ Ad, const double* __restrict__ Bd, const double* __restrict__ Kd,
    const double* __restrict__ Sd, const double* __restrict__ lo, const double* __restrict__ hi,
    double* __restrict__ dbuf, const double* __restrict__ rhov, int N, int pitch,
    const double* __restrict__ Omd, const int* __restrict__ seg_start, double* __restrict__ tseg, double* __restrict__ eseg,
    const double* __restrict__ loT = nullptr, const double* __restrict__ hiT = nullptr, const double* __restrict__ ubd = nullptr) {
  constexpr int NB = NX + NU, D = ProwsDepth<NX>::D, QPW = PscanShape<NX>::QPW;
  static_assert(!SOC || VFORM, "the (z, y) form carries the projected z: only the v-form projects");
  static_assert(NU <= NX && QPW * NX <= PI_THREADS, "rows over lanes: m <= n, n x QPW lanes");
  const RowsLane<NX> ln;
  const int ir = ln.ir, col = ln.col;
  if (col >= pitch) return;                                 // (a whole wave: pitch is a multiple of 64, hence of QPW)
  const bool live_x = ir < NX, live_u = ir < NU;
  const int i = live_x ? ir : 0, j = live_u ? ir : 0;       // (other lanes shadow row 0: loads and arithmetic only)
  const size_t P_ = (size_t)pitch;
  const double rho = rhov[col];
  const int sg = SEG ? (int)blockIdx.y : 0;
  const int ka = SEG ? seg_start[sg] : 0, kb = SEG ? seg_start[sg + 1] : N;
  auto across = [&](double v, int l) { return ln.across(v, l); };   // row l's value of this QP
#ifdef ADMM_NO_TILED_BOUNDS
  constexpr bool TB_ = false;
#else
  constexpr bool TB_ = TILED && PB && VFORM;      // per-instance box read as staged tiles
#endif
  struct OpsM { double Acol[NX], Kcol[NU], Om[SEG ? NU : 1], Bcol[NX], Si[NU], lox, hix, lou, hiu; };      // matrix operands of a stage (+ its box, TB_)
  struct OpsS { double x0, x1, u0, u1, qx, qu, lox, hix, lou, hiu; };                  // its state / bound scalars
  auto loadS = [&](OpsS& o, int k) {
    const size_t ox = ((size_t)k * NB + NU + i) * P_ + col, ou = ((size_t)k * NB + j) * P_ + col;
    o.x0 = z[ox]; o.u0 = z[ou];
    o.x1 = VFORM ? 0.0 : y[ox]; o.u1 = VFORM ? 0.0 : y[ou];
    o.qx = HASQ ? q[ox] : 0.0; o.qu = HASQ ? q[ou] : 0.0;
    if (VFORM) {
      if (TB_) { }                               // (the box comes with the staged tiles: lread)
      else if (PB) { o.lox = lo[ox]; o.hix = hi[ox]; o.lou = lo[ou]; o.hiu = hi[ou]; }
      else {
        o.lox = lo[(size_t)k * NB + NU + i]; o.hix = hi[(size_t)k * NB + NU + i];
        o.lou = lo[(size_t)k * NB + j]; o.hiu = hi[(size_t)k * NB + j];
      }
    } else {
      o.lox = o.hix = o.lou = o.hiu = 0.0;
    }
  };
  double t = 0.0, es = 0.0;
  // `pre`: the value that is clipped -- s0, or s0 scaled onto the thrust-magnitude ball (control rows, SOC)
  auto gterm = [&](double s0, double pre, double s1, double lo_, double hi_, double qv) {
    double zz = s0, yy;
    if (VFORM) { zz = fmin(fmax(pre, lo_), hi_); yy = s0 - zz; } else { yy = s1; }
    double g = -rho * (zz - yy);
    if (HASQ) g += qv;
    return g;
  };
  auto body = [&](const OpsM& m, const OpsS& o, int k, bool valid) {
    // thrust-magnitude bound of this stage (branch-free: ub = +inf gives exactly 1): ||v_u|| over the QP's control
    // rows, read across the lanes and summed in row order -- the one-lane kernels' (and the oracle's) operations
    double cs = 1.0;
    if constexpr (SOC) {
      double ss = 0.0;
#pragma unroll
      for (int l = 0; l < NU; ++l) { const double vu = across(o.u0, l); ss = fma(vu, vu, ss); }
      const double nrm = sqrt(ss), ub = ubd[k];
      cs = nrm > ub ? ub / nrm : 1.0;
    }
    const double gx = gterm(o.x0, o.x0, o.x1, TB_ ? m.lox : o.lox, TB_ ? m.hix : o.hix, o.qx);
    const double gu = gterm(o.u0, SOC ? o.u0 * cs : o.u0, o.u1, TB_ ? m.lou : o.lou, TB_ ? m.hiu : o.hiu, o.qu);
    const double p = gx + t;
    double pv[NX], hv[NU];
#pragma unroll
    for (int l = 0; l < NX; ++l) pv[l] = across(p, l);
    double h = gu;
#pragma unroll
    for (int l = 0; l < NX; ++l) h = fma(m.Bcol[l], pv[l], h);
#pragma unroll
    for (int l = 0; l < NU; ++l) hv[l] = across(h, l);
    double d = 0.0;
#pragma unroll
    for (int l = 0; l < NU; ++l) d = fma(m.Si[l], hv[l], d);
    if (valid && live_u) dbuf[((size_t)k * NU + j) * P_ + col] = d;
    if (SEG) {
      double a = es;
#pragma unroll
      for (int l = 0; l < NU; ++l) a = fma(m.Om[l], across(d, l), a);
      es = valid ? a : es;
    }
    double a = 0.0;
#pragma unroll
    for (int l = 0; l < NX; ++l) a = fma(m.Acol[l], pv[l], a);
#pragma unroll
    for (int l = 0; l < NU; ++l) a = fma(-m.Kcol[l], hv[l], a);
    t = valid ? a : t;
  };
  const int len = kb - ka;                    // stage u of the segment is k = kb - 1 - u
  if constexpr (!TILED) {
    const Operand<NX, false> opA(Ad, NX * NX, P_, col), opB(Bd, NX * NU, P_, col), opK(Kd, NU * NX, P_, col), opS(Sd, NU * NU, P_, col),
        opO(SEG ? Omd : Ad, NX * NU, P_, col);
    struct Ops { OpsM m; OpsS s; };
    auto load = [&](Ops& o, int k) {
      const double *Ak = opA.stage(k), *Bk = opB.stage(k), *Kk = opK.stage(k), *Sk = opS.stage(k), *Ok = opO.stage(k);
#pragma unroll
      for (int l = 0; l < NX; ++l) {
        o.m.Acol[l] = Ak[opA.el(i * NX + l)];      // A[l][i]
        o.m.Bcol[l] = Bk[opB.el(j * NX + l)];      // B[l][j]
      }
#pragma unroll
      for (int l = 0; l < NU; ++l) {
        o.m.Kcol[l] = Kk[opK.el(l * NX + i)];      // K[l][i]
        o.m.Si[l] = Sk[opS.el(j * NU + l)];
        if (SEG) o.m.Om[l] = Ok[opO.el(i * NU + l)];
      }
      loadS(o.s, k);
    };
    // groups of D stages, two register sets alternate
    const int ngroups = (len + D - 1) / D;
    auto load_group = [&](Ops (&o)[D], int u0) {
#pragma unroll
      for (int u = 0; u < D; ++u) load(o[u], (u0 + u < len) ? kb - 1 - (u0 + u) : ka);
    };
    auto run_group = [&](const Ops (&o)[D], int u0) {
#pragma unroll
      for (int u = 0; u < D; ++u) body(o[u].m, o[u].s, kb - 1 - (u0 + u), u0 + u < len);
    };
    Ops A[D], B[D];
    load_group(A, 0);
    for (int g = 0; g < ngroups; g += 2) {
      if (g + 1 < ngroups) load_group(B, (g + 1) * D);
      run_group(A, g * D);
      if (g + 1 >= ngroups) break;
      if (g + 2 < ngroups) load_group(A, (g + 2) * D);
      run_group(B, (g + 1) * D);
    }
  } else {
    // Wide shapes: the matrix operands staged through LDS (admm_pinst.hpp, StageTile).  Slot u % 2 of the wave's LDS holds stage u;
    // two register sets hold the raw tiles of stages u + 1 and u + 2, in flight.  (Stages past the segment's end re-read its last.)
    typedef StageTile<NX * NX, NX, QPW> TA;
    typedef StageTile<NX * NU, NX, QPW> TB;
    typedef StageTile<NU * NX, NX, QPW> TK;
    typedef StageTile<NU * NU, NU, QPW> TS;
    typedef StageTile<NX * NU, NU, QPW> TO;
    typedef StageTile<NB, NB, QPW> TX;             // the box: [n + m][c] per stage and QP group
    constexpr int offB = TA::LWORDS, offK = offB + TB::LWORDS, offS = offK + TK::LWORDS, offO = offS + TS::LWORDS,
                  offL = offO + (SEG ? TO::LWORDS : 0), offH = offL + (TB_ ? TX::LWORDS : 0), SLOT = offH + (TB_ ? TX::LWORDS : 0);
    extern __shared__ pair_t prows_lds[];
    double* slot0 = reinterpret_cast<double*>(prows_lds) + (size_t)(threadIdx.x / PI_THREADS) * 2 * SLOT;
    double* slot1 = slot0 + SLOT;
    const int lane = threadIdx.x % PI_THREADS, c = ln.c;
    const size_t G = P_ / QPW, g = (size_t)(col / QPW);
    struct Raw { pair_t a[TA::NLD], b[TB::NLD], kk[TK::NLD], s[TS::NLD], o[SEG ? TO::NLD : 1], xl[TB_ ? TX::NLD : 1], xh[TB_ ? TX::NLD : 1]; };
    auto gload = [&](Raw& r, int k) {
      const size_t kg = ((size_t)k * G + g) * QPW;
      TA::gload(r.a, Ad + kg * (NX * NX), lane);
      TB::gload(r.b, Bd + kg * (NX * NU), lane);
      TK::gload(r.kk, Kd + kg * (NU * NX), lane);
      TS::gload(r.s, Sd + kg * (NU * NU), lane);
      if constexpr (SEG) TO::gload(r.o, Omd + kg * (NX * NU), lane);
      if constexpr (TB_) { TX::gload(r.xl, loT + kg * NB, lane); TX::gload(r.xh, hiT + kg * NB, lane); }
    };
    auto lstore = [&](const Raw& r, double* slot) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      TA::lstore(r.a, slot, lane);
      TB::lstore(r.b, slot + offB, lane);
      TK::lstore(r.kk, slot + offK, lane);
      TS::lstore(r.s, slot + offS, lane);
      if constexpr (SEG) TO::lstore(r.o, slot + offO, lane);
      if constexpr (TB_) { TX::lstore(r.xl, slot + offL, lane); TX::lstore(r.xh, slot + offH, lane); }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    };
    auto lread = [&](OpsM& m, const double* slot) {
#pragma unroll
      for (int l = 0; l < NX; ++l) {
        m.Acol[l] = slot[TA::at(i, l, c)];             // A[l][i]: element i * n + l
        m.Bcol[l] = slot[offB + TB::at(j, l, c)];      // B[l][j]: element j * n + l
      }
#pragma unroll
      for (int l = 0; l < NU; ++l) {
        m.Kcol[l] = slot[offK + TK::at(l, i, c)];      // K[l][i]: element l * n + i
        m.Si[l] = slot[offS + TS::at(j, l, c)];
        if (SEG) m.Om[l] = slot[offO + TO::at(i, l, c)];
      }
      if constexpr (TB_) {
        m.lox = slot[offL + TX::at(0, NU + i, c)]; m.hix = slot[offH + TX::at(0, NU + i, c)];
        m.lou = slot[offL + TX::at(0, j, c)];      m.hiu = slot[offH + TX::at(0, j, c)];
      }
    };
    auto ku = [&](int u) { return u < len ? kb - 1 - u : ka; };
    Raw R0, R1;
    OpsS s0, s1;
    gload(R0, ku(0));
    gload(R1, ku(1));
    loadS(s0, ku(0));
    lstore(R0, slot0);
    gload(R0, ku(2));
    for (int u = 0; u < len; u += 2) {
      // the waves of a block read the same 128-byte lines of the batch-minor arrays (state, bounds, d): kept in step, the line one
      // wave has fetched is still cached when the others ask (measured: 2.73 -> 2.25 ms per sweep at (12, 6), 4096 x 1000)
      __syncthreads();
      loadS(s1, ku(u + 1));
      lstore(R1, slot1);
      gload(R1, ku(u + 3));
      {
        OpsM m;
        lread(m, slot0);
        body(m, s0, ku(u), true);
      }
      if (u + 1 >= len) break;
      __syncthreads();
      loadS(s0, ku(u + 2));
      lstore(R0, slot0);
      gload(R0, ku(u + 4));
      {
        OpsM m;
        lread(m, slot1);
        body(m, s1, ku(u + 1), true);
      }
    }
  }
  if (SEG && live_x) {
    tseg[((size_t)sg * NX + i) * P_ + col] = t;
    eseg[((size_t)sg * NX + i) * P_ + col] = es;
  }
}

// ---------------------------------------------------------------------------
// Forward rollout (+ z-update, dual ascent, residual partials when ZUP; + w stored when STOREW), rows over lanes
// (the arithmetic of pxfz_kernel):   u = -K x - d [- Psi t_in];  x <- A x + B u;   ZUP: v+ = wh + y_old -> v
// ---------------------------------------------------------------------------
template <int NX, int NU, bool ZUP, bool RESID, bool RELAX, bool VIN, bool PB, bool STOREW, bool SEG, bool TILED = false, bool SOC = false>
__global__ __launch_bounds__(PROWS_BLOCK) void pxfz_rows_kernel(
    const double* __restrict__ dbuf, const double* __restrict__ x0, const double* __restrict__ Ad,
    const double* __restrict__ Bd, const double* __restrict__ Kd, const double* __restrict__ lo,
    const double* __restrict__ hi, const double* __restrict__ zin, const double* __restrict__ yin,
    double* __restrict__ v, double* __restrict__ w, double* __restrict__ part, double alpha, int N, int pitch,
    const double* __restrict__ Psd, const int* __restrict__ seg_start, const double* __restrict__ tin,
    const double* __restrict__ xin, const double* __restrict__ loT = nullptr, const double* __restrict__ hiT = nullptr,
    const double* __restrict__ ubd = nullptr) {
  constexpr int NB = NX + NU, D = ProwsDepth<NX>::D, QPW = PscanShape<NX>::QPW;
  static_assert(!SOC || ZUP, "the read-out form projects nothing");
  constexpr bool NEEDZ = RESID || RELAX;
  static_assert(NU <= NX && QPW * NX <= PI_THREADS, "rows over lanes: m <= n, n x QPW lanes");
  const RowsLane<NX> ln;
  const int ir = ln.ir, col = ln.col;
  if (col >= pitch) return;
  const bool live_x = ir < NX, live_u = ir < NU;
  const int i = live_x ? ir : 0, j = live_u ? ir : 0;
  const size_t P_ = (size_t)pitch;
  const int sg = SEG ? (int)blockIdx.y : 0;
  const int ka = SEG ? seg_start[sg] : 0, kb = SEG ? seg_start[sg + 1] : N;
  auto across = [&](double val, int l) { return ln.across(val, l); };
  double x = SEG ? xin[((size_t)sg * NX + i) * P_ + col] : x0[(size_t)i * P_ + col];
  double ti[SEG ? NX : 1];
#pragma unroll
  for (int l = 0; l < (SEG ? NX : 1); ++l) ti[l] = SEG ? tin[((size_t)sg * NX + l) * P_ + col] : 0.0;
  // state operands as in pxfz_kernel: VIN: s0 = v;  else s0 = y and (NEEDZ) s1 = z
  const double* st0 = ZUP ? (VIN ? v : yin) : dbuf;
#ifdef ADMM_NO_TILED_BOUNDS
  constexpr bool TB_ = false;
#else
  constexpr bool TB_ = TILED && PB && ZUP;        // per-instance box read as staged tiles
#endif
  struct OpsM { double Krow[NX], Ps[SEG ? NX : 1], Arow[NX], Brow[NU], lox, hix, lou, hiu; };      // matrix operands of a stage (+ its box, TB_)
  struct OpsS { double dj, x0, x1, u0, u1, lox, hix, lou, hiu; };               // its feed-forward / state / bound scalars
  auto loadS = [&](OpsS& o, int k) {
    o.dj = dbuf[((size_t)k * NU + j) * P_ + col];
    const size_t ox = ((size_t)k * NB + NU + i) * P_ + col, ou = ((size_t)k * NB + j) * P_ + col;
    if (ZUP) {
      o.x0 = st0[ox]; o.u0 = st0[ou];
      const bool two = !VIN && NEEDZ;
      o.x1 = two ? zin[ox] : 0.0; o.u1 = two ? zin[ou] : 0.0;
      if (TB_) { }                               // (the box comes with the staged tiles: lread)
      else if (PB) { o.lox = lo[ox]; o.hix = hi[ox]; o.lou = lo[ou]; o.hiu = hi[ou]; }
      else {
        o.lox = lo[(size_t)k * NB + NU + i]; o.hix = hi[(size_t)k * NB + NU + i];
        o.lou = lo[(size_t)k * NB + j]; o.hiu = hi[(size_t)k * NB + j];
      }
    } else {
      o.x0 = o.x1 = o.u0 = o.u1 = o.lox = o.hix = o.lou = o.hiu = 0.0;
    }
  };
  double accx[5] = {0, 0, 0, 0, 0}, accu[5] = {0, 0, 0, 0, 0};
  // cs_old / cs_new: the thrust-magnitude factors of the state before / after this z-update (1 unless SOC and a control row)
  auto zrow = [&](double wv, double s0, double s1, double lo_, double hi_, size_t o, bool store, double (&acc)[5], double cs_old, double cs_new) {
    if (STOREW && store) w[o] = wv;
    if (ZUP) {
      double zo, yo;
      if (VIN) { zo = fmin(fmax(SOC ? s0 * cs_old : s0, lo_), hi_); yo = s0 - zo; }
      else { yo = s0; zo = NEEDZ ? s1 : 0.0; }
      double wh = wv;
      if (RELAX) wh = fma(alpha, wv, (1.0 - alpha) * zo);
      const double vn = wh + yo;
      if (store) v[o] = vn;
      if (RESID && store) {
        const double zn = fmin(fmax(SOC ? vn * cs_new : vn, lo_), hi_);
        const double yn = vn - zn;
        const double dr = wv - zn, ds = zn - zo;
        acc[0] = fma(dr, dr, acc[0]);
        acc[1] = fma(ds, ds, acc[1]);
        acc[2] = fma(wv, wv, acc[2]);
        acc[3] = fma(zn, zn, acc[3]);
        acc[4] = fma(yn, yn, acc[4]);
      }
    }
  };
  auto body = [&](const OpsM& m, const OpsS& o, int k, bool valid) {
    double xv[NX], uv[NU];
#pragma unroll
    for (int l = 0; l < NX; ++l) xv[l] = across(x, l);
    double a = o.dj;
    if (SEG) {
#pragma unroll
      for (int l = 0; l < NX; ++l) a = fma(m.Ps[l], ti[l], a);
    }
#pragma unroll
    for (int l = 0; l < NX; ++l) a = fma(m.Krow[l], xv[l], a);
    const double wu = -a;
#pragma unroll
    for (int l = 0; l < NU; ++l) uv[l] = across(wu, l);
    double b = 0.0;
#pragma unroll
    for (int l = 0; l < NX; ++l) b = fma(m.Arow[l], xv[l], b);
#pragma unroll
    for (int l = 0; l < NU; ++l) b = fma(m.Brow[l], uv[l], b);
    x = valid ? b : x;
    const size_t ox = ((size_t)k * NB + NU + i) * P_ + col, ou = ((size_t)k * NB + j) * P_ + col;
    const double lou_ = TB_ ? m.lou : o.lou, hiu_ = TB_ ? m.hiu : o.hiu;
    // thrust-magnitude bound (as in pxfz_kernel): the factors of the state before and after this z-update, branch-free; the norms run
    // over the QP's control rows, read across the lanes in row order
    double cs_old = 1.0, cs_new = 1.0;
    if constexpr (SOC) {
      const double ub = ubd[k];
      if (VIN) {
        double ss = 0.0;
#pragma unroll
        for (int l = 0; l < NU; ++l) { const double vu = across(o.u0, l); ss = fma(vu, vu, ss); }
        const double nrm = sqrt(ss);
        cs_old = nrm > ub ? ub / nrm : 1.0;
      }
      if (RESID) {                                  // z+ needs ||v+_u||: this lane's control row of v+ first
        double zo, yo;
        if (VIN) { zo = fmin(fmax(o.u0 * cs_old, lou_), hiu_); yo = o.u0 - zo; }
        else { yo = o.u0; zo = NEEDZ ? o.u1 : 0.0; }
        double wh = wu;
        if (RELAX) wh = fma(alpha, wu, (1.0 - alpha) * zo);
        const double vnu = wh + yo;
        double ss = 0.0;
#pragma unroll
        for (int l = 0; l < NU; ++l) { const double vu = across(vnu, l); ss = fma(vu, vu, ss); }
        const double nrm = sqrt(ss);
        cs_new = nrm > ub ? ub / nrm : 1.0;
      }
    }
    zrow(wu, o.u0, o.u1, lou_, hiu_, ou, valid && live_u, accu, cs_old, cs_new);
    zrow(b, o.x0, o.x1, TB_ ? m.lox : o.lox, TB_ ? m.hix : o.hix, ox, valid && live_x, accx, 1.0, 1.0);
  };
  const int len = kb - ka;                    // stage u of the segment is k = ka + u
  if constexpr (!TILED) {
    const Operand<NX, false> opA(Ad, NX * NX, P_, col), opB(Bd, NX * NU, P_, col), opK(Kd, NU * NX, P_, col),
        opP(SEG ? Psd : Kd, NU * NX, P_, col);
    struct Ops { OpsM m; OpsS s; };
    auto load = [&](Ops& o, int k) {
      const double *Ak = opA.stage(k), *Bk = opB.stage(k), *Kk = opK.stage(k), *Pk = opP.stage(k);
#pragma unroll
      for (int l = 0; l < NX; ++l) {
        o.m.Krow[l] = Kk[opK.el(j * NX + l)];      // K[j][l]
        o.m.Arow[l] = Ak[opA.el(l * NX + i)];      // A[i][l]
        if (SEG) o.m.Ps[l] = Pk[opP.el(j * NX + l)];
      }
#pragma unroll
      for (int l = 0; l < NU; ++l) o.m.Brow[l] = Bk[opB.el(l * NX + i)];   // B[i][l]
      loadS(o.s, k);
    };
    const int ngroups = (len + D - 1) / D;
    auto load_group = [&](Ops (&o)[D], int u0) {
#pragma unroll
      for (int u = 0; u < D; ++u) load(o[u], (u0 + u < len) ? ka + u0 + u : kb - 1);
    };
    auto run_group = [&](const Ops (&o)[D], int u0) {
#pragma unroll
      for (int u = 0; u < D; ++u) body(o[u].m, o[u].s, ka + u0 + u, u0 + u < len);
    };
    Ops A[D], B[D];
    load_group(A, 0);
    for (int g = 0; g < ngroups; g += 2) {
      if (g + 1 < ngroups) load_group(B, (g + 1) * D);
      run_group(A, g * D);
      if (g + 1 >= ngroups) break;
      if (g + 2 < ngroups) load_group(A, (g + 2) * D);
      run_group(B, (g + 1) * D);
    }
  } else {
    // Wide shapes: the matrix operands staged through LDS (as in pxb_rows_kernel)
    typedef StageTile<NX * NX, NX, QPW> TA;
    typedef StageTile<NX * NU, NX, QPW> TB;
    typedef StageTile<NU * NX, NX, QPW> TK;      // K and Psi: [m][n], row-major
    typedef StageTile<NB, NB, QPW> TX;             // the box: [n + m][c] per stage and QP group
    constexpr int offB = TA::LWORDS, offK = offB + TB::LWORDS, offP = offK + TK::LWORDS, offL = offP + (SEG ? TK::LWORDS : 0),
                  offH = offL + (TB_ ? TX::LWORDS : 0), SLOT = offH + (TB_ ? TX::LWORDS : 0);
    extern __shared__ pair_t prows_lds[];
    double* slot0 = reinterpret_cast<double*>(prows_lds) + (size_t)(threadIdx.x / PI_THREADS) * 2 * SLOT;
    double* slot1 = slot0 + SLOT;
    const int lane = threadIdx.x % PI_THREADS, c = ln.c;
    const size_t G = P_ / QPW, g = (size_t)(col / QPW);
    struct Raw { pair_t a[TA::NLD], b[TB::NLD], kk[TK::NLD], ps[SEG ? TK::NLD : 1], xl[TB_ ? TX::NLD : 1], xh[TB_ ? TX::NLD : 1]; };
    auto gload = [&](Raw& r, int k) {
      const size_t kg = ((size_t)k * G + g) * QPW;
      TA::gload(r.a, Ad + kg * (NX * NX), lane);
      TB::gload(r.b, Bd + kg * (NX * NU), lane);
      TK::gload(r.kk, Kd + kg * (NU * NX), lane);
      if constexpr (SEG) TK::gload(r.ps, Psd + kg * (NU * NX), lane);
      if constexpr (TB_) { TX::gload(r.xl, loT + kg * NB, lane); TX::gload(r.xh, hiT + kg * NB, lane); }
    };
    auto lstore = [&](const Raw& r, double* slot) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      TA::lstore(r.a, slot, lane);
      TB::lstore(r.b, slot + offB, lane);
      TK::lstore(r.kk, slot + offK, lane);
      if constexpr (SEG) TK::lstore(r.ps, slot + offP, lane);
      if constexpr (TB_) { TX::lstore(r.xl, slot + offL, lane); TX::lstore(r.xh, slot + offH, lane); }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    };
    auto lread = [&](OpsM& m, const double* slot) {
#pragma unroll
      for (int l = 0; l < NX; ++l) {
        m.Krow[l] = slot[offK + TK::at(j, l, c)];      // K[j][l]: element j * n + l
        m.Arow[l] = slot[TA::at(l, i, c)];             // A[i][l]: element l * n + i
        if (SEG) m.Ps[l] = slot[offP + TK::at(j, l, c)];
      }
#pragma unroll
      for (int l = 0; l < NU; ++l) m.Brow[l] = slot[offB + TB::at(l, i, c)];     // B[i][l]: element l * n + i
      if constexpr (TB_) {
        m.lox = slot[offL + TX::at(0, NU + i, c)]; m.hix = slot[offH + TX::at(0, NU + i, c)];
        m.lou = slot[offL + TX::at(0, j, c)];      m.hiu = slot[offH + TX::at(0, j, c)];
      }
    };
    auto ku = [&](int u) { return u < len ? ka + u : kb - 1; };
    Raw R0, R1;
    OpsS s0, s1;
    gload(R0, ku(0));
    gload(R1, ku(1));
    loadS(s0, ku(0));
    lstore(R0, slot0);
    gload(R0, ku(2));
    for (int u = 0; u < len; u += 2) {
      // the waves of a block read the same 128-byte lines of the batch-minor arrays (state, bounds, d): kept in step, the line one
      // wave has fetched is still cached when the others ask (measured: 2.73 -> 2.25 ms per sweep at (12, 6), 4096 x 1000)
      __syncthreads();
      loadS(s1, ku(u + 1));
      lstore(R1, slot1);
      gload(R1, ku(u + 3));
      {
        OpsM m;
        lread(m, slot0);
        body(m, s0, ku(u), true);
      }
      if (u + 1 >= len) break;
      __syncthreads();
      loadS(s0, ku(u + 2));
      lstore(R0, slot0);
      gload(R0, ku(u + 4));
      {
        OpsM m;
        lread(m, slot1);
        body(m, s1, ku(u + 1), true);
      }
    }
  }
  if (ZUP && RESID) {
    // a QP's partial sums: its m control rows, then its n state rows, added in that fixed order
    double* pp = part + (size_t)sg * 5 * P_;
#pragma unroll
    for (int q5 = 0; q5 < 5; ++q5) {
      double tot = 0.0;
#pragma unroll
      for (int l = 0; l < NU; ++l) tot += across(accu[q5], l);
#pragma unroll
      for (int l = 0; l < NX; ++l) tot += across(accx[q5], l);
      if (ir == 0) pp[(size_t)q5 * P_ + col] = tot;
    }
  }
}

// dynamic LDS of the staged (TILED) sweeps: doubles per wave (two slots)
template <int NX, int NU, bool SEG>
constexpr int pxb_rows_lds_words() {
  constexpr int Q = PscanShape<NX>::QPW;
  return 2 * (StageTile<NX * NX, NX, Q>::LWORDS + StageTile<NX * NU, NX, Q>::LWORDS + StageTile<NU * NX, NX, Q>::LWORDS +
              StageTile<NU * NU, NU, Q>::LWORDS + (SEG ? StageTile<NX * NU, NU, Q>::LWORDS : 0) + 2 * StageTile<NX + NU, NX + NU, Q>::LWORDS);
}
template <int NX, int NU, bool SEG>
constexpr int pxfz_rows_lds_words() {
  constexpr int Q = PscanShape<NX>::QPW;
  return 2 * (StageTile<NX * NX, NX, Q>::LWORDS + StageTile<NX * NU, NX, Q>::LWORDS + (SEG ? 2 : 1) * StageTile<NU * NX, NX, Q>::LWORDS +
              2 * StageTile<NX + NU, NX + NU, Q>::LWORDS);
}

}  // namespace admm

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
template <int NX, int NU, bool HASQ, bool VFORM, bool PB, bool SEG, bool TILED = false>
__global__ __launch_bounds__(PROWS_BLOCK) void pxb_rows_kernel(
    const double* __restrict__ z, const double* __restrict__ y, const double* __restrict__ q,
    const double* __restrict__ Ad, const double* __restrict__ Bd, const double* __restrict__ Kd,
    const double* __restrict__ Sd, const double* __restrict__ lo, const double* __restrict__ hi,
    double* __restrict__ dbuf, const double* __restrict__ rhov, int N, int pitch,
    const double* __restrict__ Omd, const int* __restrict__ seg_start, double* __restrict__ tseg, double* __restrict__ eseg) {
  constexpr int NB = NX + NU, D = ProwsDepth<NX>::D, QPW = PscanShape<NX>::QPW;
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
  struct Ops {
    double Acol[NX], Kcol[NU], Om[SEG ? NU : 1], Bcol[NX], Si[NU];
    double x0, x1, u0, u1, qx, qu, lox, hix, lou, hiu;
  };
  const Operand<NX, TILED> opA(Ad, NX * NX, P_, col), opB(Bd, NX * NU, P_, col), opK(Kd, NU * NX, P_, col), opS(Sd, NU * NU, P_, col),
      opO(SEG ? Omd : Ad, NX * NU, P_, col);
  auto load = [&](Ops& o, int k) {
    const double *Ak = opA.stage(k), *Bk = opB.stage(k), *Kk = opK.stage(k), *Sk = opS.stage(k), *Ok = opO.stage(k);
#pragma unroll
    for (int l = 0; l < NX; ++l) {
      o.Acol[l] = Ak[opA.el(i * NX + l)];      // A[l][i]
      o.Bcol[l] = Bk[opB.el(j * NX + l)];      // B[l][j]
    }
#pragma unroll
    for (int l = 0; l < NU; ++l) {
      o.Kcol[l] = Kk[opK.el(l * NX + i)];      // K[l][i]
      o.Si[l] = Sk[opS.el(j * NU + l)];
      if (SEG) o.Om[l] = Ok[opO.el(i * NU + l)];
    }
    const size_t ox = ((size_t)k * NB + NU + i) * P_ + col, ou = ((size_t)k * NB + j) * P_ + col;
    o.x0 = z[ox]; o.u0 = z[ou];
    o.x1 = VFORM ? 0.0 : y[ox]; o.u1 = VFORM ? 0.0 : y[ou];
    o.qx = HASQ ? q[ox] : 0.0; o.qu = HASQ ? q[ou] : 0.0;
    if (VFORM) {
      if (PB) { o.lox = lo[ox]; o.hix = hi[ox]; o.lou = lo[ou]; o.hiu = hi[ou]; }
      else {
        o.lox = lo[(size_t)k * NB + NU + i]; o.hix = hi[(size_t)k * NB + NU + i];
        o.lou = lo[(size_t)k * NB + j]; o.hiu = hi[(size_t)k * NB + j];
      }
    } else {
      o.lox = o.hix = o.lou = o.hiu = 0.0;
    }
  };
  double t = 0.0, es = 0.0;
  auto gterm = [&](double s0, double s1, double lo_, double hi_, double qv) {
    double zz = s0, yy;
    if (VFORM) { zz = fmin(fmax(s0, lo_), hi_); yy = s0 - zz; } else { yy = s1; }
    double g = -rho * (zz - yy);
    if (HASQ) g += qv;
    return g;
  };
  auto body = [&](const Ops& o, int k, bool valid) {
    const double gx = gterm(o.x0, o.x1, o.lox, o.hix, o.qx), gu = gterm(o.u0, o.u1, o.lou, o.hiu, o.qu);
    const double p = gx + t;
    double pv[NX], hv[NU];
#pragma unroll
    for (int l = 0; l < NX; ++l) pv[l] = across(p, l);
    double h = gu;
#pragma unroll
    for (int l = 0; l < NX; ++l) h = fma(o.Bcol[l], pv[l], h);
#pragma unroll
    for (int l = 0; l < NU; ++l) hv[l] = across(h, l);
    double d = 0.0;
#pragma unroll
    for (int l = 0; l < NU; ++l) d = fma(o.Si[l], hv[l], d);
    if (valid && live_u) dbuf[((size_t)k * NU + j) * P_ + col] = d;
    if (SEG) {
      double a = es;
#pragma unroll
      for (int l = 0; l < NU; ++l) a = fma(o.Om[l], across(d, l), a);
      es = valid ? a : es;
    }
    double a = 0.0;
#pragma unroll
    for (int l = 0; l < NX; ++l) a = fma(o.Acol[l], pv[l], a);
#pragma unroll
    for (int l = 0; l < NU; ++l) a = fma(-o.Kcol[l], hv[l], a);
    t = valid ? a : t;
  };
  // stage u of the segment is k = kb - 1 - u; groups of D stages, two register sets alternate
  const int len = kb - ka, ngroups = (len + D - 1) / D;
  auto load_group = [&](Ops (&o)[D], int u0) {
#pragma unroll
    for (int u = 0; u < D; ++u) load(o[u], (u0 + u < len) ? kb - 1 - (u0 + u) : ka);
  };
  auto run_group = [&](const Ops (&o)[D], int u0) {
#pragma unroll
    for (int u = 0; u < D; ++u) body(o[u], kb - 1 - (u0 + u), u0 + u < len);
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
  if (SEG && live_x) {
    tseg[((size_t)sg * NX + i) * P_ + col] = t;
    eseg[((size_t)sg * NX + i) * P_ + col] = es;
  }
}

// ---------------------------------------------------------------------------
// Forward rollout (+ z-update, dual ascent, residual partials when ZUP; + w stored when STOREW), rows over lanes
// (the arithmetic of pxfz_kernel):   u = -K x - d [- Psi t_in];  x <- A x + B u;   ZUP: v+ = wh + y_old -> v
// ---------------------------------------------------------------------------
template <int NX, int NU, bool ZUP, bool RESID, bool RELAX, bool VIN, bool PB, bool STOREW, bool SEG, bool TILED = false>
__global__ __launch_bounds__(PROWS_BLOCK) void pxfz_rows_kernel(
    const double* __restrict__ dbuf, const double* __restrict__ x0, const double* __restrict__ Ad,
    const double* __restrict__ Bd, const double* __restrict__ Kd, const double* __restrict__ lo,
    const double* __restrict__ hi, const double* __restrict__ zin, const double* __restrict__ yin,
    double* __restrict__ v, double* __restrict__ w, double* __restrict__ part, double alpha, int N, int pitch,
    const double* __restrict__ Psd, const int* __restrict__ seg_start, const double* __restrict__ tin,
    const double* __restrict__ xin) {
  constexpr int NB = NX + NU, D = ProwsDepth<NX>::D, QPW = PscanShape<NX>::QPW;
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
  struct Ops {
    double Krow[NX], Ps[SEG ? NX : 1], Arow[NX], Brow[NU], dj;
    double x0, x1, u0, u1, lox, hix, lou, hiu;
  };
  const Operand<NX, TILED> opA(Ad, NX * NX, P_, col), opB(Bd, NX * NU, P_, col), opK(Kd, NU * NX, P_, col),
      opP(SEG ? Psd : Kd, NU * NX, P_, col);
  auto load = [&](Ops& o, int k) {
    const double *Ak = opA.stage(k), *Bk = opB.stage(k), *Kk = opK.stage(k), *Pk = opP.stage(k);
#pragma unroll
    for (int l = 0; l < NX; ++l) {
      o.Krow[l] = Kk[opK.el(j * NX + l)];      // K[j][l]
      o.Arow[l] = Ak[opA.el(l * NX + i)];      // A[i][l]
      if (SEG) o.Ps[l] = Pk[opP.el(j * NX + l)];
    }
#pragma unroll
    for (int l = 0; l < NU; ++l) o.Brow[l] = Bk[opB.el(l * NX + i)];   // B[i][l]
    o.dj = dbuf[((size_t)k * NU + j) * P_ + col];
    const size_t ox = ((size_t)k * NB + NU + i) * P_ + col, ou = ((size_t)k * NB + j) * P_ + col;
    if (ZUP) {
      o.x0 = st0[ox]; o.u0 = st0[ou];
      const bool two = !VIN && NEEDZ;
      o.x1 = two ? zin[ox] : 0.0; o.u1 = two ? zin[ou] : 0.0;
      if (PB) { o.lox = lo[ox]; o.hix = hi[ox]; o.lou = lo[ou]; o.hiu = hi[ou]; }
      else {
        o.lox = lo[(size_t)k * NB + NU + i]; o.hix = hi[(size_t)k * NB + NU + i];
        o.lou = lo[(size_t)k * NB + j]; o.hiu = hi[(size_t)k * NB + j];
      }
    } else {
      o.x0 = o.x1 = o.u0 = o.u1 = o.lox = o.hix = o.lou = o.hiu = 0.0;
    }
  };
  double accx[5] = {0, 0, 0, 0, 0}, accu[5] = {0, 0, 0, 0, 0};
  auto zrow = [&](double wv, double s0, double s1, double lo_, double hi_, size_t o, bool store, double (&acc)[5]) {
    if (STOREW && store) w[o] = wv;
    if (ZUP) {
      double zo, yo;
      if (VIN) { zo = fmin(fmax(s0, lo_), hi_); yo = s0 - zo; }
      else { yo = s0; zo = NEEDZ ? s1 : 0.0; }
      double wh = wv;
      if (RELAX) wh = fma(alpha, wv, (1.0 - alpha) * zo);
      const double vn = wh + yo;
      if (store) v[o] = vn;
      if (RESID && store) {
        const double zn = fmin(fmax(vn, lo_), hi_);
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
  auto body = [&](const Ops& o, int k, bool valid) {
    double xv[NX], uv[NU];
#pragma unroll
    for (int l = 0; l < NX; ++l) xv[l] = across(x, l);
    double a = o.dj;
    if (SEG) {
#pragma unroll
      for (int l = 0; l < NX; ++l) a = fma(o.Ps[l], ti[l], a);
    }
#pragma unroll
    for (int l = 0; l < NX; ++l) a = fma(o.Krow[l], xv[l], a);
    const double wu = -a;
#pragma unroll
    for (int l = 0; l < NU; ++l) uv[l] = across(wu, l);
    double b = 0.0;
#pragma unroll
    for (int l = 0; l < NX; ++l) b = fma(o.Arow[l], xv[l], b);
#pragma unroll
    for (int l = 0; l < NU; ++l) b = fma(o.Brow[l], uv[l], b);
    x = valid ? b : x;
    const size_t ox = ((size_t)k * NB + NU + i) * P_ + col, ou = ((size_t)k * NB + j) * P_ + col;
    zrow(wu, o.u0, o.u1, o.lou, o.hiu, ou, valid && live_u, accu);
    zrow(b, o.x0, o.x1, o.lox, o.hix, ox, valid && live_x, accx);
  };
  const int len = kb - ka, ngroups = (len + D - 1) / D;
  auto load_group = [&](Ops (&o)[D], int u0) {
#pragma unroll
    for (int u = 0; u < D; ++u) load(o[u], (u0 + u < len) ? ka + u0 + u : kb - 1);
  };
  auto run_group = [&](const Ops (&o)[D], int u0) {
#pragma unroll
    for (int u = 0; u < D; ++u) body(o[u], ka + u0 + u, u0 + u < len);
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

}  // namespace admm

// admm_kernels_alt.hpp -- the two fused kernels of the alternating-direction iteration
// (DESIGN.md §4.8).  No reference counterpart exists (README.md:1-2 only).
//
// The plain iteration sweeps the horizon twice per x-update in opposite directions
// (xb_kernel backward, xfz_kernel forward), and the next iteration's backward sweep has to
// re-read the state v the forward sweep has just written.  Here consecutive iterations solve
// the same KKT system in opposite elimination orders:
//     even iteration:  backward elimination (Riccati)            + forward substitution (feedback + rollout)
//     odd iteration:   forward elimination (its mirror in time)   + backward substitution (feedback + backward rollout)
// so every substitution sweep runs in the direction of the NEXT iteration's elimination sweep
// and the two are one kernel: the fresh v+ of a stage is eliminated while still in registers.
//     xfze_kernel  (forward):   rollout (as xfz_kernel) | z-update | forward elimination of v+
//     xbze_kernel  (backward):  backward rollout        | z-update | backward elimination of v+
// One kernel + one scan per iteration.  Algorithmic HBM bytes per stacked element (n = 6, m = 3):
//     xfze: d read 2.67 + v read 8 + v+ written 8 + db written 2.67 = 21.33   (+8 with a linear term q)
//     xbze: db read 2.67 + v read 8 + v+ written 8 + d written 2.67 = 21.33   (+8 with q)
// against 29.33 for xb + xfz (each elimination leaves only its m-row feed-forward term per stage).
// Same layout, staging and addressing as admm_kernels.hpp.
#pragma once

#include "admm_kernels.hpp"

namespace admm {

// Cache policy of the fused kernels' state loads / stores (RowView AUX bits; A/B in DESIGN.md §4.8).
#ifndef ADMM_ALT_LOAD_AUX
#define ADMM_ALT_LOAD_AUX 0
#endif
#ifndef ADMM_ALT_STORE_AUX
#define ADMM_ALT_STORE_AUX 0
#endif

// Tuning constants of the fused kernels by block size nb = n + m (measured at (6, 3); larger blocks by
// register fit: 0 spills / no accumulator-register traffic on the operand ring, tools/alt_sweep.sh).
// A macro, where defined, overrides the rule (A/B sweeps).
//   prefetch depth (stages) and LDS operand pairs read ahead of their FMAs, forward / backward kernel
constexpr int alt_pf_f(int nb, bool hasq, bool soc, int xfree = 0) {
#ifdef ADMM_ALT_PF_F_XFREE
  if (xfree && !hasq && !soc) return ADMM_ALT_PF_F_XFREE;
#endif
#ifdef ADMM_ALT_PF_F
  return ADMM_ALT_PF_F;
#else
  (void)soc; (void)xfree;
#if ADMM_DPP_OPERANDS
  // With the operators distributed over the lanes (dpp_matvec_acc) the kernels no longer wait on the LDS return path, and a
  // second stage of state rows in flight buys nothing (round 3 A/B at configs[2], residual / XFREE = 2 forms: depth 2
  // 131.9 / 74.9 us, depth 1 130.3 / 73.9 us; depths 2..4 of the XFREE forms were level before that too) -- while the
  // 2-stage unrolled body of the general non-residual form needs 256 registers + scratch next to the 30 operator registers.
  (void)nb; (void)hasq;
  return 1;
#else
  return (nb <= 9 && !hasq) ? 2 : 1;        // the q rows ride in the ring too
#endif
#endif
}
constexpr int alt_pf_b(int nb, int xfree = 0, bool hasq = false, bool soc = false) {
#ifdef ADMM_ALT_PF_B_XFREE
  if (xfree && !hasq && !soc) return ADMM_ALT_PF_B_XFREE;
#endif
#ifdef ADMM_ALT_PF_B
  return ADMM_ALT_PF_B;
#else
  (void)xfree; (void)hasq; (void)soc;
  return 1;
#endif
}
constexpr int alt_g_f(int nb) {
#ifdef ADMM_ALT_G_F
  return ADMM_ALT_G_F;
#else
  return nb <= 9 ? 9 : 12;
#endif
}
constexpr int alt_g_b(int nb) {
#ifdef ADMM_ALT_G_B
  return ADMM_ALT_G_B;
#else
  return 12;
#endif
}
// Register budget: asking for two waves per SIMD caps a wave at 256 registers (arch + accumulator),
// which keeps the whole working set in arch VGPRs.  Left at (1, 2) for the small blocks the scheduler
// spends up to 512, parks the operand prefetch ring in accumulator registers and then serialises every
// prefetch load behind an `s_waitcnt vmcnt(0)` + v_accvgpr_write (measured: 3x slower).  Blocks above
// 14 rows (9 with q) do not fit 256 registers and take the 512 budget (the ring stays in arch VGPRs there).
constexpr int alt_min_waves(int nb, bool hasq, bool soc) {
#ifdef ADMM_ALT_MIN_WAVES
  return ADMM_ALT_MIN_WAVES;
#else
  // the q rows cost 2 nb registers per ring slot plus 2 nb live; q AND the thrust-magnitude bound together do not
  // fit 256 registers at n + m = 9 (37 registers went to scratch): those forms take the 512 budget (about 30
  // values parked in accumulator registers, no scratch)
  return nb <= (hasq ? (soc ? 6 : 9) : (soc ? 12 : 14)) ? 2 : 1;
#endif
}
#define ADMM_ALT_OCCUPANCY(NB_, HQ_, SOC_) __attribute__((amdgpu_waves_per_eu(alt_min_waves(NB_, HQ_, SOC_), 2)))

// the alternating kernels are compiled for every (n, m) pair of admm_dims_g*.hip
constexpr bool alt_dims(int nx, int nu) { return nx >= 1 && nu >= 1; }

// Sum of the (<= 8) split-K slabs of one scan output row (see xf_kernel).
__device__ __forceinline__ double scan_row(const double* base, size_t o, int nsplit, size_t split_stride) {
  double p[8];
#pragma unroll
  for (int sp = 0; sp < 8; ++sp) p[sp] = base[(size_t)(sp < nsplit ? sp : 0) * split_stride + o];
  double a = p[0];
#pragma unroll
  for (int sp = 1; sp < 8; ++sp) a += (sp < nsplit) ? p[sp] : 0.0;
  return a;
}

// ---------------------------------------------------------------------------
// Forward fused kernel.  One lane = one QP, blockIdx.y = segment, stages k = a .. b-1:
//     d  = d0_k + Psi_k t_in;  u = -K_k x - d;  x = A_k x + B_k u         (w block k = (u, x))
//     (z, y) = (clip(v), v - clip(v));  wh = alpha w + (1 - alpha) z;  v+ = wh + y   -> v (in place)
//     z+ = clip(v+), y+ = v+ - z+;  RESID: the five per-QP partial sums           -> part
//     g  = q - rho (z+ - y+)                              (linear term of the NEXT x-update; q if HASQ)
//     db  = DK_k mu + DG_k g^u                            (mu = 0 on entry)        -> dbb block k (m rows)
//     eb += OB_k db
//     mu  = FM_k mu + GA_k g^u + PI_k g^x
// and on exit mu -> mseg[s], eb -> ebseg[s].
// ---------------------------------------------------------------------------
template <int NX, int NU, bool RESID, bool RELAX, bool HASQ, bool SOC, int XFREE = 0>
__global__ __launch_bounds__(XB_THREADS) ADMM_ALT_OCCUPANCY(NX + NU, HASQ, SOC) void xfze_kernel(
    const double* __restrict__ dbuf, const double* __restrict__ tin, const double* __restrict__ xin,
    const double* __restrict__ recFE, const int* __restrict__ seg_start_, const double* __restrict__ q,
    double* __restrict__ v,
    double* __restrict__ dbb, double* __restrict__ mseg, double* __restrict__ epsseg,
    double* __restrict__ part, double alpha, double rho, int pitch, int nsplit, size_t split_stride) {
  constexpr int NB = NX + NU;
  constexpr RecFELayout LF = rec_fe_layout(NX, NU);
  constexpr int RF = LF.SIZE;
  constexpr int PF = alt_pf_f(NB, HASQ, SOC, XFREE);
  [[maybe_unused]] constexpr int ALT_G = alt_g_f(NB);
  constexpr int CH = stage_chunk(RF, PF);
  __shared__ __attribute__((aligned(16))) double rec[CH * RF + 16];      // (+ 16: the distributed read of the last stage's last register)
  // operators distributed over the lanes of a row (dpp_matvec_acc, admm_kernels.hpp), read from LDS two operators ahead of
  // their FMAs (ADMM_LD): PSI, K, A, B | z-update | DK, DG, OB, FM, GA, PI, then PSI, K of the next stage
  constexpr int NREG = (LF.LO + 15) / 16;
  double ops[NREG];
  const double* rec16 = rec + (threadIdx.x & 15);
#if ADMM_DPP_OPERANDS
#define ADMM_MV(R_, C_, NEG_, BLK_, X_, ACC_) dpp_matvec_acc<R_, C_, NEG_, LF.BLK_, NREG>(ops, X_, ACC_)
#define ADMM_LD(R_, C_, BLK_, PTR_) dpp_op_load<LF.BLK_, (R_) * (C_), NREG>(PTR_, ops)
#else
#define ADMM_MV(R_, C_, NEG_, BLK_, X_, ACC_) lds_matvec_acc<R_, C_, NEG_, ALT_G>(rf + LF.BLK_, X_, ACC_)
#define ADMM_LD(R_, C_, BLK_, PTR_) (void)(PTR_)
  (void)rec16; (void)ops;
#endif

  const int col_raw = grid_col_block() * XB_THREADS + threadIdx.x;
  const int col = col_raw < pitch ? col_raw : pitch - 1;      // clamped lanes: loads only (see xb_kernel)
  const unsigned lb_st = col_raw < pitch ? (unsigned)col * 8u : ROWVIEW_OOB;
  const int s = grid_segment();
  cint_p seg_start = as_const(seg_start_);
  const int k0 = seg_start[s], k1 = seg_start[s + 1];
  const size_t P = (size_t)pitch;
  const unsigned lb = (unsigned)col * 8u;
  const unsigned PB = (unsigned)pitch * 8u;
  // XFREE: every STATE row is unbounded at every stage (the input-box-only problems of configs[1..4]).  Such a row has
  // z = v, y = 0 identically, so v+ = w^ + y_old = w^ whatever v_old was: when neither the residuals nor the
  // over-relaxation need z_old (XFREE is only instantiated for !RESID && !RELAX), v_old of the state rows is NOT READ --
  // c0 = 0 stands in, and the unchanged arithmetic gives y_old = 0 - clip(0) = 0: bit-identical iterates for
  // 8 n / (n + m) fewer bytes per element (21.33 -> 16 B at n = 6, m = 3) on every iteration without residuals, and n
  // fewer registers per ring slot.
  // (round 3) XFREE = 1 together with RESID: the unbounded state rows are READ (the dual residual needs z_old = v_old) and
  // written, but skip the clip / dual arithmetic and the two sums that are identically zero -- for the blocks whose kernels are
  // bound by fp64 issue rather than HBM (n + m >= 12).
  static_assert(!XFREE || !RELAX, "XFREE needs no over-relaxation");
  static_assert(!(XFREE == 2 && RESID), "the residuals need v of every row");
  const RowView vv(v, (size_t)k0 * NB * P, (size_t)(k1 - k0) * NB * P * 8);
  const RowView vd(dbuf, (size_t)k0 * NU * P, (size_t)(k1 - k0) * NU * P * 8);
  const RowView vm(dbb, (size_t)k0 * NU * P, (size_t)(k1 - k0) * NU * P * 8);
  const RowView vq(HASQ ? q : v, (size_t)k0 * NB * P, (size_t)(k1 - k0) * NB * P * 8);
  double t[NX], x[NX], mu[NX], eps[NX];
  {
    const size_t o = (size_t)s * NX * P + col;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      t[i] = scan_row(tin, o + i * P, nsplit, split_stride);
      x[i] = scan_row(xin, o + i * P, nsplit, split_stride);
      mu[i] = 0.0;
      eps[i] = 0.0;
    }
  }
  double ld[PF][NU], l0[PF][NB], lq[PF][NB];
#pragma unroll
  for (int j = 0; j < PF; ++j) {
    const int kj = (k0 + j < k1) ? k0 + j : k1 - 1;
    const unsigned d0 = (unsigned)(kj - k0) * NU * PB;
#pragma unroll
    for (int jj = 0; jj < NU; ++jj) ld[j][jj] = vd.load<ADMM_ALT_LOAD_AUX>(lb, d0 + jj * PB);
    const unsigned r0 = (unsigned)(kj - k0) * NB * PB;
#pragma unroll
    for (int r = 0; r < NB; ++r) {
      l0[j][r] = (XFREE && !RESID && r >= NU) ? 0.0 : vv.load<ADMM_ALT_LOAD_AUX>(lb, r0 + r * PB);
      if (HASQ) lq[j][r] = vq.load<ADMM_ALT_LOAD_AUX>(lb, r0 + r * PB);
    }
  }
  double a_r = 0, a_s = 0, a_w = 0, a_z = 0, a_y = 0;
  for (int kc = k0; kc < k1; kc += CH) {
    const int khi = (kc + CH - 1 < k1 - 1) ? kc + CH - 1 : k1 - 1;
    __syncthreads();
    stage_records<XB_THREADS>(rec, recFE + (size_t)kc * RF, (khi - kc + 1) * RF, threadIdx.x);
    __syncthreads();
    ADMM_LD(NU, NX, PSI, rec16);                       // the chunk's first stage (later ones: at the end of their predecessor)
    ADMM_LD(NU, NX, K, rec16);
    for (int kb = kc; kb <= khi; kb += PF) {
#pragma unroll
      for (int j = 0; j < PF; ++j) {
        const int k = kb + j;
        if (k > khi) break;
        [[maybe_unused]] const double* rf = rec + (k - kc) * RF;
        const double* rf16 = rec16 + (k - kc) * RF;
        const double* rn16 = rec16 + ((k < khi ? k + 1 : khi) - kc) * RF;      // the next stage's record (clamped: re-read, unused)
        double d[NU], c0[NB], cq[NB];
#pragma unroll
        for (int jj = 0; jj < NU; ++jj) d[jj] = ld[j][jj];
#pragma unroll
        for (int r = 0; r < NB; ++r) {
          c0[r] = l0[j][r];
          if (HASQ) cq[r] = lq[j][r];
        }
        {  // refill this slot with stage k + PF (clamped; see xfz_kernel)
          const int kn = (k + PF < k1) ? k + PF : k1 - 1;
          const unsigned d0 = (unsigned)(kn - k0) * NU * PB;
#pragma unroll
          for (int jj = 0; jj < NU; ++jj) ld[j][jj] = vd.load<ADMM_ALT_LOAD_AUX>(lb, d0 + jj * PB);
          const unsigned r0 = (unsigned)(kn - k0) * NB * PB;
#pragma unroll
          for (int r = 0; r < NB; ++r) {
            l0[j][r] = (XFREE && !RESID && r >= NU) ? 0.0 : vv.load<ADMM_ALT_LOAD_AUX>(lb, r0 + r * PB);
            if (HASQ) lq[j][r] = vq.load<ADMM_ALT_LOAD_AUX>(lb, r0 + r * PB);
          }
        }
        // ---- substitution: rollout of stage k ----
        double wv[NB];
        {
          double uacc[NU], xn[NX];
#pragma unroll
          for (int jj = 0; jj < NU; ++jj) uacc[jj] = d[jj];
          ADMM_LD(NX, NX, A, rf16);
          ADMM_MV(NU, NX, false, PSI, t, uacc);
          ADMM_LD(NX, NU, B, rf16);
          ADMM_MV(NU, NX, false, K, x, uacc);
          double uu[NU];
#pragma unroll
          for (int jj = 0; jj < NU; ++jj) { wv[jj] = -uacc[jj]; uu[jj] = wv[jj]; }
#pragma unroll
          for (int i = 0; i < NX; ++i) xn[i] = 0.0;
          ADMM_LD(NU, NX, DK, rf16);
          ADMM_MV(NX, NX, false, A, x, xn);
          ADMM_LD(NU, NU, DG, rf16);
          ADMM_MV(NX, NU, false, B, uu, xn);
#pragma unroll
          for (int i = 0; i < NX; ++i) { wv[NU + i] = xn[i]; x[i] = xn[i]; }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- z-update, dual ascent, residual partials; g = linear term of the next x-update ----
        // the box of the block: read as two 16-byte-pair blocks up front -- except in the SOC forms, whose two
        // ball-projection factors (sqrt + division each) keep every row live at once: there the 4 (n + m) registers
        // of the box are what pushed the kernel into scratch, and each row reads its own bounds instead
        double mLO[SOC ? 2 : even_up(NB)], mHI[SOC ? 2 : even_up(NB)], g[NB];
        if constexpr (!SOC) {
          lds_block(rf + LF.LO, mLO);
          lds_block(rf + LF.HI, mHI);
        }
        const unsigned r0 = (unsigned)(k - k0) * NB * PB;
        // thrust-magnitude bound on this stage's control rows (SOC forms only): the two
        // ball-projection factors, of the old state and of v+.  BRANCH-FREE: a stage without the bound carries
        // ub = +inf, for which both factors are exactly 1 (norm > inf is false), and where the bound is finite the
        // box of the control rows is (-inf, inf) (validated at setup), so  z_u = clip(c v_u, lo, hi)  is the ball
        // projection in the one case and the box projection in the other, bit for bit.  (The first version branched
        // on the wave-uniform `ub < inf`: every row of the block then stayed live across the branch and the two
        // sqrt / division sequences inside it, and the kernels went to scratch at most shapes.)
        double cs_old = 1.0, cs_new = 1.0;
        if constexpr (SOC) {
          const double ub = rf[LF.UB];
          cs_old = soc_scale<NU, NB>(c0, ub);
          double vnu[NB];
#pragma unroll
          for (int r = 0; r < NB; ++r) {
            if (r < NU) {
              const double zo = fmin(fmax(c0[r] * cs_old, rf[LF.LO + r]), rf[LF.HI + r]), yo = c0[r] - zo;
              double wh = wv[r];
              if (RELAX) wh = fma(alpha, wv[r], (1.0 - alpha) * zo);
              vnu[r] = wh + yo;
            } else {
              vnu[r] = 0.0;
            }
          }
          cs_new = soc_scale<NU, NB>(vnu, ub);
        }
#pragma unroll
        for (int r = 0; r < NB; ++r) {
          if (XFREE && r >= NU) {
            // an unbounded state row: z = v, y = 0 identically, so v+ = w^ = w (no relaxation here), z+ = v+, y+ = 0 and
            // g = -rho z+ -- the clip / dual arithmetic of the general row (2 max, 2 min, 4 adds) would reproduce exactly
            // these values (up to the sign of a zero); the kernels without residuals are bound by fp64 issue, not by HBM
            if (XFREE != 2) vv.store<ADMM_ALT_STORE_AUX>(wv[r], lb_st, r0 + r * PB);
            g[r] = -rho * wv[r];
            if (HASQ) g[r] += cq[r];
            if (RESID) {          // w - z+ = 0 and y+ = 0 on this row; z+ = w, z_old = v_old
              const double ds = wv[r] - c0[r];
              a_s = fma(ds, ds, a_s);
              a_w = fma(wv[r], wv[r], a_w);
              a_z = fma(wv[r], wv[r], a_z);
            }
            continue;
          }
          const bool ball = SOC && r < NU;
          const double lo_r = SOC ? rf[LF.LO + r] : mLO[SOC ? 0 : r], hi_r = SOC ? rf[LF.HI + r] : mHI[SOC ? 0 : r];
          const double zo = fmin(fmax(ball ? c0[r] * cs_old : c0[r], lo_r), hi_r);
          const double yo = c0[r] - zo;
          double wh = wv[r];
          if (RELAX) wh = fma(alpha, wv[r], (1.0 - alpha) * zo);
          const double vn = wh + yo;
          if (!(XFREE == 2 && r >= NU))      // XFREE = 2: the next iteration does not read these rows either
            vv.store<ADMM_ALT_STORE_AUX>(vn, lb_st, r0 + r * PB);
          const double zn = fmin(fmax(ball ? vn * cs_new : vn, lo_r), hi_r);
          const double yn = vn - zn;
          g[r] = -rho * (zn - yn);
          if (HASQ) g[r] += cq[r];
          if (RESID) {
            const double dr = wv[r] - zn, ds = zn - zo;
            a_r = fma(dr, dr, a_r);
            a_s = fma(ds, ds, a_s);
            a_w = fma(wv[r], wv[r], a_w);
            a_z = fma(zn, zn, a_z);
            a_y = fma(yn, yn, a_y);
          }
          if (r % 3 == 2) __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- forward elimination of stage k for the next x-update ----
        {
          double gu[NU], gx[NX], dn[NU], mn[NX];
#pragma unroll
          for (int jj = 0; jj < NU; ++jj) { gu[jj] = g[jj]; dn[jj] = 0.0; }
#pragma unroll
          for (int i = 0; i < NX; ++i) { gx[i] = g[NU + i]; mn[i] = 0.0; }
          ADMM_LD(NX, NU, OB, rf16);
          ADMM_MV(NU, NX, false, DK, mu, dn);       // db = DK mu
          ADMM_LD(NX, NX, FM, rf16);
          ADMM_MV(NU, NU, false, DG, gu, dn);       //      + DG g^u
          const unsigned m0 = (unsigned)(k - k0) * NU * PB;
#pragma unroll
          for (int jj = 0; jj < NU; ++jj) vm.store<ADMM_ALT_STORE_AUX>(dn[jj], lb_st, m0 + jj * PB);
          ADMM_LD(NX, NU, GA, rf16);
          ADMM_MV(NX, NU, false, OB, dn, eps);      // eb += OB db
          __builtin_amdgcn_sched_barrier(0);
          ADMM_LD(NX, NX, PI, rf16);
          ADMM_MV(NX, NX, false, FM, mu, mn);
          // the next stage's first two operators, while this stage's last two run -- unless they share a register with them
          // (blocks of a few rows only: the whole record is one or two registers there)
          constexpr bool AHEAD = (LF.K + NU * NX - 1) / 16 < LF.FM / 16;
          if constexpr (AHEAD) ADMM_LD(NU, NX, PSI, rn16);
          ADMM_MV(NX, NU, false, GA, gu, mn);
          if constexpr (AHEAD) ADMM_LD(NU, NX, K, rn16);
          ADMM_MV(NX, NX, false, PI, gx, mn);
          if constexpr (!AHEAD) {
            __builtin_amdgcn_sched_barrier(0);
            ADMM_LD(NU, NX, PSI, rn16);
            ADMM_LD(NU, NX, K, rn16);
          }
#pragma unroll
          for (int i = 0; i < NX; ++i) mu[i] = mn[i];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if (col_raw < pitch) {
    const size_t o = (size_t)s * NX * P + col;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      mseg[o + i * P] = mu[i];
      epsseg[o + i * P] = eps[i];
    }
    if (RESID) {
      const size_t op = (size_t)s * 5 * P + col;
      part[op + 0 * P] = a_r;
      part[op + 1 * P] = a_s;
      part[op + 2 * P] = a_w;
      part[op + 3 * P] = a_z;
      part[op + 4 * P] = a_y;
    }
  }
}

#undef ADMM_MV
#undef ADMM_LD

// ---------------------------------------------------------------------------
// Backward fused kernel.  One lane = one QP, blockIdx.y = segment, stages k = b-1 .. a, with
// x = x_end(s) (the state at the segment's end), m_in = m_in(s) from the scan and t = e = 0 on entry:
//     d  = db_k + PSB_k m_in;   u = -KB_k x - d                          (w block k = (u, x);  x = x_{k+1})
//     x  <- AI_k x + AIB_k u                                             (x_k: backward rollout)
//     (z, y) = (clip(v), v - clip(v));  wh = alpha w + (1 - alpha) z;  v+ = wh + y   -> v (in place);
//     z+, y+, RESID partials as above
//     g  = -rho (z+ - y+)
//     p = g^x + t;  h = BT_k p + g^u;  d0_k = SI_k h -> dbuf;  t = AT_k p - KT_k h;  e += OM_k d0_k
// and on exit t -> tseg[s], e -> eseg[s]: exactly what xb_kernel leaves for the plain scan.
// ---------------------------------------------------------------------------
template <int NX, int NU, bool RESID, bool RELAX, bool HASQ, bool SOC, int XFREE = 0>
__global__ __launch_bounds__(XB_THREADS) ADMM_ALT_OCCUPANCY(NX + NU, HASQ, SOC) void xbze_kernel(
    const double* __restrict__ dbb, const double* __restrict__ min_, const double* __restrict__ xend,
    const double* __restrict__ recBE, const int* __restrict__ seg_start_, const double* __restrict__ q,
    double* __restrict__ v,
    double* __restrict__ dbuf, double* __restrict__ tseg, double* __restrict__ eseg,
    double* __restrict__ part, double alpha, double rho, int pitch, int nsplit, size_t split_stride) {
  constexpr int NB = NX + NU;
  constexpr RecBELayout LB = rec_be_layout(NX, NU);
  constexpr int RB = LB.SIZE;
  constexpr int PF = alt_pf_b(NB, XFREE, HASQ, SOC);
  [[maybe_unused]] constexpr int ALT_G = alt_g_b(NB);
  constexpr int CH = stage_chunk(RB, PF);
  __shared__ __attribute__((aligned(16))) double rec[CH * RB + 16];
  // operators distributed over the lanes of a row, read two operators ahead of their FMAs (see xfze_kernel):
  // PSB, KB, AI, AIB | z-update | BT, SI, AT, KT, OM, then PSB, KB of the next (= previous in time) stage
  constexpr int NREG = (LB.LO + 15) / 16;
  double ops[NREG];
  const double* rec16 = rec + (threadIdx.x & 15);
#if ADMM_DPP_OPERANDS
#define ADMM_MV(R_, C_, NEG_, BLK_, X_, ACC_) dpp_matvec_acc<R_, C_, NEG_, LB.BLK_, NREG>(ops, X_, ACC_)
#define ADMM_LD(R_, C_, BLK_, PTR_) dpp_op_load<LB.BLK_, (R_) * (C_), NREG>(PTR_, ops)
#else
#define ADMM_MV(R_, C_, NEG_, BLK_, X_, ACC_) lds_matvec_acc<R_, C_, NEG_, ALT_G>(rb + LB.BLK_, X_, ACC_)
#define ADMM_LD(R_, C_, BLK_, PTR_) (void)(PTR_)
  (void)rec16; (void)ops;
#endif

  const int col_raw = grid_col_block() * XB_THREADS + threadIdx.x;
  const int col = col_raw < pitch ? col_raw : pitch - 1;
  const unsigned lb_st = col_raw < pitch ? (unsigned)col * 8u : ROWVIEW_OOB;
  const int s = grid_segment();
  cint_p seg_start = as_const(seg_start_);
  const int k0 = seg_start[s], k1 = seg_start[s + 1];
  const size_t P = (size_t)pitch;
  const unsigned lb = (unsigned)col * 8u;
  const unsigned PB = (unsigned)pitch * 8u;
  // XFREE: every STATE row is unbounded at every stage (the input-box-only problems of configs[1..4]).  Such a row has
  // z = v, y = 0 identically, so v+ = w^ + y_old = w^ whatever v_old was: when neither the residuals nor the
  // over-relaxation need z_old (XFREE is only instantiated for !RESID && !RELAX), v_old of the state rows is NOT READ --
  // c0 = 0 stands in, and the unchanged arithmetic gives y_old = 0 - clip(0) = 0: bit-identical iterates for
  // 8 n / (n + m) fewer bytes per element (21.33 -> 16 B at n = 6, m = 3) on every iteration without residuals, and n
  // fewer registers per ring slot.
  // (round 3) XFREE = 1 together with RESID: the unbounded state rows are READ (the dual residual needs z_old = v_old) and
  // written, but skip the clip / dual arithmetic and the two sums that are identically zero -- for the blocks whose kernels are
  // bound by fp64 issue rather than HBM (n + m >= 12).
  static_assert(!XFREE || !RELAX, "XFREE needs no over-relaxation");
  static_assert(!(XFREE == 2 && RESID), "the residuals need v of every row");
  const RowView vv(v, (size_t)k0 * NB * P, (size_t)(k1 - k0) * NB * P * 8);
  const RowView vd(dbuf, (size_t)k0 * NU * P, (size_t)(k1 - k0) * NU * P * 8);
  const RowView vm(dbb, (size_t)k0 * NU * P, (size_t)(k1 - k0) * NU * P * 8);
  const RowView vq(HASQ ? q : v, (size_t)k0 * NB * P, (size_t)(k1 - k0) * NB * P * 8);
#ifdef ADMM_ABLATE_REVERSE   // timing-only diagnostic: the backward sweep walks ASCENDING addresses (wrong results)
#define SIDX(kk) (k1 - 1 - (kk))
#else
#define SIDX(kk) ((kk) - k0)
#endif
  double t[NX], e[NX], x[NX], mi[NX];
  {
    const size_t o = (size_t)s * NX * P + col;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      mi[i] = scan_row(min_, o + i * P, nsplit, split_stride);
      x[i] = scan_row(xend, o + i * P, nsplit, split_stride);
      t[i] = 0.0;
      e[i] = 0.0;
    }
  }
  double lm[PF][NU], l0[PF][NB], lq[PF][NB];
#pragma unroll
  for (int j = 0; j < PF; ++j) {
    const int kj = (k1 - 1 - j > k0) ? k1 - 1 - j : k0;
    const unsigned m0 = (unsigned)SIDX(kj) * NU * PB;
#pragma unroll
    for (int i = 0; i < NU; ++i) lm[j][i] = vm.load<ADMM_ALT_LOAD_AUX>(lb, m0 + i * PB);
    const unsigned r0 = (unsigned)SIDX(kj) * NB * PB;
#pragma unroll
    for (int r = 0; r < NB; ++r) {
      l0[j][r] = (XFREE && !RESID && r >= NU) ? 0.0 : vv.load<ADMM_ALT_LOAD_AUX>(lb, r0 + r * PB);
      if (HASQ) lq[j][r] = vq.load<ADMM_ALT_LOAD_AUX>(lb, r0 + r * PB);
    }
  }
  double a_r = 0, a_s = 0, a_w = 0, a_z = 0, a_y = 0;
  for (int kc = k1 - 1; kc >= k0; kc -= CH) {
    const int klo = (kc - CH + 1 > k0) ? kc - CH + 1 : k0;
    __syncthreads();
    stage_records<XB_THREADS>(rec, recBE + (size_t)klo * RB, (kc - klo + 1) * RB, threadIdx.x);
    __syncthreads();
    ADMM_LD(NU, NX, PSB, rec16 + (kc - klo) * RB);       // the chunk's first (= last in time) stage
    ADMM_LD(NU, NX, KB, rec16 + (kc - klo) * RB);
    for (int kb = kc; kb >= klo; kb -= PF) {
#pragma unroll
      for (int j = 0; j < PF; ++j) {
        const int k = kb - j;
        if (k < klo) break;
        [[maybe_unused]] const double* rb = rec + (k - klo) * RB;
        const double* rb16 = rec16 + (k - klo) * RB;
        const double* rn16 = rec16 + ((k > klo ? k - 1 : klo) - klo) * RB;     // the next stage's record (clamped)
        double c0[NB], d[NU], cq[NB];
#pragma unroll
        for (int r = 0; r < NB; ++r) {
          c0[r] = l0[j][r];
          if (HASQ) cq[r] = lq[j][r];
        }
#pragma unroll
        for (int jj = 0; jj < NU; ++jj) d[jj] = lm[j][jj];
        {  // refill this slot with stage k - PF (clamped: the re-read rows near the segment start
           // are overwritten by this lane only later, in program order, and the values are unused)
          const int kn = (k - PF > k0) ? k - PF : k0;
          const unsigned m0 = (unsigned)SIDX(kn) * NU * PB;
#pragma unroll
          for (int i = 0; i < NU; ++i) lm[j][i] = vm.load<ADMM_ALT_LOAD_AUX>(lb, m0 + i * PB);
          const unsigned r0 = (unsigned)SIDX(kn) * NB * PB;
#pragma unroll
          for (int r = 0; r < NB; ++r) {
            l0[j][r] = (XFREE && !RESID && r >= NU) ? 0.0 : vv.load<ADMM_ALT_LOAD_AUX>(lb, r0 + r * PB);
            if (HASQ) lq[j][r] = vq.load<ADMM_ALT_LOAD_AUX>(lb, r0 + r * PB);
          }
        }
        // ---- substitution: feedback law and backward rollout of stage k ----
        double wv[NB];
        {
          double uu[NU], xk[NX];
          ADMM_LD(NX, NX, AI, rb16);
          ADMM_MV(NU, NX, false, PSB, mi, d);       // d = db + PSB m_in
          ADMM_LD(NX, NU, AIB, rb16);
#pragma unroll
          for (int jj = 0; jj < NU; ++jj) uu[jj] = -d[jj];
          ADMM_MV(NU, NX, true, KB, x, uu);         // u = -d - KB x
#pragma unroll
          for (int jj = 0; jj < NU; ++jj) wv[jj] = uu[jj];
#pragma unroll
          for (int i = 0; i < NX; ++i) { wv[NU + i] = x[i]; xk[i] = 0.0; }
          __builtin_amdgcn_sched_barrier(0);
          ADMM_LD(NU, NX, BT, rb16);
          ADMM_MV(NX, NX, false, AI, x, xk);        // x_k = AI x_{k+1}
          ADMM_LD(NU, NU, SI, rb16);
          ADMM_MV(NX, NU, false, AIB, uu, xk);      //       + AIB u
#pragma unroll
          for (int i = 0; i < NX; ++i) x[i] = xk[i];
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- z-update, dual ascent, residual partials ----
        double g[NB];
        const unsigned r0 = (unsigned)SIDX(k) * NB * PB;
        double cs_old = 1.0, cs_new = 1.0;     // thrust-magnitude bound, branch-free (see xfze_kernel)
        if constexpr (SOC) {
          const double ub = rb[LB.UB];
          cs_old = soc_scale<NU, NB>(c0, ub);
          double vnu[NB];
#pragma unroll
          for (int r = 0; r < NB; ++r) {
            if (r < NU) {
              const double zo = fmin(fmax(c0[r] * cs_old, rb[LB.LO + r]), rb[LB.HI + r]), yo = c0[r] - zo;
              double wh = wv[r];
              if (RELAX) wh = fma(alpha, wv[r], (1.0 - alpha) * zo);
              vnu[r] = wh + yo;
            } else {
              vnu[r] = 0.0;
            }
          }
          cs_new = soc_scale<NU, NB>(vnu, ub);
        }
#pragma unroll
        for (int r3 = 0; r3 < NB; r3 += 3) {
#pragma unroll
          for (int r = r3; r < r3 + 3 && r < NB; ++r) {
            if (XFREE && r >= NU) {             // unbounded state row: see xfze_kernel
              if (XFREE != 2) vv.store<ADMM_ALT_STORE_AUX>(wv[r], lb_st, r0 + r * PB);
              g[r] = -rho * wv[r];
              if (HASQ) g[r] += cq[r];
              if (RESID) {
                const double ds = wv[r] - c0[r];
                a_s = fma(ds, ds, a_s);
                a_w = fma(wv[r], wv[r], a_w);
                a_z = fma(wv[r], wv[r], a_z);
              }
              continue;
            }
            const bool ball = SOC && r < NU;
            const double lo = rb[LB.LO + r], hi = rb[LB.HI + r];
            const double zo = fmin(fmax(ball ? c0[r] * cs_old : c0[r], lo), hi);
            const double yo = c0[r] - zo;
            double wh = wv[r];
            if (RELAX) wh = fma(alpha, wv[r], (1.0 - alpha) * zo);
            const double vn = wh + yo;
            if (!(XFREE == 2 && r >= NU))
              vv.store<ADMM_ALT_STORE_AUX>(vn, lb_st, r0 + r * PB);
            const double zn = fmin(fmax(ball ? vn * cs_new : vn, lo), hi);
            const double yn = vn - zn;
            g[r] = -rho * (zn - yn);
            if (HASQ) g[r] += cq[r];
            if (RESID) {
              const double dr = wv[r] - zn, ds = zn - zo;
              a_r = fma(dr, dr, a_r);
              a_s = fma(ds, ds, a_s);
              a_w = fma(wv[r], wv[r], a_w);
              a_z = fma(zn, zn, a_z);
              a_y = fma(yn, yn, a_y);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        // ---- backward elimination of stage k for the next x-update (as xb_kernel) ----
        {
          double p[NX], h[NU], d[NU];
#pragma unroll
          for (int i = 0; i < NX; ++i) p[i] = g[NU + i] + t[i];
#pragma unroll
          for (int jj = 0; jj < NU; ++jj) { h[jj] = g[jj]; d[jj] = 0.0; }
          ADMM_LD(NX, NX, AT, rb16);
          ADMM_MV(NU, NX, false, BT, p, h);
          ADMM_LD(NX, NU, KT, rb16);
          ADMM_MV(NU, NU, false, SI, h, d);
          const unsigned d0 = (unsigned)SIDX(k) * NU * PB;
#pragma unroll
          for (int jj = 0; jj < NU; ++jj) vd.store<ADMM_ALT_STORE_AUX>(d[jj], lb_st, d0 + jj * PB);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < NX; ++i) t[i] = 0.0;
          ADMM_LD(NX, NU, OM, rb16);
          ADMM_MV(NX, NX, false, AT, p, t);
          constexpr bool AHEAD = (LB.KB + NU * NX - 1) / 16 < LB.AT / 16;      // see xfze_kernel
          if constexpr (AHEAD) ADMM_LD(NU, NX, PSB, rn16);
          ADMM_MV(NX, NU, true, KT, h, t);
          if constexpr (AHEAD) ADMM_LD(NU, NX, KB, rn16);
          ADMM_MV(NX, NU, false, OM, d, e);
          if constexpr (!AHEAD) {
            __builtin_amdgcn_sched_barrier(0);
            ADMM_LD(NU, NX, PSB, rn16);
            ADMM_LD(NU, NX, KB, rn16);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if (col_raw < pitch) {
    const size_t o = (size_t)s * NX * P + col;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      tseg[o + i * P] = t[i];
      eseg[o + i * P] = e[i];
    }
    if (RESID) {
      const size_t op = (size_t)s * 5 * P + col;
      part[op + 0 * P] = a_r;
      part[op + 1 * P] = a_s;
      part[op + 2 * P] = a_w;
      part[op + 3 * P] = a_z;
      part[op + 4 * P] = a_y;
    }
  }
}

#undef ADMM_MV
#undef ADMM_LD

}  // namespace admm

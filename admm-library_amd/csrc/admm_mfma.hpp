// admm_mfma.hpp -- MFMA form of the fused iteration kernels (BASELINE.json configs[4]:
// "x-update with MFMA batched-GEMM reformulation").  No reference counterpart exists (README.md:1-2 only).
//
// Same iteration, same HBM arrays and the same segment scan as the one-lane-per-QP kernels of admm_kernels.hpp /
// admm_kernels_alt.hpp; what changes is how a stage's operators reach the arithmetic.  There every lane owns a QP
// and the stage matrices arrive as LDS broadcasts (2 LDS cycles per double per wave: 7200 cycles per stage and CU at
// n = 12, m = 6 -- the bound of those kernels at that size).  Here 16 QPs x one stage is a chain of
// v_mfma_{f64,f32}_16x16x4 products: the batch-minor panel is the B operand, the stage operators are pre-packed on
// the host as A fragments (one element per lane per MFMA, admm_mfma_layout.hpp), and the accumulator tile of one
// stage is, register for register, the B operand of the next.
//
//   xfzem_kernel<NX, NU, TS, TE, RESID, RELAX, ELIM>   forward:   SUB_F | z-update, dual, residual partials | ELIM_F
//       ELIM = true : the alternating path's forward fused kernel (xfze_kernel)
//       ELIM = false: the plain path's fused forward kernel (xfz_kernel, state in v-form)
//   xbzem_kernel<NX, NU, TS, TE, RESID, RELAX, SUBST>  backward:  SUB_B | z-update, dual, residual partials | ELIM_B
//       SUBST = true : the alternating path's backward fused kernel (xbze_kernel)
//       SUBST = false: the plain path's backward sweep (xb_kernel, state in v-form; v is only read)
//   TS / TE = element type of the SUB / ELIM product (operands, chains and accumulators):
//   all double (ADMM_PRECISION_FP64_MFMA): exact -- iterates agree with the fp64 one-lane kernels to rounding (the
//       operators are folded on the host, e.g. A - B K, so the rounding differs at the 1e-16 level).
//   mixed (ADMM_PRECISION_MIXED): the Riccati form's two products, SUB_F and ELIM_B (operators O(1)), in fp32 on the
//       fp32 matrix pipe (half the cycles per MFMA); the forward-elimination form's ELIM_F and SUB_B, whose
//       early-stage gains reach 2.5e4, in fp64.  So xfzem<float, double> / xbzem<double, float> on the alternating
//       path; v, the z-update, the dual and the residuals are fp64 in every mode.
// Thrust-magnitude bound and linear term q: not in this form (admm_setup refuses the combination).
//
// Geometry: workgroup = 512 threads = 8 waves, 2 waves per SIMD (<= 256 registers each); a wave owns MF_NT = 2
// tiles of 16 QPs, so a workgroup covers 256 QPs of one segment -- the same grid as the one-lane kernels,
// (ceil(pitch / 256), S), one workgroup per CU at configs[2..4]'s batch.
#pragma once

#include "admm_kernels.hpp"
#include "admm_kernels_alt.hpp"
#include "admm_mfma_layout.hpp"

namespace admm {

constexpr int MF_THREADS = 512;
// 16-QP tiles per wave: template parameter NT of the kernels.  2 for batches that fill the chip (a workgroup then covers
// 256 QPs, the one-lane kernels' grid); 1 for small batches, where waves whose columns lie beyond the batch skip the
// arithmetic altogether -- a single QP is then ONE wave per segment running 22 dependent MFMAs per stage (0.65 us)
// instead of ~450 dependent vector instructions (2.2 us): the small-batch form of the iteration (configs[1]).
#ifndef ADMM_MF_PF
#define ADMM_MF_PF 2
#endif
#ifndef ADMM_MF_PF_B
#define ADMM_MF_PF_B 1
#endif
constexpr int MF_PF_F = ADMM_MF_PF;            // stages of operand prefetch (register ring) of the forward kernel ...
constexpr int MF_PF_B = ADMM_MF_PF_B;          // ... and of the backward one (2 spilled ~100 registers there); both divide the LDS chunk
constexpr int mf_cols(int nt) { return (MF_THREADS / 64) * nt * 16; }   // QPs per workgroup

typedef float mfma_f4 __attribute__((ext_vector_type(4)));

template <class T> struct MfmaOps;
template <> struct MfmaOps<double> {
  typedef mfma_d4 acc_t;
  static __device__ __forceinline__ acc_t mfma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
};
template <> struct MfmaOps<float> {
  typedef mfma_f4 acc_t;
  static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
};

// stages whose records sit in one LDS buffer (two buffers: <= ~80 KB of the CU's 160 KB)
constexpr int mfma_chunk(int rec_bytes, int pf) {
  int ch = 40000 / rec_bytes;
  ch = ch < 1 ? 1 : (ch > 4 ? 4 : ch);
  ch = (ch / pf) * pf;                         // ring slots stay aligned across chunks
  return ch < pf ? pf : ch;
}

// Asynchronous global -> LDS copy of one chunk of records (LDS-DMA, global_load_lds_dwordx4: no staging registers,
// no ds_write pass).  A wave-instruction moves 64 lanes x 16 B = one contiguous KiB to "M0 (wave-uniform LDS
// address) + lane x 16", so the chunk is copied in whole KiB pieces, wave w taking pieces w, w + 8, ...: the LDS
// buffers are rounded up to a KiB and the device arrays carry a KiB of slack, so the last piece may run past the
// chunk.
// Issued through inline asm ON PURPOSE: with an LDS-DMA it knows of in flight, hipcc (ROCm 7.2) turns every counted
// `s_waitcnt vmcnt(N)` into vmcnt(0) (cdna_hip_programming.md §5, "Pipelining across barriers") -- in the stage loop
// that made each stage wait for the previous stage's v+ STORES before touching its prefetched operands (measured:
// 371 us per launch instead of ~300 at n = 12).  Hidden from the compiler, the DMA is still counted by the hardware:
// a counted wait can only over-wait (the DMA is younger than the loads it is waiting for), never under-wait, and the
// chunk that is being filled is not read before glds_retire() + the barrier that ends the current chunk.
constexpr int mfma_lds_bytes(int chunk_bytes) { return ((chunk_bytes + 1023) / 1024) * 1024; }

__device__ __forceinline__ void glds_chunk(const unsigned char* src, unsigned char* lds_dst, int bytes, int wave, int lane) {
  const int pieces = (bytes + 1023) >> 10;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds_dst;
  for (int p = wave; p < pieces; p += MF_THREADS / 64) {
    const unsigned char* gsrc = src + (size_t)p * 1024 + lane * 16;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)p * 1024u);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
  }
}
// every LDS-DMA of this wave has landed (then a barrier makes the other waves' pieces visible too)
__device__ __forceinline__ void glds_retire() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// One register (4 rows x 16 QPs) of the z-update, dual ascent and residual partials; returns v+ and the linear
// term g = -rho (z+ - y+) of the next x-update.  Padding slots carry c0 = w = 0 and the box (-inf, inf): every
// result is then (+-) 0 and the sums are untouched.
template <bool RESID, bool RELAX>
__device__ __forceinline__ void mf_zupdate(double c0, double w, double lo, double hi, double alpha, double rho,
                                           double& vn, double& gg, double (&acc)[5]) {
  const double zo = fmin(fmax(c0, lo), hi);
  const double yo = c0 - zo;
  double wh = w;
  if (RELAX) wh = fma(alpha, w, (1.0 - alpha) * zo);
  vn = wh + yo;
  const double zn = fmin(fmax(vn, lo), hi);
  const double yn = vn - zn;
  gg = -rho * (zn - yn);
  if (RESID) {
    const double dr = w - zn, ds = zn - zo;
    acc[0] = fma(dr, dr, acc[0]);
    acc[1] = fma(ds, ds, acc[1]);
    acc[2] = fma(w, w, acc[2]);
    acc[3] = fma(zn, zn, acc[3]);
    acc[4] = fma(yn, yn, acc[4]);
  }
}

// One row of the scan output; the common case of one slab (every batch that fills the chip) takes ONE load instead of
// scan_row's eight (its unrolled form serialised the prologue here: 48 rows x 8 loads per lane).
__device__ __forceinline__ double mf_scan_row(const double* base, size_t o, int nsplit, size_t split_stride) {
  if (nsplit == 1) return base[o];          // wave-uniform branch
  return scan_row(base, o, nsplit, split_stride);
}

// Rows 4.. of u (admm_mfma_layout.hpp): ur[j] holds this lane's partial sum of row 4 + j; the four lane groups of a
// column are added, and lane group g keeps row 4 + g (slot (4, g)).
template <int UR>
__device__ __forceinline__ double mf_urow_place(const double (&ur)[UR > 0 ? UR : 1], int g) {
  double out = 0.0;
#pragma unroll
  for (int j = 0; j < UR; ++j) {
    double t = ur[j];
    t += __shfl_xor(t, 16, 64);
    t += __shfl_xor(t, 32, 64);
    out = (g == j) ? t : out;
  }
  return out;
}

// per-QP residual partials: a QP's rows live in the four lane groups of its column -> two cross-lane adds
__device__ __forceinline__ double mf_colsum(double x) {
  x += __shfl_xor(x, 16, 64);
  x += __shfl_xor(x, 32, 64);
  return x;
}

// ---------------------------------------------------------------------------
// Forward kernel.  Per 16-QP tile and stage k = a .. b-1 (block k = (u_k, x_{k+1})):
//     SUB_F :  [x+ ; u] = M [x ; t_in ; d_k]                       (d_k from dbuf; w block k = (u, x+))
//     z-update on block k with v (in place) -> v+, g
//     ELIM_F:  [mu+ ; deps ; db_k] = M [mu ; g^x ; g^u],  eps += deps,  db_k -> dbb          (ELIM only)
// and on exit mu -> mseg[s], eps -> epsseg[s] (ELIM), residual partials -> part (RESID).
// ---------------------------------------------------------------------------
// HASQ: a linear cost term q (block rows like v): g = -rho (z+ - y+) + q.
template <int NX, int NU, int NT, class TS, class TE, bool RESID, bool RELAX, bool ELIM, int XFREE = 0, bool HASQ = false>
__global__ __launch_bounds__(MF_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void xfzem_kernel(
    const double* __restrict__ dbuf, const double* __restrict__ tin, const double* __restrict__ xin,
    const unsigned char* __restrict__ recMF, const int* __restrict__ seg_start_, double* __restrict__ v,
    double* __restrict__ dbb, double* __restrict__ mseg, double* __restrict__ epsseg, double* __restrict__ part,
    double alpha, double rho, int pitch, int nsplit, size_t split_stride, int batch, const double* __restrict__ qlin = nullptr) {
  typedef MfmaOps<TS> OpsS;
  typedef MfmaOps<TE> OpsE;
  typedef typename OpsS::acc_t accs_t;
  typedef typename OpsE::acc_t acce_t;
  constexpr int MODE = sizeof(TS) == 4 ? 1 : 2;
  static_assert(mfma_dims(NX, NU), "MFMA form: n <= 12, m <= 8");
  constexpr int NB = NX + NU;
  constexpr MfmaLayout ML = mfma_layout(NX, NU);
  constexpr int XT = ML.xt, UR = ML.urows, KS = ML.ks_sub, KE = ML.ks_elim_f, NR = ML.nr;
  constexpr int RM = mfma_rec_bytes_fwd(NX, NU, MODE);
  static_assert(sizeof(TS) == mfma_es_sub_f(MODE) && (!ELIM || sizeof(TE) == mfma_es_elim_f(MODE)), "element types vs record layout");
  constexpr int O_ELIM = ML.nf_sub * 64 * mfma_es_sub_f(MODE);     // byte offset of the ELIM fragments
  constexpr int O_LOHI = O_ELIM + ML.nf_elim_f * 64 * mfma_es_elim_f(MODE);   // byte offset of lo / hi
  constexpr int MF_PF = MF_PF_F;
  constexpr int CH = mfma_chunk(RM, MF_PF);
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][mfma_lds_bytes(CH * RM)];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int s = grid_segment();
  cint_p seg_start = as_const(seg_start_);
  const int k0 = seg_start[s], k1 = seg_start[s + 1];
  const size_t P = (size_t)pitch;
  const unsigned PB = (unsigned)pitch * 8u;
  const RowView vv(v, (size_t)k0 * NB * P, (size_t)(k1 - k0) * NB * P * 8);
  const RowView vd(dbuf, (size_t)k0 * NU * P, (size_t)(k1 - k0) * NU * P * 8);
  const RowView vm(dbb, (size_t)k0 * NU * P, (size_t)(k1 - k0) * NU * P * 8);
  const RowView vq(HASQ ? qlin : v, (size_t)k0 * NB * P, (size_t)(k1 - k0) * NB * P * 8);

  // slot validity of this lane group (compile-time in r, run-time in g)
  bool okx[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) okx[r] = 4 * r + g < NX;
  const bool oku0 = g < NU, oku1 = 4 + g < NU;

  int col[NT];
  unsigned lbl[NT], lbs[NT];             // lane byte offsets for loads (clamped column) / stores (dropped if clamped)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col_raw = grid_col_block() * mf_cols(NT) + (wave * NT + nt) * 16 + c;
    col[nt] = col_raw < pitch ? col_raw : pitch - 1;
    lbl[nt] = ((unsigned)g * (unsigned)pitch + (unsigned)col[nt]) * 8u;
    lbs[nt] = col_raw < pitch ? lbl[nt] : ROWVIEW_OOB;
  }
  // a wave all of whose columns lie beyond the batch (pad columns of the last 64, or the unused waves of a small
  // batch) takes part in the record staging and the barriers only
  const bool wave_active = grid_col_block() * mf_cols(NT) + wave * NT * 16 < batch;
  // XFREE: the state rows are unbounded at every stage and neither residuals nor over-relaxation need z_old: their v is
  // not read (xfze_kernel explains why that is exact)
  static_assert(!XFREE || (!RESID && !RELAX), "XFREE needs z_old of no row");

  TS X[NT][NR], Tin[NT][NR];
  TE Mu[NT][NR], Eps[NT][NR];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const size_t o = ((size_t)s * NX + (okx[r] ? 4 * r + g : 0)) * P + col[nt];
      const double tv = mf_scan_row(tin, o, nsplit, split_stride), xv = mf_scan_row(xin, o, nsplit, split_stride);
      Tin[nt][r] = okx[r] ? (TS)tv : (TS)0;
      X[nt][r] = okx[r] ? (TS)xv : (TS)0;
      Mu[nt][r] = (TE)0;
      Eps[nt][r] = (TE)0;
    }
  double racc[NT][5];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int q = 0; q < 5; ++q) racc[nt][q] = 0.0;

  // operand prefetch ring, MF_PF stages deep: v rows (5 registers) and d rows (2) per tile and slot
  double pv[MF_PF][NT][5], pd[MF_PF][NT][2], pq[HASQ ? MF_PF : 1][NT][5];
  auto load_stage = [&](int k, int j, int nt) {
    const int kk = k < k1 ? k : k1 - 1;
    const unsigned r0 = (unsigned)(kk - k0) * NB * PB;
#pragma unroll
    for (int r = 0; r < NR; ++r) pv[j][nt][r] = XFREE ? 0.0 : vv.load(lbl[nt], r0 + (unsigned)(NU + 4 * r) * PB);
    pv[j][nt][3] = vv.load(lbl[nt], r0);
    pv[j][nt][4] = XT ? vv.load(lbl[nt], r0 + 4u * PB) : 0.0;
    if (HASQ) {
#pragma unroll
      for (int r = 0; r < NR; ++r) pq[HASQ ? j : 0][nt][r] = vq.load(lbl[nt], r0 + (unsigned)(NU + 4 * r) * PB);
      pq[HASQ ? j : 0][nt][3] = vq.load(lbl[nt], r0);
      pq[HASQ ? j : 0][nt][4] = XT ? vq.load(lbl[nt], r0 + 4u * PB) : 0.0;
    }
    const unsigned d0 = (unsigned)(kk - k0) * NU * PB;
    pd[j][nt][0] = vd.load(lbl[nt], d0);
    pd[j][nt][1] = XT ? vd.load(lbl[nt], d0 + 4u * PB) : 0.0;
  };
#pragma unroll
  for (int j = 0; j < MF_PF; ++j)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) load_stage(k0 + j, j, nt);

  glds_chunk(recMF + (size_t)k0 * RM, lds[0], ((k1 - k0 < CH) ? k1 - k0 : CH) * RM, wave, lane);
  glds_retire();
  __syncthreads();
  int buf = 0;
  for (int kc = k0; kc < k1; kc += CH) {
    const int khi = (kc + CH < k1) ? kc + CH : k1;               // this chunk: stages kc .. khi-1
    const int nnext = (khi < k1) ? ((khi + CH < k1 ? CH : k1 - khi) * RM) : 0;
    if (nnext) glds_chunk(recMF + (size_t)khi * RM, lds[buf ^ 1], nnext, wave, lane);   // next chunk, in flight during this one
    for (int kb = kc; kb < khi; kb += MF_PF) {
#pragma unroll
     for (int j = 0; j < MF_PF; ++j) {
      const int k = kb + j;
      if (k >= khi) break;
      if (!wave_active) continue;
      const unsigned char* rec = lds[buf] + (k - kc) * RM;
      const TS* af = reinterpret_cast<const TS*>(rec);
      const double* lohi = reinterpret_cast<const double*>(rec + O_LOHI);
      double c0[NT][5], dk[NT][2], qk[HASQ ? NT : 1][5];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int r = 0; r < NR; ++r) c0[nt][r] = okx[r] ? pv[j][nt][r] : 0.0;
        c0[nt][3] = oku0 ? pv[j][nt][3] : 0.0;
        c0[nt][4] = (XT && oku1) ? pv[j][nt][4] : 0.0;
        dk[nt][0] = oku0 ? pd[j][nt][0] : 0.0;
        dk[nt][1] = (XT && oku1) ? pd[j][nt][1] : 0.0;
        if (HASQ) {
#pragma unroll
          for (int r = 0; r < NR; ++r) qk[HASQ ? nt : 0][r] = okx[r] ? pq[HASQ ? j : 0][nt][r] : 0.0;
          qk[HASQ ? nt : 0][3] = oku0 ? pq[HASQ ? j : 0][nt][3] : 0.0;
          qk[HASQ ? nt : 0][4] = (XT && oku1) ? pq[HASQ ? j : 0][nt][4] : 0.0;
        }
        load_stage(k + MF_PF, j, nt);
      }
      // ---- SUB_F ----
      accs_t a0[NT];
      double ur[NT][UR > 0 ? UR : 1];
      const double* urow = lohi + 40;                          // [row - 4][ks][g]
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        a0[nt] = accs_t{0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < (UR > 0 ? UR : 1); ++j) ur[nt][j] = 0.0;
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const TS fa0 = af[ks * 64 + lane];
        double uc[UR > 0 ? UR : 1];
#pragma unroll
        for (int j = 0; j < UR; ++j) uc[j] = urow[(j * KS + ks) * 4 + g];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const TS b = ks < NR ? X[nt][ks < NR ? ks : 0] : (ks < 2 * NR ? Tin[nt][(ks >= NR && ks < 2 * NR) ? ks - NR : 0] : (TS)dk[nt][ks == 2 * NR ? 0 : 1]);
          a0[nt] = OpsS::mfma(fa0, b, a0[nt]);
#pragma unroll
          for (int j = 0; j < UR; ++j) ur[nt][j] = fma(uc[j], (double)b, ur[nt][j]);
        }
      }
      double u1[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) u1[nt] = XT ? mf_urow_place<UR>(ur[nt], g) : 0.0;
      __builtin_amdgcn_sched_barrier(0);
      // ---- z-update, dual ascent, residual partials; v+ stored in place ----
      double gg[NT][5];
      const unsigned r0 = (unsigned)(k - k0) * NB * PB;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        double vn;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          if (XFREE) {            // unbounded state rows: v+ = w, z+ = v+, y+ = 0, g = -rho w (see xfze_kernel): no clip / dual arithmetic
            vn = (double)a0[nt][r];
            gg[nt][r] = -rho * vn;
          } else {
            mf_zupdate<RESID, RELAX>(c0[nt][r], (double)a0[nt][r], lohi[r * 4 + g], lohi[20 + r * 4 + g], alpha, rho, vn, gg[nt][r], racc[nt]);
          }
          if (XFREE != 2) vv.store(vn, okx[r] ? lbs[nt] : ROWVIEW_OOB, r0 + (unsigned)(NU + 4 * r) * PB);
          X[nt][r] = a0[nt][r];
        }
        mf_zupdate<RESID, RELAX>(c0[nt][3], (double)a0[nt][3], lohi[12 + g], lohi[32 + g], alpha, rho, vn, gg[nt][3], racc[nt]);
        vv.store(vn, oku0 ? lbs[nt] : ROWVIEW_OOB, r0);
        if (XT) {
          mf_zupdate<RESID, RELAX>(c0[nt][4], u1[nt], lohi[16 + g], lohi[36 + g], alpha, rho, vn, gg[nt][4], racc[nt]);
          vv.store(vn, oku1 ? lbs[nt] : ROWVIEW_OOB, r0 + 4u * PB);
        } else {
          gg[nt][4] = 0.0;
        }
        if (HASQ) {
#pragma unroll
          for (int q5 = 0; q5 < 5; ++q5)
            if (q5 < NR || q5 >= 3) gg[nt][q5] += qk[HASQ ? nt : 0][q5];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- ELIM_F ----
      if (ELIM) {
        const TE* ae = reinterpret_cast<const TE*>(rec + O_ELIM);
        acce_t e0[NT], e1[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          e0[nt] = acce_t{0, 0, 0, 0};
          e1[nt] = acce_t{Eps[nt][0], NR > 1 ? Eps[nt][NR > 1 ? 1 : 0] : (TE)0, NR > 2 ? Eps[nt][NR > 2 ? 2 : 0] : (TE)0, 0};
        }
#pragma unroll
        for (int ks = 0; ks < KE; ++ks) {
          const TE fa0 = ae[(ks * 2 + 0) * 64 + lane], fa1 = ae[(ks * 2 + 1) * 64 + lane];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const TE b = ks < NR ? Mu[nt][ks < NR ? ks : 0] : (TE)gg[nt][ks < 2 * NR ? (ks >= NR ? ks - NR : 0) : (ks == 2 * NR ? 3 : 4)];
#ifndef ADMM_MF_ABLATE_ELIM     // timing-only diagnostic: no ELIM MFMAs (wrong results)
            e0[nt] = OpsE::mfma(fa0, b, e0[nt]);
            e1[nt] = OpsE::mfma(fa1, b, e1[nt]);
#else
            e0[nt][0] += fa0 * b; e1[nt][0] += fa1 * b;
#endif
          }
        }
        const unsigned m0 = (unsigned)(k - k0) * NU * PB;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
          for (int r = 0; r < NR; ++r) { Mu[nt][r] = e0[nt][r]; Eps[nt][r] = e1[nt][r]; }
          vm.store((double)e0[nt][3], oku0 ? lbs[nt] : ROWVIEW_OOB, m0);
          if (XT) vm.store((double)e1[nt][3], oku1 ? lbs[nt] : ROWVIEW_OOB, m0 + 4u * PB);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
     }
    }
    if (nnext) glds_retire();
    __syncthreads();
    buf ^= 1;
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const bool st = wave_active && lbs[nt] != ROWVIEW_OOB;
    if (ELIM) {
#pragma unroll
      for (int r = 0; r < NR; ++r)
        if (st && okx[r]) {
          const size_t o = ((size_t)s * NX + 4 * r + g) * P + col[nt];
          mseg[o] = (double)Mu[nt][r];
          epsseg[o] = (double)Eps[nt][r];
        }
    }
    if (RESID) {
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        const double t = mf_colsum(racc[nt][q]);
        if (st && g == 0) part[((size_t)s * 5 + q) * P + col[nt]] = t;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Backward kernel.  Per 16-QP tile and stage k = b-1 .. a (x = x_{k+1} on entry of a stage):
//     SUB_B :  [x_k ; u] = M [x ; m_in ; db_k]      (db_k from dbb; w block k = (u, x_{k+1}))          (SUBST only)
//     z-update on block k -> v+, g                   (SUBST = false: g = -rho (z - y) of the CURRENT state, v only read)
//     ELIM_B:  [t+ ; de ; d0_k] = M [g^x + t ; g^u],  e += de,  d0_k -> dbuf
// and on exit t -> tseg[s], e -> eseg[s]: what xb_kernel / xbze_kernel leave for the plain scan.
// ---------------------------------------------------------------------------
template <int NX, int NU, int NT, class TS, class TE, bool RESID, bool RELAX, bool SUBST, int XFREE = 0, bool HASQ = false>
__global__ __launch_bounds__(MF_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void xbzem_kernel(
    const double* __restrict__ dbb, const double* __restrict__ min_, const double* __restrict__ xend,
    const unsigned char* __restrict__ recMB, const int* __restrict__ seg_start_, double* __restrict__ v,
    double* __restrict__ dbuf, double* __restrict__ tseg, double* __restrict__ eseg, double* __restrict__ part,
    double alpha, double rho, int pitch, int nsplit, size_t split_stride, int batch, const double* __restrict__ qlin = nullptr) {
  typedef MfmaOps<TS> OpsS;
  typedef MfmaOps<TE> OpsE;
  typedef typename OpsS::acc_t accs_t;
  typedef typename OpsE::acc_t acce_t;
  constexpr int MODE = sizeof(TE) == 4 ? 1 : 2;
  static_assert(mfma_dims(NX, NU), "MFMA form: n <= 12, m <= 8");
  constexpr int NB = NX + NU;
  constexpr MfmaLayout ML = mfma_layout(NX, NU);
  constexpr int XT = ML.xt, UR = ML.urows, KS = ML.ks_sub, KE = ML.ks_elim_b, NR = ML.nr;
  constexpr int RM = mfma_rec_bytes_bwd(NX, NU, MODE);
  static_assert(sizeof(TE) == mfma_es_elim_b(MODE) && (!SUBST || sizeof(TS) == mfma_es_sub_b(MODE)), "element types vs record layout");
  constexpr int O_ELIM = ML.nf_sub * 64 * mfma_es_sub_b(MODE);
  constexpr int O_LOHI = O_ELIM + ML.nf_elim_b * 64 * mfma_es_elim_b(MODE);
  constexpr int MF_PF = MF_PF_B;
  constexpr int CH = mfma_chunk(RM, MF_PF);
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][mfma_lds_bytes(CH * RM)];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int s = grid_segment();
  cint_p seg_start = as_const(seg_start_);
  const int k0 = seg_start[s], k1 = seg_start[s + 1];
  const size_t P = (size_t)pitch;
  const unsigned PB = (unsigned)pitch * 8u;
  const RowView vv(v, (size_t)k0 * NB * P, (size_t)(k1 - k0) * NB * P * 8);
  const RowView vd(dbuf, (size_t)k0 * NU * P, (size_t)(k1 - k0) * NU * P * 8);
  const RowView vm(dbb, (size_t)k0 * NU * P, (size_t)(k1 - k0) * NU * P * 8);
  const RowView vq(HASQ ? qlin : v, (size_t)k0 * NB * P, (size_t)(k1 - k0) * NB * P * 8);

  bool okx[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) okx[r] = 4 * r + g < NX;
  const bool oku0 = g < NU, oku1 = 4 + g < NU;

  int col[NT];
  unsigned lbl[NT], lbs[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col_raw = grid_col_block() * mf_cols(NT) + (wave * NT + nt) * 16 + c;
    col[nt] = col_raw < pitch ? col_raw : pitch - 1;
    lbl[nt] = ((unsigned)g * (unsigned)pitch + (unsigned)col[nt]) * 8u;
    lbs[nt] = col_raw < pitch ? lbl[nt] : ROWVIEW_OOB;
  }
  // a wave all of whose columns lie beyond the batch (pad columns of the last 64, or the unused waves of a small
  // batch) takes part in the record staging and the barriers only
  const bool wave_active = grid_col_block() * mf_cols(NT) + wave * NT * 16 < batch;
  // XFREE: the state rows are unbounded at every stage and neither residuals nor over-relaxation need z_old: their v is
  // not read (xfze_kernel explains why that is exact)
  static_assert(!XFREE || (!RESID && !RELAX), "XFREE needs z_old of no row");

  TS X[NT][NR], Min[NT][NR];
  TE Tt[NT][NR], Ee[NT][NR];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      if (SUBST) {
        const size_t o = ((size_t)s * NX + (okx[r] ? 4 * r + g : 0)) * P + col[nt];
        const double mv = mf_scan_row(min_, o, nsplit, split_stride), xv = mf_scan_row(xend, o, nsplit, split_stride);
        Min[nt][r] = okx[r] ? (TS)mv : (TS)0;
        X[nt][r] = okx[r] ? (TS)xv : (TS)0;
      } else {
        Min[nt][r] = (TS)0;
        X[nt][r] = (TS)0;
      }
      Tt[nt][r] = (TE)0;
      Ee[nt][r] = (TE)0;
    }
  double racc[NT][5];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int q = 0; q < 5; ++q) racc[nt][q] = 0.0;

  double pv[MF_PF][NT][5], pd[MF_PF][NT][2], pq[HASQ ? MF_PF : 1][NT][5];
  auto load_stage = [&](int k, int j, int nt) {
    const int kk = k > k0 ? k : k0;
    const unsigned r0 = (unsigned)(kk - k0) * NB * PB;
#pragma unroll
    for (int r = 0; r < NR; ++r) pv[j][nt][r] = XFREE ? 0.0 : vv.load(lbl[nt], r0 + (unsigned)(NU + 4 * r) * PB);
    pv[j][nt][3] = vv.load(lbl[nt], r0);
    pv[j][nt][4] = XT ? vv.load(lbl[nt], r0 + 4u * PB) : 0.0;
    if (HASQ) {
#pragma unroll
      for (int r = 0; r < NR; ++r) pq[HASQ ? j : 0][nt][r] = vq.load(lbl[nt], r0 + (unsigned)(NU + 4 * r) * PB);
      pq[HASQ ? j : 0][nt][3] = vq.load(lbl[nt], r0);
      pq[HASQ ? j : 0][nt][4] = XT ? vq.load(lbl[nt], r0 + 4u * PB) : 0.0;
    }
    if (SUBST) {
      const unsigned d0 = (unsigned)(kk - k0) * NU * PB;
      pd[j][nt][0] = vm.load(lbl[nt], d0);
      pd[j][nt][1] = XT ? vm.load(lbl[nt], d0 + 4u * PB) : 0.0;
    } else {
      pd[j][nt][0] = pd[j][nt][1] = 0.0;
    }
  };
#pragma unroll
  for (int j = 0; j < MF_PF; ++j)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) load_stage(k1 - 1 - j, j, nt);

  {  // first chunk: stages k1-CH .. k1-1 (clipped at k0); LDS slot j = k - klo
    const int klo = (k1 - CH > k0) ? k1 - CH : k0;
    glds_chunk(recMB + (size_t)klo * RM, lds[0], (k1 - klo) * RM, wave, lane);
  }
  glds_retire();
  __syncthreads();
  int buf = 0;
  for (int kc = k1; kc > k0; kc -= CH) {                       // this chunk: stages klo .. kc-1, descending
    const int klo = (kc - CH > k0) ? kc - CH : k0;
    const int nlo = (klo - CH > k0) ? klo - CH : k0;           // next chunk: stages nlo .. klo-1
    const int nnext = (klo > k0) ? (klo - nlo) * RM : 0;
    if (nnext) glds_chunk(recMB + (size_t)nlo * RM, lds[buf ^ 1], nnext, wave, lane);
    for (int kb = kc - 1; kb >= klo; kb -= MF_PF) {
#pragma unroll
     for (int j = 0; j < MF_PF; ++j) {
      const int k = kb - j;
      if (k < klo) break;
      if (!wave_active) continue;
      const unsigned char* rec = lds[buf] + (k - klo) * RM;
      const TS* af = reinterpret_cast<const TS*>(rec);
      const double* lohi = reinterpret_cast<const double*>(rec + O_LOHI);
      double c0[NT][5], dk[NT][2], qk[HASQ ? NT : 1][5];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int r = 0; r < NR; ++r) c0[nt][r] = okx[r] ? pv[j][nt][r] : 0.0;
        c0[nt][3] = oku0 ? pv[j][nt][3] : 0.0;
        c0[nt][4] = (XT && oku1) ? pv[j][nt][4] : 0.0;
        dk[nt][0] = oku0 ? pd[j][nt][0] : 0.0;
        dk[nt][1] = (XT && oku1) ? pd[j][nt][1] : 0.0;
        if (HASQ) {
#pragma unroll
          for (int r = 0; r < NR; ++r) qk[HASQ ? nt : 0][r] = okx[r] ? pq[HASQ ? j : 0][nt][r] : 0.0;
          qk[HASQ ? nt : 0][3] = oku0 ? pq[HASQ ? j : 0][nt][3] : 0.0;
          qk[HASQ ? nt : 0][4] = (XT && oku1) ? pq[HASQ ? j : 0][nt][4] : 0.0;
        }
        load_stage(k - MF_PF, j, nt);
      }
      double gg[NT][5];
      if (SUBST) {
        // ---- SUB_B ----
        accs_t a0[NT];
        double ur[NT][UR > 0 ? UR : 1];
        const double* urow = lohi + 40;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          a0[nt] = accs_t{0, 0, 0, 0};
#pragma unroll
          for (int j = 0; j < (UR > 0 ? UR : 1); ++j) ur[nt][j] = 0.0;
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const TS fa0 = af[ks * 64 + lane];
          double uc[UR > 0 ? UR : 1];
#pragma unroll
          for (int j = 0; j < UR; ++j) uc[j] = urow[(j * KS + ks) * 4 + g];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const TS b = ks < NR ? X[nt][ks < NR ? ks : 0] : (ks < 2 * NR ? Min[nt][(ks >= NR && ks < 2 * NR) ? ks - NR : 0] : (TS)dk[nt][ks == 2 * NR ? 0 : 1]);
            a0[nt] = OpsS::mfma(fa0, b, a0[nt]);
#pragma unroll
            for (int j = 0; j < UR; ++j) ur[nt][j] = fma(uc[j], (double)b, ur[nt][j]);
          }
        }
        double u1[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) u1[nt] = XT ? mf_urow_place<UR>(ur[nt], g) : 0.0;
        __builtin_amdgcn_sched_barrier(0);
        // ---- z-update: w block k = (u_k, x_{k+1}) ----
        const unsigned r0 = (unsigned)(k - k0) * NB * PB;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          double vn;
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            if (XFREE) {          // unbounded state rows: v+ = w, z+ = v+, y+ = 0, g = -rho w (see xfze_kernel)
              vn = (double)X[nt][r];
              gg[nt][r] = -rho * vn;
            } else {
              mf_zupdate<RESID, RELAX>(c0[nt][r], (double)X[nt][r], lohi[r * 4 + g], lohi[20 + r * 4 + g], alpha, rho, vn, gg[nt][r], racc[nt]);
            }
            if (XFREE != 2) vv.store(vn, okx[r] ? lbs[nt] : ROWVIEW_OOB, r0 + (unsigned)(NU + 4 * r) * PB);
            X[nt][r] = a0[nt][r];                                 // x_k
          }
          mf_zupdate<RESID, RELAX>(c0[nt][3], (double)a0[nt][3], lohi[12 + g], lohi[32 + g], alpha, rho, vn, gg[nt][3], racc[nt]);
          vv.store(vn, oku0 ? lbs[nt] : ROWVIEW_OOB, r0);
          if (XT) {
            mf_zupdate<RESID, RELAX>(c0[nt][4], u1[nt], lohi[16 + g], lohi[36 + g], alpha, rho, vn, gg[nt][4], racc[nt]);
            vv.store(vn, oku1 ? lbs[nt] : ROWVIEW_OOB, r0 + 4u * PB);
          } else {
            gg[nt][4] = 0.0;
          }
        }
      } else {
        // the plain path's backward sweep: linear term of the CURRENT state, z = clip(v), y = v - z
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int q = 0; q < 5; ++q) {
            const int slot = q * 4 + g;
            const double zz = fmin(fmax(c0[nt][q], lohi[slot]), lohi[20 + slot]);
            gg[nt][q] = ((q < NR || q == 3) || (q == 4 && XT)) ? -rho * (zz - (c0[nt][q] - zz)) : 0.0;
          }
      }
      if (HASQ) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int q5 = 0; q5 < 5; ++q5)
            if (q5 < NR || q5 == 3 || (q5 == 4 && XT)) gg[nt][q5] += qk[HASQ ? nt : 0][q5];
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- ELIM_B ----
      {
        const TE* ae = reinterpret_cast<const TE*>(rec + O_ELIM);
        acce_t e0[NT], e1[NT];
        TE pp[NT][NR];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          e0[nt] = acce_t{0, 0, 0, 0};
          e1[nt] = acce_t{Ee[nt][0], NR > 1 ? Ee[nt][NR > 1 ? 1 : 0] : (TE)0, NR > 2 ? Ee[nt][NR > 2 ? 2 : 0] : (TE)0, 0};
#pragma unroll
          for (int r = 0; r < NR; ++r) pp[nt][r] = (TE)gg[nt][r] + Tt[nt][r];
        }
#pragma unroll
        for (int ks = 0; ks < KE; ++ks) {
          const TE fa0 = ae[(ks * 2 + 0) * 64 + lane], fa1 = ae[(ks * 2 + 1) * 64 + lane];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const TE b = ks < NR ? pp[nt][ks < NR ? ks : 0] : (TE)gg[nt][ks == NR ? 3 : 4];
#ifndef ADMM_MF_ABLATE_ELIM     // timing-only diagnostic: no ELIM MFMAs (wrong results)
            e0[nt] = OpsE::mfma(fa0, b, e0[nt]);
            e1[nt] = OpsE::mfma(fa1, b, e1[nt]);
#else
            e0[nt][0] += fa0 * b; e1[nt][0] += fa1 * b;
#endif
          }
        }
        const unsigned d0 = (unsigned)(k - k0) * NU * PB;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
          for (int r = 0; r < NR; ++r) { Tt[nt][r] = e0[nt][r]; Ee[nt][r] = e1[nt][r]; }
          vd.store((double)e0[nt][3], oku0 ? lbs[nt] : ROWVIEW_OOB, d0);
          if (XT) vd.store((double)e1[nt][3], oku1 ? lbs[nt] : ROWVIEW_OOB, d0 + 4u * PB);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
     }
    }
    if (nnext) glds_retire();
    __syncthreads();
    buf ^= 1;
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const bool st = wave_active && lbs[nt] != ROWVIEW_OOB;
#pragma unroll
    for (int r = 0; r < NR; ++r)
      if (st && okx[r]) {
        const size_t o = ((size_t)s * NX + 4 * r + g) * P + col[nt];
        tseg[o] = (double)Tt[nt][r];
        eseg[o] = (double)Ee[nt][r];
      }
    if (RESID && SUBST) {
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        const double t = mf_colsum(racc[nt][q]);
        if (st && g == 0) part[((size_t)s * 5 + q) * P + col[nt]] = t;
      }
    }
  }
}

}  // namespace admm

// admm_kernels.hpp -- hand-written HIP kernels (gfx950 / CDNA4) of the ADMM
// inner loop.  DESIGN.md §4 describes the data layout and each kernel's
// roofline.  No reference counterpart exists (README.md:1-2 only).
//
// Device layout ("batch-minor"): every per-QP vector is stored transposed,
//     v[row][col],  row = stacked index e in [0, L),  col = QP b in [0, pitch)
// with pitch = batch rounded up to 64.  One lane owns one QP (x-update) or two
// adjacent QPs (z/dual kernel), so every wave-level access is one contiguous
// 512-B / 1-KiB segment of a row, and all per-stage matrices and bounds are
// wave-uniform: they travel through the scalar unit (s_load -> SGPR operand of
// v_fma_f64) and cost no VGPRs and no LDS bandwidth.
// (Non-template kernels are `static`: this header is included by several translation units.)
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "admm_layout.hpp"

namespace admm {

#ifndef ADMM_XB_THREADS
#define ADMM_XB_THREADS 256
#endif
#ifndef ADMM_XB_PREFETCH
#define ADMM_XB_PREFETCH 0      // 0 = by block size (prefetch_depth below)
#endif
constexpr int XB_THREADS = ADMM_XB_THREADS;   // x-update workgroup: 4 waves = 256 QPs of one segment
constexpr int Z_THREADS = 256;    // z/dual workgroup: 256 lanes x 2 QPs = 512 columns
constexpr int T_TILE = 32;        // transpose tile
// Stages of operand prefetch in xb / xfz (register ring depth).  Measured at n = 6, m = 3:
// 1 -> 4721, 2 -> 4844, 3 -> 4861 iterations/s; larger blocks get a shallower ring (registers).
// (Round 3, operators distributed over the lanes -- dpp_matvec_acc below: depths 1 / 2 / 3 -> 4899 / 5042 / 5091 iterations/s on the
// plain path; depth 3 parks the ring of the q forms in accumulator registers, depth 2 keeps every form of (6, 3) in 256.)
constexpr int prefetch_depth(int nb) {
  return ADMM_XB_PREFETCH > 0 ? ADMM_XB_PREFETCH : (nb <= 12 ? 2 : 1);
}

// The x kernels run at 1-2 waves per SIMD (grid size and the LDS record slab decide that, not
// registers).  Saying so lets the scheduler spend registers on instruction-level parallelism:
// left at the default it minimises register pressure and serialises every LDS operand read
// (`ds_read_b128 v[2:5]; s_waitcnt lgkmcnt(0); use; ...` -- measured 30 % slower on xfz).
#ifndef ADMM_X_MAX_WAVES
#define ADMM_X_MAX_WAVES 2
#endif
#define ADMM_X_OCCUPANCY __attribute__((amdgpu_waves_per_eu(1, ADMM_X_MAX_WAVES)))

// Grid of the sweep kernels: (column block, segment).  Workgroups are dealt to the 8 XCDs round-robin in linear order, and the 16
// column blocks of a segment all read that segment's stage records; with the column block as the fast index (the default) a
// segment's workgroups are spread over all XCDs and every XCD's L2 fetches every segment's records (~14 MB of the 28-32 MB
// read-side excess of a launch at configs[2]).  ADMM_SEGMENT_MAJOR_GRID=1 makes the SEGMENT the fast index -- linear id % 8 =
// segment % 8: two segments per XCD at S = 16, their records fetched once -- the XCD-aware mapping.  Measured round 3, same box,
// two runs each (tools/ab_variants.sh colmaj segmaj): 6766 / 6608 vs 6700 / 6695 batch-iterations/s with residuals, 9930 / 9834 vs
// 9717 / 9644 every 10th, n = 12: 3283 vs 3333 -- inside the run-to-run spread either way (the records are < 2 % of the traffic and
// sit in the infinity cache), so the default stays.
#ifndef ADMM_SEGMENT_MAJOR_GRID
#define ADMM_SEGMENT_MAJOR_GRID 0
#endif
__device__ __forceinline__ int grid_col_block() { return ADMM_SEGMENT_MAJOR_GRID ? blockIdx.y : blockIdx.x; }
__device__ __forceinline__ int grid_segment() { return ADMM_SEGMENT_MAJOR_GRID ? blockIdx.x : blockIdx.y; }
inline dim3 sweep_grid(int col_blocks, int segments) {
  return ADMM_SEGMENT_MAJOR_GRID ? dim3(segments, col_blocks) : dim3(col_blocks, segments);
}

// Stage records (per-stage matrices + box, shared by the batch) are staged into LDS in
// chunks of this many stages: <= 64 KiB per workgroup (two workgroups per CU still fit), a
// multiple of the prefetch depth so that ring slots stay aligned across refills.
constexpr int stage_chunk(int rec_doubles, int pf) {
  int ch = 8192 / rec_doubles;
  if (ch > 128) ch = 128;
  ch = (ch / pf) * pf;          // AFTER the cap: 128 is not a multiple of a 3-deep ring (tiny blocks, segments > 128 stages)
  if (ch < pf) ch = pf;
  return ch;
}

// Cooperative global -> LDS copy of `cnt` doubles (cnt even, both sides 16-byte aligned).
// Four independent 16-byte loads per thread are issued before the first LDS write: a naive
// element loop compiles to load / s_waitcnt vmcnt(0) / ds_write per element, i.e. one exposed
// memory round trip per 2 KiB of records.
template <int THREADS>
__device__ __forceinline__ void stage_records(double* lds, const double* src, int cnt, int tid) {
  const double2* s2 = reinterpret_cast<const double2*>(src);
  double2* d2 = reinterpret_cast<double2*>(lds);
  const int cnt2 = cnt >> 1;
  constexpr int U = 4;
  for (int i0 = 0; i0 < cnt2; i0 += THREADS * U) {
    double2 tmp[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * THREADS + tid;
      tmp[u] = (i < cnt2) ? s2[i] : double2{0.0, 0.0};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * THREADS + tid;
      if (i < cnt2) d2[i] = tmp[u];
    }
  }
}

// Per-stage matrices are wave-uniform; reading them through the constant
// address space makes hipcc select scalar (SMEM) loads for them.
typedef const __attribute__((address_space(4))) double* cdouble_p;
typedef const __attribute__((address_space(4))) int* cint_p;

__device__ __forceinline__ cdouble_p as_const(const double* p) {
  return (cdouble_p)(uintptr_t)p;
}
__device__ __forceinline__ cint_p as_const(const int* p) { return (cint_p)(uintptr_t)p; }

// Rows of a batch-minor array through a buffer resource: the 128-bit descriptor and the row's
// byte offset are wave-uniform (SGPRs), the lane part is ONE 32-bit byte offset shared by every
// access of the kernel -- `buffer_load_dwordx2 v, v_off, s[rsrc], s_row offen`.  No per-access
// 64-bit VALU address arithmetic (it was ~18 % of the VALU instructions of a stage) and no
// per-row address VGPRs.  The descriptor spans one segment of the array (< 2 GiB, checked at
// setup); accesses past it are dropped by the hardware instead of faulting.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// Lane offset of a lane that must not store.  A view spans < 2 GiB (checked at setup), so 2^31 is
// out of range whether or not the hardware adds the scalar row offset before its range check
// (and 2^31 + row offset cannot wrap): the store is discarded.
constexpr unsigned ROWVIEW_OOB = 0x80000000u;

struct RowView {
  __amdgpu_buffer_rsrc_t rsrc;
  __device__ __forceinline__ RowView(const double* base, size_t first_elem, size_t span_bytes)
      : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(base + first_elem), 0,
                                               (int)(unsigned)span_bytes, 0x00020000)) {}
  // AUX = cache policy bits of the buffer instruction (gfx940+: 1 = sc0, 2 = nt, 16 = sc1)
  template <int AUX = 0>
  __device__ __forceinline__ double load(unsigned lane_bytes, unsigned row_bytes) const {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc, lane_bytes, row_bytes, AUX));
  }
  template <int AUX = 0>
  __device__ __forceinline__ void store(double v, unsigned lane_bytes, unsigned row_bytes) const {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rsrc, lane_bytes, row_bytes, AUX);
  }
};

// LEN (even) doubles from a 16-byte-aligned, wave-uniform LDS address, as 16-byte pairs
// (ds_read_b128 broadcasts).
template <int LEN>
__device__ __forceinline__ void lds_block(const double* p, double (&o)[LEN]) {
  static_assert(LEN % 2 == 0, "blocks are padded to even length");
#ifdef ADMM_ABLATE_LDS      // timing-only diagnostic build: no LDS reads, wrong results
#pragma unroll
  for (int c = 0; c < LEN; ++c) o[c] = 0.25 + 0.001 * c;
  (void)p;
#else
  const double2* q = reinterpret_cast<const double2*>(p);
#pragma unroll
  for (int c = 0; c < LEN / 2; ++c) {
    const double2 v = q[c];
    o[2 * c] = v.x;
    o[2 * c + 1] = v.y;
  }
#endif
}

// acc[r] (+|-)= sum_c M[r][c] x[c] with M (ROWS x COLS, row-major, zero-padded to even length)
// at a 16-byte-aligned, wave-uniform LDS address.  The matrix is consumed as a flat stream of
// 16-byte pairs; per row the products are accumulated in column order (the same order as a plain
// row loop).  For large blocks a scheduling barrier every LDS_GROUP_PAIRS pairs stops the
// scheduler from hoisting the whole matrix into registers (n = 12 spilled to scratch without it);
// blocks of up to 2 * LDS_GROUP_PAIRS doubles (all of n = 6, m = 3) are unaffected.
constexpr int LDS_GROUP_PAIRS = 24;

template <int ROWS, int COLS, bool NEG, int G = LDS_GROUP_PAIRS>
__device__ __forceinline__ void lds_matvec_acc(const double* blk, const double (&x)[COLS], double (&acc)[ROWS]) {
  constexpr int PAIRS = even_up(ROWS * COLS) / 2;
#ifndef ADMM_ABLATE_LDS
  const double2* q = reinterpret_cast<const double2*>(blk);
#else
  (void)blk;
#endif
  // Group by group: first every 16-byte read of the group (source order = issue order, so the
  // reads pipeline and the waits are counted), then its FMAs.  Interleaving read/use in the
  // source made the scheduler wait for each read individually.
#pragma unroll
  for (int g0 = 0; g0 < PAIRS; g0 += G) {
    double2 m[G];
#pragma unroll
    for (int c = 0; c < G; ++c)
      if (g0 + c < PAIRS) {
#ifdef ADMM_ABLATE_LDS      // timing-only diagnostic build: no LDS reads, wrong results
        m[c] = double2{0.25 + 0.001 * (g0 + c), 0.125 + 0.002 * (g0 + c)};
#else
        m[c] = q[g0 + c];
#endif
      }
#pragma unroll
    for (int c = 0; c < G; ++c)
      if (g0 + c < PAIRS) {
        const int e0 = 2 * (g0 + c), e1 = e0 + 1;
        if (e0 < ROWS * COLS) acc[e0 / COLS] = fma(NEG ? -m[c].x : m[c].x, x[e0 % COLS], acc[e0 / COLS]);
        if (e1 < ROWS * COLS) acc[e1 / COLS] = fma(NEG ? -m[c].y : m[c].y, x[e1 % COLS], acc[e1 / COLS]);
      }
    if (g0 + G < PAIRS) __builtin_amdgcn_sched_barrier(0);   // large blocks only: bound the hoisting
  }
}

// ---------------------------------------------------------------------------
// Stage operators DISTRIBUTED over the lanes of a row, applied with DPP (round 3; DESIGN.md §4.5 "operand delivery").
//
// lds_matvec_acc above delivers every wave-uniform operator entry to all 64 lanes: one ds_read_b128 writes 1 KiB of
// registers for two FMAs, and the LDS return path (measured: >= 6.8 cycles per broadcast b128 and CU with four waves,
// tools/micro/lds_bcast.hip) -- 120 such reads per stage and wave at n = 6, m = 3 -- is what the fused kernels wait for
// once their HBM bytes are cut (XFREE forms).  gfx90a+ has ONE data-parallel-primitive control for fp64:
//     v_fmac_f64_dpp  acc, op, x  row_newbcast:k      acc += op[lane k of this lane's 16-lane row] * x
// at the full fp64 FMA rate (tools/micro/dpp_fma_rate.hip: 2.35 vs 2.38 ns).  So lane l keeps record entry 16 i + (l & 15)
// in register i -- ONE ds_read_b64 per 16 entries and wave (the four rows read the same words: LDS broadcast within a bank)
// -- and every FMA names its entry by (register, k).  The record of a stage at (6, 3) is 15 registers instead of 120
// 16-byte reads; products and their order are those of lds_matvec_acc (bit-identical results).
// Only v_fmac_f64 and v_mov_b64 take the control, so the box (v_min / v_max operands) stays on the broadcast path.
// ---------------------------------------------------------------------------
#ifndef ADMM_NO_DPP_OPERANDS
#define ADMM_DPP_OPERANDS 1
#else
#define ADMM_DPP_OPERANDS 0
#endif

template <int K, bool NEG>
__device__ __forceinline__ void fmac_row_bcast(double& acc, double op, double x) {
  static_assert(K >= 0 && K < 16, "row_newbcast lane");
  // (not volatile: a pure register-to-register operation the scheduler may move like any FMA; the LDS wait for `op` is
  //  inserted by the compiler, which tracks the registers of inline-asm operands)
  if (NEG) asm("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(op), "v"(x), "n"(K));
  else     asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(op), "v"(x), "n"(K));
}

// Registers [R0, R1) of a stage record: v[i] = rec[16 i + (lane & 15)].  `rec_l16` = the stage's LDS address + (lane & 15).
template <int R0, int R1, int NREG>
__device__ __forceinline__ void dpp_rec_load(const double* rec_l16, double (&v)[NREG]) {
  static_assert(R0 >= 0 && R1 <= NREG, "register range");
#pragma unroll
  for (int i = R0; i < R1; ++i) v[i] = rec_l16[16 * i];
}

// The registers that hold the SIZE entries of one operator at record offset OFF.  Kernels issue these one or two operators
// AHEAD of the operator's FMAs (software pipeline over the operator list of a stage): at most three operators' registers are
// live at a time -- 6-8 registers at (6, 3), ~20 at (12, 6), where the whole record would be 57 -- and the LDS latency
// (~105 cycles) hides under the FMAs of the operators in between.  A register shared by two neighbouring operators is simply
// read twice (same stage, same value).
template <int OFF, int SIZE, int NREG>
__device__ __forceinline__ void dpp_op_load(const double* rec_l16, double (&v)[NREG]) {
  dpp_rec_load<OFF / 16, (OFF + SIZE - 1) / 16 + 1, NREG>(rec_l16, v);
}

namespace detail {
template <int ROWS, int COLS, bool NEG, int OFF, int E, int NREG>
__device__ __forceinline__ void dpp_mv_step(const double (&v)[NREG], const double (&x)[COLS], double (&acc)[ROWS]) {
  if constexpr (E < ROWS * COLS) {
    // column-major over the matrix: consecutive FMAs go to different accumulators (independent chains), while each
    // accumulator still receives its products in column order -- the order of a plain row loop and of lds_matvec_acc
    constexpr int c = E / ROWS, r = E % ROWS, flat = OFF + r * COLS + c;
    static_assert(flat / 16 < NREG, "operator entry beyond the loaded registers");
    fmac_row_bcast<flat % 16, NEG>(acc[r], v[flat / 16], x[c]);
    dpp_mv_step<ROWS, COLS, NEG, OFF, E + 1, NREG>(v, x, acc);
  }
}
}  // namespace detail

// acc[r] (+|-)= sum_c M[r][c] x[c] with M (ROWS x COLS, row-major) at record offset OFF of the distributed record v.
template <int ROWS, int COLS, bool NEG, int OFF, int NREG>
__device__ __forceinline__ void dpp_matvec_acc(const double (&v)[NREG], const double (&x)[COLS], double (&acc)[ROWS]) {
  detail::dpp_mv_step<ROWS, COLS, NEG, OFF, 0, NREG>(v, x, acc);
}

// Thrust-magnitude (second-order-cone) projection factor of one block: the control
// rows u of a stage with a finite bound ub are scaled onto the ball ||u||_2 <= ub,
//     c = ||u|| > ub ? ub / ||u|| : 1,      z_u = c u.
// sqrt and the division are the correctly rounded fp64 forms, accumulated in row order with fma --
// the same operations as the CPU oracle.
template <int NU, int NB>
__device__ __forceinline__ double soc_scale(const double (&vblk)[NB], double ub) {
  double ss = 0.0;
#pragma unroll
  for (int j = 0; j < NU; ++j) ss = fma(vblk[j], vblk[j], ss);
  const double nrm = sqrt(ss);
  return nrm > ub ? ub / nrm : 1.0;
}

// ---------------------------------------------------------------------------
// State compression ("v-form").  After any z-update,
//     z = clip(v, lo, hi),   y = v - z        with  v = w^ + y_old
// so the pair (z, y) is a function of the single vector v.  The steady-state
// kernels therefore keep only v in HBM and rebuild z and y in registers with
// the very operations that produced them -- bit-identical iterates, half the
// state traffic.  Kernels take a VFORM / VIN flag: false = read the separate
// z, y arrays (first iteration after admm_setup / admm_set_state, unfused path).
// ---------------------------------------------------------------------------

// ---------------------------------------------------------------------------
// x-update, backward sweep (segment-local).  One lane = one QP, blockIdx.y =
// segment.  For stages k = b-1 .. a of the segment, with tail t = 0 on entry:
//     (z, y) = VFORM ? (clip(v), v - clip(v)) : loaded
//     g    = q - rho (z - y)                    (block k: g^u (m), g^x (n))
//     p    = g^x + t
//     h    = B_k' p + g^u
//     d0_k = Si_k h                    -> dbuf  (m rows per stage)
//     t    = A_k' p - K_k' h
//     e   += Omega_k d0_k
// and on exit t -> tseg[s], e -> eseg[s] (n rows each).
// HBM per stacked element: VFORM: read v 8 (+8 with q), write d 8 m/(n+m);
// otherwise read z, y 16.
// ---------------------------------------------------------------------------
template <int NX, int NU, bool HASQ, bool VFORM, bool SOC>
__global__ __launch_bounds__(XB_THREADS) ADMM_X_OCCUPANCY void xb_kernel(
    const double* __restrict__ z, const double* __restrict__ y, const double* __restrict__ q,
    const double* __restrict__ recB, const int* __restrict__ seg_start_,
    double* __restrict__ dbuf, double* __restrict__ tseg, double* __restrict__ eseg,
    double rho, int pitch) {
  // VFORM: `z` is the v array and `y` is unused.
  constexpr int NB = NX + NU;
  constexpr RecBLayout LB = rec_b_layout(NX, NU);
  constexpr int RB = LB.SIZE;
  constexpr int PF = prefetch_depth(NB);
  constexpr int CH = stage_chunk(RB, PF);        // stages whose records are staged in LDS at once
  __shared__ __attribute__((aligned(16))) double rec[CH * RB + 16];
  // operators distributed over the lanes of a row and read two operators ahead of their FMAs (dpp_matvec_acc above):
  // BT, SI, AT, KT, OM, then BT, SI of the next (= previous in time) stage
  constexpr int NREG = (LB.LO + 15) / 16;
  double ops[NREG];
  const double* rec16 = rec + (threadIdx.x & 15);
#if ADMM_DPP_OPERANDS
#define ADMM_MV(R_, C_, NEG_, BLK_, X_, ACC_) dpp_matvec_acc<R_, C_, NEG_, LB.BLK_, NREG>(ops, X_, ACC_)
#define ADMM_LD(R_, C_, BLK_, PTR_) dpp_op_load<LB.BLK_, (R_) * (C_), NREG>(PTR_, ops)
#else
#define ADMM_MV(R_, C_, NEG_, BLK_, X_, ACC_) lds_matvec_acc<R_, C_, NEG_>(rb + LB.BLK_, X_, ACC_)
#define ADMM_LD(R_, C_, BLK_, PTR_) (void)(PTR_)
  (void)rec16; (void)ops;
#endif

  // No early return (every wave must reach the barriers) and no store branches: lanes past the
  // pitch are clamped onto the last column for their LOADS (so they compute finite values) and
  // their STORES are dropped (out-of-range buffer offset) -- the stage loop stays branch-free.
  const int col_raw = grid_col_block() * XB_THREADS + threadIdx.x;
  const int col = col_raw < pitch ? col_raw : pitch - 1;
#ifdef ADMM_STORE_PRED      // A/B diagnostic: the older predicated-store form
  const bool st = col_raw < pitch;
#else
  constexpr bool st = true;
#endif
  // Stores of clamped lanes go to an out-of-range buffer offset and are dropped by the hardware
  // (see RowView): still branch-free, and no lane ever writes a column it does not own.
  const unsigned lb_st = col_raw < pitch ? (unsigned)col * 8u : ROWVIEW_OOB;
  const int s = grid_segment();
  cint_p seg_start = as_const(seg_start_);
  const int k0 = seg_start[s], k1 = seg_start[s + 1];
  const size_t P = (size_t)pitch;
  const unsigned lb = (unsigned)col * 8u;          // this lane's byte offset inside a row
  const unsigned PB = (unsigned)pitch * 8u;        // bytes per row
  // views of this segment's rows (row offsets below are relative to stage k0)
  const RowView vz(z, (size_t)k0 * NB * P, (size_t)(k1 - k0) * NB * P * 8);
  const RowView vy(VFORM ? z : y, (size_t)k0 * NB * P, (size_t)(k1 - k0) * NB * P * 8);
  const RowView vq(HASQ ? q : z, (size_t)k0 * NB * P, (size_t)(k1 - k0) * NB * P * 8);
  const RowView vd(dbuf, (size_t)k0 * NU * P, (size_t)(k1 - k0) * NU * P * 8);

  double t[NX], e[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) { t[i] = 0.0; e[i] = 0.0; }

  // Operand prefetch ring, PF stages deep (slot = stage offset inside the unrolled group, so
  // slots are compile-time registers).
  double lz[PF][NB], ly[PF][NB], lq[PF][NB];
#pragma unroll
  for (int j = 0; j < PF; ++j) {
    const int kj = (k1 - 1 - j > k0) ? k1 - 1 - j : k0;
    const unsigned r0 = (unsigned)(kj - k0) * NB * PB;
#pragma unroll
    for (int r = 0; r < NB; ++r) {
      lz[j][r] = vz.load(lb, r0 + r * PB);
      if (!VFORM) ly[j][r] = vy.load(lb, r0 + r * PB);
      if (HASQ) lq[j][r] = vq.load(lb, r0 + r * PB);
    }
  }

  for (int kc = k1 - 1; kc >= k0; kc -= CH) {      // LDS refill: stages kc, kc-1, ..., klo
    const int klo = (kc - CH + 1 > k0) ? kc - CH + 1 : k0;
    __syncthreads();
    stage_records<XB_THREADS>(rec, recB + (size_t)klo * RB, (kc - klo + 1) * RB, threadIdx.x);
    __syncthreads();
    for (int kb = kc; kb >= klo; kb -= PF) {
#pragma unroll
      for (int j = 0; j < PF; ++j) {
        const int k = kb - j;
        if (k < klo) break;
        [[maybe_unused]] const double* rb = rec + (k - klo) * RB;     // wave-uniform address: LDS broadcast reads
        const double* rb16 = rec16 + (k - klo) * RB;
        ADMM_LD(NU, NX, BT, rb16);                   // (the z-update below covers their latency)
        ADMM_LD(NU, NU, SI, rb16);
        double mLO[even_up(NB)], mHI[even_up(NB)];
        double cs = 1.0;
        if (VFORM) {
          lds_block(rb + LB.LO, mLO);
          lds_block(rb + LB.HI, mHI);
          // compiled only for problems with a thrust-magnitude bound.  Branch-free: ub = +inf (no bound at this
          // stage) gives cs = 1 exactly, and where ub is finite the control rows' box is (-inf, inf), so
          // clip(cs v_u) is the ball projection there and the box projection elsewhere (see xfze_kernel)
          if (SOC) cs = soc_scale<NU, NB>(lz[j], rb[LB.UB]);
        }
        double g[NB];
#pragma unroll
        for (int r = 0; r < NB; ++r) {
          double zz = lz[j][r], yy;
          if (VFORM) {
            zz = fmin(fmax((SOC && r < NU) ? lz[j][r] * cs : lz[j][r], mLO[r]), mHI[r]);
            yy = lz[j][r] - zz;
          } else {
            yy = ly[j][r];
          }
          g[r] = -rho * (zz - yy);
          if (HASQ) g[r] += lq[j][r];
        }
        {  // refill this slot with stage k - PF (clamped: a harmless re-read near the segment start)
          const int kn = (k - PF > k0) ? k - PF : k0;
          const unsigned r0 = (unsigned)(kn - k0) * NB * PB;
#pragma unroll
          for (int r = 0; r < NB; ++r) {
#ifdef ADMM_ABLATE_GLOBAL   // timing-only diagnostic build: no operand loads in the stage loop
            lz[j][r] = lz[j][r] * 0.5 + (double)r0;
#else
            lz[j][r] = vz.load(lb, r0 + r * PB);
#endif
            if (!VFORM) ly[j][r] = vy.load(lb, r0 + r * PB);
            if (HASQ) lq[j][r] = vq.load(lb, r0 + r * PB);
          }
        }
        double p[NX], h[NU], d[NU];
#pragma unroll
        for (int i = 0; i < NX; ++i) p[i] = g[NU + i] + t[i];
#pragma unroll
        for (int jj = 0; jj < NU; ++jj) h[jj] = g[jj];
        __builtin_amdgcn_sched_barrier(0);                        // region boundary (see xfz_kernel)
        ADMM_LD(NX, NX, AT, rb16);
        ADMM_MV(NU, NX, false, BT, p, h);          // h = g^u + B' p
#pragma unroll
        for (int jj = 0; jj < NU; ++jj) d[jj] = 0.0;
        ADMM_LD(NX, NU, KT, rb16);
        ADMM_MV(NU, NU, false, SI, h, d);          // d = Si h
        if (st) {
          const unsigned d0 = (unsigned)(k - k0) * NU * PB;
#pragma unroll
          for (int jj = 0; jj < NU; ++jj) vd.store(d[jj], lb_st, d0 + jj * PB);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NX; ++i) t[i] = 0.0;
        ADMM_LD(NX, NU, OM, rb16);
        ADMM_MV(NX, NX, false, AT, p, t);          // t = A' p
        ADMM_MV(NX, NU, true, KT, h, t);           //       - K' h
        ADMM_MV(NX, NU, false, OM, d, e);          // e += Omega d
        // keep the unrolled stages apart: without this the scheduler hoists the next stage's
        // LDS matrix reads across the boundary and the register file overflows into AGPR moves
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
  }
}

#undef ADMM_MV
#undef ADMM_LD

// ---------------------------------------------------------------------------
// Segment scan.  One lane = one QP, all segments, sequential in S (S is small):
//     t_in(S-1) = 0;   t_in(s-1) = tseg(s) + Phi_s t_in(s)
//     e'(s)     = eseg(s) + Xi_s t_in(s)                     (in place in eseg)
//     x_in(0)   = x0;  x_in(s+1) = e'(s) + Th_s x_in(s)
// Touches only 4 S n rows: negligible HBM traffic, but a dependent chain of
// 2 S small mat-vecs run by only pitch/64 waves, so every exposed latency
// counts twice S times.  Two measures keep the chain at ALU latency:
//   - the vector operands of step s (tseg / eseg rows) do not depend on the
//     chain, so they are prefetched D steps ahead into a register ring;
//   - the segment matrices (3 n^2 doubles per segment) are staged into LDS once
//     (SCAN_LDS_DOUBLES at a time) and read back as wave-uniform broadcasts:
//     with one wave per CU a scalar-cache miss per step would not be hidden.
// ---------------------------------------------------------------------------
constexpr int SCAN_LDS_DOUBLES = 6144;   // 48 KiB of segment matrices per refill

template <int NX>
__global__ __launch_bounds__(64) void xscan_kernel(
    const double* __restrict__ tseg, double* __restrict__ eseg, const double* __restrict__ x0,
    const double* __restrict__ recS, double* __restrict__ tin, double* __restrict__ xin,
    int S, int pitch) {
  constexpr int RS = 3 * NX * NX;
  constexpr int D = (NX <= 4) ? 8 : (NX <= 8 ? 6 : 4);     // prefetch depth (register ring)
  constexpr int CS = (SCAN_LDS_DOUBLES / RS / D) * D;      // segments per LDS refill, multiple of D
  static_assert(CS >= D, "segment records do not fit the LDS staging buffer");
  __shared__ double mats[CS * RS];
  const int col = blockIdx.x * 64 + threadIdx.x;           // pitch is a multiple of 64
  const size_t P = (size_t)pitch;
  double t[NX], x[NX];
  double rt[D][NX], re[D][NX];

  // ---- backward chain over s = S-1 .. 0, in LDS refills of CS segments ----
#pragma unroll
  for (int j = 0; j < D; ++j) {
    const int sj = S - 1 - j;
    const size_t o = (size_t)(sj > 0 ? sj : 0) * NX * P + col;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      rt[j][i] = tseg[o + i * P];
      re[j][i] = eseg[o + i * P];
    }
  }
#pragma unroll
  for (int i = 0; i < NX; ++i) t[i] = 0.0;
  for (int hi_s = S - 1; hi_s >= 0; hi_s -= CS) {
    const int lo_s = (hi_s - CS + 1 > 0) ? hi_s - CS + 1 : 0;     // this refill covers [lo_s, hi_s]
    __syncthreads();
    for (int i = threadIdx.x; i < (hi_s - lo_s + 1) * RS; i += 64) mats[i] = recS[(size_t)lo_s * RS + i];
    __syncthreads();
    for (int base = hi_s; base >= lo_s; base -= D) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const int s = base - j;
        if (s >= lo_s) {
          const size_t o = (size_t)s * NX * P + col;
          const double* rs = mats + (s - lo_s) * RS;
          double tn[NX], en[NX];
#pragma unroll
          for (int i = 0; i < NX; ++i) {
            double a = rt[j][i], b = re[j][i];
#pragma unroll
            for (int l = 0; l < NX; ++l) {
              a = fma(rs[i * NX + l], t[l], a);
              b = fma(rs[NX * NX + i * NX + l], t[l], b);
            }
            tn[i] = a;
            en[i] = b;
          }
#pragma unroll
          for (int i = 0; i < NX; ++i) {
            tin[o + i * P] = t[i];       // t_in(s)
            eseg[o + i * P] = en[i];     // e'(s)
            t[i] = tn[i];                // t_out(s) = t_in(s-1)
          }
          const int sn = s - D;          // refill this ring slot
          if (sn >= 0) {
            const size_t on = (size_t)sn * NX * P + col;
#pragma unroll
            for (int i = 0; i < NX; ++i) {
              rt[j][i] = tseg[on + i * P];
              re[j][i] = eseg[on + i * P];
            }
          }
        }
      }
    }
  }

  // ---- forward chain over s = 0 .. S-1 (reads the e' this lane just wrote) ----
#pragma unroll
  for (int j = 0; j < D; ++j) {
    const int sj = (j < S) ? j : S - 1;
    const size_t o = (size_t)sj * NX * P + col;
#pragma unroll
    for (int i = 0; i < NX; ++i) re[j][i] = eseg[o + i * P];
  }
#pragma unroll
  for (int i = 0; i < NX; ++i) x[i] = x0[(size_t)i * P + col];
  for (int lo_s = 0; lo_s < S; lo_s += CS) {
    const int hi_s = (lo_s + CS - 1 < S - 1) ? lo_s + CS - 1 : S - 1;
    if (S > CS) {      // otherwise the one refill of the backward chain still holds every segment
      __syncthreads();
      for (int i = threadIdx.x; i < (hi_s - lo_s + 1) * RS; i += 64) mats[i] = recS[(size_t)lo_s * RS + i];
      __syncthreads();
    }
    for (int base = lo_s; base <= hi_s; base += D) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const int s = base + j;
        if (s <= hi_s) {
          const size_t o = (size_t)s * NX * P + col;
          const double* rs = mats + (s - lo_s) * RS + 2 * NX * NX;
          double xn[NX];
#pragma unroll
          for (int i = 0; i < NX; ++i) {
            double a = re[j][i];
#pragma unroll
            for (int l = 0; l < NX; ++l) a = fma(rs[i * NX + l], x[l], a);
            xn[i] = a;
          }
#pragma unroll
          for (int i = 0; i < NX; ++i) {
            xin[o + i * P] = x[i];       // x_in(s)
            x[i] = xn[i];                // x_in(s+1)
          }
          const int sn = s + D;
          if (sn < S) {
            const size_t on = (size_t)sn * NX * P + col;
#pragma unroll
            for (int i = 0; i < NX; ++i) re[j][i] = eseg[on + i * P];
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Residual finalise + stopping rule (device body shared by resid_finalize_kernel and by the
// finalise role of xscan_mfma_kernel).  A workgroup = 64 QPs x GROUPS chunk groups: each lane sums
// its group's chunk partials, the groups are combined through LDS in a fixed order (bitwise
// reproducible), then one wave takes square roots, applies
//   r <= sqrt(L) eps_abs + eps_rel max(|w|, |z|),  s <= sqrt(L) eps_abs + eps_rel rho |y|
// records the first iteration at which the QP met it, and counts converged QPs of the real batch
// (wave-shuffle reduction, one atomic per wave).
// ---------------------------------------------------------------------------
constexpr int FIN_COLS = 64;     // columns (QPs) per finalise workgroup: one wave wide
constexpr int FIN_GROUPS = 16;   // chunk groups per workgroup of resid_finalize_kernel (one wave each)

struct FinArgs {
  const double* part;
  double* resid;
  int *status, *iters, *nconv;
  double rho, eps_abs, eps_rel, sqrtL;
  int nchunks, batch, it;
  const double* rhov;     // per-QP rho [pitch] (per-instance dynamics), or NULL: rho above for every QP
};

template <int GROUPS>
__device__ __forceinline__ void finalize_body(double (*red)[5][FIN_COLS], const FinArgs& fa, int block, int pitch) {
  const int lane = threadIdx.x & (FIN_COLS - 1), g = threadIdx.x / FIN_COLS;
  const int col = block * FIN_COLS + lane;      // pitch is a multiple of 64
  const size_t P = (size_t)pitch;
  double rho = fa.rho;
  if (fa.rhov != nullptr && g == 0) rho = fa.rhov[col];
  double a[5] = {0, 0, 0, 0, 0};
  for (int c = g; c < fa.nchunks; c += GROUPS) {
    const size_t o = (size_t)c * 5 * P + col;
#pragma unroll
    for (int v = 0; v < 5; ++v) a[v] += fa.part[o + v * P];
  }
#pragma unroll
  for (int v = 0; v < 5; ++v) red[g][v][lane] = a[v];
  __syncthreads();
  if (g != 0) return;
  // fixed combination order (group 0, 1, ...): bitwise reproducible run to run
#pragma unroll
  for (int v = 0; v < 5; ++v) {
    double t = red[0][v][lane];
    for (int gg = 1; gg < GROUPS; ++gg) t += red[gg][v][lane];
    a[v] = t;
  }
  const double r = sqrt(a[0]), s = rho * sqrt(a[1]);
  const double nw = sqrt(a[2]), nz = sqrt(a[3]), ny = rho * sqrt(a[4]);
  fa.resid[0 * P + col] = r;
  fa.resid[1 * P + col] = s;
  fa.resid[2 * P + col] = nw;
  fa.resid[3 * P + col] = nz;
  fa.resid[4 * P + col] = ny;
  if (fa.it > 0) {
    int ok_now = 0;
    if (col < fa.batch) {
      const double e_pri = fa.sqrtL * fa.eps_abs + fa.eps_rel * fmax(nw, nz);
      const double e_dua = fa.sqrtL * fa.eps_abs + fa.eps_rel * ny;
      int st = fa.status[col];
      if (!st && r <= e_pri && s <= e_dua) {
        st = 1;
        fa.status[col] = 1;
        fa.iters[col] = fa.it;
      }
      ok_now = st;
    }
    // wave64 shuffle reduction of the converged count, then one atomic per wave
    int v = ok_now;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0 && v) atomicAdd(fa.nconv, v);
  }
}

// ---------------------------------------------------------------------------
// Segment scan as a dense fp64 GEMM on the matrix cores (the default scan).
// Both chains of xscan_kernel are linear with batch-shared matrices, so their
// composition is one constant matrix W (built on the host, admm_factor.cpp):
//     [t_in(0..S-1); x_in(0..S-1)] = W [tseg(0..S-1); x0; eseg(0..S-1)]
// i.e. out[M][pitch] = W[M][K] * in[K][pitch] with M = 2 S n (padded), K = (2S+1) n
// (padded) -- for S = 32, n = 6: 384 x 392 times a 392 x 4096 panel.  The
// batch-minor layout IS the row-major B operand.  This trades 2S dependent
// latency-bound steps for one pass of v_mfma_f64_16x16x4_f64.
//   wave  = one 16-column N-tile x SCAN_MT M-tiles (4 accumulators: covers the MFMA
//           dependent-issue latency); workgroup = 4 waves = 4 N-tiles of one M-group
//   A     = W pre-packed on the host in fragment order (coalesced, L2-resident: 1.4 MB),
//           fetched once per workgroup into a double-buffered LDS slab;
//           B = in[4 ks + (lane>>4)][n0 + (lane&15)], straight to registers
//   sparsity: per M-group the [k_begin, k_end) step range outside which its rows of
//           W vanish (block-triangular structure) comes from the host; ~50 % skipped.
// f64 MFMA operand maps (cdna_hip_programming.md §3): A[i = lane&15][k = lane>>4],
// B[k = lane>>4][j = lane&15], D: col = lane&15, row = (lane>>4) + 4 reg.
// ---------------------------------------------------------------------------
typedef double mfma_d4 __attribute__((ext_vector_type(4)));

#ifndef ADMM_SCAN_U
#define ADMM_SCAN_U 8
#endif
constexpr int SCAN_U = ADMM_SCAN_U;   // k-steps per batch; the host pads K and the ranges to multiples of it

// Finalise role: launched with gridDim.y = ngroups + 1, the extra row of workgroups (64 QPs each,
// 4 chunk groups) finalises the residual partials of the iteration that has just finished (fa) while
// the others run the scan -- the two are independent and both too small to fill the chip, so the
// 6 us resid_finalize_kernel launch disappears from every checked iteration but the last of a call.
template <int MT>
__global__ __launch_bounds__(256) void xscan_mfma_kernel(
    const double* __restrict__ Wp, const double* __restrict__ in, double* __restrict__ out,
    const int* __restrict__ krange_, int mtiles, int ngroups, int pitch, int nsplit, size_t split_stride, FinArgs fa) {
  // Workgroup = 4 waves = 4 adjacent N-tiles (64 columns) of ONE M-group, so the A
  // fragments (the same for every N-tile) are fetched from L2 once per workgroup and
  // shared through a double-buffered LDS slab; each wave loads its own B fragments.
  constexpr int U = SCAN_U;
  constexpr int SLAB = U * MT * 64;                  // doubles of A per batch (16 KiB at MT = 4)
  __shared__ __attribute__((aligned(16))) double abuf[2][SLAB];
  static_assert(2 * SLAB >= 4 * 5 * FIN_COLS, "finalise role reuses the A slab");
  if ((int)blockIdx.y >= ngroups) {                  // finalise role (workgroup-uniform)
    if (blockIdx.z == 0) finalize_body<4>(reinterpret_cast<double(*)[5][FIN_COLS]>(&abuf[0][0]), fa, blockIdx.x, pitch);
    return;
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int group = ngroups - 1 - (int)blockIdx.y;   // long k-ranges (x_in rows) are dispatched first
  cint_p krange = as_const(krange_);
  int kb = krange[2 * group], ke = krange[2 * group + 1];          // multiples of U
  // Split-K (blockIdx.z): for small batches the grid has few workgroups and each would walk a
  // long K range alone; the range is cut into nsplit slices whose partial results go to
  // separate output slabs (the consumers add them in split order: deterministic).
  if (nsplit > 1) {
    const int per = (((ke - kb) / U + nsplit - 1) / nsplit) * U;
    const int sb = kb + (int)blockIdx.z * per;
    ke = (sb + per < ke) ? sb + per : ke;
    kb = sb;
    out += (size_t)blockIdx.z * split_stride;
  }
  const size_t P = (size_t)pitch;
  const int n0 = (blockIdx.x * 4 + wave) * 16;       // pitch is a multiple of 64
  const double* bptr = in + (size_t)(lane >> 4) * P + n0 + (lane & 15);
  const double* aptr = Wp + (size_t)group * MT * 64 + tid;         // this thread's element of each k-step's slab row
  const size_t astep = (size_t)mtiles * 64, bstep = 4 * P;
  mfma_d4 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = mfma_d4{0.0, 0.0, 0.0, 0.0};
  if (kb < ke) {
    double ar[U], bc[U], bn[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ar[u] = aptr[(size_t)(kb + u) * astep];
      bc[u] = bptr[(size_t)(kb + u) * bstep];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) abuf[0][u * MT * 64 + tid] = ar[u];
    __syncthreads();
    int buf = 0;
    for (int ks = kb; ks < ke; ks += U) {
      const bool has_next = ks + U < ke;
      if (has_next) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          ar[u] = aptr[(size_t)(ks + U + u) * astep];
          bn[u] = bptr[(size_t)(ks + U + u) * bstep];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int t = 0; t < MT; ++t)
          acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(abuf[buf][(u * MT + t) * 64 + lane], bc[u], acc[t], 0, 0, 0);
      }
      if (has_next) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          abuf[buf ^ 1][u * MT * 64 + tid] = ar[u];
          bc[u] = bn[u];
        }
      }
      __syncthreads();
      buf ^= 1;
    }
  }
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int row0 = (group * MT + t) * 16 + (lane >> 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(size_t)(row0 + 4 * r) * P + n0 + (lane & 15)] = acc[t][r];
  }
}

// ---------------------------------------------------------------------------
// Segment scan for a handful of QPs (batch <= SCAN_GEMV_MAXCOLS): the same product out = W in as a matrix-VECTOR
// product per column.  With one QP the MFMA form above wastes 15 of its 16 panel columns and walks each M-group's k range
// in dependent load -> LDS -> MFMA rounds (10.5 us at S = 64 even with split-K: 40 % of a one-QP iteration; this form:
// configs[1] 26.0 -> 22.2 us per iteration with residuals, 25.0 -> 18.8 us without).  Here one wave
// owns one output row: its lanes stride over the row's non-zero k range (W dense row-major, coalesced; the range comes
// from the host), every load of the row is issued before the first use, and the 64 partial sums are combined with a
// fixed xor-shuffle tree (bitwise reproducible).  The matrix (2.4 MB non-zero at S = 64, n = 6) stays in the infinity cache.
// blockIdx.y == 1: the finalise role, as in xscan_mfma_kernel.
// ---------------------------------------------------------------------------
constexpr int SCAN_GEMV_MAXCOLS = 4;     // measured: 8 columns run as fast on the MFMA form (11.9 vs 10.7 us)

template <int NC>
__global__ __launch_bounds__(256) void xscan_gemv_kernel(
    const double* __restrict__ W, const int* __restrict__ rowrange_, const double* __restrict__ in, double* __restrict__ out,
    int M, int K, int pitch, FinArgs fa) {
  __shared__ double red[4][5][FIN_COLS];
  if (blockIdx.y == 1) {                                // finalise role (workgroup-uniform): one workgroup per 64 QPs
    if ((int)blockIdx.x * FIN_COLS < pitch) finalize_body<4>(red, fa, blockIdx.x, pitch);
    return;
  }
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (row >= M) return;
  cint_p rowrange = as_const(rowrange_);
  const int kb = rowrange[2 * row], ke = rowrange[2 * row + 1];
  const size_t P = (size_t)pitch;
  const double* wr = W + (size_t)row * K;
  double acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = 0.0;
  constexpr int UN = 4;                                 // 4 x 64 = 256 columns of W per round, all loads in flight together
  for (int k0 = kb; k0 < ke; k0 += 64 * UN) {
    double wv[UN], xv[UN][NC];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int k = k0 + 64 * u + lane;
      const bool ok = k < ke;
      const int kc = ok ? k : kb;                       // clamped: loads only
      wv[u] = ok ? wr[kc] : 0.0;
#pragma unroll
      for (int c = 0; c < NC; ++c) xv[u][c] = in[(size_t)kc * P + c];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u)
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[c] = fma(wv[u], xv[u][c], acc[c]);
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    double a = acc[c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
    if (lane == 0) out[(size_t)row * P + c] = a;
  }
}

// ---------------------------------------------------------------------------
// x-update, forward rollout (segment-local, exact once t_in/x_in are known).
//     d   = d0_k + Psi_k t_in
//     u   = -K_k x - d
//     x   = A_k x + B_k u
//     w block k = (u, x)
// HBM per stacked element: reads d (8 m/(n+m)), writes w (8 B).
// ---------------------------------------------------------------------------
template <int NX, int NU>
__global__ __launch_bounds__(XB_THREADS) ADMM_X_OCCUPANCY void xf_kernel(
    const double* __restrict__ dbuf, const double* __restrict__ tin, const double* __restrict__ xin,
    const double* __restrict__ recF, const int* __restrict__ seg_start_,
    double* __restrict__ w, int pitch, int nsplit, size_t split_stride) {
  // Read-out (admm_get's w, admm_step_x, the unfused path): the rollout of xfz_kernel on the same LDS-staged records
  // and buffer addressing, storing w instead of updating v.  (Round 1's version read every operand through the
  // scalar unit: 165 us at configs[2] against xfz's 112 us for MORE traffic.)
  constexpr int NB = NX + NU;
  constexpr RecFLayout LF = rec_f_layout(NX, NU);
  constexpr int RF = LF.SIZE;
  constexpr int PF = prefetch_depth(NB);
  constexpr int CH = stage_chunk(RF, PF);
  __shared__ __attribute__((aligned(16))) double rec[CH * RF + 16];
  // operators distributed over the lanes of a row, read ahead of their FMAs (dpp_matvec_acc above): PSI, K, A, B
  constexpr int NREG = (LF.LO + 15) / 16;
  double ops[NREG];
  const double* rec16 = rec + (threadIdx.x & 15);
#if ADMM_DPP_OPERANDS
#define ADMM_MV(R_, C_, NEG_, BLK_, X_, ACC_) dpp_matvec_acc<R_, C_, NEG_, LF.BLK_, NREG>(ops, X_, ACC_)
#define ADMM_LD(R_, C_, BLK_, PTR_) dpp_op_load<LF.BLK_, (R_) * (C_), NREG>(PTR_, ops)
#else
#define ADMM_MV(R_, C_, NEG_, BLK_, X_, ACC_) lds_matvec_acc<R_, C_, NEG_>(rf + LF.BLK_, X_, ACC_)
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
  const RowView vw(w, (size_t)k0 * NB * P, (size_t)(k1 - k0) * NB * P * 8);
  const RowView vd(dbuf, (size_t)k0 * NU * P, (size_t)(k1 - k0) * NU * P * 8);
  double t[NX], x[NX];
  {
    const size_t o = (size_t)s * NX * P + col;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      // the scan may have been split over K (small batches, <= 8 slabs): partial sums, added in split order
      double tp[8], xp[8];
#pragma unroll
      for (int sp = 0; sp < 8; ++sp) {
        const size_t so = (size_t)(sp < nsplit ? sp : 0) * split_stride + o + i * P;
        tp[sp] = tin[so];
        xp[sp] = xin[so];
      }
      double ta = tp[0], xa = xp[0];
#pragma unroll
      for (int sp = 1; sp < 8; ++sp) {
        ta += (sp < nsplit) ? tp[sp] : 0.0;
        xa += (sp < nsplit) ? xp[sp] : 0.0;
      }
      t[i] = ta;
      x[i] = xa;
    }
  }
  double ld[PF][NU];
#pragma unroll
  for (int j = 0; j < PF; ++j) {
    const int kj = (k0 + j < k1) ? k0 + j : k1 - 1;
    const unsigned d0 = (unsigned)(kj - k0) * NU * PB;
#pragma unroll
    for (int jj = 0; jj < NU; ++jj) ld[j][jj] = vd.load(lb, d0 + jj * PB);
  }
  for (int kc = k0; kc < k1; kc += CH) {
    const int khi = (kc + CH - 1 < k1 - 1) ? kc + CH - 1 : k1 - 1;
    __syncthreads();
    stage_records<XB_THREADS>(rec, recF + (size_t)kc * RF, (khi - kc + 1) * RF, threadIdx.x);
    __syncthreads();
    ADMM_LD(NU, NX, PSI, rec16);
    for (int kb = kc; kb <= khi; kb += PF) {
#pragma unroll
      for (int j = 0; j < PF; ++j) {
        const int k = kb + j;
        if (k > khi) break;
        [[maybe_unused]] const double* rf = rec + (k - kc) * RF;
        const double* rf16 = rec16 + (k - kc) * RF;
        ADMM_LD(NU, NX, K, rf16);
        ADMM_LD(NX, NX, A, rf16);
        double uacc[NU], xn[NX], uu[NU];
#pragma unroll
        for (int jj = 0; jj < NU; ++jj) uacc[jj] = ld[j][jj];
        {
          const int kn = (k + PF < k1) ? k + PF : k1 - 1;
          const unsigned d0 = (unsigned)(kn - k0) * NU * PB;
#pragma unroll
          for (int jj = 0; jj < NU; ++jj) ld[j][jj] = vd.load(lb, d0 + jj * PB);
        }
        ADMM_MV(NU, NX, false, PSI, t, uacc);    // d0 + Psi t_in
        ADMM_LD(NX, NU, B, rf16);
        ADMM_MV(NU, NX, false, K, x, uacc);      //      + K x
#pragma unroll
        for (int jj = 0; jj < NU; ++jj) uu[jj] = -uacc[jj];
#pragma unroll
        for (int i = 0; i < NX; ++i) xn[i] = 0.0;
        ADMM_MV(NX, NX, false, A, x, xn);        // x+ = A x
        ADMM_MV(NX, NU, false, B, uu, xn);       //      + B u
        __builtin_amdgcn_sched_barrier(0);
        ADMM_LD(NU, NX, PSI, rec16 + ((k < khi ? k + 1 : khi) - kc) * RF);      // the next stage's first operator (after A, B: they may share a register)
        const unsigned r0 = (unsigned)(k - k0) * NB * PB;
#pragma unroll
        for (int jj = 0; jj < NU; ++jj) vw.store(uu[jj], lb_st, r0 + jj * PB);
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          x[i] = xn[i];
          vw.store(xn[i], lb_st, r0 + (NU + i) * PB);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

#undef ADMM_MV
#undef ADMM_LD

// ---------------------------------------------------------------------------
// Fused forward rollout + z-update + dual ascent + residual partials (the
// default iteration path).  Same rollout as xf_kernel, but block k of w is
// consumed in registers instead of being written to HBM and re-read, and the
// result is stored in v-form:
//     (z, y) = VIN ? (clip(v), v - clip(v)) : loaded from the z / y arrays
//     wh = alpha w + (1 - alpha) z       (RELAX only)
//     v+ = wh + y                                               -> v (in place)
//     RESID: z+ = clip(v+), y+ = v+ - z+ and per-QP sums over the segment's rows
//            of (w - z+)^2, (z+ - z)^2, w^2, z+^2, y+^2  -> part[segment][5][pitch]
// Neither w nor z+ / y+ is stored: nothing in the next iteration reads w, and
// z+, y+ are functions of v+ (admm_get re-materialises all three on demand).
// Algorithmic HBM bytes per stacked element, fp64, steady state (VIN):
//     d read 8 m/(n+m) + v read 8 + v+ written 8        (= 18.67 for n = 6, m = 3)
// ---------------------------------------------------------------------------
template <int NX, int NU, bool RESID, bool RELAX, bool VIN, bool SOC>
__global__ __launch_bounds__(XB_THREADS) ADMM_X_OCCUPANCY void xfz_kernel(
    const double* __restrict__ dbuf, const double* __restrict__ tin, const double* __restrict__ xin,
    const double* __restrict__ recF, const int* __restrict__ seg_start_,
    const double* __restrict__ zin, const double* __restrict__ yin, double* __restrict__ v,
    double* __restrict__ part, double alpha, int pitch, int nsplit, size_t split_stride) {
  // VIN: the state is read from v (zin, yin unused); otherwise from zin, yin.  Either
  // way v+ is written to v.  (zin / yin never alias v.)
  constexpr int NB = NX + NU;
  constexpr RecFLayout LF = rec_f_layout(NX, NU);
  constexpr int RF = LF.SIZE;
  constexpr bool NEEDZ = RESID || RELAX;
  constexpr int PF = prefetch_depth(NB);
  constexpr int CH = stage_chunk(RF, PF);
  __shared__ __attribute__((aligned(16))) double rec[CH * RF + 16];
  // operators distributed over the lanes of a row, read ahead of their FMAs (dpp_matvec_acc above): PSI, K, A, B; the next
  // stage's PSI, K under this stage's z-update
  constexpr int NREG = (LF.LO + 15) / 16;
  double ops[NREG];
  const double* rec16 = rec + (threadIdx.x & 15);
#if ADMM_DPP_OPERANDS
#define ADMM_MV(R_, C_, NEG_, BLK_, X_, ACC_) dpp_matvec_acc<R_, C_, NEG_, LF.BLK_, NREG>(ops, X_, ACC_)
#define ADMM_LD(R_, C_, BLK_, PTR_) dpp_op_load<LF.BLK_, (R_) * (C_), NREG>(PTR_, ops)
#else
#define ADMM_MV(R_, C_, NEG_, BLK_, X_, ACC_) lds_matvec_acc<R_, C_, NEG_>(rf + LF.BLK_, X_, ACC_)
#define ADMM_LD(R_, C_, BLK_, PTR_) (void)(PTR_)
  (void)rec16; (void)ops;
#endif

  const int col_raw = grid_col_block() * XB_THREADS + threadIdx.x;
  const int col = col_raw < pitch ? col_raw : pitch - 1;      // clamped lanes: loads only (see xb_kernel)
#ifdef ADMM_STORE_PRED
  const bool st = col_raw < pitch;
#else
  constexpr bool st = true;
#endif
  // v is updated IN PLACE, so a clamped lane must never store: a duplicate of the last column
  // running ahead would overwrite v rows its owner has not read yet.  Out-of-range offset = dropped.
  const unsigned lb_st = col_raw < pitch ? (unsigned)col * 8u : ROWVIEW_OOB;
  const int s = grid_segment();
  cint_p seg_start = as_const(seg_start_);
  const int k0 = seg_start[s], k1 = seg_start[s + 1];
  const size_t P = (size_t)pitch;
  const unsigned lb = (unsigned)col * 8u;          // this lane's byte offset inside a row
  const unsigned PB = (unsigned)pitch * 8u;        // bytes per row
  // views of this segment's rows (row offsets below are relative to stage k0)
  const RowView vv(v, (size_t)k0 * NB * P, (size_t)(k1 - k0) * NB * P * 8);
  const RowView vy(VIN ? v : yin, (size_t)k0 * NB * P, (size_t)(k1 - k0) * NB * P * 8);
  const RowView vz(VIN ? v : zin, (size_t)k0 * NB * P, (size_t)(k1 - k0) * NB * P * 8);
  const RowView vd(dbuf, (size_t)k0 * NU * P, (size_t)(k1 - k0) * NU * P * 8);
  double t[NX], x[NX];
  {
    const size_t o = (size_t)s * NX * P + col;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      // the scan may have been split over K (small batches, <= 8 slabs): partial sums, added in
      // split order.  Unrolled so that every slab's loads are in flight together.
      double tp[8], xp[8];
#pragma unroll
      for (int sp = 0; sp < 8; ++sp) {
        const size_t so = (size_t)(sp < nsplit ? sp : 0) * split_stride + o + i * P;
        tp[sp] = tin[so];
        xp[sp] = xin[so];
      }
      double ta = tp[0], xa = xp[0];
#pragma unroll
      for (int sp = 1; sp < 8; ++sp) {
        ta += (sp < nsplit) ? tp[sp] : 0.0;
        xa += (sp < nsplit) ? xp[sp] : 0.0;
      }
      t[i] = ta;
      x[i] = xa;
    }
  }
  // prefetch ring (see xb_kernel): l0 = v (VIN) or y; l1 = z (only !VIN && NEEDZ)
  double ld[PF][NU], l0[PF][NB], l1[PF][NB];
#pragma unroll
  for (int j = 0; j < PF; ++j) {
    const int kj = (k0 + j < k1) ? k0 + j : k1 - 1;
    const unsigned d0 = (unsigned)(kj - k0) * NU * PB;
#pragma unroll
    for (int jj = 0; jj < NU; ++jj) ld[j][jj] = vd.load(lb, d0 + jj * PB);
    const unsigned r0 = (unsigned)(kj - k0) * NB * PB;
#pragma unroll
    for (int r = 0; r < NB; ++r) {
      if (VIN) {
        l0[j][r] = vv.load(lb, r0 + r * PB);
      } else {
        l0[j][r] = vy.load(lb, r0 + r * PB);
        if (NEEDZ) l1[j][r] = vz.load(lb, r0 + r * PB);
      }
    }
  }
  double a_r = 0, a_s = 0, a_w = 0, a_z = 0, a_y = 0;
  for (int kc = k0; kc < k1; kc += CH) {           // LDS refill: stages kc .. khi
    const int khi = (kc + CH - 1 < k1 - 1) ? kc + CH - 1 : k1 - 1;
    __syncthreads();
    stage_records<XB_THREADS>(rec, recF + (size_t)kc * RF, (khi - kc + 1) * RF, threadIdx.x);
    __syncthreads();
    ADMM_LD(NU, NX, PSI, rec16);
    ADMM_LD(NU, NX, K, rec16);
    for (int kb = kc; kb <= khi; kb += PF) {
#pragma unroll
      for (int j = 0; j < PF; ++j) {
        const int k = kb + j;
        if (k > khi) break;
        [[maybe_unused]] const double* rf = rec + (k - kc) * RF;      // wave-uniform address: LDS broadcast reads
        const double* rf16 = rec16 + (k - kc) * RF;
        double d[NU], c0[NB], c1[NB];
#pragma unroll
        for (int jj = 0; jj < NU; ++jj) d[jj] = ld[j][jj];
#pragma unroll
        for (int r = 0; r < NB; ++r) {
          c0[r] = l0[j][r];
          if (!VIN && NEEDZ) c1[r] = l1[j][r];
        }
        {  // refill this slot with stage k + PF (clamped to the segment; a clamped re-read of
           // rows this lane overwrites later is harmless: same lane, program order, value unused)
          const int kn = (k + PF < k1) ? k + PF : k1 - 1;
          const unsigned d0 = (unsigned)(kn - k0) * NU * PB;
#pragma unroll
          for (int jj = 0; jj < NU; ++jj) ld[j][jj] = vd.load(lb, d0 + jj * PB);
          const unsigned r0 = (unsigned)(kn - k0) * NB * PB;
#pragma unroll
          for (int r = 0; r < NB; ++r) {
            if (VIN) {
#ifdef ADMM_ABLATE_XFZ_LOAD    // timing-only diagnostic: no v loads in the stage loop
              l0[j][r] = l0[j][r] * 0.5 + 0.125;
#else
              l0[j][r] = vv.load(lb, r0 + r * PB);
#endif
            } else {
              l0[j][r] = vy.load(lb, r0 + r * PB);
              if (NEEDZ) l1[j][r] = vz.load(lb, r0 + r * PB);
            }
          }
        }
        double wv[NB];
        {
          double uacc[NU], xn[NX];
#pragma unroll
          for (int jj = 0; jj < NU; ++jj) uacc[jj] = d[jj];
          ADMM_LD(NX, NX, A, rf16);
          ADMM_MV(NU, NX, false, PSI, t, uacc);    // d0 + Psi t_in
          ADMM_LD(NX, NU, B, rf16);
          ADMM_MV(NU, NX, false, K, x, uacc);      //      + K x
#pragma unroll
          for (int jj = 0; jj < NU; ++jj) wv[jj] = -uacc[jj];     // u = -(...)
          double uu[NU];
#pragma unroll
          for (int jj = 0; jj < NU; ++jj) uu[jj] = wv[jj];
#pragma unroll
          for (int i = 0; i < NX; ++i) xn[i] = 0.0;
          ADMM_MV(NX, NX, false, A, x, xn);        // x+ = A x
          ADMM_MV(NX, NU, false, B, uu, xn);       //      + B u
#pragma unroll
          for (int i = 0; i < NX; ++i) wv[NU + i] = xn[i];
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = wv[NU + i];
        // Scheduling-region boundaries: one stage as a single huge region makes the scheduler
        // fall back to its minimum-register order (every LDS read waited for immediately);
        // split into rollout | row | row | ... each region gets a latency-aware schedule.
        __builtin_amdgcn_sched_barrier(0);
        {  // the next stage's first two operators (after A, B: K and A may share a register), under the z-update
          const double* rn16 = rec16 + ((k < khi ? k + 1 : khi) - kc) * RF;
          ADMM_LD(NU, NX, PSI, rn16);
          ADMM_LD(NU, NX, K, rn16);
        }
        const unsigned r0 = (unsigned)(k - k0) * NB * PB;
        // thrust-magnitude bound on this stage's control rows, branch-free: ub = +inf (no bound here) gives both
        // factors = 1 exactly; where ub is finite the control rows' box is (-inf, inf) (see xfze_kernel)
        double cs_old = 1.0, cs_new = 1.0;
        if constexpr (SOC) {
          const double ub = rf[LF.UB];
          if (VIN) cs_old = soc_scale<NU, NB>(c0, ub);
          if (RESID) {                           // z+ needs ||v+_u||: form the control rows of v+ first
            double vnew[NB];
#pragma unroll
            for (int r = 0; r < NB; ++r) {
              if (r < NU) {
                double zo, yo;
                if (VIN) { zo = fmin(fmax(c0[r] * cs_old, rf[LF.LO + r]), rf[LF.HI + r]); yo = c0[r] - zo; }
                else { yo = c0[r]; zo = NEEDZ ? c1[r] : 0.0; }
                double wh = wv[r];
                if (RELAX) wh = fma(alpha, wv[r], (1.0 - alpha) * zo);
                vnew[r] = wh + yo;
              } else {
                vnew[r] = 0.0;
              }
            }
            cs_new = soc_scale<NU, NB>(vnew, ub);
          }
        }
#pragma unroll
        for (int r = 0; r < NB; ++r) {
          const double l = rf[LF.LO + r], h = rf[LF.HI + r];
          const bool ball = SOC && r < NU;
          double zo, yo;                       // state before this z-update
          if (VIN) {
            zo = fmin(fmax(ball ? c0[r] * cs_old : c0[r], l), h);
            yo = c0[r] - zo;
          } else {
            yo = c0[r];
            zo = NEEDZ ? c1[r] : 0.0;
          }
          double wh = wv[r];
          if (RELAX) wh = fma(alpha, wv[r], (1.0 - alpha) * zo);
          const double vn = wh + yo;
#ifdef ADMM_ABLATE_XFZ_STORE   // timing-only diagnostic: no v+ store (keep vn alive)
          asm volatile("" ::"v"(vn));
#else
          if (st) vv.store(vn, lb_st, r0 + r * PB);
#endif
          if (RESID) {
            const double zn = fmin(fmax(ball ? vn * cs_new : vn, l), h);
            const double yn = vn - zn;
            const double dr = wv[r] - zn, ds = zn - zo;
            a_r = fma(dr, dr, a_r);
            a_s = fma(ds, ds, a_s);
            a_w = fma(wv[r], wv[r], a_w);
            a_z = fma(zn, zn, a_z);
            a_y = fma(yn, yn, a_y);
          }
          if (r % 3 == 2) __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);   // see xb_kernel
      }
    }
  }
  if (RESID && col_raw < pitch) {
    const size_t o = (size_t)s * 5 * P + col;
    part[o + 0 * P] = a_r;
    part[o + 1 * P] = a_s;
    part[o + 2 * P] = a_w;
    part[o + 3 * P] = a_z;
    part[o + 4 * P] = a_y;
  }
}

// y *= c over the whole array (rho change: the scaled dual y = lambda / rho is rescaled so that
// the multiplier lambda is unchanged).  16-byte accesses, grid-stride.
static __global__ __launch_bounds__(256) void scale_kernel(double* __restrict__ y, double c, size_t count2) {
  double2* y2 = reinterpret_cast<double2*>(y);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count2; i += (size_t)gridDim.x * 256) {
    double2 v = y2[i];
    v.x *= c;
    v.y *= c;
    y2[i] = v;
  }
}

// v -> (z, y): z = clip(v), y = v - z.  Read-out / mode switches only.  One lane =
// two adjacent QPs, blockIdx.y = row chunk (same geometry as zdual_kernel).
static __global__ __launch_bounds__(Z_THREADS) void v_to_zy_kernel(
    const double* __restrict__ v, double* __restrict__ z, double* __restrict__ y,
    const double* __restrict__ lo_, const double* __restrict__ hi_, int L, int zrows, int pitch) {
  const int col = (blockIdx.x * Z_THREADS + threadIdx.x) * 2;
  if (col >= pitch) return;
  const int r_begin = blockIdx.y * zrows;
  const int r_end = (r_begin + zrows < L) ? r_begin + zrows : L;
  cdouble_p lo = as_const(lo_);
  cdouble_p hi = as_const(hi_);
  constexpr int U = 4;                 // rows in flight per lane (a row-at-a-time loop ran at 3.5 TB/s: 250 us at configs[2])
  int r = r_begin;
  for (; r + U <= r_end; r += U) {
    double2 vv[U];
#pragma unroll
    for (int i = 0; i < U; ++i) vv[i] = *reinterpret_cast<const double2*>(v + (size_t)(r + i) * pitch + col);
#pragma unroll
    for (int i = 0; i < U; ++i) {
      const size_t o = (size_t)(r + i) * pitch + col;
      double2 zz, yy;
      zz.x = fmin(fmax(vv[i].x, lo[r + i]), hi[r + i]);
      zz.y = fmin(fmax(vv[i].y, lo[r + i]), hi[r + i]);
      yy.x = vv[i].x - zz.x;
      yy.y = vv[i].y - zz.y;
      *reinterpret_cast<double2*>(z + o) = zz;
      *reinterpret_cast<double2*>(y + o) = yy;
    }
  }
  for (; r < r_end; ++r) {
    const size_t o = (size_t)r * pitch + col;
    const double2 vv = *reinterpret_cast<const double2*>(v + o);
    double2 zz, yy;
    zz.x = fmin(fmax(vv.x, lo[r]), hi[r]);
    zz.y = fmin(fmax(vv.y, lo[r]), hi[r]);
    yy.x = vv.x - zz.x;
    yy.y = vv.y - zz.y;
    *reinterpret_cast<double2*>(z + o) = zz;
    *reinterpret_cast<double2*>(y + o) = yy;
  }
}

#undef ADMM_MV
#undef ADMM_LD

// ---------------------------------------------------------------------------
// Fused z-update + dual ascent + residual partial sums -- the HBM-bound kernel
// the project is graded on.  One lane = two adjacent QPs (16-B accesses), a
// workgroup = 512 columns x `zrows` rows (blockIdx.y = row chunk).
//     wh = alpha w + (1 - alpha) z        (RELAX only)
//     v  = wh + y;  z+ = min(max(v, lo), hi);  y+ = v - z+       (in place)
//     RESID: per-QP partial sums over the chunk's rows of
//            (w - z+)^2, (z+ - z)^2, w^2, z+^2, y+^2  -> part[chunk][5][pitch]
// The row index is wave-uniform, so lo/hi are scalar loads and the per-QP sums
// need no cross-lane step at all: each lane owns its QPs' accumulators.
// Algorithmic HBM bytes per stacked element: RESID/RELAX: 3 reads + 2 writes =
// 40 B; plain: 2 reads + 2 writes = 32 B.
// ---------------------------------------------------------------------------
template <bool RESID, bool RELAX>
__global__ __launch_bounds__(Z_THREADS) void zdual_kernel(
    const double* __restrict__ w, double* __restrict__ z, double* __restrict__ y,
    const double* __restrict__ lo_, const double* __restrict__ hi_,
    double* __restrict__ part, double alpha, int L, int zrows, int pitch) {
  const int col = (blockIdx.x * Z_THREADS + threadIdx.x) * 2;
  if (col >= pitch) return;
  const int chunk = blockIdx.y;
  const int r_begin = chunk * zrows;
  const int r_end = (r_begin + zrows < L) ? r_begin + zrows : L;
  const size_t P = (size_t)pitch;
  cdouble_p lo = as_const(lo_);
  cdouble_p hi = as_const(hi_);
  double2 a_r = {0, 0}, a_s = {0, 0}, a_w = {0, 0}, a_z = {0, 0}, a_y = {0, 0};
#ifndef ADMM_Z_UNROLL
#define ADMM_Z_UNROLL 4
#endif
  constexpr int U = ADMM_Z_UNROLL;
  int r = r_begin;
  for (; r + U <= r_end; r += U) {
    double2 wv[U], yv[U], zv[U];
#pragma unroll
    for (int i = 0; i < U; ++i) {
      const size_t o = (size_t)(r + i) * P + col;
#ifdef ADMM_Z_NT          // A/B: non-temporal hint on the once-read operand w
      wv[i] = __builtin_nontemporal_load(reinterpret_cast<const double2*>(w + o));
#else
      wv[i] = *reinterpret_cast<const double2*>(w + o);
#endif
      yv[i] = *reinterpret_cast<const double2*>(y + o);
      if (RESID || RELAX) zv[i] = *reinterpret_cast<const double2*>(z + o);
    }
#pragma unroll
    for (int i = 0; i < U; ++i) {
      const size_t o = (size_t)(r + i) * P + col;
      const double l = lo[r + i], h = hi[r + i];
      double2 wh = wv[i];
      if (RELAX) {
        wh.x = fma(alpha, wv[i].x, (1.0 - alpha) * zv[i].x);
        wh.y = fma(alpha, wv[i].y, (1.0 - alpha) * zv[i].y);
      }
      double2 v, zn, yn;
      v.x = wh.x + yv[i].x;
      v.y = wh.y + yv[i].y;
      zn.x = fmin(fmax(v.x, l), h);
      zn.y = fmin(fmax(v.y, l), h);
      yn.x = v.x - zn.x;
      yn.y = v.y - zn.y;
      *reinterpret_cast<double2*>(z + o) = zn;
      *reinterpret_cast<double2*>(y + o) = yn;
      if (RESID) {
        double dx = wv[i].x - zn.x, dy = wv[i].y - zn.y;
        a_r.x = fma(dx, dx, a_r.x); a_r.y = fma(dy, dy, a_r.y);
        dx = zn.x - zv[i].x; dy = zn.y - zv[i].y;
        a_s.x = fma(dx, dx, a_s.x); a_s.y = fma(dy, dy, a_s.y);
        a_w.x = fma(wv[i].x, wv[i].x, a_w.x); a_w.y = fma(wv[i].y, wv[i].y, a_w.y);
        a_z.x = fma(zn.x, zn.x, a_z.x); a_z.y = fma(zn.y, zn.y, a_z.y);
        a_y.x = fma(yn.x, yn.x, a_y.x); a_y.y = fma(yn.y, yn.y, a_y.y);
      }
    }
  }
  for (; r < r_end; ++r) {
    const size_t o = (size_t)r * P + col;
    const double2 wv = *reinterpret_cast<const double2*>(w + o);
    const double2 yv = *reinterpret_cast<const double2*>(y + o);
    double2 zv = {0, 0};
    if (RESID || RELAX) zv = *reinterpret_cast<const double2*>(z + o);
    const double l = lo[r], h = hi[r];
    double2 wh = wv;
    if (RELAX) {
      wh.x = fma(alpha, wv.x, (1.0 - alpha) * zv.x);
      wh.y = fma(alpha, wv.y, (1.0 - alpha) * zv.y);
    }
    double2 v, zn, yn;
    v.x = wh.x + yv.x;
    v.y = wh.y + yv.y;
    zn.x = fmin(fmax(v.x, l), h);
    zn.y = fmin(fmax(v.y, l), h);
    yn.x = v.x - zn.x;
    yn.y = v.y - zn.y;
    *reinterpret_cast<double2*>(z + o) = zn;
    *reinterpret_cast<double2*>(y + o) = yn;
    if (RESID) {
      double dx = wv.x - zn.x, dy = wv.y - zn.y;
      a_r.x = fma(dx, dx, a_r.x); a_r.y = fma(dy, dy, a_r.y);
      dx = zn.x - zv.x; dy = zn.y - zv.y;
      a_s.x = fma(dx, dx, a_s.x); a_s.y = fma(dy, dy, a_s.y);
      a_w.x = fma(wv.x, wv.x, a_w.x); a_w.y = fma(wv.y, wv.y, a_w.y);
      a_z.x = fma(zn.x, zn.x, a_z.x); a_z.y = fma(zn.y, zn.y, a_z.y);
      a_y.x = fma(yn.x, yn.x, a_y.x); a_y.y = fma(yn.y, yn.y, a_y.y);
    }
  }
  if (RESID) {
    const size_t o = (size_t)chunk * 5 * P + col;
    *reinterpret_cast<double2*>(part + o + 0 * P) = a_r;
    *reinterpret_cast<double2*>(part + o + 1 * P) = a_s;
    *reinterpret_cast<double2*>(part + o + 2 * P) = a_w;
    *reinterpret_cast<double2*>(part + o + 3 * P) = a_z;
    *reinterpret_cast<double2*>(part + o + 4 * P) = a_y;
  }
}

// ---------------------------------------------------------------------------
// Block-structured forms of zdual_kernel / v_to_zy_kernel for problems with a thrust-magnitude
// bound: the projection of a stage's control rows needs their joint norm, so a
// chunk is a whole number of blocks (zrows % nb == 0) and each block is visited twice -- once
// over its m control rows for ||v_u|| (the re-read hits L1/L2), once to apply.  Unfused path and
// read-out only; the fused xfz / xb kernels do the same work on register-resident blocks.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double2 soc_scale2(double2 ss, double ub) {
  double2 c;
  const double nx = sqrt(ss.x), ny = sqrt(ss.y);
  c.x = nx > ub ? ub / nx : 1.0;
  c.y = ny > ub ? ub / ny : 1.0;
  return c;
}

template <bool RESID, bool RELAX>
__global__ __launch_bounds__(Z_THREADS) void zdual_soc_kernel(
    const double* __restrict__ w, double* __restrict__ z, double* __restrict__ y,
    const double* __restrict__ lo_, const double* __restrict__ hi_, const double* __restrict__ ub_,
    double* __restrict__ part, double alpha, int L, int zrows, int pitch, int nb, int m) {
  const int col = (blockIdx.x * Z_THREADS + threadIdx.x) * 2;
  if (col >= pitch) return;
  const int chunk = blockIdx.y;
  const int r_begin = chunk * zrows;
  const int r_end = (r_begin + zrows < L) ? r_begin + zrows : L;
  const size_t P = (size_t)pitch;
  cdouble_p lo = as_const(lo_);
  cdouble_p hi = as_const(hi_);
  cdouble_p ubv = as_const(ub_);
  double2 a_r = {0, 0}, a_s = {0, 0}, a_w = {0, 0}, a_z = {0, 0}, a_y = {0, 0};
  for (int rb = r_begin; rb < r_end; rb += nb) {
    const double ub = ubv[rb / nb];
    const bool soc = ub < INFINITY;
    double2 cs = {1.0, 1.0};
    if (soc) {
      double2 ss = {0.0, 0.0};
      for (int j = 0; j < m; ++j) {
        const size_t o = (size_t)(rb + j) * P + col;
        const double2 wv = *reinterpret_cast<const double2*>(w + o);
        const double2 yv = *reinterpret_cast<const double2*>(y + o);
        double2 wh = wv;
        if (RELAX) {
          const double2 zv = *reinterpret_cast<const double2*>(z + o);
          wh.x = fma(alpha, wv.x, (1.0 - alpha) * zv.x);
          wh.y = fma(alpha, wv.y, (1.0 - alpha) * zv.y);
        }
        const double vx = wh.x + yv.x, vy = wh.y + yv.y;
        ss.x = fma(vx, vx, ss.x);
        ss.y = fma(vy, vy, ss.y);
      }
      cs = soc_scale2(ss, ub);
    }
    for (int r = 0; r < nb; ++r) {
      const int row = rb + r;
      const size_t o = (size_t)row * P + col;
      const double2 wv = *reinterpret_cast<const double2*>(w + o);
      const double2 yv = *reinterpret_cast<const double2*>(y + o);
      double2 zv = {0, 0};
      if (RESID || RELAX) zv = *reinterpret_cast<const double2*>(z + o);
      const double l = lo[row], h = hi[row];
      double2 wh = wv;
      if (RELAX) {
        wh.x = fma(alpha, wv.x, (1.0 - alpha) * zv.x);
        wh.y = fma(alpha, wv.y, (1.0 - alpha) * zv.y);
      }
      double2 v, zn, yn;
      v.x = wh.x + yv.x;
      v.y = wh.y + yv.y;
      if (soc && r < m) {
        zn.x = v.x * cs.x;
        zn.y = v.y * cs.y;
      } else {
        zn.x = fmin(fmax(v.x, l), h);
        zn.y = fmin(fmax(v.y, l), h);
      }
      yn.x = v.x - zn.x;
      yn.y = v.y - zn.y;
      *reinterpret_cast<double2*>(z + o) = zn;
      *reinterpret_cast<double2*>(y + o) = yn;
      if (RESID) {
        double dx = wv.x - zn.x, dy = wv.y - zn.y;
        a_r.x = fma(dx, dx, a_r.x); a_r.y = fma(dy, dy, a_r.y);
        dx = zn.x - zv.x; dy = zn.y - zv.y;
        a_s.x = fma(dx, dx, a_s.x); a_s.y = fma(dy, dy, a_s.y);
        a_w.x = fma(wv.x, wv.x, a_w.x); a_w.y = fma(wv.y, wv.y, a_w.y);
        a_z.x = fma(zn.x, zn.x, a_z.x); a_z.y = fma(zn.y, zn.y, a_z.y);
        a_y.x = fma(yn.x, yn.x, a_y.x); a_y.y = fma(yn.y, yn.y, a_y.y);
      }
    }
  }
  if (RESID) {
    const size_t o = (size_t)chunk * 5 * P + col;
    *reinterpret_cast<double2*>(part + o + 0 * P) = a_r;
    *reinterpret_cast<double2*>(part + o + 1 * P) = a_s;
    *reinterpret_cast<double2*>(part + o + 2 * P) = a_w;
    *reinterpret_cast<double2*>(part + o + 3 * P) = a_z;
    *reinterpret_cast<double2*>(part + o + 4 * P) = a_y;
  }
}

static __global__ __launch_bounds__(Z_THREADS) void v_to_zy_soc_kernel(
    const double* __restrict__ v, double* __restrict__ z, double* __restrict__ y,
    const double* __restrict__ lo_, const double* __restrict__ hi_, const double* __restrict__ ub_,
    int L, int zrows, int pitch, int nb, int m) {
  const int col = (blockIdx.x * Z_THREADS + threadIdx.x) * 2;
  if (col >= pitch) return;
  const int r_begin = blockIdx.y * zrows;
  const int r_end = (r_begin + zrows < L) ? r_begin + zrows : L;
  cdouble_p lo = as_const(lo_);
  cdouble_p hi = as_const(hi_);
  cdouble_p ubv = as_const(ub_);
  for (int rb = r_begin; rb < r_end; rb += nb) {
    const double ub = ubv[rb / nb];
    const bool soc = ub < INFINITY;
    double2 cs = {1.0, 1.0};
    if (soc) {
      double2 ss = {0.0, 0.0};
      for (int j = 0; j < m; ++j) {
        const double2 vv = *reinterpret_cast<const double2*>(v + (size_t)(rb + j) * pitch + col);
        ss.x = fma(vv.x, vv.x, ss.x);
        ss.y = fma(vv.y, vv.y, ss.y);
      }
      cs = soc_scale2(ss, ub);
    }
    for (int r = 0; r < nb; ++r) {
      const int row = rb + r;
      const size_t o = (size_t)row * pitch + col;
      const double2 vv = *reinterpret_cast<const double2*>(v + o);
      double2 zz, yy;
      if (soc && r < m) {
        zz.x = vv.x * cs.x;
        zz.y = vv.y * cs.y;
      } else {
        zz.x = fmin(fmax(vv.x, lo[row]), hi[row]);
        zz.y = fmin(fmax(vv.y, lo[row]), hi[row]);
      }
      yy.x = vv.x - zz.x;
      yy.y = vv.y - zz.y;
      *reinterpret_cast<double2*>(z + o) = zz;
      *reinterpret_cast<double2*>(y + o) = yy;
    }
  }
}

// Standalone residual finalise (see finalize_body above).
static __global__ __launch_bounds__(FIN_COLS * FIN_GROUPS) void resid_finalize_kernel(FinArgs fa, int pitch) {
  __shared__ double red[FIN_GROUPS][5][FIN_COLS];
  finalize_body<FIN_GROUPS>(red, fa, blockIdx.x, pitch);
}

// ---------------------------------------------------------------------------
// Layout conversion between the ABI's QP-major arrays  src[b][e]  (b < batch,
// e < L) and the device's batch-minor  dst[e][col]  (col < pitch).  32x32 tiles
// through padded LDS so both sides are coalesced.  Setup / read-out only.
// ---------------------------------------------------------------------------
// nr > 0: the source rows are stacks of ROW-major nr x nc blocks (ADMM_FLAG_ROW_MAJOR): source row e = k * nr * nc + i * nc + j goes
// to destination row k * nr * nc + j * nr + i, the block's column-major place.
static __global__ __launch_bounds__(T_TILE * 8) void to_batch_minor_kernel(
    const double* __restrict__ src, double* __restrict__ dst, int batch, int L, int pitch, int nr = 0, int nc = 0) {
  __shared__ double tile[T_TILE][T_TILE + 1];
  const int e0 = blockIdx.x * T_TILE, b0 = blockIdx.y * T_TILE;
  const int tx = threadIdx.x & (T_TILE - 1), ty = threadIdx.x / T_TILE;
  for (int j = ty; j < T_TILE; j += 8) {
    const int b = b0 + j, e = e0 + tx;
    tile[j][tx] = (b < batch && e < L) ? src[(size_t)b * L + e] : 0.0;
  }
  __syncthreads();
  for (int j = ty; j < T_TILE; j += 8) {
    int e = e0 + j;
    const int b = b0 + tx;
    if (e < L && b < pitch) {
      if (nr > 0) {
        const int E = nr * nc, kk = e / E, ee = e % E;
        e = kk * E + (ee % nc) * nr + ee / nc;
      }
      dst[(size_t)e * pitch + b] = tile[tx][j];
    }
  }
}

static __global__ __launch_bounds__(T_TILE * 8) void from_batch_minor_kernel(
    const double* __restrict__ src, double* __restrict__ dst, int batch, int L, int pitch) {
  __shared__ double tile[T_TILE][T_TILE + 1];
  const int e0 = blockIdx.x * T_TILE, b0 = blockIdx.y * T_TILE;
  const int tx = threadIdx.x & (T_TILE - 1), ty = threadIdx.x / T_TILE;
  for (int j = ty; j < T_TILE; j += 8) {
    const int e = e0 + j, b = b0 + tx;
    tile[j][tx] = (e < L && b < pitch) ? src[(size_t)e * pitch + b] : 0.0;
  }
  __syncthreads();
  for (int j = ty; j < T_TILE; j += 8) {
    const int b = b0 + j, e = e0 + tx;
    if (b < batch && e < L) dst[(size_t)b * L + e] = tile[tx][j];
  }
}

}  // namespace admm

// admm_mfma_layout.hpp -- layout of the MFMA form of the fused stage operators, shared by the
// host packing (admm_factor.cpp, runtime n, m) and the kernels (admm_mfma.hpp, compile-time NX, NU).
// No reference counterpart exists (README.md:1-2 only).
//
// Why: with one lane per QP the per-stage matrices reach the lanes as LDS BROADCASTS, 2 LDS cycles per double
// per wave.  At n = 12, m = 6 a stage of the fused forward kernel reads 900 doubles = 1800 LDS cycles per wave,
// 7200 per CU with four waves, against 3600 cycles of fp64 FMA issue per wave: the kernel is bound by operand
// delivery, not by arithmetic or HBM.  An MFMA takes its matrix operand DISTRIBUTED over the lanes (one element
// per lane per instruction), so the same operators cost 32 ds_read_b64 per wave per stage.
//
// Form: per stage the block-banded solve is a chain of small dense products
//     [outputs](rows) x 16 QPs  =  M_k (rows x cols)  x  [inputs](cols) x 16 QPs
// with the batch-minor panel of 16 QPs as the B operand and the accumulator tile of one stage used, register for
// register, as the B operand of the next (no lane movement: see `slot` below).
//
// Vector tile ("slot" addressing).  A 16-row x 16-QP accumulator tile is 4 registers per lane; lane l serves
// QP column c = l & 15 and lane group g = l >> 4.  Register r of lane group g is called slot (r, g).  A block
// vector (u (m), x (n)), n <= 12, m <= 8, is held as
//     main tile:   slots (r, g), r < NR = ceil(n / 4)  ->  x row 4 r + g        slot (3, g)  ->  u row g
//     extra reg:   slot  (4, g)                        ->  u row 4 + g           (only when m > 4)
// (registers NR .. 2 of the main tile stand for nothing when n <= 8: they are never loaded, stored or multiplied)
// As the B operand of v_mfma_*_16x16x4 a register IS one k-step: lane group g supplies k = g, so k-step "register r"
// multiplies matrix columns {slot (r, 0..3)}.  As a D result the hardware row of slot (r, g) is
//     fp64 (v_mfma_f64_16x16x4_f64):  row = g + 4 r        fp32 (v_mfma_f32_16x16x4_f32):  row = 4 g + r
// (cdna_hip_programming.md §3); the host writes each A fragment with the matrix row the hardware row stands for,
// so the kernels never convert between the two.
//
// Products (matrices folded on the host in fp64 from the blocks of admm_layout.hpp):
//   forward kernel   SUB_F : [x+ ; u]          = [A-BK, -B Psi, -B ; -K, -Psi, -I]        [x ; t_in ; d]
//                    ELIM_F: [mu+ ; deps ; db] = [FM, PI, GA ; OB DK, 0, OB DG ; DK, 0, DG] [mu ; g^x ; g^u]
//   backward kernel  SUB_B : [x_k ; u]         = [AI-AIB' KB, -AIB' PSB, -AIB' ; -KB, -PSB, -I] [x_{k+1} ; m_in ; db]   (AIB' = A^-1 B)
//                    ELIM_B: [t+ ; de ; d0]    = [AT-KT BT, -KT ; OM SI BT, OM SI ; SI BT, SI]  [p ; g^u]
// k-steps: each input n-vector is NR registers, the m-vector 1 (+1 when m > 4).  Output tiles: tile 0 = first
// n-vector + rows 0..3 of the m-vector (slot (3, g)), tile 1 = second n-vector + rows 4..7 of the m-vector (ELIM).
// SUB has ONE output tile (x + u rows 0..3 = 16 rows exactly at n = 12); the m - 4 remaining rows of u (m > 4) would
// cost a whole second tile of MFMAs for two rows at m = 6 (a quarter of the forward kernel's matrix work), so they are
// formed on the vector unit instead: each lane multiplies its own slots by the rows' coefficients (UROW table below,
// indexed [row - 4][k-step][g]) and the four lane groups of a column are added with two cross-lane steps.
//
// Per-stage record, in the order the kernel consumes it (element type per product: mfma_es_* below):
//     SUB  fragments  [ks][64 lanes]      ELIM fragments [ks][2][64 lanes]      lo [20], hi [20]      UROW [m-4][ks][4]
// lo / hi / UROW are fp64.  lo / hi are indexed by slot (r * 4 + g, r = 0..4); slots that stand for no row carry
// (-inf, +inf).
#pragma once

namespace admm {

struct MfmaLayout {
  int nr;                    // registers of an n-vector: ceil(n / 4)
  int xt;                    // 1 when m > 4 (extra u register)
  int urows;                 // m - 4 rows of u formed on the vector unit (0 when m <= 4)
  int ks_sub;                // k-steps of SUB_F / SUB_B (one output tile)
  int ks_elim_f, ks_elim_b;  // k-steps of ELIM_F / ELIM_B (two output tiles each)
  int nf_sub, nf_elim_f, nf_elim_b;   // fragments (of 64 elements) per stage
  int lohi_doubles;          // 40
};
constexpr MfmaLayout mfma_layout(int n, int m) {
  MfmaLayout l{};
  l.nr = (n + 3) / 4;
  l.xt = m > 4 ? 1 : 0;
  l.urows = m > 4 ? m - 4 : 0;
  l.ks_sub = 2 * l.nr + 1 + l.xt;
  l.ks_elim_f = 2 * l.nr + 1 + l.xt;
  l.ks_elim_b = l.nr + 1 + l.xt;
  l.nf_sub = l.ks_sub;
  l.nf_elim_f = l.ks_elim_f * 2;
  l.nf_elim_b = l.ks_elim_b * 2;
  l.lohi_doubles = 40;
  return l;
}
constexpr bool mfma_dims(int n, int m) { return n >= 1 && n <= 12 && m >= 1 && m <= 8; }

// Element sizes by mode.  mode 2 (ADMM_PRECISION_FP64_MFMA): every product fp64.  mode 1 (ADMM_PRECISION_MIXED): the
// two products of the Riccati form -- SUB_F (forward rollout) and ELIM_B (backward elimination), whose operators are
// O(1) -- in fp32; the two of the forward-elimination form -- ELIM_F and SUB_B, whose early-stage gains reach 2.5e4
//  -- stay fp64.
constexpr int mfma_es_sub_f(int mode) { return mode == 1 ? 4 : 8; }
constexpr int mfma_es_elim_f(int) { return 8; }
constexpr int mfma_es_sub_b(int) { return 8; }
constexpr int mfma_es_elim_b(int mode) { return mode == 1 ? 4 : 8; }

// bytes of one stage's record (multiples of 16)
constexpr int mfma_tail_bytes(int n, int m) { return (40 + mfma_layout(n, m).urows * mfma_layout(n, m).ks_sub * 4) * 8; }
constexpr int mfma_rec_bytes_fwd(int n, int m, int mode) {
  return mfma_layout(n, m).nf_sub * 64 * mfma_es_sub_f(mode) + mfma_layout(n, m).nf_elim_f * 64 * mfma_es_elim_f(mode) + mfma_tail_bytes(n, m);
}
constexpr int mfma_rec_bytes_bwd(int n, int m, int mode) {
  return mfma_layout(n, m).nf_sub * 64 * mfma_es_sub_b(mode) + mfma_layout(n, m).nf_elim_b * 64 * mfma_es_elim_b(mode) + mfma_tail_bytes(n, m);
}

}  // namespace admm

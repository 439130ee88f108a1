// admm_factor.hpp -- host-side (fp64) pre-factorisation of the x-update's KKT
// system and the per-segment transfer matrices of the parallel-in-time sweep.
// DESIGN.md §2.2 (Riccati form) and §4.2 (segment algebra).  No reference
// counterpart exists (/root/reference is README.md:1-2 + LICENSE).
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/admm_hip.h"
#include "admm_layout.hpp"
#include "admm_mfma_layout.hpp"

namespace admm {

// Packed per-stage records, laid out in the order the kernels consume them.
// All row-major, fp64; block offsets (even, zero-padded) come from admm_layout.hpp.
//
// Backward record (stage k), RB doubles:
//   AT  [n][n]   AT[i][l]  = A_k[l][i]
//   BT  [m][n]   BT[j][i]  = B_k[i][j]
//   Si  [m][m]   inverse of S_k = R + rho I + B_k' P_{k+1} B_k
//   KT  [n][m]   KT[i][j]  = K_k[j][i]
//   Om  [n][m]   Omega_k   = -(Acl_{b-1} ... Acl_{k+1}) B_k     (b = segment end)
//   LO  [m+n], HI [m+n]    the box of block k (u rows, then x rows)
// Forward record (stage k), RF doubles:
//   Psi [m][n]   Psi_k     = Si_k B_k' (Acl_{b-1} ... Acl_{k+1})'
//   K   [m][n]
//   A   [n][n]
//   B   [n][m]
//   LO  [m+n], HI [m+n]
// (the box rides in both records so that a stage's whole operand set is one
//  contiguous block, staged into LDS by one coalesced copy)
// Segment record (segment s), 3 n*n doubles:
//   Phi [n][n]   (Acl_{b-1} ... Acl_a)'          tail map   t_out = t_out0 + Phi t_in
//   Xi  [n][n]   sum_k Omega_k Psi_k            x_out += Xi t_in
//   Th  [n][n]   Acl_{b-1} ... Acl_a            x_out += Th x_in
struct Factor {
  int N = 0, n = 0, m = 0, S = 0;
  double rho = 0.0;                 // the rho this factor was computed for
  int ts_ranks = 0;                 // > 1: scan inputs in the rank-by-rank layout of time-sharded handles (see factorise)
  int RB = 0, RF = 0, RS = 0;
  std::vector<int32_t> seg_start;   // S + 1 entries, seg_start[S] = N
  std::vector<double> recB;         // N * RB
  std::vector<double> recF;         // N * RF
  std::vector<double> recS;         // S * RS
  std::vector<double> K;            // N * m * n (for tests)
  std::vector<double> Sinv;         // N * m * m (for tests)

  // Segment scan as ONE dense product  out = W * in  per QP:
  //   in  rows: tseg(0..S-1) [S n] | x0 [n] | eseg(0..S-1) [S n] | zero pad      -> scanK rows
  //   out rows: t_in(0..S-1) [S n] | pad to scanMt | x_in(0..S-1) [S n] | pad    -> scanM rows
  // scanW is dense row-major scanM x scanK; scanWp is the same matrix packed in
  // v_mfma_f64_16x16x4 A-fragment order: Wp[kstep][mtile][lane] =
  // W[16 mtile + (lane & 15)][4 kstep + (lane >> 4)].  scanRange holds, per group of
  // SCAN_MT M-tiles, the [begin, end) k-step range outside which that group's rows of W
  // are identically zero (block-triangular structure).
  int scanM = 0, scanMt = 0, scanK = 0;
  std::vector<double> scanW, scanWp;
  std::vector<int32_t> scanRange;   // 2 * (scanM / 16 / SCAN_MT)

  // Alternating-direction iteration: records of the two fused kernels
  // (layouts in admm_layout.hpp) and the scan matrix of the forward-elimination form, same
  // shape and row/column layout as scanW:
  //   in  rows: mseg(0..S-1) | x0 | ebseg(0..S-1)           out rows: m_in(0..S-1) | x_end(0..S-1)
  // alt_ok is false when that form could not be built (then only the plain iteration runs).
  bool alt_ok = false;
  double alt_check = -1.0;          // relative mismatch of the two forms on the host verification vector (-1: not run)
  int RFE = 0, RBE = 0;
  std::vector<double> recFE;        // N * RFE
  std::vector<double> recBE;        // N * RBE
  std::vector<double> scanWB, scanWpB;
  std::vector<int32_t> scanRangeB;

  // MFMA form of the fused stage operators (admm_mfma_layout.hpp): per-stage records of
  // pre-packed A fragments + the box in slot order, as raw bytes (element type per product: mfma_es_*(mode)).
  // mfma_mode = 0: not built; 1 = mixed (SUB_F, ELIM_B fp32; ELIM_F, SUB_B fp64); 2 = all fp64.
  // recMF = forward kernel (SUB_F, ELIM_F), recMB = backward kernel (SUB_B, ELIM_B);
  // the ELIM_F / SUB_B fragments are zero when alt_ok is false (only the plain path may then use the records).
  int mfma_mode = 0;
  int RMF = 0, RMB = 0;             // bytes per stage
  std::vector<unsigned char> recMF; // N * RMF
  std::vector<unsigned char> recMB; // N * RMB
  // mfma_mode 1 only: the all-fp64 records as well, for the fp64 refinement phase of a mixed solve
  int RMF64 = 0, RMB64 = 0;
  std::vector<unsigned char> recMF64, recMB64;
};

#ifndef ADMM_SCAN_MT
#define ADMM_SCAN_MT 4
#endif
#ifndef ADMM_SCAN_U
#define ADMM_SCAN_U 8
#endif
constexpr int SCAN_MT = ADMM_SCAN_MT;      // M-tiles (of 16 rows) per wave in xscan_mfma_kernel
constexpr int SCAN_KALIGN = ADMM_SCAN_U;   // k-step ranges and K/4 are padded to the kernel's batch (SCAN_U k-steps)

inline int rec_b_size(int n, int m) { return rec_b_layout(n, m).SIZE; }
inline int rec_f_size(int n, int m) { return rec_f_layout(n, m).SIZE; }
inline int rec_s_size(int n) { return 3 * n * n; }
inline int rec_fe_size(int n, int m) { return rec_fe_layout(n, m).SIZE; }
inline int rec_be_size(int n, int m) { return rec_be_layout(n, m).SIZE; }

// Validates nothing about the batch; only dynamics/weights.  Returns an
// admm_status; err receives a message on failure.
// mfma_mode: 0 = no MFMA records; 1 / 2 = also pack the MFMA form, mixed / fp64 (needs mfma_dims(n, m)).
// pack_scan_mfma = false leaves scanWp / scanRange (and the B pair) empty: handles whose scan runs as a matrix-vector
// product (batches of <= 4 QPs) only need the dense matrices.
// ts_ranks > 1 (time-sharded handles, include/admm_hip.h): the INPUT rows of both scan products are laid out rank by rank --
//   [ rank 0: tseg of its S / ts_ranks segments | eseg of them ] [ rank 1: ... ] ... | x0 | pad
// instead of tseg(0..S-1) | x0 | eseg(0..S-1), so that the segment summaries of all ranks are completed by ONE all-gather of
// contiguous, equal slices (the columns of scanW / scanWB are permuted accordingly before they are packed).
int factorise(const admm_problem& p, double rho, int segments, Factor& out, std::string& err, int mfma_mode = 0,
              bool pack_scan_mfma = true, int ts_ranks = 0);

// The per-stage / per-segment loops of factorise run on host threads (ADMM_FACTOR_THREADS, default min(cores, 16)); this
// caps the count for factorisations started from the calling thread (0 = no cap).  Used by the background refactors of
// the adaptive-rho rule so that they leave cores to the thread that launches kernels.
void set_factor_thread_cap(int cap);

}  // namespace admm

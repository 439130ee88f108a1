// admm_layout.hpp -- layout of the packed per-stage records, shared by the host
// factorisation (runtime n, m) and the kernels (compile-time NX, NU).
//
// Every block starts at an even offset and every record has even length, so that
// with a 16-byte-aligned base all blocks can be read from LDS as 16-byte pairs
// (ds_read_b128: 2 LDS cycles per double; the 8-byte-aligned ds_read2_b64 the
// compiler would otherwise pick costs 4).  Pad entries are zero and never used.
#pragma once

namespace admm {

constexpr int even_up(int v) { return (v + 1) & ~1; }

// Backward record:  AT [n][n] | BT [m][n] | SI [m][m] | KT [n][m] | OM [n][m] | LO [m+n] | HI [m+n] | UB [1]
// (UB = thrust-magnitude bound of the stage, +inf when the control rows use their box)
struct RecBLayout {
  int AT, BT, SI, KT, OM, LO, HI, UB, SIZE;
};
constexpr RecBLayout rec_b_layout(int n, int m) {
  RecBLayout l{};
  l.AT = 0;
  l.BT = l.AT + even_up(n * n);
  l.SI = l.BT + even_up(m * n);
  l.KT = l.SI + even_up(m * m);
  l.OM = l.KT + even_up(n * m);
  l.LO = l.OM + even_up(n * m);
  l.HI = l.LO + even_up(n + m);
  l.UB = l.HI + even_up(n + m);
  l.SIZE = l.UB + 2;
  return l;
}

// Forward record:  PSI [m][n] | K [m][n] | A [n][n] | B [n][m] | LO [m+n] | HI [m+n] | UB [1]
struct RecFLayout {
  int PSI, K, A, B, LO, HI, UB, SIZE;
};
constexpr RecFLayout rec_f_layout(int n, int m) {
  RecFLayout l{};
  l.PSI = 0;
  l.K = l.PSI + even_up(m * n);
  l.A = l.K + even_up(m * n);
  l.B = l.A + even_up(n * n);
  l.LO = l.B + even_up(n * m);
  l.HI = l.LO + even_up(n + m);
  l.UB = l.HI + even_up(n + m);
  l.SIZE = l.UB + 2;
  return l;
}

// Records of the alternating-direction iteration.  Even iterations solve the
// x-update by backward elimination / forward substitution (the Riccati form above), odd ones by
// forward elimination / backward substitution (its time-reversed mirror), so that each substitution
// sweep runs in the same direction as -- and is fused with -- the next iteration's elimination.
//
// Forward fused record (xfze_kernel):  the forward record's blocks, then
//   FM [n][n]  F_k   = (I - G_{k+1}) A_k                 mu_{k+1} = F mu_k + GA g^u + PI g^x
//   GA [n][m]  Gam_k = -(I - G_{k+1}) B_k Rr^{-1}
//   PI [n][n]  Pi_k  = -G_{k+1} Qr_{k+1}^{-1}
//   DK [m][n], DG [m][m]    db_k = DK mu_k + DG g^u        (the stored m rows of the stage)
//   OB [n][m]  Omb_k                                       ebseg += OB db_k
struct RecFELayout {
  int PSI, K, A, B, FM, GA, PI, DK, DG, OB, LO, HI, UB, SIZE;
};
constexpr RecFELayout rec_fe_layout(int n, int m) {
  RecFELayout l{};
  l.PSI = 0;
  l.K = l.PSI + even_up(m * n);
  l.A = l.K + even_up(m * n);
  l.B = l.A + even_up(n * n);
  l.FM = l.B + even_up(n * m);
  l.GA = l.FM + even_up(n * n);
  l.PI = l.GA + even_up(n * m);
  l.DK = l.PI + even_up(n * n);
  l.DG = l.DK + even_up(m * n);
  l.OB = l.DG + even_up(m * m);
  l.LO = l.OB + even_up(n * m);
  l.HI = l.LO + even_up(n + m);
  l.UB = l.HI + even_up(n + m);
  l.SIZE = l.UB + 2;
  return l;
}

// Backward fused record (xbze_kernel):
//   PSB [m][n]  Psb_k                  d    = db_k + PSB m_in
//   KB  [m][n]  Kb_k                   u_k  = -KB x_{k+1} - d
//   AI  [n][n]  A_k^{-1}               x_k  = AI x_{k+1} + AIB u_k
//   AIB [n][m]  -A_k^{-1} B_k
//   AT, BT, SI, KT, OM                  as in the backward record (the elimination half)
struct RecBELayout {
  int PSB, KB, AI, AIB, AT, BT, SI, KT, OM, LO, HI, UB, SIZE;
};
constexpr RecBELayout rec_be_layout(int n, int m) {
  RecBELayout l{};
  l.PSB = 0;
  l.KB = l.PSB + even_up(m * n);
  l.AI = l.KB + even_up(m * n);
  l.AIB = l.AI + even_up(n * n);
  l.AT = l.AIB + even_up(n * m);
  l.BT = l.AT + even_up(n * n);
  l.SI = l.BT + even_up(m * n);
  l.KT = l.SI + even_up(m * m);
  l.OM = l.KT + even_up(n * m);
  l.LO = l.OM + even_up(n * m);
  l.HI = l.LO + even_up(n + m);
  l.UB = l.HI + even_up(n + m);
  l.SIZE = l.UB + 2;
  return l;
}

}  // namespace admm

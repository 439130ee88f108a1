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
// forward elimination / backward substitution (information-filter form), so that each substitution
// sweep runs in the same direction as -- and is fused with -- the next iteration's elimination.
//
// Forward fused record (xfze_kernel):  the forward record's blocks, then
//   FM [n][n]  F_k   = (I - G_{k+1}) A_k                 m_{k+1} = F m_k + GA g^u + PI g^x
//   GA [n][m]  Gam_k = -(I - G_{k+1}) B_k Rr^{-1}
//   PI [n][n]  Pi_k  = -G_{k+1} Qr_{k+1}^{-1}
//   YU [n][m], YX [n][n]   the segment's costate summary  eps += YU g^u + YX g^x
struct RecFELayout {
  int PSI, K, A, B, FM, GA, PI, YU, YX, LO, HI, UB, SIZE;
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
  l.YU = l.PI + even_up(n * n);
  l.YX = l.YU + even_up(n * m);
  l.LO = l.YX + even_up(n * n);
  l.HI = l.LO + even_up(n + m);
  l.UB = l.HI + even_up(n + m);
  l.SIZE = l.UB + 2;
  return l;
}

// Backward fused record (xbze_kernel):
//   PHF [n][n]  F_k ... F_a            m_{k+1} = m0_{k+1} + PHF m_in        (a = segment start)
//   CM  [n][n]  C_{k+1}                x_{k+1} = m_{k+1} + CM lam
//   QM  [n][n]  Qr_{k+1}               nu      = lam - QM x_{k+1} - g^x
//   RB  [m][n]  Rr^{-1} B_k'           u_k     = RB nu - RI g^u
//   RI  [m][m]  Rr^{-1}
//   AT  [n][n]  A_k'                   lam'    = AT nu        (also the elimination's A' p)
//   BT, SI, KT, OM                      as in the backward record
struct RecBELayout {
  int PHF, CM, QM, RB, RI, AT, BT, SI, KT, OM, LO, HI, UB, SIZE;
};
constexpr RecBELayout rec_be_layout(int n, int m) {
  RecBELayout l{};
  l.PHF = 0;
  l.CM = l.PHF + even_up(n * n);
  l.QM = l.CM + even_up(n * n);
  l.RB = l.QM + even_up(n * n);
  l.RI = l.RB + even_up(m * n);
  l.AT = l.RI + even_up(m * m);
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

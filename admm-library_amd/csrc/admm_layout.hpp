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

}  // namespace admm

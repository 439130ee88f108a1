// Per-instance dynamics: the WIDE shapes (admm_pinst_wide.hpp) -- every kernel with a QP's rows spread over the lanes of a wave,
// whatever the batch: one lane cannot hold a stage's operands from n = 8 on; operand arrays TILED (admm_pinst.hpp, Operand).
// The thrust-magnitude forms project over a QP's control rows with cross-lane reads (admm_pinst_rows.hpp, SOC).
#include "admm_pinst_launch.hpp"

#define ADMM_PINST_DIMS_G2(X) X(12, 6) X(8, 4) X(12, 3) X(9, 3)
ADMM_PINST_GROUP_WIDE(g2, ADMM_PINST_DIMS_G2)

namespace admm {
// LDS a workgroup of the wide sweeps asks for at most (bytes; 0 = not a wide shape): checked against the device at set-up
size_t pinst_wide_lds_bytes(int n, int m) {
#define X(NX, NU)                                                                                                      \
  if (n == NX && m == NU) {                                                                                            \
    const size_t w = PROWS_BLOCK / PI_THREADS;                                                                         \
    const size_t b = w * pxb_rows_lds_words<NX, NU, true>(), f = w * pxfz_rows_lds_words<NX, NU, true>();              \
    return sizeof(double) * (b > f ? b : f);                                                                           \
  }
  ADMM_PINST_DIMS_G2(X)
#undef X
  return 0;
}

// (the one-lane factor / segment kernels' twins at (6, 3): tests/test_gpu_pinst.py compares their output bit for bit)
void launch_pinst_rows_twin_6_3(const PLaunch& l, PKernel k) { launch_dim_wide<6, 3, false>(l, k); }
}  // namespace admm

// Per-instance dynamics: the second group of (n, m) shapes (see admm_pinst.hip).  (8, 4) and beyond do not fit these one-lane kernels
// (the device factorisation and the sweeps keep a stage's operands in registers; (8, 4) spilled 1.2 KB per lane): they are the wide
// shapes of admm_pinst_g2.hip, with a QP's rows spread over the lanes of a wave.
#include "admm_pinst_launch.hpp"

#define ADMM_PINST_DIMS_G1(X) X(1, 1) X(2, 2) X(4, 1) X(6, 1) X(6, 2) X(6, 4)
ADMM_PINST_GROUP(g1, ADMM_PINST_DIMS_G1)

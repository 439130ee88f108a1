// MFMA form of the fused iteration kernels (admm_mfma.hpp): instantiations and launcher.
// Adding a shape = adding X(n, m) to ADMM_MFMA_DIMS (n <= 12, m <= 8).
#include "admm_dispatch.hpp"
#include "admm_mfma.hpp"

#define ADMM_MFMA_DIMS(X) X(12, 6) X(6, 3) X(10, 4)

namespace admm {

const char* dims_mfma() {
#define ADMM_STR2(x) #x
#define ADMM_STR(x) ADMM_STR2(x)
#define X(NX, NU) "(" ADMM_STR(NX) "," ADMM_STR(NU) ") "
  return ADMM_MFMA_DIMS(X);
#undef X
}

namespace {

template <int NX, int NU>
bool launch_dim(const XLaunch& l, XKernel k, bool resid, bool query_only) {
  const bool f32 = l.mfma_elem == 4;
  // fp32: the plain path only (xb / xfz roles); fp64: the alternating pair
  const bool ok = f32 ? (k == XKernel::XB || k == XKernel::XFZ) : (k == XKernel::XFZE || k == XKernel::XBZE);
  if (!ok) return false;
  if (query_only) return true;
  const dim3 grid((l.pitch + MF_COLS - 1) / MF_COLS, l.S), block(MF_THREADS);
  const bool relax = l.alpha != 1.0;
#define FWD(T, RS, RX, EL)                                                                                          \
  hipLaunchKernelGGL((xfzem_kernel<NX, NU, T, RS, RX, EL>), grid, block, 0, l.stream, l.dbuf, l.tin, l.xin, l.recMF, \
                     l.seg_start, l.v, l.mvec, l.tseg, l.eseg, l.part, l.alpha, l.rho, l.pitch, l.nsplit, l.split_stride)
#define BWD(T, RS, RX, SB)                                                                                          \
  hipLaunchKernelGGL((xbzem_kernel<NX, NU, T, RS, RX, SB>), grid, block, 0, l.stream, l.mvec, l.tin, l.xin, l.recMB, \
                     l.seg_start, l.v, l.dbuf, l.tseg, l.eseg, l.part, l.alpha, l.rho, l.pitch, l.nsplit, l.split_stride)
#define BY_FLAGS(CALL, T, LAST)                                                   \
  do {                                                                            \
    if (resid) { if (relax) CALL(T, true, true, LAST); else CALL(T, true, false, LAST); }   \
    else       { if (relax) CALL(T, false, true, LAST); else CALL(T, false, false, LAST); } \
  } while (0)
  switch (k) {
    case XKernel::XFZE: BY_FLAGS(FWD, double, true); break;
    case XKernel::XBZE: BY_FLAGS(BWD, double, true); break;
    case XKernel::XFZ:  BY_FLAGS(FWD, float, false); break;
    case XKernel::XB:   BWD(float, false, false, false); break;
    default: return false;
  }
#undef BY_FLAGS
#undef BWD
#undef FWD
  return true;
}

}  // namespace

bool launch_mfma(const XLaunch& l, XKernel k, bool resid, bool query_only) {
#define X(NX, NU) if (l.n == NX && l.m == NU) return launch_dim<NX, NU>(l, k, resid, query_only);
  ADMM_MFMA_DIMS(X)
#undef X
  return false;
}

}  // namespace admm

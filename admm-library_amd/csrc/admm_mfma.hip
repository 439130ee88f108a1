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
  const bool mixed = l.mfma_mode == 1;
  // mode 2 (fp64): the alternating pair; mode 1 (mixed): the pair, and the plain path's v-form kernels (xb / xfz roles)
  const bool ok = (k == XKernel::XFZE || k == XKernel::XBZE) || (mixed && (k == XKernel::XB || k == XKernel::XFZ));
  if (!ok) return false;
  if (query_only) return true;
  // one 16-QP tile per wave for small batches (waves beyond the batch idle), two otherwise
  const bool nt1 = l.pitch <= 128;
  const dim3 grid = sweep_grid((l.pitch + mf_cols(nt1 ? 1 : 2) - 1) / mf_cols(nt1 ? 1 : 2), l.S), block(MF_THREADS);
  const bool relax = l.alpha != 1.0;
  if (l.has_q) {
    // A linear term q: HASQ forms of the fp64 alternating pair with one tile per wave (small batches), for the shape of the
    // successive-convexification QPs only -- every form doubles the compile time of this unit, and at n = 12 the extra
    // prefetch registers do not fit under the 256-register cap of two waves per SIMD.  (admm_setup picks the one-lane kernels
    // for every other problem with q.)
    if constexpr (NX == 6 && NU == 3) {
      if (!nt1 || mixed || !(k == XKernel::XFZE || k == XKernel::XBZE)) return false;
#define QF(RS, RX, XF) hipLaunchKernelGGL((xfzem_kernel<NX, NU, 1, double, double, RS, RX, true, XF, true>), grid, block, 0, l.stream, \
                                          l.dbuf, l.tin, l.xin, l.recMF, l.seg_start, l.v, l.mvec, l.tseg, l.eseg, l.part, l.alpha, l.rho, \
                                          l.pitch, l.nsplit, l.split_stride, l.batch, l.q)
#define QB(RS, RX, XF) hipLaunchKernelGGL((xbzem_kernel<NX, NU, 1, double, double, RS, RX, true, XF, true>), grid, block, 0, l.stream, \
                                          l.mvec, l.tin, l.xin, l.recMB, l.seg_start, l.v, l.dbuf, l.tseg, l.eseg, l.part, l.alpha, l.rho, \
                                          l.pitch, l.nsplit, l.split_stride, l.batch, l.q)
#define QK(RS, RX, XF) do { if (k == XKernel::XFZE) QF(RS, RX, XF); else QB(RS, RX, XF); } while (0)
      if (resid) { if (relax) QK(true, true, 0); else QK(true, false, 0); }
      else if (relax) QK(false, true, 0);
      else if (l.xfree == 2) QK(false, false, 2);
      else if (l.xfree == 1) QK(false, false, 1);
      else QK(false, false, 0);
#undef QK
#undef QB
#undef QF
      return true;
    } else {
      return false;
    }
  }
#define FWD0(NT_, TS, TE, RS, RX, EL, XF)                                                                               \
  hipLaunchKernelGGL((xfzem_kernel<NX, NU, NT_, TS, TE, RS, RX, EL, XF>), grid, block, 0, l.stream, l.dbuf, l.tin, l.xin, l.recMF, \
                     l.seg_start, l.v, l.mvec, l.tseg, l.eseg, l.part, l.alpha, l.rho, l.pitch, l.nsplit, l.split_stride, l.batch)
#define BWD0(NT_, TS, TE, RS, RX, SB, XF)                                                                               \
  hipLaunchKernelGGL((xbzem_kernel<NX, NU, NT_, TS, TE, RS, RX, SB, XF>), grid, block, 0, l.stream, l.mvec, l.tin, l.xin, l.recMB, \
                     l.seg_start, l.v, l.dbuf, l.tseg, l.eseg, l.part, l.alpha, l.rho, l.pitch, l.nsplit, l.split_stride, l.batch)
// XFREE forms for the non-residual, non-relaxed kernels that update v (not the plain path's backward sweep)
#define FWD1(NT_, TS, TE, RS, RX, EL) do { if constexpr (!(RS) && !(RX)) { if (l.xfree == 2) FWD0(NT_, TS, TE, RS, RX, EL, 2); else if (l.xfree == 1) FWD0(NT_, TS, TE, RS, RX, EL, 1); else FWD0(NT_, TS, TE, RS, RX, EL, 0); } else FWD0(NT_, TS, TE, RS, RX, EL, 0); } while (0)
#define BWD1(NT_, TS, TE, RS, RX, SB) do { if constexpr (!(RS) && !(RX) && (SB)) { if (l.xfree == 2) BWD0(NT_, TS, TE, RS, RX, SB, 2); else if (l.xfree == 1) BWD0(NT_, TS, TE, RS, RX, SB, 1); else BWD0(NT_, TS, TE, RS, RX, SB, 0); } else BWD0(NT_, TS, TE, RS, RX, SB, 0); } while (0)
#define FWD(TS, TE, RS, RX, EL) do { if (nt1) FWD1(1, TS, TE, RS, RX, EL); else FWD1(2, TS, TE, RS, RX, EL); } while (0)
#define BWD(TS, TE, RS, RX, SB) do { if (nt1) BWD1(1, TS, TE, RS, RX, SB); else BWD1(2, TS, TE, RS, RX, SB); } while (0)
#define BY_FLAGS(CALL, TS, TE, LAST)                                                      \
  do {                                                                                    \
    if (resid) { if (relax) CALL(TS, TE, true, true, LAST); else CALL(TS, TE, true, false, LAST); }   \
    else       { if (relax) CALL(TS, TE, false, true, LAST); else CALL(TS, TE, false, false, LAST); } \
  } while (0)
  switch (k) {
    case XKernel::XFZE: if (mixed) BY_FLAGS(FWD, float, double, true); else BY_FLAGS(FWD, double, double, true); break;
    case XKernel::XBZE: if (mixed) BY_FLAGS(BWD, double, float, true); else BY_FLAGS(BWD, double, double, true); break;
    case XKernel::XFZ:  BY_FLAGS(FWD, float, double, false); break;
    case XKernel::XB:   BWD(double, float, false, false, false); break;
    default: return false;
  }
#undef FWD1
#undef BWD1
#undef FWD0
#undef BWD0
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

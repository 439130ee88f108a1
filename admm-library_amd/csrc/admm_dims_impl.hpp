// admm_dims_impl.hpp -- body shared by admm_dims_g*.hip.  The including file defines
//   ADMM_GROUP_FN    launch_groupK
//   ADMM_GROUP_LIST  dims_groupK
//   ADMM_GROUP_DIMS(X)   X(n, m) X(n, m) ...
#include "admm_dispatch.hpp"
#include "admm_kernels.hpp"
#include "admm_kernels_alt.hpp"

namespace admm {

#define ADMM_STR2(x) #x
#define ADMM_STR(x) ADMM_STR2(x)

const char* ADMM_GROUP_LIST() {
#define X(NX, NU) "(" ADMM_STR(NX) "," ADMM_STR(NU) ") "
  return ADMM_GROUP_DIMS(X);
#undef X
}

namespace {

template <int NX, int NU>
void launch_dim(const XLaunch& l, XKernel k, bool a, bool b) {
  const dim3 grid = sweep_grid((l.pitch + XB_THREADS - 1) / XB_THREADS, l.S), block(XB_THREADS);
  switch (k) {
    case XKernel::XB: {
      const double* zin = a ? l.v : l.z;
#define XB1(HQ, VF, SC)                                                                                  \
  hipLaunchKernelGGL((xb_kernel<NX, NU, HQ, VF, SC>), grid, block, 0, l.stream, zin, l.y, l.q, l.recB,   \
                     l.seg_start, l.dbuf, l.tseg, l.eseg, l.rho, l.pitch)
// the SOC form differs only where z is rebuilt from v (VFORM)
#define XB(HQ, VF) do { if constexpr (VF) { if (l.has_soc) XB1(HQ, VF, true); else XB1(HQ, VF, false); } else XB1(HQ, VF, false); } while (0)
      if (l.has_q) { if (a) XB(true, true); else XB(true, false); }
      else         { if (a) XB(false, true); else XB(false, false); }
#undef XB
#undef XB1
      break;
    }
    case XKernel::XF:
      hipLaunchKernelGGL((xf_kernel<NX, NU>), grid, block, 0, l.stream, l.dbuf, l.tin, l.xin, l.recF,
                         l.seg_start, l.w, l.pitch, l.nsplit, l.split_stride);
      break;
    case XKernel::XFZ: {
      const bool relax = l.alpha != 1.0;
#define XFZ1(RS, RX, VI, SC)                                                                                 \
  hipLaunchKernelGGL((xfz_kernel<NX, NU, RS, RX, VI, SC>), grid, block, 0, l.stream, l.dbuf, l.tin, l.xin,   \
                     l.recF, l.seg_start, l.z, l.y, l.v, l.part, l.alpha, l.pitch, l.nsplit, l.split_stride)
// the SOC form matters where z is rebuilt from v (VIN) or z+ is formed (RESID)
#define XFZ(RS, RX, VI) do { if constexpr (RS || VI) { if (l.has_soc) XFZ1(RS, RX, VI, true); else XFZ1(RS, RX, VI, false); } else XFZ1(RS, RX, VI, false); } while (0)
#define XFZ2(RS, RX) do { if (a) XFZ(RS, RX, true); else XFZ(RS, RX, false); } while (0)
      if (b) { if (relax) XFZ2(true, true); else XFZ2(true, false); }
      else   { if (relax) XFZ2(false, true); else XFZ2(false, false); }
#undef XFZ2
#undef XFZ
#undef XFZ1
      break;
    }
    case XKernel::XSCAN_CHAIN:
      hipLaunchKernelGGL((xscan_kernel<NX>), dim3(l.pitch / 64), dim3(64), 0, l.stream, l.tseg, l.eseg, l.x0,
                         l.recS, l.tin, l.xin, l.S, l.pitch);
      break;
    case XKernel::XFZE:
    case XKernel::XBZE:
      if constexpr (alt_dims(NX, NU)) {
        const bool relax = l.alpha != 1.0;
        // the scan's input / output slots are shared by both directions: (tseg, eseg) = (mseg, epsseg),
        // (tin, xin) = (m_in, x_end)
#define ALT4(RS, RX, HQ, SC, XF)                                                                                \
  do {                                                                                                       \
    if (k == XKernel::XFZE)                                                                                  \
      hipLaunchKernelGGL((xfze_kernel<NX, NU, RS, RX, HQ, SC, XF>), grid, block, 0, l.stream, l.dbuf, l.tin, l.xin,  \
                         l.recFE, l.seg_start, l.q, l.v, l.mvec, l.tseg, l.eseg, l.part, l.alpha, l.rho,     \
                         l.pitch, l.nsplit, l.split_stride);                                                 \
    else                                                                                                     \
      hipLaunchKernelGGL((xbze_kernel<NX, NU, RS, RX, HQ, SC, XF>), grid, block, 0, l.stream, l.mvec, l.tin, l.xin,  \
                         l.recBE, l.seg_start, l.q, l.v, l.dbuf, l.tseg, l.eseg, l.part, l.alpha, l.rho,     \
                         l.pitch, l.nsplit, l.split_stride);                                                 \
  } while (0)
// XFREE forms (state rows unbounded everywhere: their v is not read -- 1 -- and, when the next iteration is of the same kind,
// not written either -- 2) exist for the non-residual, non-relaxed kernels only
#define ALT3(RS, RX, HQ, SC)                                                                                  \
  do {                                                                                                       \
    if constexpr (!(RS) && !(RX)) {                                                                          \
      if (l.xfree == 2) ALT4(RS, RX, HQ, SC, 2); else if (l.xfree == 1) ALT4(RS, RX, HQ, SC, 1); else ALT4(RS, RX, HQ, SC, 0); \
    } else if constexpr ((RS) && !(RX) && NX + NU >= 12) {   /* residual form with the state rows' shortcut: fp64-issue-bound blocks only */ \
      if (l.xfree) ALT4(RS, RX, HQ, SC, 1); else ALT4(RS, RX, HQ, SC, 0);                                     \
    } else ALT4(RS, RX, HQ, SC, 0);                                                                          \
  } while (0)
#define ALT2(RS, RX, HQ) do { if (l.has_soc) ALT3(RS, RX, HQ, true); else ALT3(RS, RX, HQ, false); } while (0)
#define ALT1(RS, RX) do { if (l.has_q) ALT2(RS, RX, true); else ALT2(RS, RX, false); } while (0)
        if (b) { if (relax) ALT1(true, true); else ALT1(true, false); }
        else   { if (relax) ALT1(false, true); else ALT1(false, false); }
#undef ALT2
#undef ALT3
#undef ALT4
#undef ALT1
      }
      break;
  }
}

}  // namespace

bool ADMM_GROUP_FN(const XLaunch& l, XKernel k, bool a, bool b, bool query_only) {
#define X(NX, NU)                               \
  if (l.n == NX && l.m == NU) {                 \
    if ((k == XKernel::XFZE || k == XKernel::XBZE) && !alt_dims(NX, NU)) return false; \
    if (!query_only) launch_dim<NX, NU>(l, k, a, b); \
    return true;                                \
  }
  ADMM_GROUP_DIMS(X)
#undef X
  return false;
}

}  // namespace admm

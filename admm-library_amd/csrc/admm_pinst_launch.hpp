// admm_pinst_launch.hpp -- the (n, m)-templated launcher of the per-instance kernels, shared by the translation units that
// instantiate them (admm_pinst.hip, admm_pinst_g1.hip: the shapes are split so that the build compiles them in parallel).
#pragma once

#include <atomic>

#include "admm_dispatch.hpp"
#include "admm_pinst.hpp"
#include "admm_pinst_rows.hpp"
#include "admm_pinst_wide.hpp"

namespace admm {
namespace {

// Launch with dynamic LDS; beyond the 64 KB a kernel may use by default the limit is raised first (once per kernel and device).
template <auto Kernel, class... Args>
void launch_with_lds(dim3 grid, dim3 block, size_t lds_bytes, hipStream_t stream, Args... args) {
  if (lds_bytes > 64 * 1024) {
    static std::atomic<unsigned> raised{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((raised.load(std::memory_order_relaxed) >> dev) & 1u)) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      raised.fetch_or(1u << dev, std::memory_order_relaxed);
    }
  }
  hipLaunchKernelGGL(Kernel, grid, block, lds_bytes, stream, args...);
}

// Grid of the rows-over-lanes kernels: a wave serves QPW QPs; from 65 QPs on a workgroup takes the 16 / QPW waves whose
// accesses share the 128-byte lines of every array (their 8 * QPW-byte pieces would otherwise be fetched by different XCDs).
template <int NX>
struct RowsGrid {
  dim3 grid, block, grid1;     // grid1: without the segment dimension (factorisation)
  int waves_per_block;
  explicit RowsGrid(const PLaunch& l) {
    constexpr int QPW = PscanShape<NX>::QPW;
    const int wpb = l.pitch > 64 ? (16 / QPW > 1 ? 16 / QPW : 1) : 1;
    const int waves = l.pitch / QPW;
    block = dim3(PI_THREADS * wpb);
    waves_per_block = wpb;
    grid1 = dim3((waves + wpb - 1) / wpb);
    grid = dim3(grid1.x, l.S > 0 ? l.S : 1);
  }
};

// Wide shapes (admm_pinst_wide.hpp): every kernel in its rows-over-lanes form (SOC: the thrust-magnitude forms of the sweeps).  TILED: the operand
// arrays in the tiled layout (admm_pinst.hpp, Operand) -- the wide shapes' handles; false = the twin at (6, 3) on a batch-minor handle.
template <int NX, int NU, bool TILED>
void launch_dim_wide(const PLaunch& l, PKernel k) {
  const bool seg = l.S > 1;
  const RowsGrid<NX> rg(l);
  const dim3 rgrid = rg.grid, rblock = rg.block;
  const bool relax = l.alpha != 1.0;
  switch (k) {
    case PKernel::SEGMENTS:
      hipLaunchKernelGGL((pseg_rows_kernel<NX, NU, TILED>), rgrid, rblock, 0, l.stream, l.Ad, l.Bd, l.Kd, l.Sd, l.seg_start, l.todo, l.Omd,
                         l.Psd, l.Segd, l.grow, l.pitch, l.batch, l.qflag);
      break;
    case PKernel::SCAN:
      hipLaunchKernelGGL((pscan_kernel<NX>), dim3(l.pitch / PscanShape<NX>::QPW), dim3(PI_THREADS), 0, l.stream, l.Segd, l.tseg,
                         l.eseg, l.x0, l.tin, l.xin, l.S, l.pitch);
      break;
    case PKernel::FACTOR:
      hipLaunchKernelGGL((pfactor_rows_kernel<NX, NU, TILED>), rg.grid1, rblock, 0, l.stream, l.Ad, l.Bd, l.Q, l.R, l.QN, l.rhov, l.todo,
                         l.Kd, l.Sd, l.fail, l.N, l.pitch, l.batch, l.qflag);
      break;
    case PKernel::XB: {
#define XB(HQ, VF, PB_, SC)                                                                                                        \
  do {                                                                                                                           \
    if (seg)                                                                                                                     \
      launch_with_lds<pxb_rows_kernel<NX, NU, HQ, VF, PB_, true, TILED, SC>>(rgrid, rblock, TILED ? rg.waves_per_block * pxb_rows_lds_words<NX, NU, true>() * sizeof(double) : 0, l.stream, l.vform ? l.v : l.z, l.y, l.q, \
                         l.Ad, l.Bd, l.Kd, l.Sd, l.lo, l.hi, l.dbuf, l.rhov, l.N, l.pitch, l.Omd, l.seg_start, l.tseg, l.eseg, l.loT, l.hiT, l.ub);   \
    else                                                                                                                         \
      launch_with_lds<pxb_rows_kernel<NX, NU, HQ, VF, PB_, false, TILED, SC>>(rgrid, rblock, TILED ? rg.waves_per_block * pxb_rows_lds_words<NX, NU, false>() * sizeof(double) : 0, l.stream, l.vform ? l.v : l.z, l.y,    \
                         l.q, l.Ad, l.Bd, l.Kd, l.Sd, l.lo, l.hi, l.dbuf, l.rhov, l.N, l.pitch, nullptr, nullptr, nullptr,       \
                         nullptr, l.loT, l.hiT, l.ub);                                                                                               \
  } while (0)
#define XB3(HQ, VF, PB_) do { if constexpr (VF && TILED) { if (l.has_soc) XB(HQ, VF, PB_, true); else XB(HQ, VF, PB_, false); } else XB(HQ, VF, PB_, false); } while (0)
#define XB2(HQ, VF) do { if (l.pbounds) XB3(HQ, VF, true); else XB3(HQ, VF, false); } while (0)
      if (l.has_q) { if (l.vform) XB2(true, true); else XB2(true, false); }
      else         { if (l.vform) XB2(false, true); else XB2(false, false); }
#undef XB2
#undef XB3
#undef XB
      break;
    }
    case PKernel::XF:
      if (seg)
        launch_with_lds<pxfz_rows_kernel<NX, NU, false, false, false, false, false, true, true, TILED>>(rgrid, rblock, TILED ? rg.waves_per_block * pxfz_rows_lds_words<NX, NU, true>() * sizeof(double) : 0, l.stream,
                           l.dbuf, l.x0, l.Ad, l.Bd, l.Kd, l.lo, l.hi, l.z, l.y, l.v, l.w, l.part, l.alpha, l.N, l.pitch, l.Psd,
                           l.seg_start, l.tin, l.xin, l.loT, l.hiT, l.ub);
      else
        launch_with_lds<pxfz_rows_kernel<NX, NU, false, false, false, false, false, true, false, TILED>>(rgrid, rblock, TILED ? rg.waves_per_block * pxfz_rows_lds_words<NX, NU, false>() * sizeof(double) : 0, l.stream,
                           l.dbuf, l.x0, l.Ad, l.Bd, l.Kd, l.lo, l.hi, l.z, l.y, l.v, l.w, l.part, l.alpha, l.N, l.pitch, nullptr,
                           nullptr, nullptr, nullptr, l.loT, l.hiT, l.ub);
      break;
    case PKernel::XFZ: {
#define XFZ(RS, RX, VI, PB_, SC)                                                                                                     \
  do {                                                                                                                             \
    if (seg)                                                                                                                       \
      launch_with_lds<pxfz_rows_kernel<NX, NU, true, RS, RX, VI, PB_, false, true, TILED, SC>>(rgrid, rblock, TILED ? rg.waves_per_block * pxfz_rows_lds_words<NX, NU, true>() * sizeof(double) : 0, l.stream, l.dbuf, l.x0, \
                         l.Ad, l.Bd, l.Kd, l.lo, l.hi, l.z, l.y, l.v, l.w, l.part, l.alpha, l.N, l.pitch, l.Psd, l.seg_start,      \
                         l.tin, l.xin, l.loT, l.hiT, l.ub);                                                                                          \
    else                                                                                                                           \
      launch_with_lds<pxfz_rows_kernel<NX, NU, true, RS, RX, VI, PB_, false, false, TILED, SC>>(rgrid, rblock, TILED ? rg.waves_per_block * pxfz_rows_lds_words<NX, NU, false>() * sizeof(double) : 0, l.stream, l.dbuf,      \
                         l.x0, l.Ad, l.Bd, l.Kd, l.lo, l.hi, l.z, l.y, l.v, l.w, l.part, l.alpha, l.N, l.pitch, nullptr, nullptr,  \
                         nullptr, nullptr, l.loT, l.hiT, l.ub);                                                                                      \
  } while (0)
#define XFZ4(RS, RX, VI, PB_) do { if constexpr ((RS || VI) && TILED) { if (l.has_soc) XFZ(RS, RX, VI, PB_, true); else XFZ(RS, RX, VI, PB_, false); } else XFZ(RS, RX, VI, PB_, false); } while (0)
#define XFZ3(RS, RX, VI) do { if (l.pbounds) XFZ4(RS, RX, VI, true); else XFZ4(RS, RX, VI, false); } while (0)
#define XFZ2(RS, RX) do { if (l.vform) XFZ3(RS, RX, true); else XFZ3(RS, RX, false); } while (0)
      if (l.resid) { if (relax) XFZ2(true, true); else XFZ2(true, false); }
      else         { if (relax) XFZ2(false, true); else XFZ2(false, false); }
#undef XFZ2
#undef XFZ3
#undef XFZ4
#undef XFZ
      break;
    }
  }
}

template <int NX, int NU>
void launch_dim(const PLaunch& l, PKernel k) {
  const bool seg = l.S > 1;
  const dim3 grid((l.pitch + PI_THREADS - 1) / PI_THREADS), block(PI_THREADS);
  const dim3 sgrid((l.pitch + PI_THREADS - 1) / PI_THREADS, l.S > 0 ? l.S : 1);     // one wave per (64 QPs, segment)
  const RowsGrid<NX> rg(l);                                                          // rows over lanes: QPW QPs per wave
  const dim3 rgrid = rg.grid, rblock = rg.block;
  const bool relax = l.alpha != 1.0;
  switch (k) {
    case PKernel::SEGMENTS:
      hipLaunchKernelGGL((pseg_kernel<NX, NU>), sgrid, block, 0, l.stream, l.Ad, l.Bd, l.Kd, l.Sd, l.seg_start, l.todo, l.Omd,
                         l.Psd, l.Segd, l.grow, l.pitch, l.batch, l.qflag);
      break;
    case PKernel::SCAN:
      hipLaunchKernelGGL((pscan_kernel<NX>), dim3(l.pitch / PscanShape<NX>::QPW), block, 0, l.stream, l.Segd, l.tseg, l.eseg, l.x0,
                         l.tin, l.xin, l.S, l.pitch);
      break;
    case PKernel::FACTOR:
      hipLaunchKernelGGL((pfactor_kernel<NX, NU>), grid, block, 0, l.stream, l.Ad, l.Bd, l.Q, l.R, l.QN, l.rhov, l.todo, l.Kd, l.Sd,
                         l.fail, l.N, l.pitch, l.batch, l.qflag);
      break;
    case PKernel::XB: {
#define XB(HQ, VF, PB_)                                                                                                          \
  do {                                                                                                                           \
    if (l.rows && seg)                                                                                                           \
      hipLaunchKernelGGL((pxb_rows_kernel<NX, NU, HQ, VF, PB_, true>), rgrid, rblock, 0, l.stream, l.vform ? l.v : l.z, l.y, l.q,  \
                         l.Ad, l.Bd, l.Kd, l.Sd, l.lo, l.hi, l.dbuf, l.rhov, l.N, l.pitch, l.Omd, l.seg_start, l.tseg, l.eseg, nullptr, nullptr, nullptr);   \
    else if (l.rows)                                                                                                             \
      hipLaunchKernelGGL((pxb_rows_kernel<NX, NU, HQ, VF, PB_, false>), rgrid, rblock, 0, l.stream, l.vform ? l.v : l.z, l.y, l.q, \
                         l.Ad, l.Bd, l.Kd, l.Sd, l.lo, l.hi, l.dbuf, l.rhov, l.N, l.pitch, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);   \
    else if (l.has_soc && (VF) && seg)                                                                                           \
      hipLaunchKernelGGL((pxb_kernel<NX, NU, HQ, VF, PB_, true, (VF)>), sgrid, block, 0, l.stream, l.vform ? l.v : l.z, l.y, l.q, \
                         l.Ad, l.Bd, l.Kd, l.Sd, l.lo, l.hi, l.dbuf, l.rhov, l.N, l.pitch, l.Omd, l.seg_start, l.tseg, l.eseg,    \
                         l.ub);                                                                                                  \
    else if (l.has_soc && (VF))                                                                                                  \
      hipLaunchKernelGGL((pxb_kernel<NX, NU, HQ, VF, PB_, false, (VF)>), grid, block, 0, l.stream, l.vform ? l.v : l.z, l.y, l.q, \
                         l.Ad, l.Bd, l.Kd, l.Sd, l.lo, l.hi, l.dbuf, l.rhov, l.N, l.pitch, nullptr, nullptr, nullptr, nullptr,    \
                         l.ub);                                                                                                  \
    else if (seg)                                                                                                                \
      hipLaunchKernelGGL((pxb_kernel<NX, NU, HQ, VF, PB_, true>), sgrid, block, 0, l.stream, l.vform ? l.v : l.z, l.y, l.q, l.Ad, \
                         l.Bd, l.Kd, l.Sd, l.lo, l.hi, l.dbuf, l.rhov, l.N, l.pitch, l.Omd, l.seg_start, l.tseg, l.eseg);         \
    else                                                                                                                         \
      hipLaunchKernelGGL((pxb_kernel<NX, NU, HQ, VF, PB_, false>), grid, block, 0, l.stream, l.vform ? l.v : l.z, l.y, l.q, l.Ad, \
                         l.Bd, l.Kd, l.Sd, l.lo, l.hi, l.dbuf, l.rhov, l.N, l.pitch, nullptr, nullptr, nullptr, nullptr, nullptr);          \
  } while (0)
#define XB2(HQ, VF) do { if (l.pbounds) XB(HQ, VF, true); else XB(HQ, VF, false); } while (0)
      if (l.has_q) { if (l.vform) XB2(true, true); else XB2(true, false); }
      else         { if (l.vform) XB2(false, true); else XB2(false, false); }
#undef XB2
#undef XB
      break;
    }
    case PKernel::XF:      // read-out: w of the last x-update
      if (l.rows && seg)
        hipLaunchKernelGGL((pxfz_rows_kernel<NX, NU, false, false, false, false, false, true, true>), rgrid, rblock, 0, l.stream,
                           l.dbuf, l.x0, l.Ad, l.Bd, l.Kd, l.lo, l.hi, l.z, l.y, l.v, l.w, l.part, l.alpha, l.N, l.pitch, l.Psd,
                           l.seg_start, l.tin, l.xin, nullptr, nullptr, nullptr);
      else if (l.rows)
        hipLaunchKernelGGL((pxfz_rows_kernel<NX, NU, false, false, false, false, false, true, false>), rgrid, rblock, 0, l.stream,
                           l.dbuf, l.x0, l.Ad, l.Bd, l.Kd, l.lo, l.hi, l.z, l.y, l.v, l.w, l.part, l.alpha, l.N, l.pitch, nullptr,
                           nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
      else if (seg)
        hipLaunchKernelGGL((pxfz_kernel<NX, NU, false, false, false, false, false, true, true>), sgrid, block, 0, l.stream, l.dbuf,
                           l.x0, l.Ad, l.Bd, l.Kd, l.lo, l.hi, l.z, l.y, l.v, l.w, l.part, l.alpha, l.N, l.pitch, l.Psd,
                           l.seg_start, l.tin, l.xin);
      else
        hipLaunchKernelGGL((pxfz_kernel<NX, NU, false, false, false, false, false, true>), grid, block, 0, l.stream, l.dbuf,
                           l.x0, l.Ad, l.Bd, l.Kd, l.lo, l.hi, l.z, l.y, l.v, l.w, l.part, l.alpha, l.N, l.pitch, nullptr,
                           nullptr, nullptr, nullptr, nullptr);
      break;
    case PKernel::XFZ: {
#define XFZ(RS, RX, VI, PB_)                                                                                                       \
  do {                                                                                                                             \
    if (l.rows && seg)                                                                                                             \
      hipLaunchKernelGGL((pxfz_rows_kernel<NX, NU, true, RS, RX, VI, PB_, false, true>), rgrid, rblock, 0, l.stream, l.dbuf, l.x0,  \
                         l.Ad, l.Bd, l.Kd, l.lo, l.hi, l.z, l.y, l.v, l.w, l.part, l.alpha, l.N, l.pitch, l.Psd, l.seg_start,      \
                         l.tin, l.xin, nullptr, nullptr, nullptr);                                                                                            \
    else if (l.rows)                                                                                                               \
      hipLaunchKernelGGL((pxfz_rows_kernel<NX, NU, true, RS, RX, VI, PB_, false, false>), rgrid, rblock, 0, l.stream, l.dbuf, l.x0, \
                         l.Ad, l.Bd, l.Kd, l.lo, l.hi, l.z, l.y, l.v, l.w, l.part, l.alpha, l.N, l.pitch, nullptr, nullptr,        \
                         nullptr, nullptr, nullptr, nullptr, nullptr);                                                                                        \
    else if (l.has_soc && seg)                                                                                                     \
      hipLaunchKernelGGL((pxfz_kernel<NX, NU, true, RS, RX, VI, PB_, false, true, true>), sgrid, block, 0, l.stream, l.dbuf, l.x0, \
                         l.Ad, l.Bd, l.Kd, l.lo, l.hi, l.z, l.y, l.v, l.w, l.part, l.alpha, l.N, l.pitch, l.Psd, l.seg_start,      \
                         l.tin, l.xin, l.ub);                                                                                      \
    else if (l.has_soc)                                                                                                            \
      hipLaunchKernelGGL((pxfz_kernel<NX, NU, true, RS, RX, VI, PB_, false, false, true>), grid, block, 0, l.stream, l.dbuf, l.x0, \
                         l.Ad, l.Bd, l.Kd, l.lo, l.hi, l.z, l.y, l.v, l.w, l.part, l.alpha, l.N, l.pitch, nullptr, nullptr,        \
                         nullptr, nullptr, l.ub);                                                                                  \
    else if (seg)                                                                                                                  \
      hipLaunchKernelGGL((pxfz_kernel<NX, NU, true, RS, RX, VI, PB_, false, true>), sgrid, block, 0, l.stream, l.dbuf, l.x0, l.Ad, \
                         l.Bd, l.Kd, l.lo, l.hi, l.z, l.y, l.v, l.w, l.part, l.alpha, l.N, l.pitch, l.Psd, l.seg_start, l.tin,    \
                         l.xin);                                                                                                   \
    else                                                                                                                           \
      hipLaunchKernelGGL((pxfz_kernel<NX, NU, true, RS, RX, VI, PB_, false>), grid, block, 0, l.stream, l.dbuf, l.x0, l.Ad,        \
                         l.Bd, l.Kd, l.lo, l.hi, l.z, l.y, l.v, l.w, l.part, l.alpha, l.N, l.pitch, nullptr, nullptr, nullptr,     \
                         nullptr);                                                                                                 \
  } while (0)
#define XFZ3(RS, RX, VI) do { if (l.pbounds) XFZ(RS, RX, VI, true); else XFZ(RS, RX, VI, false); } while (0)
#define XFZ2(RS, RX) do { if (l.vform) XFZ3(RS, RX, true); else XFZ3(RS, RX, false); } while (0)
      if (l.resid) { if (relax) XFZ2(true, true); else XFZ2(true, false); }
      else         { if (relax) XFZ2(false, true); else XFZ2(false, false); }
#undef XFZ2
#undef XFZ3
#undef XFZ
      break;
    }
  }
}

}  // namespace

}  // namespace admm

// bool launch_pinst_<group>(l, k, query_only) over the X(n, m) list DIMS, and the group's " (n,m) ..." string
#define ADMM_PINST_GROUP(NAME, DIMS)                                        \
  namespace admm {                                                          \
  bool launch_pinst_##NAME(const PLaunch& l, PKernel k, bool query_only) {  \
    DIMS(ADMM_PINST_TRY)                                                    \
    return false;                                                           \
  }                                                                         \
  const char* dims_pinst_##NAME() { return DIMS(ADMM_PINST_NAME); }         \
  }
#define ADMM_PINST_GROUP_WIDE(NAME, DIMS)                                    \
  namespace admm {                                                          \
  bool launch_pinst_##NAME(const PLaunch& l, PKernel k, bool query_only) {  \
    DIMS(ADMM_PINST_TRY_WIDE)                                               \
    return false;                                                           \
  }                                                                         \
  const char* dims_pinst_##NAME() { return DIMS(ADMM_PINST_NAME); }         \
  }
#define ADMM_PINST_TRY_WIDE(NX, NU)                 \
  if (l.n == NX && l.m == NU) {                     \
    if (!query_only) launch_dim_wide<NX, NU, true>(l, k); \
    return true;                                    \
  }
#define ADMM_PINST_TRY(NX, NU)                 \
  if (l.n == NX && l.m == NU) {                \
    if (!query_only) launch_dim<NX, NU>(l, k); \
    return true;                               \
  }
#define ADMM_PINST_STR2(x) #x
#define ADMM_PINST_STR(x) ADMM_PINST_STR2(x)
#define ADMM_PINST_NAME(NX, NU) "(" ADMM_PINST_STR(NX) "," ADMM_PINST_STR(NU) ") "

// Per-instance dynamics (admm_pinst.hpp): instantiations and launcher.  Adding a shape = adding X(n, m) to one of the two
// groups (this file, admm_pinst_g1.hip); both are compiled in parallel by build().
#include <algorithm>
#include <string>

#include "admm_pinst_launch.hpp"

#define ADMM_PINST_DIMS_G0(X) X(6, 3) X(2, 1) X(4, 2) X(3, 2)
ADMM_PINST_GROUP(g0, ADMM_PINST_DIMS_G0)

namespace admm {

bool launch_pinst_g1(const PLaunch& l, PKernel k, bool query_only);
bool launch_pinst_g2(const PLaunch& l, PKernel k, bool query_only);
void launch_pinst_rows_twin_6_3(const PLaunch& l, PKernel k);
const char* dims_pinst_g1();
const char* dims_pinst_g2();

bool launch_pinst(const PLaunch& l, PKernel k, bool query_only) {
  // (6, 3) with l.rows: factorisation and transfer matrices in their rows-over-lanes form too (ADMM_PI_ROWS_FACTOR=1: the
  // twin of the wide shapes' kernels at a shape where the one-lane kernels exist)
  if (l.rows_factor && l.n == 6 && l.m == 3 && (k == PKernel::FACTOR || k == PKernel::SEGMENTS)) {
    if (!query_only) launch_pinst_rows_twin_6_3(l, k);
    return true;
  }
  return launch_pinst_g0(l, k, query_only) || launch_pinst_g1(l, k, query_only) || launch_pinst_g2(l, k, query_only);
}

// the wide shapes exist in the rows-over-lanes form only
bool pinst_rows_only(int n, int m) {
  PLaunch l{};
  l.n = n; l.m = m;
  return launch_pinst_g2(l, PKernel::XB, true);
}

const char* dims_pinst() {
  static const std::string all = std::string(dims_pinst_g0()) + dims_pinst_g1() + dims_pinst_g2();
  return all.c_str();
}

void launch_padapt(hipStream_t stream, const double* resid, const int* status, double* rhov, int* nupd, int* todo,
                   double* cscale, int* nchanged, double mu2, double tau, int adapt_max, int pitch, int batch, double* rho_prev) {
  hipLaunchKernelGGL(padapt_kernel, dim3((pitch + 255) / 256), dim3(256), 0, stream, resid, status, rhov, nupd, todo, cscale,
                     nchanged, mu2, tau, adapt_max, pitch, batch, rho_prev);
}

void launch_padapt_veto(hipStream_t stream, const int* qflag, const double* rho_prev, double* rhov, int* nupd, int* todo,
                        double* cscale, int* nchanged, int* nveto, int adapt_max, int pitch) {
  hipLaunchKernelGGL(padapt_veto_kernel, dim3((pitch + 255) / 256), dim3(256), 0, stream, qflag, rho_prev, rhov, nupd, todo, cscale,
                     nchanged, nveto, adapt_max, pitch);
}

void launch_padapt_scale(hipStream_t stream, double* y, const double* cscale, const int* todo, int rows, int pitch) {
  hipLaunchKernelGGL(padapt_scale_kernel, dim3((pitch + 255) / 256, 64), dim3(256), 0, stream, y, cscale, todo, rows, pitch);
}

void launch_pv_to_zy_soc(hipStream_t stream, const double* v, double* z, double* y, const double* lo, const double* hi,
                         const double* ub, int N, int nb, int m, int pitch) {
  hipLaunchKernelGGL(pv_to_zy_soc_kernel, dim3((pitch + 255) / 256, 64), dim3(256), 0, stream, v, z, y, lo, hi, ub, N, nb, m, pitch);
}

void launch_to_tiled(hipStream_t stream, const double* src, double* dst, int batch, int N, int E, int n, int pitch, int nr, int nc) {
  const int qpw = n <= 2 ? 32 : (n <= 4 ? 16 : (n <= 8 ? 8 : 4));                 // PscanShape<n>::QPW
  const size_t tiles = (size_t)N * (pitch / qpw);
  hipLaunchKernelGGL(to_tiled_kernel, dim3((unsigned)std::min<size_t>(tiles, 1u << 20)), dim3(256), 0, stream, src, dst, batch, N, E, qpw, pitch, nr, nc);
}

void launch_pv_to_zy(hipStream_t stream, const double* v, double* z, double* y, const double* lo, const double* hi, size_t count) {
  hipLaunchKernelGGL(pv_to_zy_kernel, dim3(2048), dim3(256), 0, stream, v, z, y, lo, hi, count);
}

}  // namespace admm

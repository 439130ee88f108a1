// Per-instance dynamics (admm_pinst.hpp): instantiations and launcher.  Adding a shape = adding X(n, m) to one of the two
// groups (this file, admm_pinst_g1.hip); both are compiled in parallel by build().
#include <string>

#include "admm_pinst_launch.hpp"

#define ADMM_PINST_DIMS_G0(X) X(6, 3) X(2, 1) X(4, 2) X(3, 2)
ADMM_PINST_GROUP(g0, ADMM_PINST_DIMS_G0)

namespace admm {

bool launch_pinst_g1(const PLaunch& l, PKernel k, bool query_only);
const char* dims_pinst_g1();

bool launch_pinst(const PLaunch& l, PKernel k, bool query_only) {
  return launch_pinst_g0(l, k, query_only) || launch_pinst_g1(l, k, query_only);
}

const char* dims_pinst() {
  static const std::string all = std::string(dims_pinst_g0()) + dims_pinst_g1();
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

void launch_pv_to_zy(hipStream_t stream, const double* v, double* z, double* y, const double* lo, const double* hi, size_t count) {
  hipLaunchKernelGGL(pv_to_zy_kernel, dim3(2048), dim3(256), 0, stream, v, z, y, lo, hi, count);
}

}  // namespace admm

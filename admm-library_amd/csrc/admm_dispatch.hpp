// admm_dispatch.hpp -- boundary between the solver runtime (admm_api.hip) and the
// dimension-templated kernels.  The (n, m) instantiations are split over several translation
// units (admm_dims_g*.hip) so that they compile in parallel; each exports one launcher.
#pragma once

#include <hip/hip_runtime.h>

namespace admm {

struct XLaunch {
  hipStream_t stream;
  int n, m, S, pitch;
  int batch;                    // QPs really present (columns batch .. pitch-1 are padding)
  int xfree;                    // every state row is unbounded at every stage: z = v, y = 0 there (XFREE kernel forms):
                                // 1 = their v is not read, 2 = nor written (the next iteration does not read it either)
  bool has_q;
  bool has_soc;                 // thrust-magnitude bound on some stage: SOC kernel forms
  double rho, alpha;
  const double *z, *y, *q;      // state in (z, y) form and the linear term
  double *v, *w;                // state in v-form; materialised w
  const double *recB, *recF, *recS;
  const double *recFE, *recBE;  // records of the alternating-direction kernels (admm_kernels_alt.hpp)
  double* mvec;                 // the forward elimination's feed-forward rows db (m rows per stage)
  const int* seg_start;
  double *dbuf, *tseg, *eseg, *tin, *xin, *part;
  const double* x0;
  int nsplit;                   // split-K slabs of the scan output (t_in / x_in), 1 = none
  size_t split_stride;          // elements between slabs
  // MFMA form (admm_mfma.hpp): per-stage fragment records; mode 1 = mixed, 2 = fp64, 0 = not in use
  const unsigned char *recMF, *recMB;
  int mfma_mode;
};

enum class XKernel { XB, XF, XFZ, XSCAN_CHAIN, XFZE, XBZE };

// Each returns true if (n, m) is compiled in that group (and, unless query_only, the kernel
// was enqueued on l.stream).  a = VFORM (XB) / VIN (XFZ); b = RESID (XFZ, XFZE, XBZE).
// XFZE / XBZE exist only for the pairs alt_dims() accepts: false otherwise.
bool launch_group0(const XLaunch& l, XKernel k, bool a, bool b, bool query_only);
bool launch_group1(const XLaunch& l, XKernel k, bool a, bool b, bool query_only);
bool launch_group2(const XLaunch& l, XKernel k, bool a, bool b, bool query_only);
bool launch_group3(const XLaunch& l, XKernel k, bool a, bool b, bool query_only);
// MFMA form of XFZE / XBZE (modes 1, 2) and of XB / XFZ in v-form (mode 1), by l.mfma_mode; false if the
// (n, m) pair or the kernel has no MFMA instantiation.
bool launch_mfma(const XLaunch& l, XKernel k, bool resid, bool query_only);
const char* dims_mfma();
// ---- per-instance dynamics (admm_pinst.hpp): everything per QP, batch-minor; one lane sweeps the whole horizon ----
struct PLaunch {
  hipStream_t stream;
  int n, m, N, pitch, batch;
  bool has_q, vform, resid, pbounds;   // vform: state read from v; pbounds: lo / hi per instance ([k][n+m][pitch])
  double alpha;
  const double* rhov;                  // rho per QP [pitch] (all equal unless the per-QP adaptive rule has moved them)
  const int* todo;                     // FACTOR only: refactor the marked QPs (NULL = all)
  const double *Ad, *Bd, *Q, *R, *QN;  // Q, R, QN: shared, row-major, on the device
  double *Kd, *Sd;
  int* fail;
  int* qflag;                          // FACTOR / SEGMENTS trial runs: per-QP verdict [pitch] (bit 0: S_k not PD, bit 1: conditioning bound), or NULL
  const double *lo, *hi;
  const double *loT, *hiT;             // wide shapes with per-instance bounds: the box in the tiled layout, [k][g][n + m][c] (else NULL)
  const double *z, *y, *q, *x0;
  double *v, *w, *dbuf, *part;
  // segments in time (S > 1; admm_pinst.hpp, pseg_kernel): per-QP transfer matrices and the segment vectors
  int S;
  const int* seg_start;                // [S + 1], device
  double *Omd, *Psd, *Segd;            // Omega_k [N][n*m], Psi_k [N][m*n], (Phi, Xi, Th) [S][3][n*n], all x pitch
  double *tseg, *eseg, *tin, *xin;     // [S][n][pitch]
  int* grow;                           // SEGMENTS: set if a transfer matrix exceeds the conditioning bound
  bool rows;                           // small batches: the sweeps with a QP's rows spread over lanes (admm_pinst_rows.hpp)
  bool rows_factor;                    // (6, 3) only: FACTOR / SEGMENTS through the wide shapes' kernels (admm_pinst_wide.hpp)
  bool has_soc;                        // thrust-magnitude bound ub [N] on the control rows (one-lane kernels; rows-over-lanes kernels of the wide shapes)
  const double* ub;
};
enum class PKernel { FACTOR, XB, XF, XFZ, SEGMENTS, SCAN };
bool launch_pinst(const PLaunch& l, PKernel k, bool query_only);
size_t pinst_wide_lds_bytes(int n, int m);   // LDS per workgroup of the wide shapes' sweeps (0: not a wide shape)
bool pinst_rows_only(int n, int m);   // a wide shape: rows-over-lanes kernels whatever the batch
void launch_padapt(hipStream_t stream, const double* resid, const int* status, double* rhov, int* nupd, int* todo,
                   double* cscale, int* nchanged, double mu2, double tau, int adapt_max, int pitch, int batch, double* rho_prev);
void launch_padapt_veto(hipStream_t stream, const int* qflag, const double* rho_prev, double* rhov, int* nupd, int* todo,
                        double* cscale, int* nchanged, int* nveto, int adapt_max, int pitch);
void launch_padapt_scale(hipStream_t stream, double* y, const double* cscale, const int* todo, int rows, int pitch);
void launch_pv_to_zy(hipStream_t stream, const double* v, double* z, double* y, const double* lo, const double* hi, size_t count);
void launch_pv_to_zy_soc(hipStream_t stream, const double* v, double* z, double* y, const double* lo, const double* hi,
                         const double* ub, int N, int nb, int m, int pitch);
// QP-major staged operand (batch x N x E) -> the wide shapes' tiled layout (admm_pinst.hpp, Operand / to_tiled_kernel)
// (nr x nc > 0: the source blocks are row-major, ADMM_FLAG_ROW_MAJOR)
void launch_to_tiled(hipStream_t stream, const double* src, double* dst, int batch, int N, int E, int n, int pitch, int nr, int nc);
const char* dims_pinst();
// " (n,m) (n,m) ..." of a group, for error messages
const char* dims_group0();
const char* dims_group1();
const char* dims_group2();
const char* dims_group3();

}  // namespace admm

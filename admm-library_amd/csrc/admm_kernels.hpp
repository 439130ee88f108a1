// admm_kernels.hpp -- hand-written HIP kernels (gfx950 / CDNA4) of the ADMM
// inner loop.  DESIGN.md §4 describes the data layout and each kernel's
// roofline.  No reference counterpart exists (README.md:1-2 only).
//
// Device layout ("batch-minor"): every per-QP vector is stored transposed,
//     v[row][col],  row = stacked index e in [0, L),  col = QP b in [0, pitch)
// with pitch = batch rounded up to 64.  One lane owns one QP (x-update) or two
// adjacent QPs (z/dual kernel), so every wave-level access is one contiguous
// 512-B / 1-KiB segment of a row, and all per-stage matrices and bounds are
// wave-uniform: they travel through the scalar unit (s_load -> SGPR operand of
// v_fma_f64) and cost no VGPRs and no LDS bandwidth.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace admm {

constexpr int XB_THREADS = 256;   // x-update workgroup: 4 waves = 256 QPs of one segment
constexpr int Z_THREADS = 256;    // z/dual workgroup: 256 lanes x 2 QPs = 512 columns
constexpr int T_TILE = 32;        // transpose tile

// Per-stage matrices are wave-uniform; reading them through the constant
// address space makes hipcc select scalar (SMEM) loads for them.
typedef const __attribute__((address_space(4))) double* cdouble_p;
typedef const __attribute__((address_space(4))) int* cint_p;

__device__ __forceinline__ cdouble_p as_const(const double* p) {
  return (cdouble_p)(uintptr_t)p;
}
__device__ __forceinline__ cint_p as_const(const int* p) { return (cint_p)(uintptr_t)p; }

// ---------------------------------------------------------------------------
// x-update, backward sweep (segment-local).  One lane = one QP, blockIdx.y =
// segment.  For stages k = b-1 .. a of the segment, with tail t = 0 on entry:
//     g    = q - rho (z - y)                    (block k: g^u (m), g^x (n))
//     p    = g^x + t
//     h    = B_k' p + g^u
//     d0_k = Si_k h                    -> dbuf  (m rows per stage)
//     t    = A_k' p - K_k' h
//     e   += Omega_k d0_k
// and on exit t -> tseg[s], e -> eseg[s] (n rows each).
// HBM per stacked element: reads z, y (16 B, +8 with q), writes d (8 m/(n+m)).
// ---------------------------------------------------------------------------
template <int NX, int NU, bool HASQ>
__global__ __launch_bounds__(XB_THREADS) void xb_kernel(
    const double* __restrict__ z, const double* __restrict__ y, const double* __restrict__ q,
    const double* __restrict__ recB_, const int* __restrict__ seg_start_,
    double* __restrict__ dbuf, double* __restrict__ tseg, double* __restrict__ eseg,
    double rho, int pitch) {
  constexpr int NB = NX + NU;
  constexpr int RB = NX * NX + NU * NX + NU * NU + NX * NU + NX * NU;
  constexpr int O_AT = 0, O_BT = NX * NX, O_SI = O_BT + NU * NX, O_KT = O_SI + NU * NU, O_OM = O_KT + NX * NU;
  const int col = blockIdx.x * XB_THREADS + threadIdx.x;
  const int s = blockIdx.y;
  if (col >= pitch) return;
  cint_p seg_start = as_const(seg_start_);
  const int k0 = seg_start[s], k1 = seg_start[s + 1];
  const size_t P = (size_t)pitch;

  double t[NX], e[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) { t[i] = 0.0; e[i] = 0.0; }

  double lz[NB], ly[NB], lq[NB];
  {
    const size_t r0 = (size_t)(k1 - 1) * NB * P + col;
#pragma unroll
    for (int r = 0; r < NB; ++r) {
      lz[r] = z[r0 + r * P];
      ly[r] = y[r0 + r * P];
      if (HASQ) lq[r] = q[r0 + r * P];
    }
  }

  for (int k = k1 - 1; k >= k0; --k) {
    double g[NB];
#pragma unroll
    for (int r = 0; r < NB; ++r) {
      g[r] = -rho * (lz[r] - ly[r]);
      if (HASQ) g[r] += lq[r];
    }
    {  // prefetch the next (earlier) stage; clamped so the last one is a harmless re-read
      const int kn = (k > k0) ? k - 1 : k0;
      const size_t r0 = (size_t)kn * NB * P + col;
#pragma unroll
      for (int r = 0; r < NB; ++r) {
        lz[r] = z[r0 + r * P];
        ly[r] = y[r0 + r * P];
        if (HASQ) lq[r] = q[r0 + r * P];
      }
    }
    cdouble_p rb = as_const(recB_) + (size_t)k * RB;
    double p[NX], h[NU], d[NU];
#pragma unroll
    for (int i = 0; i < NX; ++i) p[i] = g[NU + i] + t[i];
#pragma unroll
    for (int j = 0; j < NU; ++j) {
      double a = g[j];
#pragma unroll
      for (int i = 0; i < NX; ++i) a = fma(rb[O_BT + j * NX + i], p[i], a);
      h[j] = a;
    }
#pragma unroll
    for (int j = 0; j < NU; ++j) {
      double a = 0.0;
#pragma unroll
      for (int l = 0; l < NU; ++l) a = fma(rb[O_SI + j * NU + l], h[l], a);
      d[j] = a;
    }
    {
      const size_t d0 = (size_t)k * NU * P + col;
#pragma unroll
      for (int j = 0; j < NU; ++j) dbuf[d0 + j * P] = d[j];
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      double a = 0.0;
#pragma unroll
      for (int l = 0; l < NX; ++l) a = fma(rb[O_AT + i * NX + l], p[l], a);
#pragma unroll
      for (int j = 0; j < NU; ++j) a = fma(-rb[O_KT + i * NU + j], h[j], a);
      t[i] = a;
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      double a = e[i];
#pragma unroll
      for (int j = 0; j < NU; ++j) a = fma(rb[O_OM + i * NU + j], d[j], a);
      e[i] = a;
    }
  }
  const size_t o = (size_t)s * NX * P + col;
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    tseg[o + i * P] = t[i];
    eseg[o + i * P] = e[i];
  }
}

// ---------------------------------------------------------------------------
// Segment scan.  One lane = one QP, all segments, sequential in S (S is small):
//     t_in(S-1) = 0;   t_in(s-1) = tseg(s) + Phi_s t_in(s)
//     x_in(0)  = x0;   x_in(s+1) = eseg(s) + Xi_s t_in(s) + Th_s x_in(s)
// Touches only 4 S n rows: negligible HBM traffic; latency-bound by design.
// ---------------------------------------------------------------------------
template <int NX>
__global__ __launch_bounds__(XB_THREADS) void xscan_kernel(
    const double* __restrict__ tseg, const double* __restrict__ eseg, const double* __restrict__ x0,
    const double* __restrict__ recS_, double* __restrict__ tin, double* __restrict__ xin,
    int S, int pitch) {
  constexpr int RS = 3 * NX * NX;
  const int col = blockIdx.x * XB_THREADS + threadIdx.x;
  if (col >= pitch) return;
  const size_t P = (size_t)pitch;
  double t[NX], x[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) t[i] = 0.0;
  for (int s = S - 1; s >= 0; --s) {
    const size_t o = (size_t)s * NX * P + col;
#pragma unroll
    for (int i = 0; i < NX; ++i) tin[o + i * P] = t[i];
    if (s == 0) break;
    cdouble_p rs = as_const(recS_) + (size_t)s * RS;
    double tn[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      double a = tseg[o + i * P];
#pragma unroll
      for (int l = 0; l < NX; ++l) a = fma(rs[i * NX + l], t[l], a);
      tn[i] = a;
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) t[i] = tn[i];
  }
#pragma unroll
  for (int i = 0; i < NX; ++i) x[i] = x0[(size_t)i * P + col];
  for (int s = 0; s < S; ++s) {
    const size_t o = (size_t)s * NX * P + col;
#pragma unroll
    for (int i = 0; i < NX; ++i) xin[o + i * P] = x[i];
    if (s == S - 1) break;
    cdouble_p rs = as_const(recS_) + (size_t)s * RS;
    double ts[NX], xn[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) ts[i] = tin[o + i * P];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      double a = eseg[o + i * P];
#pragma unroll
      for (int l = 0; l < NX; ++l) a = fma(rs[NX * NX + i * NX + l], ts[l], a);
#pragma unroll
      for (int l = 0; l < NX; ++l) a = fma(rs[2 * NX * NX + i * NX + l], x[l], a);
      xn[i] = a;
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = xn[i];
  }
}

// ---------------------------------------------------------------------------
// x-update, forward rollout (segment-local, exact once t_in/x_in are known).
//     d   = d0_k + Psi_k t_in
//     u   = -K_k x - d
//     x   = A_k x + B_k u
//     w block k = (u, x)
// HBM per stacked element: reads d (8 m/(n+m)), writes w (8 B).
// ---------------------------------------------------------------------------
template <int NX, int NU>
__global__ __launch_bounds__(XB_THREADS) void xf_kernel(
    const double* __restrict__ dbuf, const double* __restrict__ tin, const double* __restrict__ xin,
    const double* __restrict__ recF_, const int* __restrict__ seg_start_,
    double* __restrict__ w, int pitch) {
  constexpr int NB = NX + NU;
  constexpr int RF = NU * NX + NU * NX + NX * NX + NX * NU;
  constexpr int O_PSI = 0, O_K = NU * NX, O_A = O_K + NU * NX, O_B = O_A + NX * NX;
  const int col = blockIdx.x * XB_THREADS + threadIdx.x;
  const int s = blockIdx.y;
  if (col >= pitch) return;
  cint_p seg_start = as_const(seg_start_);
  const int k0 = seg_start[s], k1 = seg_start[s + 1];
  const size_t P = (size_t)pitch;
  double t[NX], x[NX];
  {
    const size_t o = (size_t)s * NX * P + col;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      t[i] = tin[o + i * P];
      x[i] = xin[o + i * P];
    }
  }
  double ld[NU];
  {
    const size_t d0 = (size_t)k0 * NU * P + col;
#pragma unroll
    for (int j = 0; j < NU; ++j) ld[j] = dbuf[d0 + j * P];
  }
  for (int k = k0; k < k1; ++k) {
    double d[NU];
#pragma unroll
    for (int j = 0; j < NU; ++j) d[j] = ld[j];
    {
      const int kn = (k + 1 < k1) ? k + 1 : k;
      const size_t d0 = (size_t)kn * NU * P + col;
#pragma unroll
      for (int j = 0; j < NU; ++j) ld[j] = dbuf[d0 + j * P];
    }
    cdouble_p rf = as_const(recF_) + (size_t)k * RF;
    double u[NU], xn[NX];
#pragma unroll
    for (int j = 0; j < NU; ++j) {
      double a = d[j];
#pragma unroll
      for (int i = 0; i < NX; ++i) a = fma(rf[O_PSI + j * NX + i], t[i], a);
#pragma unroll
      for (int i = 0; i < NX; ++i) a = fma(rf[O_K + j * NX + i], x[i], a);
      u[j] = -a;
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      double a = 0.0;
#pragma unroll
      for (int l = 0; l < NX; ++l) a = fma(rf[O_A + i * NX + l], x[l], a);
#pragma unroll
      for (int j = 0; j < NU; ++j) a = fma(rf[O_B + i * NU + j], u[j], a);
      xn[i] = a;
    }
    const size_t r0 = (size_t)k * NB * P + col;
#pragma unroll
    for (int j = 0; j < NU; ++j) w[r0 + j * P] = u[j];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      x[i] = xn[i];
      w[r0 + (NU + i) * P] = xn[i];
    }
  }
}

// ---------------------------------------------------------------------------
// Fused z-update + dual ascent + residual partial sums -- the HBM-bound kernel
// the project is graded on.  One lane = two adjacent QPs (16-B accesses), a
// workgroup = 512 columns x `zrows` rows (blockIdx.y = row chunk).
//     wh = alpha w + (1 - alpha) z        (RELAX only)
//     v  = wh + y;  z+ = min(max(v, lo), hi);  y+ = v - z+       (in place)
//     RESID: per-QP partial sums over the chunk's rows of
//            (w - z+)^2, (z+ - z)^2, w^2, z+^2, y+^2  -> part[chunk][5][pitch]
// The row index is wave-uniform, so lo/hi are scalar loads and the per-QP sums
// need no cross-lane step at all: each lane owns its QPs' accumulators.
// Algorithmic HBM bytes per stacked element: RESID/RELAX: 3 reads + 2 writes =
// 40 B; plain: 2 reads + 2 writes = 32 B.
// ---------------------------------------------------------------------------
template <bool RESID, bool RELAX>
__global__ __launch_bounds__(Z_THREADS) void zdual_kernel(
    const double* __restrict__ w, double* __restrict__ z, double* __restrict__ y,
    const double* __restrict__ lo_, const double* __restrict__ hi_,
    double* __restrict__ part, double alpha, int L, int zrows, int pitch) {
  const int col = (blockIdx.x * Z_THREADS + threadIdx.x) * 2;
  if (col >= pitch) return;
  const int chunk = blockIdx.y;
  const int r_begin = chunk * zrows;
  const int r_end = (r_begin + zrows < L) ? r_begin + zrows : L;
  const size_t P = (size_t)pitch;
  cdouble_p lo = as_const(lo_);
  cdouble_p hi = as_const(hi_);
  double2 a_r = {0, 0}, a_s = {0, 0}, a_w = {0, 0}, a_z = {0, 0}, a_y = {0, 0};
  constexpr int U = 4;
  int r = r_begin;
  for (; r + U <= r_end; r += U) {
    double2 wv[U], yv[U], zv[U];
#pragma unroll
    for (int i = 0; i < U; ++i) {
      const size_t o = (size_t)(r + i) * P + col;
      wv[i] = *reinterpret_cast<const double2*>(w + o);
      yv[i] = *reinterpret_cast<const double2*>(y + o);
      if (RESID || RELAX) zv[i] = *reinterpret_cast<const double2*>(z + o);
    }
#pragma unroll
    for (int i = 0; i < U; ++i) {
      const size_t o = (size_t)(r + i) * P + col;
      const double l = lo[r + i], h = hi[r + i];
      double2 wh = wv[i];
      if (RELAX) {
        wh.x = fma(alpha, wv[i].x, (1.0 - alpha) * zv[i].x);
        wh.y = fma(alpha, wv[i].y, (1.0 - alpha) * zv[i].y);
      }
      double2 v, zn, yn;
      v.x = wh.x + yv[i].x;
      v.y = wh.y + yv[i].y;
      zn.x = fmin(fmax(v.x, l), h);
      zn.y = fmin(fmax(v.y, l), h);
      yn.x = v.x - zn.x;
      yn.y = v.y - zn.y;
      *reinterpret_cast<double2*>(z + o) = zn;
      *reinterpret_cast<double2*>(y + o) = yn;
      if (RESID) {
        double dx = wv[i].x - zn.x, dy = wv[i].y - zn.y;
        a_r.x = fma(dx, dx, a_r.x); a_r.y = fma(dy, dy, a_r.y);
        dx = zn.x - zv[i].x; dy = zn.y - zv[i].y;
        a_s.x = fma(dx, dx, a_s.x); a_s.y = fma(dy, dy, a_s.y);
        a_w.x = fma(wv[i].x, wv[i].x, a_w.x); a_w.y = fma(wv[i].y, wv[i].y, a_w.y);
        a_z.x = fma(zn.x, zn.x, a_z.x); a_z.y = fma(zn.y, zn.y, a_z.y);
        a_y.x = fma(yn.x, yn.x, a_y.x); a_y.y = fma(yn.y, yn.y, a_y.y);
      }
    }
  }
  for (; r < r_end; ++r) {
    const size_t o = (size_t)r * P + col;
    const double2 wv = *reinterpret_cast<const double2*>(w + o);
    const double2 yv = *reinterpret_cast<const double2*>(y + o);
    double2 zv = {0, 0};
    if (RESID || RELAX) zv = *reinterpret_cast<const double2*>(z + o);
    const double l = lo[r], h = hi[r];
    double2 wh = wv;
    if (RELAX) {
      wh.x = fma(alpha, wv.x, (1.0 - alpha) * zv.x);
      wh.y = fma(alpha, wv.y, (1.0 - alpha) * zv.y);
    }
    double2 v, zn, yn;
    v.x = wh.x + yv.x;
    v.y = wh.y + yv.y;
    zn.x = fmin(fmax(v.x, l), h);
    zn.y = fmin(fmax(v.y, l), h);
    yn.x = v.x - zn.x;
    yn.y = v.y - zn.y;
    *reinterpret_cast<double2*>(z + o) = zn;
    *reinterpret_cast<double2*>(y + o) = yn;
    if (RESID) {
      double dx = wv.x - zn.x, dy = wv.y - zn.y;
      a_r.x = fma(dx, dx, a_r.x); a_r.y = fma(dy, dy, a_r.y);
      dx = zn.x - zv.x; dy = zn.y - zv.y;
      a_s.x = fma(dx, dx, a_s.x); a_s.y = fma(dy, dy, a_s.y);
      a_w.x = fma(wv.x, wv.x, a_w.x); a_w.y = fma(wv.y, wv.y, a_w.y);
      a_z.x = fma(zn.x, zn.x, a_z.x); a_z.y = fma(zn.y, zn.y, a_z.y);
      a_y.x = fma(yn.x, yn.x, a_y.x); a_y.y = fma(yn.y, yn.y, a_y.y);
    }
  }
  if (RESID) {
    const size_t o = (size_t)chunk * 5 * P + col;
    *reinterpret_cast<double2*>(part + o + 0 * P) = a_r;
    *reinterpret_cast<double2*>(part + o + 1 * P) = a_s;
    *reinterpret_cast<double2*>(part + o + 2 * P) = a_w;
    *reinterpret_cast<double2*>(part + o + 3 * P) = a_z;
    *reinterpret_cast<double2*>(part + o + 4 * P) = a_y;
  }
}

// ---------------------------------------------------------------------------
// Residual finalise + stopping rule.  One lane = one QP: sums the chunk
// partials in chunk order (bitwise reproducible), takes square roots, applies
//   r <= sqrt(L) eps_abs + eps_rel max(|w|, |z|),  s <= sqrt(L) eps_abs + eps_rel rho |y|
// records the first iteration at which the QP met it, and counts converged QPs
// of the real batch (wave-shuffle reduction, one atomic per wave).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resid_finalize_kernel(
    const double* __restrict__ part, double* __restrict__ resid, int* __restrict__ status,
    int* __restrict__ iters, int* __restrict__ nconv, double rho, double eps_abs, double eps_rel,
    double sqrtL, int zchunks, int batch, int pitch, int it) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  int ok_now = 0;
  if (col < pitch) {
    const size_t P = (size_t)pitch;
    double a[5] = {0, 0, 0, 0, 0};
    for (int c = 0; c < zchunks; ++c) {
      const size_t o = (size_t)c * 5 * P + col;
#pragma unroll
      for (int v = 0; v < 5; ++v) a[v] += part[o + v * P];
    }
    const double r = sqrt(a[0]), s = rho * sqrt(a[1]);
    const double nw = sqrt(a[2]), nz = sqrt(a[3]), ny = rho * sqrt(a[4]);
    resid[0 * P + col] = r;
    resid[1 * P + col] = s;
    resid[2 * P + col] = nw;
    resid[3 * P + col] = nz;
    resid[4 * P + col] = ny;
    if (col < batch && it > 0) {
      const double e_pri = sqrtL * eps_abs + eps_rel * fmax(nw, nz);
      const double e_dua = sqrtL * eps_abs + eps_rel * ny;
      int st = status[col];
      if (!st && r <= e_pri && s <= e_dua) {
        st = 1;
        status[col] = 1;
        iters[col] = it;
      }
      ok_now = st;
    }
  }
  if (it > 0) {
    // wave64 reduction of the converged count, then one atomic per wave
    int v = ok_now;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(nconv, v);
  }
}

// ---------------------------------------------------------------------------
// Layout conversion between the ABI's QP-major arrays  src[b][e]  (b < batch,
// e < L) and the device's batch-minor  dst[e][col]  (col < pitch).  32x32 tiles
// through padded LDS so both sides are coalesced.  Setup / read-out only.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(T_TILE * 8) void to_batch_minor_kernel(
    const double* __restrict__ src, double* __restrict__ dst, int batch, int L, int pitch) {
  __shared__ double tile[T_TILE][T_TILE + 1];
  const int e0 = blockIdx.x * T_TILE, b0 = blockIdx.y * T_TILE;
  const int tx = threadIdx.x & (T_TILE - 1), ty = threadIdx.x / T_TILE;
  for (int j = ty; j < T_TILE; j += 8) {
    const int b = b0 + j, e = e0 + tx;
    tile[j][tx] = (b < batch && e < L) ? src[(size_t)b * L + e] : 0.0;
  }
  __syncthreads();
  for (int j = ty; j < T_TILE; j += 8) {
    const int e = e0 + j, b = b0 + tx;
    if (e < L && b < pitch) dst[(size_t)e * pitch + b] = tile[tx][j];
  }
}

__global__ __launch_bounds__(T_TILE * 8) void from_batch_minor_kernel(
    const double* __restrict__ src, double* __restrict__ dst, int batch, int L, int pitch) {
  __shared__ double tile[T_TILE][T_TILE + 1];
  const int e0 = blockIdx.x * T_TILE, b0 = blockIdx.y * T_TILE;
  const int tx = threadIdx.x & (T_TILE - 1), ty = threadIdx.x / T_TILE;
  for (int j = ty; j < T_TILE; j += 8) {
    const int e = e0 + j, b = b0 + tx;
    tile[j][tx] = (e < L && b < pitch) ? src[(size_t)e * pitch + b] : 0.0;
  }
  __syncthreads();
  for (int j = ty; j < T_TILE; j += 8) {
    const int b = b0 + j, e = e0 + tx;
    if (b < batch && e < L) dst[(size_t)b * L + e] = tile[tx][j];
  }
}

}  // namespace admm

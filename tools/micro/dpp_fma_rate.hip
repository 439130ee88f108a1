// Microbenchmark: v_fmac_f64_dpp row_newbcast:k on gfx950 -- an fp64 FMA whose first source is lane k of each 16-lane row,
// broadcast to the row (the only DPP control the fp64 ALU accepts).  Checks the semantics (which lane is read) and measures the
// issue rate against the plain v_fmac_f64, with 8 independent accumulator chains, 1 / 2 / 4 waves per SIMD.
// Build: hipcc -w -O3 --offload-arch=gfx950 dpp_fma_rate.hip -o dpp_fma_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define FMAC_DPP(acc, op, x, K) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(op), "v"(x))
#define FMAC(acc, op, x) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc) : "v"(op), "v"(x))

__global__ void semantics(double* out) {
  double op = (double)threadIdx.x, x = 1.0, a3 = 0.0, a15 = 0.0, a0 = 0.0;
  FMAC_DPP(a3, op, x, 3);
  FMAC_DPP(a15, op, x, 15);
  FMAC_DPP(a0, op, x, 0);
  out[threadIdx.x] = a3;
  out[64 + threadIdx.x] = a15;
  out[128 + threadIdx.x] = a0;
}

template <bool DPP>
__global__ __launch_bounds__(256) void chain(double* out, int iters) {
  double acc[8], op = 1.0 + 1e-9 * (threadIdx.x & 15), x = 1.0000001;
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = threadIdx.x * 1e-3 + c;
  for (int i = 0; i < iters; ++i) {
    if (DPP) {
      FMAC_DPP(acc[0], op, x, 0); FMAC_DPP(acc[1], op, x, 1); FMAC_DPP(acc[2], op, x, 2); FMAC_DPP(acc[3], op, x, 3);
      FMAC_DPP(acc[4], op, x, 4); FMAC_DPP(acc[5], op, x, 5); FMAC_DPP(acc[6], op, x, 6); FMAC_DPP(acc[7], op, x, 7);
      FMAC_DPP(acc[0], op, x, 8); FMAC_DPP(acc[1], op, x, 9); FMAC_DPP(acc[2], op, x, 10); FMAC_DPP(acc[3], op, x, 11);
      FMAC_DPP(acc[4], op, x, 12); FMAC_DPP(acc[5], op, x, 13); FMAC_DPP(acc[6], op, x, 14); FMAC_DPP(acc[7], op, x, 15);
    } else {
      FMAC(acc[0], op, x); FMAC(acc[1], op, x); FMAC(acc[2], op, x); FMAC(acc[3], op, x);
      FMAC(acc[4], op, x); FMAC(acc[5], op, x); FMAC(acc[6], op, x); FMAC(acc[7], op, x);
      FMAC(acc[0], op, x); FMAC(acc[1], op, x); FMAC(acc[2], op, x); FMAC(acc[3], op, x);
      FMAC(acc[4], op, x); FMAC(acc[5], op, x); FMAC(acc[6], op, x); FMAC(acc[7], op, x);
    }
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) s += acc[c];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <bool DPP>
void run(double* d, int wgs) {
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(chain<DPP>, dim3(wgs), dim3(256), 0, 0, d, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(chain<DPP>, dim3(wgs), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double per_wave = (double)iters * 16;
  printf("%s  waves/SIMD %d: %.2f ns per wave-FMA per SIMD (%.1f cycles at 2.4 GHz)\n", DPP ? "v_fmac_f64_dpp row_newbcast" : "v_fmac_f64                 ",
         wgs / 256, ms * 1e6 / per_wave / (wgs / 256), ms * 1e6 / per_wave / (wgs / 256) * 2.4);
}

int main() {
  double* d; hipMalloc(&d, sizeof(double) * 256 * 1024);
  hipLaunchKernelGGL(semantics, dim3(1), dim3(64), 0, 0, d);
  double h[192]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  bool ok = true;
  for (int l = 0; l < 64; ++l) ok = ok && h[l] == (l / 16) * 16 + 3 && h[64 + l] == (l / 16) * 16 + 15 && h[128 + l] == (l / 16) * 16;
  printf("row_newbcast:k reads lane k of the lane's own 16-lane row: %s (lane 37 -> %g, %g, %g)\n", ok ? "yes" : "NO", h[37], h[64 + 37], h[128 + 37]);
  for (int wgs : {256, 512, 1024}) { run<false>(d, wgs); run<true>(d, wgs); }
  return ok ? 0 : 1;
}

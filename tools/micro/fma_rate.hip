// Microbenchmark: issue rate of v_fma_f64 on gfx950 with 1..16 independent accumulator chains per wave,
// one, two and four waves per SIMD (256 / 512 / 1024 workgroups of 256 threads).
// Build: hipcc -w -O3 --offload-arch=gfx950 fma_rate.hip   (DESIGN.md §4.8 quotes the result)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH>
__global__ __launch_bounds__(256) void fma_chain(double* out, double a, double b, int iters) {
  double acc[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) acc[c] = threadIdx.x * 1e-3 + c;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = fma(acc[c], a, b);
    }
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += acc[c];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CH>
void run(double* d, int wgs) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(fma_chain<CH>, dim3(wgs), dim3(256), 0, 0, d, 1.0000001, 1e-9, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(fma_chain<CH>, dim3(wgs), dim3(256), 0, 0, d, 1.0000001, 1e-9, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fmas_per_wave = (double)iters * 16 * CH;
  const double ns_per_fma = ms * 1e6 / fmas_per_wave;           // one wave per SIMD: wall time / FMAs issued by a wave
  const double tflops = 2.0 * fmas_per_wave * 64 * 4 * wgs / (ms * 1e-3) / 1e12;
  printf("chains %2d  wgs %4d: %.3f ms  %.2f ns per wave-FMA (%.1f cycles at 2.4 GHz)  %.1f TFLOP/s\n", CH, wgs, ms, ns_per_fma, ns_per_fma * 2.4, tflops);
}
int main() {
  double* d; hipMalloc(&d, sizeof(double) * 256 * 2048);
  for (int wgs : {256, 512, 1024}) {
    run<1>(d, wgs); run<2>(d, wgs); run<3>(d, wgs); run<4>(d, wgs); run<6>(d, wgs); run<8>(d, wgs); run<12>(d, wgs); run<16>(d, wgs);
  }
  return 0;
}

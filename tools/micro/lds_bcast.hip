// Microbenchmark: cost of wave-uniform (broadcast) ds_read_b128 on gfx950 -- the operand-delivery path of the one-lane
// fused kernels: every lane reads the SAME 16 bytes.  Measures, per wave, the time per read when reads are
// issued back to back in batches of 8 (throughput) and one at a time (latency), with 1, 2 and 4 waves per CU sharing the LDS,
// and the same batches interleaved with 16 independent fp64 FMAs (do the two overlap inside one wave?).
// Build: hipcc -w -O3 --offload-arch=gfx950 lds_bcast.hip -o lds_bcast
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));

template <int MODE>   // 0: batches of 8 reads; 1: single dependent reads; 2: batches of 8 reads + 16 FMAs issued after the reads of the NEXT batch
__global__ __launch_bounds__(256) void lds_kernel(double* out, int iters, double a) {
  __shared__ __attribute__((aligned(16))) double rec[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) rec[i] = 1.0 + 1e-6 * i;
  __syncthreads();
  unsigned addr = (unsigned)(size_t)rec;     // LDS byte address (low 32 bits of the generic->local cast below)
  addr = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(uintptr_t)(__attribute__((address_space(3))) double*)rec));
  double acc[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) acc[c] = threadIdx.x * 1e-3 + c;
  d2 r0, r1, r2, r3, r4, r5, r6, r7;
  double s = 0.0;
  unsigned off = 0;
  for (int i = 0; i < iters; ++i) {
    const unsigned p = addr + (off & 0x3fff);
    off += 128;
    if (MODE == 1) {
      asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r0) : "v"(p) : "memory");
      s += r0.x;
      asm volatile("ds_read_b128 %0, %1 offset:16\n s_waitcnt lgkmcnt(0)" : "=v"(r1) : "v"(p + (unsigned)(s == 12345.0)) : "memory");
      s += r1.x;
    } else {
      asm volatile(
          "ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:16\n ds_read_b128 %2, %8 offset:32\n ds_read_b128 %3, %8 offset:48\n"
          "ds_read_b128 %4, %8 offset:64\n ds_read_b128 %5, %8 offset:80\n ds_read_b128 %6, %8 offset:96\n ds_read_b128 %7, %8 offset:112\n"
          : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(p) : "memory");
      if (MODE == 2) {
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = fma(acc[c], a, 1e-9);      // independent of the reads in flight
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      s += r0.x + r1.y + r2.x + r3.y + r4.x + r5.y + r6.x + r7.y;
    }
  }
#pragma unroll
  for (int c = 0; c < 16; ++c) s += acc[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(double* d, int threads, const char* what) {
  const int iters = 20000, wgs = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(lds_kernel<MODE>, dim3(wgs), dim3(threads), 0, 0, d, 10, 1.0000001);
  hipEventRecord(e0);
  hipLaunchKernelGGL(lds_kernel<MODE>, dim3(wgs), dim3(threads), 0, 0, d, iters, 1.0000001);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double reads = (double)iters * (MODE == 1 ? 2 : 8);
  printf("%-46s waves/CU %d: %.2f ns per wave-read (%.1f cycles at 2.4 GHz), %.1f cycles per read per CU\n", what, threads / 64, ms * 1e6 / reads,
         ms * 1e6 / reads * 2.4, ms * 1e6 / reads * 2.4 / (threads / 64));
}
int main() {
  double* d; hipMalloc(&d, sizeof(double) * 256 * 256);
  for (int t : {64, 128, 256}) {
    run<0>(d, t, "batches of 8 broadcast ds_read_b128");
    run<1>(d, t, "single dependent ds_read_b128 (latency)");
    run<2>(d, t, "batches of 8 reads + 16 independent FMAs");
  }
  return 0;
}

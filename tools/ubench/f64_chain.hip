// Micro-benchmark: f64 VALU issue on gfx950 as a function of (a) independent dependency chains per wavefront and
// (b) wavefronts per SIMD.  One workgroup per CU (LDS-limited), 256 CUs; reports cycles per f64 instruction per SIMD.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off f64_chain.hip -o f64_chain && ./f64_chain
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CHAINS, int OP>   // OP 0: fma, 1: mul+add pairs (two dependent instructions)
__global__ void k(double* out, int iters, double m, double c) {
  extern __shared__ char lds[];
  double a[CHAINS];
#pragma unroll
  for (int j = 0; j < CHAINS; j++) a[j] = threadIdx.x * 1e-3 + j;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
#pragma unroll
      for (int j = 0; j < CHAINS; j++) {
        if (OP == 0) a[j] = __builtin_fma(a[j], m, c);
        else { a[j] = a[j] * m; a[j] = a[j] + c; }
      }
    }
  }
  double s = 0;
#pragma unroll
  for (int j = 0; j < CHAINS; j++) s += a[j];
  if (s == 12345.678) out[0] = s + (lds[0] ? 1 : 0);
}

template <int CHAINS, int OP> void run(double* d, int threads) {
  const int iters = 4000;
  const size_t lds = 100 * 1024;     // one workgroup per CU
  (void)hipFuncSetAttribute((const void*)k<CHAINS, OP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<CHAINS, OP>), dim3(256), dim3(threads), lds, 0, d, 10, 1.0000001, 1e-9);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<CHAINS, OP>), dim3(256), dim3(threads), lds, 0, d, iters, 1.0000001, 1e-9);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_wave = (double)iters * 8 * CHAINS * (OP == 0 ? 1 : 2);
  const int waves_per_simd = threads / 256;
  const double cycles = ms * 1e-3 * 2.4e9;
  printf("chains %d op %s waves/SIMD %d: %.3f ms  -> %.2f cycles per instruction per wave, %.2f per SIMD\n", CHAINS, OP == 0 ? "fma" : "mul+add",
         waves_per_simd, ms, cycles / instr_per_wave, cycles / (instr_per_wave * waves_per_simd));
}
int main() {
  double* d; (void)hipMalloc(&d, 64);
  for (int threads : {256, 512, 1024}) {
    run<1, 0>(d, threads); run<2, 0>(d, threads); run<4, 0>(d, threads); run<8, 0>(d, threads);
    run<1, 1>(d, threads); run<2, 1>(d, threads); run<4, 1>(d, threads);
  }
  return 0;
}

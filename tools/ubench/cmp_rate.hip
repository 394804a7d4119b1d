// Micro-benchmark: issue rate of 64-bit vs 32-bit integer compares / selects on gfx950 (one wave per SIMD slot).
// hipcc --offload-arch=gfx950 -O3 cmp_rate.hip -o cmp_rate && ./cmp_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void __launch_bounds__(256) k(unsigned long long* out, int iters) {
  unsigned long long a = threadIdx.x * 0x9E3779B97F4A7C15ull + 1, b = a ^ 0x1234567ull, c = 0;
  unsigned long long k0 = a + 3, k1 = a + 5, k2 = a + 7, k3 = a + 9;
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) {          // four independent u64 compares
      c += (a < k0) + (a < k1) + (a < k2) + (a < k3);
    } else if (MODE == 1) {   // four independent u32 compares
      c += ((unsigned)a < (unsigned)k0) + ((unsigned)a < (unsigned)k1) + ((unsigned)a < (unsigned)k2) + ((unsigned)a < (unsigned)k3);
    } else {                  // four (hi, lo) lexicographic compares spelled in 32-bit ops
      const unsigned ah = a >> 32, al = (unsigned)a;
      c += ((ah < (unsigned)(k0 >> 32)) | ((ah == (unsigned)(k0 >> 32)) & (al < (unsigned)k0))) +
           ((ah < (unsigned)(k1 >> 32)) | ((ah == (unsigned)(k1 >> 32)) & (al < (unsigned)k1))) +
           ((ah < (unsigned)(k2 >> 32)) | ((ah == (unsigned)(k2 >> 32)) & (al < (unsigned)k2))) +
           ((ah < (unsigned)(k3 >> 32)) | ((ah == (unsigned)(k3 >> 32)) & (al < (unsigned)k3)));
    }
    a += b; k0 ^= a; k1 += c; k2 ^= b; k3 += a;     // keep everything live and loop-carried
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c + a + k0 + k1 + k2 + k3;
}

template <int MODE> float run(unsigned long long* d, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(256), 0, 0, d, 16);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  unsigned long long* d; hipMalloc(&d, 1024 * 256 * 8);
  const int iters = 20000;
  printf("u64 compares x4/iter : %.3f ms\n", run<0>(d, iters));
  printf("u32 compares x4/iter : %.3f ms\n", run<1>(d, iters));
  printf("hi/lo spelled x4/iter: %.3f ms\n", run<2>(d, iters));
  return 0;
}

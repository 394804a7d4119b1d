// How many dependent kernel boundaries per second does the chip sustain, over all queues?  K host threads, one stream each, N tiny
// kernels back to back per stream (every kernel waits for its predecessor on the stream: the shape of the per-scan SLAM step).
//   hipcc --offload-arch=gfx950 -O2 -pthread tools/ubench/launch_rate.hip -o /tmp/launch_rate && /tmp/launch_rate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

__global__ void tiny(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 20000;
  int* d = nullptr;
  hipMalloc(&d, 4096 * sizeof(int));
  hipMemset(d, 0, 4096 * sizeof(int));
  for (int K : {1, 2, 4, 8, 16, 32}) {
    std::vector<hipStream_t> st(K);
    for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int k = 0; k < K; k++) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st[k], d + 64 * k);
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int k = 0; k < K; k++)
      th.emplace_back([&, k] {
        for (int i = 0; i < N; i++) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st[k], d + 64 * k);
        hipStreamSynchronize(st[k]);
      });
    for (auto& t : th) t.join();
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("{\"streams\": %d, \"kernels_per_stream\": %d, \"us_per_kernel_per_stream\": %.3f, \"kernels_per_s_all_streams\": %.0f}\n", K, N, 1e6 * s / N, K * (double)N / s);
    for (auto& s2 : st) hipStreamDestroy(s2);
  }
  return 0;
}

// Host-side cost of the HIP calls the per-scan SLAM step is made of (the step is enqueue-bound in pipelined use):
// wall clock per call, averaged over batches that are small enough not to fill any queue, with the stream drained
// between batches.  hipcc --offload-arch=gfx950 -O2 tools/ubench/host_api_cost.hip -o /tmp/host_api_cost && /tmp/host_api_cost
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <chrono>
#include <cstdio>
#include <functional>

__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 12345) *p = 1; }
struct Big { int v[64]; };
__global__ void bigarg_kernel(Big b, int* p) { if (p && threadIdx.x == 12345) *p = b.v[3]; }

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

static double per_call_us(const std::function<void()>& f, hipStream_t* drain, int n_drain, int batch = 64, int reps = 30) {
  double total = 0;
  for (int r = 0; r < reps + 3; r++) {
    for (int i = 0; i < n_drain; i++) (void)hipStreamSynchronize(drain[i]);
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < batch; i++) f();
    const auto t1 = std::chrono::steady_clock::now();
    if (r >= 3) total += std::chrono::duration<double>(t1 - t0).count();
  }
  return 1e6 * total / (reps * batch);
}

int main() {
  hipStream_t s[2];
  CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
  int* d = nullptr; int* hp = nullptr;
  const size_t n = 32768;
  CK(hipMalloc(&d, 4 * n * sizeof(unsigned long long) + 1024));
  CK(hipHostMalloc((void**)&hp, 1 << 20, hipHostMallocDefault));
  hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  unsigned long long* k1 = reinterpret_cast<unsigned long long*>(d); unsigned long long* k2 = k1 + n;
  int* v1 = reinterpret_cast<int*>(k2 + n); int* v2 = v1 + n; int* sc = v2 + n; int* so = sc + n;
  CK(hipMemset(d, 0x5a, 4 * n * sizeof(unsigned long long)));
  size_t tb = 0, tb2 = 0;
  CK(rocprim::radix_sort_pairs(nullptr, tb, k1, k2, v1, v2, n, 0u, 64u, s[0]));
  CK(rocprim::exclusive_scan(nullptr, tb2, sc, so, 0, n, rocprim::plus<int>(), s[0]));
  void* tmp = nullptr; CK(hipMalloc(&tmp, tb + tb2 + 256));
  Big big{};
  struct Row { const char* name; std::function<void()> f; int batch; };
  Row rows[] = {
      {"kernel launch, 1 x 64 threads, one pointer argument", [&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s[0], d); }, 64},
      {"kernel launch, 256-byte argument block", [&] { hipLaunchKernelGGL(bigarg_kernel, dim3(1), dim3(64), 0, s[0], big, d); }, 64},
      {"hipMemsetAsync, 4 bytes", [&] { (void)hipMemsetAsync(d, 0, 4, s[0]); }, 64},
      {"hipMemsetAsync, 480 bytes", [&] { (void)hipMemsetAsync(d, 0, 480, s[0]); }, 64},
      {"hipMemcpyAsync H2D, 8 bytes from pinned memory", [&] { (void)hipMemcpyAsync(d, hp, 8, hipMemcpyHostToDevice, s[0]); }, 64},
      {"hipMemcpyAsync H2D, 460 KB from pinned memory", [&] { (void)hipMemcpyAsync(d, hp, 460 * 1024, hipMemcpyHostToDevice, s[0]); }, 16},
      {"hipMemcpyAsync D2H, 4 bytes to pinned memory", [&] { (void)hipMemcpyAsync(hp, d, 4, hipMemcpyDeviceToHost, s[0]); }, 64},
      {"hipMemcpyAsync D2H, 480 bytes to pinned memory", [&] { (void)hipMemcpyAsync(hp, d, 480, hipMemcpyDeviceToHost, s[0]); }, 64},
      {"hipEventRecord", [&] { (void)hipEventRecord(ev, s[0]); }, 64},
      {"hipEventRecord + hipStreamWaitEvent on a second stream", [&] { (void)hipEventRecord(ev, s[0]); (void)hipStreamWaitEvent(s[1], ev, 0); }, 64},
      {"hipEventQuery", [&] { (void)hipEventQuery(ev); }, 64},
      {"hipEventSynchronize on a completed event", [&] { (void)hipEventSynchronize(ev); }, 64},
      {"rocprim::radix_sort_pairs, 28 800 x (u64, int): size query + run", [&] {
         size_t t = 0; (void)rocprim::radix_sort_pairs(nullptr, t, k1, k2, v1, v2, (size_t)28800, 0u, 64u, s[0]);
         (void)rocprim::radix_sort_pairs(tmp, t, k1, k2, v1, v2, (size_t)28800, 0u, 64u, s[0]); }, 16},
      {"rocprim::exclusive_scan, 28 800 ints: size query + run", [&] {
         size_t t = 0; (void)rocprim::exclusive_scan(nullptr, t, sc, so, 0, (size_t)28800, rocprim::plus<int>(), s[0]);
         (void)rocprim::exclusive_scan(tmp, t, sc, so, 0, (size_t)28800, rocprim::plus<int>(), s[0]); }, 16},
      {"memcpy 518 KB pageable -> pinned (the scan's staging copy)", [&] { std::memcpy(hp, reinterpret_cast<char*>(hp) + (1 << 19), 512 * 1024 - 64); }, 16},
  };
  std::printf("| call | host time per call, us |\n|---|---|\n");
  for (auto& r : rows) std::printf("| %s | %.2f |\n", r.name, per_call_us(r.f, s, 2, r.batch));
  (void)hipDeviceSynchronize();
  return 0;
}

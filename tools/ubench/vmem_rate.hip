// vmem_rate.hip — what a vector load instruction costs the texture addresser / L1 of a CU on gfx950 (MI355X), as a function of
// its width, its address pattern and the number of active lanes.
//
// Question (round 4b): the 5-NN kernel's texture addresser is 78 % busy (TA_BUSY_avr) with ~84 vector loads per wavefront.
// Does a 16-byte-per-lane gather cost the CU by the instruction, by the byte or by the L1 request?  This probe issues
// L1-resident loads (a 16 KB window per workgroup) from every wavefront slot of the chip and reports
//     CU-clocks per wave-load-instruction = (CUs x clock x time) / (wavefronts x loads per wavefront)
// for  width    : dword, dwordx2, dwordx3, dwordx4
//      pattern  : coalesced (lane i reads element i), scattered (every lane its own 128-byte line),
//                 pairs (lanes 2k and 2k+1 read the SAME element: what neighbouring queries walking one range do),
//                 uniform (all lanes one address: a per-scan pose read with per-lane addresses)
//      active   : 64, 32, 16, 8, 4 lanes (the rest masked off by a branch)
// The clock is taken as 2.4 GHz (the kernels of the bench run at it, GRBM_GUI_ACTIVE / duration).
//
// hipcc --offload-arch=gfx950 -O3 vmem_rate.hip -o vmem_rate && ./vmem_rate [out.json]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f3 __attribute__((ext_vector_type(3)));
typedef float f4 __attribute__((ext_vector_type(4)));

enum Pattern { COALESCED = 0, SCATTERED, PAIRS, UNIFORM, N_PATTERNS };
static const char* kPatternName[N_PATTERNS] = {"coalesced", "scattered (one 128-B line per lane)", "pairs (two lanes per element)", "uniform (one address)"};

constexpr int kWindow = 16384;         // bytes per workgroup: stays in the 32 KB vector L1
constexpr int kLoadsPerIter = 8;

template <int WIDTH> struct Vec;
template <> struct Vec<1> { typedef float T; };
template <> struct Vec<2> { typedef f2 T; };
template <> struct Vec<3> { typedef f3 T; };
template <> struct Vec<4> { typedef f4 T; };
__device__ __forceinline__ float fold(float v) { return v; }
__device__ __forceinline__ float fold(f2 v) { return v.x + v.y; }
__device__ __forceinline__ float fold(f3 v) { return v.x + v.y + v.z; }
__device__ __forceinline__ float fold(f4 v) { return v.x + v.y + v.z + v.w; }

template <int WIDTH>
__global__ void __launch_bounds__(256) vmem_kernel(const char* __restrict__ buf, int pattern, int active, int iters, float* __restrict__ out) {
  typedef typename Vec<WIDTH>::T V;
  const int lane = threadIdx.x & 63;
  const char* base = buf + (size_t)(blockIdx.x % 64) * kWindow;           // 64 windows = 1 MB: L2 resident, one window per workgroup in L1
  // byte offset of this lane's element inside a 2 KB group; the eight loads of an iteration walk eight groups of the window
  int off;
  if (pattern == COALESCED) off = lane * (int)sizeof(V);       // one contiguous span (f3 has the stride of f4)
  else if (pattern == SCATTERED) off = lane * 128;             // 64 different 128-byte lines
  else if (pattern == PAIRS) off = (lane >> 1) * 128;          // 32 different lines, two lanes on each element
  else off = 0;
  float acc = 0.0f;
  if (lane < active) {
    for (int it = 0; it < iters; it++) {
      V v[kLoadsPerIter];
#pragma unroll
      for (int k = 0; k < kLoadsPerIter; k++) {
        const char* p = base + ((off + k * 2048 + it * 16) & (kWindow - 1) & ~15);
        v[k] = *(const V*)p;
        asm volatile("" : "+v"(v[k]));
      }
#pragma unroll
      for (int k = 0; k < kLoadsPerIter; k++) acc += fold(v[k]);
    }
  }
  if (acc == 123.456f) out[0] = acc;        // never true: keeps the loads
}

template <int WIDTH>
double run(const char* buf, float* out, int pattern, int active, int blocks, int iters) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(vmem_kernel<WIDTH>, dim3(blocks), dim3(256), 0, 0, buf, pattern, active, iters / 4, out);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(vmem_kernel<WIDTH>, dim3(blocks), dim3(256), 0, 0, buf, pattern, active, iters, out);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e-3;
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double clock = 2.4e9;
  char* buf; float* out;
  CHECK(hipMalloc(&buf, 64 * kWindow + 4096)); CHECK(hipMemset(buf, 0, 64 * kWindow + 4096));
  CHECK(hipMalloc(&out, 64));
  const int blocks = cus * 8;                 // 8 workgroups of 4 wavefronts per CU: every wavefront slot (8 per SIMD)
  const int iters = 2000;
  std::string json = "{\"cus\": " + std::to_string(cus) + ", \"clock_hz\": 2.4e9, \"rows\": [";
  printf("%-40s %6s %7s %28s\n", "pattern", "width", "active", "CU-clocks per wave-load instr");
  bool first = true;
  for (int pattern = 0; pattern < N_PATTERNS; pattern++)
    for (int width = 1; width <= 4; width++)
      for (int active : {64, 32, 16, 8, 4}) {
        double t;
        if (width == 1) t = run<1>(buf, out, pattern, active, blocks, iters);
        else if (width == 2) t = run<2>(buf, out, pattern, active, blocks, iters);
        else if (width == 3) t = run<3>(buf, out, pattern, active, blocks, iters);
        else t = run<4>(buf, out, pattern, active, blocks, iters);
        const double insts = (double)blocks * 4 * iters * kLoadsPerIter;       // wave-load instructions
        const double cu_clocks = cus * clock * t / insts;
        printf("%-40s %6d %7d %28.2f\n", kPatternName[pattern], width, active, cu_clocks);
        char row[256];
        snprintf(row, sizeof(row), "%s{\"pattern\": \"%s\", \"dwords\": %d, \"active_lanes\": %d, \"cu_clocks_per_load\": %.3f}", first ? "" : ", ", kPatternName[pattern], width, active, cu_clocks);
        json += row; first = false;
      }
  json += "]}";
  if (argc > 1) { FILE* f = fopen(argv[1], "w"); if (f) { fputs(json.c_str(), f); fputc('\n', f); fclose(f); } }
  return 0;
}

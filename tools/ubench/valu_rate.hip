// valu_rate.hip — issue rate of the VALU instructions the 5-NN kernel is made of, on gfx950 (MI355X).
//
// Question (What's weak #3): does a wave64 VALU instruction occupy its SIMD for 4 clocks or for 2?
// DESIGN.md read SQ_ACTIVE_INST_VALU (quad-cycles) ~= SQ_INSTS_VALU as "4 clocks each => VALU 94 % busy"; a
// quad-cycle counter reports >= 1 per instruction either way.  This probe measures it directly:
//   * one workgroup on one CU, W wavefronts per SIMD (W = 1, 2, 4, 8), every wavefront runs `iters` blocks of 32
//     identical instructions, either 8 independent chains or 1 dependent chain, written as inline asm so that the
//     compiler cannot fuse, reorder or drop anything;
//   * clocks by s_memtime (shader clock) read by every wavefront around its loop; reported as
//       cycles per instruction per wavefront            (latency view)
//       cycles per instruction per SIMD = that / W      (throughput view: 4.0 = one wave64 op per 4 clocks, 2.0 = per 2)
//   * the same kernels over the whole chip (256 CUs x 4 SIMDs x W wavefronts) under HIP events: instructions / s.
// Run under `rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES` to see what the
// counters report per instruction for a kernel whose instruction count is known exactly.
//
// hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

enum Op { FMA_F32 = 0, PK_MUL_F32, PK_ADD_F32, CNDMASK, CMP_U64, FMA_F64, ADD_U32, CNDMASK_SGPR, CNDMASK_VCC_ONCE, BFI_B32, MIN_U32, MED3_U32, AND_B32, MOV_B32, CMP_F32, CMP_U32, ADD_F32, MUL_F32, LSHL_ADD_U64, MAX_F32, CNDMASK_E64_VCC, CMP_U64_SGPR, CNDMASK_VCC_DISTINCT, N_OPS };
static const char* kOpName[N_OPS] = {"v_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_cndmask_b32(vcc,cmp/32)", "v_cmp_lt_u64", "v_fma_f64", "v_add_u32", "v_cndmask_b32_e64(sgpr)", "v_cndmask_b32(vcc const)", "v_bfi_b32", "v_min_u32", "v_med3_u32", "v_and_b32", "v_mov_b32", "v_cmp_lt_f32", "v_cmp_lt_u32", "v_add_f32", "v_mul_f32", "v_lshl_add_u64", "v_max_f32", "v_cndmask_b32_e64(vcc)", "v_cmp_lt_u64_e64(sgpr dst)", "v_cndmask_b32(vcc, dst!=src)"};

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)


#define GEN_OPP(PRE, INSTR_DEP, INSTR_IND)                                                                                        \
  {                                                                                                                          \
    unsigned bi = __float_as_uint(b), ci = __float_as_uint(c);                                                               \
    unsigned* ai = reinterpret_cast<unsigned*>(a);                                                                           \
    if (DEP) asm volatile(PRE REP32(INSTR_DEP) : "+v"(ai[0]) : "v"(bi), "v"(ci) : "vcc", "s20", "s21");                             \
    else asm volatile(PRE REP4(INSTR_IND) : "+v"(ai[0]), "+v"(ai[1]), "+v"(ai[2]), "+v"(ai[3]), "+v"(ai[4]), "+v"(ai[5]), "+v"(ai[6]), "+v"(ai[7]) \
                      : "v"(bi), "v"(ci) : "vcc", "s20", "s21");                                                             \
  }
#define GEN_OP(INSTR_DEP, INSTR_IND) GEN_OPP("", INSTR_DEP, INSTR_IND)
#define IND8(op, tail) op " %0, %0" tail "\n" op " %1, %1" tail "\n" op " %2, %2" tail "\n" op " %3, %3" tail "\n" op " %4, %4" tail "\n" op " %5, %5" tail "\n" op " %6, %6" tail "\n" op " %7, %7" tail "\n"

// 32 instructions per call; DEP: all on accumulator 0, else round-robin over 8 accumulators
template <int OP, bool DEP>
__device__ __forceinline__ void block32(float (&a)[8], f2 (&p)[8], double (&d)[8], unsigned long long (&u)[8], float b, float c) {
  if (OP == FMA_F32) {
    if (DEP) asm volatile(REP32("v_fma_f32 %0, %0, %1, %2\n") : "+v"(a[0]) : "v"(b), "v"(c));
    else asm volatile(REP4("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                           "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                      : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));
  } else if (OP == PK_MUL_F32 || OP == PK_ADD_F32) {
    const f2 bb = {b, c};
    if (OP == PK_MUL_F32) {
      if (DEP) asm volatile(REP32("v_pk_mul_f32 %0, %0, %1\n") : "+v"(p[0]) : "v"(bb));
      else asm volatile(REP4("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                             "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n")
                        : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(bb));
    } else {
      if (DEP) asm volatile(REP32("v_pk_add_f32 %0, %0, %1\n") : "+v"(p[0]) : "v"(bb));
      else asm volatile(REP4("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                             "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n")
                        : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(bb));
    }
  } else if (OP == CNDMASK) {
    // vcc is set once per block by a compare the block does not count (33rd instruction, noted in the table)
    if (DEP) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n" REP32("v_cndmask_b32 %0, %0, %1, vcc\n") : "+v"(a[0]) : "v"(b), "v"(c) : "vcc");
    else asm volatile("v_cmp_lt_f32 vcc, %8, %9\n"
                      REP4("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                           "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
                      : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c) : "vcc");
  } else if (OP == CMP_U64) {
    // compares have no register chain: DEP and independent are the same stream (vcc is rewritten every time)
    asm volatile(REP4("v_cmp_lt_u64 vcc, %0, %1\n v_cmp_lt_u64 vcc, %1, %2\n v_cmp_lt_u64 vcc, %2, %3\n v_cmp_lt_u64 vcc, %3, %4\n"
                      "v_cmp_lt_u64 vcc, %4, %5\n v_cmp_lt_u64 vcc, %5, %6\n v_cmp_lt_u64 vcc, %6, %7\n v_cmp_lt_u64 vcc, %7, %0\n")
                 : : "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]), "v"(u[4]), "v"(u[5]), "v"(u[6]), "v"(u[7]) : "vcc");
  } else if (OP == FMA_F64) {
    const double bd = (double)b, cd = (double)c;
    if (DEP) asm volatile(REP32("v_fma_f64 %0, %0, %1, %2\n") : "+v"(d[0]) : "v"(bd), "v"(cd));
    else asm volatile(REP4("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                           "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n")
                      : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) : "v"(bd), "v"(cd));
  } else if (OP == CNDMASK_SGPR) {
    GEN_OPP("s_mov_b64 s[20:21], 0x55555555\n", "v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n", IND8("v_cndmask_b32_e64", ", %8, s[20:21]"))
  } else if (OP == CNDMASK_VCC_ONCE) {
    GEN_OPP("s_mov_b64 vcc, 0x55555555\n", "v_cndmask_b32 %0, %0, %1, vcc\n", IND8("v_cndmask_b32", ", %8, vcc"))
  } else if (OP == CNDMASK_E64_VCC) {
    GEN_OPP("s_mov_b64 vcc, 0x55555555\n", "v_cndmask_b32_e64 %0, %0, %1, vcc\n", IND8("v_cndmask_b32_e64", ", %8, vcc"))
  } else if (OP == CNDMASK_VCC_DISTINCT) {
    GEN_OPP("s_mov_b64 vcc, 0x55555555\n", "v_cndmask_b32 %0, %1, %2, vcc\n", "v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %8, %9, vcc\n v_cndmask_b32 %2, %8, %9, vcc\n v_cndmask_b32 %3, %8, %9, vcc\n v_cndmask_b32 %4, %8, %9, vcc\n v_cndmask_b32 %5, %8, %9, vcc\n v_cndmask_b32 %6, %8, %9, vcc\n v_cndmask_b32 %7, %8, %9, vcc\n")
  } else if (OP == CMP_U64_SGPR) {
    asm volatile(REP4("v_cmp_lt_u64_e64 s[20:21], %0, %1\n v_cmp_lt_u64_e64 s[22:23], %1, %2\n v_cmp_lt_u64_e64 s[20:21], %2, %3\n v_cmp_lt_u64_e64 s[22:23], %3, %4\n"
                      "v_cmp_lt_u64_e64 s[20:21], %4, %5\n v_cmp_lt_u64_e64 s[22:23], %5, %6\n v_cmp_lt_u64_e64 s[20:21], %6, %7\n v_cmp_lt_u64_e64 s[22:23], %7, %0\n")
                 : : "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]), "v"(u[4]), "v"(u[5]), "v"(u[6]), "v"(u[7]) : "s20", "s21", "s22", "s23");
  } else if (OP == BFI_B32) {
    GEN_OP("v_bfi_b32 %0, %1, %0, %2\n", IND8("v_bfi_b32", ", %8, %9"))
  } else if (OP == MIN_U32) {
    GEN_OP("v_min_u32 %0, %0, %1\n", IND8("v_min_u32", ", %8"))
  } else if (OP == MED3_U32) {
    GEN_OP("v_med3_u32 %0, %0, %1, %2\n", IND8("v_med3_u32", ", %8, %9"))
  } else if (OP == AND_B32) {
    GEN_OP("v_and_b32 %0, %0, %1\n", IND8("v_and_b32", ", %8"))
  } else if (OP == MOV_B32) {
    GEN_OP("v_mov_b32 %0, %1\n", "v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n")
  } else if (OP == CMP_F32) {
    GEN_OP("v_cmp_lt_f32 vcc, %0, %1\n", "v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8\n")
  } else if (OP == CMP_U32) {
    GEN_OP("v_cmp_lt_u32 vcc, %0, %1\n", "v_cmp_lt_u32 vcc, %0, %8\n v_cmp_lt_u32 vcc, %1, %8\n v_cmp_lt_u32 vcc, %2, %8\n v_cmp_lt_u32 vcc, %3, %8\n v_cmp_lt_u32 vcc, %4, %8\n v_cmp_lt_u32 vcc, %5, %8\n v_cmp_lt_u32 vcc, %6, %8\n v_cmp_lt_u32 vcc, %7, %8\n")
  } else if (OP == ADD_F32) {
    GEN_OP("v_add_f32 %0, %0, %1\n", IND8("v_add_f32", ", %8"))
  } else if (OP == MUL_F32) {
    GEN_OP("v_mul_f32 %0, %0, %1\n", IND8("v_mul_f32", ", %8"))
  } else if (OP == MAX_F32) {
    GEN_OP("v_max_f32 %0, %0, %1\n", IND8("v_max_f32", ", %8"))
  } else if (OP == LSHL_ADD_U64) {
    if (DEP) asm volatile(REP32("v_lshl_add_u64 %0, %0, 0, %1\n") : "+v"(u[0]) : "v"(u[1]));
    else asm volatile(REP4("v_lshl_add_u64 %0, %0, 0, %8\n v_lshl_add_u64 %1, %1, 0, %8\n v_lshl_add_u64 %2, %2, 0, %8\n v_lshl_add_u64 %3, %3, 0, %8\n"
                           "v_lshl_add_u64 %4, %4, 0, %8\n v_lshl_add_u64 %5, %5, 0, %8\n v_lshl_add_u64 %6, %6, 0, %8\n v_lshl_add_u64 %7, %7, 0, %8\n")
                      : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]) : "v"(d[0]));
  } else {
    unsigned bi = __float_as_uint(b);
    unsigned* ai = reinterpret_cast<unsigned*>(a);
    if (DEP) asm volatile(REP32("v_add_u32 %0, %0, %1\n") : "+v"(ai[0]) : "v"(bi));
    else asm volatile(REP4("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                           "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n")
                      : "+v"(ai[0]), "+v"(ai[1]), "+v"(ai[2]), "+v"(ai[3]), "+v"(ai[4]), "+v"(ai[5]), "+v"(ai[6]), "+v"(ai[7]) : "v"(bi));
  }
}

template <int OP, bool DEP>
__global__ void __launch_bounds__(1024) rate_kernel(unsigned long long* __restrict__ clocks, float* __restrict__ sink, int iters, float b, float c) {
  float a[8]; f2 p[8]; double d[8]; unsigned long long u[8];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    a[k] = 1.0f + 0.001f * (threadIdx.x + k); p[k] = f2{a[k], a[k] + 0.5f}; d[k] = (double)a[k];
    u[k] = 0x9E3779B97F4A7C15ull * (threadIdx.x + k + 1);
  }
  __syncthreads();
  const unsigned long long t0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; i++) block32<OP, DEP>(a, p, d, u, b, c);
  const unsigned long long t1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; k++) s += a[k] + p[k].x + p[k].y + (float)d[k];
  if (s == 12345.678f) sink[0] = s;                         // keeps the chains live
  if ((threadIdx.x & 63) == 0) {
    const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    clocks[2 * w] = t1 - t0; clocks[2 * w + 1] = w1 - w0;
  }
}

struct Result { double cyc_per_inst_wave, cyc_per_inst_simd, mhz, chip_ginst; };

template <int OP, bool DEP>
Result run(int waves_per_simd, unsigned long long* d_clk, float* d_sink, int n_cu) {
  const int iters = 4096;
  const int threads = 64 * 4 * waves_per_simd;                // 4 SIMDs per CU: consecutive wavefronts of a workgroup go round-robin
  std::vector<unsigned long long> h(2 * 4 * waves_per_simd);
  hipLaunchKernelGGL((rate_kernel<OP, DEP>), dim3(1), dim3(threads), 0, 0, d_clk, d_sink, 64, 1.0000001f, 1e-9f);   // warm-up
  hipLaunchKernelGGL((rate_kernel<OP, DEP>), dim3(1), dim3(threads), 0, 0, d_clk, d_sink, iters, 1.0000001f, 1e-9f);
  hipDeviceSynchronize();
  hipMemcpy(h.data(), d_clk, h.size() * 8, hipMemcpyDeviceToHost);
  double cyc = 0, wall = 0;
  for (int w = 0; w < 4 * waves_per_simd; w++) { cyc = std::max(cyc, (double)h[2 * w]); wall = std::max(wall, (double)h[2 * w + 1]); }
  Result r;
  const double n_inst = (double)iters * 32.0;
  r.cyc_per_inst_wave = cyc / n_inst;
  r.cyc_per_inst_simd = cyc / n_inst / waves_per_simd;
  r.mhz = wall > 0 ? cyc / (wall / 100.0) : 0;               // s_memrealtime ticks at 100 MHz
  // whole chip: n_cu workgroups (one per CU if the dispatcher spreads them, which it does for <= n_cu workgroups of this size)
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int chip_iters = 16384;
  hipEventRecord(e0);
  hipLaunchKernelGGL((rate_kernel<OP, DEP>), dim3(n_cu), dim3(threads), 0, 0, d_clk, d_sink, chip_iters, 1.0000001f, 1e-9f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  r.chip_ginst = (double)n_cu * 4 * waves_per_simd * chip_iters * 32.0 / (ms * 1e-3) / 1e9;   // wave64 instructions per second, 1e9
  hipEventDestroy(e0); hipEventDestroy(e1);
  return r;
}

template <int OP> void table(unsigned long long* d_clk, float* d_sink, int n_cu, FILE* js, bool& first) {
  for (int dep = 0; dep < 2; dep++) {
    for (int w : {1, 2, 4}) {                                  // a workgroup holds at most 16 wavefronts = 4 per SIMD
      const Result r = dep ? run<OP, true>(w, d_clk, d_sink, n_cu) : run<OP, false>(w, d_clk, d_sink, n_cu);
      printf("%-14s %-11s waves/SIMD %d : %6.2f cyc/inst/wave  %5.2f cyc/inst/SIMD  clock %4.0f MHz  chip %7.1f G wave-inst/s\n", kOpName[OP],
             dep ? "dependent" : "independent", w, r.cyc_per_inst_wave, r.cyc_per_inst_simd, r.mhz, r.chip_ginst);
      fprintf(js, "%s{\"op\":\"%s\",\"chain\":\"%s\",\"waves_per_simd\":%d,\"cyc_per_inst_wave\":%.3f,\"cyc_per_inst_simd\":%.3f,\"mhz\":%.0f,\"chip_g_wave_inst_per_s\":%.1f}",
              first ? "" : ",\n", kOpName[OP], dep ? "dependent" : "independent", w, r.cyc_per_inst_wave, r.cyc_per_inst_simd, r.mhz, r.chip_ginst);
      first = false;
    }
  }
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int n_cu = prop.multiProcessorCount;
  printf("%s, %d CUs, clockRate %d kHz\n", prop.gcnArchName, n_cu, prop.clockRate);
  unsigned long long* d_clk; float* d_sink;
  hipMalloc(&d_clk, (size_t)n_cu * 64 * 2 * 8); hipMalloc(&d_sink, 64);
  const char* out = argc > 1 ? argv[1] : "valu_rate.json";
  FILE* js = fopen(out, "w");
  if (!js) { js = stdout; }
  fprintf(js, "{\"device\":\"%s\",\"cus\":%d,\"rows\":[\n", prop.gcnArchName, n_cu);
  bool first = true;
  if (argc > 2 && !strcmp(argv[2], "pmc")) {                   // counter run: two kernels with exactly known instruction counts
    table<FMA_F32>(d_clk, d_sink, n_cu, js, first);
    table<PK_MUL_F32>(d_clk, d_sink, n_cu, js, first);
  } else {
    table<FMA_F32>(d_clk, d_sink, n_cu, js, first);
    table<PK_MUL_F32>(d_clk, d_sink, n_cu, js, first);
    table<PK_ADD_F32>(d_clk, d_sink, n_cu, js, first);
    table<CNDMASK>(d_clk, d_sink, n_cu, js, first);
    table<CMP_U64>(d_clk, d_sink, n_cu, js, first);
    table<ADD_U32>(d_clk, d_sink, n_cu, js, first);
    table<FMA_F64>(d_clk, d_sink, n_cu, js, first);
    table<CNDMASK_SGPR>(d_clk, d_sink, n_cu, js, first);
    table<CNDMASK_VCC_ONCE>(d_clk, d_sink, n_cu, js, first);
    table<BFI_B32>(d_clk, d_sink, n_cu, js, first);
    table<MIN_U32>(d_clk, d_sink, n_cu, js, first);
    table<MED3_U32>(d_clk, d_sink, n_cu, js, first);
    table<AND_B32>(d_clk, d_sink, n_cu, js, first);
    table<MOV_B32>(d_clk, d_sink, n_cu, js, first);
    table<CMP_F32>(d_clk, d_sink, n_cu, js, first);
    table<CMP_U32>(d_clk, d_sink, n_cu, js, first);
    table<ADD_F32>(d_clk, d_sink, n_cu, js, first);
    table<MUL_F32>(d_clk, d_sink, n_cu, js, first);
    table<MAX_F32>(d_clk, d_sink, n_cu, js, first);
    table<LSHL_ADD_U64>(d_clk, d_sink, n_cu, js, first);
    table<CNDMASK_E64_VCC>(d_clk, d_sink, n_cu, js, first);
    table<CNDMASK_VCC_DISTINCT>(d_clk, d_sink, n_cu, js, first);
    table<CMP_U64_SGPR>(d_clk, d_sink, n_cu, js, first);
  }
  fprintf(js, "\n]}\n");
  if (js != stdout) fclose(js);
  return 0;
}

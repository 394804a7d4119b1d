// msfl_api.hip — C-ABI implementation (include/msfl_c_api.h) over the HIP kernels.
//
// One handle = one HIP stream + all device scratch; no global state, so the reference's two
// matcher instances (odometry thread / mapping thread) map to two independent handles.
// There is NO CPU fallback: every entry point either runs the gfx950 kernels or returns an
// error status.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <chrono>
#include <string>
#include <vector>

#include <condition_variable>
#include <mutex>
#include <thread>

#include "../../include/msfl_c_api.h"
#include "msfl_kernels.cuh"
#include "msfl_extract.cuh"
#include "msfl_odom.cuh"
#include "msfl_grid.cuh"
#include "msfl_deskew.cuh"

using namespace msfl;

namespace {

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    size_t want = std::max(bytes, cap + cap / 2);          // geometric growth: a buffer that grows a little per call
    want = (want + 255) & ~size_t(255);                    // (the map store's) is not reallocated on every call
    if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) return e; p = nullptr; cap = 0; }
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// Ring of pinned host staging slots for small metadata uploads (offset tables, init words).
// hipMemcpyAsync from PAGEABLE memory may read the source after the call returns when the stream
// is busy, so metadata must live in pinned memory that stays untouched until its copy has run.
struct PinSlot { void* p = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool pending = false; };
struct PinRing {
  static constexpr int kSlots = 8;
  PinSlot slot[kSlots];
  int next = 0;
  hipError_t upload(void* dst, const void* src, size_t bytes, hipStream_t st) {
    PinSlot& s = slot[next];
    next = (next + 1) % kSlots;
    hipError_t e;
    if (s.pending) { e = hipEventSynchronize(s.ev); if (e != hipSuccess) return e; s.pending = false; }
    if (bytes > s.cap) {
      // generous floor and geometric growth: growing a slot costs a hipHostFree + hipHostMalloc (tens of milliseconds, and
      // it happened once per slot as soon as a (B+1)-int offset table of a 1 024-scan batch met a 4 KB slot)
      const size_t want = (std::max<size_t>(std::max(bytes, s.cap + s.cap / 2), 256 * 1024) + 4095) & ~size_t(4095);
      if (s.p) { e = hipHostFree(s.p); if (e != hipSuccess) return e; s.p = nullptr; s.cap = 0; }
      e = hipHostMalloc(&s.p, want, hipHostMallocDefault); if (e != hipSuccess) return e;
      s.cap = want;
    }
    if (!s.ev) { e = hipEventCreateWithFlags(&s.ev, hipEventDisableTiming); if (e != hipSuccess) return e; }
    std::memcpy(s.p, src, bytes);
    e = hipMemcpyAsync(dst, s.p, bytes, hipMemcpyHostToDevice, st); if (e != hipSuccess) return e;
    e = hipEventRecord(s.ev, st); if (e != hipSuccess) return e;
    s.pending = true;
    return hipSuccess;
  }
  void release() {
    for (auto& s : slot) {
      if (s.pending) (void)hipEventSynchronize(s.ev);
      if (s.ev) (void)hipEventDestroy(s.ev);
      if (s.p) (void)hipHostFree(s.p);
      s = PinSlot{};
    }
  }
};

// Pinned host buffer for small synchronous read-backs (counts, offsets, descriptors): a D2H copy into pageable
// memory goes through the runtime's staging path, which stalls for tens of milliseconds once in a while.
struct PinBuf {
  void* p = nullptr; size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    const size_t want = (std::max<size_t>(std::max(bytes, cap + cap / 2), 4096) + 4095) & ~size_t(4095);
    if (p) { hipError_t e = hipHostFree(p); if (e != hipSuccess) return e; p = nullptr; cap = 0; }
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct MapIndex {
  DevBuf gdesc;        // GridDesc, computed on the device (no host round trip in msfl_set_map)
  DevBuf bbox;         // 6 ordered ints, armed once and re-armed by grid_scatter_kernel (grid_setup_kernel for an empty cloud)
  int cap_cells = 0;   // capacity of cell_start / count (cells)
  // feedback for the table span of the next build (asynchronous read-back, never waited for)
  int* want_host = nullptr; hipEvent_t want_ev = nullptr; bool want_pending = false; int span = 0;
  DevBuf sorted;      // float4[n]
  DevBuf pos_of;      // int[n]: original index -> position in `sorted`
  DevBuf cell_start;  // int[n_cells + 1]
  int n_input = 0;    // points handed to msfl_set_map
};

enum TimerClass { T_ASSOC = 0, T_SOLVE, T_INDEX, T_EXTRACT, T_ODOM, T_FIT, T_ASSOC_SEEDED /* a subset of T_ASSOC */, T_COUNT };

struct TimedSpan { hipEvent_t a, b; int cls; };

}  // namespace

struct msfl_handle_s {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  msfl_params prm{};
  std::string last_error;

  bool have_map = false;
  MapIndex map_c, map_s;
  int grid_cap_cells = 64 * 1024 * 1024;  // hard limit of the dense cell table (MSFL_GRID_CAP_CELLS)
  int h2d_chunk_scans = 512;              // MSFL_H2D_CHUNK_SCANS: scans per PCIe chunk of a host-buffer batch (>= 2 chunks to pipeline)
  int h2d_sub_chunks = 2;                 // MSFL_H2D_SUB_CHUNKS: pieces a chunk arrives in (its first association pass follows them)
  // set by msfl_voxel_downsample_batch when it refuses a cloud for a non-finite point: the cloud's number (-1: none) and where its
  // points are on the device (the staged copy of a host batch, or the caller's device array).  msfl_voxel_downsample reads these,
  // not the error text, to run PCL's drop-the-non-finite-points path.
  int vox_nonfinite_cloud = -1;
  const float4* vox_staged_pts = nullptr;
  bool voxel_no_big = false;              // MSFL_VOXEL_NO_BIG=1: lists beyond the LDS forms go straight to the device-wide form (A/B of the round-5 big form)
  int odom_bin_split = -1;         // the scan-to-scan index build: -1 = several workgroups per cloud when the call holds few pairs with long clouds, 0 / 1 = never / always (MSFL_ODOM_BIN_SPLIT)
  int prep_split = -1;             // workgroups per scan of the extraction's ring split: -1 = by batch size, 1 = the one-workgroup kernel (MSFL_PREP_SPLIT)
  bool voxel_force_global = false;        // MSFL_VOXEL_GLOBAL=1: the batched voxel filter keeps the device-wide radix-sort form (A/B testing)
  long long odom_wave_max_targets = -1;   // MSFL_ODOM_WAVE_MAX_TARGETS: previous-scan points up to which a small batch takes the one-wavefront-per-query kernel (default 4096 per pair)
  bool odom_force_brute = false;          // MSFL_ODOM_BRUTE=1: stage B plane queries stay on the brute-force kernel (A/B testing)
  bool nn_by_feature = false;             // numbering of `nn` left by the last association pass (s_launch_assoc)
  bool knn_seed = false;                  // MSFL_KNN_SEED=1: the second outer iteration's 5-NN search starts from the bound the first one's neighbours give
                                          // (exact; measured slower, docs/rejected_experiments.md: -16 % candidates, +13 % launch time)
  int knn_form = 0;                       // MSFL_KNN_FORM: 0 auto (row-parallel latency form for launches of <= kKnnRowsMaxRecords queries),
                                          // 1 "lane" (one lane per query always), 2 "rows" (row-parallel always); results are identical

  // scratch
  DevBuf in_corner, in_surf, in_off, poses, status, info, records, pprime, nn;
  std::vector<int> in_off_host; const void* in_off_host_ptr = nullptr;   // what in_off holds on the device (upload_in_off)
  DevBuf idx_cell_of, idx_count, idx_scanned, idx_bbox, idx_cub, idx_stage;
  bool index_single = false;   // MSFL_INDEX_SINGLE=1: msfl_set_map builds its two indexes one after the other (the rounds 1-5 form; A/B and parity of the pair build)
  DevBuf knn_count;            // one u64: candidates evaluated by the counting 5-NN instantiation (timing mode 3)
  size_t idx_count_zero = 0;   // leading ints of idx_count known to be zero on the stream
  DevBuf dk[5];
  DevBuf ex[16];
  DevBuf od[20];
  DevBuf vb[14];  // batched voxel filter
  DevBuf fit_fallback;                    // two {count, record numbers} lists of the whole-batch fit kernel's deferred pivoted-QR planes
  size_t fit_fallback_words = 0; int fit_fallback_parity = 0;
  DevBuf vox_big_scratch, vox_big_list;   // the one-workgroup form for lists of 65 536 .. 131 071 points (round 5): per-workgroup run scratch, list of refused clouds
  DevBuf vb2[6];  // second scratch set of the pair form: staging, run sums, counts/flags/offsets, offsets
  DevBuf pp[5];   // per-point passes: pre-integration samples, staged points, dq, dp, flag
  DevBuf pr[5];   // batched (map, scan) pairs: map offsets and cell-table bases per cloud kind, preset status

  PinRing pin;
  PinBuf readback;
  hipStream_t copy_stream = nullptr;      // host-buffer batches: H2D of chunk k+1 under the compute of chunk k
  hipEvent_t copy_ev[9] = {};             // [8] = fork event

  // timing
  int timing = 0;            // 0 off, 1 every kernel class, 2 the association (5-NN) kernel only
  std::vector<TimedSpan> spans;
  std::vector<hipEvent_t> free_events;
  double t_ms[T_COUNT] = {0};
  int t_n[T_COUNT] = {0};
};

namespace {

msfl_status fail(msfl_handle* h, msfl_status s, const std::string& msg) {
  if (h) h->last_error = msg;
  return s;
}

#define HIPCHK(h, expr)                                                                            \
  do {                                                                                             \
    hipError_t e__ = (expr);                                                                       \
    if (e__ != hipSuccess)                                                                         \
      return fail(h, MSFL_HIP_ERROR, std::string(#expr) + ": " + hipGetErrorString(e__));          \
  } while (0)

hipEvent_t get_event(msfl_handle* h) {
  if (!h->free_events.empty()) { hipEvent_t e = h->free_events.back(); h->free_events.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

struct ScopedTimer {
  msfl_handle* h; TimedSpan s{}; bool on;
  ScopedTimer(msfl_handle* h_, int cls) : h(h_), on(h_->timing == 1 || h_->timing == 3 || (h_->timing == 2 && (cls == T_ASSOC || cls == T_ASSOC_SEEDED))) {
    if (on) { s.a = get_event(h); s.b = get_event(h); s.cls = cls; (void)hipEventRecord(s.a, h->stream); }
  }
  ~ScopedTimer() {
    if (on) { (void)hipEventRecord(s.b, h->stream); h->spans.push_back(s); }
  }
};

void collect_timing(msfl_handle* h) {
  for (auto& s : h->spans) {
    (void)hipEventSynchronize(s.b);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
      h->t_ms[s.cls] += ms; h->t_n[s.cls]++;
      if (s.cls == T_ASSOC_SEEDED) { h->t_ms[T_ASSOC] += ms; h->t_n[T_ASSOC]++; }
    }
    h->free_events.push_back(s.a); h->free_events.push_back(s.b);
  }
  h->spans.clear();
}

// The offset table of a batch on the device (h->in_off).  Registering the same batch layout again (a replayed or re-registered batch: the
// bench step, a retry after msfl_set_map) skips the copy: it is one more dependent operation on the stream in front of the first kernel.
msfl_status upload_in_off(msfl_handle* h, const int* data, size_t count, hipStream_t st) {
  HIPCHK(h, h->in_off.reserve(count * sizeof(int)));
  if (h->in_off_host.size() == count && h->in_off_host_ptr == h->in_off.p && std::memcmp(h->in_off_host.data(), data, count * sizeof(int)) == 0)
    return MSFL_OK;
  h->in_off_host.clear();                       // (stays empty if the upload fails)
  HIPCHK(h, h->pin.upload(h->in_off.p, data, count * sizeof(int), st));
  h->in_off_host.assign(data, data + count); h->in_off_host_ptr = h->in_off.p;
  return MSFL_OK;
}

SolverParams solver_params(const msfl_params& p, int min_corr) {
  SolverParams s;
  s.max_iterations = p.max_lm_iterations;
  s.huber = p.huber_delta;
  s.radius0 = p.initial_trust_region_radius;
  s.radius_max = p.max_trust_region_radius;
  s.radius_min = p.min_trust_region_radius;
  s.min_relative_decrease = p.min_relative_decrease;
  s.min_diag = p.min_lm_diagonal;
  s.max_diag = p.max_lm_diagonal;
  s.ftol = p.function_tolerance;
  s.gtol = p.gradient_tolerance;
  s.ptol = p.parameter_tolerance;
  s.max_invalid = p.max_consecutive_invalid_steps;
  s.min_correspondences = min_corr;
  return s;
}

inline int div_up(int a, int b) { return (a + b - 1) / b; }

__global__ void __launch_bounds__(256) zero_ints_kernel(int* __restrict__ p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0;
}

// Build the exact-kNN grid over `pts` (device pointer, n points).  Fully asynchronous: the grid
// descriptor is computed and kept on the device; the dense cell table has a fixed capacity
// (default 4 M cells, MSFL_GRID_CAP_CELLS) and the device grows the cell edge if the map's bounding
// box would need more (larger cells stay exact).
msfl_status build_index(msfl_handle* h, const float4* pts, int n, MapIndex& mi, const int* n_dev = nullptr) {
  ScopedTimer timer(h, T_INDEX);
  // the 5-NN walk addresses candidates by 32-bit byte offsets into the sorted copy (knn5_grid_k32)
  if (n >= (1 << 28)) return fail(h, MSFL_BAD_ARG, "map cloud of 2^28 points or more (the index addresses 16-byte points by 32-bit byte offsets)");
  mi.n_input = n;
  hipStream_t st = h->stream;
  // span of the dense cell table: what the previous build of this map said it needs (+25 %), else 1 M
  // cells.  The device grows the cell edge when the bounding box needs more than `cap` (still exact,
  // just more candidates per cell), and reports the wanted size for the next build.
  if (!mi.want_host) {
    HIPCHK(h, hipHostMalloc((void**)&mi.want_host, sizeof(int), hipHostMallocDefault));
    HIPCHK(h, hipEventCreateWithFlags(&mi.want_ev, hipEventDisableTiming));
    *mi.want_host = 0;
  }
  if (mi.want_pending && hipEventQuery(mi.want_ev) == hipSuccess) {
    mi.want_pending = false;
    const long long w = (long long)(*mi.want_host) + (*mi.want_host) / 4 + 1024;
    mi.span = (int)std::min<long long>(std::max<long long>(w, 65536), h->grid_cap_cells);
  }
  if (mi.span <= 0) mi.span = std::min(1 << 20, h->grid_cap_cells);
  const int cap = mi.span;
  HIPCHK(h, mi.gdesc.reserve(sizeof(GridDesc)));
  HIPCHK(h, h->idx_cell_of.reserve(std::max<size_t>(1, (size_t)n) * sizeof(int)));
  const void* count_before = h->idx_count.p;
  HIPCHK(h, h->idx_count.reserve(((size_t)cap + 1) * sizeof(int)));
  if (h->idx_count.p != count_before) h->idx_count_zero = 0;
  HIPCHK(h, mi.cell_start.reserve(((size_t)cap + 1) * sizeof(int)));
  HIPCHK(h, mi.sorted.reserve(std::max<size_t>(1, (size_t)n) * sizeof(float4)));
  HIPCHK(h, mi.pos_of.reserve(std::max<size_t>(1, (size_t)n) * sizeof(int)));
  mi.cap_cells = cap;
  if (!mi.bbox.p) {                       // armed once; the build re-arms it after the last read
    HIPCHK(h, mi.bbox.reserve(6 * sizeof(int)));
    const int init[6] = {INT32_MAX, INT32_MAX, INT32_MAX, INT32_MIN, INT32_MIN, INT32_MIN};
    HIPCHK(h, h->pin.upload(mi.bbox.p, init, sizeof(init), st));
  }
  if (n > 0) {
    const int blocks = std::min(div_up(n, 1024), 256);   // 6 atomics per workgroup, all on one line: four points per lane and pass
    hipLaunchKernelGGL(grid_bbox_kernel, dim3(blocks), dim3(256), 0, st, pts, n, mi.bbox.as<int>(), n_dev);
  }
  const double radius = std::sqrt((double)h->prm.map_knn_max_sq_dist);
  if (n == 0)
    hipLaunchKernelGGL(grid_setup_kernel, dim3(1), dim3(1), 0, st, mi.bbox.as<int>(), radius, cap, mi.gdesc.as<GridDesc>());
  // the table is cleared / scanned over the cells actually used last time (+ margin) when known,
  // else over the full capacity; the device never indexes beyond n_cells <= cap.
  const size_t span = (size_t)cap + 1;
  // the scatter kernel counts every cell back down to zero, so the table only needs clearing when it
  // is new or larger than what has been cleared before
  const size_t zeroed = h->idx_count_zero;
  h->idx_count_zero = 0;                  // unknown until this build has been enqueued completely
  if (span > zeroed) HIPCHK(h, hipMemsetAsync(h->idx_count.p, 0, span * sizeof(int), st));
  if (n > 0)
    hipLaunchKernelGGL(grid_count_kernel, dim3(div_up(n, 256)), dim3(256), 0, st, pts, n, (const int*)mi.bbox.as<int>(), radius, cap,
                       mi.gdesc.as<GridDesc>(), h->idx_cell_of.as<int>(), h->idx_count.as<int>(), n_dev);
  size_t tmp_bytes = 0;
  HIPCHK(h, rocprim::exclusive_scan(nullptr, tmp_bytes, h->idx_count.as<int>(), mi.cell_start.as<int>(), 0, (size_t)((int)span), rocprim::plus<int>(), st));
  HIPCHK(h, h->idx_cub.reserve(tmp_bytes));
  HIPCHK(h, rocprim::exclusive_scan(h->idx_cub.p, tmp_bytes, h->idx_count.as<int>(), mi.cell_start.as<int>(), 0, (size_t)((int)span), rocprim::plus<int>(), st));
  if (n > 0)
    hipLaunchKernelGGL(grid_scatter_kernel, dim3(div_up(n, 256)), dim3(256), 0, st, pts, n, h->idx_cell_of.as<int>(),
                       mi.cell_start.as<int>(), h->idx_count.as<int>(), mi.sorted.as<float4>(), mi.pos_of.as<int>(),
                       mi.gdesc.as<GridDesc>(), mi.bbox.as<int>(), n_dev);
  HIPCHK(h, hipGetLastError());
  h->idx_count_zero = std::max(zeroed, span);
  if (!mi.want_pending) {
    HIPCHK(h, hipMemcpyAsync(mi.want_host, &mi.gdesc.as<GridDesc>()->want_cells, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipEventRecord(mi.want_ev, st));
    mi.want_pending = true;
  }
  return MSFL_OK;
}

// Both maps of msfl_set_map through one chain of five launches (msfl_kernels.cuh: GridPairJob).  Same results as two build_index calls;
// used when both clouds are non-empty and their sizes are host-side numbers (the per-scan SLAM step builds its maps on streams of their own).
// The wanted table size comes back through a pinned host word the scatter launch writes (no copy, no event): it is a hint for the NEXT
// build's table span, so whichever value the host happens to see -- the previous build's or this one's -- is fine.
msfl_status build_index_pair(msfl_handle* h, const float4* pts_c, int n_c, const float4* pts_s, int n_s) {
  ScopedTimer timer(h, T_INDEX);
  if ((long long)n_c >= (1 << 28) || (long long)n_s >= (1 << 28))
    return fail(h, MSFL_BAD_ARG, "map cloud of 2^28 points or more (the index addresses 16-byte points by 32-bit byte offsets)");
  hipStream_t st = h->stream;
  MapIndex* mi[2] = {&h->map_c, &h->map_s};
  const float4* pts[2] = {pts_c, pts_s};
  const int n[2] = {n_c, n_s};
  GridPairJob j;
  for (int m = 0; m < 2; m++) {
    MapIndex& x = *mi[m];
    x.n_input = n[m];
    if (!x.want_host) {
      HIPCHK(h, hipHostMalloc((void**)&x.want_host, sizeof(int), hipHostMallocDefault));
      HIPCHK(h, hipEventCreateWithFlags(&x.want_ev, hipEventDisableTiming));
      *x.want_host = 0;
    }
    if (x.want_pending) { HIPCHK(h, hipEventSynchronize(x.want_ev)); x.want_pending = false; }   // a read-back copy of an earlier single build
    const int seen = *(volatile int*)x.want_host;
    if (seen > 0) {
      const long long w = (long long)seen + seen / 4 + 1024;
      x.span = (int)std::min<long long>(std::max<long long>(w, 65536), h->grid_cap_cells);
    }
    if (x.span <= 0) x.span = std::min(1 << 20, h->grid_cap_cells);
    x.cap_cells = x.span;
    HIPCHK(h, x.gdesc.reserve(sizeof(GridDesc)));
    HIPCHK(h, x.cell_start.reserve(((size_t)x.span + 1) * sizeof(int)));
    HIPCHK(h, x.sorted.reserve((size_t)n[m] * sizeof(float4)));
    HIPCHK(h, x.pos_of.reserve((size_t)n[m] * sizeof(int)));
    if (!x.bbox.p) {                       // armed once; every build re-arms it after the last read
      HIPCHK(h, x.bbox.reserve(6 * sizeof(int)));
      const int init[6] = {INT32_MAX, INT32_MAX, INT32_MAX, INT32_MIN, INT32_MIN, INT32_MIN};
      HIPCHK(h, h->pin.upload(x.bbox.p, init, sizeof(init), st));
    }
    j.pts[m] = pts[m]; j.n[m] = n[m]; j.bbox[m] = x.bbox.as<int>(); j.gdesc[m] = x.gdesc.as<GridDesc>(); j.cap[m] = x.span;
    j.cell_start[m] = x.cell_start.as<int>(); j.sorted[m] = x.sorted.as<float4>(); j.pos_of[m] = x.pos_of.as<int>();
    j.want_host[m] = x.want_host;
  }
  const size_t span = (size_t)j.cap[0] + 1 + (size_t)j.cap[1] + 1;
  HIPCHK(h, h->idx_cell_of.reserve(((size_t)n_c + (size_t)n_s) * sizeof(int)));
  const void* count_before = h->idx_count.p;
  HIPCHK(h, h->idx_count.reserve(span * sizeof(int)));
  if (h->idx_count.p != count_before) h->idx_count_zero = 0;
  HIPCHK(h, h->idx_scanned.reserve(span * sizeof(int)));
  j.cell_of = h->idx_cell_of.as<int>(); j.count = h->idx_count.as<int>(); j.scanned = h->idx_scanned.as<int>();
  j.blocks0 = div_up(n_c, 256); j.bbox_blocks0 = std::min(div_up(n_c, 1024), 256);
  j.radius = std::sqrt((double)h->prm.map_knn_max_sq_dist);
  const int bbox_blocks1 = std::min(div_up(n_s, 1024), 256), blocks1 = div_up(n_s, 256);
  hipLaunchKernelGGL(grid_bbox_pair_kernel, dim3(j.bbox_blocks0 + bbox_blocks1), dim3(256), 0, st, j);
  // the scatter launch counts every cell back down to zero: the table only needs clearing when it is new or larger than what has been cleared
  const size_t zeroed = h->idx_count_zero;
  h->idx_count_zero = 0;
  if (span > zeroed) HIPCHK(h, hipMemsetAsync(h->idx_count.p, 0, span * sizeof(int), st));
  hipLaunchKernelGGL(grid_count_pair_kernel, dim3(j.blocks0 + blocks1), dim3(256), 0, st, j);
  size_t tmp_bytes = 0;
  HIPCHK(h, rocprim::exclusive_scan(nullptr, tmp_bytes, j.count, j.scanned, 0, span, rocprim::plus<int>(), st));
  HIPCHK(h, h->idx_cub.reserve(tmp_bytes));
  HIPCHK(h, rocprim::exclusive_scan(h->idx_cub.p, tmp_bytes, j.count, j.scanned, 0, span, rocprim::plus<int>(), st));
  hipLaunchKernelGGL(grid_scatter_pair_kernel, dim3(j.blocks0 + blocks1), dim3(256), 0, st, j);
  HIPCHK(h, hipGetLastError());
  h->idx_count_zero = std::max(zeroed, span);
  return MSFL_OK;
}

// the two fallback lists of the whole-batch fit kernel (msfl_kernels.cuh: fit_fallback_kernel), n_surf + 1 ints each, zeroed when (re)allocated
msfl_status ensure_fit_fallback(msfl_handle* h, int n_surf) {
  const size_t words = (size_t)std::max(n_surf, 0) + 1;
  if (h->fit_fallback_words >= words) return MSFL_OK;
  HIPCHK(h, hipStreamSynchronize(h->stream));            // a launch in flight may still read the old lists
  HIPCHK(h, h->fit_fallback.reserve(2 * words * sizeof(int)));
  HIPCHK(h, hipMemsetAsync(h->fit_fallback.p, 0, 2 * words * sizeof(int), h->stream));
  h->fit_fallback_words = words; h->fit_fallback_parity = 0;
  return MSFL_OK;
}

// one data-association pass = kNN kernel + fit kernel
// (records [rec_begin, rec_end) of the batch; rec_end < 0: all of them)
// seed: `nn` still holds this batch's neighbours from the previous outer iteration (same records, same map index): the 5-NN search
// starts from the bound they give (knn5_seed_bound; exact, MSFL_KNN_SEED=0 switches it off for A/B)
void s_launch_assoc(msfl_handle* h, const BatchView& bv_all, const double* d_poses, const int* d_status, bool deskew,
                    const DeskewView& dv, int n_rec, double* full = nullptr, int rec_begin = 0, int rec_end = -1, bool second_pass = false) {
  // the whole-batch kernels number `nn` by feature slot, the per-record ones by record (msfl_kernels.cuh: feature_slot); a seeded second pass
  // reads the first pass's lists, so it is only seeded when both passes use the same numbering (a host-buffer batch's first pass runs chunk by chunk)
  const bool whole_batch = !deskew && !bv_all.dyn && rec_end < 0 && (rec_end >= 0 ? rec_end - rec_begin : n_rec) >= 65536 && !full && h->timing != 3;
  const bool seed = second_pass && h->knn_seed && h->nn_by_feature == whole_batch;
  h->nn_by_feature = whole_batch;
  hipStream_t st = h->stream;
  int* nn = h->nn.as<int>();
  BatchView bv = bv_all;
  if (rec_end >= 0) { bv.rec_begin = rec_begin; bv.n_records = rec_end; n_rec = rec_end - rec_begin; }
  if (n_rec <= 0) return;
  const dim3 grid(div_up(n_rec, kAssocBlock)), block(kAssocBlock);
  {
    ScopedTimer timer(h, second_pass ? T_ASSOC_SEEDED : T_ASSOC);      // second-pass launches are timed (and counted) on their own, seeded or not
    if (h->timing == 3) {
      unsigned long long* cnt = h->knn_count.as<unsigned long long>();
      if (deskew)
        hipLaunchKernelGGL((knn5_scan2map_kernel<true, true>), grid, block, 0, st, bv, d_poses, d_status,
                           (const GridDesc*)h->map_c.gdesc.as<GridDesc>(), h->map_c.sorted.as<float4>(), h->map_c.cell_start.as<int>(),
                           (const GridDesc*)h->map_s.gdesc.as<GridDesc>(), h->map_s.sorted.as<float4>(), h->map_s.cell_start.as<int>(),
                           (const int*)h->map_c.pos_of.as<int>(), (const int*)h->map_s.pos_of.as<int>(),
                           h->prm.map_knn_max_sq_dist, dv, nn, cnt);
      else if (seed)
        hipLaunchKernelGGL((knn5_scan2map_kernel<false, true, true>), grid, block, 0, st, bv, d_poses, d_status,
                           (const GridDesc*)h->map_c.gdesc.as<GridDesc>(), h->map_c.sorted.as<float4>(), h->map_c.cell_start.as<int>(),
                           (const GridDesc*)h->map_s.gdesc.as<GridDesc>(), h->map_s.sorted.as<float4>(), h->map_s.cell_start.as<int>(),
                           (const int*)h->map_c.pos_of.as<int>(), (const int*)h->map_s.pos_of.as<int>(),
                           h->prm.map_knn_max_sq_dist, dv, nn, cnt + 1);
      else
        hipLaunchKernelGGL((knn5_scan2map_kernel<false, true>), grid, block, 0, st, bv, d_poses, d_status,
                           (const GridDesc*)h->map_c.gdesc.as<GridDesc>(), h->map_c.sorted.as<float4>(), h->map_c.cell_start.as<int>(),
                           (const GridDesc*)h->map_s.gdesc.as<GridDesc>(), h->map_s.sorted.as<float4>(), h->map_s.cell_start.as<int>(),
                           (const int*)h->map_c.pos_of.as<int>(), (const int*)h->map_s.pos_of.as<int>(),
                           h->prm.map_knn_max_sq_dist, dv, nn, cnt + (second_pass ? 1 : 0));
    } else if (whole_batch) {       // a whole large batch: one body per feature kind (-2 %)
      const int n_s = bv.n_surf_total, n_c = n_rec - n_s;
      const int edge_blocks = div_up(n_c, kAssocBlock), plane_blocks = div_up(n_s, kAssocBlock);
      if (seed)
        hipLaunchKernelGGL(knn5_scan2map_split_kernel<true>, dim3(edge_blocks + plane_blocks), block, 0, st, bv, d_poses, d_status,
                           (const GridDesc*)h->map_c.gdesc.as<GridDesc>(), h->map_c.sorted.as<float4>(), h->map_c.cell_start.as<int>(),
                           (const GridDesc*)h->map_s.gdesc.as<GridDesc>(), h->map_s.sorted.as<float4>(), h->map_s.cell_start.as<int>(),
                           (const int*)h->map_c.pos_of.as<int>(), (const int*)h->map_s.pos_of.as<int>(), h->prm.map_knn_max_sq_dist, nn, edge_blocks);
      else
        hipLaunchKernelGGL(knn5_scan2map_split_kernel<false>, dim3(edge_blocks + plane_blocks), block, 0, st, bv, d_poses, d_status,
                           (const GridDesc*)h->map_c.gdesc.as<GridDesc>(), h->map_c.sorted.as<float4>(), h->map_c.cell_start.as<int>(),
                           (const GridDesc*)h->map_s.gdesc.as<GridDesc>(), h->map_s.sorted.as<float4>(), h->map_s.cell_start.as<int>(),
                           (const int*)h->map_c.pos_of.as<int>(), (const int*)h->map_s.pos_of.as<int>(), h->prm.map_knn_max_sq_dist, nn, edge_blocks);
    } else if (!deskew && (h->knn_form == 2 || (h->knn_form == 0 && (n_rec <= kKnnRowsMaxRecords ||
                                                                       // the SLAM step launches over the list CAPACITIES (device-side counts): a 64-beam scan's ~11-17 k
                                                                       // queries sit in a ~150 k-slot launch whose surplus workgroups leave at once
                                                                       (bv.dyn && n_rec <= 8 * kKnnRowsMaxRecords)))))
      hipLaunchKernelGGL(knn5_scan2map_rows_kernel, dim3(div_up(n_rec, kKnnRowsBlock / kKnnRowLanes)), dim3(kKnnRowsBlock), 0, st, bv, d_poses, d_status,
                         (const GridDesc*)h->map_c.gdesc.as<GridDesc>(), h->map_c.sorted.as<float4>(), h->map_c.cell_start.as<int>(),
                         (const GridDesc*)h->map_s.gdesc.as<GridDesc>(), h->map_s.sorted.as<float4>(), h->map_s.cell_start.as<int>(),
                         (const int*)h->map_c.pos_of.as<int>(), (const int*)h->map_s.pos_of.as<int>(), h->prm.map_knn_max_sq_dist, nn);
    else if (deskew)
      hipLaunchKernelGGL(knn5_scan2map_kernel<true>, grid, block, 0, st, bv, d_poses, d_status,
                         (const GridDesc*)h->map_c.gdesc.as<GridDesc>(), h->map_c.sorted.as<float4>(), h->map_c.cell_start.as<int>(),
                         (const GridDesc*)h->map_s.gdesc.as<GridDesc>(), h->map_s.sorted.as<float4>(), h->map_s.cell_start.as<int>(),
                         (const int*)h->map_c.pos_of.as<int>(), (const int*)h->map_s.pos_of.as<int>(),
                         h->prm.map_knn_max_sq_dist, dv, nn);
    else if (seed)
      hipLaunchKernelGGL((knn5_scan2map_kernel<false, false, true>), grid, block, 0, st, bv, d_poses, d_status,
                         (const GridDesc*)h->map_c.gdesc.as<GridDesc>(), h->map_c.sorted.as<float4>(), h->map_c.cell_start.as<int>(),
                         (const GridDesc*)h->map_s.gdesc.as<GridDesc>(), h->map_s.sorted.as<float4>(), h->map_s.cell_start.as<int>(),
                         (const int*)h->map_c.pos_of.as<int>(), (const int*)h->map_s.pos_of.as<int>(),
                         h->prm.map_knn_max_sq_dist, dv, nn);
    else
      hipLaunchKernelGGL(knn5_scan2map_kernel<false>, grid, block, 0, st, bv, d_poses, d_status,
                         (const GridDesc*)h->map_c.gdesc.as<GridDesc>(), h->map_c.sorted.as<float4>(), h->map_c.cell_start.as<int>(),
                         (const GridDesc*)h->map_s.gdesc.as<GridDesc>(), h->map_s.sorted.as<float4>(), h->map_s.cell_start.as<int>(),
                         (const int*)h->map_c.pos_of.as<int>(), (const int*)h->map_s.pos_of.as<int>(),
                         h->prm.map_knn_max_sq_dist, dv, nn);
  }
  {
    ScopedTimer timer(h, T_FIT);
    if (deskew)
      hipLaunchKernelGGL(fit_scan2map_kernel<true>, grid, block, 0, st, bv, h->map_c.sorted.as<float4>(),
                         h->map_s.sorted.as<float4>(), (const int*)nn, h->prm.line_eigen_ratio, h->prm.plane_tolerance, dv,
                         h->records.as<double>(), full);
    else
      if (whole_batch) {    // (the same condition as the 5-NN launch above: the two share a numbering of `nn`)
        // a whole large batch: corner and surf features through their own specialisations of the fit (msfl_kernels.cuh)
        const int n_c = n_rec - bv.n_surf_total, n_s = bv.n_surf_total;
        const int edge_blocks = div_up(n_c, kAssocBlock), plane_blocks = div_up(n_s, kAssocBlock);
        // fallback lists (count + record numbers) of the deferred pivoted-QR fits: two, used alternately, each re-armed by the OTHER launch's
        // fallback kernel (zeroed at allocation)
        // (sized by ensure_fit_fallback in the callers, which can report an allocation failure)
        int* fb = h->fit_fallback.as<int>() + (size_t)h->fit_fallback_parity * h->fit_fallback_words;
        int* fb_next = h->fit_fallback.as<int>() + (size_t)(1 - h->fit_fallback_parity) * h->fit_fallback_words; (void)fb_next;
        h->fit_fallback_parity ^= 1;
        hipLaunchKernelGGL(fit_scan2map_split_kernel, dim3(edge_blocks + plane_blocks), block, 0, st, bv, h->map_c.sorted.as<float4>(),
                           h->map_s.sorted.as<float4>(), (const int*)nn, h->prm.line_eigen_ratio, h->prm.plane_tolerance, dv,
                           h->records.as<double>(), full, edge_blocks, fb);
#if MSFL_FIT_DEFER
        hipLaunchKernelGGL(fit_fallback_kernel, dim3(64), dim3(64), 0, st, bv, h->map_s.sorted.as<float4>(), (const int*)nn, h->prm.plane_tolerance,
                           h->records.as<double>(), full, (const int*)fb, fb_next);
#endif
      } else
      hipLaunchKernelGGL(fit_scan2map_kernel<false>, grid, block, 0, st, bv, h->map_c.sorted.as<float4>(),
                         h->map_s.sorted.as<float4>(), (const int*)nn, h->prm.line_eigen_ratio, h->prm.plane_tolerance, dv,
                         h->records.as<double>(), full);
  }
}

// Core of stage C.  All pointers are device pointers except the offsets (host).
// `n_chunks` > 1: the features of scans [chunk_b[c], chunk_b[c + 1]) are valid on the device once `chunk_ev[c]` has
// fired (host-buffer batches: they arrive over PCIe on another stream).  The first association pass then runs chunk by
// chunk behind the copies; everything after it sees the whole batch.
msfl_status match_scan2map_device(msfl_handle* h, int B, const float4* d_corner, const int* h_corner_off,
                                  const float4* d_surf, const int* h_surf_off, double* d_poses, int* d_status,
                                  DevMatchInfo* d_info, const DeskewView* deskew, int n_chunks = 1, const int* chunk_b = nullptr,
                                  hipEvent_t* chunk_ev = nullptr, const std::function<hipError_t(int)>* enqueue_chunk = nullptr) {
  hipStream_t st = h->stream;
  // offsets -> device: [corner_off (B+1) | surf_off (B+1) | rec_off (B+1)]
  std::vector<int> offs(3 * (size_t)(B + 1));
  for (int b = 0; b <= B; b++) {
    offs[b] = h_corner_off[b];
    offs[(B + 1) + b] = h_surf_off[b];
    offs[2 * (B + 1) + b] = h_corner_off[b] - h_corner_off[0] + h_surf_off[b] - h_surf_off[0];
    if (b > 0 && (h_corner_off[b] < h_corner_off[b - 1] || h_surf_off[b] < h_surf_off[b - 1]))
      return fail(h, MSFL_BAD_ARG, "offset arrays must be non-decreasing");
  }
  const int n_rec = offs[2 * (B + 1) + B];
  { const msfl_status us = upload_in_off(h, offs.data(), offs.size(), st); if (us) return us; }
  HIPCHK(h, h->records.reserve(std::max<size_t>(1, (size_t)n_rec) * 6 * sizeof(double)));
  HIPCHK(h, h->nn.reserve(std::max<size_t>(1, (size_t)n_rec) * 5 * sizeof(int)));
  BatchView bv;
  bv.corner = d_corner; bv.corner_off = h->in_off.as<int>();
  bv.surf = d_surf; bv.surf_off = h->in_off.as<int>() + (B + 1);
  bv.rec_off = h->in_off.as<int>() + 2 * (B + 1);
  bv.n_scans = B; bv.n_records = n_rec;
  bv.c0 = h_corner_off[0]; bv.s0 = h_surf_off[0]; bv.n_surf_total = h_surf_off[B] - h_surf_off[0];
  DeskewView dv{};
  if (deskew) {
    dv = *deskew;
    HIPCHK(h, h->pprime.reserve(std::max<size_t>(1, (size_t)n_rec) * 3 * sizeof(double)));
    dv.pprime = h->pprime.as<double>();
  }
  const SolverParams sp = solver_params(h->prm, 0);
  { const msfl_status fs = ensure_fit_fallback(h, bv.n_surf_total); if (fs) return fs; }
  for (int it = 0; it < h->prm.outer_iterations; it++) {
    if (it == 0 && (n_chunks > 1 || enqueue_chunk)) {
      for (int c = 0; c < n_chunks; c++) {
        // the copy of chunk c is ENQUEUED here, right before the kernels that wait for it: a copy from pageable memory
        // holds the calling thread until it is staged, and the kernels of chunk c - 1 must be in the queue by then
        if (enqueue_chunk) HIPCHK(h, (*enqueue_chunk)(c));
        HIPCHK(h, hipStreamWaitEvent(st, chunk_ev[c], 0));
        s_launch_assoc(h, bv, d_poses, d_status, deskew != nullptr, dv, n_rec, nullptr, offs[2 * (B + 1) + chunk_b[c]], offs[2 * (B + 1) + chunk_b[c + 1]]);
      }
    } else if (n_rec > 0) {
      s_launch_assoc(h, bv, d_poses, d_status, deskew != nullptr, dv, n_rec, nullptr, 0, -1, it > 0);
    }
    {
      ScopedTimer timer(h, T_SOLVE);
      hipLaunchKernelGGL(lm_solve_kernel<kLmBlock>, dim3(B), dim3(kLmBlock), 0, st, bv,
                         deskew ? (const double*)dv.pprime : (const double*)nullptr,
                         (const double*)h->records.as<double>(), d_poses, d_status, d_info, it, sp);
    }
  }
  HIPCHK(h, hipGetLastError());
  return MSFL_OK;
}

msfl_status check_map(msfl_handle* h) {
  if (!h->have_map) return fail(h, MSFL_NO_MAP, "msfl_set_map has not been called on this handle");
  // host-known input sizes; non-finite map points only ever reduce the candidate set (a query
  // with fewer than 5 finite neighbours is rejected on the device like any other)
  if (h->map_c.n_input < 5 || h->map_s.n_input < 5)
    return fail(h, MSFL_MAP_TOO_SMALL, "map corner/surf cloud has fewer than 5 points (mapping_scan_matcher.cc:128,198 would read out of bounds)");
  return MSFL_OK;
}

msfl_status enter(msfl_handle* h) {
  if (!h) return MSFL_BAD_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  return MSFL_OK;
}

}  // namespace

// =============================================================================================
// lifecycle
// =============================================================================================

extern "C" {

void msfl_default_params(msfl_params* p) {
  if (!p) return;
  p->scan_period = 0.1;
  p->min_range = 0.3;
  p->curvature_threshold = 0.1f;
  p->neighbor_gap_sq = 0.05f;
  p->sectors_per_ring = 6;
  p->max_sharp_per_sector = 2;
  p->max_less_sharp_per_sector = 20;
  p->max_flat_per_sector = 4;
  p->odom_distance_sq_threshold = 25.0;
  p->odom_nearby_scan = 2.5;
  p->odom_min_correspondences = 10;
  p->map_knn = 5;
  p->map_knn_max_sq_dist = 1.0f;
  p->line_eigen_ratio = 3.0;
  p->plane_tolerance = 0.2;
  p->outer_iterations = 2;
  p->max_lm_iterations = 6;
  p->huber_delta = 0.1;
  p->initial_trust_region_radius = 1e4;
  p->max_trust_region_radius = 1e16;
  p->min_trust_region_radius = 1e-32;
  p->min_relative_decrease = 1e-3;
  p->min_lm_diagonal = 1e-6;
  p->max_lm_diagonal = 1e32;
  p->function_tolerance = 1e-6;
  p->gradient_tolerance = 1e-10;
  p->parameter_tolerance = 1e-8;
  p->max_consecutive_invalid_steps = 5;
}

int msfl_api_version(void) { return MSFL_API_VERSION; }

msfl_status msfl_create(const msfl_params* params, int device, msfl_handle** out) {
  if (!out) return MSFL_BAD_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return MSFL_HIP_ERROR;   // no GPU: fail loudly, no fallback
  if (device < 0 || device >= ndev) return MSFL_BAD_ARG;
  msfl_handle* h = new msfl_handle_s();
  h->device = device;
  if (params) h->prm = *params; else msfl_default_params(&h->prm);
  if (h->prm.map_knn != 5 || h->prm.outer_iterations < 1 || h->prm.outer_iterations > 2 ||
      h->prm.max_lm_iterations < 0 || !(h->prm.map_knn_max_sq_dist > 0.f) ||
      h->prm.sectors_per_ring < 1 || h->prm.sectors_per_ring > 16) {
    delete h;
    return MSFL_BAD_ARG;
  }
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete h;
    return MSFL_HIP_ERROR;
  }
  h->stream = h->own_stream;
  if (const char* e = std::getenv("MSFL_GRID_CAP_CELLS")) { const int c = std::atoi(e); if (c >= 8 && c <= (1 << 28)) h->grid_cap_cells = c; }
  if (const char* e = std::getenv("MSFL_H2D_CHUNK_SCANS")) { const int c = std::atoi(e); if (c >= 1) h->h2d_chunk_scans = c; }
  if (const char* e = std::getenv("MSFL_H2D_SUB_CHUNKS")) { const int c = std::atoi(e); if (c >= 1) h->h2d_sub_chunks = c; }
  if (const char* e = std::getenv("MSFL_ODOM_BRUTE")) h->odom_force_brute = std::atoi(e) != 0;
  if (const char* e = std::getenv("MSFL_INDEX_SINGLE")) h->index_single = std::atoi(e) != 0;
  if (const char* e = std::getenv("MSFL_KNN_SEED")) h->knn_seed = std::atoi(e) != 0;
  if (const char* e = std::getenv("MSFL_KNN_FORM")) h->knn_form = !std::strcmp(e, "lane") ? 1 : !std::strcmp(e, "rows") ? 2 : 0;
  if (const char* e = std::getenv("MSFL_ODOM_WAVE_MAX_TARGETS")) h->odom_wave_max_targets = std::atoll(e);
  if (const char* e = std::getenv("MSFL_VOXEL_GLOBAL")) h->voxel_force_global = std::atoi(e) != 0;
  if (const char* e = std::getenv("MSFL_VOXEL_NO_BIG")) h->voxel_no_big = std::atoi(e) != 0;
  if (const char* e = std::getenv("MSFL_ODOM_BIN_SPLIT")) h->odom_bin_split = std::atoi(e) != 0 ? 1 : 0;
  if (const char* e = std::getenv("MSFL_PREP_SPLIT")) { const int g = std::atoi(e); if (g == 1 || g == 2 || g == 4 || g == 8 || g == 16) h->prep_split = g; }
  *out = h;
  return MSFL_OK;
}

void msfl_destroy(msfl_handle* h) {
  if (!h) return;
#ifdef MSFL_LM_PROFILE
  {
    unsigned long long v[8] = {0};
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(v, HIP_SYMBOL(msfl::g_lm_prof), sizeof(v));
    if (v[5]) fprintf(stderr, "[lm profile] per solve (100 MHz ticks): eval %.0f reduce %.0f serial %.0f total %.0f passes %.2f solves %llu\n",
                      (double)v[0] / v[5], (double)v[1] / v[5], (double)v[2] / v[5], (double)v[3] / v[5], (double)v[4] / v[5], v[5]);
    if (v[5]) fprintf(stderr, "[lm profile] lane-0 logic: tr_decide %.0f tr_propose %.0f ticks per solve\n", (double)v[6] / v[5], (double)v[7] / v[5]);
  }
#endif
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  collect_timing(h);
  for (auto e : h->free_events) (void)hipEventDestroy(e);
  h->pin.release();
  h->readback.release();
  if (h->copy_stream) { (void)hipStreamDestroy(h->copy_stream); for (auto e : h->copy_ev) if (e) (void)hipEventDestroy(e); }
  for (MapIndex* mi : {&h->map_c, &h->map_s}) {
    if (mi->want_pending) (void)hipEventSynchronize(mi->want_ev);
    if (mi->want_ev) (void)hipEventDestroy(mi->want_ev);
    if (mi->want_host) (void)hipHostFree(mi->want_host);
  }
  DevBuf* bufs[] = {&h->map_c.sorted, &h->map_c.cell_start, &h->map_s.sorted, &h->map_s.cell_start, &h->map_c.pos_of, &h->map_s.pos_of,
                    &h->map_c.gdesc, &h->map_s.gdesc, &h->map_c.bbox, &h->map_s.bbox, &h->in_corner,
                    &h->in_surf, &h->in_off, &h->poses, &h->status, &h->info, &h->records, &h->pprime, &h->nn,
                    &h->idx_cell_of, &h->idx_count, &h->idx_scanned, &h->idx_bbox, &h->idx_cub, &h->idx_stage, &h->knn_count};
  for (auto* b : bufs) b->release();
  for (auto& b : h->dk) b.release();
  for (auto& b : h->ex) b.release();
  for (auto& b : h->od) b.release();
  for (auto& b : h->pp) b.release();
  for (auto& b : h->pr) b.release();
  for (auto& b : h->vb) b.release();
  for (auto& b : h->vb2) b.release();
  h->vox_big_scratch.release(); h->vox_big_list.release(); h->fit_fallback.release();
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
}

msfl_status msfl_set_stream(msfl_handle* h, void* hip_stream) {
  msfl_status s = enter(h); if (s) return s;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->stream = reinterpret_cast<hipStream_t>(hip_stream);   // NULL = HIP null stream
  return MSFL_OK;
}

msfl_status msfl_reset_stream(msfl_handle* h) {
  msfl_status s = enter(h); if (s) return s;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->stream = h->own_stream;
  return MSFL_OK;
}

msfl_status msfl_synchronize(msfl_handle* h) {
  msfl_status s = enter(h); if (s) return s;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MSFL_OK;
}

const char* msfl_status_string(int status) {
  switch (status) {
    case MSFL_OK: return "OK";
    case MSFL_TOO_FEW_CORRESPONDENCES: return "TOO_FEW_CORRESPONDENCES";
    case MSFL_MAP_TOO_SMALL: return "MAP_TOO_SMALL";
    case MSFL_BAD_ARG: return "BAD_ARG";
    case MSFL_HIP_ERROR: return "HIP_ERROR";
    case MSFL_BAD_RING: return "BAD_RING";
    case MSFL_NO_MAP: return "NO_MAP";
    case MSFL_CAPACITY: return "CAPACITY";
    default: return "UNKNOWN";
  }
}

const char* msfl_last_error(const msfl_handle* h) { return h ? h->last_error.c_str() : ""; }

msfl_status msfl_set_timing(msfl_handle* h, int enabled) {
  msfl_status s = enter(h); if (s) return s;
  h->timing = enabled < 0 ? 0 : (enabled > 3 ? 1 : enabled);
  if (h->timing == 3 && !h->knn_count.p) {
    HIPCHK(h, h->knn_count.reserve(2 * sizeof(unsigned long long)));     // [first-pass launches, seeded launches]
    HIPCHK(h, hipMemsetAsync(h->knn_count.p, 0, 2 * sizeof(unsigned long long), h->stream));
  }
  return MSFL_OK;
}

msfl_status msfl_get_timing(msfl_handle* h, msfl_timing* out, int reset) {
  msfl_status s = enter(h); if (s) return s;
  if (!out) return MSFL_BAD_ARG;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  collect_timing(h);
  out->launches_assoc = h->t_n[T_ASSOC];     out->ms_assoc = h->t_ms[T_ASSOC];
  out->launches_solve = h->t_n[T_SOLVE];     out->ms_solve = h->t_ms[T_SOLVE];
  out->launches_index = h->t_n[T_INDEX];     out->ms_index = h->t_ms[T_INDEX];
  out->launches_extract = h->t_n[T_EXTRACT]; out->ms_extract = h->t_ms[T_EXTRACT];
  out->launches_odom = h->t_n[T_ODOM];       out->ms_odom = h->t_ms[T_ODOM];
  out->launches_fit = h->t_n[T_FIT];         out->ms_fit = h->t_ms[T_FIT];
  out->launches_assoc_seeded = h->t_n[T_ASSOC_SEEDED]; out->ms_assoc_seeded = h->t_ms[T_ASSOC_SEEDED];
  out->knn_candidates = 0; out->knn_candidates_seeded = 0;
  if (h->knn_count.p) {
    unsigned long long c2[2] = {0, 0};
    HIPCHK(h, hipMemcpy(c2, h->knn_count.p, sizeof(c2), hipMemcpyDeviceToHost));
    out->knn_candidates = c2[0]; out->knn_candidates_seeded = c2[1];
    if (reset) HIPCHK(h, hipMemset(h->knn_count.p, 0, sizeof(c2)));
  }
  if (reset) for (int i = 0; i < T_COUNT; i++) { h->t_ms[i] = 0; h->t_n[i] = 0; }
  return MSFL_OK;
}

// =============================================================================================
// stage C
// =============================================================================================

msfl_status msfl_set_map(msfl_handle* h, const msfl_point* corner, int n_corner, const msfl_point* surf, int n_surf,
                         msfl_mem mem) {
  msfl_status s = enter(h); if (s) return s;
  if (n_corner < 0 || n_surf < 0 || (n_corner > 0 && !corner) || (n_surf > 0 && !surf))
    return fail(h, MSFL_BAD_ARG, "msfl_set_map: null cloud with positive size");
  h->have_map = false;
  const float4* dc = reinterpret_cast<const float4*>(corner);
  const float4* ds = reinterpret_cast<const float4*>(surf);
  if (mem == MSFL_MEM_HOST) {
    HIPCHK(h, h->idx_stage.reserve(std::max<size_t>(1, (size_t)n_corner + (size_t)n_surf) * sizeof(float4)));
    float4* stage = h->idx_stage.as<float4>();
    if (n_corner) HIPCHK(h, hipMemcpyAsync(stage, corner, (size_t)n_corner * sizeof(float4), hipMemcpyHostToDevice, h->stream));
    if (n_surf) HIPCHK(h, hipMemcpyAsync(stage + n_corner, surf, (size_t)n_surf * sizeof(float4), hipMemcpyHostToDevice, h->stream));
    dc = stage; ds = stage + n_corner;
  }
  if (n_corner > 0 && n_surf > 0 && !h->index_single) {
    s = build_index_pair(h, dc, n_corner, ds, n_surf); if (s) return s;
  } else {
    s = build_index(h, dc, n_corner, h->map_c); if (s) return s;
    s = build_index(h, ds, n_surf, h->map_s); if (s) return s;
  }
  // host arrays were staged with asynchronous copies from pageable memory: they must have been read
  // before the caller may touch them again.  Device-resident maps stay fully asynchronous.
  if (mem == MSFL_MEM_HOST) HIPCHK(h, hipStreamSynchronize(h->stream));
  h->have_map = true;
  return MSFL_OK;
}

static msfl_status match_batch_impl(msfl_handle* h, int B, const msfl_point* corner, const int* corner_off,
                                    const msfl_point* surf, const int* surf_off, double* poses_io, int* status,
                                    msfl_match_info* info, msfl_mem mem, const msfl_deskew_batch* deskew) {
  if (B < 0 || (B > 0 && (!corner_off || !surf_off || !poses_io)))
    return fail(h, MSFL_BAD_ARG, "msfl_match_scan2map_batch: null argument");
  if (B == 0) return MSFL_OK;
  msfl_status s = check_map(h); if (s) return s;
  hipStream_t st = h->stream;
  const int c0 = corner_off[0], s0 = surf_off[0];
  const int ncp = corner_off[B] - c0, nsp = surf_off[B] - s0;
  if (ncp < 0 || nsp < 0 || (ncp > 0 && !corner) || (nsp > 0 && !surf))
    return fail(h, MSFL_BAD_ARG, "msfl_match_scan2map_batch: bad offsets or null feature array");
  const float4* d_corner; const float4* d_surf; double* d_poses; int* d_status;
  std::vector<int> co(corner_off, corner_off + B + 1), so(surf_off, surf_off + B + 1);
  // Host buffers, large batch, plain branch: the features go over PCIe in chunks of >= 512 scans on a second
  // stream while the previous chunk is being registered (82 MB per 1 024 scans is as long as the compute).
  const int n_chunks = (mem == MSFL_MEM_HOST && !deskew && B >= 2 * h->h2d_chunk_scans) ? std::min(8, B / h->h2d_chunk_scans) : 1;   // copy_ev[8] is the fork event
  if (mem == MSFL_MEM_HOST) {
    HIPCHK(h, h->in_corner.reserve(std::max<size_t>(1, (size_t)ncp) * sizeof(float4)));
    HIPCHK(h, h->in_surf.reserve(std::max<size_t>(1, (size_t)nsp) * sizeof(float4)));
    HIPCHK(h, h->poses.reserve((size_t)B * 7 * sizeof(double)));
    if (n_chunks == 1) {
      if (ncp) HIPCHK(h, hipMemcpyAsync(h->in_corner.p, corner + c0, (size_t)ncp * sizeof(float4), hipMemcpyHostToDevice, st));
      if (nsp) HIPCHK(h, hipMemcpyAsync(h->in_surf.p, surf + s0, (size_t)nsp * sizeof(float4), hipMemcpyHostToDevice, st));
    }
    HIPCHK(h, hipMemcpyAsync(h->poses.p, poses_io, (size_t)B * 7 * sizeof(double), hipMemcpyHostToDevice, st));
    for (int b = 0; b <= B; b++) { co[b] -= c0; so[b] -= s0; }
    d_corner = h->in_corner.as<float4>(); d_surf = h->in_surf.as<float4>(); d_poses = h->poses.as<double>();
  } else {
    d_corner = reinterpret_cast<const float4*>(corner); d_surf = reinterpret_cast<const float4*>(surf);
    d_poses = poses_io;
  }
  if (mem == MSFL_MEM_DEVICE && status) {
    d_status = status;
  } else {
    HIPCHK(h, h->status.reserve((size_t)B * sizeof(int)));
    d_status = h->status.as<int>();
  }
  // (a kernel, not hipMemsetAsync: replayed from a captured HIP graph the memset node of a 24-byte status array wrote host-pointer
  // patterns into it from the second replay on -- ROCm 7.2, tests/test_gpu_scan2map.py::test_the_batch_step_replays_from_a_captured_graph)
  hipLaunchKernelGGL(zero_ints_kernel, dim3(div_up(B, 256)), dim3(256), 0, st, d_status, B);
  DevMatchInfo* d_info = nullptr;
  if (info) {
    static_assert(sizeof(DevMatchInfo) == sizeof(msfl_match_info), "info layout");
    HIPCHK(h, h->info.reserve((size_t)B * sizeof(DevMatchInfo)));
    HIPCHK(h, hipMemsetAsync(h->info.p, 0, (size_t)B * sizeof(DevMatchInfo), st));
    d_info = h->info.as<DevMatchInfo>();
  }
  DeskewView dv{}; const DeskewView* dvp = nullptr;
  if (deskew) {
    // the four per-feature arrays are indexed like the feature arrays (corner_off / surf_off), velocity per scan
    const double* src[5] = {deskew->corner_dq, deskew->corner_dp, deskew->surf_dq, deskew->surf_dp, deskew->velocity};
    const size_t nb[5] = {(size_t)ncp * 4, (size_t)ncp * 3, (size_t)nsp * 4, (size_t)nsp * 3, (size_t)B * 3};
    const size_t skip[5] = {(size_t)c0 * 4, (size_t)c0 * 3, (size_t)s0 * 4, (size_t)s0 * 3, 0};
    const double* dev[5];
    for (int k = 0; k < 5; k++) {
      if (nb[k] && !src[k]) return fail(h, MSFL_BAD_ARG, "msfl_match_scan2map_deskew: null deskew array");
      if (mem == MSFL_MEM_HOST) {
        HIPCHK(h, h->dk[k].reserve(std::max<size_t>(1, nb[k]) * sizeof(double)));
        if (nb[k]) HIPCHK(h, hipMemcpyAsync(h->dk[k].p, src[k] + skip[k], nb[k] * sizeof(double), hipMemcpyHostToDevice, st));
        dev[k] = h->dk[k].as<double>();
      } else {
        dev[k] = src[k];
      }
    }
    dv.corner_dq = dev[0]; dv.corner_dp = dev[1]; dv.surf_dq = dev[2]; dv.surf_dp = dev[3]; dv.V = dev[4];
    for (int a = 0; a < 3; a++) dv.G[a] = deskew->gravity[a];
    dvp = &dv;
  }
  if (n_chunks == 1) {
    s = match_scan2map_device(h, B, d_corner, co.data(), d_surf, so.data(), d_poses, d_status, d_info, dvp);
    if (s) return s;
  } else {
    if (!h->copy_stream) {
      HIPCHK(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
      for (auto& e : h->copy_ev) HIPCHK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    // the staging buffers may still be read by work queued earlier on the compute stream
    HIPCHK(h, hipEventRecord(h->copy_ev[8], st));
    HIPCHK(h, hipStreamWaitEvent(h->copy_stream, h->copy_ev[8], 0));
    // Two levels.  PARTS of >= h2d_chunk_scans scans are registered one after the other, each completely (a solve over a
    // fraction of the batch takes as long as one over all of it, so there are few parts): part p + 1 crosses PCIe while
    // part p is being registered.  Inside a part the FIRST association pass follows its SUB-chunks as they land, so a
    // part's kernels do not wait for all of its features either.
    const int n_sub = std::max(1, std::min(h->h2d_sub_chunks, 8));
    for (int p = 0; p < n_chunks; p++) {
      const int pb0 = (int)((long long)B * p / n_chunks), pb1 = (int)((long long)B * (p + 1) / n_chunks), Bp = pb1 - pb0;
      const int ns_p = std::max(1, std::min(n_sub, Bp / 32));
      int sub_b[9];
      for (int c = 0; c <= ns_p; c++) sub_b[c] = (int)((long long)Bp * c / ns_p);
      const std::function<hipError_t(int)> enqueue_chunk = [&](int c) -> hipError_t {
        const int b0 = pb0 + sub_b[c], b1 = pb0 + sub_b[c + 1];
        const size_t nc = (size_t)(co[b1] - co[b0]), ns = (size_t)(so[b1] - so[b0]);
        hipError_t e = hipSuccess;
        if (nc) e = hipMemcpyAsync(h->in_corner.as<float4>() + co[b0], corner + c0 + co[b0], nc * sizeof(float4), hipMemcpyHostToDevice, h->copy_stream);
        if (ns && e == hipSuccess) e = hipMemcpyAsync(h->in_surf.as<float4>() + so[b0], surf + s0 + so[b0], ns * sizeof(float4), hipMemcpyHostToDevice, h->copy_stream);
        if (e == hipSuccess) e = hipEventRecord(h->copy_ev[c], h->copy_stream);
        return e;
      };
      s = match_scan2map_device(h, Bp, d_corner, co.data() + pb0, d_surf, so.data() + pb0, d_poses + 7 * (size_t)pb0, d_status + pb0,
                                d_info ? d_info + pb0 : nullptr, nullptr, ns_p, sub_b, h->copy_ev, &enqueue_chunk);
      if (s) return s;
    }
  }
  if (mem == MSFL_MEM_HOST) {
    HIPCHK(h, hipMemcpyAsync(poses_io, d_poses, (size_t)B * 7 * sizeof(double), hipMemcpyDeviceToHost, st));
    if (status) HIPCHK(h, hipMemcpyAsync(status, d_status, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, st));
  }
  if (info) HIPCHK(h, hipMemcpyAsync(info, d_info, (size_t)B * sizeof(DevMatchInfo), hipMemcpyDeviceToHost, st));
  if (mem == MSFL_MEM_HOST || info) HIPCHK(h, hipStreamSynchronize(st));
  return MSFL_OK;
}

msfl_status msfl_match_scan2map_batch(msfl_handle* h, int n_scans, const msfl_point* corner, const int* corner_off,
                                      const msfl_point* surf, const int* surf_off, double* poses_io, int* status,
                                      msfl_match_info* info, msfl_mem mem) {
  msfl_status s = enter(h); if (s) return s;
  return match_batch_impl(h, n_scans, corner, corner_off, surf, surf_off, poses_io, status, info, mem, nullptr);
}

msfl_status msfl_match_scan2map(msfl_handle* h, const msfl_point* corner, int n_corner, const msfl_point* surf,
                                int n_surf, double pose_io[7], msfl_match_info* info, msfl_mem mem) {
  msfl_status s = enter(h); if (s) return s;
  if (n_corner < 0 || n_surf < 0 || !pose_io) return fail(h, MSFL_BAD_ARG, "msfl_match_scan2map: bad argument");
  const int co[2] = {0, n_corner}, so[2] = {0, n_surf};
  int st = 0;
  int* stp = (mem == MSFL_MEM_HOST) ? &st : nullptr;
  s = match_batch_impl(h, 1, corner, co, surf, so, pose_io, stp, info, mem, nullptr);
  if (s) return s;
  return (msfl_status)st;
}

msfl_status msfl_match_scan2map_deskew(msfl_handle* h, const msfl_point* corner, int n_corner, const msfl_point* surf,
                                       int n_surf, const msfl_deskew* deskew, double pose_io[7], msfl_match_info* info) {
  msfl_status s = enter(h); if (s) return s;
  if (n_corner < 0 || n_surf < 0 || !pose_io || !deskew) return fail(h, MSFL_BAD_ARG, "msfl_match_scan2map_deskew: bad argument");
  const int co[2] = {0, n_corner}, so[2] = {0, n_surf};
  int st = 0;
  msfl_deskew_batch db;
  db.corner_dq = deskew->corner_dq; db.corner_dp = deskew->corner_dp; db.surf_dq = deskew->surf_dq; db.surf_dp = deskew->surf_dp;
  db.velocity = deskew->velocity;
  for (int a = 0; a < 3; a++) db.gravity[a] = deskew->gravity[a];
  s = match_batch_impl(h, 1, corner, co, surf, so, pose_io, &st, info, MSFL_MEM_HOST, &db);
  if (s) return s;
  return (msfl_status)st;
}

msfl_status msfl_match_scan2map_deskew_batch(msfl_handle* h, int n_scans, const msfl_point* corner, const int* corner_off,
                                             const msfl_point* surf, const int* surf_off, const msfl_deskew_batch* deskew,
                                             double* poses_io, int* status, msfl_match_info* info, msfl_mem mem) {
  msfl_status s = enter(h); if (s) return s;
  if (!deskew) return fail(h, MSFL_BAD_ARG, "msfl_match_scan2map_deskew_batch: null deskew");
  return match_batch_impl(h, n_scans, corner, corner_off, surf, surf_off, poses_io, status, info, mem, deskew);
}

static msfl_status stage_single(msfl_handle* h, const msfl_point* corner, int n_corner, const msfl_point* surf, int n_surf,
                                const double* pose, BatchView& bv) {
  hipStream_t st = h->stream;
  HIPCHK(h, h->in_corner.reserve(std::max<size_t>(1, (size_t)n_corner) * sizeof(float4)));
  HIPCHK(h, h->in_surf.reserve(std::max<size_t>(1, (size_t)n_surf) * sizeof(float4)));
  HIPCHK(h, h->poses.reserve(7 * sizeof(double)));
  HIPCHK(h, h->status.reserve(sizeof(int)));

  HIPCHK(h, h->records.reserve(std::max<size_t>(1, (size_t)(n_corner + n_surf)) * 6 * sizeof(double)));
  HIPCHK(h, h->nn.reserve(std::max<size_t>(1, (size_t)(n_corner + n_surf)) * 5 * sizeof(int)));
  if (n_corner) HIPCHK(h, hipMemcpyAsync(h->in_corner.p, corner, (size_t)n_corner * sizeof(float4), hipMemcpyHostToDevice, st));
  if (n_surf) HIPCHK(h, hipMemcpyAsync(h->in_surf.p, surf, (size_t)n_surf * sizeof(float4), hipMemcpyHostToDevice, st));
  HIPCHK(h, hipMemcpyAsync(h->poses.p, pose, 7 * sizeof(double), hipMemcpyHostToDevice, st));
  HIPCHK(h, hipMemsetAsync(h->status.p, 0, sizeof(int), st));
  const int offs[6] = {0, n_corner, 0, n_surf, 0, n_corner + n_surf};
  { const msfl_status us = upload_in_off(h, offs, 6, st); if (us) return us; }
  bv.corner = h->in_corner.as<float4>(); bv.corner_off = h->in_off.as<int>();
  bv.surf = h->in_surf.as<float4>(); bv.surf_off = h->in_off.as<int>() + 2;
  bv.rec_off = h->in_off.as<int>() + 4;
  bv.n_scans = 1; bv.n_records = n_corner + n_surf;
  bv.c0 = 0; bv.s0 = 0; bv.n_surf_total = n_surf;
  return MSFL_OK;
}

msfl_status msfl_associate_scan2map(msfl_handle* h, const msfl_point* corner, int n_corner, const msfl_point* surf,
                                    int n_surf, const double pose[7], double* records_out) {
  msfl_status s = enter(h); if (s) return s;
  if (n_corner < 0 || n_surf < 0 || !pose || (n_corner + n_surf > 0 && !records_out) || (n_corner && !corner) || (n_surf && !surf))
    return fail(h, MSFL_BAD_ARG, "msfl_associate_scan2map: bad argument");
  s = check_map(h); if (s) return s;
  const int n = n_corner + n_surf;
  if (n == 0) return MSFL_OK;
  BatchView bv;
  s = stage_single(h, corner, n_corner, surf, n_surf, pose, bv); if (s) return s;
  DeskewView dv{};
  HIPCHK(h, h->pprime.reserve((size_t)n * 6 * sizeof(double)));   // {C,N} staging for the host-format output
  s = ensure_fit_fallback(h, bv.n_surf_total); if (s) return s;
  s_launch_assoc(h, bv, h->poses.as<double>(), h->status.as<int>(), false, dv, n, h->pprime.as<double>());
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemcpyAsync(records_out, h->pprime.p, (size_t)n * 6 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MSFL_OK;
}

msfl_status msfl_solve_records(msfl_handle* h, const msfl_point* corner, int n_corner, const msfl_point* surf, int n_surf,
                               const double* records, double pose_io[7], msfl_match_info* info) {
  msfl_status s = enter(h); if (s) return s;
  if (n_corner < 0 || n_surf < 0 || !pose_io || (n_corner + n_surf > 0 && !records) || (n_corner && !corner) || (n_surf && !surf))
    return fail(h, MSFL_BAD_ARG, "msfl_solve_records: bad argument");
  const int n = n_corner + n_surf;
  BatchView bv;
  s = stage_single(h, corner, n_corner, surf, n_surf, pose_io, bv); if (s) return s;
  if (n) {
    HIPCHK(h, h->pprime.reserve((size_t)n * 6 * sizeof(double)));
    HIPCHK(h, hipMemcpyAsync(h->pprime.p, records, (size_t)n * 6 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(pack_records_kernel, dim3(div_up(n, 256)), dim3(256), 0, h->stream, bv, (const double*)h->pprime.as<double>(),
                       h->records.as<double>());
  }
  DevMatchInfo* d_info = nullptr;
  if (info) {
    HIPCHK(h, h->info.reserve(sizeof(DevMatchInfo)));
    HIPCHK(h, hipMemsetAsync(h->info.p, 0, sizeof(DevMatchInfo), h->stream));
    d_info = h->info.as<DevMatchInfo>();
  }
  {
    ScopedTimer timer(h, T_SOLVE);
    hipLaunchKernelGGL(lm_solve_kernel<kLmBlock>, dim3(1), dim3(kLmBlock), 0, h->stream, bv, (const double*)nullptr,
                       (const double*)h->records.as<double>(), h->poses.as<double>(), h->status.as<int>(), d_info, 0,
                       solver_params(h->prm, 0));
  }
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemcpyAsync(pose_io, h->poses.p, 7 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (info) HIPCHK(h, hipMemcpyAsync(info, d_info, sizeof(DevMatchInfo), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MSFL_OK;
}

}  // extern "C"

#include "msfl_api_stage_ab.inc"
#include "msfl_api_grid.inc"
#include "msfl_api_deskew.inc"
#include "msfl_api_slam.inc"
#include "msfl_api_pairs.inc"

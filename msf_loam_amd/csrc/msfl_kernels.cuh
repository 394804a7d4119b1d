// msfl_kernels.cuh — HIP kernels of the scan-to-map registration path (gfx950 / CDNA4).
//
//   K3  map grid index      replaces pcl::KdTreeFLANN::setInputCloud   mapping_scan_matcher.cc:66-73
//   K4  assoc_scan2map      replaces the two association loops         mapping_scan_matcher.cc:109-246
//   K5+K6 lm_solve          replaces ceres::Solve (LM, Huber)          mapping_scan_matcher.cc:250-272
//
// Data layout in HBM:
//   features     float4 {x,y,z,t} per point, scans concatenated, B+1 prefix offsets
//   map (sorted) float4 {x,y,z, bits(original index)} ordered by grid cell (x fastest)
//   cell_start   int[n_cells+1]
//   records      per scan: n_corner edge records {C[3], N[3]} (48 B) followed by n_surf plane records
//                {N[3], N.C} (32 B); N == 0 marks a rejected correspondence.  Scan b starts at
//                6*(corner_off[b]-corner_off[0]) + 4*(surf_off[b]-surf_off[0]) doubles.
//   poses        7 doubles per scan
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "msfl_math.cuh"

namespace msfl {

// ---------------------------------------------------------------------------------------------
// K3: uniform-grid index.  Cell edge >= 1.001 * sqrt(max_sq_dist) so the 27-cell neighbourhood
// of a query contains every map point whose f32 distance can pass the reference's
// `pointSearchSqDis[4] < 1.0` gate: exact-kNN-equivalent on every ACCEPTED query.
// ---------------------------------------------------------------------------------------------
#ifndef MSFL_GRID_XSUB
#define MSFL_GRID_XSUB 3
#endif
constexpr int kGridXSub = MSFL_GRID_XSUB;

struct GridDesc {
  float ox, oy, oz;   // origin = bbox min
  float inv_cell;     // 1 / cell edge in y and z
  float inv_cell_x;   // cells are kGridXSub times finer along x (the fastest index): a (y,z) row of 2*kGridXSub+1 cells
                      // is still ONE contiguous range, but its ends can be trimmed in finer steps
  int dx, dy, dz;
  int n_pts;          // finite points indexed
  int n_cells;
  int reach;          // cells scanned on each side of the query cell (1)
  int want_cells;     // cells the bbox needs at the base cell edge (saturated); host feedback for the next build
  float cell2, cellx2;  // (1 / inv_cell)^2 and (1 / inv_cell_x)^2 in f32: every query needs them, computed once per build
};

// order-preserving float <-> int encoding for atomicMin/Max
__device__ __forceinline__ int float_to_ordered(float f) {
  const int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7fffffff;
}
__host__ __device__ __forceinline__ float ordered_to_float(int i) {
  const int j = i >= 0 ? i : i ^ 0x7fffffff;
#if defined(__HIP_DEVICE_COMPILE__)
  return __int_as_float(j);
#else
  float f; __builtin_memcpy(&f, &j, 4); return f;
#endif
}

// bbox[0..2] = min (ordered ints), bbox[3..5] = max; pre-initialised to INT_MAX / INT_MIN
// n_dev (optional): the number of valid points lives on the device (a surrounded cloud that never visits the host);
// `n` is then the launch's upper bound
__device__ __forceinline__ void grid_bbox_body(const float4* __restrict__ pts, int n, int* __restrict__ bbox, int block, int n_blocks) {
  __shared__ float s_mn[4][3], s_mx[4][3];
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  auto take = [&](float4 p) {
    if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
      mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
      mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
    }
  };
  const int stride = n_blocks * blockDim.x;
  int i = block * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {       // four independent loads in flight per lane
    const float4 p0 = pts[i], p1 = pts[i + stride], p2 = pts[i + 2 * stride], p3 = pts[i + 3 * stride];
    take(p0); take(p1); take(p2); take(p3);
  }
  for (; i < n; i += stride) take(pts[i]);
#pragma unroll
  for (int a = 0; a < 3; a++) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mn[a] = fminf(mn[a], __shfl_xor(mn[a], o));
      mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o));
    }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < 3; a++) { s_mn[wave][a] = mn[a]; s_mx[wave][a] = mx[a]; }
  }
  __syncthreads();
  // six lanes, one atomic each per workgroup (they all land on one cache line: keep them few)
  if (threadIdx.x < 3) {
    const int a = threadIdx.x;
    const float v = fminf(fminf(s_mn[0][a], s_mn[1][a]), fminf(s_mn[2][a], s_mn[3][a]));
    if (v != INFINITY) atomicMin(&bbox[a], float_to_ordered(v));
  } else if (threadIdx.x < 6) {
    const int a = threadIdx.x - 3;
    const float v = fmaxf(fmaxf(s_mx[0][a], s_mx[1][a]), fmaxf(s_mx[2][a], s_mx[3][a]));
    if (v != -INFINITY) atomicMax(&bbox[3 + a], float_to_ordered(v));
  }
}
__global__ void __launch_bounds__(256) grid_bbox_kernel(const float4* __restrict__ pts, int n, int* __restrict__ bbox, const int* __restrict__ n_dev = nullptr) {
  if (n_dev) n = min(n, max(*n_dev, 0));
  grid_bbox_body(pts, n, bbox, (int)blockIdx.x, (int)gridDim.x);
}

__device__ __forceinline__ int grid_coord(float v, float o, float inv, int dim) {
  // clamp in float first: far-away queries must not overflow the int conversion
  float u = floorf((v - o) * inv);
  u = fminf(fmaxf(u, -2.0f), (float)dim + 1.0f);
  return (int)u;
}

__device__ __forceinline__ void grid_desc_squares(GridDesc& g) {
  const float cell = 1.0f / g.inv_cell, cellx = 1.0f / g.inv_cell_x;
  g.cell2 = cell * cell; g.cellx2 = cellx * cellx;
}
// bbox -> grid descriptor, entirely on the device so that msfl_set_map needs no host round trip.
// Cell edge = 1.001 * acceptance radius, grown by 26 % steps until the dense table fits `cap_cells`
// (larger cells stay exact).  An empty cloud yields n_cells = 1, n_pts = 0.
__device__ __forceinline__ GridDesc grid_desc_from_bbox(const int* __restrict__ bbox, double radius, int cap_cells) {
  GridDesc g;
  const int b0 = bbox[0];
  g.n_pts = 0; g.reach = 1;
  if (b0 == 0x7fffffff) {            // no finite point
    g.ox = g.oy = g.oz = 0.f; g.inv_cell = 1.f; g.inv_cell_x = (float)kGridXSub; g.dx = g.dy = g.dz = 1; g.n_cells = 1; g.want_cells = 1;
    grid_desc_squares(g);
    return g;
  }
  float mn[3], mx[3];
  for (int a = 0; a < 3; a++) { mn[a] = ordered_to_float(bbox[a]); mx[a] = ordered_to_float(bbox[3 + a]); }
  double cell = 1.001 * radius;
  int dims[3];
  bool first = true;
  g.want_cells = 1;
  for (;;) {
    double total = 1.0;
    for (int a = 0; a < 3; a++) {
      const double edge = a == 0 ? cell / kGridXSub : cell;
      dims[a] = (int)floor(((double)mx[a] - (double)mn[a]) / edge) + 2;   // +1 spare cell: f32 rounding of (v-o)*inv
      if (dims[a] < 2) dims[a] = 2;
      total *= dims[a];
    }
    if (first) { g.want_cells = total < 2.0e9 ? (int)total : 2000000000; first = false; }
    if (total <= (double)cap_cells) break;
    cell *= 1.26;
  }
  g.ox = mn[0]; g.oy = mn[1]; g.oz = mn[2];
  g.inv_cell = (float)(1.0 / cell);
  g.inv_cell_x = (float)((double)kGridXSub / cell);
  g.dx = dims[0]; g.dy = dims[1]; g.dz = dims[2];
  g.n_cells = g.dx * g.dy * g.dz;
  grid_desc_squares(g);
  return g;
}

__device__ __forceinline__ void grid_bbox_rearm(int* __restrict__ bbox) {
  for (int a = 0; a < 3; a++) { bbox[a] = 0x7fffffff; bbox[3 + a] = (int)0x80000000; }
}

// Only launched for an EMPTY cloud (nothing else runs then); a non-empty build derives the descriptor
// inside grid_count_kernel and re-arms the bbox in grid_scatter_kernel.
__global__ void grid_setup_kernel(int* __restrict__ bbox, double radius, int cap_cells, GridDesc* __restrict__ out) {
  *out = grid_desc_from_bbox(bbox, radius, cap_cells);
  grid_bbox_rearm(bbox);
}

// Every workgroup derives the (identical) descriptor from the finished bbox itself: one launch less
// per build than a separate one-thread setup kernel; workgroup 0 publishes it.
__device__ __forceinline__ void grid_count_body(const float4* __restrict__ pts, int n, const int* __restrict__ bbox,
                                                double radius, int cap_cells, GridDesc* __restrict__ gout,
                                                int* __restrict__ cell_of, int* __restrict__ count, int block) {
  __shared__ GridDesc s_g;
  if (threadIdx.x == 0) {
    s_g = grid_desc_from_bbox(bbox, radius, cap_cells);
    if (block == 0) *gout = s_g;
  }
  __syncthreads();
  const int i = block * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const GridDesc g = s_g;
  const float4 p = pts[i];
  int c = -1;
  if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
    int cx = grid_coord(p.x, g.ox, g.inv_cell_x, g.dx); cx = min(max(cx, 0), g.dx - 1);
    int cy = grid_coord(p.y, g.oy, g.inv_cell, g.dy); cy = min(max(cy, 0), g.dy - 1);
    int cz = grid_coord(p.z, g.oz, g.inv_cell, g.dz); cz = min(max(cz, 0), g.dz - 1);
    c = (cz * g.dy + cy) * g.dx + cx;
    atomicAdd(&count[c], 1);
  }
  cell_of[i] = c;
}
__global__ void __launch_bounds__(256) grid_count_kernel(const float4* __restrict__ pts, int n, const int* __restrict__ bbox,
                                                          double radius, int cap_cells, GridDesc* __restrict__ gout,
                                                          int* __restrict__ cell_of, int* __restrict__ count, const int* __restrict__ n_dev = nullptr) {
  if (n_dev) n = min(n, max(*n_dev, 0));
  grid_count_body(pts, n, bbox, radius, cap_cells, gout, cell_of, count, (int)blockIdx.x);
}

// cursor[] holds the per-cell counts on entry and is consumed — it is all zeros again afterwards, so
// the next build needs no memset; the order inside a cell is arbitrary, which is harmless because
// the kNN selection uses the total order (d2, original index).
__global__ void __launch_bounds__(256) grid_scatter_kernel(const float4* __restrict__ pts, int n,
                                                            const int* __restrict__ cell_of,
                                                            const int* __restrict__ cell_start, int* __restrict__ cursor,
                                                            float4* __restrict__ sorted, int* __restrict__ pos_of,
                                                            GridDesc* __restrict__ g, int* __restrict__ bbox, const int* __restrict__ n_dev = nullptr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (n_dev) n = min(n, max(*n_dev, 0));
  if (i == 0) { g->n_pts = cell_start[g->n_cells]; grid_bbox_rearm(bbox); }     // number of indexed (finite) points; bbox ready for the next build
  if (i >= n) return;
  const int c = cell_of[i];
  if (c < 0) { pos_of[i] = -1; return; }
  const int k = atomicSub(&cursor[c], 1) - 1;
  float4 p = pts[i];
  p.w = __int_as_float(i);
  sorted[cell_start[c] + k] = p;
  pos_of[i] = cell_start[c] + k;      // original index -> position (the fit kernel fetches by index)
}

// ---- both maps of msfl_set_map through ONE chain of launches (round 6) -------------------------------------------------------
// The corner map (a few thousand points) and the surf map used to be indexed one after the other: ten launches and two 4-byte
// read-back copies in front of every batch, ~45 us of mostly launch-to-launch latency.  Here every launch serves both clouds
// (blocks [0, blocks0) the first, the rest the second), the two count tables sit back to back and are prefix-summed by ONE
// rocPRIM scan, and the scatter launch also writes each map's own cell table from the combined sums (the second map's minus the
// first map's total) and the grids' wanted size straight into pinned host memory.  Same descriptors, same tables, same sorted
// copies up to the (arbitrary, never observable) order inside a cell as two single builds.
struct GridPairJob {
  const float4* pts[2]; int n[2];
  int* bbox[2]; GridDesc* gdesc[2]; int cap[2];
  int* cell_of;              // [n[0] | n[1]]
  int* count;                // [cap[0] + 1 | cap[1] + 1], zero on entry, zero again on exit
  int* scanned;              // exclusive prefix sums of `count` over both tables (the scatter launch reads them)
  int* cell_start[2]; float4* sorted[2]; int* pos_of[2];
  int* want_host[2];         // pinned host words: cells the bounding box needs at the base cell edge (feedback for the next build)
  int blocks0;               // workgroups of the per-point launches that serve map 0
  int bbox_blocks0;          // the same for the bounding-box launch
  double radius;
};
__global__ void __launch_bounds__(256) grid_bbox_pair_kernel(GridPairJob j) {
  const int m = (int)blockIdx.x < j.bbox_blocks0 ? 0 : 1;
  const int block = m ? (int)blockIdx.x - j.bbox_blocks0 : (int)blockIdx.x;
  grid_bbox_body(j.pts[m], j.n[m], j.bbox[m], block, m ? (int)gridDim.x - j.bbox_blocks0 : j.bbox_blocks0);
}
__global__ void __launch_bounds__(256) grid_count_pair_kernel(GridPairJob j) {
  const int m = (int)blockIdx.x < j.blocks0 ? 0 : 1;
  const int block = m ? (int)blockIdx.x - j.blocks0 : (int)blockIdx.x;
  grid_count_body(j.pts[m], j.n[m], j.bbox[m], j.radius, j.cap[m], j.gdesc[m], j.cell_of + (m ? j.n[0] : 0), j.count + (m ? j.cap[0] + 1 : 0), block);
}
__global__ void __launch_bounds__(256) grid_scatter_pair_kernel(GridPairJob j) {
  const int t0 = j.cap[0] + 1, t_all = t0 + j.cap[1] + 1;
  const int base1 = j.scanned[t0];                         // = the first map's indexed points (its table's last entries count nothing)
  // (a) each map's own cell table, by every thread of the launch in turn
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < t_all; t += gridDim.x * blockDim.x) {
    if (t < t0) j.cell_start[0][t] = j.scanned[t];
    else j.cell_start[1][t - t0] = j.scanned[t] - base1;
  }
  // (b) the points
  const int m = (int)blockIdx.x < j.blocks0 ? 0 : 1;
  const int block = m ? (int)blockIdx.x - j.blocks0 : (int)blockIdx.x;
  const int i = block * blockDim.x + threadIdx.x;
  const int* scanned = j.scanned + (m ? t0 : 0);
  const int sub = m ? base1 : 0;
  if (i == 0) {
    GridDesc* g = j.gdesc[m];
    g->n_pts = scanned[g->n_cells] - sub;
    *j.want_host[m] = g->want_cells;
    grid_bbox_rearm(j.bbox[m]);
  }
  if (i >= j.n[m]) return;
  const int c = (j.cell_of + (m ? j.n[0] : 0))[i];
  if (c < 0) { j.pos_of[m][i] = -1; return; }
  const int k = atomicSub(&(j.count + (m ? t0 : 0))[c], 1) - 1;
  float4 p = j.pts[m][i];
  p.w = __int_as_float(i);
  const int at = scanned[c] - sub + k;
  j.sorted[m][at] = p;
  j.pos_of[m][i] = at;
}

// ---------------------------------------------------------------------------------------------
// K4: association = transform + exact 5-NN + line / plane fit -> {C, N} record
// ---------------------------------------------------------------------------------------------

// Running 5 best as packed 64-bit keys (f32 distance bits << 32 | original map index): distances
// are >= 0 so the bit pattern is monotone, and one unsigned compare realises the total order
// (distance, index).  Branch-free sorted insertion: 5 compares + 10 selects on register pairs.
struct Top5 {
  unsigned long long k0, k1, k2, k3, k4;
};

// Every slot starts at (bound, 0xffffffff): a result is only ACCEPTED when its 5th distance is below the
// reference's gate (< 1.0, mapping_scan_matcher.cc:128,198), so candidates at or beyond the gate can be
// dropped at the pre-filter and rows / end cells farther than it pruned from the first row on.  A query
// with fewer than five neighbours inside the gate keeps the sentinel in k4 and is rejected.
__device__ __forceinline__ void top5_init(Top5& t, float bound) {
  t.k0 = t.k1 = t.k2 = t.k3 = t.k4 = ((unsigned long long)__float_as_uint(bound) << 32) | 0xffffffffull;
}
__device__ __forceinline__ float top5_d4(const Top5& t) { return __uint_as_float((unsigned int)(t.k4 >> 32)); }  // the gate while < 5 found
__device__ __forceinline__ float top_d4(const Top5& t) { return top5_d4(t); }

#ifndef MSFL_TOP5_SELECT
#define MSFL_TOP5_SELECT 0          /* 1: the compare / select insertion network of rounds 1-3 (A/B) */
#endif
// The inline assembly below (v_min_f64 / v_max_f64 / v_med3_u32 by their gfx950 mnemonics) is written for the one target this library
// has: a device pass for anything else stops here with a message instead of an assembler error deep in a template.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libmsfl_hip is written for gfx950 (MI355X) only: build with --offload-arch=gfx950"
#endif
// unsigned 64-bit min / max of two keys whose high words are < 2^31, on the f64 min / max unit (see top5_insert)
__device__ __forceinline__ unsigned long long u64_min_f(unsigned long long a, unsigned long long b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(__longlong_as_double((long long)a)), "v"(__longlong_as_double((long long)b)));
  return (unsigned long long)__double_as_longlong(r);
}
__device__ __forceinline__ unsigned long long u64_max_f(unsigned long long a, unsigned long long b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(__longlong_as_double((long long)a)), "v"(__longlong_as_double((long long)b)));
  return (unsigned long long)__double_as_longlong(r);
}
__device__ __forceinline__ void top5_insert(Top5& t, float d, int idx) {
  // cheap pre-filter on the distance word alone (a full-rate 32-bit compare; the bit pattern of a
  // non-negative float is monotone), then the exact 64-bit (distance, index) order
  if (__float_as_uint(d) > (unsigned int)(t.k4 >> 32)) return;
  const unsigned long long x = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned int)idx;
  if (!(x < t.k4)) return;
#if MSFL_TOP5_SELECT
  const bool c3 = x < t.k3, c2 = x < t.k2, c1 = x < t.k1, c0 = x < t.k0;
  t.k4 = c3 ? t.k3 : x;
  t.k3 = c3 ? (c2 ? t.k2 : x) : t.k3;
  t.k2 = c2 ? (c1 ? t.k1 : x) : t.k2;
  t.k1 = c1 ? (c0 ? t.k0 : x) : t.k1;
  t.k0 = c0 ? x : t.k0;
#else
  // Sorted insertion as a min / max chain on the keys READ AS DOUBLES.  A key's high word is the bit pattern of a non-negative f32
  // (< 2^31), so as an f64 it is a non-negative finite number or denormal (exponent field = the top 11 bits of the f32 pattern:
  // <= 0x7fc even for an f32 NaN), and for non-negative doubles the IEEE order IS the unsigned order of the bit patterns; f64
  // denormals are preserved (the default f64 mode), min / max select one operand bit for bit.  8 instructions instead of
  // 5 x v_cmp_lt_u64 + 16 x v_cndmask_b32 (all of the 4-clock class): the insertion pass runs for the whole wavefront whenever one
  // lane inserts and was ~30 % of the kernel.
  const unsigned long long c0 = u64_max_f(t.k0, x); t.k0 = u64_min_f(t.k0, x);
  const unsigned long long c1 = u64_max_f(t.k1, c0); t.k1 = u64_min_f(t.k1, c0);
  const unsigned long long c2 = u64_max_f(t.k2, c1); t.k2 = u64_min_f(t.k2, c1);
  const unsigned long long c3 = u64_max_f(t.k3, c2); t.k3 = u64_min_f(t.k3, c2);
  t.k4 = c3;                    // = min(k4, c3): x < k4 (the guard above) and k3 <= k4, so the largest of {k0..k3, x} is below k4
#endif
}

// Round 4, the walk's own top list: SIX 32-bit keys (distance bits & ~7) | slot, sorted, and the candidates' positions in the
// sorted map kept per (slot, lane) in LDS.  A candidate that passes the pre-filter takes the slot of the key it can only evict
// (the 6th), stores its position there and goes through 1 x v_min_u32 + 5 x v_med3_u32 — ~45 clocks per insertion pass of the
// wavefront instead of ~78 for the 64-bit min / max chain above (the pass runs for all 64 lanes whenever one lane inserts:
// ~47 passes per wavefront).  The keys order candidates by their distance truncated to 29 bits (t = bits >> 3); the result is
// the exact top-5 by (distance, index) whenever the six final keys have six different t:
//  * a candidate dropped at the pre-filter had t > t(5th at that time) >= t(final 5th);
//  * every other candidate was offered to the list, which keeps the six smallest keys offered, so a candidate outside the final
//    top-5 with t == t(final 5th) implies t(final 6th) == t(final 5th);
//  * with different t the truncated order IS the order of the exact distances, and the index word is never consulted.
// A lane whose final keys show an equal-t neighbour pair (or whose 5th shares its t with the acceptance gate) is AMBIGUOUS and
// is searched again with the exact 64-bit keys (Top5): two distances within 2^-21 relative of each other, ~1e-6 of the queries
// on a scanned surface, every query on an exact lattice.  The 6th key's stored position may be stale (a candidate that tied
// with it overwrote the slot without entering); the 6th is never output and can only leave the list.
#ifndef MSFL_KNN_KEY32
#define MSFL_KNN_KEY32 1            /* 0: the 64-bit (distance, index) keys of rounds 1-4a in every kernel (A/B) */
#endif
constexpr unsigned int kTopSentinel = 0xffffff00u;      // sentinels 0xffffff00 + 9 i: six different t, slots 0..5, above every f32 distance pattern
// k = med3(below, k, x) IN PLACE (below <= k): the sorted list's element after x has been inserted somewhere
__device__ __forceinline__ void u32_med3_into(unsigned int& k, unsigned int below, unsigned int x) {
  asm("v_med3_u32 %0, %1, %0, %2" : "+v"(k) : "v"(below), "v"(x));
}
struct Top6K {
  unsigned int k0, k1, k2, k3, k4, k5;
  unsigned int bound;               // min(k4 | 7, gate bits): the pre-filter, a conservative 5th distance for the row / cell bounds
  unsigned int gate;                // acceptance gate bits (or the seeded initial bound)
  int* col;                         // this lane's column of the slot table: col[slot * kTopSlotStride]
};
constexpr int kTopSlots = 6;
__device__ __forceinline__ void top_init(Top6K& t, float bound, int* col) {
  t.k0 = kTopSentinel; t.k1 = kTopSentinel + 9; t.k2 = kTopSentinel + 18; t.k3 = kTopSentinel + 27; t.k4 = kTopSentinel + 36; t.k5 = kTopSentinel + 45;
  t.gate = __float_as_uint(bound); t.bound = t.gate; t.col = col;
}
__device__ __forceinline__ float top_d4(const Top6K& t) { return __uint_as_float(t.bound); }
// true: the five nearest and their order are decided by the truncated keys (see above); false: search again with Top5
__device__ __forceinline__ bool top_settled(const Top6K& t, float accept_gate) {
  const unsigned int amb = min(min(min(t.k0 ^ t.k1, t.k1 ^ t.k2), min(t.k2 ^ t.k3, t.k3 ^ t.k4)), min(t.k4 ^ t.k5, t.k4 ^ __float_as_uint(accept_gate)));
  return amb >= 8u;
}
__device__ __forceinline__ bool top_found(const Top6K& t) { return t.k4 < kTopSentinel; }     // five real candidates within the gate
template <int STRIDE>
__device__ __forceinline__ int top_pos(const Top6K& t, unsigned int k) { return (int)((unsigned int)t.col[(k & 7u) * STRIDE] >> 4); }   // stored: byte offset

template <int STRIDE>
__device__ __forceinline__ void top_insert(Top5& t, float d, int idx, int) { top5_insert(t, d, idx); }

// flann::L2_Simple<float>: ((dx*dx) + dy*dy) + dz*dz, every operation rounded to f32
__device__ __forceinline__ float l2_simple(float4 a, float3 q) {
  const float dx = a.x - q.x, dy = a.y - q.y, dz = a.z - q.z;
  float r = dx * dx;
  r = r + dy * dy;
  r = r + dz * dz;
  return r;
}
// the same arithmetic with x and y in one packed-f32 lane pair (v_pk_add_f32 / v_pk_mul_f32: two
// IEEE operations per instruction, individually rounded, so the result is bit-identical)
typedef float msfl_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float l2_simple_pk(float4 a, msfl_f2 qxy, float qz) {
  const msfl_f2 axy = {a.x, a.y};
  const msfl_f2 dxy = axy - qxy;
  const msfl_f2 sxy = dxy * dxy;
  const float dz = a.z - qz;
  float r = sxy.x + sxy.y;
  r = r + dz * dz;
  return r;
}

// Exact 5-NN over the 27-cell neighbourhood.  Rows (y,z) are visited centre-first and a row / an
// end cell of a row is skipped when a LOWER BOUND of its distance to the query already exceeds the
// current 5th-best distance, so the result is identical to scanning all 27 cells.  The bound is
// shrunk by 1e-3 cell to stay conservative under the f32 rounding of cell coordinates.
__device__ __forceinline__ float axis_gap(float u, int c) {
  // distance (in cell units) from coordinate u to the interval [c, c+1], minus slack, floored at 0
  const float g = fmaxf((float)c - u, u - (float)(c + 1));
  return fmaxf(g - 1e-3f, 0.0f);
}

// max_sq_dist here is the INITIAL bound of the search: the acceptance gate, or (second outer iteration) something tighter
// that five real map points are known to meet (knn5_seed_bound).
// TOP: Top5 (exact 64-bit keys) or Top6K (32-bit keys + slot table with lane stride STRIDE ints, exact unless !top_settled); the caller
// initialises it with the search's initial bound.
template <class TOP, int STRIDE = 64>
__device__ __forceinline__ void knn5_grid(const GridDesc& g, const float4* __restrict__ sorted,
                                          const int* __restrict__ cell_start, float3 q, TOP& t,
                                          int& n_cand) {      // n_cand: candidates evaluated (dead code unless the caller reads it)
  const float ux = (q.x - g.ox) * g.inv_cell_x, uy = (q.y - g.oy) * g.inv_cell, uz = (q.z - g.oz) * g.inv_cell;
  const int cx = grid_coord(q.x, g.ox, g.inv_cell_x, g.dx);
  const int cy = grid_coord(q.y, g.oy, g.inv_cell, g.dy);
  const int cz = grid_coord(q.z, g.oz, g.inv_cell, g.dz);
  const int xs = max(cx - kGridXSub, 0), xe = min(cx + kGridXSub, g.dx - 1);
  if (xs > xe) return;
  const float cell2 = g.cell2;
  // per-axis lower bounds for the three y and three z cell offsets, computed once
  const float gy0 = axis_gap(uy, cy - 1), gy1 = axis_gap(uy, cy), gy2 = axis_gap(uy, cy + 1);
  const float gz0 = axis_gap(uz, cz - 1), gz1 = axis_gap(uz, cz), gz2 = axis_gap(uz, cz + 1);
  const float cellx2 = g.cellx2;
  // squared lower bounds of the kGridXSub cells left of / right of the query cell, outermost first: they
  // are the same for all nine rows, so the per-row trimming is two adds and two compares per side
  float gxa[kGridXSub], gxb[kGridXSub];
#pragma unroll
  for (int k = 0; k < kGridXSub; k++) {
    const float ga = axis_gap(ux, xs + k), gb = axis_gap(ux, xe - k);
    gxa[k] = ga * ga * cellx2; gxb[k] = gb * gb * cellx2;
  }
  const msfl_f2 qxy = {q.x, q.y};
  // Visit order of the 9 (dy, dz) rows, PER QUERY: centre, then the two side rows on the query's near sides (smaller
  // gap first), the two far side rows, the near-near diagonal, the two mixed diagonals, the far-far diagonal.  Any
  // order is exact (a row is only skipped on a lower bound); this one makes the 64 queries of a wavefront need the
  // same loop POSITIONS: with a fixed (-y, +y, -z, +z) order each position ran for the ~19 lanes whose near side it
  // happened to be and cost the longest of their ranges, now the first two positions carry nearly all of that work
  // and the later ones are skipped by the whole wavefront; the 5th-best distance also tightens sooner.
  const bool y_lo = gy0 <= gy2, z_lo = gz0 <= gz2;                 // near side: the smaller gap
  const int sy = y_lo ? -1 : 1, sz = z_lo ? -1 : 1;
  const float g_ny = y_lo ? gy0 : gy2, g_fy = y_lo ? gy2 : gy0;
  const float g_nz = z_lo ? gz0 : gz2, g_fz = z_lo ? gz2 : gz0;
  const bool ny_first = g_ny <= g_nz, fy_first = g_fy <= g_fz;
  const bool e_first = g_ny * g_ny + g_fz * g_fz <= g_fy * g_fy + g_nz * g_nz;   // (near y, far z) before (far y, near z)
  // candidates of the cells [a, b] of a row: x-adjacent cells are contiguous in the sorted array
  auto scan = [&](int row, int a, int b) __attribute__((always_inline)) {
    const int s = cell_start[row + a], e = cell_start[row + b + 1];
    n_cand += e - s;
    const float4* p = sorted + s;
    const float4* const pe = sorted + e;
    int pos = s;                            // position in the sorted map (Top6K's payload; dead code for Top5)
    for (; p + 1 < pe; p += 2, pos += 2) {  // two loads in flight, one address register
      float4 m0 = p[0], m1 = p[1];
      // keep the index word in the 16-byte load: left alone, the compiler splits it off into a second,
      // dependent load inside the (latency-critical) insertion path
      asm volatile("" : "+v"(m0.w));
      top_insert<STRIDE>(t, l2_simple_pk(m0, qxy, q.z), __float_as_int(m0.w), pos);
      asm volatile("" : "+v"(m1.w));
      top_insert<STRIDE>(t, l2_simple_pk(m1, qxy, q.z), __float_as_int(m1.w), pos + 1);
    }
    if (p < pe) { float4 m = p[0]; asm volatile("" : "+v"(m.w)); top_insert<STRIDE>(t, l2_simple_pk(m, qxy, q.z), __float_as_int(m.w), pos); }
  };
#pragma unroll
  for (int r = 0; r < 9; r++) {
    int dy, dz; float gy, gz;
    if (r == 0) { dy = 0; dz = 0; gy = gy1; gz = gz1; }
    else if (r == 1 || r == 2) {                                    // near side rows
      const bool yrow = (r == 1) == ny_first;
      dy = yrow ? sy : 0; dz = yrow ? 0 : sz; gy = yrow ? g_ny : gy1; gz = yrow ? gz1 : g_nz;
    } else if (r == 3 || r == 4) {                                  // far side rows
      const bool yrow = (r == 3) == fy_first;
      dy = yrow ? -sy : 0; dz = yrow ? 0 : -sz; gy = yrow ? g_fy : gy1; gz = yrow ? gz1 : g_fz;
    } else if (r == 5) { dy = sy; dz = sz; gy = g_ny; gz = g_nz; }
    else if (r == 6 || r == 7) {                                    // mixed diagonals
      const bool e = (r == 6) == e_first;                           // e: (near y, far z)
      dy = e ? sy : -sy; dz = e ? -sz : sz; gy = e ? g_ny : g_fy; gz = e ? g_fz : g_nz;
    } else { dy = -sy; dz = -sz; gy = g_fy; gz = g_fz; }
    const int y = cy + dy, z = cz + dz;
    if (y < 0 || y >= g.dy || z < 0 || z >= g.dz) continue;
    const float row2 = (gy * gy + gz * gz) * cell2;
    const int row = (z * g.dy + y) * g.dx;
    const float d4 = top_d4(t);
    if (row2 > d4) continue;                  // d4 is the acceptance gate until five neighbours are known
    // trim the x range: drop end cells whose lower bound exceeds the 5th-best distance
    // a cell is dropped when its lower bound exceeds d4; the bounds shrink towards the query, so count the
    // leading run of dropped cells on each side
    int a = xs, b = xe;
    bool da = true, db = true;
#pragma unroll
    for (int k = 0; k < kGridXSub; k++) {
      da = da && (row2 + gxa[k] > d4); a += da ? 1 : 0;
      db = db && (row2 + gxb[k] > d4); b -= db ? 1 : 0;
    }
    if (a > b) continue;                      // only near the grid border: every remaining cell is out of reach
    scan(row, a, b);
  }
}

// The same walk for Top6K, written for the instruction count (the kernel is VALU-issue bound: every wave instruction of a
// visited row is paid 64 lanes wide).  Differences to knn5_grid, none of which changes the result:
//  * candidates are addressed by 32-bit BYTE offsets from the (scalar) map base (saddr loads, 32-bit loop control; a map is
//    < 2^28 points: msfl_set_map refuses more) and the slot table stores the byte offset (position = offset >> 4 at the end);
//  * a row's end cells are trimmed by COUNTING the cells whose bound exceeds (5th distance - row bound): the per-side bounds
//    are made monotone at set-up (a cell at or beyond the query's own cell gets 0), so the count is the leading run;
//  * row validity (y, z inside the grid) and the row bases come from six set-up compares / two offsets instead of four
//    compares and two 32-bit multiplies per row.
template <int STRIDE>
__device__ __forceinline__ void top_insert_off(Top6K& t, float d, unsigned int off) {
  const unsigned int db = __float_as_uint(d);
  if (db > t.bound) return;
  const unsigned int slot = t.k5 & 7u;
  const unsigned int x = (db & ~7u) | slot;
  t.col[slot * STRIDE] = (int)off;
  u32_med3_into(t.k5, t.k4, x); u32_med3_into(t.k4, t.k3, x); u32_med3_into(t.k3, t.k2, x);
  u32_med3_into(t.k2, t.k1, x); u32_med3_into(t.k1, t.k0, x);
  t.k0 = min(t.k0, x);
  t.bound = min(t.k4 | 7u, t.gate);
}
template <int STRIDE>
__device__ __forceinline__ void knn5_grid_k32(const GridDesc& g, const float4* __restrict__ sorted,
                                              const int* __restrict__ cell_start, float3 q, Top6K& t, int& n_cand,
                                              int cell_base = 0) {        // cell_base: this map's slice of a shared cell table (pairs)
  const float ux = (q.x - g.ox) * g.inv_cell_x, uy = (q.y - g.oy) * g.inv_cell, uz = (q.z - g.oz) * g.inv_cell;
  const int cx = grid_coord(q.x, g.ox, g.inv_cell_x, g.dx);
  const int cy = grid_coord(q.y, g.oy, g.inv_cell, g.dy);
  const int cz = grid_coord(q.z, g.oz, g.inv_cell, g.dz);
  const int xs = max(cx - kGridXSub, 0), xe = min(cx + kGridXSub, g.dx - 1);
  if (xs > xe) return;
  const float cell2 = g.cell2, cellx2 = g.cellx2;
  const float gy0 = axis_gap(uy, cy - 1), gy1 = axis_gap(uy, cy), gy2 = axis_gap(uy, cy + 1);
  const float gz0 = axis_gap(uz, cz - 1), gz1 = axis_gap(uz, cz), gz2 = axis_gap(uz, cz + 1);
  float gxa[kGridXSub], gxb[kGridXSub];       // outermost first, non-increasing
#pragma unroll
  for (int k = 0; k < kGridXSub; k++) {
    const float ga = axis_gap(ux, xs + k), gb = axis_gap(ux, xe - k);
    gxa[k] = xs + k < cx ? ga * ga * cellx2 : 0.0f;
    gxb[k] = xe - k > cx ? gb * gb * cellx2 : 0.0f;
  }
  const msfl_f2 qxy = {q.x, q.y};
  const bool y_lo = gy0 <= gy2, z_lo = gz0 <= gz2;                 // near side: the smaller gap
  const float g_ny = y_lo ? gy0 : gy2, g_fy = y_lo ? gy2 : gy0;
  const float g_nz = z_lo ? gz0 : gz2, g_fz = z_lo ? gz2 : gz0;
  const float q_ny = g_ny * g_ny, q_fy = g_fy * g_fy, q_nz = g_nz * g_nz, q_fz = g_fz * g_fz, q_y1 = gy1 * gy1, q_z1 = gz1 * gz1;
  const bool ny_first = g_ny <= g_nz, fy_first = g_fy <= g_fz;
  const bool e_first = q_ny + q_fz <= q_fy + q_nz;                 // (near y, far z) before (far y, near z)
  // validity of the three y and three z cell coordinates, near / centre / far
  const int sy = y_lo ? -1 : 1, sz = z_lo ? -1 : 1;
  const bool v_y1 = (unsigned int)cy < (unsigned int)g.dy, v_ny = (unsigned int)(cy + sy) < (unsigned int)g.dy, v_fy = (unsigned int)(cy - sy) < (unsigned int)g.dy;
  const bool v_z1 = (unsigned int)cz < (unsigned int)g.dz, v_nz = (unsigned int)(cz + sz) < (unsigned int)g.dz, v_fz = (unsigned int)(cz - sz) < (unsigned int)g.dz;
  const int base0 = (cz * g.dy + cy) * g.dx + cell_base;           // only used where the row is valid
  const int off_y = y_lo ? -g.dx : g.dx;                           // near-side steps
  const int zstep = g.dy * g.dx;
  const int off_z = z_lo ? -zstep : zstep;
  const char* const mapb = (const char*)sorted;
  auto visit = [&](bool valid, int row, float rowq) __attribute__((always_inline)) {
    const float row2 = rowq * cell2;
    const float d4 = top_d4(t);
    if (!valid || row2 > d4) return;           // d4: the acceptance gate until five neighbours are known, then >= the 5th distance
    const float room = d4 - row2;
    int a = xs, b = xe;
#pragma unroll
    for (int k = 0; k < kGridXSub; k++) { a += gxa[k] > room ? 1 : 0; b -= gxb[k] > room ? 1 : 0; }
    if (a > b) return;
    const unsigned int s = (unsigned int)cell_start[(unsigned int)(row + a)], e = (unsigned int)cell_start[(unsigned int)(row + b + 1)];
    if (s == e) return;
    n_cand += (int)(e - s);
    unsigned int off = s << 4;
    const unsigned int end = e << 4, last = end - 16u;      // last: offset of the range's last point (>= off)
    for (; off < last; off += 32u) {           // two loads in flight
      float4 m0 = *(const float4*)(mapb + off), m1 = *(const float4*)(mapb + off + 16u);
      asm volatile("" : "+v"(m0.w));           // keep the 16-byte loads (12-byte loads: slower in rounds 1-2, no change in round 4b)
      top_insert_off<STRIDE>(t, l2_simple_pk(m0, qxy, q.z), off);
      asm volatile("" : "+v"(m1.w));
      top_insert_off<STRIDE>(t, l2_simple_pk(m1, qxy, q.z), off + 16u);
    }
    if (off < end) { float4 m = *(const float4*)(mapb + off); asm volatile("" : "+v"(m.w)); top_insert_off<STRIDE>(t, l2_simple_pk(m, qxy, q.z), off); }
  };
  // the same per-query row order as knn5_grid: centre, near sides (smaller gap first), far sides, near-near, mixed diagonals, far-far
  visit(v_y1 && v_z1, base0, q_y1 + q_z1);
  {
    const bool va = v_ny && v_z1, vb = v_y1 && v_nz;               // A: near-y side row, B: near-z side row
    const float ra = q_ny + q_z1, rb = q_y1 + q_nz;
    visit(ny_first ? va : vb, base0 + (ny_first ? off_y : off_z), ny_first ? ra : rb);
    visit(ny_first ? vb : va, base0 + (ny_first ? off_z : off_y), ny_first ? rb : ra);
  }
  {
    const bool va = v_fy && v_z1, vb = v_y1 && v_fz;
    const float ra = q_fy + q_z1, rb = q_y1 + q_fz;
    visit(fy_first ? va : vb, base0 - (fy_first ? off_y : off_z), fy_first ? ra : rb);
    visit(fy_first ? vb : va, base0 - (fy_first ? off_z : off_y), fy_first ? rb : ra);
  }
  visit(v_ny && v_nz, base0 + off_y + off_z, q_ny + q_nz);
  {
    const bool va = v_ny && v_fz, vb = v_fy && v_nz;               // A: (near y, far z)
    const float ra = q_ny + q_fz, rb = q_fy + q_nz;
    const int oa = off_y - off_z, ob = off_z - off_y;
    visit(e_first ? va : vb, base0 + (e_first ? oa : ob), e_first ? ra : rb);
    visit(e_first ? vb : va, base0 + (e_first ? ob : oa), e_first ? rb : ra);
  }
  visit(v_fy && v_fz, base0 - off_y - off_z, q_fy + q_fz);
}

// Second outer iteration (mapping_scan_matcher.cc:75: the association loop runs kOptimalNum = 2 times on the same clouds): the five
// neighbours the first pass found for this feature (positions in the sorted map, still in `nn`) are five DISTINCT real map
// points, so the largest of their f32 distances to the re-transformed query is an upper bound of the new 5th-nearest
// distance.  Starting the top-5 from (min(gate, that distance), 0xffffffff) sentinels trims rows and x-cells from the first
// row on and the result is still the exact top-5 by (distance, index): a candidate AT the bound compares below the sentinel's
// index word, and at least five real candidates are within the bound.  A feature the first pass rejected keeps the gate.
__device__ __forceinline__ float knn5_seed_bound(const float4* __restrict__ sorted, const int* __restrict__ prev, float3 q, float gate) {
  const int p0 = prev[0];
  if (p0 < 0) return gate;
  const int p1 = prev[1], p2 = prev[2], p3 = prev[3], p4 = prev[4];
  const float4 m0 = sorted[p0], m1 = sorted[p1], m2 = sorted[p2], m3 = sorted[p3], m4 = sorted[p4];
  const msfl_f2 qxy = {q.x, q.y};
  float m = l2_simple_pk(m0, qxy, q.z);
  float d;
  d = l2_simple_pk(m1, qxy, q.z); m = d > m ? d : m;
  d = l2_simple_pk(m2, qxy, q.z); m = d > m ? d : m;
  d = l2_simple_pk(m3, qxy, q.z); m = d > m ? d : m;
  d = l2_simple_pk(m4, qxy, q.z); m = d > m ? d : m;
  return m < gate ? m : gate;         // a NaN distance (non-finite query) leaves the gate in place
}

struct FitOut { d3 C, N; bool ok; };

// mapping_scan_matcher.cc:130-151
__device__ __forceinline__ FitOut edge_fit(const float4 (&nb)[5], double ratio) {
  FitOut o; o.ok = false; o.C = mk3(0, 0, 0); o.N = mk3(0, 0, 0);
  d3 m[5];
  d3 c = mk3(0, 0, 0);
#pragma unroll
  for (int j = 0; j < 5; j++) { m[j] = mk3((double)nb[j].x, (double)nb[j].y, (double)nb[j].z); c = c + m[j]; }
  c = mk3(c.x / 5.0, c.y / 5.0, c.z / 5.0);
  sym3 S = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < 5; j++) {
    const d3 d = m[j] - c;
    S.a00 += d.x * d.x; S.a01 += d.x * d.y; S.a02 += d.x * d.z;
    S.a11 += d.y * d.y; S.a12 += d.y * d.z; S.a22 += d.z * d.z;
  }
  double e1, e2; d3 dir;
  sym_eigen3_top(S, e1, e2, dir);
  if (!(e2 > ratio * e1)) return o;
  const d3 pa = mk3(0.1 * dir.x + c.x, 0.1 * dir.y + c.y, 0.1 * dir.z + c.z);
  const d3 pb = mk3(-0.1 * dir.x + c.x, -0.1 * dir.y + c.y, -0.1 * dir.z + c.z);
  o.N = normalized(pa - pb);
  o.C = pa;
  o.ok = true;
  return o;
}

// mapping_scan_matcher.cc:199-222
#ifndef MSFL_PLANE_ADJ
#define MSFL_PLANE_ADJ 1            /* 0: the unpivoted-QR fast path of round 3 instead of the centred adjugate form (A/B) */
#endif
// DEFER (round 5, the whole-batch fit kernel): an ill-conditioned neighbourhood does not run the pivoted QR here but is reported through
// `deferred` and fitted by fit_fallback_kernel; with the fallback out of this binary the hot path needs fewer registers (more
// wavefronts per SIMD).  The result is the same function of the five points either way.
template <bool DEFER = false>
__device__ __forceinline__ FitOut plane_fit(const float4 (&nb)[5], double tol, bool* deferred = nullptr) {
  FitOut o; o.ok = false; o.C = mk3(0, 0, 0); o.N = mk3(0, 0, 0);
  d3 c = mk3(0, 0, 0);
#pragma unroll
  for (int j = 0; j < 5; j++) c = c + mk3((double)nb[j].x, (double)nb[j].y, (double)nb[j].z);
  c = mk3(c.x / 5.0, c.y / 5.0, c.z / 5.0);
  bool well = false;
  d3 x = mk3(0, 0, 0);
#if MSFL_PLANE_ADJ
  if (!MSFL_IEEE_DIV) {                            // direction of the least-squares solution only: all the fit uses
    double q[5][3];                                // centred points (not kept across the fallback: the distance test re-forms them)
#pragma unroll
    for (int j = 0; j < 5; j++) { q[j][0] = (double)nb[j].x - c.x; q[j][1] = (double)nb[j].y - c.y; q[j][2] = (double)nb[j].z - c.z; }
    x = plane_normal_centred(q, c, well);
  }
#else
  {
    double A[5][3], b[5];
#pragma unroll
    for (int j = 0; j < 5; j++) { A[j][0] = (double)nb[j].x; A[j][1] = (double)nb[j].y; A[j][2] = (double)nb[j].z; b[j] = -1.0; }
    if (!MSFL_IEEE_DIV) x = lstsq5x3_fast(A, b, well);             // overwrites A, b
  }
#endif
  if (DEFER && !well) { *deferred = true; return o; }
  if (!DEFER && !well) {                           // ill-conditioned or rank-deficient: the reference's pivoted QR decides
    double A[5][3], b[5];
#pragma unroll
    for (int j = 0; j < 5; j++) { A[j][0] = (double)nb[j].x; A[j][1] = (double)nb[j].y; A[j][2] = (double)nb[j].z; b[j] = -1.0; }
    x = lstsq5x3(A, b);
  }
  const d3 n = normalized_rsq(x);
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 5; j++) {
    const double d = n.x * ((double)nb[j].x - c.x) + n.y * ((double)nb[j].y - c.y) + n.z * ((double)nb[j].z - c.z);
    if (fabs(d) > tol) ok = false;
  }
  if (!ok) return o;
  o.C = c; o.N = n; o.ok = true;
  return o;
}

struct BatchView {
  const float4* corner; const int* corner_off;
  const float4* surf;   const int* surf_off;
  const int* rec_off;       // rec_off[b] = corner_off[b] + surf_off[b]
  int n_scans;
  int n_records;            // rec_off[n_scans]; for the association kernels: one past the last record of this launch
  int rec_begin = 0;        // association kernels: first record of this launch (a host-buffer batch is associated chunk by chunk as it arrives)
  int c0, s0;               // corner_off[0], surf_off[0] (host copies)
  int n_surf_total;         // surf_off[n_scans] - surf_off[0]
  int dyn = 0;              // 1: the offsets were written on the device (per-scan SLAM step); n_records is then an upper bound and the
                            // record count is rec_off[n_scans]; c0 = s0 = 0 and n_surf_total is the surf cloud's CAPACITY (a layout constant)
};
__device__ __forceinline__ int batch_records(const BatchView& bv) {
  return bv.dyn ? min(bv.n_records, bv.rec_off[bv.n_scans]) : bv.n_records;
}

// scan owning global record index g (upper bound - 1 over rec_off)
__device__ __forceinline__ int find_scan(const int* __restrict__ rec_off, int n_scans, int g);
// the same for a whole wavefront of consecutive g: the binary search runs once on the scalar unit for
// the first lane's g, every lane then steps forward over the (rare) scan boundaries inside the wave
__device__ __forceinline__ int find_scan_wave(const int* __restrict__ rec_off, int n_scans, int g) {
  const int g0 = __builtin_amdgcn_readfirstlane(g);
  int b = __builtin_amdgcn_readfirstlane(find_scan(rec_off, n_scans, g0));
  while (b + 1 < n_scans && g >= rec_off[b + 1]) b++;
  return b;
}
__device__ __forceinline__ int find_scan(const int* __restrict__ rec_off, int n_scans, int g) {
  int lo = 0, hi = n_scans;      // invariant: rec_off[lo] <= g < rec_off[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (rec_off[mid] <= g) lo = mid; else hi = mid;
  }
  return lo;
}

// Neighbour lists of the WHOLE-BATCH kernels (knn5_scan2map_split_kernel -> fit_scan2map_split_kernel; round 6): a feature's five neighbours
// sit at slot (f - c0) for corner feature f and (all corner features) + (f - s0) for surf feature f -- a function of the feature's index in
// its cloud array alone, like the record offsets below.  The per-record kernels number `nn` by record (scan by scan, corners then
// surfs), which costs a wavefront the scan search and the offset loads before its first useful load; the fit kernel is a short
// latency chain at four wavefronts per SIMD and that prologue was a tenth of it.  Producer and consumer of a batch always use the same numbering.
__device__ __forceinline__ size_t feature_slot(const BatchView& bv, bool edge, int f) {
  return edge ? (size_t)(f - bv.c0) : (size_t)(bv.n_records - bv.n_surf_total) + (size_t)(f - bv.s0);
}

// Record buffer: the PLANE records of the whole batch first (4 doubles each, so every record is one aligned 32-byte
// line), then the edge records (6 doubles each); both are indexed by the feature's index in its cloud, so a record can be
// written from any processing order.  Offsets in doubles:
__device__ __forceinline__ size_t plane_rec_off(const BatchView& bv, int fi_surf) { return 4 * (size_t)(fi_surf - bv.s0); }
__device__ __forceinline__ size_t edge_rec_off(const BatchView& bv, int fi_corner) {
  return 4 * (size_t)bv.n_surf_total + 6 * (size_t)(fi_corner - bv.c0);
}

struct DeskewView {
  // all null for the plain (LiDAR-only) branch
  const double* corner_dq; const double* corner_dp;
  const double* surf_dq;   const double* surf_dp;
  const double* V;          // n_scans x 3 (device): Vi of every scan
  double G[3];
  const double* G_dev;      // gravity in device memory (the SLAM step's per-scan IMU block); null: G above
  double* pprime;           // out: n_records x 3, p' = dq*p + dp
};

// K4a: transform + exact 5-NN.  Low register count (no f64 fits here) -> 8 waves/SIMD to hide the
// latency of the scattered 16-byte candidate loads.  nn[5*g..] = positions in the sorted map (nearest first),
// nn[5*g] = -1 when the feature is rejected by the `pointSearchSqDis[4] < 1.0` gate (:128 / :198).
#ifndef MSFL_ASSOC_BLOCK
#define MSFL_ASSOC_BLOCK 64
#endif
constexpr int kAssocBlock = MSFL_ASSOC_BLOCK;      // threads per workgroup of the 5-NN and fit kernels
// PAIRS: scan b is registered against ITS OWN map (msfl_pairs.cuh): descriptor gcp[b] / gsp[b], cell table slice at cbase_*[b]
template <bool DESKEW, bool COUNT = false, bool SEED = false, bool PAIRS = false>
__global__ void __launch_bounds__(kAssocBlock)
knn5_scan2map_kernel(BatchView bv, const double* __restrict__ poses, const int* __restrict__ status,
                     const GridDesc* __restrict__ gcp, const float4* __restrict__ map_c, const int* __restrict__ cs_c,
                     const GridDesc* __restrict__ gsp, const float4* __restrict__ map_s, const int* __restrict__ cs_s,
                     const int* __restrict__ pos_c, const int* __restrict__ pos_s,
                     float max_sq_dist, DeskewView dv, int* __restrict__ nn, unsigned long long* __restrict__ n_candidates = nullptr,
                     const int* __restrict__ cbase_c = nullptr, const int* __restrict__ cbase_s = nullptr) {
  const int g = bv.rec_begin + blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= batch_records(bv)) return;
  const int b = find_scan_wave(bv.rec_off, bv.n_scans, g);
  int* out = nn + 5 * (size_t)g;
  if (status[b] != 0) { out[0] = -1; out[1] = -1; out[2] = -1; out[3] = -1; out[4] = -1; return; }
  const int local = g - bv.rec_off[b];
  const int nc = bv.corner_off[b + 1] - bv.corner_off[b];
  const bool is_edge = local < nc;
  const int fi = is_edge ? bv.corner_off[b] + local : bv.surf_off[b] + (local - nc);
  const float4 f = is_edge ? bv.corner[fi] : bv.surf[fi];
  const pose7 T = load_pose(poses + 7 * b);
  float3 q;
  if (DESKEW) {
    // mapping_scan_matcher.cc:120 / :190: pose * Rigid3d{q^-1 (Vi dt - G dt^2/2) + dp, dq}
    const double* dqp = is_edge ? dv.corner_dq + 4 * (size_t)fi : dv.surf_dq + 4 * (size_t)fi;
    const double* dpp = is_edge ? dv.corner_dp + 3 * (size_t)fi : dv.surf_dp + 3 * (size_t)fi;
    quat dq; dq.x = dqp[0]; dq.y = dqp[1]; dq.z = dqp[2]; dq.w = dqp[3];
    const d3 dp = mk3(dpp[0], dpp[1], dpp[2]);
    const double dt = (double)f.w;
    const double* Vb = dv.V + 3 * (size_t)b;
    const double Gv[3] = {dv.G_dev ? dv.G_dev[0] : dv.G[0], dv.G_dev ? dv.G_dev[1] : dv.G[1], dv.G_dev ? dv.G_dev[2] : dv.G[2]};
    const d3 shift = mk3(Vb[0] * dt - 0.5 * Gv[0] * dt * dt, Vb[1] * dt - 0.5 * Gv[1] * dt * dt,
                         Vb[2] * dt - 0.5 * Gv[2] * dt * dt);
    quat qc; qc.x = -T.q.x; qc.y = -T.q.y; qc.z = -T.q.z; qc.w = T.q.w;
    pose7 full;
    full.t = quat_rotate(T.q, quat_rotate(qc, shift) + dp) + T.t;       // Rigid3d operator*
    full.q = quat_normalized(quat_mul(T.q, dq));
    q = transform_point_f32(full, f.x, f.y, f.z);
    const d3 pp = quat_rotate(dq, mk3((double)f.x, (double)f.y, (double)f.z)) + dp;
    dv.pprime[3 * (size_t)g + 0] = pp.x; dv.pprime[3 * (size_t)g + 1] = pp.y; dv.pprime[3 * (size_t)g + 2] = pp.z;
  } else {
    q = transform_point_f32(T, f.x, f.y, f.z);                          // :123 / :193
  }
  Top5 t;
  int n_cand = 0;
  const float bound = SEED ? knn5_seed_bound(is_edge ? map_c : map_s, out, q, max_sq_dist) : max_sq_dist;
#if MSFL_KNN_KEY32
  // the walk on truncated 32-bit keys first (Top6K); a lane it leaves ambiguous is searched again with the exact keys below
  __shared__ int s_slot[kTopSlots * kAssocBlock];
  Top6K tk;
  top_init(tk, bound, s_slot + threadIdx.x);
  if (is_edge) { const GridDesc gc = gcp[PAIRS ? b : 0]; knn5_grid_k32<kAssocBlock>(gc, map_c, cs_c, q, tk, n_cand, PAIRS ? cbase_c[b] : 0); }
  else { const GridDesc gs = gsp[PAIRS ? b : 0]; knn5_grid_k32<kAssocBlock>(gs, map_s, cs_s, q, tk, n_cand, PAIRS ? cbase_s[b] : 0); }
  const bool settled = top_settled(tk, max_sq_dist);
  if (!settled) {
    int n_again = 0;                   // the candidate counter reports the first walk
    top5_init(t, bound);
    if (is_edge) { const GridDesc gc = gcp[PAIRS ? b : 0]; knn5_grid(gc, map_c, cs_c + (PAIRS ? cbase_c[b] : 0), q, t, n_again); }
    else { const GridDesc gs = gsp[PAIRS ? b : 0]; knn5_grid(gs, map_s, cs_s + (PAIRS ? cbase_s[b] : 0), q, t, n_again); }
  }
#else
  const bool settled = false;
  top5_init(t, bound);
  if (is_edge) { const GridDesc gc = gcp[PAIRS ? b : 0]; knn5_grid(gc, map_c, cs_c + (PAIRS ? cbase_c[b] : 0), q, t, n_cand); }
  else { const GridDesc gs = gsp[PAIRS ? b : 0]; knn5_grid(gs, map_s, cs_s + (PAIRS ? cbase_s[b] : 0), q, t, n_cand); }
#endif
  if (COUNT) {                                                  // one atomic per wavefront: sum over the lanes still here
    const unsigned long long act = __ballot(1);
    unsigned long long m = act;
    int tot = 0;
    while (m) { const int l = __ffsll((long long)m) - 1; tot += __shfl(n_cand, l); m &= m - 1; }
    if ((int)(threadIdx.x & 63) == __ffsll((long long)act) - 1) atomicAdd(n_candidates, (unsigned long long)tot);
  }
#if MSFL_KNN_KEY32
  if (settled) {
    if (top_found(tk)) {                                                              // :128 / :198: the 5th is below the gate
      out[0] = top_pos<kAssocBlock>(tk, tk.k0); out[1] = top_pos<kAssocBlock>(tk, tk.k1); out[2] = top_pos<kAssocBlock>(tk, tk.k2);
      out[3] = top_pos<kAssocBlock>(tk, tk.k3); out[4] = top_pos<kAssocBlock>(tk, tk.k4);   // nearest first, positions in the sorted map
    } else {
      out[0] = -1; out[1] = -1; out[2] = -1; out[3] = -1; out[4] = -1;
    }
    return;
  }
#endif
  if ((unsigned int)t.k4 != 0xffffffffu && (double)top5_d4(t) < (double)max_sq_dist) {      // :128 / :198
    // original map index -> position in the sorted map array (the fit kernel then gathers directly);
    // done here because this kernel runs at 8 waves/SIMD and hides the extra dependent load
    const int* po = is_edge ? pos_c : pos_s;
    out[0] = po[(unsigned int)t.k0]; out[1] = po[(unsigned int)t.k1]; out[2] = po[(unsigned int)t.k2];
    out[3] = po[(unsigned int)t.k3]; out[4] = po[(unsigned int)t.k4];       // nearest first
  } else {
    out[0] = -1; out[1] = -1; out[2] = -1; out[3] = -1; out[4] = -1;   // all five: the fit kernel loads them unconditionally
  }
}

// a load the caller knows to be wave-uniform and of memory no thread of this kernel writes: emitted as a scalar load
template <class V>
__device__ __forceinline__ V uniform_load(const V* p) {
  return *(const __attribute__((address_space(4))) V*)(unsigned long long)p;
}

// K4a for a whole large batch of the plain branch: blocks [0, edge_blocks) serve the corner features (corner map index), the rest
// the surf features, so that a wavefront never holds both kinds and each body knows its map at compile time.
template <bool EDGE, bool SEED>
__device__ __forceinline__ void knn5_one_kind(const BatchView& bv, const double* __restrict__ poses, const int* __restrict__ status,
                                              const GridDesc* __restrict__ gp, const float4* __restrict__ map, const int* __restrict__ cs,
                                              const int* __restrict__ po, float max_sq_dist, int* __restrict__ nn, int block, int* s_slot) {
  const int* off = EDGE ? bv.corner_off : bv.surf_off;
  const int f_i = off[0] + block * (int)blockDim.x + (int)threadIdx.x;
  if (f_i >= off[bv.n_scans]) return;
  // Nearly every wavefront lies inside ONE scan (two boundaries per scan in ~78 wavefronts): its scan number, offsets, status and
  // pose are then read through the scalar unit — the kernel is bound by vector-memory instruction issue as much as by VALU issue
  // (texture addresser ~80 % busy, profiles/r04b_knn_ta.md), and these were ten of its ~92 vector loads per wavefront.
  const int b0 = __builtin_amdgcn_readfirstlane(find_scan(off, bv.n_scans, __builtin_amdgcn_readfirstlane(f_i)));
  int b, st; pose7 T;
  if (__all(f_i < off[b0 + 1])) {
    // loads through the constant address space (written by earlier kernels only), so that they stay scalar loads: with plain loads
    // the optimiser merges the two branches into one set of per-lane loads again
    b = b0;
    st = uniform_load(status + b0);
    const double* pp = poses + 7 * b0;
    T.t = mk3(uniform_load(pp), uniform_load(pp + 1), uniform_load(pp + 2));
    T.q.x = uniform_load(pp + 3); T.q.y = uniform_load(pp + 4); T.q.z = uniform_load(pp + 5); T.q.w = uniform_load(pp + 6);
  } else {
    b = b0;
    while (b + 1 < bv.n_scans && f_i >= off[b + 1]) b++;
    st = status[b];
    T = load_pose(poses + 7 * b);
  }
  int* out = nn + 5 * feature_slot(bv, EDGE, f_i);
  if (st != 0) { out[0] = -1; out[1] = -1; out[2] = -1; out[3] = -1; out[4] = -1; return; }
  const float4 f = EDGE ? bv.corner[f_i] : bv.surf[f_i];
  const float3 q = transform_point_f32(T, f.x, f.y, f.z);                          // :123 / :193
  int n_cand = 0;
  const GridDesc gd = *gp;
  const float bound = SEED ? knn5_seed_bound(map, out, q, max_sq_dist) : max_sq_dist;
#if MSFL_KNN_KEY32
  Top6K tk;
  top_init(tk, bound, s_slot + threadIdx.x);
  knn5_grid_k32<kAssocBlock>(gd, map, cs, q, tk, n_cand);
  if (top_settled(tk, max_sq_dist)) {
    if (top_found(tk)) {                                                              // :128 / :198: the 5th is below the gate (t differs from the gate's)
      out[0] = top_pos<kAssocBlock>(tk, tk.k0); out[1] = top_pos<kAssocBlock>(tk, tk.k1); out[2] = top_pos<kAssocBlock>(tk, tk.k2);
      out[3] = top_pos<kAssocBlock>(tk, tk.k3); out[4] = top_pos<kAssocBlock>(tk, tk.k4);   // nearest first, positions in the sorted map
    } else {
      out[0] = -1; out[1] = -1; out[2] = -1; out[3] = -1; out[4] = -1;
    }
    return;
  }
  // ambiguous under the truncated keys: the exact (distance, index) search
#endif
  Top5 t;
  top5_init(t, bound);
  knn5_grid(gd, map, cs, q, t, n_cand);
  if ((unsigned int)t.k4 != 0xffffffffu && (double)top5_d4(t) < (double)max_sq_dist) {      // :128 / :198
    out[0] = po[(unsigned int)t.k0]; out[1] = po[(unsigned int)t.k1]; out[2] = po[(unsigned int)t.k2];
    out[3] = po[(unsigned int)t.k3]; out[4] = po[(unsigned int)t.k4];       // nearest first
  } else {
    out[0] = -1; out[1] = -1; out[2] = -1; out[3] = -1; out[4] = -1;
  }
}
template <bool SEED>
__global__ void __launch_bounds__(kAssocBlock)
knn5_scan2map_split_kernel(BatchView bv, const double* __restrict__ poses, const int* __restrict__ status,
                           const GridDesc* __restrict__ gcp, const float4* __restrict__ map_c, const int* __restrict__ cs_c,
                           const GridDesc* __restrict__ gsp, const float4* __restrict__ map_s, const int* __restrict__ cs_s,
                           const int* __restrict__ pos_c, const int* __restrict__ pos_s, float max_sq_dist, int* __restrict__ nn, int edge_blocks) {
  __shared__ int s_slot[MSFL_KNN_KEY32 ? kTopSlots * kAssocBlock : 1];        // Top6K's slot table: 6 positions per lane
  if ((int)blockIdx.x < edge_blocks) knn5_one_kind<true, SEED>(bv, poses, status, gcp, map_c, cs_c, pos_c, max_sq_dist, nn, (int)blockIdx.x, s_slot);
  else knn5_one_kind<false, SEED>(bv, poses, status, gsp, map_s, cs_s, pos_s, max_sq_dist, nn, (int)blockIdx.x - edge_blocks, s_slot);
}

// K4a, latency form: the same exact 5-NN for a launch too small to fill the machine (one scan per call: ~5 000 queries are
// 80 wavefronts on 1 024 SIMDs, and a query is a chain of up to nine dependent row-bound -> candidate load round trips,
// ~35 us).  Sixteen lanes serve one query: lane r < 9 owns row r of the 3 x 3 (y, z) neighbourhood, reads its two row
// bounds and scans its candidates into a private top-5 against the acceptance gate only (rows are not pruned against
// each other: that would serialise them again), then the nine sorted lists are merged by five rounds of "smallest head
// of the group".  Keys are (distance bits, map index), so the five survivors and their order are exactly those of
// knn5_grid: an exact top-5 does not depend on the visit order.  Lanes 0..4 translate and store one neighbour each.
constexpr int kKnnRowLanes = 16;
constexpr int kKnnRowsBlock = 256;
#ifndef MSFL_KNN_ROWS_MAX
#define MSFL_KNN_ROWS_MAX 32768
#endif
constexpr int kKnnRowsMaxRecords = MSFL_KNN_ROWS_MAX;   // launches of up to this many queries take the latency form (measured crossover: DESIGN.md)
__global__ void __launch_bounds__(kKnnRowsBlock)
knn5_scan2map_rows_kernel(BatchView bv, const double* __restrict__ poses, const int* __restrict__ status,
                          const GridDesc* __restrict__ gcp, const float4* __restrict__ map_c, const int* __restrict__ cs_c,
                          const GridDesc* __restrict__ gsp, const float4* __restrict__ map_s, const int* __restrict__ cs_s,
                          const int* __restrict__ pos_c, const int* __restrict__ pos_s, float max_sq_dist, int* __restrict__ nn) {
  const int sl = threadIdx.x & (kKnnRowLanes - 1);
  const int g = bv.rec_begin + (int)((blockIdx.x * kKnnRowsBlock + threadIdx.x) / kKnnRowLanes);
  if (g >= batch_records(bv)) return;                         // whole groups leave together
  const int b = find_scan(bv.rec_off, bv.n_scans, g);
  int* out = nn + 5 * (size_t)g;
  if (status[b] != 0) { if (sl < 5) out[sl] = -1; return; }
  const int local = g - bv.rec_off[b];
  const int nc = bv.corner_off[b + 1] - bv.corner_off[b];
  const bool is_edge = local < nc;
  const int fi = is_edge ? bv.corner_off[b] + local : bv.surf_off[b] + (local - nc);
  const float4 f = is_edge ? bv.corner[fi] : bv.surf[fi];
  const pose7 T = load_pose(poses + 7 * b);
  const float3 q = transform_point_f32(T, f.x, f.y, f.z);    // :123 / :193
  const GridDesc gd = is_edge ? *gcp : *gsp;
  const float4* __restrict__ sorted = is_edge ? map_c : map_s;
  const int* __restrict__ cell_start = is_edge ? cs_c : cs_s;
  Top5 t;
  top5_init(t, max_sq_dist);
  const unsigned long long sentinel = t.k0;
  const float ux = (q.x - gd.ox) * gd.inv_cell_x, uy = (q.y - gd.oy) * gd.inv_cell, uz = (q.z - gd.oz) * gd.inv_cell;
  const int cx = grid_coord(q.x, gd.ox, gd.inv_cell_x, gd.dx);
  const int cy = grid_coord(q.y, gd.oy, gd.inv_cell, gd.dy);
  const int cz = grid_coord(q.z, gd.oz, gd.inv_cell, gd.dz);
  const int xs = max(cx - kGridXSub, 0), xe = min(cx + kGridXSub, gd.dx - 1);
  const int y = cy + (sl % 3) - 1, z = cz + (sl / 3) - 1;
  if (sl < 9 && xs <= xe && y >= 0 && y < gd.dy && z >= 0 && z < gd.dz) {
    const float gy = axis_gap(uy, y), gz = axis_gap(uz, z);
    const float row2 = (gy * gy + gz * gz) * gd.cell2;
    if (!(row2 > max_sq_dist)) {
      // end cells whose lower bound is beyond the gate hold no acceptable candidate
      int a = xs, e_cell = xe;
      bool da = true, db = true;
#pragma unroll
      for (int k = 0; k < kGridXSub; k++) {
        const float ga = axis_gap(ux, xs + k), gb = axis_gap(ux, xe - k);
        da = da && (row2 + ga * ga * gd.cellx2 > max_sq_dist); a += da ? 1 : 0;
        db = db && (row2 + gb * gb * gd.cellx2 > max_sq_dist); e_cell -= db ? 1 : 0;
      }
      if (a <= e_cell) {
        const int row = (z * gd.dy + y) * gd.dx;
        const int s = cell_start[row + a], e = cell_start[row + e_cell + 1];
        const msfl_f2 qxy = {q.x, q.y};
        for (int i = s; i < e; i += 4) {                    // four loads requested together (clamped index)
          float4 m[4];
#pragma unroll
          for (int u = 0; u < 4; u++) { m[u] = sorted[min(i + u, e - 1)]; asm volatile("" : "+v"(m[u].w)); }
#pragma unroll
          for (int u = 0; u < 4; u++)
            if (i + u < e) top5_insert(t, l2_simple_pk(m[u], qxy, q.z), __float_as_int(m[u].w));
        }
      }
    }
  }
  // merge: five times the smallest head of the group's (sorted) lists; the owner pops.  Map indices are unique, so
  // real keys are too; equal heads are sentinels, and popping a sentinel list changes nothing.
  unsigned long long best[5];
#pragma unroll
  for (int j = 0; j < 5; j++) {
    unsigned long long m = t.k0;
#pragma unroll
    for (int o = kKnnRowLanes / 2; o > 0; o >>= 1) {
      const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)m, o, kKnnRowLanes), hi = (unsigned)__shfl_xor((int)(unsigned)(m >> 32), o, kKnnRowLanes);
      const unsigned long long other = ((unsigned long long)hi << 32) | lo;
      m = other < m ? other : m;
    }
    best[j] = m;
    if (t.k0 == m) { t.k0 = t.k1; t.k1 = t.k2; t.k2 = t.k3; t.k3 = t.k4; t.k4 = sentinel; }
  }
  if (sl >= 5) return;
  const bool accepted = (unsigned int)best[4] != 0xffffffffu && (double)__uint_as_float((unsigned int)(best[4] >> 32)) < (double)max_sq_dist;   // :128 / :198
  const unsigned long long mine = sl == 0 ? best[0] : sl == 1 ? best[1] : sl == 2 ? best[2] : sl == 3 ? best[3] : best[4];
  const int* po = is_edge ? pos_c : pos_s;
  out[sl] = accepted ? po[(unsigned int)mine] : -1;           // nearest first
}

// K4b: 5 neighbours -> line fit (3x3 Jacobi eigen) / plane fit (5x3 Householder QR) -> record.
// Edge: {C, N}.  Plane: {N, N.C} (the residual N.(Rp+t-C) only needs the offset N.C).
// `full` (optional, debug/parity API) receives {C, N} for every feature.
#ifndef MSFL_FIT_WAVES
#define MSFL_FIT_WAVES 1
#endif
// KIND 0: every record of [rec_begin, n_records) (one thread each, edges and planes as they come); KIND 1 / 2: the launch covers
// the batch's corner / surf features only (thread = feature index in its cloud), so that the compiler sees one of the two fits.
template <bool DESKEW, int KIND, bool DEFER = false>
__device__ __forceinline__ void fit_one(const BatchView& bv, const float4* __restrict__ map_c, const float4* __restrict__ map_s,
                                        const int* __restrict__ nn, double line_ratio, double plane_tol, const DeskewView& dv,
                                        double* __restrict__ rec, double* __restrict__ full, int block, int* __restrict__ fallback = nullptr) {
  int g = 0, b = 0, local = 0, nc = 0, fi = 0;
  size_t slot;
  if (KIND == 0) {
    g = bv.rec_begin + block * blockDim.x + threadIdx.x;
    if (g >= batch_records(bv)) return;
    b = find_scan_wave(bv.rec_off, bv.n_scans, g);
    local = g - bv.rec_off[b];
    nc = bv.corner_off[b + 1] - bv.corner_off[b];
    slot = (size_t)g;
  } else {
    // whole-batch form: neighbour slot and record offset are functions of the feature's index alone (feature_slot): no scan search,
    // the wavefront's first load is its neighbour list
    fi = (KIND == 1 ? bv.c0 : bv.s0) + block * blockDim.x + threadIdx.x;
    if (fi >= (KIND == 1 ? bv.c0 + (bv.n_records - bv.n_surf_total) : bv.s0 + bv.n_surf_total)) return;
    slot = feature_slot(bv, KIND == 1, fi);
  }
  const int* in = nn + 5 * slot;
  const bool is_edge = KIND == 0 ? local < nc : KIND == 1;
  FitOut fo; fo.ok = false; fo.C = mk3(0, 0, 0); fo.N = mk3(0, 0, 0);
  // all five indices at once and the five neighbours unconditionally (index 0 stands in when there is no match): behind the
  // `p0 >= 0` test the loads came as three dependent round trips (first index, the other four, the points)
  const int p0 = in[0], p1 = in[1], p2 = in[2], p3 = in[3], p4 = in[4];
  const float4* mp = is_edge ? map_c : map_s;
  const bool have = p0 >= 0;                                  // no match: the other four slots were never written
  const float4 nb[5] = {mp[have ? p0 : 0], mp[have ? p1 : 0], mp[have ? p2 : 0], mp[have ? p3 : 0], mp[have ? p4 : 0]};
  if (p0 >= 0) {
    bool deferred = false;
    fo = is_edge ? edge_fit(nb, line_ratio) : plane_fit<DEFER>(nb, plane_tol, &deferred);
    if (DEFER && deferred) {                                    // fallback[0]: count, then the surf features' indices; the record written
      const int at = atomicAdd(fallback, 1);                    // below (rejected) is overwritten by fit_fallback_kernel
      fallback[1 + at] = fi;
    }
    if (DESKEW && fo.ok) {
      // C' = C - (Vi dt - G dt^2/2): the velocity block is constant (mapping_scan_matcher.cc:94)
      static_assert(!DESKEW || KIND == 0, "the de-skew branch runs the per-record kernel");
      const int fd = is_edge ? bv.corner_off[b] + local : bv.surf_off[b] + (local - nc);
      const double dt = (double)(is_edge ? bv.corner[fd].w : bv.surf[fd].w);
      const double* Vb = dv.V + 3 * (size_t)b;
      const double Gv[3] = {dv.G_dev ? dv.G_dev[0] : dv.G[0], dv.G_dev ? dv.G_dev[1] : dv.G[1], dv.G_dev ? dv.G_dev[2] : dv.G[2]};
      fo.C = fo.C - mk3(Vb[0] * dt - 0.5 * Gv[0] * dt * dt, Vb[1] * dt - 0.5 * Gv[1] * dt * dt,
                        Vb[2] * dt - 0.5 * Gv[2] * dt * dt);
    }
  }
  if (is_edge) {
    double* out = rec + edge_rec_off(bv, KIND == 0 ? bv.corner_off[b] + local : fi);
    out[0] = fo.C.x; out[1] = fo.C.y; out[2] = fo.C.z;
    out[3] = fo.N.x; out[4] = fo.N.y; out[5] = fo.N.z;
  } else {
    double* out = rec + plane_rec_off(bv, KIND == 0 ? bv.surf_off[b] + (local - nc) : fi);
    out[0] = fo.N.x; out[1] = fo.N.y; out[2] = fo.N.z; out[3] = dot(fo.N, fo.C);
  }
  if (KIND == 0 && full) {                                      // (the whole-batch form is not launched with a `full` output)
    double* o = full + 6 * (size_t)g;
    o[0] = fo.C.x; o[1] = fo.C.y; o[2] = fo.C.z; o[3] = fo.N.x; o[4] = fo.N.y; o[5] = fo.N.z;
  }
}

template <bool DESKEW>
__global__ void __launch_bounds__(kAssocBlock, MSFL_FIT_WAVES)
fit_scan2map_kernel(BatchView bv, const float4* __restrict__ map_c, const float4* __restrict__ map_s,
                    const int* __restrict__ nn, double line_ratio, double plane_tol, DeskewView dv,
                    double* __restrict__ rec, double* __restrict__ full) {
  fit_one<DESKEW, 0>(bv, map_c, map_s, nn, line_ratio, plane_tol, dv, rec, full, (int)blockIdx.x);
}
// A whole large batch (plain branch): blocks [0, edge_blocks) fit the corner features, the rest the surf features, each
// through its own specialisation of the body, at four wavefronts per SIMD (the plane fit alone needs 131 registers, the
// mixed body 135: three wavefronts).  The edge blocks come first so that their longer per-record chain (the Jacobi
// eigen-solver) runs under the plane blocks instead of forming the launch's tail.
#ifndef MSFL_FIT_DEFER
#define MSFL_FIT_DEFER 0            /* 1: the pivoted-QR fallback of the whole-batch fit kernel deferred to fit_fallback_kernel (round 5: the hot kernel
                                       drops to 92 VGPRs / 5 waves per SIMD and 0.098 ms, but the extra launch costs what it saves: 0.1063 vs 0.1045 ms
                                       per pass for both; docs/rejected_experiments.md).  `make defer` builds it; the test of the path runs on either. */
#endif
#ifndef MSFL_FIT_SPLIT_WAVES
#define MSFL_FIT_SPLIT_WAVES (MSFL_FIT_DEFER ? 5 : 4)
#endif
__global__ void __launch_bounds__(kAssocBlock, MSFL_FIT_SPLIT_WAVES)
fit_scan2map_split_kernel(BatchView bv, const float4* __restrict__ map_c, const float4* __restrict__ map_s,
                          const int* __restrict__ nn, double line_ratio, double plane_tol, DeskewView dv,
                          double* __restrict__ rec, double* __restrict__ full, int edge_blocks, int* __restrict__ fallback) {
  if ((int)blockIdx.x < edge_blocks) fit_one<false, 1>(bv, map_c, map_s, nn, line_ratio, plane_tol, dv, rec, full, (int)blockIdx.x);
  else fit_one<false, 2, MSFL_FIT_DEFER != 0>(bv, map_c, map_s, nn, line_ratio, plane_tol, dv, rec, full, (int)blockIdx.x - edge_blocks, fallback);
}
// The planes whose neighbourhood failed the guards of plane_normal_centred (rank-deficient or ill-conditioned: a handful per million on
// scanned surfaces): the reference's pivoted QR (plane_fit's non-deferred body), one thread each over a fixed small grid, the count
// read on the device.  Re-arms the OTHER counter of the ping-pong pair for the next launch (this one is still being read).
__global__ void __launch_bounds__(64)
fit_fallback_kernel(BatchView bv, const float4* __restrict__ map_s, const int* __restrict__ nn, double plane_tol,
                    double* __restrict__ rec, double* __restrict__ full, const int* __restrict__ fallback, int* __restrict__ next_counter) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *next_counter = 0;
  const int n = fallback[0];
  (void)full;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    const int fi = fallback[1 + e];                              // a surf feature's index in its cloud array
    const int* in = nn + 5 * feature_slot(bv, false, fi);
    const float4 nb[5] = {map_s[in[0]], map_s[in[1]], map_s[in[2]], map_s[in[3]], map_s[in[4]]};
    const FitOut fo = plane_fit<false>(nb, plane_tol);
    double* out = rec + plane_rec_off(bv, fi);
    out[0] = fo.N.x; out[1] = fo.N.y; out[2] = fo.N.z; out[3] = dot(fo.N, fo.C);
  }
}

// {C, N} x n_records (host/debug format) -> compact internal records
__global__ void __launch_bounds__(256) pack_records_kernel(BatchView bv, const double* __restrict__ full, double* __restrict__ rec) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= batch_records(bv)) return;
  const int b = find_scan_wave(bv.rec_off, bv.n_scans, g);
  const int local = g - bv.rec_off[b];
  const int nc = bv.corner_off[b + 1] - bv.corner_off[b];
  const double* in = full + 6 * (size_t)g;
  if (local < nc) {
    double* out = rec + edge_rec_off(bv, bv.corner_off[b] + local);
#pragma unroll
    for (int k = 0; k < 6; k++) out[k] = in[k];
  } else {
    double* out = rec + plane_rec_off(bv, bv.surf_off[b] + (local - nc));
    out[0] = in[3]; out[1] = in[4]; out[2] = in[5]; out[3] = in[3] * in[0] + in[4] * in[1] + in[5] * in[2];
  }
}

// ---------------------------------------------------------------------------------------------
// K5 + K6: persistent per-scan trust-region LM with Huber loss (Ceres semantics)
// ---------------------------------------------------------------------------------------------

struct SolverParams {
  int    max_iterations;
  double huber;
  double radius0, radius_max, radius_min;
  double min_relative_decrease, min_diag, max_diag;
  double ftol, gtol, ptol;
  int    max_invalid;
  int    min_correspondences;   // 0 for the mapping matcher, 10 for odometry (.cc:262)
};

struct DevMatchInfo {   // mirrors msfl_match_info
  int status;
  int n_edge[2], n_plane[2];
  int lm_iterations[2], lm_successful[2];
  double initial_cost[2], final_cost[2];
};

constexpr int kAcc = 28;   // cost + g[6] + H upper[21]
static_assert(kAcc + 2 == 30, "block_reduce folds 30 values");

// ---- robustified normal equations, one residual row at a time ----------------------------------------------
// Ceres scales residual and Jacobian of a block by sqrt(rho') (Corrector with rho'' <= 0) and then forms
// J^T J / J^T r.  The products only ever contain sqrt(rho')^2, so the accumulation below weights the row
// with w = rho' directly: no square root per record, results equal up to the rounding of sqrt(w)^2 vs w.
// j[0..2] = d r/d t, j[3..5] = d r/d theta (tangent space, lidar_factor.cc:19,39 in closed form).
__device__ __forceinline__ void acc_row_w(double (&acc)[kAcc], const double (&j)[6], double r, double w) {
  double jw[6];
#pragma unroll
  for (int k = 0; k < 6; k++) jw[k] = w * j[k];
#pragma unroll
  for (int k = 0; k < 6; k++) acc[1 + k] = __builtin_fma(jw[k], r, acc[1 + k]);
  int n = 7;
#pragma unroll
  for (int p2 = 0; p2 < 6; p2++)
#pragma unroll
    for (int q = p2; q < 6; q++) { acc[n] = __builtin_fma(jw[p2], j[q], acc[n]); n++; }
}

// HuberLoss(a) on s = |r|^2 given |r|: rho0 = rho(s), w = rho'(s) (Ceres loss_function.cc; the Corrector's
// max(DBL_MIN, .) guard is kept)
__device__ __forceinline__ void huber_weight(double a, double s, double abs_r, double& rho0, double& w) {
  const double b = a * a;
  if (s > b) {
    rho0 = 2.0 * a * abs_r - b;
    w = fmax(2.2250738585072014e-308, a * fast_rcp(abs_r));
  } else {
    rho0 = s; w = 1.0;
  }
}

// One evaluation pass of a scan's records at pose T: cost, g = J^T r, H = J^T J (robustified,
// tangent space).  lidar_factor.cc:7-44 + Ceres HuberLoss/Corrector.
// LDS-resident copy of the first K plane records + their points (44 B each, structure of arrays so
// that consecutive lanes hit consecutive banks).  A solve re-reads every record in each of its ~4-7
// evaluation passes: what fits here is fetched from HBM once per solve instead of once per pass.
#ifndef MSFL_LM_BLOCK
#define MSFL_LM_BLOCK 128
#endif
constexpr int kLmBlock = MSFL_LM_BLOCK;                        // threads per scan in the scan-to-map LM solve
#ifndef MSFL_ODOM_LM_BLOCK
#define MSFL_ODOM_LM_BLOCK 128
#endif
constexpr int kOdomLmBlock = MSFL_ODOM_LM_BLOCK;               // scan-to-scan: ~500 records per pair
#ifndef MSFL_LM_EDGE_CACHE
#define MSFL_LM_EDGE_CACHE 0
#endif
// Optional LDS cache for the edge records (build switch, default off): the edge loop has only 2-3
// iterations per lane and takes 26 % of the evaluation time for 11 % of the records, but caching them
// (at the planes' expense) measured 0.336 / 0.338 ms vs 0.331 ms without.
// cache sizes per workgroup size: 76 KB x 2 workgroups per CU (256 threads), 36.6 KB x 4 (128: the small
// scan-to-scan problems, measured 0.365 ms per call vs 0.448 with 256 threads and 0.382 with 64), 17 KB x 8 (64:
// one wavefront per problem, no cross-wave barrier)
constexpr int lm_edge_cache(int block) { return block == 256 ? MSFL_LM_EDGE_CACHE : 0; }      // 60 B each
// 512 threads: the one-solve-per-call SLAM step, where the machine is empty and a solve is as long as its chain of passes
// (3 072 planes x 44 B = 132 KB, one workgroup per CU)
// trips of the streamed plane loop whose loads are requested together (see evaluate_pass): 512 threads have <= 10 trips in all
#ifndef MSFL_LM_GROUP_SMALL
#define MSFL_LM_GROUP_SMALL 8
#endif
#ifndef MSFL_LM_GROUP_LARGE
#define MSFL_LM_GROUP_LARGE 4
#endif
constexpr int lm_load_group(int block) { return block >= 512 ? MSFL_LM_GROUP_LARGE : MSFL_LM_GROUP_SMALL; }
constexpr int lm_plane_cache(int block) {
#ifdef MSFL_LM_PLANE_CACHE
  return block == 256 ? MSFL_LM_PLANE_CACHE : block == 128 ? 832 : block == 512 ? 3072 : 384;
#else
  return block == 256 ? ((1728 - (lm_edge_cache(256) * 60 + 43) / 44) & ~63) : block == 128 ? 832 : block == 512 ? 3072 : 384;
#endif
}
template <int BLOCK>
struct PlaneCache {
  static constexpr int kPlanes = lm_plane_cache(BLOCK), kEdges = lm_edge_cache(BLOCK);
  double nx[kPlanes], ny[kPlanes], nz[kPlanes], d0[kPlanes];
  float px[kPlanes], py[kPlanes], pz[kPlanes];
  double ecx[kEdges + 1], ecy[kEdges + 1], ecz[kEdges + 1], enx[kEdges + 1], eny[kEdges + 1], enz[kEdges + 1];
  float epx[kEdges + 1], epy[kEdges + 1], epz[kEdges + 1];
};

// Accepted edge correspondences of a solve, in index order.  Only ~40 % of the corner features pass the line test,
// and an edge row costs 2.7 plane rows: the first pass marks the accepted ones in a bit mask, wavefront 0 turns the
// mask into a dense index list, and the later passes (4 of 5) walk the list, so their lanes are all busy.
constexpr int kEdgeListMax = 1024;                 // edges beyond this index keep the checked walk
struct EdgeList {
  unsigned mask[kEdgeListMax / 32];
  unsigned short idx[kEdgeListMax];
  int n;
};

// FILL: first pass of a solve (records come from global memory and are copied into the cache);
// later passes read the cached part from LDS.  A thread only ever re-reads entries it wrote itself
// (same i -> thread mapping in every pass), so no barrier is needed around the cache.
#ifdef MSFL_LM_PROFILE
__device__ unsigned long long g_lm_prof[8];   // cycles (lane 0, summed over workgroups): eval, reduce, serial, total, passes
#define LM_T(x) const unsigned long long x = wall_clock64()
#ifndef MSFL_LM_PROFILE_BLOCK
#define MSFL_LM_PROFILE_BLOCK 0      /* 0: every instantiation; else only the one with this many threads */
#endif
#define LM_ADD(k, v) if (threadIdx.x == 0 && (MSFL_LM_PROFILE_BLOCK == 0 || MSFL_LM_PROFILE_BLOCK == (int)blockDim.x)) atomicAdd(&g_lm_prof[k], (unsigned long long)(v))
#else
#define LM_T(x)
#define LM_ADD(k, v)
#endif

__device__ __forceinline__ d3 lm_rotate(const quat& q, d3 v) { return quat_rotate(q, v); }

template <int BLOCK, bool FILL>
__device__ __forceinline__ void evaluate_pass(const pose7& T, double huber,
                                              const float4* __restrict__ corner, int nc,
                                              const float4* __restrict__ surf, int ns,
                                              const double* __restrict__ pprime,   // may be null
                                              const double* __restrict__ rec,      // this scan's edge records
                                              const double* __restrict__ recp,     // this scan's plane records
                                              PlaneCache<BLOCK>& pc, EdgeList& el,
                                              double (&acc)[kAcc], int& n_edge, int& n_plane) {
#pragma unroll
  for (int k = 0; k < kAcc; k++) acc[k] = 0.0;
  n_edge = 0; n_plane = 0;
  const mat3 R = quat_to_matrix(T.q);
  LM_T(t_eval_begin);
  // edges: {C, N}, r = N x (R p + t - C)                                       lidar_factor.cc:12
  // FILL: every edge, accepted ones marked; later passes: the dense list first, then the unlisted tail
  const int n_listed = FILL ? 0 : el.n;
  const int n_walk = FILL ? nc : n_listed + max(nc - kEdgeListMax, 0);
  auto edge_row = [&](int i, d3 C, d3 N, d3 p) __attribute__((always_inline)) {
    if (N.x == 0.0 && N.y == 0.0 && N.z == 0.0) return;        // rejected correspondence (never a listed one)
    if (FILL && i < kEdgeListMax) atomicOr(&el.mask[i >> 5], 1u << (i & 31));
    n_edge++;
    // d = R p + t - C through the rotation matrix (the quaternion sandwich costs three times the instructions; the two agree
    // to rounding), r = N x d
    const d3 d = mk3(__builtin_fma(R.m[0], p.x, __builtin_fma(R.m[1], p.y, R.m[2] * p.z)) + (T.t.x - C.x),
                     __builtin_fma(R.m[3], p.x, __builtin_fma(R.m[4], p.y, R.m[5] * p.z)) + (T.t.y - C.y),
                     __builtin_fma(R.m[6], p.x, __builtin_fma(R.m[7], p.y, R.m[8] * p.z)) + (T.t.z - C.z));
    const d3 r = mk3(__builtin_fma(N.y, d.z, -(N.z * d.y)), __builtin_fma(N.z, d.x, -(N.x * d.z)), __builtin_fma(N.x, d.y, -(N.y * d.x)));
    const double s = __builtin_fma(r.x, r.x, __builtin_fma(r.y, r.y, r.z * r.z));
    double rho0 = s, w = 1.0;
    if (s > huber * huber) {                                     // rho acts on the 3-vector norm of the block
      const double inv = fast_rsqrt(s);
      rho0 = 2.0 * huber * (s * inv) - huber * huber;
      w = fmax(2.2250738585072014e-308, huber * inv);
    }
    acc[0] += 0.5 * rho0;
    // rows of skew(N) (:18-19): a0 = (0, -Nz, Ny), a1 = (Nz, 0, -Nx), a2 = (-Ny, Nx, 0); rotation part p x (R^T a_k),
    // R^T a_k from two rows of R each
    const d3 l0 = mk3(__builtin_fma(N.y, R.m[6], -(N.z * R.m[3])), __builtin_fma(N.y, R.m[7], -(N.z * R.m[4])), __builtin_fma(N.y, R.m[8], -(N.z * R.m[5])));
    const d3 l1 = mk3(__builtin_fma(N.z, R.m[0], -(N.x * R.m[6])), __builtin_fma(N.z, R.m[1], -(N.x * R.m[7])), __builtin_fma(N.z, R.m[2], -(N.x * R.m[8])));
    const d3 l2 = mk3(__builtin_fma(N.x, R.m[3], -(N.y * R.m[0])), __builtin_fma(N.x, R.m[4], -(N.y * R.m[1])), __builtin_fma(N.x, R.m[5], -(N.y * R.m[2])));
    const d3 b0 = cross(p, l0), b1 = cross(p, l1), b2 = cross(p, l2);
    const double j0[6] = {0.0, -N.z, N.y, b0.x, b0.y, b0.z};
    const double j1[6] = {N.z, 0.0, -N.x, b1.x, b1.y, b1.z};
    const double j2[6] = {-N.y, N.x, 0.0, b2.x, b2.y, b2.z};
    acc_row_w(acc, j0, r.x, w);
    acc_row_w(acc, j1, r.y, w);
    acc_row_w(acc, j2, r.z, w);
  };
  // planes: {N, N.C}, r = N.(R p + t) - N.C                                    lidar_factor.cc:32
  const bool use_cache = (pprime == nullptr);       // deskew keeps f64 points in global memory
  auto plane_row = [&](d3 N, double d0, d3 p) __attribute__((always_inline)) {
    if (N.x == 0.0 && N.y == 0.0 && N.z == 0.0) return;        // rejected correspondence
    n_plane++;
    // r = N.(R p + t) - d0 = (R^T N).p + (N.t - d0): the vector l = R^T N is what the Jacobian's rotation part
    // p x (R^T N) needs anyway (:38-39), so the residual costs six fused multiply-adds on top of it
    const d3 l = mk3(__builtin_fma(R.m[0], N.x, __builtin_fma(R.m[3], N.y, R.m[6] * N.z)),
                     __builtin_fma(R.m[1], N.x, __builtin_fma(R.m[4], N.y, R.m[7] * N.z)),
                     __builtin_fma(R.m[2], N.x, __builtin_fma(R.m[5], N.y, R.m[8] * N.z)));
    const double nt = __builtin_fma(N.x, T.t.x, __builtin_fma(N.y, T.t.y, __builtin_fma(N.z, T.t.z, -d0)));
    const double r = __builtin_fma(l.x, p.x, __builtin_fma(l.y, p.y, __builtin_fma(l.z, p.z, nt)));
    const d3 b = cross(p, l);
    double rho0, w; huber_weight(huber, r * r, fabs(r), rho0, w);
    acc[0] += 0.5 * rho0;
    const double j[6] = {N.x, N.y, N.z, b.x, b.y, b.z};
    acc_row_w(acc, j, r, w);
  };
  // (a) the LDS-resident head of the plane list (later passes; a thread reads back what it wrote itself)
  int i = threadIdx.x;
  auto cached_planes = [&]() __attribute__((always_inline)) {
    if (!FILL && use_cache) {
      for (; i < min(ns, PlaneCache<BLOCK>::kPlanes); i += BLOCK)
        plane_row(mk3(pc.nx[i], pc.ny[i], pc.nz[i]), pc.d0[i], mk3((double)pc.px[i], (double)pc.py[i], (double)pc.pz[i]));
    }
  };
  if (pprime == nullptr && PlaneCache<BLOCK>::kEdges == 0) {
    // Round 6 (profiles/r06_lm_ablation.md: the edge rows were two to four DEPENDENT memory round trips per pass, 3.4 us of a 28 us pass and
    // 8 us of the first one): the loads of up to kEdgeGroup trips are requested together (clamped index, no branch around them), and in the
    // later passes the LDS-resident plane rows -- which wait for nothing -- are accumulated while those loads are in flight.  The rows of a
    // pass are therefore summed as {cached planes, edges, streamed planes} (first pass: {edges, planes} as before): a fixed order, a
    // function of the records alone.
    constexpr int kEdgeGroup = BLOCK >= 512 ? 1 : 4;
    bool first = true;
    for (int k0 = threadIdx.x; first || k0 < n_walk; k0 += kEdgeGroup * BLOCK) {
      double e6[kEdgeGroup][6];
      float4 ef[kEdgeGroup];
      int ei[kEdgeGroup];
#pragma unroll
      for (int u = 0; u < kEdgeGroup; u++) {
        const int k = min(k0 + u * BLOCK, max(n_walk - 1, 0));
        ei[u] = FILL ? k : (k < n_listed ? (int)el.idx[k] : kEdgeListMax + (k - n_listed));
      }
      if (n_walk > 0) {
#pragma unroll
        for (int u = 0; u < kEdgeGroup; u++) {
          const double* r6 = rec + 6 * (size_t)ei[u];
          e6[u][0] = r6[0]; e6[u][1] = r6[1]; e6[u][2] = r6[2]; e6[u][3] = r6[3]; e6[u][4] = r6[4]; e6[u][5] = r6[5];
          ef[u] = corner[ei[u]];                                   // curr_point: untransformed (:146)
        }
      }
      if (first) { cached_planes(); first = false; }
#pragma unroll
      for (int u = 0; u < kEdgeGroup; u++) {
        if (k0 + u * BLOCK >= n_walk) break;
        edge_row(ei[u], mk3(e6[u][0], e6[u][1], e6[u][2]), mk3(e6[u][3], e6[u][4], e6[u][5]), mk3((double)ef[u].x, (double)ef[u].y, (double)ef[u].z));
      }
    }
  } else {
    for (int k = threadIdx.x; k < n_walk; k += BLOCK) {
      const bool listed = !FILL && k < n_listed;
      const int i = FILL ? k : (listed ? (int)el.idx[k] : kEdgeListMax + (k - n_listed));
      d3 C, N, p;
      if (!FILL && pprime == nullptr && i < PlaneCache<BLOCK>::kEdges) {
        C = mk3(pc.ecx[i], pc.ecy[i], pc.ecz[i]); N = mk3(pc.enx[i], pc.eny[i], pc.enz[i]);
        p = mk3((double)pc.epx[i], (double)pc.epy[i], (double)pc.epz[i]);
      } else {
        const double* r6 = rec + 6 * (size_t)i;
        C = mk3(r6[0], r6[1], r6[2]);
        N = mk3(r6[3], r6[4], r6[5]);
        if (pprime) p = mk3(pprime[3 * (size_t)i], pprime[3 * (size_t)i + 1], pprime[3 * (size_t)i + 2]);
        else {
          const float4 f = corner[i];                              // curr_point: untransformed (:146)
          p = mk3((double)f.x, (double)f.y, (double)f.z);
          if (FILL && i < PlaneCache<BLOCK>::kEdges) {
            pc.ecx[i] = C.x; pc.ecy[i] = C.y; pc.ecz[i] = C.z; pc.enx[i] = N.x; pc.eny[i] = N.y; pc.enz[i] = N.z;
            pc.epx[i] = f.x; pc.epy[i] = f.y; pc.epz[i] = f.z;
          }
        }
      }
      edge_row(i, C, N, p);
    }
    cached_planes();
  }
  LM_T(t_edges_done);
  // (b) the streamed rest (everything in the FILL pass): ~1 GB per launch with all 1 024 solves resident (PMC r02).  A thread's
  // trips are a chain of dependent load round trips at two wavefronts per SIMD, so the loads of kGroup trips are requested
  // together (clamped index, no branch around them) and the rows then accumulated in ascending i as before: the sums are those
  // of the plain loop bit for bit.  Measured on the bench batch: 0.227 ms per launch ungrouped, 0.221 / 0.202 / 0.190 / 0.186 with
  // 2 / 4 / 6 / 8 trips per group (248 VGPRs, no spills).  Measured and rejected earlier: software pipelining the loads 1 / 2 / 3
  // trips ahead across the loop's back edge (0.249 / 0.258 / 0.264 ms); solving the batch in 2 / 4 launches of fewer scans
  // (0.41 / 0.71 ms: a launch takes ~0.2 ms however few problems it holds).
  constexpr int kGroup = lm_load_group(BLOCK);
  if (kGroup > 1 && use_cache) {
    for (; i < ns; i += kGroup * BLOCK) {
      double gn[kGroup][4];
      float4 gf[kGroup];
#pragma unroll
      for (int u = 0; u < kGroup; u++) {
        const int iu = min(i + u * BLOCK, ns - 1);
        const double* r4 = recp + 4 * (size_t)iu;
        gn[u][0] = r4[0]; gn[u][1] = r4[1]; gn[u][2] = r4[2]; gn[u][3] = r4[3];
        gf[u] = surf[iu];
      }
#pragma unroll
      for (int u = 0; u < kGroup; u++) {
        const int iu = i + u * BLOCK;
        if (iu >= ns) break;
        if (FILL && iu < PlaneCache<BLOCK>::kPlanes) {
          pc.nx[iu] = gn[u][0]; pc.ny[iu] = gn[u][1]; pc.nz[iu] = gn[u][2]; pc.d0[iu] = gn[u][3];
          pc.px[iu] = gf[u].x; pc.py[iu] = gf[u].y; pc.pz[iu] = gf[u].z;
        }
        plane_row(mk3(gn[u][0], gn[u][1], gn[u][2]), gn[u][3], mk3((double)gf[u].x, (double)gf[u].y, (double)gf[u].z));
      }
    }
  }
  for (; i < ns; i += BLOCK) {
    const double* r4 = recp + 4 * (size_t)i;
    const d3 N = mk3(r4[0], r4[1], r4[2]); const double d0 = r4[3];
    d3 p;
    if (pprime) { const size_t q = (size_t)(nc + i); p = mk3(pprime[3 * q], pprime[3 * q + 1], pprime[3 * q + 2]); }
    else {
      const float4 f = surf[i];                                  // curr_point: untransformed (:221)
      p = mk3((double)f.x, (double)f.y, (double)f.z);
      if (FILL && i < PlaneCache<BLOCK>::kPlanes) {
        pc.nx[i] = N.x; pc.ny[i] = N.y; pc.nz[i] = N.z; pc.d0[i] = d0;
        pc.px[i] = f.x; pc.py[i] = f.y; pc.pz[i] = f.z;
      }
    }
    plane_row(N, d0, p);
  }
  LM_T(t_planes_done);
}

// Trust-region state of one solve.  Lives in LDS so that the evaluation passes (which every lane
// runs) keep a small register footprint; only lane 0 of the workgroup touches it.
struct TrState {
  double sys[kAcc];          // {cost, g, H upper} at the current iterate x
  double scale[6], diagonal[6];
  double x[7], cand[7];
  double cost, gmax, x_norm, radius, decrease_factor, model_cost_change;
  int iteration, invalid, successful, reuse_diagonal, step_ok;
};

template <int BLOCK>
struct LmShared {
  double part[BLOCK / 64][kAcc + 2];   // per wavefront: {cost, g, H} and the two correspondence counts (as doubles: exact)
  double red[kAcc];          // reduced {cost, g, H} of the last pass
  int    cnt[2];
  TrState tr;
  int    go;                 // 1: evaluate candidate, 0: finished
};

// deterministic block reduction, fixed order inside the wave and across waves.  Inside the wave the 30 values are not
// reduced one butterfly each (30 x 6 exchanges): every exchange step halves the number of values a lane carries — at
// offset 32 the lower half-wave keeps values 0..14 and receives the upper half's share of them while the upper half keeps
// 15..29, and so on down to one value per lane pair: 15 + 8 + 4 + 2 + 1 + 1 = 31 exchanges.  The lane whose bits say
// (b5 b4 b3 b2 b1) ends up with value 15 b5 + 8 b4 + 4 b3 + 2 b2 + b1 (slot 15 of each half is padding).
template <int BLOCK>
__device__ __forceinline__ void block_reduce(LmShared<BLOCK>& sh, double (&acc)[kAcc], int n_edge, int n_plane) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double v16[16], v8[8], v4[4], v2[2], x;
  {
    const bool up = lane & 32;
#pragma unroll
    for (int k = 0; k < 15; k++) {
      const double a = acc[k < kAcc ? k : 0];
      const double b = k + 15 < kAcc ? acc[k + 15 < kAcc ? k + 15 : 0] : (k + 15 == kAcc ? (double)n_edge : (double)n_plane);
      v16[k] = (up ? b : a) + __shfl_xor(up ? a : b, 32);
    }
    v16[15] = 0.0;
  }
  {
    const bool up = lane & 16;
#pragma unroll
    for (int k = 0; k < 8; k++) v8[k] = (up ? v16[k + 8] : v16[k]) + __shfl_xor(up ? v16[k] : v16[k + 8], 16);
  }
  {
    const bool up = lane & 8;
#pragma unroll
    for (int k = 0; k < 4; k++) v4[k] = (up ? v8[k + 4] : v8[k]) + __shfl_xor(up ? v8[k] : v8[k + 4], 8);
  }
  {
    const bool up = lane & 4;
#pragma unroll
    for (int k = 0; k < 2; k++) v2[k] = (up ? v4[k + 2] : v4[k]) + __shfl_xor(up ? v4[k] : v4[k + 2], 4);
  }
  {
    const bool up = lane & 2;
    x = (up ? v2[1] : v2[0]) + __shfl_xor(up ? v2[0] : v2[1], 2);
  }
  x += __shfl_xor(x, 1);
  const int slot = (lane >> 1) & 15;
  if (!(lane & 1) && slot != 15) sh.part[wave][15 * (lane >> 5) + slot] = x;
  __syncthreads();
  if (threadIdx.x < kAcc + 2) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; w++) s += sh.part[w][threadIdx.x];
    if (threadIdx.x < kAcc) sh.red[threadIdx.x] = s;
    else sh.cnt[threadIdx.x - kAcc] = (int)s;
  }
  // no barrier here: the sums are written and then read (trust-region logic, lane 0) inside wavefront 0, in program
  // order; every other reader comes after the barrier that ends the serial section
}

// 6x6 SPD solve by Cholesky, fully unrolled; A symmetric (full storage), returns false if not PD
__device__ __forceinline__ bool chol_solve6(const double (&A)[6][6], const double (&b)[6], double (&x)[6]) {
  double L[6][6];
  bool ok = true;
#pragma unroll
  for (int i = 0; i < 6; i++) {
#pragma unroll
    for (int j = 0; j <= i; j++) {
      double s = A[i][j];
#pragma unroll
      for (int k = 0; k < j; k++) s -= L[i][k] * L[j][k];
      if (i == j) {
        if (!(s > 0.0)) ok = false;
        L[i][i] = sqrt(s);
      } else {
        L[i][j] = s / L[j][j];
      }
    }
  }
  double y[6];
#pragma unroll
  for (int i = 0; i < 6; i++) {
    double s = b[i];
#pragma unroll
    for (int k = 0; k < i; k++) s -= L[i][k] * y[k];
    y[i] = s / L[i][i];
  }
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    double s = y[i];
#pragma unroll
    for (int k = i + 1; k < 6; k++) s -= L[k][i] * x[k];
    x[i] = s / L[i][i];
  }
#pragma unroll
  for (int i = 0; i < 6; i++) if (!isfinite(x[i])) ok = false;
  return ok;
}

// The only use of the gradient max-norm is the test `gmax <= gradient_tolerance`.  Its translation components are
// |x_i - fl(x_i - g_i)| >= |g_i| - |x_i - g_i| 2^-53 (1 + 2^-53): for |x_i| < 1e6 that can only be <= gtol when
// |g_i| <= gtol + 2.3e-10, so a translation gradient component above 2 gtol + 1e-9 decides the test without the
// manifold step (sincos, quaternion product, normalisation) that the exact value needs.  Returns a value > gtol then.
__device__ __forceinline__ double gradient_max_norm(const pose7& x, const double* g);
__device__ __forceinline__ double gradient_max_norm_for_test(const pose7& x, const double* g, double gtol) {
  const double gt = fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2])));
  const double xt = fmax(fabs(x.t.x), fmax(fabs(x.t.y), fabs(x.t.z)));
  if (gt > 2.0 * gtol + 1e-9 && xt < 1e6) return gt;
  return gradient_max_norm(x, g);
}
__device__ __forceinline__ double gradient_max_norm(const pose7& x, const double* g) {
  const pose7 xp = pose_plus(x, mk3(-g[0], -g[1], -g[2]), mk3(-g[3], -g[4], -g[5]));
  double m = fabs(x.t.x - xp.t.x);
  m = fmax(m, fabs(x.t.y - xp.t.y)); m = fmax(m, fabs(x.t.z - xp.t.z));
  m = fmax(m, fabs(x.q.x - xp.q.x)); m = fmax(m, fabs(x.q.y - xp.q.y));
  m = fmax(m, fabs(x.q.z - xp.q.z)); m = fmax(m, fabs(x.q.w - xp.q.w));
  return m;
}
__device__ __forceinline__ double pose_norm(const pose7& x) {
  return sqrt(x.t.x * x.t.x + x.t.y * x.t.y + x.t.z * x.t.z + x.q.x * x.q.x + x.q.y * x.q.y + x.q.z * x.q.z + x.q.w * x.q.w);
}

// unpack the packed accumulator into full H (6x6), g
__device__ __forceinline__ void unpack_system(const double* sys, double (&H)[6][6], double (&g)[6]) {
#pragma unroll
  for (int k = 0; k < 6; k++) g[k] = sys[1 + k];
  int n = 7;
#pragma unroll
  for (int p = 0; p < 6; p++)
#pragma unroll
    for (int q = p; q < 6; q++) { H[p][q] = sys[n]; H[q][p] = sys[n]; n++; }
}

// FinalizeIterationAndCheckIfMinimizerCanContinue + ComputeTrustRegionStep (+ HandleInvalidStep,
// ParameterToleranceReached), lane 0 only.  Returns 1 when tr.cand holds a candidate to evaluate.
// `prm` BY VALUE: through a reference the (kernel-argument) structure is read with flat loads, and because the stores to
// the LDS state in between might alias it, every use re-read its field from memory behind an `s_waitcnt vmcnt(0)` — a
// dozen global round trips per call on a single lane (found with clock reads inside the function: the six clamps of the
// LM diagonal alone took 7 us).
__device__ __noinline__ int tr_propose(TrState& tr, const SolverParams prm) {
  for (;;) {
    if (tr.iteration >= prm.max_iterations) return 0;
    if (tr.step_ok && tr.gmax <= prm.gtol) return 0;
    if (tr.radius < prm.radius_min) return 0;
    tr.iteration++;
    // A = S H S (+ LM damping); its Cholesky factor goes to L, A itself stays readable for the model cost change.
    double A[6][6], gs[6], y[6], lm2[6];
    {
      int n = 7;
#pragma unroll
      for (int p = 0; p < 6; p++)
#pragma unroll
        for (int q = p; q < 6; q++) { const double v = tr.sys[n++] * tr.scale[p] * tr.scale[q]; A[p][q] = v; A[q][p] = v; }
    }
#pragma unroll
    for (int i = 0; i < 6; i++) gs[i] = tr.sys[1 + i] * tr.scale[i];
    if (!tr.reuse_diagonal) {
#pragma unroll
      for (int i = 0; i < 6; i++) tr.diagonal[i] = fmin(fmax(A[i][i], prm.min_diag), prm.max_diag);
    }
    double hs_diag[6];                     // undamped diagonal of S H S, for the model cost change below
    // Ceres' LM strategy appends sqrt(diagonal / radius) as extra Jacobian rows, i.e. adds diagonal / radius to the
    // normal equations: formed directly here (one division for all six), equal up to the rounding of sqrt(.)^2
    const double inv_radius = 1.0 / tr.radius;
#pragma unroll
    for (int i = 0; i < 6; i++) {
      lm2[i] = tr.diagonal[i] * inv_radius;
      hs_diag[i] = A[i][i];
      A[i][i] += lm2[i];
    }
    // factorise, solve, then step^T Hs step from the very values A was formed from (off-diagonal entries of A and the
    // diagonal saved before damping): no second pass over the packed system in LDS.  The factor is kept as
    // L[i][j] (i > j) and the RECIPROCALS of its diagonal: one reciprocal square root per column instead of a
    // square root and 2 x (5 - j) + 2 divisions (this serial, single-lane chain is latency bound: ~200 clocks each)
    double L[6][6], dinv[6];
    bool ok = true;
#pragma unroll
    for (int jc = 0; jc < 6; jc++) {
      double s = A[jc][jc];
#pragma unroll
      for (int k = 0; k < jc; k++) s = __builtin_fma(-L[jc][k], L[jc][k], s);
      if (!(s > 0.0)) ok = false;
      dinv[jc] = fast_rsqrt(s);
#pragma unroll
      for (int i = jc + 1; i < 6; i++) {
        double v = A[i][jc];
#pragma unroll
        for (int k = 0; k < jc; k++) v = __builtin_fma(-L[i][k], L[jc][k], v);
        L[i][jc] = v * dinv[jc];
      }
    }
#pragma unroll
    for (int i = 0; i < 6; i++) {
      double s = gs[i];
#pragma unroll
      for (int k = 0; k < i; k++) s = __builtin_fma(-L[i][k], y[k], s);
      y[i] = s * dinv[i];
    }
    double step[6];
#pragma unroll
    for (int i = 5; i >= 0; i--) {
      double s = y[i];
#pragma unroll
      for (int k = i + 1; k < 6; k++) s = __builtin_fma(-L[k][i], step[k], s);
      step[i] = s * dinv[i];
    }
#pragma unroll
    for (int i = 0; i < 6; i++) { if (!isfinite(step[i])) ok = false; step[i] = -step[i]; }
    tr.reuse_diagonal = 1;
    double mcc = 0.0;
    if (ok) {
      double gts = 0.0, shs = 0.0;
#pragma unroll
      for (int i = 0; i < 6; i++) {
        gts += gs[i] * step[i];
#pragma unroll
        for (int j = 0; j < 6; j++) {
          // Hs[i][j]: the value A was formed from (A itself off the diagonal, the saved entry on it)
          shs += step[i] * (i == j ? hs_diag[i] : A[i][j]) * step[j];
        }
      }
      mcc = -gts - 0.5 * shs;
    }
    tr.model_cost_change = mcc;
    if (!ok || !(mcc > 0.0)) {          // HandleInvalidStep
      if (++tr.invalid >= prm.max_invalid) return 0;
      tr.radius *= 0.5;
      tr.step_ok = 0;
      continue;
    }
    tr.invalid = 0;
    const pose7 x = load_pose(tr.x);
    const pose7 cand = pose_plus(x, mk3(step[0] * tr.scale[0], step[1] * tr.scale[1], step[2] * tr.scale[2]),
                                 mk3(step[3] * tr.scale[3], step[4] * tr.scale[4], step[5] * tr.scale[5]));
    store_pose(tr.cand, cand);
    double sn = 0.0;
#pragma unroll
    for (int i = 0; i < 7; i++) { const double d = tr.x[i] - tr.cand[i]; sn += d * d; }
    sn = sqrt(sn);
    if (sn <= prm.ptol * (tr.x_norm + prm.ptol)) return 0;   // ParameterToleranceReached
    return 1;
  }
}

// FunctionToleranceReached / IsStepSuccessful / HandleSuccessfulStep / HandleUnsuccessfulStep,
// lane 0 only; `red` = {cost, g, H} evaluated at tr.cand.  Returns 1 to continue.
__device__ __noinline__ int tr_decide(TrState& tr, const double* red, const SolverParams prm) {
  const double cand_cost = red[0];
  const double cost_change = tr.cost - cand_cost;
  if (fabs(cost_change) <= prm.ftol * tr.cost) return 0;
  const double rel = cost_change / tr.model_cost_change;
  if (rel > prm.min_relative_decrease) {
#pragma unroll
    for (int i = 0; i < 7; i++) tr.x[i] = tr.cand[i];
#pragma unroll
    for (int k = 0; k < kAcc; k++) tr.sys[k] = red[k];
    const pose7 x = load_pose(tr.x);
    tr.x_norm = pose_norm(x);
    tr.cost = cand_cost;
    tr.gmax = gradient_max_norm_for_test(x, tr.sys + 1, prm.gtol);
    const double t = 2.0 * rel - 1.0;
    tr.radius = fmin(tr.radius / fmax(1.0 / 3.0, 1.0 - t * t * t), prm.radius_max);
    tr.decrease_factor = 2.0;
    tr.reuse_diagonal = 0;
    tr.step_ok = 1;
    tr.successful++;
  } else {
    tr.radius = tr.radius / tr.decrease_factor;
    tr.decrease_factor *= 2.0;
    tr.reuse_diagonal = 1;
    tr.step_ok = 0;
  }
  return 1;
}


// One workgroup per scan, persistent over all trust-region iterations of one ceres::Solve.
// Lane 0 runs the (serial, tiny) trust-region logic between evaluation passes; every pass
// evaluates cost AND the normal equations at the candidate, so an accepted step needs no second
// pass (Ceres re-evaluates; the values are identical).
#ifndef MSFL_LM_WAVES
#define MSFL_LM_WAVES 2
#endif
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK, MSFL_LM_WAVES)
lm_solve_kernel(BatchView bv, const double* __restrict__ pprime_all, const double* __restrict__ rec_all,
                double* __restrict__ poses, int* __restrict__ status, DevMatchInfo* __restrict__ info,
                int outer_it, SolverParams prm) {
  __shared__ LmShared<BLOCK> sh;
  __shared__ PlaneCache<BLOCK> s_cache;
  __shared__ EdgeList s_edges;
  const int b = blockIdx.x;
  if (status[b] != 0) return;
  if (threadIdx.x < kEdgeListMax / 32) s_edges.mask[threadIdx.x] = 0;
  __syncthreads();
  const int nc = bv.corner_off[b + 1] - bv.corner_off[b];
  const int ns = bv.surf_off[b + 1] - bv.surf_off[b];
  const float4* corner = bv.corner + bv.corner_off[b];
  const float4* surf = bv.surf + bv.surf_off[b];
  const size_t r0 = (size_t)bv.rec_off[b];
  const double* rec = rec_all + edge_rec_off(bv, bv.corner_off[b]);
  const double* recp = rec_all + plane_rec_off(bv, bv.surf_off[b]);
  const double* pprime = pprime_all ? pprime_all + 3 * r0 : nullptr;
  double* pose_g = poses + 7 * (size_t)b;
  TrState& tr = sh.tr;
  LM_T(t_begin);
  {
    double acc[kAcc];
    int ne, np;
    const pose7 T = load_pose(pose_g);
    LM_T(t0);
    evaluate_pass<BLOCK, true>(T, prm.huber, corner, nc, surf, ns, pprime, rec, recp, s_cache, s_edges, acc, ne, np);
    LM_T(t1);
    block_reduce<BLOCK>(sh, acc, ne, np);
    LM_T(t2);
    LM_ADD(0, t1 - t0); LM_ADD(1, t2 - t1); LM_ADD(4, 1);
  }
  LM_T(t_s0);
  // mask -> index list (32 lanes of wavefront 0, one mask word each; ascending index order, so the list and with it the
  // summation order of the later passes is a function of the records alone)
  if (threadIdx.x < kEdgeListMax / 32) {
    unsigned m = s_edges.mask[threadIdx.x];
    const int c = __popc(m);
    int incl = c;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) { const int v = __shfl_up(incl, o); if ((int)threadIdx.x >= o) incl += v; }
    int at = incl - c;
    while (m) { const int bit = __ffs((int)m) - 1; s_edges.idx[at++] = (unsigned short)(32 * threadIdx.x + bit); m &= m - 1; }
    if (threadIdx.x == kEdgeListMax / 32 - 1) s_edges.n = incl;
  }
  if (threadIdx.x == 0) {
    const int n_edge = sh.cnt[0], n_plane = sh.cnt[1];
    int go = 1;
    if (info) { info[b].n_edge[outer_it] = n_edge; info[b].n_plane[outer_it] = n_plane; }
    tr.cost = 0.0; tr.iteration = 0; tr.successful = 0;
    if (n_edge + n_plane < prm.min_correspondences) {
      status[b] = 1;                       // MSFL_TOO_FEW_CORRESPONDENCES (odometry_scan_matcher.cc:262-267)
      if (info) info[b].status = 1;
      go = 0;
    } else if (n_edge + n_plane == 0) {
      go = 0;                              // Ceres: empty problem, parameters untouched
    } else {
#pragma unroll
      for (int k = 0; k < kAcc; k++) tr.sys[k] = sh.red[k];
#pragma unroll
      for (int i = 0; i < 7; i++) tr.x[i] = pose_g[i];
      tr.cost = sh.red[0];
      if (info) info[b].initial_cost[outer_it] = tr.cost;
      // jacobi_scaling from iteration 0: 1 / (1 + sqrt(diag(J^T J)))
      const int dg[6] = {7, 13, 18, 22, 25, 27};   // packed positions of H[i][i]
#pragma unroll
      for (int i = 0; i < 6; i++) tr.scale[i] = 1.0 / (1.0 + sqrt(sh.red[dg[i]]));
      const pose7 x = load_pose(tr.x);
      tr.gmax = gradient_max_norm_for_test(x, tr.sys + 1, prm.gtol);
      tr.x_norm = pose_norm(x);
      tr.radius = prm.radius0; tr.decrease_factor = 2.0; tr.model_cost_change = 0.0;
      tr.invalid = 0; tr.reuse_diagonal = 0; tr.step_ok = 1;
      go = tr_propose(tr, prm);
    }
    sh.go = go;
  }
  __syncthreads();
  LM_T(t_s1);
  LM_ADD(2, t_s1 - t_s0);
  bool solved = (sh.cnt[0] + sh.cnt[1] >= prm.min_correspondences) && (sh.cnt[0] + sh.cnt[1] > 0);
  while (sh.go) {
    double acc[kAcc];
    int ne, np;
    const pose7 T = load_pose(tr.cand);   // lane 0 overwrites go / cand only after the reduction's barrier, which every
                                           // thread reaches after this read: no barrier of its own needed
    LM_T(t0);
    evaluate_pass<BLOCK, false>(T, prm.huber, corner, nc, surf, ns, pprime, rec, recp, s_cache, s_edges, acc, ne, np);
    LM_T(t1);
    block_reduce<BLOCK>(sh, acc, ne, np);
    LM_T(t2);
#ifdef MSFL_LM_PROFILE
    if (threadIdx.x == 0) {
      const unsigned long long c0 = wall_clock64();
      const int cont = tr_decide(tr, sh.red, prm);
      const unsigned long long c1 = wall_clock64();
      sh.go = cont ? tr_propose(tr, prm) : 0;
      const unsigned long long c2 = wall_clock64();
      atomicAdd(&g_lm_prof[6], c1 - c0); atomicAdd(&g_lm_prof[7], c2 - c1);
    }
#else
    if (threadIdx.x == 0) sh.go = tr_decide(tr, sh.red, prm) ? tr_propose(tr, prm) : 0;
#endif
    __syncthreads();
    LM_T(t3);
    LM_ADD(0, t1 - t0); LM_ADD(1, t2 - t1); LM_ADD(2, t3 - t2); LM_ADD(4, 1);
  }
  LM_T(t_end);
  LM_ADD(3, t_end - t_begin); LM_ADD(5, 1);
  if (threadIdx.x == 0) {
    if (solved) {
#pragma unroll
      for (int i = 0; i < 7; i++) pose_g[i] = tr.x[i];
    }
    if (info) {
      info[b].lm_iterations[outer_it] = tr.iteration;
      info[b].lm_successful[outer_it] = tr.successful;
      info[b].final_cost[outer_it] = tr.cost;
    }
  }
}

}  // namespace msfl

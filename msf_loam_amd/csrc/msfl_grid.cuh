// msfl_grid.cuh — N1: device-resident local map store.
//
// Replaces HybridGridImpl (src/slam/map/hybrid_grid.cc:462-534), the Cartographer-derived
// DynamicGrid<NestedGrid<FlatGrid>> of per-3 m-cell point clouds behind
//   HybridGrid::InsertScan          (:503-521)  append to cells, VoxelGrid-filter the touched cells
//   HybridGrid::GetSurroundedCloud  (:470-501)  union of the cells around the transformed scan points
//
// GPU formulation (round 3: insert cost proportional to the SCAN, not to the map).
//   * The points live in one POOL; a cell is a slab [start, start + count) of it, its points in
//     pcl::VoxelGrid's output order (voxel index z-major, x-minor).
//   * A CELL TABLE sorted by the 42-bit cell key (iz, iy, ix: 14 bits each, the reference's +-8192
//     limit, hybrid_grid.cc:460) holds {key, start, count}.
//   * InsertScan sorts only the NEW points by (cell, voxel) key, finds the cells they touch, and
//     rebuilds exactly those cells: one workgroup per touched cell merges the cell's old points
//     (voxel re-derived from their coordinates, because the reference re-runs the filter over the
//     cell's own cloud, :513-520) with the new ones in LDS, one centroid per voxel run, f32 sums in
//     the order pcl's CentroidPoint sees (old points first, then the new ones in scan order).  The
//     rebuilt cells go to fresh slabs at the top of the pool, the old slabs become garbage, the
//     cell table is rebuilt by a merge (O(cells)); the pool is compacted when it fills up.
//     Untouched cells are never read or written.
//   * Every kernel takes its sizes from a device-resident GridState / device counters and is launched
//     over a host-known upper bound, so a whole insert (or surround query) is stream-ordered with no
//     host round trip: the per-scan SLAM step (msfl_slam.inc) chains them without synchronising.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "msfl_math.cuh"

namespace msfl {

constexpr int kGridCellBits = 14;                 // +-8192 cells, the reference's hard limit (hybrid_grid.cc:460)
constexpr int kGridVoxBits = 7;
constexpr unsigned long long kGridBadKey = ~0ull;
constexpr unsigned kGridVoxMask = (1u << (3 * kGridVoxBits)) - 1u;

struct GridStoreDesc {
  float resolution;       // 3.0
  float inv_leaf;         // 1 / leaf, f32 like pcl::VoxelGrid::inverse_leaf_size_
  double leaf;
};

// device-resident bookkeeping of one map store
struct GridState {
  int n_cells;            // live cells
  int n_points;           // live points
  int pool_top;           // first free pool slot
  int epoch;              // surround queries stamp hit cells with it (no clearing pass)
  // --- the insert in flight (reset by grid_begin_kernel) ---
  int bad;                // a new point lies outside the +-8192-cell range: the insert is dropped, map unchanged
  int overflow;           // pool / cell-table capacity exceeded (the host sizes both so that this cannot happen)
  int n_touched;          // cells the scan touches
  int n_new_cells;        // of which not yet in the table
  int work;               // pool slots the rebuilt cells take (their slabs' capacities)
  int delta_points;       // sum over touched cells of (points after - points before)
  int n_big;              // touched cells the small rebuild kernel left to the large one
  int surround_total;     // points delivered by the last surround query
};

// HybridGridBase::GetCellIndex (:422-426): lround(double(p / resolution)), division in f32
__device__ __forceinline__ int grid_cell_index(float v, float resolution) {
  return (int)lround((double)(v / resolution));
}

__device__ __forceinline__ unsigned long long grid_cell_key(int ix, int iy, int iz) {
  const int lim = 1 << (kGridCellBits - 1);
  if (ix < -lim || ix >= lim || iy < -lim || iy >= lim || iz < -lim || iz >= lim) return kGridBadKey;
  return ((unsigned long long)(iz + lim) << (2 * kGridCellBits)) | ((unsigned long long)(iy + lim) << kGridCellBits) |
         (unsigned long long)(ix + lim);
}
__device__ __forceinline__ void grid_cell_of_key(unsigned long long cell, int& ix, int& iy, int& iz) {
  const int lim = 1 << (kGridCellBits - 1);
  ix = (int)(cell & ((1u << kGridCellBits) - 1u)) - lim;
  iy = (int)((cell >> kGridCellBits) & ((1u << kGridCellBits) - 1u)) - lim;
  iz = (int)(cell >> (2 * kGridCellBits)) - lim;
}

// voxel coordinate of pcl::VoxelGrid (floor(p * inv_leaf) in f32) relative to a per-cell base that
// only has to be monotone: rel = k - (floor((c - 0.5) * resolution / leaf) - 2)
__device__ __forceinline__ int grid_vox_rel(float v, int c, const GridStoreDesc& d) {
  const int k = (int)floorf(v * d.inv_leaf);
  const int base = (int)floor(((double)c - 0.5) * (double)d.resolution / d.leaf) - 2;
  return k - base;
}
// 21-bit voxel-in-cell key (z major) of a point filed under cell (ix, iy, iz).  An old centroid is the mean of points
// of its cell, so its voxel lies inside the cell's range up to rounding; the clamp only keeps a stray value addressable.
__device__ __forceinline__ unsigned grid_vox_key(float4 p, int ix, int iy, int iz, const GridStoreDesc& d) {
  const int lim = (1 << kGridVoxBits) - 1;
  const int rx = min(max(grid_vox_rel(p.x, ix, d), 0), lim), ry = min(max(grid_vox_rel(p.y, iy, d), 0), lim),
            rz = min(max(grid_vox_rel(p.z, iz, d), 0), lim);
  return ((unsigned)rz << (2 * kGridVoxBits)) | ((unsigned)ry << kGridVoxBits) | (unsigned)rx;
}

__device__ __forceinline__ int grid_dev_n(int n_cap, const int* __restrict__ n_dev) {
  return n_dev ? min(max(*n_dev, 0), n_cap) : n_cap;
}

// (the per-insert fields of GridState are left zeroed by grid_finish_kernel, so an insert needs no reset launch)

// (cell key << 21 | voxel inside the cell) of a point, kGridBadKey when the cell or the voxel is out of range
__device__ __forceinline__ unsigned long long grid_point_key(float4 p, const GridStoreDesc& d) {
  const int ix = grid_cell_index(p.x, d.resolution), iy = grid_cell_index(p.y, d.resolution), iz = grid_cell_index(p.z, d.resolution);
  unsigned long long key = grid_cell_key(ix, iy, iz);
  if (key != kGridBadKey) {
    const int rx = grid_vox_rel(p.x, ix, d), ry = grid_vox_rel(p.y, iy, d), rz = grid_vox_rel(p.z, iz, d);
    const int lim = 1 << kGridVoxBits;
    if (rx < 0 || rx >= lim || ry < 0 || ry >= lim || rz < 0 || rz >= lim) key = kGridBadKey;
    else key = (key << (3 * kGridVoxBits)) | ((unsigned long long)rz << (2 * kGridVoxBits)) | ((unsigned long long)ry << kGridVoxBits) |
               (unsigned long long)rx;
  }
  return key;
}

// ---- insert, step 1: (optional) rigid transform + 63-bit key of every new point -------------------------------------
// pose != null: p <- TransformPoint(pose, p) (laser_mapping.cc:24-31 / rigid_transform.h:131-137: f32 -> f64 -> q p + t -> f32),
// written to xf (what the rest of the insert reads); pose == null: xf is not touched and the caller passes pts as xf.
// Slots at and beyond the device-side count get the bad key so that they sort to the end.
__global__ void __launch_bounds__(256)
grid_key_kernel(const float4* __restrict__ pts, int n_cap, const int* __restrict__ n_dev, const double* __restrict__ pose,
                float4* __restrict__ xf, GridStoreDesc d, unsigned long long* __restrict__ keys, int* __restrict__ vals,
                GridState* __restrict__ st) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_cap) return;
  const int n = grid_dev_n(n_cap, n_dev);
  vals[i] = i;
  if (i >= n) { keys[i] = kGridBadKey; return; }
  float4 p = pts[i];
  if (pose) {
    const float3 q = transform_point_f32(load_pose(pose), p.x, p.y, p.z);
    p = make_float4(q.x, q.y, q.z, p.w);
    xf[i] = p;
  }
  const unsigned long long key = grid_point_key(p, d);
  if (key == kGridBadKey) st->bad = 1;              // a non-finite coordinate lands here too (lround of NaN / inf is out of range)
  keys[i] = key;
}

// ---- insert, step 2 (after the stable sort of the new keys): heads of the cell runs ----------------------------------
__global__ void __launch_bounds__(256)
grid_touch_flag_kernel(const unsigned long long* __restrict__ skeys, int n_cap, const int* __restrict__ n_dev, int* __restrict__ head) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_cap) return;
  const int n = grid_dev_n(n_cap, n_dev);
  int h = 0;
  if (i < n) {
    const unsigned long long c = skeys[i] >> (3 * kGridVoxBits);
    h = (i == 0 || c != (skeys[i - 1] >> (3 * kGridVoxBits))) ? 1 : 0;
  }
  head[i] = h;
}

// ---- insert, step 3 (after the inclusive scan of the heads): the sorted list of touched cells ------------------------
// t_key[t] = cell key, t_ns[t] = first position of the cell's new points in the sorted order (t_ns[n_touched] = n).
// A dropped insert (bad point) touches nothing.
__global__ void __launch_bounds__(256)
grid_touch_list_kernel(const unsigned long long* __restrict__ skeys, const int* __restrict__ head, const int* __restrict__ tpos, int n_cap,
                       const int* __restrict__ n_dev, unsigned long long* __restrict__ t_key, int* __restrict__ t_ns, GridState* __restrict__ st) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_cap) return;
  const int n = grid_dev_n(n_cap, n_dev);
  const bool dropped = st->bad != 0;
  if (i == 0 && (n == 0 || dropped)) { st->n_touched = 0; t_ns[0] = 0; }
  if (i >= n || dropped) return;
  if (head[i]) { const int t = tpos[i] - 1; t_key[t] = skeys[i] >> (3 * kGridVoxBits); t_ns[t] = i; }
  if (i == n - 1) { st->n_touched = tpos[i]; t_ns[tpos[i]] = n; }
}

// ---- one-workgroup forms for the per-scan SLAM step ------------------------------------------------------------------------
// A scan's insert / surround works on tens of thousands of keys and hundreds of cells: the parallel primitives (a device-wide
// scan is two launches and a host-side planning call) cost more in launches than in work.  These kernels do the same job in
// one 1 024-thread workgroup; results are the same integers.  (Not the kNN index's cell table: a one-workgroup scan of its
// ~100 k counters measured 71 us against the device-wide scan's 9.)

// inclusive scan of one int per thread over the workgroup (1 024 threads); returns the inclusive value, total = workgroup sum
__device__ __forceinline__ int block_incl_scan_1024(int t, int* s_wave /* [16] */, int& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = t;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(incl, d); if (lane >= d) incl += u; }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  int pre = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 16; w++) { const int x = s_wave[w]; tot += x; pre += w < wave ? x : 0; }
  __syncthreads();                                      // s_wave is written again by the caller's next round
  total = tot;
  return incl + pre;
}

// insert, steps 2 + 3 in one launch (grid_touch_flag_kernel, the inclusive scan of the heads, grid_touch_list_kernel)
__global__ void __launch_bounds__(1024)
grid_touch_kernel(const unsigned long long* __restrict__ skeys, int n_cap, const int* __restrict__ n_dev, unsigned long long* __restrict__ t_key,
                  int* __restrict__ t_ns, GridState* __restrict__ st) {
  __shared__ int s_wave[16];
  const int n = grid_dev_n(n_cap, n_dev);
  if (n == 0 || st->bad != 0) {                       // a dropped insert (bad point) touches nothing
    if (threadIdx.x == 0) { st->n_touched = 0; t_ns[0] = 0; }
    return;
  }
  int carry = 0;
  for (int base = 0; base < n; base += 4 * 1024) {
    const int i0 = base + 4 * (int)threadIdx.x;
    unsigned long long c[5];
#pragma unroll
    for (int u = 0; u < 5; u++) { const int i = i0 - 1 + u; c[u] = (i >= 0 && i < n) ? skeys[i] >> (3 * kGridVoxBits) : ~0ull; }
    int h[4], t = 0;
#pragma unroll
    for (int u = 0; u < 4; u++) { h[u] = (i0 + u < n && (i0 + u == 0 || c[u + 1] != c[u])) ? 1 : 0; t += h[u]; }
    int total;
    int run = carry + block_incl_scan_1024(t, s_wave, total) - t;
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (h[u]) { t_key[run] = c[u + 1]; t_ns[run] = i0 + u; run++; }
    carry += total;
  }
  if (threadIdx.x == 0) { st->n_touched = carry; t_ns[carry] = n; }
}

// surround, per-cell emit counts + their exclusive scan in one launch (grid_emit_count_kernel + the device-wide scan)
__global__ void __launch_bounds__(1024)
grid_emit_scan_kernel(const int* __restrict__ stamp, const int* __restrict__ cell_cnt, int bound, int* __restrict__ cnt, int* __restrict__ off,
                      const GridState* __restrict__ st) {
  __shared__ int s_wave[16];
  const int n_cells = st->n_cells, epoch = st->epoch;
  bound = min(bound, n_cells);                          // grid_emit_kernel reads neither array beyond the live cells
  int carry = 0;
  for (int base = 0; base < bound; base += 4 * 1024) {
    const int c0 = base + 4 * (int)threadIdx.x;
    int v[4], t = 0;
#pragma unroll
    for (int u = 0; u < 4; u++) { const int c = c0 + u; v[u] = (c < bound && c < n_cells && stamp[c] == epoch) ? cell_cnt[c] : 0; t += v[u]; }
    int total;
    int run = carry + block_incl_scan_1024(t, s_wave, total) - t;
#pragma unroll
    for (int u = 0; u < 4; u++) { const int c = c0 + u; if (c < bound) { cnt[c] = v[u]; off[c] = run; } run += v[u]; }
    carry += total;
  }
}

// insert, step 1 + the stable sort for a list of at most 2 048 points (the corner side of a scan): keys as grid_key_kernel
// makes them, then a bitonic sort in LDS of (key, ordinal) — the ordinal as the tie-break is what a stable sort gives.
constexpr int kGridSortSmall = 2048;
constexpr int kGridOneBlockMax = 1 << 17;          // lists / cell tables up to this size take the one-workgroup forms
constexpr int kGridTouchOneBlockMax = 1 << 15;     // the touched-cell list of an insert: longer point lists (a 64-beam less-flat list) take the flag / scan / list launches
__global__ void __launch_bounds__(1024)
grid_key_sort_small_kernel(const float4* __restrict__ pts, int n_cap, const int* __restrict__ n_dev, const double* __restrict__ pose,
                           float4* __restrict__ xf, GridStoreDesc d, unsigned long long* __restrict__ skeys, int* __restrict__ svals,
                           float4* __restrict__ xs, GridState* __restrict__ st) {
  __shared__ unsigned long long s_key[kGridSortSmall];
  __shared__ unsigned short s_val[kGridSortSmall];
  const int n = grid_dev_n(n_cap, n_dev);
  for (int i = threadIdx.x; i < kGridSortSmall; i += 1024) {
    unsigned long long key = kGridBadKey;
    if (i < n) {
      float4 p = pts[i];
      if (pose) {
        const float3 q = transform_point_f32(load_pose(pose), p.x, p.y, p.z);
        p = make_float4(q.x, q.y, q.z, p.w);
        xf[i] = p;
      }
      key = grid_point_key(p, d);
      if (key == kGridBadKey) st->bad = 1;
    }
    s_key[i] = key; s_val[i] = (unsigned short)i;
  }
  __syncthreads();
  for (int k = 2; k <= kGridSortSmall; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      const int t = threadIdx.x;                                   // 1 024 compare-exchanges per stage
      const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
      const bool up = (lo & k) == 0;
      const unsigned long long ka = s_key[lo], kb = s_key[hi];
      const unsigned short va = s_val[lo], vb = s_val[hi];
      const bool a_after_b = ka > kb || (ka == kb && va > vb);
      if (a_after_b == up) { s_key[lo] = kb; s_key[hi] = ka; s_val[lo] = vb; s_val[hi] = va; }
      __syncthreads();
    }
  __threadfence_block();                                           // xf (written above) is read back below by other threads of the workgroup
  __syncthreads();
  const float4* src = pose ? xf : pts;
  for (int i = threadIdx.x; i < n_cap; i += 1024) {
    skeys[i] = s_key[i]; svals[i] = (int)s_val[i];
    if (i < n && (int)s_val[i] < n_cap) xs[i] = src[s_val[i]];                               // the points in sorted order (see grid_gather_sorted_kernel)
  }
}

// The new points in sorted order: a cell's new points are then one contiguous piece [t_ns, t_ns + n_new) and a voxel run inside it
// consecutive addresses, so the rebuild reads a point with ONE load that coalesces across a wavefront instead of the
// index load -> scattered point load chain (two dependent latencies per entry on the longest voxel run of the slowest cell).
__global__ void __launch_bounds__(256)
grid_gather_sorted_kernel(const float4* __restrict__ xf, const int* __restrict__ svals, int n_cap, const int* __restrict__ n_dev,
                          float4* __restrict__ xs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < grid_dev_n(n_cap, n_dev)) xs[i] = xf[svals[i]];
}

__device__ __forceinline__ int grid_find_cell(const unsigned long long* __restrict__ cell_keys, int n_cells, unsigned long long key) {
  int lo = 0, hi = n_cells;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const unsigned long long k = cell_keys[mid];
    if (k == key) return mid;
    if (k < key) lo = mid + 1; else hi = mid;
  }
  return -1;
}
__device__ __forceinline__ int grid_lower_bound(const unsigned long long* __restrict__ keys, int n, unsigned long long key) {
  int lo = 0, hi = n;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[mid] < key) lo = mid + 1; else hi = mid; }
  return lo;
}

// LDS capacities of the two rebuild instantiations (entries = old + new points of one cell, padded to a power of two)
#ifndef MSFL_GRID_SMALL_CAP
#define MSFL_GRID_SMALL_CAP 4096
#endif
#ifndef MSFL_GRID_SMALL_THREADS
#define MSFL_GRID_SMALL_THREADS 512
#endif
constexpr int kGridSmallCap = MSFL_GRID_SMALL_CAP;         // the LDS form: 512 threads, 32 KB of sort entries + 64 KB of points (round 5: 256 threads / 2 048
                                                           // entries left the 2 000 - 4 000-entry cells of a 64-beam sweep to the large form)
constexpr int kGridSmallThreads = MSFL_GRID_SMALL_THREADS;
constexpr int kGridLargeCap = 16384;              // 1024 threads, 128 KB; beyond that the sort runs in a global scratch slab
__host__ __device__ __forceinline__ int grid_pow2_at_least(int v) { int p = 1; while (p < v) p <<= 1; return p; }

// ---- insert, step 4: plan — one workgroup walks the touched cells: table lookup, slab sizes, prefix sums --------------
// t_old[t]   = index in the current cell table (-1: new cell)
// t_woff[t]  = pool slot of the rebuilt cell's slab (capacity old + new, + scratch for cells beyond the LDS capacity)
// t_rank[t]  = number of NEW cells among touched cells [0, t)
__global__ void __launch_bounds__(1024)
grid_plan_kernel(const unsigned long long* __restrict__ cell_keys, const int* __restrict__ cell_cnt, const unsigned long long* __restrict__ t_key,
                 const int* __restrict__ t_ns, int* __restrict__ t_old, int* __restrict__ t_woff, int* __restrict__ t_rank,
                 int pool_cap, int cell_cap, GridState* __restrict__ st) {
  __shared__ int s_a[16], s_b[16];
  __shared__ int s_carry_a, s_carry_b;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nt = st->n_touched, nc = st->n_cells, top = st->pool_top;
  if (tid == 0) { s_carry_a = 0; s_carry_b = 0; }
  __syncthreads();
  for (int base = 0; base < nt; base += 1024) {
    const int t = base + tid;
    int alloc = 0, is_new = 0, e = -1;
    if (t < nt) {
      e = grid_find_cell(cell_keys, nc, t_key[t]);
      const int cap = (e >= 0 ? cell_cnt[e] : 0) + (t_ns[t + 1] - t_ns[t]);
      alloc = cap > kGridLargeCap ? cap + grid_pow2_at_least(cap) / 2 : cap;      // 8-byte sort entries in 16-byte slots
      is_new = e < 0 ? 1 : 0;
    }
    int ia = alloc, ib = is_new;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int ua = __shfl_up(ia, o), ub = __shfl_up(ib, o); if (lane >= o) { ia += ua; ib += ub; } }
    if (lane == 63) { s_a[wave] = ia; s_b[wave] = ib; }
    __syncthreads();
    int pa = s_carry_a + ia - alloc, pb = s_carry_b + ib - is_new;
    for (int w = 0; w < wave; w++) { pa += s_a[w]; pb += s_b[w]; }
    if (t < nt) { t_old[t] = e; t_woff[t] = top + pa; t_rank[t] = pb; }
    __syncthreads();
    if (tid == 1023) { s_carry_a = pa + alloc; s_carry_b = pb + is_new; }
    __syncthreads();
  }
  if (tid == 0) {
    t_rank[nt] = s_carry_b;
    st->work = s_carry_a; st->n_new_cells = s_carry_b;
    if ((long long)top + s_carry_a > (long long)pool_cap || nc + s_carry_b > cell_cap) { st->overflow = 1; st->n_touched = 0; st->work = 0; st->n_new_cells = 0; }
  }
}

// ---- insert, step 5: rebuild the touched cells ------------------------------------------------------------------------
// Sort entries are (voxel key << 32 | ordinal): ordinals [0, n_old) = the cell's current points in their stored order,
// [n_old, E) = its new points in sorted-new order (= scan order inside a voxel: the sort of the new keys is stable), so
// the unique 64-bit keys sort into exactly the order pcl's stable view gives: per voxel old first, then new in arrival order.
template <int THREADS>
__device__ __forceinline__ void grid_bitonic(unsigned long long* __restrict__ s, int P, int k_first) {
  for (int k = k_first; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < (P >> 1); i += THREADS) {
        const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1)), hi = lo | j;       // the pair (lo, lo + j)
        const bool up = (lo & k) == 0;
        const unsigned long long a = s[lo], b = s[hi];
        if ((a > b) == up) { s[lo] = b; s[hi] = a; }
      }
      __syncthreads();
    }
  }
}

#ifdef MSFL_GRID_PROF
__device__ unsigned long long g_grid_prof[1024 * 8];      // profile build: per touched cell, phase durations in 10 ns ticks
#endif
template <int THREADS, int LDS_CAP, bool LARGE>
__global__ void __launch_bounds__(THREADS)
grid_rebuild_kernel(const float4* __restrict__ pool_in, float4* __restrict__ pool_out, const int* __restrict__ cell_start,
                    const int* __restrict__ cell_cnt, const float4* __restrict__ xs, const unsigned long long* __restrict__ skeys,
                    const unsigned long long* __restrict__ t_key, const int* __restrict__ t_ns,
                    const int* __restrict__ t_old, const int* __restrict__ t_woff, int* __restrict__ t_cnt, GridStoreDesc d,
                    GridState* __restrict__ st) {
  __shared__ unsigned long long s_ent[LDS_CAP];
  __shared__ int s_hcnt[LDS_CAP / 64];                  // run heads per (chunk, wavefront) of a piece, then their exclusive scan
  constexpr int kLongCap = LDS_CAP / 16 > 1024 ? 1024 : LDS_CAP / 16;      // runs the head thread hands to a wavefront
  __shared__ int s_long[kLongCap][3];
  __shared__ int s_nlong;
  static_assert(LARGE || LDS_CAP <= 32768, "s_run packs a run as (length << 16 | first entry)");
  __shared__ float4 s_pts[LARGE ? 1 : LDS_CAP];         // the LDS form: the cell's points in entry order
  __shared__ unsigned s_run[LARGE ? 1 : LDS_CAP];       //               its runs by rank: length << 16 | first entry
  __shared__ int s_carry, s_unsorted;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nt = st->n_touched;
  if (LARGE && st->n_big == 0) return;
  for (int t = blockIdx.x; t < nt; t += gridDim.x) {
    const int e = t_old[t];
    const int n_old = e >= 0 ? cell_cnt[e] : 0, o_start = e >= 0 ? cell_start[e] : 0;
    const int ns = t_ns[t], n_new = t_ns[t + 1] - ns;
    const int E = n_old + n_new;
    if (LARGE ? E <= kGridSmallCap : E > kGridSmallCap) {            // the other instantiation's cell
      if (!LARGE && tid == 0) atomicAdd(&st->n_big, 1);
      continue;
    }
    const int P = grid_pow2_at_least(E);
    const int woff = t_woff[t];
#ifdef MSFL_GRID_PROF
    unsigned long long tq[6]; tq[0] = wall_clock64();
#endif
    // entries beyond the LDS capacity (LARGE only): the sort runs in the scratch half of the cell's slab
    unsigned long long* ent = s_ent;
    if (LARGE && P > LDS_CAP) ent = reinterpret_cast<unsigned long long*>(pool_out + woff + E);
    int cx, cy, cz;
    grid_cell_of_key(t_key[t], cx, cy, cz);
    if (tid == 0) { s_unsorted = 0; s_carry = 0; }
    __syncthreads();
    // fill: old points ascending at the front, new points DESCENDING at the back, padding (max key) between them.  The new
    // keys are sorted; the old ones are too unless a centroid's f32 rounding moved it into another voxel, so the sequence
    // is bitonic and ONE merge network (log2 P steps) sorts it; an unsorted old part takes the full network instead.
    for (int i = tid; i < P; i += THREADS) {
      unsigned long long key = ~0ull;
      if (i < n_old) {
        const float4 p = pool_in[o_start + i];
        const unsigned v = grid_vox_key(p, cx, cy, cz, d);
        key = ((unsigned long long)v << 32) | (unsigned)i;
        if (i + 1 < n_old) { const unsigned vn = grid_vox_key(pool_in[o_start + i + 1], cx, cy, cz, d); if (vn < v) s_unsorted = 1; }
      } else if (i >= P - n_new) {
        const int j = P - 1 - i;                                       // new point j (sorted-new order)
        key = ((unsigned long long)((unsigned)skeys[ns + j] & kGridVoxMask) << 32) | (unsigned)(n_old + j);
      }
      ent[i] = key;
    }
    __syncthreads();
#ifdef MSFL_GRID_PROF
    tq[1] = wall_clock64();
#endif
    grid_bitonic<THREADS>(ent, P, s_unsorted ? 2 : P);
#ifdef MSFL_GRID_PROF
    tq[2] = wall_clock64();
#endif
    // voxel runs -> one centroid per run: f32 sums in entry order (old points first, then new ones in arrival order), as pcl's
    // CentroidPoint accumulates them.  The entries are walked in pieces of LDS_CAP (one piece unless the cell is beyond the LDS
    // form).  Pass 1 counts the run heads of every (chunk of THREADS entries, wavefront); one wavefront scans those counts; pass 2
    // recomputes a head's rank from them and sums its run -- no barrier between chunks, so the gathers of all short runs of the
    // cell are in flight together (rounds 3-5a ranked and summed chunk by chunk, three barriers each: 22 us of the 56 that a
    // 1 800-entry cell of a 64-beam sweep took).
    //   run of up to kThreadRun entries: its head's thread, four gathers in flight (the new points of a run are consecutive in xs)
    //   longer run (a 0.4 m voxel on a corridor wall 1.5 m from the sensor receives hundreds of points per scan; one thread chasing
    //   them set the whole kernel: 279 us per insert): queued, summed by a wavefront afterwards -- 64 entries per coalesced
    //   gather, the next 64 (of this run or of the wavefront's next run) loading meanwhile, then the same sequential additions
    //   through scalar broadcasts (bit-identical).
    // Rejected (round 5, docs/rejected_experiments.md): every run on a wavefront's scalar-broadcast walk (one useful addition per
    // 64-lane instruction: 2x slower on 16-beam cells), head threads taking runs of up to 64 entries (their gathers serialise).
    //   The LDS form (!LARGE) first copies the cell's points into LDS in entry order -- every thread's loads in flight together,
    //   one memory latency for the whole cell -- lists the runs by rank, and run r is summed by thread r from LDS (four reads in
    //   flight, all lanes at work; a wavefront's scalar-broadcast walk spends a 64-lane instruction per addition and measured
    //   ~4 us per 64 entries even from LDS).
    float4* out = pool_out + woff;
    auto entry_global = [&](int j) __attribute__((always_inline)) {
      const int ord = (int)(unsigned)ent[j];
      return ord < n_old ? pool_in[o_start + ord] : xs[ns + ord - n_old];
    };
    constexpr int kWaves = THREADS / 64, kPer = LDS_CAP / THREADS, kThreadRun = 16;
    if (!LARGE) {
      float4 pv[kPer];
#pragma unroll
      for (int c = 0; c < kPer; c++) { const int i = c * THREADS + tid; if (i < E) pv[c] = entry_global(i); }
#pragma unroll
      for (int c = 0; c < kPer; c++) { const int i = c * THREADS + tid; if (i < E) s_pts[i] = pv[c]; }
    }
    auto entry_point = [&](int j) __attribute__((always_inline)) { return LARGE ? entry_global(j) : s_pts[j]; };
    if (tid == 0) s_nlong = 0;
    int carry = 0;
    for (int sb = 0; sb < E; sb += LDS_CAP) {
      const int se = min(E, sb + LDS_CAP);
#pragma unroll
      for (int c = 0; c < kPer; c++) {
        const int i = sb + c * THREADS + tid;
        const bool head = i < se && (i == 0 || (unsigned)(ent[i] >> 32) != (unsigned)(ent[i - 1] >> 32));
        const unsigned long long hm = __ballot(head);
        if (lane == 0) s_hcnt[c * kWaves + wave] = __popcll(hm);
      }
      __syncthreads();
      if (wave == 0) {                                                     // exclusive scan of the kPer x kWaves counts (chunk-major = entry order)
        constexpr int kCnt = kPer * kWaves, kEach = (kCnt + 63) / 64;
        int v[kEach], sum = 0;
#pragma unroll
        for (int u = 0; u < kEach; u++) { const int k = lane * kEach + u; v[u] = k < kCnt ? s_hcnt[k] : 0; sum += v[u]; }
        int incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t2 = __shfl_up(incl, o); if (lane >= o) incl += t2; }
        int run = incl - sum;
#pragma unroll
        for (int u = 0; u < kEach; u++) { const int k = lane * kEach + u; if (k < kCnt) s_hcnt[k] = run; run += v[u]; }
        if (lane == 63) s_carry = incl;                                    // runs that start in this piece
      }
      __syncthreads();
#pragma unroll
      for (int c = 0; c < kPer; c++) {
        const int i = sb + c * THREADS + tid;
        if (sb + c * THREADS >= se) break;                                 // uniform
        unsigned v = 0; bool head = false;
        if (i < se) { v = (unsigned)(ent[i] >> 32); head = i == 0 || v != (unsigned)(ent[i - 1] >> 32); }
        const unsigned long long hm = __ballot(head);
        if (head) {
          const int r = carry + s_hcnt[c * kWaves + wave] + __popcll(hm & ((1ull << lane) - 1ull));
          // end of the run: the entries are sorted, so gallop (1, 2, 4, ... entries ahead) and bisect
          int lo = i, step = 1;
          while (lo + step < E && (unsigned)(ent[lo + step] >> 32) == v) { lo += step; step <<= 1; }
          int j = min(lo + step, E);
          while (j - lo > 1) { const int mid = (lo + j) >> 1; if ((unsigned)(ent[mid] >> 32) == v) lo = mid; else j = mid; }
          const int L = j - i;
          int slot = -1;
          if (LARGE && L > kThreadRun) { slot = atomicAdd(&s_nlong, 1); if (slot >= kLongCap) slot = -1; }
          if (!LARGE) s_run[r] = ((unsigned)L << 16) | (unsigned)i;        // the LDS form: the runs listed by rank, summed below
          else if (slot >= 0) { s_long[slot][0] = r; s_long[slot][1] = i; s_long[slot][2] = L; }
          else {
            float sx = 0.f, sy = 0.f, sz = 0.f, sw = 0.f;
            for (int q = i; q < j; q += 4) {                               // four gathers in flight, added in run order
              const float4 p0 = entry_point(q), p1 = entry_point(min(q + 1, j - 1)), p2 = entry_point(min(q + 2, j - 1)), p3 = entry_point(min(q + 3, j - 1));
              sx += p0.x; sy += p0.y; sz += p0.z; sw += p0.w;
              if (q + 1 < j) { sx += p1.x; sy += p1.y; sz += p1.z; sw += p1.w; }
              if (q + 2 < j) { sx += p2.x; sy += p2.y; sz += p2.z; sw += p2.w; }
              if (q + 3 < j) { sx += p3.x; sy += p3.y; sz += p3.z; sw += p3.w; }
            }
            const float cnt = (float)L;
            out[r] = make_float4(sx / cnt, sy / cnt, sz / cnt, sw / cnt);
          }
        }
      }
      carry += s_carry;
      __syncthreads();                                                     // s_hcnt / s_carry are rewritten by the next piece
    }
    if (!LARGE) {
      // the LDS form: run r to thread r (the run heads are sparse in the entries -- one in ~30 on a 64-beam sweep's cell -- so summing
      // them where they were found left two or three lanes of a wavefront at work, eight chunks in turn: 48 us on a 4 000-entry cell)
      for (int r = tid; r < carry; r += THREADS) {
        const unsigned w = s_run[r];
        const int i = (int)(w & 0xffffu), L = (int)(w >> 16), j = i + L;
        float sx = 0.f, sy = 0.f, sz = 0.f, sw = 0.f;
        for (int q = i; q < j; q += 4) {                                   // four LDS reads in flight, added in run order
          const float4 p0 = s_pts[q], p1 = s_pts[min(q + 1, j - 1)], p2 = s_pts[min(q + 2, j - 1)], p3 = s_pts[min(q + 3, j - 1)];
          sx += p0.x; sy += p0.y; sz += p0.z; sw += p0.w;
          if (q + 1 < j) { sx += p1.x; sy += p1.y; sz += p1.z; sw += p1.w; }
          if (q + 2 < j) { sx += p2.x; sy += p2.y; sz += p2.z; sw += p2.w; }
          if (q + 3 < j) { sx += p3.x; sy += p3.y; sz += p3.z; sw += p3.w; }
        }
        const float cnt = (float)L;
        out[r] = make_float4(sx / cnt, sy / cnt, sz / cnt, sw / cnt);
      }
    }
#ifdef MSFL_GRID_PROF
    tq[3] = wall_clock64();
#endif
    {
      const int n_long = min(s_nlong, kLongCap);
      int q = wave, c0 = 0;
      float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q < n_long) p = entry_point(s_long[q][1] + min(lane, s_long[q][2] - 1));
      float sx = 0.f, sy = 0.f, sz = 0.f, sw = 0.f;
      while (q < n_long) {
        const int r = s_long[q][0], L = s_long[q][2];
        const float4 pc = p;
        const int m = min(64, L - c0);
        int nq = q, nc0 = c0 + 64;
        if (nc0 >= L) { nq = q + kWaves; nc0 = 0; }
        if (nq < n_long) { const int nL = s_long[nq][2]; p = entry_point(s_long[nq][1] + nc0 + min(lane, nL - nc0 - 1)); }
        for (int e = 0; e < m; e++) {
          sx += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pc.x), e)); sy += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pc.y), e));
          sz += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pc.z), e)); sw += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pc.w), e));
        }
        if (nc0 == 0) {                                                    // the run is complete
          if (lane == 0) { const float cnt = (float)L; out[r] = make_float4(sx / cnt, sy / cnt, sz / cnt, sw / cnt); }
          sx = 0.f; sy = 0.f; sz = 0.f; sw = 0.f;
        }
        q = nq; c0 = nc0;
      }
    }
    __syncthreads();
#ifdef MSFL_GRID_PROF
    tq[4] = wall_clock64();
    if (tid == 0 && t < 1024) {
      unsigned long long* q = g_grid_prof + (size_t)t * 8;
      q[0] = (unsigned long long)E | ((unsigned long long)THREADS << 32); q[1] = tq[1] - tq[0]; q[2] = tq[2] - tq[1]; q[3] = tq[3] - tq[2]; q[4] = tq[4] - tq[3];
      q[5] = tq[0]; q[6] = tq[4]; q[7] = (unsigned long long)carry | ((unsigned long long)blockIdx.x << 32);
    }
#endif
    if (tid == 0) { const int R = carry; t_cnt[t] = R; atomicAdd(&st->delta_points, R - n_old); }
    __syncthreads();
  }
}

// ---- insert, step 6: the new cell table = old table merged with the new cells; touched cells point at their new slabs ---
__global__ void __launch_bounds__(256)
grid_commit_kernel(const unsigned long long* __restrict__ keys_in, const int* __restrict__ start_in, const int* __restrict__ cnt_in,
                   const int* __restrict__ stamp_in, unsigned long long* __restrict__ keys_out, int* __restrict__ start_out,
                   int* __restrict__ cnt_out, int* __restrict__ stamp_out, const unsigned long long* __restrict__ t_key,
                   const int* __restrict__ t_old, const int* __restrict__ t_woff, const int* __restrict__ t_rank, const int* __restrict__ t_cnt,
                   int bound, const GridState* __restrict__ st) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= bound) return;
  const int nc = st->n_cells, nt = st->n_touched;
  if (i < nc) {                                                        // an existing cell moves up by the new cells before it
    const unsigned long long k = keys_in[i];
    const int p = grid_lower_bound(t_key, nt, k);
    const bool touched = p < nt && t_key[p] == k;
    const int o = i + t_rank[p];
    keys_out[o] = k;
    start_out[o] = touched ? t_woff[p] : start_in[i];
    cnt_out[o] = touched ? t_cnt[p] : cnt_in[i];
    stamp_out[o] = stamp_in[i];
  } else if (i - nc < nt) {                                            // a touched cell that is new
    const int t = i - nc;
    if (t_old[t] >= 0) return;
    const unsigned long long k = t_key[t];
    const int o = grid_lower_bound(keys_in, nc, k) + t_rank[t];
    keys_out[o] = k; start_out[o] = t_woff[t]; cnt_out[o] = t_cnt[t]; stamp_out[o] = 0;
  }
}

// one thread: fold the insert into the state, publish {points, cells, pool_top, bad, overflow, work} for the host
__global__ void grid_finish_kernel(GridState* __restrict__ st, int* __restrict__ report) {
  st->n_cells += st->n_new_cells;
  st->n_points += st->delta_points;
  st->pool_top += st->work;
  if (report) {
    report[0] = st->n_points; report[1] = st->n_cells; report[2] = st->pool_top; report[3] = st->bad; report[4] = st->overflow;
    report[5] = st->n_touched; report[6] = st->work; report[7] = st->surround_total;
  }
  st->bad = 0; st->overflow = 0; st->n_touched = 0; st->n_new_cells = 0; st->work = 0; st->delta_points = 0; st->n_big = 0;
}

// ---- pool compaction / dump: the live cells back to back in table order -----------------------------------------------
// (after an exclusive scan of the counts into new_start)
__global__ void __launch_bounds__(256)
grid_zero_tail_kernel(int* __restrict__ cnt, int bound, const GridState* __restrict__ st) {   // counts beyond n_cells read as 0 by the scan
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < bound && i >= st->n_cells) cnt[i] = 0;
}
__global__ void __launch_bounds__(256)
grid_compact_kernel(const float4* __restrict__ pool_in, const int* __restrict__ start, const int* __restrict__ cnt, const int* __restrict__ new_start,
                    float4* __restrict__ out, int capacity, int* __restrict__ start_out, int bound, GridState* __restrict__ st, int set_top) {
  const int nc = st->n_cells;
  for (int c = blockIdx.x; c < min(nc, bound); c += gridDim.x) {
    const int s = start[c], m = cnt[c], o = new_start[c];
    for (int i = threadIdx.x; i < m; i += blockDim.x) if (o + i < capacity) out[o + i] = pool_in[s + i];
    if (start_out && threadIdx.x == 0) start_out[c] = o;
  }
  if (set_top && blockIdx.x == 0 && threadIdx.x == 0) st->pool_top = st->n_points;
}

// ---- GetSurroundedCloud ---------------------------------------------------------------------------------------------
// (the epoch a query stamps with is bumped at the END of the previous query's emit kernel; it starts at 1)

// pass 1: stamp the cells hit by pose_f32 * p + (i,j,k) metres
__global__ void __launch_bounds__(256)
grid_mark_kernel(const float4* __restrict__ scan, const int* __restrict__ idx, int n_cap, const int* __restrict__ n_dev,
                 const double* __restrict__ pose, float resolution, const unsigned long long* __restrict__ cell_keys, int* __restrict__ stamp,
                 const GridState* __restrict__ st) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= grid_dev_n(n_cap, n_dev)) return;
  const int n_cells = st->n_cells, epoch = st->epoch;
  const float4 p = scan[idx ? idx[t] : t];
  const float nrm = sqrtf(p.x * p.x + p.y * p.y + p.z * p.z);
  if ((double)nrm > 60.0) return;                                   // kDist (:474, :532)
  // pose.cast<float>() * point: Eigen Quaternionf _transformVector + translation, all f32 (:478)
  const float qx = (float)pose[3], qy = (float)pose[4], qz = (float)pose[5], qw = (float)pose[6];
  float ux = qy * p.z - qz * p.y, uy = qz * p.x - qx * p.z, uz = qx * p.y - qy * p.x;
  ux += ux; uy += uy; uz += uz;
  const float cx = qy * uz - qz * uy, cy = qz * ux - qx * uz, cz = qx * uy - qy * ux;
  const float wx = (p.x + qw * ux + cx) + (float)pose[0];
  const float wy = (p.y + qw * uy + cy) + (float)pose[1];
  const float wz = (p.z + qw * uz + cz) + (float)pose[2];
  // The 27 probes (wx + i, wy + j, wz + k), i, j, k in {-1, 0, 1} (:476-485) hit few DISTINCT cells: per axis the three
  // indices lround((w + o) / resolution) take one or two values when the cells are wider than 2 m (3 when narrower), so the
  // distinct (x, y, z) combinations are looked up once each — 8 binary searches instead of 27 for the reference's 3 m cells.
  int cxs[3], cys[3], czs[3], nx = 0, ny = 0, nz = 0;
  for (int o = -1; o <= 1; ++o) {
    const int a = grid_cell_index(wx + (float)o, resolution), b = grid_cell_index(wy + (float)o, resolution), c = grid_cell_index(wz + (float)o, resolution);
    bool da = false, db = false, dc = false;
    for (int q = 0; q < nx; q++) da = da || cxs[q] == a;
    for (int q = 0; q < ny; q++) db = db || cys[q] == b;
    for (int q = 0; q < nz; q++) dc = dc || czs[q] == c;
    if (!da) cxs[nx++] = a;
    if (!db) cys[ny++] = b;
    if (!dc) czs[nz++] = c;
  }
  for (int i = 0; i < nx; ++i)
    for (int j = 0; j < ny; ++j)
      for (int k = 0; k < nz; ++k) {
        const unsigned long long key = grid_cell_key(cxs[i], cys[j], czs[k]);
        if (key == kGridBadKey) continue;
        const int c = grid_find_cell(cell_keys, n_cells, key);
        if (c >= 0 && stamp[c] != epoch) stamp[c] = epoch;          // TryInsertGrid (:524-529)
      }
}

// per cell: number of points to emit (0 when not hit, 0 beyond the live cells)
__global__ void __launch_bounds__(256)
grid_emit_count_kernel(const int* __restrict__ stamp, const int* __restrict__ cell_cnt, int bound, int* __restrict__ cnt, const GridState* __restrict__ st) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= bound) return;
  cnt[c] = (c < st->n_cells && stamp[c] == st->epoch) ? cell_cnt[c] : 0;
}

// copy the hit cells in ascending cell order (after the exclusive scan of cnt into off); *n_out = total
__global__ void __launch_bounds__(256)
grid_emit_kernel(const float4* __restrict__ pool, const int* __restrict__ cell_start, const int* __restrict__ cnt, const int* __restrict__ off,
                 int bound, int capacity, float4* __restrict__ out, int* __restrict__ n_out, GridState* __restrict__ st) {
  const int nc = min(st->n_cells, bound);
  for (int c = blockIdx.x; c < nc; c += gridDim.x) {
    const int m = cnt[c];
    if (c == nc - 1 && threadIdx.x == 0) { const int total = off[c] + m; st->surround_total = total; if (n_out) *n_out = total; }
    if (m == 0) continue;
    const int s = cell_start[c], o = off[c];
    for (int i = threadIdx.x; i < m; i += blockDim.x) if (o + i < capacity) out[o + i] = pool[s + i];
  }
  if (nc == 0 && blockIdx.x == 0 && threadIdx.x == 0) { st->surround_total = 0; if (n_out) *n_out = 0; }
  if (blockIdx.x == 0 && threadIdx.x == 0) st->epoch += 1;         // nothing in this kernel reads it: the next query stamps with a fresh value
}

}  // namespace msfl

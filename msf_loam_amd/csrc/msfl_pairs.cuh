// msfl_pairs.cuh — one grid index per (map, scan) PAIR, built for a whole batch of pairs at once.
//
// The reference registers one scan against one map per MatchScan2Map call and rebuilds both kd-trees inside the call
// (mapping_scan_matcher.cc:66-73; call site laser_mapping.cc:304-311).  `msfl_match_scan2map_batch` covers "many scans, one map";
// this is the other shape the north star names, "many map-submap pairs": P maps (corner + surf cloud each), P scans, scan p is
// registered against map p only.  The P exact-kNN grids (msfl_kernels.cuh, K3) live in ONE set of arrays:
//   sorted map   all P clouds concatenated, each pair's points ordered by ITS grid's cells; the index word keeps the point's
//                position in the concatenated input, so ties break exactly as in a single call on that pair's cloud
//   cell table   pair p owns the slice [cell_base[p], cell_base[p + 1]) of one dense count / start array (the slice sizes are
//                host-known bounds, 2 x points + 4 096; the device grows a pair's cell edge until its grid fits its slice),
//                unused cells count zero, so ONE exclusive scan over the whole array yields every pair's cell starts
//   descriptors  GridDesc[P]
// Five launches + one rocPRIM scan per cloud kind, whatever P is.
#pragma once
#include <hip/hip_runtime.h>

#include "msfl_kernels.cuh"

namespace msfl {

// per-pair bounding boxes as six order-preserving ints (the encoding of grid_bbox_kernel): armed, then one thread per map point
// of the concatenated clouds; a workgroup that lies inside one pair (nearly all do) reduces first and issues six atomics
__global__ void __launch_bounds__(256) pairs_arm_kernel(int* __restrict__ bbox, int n_pairs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 6 * n_pairs) bbox[i] = (i % 6) < 3 ? 0x7fffffff : (int)0x80000000;
}

__global__ void __launch_bounds__(256)
pairs_bbox_kernel(const float4* __restrict__ pts, int n, const int* __restrict__ off, int n_pairs, int* __restrict__ bbox) {
  __shared__ float s_mn[4][3], s_mx[4][3];
  __shared__ int s_pair[2];
  const int i = off[0] + blockIdx.x * blockDim.x + threadIdx.x;
  const bool in = i < off[0] + n;
  const int p = in ? find_scan_wave(off, n_pairs, i) : -1;
  if (threadIdx.x == 0) s_pair[0] = p;
  const int last = min((int)(blockIdx.x * blockDim.x) + 255, n - 1);
  if ((int)(blockIdx.x * blockDim.x + threadIdx.x) == last) s_pair[1] = p;
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  if (in) {
    const float4 q = pts[i];
    if (isfinite(q.x) && isfinite(q.y) && isfinite(q.z)) { mn[0] = mx[0] = q.x; mn[1] = mx[1] = q.y; mn[2] = mx[2] = q.z; }
  }
  __syncthreads();
  if (s_pair[0] != s_pair[1]) {                  // a workgroup across a pair boundary: every lane for itself
    if (in && mn[0] != INFINITY)
      for (int a = 0; a < 3; a++) { atomicMin(&bbox[6 * p + a], float_to_ordered(mn[a])); atomicMax(&bbox[6 * p + 3 + a], float_to_ordered(mx[a])); }
    return;
  }
#pragma unroll
  for (int a = 0; a < 3; a++) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn[a] = fminf(mn[a], __shfl_xor(mn[a], o)); mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o)); }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { for (int a = 0; a < 3; a++) { s_mn[wave][a] = mn[a]; s_mx[wave][a] = mx[a]; } }
  __syncthreads();
  const int pp = s_pair[0];
  if (threadIdx.x < 3) {
    const int a = threadIdx.x;
    const float v = fminf(fminf(s_mn[0][a], s_mn[1][a]), fminf(s_mn[2][a], s_mn[3][a]));
    if (v != INFINITY) atomicMin(&bbox[6 * pp + a], float_to_ordered(v));
  } else if (threadIdx.x < 6) {
    const int a = threadIdx.x - 3;
    const float v = fmaxf(fmaxf(s_mx[0][a], s_mx[1][a]), fmaxf(s_mx[2][a], s_mx[3][a]));
    if (v != -INFINITY) atomicMax(&bbox[6 * pp + 3 + a], float_to_ordered(v));
  }
}

// one thread per pair: its GridDesc (cell edge grown until the grid fits the pair's slice of the cell table)
__global__ void __launch_bounds__(256)
pairs_desc_kernel(const int* __restrict__ bbox, int n_pairs, const int* __restrict__ cell_base, double radius, GridDesc* __restrict__ gdesc) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pairs) return;
  gdesc[p] = grid_desc_from_bbox(bbox + 6 * p, radius, cell_base[p + 1] - cell_base[p]);
}

// one thread per map point of the concatenated clouds: its pair, its cell in that pair's grid
__global__ void __launch_bounds__(256)
pairs_count_kernel(const float4* __restrict__ pts, int n, const int* __restrict__ off, int n_pairs, const int* __restrict__ cell_base,
                   const GridDesc* __restrict__ gdesc, int* __restrict__ cell_of, int* __restrict__ count) {
  const int i = off[0] + blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= off[0] + n) return;
  const int p = find_scan_wave(off, n_pairs, i);
  const GridDesc g = gdesc[p];
  const float4 q = pts[i];
  int c = -1;
  if (isfinite(q.x) && isfinite(q.y) && isfinite(q.z)) {
    int cx = grid_coord(q.x, g.ox, g.inv_cell_x, g.dx); cx = min(max(cx, 0), g.dx - 1);
    int cy = grid_coord(q.y, g.oy, g.inv_cell, g.dy); cy = min(max(cy, 0), g.dy - 1);
    int cz = grid_coord(q.z, g.oz, g.inv_cell, g.dz); cz = min(max(cz, 0), g.dz - 1);
    c = cell_base[p] + (cz * g.dy + cy) * g.dx + cx;
    atomicAdd(&count[c], 1);
  }
  cell_of[i - off[0]] = c;
}

// cursor[] = the counts on entry, consumed back to zero (as grid_scatter_kernel); sorted / pos_of are indexed from the
// concatenated cloud's first point (off[0])
__global__ void __launch_bounds__(256)
pairs_scatter_kernel(const float4* __restrict__ pts, int n, const int* __restrict__ off, const int* __restrict__ cell_of,
                     const int* __restrict__ cell_start, int* __restrict__ cursor, float4* __restrict__ sorted, int* __restrict__ pos_of) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int c = cell_of[k];
  if (c < 0) { pos_of[k] = -1; return; }
  const int slot = atomicSub(&cursor[c], 1) - 1;
  float4 q = pts[off[0] + k];
  q.w = __int_as_float(k);                    // index inside the concatenated cloud: unique, and ordered like the pair's own indices
  sorted[cell_start[c] + slot] = q;
  pos_of[k] = cell_start[c] + slot;
}

}  // namespace msfl

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

// Launch geometry of the three per-point kernels: blockIdx.y = pair, blockIdx.x = a chunk of kPairChunk points inside the pair
// (grid.x = the longest pair's chunk count; a pair's surplus workgroups leave at once).  A 1-D launch over the concatenated cloud
// needs every wavefront to find its pair first (a binary search over the offsets: eight dependent loads before the point is even
// requested), and that search, not the atomics, was what the first version of these kernels spent its time in.
constexpr int kPairChunk = 1024;                   // points per workgroup: four per lane, their loads issued together
__global__ void __launch_bounds__(256)
pairs_bbox_kernel(const float4* __restrict__ pts, const int* __restrict__ off, int* __restrict__ bbox) {
  __shared__ float s_mn[4][3], s_mx[4][3];
  const int pp = blockIdx.y;
  const int lo = off[pp] + (int)blockIdx.x * kPairChunk, hi = min(off[pp + 1], lo + kPairChunk);
  if (lo >= hi) return;
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  float4 q[4];
#pragma unroll
  for (int u = 0; u < 4; u++) { const int i = lo + u * 256 + (int)threadIdx.x; q[u] = pts[min(i, hi - 1)]; }
#pragma unroll
  for (int u = 0; u < 4; u++) {
    if (lo + u * 256 + (int)threadIdx.x < hi && isfinite(q[u].x) && isfinite(q[u].y) && isfinite(q[u].z)) {
      mn[0] = fminf(mn[0], q[u].x); mn[1] = fminf(mn[1], q[u].y); mn[2] = fminf(mn[2], q[u].z);
      mx[0] = fmaxf(mx[0], q[u].x); mx[1] = fmaxf(mx[1], q[u].y); mx[2] = fmaxf(mx[2], q[u].z);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; a++) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn[a] = fminf(mn[a], __shfl_xor(mn[a], o)); mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o)); }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { for (int a = 0; a < 3; a++) { s_mn[wave][a] = mn[a]; s_mx[wave][a] = mx[a]; } }
  __syncthreads();
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
pairs_count_kernel(const float4* __restrict__ pts, const int* __restrict__ off, const int* __restrict__ cell_base,
                   const GridDesc* __restrict__ gdesc, int* __restrict__ cell_of, int* __restrict__ count) {
  const int p = blockIdx.y;
  const int lo = off[p] + (int)blockIdx.x * kPairChunk, hi = min(off[p + 1], lo + kPairChunk);
  if (lo >= hi) return;
  const GridDesc g = gdesc[p];
  const int base = cell_base[p], o0 = off[0];
  float4 q[4];
#pragma unroll
  for (int u = 0; u < 4; u++) { const int i = lo + u * 256 + (int)threadIdx.x; q[u] = pts[min(i, hi - 1)]; }
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int i = lo + u * 256 + (int)threadIdx.x;
    if (i >= hi) break;
    int c = -1;
    if (isfinite(q[u].x) && isfinite(q[u].y) && isfinite(q[u].z)) {
      int cx = grid_coord(q[u].x, g.ox, g.inv_cell_x, g.dx); cx = min(max(cx, 0), g.dx - 1);
      int cy = grid_coord(q[u].y, g.oy, g.inv_cell, g.dy); cy = min(max(cy, 0), g.dy - 1);
      int cz = grid_coord(q[u].z, g.oz, g.inv_cell, g.dz); cz = min(max(cz, 0), g.dz - 1);
      c = base + (cz * g.dy + cy) * g.dx + cx;
      atomicAdd(&count[c], 1);
    }
    cell_of[i - o0] = c;
  }
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

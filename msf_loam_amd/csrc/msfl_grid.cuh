// msfl_grid.cuh — N1: device-resident local map store.
//
// Replaces HybridGridImpl (src/slam/map/hybrid_grid.cc:462-534), the Cartographer-derived
// DynamicGrid<NestedGrid<FlatGrid>> of per-3 m-cell point clouds behind
//   HybridGrid::InsertScan          (:503-521)  append to cells, VoxelGrid-filter the touched cells
//   HybridGrid::GetSurroundedCloud  (:470-501)  union of the cells around the transformed scan points
//
// GPU formulation.  The whole map is ONE point array sorted by a 63-bit key
//     key = cell(iz,iy,ix: 14 bits each) << 21 | voxel-in-cell(rz,ry,rx: 7 bits each)
// so a cell is a contiguous run and the voxels of a cell appear in pcl::VoxelGrid's output order
// (z major, x minor).  Because a filtered cell holds one centroid per voxel and the centroid of a
// single point is the point itself, re-filtering an untouched cell is the identity: InsertScan is
// therefore "append, stable-sort everything by key (old points first), one centroid per key run"
// — one rocPRIM radix sort + two light kernels, no per-cell containers, no pointer chasing.
// A map point keeps the CELL of the run it was created from (the reference keeps a point in the cell
// container it was pushed into and does not re-derive the cell from the centroid's coordinates); the
// voxel part of its key is re-derived from its coordinates whenever an insert touches its cell, because
// the reference's per-cell filter bins by coordinates (grid_rekey_touched_kernel).
// Cost per insert is O(map + scan) (one sort over everything); merging the sorted scan into the
// sorted map and re-filtering only the runs that received points would make it O(scan).
// GetSurroundedCloud marks cells through a binary search over the sorted unique cell keys.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "msfl_math.cuh"

namespace msfl {

constexpr int kGridCellBits = 14;                 // +-8192 cells, the reference's hard limit (hybrid_grid.cc:460)
constexpr int kGridVoxBits = 7;
constexpr unsigned long long kGridBadKey = ~0ull;

struct GridStoreDesc {
  float resolution;       // 3.0
  float inv_leaf;         // 1 / leaf, f32 like pcl::VoxelGrid::inverse_leaf_size_
  double leaf;
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

// voxel coordinate of pcl::VoxelGrid (floor(p * inv_leaf) in f32) relative to a per-cell base that
// only has to be monotone: rel = k - (floor((c - 0.5) * resolution / leaf) - 2)
__device__ __forceinline__ int grid_vox_rel(float v, int c, const GridStoreDesc& d) {
  const int k = (int)floorf(v * d.inv_leaf);
  const int base = (int)floor(((double)c - 0.5) * (double)d.resolution / d.leaf) - 2;
  return k - base;
}

// full 63-bit key of a map-frame point; kGridBadKey when out of range (caller reports MSFL_CAPACITY)
__global__ void __launch_bounds__(256) grid_point_key_kernel(const float4* __restrict__ pts, int n, GridStoreDesc d,
                                                              unsigned long long* __restrict__ keys, int* __restrict__ bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = pts[i];
  const int ix = grid_cell_index(p.x, d.resolution), iy = grid_cell_index(p.y, d.resolution), iz = grid_cell_index(p.z, d.resolution);
  unsigned long long key = grid_cell_key(ix, iy, iz);
  if (key != kGridBadKey) {
    const int rx = grid_vox_rel(p.x, ix, d), ry = grid_vox_rel(p.y, iy, d), rz = grid_vox_rel(p.z, iz, d);
    const int lim = 1 << kGridVoxBits;
    if (rx < 0 || rx >= lim || ry < 0 || ry >= lim || rz < 0 || rz >= lim) key = kGridBadKey;
    else key = (key << (3 * kGridVoxBits)) | ((unsigned long long)rz << (2 * kGridVoxBits)) | ((unsigned long long)ry << kGridVoxBits) |
               (unsigned long long)rx;
  }
  if (key == kGridBadKey) *bad = 1;
  keys[i] = key;
}

// cell part of the new scan's keys (sorted afterwards: the set of cells this insert touches)
__global__ void __launch_bounds__(256) grid_cell_of_key_kernel(const unsigned long long* __restrict__ keys, int n,
                                                                unsigned long long* __restrict__ cells) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) cells[i] = keys[i] >> (3 * kGridVoxBits);
}

// The reference re-runs the voxel filter over every cell the scan touched (hybrid_grid.cc:513-520): an old centroid of
// such a cell is binned by its COORDINATES again (an f32 centroid can round onto the next voxel's boundary and then
// merges with that voxel's content), while it stays in the cell container it was pushed into and nothing moves in the
// cells the scan does not touch.  So: old map points whose cell is among `touched` (sorted, duplicates allowed) get the
// voxel part of their key re-derived from their coordinates, relative to their STORED cell; all other keys stay.
__global__ void __launch_bounds__(256) grid_rekey_touched_kernel(const float4* __restrict__ pts, int n, GridStoreDesc d,
                                                                  const unsigned long long* __restrict__ touched, int n_touched,
                                                                  unsigned long long* __restrict__ keys, int* __restrict__ bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long cell = keys[i] >> (3 * kGridVoxBits);
  int lo = 0, hi = n_touched;                               // first entry >= cell
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (touched[mid] < cell) lo = mid + 1; else hi = mid; }
  if (lo == n_touched || touched[lo] != cell) return;
  const int lim_c = 1 << (kGridCellBits - 1);
  const int ix = (int)(cell & ((1u << kGridCellBits) - 1u)) - lim_c, iy = (int)((cell >> kGridCellBits) & ((1u << kGridCellBits) - 1u)) - lim_c,
            iz = (int)(cell >> (2 * kGridCellBits)) - lim_c;
  const float4 p = pts[i];
  const int rx = grid_vox_rel(p.x, ix, d), ry = grid_vox_rel(p.y, iy, d), rz = grid_vox_rel(p.z, iz, d);
  const int lim = 1 << kGridVoxBits;
  if (rx < 0 || rx >= lim || ry < 0 || ry >= lim || rz < 0 || rz >= lim) { *bad = 1; return; }
  keys[i] = (cell << (3 * kGridVoxBits)) | ((unsigned long long)rz << (2 * kGridVoxBits)) | ((unsigned long long)ry << kGridVoxBits) |
            (unsigned long long)rx;
}

// head flags of key runs (voxels) and of cell runs in the sorted key array
__global__ void __launch_bounds__(256) grid_flag_kernel(const unsigned long long* __restrict__ keys, int n,
                                                         int* __restrict__ vox_head, int* __restrict__ cell_head) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long k = keys[i];
  const bool first = (i == 0);
  vox_head[i] = (first || k != keys[i - 1]) ? 1 : 0;
  cell_head[i] = (first || (k >> (3 * kGridVoxBits)) != (keys[i - 1] >> (3 * kGridVoxBits))) ? 1 : 0;
}

// one thread per voxel head: centroid of the run in sorted (= arrival) order, f32 accumulators
// (pcl CentroidPoint); writes the filtered point, its key, and records cell starts.
__global__ void __launch_bounds__(256)
grid_centroid_kernel(const float4* __restrict__ pts, const int* __restrict__ order, const unsigned long long* __restrict__ keys,
                     const int* __restrict__ vox_head, const int* __restrict__ vox_pos, const int* __restrict__ cell_head,
                     const int* __restrict__ cell_pos, int n, float4* __restrict__ out_pts, unsigned long long* __restrict__ out_keys,
                     unsigned long long* __restrict__ cell_keys, int* __restrict__ cell_start,
                     const int* __restrict__ bad, int* __restrict__ result) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == n - 1) { result[0] = vox_pos[i]; result[1] = cell_pos[i]; result[2] = *bad; }   // {points, cells, out-of-range flag}: one read-back
  if (i >= n || !vox_head[i]) return;
  const unsigned long long k = keys[i];
  float sx = 0.f, sy = 0.f, sz = 0.f, st = 0.f;
  int j = i;
  for (; j < n && keys[j] == k; j++) {
    const float4 p = pts[order[j]];
    sx += p.x; sy += p.y; sz += p.z; st += p.w;
  }
  const float c = (float)(j - i);
  const int o = vox_pos[i] - 1;                    // inclusive scan -> index of this voxel in the new map
  out_pts[o] = make_float4(sx / c, sy / c, sz / c, st / c);
  out_keys[o] = k;
  if (cell_head[i]) {
    const int ci = cell_pos[i] - 1;
    cell_keys[ci] = k >> (3 * kGridVoxBits);
    cell_start[ci] = o;
  }
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

// GetSurroundedCloud pass 1: mark the cells hit by pose_f32 * p + (i,j,k) metres
__global__ void __launch_bounds__(256)
grid_mark_kernel(const float4* __restrict__ scan, int n, const double* __restrict__ pose, float resolution,
                 const unsigned long long* __restrict__ cell_keys, int n_cells, int* __restrict__ hit) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const float4 p = scan[t];
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
  for (int i = -1; i <= 1; ++i)
    for (int j = -1; j <= 1; ++j)
      for (int k = -1; k <= 1; ++k) {
        const unsigned long long key = grid_cell_key(grid_cell_index(wx + (float)i, resolution), grid_cell_index(wy + (float)j, resolution),
                                                     grid_cell_index(wz + (float)k, resolution));
        if (key == kGridBadKey) continue;
        const int c = grid_find_cell(cell_keys, n_cells, key);
        if (c >= 0) hit[c] = 1;                                     // TryInsertGrid (:524-529)
      }
}

// per cell: number of points to emit (0 when not hit)
__global__ void __launch_bounds__(256) grid_emit_count_kernel(const int* __restrict__ hit, const int* __restrict__ cell_start, int n_cells,
                                                               int n_points, int* __restrict__ cnt) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_cells) return;
  const int e = (c + 1 < n_cells) ? cell_start[c + 1] : n_points;
  cnt[c] = hit[c] ? (e - cell_start[c]) : 0;
}

// copy the hit cells, ascending cell order; one workgroup-stride loop per cell chunk
__global__ void __launch_bounds__(256) grid_emit_kernel(const float4* __restrict__ pts, const int* __restrict__ cell_start,
                                                         const int* __restrict__ cnt, const int* __restrict__ out_off, int n_cells,
                                                         int capacity, float4* __restrict__ out) {
  const int c = blockIdx.x;
  if (c >= n_cells) return;
  const int m = cnt[c];
  if (m == 0) return;
  const int s = cell_start[c], o = out_off[c];
  for (int i = threadIdx.x; i < m; i += blockDim.x)
    if (o + i < capacity) out[o + i] = pts[s + i];
}

}  // namespace msfl

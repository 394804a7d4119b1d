// msfl_knn_binned.cuh — throughput form of the scan-to-map 5-NN (mapping_scan_matcher.cc:123-128,193-198) for
// large batches: the queries of ALL scans are binned by map tile, sorted by fine cell inside a workgroup, and a
// wavefront then serves 64 queries that sit within a fraction of a metre of each other.
//
// Why: with one query per lane and per-lane candidate ranges (knn5_scan2map_kernel) the 64 lanes of a wavefront walk
// nine cell rows in lock-step and every loop position costs the longest of 64 ragged ranges (PMC r01h: 38 two-candidate
// steps per wavefront for 13 per lane, the running top-5 insertion network executed whenever ANY lane inserts: 2 312
// VALU instructions per wavefront for 24 useful candidates per lane).  Here the candidate set is WAVE-UNIFORM:
//   * the region walk (rows, x ranges, radius schedule) is scalar code, candidates arrive by scalar loads and are
//     broadcast operands of the 64 lanes' distance evaluations: no divergence, no per-lane bookkeeping;
//   * the running best-six is a branch-free chain of five v_med3_u32 and one v_min_u32 on 32-bit keys
//     (f32 distance bits with the low kBinOrdBits replaced by the candidate's ordinal in the wave's visit order);
//   * exactness: the key order equals the reference's (distance, index) order unless two of a lane's six best keys
//     agree in all kept distance bits; such lanes (≈1e-3 of them; always the case for exactly tied distances)
//     re-run the exact per-lane search (knn5_grid) inside the same kernel.  Results are bit-identical to
//     knn5_scan2map_kernel (tests compare the two paths).
//
// Pipeline per association pass (all on the handle's stream, nothing visits the host):
//   bin_count_kernel    per chunk of records: transform + tile id, LDS histogram -> hist[chunk][tile]
//   bin_prefix_kernel   column prefix over the chunks (16 workgroups) -> per-(chunk, tile) offsets, tile totals
//   bin_finalize_kernel tile starts, search work list (slices of <= kBinSlice queries)
//   bin_scatter_kernel  transform again, scatter {q.xyz, feature index} into tile order
//   knn5_binned_kernel  per slice: LDS counting sort by 12-bit Morton fine cell, then wave-uniform search
//   fit_binned_kernel   line / plane fit in binned order, record written to its feature's slot
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "msfl_kernels.cuh"

namespace msfl {

constexpr int kBinTiles = 1024;            // tiles of both maps together (one histogram row)
constexpr int kBinTilesCorner = 128;       // corner map: tiles [0, 128); surf map: [128, 1024)
#ifndef MSFL_BIN_EXP
#define MSFL_BIN_EXP 0      // timing experiments only, wrong results (2: statistics; 3: no distance loop, no hard pass; 4: also no staging; 5: no hard pass; 6: sort only)
#endif
#ifndef MSFL_BIN_SLICE
#define MSFL_BIN_SLICE 2048
#endif
constexpr int kBinSlice = MSFL_BIN_SLICE;  // queries one workgroup of the search kernel sorts and serves
constexpr int kBinOrdBits = 8;             // ordinal bits of a key: up to 256 candidates per wavefront tile
constexpr unsigned kBinOrdMask = (1u << kBinOrdBits) - 1u;
constexpr int kBinThreads = 512;           // threads per workgroup of the count / scatter kernels
#ifndef MSFL_BIN_BLOCK
#define MSFL_BIN_BLOCK 256
#endif
constexpr int kBinBlock = MSFL_BIN_BLOCK;  // threads per workgroup of the search kernel

struct MapView {
  const GridDesc* g;
  const float4* sorted;
  const int* cell_start;
  const int* pos_of;
};

// ---------------------------------------------------------------------------------------------
// binning
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int bin_tile_of(const GridDesc& g, float3 q) {
  int cx = grid_coord(q.x, g.ox, g.inv_cell_x, g.dx); cx = min(max(cx, 0), g.dx - 1);
  int cy = grid_coord(q.y, g.oy, g.inv_cell, g.dy); cy = min(max(cy, 0), g.dy - 1);
  int cz = grid_coord(q.z, g.oz, g.inv_cell, g.dz); cz = min(max(cz, 0), g.dz - 1);
  return ((cz / g.tk) * g.nty + (cy / g.tk)) * g.ntx + cx / (g.tk * kGridXSub);
}

struct BinItem { float3 q; int fi; int tile; };      // tile < 0: the record's scan has failed, nothing to do

// record g of the batch -> transformed query, feature index (into the corner or surf cloud) and tile.
// g must be consecutive across the lanes of a wavefront (find_scan_wave).
__device__ __forceinline__ BinItem bin_item(const BatchView& bv, const double* __restrict__ poses, const int* __restrict__ status,
                                            const GridDesc& gc, const GridDesc& gs, int g) {
  BinItem o; o.tile = -1; o.fi = 0; o.q = make_float3(0.f, 0.f, 0.f);
  const int b = find_scan_wave(bv.rec_off, bv.n_scans, g);
  if (status[b] != 0) return o;
  const int local = g - bv.rec_off[b];
  const int nc = bv.corner_off[b + 1] - bv.corner_off[b];
  const bool is_edge = local < nc;
  o.fi = is_edge ? bv.corner_off[b] + local : bv.surf_off[b] + (local - nc);
  const float4 f = is_edge ? bv.corner[o.fi] : bv.surf[o.fi];
  const pose7 T = load_pose(poses + 7 * b);
  o.q = transform_point_f32(T, f.x, f.y, f.z);                          // :123 / :193
  o.tile = is_edge ? bin_tile_of(gc, o.q) : kBinTilesCorner + bin_tile_of(gs, o.q);
  return o;
}

__global__ void __launch_bounds__(kBinThreads)
bin_count_kernel(BatchView bv, const double* __restrict__ poses, const int* __restrict__ status,
                 const GridDesc* __restrict__ gcp, const GridDesc* __restrict__ gsp, int chunk, int* __restrict__ hist) {
  __shared__ int s_hist[kBinTiles];
  for (int t = threadIdx.x; t < kBinTiles; t += kBinThreads) s_hist[t] = 0;
  __syncthreads();
  const GridDesc gc = *gcp, gs = *gsp;
  const int g0 = blockIdx.x * chunk, g1 = min(g0 + chunk, bv.n_records);
  for (int g = g0 + threadIdx.x; g < g1; g += kBinThreads) {
    const BinItem it = bin_item(bv, poses, status, gc, gs, g);
    if (it.tile >= 0) atomicAdd(&s_hist[it.tile], 1);
  }
  __syncthreads();
  int* row = hist + (size_t)blockIdx.x * kBinTiles;
  for (int t = threadIdx.x; t < kBinTiles; t += kBinThreads) row[t] = s_hist[t];
}

// hist[c][t] -> number of tile-t items in the chunks before c (in place); tile_total[t].  One workgroup per 64 tiles,
// 16 chunk groups of 64 lanes each: partial sums per group, prefix over the groups in LDS, then the running prefix.
__global__ void __launch_bounds__(1024)
bin_prefix_kernel(int* __restrict__ hist, int n_chunks, int* __restrict__ tile_total) {
  __shared__ int s_part[16][64];
  const int tl = threadIdx.x & 63, gi = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + tl;
  const int per = (n_chunks + 15) / 16;
  const int c0 = gi * per, c1 = min(c0 + per, n_chunks);
  int sum = 0;
  for (int c = c0; c < c1; c++) sum += hist[(size_t)c * kBinTiles + t];
  s_part[gi][tl] = sum;
  __syncthreads();
  int run = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) { const int v = s_part[k][tl]; if (k < gi) run += v; tot += v; }
  if (gi == 0) tile_total[t] = tot;
  for (int c = c0; c < c1; c++) {
    int* p = hist + (size_t)c * kBinTiles + t;
    const int v = *p;
    *p = run;
    run += v;
  }
}

// exclusive scan of one value per thread over a 1024-thread workgroup; returns the exclusive prefix, *total = the sum
__device__ __forceinline__ int block_excl_scan_1024(int v, int* s_wave /* [16] */, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o); if (lane >= o) incl += u; }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) { const int u = s_wave[k]; if (k < wave) base += u; tot += u; }
  __syncthreads();
  *total = tot;
  return base + incl - v;
}

// tile_start[t] (position of tile t in the binned array; [kBinTiles] = number of items) and the search work list:
// work_start[t] = first slice number of tile t ([kBinTiles] = number of slices)
__global__ void __launch_bounds__(1024)
bin_finalize_kernel(const int* __restrict__ tile_total, int* __restrict__ tile_start, int* __restrict__ work_start) {
  __shared__ int s_wave[16];
  const int t = threadIdx.x;
  const int n = tile_total[t];
  int tot;
  const int st = block_excl_scan_1024(n, s_wave, &tot);
  tile_start[t] = st;
  if (t == kBinTiles - 1) tile_start[kBinTiles] = tot;
  const int ws = block_excl_scan_1024((n + kBinSlice - 1) / kBinSlice, s_wave, &tot);
  work_start[t] = ws;
  if (t == kBinTiles - 1) work_start[kBinTiles] = tot;
}

__global__ void __launch_bounds__(kBinThreads)
bin_scatter_kernel(BatchView bv, const double* __restrict__ poses, const int* __restrict__ status,
                   const GridDesc* __restrict__ gcp, const GridDesc* __restrict__ gsp, int chunk,
                   const int* __restrict__ hist, const int* __restrict__ tile_start, float4* __restrict__ items) {
  __shared__ int s_cur[kBinTiles];
  const int* row = hist + (size_t)blockIdx.x * kBinTiles;
  for (int t = threadIdx.x; t < kBinTiles; t += kBinThreads) s_cur[t] = tile_start[t] + row[t];
  __syncthreads();
  const GridDesc gc = *gcp, gs = *gsp;
  const int g0 = blockIdx.x * chunk, g1 = min(g0 + chunk, bv.n_records);
  for (int g = g0 + threadIdx.x; g < g1; g += kBinThreads) {
    const BinItem it = bin_item(bv, poses, status, gc, gs, g);
    if (it.tile >= 0) {
      const int pos = atomicAdd(&s_cur[it.tile], 1);          // order inside a (chunk, tile) run is arbitrary: results do not depend on it
      items[pos] = make_float4(it.q.x, it.q.y, it.q.z, __int_as_float(it.fi));
    }
  }
}

// ---------------------------------------------------------------------------------------------
// wave-uniform search
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned med3_u32(unsigned a, unsigned b, unsigned c) {
  unsigned r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// wavefront reductions on the DPP network (six VALU instructions + one readlane each); every lane must be active
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xf, false); }
#define MSFL_WAVE_REDUCE(NAME, TYPE, OP, TO_I, FROM_I)                                                            \
  __device__ __forceinline__ TYPE NAME(TYPE v) {                                                                  \
    v = OP(v, FROM_I(dpp_i<0xb1, 0xf>(TO_I(v))));  /* quad_perm [1,0,3,2] */                                      \
    v = OP(v, FROM_I(dpp_i<0x4e, 0xf>(TO_I(v))));  /* quad_perm [2,3,0,1] */                                      \
    v = OP(v, FROM_I(dpp_i<0x141, 0xf>(TO_I(v)))); /* row_half_mirror */                                          \
    v = OP(v, FROM_I(dpp_i<0x140, 0xf>(TO_I(v)))); /* row_mirror */                                               \
    v = OP(v, FROM_I(dpp_i<0x142, 0xa>(TO_I(v)))); /* row_bcast:15 -> rows 1, 3 */                                \
    v = OP(v, FROM_I(dpp_i<0x143, 0xc>(TO_I(v)))); /* row_bcast:31 -> rows 2, 3 */                                \
    return FROM_I(__builtin_amdgcn_readlane(TO_I(v), 63));                                                        \
  }
__device__ __forceinline__ unsigned umax_u32(unsigned a, unsigned b) { return a > b ? a : b; }
MSFL_WAVE_REDUCE(wave_min_f32, float, fminf, __float_as_int, __int_as_float)
MSFL_WAVE_REDUCE(wave_max_f32, float, fmaxf, __float_as_int, __int_as_float)
MSFL_WAVE_REDUCE(wave_max_u32, unsigned, umax_u32, (int), (unsigned))
#undef MSFL_WAVE_REDUCE

__device__ __forceinline__ int spread4(int v) {          // abcd -> a00b00c00d
  v = (v | (v << 4)) & 0x0c3;
  return (v | (v << 2)) & 0x249;
}

// running best six of a lane: 32-bit keys, ascending; five medians and a minimum per candidate
struct Best6 { unsigned k0, k1, k2, k3, k4, k5; };
__device__ __forceinline__ void best6_insert(Best6& b, unsigned x) {
  const unsigned n5 = med3_u32(b.k4, b.k5, x), n4 = med3_u32(b.k3, b.k4, x), n3 = med3_u32(b.k2, b.k3, x);
  const unsigned n2 = med3_u32(b.k1, b.k2, x), n1 = med3_u32(b.k0, b.k1, x);
  b.k0 = min(b.k0, x); b.k1 = n1; b.k2 = n2; b.k3 = n3; b.k4 = n4; b.k5 = n5;
}

#if MSFL_BIN_EXP == 2
__device__ unsigned long long g_bin_dbg[8];   // batches, candidates, valid lanes, hard lanes, tied lanes, overflowed batches
#endif
// One workgroup per slice (<= kBinSlice queries of one tile).  out6[6 i ..] = {feature index, five positions in the
// sorted map, nearest first} for binned item i; position[0] = -1 when the 5th distance fails the acceptance gate.
//
// Per wavefront and 64 sorted queries: box of the queries in cell coordinates (six DPP reductions) -> the rows (y, z)
// and the x cell range that can hold a point within `radius` of ANY of the 64 queries; the lanes fetch the row ranges
// in parallel (one row per lane), the candidates of all rows are STAGED INTO AN LDS TILE with coalesced 16-byte loads,
// and the distance / best-six loop then reads one candidate per step as an LDS broadcast.
// A lane is SETTLED when its five best lie inside `radius` (nothing outside the staged region can beat them) and its
// six best keys are distinct in the kept distance bits.  Every other lane (further neighbours needed, fewer than five
// found, tied keys, tile overflow) is a HARD query: the workgroup collects them and serves them afterwards in dense
// wavefronts with the exact per-lane search (knn5_grid), so a few hard lanes do not make 64 lanes walk a wider region.
__global__ void __launch_bounds__(kBinBlock)
knn5_binned_kernel(const float4* __restrict__ items, const int* __restrict__ tile_start, const int* __restrict__ work_start,
                   MapView mc, MapView ms, float max_sq_dist, float radius, int* __restrict__ out6) {
  constexpr int kCap = 1 << kBinOrdBits;              // candidates per wavefront tile
  constexpr int kWaves = kBinBlock / 64;
  __shared__ unsigned short s_key[kBinSlice];
  __shared__ unsigned short s_perm[kBinSlice];        // sorted slot -> item of the slice
  __shared__ unsigned short s_hard[kBinSlice];        // sorted slots of the hard queries
  __shared__ float4 s_tile[kWaves * kCap];            // counting-sort histogram first (4096 ints), candidate tiles afterwards
  __shared__ int s_wsum[kWaves];
  __shared__ int s_nhard;
  static_assert(kWaves * kCap * sizeof(float4) >= 4096 * sizeof(int), "the candidate tiles reuse the histogram's memory");
  static_assert(4096 % kBinBlock == 0 && kBinSlice <= 65536, "counting sort layout");
  int* s_hist = reinterpret_cast<int*>(s_tile);
  const int w = blockIdx.x;
  if (w >= work_start[kBinTiles]) return;
  int lo = 0, hi = kBinTiles;                       // largest t with work_start[t] <= w (tiles without work share their successor's start)
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (work_start[mid] <= w) lo = mid; else hi = mid;
  }
  const int t = lo;
  const bool edge_tile = t < kBinTilesCorner;
  const MapView mv = edge_tile ? mc : ms;
  const GridDesc g = *mv.g;
  const int tl = edge_tile ? t : t - kBinTilesCorner;
  const int tx = tl % g.ntx, ty = (tl / g.ntx) % g.nty, tz = tl / (g.ntx * g.nty);
  const int i0 = tile_start[t] + (w - work_start[t]) * kBinSlice;
  const int n = min(kBinSlice, tile_start[t + 1] - i0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // ---- counting sort of the slice by 12-bit Morton fine cell (16 fine cells per tile edge) ----
  for (int k = tid; k < 4096; k += kBinBlock) s_hist[k] = 0;
  if (tid == 0) s_nhard = 0;
  __syncthreads();
  {
    const float fsx = 16.0f / (float)(g.tk * kGridXSub), fsyz = 16.0f / (float)g.tk;
    const float bx = (float)(tx * g.tk * kGridXSub), by = (float)(ty * g.tk), bz = (float)(tz * g.tk);
    for (int j = tid; j < n; j += kBinBlock) {
      const float4 it = items[i0 + j];
      const float ux = (it.x - g.ox) * g.inv_cell_x, uy = (it.y - g.oy) * g.inv_cell, uz = (it.z - g.oz) * g.inv_cell;
      const int fx = min(max((int)((ux - bx) * fsx), 0), 15), fy = min(max((int)((uy - by) * fsyz), 0), 15),
                fz = min(max((int)((uz - bz) * fsyz), 0), 15);
      const int key = spread4(fx) | (spread4(fy) << 1) | (spread4(fz) << 2);
      s_key[j] = (unsigned short)key;
      atomicAdd(&s_hist[key], 1);
    }
  }
  __syncthreads();
  {
    constexpr int kPer = 4096 / kBinBlock;
    int v[kPer], sum = 0;
#pragma unroll
    for (int k = 0; k < kPer; k++) { v[k] = s_hist[kPer * tid + k]; sum += v[k]; }
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o); if (lane >= o) incl += u; }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int k = 0; k < wave; k++) run += s_wsum[k];
#pragma unroll
    for (int k = 0; k < kPer; k++) { s_hist[kPer * tid + k] = run; run += v[k]; }
  }
  __syncthreads();
  for (int j = tid; j < n; j += kBinBlock) s_perm[atomicAdd(&s_hist[s_key[j]], 1)] = (unsigned short)j;
  __syncthreads();                                   // the histogram is dead from here on: its memory becomes the candidate tiles

  if (MSFL_BIN_EXP == 6) return;
  // ---- wave-uniform search: 64 consecutive sorted queries per wavefront ----
  const unsigned gate_bits = __float_as_uint(max_sq_dist);
  const float slack = 2e-3f;                                    // cell units: covers the f32 rounding of cell coordinates
  const float rho = fminf(radius, sqrtf(max_sq_dist) * 1.0001f);
  const float settle = fminf(rho * rho * 0.99998f, max_sq_dist); // 5th-best distance (upper bound) at or below this: settled
  const float rx = rho * g.inv_cell_x + slack, ryz = rho * g.inv_cell + slack;
  const int* __restrict__ cs = mv.cell_start;
  const float4* __restrict__ sorted = mv.sorted;
  float4* tile = s_tile + wave * kCap;
  for (int base = wave * 64; base < n; base += kBinBlock) {
    const int sidx = base + lane;
    const bool valid = sidx < n;
    const float4 it = items[i0 + (valid ? (int)s_perm[sidx] : 0)];
    const float3 q = make_float3(it.x, it.y, it.z);
    const float ux = (q.x - g.ox) * g.inv_cell_x, uy = (q.y - g.oy) * g.inv_cell, uz = (q.z - g.oz) * g.inv_cell;
    // a query more than one gate radius outside the grid cannot have a neighbour inside the gate (cell edge >= radius):
    // it stays out of the wave's box (also non-finite coordinates), keeps its sentinels and is rejected below
    const bool in_reach = valid && ux > -(float)kGridXSub - 0.5f && ux < (float)(g.dx + kGridXSub) + 0.5f &&
                          uy > -1.5f && uy < (float)g.dy + 1.5f && uz > -1.5f && uz < (float)g.dz + 1.5f;
    const float bx0 = wave_min_f32(in_reach ? ux : INFINITY), bx1 = wave_max_f32(in_reach ? ux : -INFINITY);
    const float by0 = wave_min_f32(in_reach ? uy : INFINITY), by1 = wave_max_f32(in_reach ? uy : -INFINITY);
    const float bz0 = wave_min_f32(in_reach ? uz : INFINITY), bz1 = wave_max_f32(in_reach ? uz : -INFINITY);
    Best6 best; best.k0 = best.k1 = best.k2 = best.k3 = best.k4 = best.k5 = 0xffffffffu;
    const msfl_f2 qxy = {q.x, q.y};
    int count = 0;                                                // candidates staged (wave-uniform)
    bool overflow = false;
    if (bx0 <= bx1) {
      const int x0 = __builtin_amdgcn_readfirstlane(max((int)floorf(bx0 - rx), 0));
      const int x1 = __builtin_amdgcn_readfirstlane(min((int)floorf(bx1 + rx), g.dx - 1));
      const int y0 = __builtin_amdgcn_readfirstlane(max((int)floorf(by0 - ryz), 0));
      const int y1 = __builtin_amdgcn_readfirstlane(min((int)floorf(by1 + ryz), g.dy - 1));
      const int z0 = __builtin_amdgcn_readfirstlane(max((int)floorf(bz0 - ryz), 0));
      const int z1 = __builtin_amdgcn_readfirstlane(min((int)floorf(bz1 + ryz), g.dz - 1));
      const int ny = y1 - y0 + 1, nrows = (x0 <= x1 && ny > 0 && z1 >= z0) ? ny * (z1 - z0 + 1) : 0;
      for (int rb = 0; rb < nrows && !overflow; rb += 64) {
        const int r = rb + lane;                                  // one row per lane: its candidate range
        int sA = 0, n1 = 0;
        if (r < nrows) {
          const int z = z0 + r / ny, y = y0 + r % ny;
          const int row = (z * g.dy + y) * g.dx;
          sA = cs[row + x0];
          n1 = cs[row + x1 + 1] - sA;
        }
        // stage the non-empty ranges: loop over the lanes that hold one (scalar), 64 candidates per load
        unsigned long long todo = __ballot(n1 > 0);
        while (todo && MSFL_BIN_EXP != 4) {
          const int l = __ffsll((long long)todo) - 1;
          todo &= todo - 1;
          const int ps = __builtin_amdgcn_readlane(sA, l), pn = __builtin_amdgcn_readlane(n1, l);
          if (count + pn > kCap) { overflow = true; break; }
          for (int o = lane; o < pn; o += 64) {
            float4 m = sorted[ps + o];
            m.w = __int_as_float(ps + o);                         // the tile keeps the POSITION in the sorted map: decoded at the end
            tile[count + o] = m;
          }
          count += pn;
        }
      }
      if (!overflow && MSFL_BIN_EXP != 3 && MSFL_BIN_EXP != 4) {
        // distance + best-six over the staged candidates: one LDS broadcast read per candidate
        int c = 0;
        for (; c + 3 < count; c += 4) {
          const float4 m0 = tile[c], m1 = tile[c + 1], m2 = tile[c + 2], m3 = tile[c + 3];
          best6_insert(best, (__float_as_uint(l2_simple_pk(m0, qxy, q.z)) & ~kBinOrdMask) | (unsigned)c);
          best6_insert(best, (__float_as_uint(l2_simple_pk(m1, qxy, q.z)) & ~kBinOrdMask) | (unsigned)(c + 1));
          best6_insert(best, (__float_as_uint(l2_simple_pk(m2, qxy, q.z)) & ~kBinOrdMask) | (unsigned)(c + 2));
          best6_insert(best, (__float_as_uint(l2_simple_pk(m3, qxy, q.z)) & ~kBinOrdMask) | (unsigned)(c + 3));
        }
        for (; c < count; c++) {
          const float4 m0 = tile[c];
          best6_insert(best, (__float_as_uint(l2_simple_pk(m0, qxy, q.z)) & ~kBinOrdMask) | (unsigned)c);
        }
      }
    }
    // ---- per lane: settled -> decode and store; hard -> the workgroup's list ----
    const unsigned t0 = best.k0 >> kBinOrdBits, t1 = best.k1 >> kBinOrdBits, t2 = best.k2 >> kBinOrdBits,
                   t3 = best.k3 >> kBinOrdBits, t4 = best.k4 >> kBinOrdBits, t5 = best.k5 >> kBinOrdBits;
    const bool tied = t0 == t1 || t1 == t2 || t2 == t3 || t3 == t4 || t4 == t5;      // two sentinels count as tied: fewer than six found
    // upper bound of the 5th-best distance (ordinal bits set); a settled lane is accepted: settle <= the gate
    const bool settled = in_reach && !overflow && !tied && __uint_as_float(best.k4 | kBinOrdMask) <= settle &&
                         (best.k4 | kBinOrdMask) < gate_bits;
#if MSFL_BIN_EXP == 2
    {
      const unsigned long long vm = __ballot(valid), hm = __ballot(valid && in_reach && !settled), tm = __ballot(valid && in_reach && tied && best.k5 != 0xffffffffu);
      if (lane == 0) { atomicAdd(&g_bin_dbg[0], 1ull); atomicAdd(&g_bin_dbg[1], (unsigned long long)count); atomicAdd(&g_bin_dbg[2], (unsigned long long)__popcll(vm));
                       atomicAdd(&g_bin_dbg[3], (unsigned long long)__popcll(hm)); atomicAdd(&g_bin_dbg[4], (unsigned long long)__popcll(tm));
                       atomicAdd(&g_bin_dbg[5], overflow ? 1ull : 0ull); }
    }
#endif
    if (valid) {
      int* out = out6 + 6 * (size_t)(i0 + sidx);
      if (settled) {
        out[0] = __float_as_int(it.w);
        out[1] = __float_as_int(tile[best.k0 & kBinOrdMask].w); out[2] = __float_as_int(tile[best.k1 & kBinOrdMask].w);
        out[3] = __float_as_int(tile[best.k2 & kBinOrdMask].w); out[4] = __float_as_int(tile[best.k3 & kBinOrdMask].w);
        out[5] = __float_as_int(tile[best.k4 & kBinOrdMask].w);
      } else if (!in_reach) {
        out[0] = __float_as_int(it.w); out[1] = -1;               // nothing inside the gate
      } else {
        s_hard[atomicAdd(&s_nhard, 1)] = (unsigned short)sidx;
      }
    }
  }
  __syncthreads();
  // ---- hard queries, dense: exact per-lane search (the arithmetic of knn5_scan2map_kernel) ----
  const int nh = (MSFL_BIN_EXP == 5 || MSFL_BIN_EXP == 3 || MSFL_BIN_EXP == 4) ? 0 : s_nhard;
  for (int k = tid; k < nh; k += kBinBlock) {
    const int sidx = s_hard[k];
    const float4 it = items[i0 + (int)s_perm[sidx]];
    Top5 tt; int n_cand = 0;
    knn5_grid(g, sorted, cs, make_float3(it.x, it.y, it.z), max_sq_dist, tt, n_cand);
    int* out = out6 + 6 * (size_t)(i0 + sidx);
    out[0] = __float_as_int(it.w);
    if ((unsigned int)tt.k4 != 0xffffffffu && (double)top5_d4(tt) < (double)max_sq_dist) {      // :128 / :198
      const int* po = mv.pos_of;
      out[1] = po[(unsigned int)tt.k0]; out[2] = po[(unsigned int)tt.k1]; out[3] = po[(unsigned int)tt.k2];
      out[4] = po[(unsigned int)tt.k3]; out[5] = po[(unsigned int)tt.k4];
    } else {
      out[1] = -1;
    }
  }
}

// K4b in binned order: out6 -> line / plane fit -> record in the feature's slot (fit_scan2map_kernel otherwise)
__global__ void __launch_bounds__(kAssocBlock, MSFL_FIT_WAVES)
fit_binned_kernel(BatchView bv, const int* __restrict__ tile_start, const int* __restrict__ out6,
                  const float4* __restrict__ map_c, const float4* __restrict__ map_s, double line_ratio, double plane_tol,
                  double* __restrict__ rec) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= tile_start[kBinTiles]) return;
  const bool is_edge = i < tile_start[kBinTilesCorner];
  const int* in = out6 + 6 * (size_t)i;
  const int fi = in[0], p0 = in[1];
  FitOut fo; fo.ok = false; fo.C = mk3(0, 0, 0); fo.N = mk3(0, 0, 0);
  if (p0 >= 0) {
    const float4* mp = is_edge ? map_c : map_s;
    const float4 nb[5] = {mp[p0], mp[in[2]], mp[in[3]], mp[in[4]], mp[in[5]]};
    fo = is_edge ? edge_fit(nb, line_ratio) : plane_fit(nb, plane_tol);
  }
  if (is_edge) {
    double* out = rec + edge_rec_off(bv, fi);
    out[0] = fo.C.x; out[1] = fo.C.y; out[2] = fo.C.z;
    out[3] = fo.N.x; out[4] = fo.N.y; out[5] = fo.N.z;
  } else {
    double* out = rec + plane_rec_off(bv, fi);
    out[0] = fo.N.x; out[1] = fo.N.y; out[2] = fo.N.z; out[3] = dot(fo.N, fo.C);
  }
}

}  // namespace msfl

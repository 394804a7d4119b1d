// msfl_odom.cuh — stage B: scan-to-scan data association on gfx950.
//
// Replaces the two association loops of OdometryScanMatcher::MatchScan2Scan
// (odometry_scan_matcher.cc:81-163 edges, :166-258 planes).  The solve is the same persistent
// LM kernel as stage C (lm_solve_kernel) with the `< 10 correspondences` gate (:262-267).
//
// One 256-thread workgroup per (scan pair, tile of 256 query features).  The target cloud of the
// previous scan (<= 1 920 less-sharp / ~20-26 k less-flat points) is streamed through LDS in
// 1024-point tiles that all queries of the workgroup share (ds_read broadcast), twice:
//   pass 1  exact 1-NN (replaces the kd-tree query :84 / :169; f32 L2_Simple distances, ties -> lower index)
//   pass 2  the two ring-window scans, restated as one ascending sweep with the reference's
//           visit-order semantics (forward scan first, strict '<' updates, break at the first
//           out-of-window ring in each direction).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "msfl_kernels.cuh"

namespace msfl {

// Record arithmetic shared by the three association kernels (one-wavefront, tiled brute-force, column grid), which must
// agree bit for bit: unit vectors by v * rsqrt(v.v) (v_rsq_f64 + two Newton steps; the zero vector of two coincident
// points stays zero) and the centroid of the three plane points by a multiplication.  The IEEE square root, the three
// divisions by it and the three divisions by 3 were a tenth of the column-grid kernel's instructions per query.
__device__ __forceinline__ d3 odom_unit(d3 v) { return normalized_rsq(v); }
__device__ __forceinline__ d3 odom_centroid3(d3 a, d3 b, d3 c) {
  const double third = 1.0 / 3.0;
  return mk3((a.x + b.x + c.x) * third, (a.y + b.y + c.y) * third, (a.z + b.z + c.z) * third);
}

struct OdomView {
  // previous scan (targets), concatenated over the batch
  const float4* last_ls; const uint16_t* last_ls_ring; const int* last_ls_off;
  const float4* last_lf; const uint16_t* last_lf_ring; const int* last_lf_off;
  double dist_sq_threshold;   // kDistanceSqThreshold = 25
  double nearby_scan;         // kNearByScan = 2.5
};

constexpr int kOdomTile = 1024;

__device__ __forceinline__ float odom_dist(float4 a, float3 q) {
  // (ax-qx)*(ax-qx) + (ay-qy)*(ay-qy) + (az-qz)*(az-qz) in f32 (:102-108); same order as L2_Simple
  // x and y as one packed-f32 pair (two individually rounded IEEE operations per instruction): same bits
  const msfl_f2 axy = {a.x, a.y}, qxy = {q.x, q.y};
  const msfl_f2 dxy = axy - qxy;
  const msfl_f2 sxy = dxy * dxy;
  const float dz = a.z - q.z;
  return sxy.x + sxy.y + dz * dz;
}

// bv.corner = curr sharp, bv.surf = curr flat; records in feature order (sharp first).
__global__ void __launch_bounds__(256)
assoc_scan2scan_kernel(BatchView bv, OdomView ov, const double* __restrict__ poses, const int* __restrict__ status,
                       double* __restrict__ rec, const int* __restrict__ plane_mode, const int* __restrict__ edge_mode) {
  __shared__ float4 s_tile[kOdomTile];
  const int b = blockIdx.y;
  const int n_sharp = bv.corner_off[b + 1] - bv.corner_off[b];
  const int n_flat = bv.surf_off[b + 1] - bv.surf_off[b];
  const int nq = n_sharp + n_flat;
  const int q0 = blockIdx.x * 256;
  if (q0 >= nq) return;
  const int qi = q0 + threadIdx.x;
  const bool has_q = qi < nq;
  // compact records: sharp (edge) features 6 doubles {C,N}, flat (plane) features 4 doubles {N, N.C}
  double* out = rec + (qi < n_sharp ? edge_rec_off(bv, bv.corner_off[b] + qi) : plane_rec_off(bv, bv.surf_off[b] + (qi - n_sharp)));
  const int out_len = qi < n_sharp ? 6 : 4;
  if (status[b] != 0) {
    if (has_q) for (int k = 0; k < out_len; k++) out[k] = 0.0;
    return;
  }
  // plane_mode[b] == 0: the column-grid kernel below owns this pair's plane queries
  const bool planes_here = plane_mode == nullptr || plane_mode[b] != 0;
  const bool edges_here = edge_mode == nullptr || edge_mode[b] != 0;       // likewise for the edge queries
  const bool is_edge = has_q && qi < n_sharp && edges_here;
  const bool is_plane = has_q && qi >= n_sharp && planes_here;
  float4 f = make_float4(0, 0, 0, 0);
  float3 q = make_float3(0, 0, 0);
  if (is_edge || is_plane) {               // queries owned by the grid kernels are not touched here
    f = is_edge ? bv.corner[bv.corner_off[b] + qi] : bv.surf[bv.surf_off[b] + (qi - n_sharp)];
    const pose7 T = load_pose(poses + 7 * b);
    q = transform_point_f32(T, f.x, f.y, f.z);        // TransformToStart with s = 1 (:21-33)
  }
  const float thr = (float)ov.dist_sq_threshold;
  // which target clouds does this workgroup need? (uniform)
  const bool wg_has_edge = q0 < n_sharp && edges_here;
  const bool wg_has_plane = (q0 + 256 > n_sharp) && (nq > n_sharp) && planes_here;
  int closest = -1, min2 = -1, min3 = -1;

  for (int which = 0; which < 2; which++) {
    const bool edge_pass = (which == 0);
    if (edge_pass ? !wg_has_edge : !wg_has_plane) continue;
    const float4* tp = edge_pass ? ov.last_ls + ov.last_ls_off[b] : ov.last_lf + ov.last_lf_off[b];
    const uint16_t* tr = edge_pass ? ov.last_ls_ring + ov.last_ls_off[b] : ov.last_lf_ring + ov.last_lf_off[b];
    const int nt = edge_pass ? ov.last_ls_off[b + 1] - ov.last_ls_off[b] : ov.last_lf_off[b + 1] - ov.last_lf_off[b];
    const bool mine = edge_pass ? is_edge : is_plane;
    // ---- pass 1: exact 1-NN ----
    float best = INFINITY; int besti = -1;
    for (int t0 = 0; t0 < nt; t0 += kOdomTile) {
      __syncthreads();
      for (int k = threadIdx.x; k < kOdomTile; k += 256) {
        const int j = t0 + k;
        if (j < nt) { float4 p = tp[j]; p.w = __int_as_float((int)tr[j]); s_tile[k] = p; }
      }
      __syncthreads();
      if (mine) {
        const int m = min(kOdomTile, nt - t0);
        for (int k = 0; k < m; k++) {
          const float d = odom_dist(s_tile[k], q);
          if (d < best) { best = d; besti = t0 + k; }
        }
      }
    }
    // ---- pass 2: ring-window sweep ----
    int id = 0;
    bool go = mine && besti >= 0 && best < thr;            // :87 / :173
    if (go) { closest = besti; id = tr[besti]; }
    const float hi_ring = (float)id + (float)ov.nearby_scan, lo_ring = (float)id - (float)ov.nearby_scan;
    float b2F = thr, b2B = thr, b3F = thr, b3B = thr;
    int i2F = -1, i2B = -1, i3F = -1, i3B = -1;
    bool fwd_done = false;
    for (int t0 = 0; t0 < nt; t0 += kOdomTile) {
      __syncthreads();
      for (int k = threadIdx.x; k < kOdomTile; k += 256) {
        const int j = t0 + k;
        if (j < nt) { float4 p = tp[j]; p.w = __int_as_float((int)tr[j]); s_tile[k] = p; }
      }
      __syncthreads();
      if (go) {
        const int m = min(kOdomTile, nt - t0);
        for (int k = 0; k < m; k++) {
          const int j = t0 + k;
          if (j == closest) continue;
          const float4 p = s_tile[k];
          const int rj = __float_as_int(p.w);
          if (j > closest) {
            // forward scan (:93-115 / :183-207)
            if (fwd_done) continue;
            if (edge_pass) {
              if (rj <= id) continue;
              if ((float)rj > hi_ring) { fwd_done = true; continue; }
              const float d = odom_dist(p, q);
              if (d < b2F) { b2F = d; i2F = j; }
            } else {
              if ((float)rj > hi_ring) { fwd_done = true; continue; }
              const float d = odom_dist(p, q);
              if (rj <= id) { if (d < b2F) { b2F = d; i2F = j; } }
              else { if (d < b3F) { b3F = d; i3F = j; } }
            }
          } else {
            // backward scan (:118-140 / :210-232), visited here in ascending order: a break point
            // invalidates everything below it, and among equal distances the later (larger) index
            // is the one the descending scan meets first
            if (edge_pass) {
              if (rj >= id) continue;
              if ((float)rj < lo_ring) { b2B = thr; i2B = -1; continue; }
              const float d = odom_dist(p, q);
              if (d < b2B || (d == b2B && i2B >= 0)) { b2B = d; i2B = j; }
            } else {
              if ((float)rj < lo_ring) { b2B = thr; i2B = -1; b3B = thr; i3B = -1; continue; }
              const float d = odom_dist(p, q);
              if (rj >= id) { if (d < b2B || (d == b2B && i2B >= 0)) { b2B = d; i2B = j; } }
              else { if (d < b3B || (d == b3B && i3B >= 0)) { b3B = d; i3B = j; } }
            }
          }
        }
      }
    }
    if (go) {
      // the backward scan continues the forward scan's running minimum with strict '<'
      min2 = (i2B >= 0 && b2B < b2F) ? i2B : i2F;
      if (!edge_pass) min3 = (i3B >= 0 && b3B < b3F) ? i3B : i3F;
      d3 C = mk3(0, 0, 0), N = mk3(0, 0, 0);
      if (edge_pass) {
        if (min2 >= 0) {                                                   // :143-162
          const float4 a = tp[closest], c = tp[min2];
          const d3 A = mk3((double)a.x, (double)a.y, (double)a.z), Bp = mk3((double)c.x, (double)c.y, (double)c.z);
          N = odom_unit(A - Bp);
          C = A;
        }
      } else {
        if (min2 >= 0 && min3 >= 0) {                                      // :234-256, lidar_factor.h:70-78
          const float4 a = tp[closest], c = tp[min2], e = tp[min3];
          const d3 A = mk3((double)a.x, (double)a.y, (double)a.z), Bp = mk3((double)c.x, (double)c.y, (double)c.z),
                   Cp = mk3((double)e.x, (double)e.y, (double)e.z);
          N = odom_unit(cross(A - Bp, A - Cp));
          C = odom_centroid3(A, Bp, Cp);
        }
      }
      if (edge_pass) { out[0] = C.x; out[1] = C.y; out[2] = C.z; out[3] = N.x; out[4] = N.y; out[5] = N.z; }
      else { out[0] = N.x; out[1] = N.y; out[2] = N.z; out[3] = dot(N, C); }
    } else if (mine) {
      for (int k = 0; k < out_len; k++) out[k] = 0.0;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Latency variant for small batches (the reference's own use: ONE scan pair per call).  With a few
// hundred queries the tiled kernel above has only 1-3 workgroups per pair and takes ~13 ms; here one
// WAVEFRONT owns one query and its 64 lanes sweep the previous scan's cloud cooperatively
// (coalesced 16-byte loads, wave reductions), ~0.1 ms per pair.  Same semantics, three sweeps:
// exact 1-NN; the two `break` boundaries; the windowed minima with the reference's tie order.
// ---------------------------------------------------------------------------------------------
struct DJ { float d; int j; };
// lexicographic min over (d, j): smallest distance, ties -> smallest j
__device__ __forceinline__ DJ wave_min_lo(DJ v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float d = __shfl_xor(v.d, o); const int j = __shfl_xor(v.j, o);
    if (d < v.d || (d == v.d && j < v.j)) { v.d = d; v.j = j; }
  }
  return v;
}
// smallest distance, ties -> LARGEST j (the descending backward scan meets it first); j = -1 means none
__device__ __forceinline__ DJ wave_min_hi(DJ v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float d = __shfl_xor(v.d, o); const int j = __shfl_xor(v.j, o);
    if (j >= 0 && (v.j < 0 || d < v.d || (d == v.d && j > v.j))) { v.d = d; v.j = j; }
  }
  return v;
}

__global__ void __launch_bounds__(256)
assoc_scan2scan_wave_kernel(BatchView bv, OdomView ov, const double* __restrict__ poses, const int* __restrict__ status,
                            double* __restrict__ rec) {
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int n_sharp = bv.corner_off[b + 1] - bv.corner_off[b];
  const int n_flat = bv.surf_off[b + 1] - bv.surf_off[b];
  const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (qi >= n_sharp + n_flat) return;
  const bool edge = qi < n_sharp;
  double* out = rec + (edge ? edge_rec_off(bv, bv.corner_off[b] + qi) : plane_rec_off(bv, bv.surf_off[b] + (qi - n_sharp)));
  const int out_len = edge ? 6 : 4;
  const float4* tp = edge ? ov.last_ls + ov.last_ls_off[b] : ov.last_lf + ov.last_lf_off[b];
  const uint16_t* tr = edge ? ov.last_ls_ring + ov.last_ls_off[b] : ov.last_lf_ring + ov.last_lf_off[b];
  const int nt = edge ? ov.last_ls_off[b + 1] - ov.last_ls_off[b] : ov.last_lf_off[b + 1] - ov.last_lf_off[b];
  const float thr = (float)ov.dist_sq_threshold;
  bool ok = status[b] == 0 && nt > 0;
  float3 q = make_float3(0, 0, 0);
  int closest = -1;
  if (ok) {
    const float4 f = edge ? bv.corner[bv.corner_off[b] + qi] : bv.surf[bv.surf_off[b] + (qi - n_sharp)];
    q = transform_point_f32(load_pose(poses + 7 * b), f.x, f.y, f.z);
    DJ best{INFINITY, 0x7fffffff};
    for (int j = lane; j < nt; j += 64) {                     // sweep 1: exact 1-NN (:84 / :169)
      const float d = odom_dist(tp[j], q);
      if (d < best.d) { best.d = d; best.j = j; }
    }
    best = wave_min_lo(best);
    ok = best.j != 0x7fffffff && best.d < thr;               // :87 / :173
    closest = best.j;
  }
  if (!ok) {
    if (lane < out_len) out[lane] = 0.0;
    return;
  }
  const int id = tr[closest];
  const float hi_ring = (float)id + (float)ov.nearby_scan, lo_ring = (float)id - (float)ov.nearby_scan;
  int fwd_end = nt, back_end = -1;                            // sweep 2: first out-of-window ring on each side
  for (int j = lane; j < nt; j += 64) {
    const float rj = (float)tr[j];
    if (j > closest && rj > hi_ring) fwd_end = min(fwd_end, j);
    if (j < closest && rj < lo_ring) back_end = max(back_end, j);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { fwd_end = min(fwd_end, __shfl_xor(fwd_end, o)); back_end = max(back_end, __shfl_xor(back_end, o)); }
  DJ f2{thr, 0x7fffffff}, f3{thr, 0x7fffffff}, b2{thr, -1}, b3{thr, -1};
  for (int j = lane; j < nt; j += 64) {                       // sweep 3: windowed minima
    if (j == closest || j >= fwd_end || j <= back_end) continue;
    const int rj = tr[j];
    if (j > closest) {
      if (edge) { if (rj <= id) continue; const float d = odom_dist(tp[j], q); if (d < f2.d) { f2.d = d; f2.j = j; } }
      else {
        const float d = odom_dist(tp[j], q);
        if (rj <= id) { if (d < f2.d) { f2.d = d; f2.j = j; } } else { if (d < f3.d) { f3.d = d; f3.j = j; } }
      }
    } else {
      if (edge) { if (rj >= id) continue; const float d = odom_dist(tp[j], q); if (d < b2.d || (d == b2.d && b2.j >= 0)) { b2.d = d; b2.j = j; } }
      else {
        const float d = odom_dist(tp[j], q);
        if (rj >= id) { if (d < b2.d || (d == b2.d && b2.j >= 0)) { b2.d = d; b2.j = j; } }
        else { if (d < b3.d || (d == b3.d && b3.j >= 0)) { b3.d = d; b3.j = j; } }
      }
    }
  }
  f2 = wave_min_lo(f2); f3 = wave_min_lo(f3); b2 = wave_min_hi(b2); b3 = wave_min_hi(b3);
  if (f2.j == 0x7fffffff) f2.j = -1;
  if (f3.j == 0x7fffffff) f3.j = -1;
  // the backward scan continues the forward scan's running minimum with strict '<'
  const int min2 = (b2.j >= 0 && b2.d < f2.d) ? b2.j : f2.j;
  const int min3 = (b3.j >= 0 && b3.d < f3.d) ? b3.j : f3.j;
  if (lane != 0) return;
  d3 C = mk3(0, 0, 0), N = mk3(0, 0, 0);
  if (edge) {
    if (min2 >= 0) {
      const float4 a = tp[closest], c = tp[min2];
      const d3 A = mk3((double)a.x, (double)a.y, (double)a.z), Bp = mk3((double)c.x, (double)c.y, (double)c.z);
      N = odom_unit(A - Bp); C = A;
    }
    out[0] = C.x; out[1] = C.y; out[2] = C.z; out[3] = N.x; out[4] = N.y; out[5] = N.z;
  } else {
    if (min2 >= 0 && min3 >= 0) {
      const float4 a = tp[closest], c = tp[min2], e = tp[min3];
      const d3 A = mk3((double)a.x, (double)a.y, (double)a.z), Bp = mk3((double)c.x, (double)c.y, (double)c.z),
               Cp = mk3((double)e.x, (double)e.y, (double)e.z);
      N = odom_unit(cross(A - Bp, A - Cp));
      C = odom_centroid3(A, Bp, Cp);
    }
    out[0] = N.x; out[1] = N.y; out[2] = N.z; out[3] = dot(N, C);
  }
}

// ---------------------------------------------------------------------------------------------
// Throughput path for the plane queries (384 flat features x ~20 k less-flat targets per pair is
// >95 % of the brute-force work).  Each previous scan's less-flat cloud is bucketed into 1 m x-y
// columns by ONE workgroup with an LDS counting sort (u16 counters, <= 40 960 columns = 80 KB):
// bounding box -> histogram -> exclusive scan -> scatter, three streaming passes over ~20 k
// points.  The column starts go to a dense per-pair table, so a row of x-adjacent columns is one
// contiguous run bounded by two table entries.  A query walks the 3x3, 5x5 and 13x13 column
// neighbourhoods until the exact lower bound of everything outside the walked square exceeds
// what it has found.
//
// The ring-window scans (:183-232) are index-ordered sweeps with `break`; on a cloud whose rings
// are non-decreasing in array order (what scan registration produces, msf_loam_node.cc:243-356)
// they select exactly {j > closest, ring <= id + 2.5} and {j < closest, ring >= id - 2.5}, and the
// running strict '<' minima become lexicographic minima over (distance, index).  Pairs whose
// cloud is not ring-monotone, has a ring >= 256, 2^24 points or more (the sorted copy's index field), more than 65 535
// points in ONE column (the u16 counters; found after the histogram: a wrapped counter makes the column total fall short
// of n), a non-finite or > 512 m coordinate, or a footprint of more than 40 960 columns keep the brute-force kernel
// above (mode[b] = 1); results are identical either way.  (Until round 5 the limit was 65 535 points per CLOUD, which
// sent every 64-beam less-flat list -- ~100 k points -- to the brute-force kernel: 27 ms per association.)
// ---------------------------------------------------------------------------------------------
constexpr int kOdomRange = 512;                         // columns are addressed relative to the cloud's bounding box
constexpr int kOdomMaxCells = 40960;                    // u16 counters: 80 KB of LDS
constexpr int kOdomTabStride = kOdomMaxCells + 1;       // per-pair table of column starts (+ end sentinel)
#ifndef MSFL_ODOM_LANES
#define MSFL_ODOM_LANES 16
#endif
constexpr int kOdomLanes = MSFL_ODOM_LANES;             // lanes cooperating on one plane query
#ifndef MSFL_ODOM_BLOCK
#define MSFL_ODOM_BLOCK 64
#endif
constexpr int kOdomBlock = MSFL_ODOM_BLOCK;             // threads per workgroup of the column-grid query kernel
// Half-widths of the walked column squares: 3x3, 5x5, 7x7, 9x9, 13x13.  Every level re-walks the inner square, so a fine
// progression only pays when few queries go far: measured 1.91 -> 1.84 ms per 1 024 pairs against the round-1 schedule
// (1, 2, 6).  Columns of 0.5 m with (1, 2, 3, 5, 8, 12) visit 24 % fewer candidates and run 35 % SLOWER (2.37 ms): the
// kernel is bound by the dependent run-bound -> run loads of each row, not by the candidates (rejected).
constexpr int kOdomLevels = 5;
__device__ __forceinline__ int odom_level_radius(int l) { return l < 4 ? l + 1 : 6; }
constexpr int kOdomMaxLevel = 6;

struct OdomPairDesc { int ox, oy, W, H; };              // column (cx, cy) = (floor(x) - ox, floor(y) - oy), 0 <= cx < W

struct OdomIndex {
  const unsigned* tab;       // B x kOdomTabStride column starts, relative to the pair's segment
  const float4* sorted;      // targets in column order: xyz + (ring << 24 | index within the pair's cloud)
  const OdomPairDesc* desc;
  const int* mode;           // per pair: 0 = column grid, 1 = brute force
  int tiles;                 // workgroups per pair in the query kernel
};

#ifndef MSFL_ODOM_BIN_KEEP
#define MSFL_ODOM_BIN_KEEP 20
#endif
constexpr int kOdomBinKeep = MSFL_ODOM_BIN_KEEP;        // points per thread the binning kernel keeps in registers (x 1 024 threads)

struct OdomBinJob {                                      // one cloud family of the batch (less-flat / less-sharp) and where its index goes
  const float4* pts; const uint16_t* ring; const int* off;
  unsigned* tab; float4* sorted; OdomPairDesc* desc; int* mode;
};

// blocks [0, n_pairs) bin job a's clouds, blocks [n_pairs, 2 n_pairs) job b's (one launch for both families)
__global__ void __launch_bounds__(1024)
odom_bin_kernel(OdomBinJob job_a, OdomBinJob job_b, int n_pairs) {
  __shared__ unsigned s_cnt[kOdomMaxCells / 2];         // two u16 counters per word
  __shared__ int s_red[16][5];
  __shared__ unsigned s_part[16];
  __shared__ int s_box[5];
  const bool second = (int)blockIdx.x >= n_pairs;
  const OdomBinJob& job = second ? job_b : job_a;
  const float4* __restrict__ pts_all = job.pts; const uint16_t* __restrict__ ring_all = job.ring; const int* __restrict__ off = job.off;
  unsigned* __restrict__ tab_all = job.tab; float4* __restrict__ sorted_all = job.sorted;
  OdomPairDesc* __restrict__ desc = job.desc; int* __restrict__ mode = job.mode;
  const int b = (int)blockIdx.x - (second ? n_pairs : 0), tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s0 = off[b], n = off[b + 1] - s0;
  const float4* pts = pts_all + s0;
  const uint16_t* ring = ring_all + s0;
  unsigned* tab = tab_all + (size_t)b * kOdomTabStride;
  // ---- bounding box in columns + the conditions the grid relies on ----
  // The first kOdomBinKeep x 1024 points stay in registers for the two later passes (xyz + the packed ring / index word the
  // sorted copy carries): their loads are all requested before the first is used, and the histogram and scatter passes
  // do not go back to memory for them.  Longer clouds run the rest through the streaming loops.
  int lox = INT32_MAX, loy = INT32_MAX, hix = INT32_MIN, hiy = INT32_MIN, bad = n >= (1 << 24) ? 1 : 0;
  float kx[kOdomBinKeep], ky[kOdomBinKeep], kz[kOdomBinKeep];
  unsigned kring[(kOdomBinKeep + 3) / 4];                        // four 8-bit ring ids per word
  auto check = [&](float x, float y, float z, int r, int rprev, int i) {
    if (!(fabsf(x) < (float)kOdomRange && fabsf(y) < (float)kOdomRange && fabsf(z) < 1e30f) || r >= 256) { bad = 1; return; }
    if (i > 0 && rprev > r) bad = 1;
    const int cx = (int)floorf(x), cy = (int)floorf(y);
    lox = min(lox, cx); hix = max(hix, cx); loy = min(loy, cy); hiy = max(hiy, cy);
  };
#pragma unroll
  for (int k = 0; k < (kOdomBinKeep + 3) / 4; k++) kring[k] = 0;
  if (n > 0) {
#pragma unroll
    for (int k = 0; k < kOdomBinKeep; k++) {
      if (1024 * k >= n) break;                                   // uniform
      const int i = tid + 1024 * k, ic = min(i, n - 1);
      const float4 p = pts[ic];
      const int r = ring[ic], rprev = ring[max(ic - 1, 0)];
      kx[k] = p.x; ky[k] = p.y; kz[k] = p.z;
      kring[k / 4] |= ((unsigned)r & 0xffu) << (8 * (k & 3));
      if (i < n) check(p.x, p.y, p.z, r, rprev, i);
    }
  }
  auto kept = [&](int k) {                                        // the sorted copy's form of slot k: xyz + (ring << 24 | index)
    return make_float4(kx[k], ky[k], kz[k],
                       __int_as_float((int)(((kring[k / 4] >> (8 * (k & 3))) & 0xffu) << 24 | (unsigned)(tid + 1024 * k))));
  };
  for (int i = tid + 1024 * kOdomBinKeep; i < n; i += 1024) { const float4 p = pts[i]; check(p.x, p.y, p.z, ring[i], ring[i - 1], i); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lox = min(lox, __shfl_xor(lox, o)); loy = min(loy, __shfl_xor(loy, o));
    hix = max(hix, __shfl_xor(hix, o)); hiy = max(hiy, __shfl_xor(hiy, o)); bad |= __shfl_xor(bad, o);
  }
  if (lane == 0) { s_red[wave][0] = lox; s_red[wave][1] = loy; s_red[wave][2] = hix; s_red[wave][3] = hiy; s_red[wave][4] = bad; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 16; w++) {
      lox = min(lox, s_red[w][0]); loy = min(loy, s_red[w][1]); hix = max(hix, s_red[w][2]); hiy = max(hiy, s_red[w][3]); bad |= s_red[w][4];
    }
    int W = 0, H = 0;
    if (n > 0 && !bad) { W = hix - lox + 1; H = hiy - loy + 1; if ((long long)W * H > kOdomMaxCells) bad = 1; }
    s_box[0] = lox; s_box[1] = loy; s_box[2] = W; s_box[3] = H; s_box[4] = bad;
    OdomPairDesc d; d.ox = lox; d.oy = loy; d.W = bad ? 0 : W; d.H = bad ? 0 : H;
    desc[b] = d;
    mode[b] = bad ? 1 : 0;
  }
  __syncthreads();
  if (s_box[4] || n == 0) return;
  const int ox = s_box[0], oy = s_box[1], W = s_box[2], cells = W * s_box[3];
  // ---- histogram ----
  for (int k = tid; k < (cells + 1) / 2; k += 1024) s_cnt[k] = 0;
  __syncthreads();
  // Consecutive points of a ring sweep mostly fall into the same column (a column is 1 m, neighbouring returns are centimetres
  // apart), so 64 single-point atomics of a wavefront would serialise on two or three counter words.  Lanes holding a run
  // of equal columns are found with one ballot; the run's first lane adds (here) or claims (scatter) the whole run.
  const int wave_i0 = tid & ~63;
  auto run_of = [&](bool valid, int c, int& head_lane, int& len) {
    const int cc = valid ? c : -1 - lane;                         // lanes past the end: runs of their own, never used
    const int cprev = __shfl_up(cc, 1);
    const bool head = lane == 0 || cc != cprev;
    const unsigned long long hm = __ballot(head);
    const unsigned long long upto = (2ull << lane) - 1ull;        // lanes 0..lane (lane 63: all ones)
    head_lane = 63 - __clzll((long long)(hm & upto));
    const unsigned long long above = hm & ~upto;
    len = (above ? __ffsll((long long)above) - 1 : 64) - head_lane;
    return head;
  };
  auto count = [&](bool valid, int c) {
    int hl, len;
    if (run_of(valid, c, hl, len) && valid) atomicAdd(&s_cnt[c >> 1], (unsigned)len << (16 * (c & 1)));
  };
#pragma unroll
  for (int k = 0; k < kOdomBinKeep; k++) {
    if (wave_i0 + 1024 * k >= n) break;                           // uniform in the wavefront
    count(tid + 1024 * k < n, ((int)floorf(ky[k]) - oy) * W + ((int)floorf(kx[k]) - ox));
  }
  for (int i0 = wave_i0 + 1024 * kOdomBinKeep; i0 < n; i0 += 1024) {
    const int i = i0 + lane;
    const float4 p = pts[min(i, n - 1)];
    count(i < n, ((int)floorf(p.y) - oy) * W + ((int)floorf(p.x) - ox));
  }
  __syncthreads();
  // ---- exclusive scan: every thread owns an even-sized run of columns ----
  const int per = (((cells + 1023) / 1024) + 1) & ~1;
  const int c0 = min(tid * per, cells), c1 = min(c0 + per, cells);
  unsigned sum = 0;
  for (int c = c0; c < c1; c++) sum += (s_cnt[c >> 1] >> (16 * (c & 1))) & 0xffffu;
  unsigned incl = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o); if (lane >= o) incl += t; }
  if (lane == 63) s_part[wave] = incl;
  __syncthreads();
  unsigned run = incl - sum, total = 0;
  for (int w = 0; w < 16; w++) { if (w == wave) run += total; total += s_part[w]; }
  if (total != (unsigned)n) {                                     // a u16 counter wrapped (every wrap loses 65 535 or 65 536): brute force
    if (tid == 0) { OdomPairDesc d; d.ox = ox; d.oy = oy; d.W = 0; d.H = 0; desc[b] = d; mode[b] = 1; }
    return;
  }
  for (int c = c0; c < c1; c++) { tab[c] = run; run += (s_cnt[c >> 1] >> (16 * (c & 1))) & 0xffffu; }
  if (tid == 0) tab[cells] = (unsigned)n;
  __threadfence_block();
  __syncthreads();
  // ---- scatter: a column's points land in its run in arbitrary order (the minima are order-free) ----
  float4* sorted = sorted_all + s0;
  auto place = [&](bool valid, const float4 p) {
    const int c = ((int)floorf(p.y) - oy) * W + ((int)floorf(p.x) - ox);
    int hl, len;
    const bool head = run_of(valid, c, hl, len);
    const unsigned sh = 16 * (c & 1);
    unsigned old = 0;
    if (head && valid) old = atomicSub(&s_cnt[c >> 1], (unsigned)len << sh);
    old = __shfl(old, hl);                                        // the run's first lane claimed slots [left - len, left)
    const unsigned k = ((old >> sh) & 0xffffu) - 1u - (unsigned)(lane - hl);
    if (valid) sorted[tab[c] + k] = p;
  };
#pragma unroll
  for (int k = 0; k < kOdomBinKeep; k++) {
    if (wave_i0 + 1024 * k >= n) break;
    place(tid + 1024 * k < n, kept(k));
  }
  for (int i0 = wave_i0 + 1024 * kOdomBinKeep; i0 < n; i0 += 1024) {
    const int i = i0 + lane, ic = min(i, n - 1);
    float4 p = pts[ic];
    p.w = __int_as_float((int)(((unsigned)ring[ic] & 0xffu) << 24 | (unsigned)i));
    place(i < n, p);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same index when the call holds FEW pairs with LONG clouds (the SLAM step's one 64-beam less-flat list: ~100 k points
// kept the one workgroup above busy for 165 us of the odometry stream while the chip idled): G workgroups per cloud and
// the counters in the table itself (u32 in global memory: no per-column limit), four launches —
//   odom_bin_box_kernel      bounding box + the grid's conditions per slice; zeroes the table
//   odom_bin_count_kernel    box verdict (desc / mode as above), histogram by atomics (one per run of equal columns)
//   odom_bin_scan_kernel     exclusive scan of the table in place (one workgroup per cloud), a copy as scatter cursors
//   odom_bin_scatter_kernel  points into their column's run
// The order inside a column differs from the kernel above and from run to run; the queries' minima are lexicographic
// (distance, index) keys and do not see it.
// ---------------------------------------------------------------------------------------------------------------------------
struct OdomBinSplit {
  int* part;               // jobs x G x 8: lox, loy, hix, hiy, bad of a slice
  unsigned* cur;           // jobs x kOdomTabStride scatter cursors
  int G;
};
constexpr int kOdomSplitThreads = 256;

__device__ __forceinline__ bool odom_run_of(bool valid, int c, int lane, int& head_lane, int& len) {
  const int cc = valid ? c : -1 - lane;
  const int cprev = __shfl_up(cc, 1);
  const bool head = lane == 0 || cc != cprev;
  const unsigned long long hm = __ballot(head);
  const unsigned long long upto = (2ull << lane) - 1ull;
  head_lane = 63 - __clzll((long long)(hm & upto));
  const unsigned long long above = hm & ~upto;
  len = (above ? __ffsll((long long)above) - 1 : 64) - head_lane;
  return head;
}

struct OdomBinSlice { const OdomBinJob* job; int b, job_i, s0, n, i0, i1; };
__device__ __forceinline__ OdomBinSlice odom_bin_slice(const OdomBinJob& job_a, const OdomBinJob& job_b, int n_pairs, int G) {
  OdomBinSlice sl;
  sl.job_i = blockIdx.y;
  const bool second = sl.job_i >= n_pairs;
  sl.job = second ? &job_b : &job_a;
  sl.b = sl.job_i - (second ? n_pairs : 0);
  sl.s0 = sl.job->off[sl.b]; sl.n = sl.job->off[sl.b + 1] - sl.s0;
  const int chunk = ((sl.n + G - 1) / G + 63) & ~63;                // whole wavefronts: the run detection works on 64 consecutive points
  sl.i0 = (int)min((long long)blockIdx.x * chunk, (long long)sl.n); sl.i1 = min(sl.i0 + chunk, sl.n);
  return sl;
}

__global__ void __launch_bounds__(kOdomSplitThreads) odom_bin_box_kernel(OdomBinJob job_a, OdomBinJob job_b, int n_pairs, OdomBinSplit sp) {
  __shared__ int s_red[kOdomSplitThreads / 64][5];
  const OdomBinSlice sl = odom_bin_slice(job_a, job_b, n_pairs, sp.G);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float4* pts = sl.job->pts + sl.s0;
  const uint16_t* ring = sl.job->ring + sl.s0;
  unsigned* tab = sl.job->tab + (size_t)sl.b * kOdomTabStride;
  for (int c = blockIdx.x * kOdomSplitThreads + tid; c < kOdomTabStride; c += sp.G * kOdomSplitThreads) tab[c] = 0u;
  int lox = INT32_MAX, loy = INT32_MAX, hix = INT32_MIN, hiy = INT32_MIN, bad = 0;
  for (int i = sl.i0 + tid; i < sl.i1; i += kOdomSplitThreads) {
    const float4 p = pts[i];
    const int r = ring[i], rprev = ring[max(i - 1, 0)];
    if (!(fabsf(p.x) < (float)kOdomRange && fabsf(p.y) < (float)kOdomRange && fabsf(p.z) < 1e30f) || r >= 256) { bad = 1; continue; }
    if (i > 0 && rprev > r) bad = 1;
    const int cx = (int)floorf(p.x), cy = (int)floorf(p.y);
    lox = min(lox, cx); hix = max(hix, cx); loy = min(loy, cy); hiy = max(hiy, cy);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lox = min(lox, __shfl_xor(lox, o)); loy = min(loy, __shfl_xor(loy, o));
    hix = max(hix, __shfl_xor(hix, o)); hiy = max(hiy, __shfl_xor(hiy, o)); bad |= __shfl_xor(bad, o);
  }
  if (lane == 0) { s_red[wave][0] = lox; s_red[wave][1] = loy; s_red[wave][2] = hix; s_red[wave][3] = hiy; s_red[wave][4] = bad; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < kOdomSplitThreads / 64; w++) {
      lox = min(lox, s_red[w][0]); loy = min(loy, s_red[w][1]); hix = max(hix, s_red[w][2]); hiy = max(hiy, s_red[w][3]); bad |= s_red[w][4];
    }
    int* part = sp.part + ((size_t)sl.job_i * sp.G + blockIdx.x) * 8;
    part[0] = lox; part[1] = loy; part[2] = hix; part[3] = hiy; part[4] = bad;
  }
}

// the box of the whole cloud from the slices' reports: {ox, oy, W, H, bad} (the one-workgroup kernel's rules)
__device__ __forceinline__ void odom_bin_verdict(const int* __restrict__ part, int G, int n, int* s_box) {
  int lox = INT32_MAX, loy = INT32_MAX, hix = INT32_MIN, hiy = INT32_MIN, bad = n >= (1 << 24) ? 1 : 0;
  for (int g = 0; g < G; g++) {
    lox = min(lox, part[8 * g]); loy = min(loy, part[8 * g + 1]); hix = max(hix, part[8 * g + 2]); hiy = max(hiy, part[8 * g + 3]); bad |= part[8 * g + 4];
  }
  int W = 0, H = 0;
  if (n > 0 && !bad) { W = hix - lox + 1; H = hiy - loy + 1; if ((long long)W * H > kOdomMaxCells) bad = 1; }
  s_box[0] = lox; s_box[1] = loy; s_box[2] = bad ? 0 : W; s_box[3] = bad ? 0 : H; s_box[4] = bad;
}

__global__ void __launch_bounds__(kOdomSplitThreads) odom_bin_count_kernel(OdomBinJob job_a, OdomBinJob job_b, int n_pairs, OdomBinSplit sp) {
  __shared__ int s_box[5];
  const OdomBinSlice sl = odom_bin_slice(job_a, job_b, n_pairs, sp.G);
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid == 0) {
    odom_bin_verdict(sp.part + (size_t)sl.job_i * sp.G * 8, sp.G, sl.n, s_box);
    if (blockIdx.x == 0) {
      OdomPairDesc d; d.ox = s_box[0]; d.oy = s_box[1]; d.W = s_box[2]; d.H = s_box[3];
      sl.job->desc[sl.b] = d;
      sl.job->mode[sl.b] = s_box[4] ? 1 : 0;
    }
  }
  __syncthreads();
  if (s_box[4] || sl.n == 0) return;
  const int ox = s_box[0], oy = s_box[1], W = s_box[2];
  const float4* pts = sl.job->pts + sl.s0;
  unsigned* tab = sl.job->tab + (size_t)sl.b * kOdomTabStride;
  for (int i0 = sl.i0 + (tid & ~63); i0 < sl.i1; i0 += kOdomSplitThreads) {
    const int i = i0 + lane;
    const bool valid = i < sl.i1;
    const float4 p = pts[min(i, sl.i1 - 1)];
    const int c = ((int)floorf(p.y) - oy) * W + ((int)floorf(p.x) - ox);
    int hl, len;
    if (odom_run_of(valid, c, lane, hl, len) && valid) atomicAdd(&tab[c], (unsigned)len);
  }
}

__global__ void __launch_bounds__(1024) odom_bin_scan_kernel(OdomBinJob job_a, OdomBinJob job_b, int n_pairs, OdomBinSplit sp) {
  __shared__ unsigned s_part[16];
  const bool second = (int)blockIdx.x >= n_pairs;
  const OdomBinJob& job = second ? job_b : job_a;
  const int b = (int)blockIdx.x - (second ? n_pairs : 0), tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = job.off[b + 1] - job.off[b];
  if (job.mode[b] != 0 || n == 0) return;
  const OdomPairDesc d = job.desc[b];
  const int cells = d.W * d.H;
  unsigned* tab = job.tab + (size_t)b * kOdomTabStride;
  unsigned* cur = sp.cur + (size_t)blockIdx.x * kOdomTabStride;
  const int per = (cells + 1023) / 1024;
  const int c0 = min(tid * per, cells), c1 = min(c0 + per, cells);
  unsigned sum = 0;
  for (int c = c0; c < c1; c++) sum += tab[c];
  unsigned incl = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o); if (lane >= o) incl += t; }
  if (lane == 63) s_part[wave] = incl;
  __syncthreads();
  unsigned run = incl - sum;
  for (int w = 0; w < wave; w++) run += s_part[w];
  for (int c = c0; c < c1; c++) { const unsigned k = tab[c]; tab[c] = run; cur[c] = run; run += k; }
  if (tid == 0) tab[cells] = (unsigned)n;
}

__global__ void __launch_bounds__(kOdomSplitThreads) odom_bin_scatter_kernel(OdomBinJob job_a, OdomBinJob job_b, int n_pairs, OdomBinSplit sp) {
  const OdomBinSlice sl = odom_bin_slice(job_a, job_b, n_pairs, sp.G);
  const int tid = threadIdx.x, lane = tid & 63;
  if (sl.job->mode[sl.b] != 0 || sl.n == 0) return;
  const OdomPairDesc d = sl.job->desc[sl.b];
  const int ox = d.ox, oy = d.oy, W = d.W;
  const float4* pts = sl.job->pts + sl.s0;
  const uint16_t* ring = sl.job->ring + sl.s0;
  unsigned* cur = sp.cur + (size_t)sl.job_i * kOdomTabStride;
  float4* sorted = sl.job->sorted + sl.s0;
  for (int i0 = sl.i0 + (tid & ~63); i0 < sl.i1; i0 += kOdomSplitThreads) {
    const int i = i0 + lane, ic = min(i, sl.i1 - 1);
    const bool valid = i < sl.i1;
    float4 p = pts[ic];
    p.w = __int_as_float((int)(((unsigned)ring[ic] & 0xffu) << 24 | (unsigned)ic));
    const int c = ((int)floorf(p.y) - oy) * W + ((int)floorf(p.x) - ox);
    int hl, len;
    const bool head = odom_run_of(valid, c, lane, hl, len);
    unsigned first = 0;
    if (head && valid) first = atomicAdd(&cur[c], (unsigned)len);
    first = __shfl(first, hl);
    if (valid) sorted[first + (unsigned)(lane - hl)] = p;
  }
}

// L lanes (a power of two, one query per L-lane group) walk the (2r+1)^2 column square around
// column (cx, cy): every row's run is read L targets at a time (coalesced); f(float4) per target.
template <int L, class F>
__device__ __forceinline__ void odom_walk_runs3(const float4* __restrict__ sorted, const int (&b0)[3], const int (&b1)[3], int sl, F&& f) {
#pragma unroll
  for (int k = 0; k < 3; k++) {
    int i = b0[k] + sl;
    for (; i + 3 * L < b1[k]; i += 4 * L) {                 // four loads in flight per lane
      const float4 p0 = sorted[i], p1 = sorted[i + L], p2 = sorted[i + 2 * L], p3 = sorted[i + 3 * L];
      f(p0); f(p1); f(p2); f(p3);
    }
    for (; i < b1[k]; i += L) f(sorted[i]);
  }
}

// run bounds of the (up to) three rows of the 3 x 3 column square: row-major table, the run of columns [x0, x1] of a
// row ends where column x1 + 1 starts; all six loads are requested together and are uniform inside the lane group
__device__ __forceinline__ void odom_runs3(const unsigned* __restrict__ tab, int W, int H, int cx, int cy, int (&b0)[3], int (&b1)[3]) {
  const int x0 = max(cx - 1, 0), x1 = min(cx + 1, W - 1);
  const int y0 = max(cy - 1, 0), y1 = min(cy + 1, H - 1);
  const bool none = x0 > x1 || y0 > y1;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    b0[k] = 0; b1[k] = 0;
    if (!none && y0 + k <= y1) { b0[k] = (int)tab[(y0 + k) * W + x0]; b1[k] = (int)tab[(y0 + k) * W + x1 + 1]; }
  }
}

template <int L, class F>
__device__ __forceinline__ void odom_walk(const unsigned* __restrict__ tab, const float4* __restrict__ sorted,
                                          int W, int H, int cx, int cy, int r, int sl, F&& f) {
  const int x0 = max(cx - r, 0), x1 = min(cx + r, W - 1);
  const int y0 = max(cy - r, 0), y1 = min(cy + r, H - 1);
  if (x0 > x1 || y0 > y1) return;
  for (int y = y0; y <= y1; y++) {
    const int b0 = (int)tab[y * W + x0], b1 = (int)tab[y * W + x1 + 1];
    for (int i = b0 + sl; i < b1; i += L) f(sorted[i]);
  }
}

// exact lower bound (squared, with a 1e-4 safety factor >> f32 rounding) on the distance from q to
// any target outside the walked square of half-width r columns
__device__ __forceinline__ float odom_gap_sq(float3 q, float fx, float fy, int r) {
  const float g = fminf(fminf(q.x - (fx - (float)r), (fx + 1.f + (float)r) - q.x), fminf(q.y - (fy - (float)r), (fy + 1.f + (float)r) - q.y));
  return g * g * 0.9999f;
}

// minimum of a packed (distance bits << 32 | tie word) key over the L lanes of a query
template <int L>
__device__ __forceinline__ unsigned long long group_min_key(unsigned long long k) {
#pragma unroll
  for (int o = L / 2; o > 0; o >>= 1) {
    const unsigned lo = __shfl_xor((unsigned)k, o), hi = __shfl_xor((unsigned)(k >> 32), o);
    const unsigned long long k2 = ((unsigned long long)hi << 32) | lo;
    k = k2 < k ? k2 : k;
  }
  return k;
}

// EDGE = false: flat queries against the less-flat cloud (:166-258); EDGE = true: sharp queries against the
// less-sharp cloud (:81-163: second point only from rings (id, id + 2.5] above / [id - 2.5, id) below).
template <int L, bool EDGE>
__device__ __forceinline__ void odom_grid_query(const BatchView& bv, const OdomView& ov, const OdomIndex& ix, const double* __restrict__ poses,
                                                const int* __restrict__ status, double* __restrict__ rec, int block) {
  // XCD-aware block -> (pair, tile) mapping.  Workgroups are dealt round-robin to the 8 XCDs (block i -> XCD i % 8)
  // and every XCD has its own L2: all `tiles` workgroups of a pair are given ids with the same residue mod 8 and
  // consecutive quotients, so one XCD reads that pair's 0.5 MB of sorted targets + column table once instead of
  // every XCD re-reading it at a different time (PMC per launch: FETCH_SIZE 1.72 -> 0.35 GB, L2 hit rate 31 -> 86 %;
  // the kernel time only moved by 4 %: it is VALU / latency bound, the misses were hidden by 8 waves/SIMD).
  const int tiles = ix.tiles;
  const int group = block / (8 * tiles), within = block - group * (8 * tiles);
  const int b = group * 8 + (within & 7);
  const int tile = within >> 3;
  if (b >= bv.n_scans) return;
  if (ix.mode[b] != 0) return;                                   // brute-force kernel owns this pair
  const int n_sharp = bv.corner_off[b + 1] - bv.corner_off[b];
  const int n_flat = bv.surf_off[b + 1] - bv.surf_off[b];
  const int sl = threadIdx.x % L;
  const int qf = tile * (kOdomBlock / L) + threadIdx.x / L;
  if (qf >= (EDGE ? n_sharp : n_flat)) return;
  constexpr int kOut = EDGE ? 6 : 4;
  double* out = rec + (EDGE ? edge_rec_off(bv, bv.corner_off[b] + qf) : plane_rec_off(bv, bv.surf_off[b] + qf));
  const int* t_off = EDGE ? ov.last_ls_off : ov.last_lf_off;
  const int s0 = t_off[b], s1 = t_off[b + 1];
  const float4* tp = (EDGE ? ov.last_ls : ov.last_lf) + s0;
  const float thr = (float)ov.dist_sq_threshold;
  bool ok = status[b] == 0 && s1 > s0;
  float3 q = make_float3(0, 0, 0);
  if (ok) {
    const float4 f = EDGE ? bv.corner[bv.corner_off[b] + qf] : bv.surf[bv.surf_off[b] + qf];
    q = transform_point_f32(load_pose(poses + 7 * b), f.x, f.y, f.z);
    ok = fabsf(q.x) < (float)(kOdomRange + 7) && fabsf(q.y) < (float)(kOdomRange + 7);   // also rejects NaN
  }
  if (!ok) { if (sl < kOut) out[sl] = 0.0; return; }
  const float fx = floorf(q.x), fy = floorf(q.y);
  const OdomPairDesc pd = ix.desc[b];
  const int cx = (int)fx - pd.ox, cy = (int)fy - pd.oy;           // may lie outside [0, W) x [0, H): the walk clips
  const unsigned* tab = ix.tab + (size_t)b * kOdomTabStride;
  const float4* sorted = ix.sorted + s0;
  // Running minima are packed keys (f32 distance bits << 32 | tie word): distances are >= 0, so one
  // unsigned compare realises (distance, tie) lexicographic order.
  int b0[3], b1[3];
  odom_runs3(tab, pd.W, pd.H, cx, cy, b0, b1);                     // the 3 x 3 square serves both walks
  // ---- exact 1-NN (:169), ties -> lower index.  Tie word = ring << 24 | index: on a ring-monotone cloud
  //      ordering by (ring, index) is ordering by index ----
  unsigned long long kbest = ~0ull;
  auto nearest = [&](const float4 p) __attribute__((always_inline)) {
    const unsigned long long k = ((unsigned long long)__float_as_uint(odom_dist(p, q)) << 32) | (unsigned)__float_as_int(p.w);
    kbest = k < kbest ? k : kbest;
  };
  int l1 = 0;                                                      // the level the 1-NN walk ended on
  for (int l = 0; l < kOdomLevels; l++) {
    const int r = odom_level_radius(l);
    l1 = l;
    if (l == 0) odom_walk_runs3<L>(sorted, b0, b1, sl, nearest);
    else odom_walk<L>(tab, sorted, pd.W, pd.H, cx, cy, r, sl, nearest);
    kbest = group_min_key<L>(kbest);
    if (__uint_as_float((unsigned)(kbest >> 32)) < odom_gap_sq(q, fx, fy, r)) break;
  }
  if (!(__uint_as_float((unsigned)(kbest >> 32)) < thr)) { if (sl < kOut) out[sl] = 0.0; return; }    // :87 / :173 (NaN / none: not <)
  const int closest = (int)((unsigned)kbest & 0xffffffu), id = (int)((unsigned)kbest >> 24 & 0xffu);
  // ---- ring-window minima (:183-232).  The reference sweeps forward from `closest` (rings up to id + nearby_scan), then
  //      backward (down to id - nearby_scan), with running minima that start at the gate and are replaced on strict '<':
  //      at equal distance a forward candidate beats a backward one, the forward ones tie to the lowest index and the
  //      backward ones to the highest.  That is ONE lexicographic minimum per output with the tie word
  //          j > closest:  j - closest            (1 .. 2^24 - 1)
  //          j < closest:  2^24 + (closest - j)
  //      On a ring-monotone cloud "ring <= id forward / ring >= id backward" (min2) is "ring == id", and the rest of the
  //      window is min3.  `none` = gate distance with tie 0: a key is below it exactly when its distance is below the gate.
  //      The window test is done on integers ----
  const float hi_ring = (float)id + (float)ov.nearby_scan, lo_ring = (float)id - (float)ov.nearby_scan;   // as the brute-force kernel forms them
  // (float)rj > hi_ring  <=>  rj > floor(hi_ring);  (float)rj < lo_ring  <=>  rj < ceil(lo_ring)  (rj an integer; a NaN bound excludes nothing)
  const int hi_i = hi_ring == hi_ring ? (int)fminf(fmaxf(floorf(hi_ring), -1048576.f), 1048576.f) : 1048576;
  const int lo_i = lo_ring == lo_ring ? (int)fminf(fmaxf(ceilf(lo_ring), -1048576.f), 1048576.f) : -1048576;
  const bool window = hi_i >= lo_i;
  const unsigned span = window ? (unsigned)(hi_i - lo_i) : 0u;
  const int lo_w = window ? lo_i : (1 << 22);                      // empty window: every ring is "outside"
  const unsigned long long none = (unsigned long long)__float_as_uint(thr) << 32;
  unsigned long long k2, k3;
  auto windowed = [&](const float4 p) __attribute__((always_inline)) {
    const int w = __float_as_int(p.w), j = w & 0xffffff, rj = (int)((unsigned)w >> 24);
    const int t = j - closest;
    const bool same = rj == id;
    const bool skip = (unsigned)(rj - lo_w) > span || t == 0 || (EDGE && same);   // edges take their second point from other rings only (:96, :121)
    const float d = odom_dist(p, q);
    const unsigned tie = t > 0 ? (unsigned)t : 0x1000000u - (unsigned)t;
    unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | tie;
    key = skip ? ~0ull : key;                                    // (a NaN distance has bits above the gate's: never below `none`)
    // branch-free updates (selects): branches here make the compiler address the minima through memory
    if (EDGE) k2 = key < k2 ? key : k2;                            // the second point of an edge plays the role of min2
    else {
      k2 = (same && key < k2) ? key : k2;
      k3 = (!same && key < k3) ? key : k3;
    }
  };
  // the windowed minima are no closer than the 1-NN: the levels that could not bound IT cannot bound them either
  for (int l = l1; l < kOdomLevels; l++) {
    const int r = odom_level_radius(l);
    k2 = k3 = none;                                                // each level re-walks the inner square too
    if (l == 0) odom_walk_runs3<L>(sorted, b0, b1, sl, windowed);
    else odom_walk<L>(tab, sorted, pd.W, pd.H, cx, cy, r, sl, windowed);
    k2 = group_min_key<L>(k2);
    if (!EDGE) k3 = group_min_key<L>(k3);
    const float g = odom_gap_sq(q, fx, fy, r);
    if (__uint_as_float((unsigned)(k2 >> 32)) < g && (EDGE || __uint_as_float((unsigned)(k3 >> 32)) < g)) break;
  }
  if (sl != 0) return;
  auto index_of = [&](unsigned long long k) {
    if (k == none) return -1;
    const unsigned tie = (unsigned)k;
    return tie < 0x1000000u ? closest + (int)tie : closest - (int)(tie - 0x1000000u);
  };
  const int min2 = index_of(k2), min3 = EDGE ? -1 : index_of(k3);
  d3 C = mk3(0, 0, 0), N = mk3(0, 0, 0);
  if (EDGE) {
    if (min2 >= 0) {                                             // :143-162
      const float4 a = tp[closest], c = tp[min2];
      const d3 A = mk3((double)a.x, (double)a.y, (double)a.z), Bp = mk3((double)c.x, (double)c.y, (double)c.z);
      N = odom_unit(A - Bp);
      C = A;
    }
    out[0] = C.x; out[1] = C.y; out[2] = C.z; out[3] = N.x; out[4] = N.y; out[5] = N.z;
    return;
  }
  if (min2 >= 0 && min3 >= 0) {                                  // :234-256, lidar_factor.h:70-78
    const float4 a = tp[closest], c = tp[min2], e = tp[min3];
    const d3 A = mk3((double)a.x, (double)a.y, (double)a.z), Bp = mk3((double)c.x, (double)c.y, (double)c.z),
             Cp = mk3((double)e.x, (double)e.y, (double)e.z);
    N = odom_unit(cross(A - Bp, A - Cp));
    C = odom_centroid3(A, Bp, Cp);
  }
  out[0] = N.x; out[1] = N.y; out[2] = N.z; out[3] = dot(N, C);
}

// One launch for both query kinds: blocks [0, plane_blocks) serve the flat queries, the rest the sharp ones.  The edge
// queries are a tenth of the work and used to run as a launch of their own behind the plane kernel's tail.
template <int L>
__global__ void __launch_bounds__(kOdomBlock)
assoc_scan2scan_grid_kernel(BatchView bv, OdomView ov, OdomIndex ix_plane, OdomIndex ix_edge, int plane_blocks,
                            const double* __restrict__ poses, const int* __restrict__ status, double* __restrict__ rec) {
  if ((int)blockIdx.x < plane_blocks) odom_grid_query<L, false>(bv, ov, ix_plane, poses, status, rec, (int)blockIdx.x);
  else odom_grid_query<L, true>(bv, ov, ix_edge, poses, status, rec, (int)blockIdx.x - plane_blocks);
}

}  // namespace msfl

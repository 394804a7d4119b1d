// msfl_extract.cuh — stage A: per-scan edge/plane feature extraction on gfx950.
//
// Replaces RealHandleLaserCloudMessage (msf_loam_node.cc:160-378).  Four kernels per batch (round 2: the sector sort is gone):
//   extract_prepare_kernel    one 1024-thread workgroup per scan: invalid-point removal (:85-111),
//                             relative time (:128-156), stable split into rings + concat (:188-195)
//   extract_curvature_kernel  one thread per point: 11-tap curvature (:213-240) + neighbour-gap flags
//   extract_pick_kernel       one wavefront per (scan, ring): the serial sharp / less-sharp / flat pick
//                             (:263-344) as repeated wave-wide arg-max / arg-min over the sector's
//                             (curvature, index) keys held in registers, neighbour suppression on LDS bitmasks
//   extract_compact_kernel    per scan: order the per-ring lists into the reference's push order,
//                             apply the lidar->imu extrinsic (:367-371)
// All index outputs are bit-exact w.r.t. the CPU oracle; the unstable std::sort tie order of the
// reference is fixed to (curvature, index) ascending.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "msfl_math.cuh"

namespace msfl {

constexpr int kMaxRings = 128;          // kMaxScanNum, msf_loam_node.cc:79
constexpr int kRingCapacity = 8128;     // points per ring handled by the LDS bitmasks (256 words each with the two spare ones: the pick kernel's
                                        // workgroup takes exactly 20 KB and eight of them share a CU)

struct ExtractParams {
  double min_range;
  double scan_period;
  double curvature_threshold;   // compared as double against the f32 curvature (:275, :312)
  double neighbor_gap_sq;       // compared as double against the f32 squared gap (:293)
  int sectors, max_sharp, max_less_sharp, max_flat;
  float min_range_sq;           // the smallest f32 s with (double)sqrtf(s) >= min_range (host: extract_params); 0 when nothing is too close
};

#ifndef MSFL_STREAM_BLOCK
#define MSFL_STREAM_BLOCK 128
#endif
#ifndef MSFL_EXTRACT_WAVES
#define MSFL_EXTRACT_WAVES 4
#endif
constexpr int kExWaves = MSFL_EXTRACT_WAVES;       // independent wavefronts per workgroup of the sector sort / pick kernels
constexpr int kStreamBlock = MSFL_STREAM_BLOCK;   // threads per workgroup of the one-thread-per-element kernels (curvature, batched voxel filter)

struct ExtractView {
  // inputs (per scan b: [off[b], off[b+1]) )
  const float4* in_pts; const uint16_t* in_ring; const int* off; int n_scans; int n_total;
  // outputs at the same offsets
  float4* full_pts; uint16_t* full_ring; float* curvature; uint8_t* label;
  int* sharp_idx; int* less_sharp_idx; int* flat_idx; int* less_flat_idx;
  int* n_full; int* n_sharp; int* n_less_sharp; int* n_flat; int* n_less_flat;
  int* status;
  // scratch
  uint8_t* gap;                // n_total: 1 if |p[i+1]-p[i]|^2 > neighbor_gap_sq
  int* ring_tab;               // n_scans x (kMaxRings + 1): ring start offsets (scan-local)
  int* tmp_idx;                // 4 x n_total: per-ring lists before compaction
  int* ring_cnt;               // n_scans x kMaxRings x 4
};

__device__ __forceinline__ bool point_valid(float4 p, float min_range_sq) {
  // RemoveInvalidPointsFromCloud: getVector3fMap().norm() < min_range || !finite.  sqrtf is correctly rounded and
  // monotone, so "(double)sqrtf(s) < min_range" is "s < min_range_sq" for the f32 threshold the host searched for
  // (no square root, no f64 compare per point); a finite s has finite coordinates, so the finite test only runs when
  // s overflowed or is NaN
  const float s = p.x * p.x + p.y * p.y + p.z * p.z;
  bool valid = !(s < min_range_sq);
  if (!(fabsf(s) < INFINITY)) valid = valid && isfinite(p.x) && isfinite(p.y) && isfinite(p.z);
  return valid;
}

// lanes of the wave that are valid and carry the same ring id (< 128): seven ballots, one per ring bit,
// instead of one loop iteration per distinct ring in the group (a 64-beam driver interleaves all rings)
__device__ __forceinline__ unsigned long long same_ring_lanes(bool valid, int r, int nbits) {
  unsigned long long m = __ballot(valid);
  for (int bit = 0; bit < nbits; bit++) {               // nbits: those of the scan's highest ring id (4 on a VLP-16), uniform
    const bool set = (r >> bit) & 1;
    const unsigned long long b = __ballot(valid && set);
    m &= set ? b : ~b;
  }
  return m;
}

// Each of the 16 wavefronts owns a CONTIGUOUS slice of the driver-order cloud, so the stable ring
// split needs no per-chunk workgroup barriers: pass A counts (wave, ring) populations, one scan
// turns them into write cursors, pass B re-walks the slice and scatters with wave-local ranks
// (ballot per distinct ring; the cursors of a wave are touched by that wave only).
__global__ void __launch_bounds__(1024) extract_prepare_kernel(ExtractView v, ExtractParams prm) {
  __shared__ int s_off[kMaxRings + 1];
  __shared__ int s_wrap[kMaxRings];
  __shared__ int s_cur[16][kMaxRings];      // pass A: population of (wave, ring); pass B: write cursor
  __shared__ int s_first[16], s_badw[16], s_rmax[16];
  __shared__ int s_flag[3];
  __shared__ double s_start_ori;
  __shared__ double s_lasta[16][kMaxRings];   // pass B: raw angle of the last point of (wave, ring) so far; NaN = none yet
  __shared__ double s_firsta[16][kMaxRings];  // raw angle of the first point of (wave, ring)
  __shared__ double s_pre_last[kMaxRings];    // early-wrap pre-pass: angle of the ring's last point so far (NaN = none)
  __shared__ int s_pre_cnt[kMaxRings];        //                       points of the ring so far
  __shared__ int s_early[kMaxRings];          // output index of the ring's wrap point if it lies in the cloud's first 256 points
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int o = v.off[b];
  const int n = v.off[b + 1] - o;
  const float4* in = v.in_pts + o;
  const uint16_t* in_ring = v.in_ring + o;
  for (int k = tid; k < 16 * kMaxRings; k += 1024) { (&s_cur[0][0])[k] = 0; (&s_lasta[0][0])[k] = __longlong_as_double(0x7ff8000000000000ll); }
  __syncthreads();
  const int slice = ((n + 15) / 16 + 63) & ~63;            // multiple of 64: a 64-point group never straddles two waves
  const int w0 = min(wave * slice, n), w1 = min(w0 + slice, n);
  // ---- pass A: populations, first valid point, ring check ----
  int first = 0x7fffffff, bad = 0, rmax = 0;
  // four 64-point groups per iteration with their loads issued up front: one workgroup per CU keeps 16 wavefronts
  // there, and one 16-byte load per lane in flight is far from what HBM needs to stay busy
#ifndef MSFL_PREP_GROUPS
#define MSFL_PREP_GROUPS 4
#endif
  constexpr int kGroups = MSFL_PREP_GROUPS;
  for (int g0 = w0; g0 < w1; g0 += 64 * kGroups) {
    float4 pp[kGroups]; int rr[kGroups];
#pragma unroll
    for (int u = 0; u < kGroups; u++) {
      const int i = g0 + 64 * u + lane;
      pp[u] = make_float4(0, 0, 0, 0); rr[u] = 0;
      if (i < w1) { pp[u] = in[i]; rr[u] = in_ring[i]; }
    }
#pragma unroll
    for (int u = 0; u < kGroups; u++) {
      const int g = g0 + 64 * u;
      if (g >= w1) break;
      const int i = g + lane;
      bool valid = false;
      int r = -1;
      if (i < w1) {
        valid = point_valid(pp[u], prm.min_range_sq);
        if (valid) { r = rr[u]; if (r >= kMaxRings) { bad = 1; valid = false; } }   // CHECK_LT(point.ring, 128), :136
      }
      const unsigned long long any_valid = __ballot(valid);
      if (any_valid && first == 0x7fffffff) first = g + (__ffsll((long long)any_valid) - 1);
      if (valid) { atomicAdd(&s_cur[wave][r], 1); rmax = max(rmax, r); }       // LDS atomic: one instruction instead of the ballot match
    }
  }
  bad = __any(bad) ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) rmax = max(rmax, __shfl_xor(rmax, o));
  if (lane == 0) { s_first[wave] = first; s_badw[wave] = bad; s_rmax[wave] = rmax; }
  __syncthreads();
  if (tid == 0) {
    int f = 0x7fffffff, bd = 0, rm = 0;
    for (int w = 0; w < 16; w++) { f = min(f, s_first[w]); bd |= s_badw[w]; rm = max(rm, s_rmax[w]); }
    s_flag[0] = f; s_flag[1] = bd; s_flag[2] = 32 - __clz(rm | 1);          // bits of the highest ring id present
  }
  // ring totals -> s_off (exclusive), then (wave, ring) cursors
  if (tid < kMaxRings) {
    int t = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) t += s_cur[w][tid];
    s_wrap[tid] = t;                                          // borrowed as the ring total
  }
  __syncthreads();
  if (s_flag[1] || s_flag[0] == 0x7fffffff) {
    if (tid == 0) {
      v.status[b] = s_flag[1] ? 5 /*MSFL_BAD_RING*/ : 3 /*MSFL_BAD_ARG: empty valid cloud, CHECK :186,:200*/;
      v.n_full[b] = 0; v.n_sharp[b] = 0; v.n_less_sharp[b] = 0; v.n_flat[b] = 0; v.n_less_flat[b] = 0;
    }
    if (tid <= kMaxRings) v.ring_tab[b * (kMaxRings + 1) + tid] = 0;
    return;
  }
  if (tid == 0) {
    int run = 0;
    for (int r = 0; r < kMaxRings; r++) { s_off[r] = run; run += s_wrap[r]; }
    s_off[kMaxRings] = run;
    const float4 f = in[s_flag[0]];
    s_start_ori = -atan2((double)f.y, (double)f.x);       // :131
  }
  __syncthreads();
  if (tid <= kMaxRings) v.ring_tab[b * (kMaxRings + 1) + tid] = s_off[tid];
  if (tid < kMaxRings) {
    int run = s_off[tid];
#pragma unroll
    for (int w = 0; w < 16; w++) { const int t = s_cur[w][tid]; s_cur[w][tid] = run; run += t; }
    s_wrap[tid] = 0x7fffffff;
  }
  __syncthreads();
  const int N = s_off[kMaxRings];
  const int ring_bits = __builtin_amdgcn_readfirstlane(s_flag[2]);
  const double start_ori = s_start_ori;
  const double two_pi = 2 * 3.14159265358979323846;
  float4* out_pts = v.full_pts + o;
  uint16_t* out_ring = v.full_ring + o;
  // relative angle of a point (:139-142): fmod(a, 2 pi) with a = ori - start_ori + 2 pi in [0, 4 pi]: fmod is exact and so is
  // a - 2 pi for 2 pi <= a <= 4 pi (Sterbenz), so two conditional subtractions give the same bits
  auto rel_angle = [&](const float4 p) __attribute__((always_inline)) {
    const double ori = -atan2((double)p.y, (double)p.x);
    double a = ori - start_ori + two_pi;
    if (a >= two_pi) a -= two_pi;
    if (a >= two_pi) a -= two_pi;
    return a;
  };
  // ---- early wraps.  A ring whose first point lies just BEHIND the cloud's first point in azimuth starts at an angle just
  //      below 2 pi and wraps at its second point: that is about half the rings of a spinning lidar, and every point of such a
  //      ring takes the +2 pi time.  Found after the scatter, that meant rewriting the time word of half the cloud (a 4-byte
  //      write into a 16-byte point = a read-modify-write of the whole line, plus the parked candidates).  So wavefront 0 runs
  //      the wrap test over the first 256 points of the cloud first, with exactly the arithmetic of the pass below; pass B
  //      then stores the final time for everything behind a wrap point found here, and the fix-up only serves later wraps ----
  if (tid < kMaxRings) { s_pre_last[tid] = __longlong_as_double(0x7ff8000000000000ll); s_pre_cnt[tid] = 0; s_early[tid] = 0x7fffffff; }
  __syncthreads();
  if (wave == 0) {
    const int pre_end = min(w1, 256);
    for (int g = 0; g < pre_end; g += 64) {
      const int i = g + lane;
      float4 p = make_float4(0, 0, 0, 0);
      int r = -1;
      bool valid = false;
      if (i < pre_end) { p = in[i]; valid = point_valid(p, prm.min_range_sq); if (valid) r = in_ring[i]; }
      const unsigned long long m = same_ring_lanes(valid, r, ring_bits);
      const unsigned long long below = m & ((1ull << lane) - 1ull);
      const double a = valid ? rel_angle(p) : 0.0;
      const int pl = below ? 63 - __clzll((long long)below) : lane;
      double a_prev = __shfl(a, pl);
      if (valid) {
        const int seen = s_pre_cnt[r];                         // every lane of a ring reads before its last lane writes
        bool has_prev = below != 0;
        if (!has_prev) { a_prev = s_pre_last[r]; has_prev = !isnan(a_prev); }
        if ((m >> lane) == 1ull) { s_pre_last[r] = a; s_pre_cnt[r] = seen + __popcll(m); }
        if (has_prev && a < a_prev) atomicMin(&s_early[r], s_off[r] + seen + __popcll(below));
      }
    }
  }
  __syncthreads();
  // ---- pass B: stable split into rings, every wave in its own slice, driver order.  The wrap test of the reference
  //      (`relative_angle < last_relative_angles[ring]`, :145-149: from the first such point on a ring gets +2 pi) compares a
  //      point with its predecessor ON ITS RING, which is the previous same-ring lane of the group, or the last same-ring
  //      point this wave has seen (LDS), or the last one of an earlier wave (fixed up after the pass): the f64 raw angles
  //      never go to memory ----
  for (int g0 = w0; g0 < w1; g0 += 64 * kGroups) {
    float4 pp[kGroups]; int rr[kGroups];
#pragma unroll
    for (int u = 0; u < kGroups; u++) {
      const int i = g0 + 64 * u + lane;
      pp[u] = make_float4(0, 0, 0, 0); rr[u] = 0;
      if (i < w1) { pp[u] = in[i]; rr[u] = in_ring[i]; }
    }
#pragma unroll
    for (int u = 0; u < kGroups; u++) {
    const int g = g0 + 64 * u;
    if (g >= w1) break;
    const int i = g + lane;
    const float4 p = pp[u];
    int r = -1;
    bool valid = false;
    if (i < w1) {
      valid = point_valid(p, prm.min_range_sq);
      if (valid) r = rr[u];
    }
    int dst = 0;
    const unsigned long long m = same_ring_lanes(valid, r, ring_bits);
    const unsigned long long below = m & ((1ull << lane) - 1ull);
    double a = 0.0;
    if (valid) {
      const int cur = s_cur[wave][r];                        // every lane of a ring reads the cursor ...
      dst = cur + __popcll(below);
      if (below == 0) s_cur[wave][r] = cur + __popcll(m);    // ... before its leader advances it
      a = rel_angle(p);
    }
    // predecessor on the ring inside this group: the highest same-ring lane below this one
    const int pl = below ? 63 - __clzll((long long)below) : lane;
    double a_prev = __shfl(a, pl);
    if (valid) {
      bool has_prev = below != 0;
      if (!has_prev) {                                       // leader of its ring in this group: what the wave saw before
        a_prev = s_lasta[wave][r];
        has_prev = !isnan(a_prev);
        if (!has_prev) s_firsta[wave][r] = a;
      }
      if ((m >> lane) == 1ull) s_lasta[wave][r] = a;         // last lane of the ring in this group (after the leader's read)
      if (has_prev && a < a_prev) atomicMin(&s_wrap[r], dst);
      // :151; behind a wrap point the angle carries +2 pi.  No ring id is stored here: a ring's ids are one constant over its
      // output range (filled below with full-width stores; 2-byte stores scattered over 16 ring ranges were written to
      // memory as partial lines, PMC: 0.83 GB written per 1 024 scans for 0.43 GB of output)
      const double aw = dst >= s_early[r] ? a + two_pi : a;
      out_pts[dst] = make_float4(p.x, p.y, p.z, (float)(aw / two_pi * prm.scan_period));
    }
    }
  }
  __syncthreads();
  // predecessor in an earlier wave: the first point of (wave, ring) against the last point of the nearest earlier wave
  // that holds the ring.  The ring's output range is split among the waves in order, so (wave, ring) starts where the
  // previous wave's cursor ended.
  if (tid < kMaxRings) {
    const int r = tid;
    int base = s_off[r];
    bool has_prev = false;
    double last = 0.0;
    for (int w = 0; w < 16; w++) {
      const int end = s_cur[w][r];
      if (end > base) {
        if (has_prev && s_firsta[w][r] < last) atomicMin(&s_wrap[r], base);
        last = s_lasta[w][r];
        has_prev = true;
      }
      base = end;
    }
  }
  __syncthreads();
  // relative time (stored in both `time` and `intensity`, :152-153): from a ring's wrap point on, the +2 pi candidate.  What
  // lies behind an EARLY wrap point has it already (s_wrap == s_early there); a later wrap (the sweep passing the start
  // direction again at the end of the scan) re-derives the angle of the few points behind it from their coordinates,
  // which are stored unchanged, with the same arithmetic as the pass above
  for (int r = wave; r < kMaxRings; r += 16) {
    const int lo = s_wrap[r], hi = min(s_early[r], s_off[r + 1]);
    if (lo >= hi) continue;                                  // no wrap on this ring (lo = INT_MAX), or an early one
    for (int i = lo + lane; i < hi; i += 64) {
      const float4 p = out_pts[i];
      out_pts[i].w = (float)((rel_angle(p) + two_pi) / two_pi * prm.scan_period);
    }
  }
  // ring ids: constant runs
  for (int r = wave; r < kMaxRings; r += 16)
    for (int i = s_off[r] + lane; i < s_off[r + 1]; i += 64) out_ring[i] = (uint16_t)r;
  if (tid == 0) v.n_full[b] = N;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same ring split for a call with FEW scans (the SLAM step hands over one): G workgroups per scan, 16 G slices.
// One 64-beam sweep (~130 k points, two f64 atan2 each) kept the single workgroup above busy for 326 us while 255 compute
// units idled (profiles/r05_slam64_*).  The two slice passes need the populations of ALL slices between them, so the
// split is three launches on the stream instead of a device-wide barrier inside one:
//   extract_prepare_count_kernel    pass A of 16 slices per workgroup -> populations of (slice, ring), per-workgroup ring totals
//   extract_prepare_scatter_kernel  cursors from those, the early-wrap pre-pass (every workgroup repeats it: 256 points), pass B
//   extract_prepare_finish_kernel   wrap test across slices, late-wrap time fix-up and ring ids of the rings r = g (mod G)
// A slice is still a contiguous piece of the driver-order cloud walked by one wavefront with the arithmetic of the kernel
// above, and a wrap point is the ring's smallest output index with a < a_prev whoever finds it: outputs are bit-identical.
// ---------------------------------------------------------------------------------------------------------------------------
struct PrepSplitView {
  int* cnt;            // B x S x kMaxRings: population of (slice, ring)
  int* start;          // B x S x kMaxRings: first output index of (slice, ring)
  double* firsta;      // B x S x kMaxRings: raw angle of the first / last point of (slice, ring); defined where cnt > 0
  double* lasta;
  int* wg_tot;         // B x G x kMaxRings: ring populations of a workgroup's 16 slices
  int* head;           // B x G x 4: first valid point, bad ring seen, highest ring id
  int* wrap;           // B x G x kMaxRings: smallest wrap index found inside the workgroup's slices (INT_MAX: none)
  int* early;          // B x kMaxRings: the early wrap points (INT_MAX: none)
  double* start_ori;   // B
  int G;
};
inline size_t prep_split_bytes(int B, int G) {
  const size_t S = 16 * (size_t)G;
  return (size_t)B * (S * kMaxRings * (2 * sizeof(int) + 2 * sizeof(double)) + (size_t)G * kMaxRings * 2 * sizeof(int) + (size_t)G * 4 * sizeof(int) +
                      kMaxRings * sizeof(int) + sizeof(double)) + 256;
}
inline PrepSplitView prep_split_view(void* base, int B, int G) {
  const size_t S = 16 * (size_t)G;
  PrepSplitView sv;
  char* p = static_cast<char*>(base);
  sv.firsta = reinterpret_cast<double*>(p); p += (size_t)B * S * kMaxRings * sizeof(double);
  sv.lasta = reinterpret_cast<double*>(p); p += (size_t)B * S * kMaxRings * sizeof(double);
  sv.start_ori = reinterpret_cast<double*>(p); p += (size_t)B * sizeof(double);
  sv.cnt = reinterpret_cast<int*>(p); p += (size_t)B * S * kMaxRings * sizeof(int);
  sv.start = reinterpret_cast<int*>(p); p += (size_t)B * S * kMaxRings * sizeof(int);
  sv.wg_tot = reinterpret_cast<int*>(p); p += (size_t)B * G * kMaxRings * sizeof(int);
  sv.wrap = reinterpret_cast<int*>(p); p += (size_t)B * G * kMaxRings * sizeof(int);
  sv.head = reinterpret_cast<int*>(p); p += (size_t)B * G * 4 * sizeof(int);
  sv.early = reinterpret_cast<int*>(p);
  sv.G = G;
  return sv;
}

constexpr int kPrepGroups = 4;           // 64-point groups per iteration with their loads issued up front (as above)

// slice [w0, w1) of wavefront `wave` of workgroup g: slices are multiples of 64 points, so a 64-point group never straddles two
__device__ __forceinline__ void prep_slice_bounds(int n, int S, int sl, int& w0, int& w1) {
  const int slice = ((n + S - 1) / S + 63) & ~63;
  w0 = (int)min((long long)sl * slice, (long long)n); w1 = min(w0 + slice, n);
}

// the verdict of pass A over the whole scan, from the workgroups' reports: [0] first valid point, [1] bad ring, [2] ring-id bits
__device__ __forceinline__ void prep_head_reduce(const int* __restrict__ head, int G, int* s_flag) {
  int f = 0x7fffffff, bd = 0, rm = 0;
  for (int g = 0; g < G; g++) { f = min(f, head[4 * g]); bd |= head[4 * g + 1]; rm = max(rm, head[4 * g + 2]); }
  s_flag[0] = f; s_flag[1] = bd; s_flag[2] = 32 - __clz(rm | 1);
}

__device__ __forceinline__ double prep_rel_angle(const float4 p, double start_ori) {
  // :139-142, see rel_angle in extract_prepare_kernel
  const double two_pi = 2 * 3.14159265358979323846;
  const double ori = -atan2((double)p.y, (double)p.x);
  double a = ori - start_ori + two_pi;
  if (a >= two_pi) a -= two_pi;
  if (a >= two_pi) a -= two_pi;
  return a;
}

__global__ void __launch_bounds__(1024) extract_prepare_count_kernel(ExtractView v, ExtractParams prm, PrepSplitView sv) {
  __shared__ int s_cur[16][kMaxRings];
  __shared__ int s_first[16], s_badw[16], s_rmax[16];
  const int g = blockIdx.x, b = blockIdx.y, G = sv.G, S = 16 * G, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int o = v.off[b];
  const int n = v.off[b + 1] - o;
  const float4* in = v.in_pts + o;
  const uint16_t* in_ring = v.in_ring + o;
  for (int k = tid; k < 16 * kMaxRings; k += 1024) (&s_cur[0][0])[k] = 0;
  __syncthreads();
  int w0, w1;
  prep_slice_bounds(n, S, 16 * g + wave, w0, w1);
  int first = 0x7fffffff, bad = 0, rmax = 0;
  for (int g0 = w0; g0 < w1; g0 += 64 * kPrepGroups) {
    float4 pp[kPrepGroups]; int rr[kPrepGroups];
#pragma unroll
    for (int u = 0; u < kPrepGroups; u++) {
      const int i = g0 + 64 * u + lane;
      pp[u] = make_float4(0, 0, 0, 0); rr[u] = 0;
      if (i < w1) { pp[u] = in[i]; rr[u] = in_ring[i]; }
    }
#pragma unroll
    for (int u = 0; u < kPrepGroups; u++) {
      const int gq = g0 + 64 * u;
      if (gq >= w1) break;
      const int i = gq + lane;
      bool valid = false;
      int r = -1;
      if (i < w1) {
        valid = point_valid(pp[u], prm.min_range_sq);
        if (valid) { r = rr[u]; if (r >= kMaxRings) { bad = 1; valid = false; } }   // CHECK_LT(point.ring, 128), :136
      }
      const unsigned long long any_valid = __ballot(valid);
      if (any_valid && first == 0x7fffffff) first = gq + (__ffsll((long long)any_valid) - 1);
      if (valid) { atomicAdd(&s_cur[wave][r], 1); rmax = max(rmax, r); }
    }
  }
  bad = __any(bad) ? 1 : 0;
#pragma unroll
  for (int o2 = 32; o2 > 0; o2 >>= 1) rmax = max(rmax, __shfl_xor(rmax, o2));
  if (lane == 0) { s_first[wave] = first; s_badw[wave] = bad; s_rmax[wave] = rmax; }
  __syncthreads();
  int* cnt = sv.cnt + ((size_t)b * S + 16 * g) * kMaxRings;
  for (int k = tid; k < 16 * kMaxRings; k += 1024) cnt[k] = (&s_cur[0][0])[k];
  if (tid < kMaxRings) {
    int t = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) t += s_cur[w][tid];
    sv.wg_tot[((size_t)b * G + g) * kMaxRings + tid] = t;
  }
  if (tid == 0) {
    int f = 0x7fffffff, bd = 0, rm = 0;
    for (int w = 0; w < 16; w++) { f = min(f, s_first[w]); bd |= s_badw[w]; rm = max(rm, s_rmax[w]); }
    int* head = sv.head + ((size_t)b * G + g) * 4;
    head[0] = f; head[1] = bd; head[2] = rm; head[3] = 0;
  }
}

__global__ void __launch_bounds__(1024) extract_prepare_scatter_kernel(ExtractView v, ExtractParams prm, PrepSplitView sv) {
  __shared__ int s_off[kMaxRings + 1];
  __shared__ int s_wrap[kMaxRings];
  __shared__ int s_cur[16][kMaxRings];
  __shared__ int s_flag[3];
  __shared__ double s_start_ori;
  __shared__ double s_lasta[16][kMaxRings];
  __shared__ double s_firsta[16][kMaxRings];
  __shared__ double s_pre_last[kMaxRings];
  __shared__ int s_pre_cnt[kMaxRings];
  __shared__ int s_early[kMaxRings];
  const int g = blockIdx.x, b = blockIdx.y, G = sv.G, S = 16 * G, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int o = v.off[b];
  const int n = v.off[b + 1] - o;
  const float4* in = v.in_pts + o;
  const uint16_t* in_ring = v.in_ring + o;
  if (tid == 0) prep_head_reduce(sv.head + (size_t)b * G * 4, G, s_flag);
  for (int k = tid; k < 16 * kMaxRings; k += 1024) (&s_lasta[0][0])[k] = __longlong_as_double(0x7ff8000000000000ll);
  // ring totals of the scan and what the workgroups in front of this one hold of each ring: at most 16 + 16 independent loads
  if (tid < kMaxRings) {
    int before = 0, total = 0;
    for (int q = 0; q < G; q++) { const int t = sv.wg_tot[((size_t)b * G + q) * kMaxRings + tid]; if (q < g) before += t; total += t; }
    const int* cnt = sv.cnt + ((size_t)b * S + 16 * g) * kMaxRings + tid;
    int c[16];
#pragma unroll
    for (int w = 0; w < 16; w++) c[w] = cnt[w * kMaxRings];
#pragma unroll
    for (int w = 0; w < 16; w++) { s_cur[w][tid] = before; before += c[w]; }
    s_wrap[tid] = total;                                      // borrowed as the ring total
  }
  __syncthreads();
  if (s_flag[1] || s_flag[0] == 0x7fffffff) {
    if (g == 0) {
      if (tid == 0) {
        v.status[b] = s_flag[1] ? 5 /*MSFL_BAD_RING*/ : 3 /*MSFL_BAD_ARG: empty valid cloud, CHECK :186,:200*/;
        v.n_full[b] = 0; v.n_sharp[b] = 0; v.n_less_sharp[b] = 0; v.n_flat[b] = 0; v.n_less_flat[b] = 0;
      }
      if (tid <= kMaxRings) v.ring_tab[b * (kMaxRings + 1) + tid] = 0;
    }
    return;
  }
  if (tid == 0) {
    int run = 0;
    for (int r = 0; r < kMaxRings; r++) { s_off[r] = run; run += s_wrap[r]; }
    s_off[kMaxRings] = run;
    const float4 f = in[s_flag[0]];
    s_start_ori = -atan2((double)f.y, (double)f.x);       // :131
  }
  __syncthreads();
  if (g == 0 && tid <= kMaxRings) v.ring_tab[b * (kMaxRings + 1) + tid] = s_off[tid];
  if (tid < kMaxRings) {
    int* start = sv.start + ((size_t)b * S + 16 * g) * kMaxRings + tid;
#pragma unroll
    for (int w = 0; w < 16; w++) { const int c = s_cur[w][tid] + s_off[tid]; s_cur[w][tid] = c; start[w * kMaxRings] = c; }
    s_wrap[tid] = 0x7fffffff;
    s_pre_last[tid] = __longlong_as_double(0x7ff8000000000000ll); s_pre_cnt[tid] = 0; s_early[tid] = 0x7fffffff;
  }
  __syncthreads();
  const int ring_bits = __builtin_amdgcn_readfirstlane(s_flag[2]);
  const double start_ori = s_start_ori;
  const double two_pi = 2 * 3.14159265358979323846;
  float4* out_pts = v.full_pts + o;
  // ---- early wraps over the first 256 points of the CLOUD (see the kernel above; any prefix gives the same outputs) ----
  if (wave == 0) {
    const int pre_end = min(n, 256);
    for (int gq = 0; gq < pre_end; gq += 64) {
      const int i = gq + lane;
      float4 p = make_float4(0, 0, 0, 0);
      int r = -1;
      bool valid = false;
      if (i < pre_end) { p = in[i]; valid = point_valid(p, prm.min_range_sq); if (valid) r = in_ring[i]; }
      const unsigned long long m = same_ring_lanes(valid, r, ring_bits);
      const unsigned long long below = m & ((1ull << lane) - 1ull);
      const double a = valid ? prep_rel_angle(p, start_ori) : 0.0;
      const int pl = below ? 63 - __clzll((long long)below) : lane;
      double a_prev = __shfl(a, pl);
      if (valid) {
        const int seen = s_pre_cnt[r];
        bool has_prev = below != 0;
        if (!has_prev) { a_prev = s_pre_last[r]; has_prev = !isnan(a_prev); }
        if ((m >> lane) == 1ull) { s_pre_last[r] = a; s_pre_cnt[r] = seen + __popcll(m); }
        if (has_prev && a < a_prev) atomicMin(&s_early[r], s_off[r] + seen + __popcll(below));
      }
    }
  }
  __syncthreads();
  // ---- pass B over this wavefront's slice ----
  int w0, w1;
  prep_slice_bounds(n, S, 16 * g + wave, w0, w1);
  for (int g0 = w0; g0 < w1; g0 += 64 * kPrepGroups) {
    float4 pp[kPrepGroups]; int rr[kPrepGroups];
#pragma unroll
    for (int u = 0; u < kPrepGroups; u++) {
      const int i = g0 + 64 * u + lane;
      pp[u] = make_float4(0, 0, 0, 0); rr[u] = 0;
      if (i < w1) { pp[u] = in[i]; rr[u] = in_ring[i]; }
    }
#pragma unroll
    for (int u = 0; u < kPrepGroups; u++) {
      const int gq = g0 + 64 * u;
      if (gq >= w1) break;
      const int i = gq + lane;
      const float4 p = pp[u];
      int r = -1;
      bool valid = false;
      if (i < w1) {
        valid = point_valid(p, prm.min_range_sq);
        if (valid) r = rr[u];
      }
      int dst = 0;
      const unsigned long long m = same_ring_lanes(valid, r, ring_bits);
      const unsigned long long below = m & ((1ull << lane) - 1ull);
      double a = 0.0;
      if (valid) {
        const int cur = s_cur[wave][r];
        dst = cur + __popcll(below);
        if (below == 0) s_cur[wave][r] = cur + __popcll(m);
        a = prep_rel_angle(p, start_ori);
      }
      const int pl = below ? 63 - __clzll((long long)below) : lane;
      double a_prev = __shfl(a, pl);
      if (valid) {
        bool has_prev = below != 0;
        if (!has_prev) {
          a_prev = s_lasta[wave][r];
          has_prev = !isnan(a_prev);
          if (!has_prev) s_firsta[wave][r] = a;
        }
        if ((m >> lane) == 1ull) s_lasta[wave][r] = a;
        if (has_prev && a < a_prev) atomicMin(&s_wrap[r], dst);
        const double aw = dst >= s_early[r] ? a + two_pi : a;
        out_pts[dst] = make_float4(p.x, p.y, p.z, (float)(aw / two_pi * prm.scan_period));
      }
    }
  }
  __syncthreads();
  double* firsta = sv.firsta + ((size_t)b * S + 16 * g) * kMaxRings;
  double* lasta = sv.lasta + ((size_t)b * S + 16 * g) * kMaxRings;
  for (int k = tid; k < 16 * kMaxRings; k += 1024) { firsta[k] = (&s_firsta[0][0])[k]; lasta[k] = (&s_lasta[0][0])[k]; }
  if (tid < kMaxRings) {
    sv.wrap[((size_t)b * G + g) * kMaxRings + tid] = s_wrap[tid];
    if (g == 0) sv.early[(size_t)b * kMaxRings + tid] = s_early[tid];
  }
  if (g == 0 && tid == 0) sv.start_ori[b] = start_ori;
}

__global__ void __launch_bounds__(1024) extract_prepare_finish_kernel(ExtractView v, ExtractParams prm, PrepSplitView sv) {
  __shared__ int s_cnt[16 * kMaxRings];          // populations of the owned rings: [j][s], j-th owned ring, S slices (S x 128 / G = 2 048 entries)
  __shared__ int s_wrap[kMaxRings];              // per owned ring j
  __shared__ int s_flag[3];
  const int g = blockIdx.x, b = blockIdx.y, G = sv.G, S = 16 * G, own = kMaxRings / G, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int o = v.off[b];
  if (tid == 0) prep_head_reduce(sv.head + (size_t)b * G * 4, G, s_flag);
  __syncthreads();
  if (s_flag[1] || s_flag[0] == 0x7fffffff) return;          // the scatter kernel reported it
  const int* cnt = sv.cnt + (size_t)b * S * kMaxRings;
  for (int k = tid; k < own * S; k += 1024) { const int j = k / S, sl = k - j * S; s_cnt[k] = cnt[sl * kMaxRings + g + G * j]; }
  if (tid < own) {
    int w = 0x7fffffff;
    for (int q = 0; q < G; q++) w = min(w, sv.wrap[((size_t)b * G + q) * kMaxRings + g + G * tid]);
    s_wrap[tid] = w;
  }
  __syncthreads();
  // predecessor in an earlier slice: the first point of (slice, ring) against the last point of the nearest earlier slice that holds the ring
  const double* firsta = sv.firsta + (size_t)b * S * kMaxRings;
  const double* lasta = sv.lasta + (size_t)b * S * kMaxRings;
  const int* start = sv.start + (size_t)b * S * kMaxRings;
  for (int k = tid; k < own * S; k += 1024) {
    const int j = k / S, sl = k - j * S, r = g + G * j;
    if (s_cnt[k] == 0) continue;
    int p = sl - 1;
    while (p >= 0 && s_cnt[j * S + p] == 0) p--;
    if (p >= 0 && firsta[sl * kMaxRings + r] < lasta[p * kMaxRings + r]) atomicMin(&s_wrap[j], start[sl * kMaxRings + r]);
  }
  __syncthreads();
  const int* tab = v.ring_tab + b * (kMaxRings + 1);
  const double start_ori = sv.start_ori[b];
  const double two_pi = 2 * 3.14159265358979323846;
  float4* out_pts = v.full_pts + o;
  uint16_t* out_ring = v.full_ring + o;
  for (int j = wave; j < own; j += 16) {
    const int r = g + G * j, r0 = tab[r], r1 = tab[r + 1];
    const int lo = s_wrap[j], hi = min(sv.early[(size_t)b * kMaxRings + r], r1);
    if (lo < hi) {                                           // a late wrap: the +2 pi time for what lies behind it (lo = INT_MAX: none)
      for (int i = lo + lane; i < hi; i += 64) {
        const float4 p = out_pts[i];
        out_pts[i].w = (float)((prep_rel_angle(p, start_ori) + two_pi) / two_pi * prm.scan_period);
      }
    }
    for (int i = r0 + lane; i < r1; i += 64) out_ring[i] = (uint16_t)r;
  }
  if (g == 0 && tid == 0) v.n_full[b] = tab[kMaxRings];
}

__device__ __forceinline__ int find_scan_off(const int* __restrict__ off, int n_scans, int g) {
  int lo = 0, hi = n_scans;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (off[mid] <= g) lo = mid; else hi = mid;
  }
  return lo;
}

// consecutive g across a wavefront: one scalar binary search for the first lane, then a forward step
__device__ __forceinline__ int find_scan_off_wave(const int* __restrict__ off, int n_scans, int g) {
  const int g0 = __builtin_amdgcn_readfirstlane(g);
  int b = __builtin_amdgcn_readfirstlane(find_scan_off(off, n_scans, g0));
  while (b + 1 < n_scans && g >= off[b + 1]) b++;
  return b;
}

// 256 consecutive points of the concatenated batch per workgroup, staged once into an LDS tile with a five-point halo on
// either side: every point is fetched from global memory once instead of twelve times (its own load, the eleven-point
// window of its neighbours and the gap test), which had the kernel waiting on the vector L1 (111 M cache accesses per
// 1 024 scans for 24 M points).  Halo slots outside the batch stay unread: the margin test keeps the window inside the scan.
__global__ void __launch_bounds__(256) extract_curvature_kernel(ExtractView v, ExtractParams prm) {
  __shared__ float4 s_p[256 + 10];
  // blockIdx.y = scan, blockIdx.x = a 256-point tile of THAT scan (grid.x = the longest scan's tile count; surplus workgroups leave
  // at once).  Launched 1-D over the concatenated batch, every wavefront had to binary-search its scan in the offset table first: ten
  // dependent scalar loads in front of one point load and ~40 instructions of work (measured on the same pattern in the pairs
  // index build: 205 -> 17 us, docs/kernels/pairs.md).
  const int b = blockIdx.y, tid = threadIdx.x;
  const int o = v.off[b];
  if ((int)blockIdx.x * 256 >= v.off[b + 1] - o) return;
  const int g0 = o + (int)blockIdx.x * 256;
  const int g = g0 + tid;
  if (g < v.n_total) s_p[tid + 5] = v.full_pts[g];
  if (tid < 10) {
    const int h = tid < 5 ? g0 - 5 + tid : g0 + 256 + (tid - 5);
    if (h >= 0 && h < v.n_total) s_p[tid < 5 ? tid : 256 + tid] = v.full_pts[h];
  }
  __syncthreads();
  if (g >= v.n_total) return;
  const int i = g - o;
  const int N = v.n_full[b];
  if (i >= N) return;
  const float4* c = s_p + tid + 5;                          // c[k] = point i + k of this scan, |k| <= 5
  float curv = 0.f;
  if (i >= 5 && i < N - 5) {
    // f32 sums in source order (:214-234), f64 squares, f32 store (:236)
    const float4 m5 = c[-5], m4 = c[-4], m3 = c[-3], m2 = c[-2], m1 = c[-1], p0 = c[0];
    const float4 p1 = c[1], p2 = c[2], p3 = c[3], p4 = c[4], p5 = c[5];
    const float dx = m5.x + m4.x + m3.x + m2.x + m1.x - 10 * p0.x + p1.x + p2.x + p3.x + p4.x + p5.x;
    const float dy = m5.y + m4.y + m3.y + m2.y + m1.y - 10 * p0.y + p1.y + p2.y + p3.y + p4.y + p5.y;
    const float dz = m5.z + m4.z + m3.z + m2.z + m1.z - 10 * p0.z + p1.z + p2.z + p3.z + p4.z + p5.z;
    const double X = dx, Y = dy, Z = dz;
    curv = (float)(X * X + Y * Y + Z * Z);
  }
  v.curvature[g] = curv;
  v.label[g] = 0;
  uint8_t gp = 1;
  if (i + 1 < N) {
    const float4 a = c[1], q = c[0];
    const float ex = a.x - q.x, ey = a.y - q.y, ez = a.z - q.z;
    const float s = ex * ex + ey * ey + ez * ez;           // Vector3f squaredNorm
    gp = ((double)s > prm.neighbor_gap_sq) ? 1 : 0;        // :293,300,326,332
  }
  v.gap[g] = gp;
}

// ---- wave-wide extremum of a 64-bit key on the DPP network (no LDS crossbar): six exchange steps, every lane active ----
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long k) {
  const int lo = __builtin_amdgcn_update_dpp((int)(unsigned)k, (int)(unsigned)k, CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(unsigned)(k >> 32), (int)(unsigned)(k >> 32), CTRL, ROW_MASK, 0xf, false);
  return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}
template <bool MAX>
__device__ __forceinline__ unsigned long long wave_extremum_u64(unsigned long long k) {
  auto pick = [](unsigned long long a, unsigned long long c) { return MAX ? (a > c ? a : c) : (a < c ? a : c); };
  k = pick(k, dpp_u64<0xb1, 0xf>(k));      // quad_perm [1,0,3,2]
  k = pick(k, dpp_u64<0x4e, 0xf>(k));      // quad_perm [2,3,0,1]
  k = pick(k, dpp_u64<0x141, 0xf>(k));     // row_half_mirror
  k = pick(k, dpp_u64<0x140, 0xf>(k));     // row_mirror
  k = pick(k, dpp_u64<0x142, 0xa>(k));     // row_bcast:15 -> rows 1, 3
  k = pick(k, dpp_u64<0x143, 0xc>(k));     // row_bcast:31 -> rows 2, 3
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)k, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(k >> 32), 63);
  return ((unsigned long long)hi << 32) | lo;
}

struct RingBits {
  unsigned int* w;   // LDS words (one spare word past the ring so that 64-bit windows never run out)
  __device__ __forceinline__ bool get(int i) const { return (w[i >> 5] >> (i & 31)) & 1u; }
  __device__ __forceinline__ void set(int i) { w[i >> 5] |= (1u << (i & 31)); }
  // bits [i, i+32) as one word
  __device__ __forceinline__ unsigned int window(int i) const {
    const unsigned long long two = (unsigned long long)w[i >> 5] | ((unsigned long long)w[(i >> 5) + 1] << 32);
    return (unsigned int)(two >> (i & 31));
  }
  // OR `bits` (<= 32 of them) in at position i
  __device__ __forceinline__ void or_window(int i, unsigned int bits) {
    const unsigned long long m = (unsigned long long)bits << (i & 31);
    w[i >> 5] |= (unsigned int)m;
    const unsigned int hi = (unsigned int)(m >> 32);
    if (hi) w[(i >> 5) + 1] |= hi;
  }
};

// Neighbour suppression of one pick at ring-local position q (msf_loam_node.cc:290-304 / :322-335):
// forward l = 1..5 while the gap between q+l-1 and q+l is small, backward l = -1..-5 likewise.
// Returns the marked span [q - back, q + fwd] as counts; bit arithmetic on one 32-bit window of
// the gap mask instead of ten dependent LDS round trips.
__device__ __forceinline__ void neighbour_span(const RingBits& gap, int q, int& back, int& fwd) {
  const unsigned int g = gap.window(q - 5);          // bit t <-> gap between (q-5+t) and (q-5+t+1)
  const unsigned int f5 = (g >> 5) & 31u;            // gaps q..q+4
  fwd = __ffs((int)(f5 | 32u)) - 1;                  // first set gap stops the run (0..5)
  const unsigned int b5 = g & 31u;                   // gaps q-5..q-1 ; walk down from q-1
  back = __clz((int)((b5 << 27) | (1u << 26)));      // leading zeros of the 5-bit field (0..5)
}

// wave-wide extremum of a 32-bit word, same six DPP steps as wave_extremum_u64
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_u32(unsigned k) { return (unsigned)__builtin_amdgcn_update_dpp((int)k, (int)k, CTRL, ROW_MASK, 0xf, false); }
// every lane takes the value of the lane above it (lane 63: 0)
__device__ __forceinline__ float dpp_wave_shl1(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x130 /* wave_shl:1 */, 0xf, 0xf, true));
}
template <bool MAX>
__device__ __forceinline__ unsigned wave_extremum_u32(unsigned k) {
  auto pick = [](unsigned a, unsigned c) { return MAX ? (a > c ? a : c) : (a < c ? a : c); };
  k = pick(k, dpp_u32<0xb1, 0xf>(k));
  k = pick(k, dpp_u32<0x4e, 0xf>(k));
  k = pick(k, dpp_u32<0x141, 0xf>(k));
  k = pick(k, dpp_u32<0x140, 0xf>(k));
  k = pick(k, dpp_u32<0x142, 0xa>(k));
  k = pick(k, dpp_u32<0x143, 0xc>(k));
  return (unsigned)__builtin_amdgcn_readlane((int)k, 63);
}

// One wavefront per (scan, ring).  The reference sorts every sector by curvature (std::sort, :263-267) and walks the
// sorted order: descending for the <= 20 corner picks, ascending for the <= 4 flat picks, skipping points that an
// earlier pick has suppressed.  Walking a sorted order and skipping the suppressed ones IS "take the largest (smallest)
// not yet suppressed key", so no sort is needed.  The sector's curvatures sit in an LDS array of f32 bit patterns
// (curvature >= 0: the u32 order is the f32 order), written once per pass from the lanes' registers: for the corner pass
// a word is 0 when the point is not above the threshold or suppressed, for the flat pass ~0 when it is not below it or
// suppressed (one array, not two: LDS is what limits the wavefronts per SIMD here, and the kernel is a latency chain).
// A pick is: every
// lane takes the extremum of its <= 8 words (position sp + lane + 64 t, so ties inside a lane resolve by t), one
// wave-wide DPP extremum, a ballot for the lane that holds it (several lanes with the same curvature bits: a second
// reduction over their positions), and the <= 11 suppressed positions are overwritten by their lanes.  Ties order like
// the sort's keys (curvature bits << 32 | index): the corner pass takes the highest index, the flat pass the lowest.
// Sectors with more than 512 points (a VLP-16 sector has 298) re-read curvature and the suppression mask per pick.
// Round 1 had a bitonic sort kernel in front of this one (0.33 ms per 1 024 scans, LDS bound) plus the key round trip.
constexpr int kPickCand = 8;
constexpr int kPickSector = 64 * kPickCand;

__global__ void __launch_bounds__(64 * kExWaves) extract_pick_kernel(ExtractView v, ExtractParams prm) {
  __shared__ unsigned int s_picked[kExWaves][kRingCapacity / 32 + 2];
  __shared__ unsigned int s_corner[kExWaves][kRingCapacity / 32 + 2];
  __shared__ unsigned int s_gap[kExWaves][kRingCapacity / 32 + 2];
  __shared__ unsigned int s_cand[kExWaves][kPickSector];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x;                               // scans on the FAST block index: only the first few ring groups of a scan
                                                          // have work (16 of 128 rings on a VLP-16); with groups on the fast index the busy
                                                          // workgroups shared a residue mod 32 and landed on 32 of the 256 CUs
  const int r = blockIdx.y * kExWaves + wave;
  if (r >= kMaxRings) return;
  int* cnt_out = v.ring_cnt + ((size_t)b * kMaxRings + r) * 4;
  const int* tab = v.ring_tab + b * (kMaxRings + 1);
  // the ring's bounds are the same in every lane: say so (readfirstlane), and the sector bounds, the pick loops' trip
  // counts and the picked positions all live in scalar registers with scalar branches around them
  const int s = __builtin_amdgcn_readfirstlane(tab[r]), len = __builtin_amdgcn_readfirstlane(tab[r + 1]) - s;
  const int o = __builtin_amdgcn_readfirstlane(v.off[b]);
  int n_sharp = 0, n_ls = 0, n_flat = 0, n_lf = 0;
  const int start = s + 5, end = s + len - 6;                               // :192-194
  const bool active = (__builtin_amdgcn_readfirstlane(v.status[b]) == 0) && len > 0 && (end - start >= 6);  // :252
  if (!active) {
    if (lane == 0) { cnt_out[0] = 0; cnt_out[1] = 0; cnt_out[2] = 0; cnt_out[3] = 0; }
    return;
  }
  if (len > kRingCapacity) {
    if (lane == 0) { v.status[b] = 7 /*MSFL_CAPACITY*/; cnt_out[0] = cnt_out[1] = cnt_out[2] = cnt_out[3] = 0; }
    return;
  }
  uint8_t* label = v.label + o;
  const uint8_t* gapb = v.gap + o;
  const float* curv = v.curvature + o;
  int* t_sharp = v.tmp_idx + 0 * (size_t)v.n_total + o + s;
  int* t_ls = v.tmp_idx + 1 * (size_t)v.n_total + o + s;
  int* t_flat = v.tmp_idx + 2 * (size_t)v.n_total + o + s;
  int* t_lf = v.tmp_idx + 3 * (size_t)v.n_total + o + s;
  RingBits picked{s_picked[wave]}, corner{s_corner[wave]}, gap{s_gap[wave]};
  unsigned int* cand = s_cand[wave];
  auto wave_sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // ring-local bitmasks: bit k <-> scan-local index s + k
  for (int w0 = 0; w0 < len + 64; w0 += 64) {       // one extra round: the spare words behind the ring
    const int k = w0 + lane;
    const bool gbit = (k < len) ? (gapb[s + k] != 0) : true;
    const unsigned long long m = __ballot(gbit);
    if (lane == 0) {
      gap.w[(w0 >> 5)] = (unsigned int)m; gap.w[(w0 >> 5) + 1] = (unsigned int)(m >> 32);
      picked.w[(w0 >> 5)] = 0; picked.w[(w0 >> 5) + 1] = 0;
      corner.w[(w0 >> 5)] = 0; corner.w[(w0 >> 5) + 1] = 0;
    }
  }
  wave_sync();
  for (int j = 0; j < prm.sectors; j++) {
    const int sp = start + (end - start) * j / prm.sectors;                 // :256-259
    const int ep = start + (end - start) * (j + 1) / prm.sectors - 1;
    const int cnt = ep - sp + 1;
    if (cnt <= 0) continue;
    const bool in_lds = cnt <= kPickSector;
    // thresholds compared as doubles (:275 / :312); points an earlier pick has suppressed are dead.  All eight loads are
    // issued before the first is used (clamped index, no branch around them)
    float cv[kPickCand];
    auto stage = [&](auto corner_pass_t) __attribute__((always_inline)) {
      constexpr bool kCorner = decltype(corner_pass_t)::value;
#pragma unroll
      for (int t = 0; t < kPickCand; t++) {
        const int k = lane + 64 * t, pos = sp + k;
        const bool live = pos <= ep && !picked.get(min(pos, ep) - s);
        const float c = cv[t];
        if (kCorner) cand[k] = (live && (double)c > prm.curvature_threshold) ? __float_as_uint(c) : 0u;
        else cand[k] = (live && (double)c < prm.curvature_threshold) ? __float_as_uint(c) : ~0u;
      }
      wave_sync();
    };
    if (in_lds) {
#pragma unroll
      for (int t = 0; t < kPickCand; t++) cv[t] = curv[min(sp + lane + 64 * t, ep)];
      stage(std::true_type{});
    }
    // The arg-max (corner pass) / arg-min (flat pass) of the live candidates as a scan-local position, -1 = none left.
    auto select = [&](auto corner_pass_t) __attribute__((always_inline)) -> int {
      constexpr bool kCorner = decltype(corner_pass_t)::value;
      if (in_lds) {
        const unsigned int* c = cand;
        unsigned int best = kCorner ? 0u : ~0u, tb = 0;
#pragma unroll
        for (int t = 0; t < kPickCand; t++) {
          const unsigned int x = c[lane + 64 * t];
          const bool take = kCorner ? x >= best : x < best;          // ties inside a lane: highest t (corner) / lowest t (flat)
          best = take ? x : best; tb = take ? (unsigned)t : tb;
        }
        const unsigned int ext = wave_extremum_u32<kCorner>(best);
        if (ext == (kCorner ? 0u : ~0u)) return -1;
        const unsigned long long holders = __ballot(best == ext);
        const unsigned int k = (unsigned)lane + 64u * tb;
        if (__popcll(holders) == 1) return sp + __builtin_amdgcn_readlane((int)k, __ffsll((long long)holders) - 1);
        return sp + (int)wave_extremum_u32<kCorner>(best == ext ? k : (kCorner ? 0u : ~0u));
      }
      unsigned long long best = kCorner ? 0ull : ~0ull;
      for (int pos = sp + lane; pos <= ep; pos += 64) {
        const float c = curv[pos];
        const bool q = kCorner ? ((double)c > prm.curvature_threshold) : ((double)c < prm.curvature_threshold);
        if (!q || picked.get(pos - s)) continue;
        const unsigned long long key = ((unsigned long long)__float_as_uint(c) << 32) | (unsigned int)pos;
        if (kCorner) best = key > best ? key : best; else best = key < best ? key : best;
      }
      best = wave_extremum_u64<kCorner>(best);
      return best == (kCorner ? 0ull : ~0ull) ? -1 : (int)(unsigned int)best;
    };
    // scan-local positions [a, a + n) have just been suppressed (n <= 11)
    auto suppress = [&](int a, int n, unsigned int dead) __attribute__((always_inline)) {
      const int pos = a + lane;
      if (in_lds && lane < n && pos >= sp && pos <= ep) cand[pos - sp] = dead;
    };
    // corner picks, descending curvature (:272-305)
    int largest = 0;
    for (;;) {
      const int pind = select(std::true_type{});
      if (pind < 0) break;                                               // nothing unsuppressed above the threshold is left
      const int pq = pind - s;
      largest++;
      if (largest > prm.max_less_sharp) break;                           // :283-285 (no marking for this one)
      int back, fwd;
      neighbour_span(gap, pq, back, fwd);                  // runs of |p[i+1]-p[i]|^2 <= 0.05 around the pick
      if (lane == 0) {
        if (largest <= prm.max_sharp) { label[pind] = 1; t_sharp[n_sharp] = pind; t_ls[n_ls] = pind; }
        else { label[pind] = 2; t_ls[n_ls] = pind; }
        const unsigned int span = (2u << (back + fwd)) - 1u;   // back + fwd + 1 ones
        picked.or_window(pq - back, span); corner.or_window(pq - back, span);
      }
      if (largest <= prm.max_sharp) n_sharp++;
      n_ls++;
      if (lane >= 1 && lane <= back) label[pind - lane] = 2;   // relabel the neighbours (:295, :302)
      if (lane >= 1 && lane <= fwd) label[pind + lane] = 2;
      suppress(pind - back, back + fwd + 1, 0u);
      wave_sync();
    }
    // flat picks, ascending curvature (:307-336)
    if (in_lds) stage(std::false_type{});
    int smallest = 0;
    for (;;) {
      const int pind = select(std::false_type{});
      if (pind < 0) break;
      const int pq = pind - s;
      if (lane == 0) { label[pind] = 3; t_flat[n_flat] = pind; }
      n_flat++;
      smallest++;
      if (smallest >= prm.max_flat) break;                     // before neighbour marking, :317-319
      int back, fwd;
      neighbour_span(gap, pq, back, fwd);
      if (lane == 0) picked.or_window(pq - back, (2u << (back + fwd)) - 1u);
      suppress(pind - back, back + fwd + 1, ~0u);
      wave_sync();
    }
    wave_sync();
    // less-flat = positions of this sector not labelled SHARP / LESS_SHARP so far (:338-344)
    for (int k0 = sp; k0 <= ep; k0 += 64) {
      const int k = k0 + lane;
      const bool keep = (k <= ep) && !corner.get(k - s);
      const unsigned long long m = __ballot(keep);
      if (keep) t_lf[n_lf + __popcll(m & ((1ull << lane) - 1ull))] = k;
      n_lf += __popcll(m);
    }
  }
  if (lane == 0) { cnt_out[0] = n_sharp; cnt_out[1] = n_ls; cnt_out[2] = n_flat; cnt_out[3] = n_lf; }
}

constexpr int kCompactThreads = 1024;     // a less-flat list is ~20 k entries per scan: 256 threads walked it in 80 dependent rounds

__global__ void __launch_bounds__(kCompactThreads) extract_compact_kernel(ExtractView v, const double* __restrict__ extrinsic) {
  static_assert(kMaxRings == 128, "two rings per lane in the prefix below");
  __shared__ int s_pref[4][kMaxRings + 1];
  __shared__ int s_tab[kMaxRings + 1];
  // grid (G, B): the G workgroups of a scan take interleaved 4 096-entry pieces of its lists (G > 1 when the call holds few scans:
  // one workgroup walked a 64-beam sweep's 100 k less-flat entries in 80 us); each repeats the 4 x 128 prefix, which is nothing
  const int g = blockIdx.x, G = gridDim.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int o = v.off[b];
  const int* cnt = v.ring_cnt + (size_t)b * kMaxRings * 4;
  const int* tab = v.ring_tab + b * (kMaxRings + 1);
  const bool ok = (v.status[b] == 0);
  if (tid >= 512 && tid - 512 <= kMaxRings) s_tab[tid - 512] = tab[tid - 512];
  if (wave < 4) {                                  // list `wave`: exclusive prefix of its 128 per-ring counts, two rings per lane
    const int c0 = ok ? cnt[(2 * lane) * 4 + wave] : 0, c1 = ok ? cnt[(2 * lane + 1) * 4 + wave] : 0;
    int incl = c0 + c1;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(incl, d); if (lane >= d) incl += u; }
    s_pref[wave][2 * lane] = incl - c0 - c1;
    s_pref[wave][2 * lane + 1] = incl - c1;
    if (lane == 63) s_pref[wave][kMaxRings] = incl;
  }
  __syncthreads();
  if (tid == 0 && g == 0) {
    v.n_sharp[b] = s_pref[0][kMaxRings]; v.n_less_sharp[b] = s_pref[1][kMaxRings];
    v.n_flat[b] = s_pref[2][kMaxRings]; v.n_less_flat[b] = s_pref[3][kMaxRings];
    if (!ok) v.n_full[b] = (v.status[b] == 7) ? v.n_full[b] : 0;
  }
  if (!ok) return;
  int* outs[4] = {v.sharp_idx + o, v.less_sharp_idx + o, v.flat_idx + o, v.less_flat_idx + o};
#pragma unroll
  for (int L = 0; L < 4; L++) {
    const int* tmp = v.tmp_idx + (size_t)L * v.n_total + o;
    // one flat loop over the output positions; the owning ring comes from a 7-step search in the LDS prefix
    // (instead of 128 per-ring loops, most of them over empty or 2-entry lists)
    const int total = s_pref[L][kMaxRings];
    // four entries per thread and round: the searches, then the four loads together, then the stores (one entry per
    // round left every round waiting on its own load)
    for (int k0 = g * 4 * kCompactThreads + tid; k0 < total; k0 += G * 4 * kCompactThreads) {
      int src[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int k = min(k0 + u * kCompactThreads, total - 1);
        int lo = 0, hi = kMaxRings;                    // largest r with s_pref[L][r] <= k and a non-empty list
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_pref[L][mid] <= k) lo = mid; else hi = mid; }
        src[u] = s_tab[lo] + (k - s_pref[L][lo]);
      }
      int val[4];
#pragma unroll
      for (int u = 0; u < 4; u++) val[u] = tmp[src[u]];
#pragma unroll
      for (int u = 0; u < 4; u++) { const int k = k0 + u * kCompactThreads; if (k < total) outs[L][k] = val[u]; }
    }
  }
  // TransformPointCloudInPlace x5 (:367-371): the five clouds are gathers of the full cloud
  if (extrinsic) {
    const pose7 T = load_pose(extrinsic);
    const int N = v.n_full[b];
    float4* c = v.full_pts + o;
    for (int i = g * kCompactThreads + tid; i < N; i += G * kCompactThreads) {
      const float4 p = c[i];
      const float3 t = transform_point_f32(T, p.x, p.y, p.z);
      c[i] = make_float4(t.x, t.y, t.z, p.w);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// pcl::VoxelGrid centroid filter (laser_mapping.cc:264-270) — caller-side helper, SURVEY.md §8f N2
// ---------------------------------------------------------------------------------------------
// ---- batched voxel filter: B clouds in one pass (pcl::VoxelGrid per cloud, laser_mapping.cc:264-270) ----
// Cloud b = count[b] points pts[off[b] + (idx ? idx[off[b] + k] : k)], k < count[b]: the index form
// reads feature lists straight out of the extraction's output (no gather pass).  Per-cloud bounding
// box -> (cloud, voxel) keys -> one stable radix sort over the whole batch -> one centroid per key
// run, accumulated in arrival order with f32 accumulators like the single-cloud kernels above.
struct VoxelBatchView {
  const float4* pts; const int* idx; const int* off; const int* count; int n_clouds; int n_total;
  float inv_leaf;
};
struct VoxelCloudDesc { int min_b[3]; int div_b[3]; int n; int bad; };

__device__ __forceinline__ float4 vb_point(const VoxelBatchView& v, int b, int k) {
  const int o = v.off[b];
  return v.pts[o + (v.idx ? v.idx[o + k] : k)];
}

// also: n_valid[b] (input to the slot scan) and max_cells[0] = the largest voxel count of any cloud (sort key width)
constexpr int kVoxDescThreads = 1024;       // one workgroup per cloud: a 64-beam less-flat list is ~100 k points
__global__ void __launch_bounds__(kVoxDescThreads) voxel_batch_desc_kernel(VoxelBatchView v, VoxelCloudDesc* __restrict__ desc, int* __restrict__ n_valid,
                                                                int* __restrict__ max_cells) {
  __shared__ float s_mn[kVoxDescThreads / 64][3], s_mx[kVoxDescThreads / 64][3];
  const int b = blockIdx.x;
  const int cap = v.off[b + 1] - v.off[b];
  const int n = v.count ? min(max(v.count[b], 0), cap) : cap;
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  __shared__ int s_nonfinite;
  if (threadIdx.x == 0) s_nonfinite = 0;
  __syncthreads();
  for (int k = threadIdx.x; k < n; k += kVoxDescThreads) {
    const float4 p = vb_point(v, b, k);
    if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
      mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
      mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
    } else {
      s_nonfinite = 1;                        // refused below: pcl::VoxelGrid would drop the point, the reference never feeds it one
    }
  }
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn[a] = fminf(mn[a], __shfl_xor(mn[a], o)); mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o)); }
  if ((threadIdx.x & 63) == 0)
    for (int a = 0; a < 3; a++) { s_mn[threadIdx.x >> 6][a] = mn[a]; s_mx[threadIdx.x >> 6][a] = mx[a]; }
  __syncthreads();
  if (threadIdx.x == 0) {
    VoxelCloudDesc d;
    d.n = n; d.bad = 0;
    long long cells = 1;
    for (int a = 0; a < 3; a++) {
      float lo = s_mn[0][a], hi = s_mx[0][a];
      for (int w = 1; w < kVoxDescThreads / 64; w++) { lo = fminf(lo, s_mn[w][a]); hi = fmaxf(hi, s_mx[w][a]); }
      if (!(lo <= hi)) { d.min_b[a] = 0; d.div_b[a] = 1; if (n > 0) d.bad = 2; continue; }       // no finite point
      d.min_b[a] = (int)floorf(lo * v.inv_leaf);
      d.div_b[a] = (int)floorf(hi * v.inv_leaf) - d.min_b[a] + 1;
      cells *= d.div_b[a];
      if (cells > 0x7fffffffLL) d.bad = 1;            // leaf too small for the extent (PCL would skip filtering)
    }
    if (s_nonfinite) d.bad = 3;
    desc[b] = d;
    n_valid[b] = d.bad ? 0 : n;
    if (!d.bad && n > 0) atomicMax(max_cells, (int)cells);
  }
}

// ---- device-wide form for batches with a cloud the LDS form cannot hold (a 64-beam less-flat list is ~100 k points) ----
// It sorts RUNS, not points: points arrive in ring / azimuth order, a run of consecutive points of one voxel is one
// (key, run number) pair, there are 4-5 x fewer runs than points, and a stable sort of the runs by (cloud, voxel) keeps a
// voxel's runs in arrival order.  Round 1 sorted every point (five 8-bit passes over 64-bit keys for 125 M points of
// 1 250 64-beam scans, then a gather of the points into key order): 9.7 ms of that configuration's 19.5.
//   head kernel        slot e of the compacted valid-point numbering (cloud b = upper_bound(in_off, e) - 1, position
//                      k = e - in_off[b]): voxel index of the point and of its predecessor in the cloud -> head flag
//   (scan)             run numbers in arrival order
//   run kernel         per head: key = cloud | voxel index [| run number in the low bits], first slot of the run
//   (stable radix sort of the runs, voxel heads, scan)
//   centroid kernel    one thread per voxel: its runs in order, their points in order, f32 sums like CentroidPoint
__device__ __forceinline__ unsigned long long vb_cell(const VoxelBatchView& v, const VoxelCloudDesc& d, const float4 p) {
  const int i0 = (int)(floorf(p.x * v.inv_leaf) - (float)d.min_b[0]);
  const int i1 = (int)(floorf(p.y * v.inv_leaf) - (float)d.min_b[1]);
  const int i2 = (int)(floorf(p.z * v.inv_leaf) - (float)d.min_b[2]);
  return (unsigned long long)((long long)i0 + (long long)i1 * d.div_b[0] + (long long)i2 * d.div_b[0] * (long long)d.div_b[1]);
}

__global__ void __launch_bounds__(256) voxel_batch_head_kernel(VoxelBatchView v, const VoxelCloudDesc* __restrict__ desc,
                                                                const int* __restrict__ in_off, int n_valid, int* __restrict__ flag) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_valid) return;
  const int b = find_scan_off_wave(in_off, v.n_clouds, e);
  const int k = e - in_off[b];
  const VoxelCloudDesc d = desc[b];
  flag[e] = (k == 0 || vb_cell(v, d, vb_point(v, b, k)) != vb_cell(v, d, vb_point(v, b, k - 1))) ? 1 : 0;
}

// run_first[r] = first slot of run r (runs numbered in arrival order), run_first[n_runs] = n_valid
__global__ void __launch_bounds__(256) voxel_batch_run_kernel(VoxelBatchView v, const VoxelCloudDesc* __restrict__ desc,
                                                               const int* __restrict__ in_off, int n_valid, const int* __restrict__ flag,
                                                               const int* __restrict__ pos, int cell_bits, int rid_bits,
                                                               unsigned long long* __restrict__ keys, unsigned* __restrict__ vals,
                                                               int* __restrict__ run_first) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_valid) return;
  if (e == n_valid - 1) run_first[pos[e]] = n_valid;
  const int b = find_scan_off_wave(in_off, v.n_clouds, e);           // by the whole wavefront (scalar search), before the heads split off
  if (!flag[e]) return;
  const int k = e - in_off[b];
  const unsigned rid = (unsigned)(pos[e] - 1);
  const unsigned long long cell = vb_cell(v, desc[b], vb_point(v, b, k));
  const unsigned long long ck = ((unsigned long long)b << cell_bits) | (cell & ((1ull << cell_bits) - 1ull));
  // rid_bits > 0: the run number rides in the low bits of the key and the (stable) radix sort only looks at the bits
  // above it: 8 bytes per run and pass instead of 12 (key + value)
  if (rid_bits > 0) keys[rid] = (ck << rid_bits) | (unsigned long long)rid;
  else { keys[rid] = ck; vals[rid] = rid; }
  run_first[rid] = e;
}

__global__ void __launch_bounds__(256) voxel_batch_flag_kernel(const unsigned long long* __restrict__ keys, int n, int idx_bits, int* __restrict__ flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  flag[i] = (i == 0 || (keys[i] >> idx_bits) != (keys[i - 1] >> idx_bits)) ? 1 : 0;
}

// positions (in the sorted run order) of the runs that open a voxel
__global__ void __launch_bounds__(256) voxel_batch_vhead_kernel(const int* __restrict__ flag, const int* __restrict__ pos, int n_runs,
                                                                 int* __restrict__ head_pos) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n_runs && flag[j]) head_pos[pos[j] - 1] = j;
}

// one thread per voxel: its runs in sorted (= arrival) order, their points in order, f32 accumulators; the voxel that
// opens a cloud also publishes the output boundaries of every cloud since the previous non-empty one
__global__ void __launch_bounds__(256) voxel_batch_run_centroid_kernel(VoxelBatchView v, const int* __restrict__ in_off,
                                                                        const unsigned long long* __restrict__ keys, const unsigned* __restrict__ vals,
                                                                        const int* __restrict__ run_first, const int* __restrict__ head_pos,
                                                                        const int* __restrict__ pos, int n_runs, int cell_bits, int rid_bits,
                                                                        float4* __restrict__ out, int* __restrict__ out_off) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  const int m = pos[n_runs - 1];                               // number of voxels in the batch
  if (o >= m) return;
  const int j0 = head_pos[o];
  const int j1 = (o + 1 < m) ? head_pos[o + 1] : n_runs;
  const int b = (int)(keys[j0] >> (cell_bits + rid_bits));
  const int base = in_off[b];
  float sx = 0.f, sy = 0.f, sz = 0.f, st = 0.f;
  int total = 0;
  // A point is four dependent loads away (sorted key -> run bounds -> index list -> point): the next run's bounds and the
  // next four indices are requested while the current four points are summed, so that a round of four points waits for
  // one memory round trip.  A 0.4 m voxel next to a 64-beam sensor holds hundreds of points: its thread's chain is what
  // its wavefront waits for.
  const int co = v.off[b];
  auto run_bounds = [&](int j, int& k, int& len) __attribute__((always_inline)) {
    const unsigned rid = rid_bits > 0 ? (unsigned)(keys[j] & ((1ull << rid_bits) - 1ull)) : vals[j];
    const int e0 = run_first[rid];
    len = run_first[rid + 1] - e0; k = e0 - base;
  };
  auto resolve = [&](int k, int e, int len, int (&src)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; i++) { const int kk = k + min(e + i, len - 1); src[i] = v.idx ? v.idx[co + kk] : kk; }
  };
  int k, len;
  run_bounds(j0, k, len);
  for (int j = j0; j < j1; j++) {
    int kn = 0, lenn = 1;
    if (j + 1 < j1) run_bounds(j + 1, kn, lenn);
    total += len;
    int src[4];
    resolve(k, 0, len, src);
    for (int e = 0; e < len; e += 4) {                                  // four loads in flight, added in arrival order
      const float4 p0 = v.pts[co + src[0]], p1 = v.pts[co + src[1]], p2 = v.pts[co + src[2]], p3 = v.pts[co + src[3]];
      if (e + 4 < len) resolve(k, e + 4, len, src);
      sx += p0.x; sy += p0.y; sz += p0.z; st += p0.w;
      if (e + 1 < len) { sx += p1.x; sy += p1.y; sz += p1.z; st += p1.w; }
      if (e + 2 < len) { sx += p2.x; sy += p2.y; sz += p2.z; st += p2.w; }
      if (e + 3 < len) { sx += p3.x; sy += p3.y; sz += p3.z; st += p3.w; }
    }
    k = kn; len = lenn;
  }
  const float c = (float)total;
  out[o] = make_float4(sx / c, sy / c, sz / c, st / c);
  const int prev = j0 == 0 ? -1 : (int)(keys[j0 - 1] >> (cell_bits + rid_bits));
  for (int cl = prev + 1; cl <= b; cl++) out_off[cl] = o;                 // empty clouds in between start (and end) here
  if (j1 == n_runs) for (int cl = b + 1; cl <= v.n_clouds; cl++) out_off[cl] = m;
}

// ---- batched voxel filter, one workgroup per cloud, everything between the two reads of the points in LDS ----
// The device-wide form above streams the batch through HBM about ten times (keys, four radix passes over 64-bit keys,
// flags, scan, gather, centroids).  A cloud's points do not fit in LDS (a VLP-16 less-flat list is ~23 k points), but
// its RUNS do: points arrive in ring / azimuth order, so consecutive points mostly fall into the same voxel, and a run of
// consecutive same-voxel points is one 8-byte record {voxel, first point, length}.  Sorting the runs by voxel id with a
// STABLE radix keeps them in arrival order inside a voxel, and summing a voxel's runs point by point reproduces
// pcl::VoxelGrid's centroid accumulation (f32 sums in arrival order) bit for bit.
//   phase 1 (global read 1)  every wavefront owns a contiguous slice of the cloud: absolute voxel coordinates
//                            floor(p * inv_leaf), head flags against the previous point, run records into the
//                            wavefront's own slot range (so run ids ascend with the arrival order), integer bounding box
//   phase 2                  bounding box -> min_b / div_b (= pcl's, floor is monotone), coordinates -> voxel index
//   phase 3                  LSD radix sort of the run ids by voxel index, 8 bits per pass, stable: per-wavefront
//                            histograms, bin-major prefix, ballot-matched ranks inside a 64-run chunk
//   phase 4                  voxel heads in the sorted order
//   phase 5 (global read 2)  one thread per voxel: its runs in order, their points in order, centroid to the staging
//                            area of the cloud (the clouds are compacted by voxel_batch_compact_kernel afterwards)
// flags[b]: 0 ok, 1 / 2 / 3 = VoxelCloudDesc::bad, 4 = does not fit (too many runs or points, coordinates beyond
// +-8191 voxels): the caller then runs the device-wide form for the whole batch; 5 = escalated by the small instantiation.
// Two instantiations: <16, 768> (1024 threads, 12 288 runs: 96 KB of records + 48 KB of order buffers, one workgroup
// per CU) for clouds like a less-flat list, <4, 512> (256 threads, 2 048 runs, 26 KB: six workgroups per CU) when every
// cloud of the batch has at most kVoxSmallPoints points (corner lists).
constexpr int kVoxMaxPoints = 65535;                // the first point of a run is a 16-bit number
constexpr int kVoxSmallPoints = 4096;

// kGlobal (third instantiation, <16, 4096, true>): the run records and the two order buffers live in a global scratch
// area (per cloud kVoxMaxRuns x 8 B + 2 x kVoxMaxRuns x 2 B, L2-resident, same code) instead of LDS.  65 536 run slots,
// 4 096 per wavefront slice, can never overflow (a slice holds at most 4 096 points), so a cloud of up to 65 535 points
// that the LDS forms refused for its RUN count (flag 4) is served without leaving the device: the per-scan SLAM step has
// no host round trip in which it could fall back to the device-wide form.  only_escalated == 2: serve flag-4 clouds.
// kBig (fourth form, <16, 4096, true, true>, round 5): a list of up to 131 071 points (a 64-beam less-flat list is ~100 k) in ONE
// workgroup with the records in global scratch: the run record packs 3 x 13 coordinate bits (+-4095 voxels) | 18 bits first point | 6 bits
// length, the chunk table has two entries per thread.  65 536 run slots (run ids stay 16-bit); more runs, wider coordinates or more
// points keep flag 4 and the batch goes to the device-wide form as before.  Launched persistently (voxel_cloud_big_kernel: a fixed
// number of workgroups walk the list of refused clouds), so the scratch is per workgroup, not per cloud.
#ifdef MSFL_GRID_PROF
__device__ unsigned long long g_vox_prof[64 * 8];     // profile build: phase clocks (10 ns ticks) of the last launches, per (form, cloud % 8)
#define VOX_TICK(k) do { __syncthreads(); if (threadIdx.x == 0) vq[k] = wall_clock64(); } while (0)
#else
#define VOX_TICK(k) do { } while (0)
#endif
template <int kVoxWaves, int kVoxRunsPerWave, bool kGlobal, bool kBig>
__device__ __forceinline__ void voxel_cloud_body(const VoxelBatchView& v, const int b, float4* __restrict__ staging, float4* __restrict__ run_sums,
                                                 int* __restrict__ m_out, int* __restrict__ flags, int only_escalated,
                                                 unsigned long long* __restrict__ g_run, unsigned short* __restrict__ g_ord) {
  constexpr int kVoxMaxRuns = kVoxWaves * kVoxRunsPerWave;
  constexpr int kThreads = 64 * kVoxWaves;
  constexpr int kLdsRuns = kGlobal ? 1 : kVoxMaxRuns;
  // run record: [coords 3 x kCB bits, later the voxel index][first point : kFB][length - 1 : 6]
  constexpr int kCB = kBig ? 13 : 14, kFB = kBig ? 18 : 16, kCellShift = kFB + 6;
  constexpr int kCoordOff = 1 << (kCB - 1);
  constexpr unsigned long long kCoordMask = (1ull << kCB) - 1ull, kLowMask = (1ull << kCellShift) - 1ull, kFirstMask = (1ull << kFB) - 1ull;
  constexpr int kMaxPoints = kBig ? 131071 : (kVoxWaves < 16 ? kVoxSmallPoints : kVoxMaxPoints);
  static_assert(!kBig || kGlobal, "the big form keeps its records in global scratch");
  __shared__ unsigned long long s_run_lds[kLdsRuns];          // (see the record layout above)
  __shared__ unsigned short s_ord_lds[2][kLdsRuns];
  unsigned long long* const s_run = kGlobal ? g_run + (size_t)blockIdx.x * kVoxMaxRuns : s_run_lds;
  unsigned short* const s_ord0 = kGlobal ? g_ord + (size_t)blockIdx.x * 2 * kVoxMaxRuns : s_ord_lds[0];
  unsigned short* const s_ord1 = kGlobal ? g_ord + (size_t)blockIdx.x * 2 * kVoxMaxRuns + kVoxMaxRuns : s_ord_lds[1];
  auto s_ord = [&](int which) __attribute__((always_inline)) { return which ? s_ord1 : s_ord0; };
  __shared__ unsigned short s_hist[kVoxWaves][256];
  // Round 5: the run slots are ONE pool per cloud, handed out chunk by chunk (a chunk = 64 consecutive points of a wavefront's slice:
  // one LDS atomic per chunk), instead of kVoxRunsPerWave slots per wavefront.  A list whose runs crowd into a few slices (the
  // upper rings of an outdoor scan are canopy: nearly every point its own voxel, 900-1 000 runs in a 1 400-point slice while the
  // whole list has 8-10 k) no longer overflows a 768-slot slice and sends the batch to the device-wide form.  The chunk table
  // (first slot, runs) restores the arrival order for the sort: chunks in (wavefront, position) order, slots ascending inside.
  constexpr int kMaxChunks = kVoxWaves * (kBig ? 128 : 64);  // <= 65 535 (4 096; big: 131 071) points over 16 (4) slices of whole chunks
  __shared__ unsigned short s_cbase[kMaxChunks], s_ccnt[kMaxChunks];
  __shared__ int s_alloc;
  __shared__ int s_bb[6];
  __shared__ int s_flag, s_total, s_wsum[kVoxWaves];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cap = v.off[b + 1] - v.off[b];
  const int n = v.count ? min(max(v.count[b], 0), cap) : cap;
  // the small instantiation runs first and ESCALATES (flag 5) what it cannot hold; the large one then only serves those
  constexpr bool kSmall = kVoxWaves < 16;
  if (only_escalated && flags[b] != (only_escalated == 2 ? 4 : 5)) return;
  if (tid == 0) { s_flag = 0; s_alloc = 0; for (int a = 0; a < 3; a++) { s_bb[a] = INT32_MAX; s_bb[3 + a] = INT32_MIN; } }
  for (int c = tid; c < kMaxChunks; c += kThreads) s_ccnt[c] = 0;
  __syncthreads();
  if (n > kMaxPoints) { if (tid == 0) { flags[b] = kSmall ? 5 : 4; m_out[b] = 0; } return; }
#ifdef MSFL_GRID_PROF
  unsigned long long vq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  VOX_TICK(0);
  // ---- phase 1 ----
  const int seg = ((n + kVoxWaves - 1) / kVoxWaves + 63) & ~63;       // points per wavefront, whole chunks
  const int k0 = wave * seg, k1 = min(k0 + seg, n);
  const int chunk0 = wave * (seg >> 6);                                // this slice's first entry of the chunk table
  int mn[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, mx[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
  int my_flag = 0;
  constexpr int kU = 4;                                                // chunks whose loads are in flight together
  // A point is reached through the index list: two dependent loads.  The indices of the NEXT round are requested while
  // this round's points are still on their way, so that a round waits for one memory round trip, not two (the points
  // one round ahead as well: no further gain, 24 more VGPRs).
  const int cloud_o = v.off[b];
  auto resolve = [&](int base0, int (&dst)[kU]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < kU; u++) {
      const int k = min(base0 + 64 * u + lane, n - 1);                   // clamped: unconditional loads
      dst[u] = v.idx ? v.idx[cloud_o + k] : k;
    }
  };
  int src[kU] = {0, 0, 0, 0};
  if (k0 < k1) resolve(k0, src);                                          // (an empty slice, or an empty cloud, reads nothing)
  for (int base0 = k0; base0 < k1; base0 += 64 * kU) {
    float4 pp[kU];
#pragma unroll
    for (int u = 0; u < kU; u++) pp[u] = v.pts[cloud_o + src[u]];           // all issued before the first use
    resolve(min(base0 + 64 * kU, max(k1 - 1, 0)), src);
#pragma unroll
    for (int u = 0; u < kU; u++) {
      const int base = base0 + 64 * u;
      if (base >= k1) break;
      const int k = base + lane;
      const bool valid = k < k1;
      int c0 = 0, c1 = 0, c2 = 0;
      if (valid) {
        const float4 p = pp[u];
        const float f0 = floorf(p.x * v.inv_leaf), f1 = floorf(p.y * v.inv_leaf), f2 = floorf(p.z * v.inv_leaf);
        if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) my_flag = max(my_flag, 3);
        else if (!(fabsf(f0) < (float)(kCoordOff - 1) && fabsf(f1) < (float)(kCoordOff - 1) && fabsf(f2) < (float)(kCoordOff - 1))) my_flag = max(my_flag, 4);
        // a finite point whose 4th field (intensity / relative time) is NaN or Inf is a point pcl keeps: its voxel's 4th sum goes non-finite, no
        // other voxel's does.  The lockstep below multiplies idle lanes' values by 0.0 (0 * NaN = NaN would reach other runs of the chunk), so
        // such a cloud is summed by the device-wide form, which adds point by point like pcl (flag 4 = "does not fit this form")
        else if (!(fabsf(p.w) < INFINITY)) my_flag = max(my_flag, 4);
        c0 = (int)f0; c1 = (int)f1; c2 = (int)f2;
        mn[0] = min(mn[0], c0); mn[1] = min(mn[1], c1); mn[2] = min(mn[2], c2);
        mx[0] = max(mx[0], c0); mx[1] = max(mx[1], c1); mx[2] = max(mx[2], c2);
      }
      // a run never crosses a chunk: lane 0 always opens one
      const int q0 = __shfl_up(c0, 1), q1 = __shfl_up(c1, 1), q2 = __shfl_up(c2, 1);
#ifndef MSFL_VOX_RUN_CAP
#define MSFL_VOX_RUN_CAP 64
#endif
      // (a run may be cut short at will: its voxel then has one more run, summed in the same order; MSFL_VOX_RUN_CAP < 64 bounds the lockstep below)
      const bool head = valid && (lane == 0 || c0 != q0 || c1 != q1 || c2 != q2 || (MSFL_VOX_RUN_CAP < 64 && (lane & (MSFL_VOX_RUN_CAP - 1)) == 0));
      const unsigned long long heads = __ballot(head), live = __ballot(valid);
      int len = 0;                                                       // of the run this lane opens
      int slot = 0;
      const int n_heads = __popcll(heads);
      int cbase = 0;
      if (lane == 0) {
        cbase = atomicAdd(&s_alloc, n_heads);                            // this chunk's slots in the cloud's pool
        const int ci = chunk0 + ((base - k0) >> 6);
        s_cbase[ci] = (unsigned short)min(cbase, 65535); s_ccnt[ci] = (unsigned short)n_heads;
      }
      cbase = __builtin_amdgcn_readfirstlane(cbase);
      if (head) {
        const unsigned long long above = heads & ~((2ull << lane) - 1ull);             // heads in higher lanes
        const int end = above ? __ffsll((long long)above) - 1 : __popcll(live);
        slot = cbase + __popcll(heads & ((1ull << lane) - 1ull));
        len = end - lane;
        if (slot < kVoxMaxRuns)
          s_run[slot] = ((unsigned long long)(unsigned)(c0 + kCoordOff) << (2 * kCB + kCellShift)) | ((unsigned long long)(unsigned)(c1 + kCoordOff) << (kCB + kCellShift)) |
                        ((unsigned long long)(unsigned)(c2 + kCoordOff) << kCellShift) | ((unsigned long long)(unsigned)k << 6) |
                        (unsigned long long)(len - 1);
      }
      // The run's points summed in arrival order while they are in registers: head lane h adds the values of lanes
      // h + 1, h + 2, ... one per step, every lane's value moving one lane down per step on the DPP network
      // (wave_shl:1).  A voxel with ONE run (93 % of a less-flat list's voxels) then needs no second look at its points;
      // the accumulators start from +0 like CentroidPoint's (0 + x, so that a lone -0 comes out as +0), and an inactive
      // step adds +0, which never changes a sum that started that way.
      {
        const float4 p = pp[u];
        // Round 5b: the conditional additions as two packed fused multiply-adds with a 1.0 / 0.0 multiplier per lane -- fma(v, 1, s) IS
        // v + s (one rounding of the exact sum) and fma(v, 0, s) is s for every finite v (a cloud with a non-finite coordinate is refused as a
        // whole, flag 3, one with a non-finite 4th field leaves for the device-wide form, flag 4, whatever the sums here hold): 4 shifts +
        // 1 select + 2 v_pk_fma_f32 per step instead of 4 + 4 + 4.
        typedef float vox_f2 __attribute__((ext_vector_type(2)));
        vox_f2 sxy = {0.f + p.x, 0.f + p.y}, szt = {0.f + p.z, 0.f + p.w};
        float vx = p.x, vy = p.y, vz = p.z, vt = p.w;
        const int longest = (int)wave_extremum_u32<true>((unsigned)len);
        for (int e = 1; e < longest; e++) {
          vx = dpp_wave_shl1(vx); vy = dpp_wave_shl1(vy); vz = dpp_wave_shl1(vz); vt = dpp_wave_shl1(vt);
          const float on = e < len ? 1.0f : 0.0f;
          const vox_f2 m = {on, on};
          sxy = __builtin_elementwise_fma((vox_f2){vx, vy}, m, sxy);
          szt = __builtin_elementwise_fma((vox_f2){vz, vt}, m, szt);
        }
        if (head && slot < kVoxMaxRuns) run_sums[v.off[b] + slot] = make_float4(sxy.x, sxy.y, szt.x, szt.y);     // a cloud has no more runs than points
      }
    }
  }
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn[a] = min(mn[a], __shfl_xor(mn[a], o)); mx[a] = max(mx[a], __shfl_xor(mx[a], o)); }
  if (lane == 0) {
    for (int a = 0; a < 3; a++) { atomicMin(&s_bb[a], mn[a]); atomicMax(&s_bb[3 + a], mx[a]); }
  }
  if (my_flag) atomicMax(&s_flag, my_flag);
  __syncthreads();
  VOX_TICK(1);
  // ---- phase 2 ----
  int flag = s_flag;
  if (s_alloc > kVoxMaxRuns) flag = max(flag, kSmall ? 5 : 4);            // more runs than the pool holds
  const int mb0 = s_bb[0], mb1 = s_bb[1], mb2 = s_bb[2];
  const long long d0 = (long long)s_bb[3] - mb0 + 1, d1 = (long long)s_bb[4] - mb1 + 1, d2 = (long long)s_bb[5] - mb2 + 1;
  long long cells = 1;
  if (n > 0 && flag == 0) {
    cells = d0; if (cells > 0x7fffffffLL) flag = 1;
    if (!flag) { cells *= d1; if (cells > 0x7fffffffLL) flag = 1; }
    if (!flag) { cells *= d2; if (cells > 0x7fffffffLL) flag = 1; }        // pcl: "leaf size too small", the cloud is not filtered
  }
  if (flag != 0 || n == 0) { if (tid == 0) { flags[b] = flag; m_out[b] = 0; } return; }
  const int E = s_alloc;                                   // all runs
  for (int id = tid; id < E; id += kThreads) {
    const unsigned long long r = s_run[id];
    const long long i0 = (long long)(int)((r >> (2 * kCB + kCellShift)) & kCoordMask) - kCoordOff - mb0,
                    i1 = (long long)(int)((r >> (kCB + kCellShift)) & kCoordMask) - kCoordOff - mb1,
                    i2 = (long long)(int)((r >> kCellShift) & kCoordMask) - kCoordOff - mb2;
    const unsigned long long cell = (unsigned long long)(i0 + i1 * d0 + i2 * d0 * d1);
    s_run[id] = (cell << kCellShift) | (r & kLowMask);
  }
  {
    // run ids in ARRIVAL order: exclusive prefix of the chunk table's counts (entry c owned by thread c; unused entries are 0), then every
    // wavefront lists the runs of its own chunks
    constexpr int kPer = kMaxChunks / kThreads > 0 ? kMaxChunks / kThreads : 1;      // chunk-table entries per thread (big form: 2)
    static_assert(kMaxChunks <= kPer * kThreads, "the chunk table is covered by the workgroup");
    int cnt_e[kPer], cnt = 0;
#pragma unroll
    for (int q = 0; q < kPer; q++) { const int e = kPer * tid + q; cnt_e[q] = e < kMaxChunks ? (int)s_ccnt[e] : 0; cnt += cnt_e[q]; }
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o); if (lane >= o) incl += u; }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    int excl = incl - cnt;
    for (int w = 0; w < wave; w++) excl += s_wsum[w];
    // thread t now knows where its chunks' runs start in the arrival order; handed to the wavefront that owns the chunk through the
    // histogram area (free until the sort)
    int* const c_at = reinterpret_cast<int*>(&s_hist[0][0]);             // kMaxChunks ints <= kVoxWaves * 256 * 2 bytes
    static_assert(kMaxChunks * 4 <= kVoxWaves * 256 * 2, "chunk prefix fits the histogram area");
#pragma unroll
    for (int q = 0; q < kPer; q++) { const int e = kPer * tid + q; if (e < kMaxChunks) c_at[e] = excl; excl += cnt_e[q]; }
    __syncthreads();
    const int n_chunks = (k1 > k0) ? ((k1 - k0 + 63) >> 6) : 0;
    for (int c = 0; c < n_chunks; c++) {
      const int ci = chunk0 + c;
      const int cnt_c = s_ccnt[ci], base_c = s_cbase[ci], at_c = c_at[ci];
      if (lane < cnt_c) s_ord(0)[at_c + lane] = (unsigned short)(base_c + lane);
    }
  }
  int nbits = 1;
  while ((1ll << nbits) < cells) nbits++;
  __syncthreads();
  VOX_TICK(2);
  // ---- phase 3: stable LSD radix over the voxel index ----
  const int segE = ((E + kVoxWaves - 1) / kVoxWaves + 63) & ~63;
  const int e0 = min(wave * segE, E), e1 = min(e0 + segE, E);
  int cur = 0;
  for (int shift = 0; shift < nbits; shift += 8) {
    for (int d = lane; d < 256; d += 64) s_hist[wave][d] = 0;
    // (no barrier: a wavefront only touches its own histogram row until the prefix)
    for (int base = e0; base < e1; base += 64) {
      const int pos = base + lane;
      const bool valid = pos < e1;
      const unsigned dg = valid ? (unsigned)((s_run[s_ord(cur)[pos]] >> (kCellShift + shift)) & 0xffu) : 0x100u;
      unsigned long long peers = __ballot(valid);
#pragma unroll
      for (int bit = 0; bit < 8; bit++) { const unsigned long long m = __ballot((dg >> bit) & 1u); peers &= ((dg >> bit) & 1u) ? m : ~m; }
      if (valid && (peers & ((1ull << lane) - 1ull)) == 0) s_hist[wave][dg] += (unsigned short)__popcll(peers);      // the group's first lane
    }
    __syncthreads();
    {
      // exclusive prefix over (digit major, wavefront minor): thread t owns entries 4t .. 4t+3 of that order
      unsigned short c[4]; int sum = 0;
#pragma unroll
      for (int q = 0; q < 4; q++) { const int e = 4 * tid + q; c[q] = s_hist[e % kVoxWaves][e / kVoxWaves]; sum += c[q]; }
      int incl = sum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o); if (lane >= o) incl += u; }
      if (lane == 63) s_wsum[wave] = incl;
      __syncthreads();
      int run = incl - sum;
      for (int w = 0; w < wave; w++) run += s_wsum[w];
#pragma unroll
      for (int q = 0; q < 4; q++) { const int e = 4 * tid + q; s_hist[e % kVoxWaves][e / kVoxWaves] = (unsigned short)run; run += c[q]; }
    }
    __syncthreads();
    for (int base = e0; base < e1; base += 64) {
      const int pos = base + lane;
      const bool valid = pos < e1;
      const unsigned short id = valid ? s_ord(cur)[pos] : (unsigned short)0;
      const unsigned dg = valid ? (unsigned)((s_run[id] >> (kCellShift + shift)) & 0xffu) : 0x100u;
      unsigned long long peers = __ballot(valid);
#pragma unroll
      for (int bit = 0; bit < 8; bit++) { const unsigned long long m = __ballot((dg >> bit) & 1u); peers &= ((dg >> bit) & 1u) ? m : ~m; }
      if (valid) {
        const int rank = __popcll(peers & ((1ull << lane) - 1ull));
        const int at = s_hist[wave][dg];
        s_ord(cur ^ 1)[at + rank] = id;
        if (rank == 0) s_hist[wave][dg] = (unsigned short)(at + __popcll(peers));
      }
    }
    cur ^= 1;
    __syncthreads();
  }
  VOX_TICK(3);
  // ---- phase 4: voxel heads (positions in the sorted order) into the other order buffer ----
  unsigned short* heads_at = s_ord(cur ^ 1);
  if (tid == 0) s_total = 0;
  __syncthreads();
  for (int base = 0; base < E; base += kThreads) {
    const int pos = base + tid;
    bool head = false;
    if (pos < E) head = pos == 0 || (s_run[s_ord(cur)[pos]] >> kCellShift) != (s_run[s_ord(cur)[pos - 1]] >> kCellShift);
    const unsigned long long hm = __ballot(head);
    if (lane == 0) s_wsum[wave] = __popcll(hm);
    __syncthreads();
    int at = s_total + __popcll(hm & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; w++) at += s_wsum[w];
    if (head) heads_at[at] = (unsigned short)pos;
    __syncthreads();
    if (tid == 0) { int t = 0; for (int w = 0; w < kVoxWaves; w++) t += s_wsum[w]; s_total += t; }
    __syncthreads();
  }
  const int m = s_total;
  VOX_TICK(4);
  // ---- phase 5: one thread per voxel ----
  float4* out = staging + v.off[b];
  // Every voxel starts from its first run's sum (phase 1); only later runs (7 % of the voxels of a less-flat list have
  // any) are read again and added point by point.  Small voxels: one thread each.  A voxel close to the sensor can hold
  // hundreds of points (a 0.4 m cube on the ground two metres out takes ~60 points of every ring that crosses it): a
  // single thread would chain that many dependent loads while its workgroup, the only one on the CU, waits.  Voxels with
  // more than kBigVoxel points after their first run are therefore collected (s_hist is free now) and summed by a whole
  // wavefront each: one coalesced load per run, then the same sequential f32 additions on every lane through broadcasts.
  // Round 5b: that walk spends eight wave-wide instructions per point (~1 us per run with four wavefronts on a SIMD), the
  // thread's chain ~0.25 us per point with four loads in flight and every other big voxel's thread running next to it: with
  // the threshold at 24 the 620 big voxels of a 64-beam list took 307 us of the list's 800 on the wavefront path; it now only
  // takes voxels a thread would need > 100 us for.  (Tried: the later points as one stream across runs, 8 loads in flight: no gain.)
#ifndef MSFL_VOX_BIG_VOXEL
#define MSFL_VOX_BIG_VOXEL 512   /* round 5b (24 until then); measured 8 / 24 / 96 / 400 / off: 64-beam lists 4.62 ms per 1 250 at 24, then 4.00 / 3.81 / 3.76; corridor list 277 / 265 / - / 246 / 231 us */
#endif
  constexpr int kBigVoxel = MSFL_VOX_BIG_VOXEL;
  // s_hist is free now: [0, kMultiCap) lists the voxels with more than one run, the rest the big ones among them
  constexpr int kMultiCap = kVoxWaves * 240, kBigCap = kVoxWaves * 16;     // of s_hist's kVoxWaves x 256 entries (192 / 64 while a big voxel began at 24 points)
  unsigned short* multi_list = &s_hist[0][0];
  unsigned short* big_list = multi_list + kMultiCap;
  __shared__ int s_nbig;
  __syncthreads();                                           // every thread has read m = s_total
  if (tid == 0) { s_total = 0; s_nbig = 0; }
  __syncthreads();
  const float4* sums = run_sums + v.off[b];
  auto sum_of = [&](int id) { return sums[id]; };            // run id = its slot in the cloud's pool
  // the later runs of voxel r added point by point to (sx, sy, sz, st); returns the voxel's point count
  auto later_runs = [&](int j0, int j1, float& sx, float& sy, float& sz, float& st) {
    int total = 0;
    auto resolve4 = [&](int k, int e, int len, int (&src)[4]) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 4; i++) { const int kk = k + min(e + i, len - 1); src[i] = v.idx ? v.idx[cloud_o + kk] : kk; }
    };
    for (int j = j0 + 1; j < j1; j++) {
      const unsigned long long rec = s_run[s_ord(cur)[j]];
      const int k = (int)((rec >> 6) & kFirstMask), len = (int)(rec & 0x3fu) + 1;
      total += len;
      int src[4];
      resolve4(k, 0, len, src);
      for (int e = 0; e < len; e += 4) {                                // four loads in flight (their indices a round ahead), added in arrival order
        const float4 p0 = v.pts[cloud_o + src[0]], p1 = v.pts[cloud_o + src[1]], p2 = v.pts[cloud_o + src[2]], p3 = v.pts[cloud_o + src[3]];
        if (e + 4 < len) resolve4(k, e + 4, len, src);
        sx += p0.x; sy += p0.y; sz += p0.z; st += p0.w;               // CentroidPoint accumulators (f32), arrival order
        if (e + 1 < len) { sx += p1.x; sy += p1.y; sz += p1.z; st += p1.w; }
        if (e + 2 < len) { sx += p2.x; sy += p2.y; sz += p2.z; st += p2.w; }
        if (e + 3 < len) { sx += p3.x; sy += p3.y; sz += p3.z; st += p3.w; }
      }
    }
    return total;
  };
  // pass 1: single-run voxels are done with one 16-byte load; the others queue up, so that no lane waits on a
  // neighbour's chain of dependent point loads five voxels in a row
  for (int r = tid; r < m; r += kThreads) {
    const int j0 = heads_at[r], j1 = (r + 1 < m) ? (int)heads_at[r + 1] : E;
    const int id0 = s_ord(cur)[j0];
    const float4 first = sum_of(id0);
    const int len0 = (int)(s_run[id0] & 0x3fu) + 1;
    float sx = first.x, sy = first.y, sz = first.z, st = first.w;
    int total = len0;
    if (j1 - j0 > 1) {
      const int at = atomicAdd(&s_total, 1);
      if (at < kMultiCap) { multi_list[at] = (unsigned short)r; continue; }
      total += later_runs(j0, j1, sx, sy, sz, st);                       // list full: finished here after all
    }
    const float c = (float)total;
    out[r] = make_float4(sx / c, sy / c, sz / c, st / c);
  }
  __syncthreads();
  // pass 2: the multi-run voxels, one thread each; the big ones move on to the wavefront-per-voxel pass
  const int n_multi = min(s_total, kMultiCap);
  for (int q = tid; q < n_multi; q += kThreads) {
    const int r = multi_list[q];
    const int j0 = heads_at[r], j1 = (r + 1 < m) ? (int)heads_at[r + 1] : E;
    const int id0 = s_ord(cur)[j0];
    const int len0 = (int)(s_run[id0] & 0x3fu) + 1;
    int later = 0;
    for (int j = j0 + 1; j < j1; j++) later += (int)(s_run[s_ord(cur)[j]] & 0x3fu) + 1;
    if (later > kBigVoxel) {
      const int at = atomicAdd(&s_nbig, 1);
      if (at < kBigCap) { big_list[at] = (unsigned short)r; continue; }
    }
    const float4 first = sum_of(id0);
    float sx = first.x, sy = first.y, sz = first.z, st = first.w;
    const float c = (float)(len0 + later_runs(j0, j1, sx, sy, sz, st));
    out[r] = make_float4(sx / c, sy / c, sz / c, st / c);
  }
  __syncthreads();
  VOX_TICK(5);
  const int n_big = min(s_nbig, kBigCap);
  for (int q = wave; q < n_big; q += kVoxWaves) {
    const int r = big_list[q];
    const int j0 = heads_at[r], j1 = (r + 1 < m) ? (int)heads_at[r + 1] : E;     // j1 - j0 >= 2 here
    const int id0 = s_ord(cur)[j0];
    const float4 first = sum_of(id0);
    float sx = first.x, sy = first.y, sz = first.z, st = first.w;
    int total = (int)(s_run[id0] & 0x3fu) + 1;
    // software pipeline over the runs: the next run's points are loading while this one is summed
    unsigned long long rec = s_run[s_ord(cur)[j0 + 1]];
    int k = (int)((rec >> 6) & kFirstMask), len = (int)(rec & 0x3fu) + 1;
    float4 p = vb_point(v, b, k + min(lane, len - 1));
    for (int j = j0 + 1; j < j1; j++) {
      const float4 pc = p; const int lc = len;
      if (j + 1 < j1) {
        rec = s_run[s_ord(cur)[j + 1]];
        k = (int)((rec >> 6) & kFirstMask); len = (int)(rec & 0x3fu) + 1;
        p = vb_point(v, b, k + min(lane, len - 1));
      }
      // e is wave-uniform: v_readlane (a scalar broadcast) instead of ds_bpermute; the additions are the same sequential f32 chain
      for (int e = 0; e < lc; e++) {
        sx += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pc.x), e)); sy += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pc.y), e));
        sz += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pc.z), e)); st += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pc.w), e));
      }
      total += lc;
    }
    if (lane == 0) { const float c = (float)total; out[r] = make_float4(sx / c, sy / c, sz / c, st / c); }
  }
  VOX_TICK(6);
#ifdef MSFL_GRID_PROF
  if (tid == 0) {
    unsigned long long* q = g_vox_prof + (size_t)((kVoxWaves == 4 ? 0 : kGlobal ? 16 : 8) + (b & 7)) * 8;
    for (int k = 0; k < 6; k++) q[k] = vq[k + 1] - vq[k];
    q[6] = ((unsigned long long)n << 32) | (unsigned)m; q[7] = ((unsigned long long)E << 32) | ((unsigned)n_multi << 16) | (unsigned)n_big;
  }
#endif
  if (tid == 0) { flags[b] = 0; m_out[b] = m; }
}

template <int kVoxWaves, int kVoxRunsPerWave, bool kGlobal = false>
__global__ void __launch_bounds__(64 * kVoxWaves) voxel_cloud_lds_kernel(VoxelBatchView v, float4* __restrict__ staging, float4* __restrict__ run_sums,
                                                                          int* __restrict__ m_out, int* __restrict__ flags, int only_escalated,
                                                                          unsigned long long* __restrict__ g_run = nullptr,
                                                                          unsigned short* __restrict__ g_ord = nullptr) {
  voxel_cloud_body<kVoxWaves, kVoxRunsPerWave, kGlobal, false>(v, (int)blockIdx.x, staging, run_sums, m_out, flags, only_escalated, g_run, g_ord);
}

// The big form, persistent: gridDim.x workgroups (each with its own 65 536-run scratch) walk the list of clouds the LDS forms refused
// (flag 4); a cloud the big form cannot hold either keeps its flag.
constexpr size_t kVoxBigScratchBytes = (size_t)65536 * 8 + (size_t)2 * 65536 * 2;     // per workgroup: run records + two order buffers
#ifndef MSFL_VOX_BIG_WAVES
#define MSFL_VOX_BIG_WAVES 8        /* waves per SIMD the big form is compiled for: 8 = two 1 024-thread workgroups per CU (64 VGPRs) */
#endif
__global__ void __launch_bounds__(1024, MSFL_VOX_BIG_WAVES) voxel_cloud_big_kernel(VoxelBatchView v, float4* __restrict__ staging, float4* __restrict__ run_sums,
                                                                int* __restrict__ m_out, int* __restrict__ flags, const int* __restrict__ cloud_list,
                                                                int n_list, unsigned long long* __restrict__ g_run, unsigned short* __restrict__ g_ord) {
  for (int j = (int)blockIdx.x; j < n_list; j += (int)gridDim.x) {
    voxel_cloud_body<16, 4096, true, true>(v, cloud_list[j], staging, run_sums, m_out, flags, 2, g_run, g_ord);
    __syncthreads();                       // the next cloud re-arms the shared state
  }
}

// staging (cloud b's centroids at off[b]) -> the clouds back to back: out[out_off[b] + r]
__global__ void __launch_bounds__(256) voxel_batch_compact_kernel(const float4* __restrict__ staging, const int* __restrict__ off,
                                                                   const int* __restrict__ out_off, int n_clouds, float4* __restrict__ out) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= out_off[n_clouds]) return;
  const int b = find_scan_off_wave(out_off, n_clouds, o);
  out[o] = staging[off[b] + (o - out_off[b])];
}

// exclusive scan of the per-cloud voxel counts (B is small: one workgroup, serial per 1024-chunk carry)
__global__ void __launch_bounds__(1024) voxel_batch_offsets_kernel(const int* __restrict__ cnt, int n, int* __restrict__ off) {
  __shared__ int s_part[16];
  __shared__ int s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int i = base + threadIdx.x;
    const int c = i < n ? cnt[i] : 0;
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if ((threadIdx.x & 63) >= o) incl += t; }
    if ((threadIdx.x & 63) == 63) s_part[threadIdx.x >> 6] = incl;
    __syncthreads();
    int run = s_carry + incl - c;
    for (int w = 0; w < (threadIdx.x >> 6); w++) run += s_part[w];
    if (i < n) off[i] = run;
    __syncthreads();
    if (threadIdx.x == 1023) s_carry = run + c;
    __syncthreads();
  }
  if (threadIdx.x == 0) off[n] = s_carry;
}

}  // namespace msfl

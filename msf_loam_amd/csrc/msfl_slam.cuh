// msfl_slam.cuh — glue kernels of the device-resident per-scan SLAM step (msfl_api_slam.inc).
//
// The stages themselves are the kernels of stage A / B / C, the voxel filter and the map store; what lives here is
// what the reference does in host code BETWEEN them (laser_odometry.cc:69-95, laser_mapping.cc:138-258,260-338) and
// therefore used to cost a PCIe round trip each: the feature clouds gathered out of the full cloud, the pose chain
// (Rigid3d operator* / inverse, rigid_transform.h:105-111), the map gate (laser_mapping.cc:284-285), the offset
// tables of the two matchers, and the result record.  Every size is read from device memory.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "msfl_math.cuh"
#include "msfl_grid.cuh"
#include "msfl_deskew.cuh"

namespace msfl {

// counts block of one scan, written by the extraction kernels: [n_full, n_sharp, n_less_sharp, n_flat, n_less_flat, status, -, -]
enum { SC_FULL = 0, SC_SHARP, SC_LESS_SHARP, SC_FLAT, SC_LESS_FLAT, SC_STATUS, SC_OVERFLOW, SC_USE_LS, SC_USE_LF, SC_IMU_BAD, SC_QUIRK_OOB,
       SC_CLOUD_BAD, SC_WORDS = 12 };
// SC_CLOUD_BAD (keep_clouds): a point of the full cloud outside the two validated lists has a time stamp outside the pre-integration span
// SC_IMU_BAD: a less-sharp / less-flat point's time lies outside the scan's pre-integration span (raised by the gather kernel's lanes,
// cleared by slam_result_kernel once the record is assembled); SC_QUIRK_OOB: reference_quirks and more less-sharp than less-flat points
// SC_USE_LS / SC_USE_LF: the less-sharp / less-flat counts the mapping thread works with: the extraction's, or 0 for a scan that is
// not processed (failed extraction, a list beyond the launch bounds), so that such a scan is neither matched nor inserted

struct SlamCaps { int sharp, less_sharp, flat, less_flat; };     // host-side bounds the launches are sized for

// The scan's IMU inputs on the device (msfl_slam_imu), one block per buffer set, uploaded in one copy: header, then the
// pre-integration samples packed as [sum_dt n | delta_q 4n | delta_p 3n].
constexpr int kSlamImuMaxSamples = 2048;
struct SlamImuDev {
  int mode;                 // 0: no IMU data; 1: UndistortScan (estimator not initialised); 2: is_initialized
  int n;                    // samples
  double velocity[3], gravity[3], presolved[7];
  double data[8 * kSlamImuMaxSamples];
};
__device__ __forceinline__ PreintView slam_preint(const SlamImuDev* __restrict__ imu) {
  PreintView pv;
  pv.n = imu->n; pv.sum_dt = imu->data; pv.dq = imu->data + imu->n; pv.dp = imu->data + 5 * (size_t)imu->n;
  return pv;
}
// The mapping thread's copy of one less-sharp / less-flat point (the odometry thread keeps the raw one):
//   mode 1: UndistortScanInternal (scan_undistortion.cc:5-19): p <- dq(p.time).cast<float>() * p, CHECK_GE(time, 0) (:12) and
//           GetDeltaQP's range CHECK (:26-30) -> *bad
//   mode 2: unchanged here (DoUndistort runs after the match, laser_mapping.cc:197-211), but the same range CHECK is due then and
//           inside the matcher (mapping_scan_matcher.cc:115,185): validated now, while nothing has been published
__device__ __forceinline__ float4 slam_map_point(float4 e, int mode, const PreintView& pv, int* __restrict__ bad) {
  if (mode == 0) return e;
  const DeltaQP o = delta_qp(pv, (double)e.w);
  if (!o.ok || (mode == 1 && !(e.w >= 0.f))) { *bad = 1; return e; }
  if (mode == 2) return e;
  const float qx = (float)o.q.x, qy = (float)o.q.y, qz = (float)o.q.z, qw = (float)o.q.w;       // as undistort_cloud_kernel
  float ux = qy * e.z - qz * e.y, uy = qz * e.x - qx * e.z, uz = qx * e.y - qy * e.x;
  ux += ux; uy += uy; uz += uz;
  const float cx = qy * uz - qz * uy, cy = qz * ux - qx * uz, cz = qx * uy - qy * ux;
  e.x = e.x + qw * ux + cx; e.y = e.y + qw * uy + cy; e.z = e.z + qw * uz + cz;
  return e;
}

// The four feature clouds of TimestampedPointCloud (timestamped_pointcloud.h:11-42) as contiguous arrays: the reference
// push_back()s copies (msf_loam_node.cc:279-344); here one gather out of cloud_full_res per list.  Also writes the five
// offset pairs of the scan-to-scan matcher for THIS scan as `curr` against `last` (the previous scan's counts):
//   odo_off = [0, last_ls | 0, last_lf | 0, sharp | 0, flat | 0, sharp + flat]
__global__ void __launch_bounds__(256)
slam_gather_kernel(const float4* __restrict__ full, const uint16_t* __restrict__ ring, const int* __restrict__ sharp_idx,
                   const int* __restrict__ ls_idx, const int* __restrict__ flat_idx, const int* __restrict__ lf_idx, int* __restrict__ cnt,
                   SlamCaps caps, float4* __restrict__ sharp_pts, float4* __restrict__ ls_pts, uint16_t* __restrict__ ls_ring,
                   float4* __restrict__ flat_pts, float4* __restrict__ lf_pts, uint16_t* __restrict__ lf_ring,
                   const int* __restrict__ last_cnt, SlamCaps last_caps, int* __restrict__ odo_off, int* __restrict__ odo_status,
                   const SlamImuDev* __restrict__ imu, float4* __restrict__ map_ls, float4* __restrict__ map_lf, int quirks) {
  const int list = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool ok = cnt[SC_STATUS] == 0;
  const int n_sharp = ok ? min(cnt[SC_SHARP], caps.sharp) : 0, n_ls = ok ? min(cnt[SC_LESS_SHARP], caps.less_sharp) : 0;
  const int n_flat = ok ? min(cnt[SC_FLAT], caps.flat) : 0, n_lf = ok ? min(cnt[SC_LESS_FLAT], caps.less_flat) : 0;
  if (list == 0 && i == 0) {
    const bool lok = last_cnt && last_cnt[SC_STATUS] == 0;
    odo_off[0] = 0; odo_off[1] = lok ? min(last_cnt[SC_LESS_SHARP], last_caps.less_sharp) : 0;
    odo_off[2] = 0; odo_off[3] = lok ? min(last_cnt[SC_LESS_FLAT], last_caps.less_flat) : 0;
    odo_off[4] = 0; odo_off[5] = n_sharp;
    odo_off[6] = 0; odo_off[7] = n_flat;
    odo_off[8] = 0; odo_off[9] = n_sharp + n_flat;
    // a list longer than the launch bound (a ring id beyond the configured ring count): reported, the scan is not matched
    const int over = ok && (cnt[SC_SHARP] > caps.sharp || cnt[SC_LESS_SHARP] > caps.less_sharp || cnt[SC_FLAT] > caps.flat ||
                            cnt[SC_LESS_FLAT] > caps.less_flat) ? 1 : 0;
    cnt[SC_OVERFLOW] = over;
    // FilterLessFlatLessCornerFeature as the reference executes it (laser_mapping.cc:340-364): the surf cloud is copied through the
    // CORNER filter's index list [0, n_less_sharp) (:359-360).  More corners than surfs: pcl::copyPointCloud reads out of bounds.
    const int oob = (quirks && ok && !over && n_ls > n_lf) ? 1 : 0;
    cnt[SC_QUIRK_OOB] = oob;
    cnt[SC_USE_LS] = (ok && !over && !oob) ? n_ls : 0;
    cnt[SC_USE_LF] = (ok && !over && !oob) ? (quirks ? min(n_lf, n_ls) : n_lf) : 0;
    *odo_status = (!ok || over) ? 3 /*MSFL_BAD_ARG: a scan without features is not matched*/ : 0;
  }
  // map_ls / map_lf: the clouds the MAPPING thread works with (UndistortScan makes new clouds, scan_undistortion.cc:45-62; the
  // odometry thread's scan_last_ keeps the raw ones, laser_odometry.cc:90)
  const int mode = imu->mode;
  if (list == 0) { if (i < n_sharp) sharp_pts[i] = full[sharp_idx[i]]; }
  else if (list == 1) {
    if (i < n_ls) { const int j = ls_idx[i]; const float4 p = full[j]; ls_pts[i] = p; ls_ring[i] = ring[j]; map_ls[i] = slam_map_point(p, mode, slam_preint(imu), cnt + SC_IMU_BAD); }
  } else if (list == 2) { if (i < n_flat) flat_pts[i] = full[flat_idx[i]]; }
  else {
    if (i < n_lf) {
      const int j = lf_idx[i]; const float4 p = full[j]; lf_pts[i] = p; lf_ring[i] = ring[j];
      // UndistortScan (mode 1) runs over the whole cloud BEFORE the truncation of reference_quirks; DoUndistort (mode 2) after it,
      // so that only the first n_less_sharp points can hit its CHECK
      map_lf[i] = (quirks && mode == 2 && i >= n_ls) ? p : slam_map_point(p, mode, slam_preint(imu), cnt + SC_IMU_BAD);
    }
  }
}

// Rigid3d operator* (rigid_transform.h:105-111) and inverse() (:66-70)
__device__ __forceinline__ pose7 pose_compose(const pose7& a, const pose7& b) {
  pose7 o;
  o.t = quat_rotate(a.q, b.t) + a.t;
  o.q = quat_normalized(quat_mul(a.q, b.q));
  return o;
}
__device__ __forceinline__ pose7 pose_inverse(const pose7& a) {
  pose7 o;
  o.q.x = -a.q.x; o.q.y = -a.q.y; o.q.z = -a.q.z; o.q.w = a.q.w;
  const d3 r = quat_rotate(o.q, a.t);
  o.t = mk3(-r.x, -r.y, -r.z);
  return o;
}

// odometry thread, after MatchScan2Scan: pose_scan2world_ = pose_scan2world_ * pose_curr2last_ (laser_odometry.cc:79;
// not on the first scan, :72-73), and the copy the mapping thread receives as odom_result.odom_pose (:84-85).
// A scan whose extraction failed leaves the chain untouched.
__global__ void slam_odom_pose_kernel(const double* __restrict__ c2l, double* __restrict__ odo_cur, double* __restrict__ pose_odom_k,
                                      double* __restrict__ c2l_k, const int* __restrict__ cnt, int first) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  pose7 odo = load_pose(odo_cur);
  if (!first && cnt[SC_STATUS] == 0 && cnt[SC_OVERFLOW] == 0) { odo = pose_compose(odo, load_pose(c2l)); store_pose(odo_cur, odo); }
  store_pose(pose_odom_k, odo);
  for (int k = 0; k < 7; k++) c2l_k[k] = c2l[k];
}

// mapping thread: mode 0 = TransformAssociateToMap (laser_mapping.h:55-57), mode 1 = TransformUpdate (:59-61)
__global__ void slam_map_pose_kernel(double* __restrict__ odom2map, const double* __restrict__ pose_odom_k, double* __restrict__ pose_map_k, int mode) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (mode == 0) store_pose(pose_map_k, pose_compose(load_pose(odom2map), load_pose(pose_odom_k)));
  else store_pose(odom2map, pose_compose(load_pose(pose_map_k), pose_inverse(load_pose(pose_odom_k))));
}

// Between GetSurroundedCloud and MatchScan2Map (laser_mapping.cc:279-311): the gate `corner > 10 && surf > 50`, the
// matcher's offset tables [0, m_c | 0, m_s | 0, m_c + m_s] and its status word (non-zero = the kernels skip the scan and
// the pose guess passes through, exactly the reference's else-branch).
__global__ void slam_map_gate_kernel(const int* __restrict__ n_map_c, const int* __restrict__ n_map_s, int min_c, int min_s,
                                     const int* __restrict__ m_c, const int* __restrict__ m_s, const int* __restrict__ vflag_c,
                                     const int* __restrict__ vflag_s, int* __restrict__ cnt, int* __restrict__ in_off,
                                     int* __restrict__ status, const SlamImuDev* __restrict__ imu, double* __restrict__ pose_map) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int mc = max(*m_c, 0), ms = max(*m_s, 0);
  in_off[0] = 0; in_off[1] = mc; in_off[2] = 0; in_off[3] = ms; in_off[4] = 0; in_off[5] = mc + ms;
  int s = 0;
  if (cnt[SC_STATUS] != 0 || cnt[SC_OVERFLOW] != 0 || cnt[SC_QUIRK_OOB] != 0) s = 3;   // MSFL_BAD_ARG: nothing to match
  else if (cnt[SC_IMU_BAD] != 0) {
    // a time stamp outside the pre-integration span: the reference CHECK-aborts before anything of this scan is published;
    // here the scan is neither matched nor inserted (the voxel filters and surrounded clouds already computed are scratch)
    s = 3; cnt[SC_USE_LS] = 0; cnt[SC_USE_LF] = 0;
  }
  else if (*vflag_c != 0 || *vflag_s != 0) s = 7;                                // MSFL_CAPACITY: a list did not fit the on-chip filter
  else if (!(*n_map_c > min_c && *n_map_s > min_s)) s = 2;                       // MSFL_MAP_TOO_SMALL: the gate
  *status = s;
  // is_initialized: the matcher starts from the IMU-only pre-solve, *pose_estimate_map_scan2world = pose_j
  // (mapping_scan_matcher.cc:57) -- only when it is called at all (gate open, laser_mapping.cc:284-311)
  if (s == 0 && imu->mode == 2) for (int k = 0; k < 7; k++) pose_map[k] = imu->presolved[k];
}

// is_initialized branch, before the match: (dq, dp) = GetDeltaQP(preintegration, pointOri.intensity) for every down-sampled
// feature (mapping_scan_matcher.cc:113-117,183-187) -> the arrays DeskewView takes.  blockIdx.y = list (0 corner, 1 surf).
__global__ void __launch_bounds__(256)
slam_delta_qp_kernel(const SlamImuDev* __restrict__ imu, const float4* __restrict__ vox_c, const float4* __restrict__ vox_s,
                     const int* __restrict__ in_off, int cap_c, int cap_s, double* __restrict__ dq_c, double* __restrict__ dp_c,
                     double* __restrict__ dq_s, double* __restrict__ dp_s) {
  const int list = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = list == 0 ? min(in_off[1], cap_c) : min(in_off[3], cap_s);
  if (i >= n) return;
  const float4 p = list == 0 ? vox_c[i] : vox_s[i];
  // a centroid of in-range times is in range (up to f32 rounding at the ends: clamped, where the reference would abort on an
  // ulp); the per-point CHECK has been applied to every point of the two lists by the gather kernel
  const PreintView pv = slam_preint(imu);
  double dt = (double)p.w;
  dt = fmin(fmax(dt, pv.sum_dt[0]), pv.sum_dt[pv.n - 1]);
  const DeltaQP o = delta_qp(pv, dt);
  double* dq = (list == 0 ? dq_c : dq_s) + 4 * (size_t)i;
  double* dp = (list == 0 ? dp_c : dp_s) + 3 * (size_t)i;
  dq[0] = o.q.x; dq[1] = o.q.y; dq[2] = o.q.z; dq[3] = o.q.w;
  dp[0] = o.p.x; dp[1] = o.p.y; dp[2] = o.p.z;
}

// is_initialized branch, after the match: DoUndistort (laser_mapping.cc:197-211) on the two clouds InsertScan2Map inserts,
//   e <- (dq(t) * e + pose_odom_scan2world_.rotation().conjugate() * (velocity_ * t - 0.5 * G * t * t) + dp(t)).cast<float>()
// blockIdx.y = list (0 less-sharp, 1 less-flat); counts from the scan's count block (0 for a scan that is not inserted).
__global__ void __launch_bounds__(256)
slam_deskew_kernel(const SlamImuDev* __restrict__ imu, const double* __restrict__ pose_odom, const int* __restrict__ cnt,
                   float4* __restrict__ map_ls, int cap_ls, float4* __restrict__ map_lf, int cap_lf) {
  const int list = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = list == 0 ? min(cnt[SC_USE_LS], cap_ls) : min(cnt[SC_USE_LF], cap_lf);
  if (i >= n || imu->mode != 2) return;
  float4* pts = list == 0 ? map_ls : map_lf;
  float4 e = pts[i];
  const double dt = (double)e.w;
  const DeltaQP o = delta_qp(slam_preint(imu), dt);
  if (!o.ok) return;                       // cannot happen: the gather kernel validated every time stamp of these lists
  quat rc; rc.x = -pose_odom[3]; rc.y = -pose_odom[4]; rc.z = -pose_odom[5]; rc.w = pose_odom[6];
  const d3 a = quat_rotate(o.q, mk3((double)e.x, (double)e.y, (double)e.z));
  const d3 m = mk3(imu->velocity[0] * dt - 0.5 * imu->gravity[0] * dt * dt, imu->velocity[1] * dt - 0.5 * imu->gravity[1] * dt * dt,
                   imu->velocity[2] * dt - 0.5 * imu->gravity[2] * dt * dt);
  const d3 b = quat_rotate(rc, m);
  e.x = (float)(a.x + b.x + o.p.x); e.y = (float)(a.y + b.y + o.p.y); e.z = (float)(a.z + b.z + o.p.z);
  pts[i] = e;
}

// msfl_slam_config.keep_clouds: the data products of LaserMapping::Run that reach neither the pose nor the map.  One thread per point of
// cloud_full_res, after TransformUpdate:
//   full_scan[i]  mode 0: the point as extracted; mode 1: UndistortScanInternal (scan_undistortion.cc:5-19, run by UndistortScan before
//                 the match, laser_mapping.cc:170-176); mode 2: DoUndistort (laser_mapping.cc:198-206) — the arithmetic of
//                 slam_map_point / slam_deskew_kernel, so a listed point equals its copy in map_ls / map_lf bit for bit
//   full_map[i]   TransformPointCloud(cloud_full_res, pose_map_scan2world_) (laser_mapping.cc:214-217, rigid_transform.h:131-137)
// A time stamp outside the pre-integration span (the reference CHECK-aborts, scan_undistortion.cc:12,26-30) leaves the point as it
// was and raises cnt[SC_CLOUD_BAD].
__global__ void __launch_bounds__(256)
slam_full_cloud_kernel(const SlamImuDev* __restrict__ imu, const double* __restrict__ pose_odom, const double* __restrict__ pose_map,
                       int* __restrict__ cnt, const float4* __restrict__ full, int n_cap, float4* __restrict__ full_scan,
                       float4* __restrict__ full_map) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = cnt[SC_STATUS] == 0 ? min(cnt[SC_FULL], n_cap) : 0;
  if (i >= n) return;
  float4 e = full[i];
  const int mode = imu->mode;
  if (mode != 0) {
    const double dt = (double)e.w;
    const DeltaQP o = delta_qp(slam_preint(imu), dt);
    if (!o.ok || (mode == 1 && !(e.w >= 0.f))) cnt[SC_CLOUD_BAD] = 1;
    else if (mode == 1) {
      const float qx = (float)o.q.x, qy = (float)o.q.y, qz = (float)o.q.z, qw = (float)o.q.w;
      float ux = qy * e.z - qz * e.y, uy = qz * e.x - qx * e.z, uz = qx * e.y - qy * e.x;
      ux += ux; uy += uy; uz += uz;
      const float cx = qy * uz - qz * uy, cy = qz * ux - qx * uz, cz = qx * uy - qy * ux;
      e.x = e.x + qw * ux + cx; e.y = e.y + qw * uy + cy; e.z = e.z + qw * uz + cz;
    } else {
      quat rc; rc.x = -pose_odom[3]; rc.y = -pose_odom[4]; rc.z = -pose_odom[5]; rc.w = pose_odom[6];
      const d3 a = quat_rotate(o.q, mk3((double)e.x, (double)e.y, (double)e.z));
      const d3 m = mk3(imu->velocity[0] * dt - 0.5 * imu->gravity[0] * dt * dt, imu->velocity[1] * dt - 0.5 * imu->gravity[1] * dt * dt,
                       imu->velocity[2] * dt - 0.5 * imu->gravity[2] * dt * dt);
      const d3 b = quat_rotate(rc, m);
      e.x = (float)(a.x + b.x + o.p.x); e.y = (float)(a.y + b.y + o.p.y); e.z = (float)(a.z + b.z + o.p.z);
    }
  }
  full_scan[i] = e;
  const float3 q = transform_point_f32(load_pose(pose_map), e.x, e.y, e.z);
  full_map[i] = make_float4(q.x, q.y, q.z, e.w);
}

// ---- voxel filter of ONE list of more than 65 535 points (a 64-beam less-flat list), device-sized --------------------
// The one-workgroup forms address a run's first point with 16 bits; a longer list is filtered here the plain way:
// pcl::VoxelGrid's (bounding-box relative) voxel index per point -> stable radix sort of (index, arrival number) ->
// one centroid per index run, f32 sums in arrival order.  Every kernel acts only when the list was refused (flag 4) and
// sizes itself from the device-side count; the host enqueues the chain when the scan capacity allows such a list at all.
struct VoxBigDesc { int mn[3]; int d[3]; int n; int active; };
constexpr int kVoxBigBoxGroups = 64;          // workgroups of the bounding-box pass (one kept a compute unit busy for 67 us on 100 k points)
struct VoxBigPart { int mn[3]; int mx[3]; int bad; int pad; };

// slice g of the list: per-slice box + verdict (0 fine, 1 leaf too small, 3 non-finite point) -> part[g]
__global__ void __launch_bounds__(256)
vox_big_bbox_kernel(const float4* __restrict__ pts, const int* __restrict__ idx, const int* __restrict__ count, int n_cap, float inv_leaf,
                    const int* __restrict__ flag, VoxBigPart* __restrict__ part) {
  __shared__ int s_mn[3], s_mx[3], s_bad;
  const int tid = threadIdx.x;
  if (*flag != 4) return;
  const int n = min(max(*count, 0), n_cap);
  if (tid == 0) { for (int a = 0; a < 3; a++) { s_mn[a] = INT32_MAX; s_mx[a] = INT32_MIN; } s_bad = 0; }
  __syncthreads();
  int mn[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, mx[3] = {INT32_MIN, INT32_MIN, INT32_MIN}, bad = 0;
  for (int k = blockIdx.x * 256 + tid; k < n; k += kVoxBigBoxGroups * 256) {
    const float4 p = pts[idx ? idx[k] : k];
    const float f0 = floorf(p.x * inv_leaf), f1 = floorf(p.y * inv_leaf), f2 = floorf(p.z * inv_leaf);
    if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) { bad = 3; continue; }
    if (!(fabsf(f0) < 1e9f && fabsf(f1) < 1e9f && fabsf(f2) < 1e9f)) { bad = max(bad, 1); continue; }
    const int c[3] = {(int)f0, (int)f1, (int)f2};
    for (int a = 0; a < 3; a++) { mn[a] = min(mn[a], c[a]); mx[a] = max(mx[a], c[a]); }
  }
  for (int a = 0; a < 3; a++) { atomicMin(&s_mn[a], mn[a]); atomicMax(&s_mx[a], mx[a]); }
  if (bad) atomicMax(&s_bad, bad);
  __syncthreads();
  if (tid == 0) {
    VoxBigPart q;
    for (int a = 0; a < 3; a++) { q.mn[a] = s_mn[a]; q.mx[a] = s_mx[a]; }
    q.bad = s_bad; q.pad = 0;
    part[blockIdx.x] = q;
  }
}

// the slices' boxes -> the list's descriptor; a list refused for good (non-finite point, leaf too small) or empty gets its final flag here
__global__ void __launch_bounds__(64)
vox_big_desc_kernel(const VoxBigPart* __restrict__ part, const int* __restrict__ count, int n_cap, int* __restrict__ flag, int* __restrict__ m_out,
                    VoxBigDesc* __restrict__ desc) {
  static_assert(kVoxBigBoxGroups == 64, "one slice per lane");
  const int lane = threadIdx.x;
  if (*flag != 4) { if (lane == 0) desc->active = 0; return; }
  const VoxBigPart q = part[lane];
  int mn[3] = {q.mn[0], q.mn[1], q.mn[2]}, mx[3] = {q.mx[0], q.mx[1], q.mx[2]}, f = q.bad;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    for (int a = 0; a < 3; a++) { mn[a] = min(mn[a], __shfl_xor(mn[a], o)); mx[a] = max(mx[a], __shfl_xor(mx[a], o)); }
    f = max(f, __shfl_xor(f, o));
  }
  if (lane == 0) {
    const int n = min(max(*count, 0), n_cap);
    long long cells = 1;
    for (int a = 0; a < 3 && !f && n > 0; a++) { cells *= (long long)mx[a] - mn[a] + 1; if (cells > 0x7fffffffLL) f = 1; }   // pcl: leaf too small
    for (int a = 0; a < 3; a++) { desc->mn[a] = mn[a]; desc->d[a] = mx[a] - mn[a] + 1; }
    desc->n = n;
    desc->active = (f == 0 && n > 0) ? 1 : 0;
    if (f != 0 || n == 0) { *flag = f; *m_out = 0; }
  }
}

__global__ void __launch_bounds__(256)
vox_big_key_kernel(const float4* __restrict__ pts, const int* __restrict__ idx, int n_cap, float inv_leaf, const VoxBigDesc* __restrict__ desc,
                   unsigned* __restrict__ keys, unsigned* __restrict__ vals) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_cap || !desc->active) return;
  vals[k] = (unsigned)k;
  if (k >= desc->n) { keys[k] = 0xffffffffu; return; }
  const float4 p = pts[idx ? idx[k] : k];
  const long long i0 = (long long)(int)floorf(p.x * inv_leaf) - desc->mn[0], i1 = (long long)(int)floorf(p.y * inv_leaf) - desc->mn[1],
                  i2 = (long long)(int)floorf(p.z * inv_leaf) - desc->mn[2];
  keys[k] = (unsigned)(i0 + i1 * desc->d[0] + i2 * (long long)desc->d[0] * desc->d[1]);
}

__global__ void __launch_bounds__(256)
vox_big_head_kernel(const unsigned* __restrict__ skeys, int n_cap, const VoxBigDesc* __restrict__ desc, int* __restrict__ head) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_cap) return;
  head[k] = (desc->active && k < desc->n && (k == 0 || skeys[k] != skeys[k - 1])) ? 1 : 0;
}

// One centroid per run of equal keys: f32 sums in arrival order (the sort is stable), as pcl's CentroidPoint accumulates them.
// The run's head thread finds its end in the sorted keys (galloping, then bisection); a run of up to 16 points is summed by that
// thread with its index loads requested together and four gathers in flight; a longer one (a 64-beam sweep puts hundreds of
// returns into a near voxel: one thread chasing them one gather at a time set the kernel's 62 us) is queued and summed by a
// wavefront: 64 coalesced index loads and gathers per step, then 64 scalar-broadcast adds in order.
constexpr int kVoxBigShortRun = 16;
__global__ void __launch_bounds__(256)
vox_big_centroid_kernel(const float4* __restrict__ pts, const int* __restrict__ idx, const unsigned* __restrict__ skeys,
                        const unsigned* __restrict__ svals, const int* __restrict__ head, const int* __restrict__ pos, int n_cap,
                        const VoxBigDesc* __restrict__ desc, float4* __restrict__ out, int* __restrict__ m_out, int* __restrict__ flag) {
  __shared__ int s_long[256][2];
  __shared__ int s_nlong;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k = blockIdx.x * blockDim.x + tid;
  if (!desc->active) return;
  const int n = desc->n;
  if (tid == 0) s_nlong = 0;
  __syncthreads();
  if (k == n - 1) { *m_out = pos[k]; *flag = 0; }
  if (k < n && head[k]) {
    const unsigned key = skeys[k];
    int lo = k, step = 1;                                    // skeys[lo] == key; gallop to the first position that is not
    while (lo + step < n && skeys[lo + step] == key) { lo += step; step <<= 1; }
    int hi = min(lo + step, n);                              // skeys[hi] != key (or hi == n)
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (skeys[mid] == key) lo = mid; else hi = mid; }
    const int len = hi - k;
    if (len <= kVoxBigShortRun) {
      int a[kVoxBigShortRun];
#pragma unroll
      for (int u = 0; u < kVoxBigShortRun; u++) { const int v = (int)svals[min(k + u, n - 1)]; a[u] = idx ? idx[v] : v; }
      float sx = 0.f, sy = 0.f, sz = 0.f, sw = 0.f;
#pragma unroll
      for (int u0 = 0; u0 < kVoxBigShortRun; u0 += 4) {
        if (u0 >= len) break;
        float4 p[4];
#pragma unroll
        for (int u = 0; u < 4; u++) p[u] = pts[a[u0 + u]];
#pragma unroll
        for (int u = 0; u < 4; u++) if (u0 + u < len) { sx += p[u].x; sy += p[u].y; sz += p[u].z; sw += p[u].w; }
      }
      const float c = (float)len;
      out[pos[k] - 1] = make_float4(sx / c, sy / c, sz / c, sw / c);
    } else {
      const int q = atomicAdd(&s_nlong, 1);
      s_long[q][0] = k; s_long[q][1] = len;
    }
  }
  __syncthreads();
  const int nl = s_nlong;
  for (int q = wave; q < nl; q += 4) {
    const int k0 = s_long[q][0], len = s_long[q][1];
    float sx = 0.f, sy = 0.f, sz = 0.f, sw = 0.f;
    for (int base = 0; base < len; base += 64) {
      const int cnt = min(64, len - base);
      float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
      if (lane < cnt) { const int v = (int)svals[k0 + base + lane]; p = pts[idx ? idx[v] : v]; }
      for (int t = 0; t < cnt; t++) {
        sx += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p.x), t));
        sy += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p.y), t));
        sz += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p.z), t));
        sw += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p.w), t));
      }
    }
    if (lane == 0) { const float c = (float)len; out[pos[k0] - 1] = make_float4(sx / c, sy / c, sz / c, sw / c); }
  }
}

}  // namespace msfl

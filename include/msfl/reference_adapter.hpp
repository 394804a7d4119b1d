// msfl/reference_adapter.hpp — the reference-side binding of libmsfl_hip.so as templates over the CALLER's types.
//
// One set of function bodies that compiles unchanged on both sides of the boundary:
//   * inside MSF_LOAM, instantiated with pcl::PointCloud<pcl::PointXYZI / PointXYZIRT> (32-byte points, common/common.h:44-62),
//     TimestampedPointCloud<T> (common/timestamped_pointcloud.h:11-42), Rigid3d = Rigid3<double> over Eigen
//     (common/rigid_transform.h:36-128), IntegrationBase (slam/imu_fusion/integration_base.h:62-69), Eigen::Vector3d;
//   * in this repository, instantiated with the dependency-free mirror PODs of msfl/scan_matcher.hpp (whose classes are now thin
//     wrappers of these functions) and, in tests/cpp/adapter_check.cpp, with PCL/Eigen-SHAPED test types that have the reference's
//     32-byte point layout and Eigen's accessor style — both run on the GPU and must agree bit for bit
//     (tests/test_cpp_host_mirror.py).
//
// What a type has to offer (duck-typed; nothing here includes PCL or Eigen):
//   cloud        size(), operator[](i) -> point, push_back(point); points with .x .y .z .intensity (and .ring, .time for the
//                sensor cloud)                                              pcl::PointCloud<T>
//   stamped      members cloud_full_res, cloud_corner_sharp, cloud_corner_less_sharp, cloud_surf_flat, cloud_surf_less_flat that
//                dereference (*p) to a cloud                                TimestampedPointCloud<T>
//   rigid        ToVector7() -> something indexable [0..6] = [t, qx qy qz qw] and default-constructible; a constructor from that
//                same vector type (NO normalisation, rigid_transform.h:47-49) Rigid3d
//   vec3         operator[](0..2) readable / writable                       Eigen::Vector3d, std::array<double, 3>
//   quaternion   .x() .y() .z() .w()  OR  operator[](0..3) in [x y z w] order Eigen::Quaterniond, std::array<double, 4>
//   integration  members sum_dt_buf_ (vector<double>), delta_q_buf_ (vector<quaternion>), delta_p_buf_ (vector<vec3>)
//
// Call sites in the reference (INTEGRATION.md has the three of them):
//   msfl::adapter::Extract(h, laser_cloud_in, g_lidar2imu_transfrom, &scan)            msf_loam_node.cc:170-371
//   msfl::adapter::MatchScan2Scan(h, scan_last, scan_curr, pose_estimate_curr2last)    odometry_scan_matcher.cc:43-285
//   msfl::adapter::MatchScan2Map(h, cloud_map, scan_curr, is_initialized, preintegration, gravity_vector,
//                                pose_estimate_map_scan2world, velocity)               mapping_scan_matcher.cc:61-278 (after :28-59)
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../msfl_c_api.h"

namespace msfl {
namespace adapter {

template <class...> struct voider { using type = void; };

// the reference aborts through glog CHECK on invariant violations; here: an exception naming the call and the library's message
inline void Check(msfl_status s, msfl_handle* h, const char* what) {
  if (s != MSFL_OK) throw std::runtime_error(std::string(what) + ": " + msfl_status_string(s) + " " + (h ? msfl_last_error(h) : ""));
}

// ---- quaternion -> [x y z w] (Eigen::Quaterniond::coeffs() order)
template <class Q, class = void>
struct QuatXYZW {
  static void Get(const Q& q, double o[4]) { for (int k = 0; k < 4; ++k) o[k] = static_cast<double>(q[k]); }
};
template <class Q>
struct QuatXYZW<Q, typename voider<decltype(std::declval<const Q&>().w())>::type> {
  static void Get(const Q& q, double o[4]) { o[0] = q.x(); o[1] = q.y(); o[2] = q.z(); o[3] = q.w(); }
};

// ---- rigid <-> 7 doubles.  The reference's ToVector7() is not const-qualified (rigid_transform.h:59): work on a copy.
template <class RigidT>
inline void RigidToArray(const RigidT& r, double o[7]) {
  RigidT c = r;
  const auto v = c.ToVector7();
  for (int k = 0; k < 7; ++k) o[k] = static_cast<double>(v[k]);
}
template <class RigidT>
inline RigidT RigidFromArray(const double a[7]) {
  using V7 = typename std::decay<decltype(std::declval<RigidT&>().ToVector7())>::type;
  V7 v;
  for (int k = 0; k < 7; ++k) v[k] = a[k];
  return RigidT(v);                                                     // no normalisation: an un-stepped pose round-trips bit for bit
}

// ---- clouds -> 16-byte msfl_point {x, y, z, t = intensity} (the reference keeps the relative time in `intensity`, msf_loam_node.cc:152-153)
template <class CloudT>
inline std::vector<msfl_point> Pack(const CloudT& c) {
  std::vector<msfl_point> o(c.size());
  for (std::size_t i = 0; i < c.size(); ++i) { const auto& p = c[i]; o[i] = msfl_point{p.x, p.y, p.z, p.intensity}; }
  return o;
}
template <class CloudT>
inline void PackWithRing(const CloudT& c, std::vector<msfl_point>* pts, std::vector<std::uint16_t>* ring) {
  pts->resize(c.size()); ring->resize(c.size());
  for (std::size_t i = 0; i < c.size(); ++i) {
    const auto& p = c[i];
    (*pts)[i] = msfl_point{p.x, p.y, p.z, p.intensity};
    (*ring)[i] = static_cast<std::uint16_t>(p.ring);
  }
}
// the point type of a cloud, and a point of it from a library point (ring / time only where the type has them)
template <class CloudT>
using PointOf = typename std::decay<decltype(std::declval<const CloudT&>()[0])>::type;
template <class P, class = void>
struct HasRing : std::false_type {};
template <class P>
struct HasRing<P, typename voider<decltype(std::declval<P&>().ring)>::type> : std::true_type {};
template <class P>
inline typename std::enable_if<HasRing<P>::value, P>::type MakePoint(const msfl_point& q, std::uint16_t ring) {
  P p{};
  p.x = q.x; p.y = q.y; p.z = q.z; p.intensity = q.t; p.ring = ring; p.time = q.t;       // time == intensity, msf_loam_node.cc:152-153
  return p;
}
template <class P>
inline typename std::enable_if<!HasRing<P>::value, P>::type MakePoint(const msfl_point& q, std::uint16_t) {
  P p{};
  p.x = q.x; p.y = q.y; p.z = q.z; p.intensity = q.t;
  return p;
}

// ---- RealHandleLaserCloudMessage between pcl::fromROSMsg and AddLaserScan (msf_loam_node.cc:170-371): fills the five clouds of
// `scan` (cleared first); `scan->time`, `frame_id` and the poses stay the caller's.
template <class CloudT, class RigidT, class StampedT>
inline void Extract(msfl_handle* h, const CloudT& laser_cloud_in, const RigidT& lidar2imu, StampedT* scan) {
  std::vector<msfl_point> p; std::vector<std::uint16_t> r;
  PackWithRing(laser_cloud_in, &p, &r);
  const std::size_t n = p.size();
  std::vector<msfl_point> full(n); std::vector<std::uint16_t> ring(n); std::vector<float> curv(n); std::vector<std::uint8_t> label(n);
  std::vector<int> idx[4];
  for (auto& v : idx) v.resize(n);
  msfl_features f{};
  f.full_pts = full.data(); f.full_ring = ring.data(); f.curvature = curv.data(); f.label = label.data();
  f.sharp_idx = idx[0].data(); f.less_sharp_idx = idx[1].data(); f.flat_idx = idx[2].data(); f.less_flat_idx = idx[3].data();
  double ext[7];
  RigidToArray(lidar2imu, ext);
  Check(msfl_extract_features(h, p.data(), r.data(), static_cast<int>(n), ext, &f, MSFL_MEM_HOST), h, "msfl_extract_features");
  using P = PointOf<typename std::decay<decltype(*scan->cloud_full_res)>::type>;
  auto fill = [&](decltype(*scan->cloud_full_res)& out, const int* list, int m) {
    out = typename std::decay<decltype(out)>::type();
    for (int k = 0; k < m; ++k) { const int i = list ? list[k] : k; out.push_back(MakePoint<P>(full[i], ring[i])); }
  };
  fill(*scan->cloud_full_res, nullptr, f.n_full);
  fill(*scan->cloud_corner_sharp, idx[0].data(), f.n_sharp);                 // msf_loam_node.cc:360-364
  fill(*scan->cloud_corner_less_sharp, idx[1].data(), f.n_less_sharp);
  fill(*scan->cloud_surf_flat, idx[2].data(), f.n_flat);
  fill(*scan->cloud_surf_less_flat, idx[3].data(), f.n_less_flat);
}

// ---- OdometryScanMatcher::MatchScan2Scan (odometry_scan_matcher.cc:43-285).  Returns false exactly where the reference does
// (< 10 correspondences, :262-267); the pose then holds the outer iterations completed so far.
template <class StampedT, class RigidT>
inline bool MatchScan2Scan(msfl_handle* h, const StampedT& scan_last, const StampedT& scan_curr, RigidT* pose_estimate_curr2last,
                           msfl_match_info* info = nullptr) {
  std::vector<msfl_point> p[4]; std::vector<std::uint16_t> r[4];
  PackWithRing(*scan_last.cloud_corner_less_sharp, &p[0], &r[0]);
  PackWithRing(*scan_last.cloud_surf_less_flat, &p[1], &r[1]);
  PackWithRing(*scan_curr.cloud_corner_sharp, &p[2], &r[2]);
  PackWithRing(*scan_curr.cloud_surf_flat, &p[3], &r[3]);
  msfl_ring_cloud c[4];
  for (int k = 0; k < 4; ++k) c[k] = msfl_ring_cloud{p[k].data(), r[k].data(), static_cast<int>(p[k].size())};
  double v[7];
  RigidToArray(*pose_estimate_curr2last, v);
  const msfl_status s = msfl_match_scan2scan(h, &c[0], &c[1], &c[2], &c[3], v, info, MSFL_MEM_HOST);
  if (s != MSFL_OK && s != MSFL_TOO_FEW_CORRESPONDENCES) Check(s, h, "msfl_match_scan2scan");
  *pose_estimate_curr2last = RigidFromArray<RigidT>(v);
  return s == MSFL_OK;
}

// ---- what GetDeltaQP reads of an IntegrationBase (scan_undistortion.cc:22-42), flattened for msfl_preintegration
struct PreintegrationView {
  std::vector<double> dq, dp;
  msfl_preintegration pre{};
  template <class IntegrationT>
  explicit PreintegrationView(const IntegrationT& ib) {
    const std::size_t n = ib.sum_dt_buf_.size();
    if (n < 2 || ib.delta_q_buf_.size() != n || ib.delta_p_buf_.size() != n)
      throw std::invalid_argument("pre-integration needs >= 2 samples and buffers of equal length");
    dq.resize(4 * n); dp.resize(3 * n);
    for (std::size_t i = 0; i < n; ++i) {
      using Q = typename std::decay<decltype(ib.delta_q_buf_[i])>::type;
      QuatXYZW<Q>::Get(ib.delta_q_buf_[i], &dq[4 * i]);
      for (int k = 0; k < 3; ++k) dp[3 * i + k] = static_cast<double>(ib.delta_p_buf_[i][k]);
    }
    pre.sum_dt = ib.sum_dt_buf_.data(); pre.delta_q = dq.data(); pre.delta_p = dp.data(); pre.n = static_cast<int>(n);
  }
};

// ---- MappingScanMatcher::MatchScan2Map (mapping_scan_matcher.cc:61-278).
// Only cloud_corner_less_sharp and cloud_surf_less_flat of both arguments are read (:71-72,109,179).
//   !is_initialized : the LiDAR-only branch (:96,123); preintegration / gravity_vector / velocity are not read.
//    is_initialized : the caller has ALREADY run its IMU-only pre-solve (:28-59, a 15-residual Ceres problem that stays on the
//                     maintainer's side) and stored pose_j / bias_j.head<3>() in *pose / *velocity (:58-59); GetDeltaQP per
//                     feature point (:112-116,182-186) runs on the GPU, then the Deskew factors with the velocity block held
//                     constant (:94).  A feature time outside the pre-integration span aborts in the reference (CHECK,
//                     scan_undistortion.cc:26-30): exception here.
// Always returns true like the reference (:277) unless the map is unusable (MAP_TOO_SMALL: the caller's gate
// laser_mapping.cc:284-285 normally prevents that) -> false, pose untouched.
template <class StampedT, class IntegrationT, class Vec3T, class RigidT>
inline bool MatchScan2Map(msfl_handle* h, const StampedT& cloud_map, const StampedT& scan_curr, const bool is_initialized,
                          const std::shared_ptr<IntegrationT>& preintegration, const Vec3T& gravity_vector,
                          RigidT* pose_estimate_map_scan2world, Vec3T* velocity, msfl_match_info* info = nullptr) {
  const std::vector<msfl_point> mc = Pack(*cloud_map.cloud_corner_less_sharp);
  const std::vector<msfl_point> ms = Pack(*cloud_map.cloud_surf_less_flat);
  Check(msfl_set_map(h, mc.data(), static_cast<int>(mc.size()), ms.data(), static_cast<int>(ms.size()), MSFL_MEM_HOST), h,
        "msfl_set_map");                                                 // the two kd-tree builds, :66-73
  const std::vector<msfl_point> c = Pack(*scan_curr.cloud_corner_less_sharp);
  const std::vector<msfl_point> s = Pack(*scan_curr.cloud_surf_less_flat);
  double v[7];
  RigidToArray(*pose_estimate_map_scan2world, v);
  msfl_status st;
  if (!is_initialized) {
    st = msfl_match_scan2map(h, c.data(), static_cast<int>(c.size()), s.data(), static_cast<int>(s.size()), v, info, MSFL_MEM_HOST);
  } else {
    if (!preintegration || !velocity) throw std::invalid_argument("MatchScan2Map: is_initialized needs a pre-integration and a velocity");
    const PreintegrationView pv(*preintegration);
    std::vector<double> cdq(4 * c.size()), cdp(3 * c.size()), sdq(4 * s.size()), sdp(3 * s.size());
    if (!c.empty()) Check(msfl_delta_qp(h, &pv.pre, c.data(), static_cast<int>(c.size()), cdq.data(), cdp.data(), MSFL_MEM_HOST), h, "msfl_delta_qp (corner)");
    if (!s.empty()) Check(msfl_delta_qp(h, &pv.pre, s.data(), static_cast<int>(s.size()), sdq.data(), sdp.data(), MSFL_MEM_HOST), h, "msfl_delta_qp (surf)");
    msfl_deskew d;
    d.corner_dq = c.empty() ? nullptr : cdq.data(); d.corner_dp = c.empty() ? nullptr : cdp.data();
    d.surf_dq = s.empty() ? nullptr : sdq.data(); d.surf_dp = s.empty() ? nullptr : sdp.data();
    for (int a = 0; a < 3; ++a) { d.velocity[a] = static_cast<double>((*velocity)[a]); d.gravity[a] = static_cast<double>(gravity_vector[a]); }
    st = msfl_match_scan2map_deskew(h, c.data(), static_cast<int>(c.size()), s.data(), static_cast<int>(s.size()), &d, v, info);
  }
  if (st == MSFL_MAP_TOO_SMALL) return false;
  Check(st, h, "msfl_match_scan2map");
  *pose_estimate_map_scan2world = RigidFromArray<RigidT>(v);               // :271 / :268 (velocity unchanged: its block is constant, :94)
  return true;
}

// ---- HybridGrid::GetSurroundedCloud / InsertScan (hybrid_grid.cc:462-534; callers laser_mapping.cc:273-278,330-338) on a
// device-resident store created with msfl_grid_create(h, resolution, leaf, &g)
template <class CloudT, class RigidT>
inline void GetSurroundedCloud(msfl_handle* h, msfl_grid* g, const CloudT& scan, const RigidT& pose, CloudT* out) {
  const std::vector<msfl_point> in = Pack(scan);
  int n_pts = 0, n_cells = 0;
  Check(msfl_grid_size(g, &n_pts, &n_cells), h, "msfl_grid_size");
  std::vector<msfl_point> buf(static_cast<std::size_t>(n_pts > 0 ? n_pts : 1));
  int n_out = 0;
  double v[7];
  RigidToArray(pose, v);
  Check(msfl_grid_get_surrounded(g, in.data(), static_cast<int>(in.size()), v, buf.data(), n_pts, &n_out, MSFL_MEM_HOST), h,
        "msfl_grid_get_surrounded");
  *out = CloudT();
  for (int i = 0; i < n_out; ++i) out->push_back(MakePoint<PointOf<CloudT>>(buf[i], 0));
}
template <class CloudT>
inline void InsertScan(msfl_handle* h, msfl_grid* g, const CloudT& scan) {
  if (scan.size() == 0) return;                                           // hybrid_grid.cc:504
  const std::vector<msfl_point> in = Pack(scan);
  Check(msfl_grid_insert_scan(g, in.data(), static_cast<int>(in.size()), MSFL_MEM_HOST), h, "msfl_grid_insert_scan");
}

// ---- DoUndistort of LaserMapping::Run (laser_mapping.cc:197-211) and ScanUndistortionUtils::DoUndistort
// (scan_undistortion.cc:5-19) on one cloud in place (x, y, z rewritten; intensity / ring / time kept)
template <class CloudT, class IntegrationT, class QuatT, class Vec3T>
inline void DeskewCloud(msfl_handle* h, const IntegrationT& preintegration, const QuatT& rot_odom_scan2world, const Vec3T& velocity,
                        const Vec3T& gravity, CloudT* cloud) {
  std::vector<msfl_point> pts = Pack(*cloud);
  const PreintegrationView pv(preintegration);
  double q[4], vel[3], g[3];
  QuatXYZW<QuatT>::Get(rot_odom_scan2world, q);
  for (int a = 0; a < 3; ++a) { vel[a] = static_cast<double>(velocity[a]); g[a] = static_cast<double>(gravity[a]); }
  Check(msfl_deskew_cloud(h, &pv.pre, pts.data(), static_cast<int>(pts.size()), q, vel, g, MSFL_MEM_HOST), h, "msfl_deskew_cloud");
  for (std::size_t i = 0; i < pts.size(); ++i) { auto& p = (*cloud)[i]; p.x = pts[i].x; p.y = pts[i].y; p.z = pts[i].z; }
}
template <class CloudT, class IntegrationT>
inline void UndistortCloud(msfl_handle* h, const IntegrationT& preintegration, CloudT* cloud) {
  std::vector<msfl_point> pts = Pack(*cloud);
  const PreintegrationView pv(preintegration);
  Check(msfl_undistort_cloud(h, &pv.pre, pts.data(), static_cast<int>(pts.size()), MSFL_MEM_HOST), h, "msfl_undistort_cloud");
  for (std::size_t i = 0; i < pts.size(); ++i) { auto& p = (*cloud)[i]; p.x = pts[i].x; p.y = pts[i].y; p.z = pts[i].z; }
}

}  // namespace adapter
}  // namespace msfl

// msfl/scan_matcher.hpp — C++ host-side mirror of MSF_LOAM's scan-matching interface over the
// C ABI (include/msfl_c_api.h).  Header-only, C++14, no PCL / Eigen / Ceres.
//
// Same class names, method names, argument meaning and error behaviour as the reference:
//   msfl::OdometryScanMatcher::MatchScan2Scan   <- src/slam/local/scan_matching/odometry_scan_matcher.h:10-12
//   msfl::MappingScanMatcher::MatchScan2Map     <- src/slam/local/scan_matching/mapping_scan_matcher.h:14-21
//   msfl::ScanRegistration::Extract             <- RealHandleLaserCloudMessage, src/msf_loam_node.cc:160-378
//   msfl::HybridGrid::{GetSurroundedCloud,InsertScan} <- src/slam/map/hybrid_grid.h:27-39 (next row N1)
//   msfl::Rigid3d                               <- src/common/rigid_transform.h:36-128
//   msfl::TimestampedPointCloud<T>              <- src/common/timestamped_pointcloud.h:11-42
//   msfl::PointXYZI / PointXYZIRT               <- pcl::PointXYZI / src/common/common.h:44-62
// The point types here are packed PODs.  The marshalling itself lives in msfl/reference_adapter.hpp as templates over the
// caller's cloud / rigid / vector types: these classes are its instantiation with the PODs below, the reference tree
// instantiates the SAME functions with its PCL / Eigen types.  Everything heavy runs in libmsfl_hip.so on the
// GPU; these classes only marshal.  A matcher object owns one msfl_handle (one HIP stream), exactly as the
// reference owns one matcher per thread (laser_odometry.h:29, laser_mapping.h:65).
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../msfl_c_api.h"
#include "reference_adapter.hpp"

namespace msfl {

struct PointXYZI { float x, y, z, intensity; };                       // 16 B == msfl_point
struct PointXYZIRT { float x, y, z, intensity; std::uint16_t ring; float time; };
using PointType = PointXYZI;                                            // common.h:64
using PointTypeOriginal = PointXYZIRT;                                  // common.h:69

template <typename T>
struct PointCloud {
  std::vector<T> points;
  std::size_t size() const { return points.size(); }
  bool empty() const { return points.empty(); }
  void push_back(const T& p) { points.push_back(p); }
  T& operator[](std::size_t i) { return points[i]; }
  const T& operator[](std::size_t i) const { return points[i]; }
};

using Vector3d = std::array<double, 3>;

// rigid_transform.h:36-128.  rotation stored [x y z w] like Eigen::Quaterniond::coeffs().
class Rigid3d {
 public:
  Rigid3d() : t_{{0, 0, 0}}, q_{{0, 0, 0, 1}} {}
  Rigid3d(const Vector3d& t, const std::array<double, 4>& q_xyzw) : t_(t), q_(q_xyzw) {}
  explicit Rigid3d(const std::array<double, 7>& v) : t_{{v[0], v[1], v[2]}}, q_{{v[3], v[4], v[5], v[6]}} {}  // :47-49, no normalisation
  std::array<double, 7> ToVector7() const { return {{t_[0], t_[1], t_[2], q_[0], q_[1], q_[2], q_[3]}}; }  // :59-64
  static Rigid3d Identity() { return Rigid3d(); }
  const Vector3d& translation() const { return t_; }
  const std::array<double, 4>& rotation() const { return q_; }
  Vector3d& translation() { return t_; }
  std::array<double, 4>& rotation() { return q_; }
  Vector3d operator*(const Vector3d& p) const {                        // :113-118
    const Vector3d r = Rotate(q_, p);
    return {{r[0] + t_[0], r[1] + t_[1], r[2] + t_[2]}};
  }
  Rigid3d inverse() const {                                            // :79-83
    const std::array<double, 4> c{{-q_[0], -q_[1], -q_[2], q_[3]}};
    const Vector3d r = Rotate(c, t_);
    return Rigid3d({{-r[0], -r[1], -r[2]}}, c);
  }
  friend Rigid3d operator*(const Rigid3d& a, const Rigid3d& b) {       // :105-111 (renormalises)
    const Vector3d r = Rotate(a.q_, b.t_);
    std::array<double, 4> q{{a.q_[3] * b.q_[0] + a.q_[0] * b.q_[3] + a.q_[1] * b.q_[2] - a.q_[2] * b.q_[1],
                             a.q_[3] * b.q_[1] + a.q_[1] * b.q_[3] + a.q_[2] * b.q_[0] - a.q_[0] * b.q_[2],
                             a.q_[3] * b.q_[2] + a.q_[2] * b.q_[3] + a.q_[0] * b.q_[1] - a.q_[1] * b.q_[0],
                             a.q_[3] * b.q_[3] - a.q_[0] * b.q_[0] - a.q_[1] * b.q_[1] - a.q_[2] * b.q_[2]}};
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (n > 0) for (double& c : q) c /= n;
    return Rigid3d({{r[0] + a.t_[0], r[1] + a.t_[1], r[2] + a.t_[2]}}, q);
  }

 private:
  static Vector3d Rotate(const std::array<double, 4>& q, const Vector3d& v) {   // Eigen _transformVector
    Vector3d uv{{q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]}};
    for (double& c : uv) c += c;
    const Vector3d c{{q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2], q[0] * uv[1] - q[1] * uv[0]}};
    return {{v[0] + q[3] * uv[0] + c[0], v[1] + q[3] * uv[1] + c[1], v[2] + q[3] * uv[2] + c[2]}};
  }
  Vector3d t_;
  std::array<double, 4> q_;
};

// timestamped_pointcloud.h:11-42
template <typename T>
struct TimestampedPointCloud {
  using PointCloudType = PointCloud<T>;
  using PointCloudTypePtr = std::shared_ptr<PointCloudType>;
  double time = 0.0;
  std::string frame_id;
  Rigid3d odom_pose, map_pose;
  PointCloudTypePtr cloud_full_res, cloud_corner_sharp, cloud_corner_less_sharp, cloud_surf_flat, cloud_surf_less_flat;
  TimestampedPointCloud()
      : cloud_full_res(new PointCloudType), cloud_corner_sharp(new PointCloudType),
        cloud_corner_less_sharp(new PointCloudType), cloud_surf_flat(new PointCloudType),
        cloud_surf_less_flat(new PointCloudType) {}
};

// Inputs of the is_initialized branch that the reference derives from `preintegration`,
// `gravity_vector` and `prev_state` (mapping_scan_matcher.cc:112-121): GetDeltaQP(preintegration, dt)
// per feature point (scan_undistortion.cc:22-42), computed by the caller.
struct DeskewInputs {
  std::vector<std::array<double, 4>> corner_delta_q, surf_delta_q;   // [x y z w]
  std::vector<Vector3d> corner_delta_p, surf_delta_p;
  Vector3d gravity_vector{{0, 0, 0}};
};

// What the scan matcher reads of the reference's IMU types (views with the reference's member names; the
// pre-integration itself — imu_fusion/integration_base.h — is outside the hot path):
//   IntegrationBase   integration_base.h:62-69: per IMU sample the cumulative time and the pre-integrated
//                     rotation / position, read by GetDeltaQP (scan_undistortion.cc:22-42)
//   RobotState        slam/estimator/estimator.h:10-19
using Quaterniond = std::array<double, 4>;                             // [x y z w] like Eigen::Quaterniond::coeffs()
struct IntegrationBase {
  std::vector<double> sum_dt_buf_;
  std::vector<Vector3d> delta_p_buf_;
  std::vector<Quaterniond> delta_q_buf_;
};
struct RobotState {
  double time = 0.0;
  Vector3d p{{0, 0, 0}}, v{{0, 0, 0}};
  Quaterniond q{{0, 0, 0, 1}};
  Vector3d bg{{0, 0, 0}}, ba{{0, 0, 0}};
  std::shared_ptr<IntegrationBase> imu_preintegration;
};

namespace detail {
using adapter::Check;
inline std::vector<msfl_point> Pack(const PointCloud<PointXYZI>& c) { return adapter::Pack(c); }
inline void Pack(const PointCloud<PointXYZIRT>& c, std::vector<msfl_point>* p, std::vector<std::uint16_t>* r) { adapter::PackWithRing(c, p, r); }
}  // namespace detail

// scan_matcher.h:13-22.  RefineByRejectOutliersWithThreshold is a no-op in the reference
// (scan_matcher.cc:13-38, body commented out) and is therefore absent.
class ScanMatcher {
 public:
  explicit ScanMatcher(int device = 0, const msfl_params* params = nullptr) {
    detail::Check(msfl_create(params, device, &h_), nullptr, "msfl_create");
  }
  virtual ~ScanMatcher() { msfl_destroy(h_); }
  ScanMatcher(const ScanMatcher&) = delete;
  ScanMatcher& operator=(const ScanMatcher&) = delete;
  msfl_handle* handle() const { return h_; }
  const msfl_match_info& last_info() const { return info_; }

 protected:
  msfl_handle* h_ = nullptr;
  msfl_match_info info_{};
};

class OdometryScanMatcher : public ScanMatcher {
 public:
  using ScanMatcher::ScanMatcher;
  // Returns false exactly where the reference does (< 10 correspondences, .cc:262-267); the pose
  // then holds the outer iterations completed so far.
  virtual bool MatchScan2Scan(const TimestampedPointCloud<PointTypeOriginal>& scan_last,
                              const TimestampedPointCloud<PointTypeOriginal>& scan_curr,
                              Rigid3d* pose_estimate_curr2last) {
    return adapter::MatchScan2Scan(h_, scan_last, scan_curr, pose_estimate_curr2last, &info_);
  }
};

class MappingScanMatcher : public ScanMatcher {
 public:
  using ScanMatcher::ScanMatcher;
  // The IMU-only pre-solve of the is_initialized branch (mapping_scan_matcher.cc:28-59): ONE IMUFactor over
  // prev_state.imu_preintegration, pose_i / bias_i constant, 6 LM iterations, whose pose_j and bias_j.head<3>() REPLACE
  // the incoming *pose_estimate_map_scan2world and *velocity (.cc:58-59) before the LiDAR loop starts.  It is a 15-residual
  // CPU problem on the maintainer's side of the boundary (Ceres + IMUFactor, SURVEY.md 8f N3: "the pre-solve stays on the
  // CPU"), so it is installed as a hook: called with prev_state, it must write the two outputs.  The eight-argument
  // MatchScan2Map refuses to run the is_initialized branch without it rather than silently start from the caller's pose.
  using ImuPresolve = std::function<void(const RobotState& prev_state, Rigid3d* pose_j, Vector3d* velocity_j)>;
  void SetImuPresolve(ImuPresolve f) { imu_presolve_ = std::move(f); }

  // cloud_map / scan_curr: only cloud_corner_less_sharp and cloud_surf_less_flat are read
  // (mapping_scan_matcher.cc:71-72,109,179).  `deskew` non-null selects the is_initialized branch
  // (per-point time comes from PointXYZI::intensity, .cc:114); `velocity` is read as Vi and written
  // back unchanged (the reference holds the velocity block constant, .cc:94,269).
  // Always returns true like the reference (.cc:277) unless the map is unusable (MAP_TOO_SMALL:
  // the caller's gate laser_mapping.cc:284-285 normally prevents that) -> false, pose untouched.
  bool MatchScan2Map(const TimestampedPointCloud<PointType>& cloud_map,
                     const TimestampedPointCloud<PointType>& scan_curr,
                     const bool is_initialized, const DeskewInputs* deskew,
                     Rigid3d* pose_estimate_map_scan2world, Vector3d* velocity) {
    const std::vector<msfl_point> mc = detail::Pack(*cloud_map.cloud_corner_less_sharp);
    const std::vector<msfl_point> ms = detail::Pack(*cloud_map.cloud_surf_less_flat);
    detail::Check(msfl_set_map(h_, mc.data(), static_cast<int>(mc.size()), ms.data(), static_cast<int>(ms.size()), MSFL_MEM_HOST),
                  h_, "msfl_set_map");                                   // kd-tree build, .cc:66-73
    const std::vector<msfl_point> c = detail::Pack(*scan_curr.cloud_corner_less_sharp);
    const std::vector<msfl_point> s = detail::Pack(*scan_curr.cloud_surf_less_flat);
    auto v = pose_estimate_map_scan2world->ToVector7();
    msfl_status st;
    if (is_initialized) {
      if (!deskew || !velocity || deskew->corner_delta_q.size() != c.size() || deskew->surf_delta_q.size() != s.size() ||
          deskew->corner_delta_p.size() != c.size() || deskew->surf_delta_p.size() != s.size())
        throw std::invalid_argument("MatchScan2Map: is_initialized needs per-point deskew inputs of matching size");
      msfl_deskew d;
      d.corner_dq = c.empty() ? nullptr : deskew->corner_delta_q[0].data();
      d.corner_dp = c.empty() ? nullptr : deskew->corner_delta_p[0].data();
      d.surf_dq = s.empty() ? nullptr : deskew->surf_delta_q[0].data();
      d.surf_dp = s.empty() ? nullptr : deskew->surf_delta_p[0].data();
      for (int a = 0; a < 3; ++a) { d.velocity[a] = (*velocity)[a]; d.gravity[a] = deskew->gravity_vector[a]; }
      st = msfl_match_scan2map_deskew(h_, c.data(), static_cast<int>(c.size()), s.data(), static_cast<int>(s.size()), &d, v.data(), &info_);
    } else {
      st = msfl_match_scan2map(h_, c.data(), static_cast<int>(c.size()), s.data(), static_cast<int>(s.size()), v.data(), &info_, MSFL_MEM_HOST);
    }
    if (st == MSFL_MAP_TOO_SMALL) return false;
    detail::Check(st, h_, "msfl_match_scan2map");
    *pose_estimate_map_scan2world = Rigid3d(v);
    return true;
  }

  // The reference's own eight-parameter signature (mapping_scan_matcher.h:14-21), argument for argument.
  //   !is_initialized: LiDAR-only branch (.cc:96,123); preintegration / gravity_vector / prev_state are not read.
  //   is_initialized : GetDeltaQP(preintegration, point.intensity) for every feature point (.cc:112-116,182-186) runs on
  //                    the GPU (msfl_delta_qp), then the Deskew factors with the velocity block held constant (.cc:94).
  //                    The IMU-only pre-solve the reference runs first (.cc:28-59) is the hook installed with
  //                    SetImuPresolve: it is called with prev_state and overwrites *pose_estimate_map_scan2world and
  //                    *velocity (pose_j, bias_j.head<3>(): what the reference's loop then reads, .cc:83,107), exactly
  //                    like .cc:58-59 — the incoming pose is NOT the starting point of this branch.  Without a hook the
  //                    call throws std::logic_error.
  // A feature time outside the pre-integration span aborts in the reference (CHECK, scan_undistortion.cc:26-30): exception here.
  bool MatchScan2Map(const TimestampedPointCloud<PointType>& cloud_map,
                     const TimestampedPointCloud<PointType>& scan_curr,
                     const bool is_initialized,
                     const std::shared_ptr<IntegrationBase>& preintegration,
                     const Vector3d& gravity_vector,
                     const RobotState& prev_state,
                     Rigid3d* pose_estimate_map_scan2world,
                     Vector3d* velocity) {
    if (!is_initialized)                                               // prev_state only feeds the pre-solve and a log line (.cc:27)
      return MatchScan2Map(cloud_map, scan_curr, false, nullptr, pose_estimate_map_scan2world, velocity);
    if (!imu_presolve_)
      throw std::logic_error("MatchScan2Map(is_initialized): no IMU pre-solve installed (SetImuPresolve); the reference starts this "
                             "branch from the IMU-only solve of prev_state (mapping_scan_matcher.cc:28-59), not from the incoming pose");
    if (!pose_estimate_map_scan2world || !velocity) throw std::invalid_argument("MatchScan2Map: null pose / velocity");
    imu_presolve_(prev_state, pose_estimate_map_scan2world, velocity);  // .cc:58-59
    return adapter::MatchScan2Map(h_, cloud_map, scan_curr, true, preintegration, gravity_vector, pose_estimate_map_scan2world, velocity, &info_);
  }

 private:
  ImuPresolve imu_presolve_;
};

// RealHandleLaserCloudMessage (msf_loam_node.cc:160-378) between pcl::fromROSMsg and AddLaserScan.
class ScanRegistration {
 public:
  explicit ScanRegistration(int device = 0, double min_range = 0.3, const Rigid3d& lidar2imu = Rigid3d::Identity())
      : lidar2imu_(lidar2imu) {
    msfl_params p; msfl_default_params(&p); p.min_range = min_range;
    detail::Check(msfl_create(&p, device, &h_), nullptr, "msfl_create");
  }
  ~ScanRegistration() { msfl_destroy(h_); }
  ScanRegistration(const ScanRegistration&) = delete;
  ScanRegistration& operator=(const ScanRegistration&) = delete;

  TimestampedPointCloud<PointTypeOriginal> Extract(const PointCloud<PointTypeOriginal>& laser_cloud_in, double stamp) {
    TimestampedPointCloud<PointTypeOriginal> scan;
    scan.time = stamp;
    adapter::Extract(h_, laser_cloud_in, lidar2imu_, &scan);
    return scan;
  }

 private:
  msfl_handle* h_ = nullptr;
  Rigid3d lidar2imu_;
};

// HybridGrid (src/slam/map/hybrid_grid.h:27-39): the local map store, resident on the device.
// The reference passes the pcl::VoxelGrid filter to InsertScan; here its leaf is fixed at
// construction (laser_mapping.cc:60-68 uses one leaf per map for the whole run).
class HybridGrid {
 public:
  HybridGrid(const float& resolution, float voxel_leaf, int device = 0) {
    detail::Check(msfl_create(nullptr, device, &h_), nullptr, "msfl_create");
    detail::Check(msfl_grid_create(h_, resolution, voxel_leaf, &g_), h_, "msfl_grid_create");
  }
  virtual ~HybridGrid() { msfl_grid_destroy(g_); msfl_destroy(h_); }
  HybridGrid(const HybridGrid&) = delete;
  HybridGrid& operator=(const HybridGrid&) = delete;

  std::shared_ptr<PointCloud<PointType>> GetSurroundedCloud(const std::shared_ptr<const PointCloud<PointType>>& scan,
                                                            const Rigid3d& pose) {
    auto cloud = std::make_shared<PointCloud<PointType>>();
    adapter::GetSurroundedCloud(h_, g_, *scan, pose, cloud.get());
    return cloud;
  }

  void InsertScan(const std::shared_ptr<PointCloud<PointType>>& scan) {
    adapter::InsertScan(h_, g_, *scan);
  }

 private:
  msfl_handle* h_ = nullptr;
  msfl_grid* g_ = nullptr;
};

// LaserOdometry + LaserMapping as one device-resident pipeline (msfl_slam_*): the raw cloud of
// RealHandleLaserCloudMessage (msf_loam_node.cc:160-167) in, pose_scan2world_ (laser_odometry.cc:79) and
// pose_map_scan2world_ (laser_mapping.cc:304-311) out; the feature clouds, the two surrounded map clouds, both HybridGrids
// and the pose chain never leave the GPU.  AddLaserScan(cloud) is the LiDAR-only form; AddLaserScan(cloud, ImuInputs) runs the
// IMU branches of LaserMapping::Run (UndistortScan, laser_mapping.cc:170-176, or the is_initialized matcher + DoUndistort,
// :197-211) from what the maintainer's estimator hands over.
class LaserSlam {
 public:
  struct Poses { Rigid3d odom, map; bool mapped; int scan_index; };
  // the per-scan products of the code that stays on the maintainer's side (msfl_slam_imu)
  struct ImuInputs {
    std::shared_ptr<IntegrationBase> preintegration;       // BuildPreintegration(imu_buf_, odom_result.time), laser_mapping.cc:191-194
    bool is_initialized = false;                            // estimator_.IsInitialized()
    Vector3d velocity{{0, 0, 0}}, gravity{{0, 0, 0}};      // bias_j.head<3>() of the pre-solve, estimator_.GetGravityVector()
    Rigid3d presolved_pose = Rigid3d::Identity();           // pose_j of the IMU-only pre-solve (mapping_scan_matcher.cc:35-59)
  };

  // reference_quirks: FilterLessFlatLessCornerFeature as the reference executes it (laser_mapping.cc:340-364), see msfl_slam_config
  // keep_clouds: also produce what LaserMapping::Run publishes / accumulates (cloud_full_res after the IMU passes, its map-frame copy), see Clouds()
  explicit LaserSlam(int device = 0, int max_scan_points = 200000, int max_rings = 128, const Rigid3d& pose_odom2map = Rigid3d::Identity(),
                     bool reference_quirks = false, bool keep_clouds = false) : max_scan_points_(max_scan_points) {
    msfl_slam_config c;
    msfl_slam_default_config(&c);
    c.max_scan_points = max_scan_points; c.max_rings = max_rings; c.reference_quirks = reference_quirks ? 1 : 0; c.keep_clouds = keep_clouds ? 1 : 0;
    const auto v = pose_odom2map.ToVector7();
    for (int k = 0; k < 7; ++k) c.pose_odom2map[k] = v[k];
    const msfl_status st = msfl_slam_create(nullptr, &c, device, &s_);
    if (st != MSFL_OK) throw std::runtime_error(std::string("msfl_slam_create: ") + msfl_status_string(st));
  }
  ~LaserSlam() { msfl_slam_destroy(s_); }
  LaserSlam(const LaserSlam&) = delete;
  LaserSlam& operator=(const LaserSlam&) = delete;

  // LaserOdometry::AddLaserScan + one LaserMapping::Run iteration; waits for this scan's mapping result.
  // `mapped` is false when the map gate (laser_mapping.cc:284-285) kept MatchScan2Map from running.
  Poses AddLaserScan(const PointCloud<PointTypeOriginal>& laser_cloud_in) {
    Enqueue(laser_cloud_in, nullptr, &rec_);
    return Unpack(rec_);
  }
  Poses AddLaserScan(const PointCloud<PointTypeOriginal>& laser_cloud_in, const ImuInputs& imu) {
    Enqueue(laser_cloud_in, &imu, &rec_);
    return Unpack(rec_);
  }
  // The two-thread form: enqueue scan k (returns its index at once), fetch results one scan late with Result(k - 1); the
  // odometry chain of scan k then runs under the mapping chain of scan k - 1.
  int AddLaserScanAsync(const PointCloud<PointTypeOriginal>& laser_cloud_in) { Enqueue(laser_cloud_in, nullptr, nullptr); return n_ - 1; }
  int AddLaserScanAsync(const PointCloud<PointTypeOriginal>& laser_cloud_in, const ImuInputs& imu) { Enqueue(laser_cloud_in, &imu, nullptr); return n_ - 1; }
  Poses Result(int scan_index) {
    const msfl_status st = msfl_slam_get_result(s_, scan_index, &rec_);
    if (st != MSFL_OK) throw std::runtime_error(std::string("msfl_slam_get_result: ") + msfl_status_string(st) + " " + msfl_slam_last_error(s_));
    return Unpack(rec_);
  }
  const msfl_slam_result& last_record() const { return rec_; }

  // What the reference publishes per scan (PublishScan, laser_mapping.cc:418-440) and accumulates for its PLY dump (:214-217), for one of the
  // last two scans fed (keep_clouds): the scan as a TimestampedPointCloud whose five clouds are the de-skewed ones, and cloud_full_res in the map frame.
  struct ScanClouds { TimestampedPointCloud<PointTypeOriginal> scan; PointCloud<PointTypeOriginal> full_res_in_map; };
  ScanClouds Clouds(int scan_index) {
    const std::size_t n = static_cast<std::size_t>(max_scan_points_);
    std::vector<msfl_point> a(n), b(n); std::vector<std::uint16_t> ring(n); std::vector<int> idx[4];
    for (auto& v : idx) v.resize(n);
    msfl_slam_clouds c{};
    c.full_scan = a.data(); c.full_map = b.data(); c.ring = ring.data();
    c.sharp_idx = idx[0].data(); c.less_sharp_idx = idx[1].data(); c.flat_idx = idx[2].data(); c.less_flat_idx = idx[3].data();
    const msfl_status st = msfl_slam_get_clouds(s_, scan_index, &c, MSFL_MEM_HOST);
    if (st != MSFL_OK) throw std::runtime_error(std::string("msfl_slam_get_clouds: ") + msfl_status_string(st) + " " + msfl_slam_last_error(s_));
    ScanClouds o;
    auto at = [&](const std::vector<msfl_point>& src, int i) { return PointXYZIRT{src[i].x, src[i].y, src[i].z, src[i].t, ring[i], src[i].t}; };
    for (int i = 0; i < c.n_full; ++i) { o.scan.cloud_full_res->push_back(at(a, i)); o.full_res_in_map.push_back(at(b, i)); }
    for (int k = 0; k < c.n_sharp; ++k) o.scan.cloud_corner_sharp->push_back(at(a, idx[0][k]));
    for (int k = 0; k < c.n_less_sharp; ++k) o.scan.cloud_corner_less_sharp->push_back(at(a, idx[1][k]));
    for (int k = 0; k < c.n_flat; ++k) o.scan.cloud_surf_flat->push_back(at(a, idx[2][k]));
    for (int k = 0; k < c.n_less_flat; ++k) o.scan.cloud_surf_less_flat->push_back(at(a, idx[3][k]));
    return o;
  }

 private:
  void Enqueue(const PointCloud<PointTypeOriginal>& in, const ImuInputs* imu, msfl_slam_result* out) {
    std::vector<msfl_point> p; std::vector<std::uint16_t> r;
    detail::Pack(in, &p, &r);
    msfl_slam_imu mi{};
    msfl_preintegration pre{};
    std::vector<double> dq, dp;
    if (imu && imu->preintegration) {
      const IntegrationBase& ib = *imu->preintegration;
      const std::size_t n = ib.sum_dt_buf_.size();
      if (ib.delta_q_buf_.size() != n || ib.delta_p_buf_.size() != n) throw std::invalid_argument("LaserSlam: pre-integration buffers of different length");
      dq.resize(4 * n); dp.resize(3 * n);
      for (std::size_t i = 0; i < n; ++i) {
        for (int k = 0; k < 4; ++k) dq[4 * i + k] = ib.delta_q_buf_[i][k];
        for (int k = 0; k < 3; ++k) dp[3 * i + k] = ib.delta_p_buf_[i][k];
      }
      pre.sum_dt = ib.sum_dt_buf_.data(); pre.delta_q = dq.data(); pre.delta_p = dp.data(); pre.n = static_cast<int>(n);
      mi.pre = &pre; mi.is_initialized = imu->is_initialized ? 1 : 0;
      const auto v = imu->presolved_pose.ToVector7();
      for (int k = 0; k < 7; ++k) mi.presolved_pose[k] = v[k];
      for (int k = 0; k < 3; ++k) { mi.velocity[k] = imu->velocity[k]; mi.gravity[k] = imu->gravity[k]; }
    }
    const msfl_status st = msfl_slam_add_scan_imu(s_, p.data(), r.data(), static_cast<int>(p.size()), MSFL_MEM_HOST, imu ? &mi : nullptr, out);
    if (st != MSFL_OK) throw std::runtime_error(std::string("msfl_slam_add_scan: ") + msfl_status_string(st) + " " + msfl_slam_last_error(s_));
    ++n_;
  }
  static Poses Unpack(const msfl_slam_result& r) {
    if (r.status_extract != MSFL_OK)          // the reference CHECK-aborts on these (msf_loam_node.cc:136,186,200)
      throw std::runtime_error(std::string("feature extraction: ") + msfl_status_string(r.status_extract));
    if (r.status_imu != MSFL_OK)              // GetDeltaQP's CHECK (scan_undistortion.cc:26-30)
      throw std::runtime_error("a point's relative time lies outside the pre-integration span");
    std::array<double, 7> o, m;
    for (int k = 0; k < 7; ++k) { o[k] = r.pose_odom[k]; m[k] = r.pose_map[k]; }
    return Poses{Rigid3d(o), Rigid3d(m), r.status_mapping == MSFL_OK, r.scan_index};
  }
  msfl_slam* s_ = nullptr;
  msfl_slam_result rec_{};
  int n_ = 0;
  int max_scan_points_ = 0;
};

}  // namespace msfl

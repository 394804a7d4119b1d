// include/msfl/reference_adapter.hpp instantiated with TWO type families, same function bodies, on the GPU:
//   -DADAPTER_TYPES=0   the dependency-free mirror PODs of msfl/scan_matcher.hpp (16-byte points, std::array poses)
//   -DADAPTER_TYPES=1   PCL / Eigen-SHAPED test types: the reference's 32-byte point layout (common/common.h:44-62: x y z pad |
//                       intensity | ring | time, 16-byte aligned), a cloud with pcl::PointCloud's `points` / size() / operator[] /
//                       push_back, vectors and quaternions with Eigen's accessor style (operator[] / x() y() z() w(), coeffs order
//                       x y z w), a Rigid3 whose ToVector7() is NOT const and whose constructor from the 7-vector does not
//                       normalise (common/rigid_transform.h:47-64)
// These test types are NOT PCL or Eigen (neither is in the image) and nothing of the reference is built with them; they only make
// the adapter's templates compile against the shapes the reference's types have.  Both binaries read the same input and must
// write the same bytes (tests/test_cpp_host_mirror.py).
// usage: adapter_check <in.bin> <out.bin>      (input layout: host_api_check.cpp)
//   out: 7 f64 odometry pose | i32 odom_ok | 5 x i32 cloud sizes (full, sharp, less_sharp, flat, less_flat)
//        | 7 f64 map pose (LiDAR-only) | 7 f64 map pose (is_initialized) | i32 n_surrounded | first 4 surrounded points (16 f32)
//        | 8 f32 = first two de-skewed less-flat points (x y z intensity)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#ifndef ADAPTER_TYPES
#define ADAPTER_TYPES 0
#endif

#if ADAPTER_TYPES == 0
#include "msfl/scan_matcher.hpp"
namespace T {
using PointI = msfl::PointXYZI;
using PointIRT = msfl::PointXYZIRT;
template <class P> using Cloud = msfl::PointCloud<P>;
template <class P> using Stamped = msfl::TimestampedPointCloud<P>;
using Rigid = msfl::Rigid3d;
using Vec3 = msfl::Vector3d;
using Quat = msfl::Quaterniond;
using Integration = msfl::IntegrationBase;
inline Rigid MakeRigid(const double* v) { return Rigid(std::array<double, 7>{{v[0], v[1], v[2], v[3], v[4], v[5], v[6]}}); }
inline Vec3 MakeVec3(double a, double b, double c) { return Vec3{{a, b, c}}; }
inline Quat MakeQuat(double x, double y, double z, double w) { return Quat{{x, y, z, w}}; }
}  // namespace T
#else
#include "msfl/reference_adapter.hpp"
namespace T {
struct alignas(16) PointI { float x, y, z, pad_; float intensity; float pad2_[3]; };                        // pcl::PointXYZI: 32 B
struct alignas(16) PointIRT { float x, y, z, pad_; float intensity; std::uint16_t ring; float time; };      // common.h:44-50: 32 B
static_assert(sizeof(PointI) == 32 && sizeof(PointIRT) == 32, "the reference's 32-byte PCL points");
static_assert(offsetof(PointIRT, intensity) == 16 && offsetof(PointIRT, ring) == 20 && offsetof(PointIRT, time) == 24, "SURVEY.md 8a T1 layout");
template <class P>
struct Cloud {                                                                                               // pcl::PointCloud<P>
  std::vector<P> points;
  unsigned width = 0, height = 1;
  bool is_dense = true;
  std::size_t size() const { return points.size(); }
  bool empty() const { return points.empty(); }
  void push_back(const P& p) { points.push_back(p); width = static_cast<unsigned>(points.size()); }
  P& operator[](std::size_t i) { return points[i]; }
  const P& operator[](std::size_t i) const { return points[i]; }
  using Ptr = std::shared_ptr<Cloud<P>>;
};
struct Vec3 {                                                                                                // Eigen::Vector3d
  double d[3];
  double& operator[](int i) { return d[i]; }
  const double& operator[](int i) const { return d[i]; }
};
struct Vec7 { double d[7]; double& operator[](int i) { return d[i]; } const double& operator[](int i) const { return d[i]; } };
struct Quat {                                                                                                // Eigen::Quaterniond (coeffs x y z w)
  double c[4];
  double x() const { return c[0]; } double y() const { return c[1]; } double z() const { return c[2]; } double w() const { return c[3]; }
};
class Rigid {                                                                                                // Rigid3<double>
 public:
  Rigid() : t_{{0, 0, 0}}, q_{{0, 0, 0, 1}} {}
  Rigid(const Vec7& v) : t_{{v[0], v[1], v[2]}}, q_{{v[3], v[4], v[5], v[6]}} {}                             // rigid_transform.h:47-49
  Vec7 ToVector7() { Vec7 v; for (int k = 0; k < 3; ++k) v[k] = t_[k]; for (int k = 0; k < 4; ++k) v[3 + k] = q_.c[k]; return v; }   // :59 (non-const)
  const Quat& rotation() const { return q_; }
 private:
  Vec3 t_; Quat q_;
};
template <class P>
struct Stamped {                                                                                             // TimestampedPointCloud<P>
  using PointCloudType = Cloud<P>;
  using PointCloudTypePtr = typename PointCloudType::Ptr;
  double time = 0;
  Rigid odom_pose, map_pose;
  PointCloudTypePtr cloud_full_res, cloud_corner_sharp, cloud_corner_less_sharp, cloud_surf_flat, cloud_surf_less_flat;
  Stamped() : cloud_full_res(new PointCloudType), cloud_corner_sharp(new PointCloudType), cloud_corner_less_sharp(new PointCloudType),
              cloud_surf_flat(new PointCloudType), cloud_surf_less_flat(new PointCloudType) {}
};
struct Integration {                                                                                         // integration_base.h:62-69
  std::vector<double> sum_dt_buf_;
  std::vector<Vec3> delta_p_buf_;
  std::vector<Quat> delta_q_buf_;
};
inline Rigid MakeRigid(const double* v) { Vec7 a; for (int k = 0; k < 7; ++k) a[k] = v[k]; return Rigid(a); }
inline Vec3 MakeVec3(double a, double b, double c) { return Vec3{{a, b, c}}; }
inline Quat MakeQuat(double x, double y, double z, double w) { return Quat{{x, y, z, w}}; }
}  // namespace T
#endif

template <class X>
static std::vector<X> rd(FILE* f, std::size_t n) { std::vector<X> v(n); if (n && fread(v.data(), sizeof(X), n, f) != n) { perror("read"); exit(2); } return v; }

int main(int argc, char** argv) {
  if (argc < 3) return 1;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 1;
  const int n = rd<int>(f, 1)[0];
  auto pts = rd<float>(f, 4 * (std::size_t)n);
  auto ring = rd<std::uint16_t>(f, n);
  const int mc = rd<int>(f, 1)[0]; auto mcp = rd<float>(f, 4 * (std::size_t)mc);
  const int ms = rd<int>(f, 1)[0]; auto msp = rd<float>(f, 4 * (std::size_t)ms);
  auto guess = rd<double>(f, 7);
  const int n_pre = rd<int>(f, 1)[0];
  auto sum_dt = rd<double>(f, n_pre); auto dq = rd<double>(f, 4 * (std::size_t)n_pre); auto dp = rd<double>(f, 3 * (std::size_t)n_pre);
  auto vel_in = rd<double>(f, 3); auto grav = rd<double>(f, 3);
  fclose(f);
  namespace A = msfl::adapter;

  msfl_handle *h_odo = nullptr, *h_map = nullptr;
  A::Check(msfl_create(nullptr, 0, &h_odo), nullptr, "msfl_create");
  A::Check(msfl_create(nullptr, 0, &h_map), nullptr, "msfl_create");

  // msf_loam_node.cc:160-378: the raw cloud -> the five feature clouds
  T::Cloud<T::PointIRT> cloud;
  for (int i = 0; i < n; ++i) { T::PointIRT p{}; p.x = pts[4 * i]; p.y = pts[4 * i + 1]; p.z = pts[4 * i + 2]; p.intensity = 0.f; p.ring = ring[i]; p.time = 0.f; cloud.push_back(p); }
  T::Stamped<T::PointIRT> scan;
  A::Extract(h_odo, cloud, T::Rigid(), &scan);

  // odometry_scan_matcher.cc:43-285: the scan against itself from a small offset
  const double rel0[7] = {0.05, -0.03, 0.01, 0, 0, 0.005, 0.9999875};
  T::Rigid rel = T::MakeRigid(rel0);
  const bool odo_ok = A::MatchScan2Scan(h_odo, scan, scan, &rel);

  // mapping_scan_matcher.cc:61-278, both branches
  T::Stamped<T::PointI> map, cur;
  for (int i = 0; i < mc; ++i) { T::PointI p{}; p.x = mcp[4 * i]; p.y = mcp[4 * i + 1]; p.z = mcp[4 * i + 2]; map.cloud_corner_less_sharp->push_back(p); }
  for (int i = 0; i < ms; ++i) { T::PointI p{}; p.x = msp[4 * i]; p.y = msp[4 * i + 1]; p.z = msp[4 * i + 2]; map.cloud_surf_less_flat->push_back(p); }
  for (std::size_t i = 0; i < scan.cloud_corner_less_sharp->size(); ++i) {
    const auto& q = (*scan.cloud_corner_less_sharp)[i]; T::PointI p{}; p.x = q.x; p.y = q.y; p.z = q.z; p.intensity = q.intensity; cur.cloud_corner_less_sharp->push_back(p);
  }
  for (std::size_t i = 0; i < scan.cloud_surf_less_flat->size(); ++i) {
    const auto& q = (*scan.cloud_surf_less_flat)[i]; T::PointI p{}; p.x = q.x; p.y = q.y; p.z = q.z; p.intensity = q.intensity; cur.cloud_surf_less_flat->push_back(p);
  }
  auto pre = std::make_shared<T::Integration>();
  pre->sum_dt_buf_ = sum_dt;
  for (int i = 0; i < n_pre; ++i) {
    pre->delta_q_buf_.push_back(T::MakeQuat(dq[4 * i], dq[4 * i + 1], dq[4 * i + 2], dq[4 * i + 3]));
    pre->delta_p_buf_.push_back(T::MakeVec3(dp[3 * i], dp[3 * i + 1], dp[3 * i + 2]));
  }
  const T::Vec3 gravity = T::MakeVec3(grav[0], grav[1], grav[2]);
  T::Rigid pose = T::MakeRigid(guess.data());
  T::Vec3 vel = T::MakeVec3(0, 0, 0);
  const bool ok = A::MatchScan2Map(h_map, map, cur, false, pre, gravity, &pose, &vel);
  T::Rigid pose_d = T::MakeRigid(guess.data());                 // "the pre-solve has run": its outputs are in *pose / *velocity (:58-59)
  T::Vec3 vel_d = T::MakeVec3(vel_in[0], vel_in[1], vel_in[2]);
  const bool ok_d = A::MatchScan2Map(h_map, map, cur, true, pre, gravity, &pose_d, &vel_d);

  // hybrid_grid.cc:462-534 through the device store: insert the map's surf cloud, ask for what surrounds the scan
  msfl_grid* g = nullptr;
  A::Check(msfl_grid_create(h_map, 3.0f, 0.4f, &g), h_map, "msfl_grid_create");
  A::InsertScan(h_map, g, *map.cloud_surf_less_flat);
  T::Cloud<T::PointI> around;
  T::Rigid pose_for_grid = pose;
  A::GetSurroundedCloud(h_map, g, *cur.cloud_surf_less_flat, pose_for_grid, &around);

  // laser_mapping.cc:197-211 on a copy of the less-flat cloud
  T::Cloud<T::PointI> lf = *cur.cloud_surf_less_flat;
  A::DeskewCloud(h_map, *pre, T::MakeQuat(0, 0, 0, 1), vel_d, gravity, &lf);

  FILE* o = fopen(argv[2], "wb");
  double v7[7];
  A::RigidToArray(rel, v7); fwrite(v7, 8, 7, o);
  const int ints[6] = {odo_ok ? 1 : 0, (int)scan.cloud_full_res->size(), (int)scan.cloud_corner_sharp->size(), (int)scan.cloud_corner_less_sharp->size(),
                       (int)scan.cloud_surf_flat->size(), (int)scan.cloud_surf_less_flat->size()};
  fwrite(ints, 4, 6, o);
  A::RigidToArray(pose, v7); fwrite(v7, 8, 7, o);
  A::RigidToArray(pose_d, v7); fwrite(v7, 8, 7, o);
  const int n_around = (int)around.size();
  fwrite(&n_around, 4, 1, o);
  for (int i = 0; i < 4; ++i) { const float q[4] = {around[i].x, around[i].y, around[i].z, around[i].intensity}; fwrite(q, 4, 4, o); }
  for (int i = 0; i < 2; ++i) { const float q[4] = {lf[i].x, lf[i].y, lf[i].z, lf[i].intensity}; fwrite(q, 4, 4, o); }
  fclose(o);
  msfl_grid_destroy(g);
  msfl_destroy(h_odo); msfl_destroy(h_map);
  return (ok && ok_d && n_around >= 4) ? 0 : 3;
}

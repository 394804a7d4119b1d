// ref_golden — runs the REFERENCE's own scan matchers on the inputs of tests/golden/msfl_golden_v1.npz and writes the
// poses in the layout tests/test_golden.py reads (tools/ref_golden/README.md).  Built only where MSF_LOAM builds
// (PCL, Ceres, Eigen, glog): see CMakeLists.txt next to this file.  Not buildable in this repository's image.
//
// Calls, with the reference's own types:
//   MappingScanMatcher::MatchScan2Map   src/slam/local/scan_matching/mapping_scan_matcher.h:14-21 (is_initialized = false)
//   OdometryScanMatcher::MatchScan2Scan src/slam/local/scan_matching/odometry_scan_matcher.h:10-12
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include <glog/logging.h>

#include "common/common.h"
#include "common/rigid_transform.h"
#include "common/timestamped_pointcloud.h"
#include "slam/estimator/estimator.h"
#include "slam/local/scan_matching/mapping_scan_matcher.h"
#include "slam/local/scan_matching/odometry_scan_matcher.h"

namespace {

template <class T>
std::vector<T> Read(FILE* f, std::size_t n) {
  std::vector<T> v(n);
  if (n && std::fread(v.data(), sizeof(T), n, f) != n) { std::perror("read"); std::exit(2); }
  return v;
}

// n x {x y z intensity} -> pcl::PointCloud<pcl::PointXYZI>
void ReadCloud(FILE* f, PointCloud* cloud) {
  const int n = Read<std::int32_t>(f, 1)[0];
  const auto p = Read<float>(f, 4 * static_cast<std::size_t>(n));
  cloud->clear();
  for (int i = 0; i < n; ++i) {
    PointType q;
    q.x = p[4 * i]; q.y = p[4 * i + 1]; q.z = p[4 * i + 2]; q.intensity = p[4 * i + 3];
    cloud->push_back(q);
  }
}

// n x {x y z intensity}, n x u16 ring -> pcl::PointCloud<PointXYZIRT>
void ReadRingCloud(FILE* f, PointCloudOriginal* cloud) {
  const int n = Read<std::int32_t>(f, 1)[0];
  const auto p = Read<float>(f, 4 * static_cast<std::size_t>(n));
  const auto r = Read<std::uint16_t>(f, static_cast<std::size_t>(n));
  cloud->clear();
  for (int i = 0; i < n; ++i) {
    PointTypeOriginal q;
    q.x = p[4 * i]; q.y = p[4 * i + 1]; q.z = p[4 * i + 2]; q.intensity = p[4 * i + 3]; q.ring = r[i]; q.time = p[4 * i + 3];
    cloud->push_back(q);
  }
}

Rigid3d ReadPose(FILE* f) {
  const auto v = Read<double>(f, 7);
  Eigen::Matrix<double, 7, 1> e;
  for (int k = 0; k < 7; ++k) e[k] = v[k];
  return Rigid3d(e);                                    // rigid_transform.h:47-49: [t, qx qy qz qw], no normalisation
}

void WritePose(FILE* o, Rigid3d pose) {
  const Eigen::Matrix<double, 7, 1> v = pose.ToVector7();
  double a[7];
  for (int k = 0; k < 7; ++k) a[k] = v[k];
  std::fwrite(a, 8, 7, o);
}

}  // namespace

int main(int argc, char** argv) {
  google::InitGoogleLogging(argv[0]);
  if (argc < 3) { std::fprintf(stderr, "usage: ref_golden <in.bin> <out.bin>\n"); return 1; }
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) { std::perror(argv[1]); return 1; }
  const auto head = Read<std::uint32_t>(f, 2);
  if (head[0] != 0x4d53464cu || head[1] != 1u) { std::fprintf(stderr, "not a ref_golden input (magic / version)\n"); return 1; }

  TimestampedPointCloud<PointType> cloud_map;
  ReadCloud(f, cloud_map.cloud_corner_less_sharp.get());
  ReadCloud(f, cloud_map.cloud_surf_less_flat.get());

  FILE* o = std::fopen(argv[2], "wb");
  if (!o) { std::perror(argv[2]); return 1; }
  const std::uint32_t out_head[2] = {0x4d534652u, 1u};
  std::fwrite(out_head, 4, 2, o);

  MappingScanMatcher mapper;
  for (int k = 0; k < 2; ++k) {
    TimestampedPointCloud<PointType> scan_curr;
    ReadCloud(f, scan_curr.cloud_corner_less_sharp.get());
    ReadCloud(f, scan_curr.cloud_surf_less_flat.get());
    Rigid3d pose = ReadPose(f);
    Vector3d velocity = Vector3d::Zero();
    RobotState prev_state;                              // only logged and copied when !is_initialized (mapping_scan_matcher.cc:27-33)
    prev_state.p = Vector3d::Zero(); prev_state.v = Vector3d::Zero(); prev_state.q = Eigen::Quaterniond::Identity();
    prev_state.bg = Vector3d::Zero(); prev_state.ba = Vector3d::Zero();
    mapper.MatchScan2Map(cloud_map, scan_curr, /*is_initialized=*/false, nullptr, Vector3d(0, 0, 9.81), prev_state, &pose, &velocity);
    WritePose(o, pose);
  }

  TimestampedPointCloud<PointTypeOriginal> scan_last, scan_curr;
  ReadRingCloud(f, scan_last.cloud_corner_less_sharp.get());
  ReadRingCloud(f, scan_last.cloud_surf_less_flat.get());
  ReadRingCloud(f, scan_curr.cloud_corner_sharp.get());
  ReadRingCloud(f, scan_curr.cloud_surf_flat.get());
  Rigid3d rel = ReadPose(f);
  OdometryScanMatcher odometry;
  const bool ok = odometry.MatchScan2Scan(scan_last, scan_curr, &rel);
  WritePose(o, rel);
  const std::int32_t ok_i = ok ? 1 : 0;
  std::fwrite(&ok_i, 4, 1, o);
  std::fclose(o);
  std::fclose(f);
  return 0;
}

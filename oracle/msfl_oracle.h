/*
 * msfl_oracle.h — CPU ORACLE for the MSF_LOAM scan-matching hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a dependency-free C restatement of the reference's
 * algorithm (kekeliu-whu/MSF_LOAM @ /root/reference) used as the parity checker.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
 * (msf_loam_amd/, libmsfl_hip.so) never links, imports or calls anything in oracle/.
 *
 * PARITY UNPINNED: the reference holds no test, golden vector or fixture on this path
 *  and its hot path cannot be compiled here (needs Ceres, PCL/FLANN, Eigen,
 * glog — none present, no network).  The third-party arithmetic is restated from the published
 * algorithms:
 *   Ceres Solver 1.14 (Ubuntu 20.04 libceres-dev; reference CMakeLists.txt:23, Dockerfile:5-9)
 *     TrustRegionMinimizer + LevenbergMarquardtStrategy + HuberLoss + Corrector
 *   PCL 1.10 KdTreeFLANN -> FLANN 1.9.1 exact kNN, L2_Simple<float> over (x,y,z)
 *   Eigen 3.3 SelfAdjointEigenSolver<Matrix3d>, colPivHouseholderQr (5x3), Quaternion
 * and each piece is guarded by an independent numpy/scipy known-answer test (tests/).
 */
#ifndef MSFL_ORACLE_H_
#define MSFL_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, z, t; } orc_point;   /* pcl::PointXYZI; t == intensity == rel. time */

/* Sensitivity tests only: flip the three choices the reference's toolchain leaves unspecified (0 = documented choice). */
void orc_set_unspecified(int sort_ties_reverse, int knn_ties_reverse, int atan2_float);

/* ---- math restatements -------------------------------------------------------------------- */

/* Eigen::Quaterniond * Vector3d (QuaternionBase::_transformVector). q = [qx qy qz qw]. */
void orc_quat_rotate(const double q[4], const double v[3], double out[3]);
/* Eigen::Quaterniond::toRotationMatrix(), row-major 3x3. */
void orc_quat_to_matrix(const double q[4], double R[9]);
/* TransformPoint (rigid_transform.h:132-138): f32 -> f64 -> rotate+translate -> f32. */
void orc_transform_point(const double pose[7], const float in[3], float out[3]);
/* PoseLocalParameterization::Plus (pose_local_parameterization.cc:6-21) with Utility::deltaQ
   (utility.h:7-31). */
void orc_pose_plus(const double x[7], const double delta[6], double out[7]);
/* Rigid3d operator* (rigid_transform.h:105-111) incl. quaternion renormalisation. */
void orc_pose_compose(const double a[7], const double b[7], double out[7]);

/* Eigen::SelfAdjointEigenSolver<Matrix3d> restated as cyclic Jacobi; eigenvalues ascending,
   eigenvectors as columns of V (row-major 3x3). */
void orc_sym_eigen3(const double A[9], double evals[3], double V[9]);
/* x = argmin |A x - b|, A 5x3 row-major, column-pivoted Householder QR
   (Eigen colPivHouseholderQr().solve()); returns the detected rank. */
int orc_lstsq_5x3(const double A[15], const double b[5], double x[3]);

/* mapping_scan_matcher.cc:130-151: 5 neighbours -> line.  Returns 1 if lambda2 > 3 lambda1;
   C = point_a (= center + 0.1 dir), N = (a-b).normalized(). */
int orc_edge_fit(const float nbr[5][3], double ratio, double C[3], double N[3]);
/* mapping_scan_matcher.cc:199-222: 5 neighbours -> plane.  Returns 1 if planeValid. */
int orc_plane_fit(const float nbr[5][3], double tol, double C[3], double N[3]);

/* lidar_factor.cc:7-24 / :26-44.  J is row-major (3x7 / 1x7) as Ceres hands it out. */
void orc_edge_factor(const double pose[7], const double p[3], const double C[3], const double N[3],
                     double r[3], double J[21]);
void orc_plane_factor(const double pose[7], const double p[3], const double C[3], const double N[3],
                      double r[1], double J[7]);

/* ---- exact kNN (FLANN semantics) ------------------------------------------------------------ */

/* k nearest in f32 L2_Simple distance ((dx*dx + dy*dy) + dz*dz), ascending, ties by index.
   Brute force: the definition. */
void orc_knn_brute(const orc_point* cloud, int n, const float q[3], int k, int* idx, float* d2);

typedef struct orc_kdtree orc_kdtree;
orc_kdtree* orc_kdtree_build(const orc_point* cloud, int n);
void orc_kdtree_free(orc_kdtree* t);
/* Same result as orc_knn_brute, O(log n). */
void orc_kdtree_knn(const orc_kdtree* t, const float q[3], int k, int* idx, float* d2);

/* ---- correspondence record + solver ---------------------------------------------------------- */

enum { ORC_KIND_NONE = 0, ORC_KIND_EDGE = 1, ORC_KIND_PLANE = 2 };

typedef struct orc_corr {
  double p[3];   /* curr_point, scan frame (untransformed, mapping_scan_matcher.cc:146,221) */
  double C[3];   /* last_line_C / last_plane_C */
  double N[3];   /* last_line_N / last_plane_N */
  int kind;
  int pad_;
} orc_corr;

typedef struct orc_solver_options {
  int    max_num_iterations;      /* 6 */
  double huber_delta;             /* 0.1 */
  double initial_trust_region_radius, max_trust_region_radius, min_trust_region_radius;
  double min_relative_decrease, min_lm_diagonal, max_lm_diagonal;
  double function_tolerance, gradient_tolerance, parameter_tolerance;
  int    max_consecutive_invalid_steps;
} orc_solver_options;
void orc_default_solver_options(orc_solver_options* o);

#define ORC_MAX_TRACE 16
typedef struct orc_solve_summary {
  int    iterations;              /* trust-region iterations executed (excl. iteration 0) */
  int    successful_steps;
  int    termination;             /* 0 max-iter, 1 gradient tol, 2 parameter tol, 3 function tol,
                                     4 min radius, 5 invalid steps, 6 no residuals */
  double initial_cost, final_cost;
  /* per executed iteration */
  double trace_cost[ORC_MAX_TRACE];        /* candidate cost */
  double trace_radius[ORC_MAX_TRACE];      /* radius used for the step */
  double trace_rel_decrease[ORC_MAX_TRACE];
  double trace_step_norm[ORC_MAX_TRACE];
  int    trace_accepted[ORC_MAX_TRACE];
} orc_solve_summary;

/* Evaluate cost (= 1/2 sum rho(|r_i|^2)), and optionally the robustified Gauss-Newton system in
   the 6-D tangent space: H = J^T J (row-major 6x6), g = J^T r, with Ceres' Corrector applied. */
double orc_evaluate(const orc_corr* corr, int n, const double pose[7], double huber_delta,
                    double* H36, double* g6);

/* ceres::Solve with default options + max_num_iterations (S1). pose in/out. */
void orc_ceres_solve(const orc_corr* corr, int n, double pose[7], const orc_solver_options* opt,
                     orc_solve_summary* summary);

/* ---- stage C -------------------------------------------------------------------------------- */

typedef struct orc_match_info {
  int n_edge[2], n_plane[2];
  int lm_iterations[2], lm_successful[2];
  double initial_cost[2], final_cost[2];
} orc_match_info;

/* Data association of one outer iteration (mapping_scan_matcher.cc:109-246): writes
   n_corner + n_surf records (kind NONE where rejected).  use_kdtree selects the search
   structure (same results). */
void orc_associate_scan2map(const orc_point* map_corner, int mc, const orc_point* map_surf, int ms,
                            const orc_point* corner, int nc, const orc_point* surf, int ns,
                            const double pose[7], int use_kdtree, orc_corr* out);

/* MappingScanMatcher::MatchScan2Map, LiDAR-only branch. pose in/out. Returns 0 on success,
   2 if a map cloud has < 5 points. */
int orc_match_scan2map(const orc_point* map_corner, int mc, const orc_point* map_surf, int ms,
                       const orc_point* corner, int nc, const orc_point* surf, int ns,
                       double pose[7], int use_kdtree, orc_match_info* info);

/* The same over B scans sharing one map; `threads` > 1 uses OpenMP across scans.  Builds the
   kd-trees once per scan like the reference when rebuild_tree_per_scan != 0 (the reference's
   cost structure, mapping_scan_matcher.cc:66-73), else once per batch. */
void orc_match_scan2map_batch(const orc_point* map_corner, int mc, const orc_point* map_surf, int ms,
                              int n_scans, const orc_point* corner, const int* corner_off,
                              const orc_point* surf, const int* surf_off,
                              double* poses, int* status, int threads, int rebuild_tree_per_scan);

/* Deskew variant (is_initialized branch; lidar_factor.cc:46-100; velocity block constant). */
void orc_match_scan2map_batch_trees(const orc_point* map_corner, int mc, const orc_kdtree* tc, const orc_point* map_surf, int ms,
                                    const orc_kdtree* ts, int n_scans, const orc_point* corner, const int* corner_off,
                                    const orc_point* surf, const int* surf_off, double* poses, int* status, int threads);
void orc_match_scan2map_batch_timed(const orc_point* map_corner, int mc, const orc_point* map_surf, int ms,
                                    int n_scans, const orc_point* corner, const int* corner_off,
                                    const orc_point* surf, const int* surf_off,
                                    double* poses, int* status, double stage_seconds[3]);
int orc_match_scan2map_deskew(const orc_point* map_corner, int mc, const orc_point* map_surf, int ms,
                              const orc_point* corner, int nc, const orc_point* surf, int ns,
                              const double* corner_dq, const double* corner_dp,
                              const double* surf_dq, const double* surf_dp,
                              const double velocity[3], const double gravity[3],
                              double pose[7], orc_match_info* info);

/* ---- stage B -------------------------------------------------------------------------------- */

/* OdometryScanMatcher::MatchScan2Scan.  Returns 0 ok, 1 when the reference returns false. */
int orc_match_scan2scan(const orc_point* last_ls, const uint16_t* last_ls_ring, int n_last_ls,
                        const orc_point* last_lf, const uint16_t* last_lf_ring, int n_last_lf,
                        const orc_point* sharp, int n_sharp,
                        const orc_point* flat, int n_flat,
                        double pose[7], orc_match_info* info);

/* Association only (one outer iteration), for kernel-level parity tests. */
void orc_associate_scan2scan(const orc_point* last_ls, const uint16_t* last_ls_ring, int n_last_ls,
                             const orc_point* last_lf, const uint16_t* last_lf_ring, int n_last_lf,
                             const orc_point* sharp, int n_sharp,
                             const orc_point* flat, int n_flat,
                             const double pose[7], orc_corr* out);

/* ---- stage A -------------------------------------------------------------------------------- */

/* RealHandleLaserCloudMessage (msf_loam_node.cc:160-378).  Output arrays have capacity n.
   Sort ties (std::sort is unstable, §3.4) are broken by ascending point index.
   Returns 0 ok, 5 bad ring, 3 empty cloud. */
int orc_extract_features(const orc_point* pts, const uint16_t* ring, int n,
                         double min_range, const double* extrinsic_pose7,
                         orc_point* full_pts, uint16_t* full_ring, float* curvature, uint8_t* label,
                         int* sharp_idx, int* less_sharp_idx, int* flat_idx, int* less_flat_idx,
                         int counts[5] /* n_full, n_sharp, n_less_sharp, n_flat, n_less_flat */);

/* ---- caller-side helper (N2) ---------------------------------------------------------------- */

/* pcl::VoxelGrid<PointXYZI>::filter with setLeafSize(leaf,leaf,leaf), downsample_all_data. */
int orc_voxel_grid(const orc_point* pts, int n, float leaf, orc_point* out);

#ifdef __cplusplus
}
#endif
#endif

/* ---- N1: the local map store, src/slam/map/hybrid_grid.cc:462-534 ------------- */
#ifdef __cplusplus
extern "C" {
#endif
typedef struct orc_grid orc_grid;
/* HybridGrid(resolution) + the pcl::VoxelGrid leaf its owner passes to InsertScan
   (laser_mapping.cc:44-45,60-68: resolution 3.0, leaf 0.2 corner / 0.4 surf). */
orc_grid* orc_grid_create(float resolution, float leaf);
void orc_grid_free(orc_grid* g);
/* HybridGridImpl::InsertScan (:503-521): append each point to the cell round(p / resolution), then
   VoxelGrid-filter every touched cell in place.  Returns 0, or 7 if a cell index leaves +-8192. */
int orc_grid_insert_scan(orc_grid* g, const orc_point* pts, int n);
/* HybridGridImpl::GetSurroundedCloud (:470-501): union of the cells hit by pose_f32 * p + (i,j,k) m,
   (i,j,k) in {-1,0,1}^3, for scan points with |p| <= 60.  Cells are emitted in ascending
   (iz, iy, ix) order (the reference iterates a boost::unordered_set of pointers: unspecified). */
int orc_grid_get_surrounded(const orc_grid* g, const orc_point* scan, int n, const double pose[7],
                            orc_point* out, int capacity);
int orc_grid_size(const orc_grid* g, int* n_cells);
/* all points, cells ascending, in-cell order as stored */
int orc_grid_dump(const orc_grid* g, orc_point* out, int capacity);
/* ---- N3: IMU deskew inputs ---------------------------------------------------------------- */
/* GetDeltaQP (src/slam/imu_fusion/scan_undistortion.cc:22-42): upper_bound over sum_dt, Eigen
   3.3 Quaternion::slerp (no normalisation, linear when |dot| >= 1 - eps), lerp of delta_p.
   Returns 0, or 1 where the reference CHECK-aborts (dt outside [front, back]).  dt == back()
   makes the reference read one past the end; restated as the last sample (s = 1). */
int orc_delta_qp(const double* sum_dt, const double* delta_q, const double* delta_p, int n_samples,
                 double dt, double q_out[4], double p_out[3]);
/* GetDeltaQP per point of a cloud (the matcher's per-feature calls, mapping_scan_matcher.cc:113-117,183-187); returns the refused count */
int orc_delta_qp_cloud(const double* sum_dt, const double* delta_q, const double* delta_p, int n_samples,
                       const orc_point* pts, int n, double* dq, double* dp);
/* laser_mapping.cc:197-211, in place; returns the number of points whose time is out of range */
int orc_deskew_cloud(const double* sum_dt, const double* delta_q, const double* delta_p, int n_samples,
                     orc_point* pts, int n, const double rot_odom[4], const double velocity[3],
                     const double gravity[3]);
/* UndistortScanInternal, scan_undistortion.cc:5-19 (rotation only, in f32) */
int orc_undistort_cloud(const double* sum_dt, const double* delta_q, const double* delta_p, int n_samples,
                        orc_point* pts, int n);
/* TransformPointCloud, laser_mapping.cc:24-31 */
void orc_transform_cloud(const orc_point* in, int n, const double pose[7], orc_point* out);

#ifdef __cplusplus
}
#endif

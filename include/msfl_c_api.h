/*
 * msfl_c_api.h — C ABI of the MI355X-native LOAM scan-matching engine (libmsfl_hip.so).
 *
 * This is the drop-in boundary for the ONE hot path of kekeliu-whu/MSF_LOAM:
 *   stage A  per-scan feature extraction      reference src/msf_loam_node.cc:160-378
 *   stage B  scan-to-scan registration        reference src/slam/local/scan_matching/odometry_scan_matcher.cc:43-285
 *   stage C  scan-to-local-map registration   reference src/slam/local/scan_matching/mapping_scan_matcher.cc:19-278
 *
 * Plain C, plain pointers and sizes, no torch / PCL / Eigen / Ceres types. Every entry point
 * cites the reference interface it replaces.  A maintainer-side binding (C++ adapter with the
 * reference's own method signatures) is shown in INTEGRATION.md and shipped as
 * include/msfl_scan_matcher.hpp.
 *
 * Conventions
 *   - Points are 16-byte records {x, y, z, t}.  `t` is what the reference keeps in
 *     pcl::PointXYZI::intensity, i.e. the per-point relative time in seconds
 *     (msf_loam_node.cc:152-153 writes time into intensity).
 *   - A pose is 7 doubles [tx ty tz qx qy qz qw], the layout of Rigid3d::ToVector7()
 *     (src/common/rigid_transform.h:59-64).  Poses are IN/OUT: initial guess in, result out,
 *     exactly like the reference's `Rigid3d *pose_estimate` arguments.
 *   - Every pointer argument is tagged by a `mem` flag: MSFL_MEM_HOST (the library stages the
 *     copy) or MSFL_MEM_DEVICE (already resident in HBM on the handle's device; nothing is copied).
 *   - All functions return an msfl_status.  No exceptions, no aborts: the reference's glog CHECK
 *     failures and out-of-bounds reads (SURVEY.md §8b "edge cases") become status codes.
 *   - A handle owns one HIP stream and all device scratch.  Handles are independent, so the
 *     odometry thread and the mapping thread of the reference (laser_odometry.h:29,
 *     laser_mapping.h:65) each own one and may call concurrently.  A single handle is not
 *     re-entrant (neither is a reference matcher instance).
 */
#ifndef MSFL_C_API_H_
#define MSFL_C_API_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSFL_API_VERSION 1

typedef struct msfl_handle_s msfl_handle;

typedef enum msfl_status {
  MSFL_OK = 0,
  /* odometry_scan_matcher.cc:262-267 returns false when corner+plane correspondences < 10.
     Per-scan status in batched calls; the pose keeps the value of the last completed solve. */
  MSFL_TOO_FEW_CORRESPONDENCES = 1,
  /* mapping_scan_matcher.cc:128,198 index pointSearchSqDis[4] unguarded; the reference relies on
     the caller's gate (laser_mapping.cc:284-285).  Here: map cloud with < 5 points. */
  MSFL_MAP_TOO_SMALL = 2,
  MSFL_BAD_ARG = 3,
  MSFL_HIP_ERROR = 4,
  /* msf_loam_node.cc:136 CHECK_LT(point.ring, 128) / :186 CHECK_GT(valid_scan_num, 0) */
  MSFL_BAD_RING = 5,
  MSFL_NO_MAP = 6,
  MSFL_CAPACITY = 7
} msfl_status;

typedef enum msfl_mem { MSFL_MEM_HOST = 0, MSFL_MEM_DEVICE = 1 } msfl_mem;

/* 16-byte point: pcl::PointXYZI as the matchers see it (x,y,z + intensity==relative time). */
typedef struct msfl_point {
  float x, y, z, t;
} msfl_point;

/*
 * All hot-path thresholds of the reference are compile-time constants (SURVEY.md §8a
 * "Constants").  They are exposed as a POD with the reference values as defaults
 * (msfl_default_params).  Parity claims hold for the defaults.
 */
typedef struct msfl_params {
  /* --- stage A, msf_loam_node.cc --- */
  double scan_period;            /* kScanPeriod = 0.1                     :80      */
  double min_range;              /* g_min_range, launch default 0.3       :434     */
  float  curvature_threshold;    /* 0.1                                   :275,312 */
  float  neighbor_gap_sq;        /* 0.05                                  :293,300 */
  int    sectors_per_ring;       /* 6                                     :255     */
  int    max_sharp_per_sector;   /* 2                                     :277     */
  int    max_less_sharp_per_sector; /* 20                                 :281     */
  int    max_flat_per_sector;    /* 4                                     :317     */
  /* --- stage B, odometry_scan_matcher.cc:15-18 --- */
  double odom_distance_sq_threshold; /* kDistanceSqThreshold = 25 */
  double odom_nearby_scan;           /* kNearByScan = 2.5         */
  int    odom_min_correspondences;   /* 10, :262                  */
  /* --- stage C, mapping_scan_matcher.cc --- */
  int    map_knn;                /* 5 (fixed by the kernels; other values -> MSFL_BAD_ARG) :125 */
  float  map_knn_max_sq_dist;    /* 1.0   :128,198 */
  double line_eigen_ratio;       /* 3.0   :147     */
  double plane_tolerance;        /* 0.2   :216     */
  /* --- shared solver settings (both matchers) --- */
  int    outer_iterations;       /* kOptimalNum = 2, mapping_scan_matcher.cc:15 */
  int    max_lm_iterations;      /* options.max_num_iterations = 6, :252 */
  double huber_delta;            /* ceres::HuberLoss(0.1), :77 */
  /* Ceres Solver::Options defaults the reference leaves untouched (third party, not in tree). */
  double initial_trust_region_radius; /* 1e4   */
  double max_trust_region_radius;     /* 1e16  */
  double min_trust_region_radius;     /* 1e-32 */
  double min_relative_decrease;       /* 1e-3  */
  double min_lm_diagonal;             /* 1e-6  */
  double max_lm_diagonal;             /* 1e32  */
  double function_tolerance;          /* 1e-6  */
  double gradient_tolerance;          /* 1e-10 */
  double parameter_tolerance;         /* 1e-8  */
  int    max_consecutive_invalid_steps; /* 5 */
} msfl_params;

/* Per-registration diagnostics (what the reference only logs via glog / Ceres BriefReport). */
typedef struct msfl_match_info {
  int    status;             /* msfl_status of this scan */
  int    n_edge[2];          /* accepted edge correspondences, outer iteration 0 / 1  (corner_num :173) */
  int    n_plane[2];         /* accepted plane correspondences (surf_num :243) */
  int    lm_iterations[2];   /* trust-region iterations executed (<= max_lm_iterations) */
  int    lm_successful[2];   /* accepted steps */
  double initial_cost[2];    /* Ceres cost (1/2 sum rho) at entry of each solve */
  double final_cost[2];      /* cost at the returned pose */
} msfl_match_info;

/* Accumulated GPU time per kernel class, measured with HIP events on the handle's stream
   (enabled by msfl_set_timing).  Used by bench.py for the live roofline figure. */
typedef struct msfl_timing {
  int    launches_assoc;   double ms_assoc;     /* transform + exact 5-NN kernel      */
  int    launches_solve;   double ms_solve;     /* persistent LM + Huber solve kernel */
  int    launches_index;   double ms_index;     /* map grid index build (all kernels) */
  int    launches_extract; double ms_extract;   /* feature extraction (all kernels)   */
  int    launches_odom;    double ms_odom;      /* scan-to-scan association kernel    */
  int    launches_fit;     double ms_fit;       /* line / plane fit kernel            */
  unsigned long long knn_candidates;            /* map points whose distance the 5-NN kernel evaluated (mode 3 only): the launches of the
                                                   first outer iteration */
  unsigned long long knn_candidates_seeded;     /* the same for the launches of the second outer iteration (which start from the bound the first
                                                   iteration's neighbours give when MSFL_KNN_SEED=1, from the acceptance gate otherwise) */
  int    launches_assoc_seeded; double ms_assoc_seeded;   /* the second iteration's 5-NN launches alone (also part of launches_assoc / ms_assoc) */
} msfl_timing;

/* ------------------------------------------------------------------------------------------ */
/* lifecycle                                                                                  */
/* ------------------------------------------------------------------------------------------ */

/* Fill `p` with the reference constants listed above. */
void msfl_default_params(msfl_params* p);

int msfl_api_version(void);

/* Replaces construction of the matcher objects (laser_odometry.cc:56, laser_mapping.cc:42).
   `params` may be NULL (defaults).  `device` is the HIP device ordinal. */
msfl_status msfl_create(const msfl_params* params, int device, msfl_handle** out);
void msfl_destroy(msfl_handle* h);

/* Run all work of this handle on a caller-provided hipStream_t (e.g. torch's current stream).
   The value is used as is: NULL means HIP's null (legacy default) stream, which is what
   torch.cuda.current_stream().cuda_stream is unless the caller switched streams.
   msfl_reset_stream goes back to the handle's own (non-blocking) stream. */
msfl_status msfl_set_stream(msfl_handle* h, void* hip_stream);
msfl_status msfl_reset_stream(msfl_handle* h);
/* Block until everything queued by this handle has finished. */
msfl_status msfl_synchronize(msfl_handle* h);

const char* msfl_status_string(int status);
/* Human-readable detail of the last non-OK status on this handle ("" if none). */
const char* msfl_last_error(const msfl_handle* h);

/* enabled: 0 off; 1 every kernel class; 2 only the association (5-NN) kernel — two events per launch
   instead of two per class, for timing the dominant kernel inside a throughput measurement; 3 like 1, and the
   5-NN kernel additionally counts the candidates it evaluates (a counting instantiation: slower, for the
   "distance evaluations per second" figure only). */
msfl_status msfl_set_timing(msfl_handle* h, int enabled);
msfl_status msfl_get_timing(msfl_handle* h, msfl_timing* out, int reset);

/* ------------------------------------------------------------------------------------------ */
/* stage C — scan-to-local-map registration                                                   */
/*   replaces MappingScanMatcher::MatchScan2Map (mapping_scan_matcher.h:14-21, .cc:19-278),    */
/*   LiDAR-only branch (!is_initialized, .cc:96,123,168-171,193,238-241,271).                  */
/* ------------------------------------------------------------------------------------------ */

/* Replaces the two pcl::KdTreeFLANN::setInputCloud calls (mapping_scan_matcher.cc:66-73):
   uploads (if host) and spatially indexes cloud_map.cloud_corner_less_sharp /
   cloud_map.cloud_surf_less_flat.  The index is an exact-kNN uniform grid.
   The map stays resident until the next msfl_set_map on this handle. */
msfl_status msfl_set_map(msfl_handle* h,
                         const msfl_point* corner, int n_corner,
                         const msfl_point* surf, int n_surf,
                         msfl_mem mem);

/* One registration against the resident map.
     corner/surf : scan_curr.cloud_corner_less_sharp / cloud_surf_less_flat (already voxel
                   down-sampled by the caller, laser_mapping.cc:264-270), scan frame.
     pose_io     : pose_estimate_map_scan2world, Vector7, in = guess, out = result.
     info        : optional diagnostics.
   Returns MSFL_OK (the reference always returns true, .cc:277) or an error status; with zero
   accepted correspondences the pose is returned unchanged (Ceres solves an empty problem). */
msfl_status msfl_match_scan2map(msfl_handle* h,
                                const msfl_point* corner, int n_corner,
                                const msfl_point* surf, int n_surf,
                                double pose_io[7], msfl_match_info* info,
                                msfl_mem mem);

/* B independent registrations against the same resident map in one call (the shape the
   north-star batches: "many scans shard embarrassingly").
     corner, corner_off : concatenated corner features and B+1 prefix offsets (scan b owns
                          [corner_off[b], corner_off[b+1]) ); same for surf.
     poses_io           : B x 7 doubles, in/out.
     status             : B ints out (msfl_status per scan), may be NULL.
     info               : B msfl_match_info out, may be NULL (host memory only).
   With mem == MSFL_MEM_DEVICE all five arrays and poses_io/status are device pointers and the
   call is asynchronous on the handle's stream. */
msfl_status msfl_match_scan2map_batch(msfl_handle* h, int n_scans,
                                      const msfl_point* corner, const int* corner_off,
                                      const msfl_point* surf, const int* surf_off,
                                      double* poses_io, int* status, msfl_match_info* info,
                                      msfl_mem mem);

/* P independent (map, scan) PAIRS in one call: scan p is registered against map p only ("many map-submap pairs").  The
   reference analogue is one MatchScan2Map call per pair, each rebuilding both kd-trees (mapping_scan_matcher.cc:66-73;
   laser_mapping.cc:304-311).  Here the P maps are indexed together (one exact-kNN grid per pair inside one set of arrays,
   four launches + one scan per cloud kind whatever P is) and the P registrations run as one batch.
     map_corner, map_corner_off : the P corner map clouds concatenated + P+1 prefix offsets (HOST array, like every batch
                                  call); same for map_surf.
     corner / surf + offsets    : the P scans' (down-sampled) feature clouds, as in msfl_match_scan2map_batch.
     poses_io, status, info     : as there.  status[p] = MSFL_MAP_TOO_SMALL for a pair whose corner or surf map has fewer
                                  than 5 points (the pose guess passes through); the call itself still returns MSFL_OK.
   Results equal P calls of msfl_set_map + msfl_match_scan2map bit for bit.  With MSFL_MEM_DEVICE the clouds, poses_io and
   status are device pointers and the call is asynchronous.  The handle's resident single map (msfl_set_map) is replaced:
   call msfl_set_map again before the next msfl_match_scan2map*. */
msfl_status msfl_match_pairs_batch(msfl_handle* h, int n_pairs,
                                   const msfl_point* map_corner, const int* map_corner_off,
                                   const msfl_point* map_surf, const int* map_surf_off,
                                   const msfl_point* corner, const int* corner_off,
                                   const msfl_point* surf, const int* surf_off,
                                   double* poses_io, int* status, msfl_match_info* info, msfl_mem mem);

/* Kernel-level entry points (the two halves of one outer iteration), exposed so that parity tests
   can pin the data association and the solver separately:
     msfl_associate_scan2map : mapping_scan_matcher.cc:109-246 at a fixed pose.  records_out gets
                               (n_corner + n_surf) x 6 doubles {C[3], N[3]} in feature order (corner
                               first); a rejected feature is all zeros.  For an edge C = point_a and
                               N = (a-b).normalized() (:150-158); for a plane C = center, N = norm (:238).
     msfl_solve_records      : one ceres::Solve (:250-259) on caller-provided records.
   Host pointers only. */
msfl_status msfl_associate_scan2map(msfl_handle* h,
                                    const msfl_point* corner, int n_corner,
                                    const msfl_point* surf, int n_surf,
                                    const double pose[7], double* records_out);
msfl_status msfl_solve_records(msfl_handle* h,
                               const msfl_point* corner, int n_corner,
                               const msfl_point* surf, int n_surf,
                               const double* records, double pose_io[7], msfl_match_info* info);

/* Optional IMU-deskew inputs for the is_initialized branch (mapping_scan_matcher.cc:84-94,
   119-121,154-166,189-191,224-236).  The per-point (delta_q, delta_p) = GetDeltaQP(preintegration,
   dt) (scan_undistortion.cc:22-42) are computed by the caller (IMU pre-integration is outside the
   hot path); the velocity block is held constant by the reference (.cc:94), so only the pose is
   optimised.  Layout: delta_q as [qx qy qz qw], one per corner then one per surf feature. */
typedef struct msfl_deskew {
  const double* corner_dq;  /* n_corner x 4 */
  const double* corner_dp;  /* n_corner x 3 */
  const double* surf_dq;    /* n_surf x 4 */
  const double* surf_dp;    /* n_surf x 3 */
  double velocity[3];       /* bias_j.head<3>() (Vi, .cc:107) */
  double gravity[3];        /* gravity_vector */
} msfl_deskew;

/* As msfl_match_scan2map with the Deskew factors (lidar_factor.cc:46-100).  Host pointers only. */
msfl_status msfl_match_scan2map_deskew(msfl_handle* h,
                                       const msfl_point* corner, int n_corner,
                                       const msfl_point* surf, int n_surf,
                                       const msfl_deskew* deskew,
                                       double pose_io[7], msfl_match_info* info);

/* The deskew branch for a batch.  The four per-feature arrays are indexed exactly like the feature arrays
   (corner feature corner_off[b] + k of scan b uses corner_dq[4 * (corner_off[b] + k)] ...), velocity holds
   one Vi per scan; all follow `mem`. */
typedef struct msfl_deskew_batch {
  const double* corner_dq;  /* (corner_off[n_scans]) x 4 */
  const double* corner_dp;  /* ... x 3 */
  const double* surf_dq;
  const double* surf_dp;
  const double* velocity;   /* n_scans x 3 */
  double gravity[3];
} msfl_deskew_batch;

msfl_status msfl_match_scan2map_deskew_batch(msfl_handle* h, int n_scans,
                                             const msfl_point* corner, const int* corner_off,
                                             const msfl_point* surf, const int* surf_off,
                                             const msfl_deskew_batch* deskew,
                                             double* poses_io, int* status, msfl_match_info* info, msfl_mem mem);

/* ------------------------------------------------------------------------------------------ */
/* stage B — scan-to-scan registration                                                        */
/*   replaces OdometryScanMatcher::MatchScan2Scan (odometry_scan_matcher.h:10-12, .cc:43-285)  */
/* ------------------------------------------------------------------------------------------ */

/* A ring-ordered feature cloud of PointXYZIRT (common.h:44-62): points + ring ids. */
typedef struct msfl_ring_cloud {
  const msfl_point* pts;
  const uint16_t*   ring;
  int               n;
} msfl_ring_cloud;

/*   last_less_sharp / last_less_flat : scan_last.cloud_corner_less_sharp / cloud_surf_less_flat
     curr_sharp / curr_flat           : scan_curr.cloud_corner_sharp / cloud_surf_flat
     pose_io                          : pose_estimate_curr2last, in/out.
   Returns MSFL_TOO_FEW_CORRESPONDENCES where the reference returns false (.cc:262-267); the
   pose then holds the result of the outer iterations completed so far. */
msfl_status msfl_match_scan2scan(msfl_handle* h,
                                 const msfl_ring_cloud* last_less_sharp,
                                 const msfl_ring_cloud* last_less_flat,
                                 const msfl_ring_cloud* curr_sharp,
                                 const msfl_ring_cloud* curr_flat,
                                 double pose_io[7], msfl_match_info* info,
                                 msfl_mem mem);

/* B independent (last, curr) pairs.  Each of the four feature sets is concatenated over the
   batch with B+1 prefix offsets. */
typedef struct msfl_ring_cloud_batch {
  const msfl_point* pts;
  const uint16_t*   ring;
  const int*        off;   /* n_scans + 1 */
} msfl_ring_cloud_batch;

msfl_status msfl_match_scan2scan_batch(msfl_handle* h, int n_scans,
                                       const msfl_ring_cloud_batch* last_less_sharp,
                                       const msfl_ring_cloud_batch* last_less_flat,
                                       const msfl_ring_cloud_batch* curr_sharp,
                                       const msfl_ring_cloud_batch* curr_flat,
                                       double* poses_io, int* status, msfl_match_info* info,
                                       msfl_mem mem);

/* ------------------------------------------------------------------------------------------ */
/* stage A — feature extraction                                                               */
/*   replaces RealHandleLaserCloudMessage (msf_loam_node.cc:160-378) between pcl::fromROSMsg   */
/*   (:166-167) and LaserOdometry::AddLaserScan (:373).                                        */
/* ------------------------------------------------------------------------------------------ */

/* Output of one extraction = the five clouds of TimestampedPointCloud<PointXYZIRT>
   (timestamped_pointcloud.h:11-42).  cloud_full_res is the ring-concatenated valid cloud
   (msf_loam_node.cc:188-195) with t = relative time; the four feature clouds are returned as
   INDEX LISTS into cloud_full_res (the reference push_back()s copies of those same points), in
   the reference's push order.  Caller allocates: full_pts/full_ring/curvature/label with capacity
   n_in, the four index arrays with capacity n_in each.  `extrinsic` (lidar->imu, :367-371) may be
   NULL = identity (the shipped config, config/lio-sam-config2.json:7-20). */
typedef struct msfl_features {
  msfl_point* full_pts;    /* out: n_full points */
  uint16_t*   full_ring;   /* out */
  float*      curvature;   /* out: cloud_curvatures (valid for i in [5, n_full-5)) */
  uint8_t*    label;       /* out: PointLabel 0 UNKNOWN 1 SHARP 2 LESS_SHARP 3 FLAT (:68-77) */
  int*        sharp_idx;       /* out */
  int*        less_sharp_idx;  /* out */
  int*        flat_idx;        /* out */
  int*        less_flat_idx;   /* out */
  int n_full, n_sharp, n_less_sharp, n_flat, n_less_flat;  /* out counts */
} msfl_features;

/*   pts/ring : the sensor cloud as pcl::fromROSMsg delivers it (driver order), n points.
   Returns MSFL_BAD_RING for ring >= 128 (CHECK at :136), MSFL_BAD_ARG for an empty valid
   cloud (CHECK at :186,200) and MSFL_CAPACITY for a ring of more than 8128 valid points (the
   per-ring pick state is held on chip; a 128-beam sensor at 2048 columns has 2048 per ring). */
msfl_status msfl_extract_features(msfl_handle* h,
                                  const msfl_point* pts, const uint16_t* ring, int n,
                                  const double* extrinsic_pose7,
                                  msfl_features* out, msfl_mem mem);

/* B scans in one call; scan b owns [off[b], off[b+1]) of pts/ring and writes its outputs at the
   same offsets of the out arrays (index lists are scan-local); counts go to the five
   n_* arrays of length B. */
typedef struct msfl_features_batch {
  msfl_point* full_pts;  uint16_t* full_ring;  float* curvature;  uint8_t* label;
  int* sharp_idx;  int* less_sharp_idx;  int* flat_idx;  int* less_flat_idx;
  int* n_full;  int* n_sharp;  int* n_less_sharp;  int* n_flat;  int* n_less_flat;  /* B each */
} msfl_features_batch;

msfl_status msfl_extract_features_batch(msfl_handle* h, int n_scans,
                                        const msfl_point* pts, const uint16_t* ring,
                                        const int* off,
                                        msfl_features_batch* out, int* status, msfl_mem mem);

/* ------------------------------------------------------------------------------------------ */
/* helpers on the caller side of the path: pcl::VoxelGrid down-sampling of  */
/* the scan features before MatchScan2Map (laser_mapping.cc:264-270) and TransformPointCloud.  */
/* ------------------------------------------------------------------------------------------ */

/* Centroid voxel filter with PCL VoxelGrid semantics (leaf cube, centroid of x,y,z,t per
   occupied voxel, output ordered by voxel index).  out has capacity n; *n_out receives the
   count.  Non-finite points are DROPPED, as pcl::VoxelGrid::applyFilter does for a cloud that is
   not dense (`if (!input_->is_dense) if (!pcl_isfinite(...)) continue;`): a stable device-side
   compaction followed by the same filter.  (For a cloud flagged dense PCL does not look and its
   result is undefined; the reference's clouds never hold such a point, msf_loam_node.cc:85-111.)
   The batch forms below REFUSE a cloud with a non-finite point (MSFL_BAD_ARG): their inputs are
   the extraction's own outputs. */
msfl_status msfl_voxel_downsample(msfl_handle* h,
                                  const msfl_point* pts, int n, float leaf,
                                  msfl_point* out, int* n_out, msfl_mem mem);

/* The same filter over n_clouds clouds in one pass (the device-resident batch pipeline between
   msfl_extract_features_batch and msfl_match_scan2map_batch).  Cloud b occupies the region
   [off[b], off[b+1]) of `pts`; with `idx` it is the points pts[off[b] + idx[off[b] + k]],
   k < count[b] — exactly how msfl_features_batch lays out its index lists and counts, so a feature
   list is filtered straight out of the extraction's output.  idx == NULL: the region itself;
   count == NULL: the whole region.  `off` is a host array (like every batch call), pts / idx /
   count / out follow `mem`.  The filtered clouds are written back to back into `out` (capacity
   off[n]-off[0] points) and out_off (HOST, n_clouds+1) receives their boundaries; the call
   synchronises once to deliver them.  A cloud holding a non-finite point makes the call return
   MSFL_BAD_ARG (the clouds before it are still delivered). */
msfl_status msfl_voxel_downsample_batch(msfl_handle* h, int n_clouds,
                                        const msfl_point* pts, const int* idx, const int* off,
                                        const int* count, float leaf,
                                        msfl_point* out, int* out_off, msfl_mem mem);

/* Two index lists over the same clouds in one call: the corner (0.2 m) and surf (0.4 m) filters that
   the mapping thread runs back to back on every scan (laser_mapping.cc:264-270).  Same results as two
   msfl_voxel_downsample_batch calls (list a, then list b; every argument as there, idx_* and count_*
   required); with device memory both filters are enqueued before the one synchronisation that
   delivers both boundary tables, so the small corner kernels do not wait for a host round trip. */
msfl_status msfl_voxel_downsample_batch_pair(msfl_handle* h, int n_clouds,
                                             const msfl_point* pts, const int* off,
                                             const int* idx_a, const int* count_a, float leaf_a,
                                             msfl_point* out_a, int* out_off_a,
                                             const int* idx_b, const int* count_b, float leaf_b,
                                             msfl_point* out_b, int* out_off_b, msfl_mem mem);

/* TransformPointCloud (laser_mapping.cc:24-31 -> TransformPoint, rigid_transform.h:131-137): every
   point goes f32 -> f64 -> q*p + t -> f32, t (relative time) is carried over.  pose7 is always a
   host array; in == out is allowed. */
msfl_status msfl_transform_cloud(msfl_handle* h,
                                 const msfl_point* in, int n, const double pose7[7],
                                 msfl_point* out, msfl_mem mem);

/* ------------------------------------------------------------------------------------------ */
/* next row N3: per-point IMU deskew.                                          */
/* ------------------------------------------------------------------------------------------ */

/* What GetDeltaQP reads of an IntegrationBase (integration_base.h:62-66): per IMU sample the
   cumulative time and the pre-integrated rotation / position.  Always host arrays (a scan spans
   ~40-50 samples); n >= 2, sum_dt non-decreasing. */
typedef struct msfl_preintegration {
  const double* sum_dt;   /* n */
  const double* delta_q;  /* n x 4, [qx qy qz qw] */
  const double* delta_p;  /* n x 3 */
  int n;
} msfl_preintegration;

/* GetDeltaQP (scan_undistortion.cc:22-42) for every point's relative time pts[i].t: upper_bound
   over sum_dt, Eigen slerp of delta_q (no normalisation), lerp of delta_p.  Writes the arrays
   msfl_deskew takes (dq n x 4 [x y z w], dp n x 3).  Returns MSFL_BAD_ARG where the reference
   CHECK-aborts (a time outside [sum_dt.front(), sum_dt.back()], :26-30).  A time equal to
   sum_dt.back() makes the reference index one past the end (:37-39); here it yields the last
   sample (s = 1). */
msfl_status msfl_delta_qp(msfl_handle* h, const msfl_preintegration* pre,
                          const msfl_point* pts, int n, double* dq, double* dp, msfl_mem mem);

/* The deskew pass of LaserMapping (laser_mapping.cc:197-211), in place on one cloud:
     p <- (dq(t) * p + R_odom^-1 * (velocity * t - 0.5 * gravity * t * t) + dp(t)).cast<float>()
   rot_odom_xyzw = pose_odom_scan2world_.rotation() as [qx qy qz qw]. */
/* Both in-place passes are all-or-nothing: when any time stamp is refused (MSFL_BAD_ARG) the cloud, host or device, is
   left exactly as it was. */
msfl_status msfl_deskew_cloud(msfl_handle* h, const msfl_preintegration* pre,
                              msfl_point* pts_io, int n, const double rot_odom_xyzw[4],
                              const double velocity[3], const double gravity[3], msfl_mem mem);

/* UndistortScanInternal (scan_undistortion.cc:5-19), in place: p <- dq(t).cast<float>() * p.
   MSFL_BAD_ARG also for a negative time (CHECK_GE, :12). */
msfl_status msfl_undistort_cloud(msfl_handle* h, const msfl_preintegration* pre,
                                 msfl_point* pts_io, int n, msfl_mem mem);

/* ------------------------------------------------------------------------------------------ */
/* next row N1: the local map store kept resident on the device.               */
/*   replaces HybridGrid (src/slam/map/hybrid_grid.h:27-39, hybrid_grid.cc:462-534), owned by   */
/*   LaserMapping as hybrid_grid_map_corner_ / hybrid_grid_map_surf_ (laser_mapping.h:77-78).   */
/* ------------------------------------------------------------------------------------------ */

typedef struct msfl_grid_s msfl_grid;

/* HybridGrid(resolution) (laser_mapping.cc:44-45: 3.0 m cells) plus the leaf of the pcl::VoxelGrid its
   owner passes to InsertScan (laser_mapping.cc:60-68: 0.2 corner / 0.4 surf).  The grid works on
   `h`'s stream and must be destroyed before `h`. */
msfl_status msfl_grid_create(msfl_handle* h, float resolution, float leaf, msfl_grid** out);
void msfl_grid_destroy(msfl_grid* g);

/* HybridGrid::InsertScan (hybrid_grid.cc:503-521): append every point to the cell
   lround(p / resolution) and VoxelGrid-filter the touched cells in place.  `pts` are in the map
   frame (the caller transforms them, laser_mapping.cc:330-338).  MSFL_CAPACITY if a point leaves
   the +-8192-cell range of the reference (the map is then left unchanged). */
msfl_status msfl_grid_insert_scan(msfl_grid* g, const msfl_point* pts, int n, msfl_mem mem);

/* HybridGrid::GetSurroundedCloud (hybrid_grid.cc:470-501): union of the cells hit by
   pose_f32 * p + (i,j,k) m, (i,j,k) in {-1,0,1}^3, over the scan points with |p| <= 60 m.
   `scan` is in the scan frame, pose7 maps scan -> map.  Cells come out in ascending (iz,iy,ix)
   order (the reference iterates an unordered set: unspecified).  *n_out receives the full count;
   MSFL_CAPACITY if it exceeds `capacity` (the first `capacity` points are still written).
   With MSFL_MEM_DEVICE the result can be handed straight to msfl_set_map. */
msfl_status msfl_grid_get_surrounded(msfl_grid* g, const msfl_point* scan, int n, const double pose7[7],
                                     msfl_point* out, int capacity, int* n_out, msfl_mem mem);

msfl_status msfl_grid_size(msfl_grid* g, int* n_points, int* n_cells);
/* all points, cells ascending (the reference dumps the map to PLY on shutdown, laser_mapping.cc:95-113) */
msfl_status msfl_grid_dump(msfl_grid* g, msfl_point* out, int capacity, int* n_out, msfl_mem mem);


/* ------------------------------------------------------------------------------------------ */
/* The per-scan SLAM step, device-resident (BASELINE configs[2]: sequential odometry + mapping).*/
/*   replaces, for one incoming scan, the chain the reference runs across its two threads:      */
/*     RealHandleLaserCloudMessage   feature extraction            msf_loam_node.cc:160-378      */
/*     LaserOdometry::AddLaserScan   MatchScan2Scan + pose chain   laser_odometry.cc:69-95       */
/*     LaserMapping::Run             TransformAssociateToMap, MatchScan2Map (voxel filters,      */
/*                                   GetSurroundedCloud x2, gate, scan-to-map), TransformUpdate, */
/*                                   InsertScan2Map x2             laser_mapping.cc:138-258,260-338 */
/*   msfl_slam_add_scan is the LiDAR-only form (no IMU data: nothing to de-skew with);           */
/*   msfl_slam_add_scan_imu takes the per-scan IMU inputs and runs what LaserMapping::Run runs    */
/*   with them: UndistortScan before the match while the estimator is not initialised             */
/*   (laser_mapping.cc:170-176), the is_initialized matcher branch from the pre-solved pose        */
/*   (mapping_scan_matcher.cc:28-59,100-121) + DoUndistort before the insert (:197-211) once it is.*/
/* Raw points in, poses out: every intermediate (features, down-sampled clouds, the surrounded    */
/* map clouds, their kNN index, the two HybridGrid stores, the pose chain) stays in HBM, all        */
/* sizes are read by the kernels from device memory, and nothing synchronises with the host        */
/* between the upload of the scan and the 480-byte result record.  Like the reference's two        */
/* threads, the odometry chain of scan k+1 (one HIP stream) overlaps the mapping chain of scan k   */
/* (another stream; its corner side and the two voxel filters run on two more) when the caller     */
/* does not wait for every result.                                                                 */
/* ------------------------------------------------------------------------------------------ */

typedef struct msfl_slam_s msfl_slam;

typedef struct msfl_slam_config {
  float  map_resolution;      /* HybridGrid(3.0)                         laser_mapping.cc:44-45 */
  float  leaf_corner;         /* downsize_filter_corner_ 0.2             :60-63 */
  float  leaf_surf;           /* downsize_filter_surf_   0.4             :64-68 */
  int    min_map_corner;      /* MatchScan2Map runs iff corner > 10 ...  :284 */
  int    min_map_surf;        /* ... && surf > 50                        :285 */
  int    max_scan_points;     /* largest scan add_scan will see (capacity of the device buffers), e.g. 28 800 for a VLP-16 */
  int    max_rings;           /* rings of the sensor (16 / 64 / 128): bounds the per-scan sharp / flat counts the launches are sized for */
  double pose_odom2map[7];    /* initial pose_odom2map_ (identity in the reference, laser_mapping.h:88) */
  /* 0 (default): the mapping thread works on the scan's whole less-flat cloud, which is what FilterLessFlatLessCornerFeature
        (laser_mapping.cc:186,340-364) evidently means to do.
     1: reproduce what that function DOES: it copies cloud_surf_less_flat through the index list of the CORNER filter
        (`indices = downsize_filter_corner_.getIndices()` at :359 after the surf filter ran), i.e. the surf cloud is cut to its
        first n_less_sharp points, and that truncated cloud is what is voxel-filtered, used for GetSurroundedCloud, matched AND
        inserted into the map.  Use it to compare trajectories / maps with the real binary.  A scan with MORE less-sharp than
        less-flat points makes the reference read out of bounds (pcl::copyPointCloud); here: status_mapping = MSFL_BAD_ARG,
        status_imu untouched, the scan is neither matched nor inserted. */
  int    reference_quirks;
  /* 1: also produce the data products of LaserMapping::Run that reach neither the pose nor the map (round 5): cloud_full_res after
        the scan's IMU passes (UndistortScan :170-176 while not initialised, DoUndistort :206 once it is; untouched without IMU data)
        and its copy in the map frame, TransformPointCloud(cloud_full_res, pose_map_scan2world_) (:214-217: what the reference
        accumulates for its PLY dump and publishes), fetched with msfl_slam_get_clouds; the sharp / flat / less-sharp / less-flat
        clouds of PublishScan (:418-440) are index lists into that cloud.  One more launch per scan.  0 (default): not computed. */
  int    keep_clouds;
} msfl_slam_config;

typedef struct msfl_slam_result {
  double pose_odom[7];        /* pose_scan2world_ of the odometry thread (laser_odometry.cc:79), odometry frame */
  double pose_map[7];         /* pose_map_scan2world_ after MatchScan2Map (laser_mapping.cc:304-311): the SLAM output */
  double pose_curr2last[7];   /* pose_curr2last_ (laser_odometry.cc:75) */
  double pose_odom2map[7];    /* pose_odom2map_ after TransformUpdate (laser_mapping.h:59-61) */
  msfl_match_info odometry;   /* MatchScan2Scan diagnostics (status MSFL_TOO_FEW_CORRESPONDENCES where the reference returns false) */
  msfl_match_info mapping;    /* MatchScan2Map diagnostics (all zero when the gate was closed) */
  int scan_index;             /* 0-based count of add_scan calls */
  int status_extract;         /* msfl_status of the extraction (MSFL_BAD_RING, MSFL_CAPACITY, ...) */
  int status_mapping;         /* 0, or MSFL_MAP_TOO_SMALL when the gate (:284-285) kept MatchScan2Map from running,
                                 or MSFL_CAPACITY when a list did not fit the on-chip voxel filter */
  int n_full, n_sharp, n_less_sharp, n_flat, n_less_flat;   /* extraction counts */
  int n_corner_ds, n_surf_ds;           /* after the 0.2 / 0.4 m voxel filters (:264-270) */
  int n_map_corner, n_map_surf;         /* GetSurroundedCloud sizes (:273-278) */
  int grid_corner[8], grid_surf[8];     /* map store after the inserts: {points, cells, pool_top, dropped, overflow, cells touched, ...} */
  int status_imu;             /* 0, or MSFL_BAD_ARG: a less-sharp / less-flat point's relative time lies outside the scan's
                                 pre-integration span (GetDeltaQP CHECK-aborts there, scan_undistortion.cc:26-30; CHECK_GE(time, 0) :12):
                                 the scan is then neither matched nor inserted (status_mapping = MSFL_BAD_ARG as well) */
  int status_insert;          /* 0, or MSFL_CAPACITY: one of the two InsertScan calls was dropped (grid_*[3] / [4] != 0: a point outside
                                 the +-8192-cell range / not finite, or an internal capacity); the map store is then unchanged */
  int status_clouds;          /* keep_clouds: 0, or MSFL_BAD_ARG: a point of cloud_full_res OUTSIDE the less-sharp / less-flat lists (ring
                                 margins, corner neighbourhoods) has a relative time outside the pre-integration span; the reference
                                 CHECK-aborts on it inside UndistortScan / DoUndistort, here that point is left as it was */
  int reserved_;
} msfl_slam_result;

void msfl_slam_default_config(msfl_slam_config* c);

/* One SLAM pipeline = the reference's LaserOdometry + LaserMapping pair (four streams, the matchers' scratch, two map stores). */
msfl_status msfl_slam_create(const msfl_params* params, const msfl_slam_config* config, int device, msfl_slam** out);
void msfl_slam_destroy(msfl_slam* s);

/* Feed one scan as pcl::fromROSMsg delivers it (driver order, `n` points + ring ids).  With mem == MSFL_MEM_HOST the
   arrays are copied into pinned staging before the call returns; with MSFL_MEM_DEVICE they must stay untouched until
   this scan's result is out.
     result != NULL : wait for this scan's mapping result and deliver it (one synchronisation per scan);
     result == NULL : enqueue only; fetch the record later with msfl_slam_get_result.  At most two scans are in flight:
                      the call first waits for scan index-2 (almost always long done).
   Returns a status for argument / HIP errors only; per-scan outcomes are in the record. */
msfl_status msfl_slam_add_scan(msfl_slam* s, const msfl_point* pts, const uint16_t* ring, int n, msfl_mem mem,
                               msfl_slam_result* result);

/* Per-scan IMU inputs of LaserMapping::Run.  Everything here is produced by code OUTSIDE the hot path (IMU pre-integration,
   the estimator, the 15-residual IMU-only pre-solve: SURVEY.md 8f N3) and handed over like the reference's members:
     pre            BuildPreintegration(imu_buf_, odom_result.time) (laser_mapping.cc:191-194, :399); NULL = no IMU data for
                    this scan (the step then behaves like msfl_slam_add_scan, whatever is_initialized says)
     is_initialized estimator_.IsInitialized() (:173,:197; the reference switches after kInitByFirstScanNums = 50 scans, estimator.h:57)
       0: UndistortScan (:170-176 -> scan_undistortion.cc:5-19,45-62): p <- dq(p.time).cast<float>() * p on the clouds the mapping
          thread works with, BEFORE FilterLessFlatLessCornerFeature, the voxel filters, GetSurroundedCloud, the match and the insert;
          velocity / gravity / presolved_pose are not read
       1: the match starts from presolved_pose (pose_j of the IMU-only Ceres problem, mapping_scan_matcher.cc:35-59; the caller solves
          it, cf. MappingScanMatcher::SetImuPresolve in include/msfl/scan_matcher.hpp) and uses the Deskew factors with
          (dq, dp) = GetDeltaQP(pre, t) per down-sampled feature, Vi = velocity held constant (:94) and `gravity` (:100-121,154-166,
          224-236); afterwards DoUndistort (:197-211) rewrites the clouds that InsertScan2Map then inserts:
            p <- (dq(t) * p + pose_odom.rotation()^-1 * (velocity * t - 0.5 * gravity * t * t) + dp(t)).cast<float>()
          GetSurroundedCloud still uses TransformAssociateToMap's pose (:185,:273-278), like the reference.
   The odometry thread (MatchScan2Scan) never sees any of this: it works on the raw feature clouds (laser_odometry.cc:69-95).
   Which clouds: cloud_corner_less_sharp and cloud_surf_less_flat, the two that reach the pose and the map.  The reference also
   rewrites cloud_full_res (published / accumulated for the PLY dump only) and, in the not-initialised branch, the sharp / flat clouds,
   which nothing reads afterwards (FilterLessFlatLessCornerFeature hands on EMPTY sharp / flat clouds, :344-345): with
   msfl_slam_config.keep_clouds = 1 the step produces those too (msfl_slam_get_clouds).
   `pre` holds at most 2 048 samples (MSFL_CAPACITY beyond); the arrays are copied before the call returns. */
typedef struct msfl_slam_imu {
  const msfl_preintegration* pre;
  int    is_initialized;
  double velocity[3];         /* velocity_ = bias_j.head<3>() of the pre-solve (mapping_scan_matcher.cc:58) */
  double gravity[3];          /* estimator_.GetGravityVector() */
  double presolved_pose[7];   /* pose_j (:57) */
} msfl_slam_imu;

/* msfl_slam_add_scan with the scan's IMU inputs (imu == NULL: identical to msfl_slam_add_scan).  With is_initialized = 1 the
   pre-solve needs the previous scan's mapping result, so a caller normally fetches that result before feeding the next scan. */
msfl_status msfl_slam_add_scan_imu(msfl_slam* s, const msfl_point* pts, const uint16_t* ring, int n, msfl_mem mem,
                                   const msfl_slam_imu* imu, msfl_slam_result* result);

/* Wait for the scan with the given index (one of the last four fed) and deliver its record. */
msfl_status msfl_slam_get_result(msfl_slam* s, int scan_index, msfl_slam_result* result);

/* msfl_slam_config.keep_clouds: the clouds of scan `scan_index` (it must be one of the last TWO fed: the buffers belong to a set that
   the scan after next reuses).  Waits for that scan's chain.
     mem == MSFL_MEM_DEVICE: the function FILLS the pointers with device addresses (valid until msfl_slam_add_scan of scan_index + 2);
     mem == MSFL_MEM_HOST  : the CALLER fills the pointers it wants (NULL = not wanted) with host arrays of max_scan_points elements
                             each and the function copies.
   Counts are always delivered.  index lists address points of `full_scan` / `full_map` / `ring`. */
typedef struct msfl_slam_clouds {
  msfl_point* full_scan;      /* cloud_full_res in the scan frame after the IMU passes of LaserMapping::Run (:170-176,:206) */
  msfl_point* full_map;       /* TransformPointCloud(cloud_full_res, pose_map_scan2world_) (:214-217), f32 */
  uint16_t*   ring;           /* ring id of every point of the full cloud */
  int *sharp_idx, *less_sharp_idx, *flat_idx, *less_flat_idx;   /* msf_loam_node.cc:279-344 as positions in the full cloud */
  int n_full, n_sharp, n_less_sharp, n_flat, n_less_flat;
} msfl_slam_clouds;
msfl_status msfl_slam_get_clouds(msfl_slam* s, int scan_index, msfl_slam_clouds* clouds, msfl_mem mem);

/* The two map stores (hybrid_grid_map_corner_ / hybrid_grid_map_surf_), e.g. for msfl_grid_dump at shutdown
   (laser_mapping.cc:95-113).  Owned by the pipeline; do not destroy.  Waits for everything in flight. */
msfl_status msfl_slam_grids(msfl_slam* s, msfl_grid** corner, msfl_grid** surf);
const char* msfl_slam_last_error(const msfl_slam* s);

#ifdef __cplusplus
} /* extern "C" */
#endif
#endif /* MSFL_C_API_H_ */

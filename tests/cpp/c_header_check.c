/* The C ABI header must be consumable from plain C (cgo / JNI / ctypes generators parse it as C):
   compiled with gcc -std=c99 -pedantic, it only takes the addresses of the entry points it names. */
#include <stddef.h>
#include <stdio.h>
#include "msfl_c_api.h"

int main(void) {
  msfl_params p;
  msfl_default_params(&p);
  const void* fns[] = {(const void*)msfl_create, (const void*)msfl_destroy, (const void*)msfl_set_map, (const void*)msfl_match_scan2map,
                       (const void*)msfl_match_scan2map_batch, (const void*)msfl_match_scan2map_deskew, (const void*)msfl_match_scan2map_deskew_batch,
                       (const void*)msfl_match_scan2scan, (const void*)msfl_match_scan2scan_batch, (const void*)msfl_extract_features,
                       (const void*)msfl_extract_features_batch, (const void*)msfl_voxel_downsample, (const void*)msfl_voxel_downsample_batch, (const void*)msfl_voxel_downsample_batch_pair,
                       (const void*)msfl_transform_cloud, (const void*)msfl_delta_qp, (const void*)msfl_deskew_cloud, (const void*)msfl_undistort_cloud,
                       (const void*)msfl_grid_create, (const void*)msfl_grid_insert_scan, (const void*)msfl_grid_get_surrounded,
                       (const void*)msfl_slam_create, (const void*)msfl_slam_add_scan, (const void*)msfl_slam_get_result, (const void*)msfl_slam_grids,
                       (const void*)msfl_slam_destroy, (const void*)msfl_slam_add_scan_imu, (const void*)msfl_match_pairs_batch,
                       (const void*)msfl_slam_get_clouds};
  size_t n = sizeof(fns) / sizeof(fns[0]), i, ok = 0;
  for (i = 0; i < n; i++) ok += fns[i] != NULL;
  printf("api %d, %zu entry points, sizeof(msfl_point)=%zu, sizeof(msfl_slam_result)=%zu, outer_iterations=%d\n", msfl_api_version(), ok, sizeof(msfl_point),
         sizeof(msfl_slam_result), p.outer_iterations);
  return (ok == n && sizeof(msfl_point) == 16) ? 0 : 1;
}

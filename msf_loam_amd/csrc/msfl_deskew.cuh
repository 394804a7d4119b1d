// msfl_deskew.cuh — N3 / N2: the per-point passes either side of the matchers.
//
//   delta_qp_kernel          GetDeltaQP (src/slam/imu_fusion/scan_undistortion.cc:22-42): upper_bound over the
//                            pre-integration's cumulative times, Eigen slerp of delta_q, lerp of delta_p
//   deskew_cloud_kernel      the IMU deskew of laser_mapping.cc:197-211
//   undistort_cloud_kernel   UndistortScanInternal (scan_undistortion.cc:5-19): rotation only, in f32
//   transform_cloud_kernel   TransformPoint (src/common/rigid_transform.h:131-137): f32 -> f64 -> f32
//
// A pre-integration holds ~40-50 samples (one per IMU message inside a 0.1 s scan); it is staged
// once per call and every point binary-searches it.  All four are streaming kernels: 16 B read,
// 16 B (or 56 B for delta_qp) written per point.
#pragma once
#include <hip/hip_runtime.h>

#include "msfl_math.cuh"

namespace msfl {

struct PreintView {
  const double* sum_dt;    // n
  const double* dq;        // n x 4, [x y z w]
  const double* dp;        // n x 3
  int n;
};

struct DeltaQP { quat q; d3 p; bool ok; };

// Eigen 3.3 QuaternionBase::slerp: no normalisation, linear blend when |dot| >= 1 - eps
__device__ __forceinline__ quat eigen_slerp(quat a, quat b, double t) {
  const double one = 1.0 - 2.220446049250313e-16;
  const double d = a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
  const double ad = fabs(d);
  double s0, s1;
  if (ad >= one) { s0 = 1.0 - t; s1 = t; }
  else {
    const double theta = acos(ad), st = sin(theta);
    s0 = sin((1.0 - t) * theta) / st;
    s1 = sin(t * theta) / st;
  }
  if (d < 0.0) s1 = -s1;
  quat r;
  r.x = s0 * a.x + s1 * b.x; r.y = s0 * a.y + s1 * b.y; r.z = s0 * a.z + s1 * b.z; r.w = s0 * a.w + s1 * b.w;
  return r;
}

__device__ __forceinline__ DeltaQP delta_qp(const PreintView& pv, double dt) {
  DeltaQP o;
  o.q.x = o.q.y = o.q.z = 0.0; o.q.w = 1.0; o.p = mk3(0, 0, 0);
  o.ok = pv.n >= 2 && dt >= pv.sum_dt[0] && dt <= pv.sum_dt[pv.n - 1];       // the reference CHECKs this (:26-30)
  if (!o.ok) return o;
  int lo = 0, hi = pv.n;                                                       // std::upper_bound (:33)
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (dt < pv.sum_dt[mid]) hi = mid; else lo = mid + 1; }
  int idx = lo - 1;
  if (idx > pv.n - 2) idx = pv.n - 2;       // dt == back(): the reference indexes one past the end (:37-39); take s = 1 instead
  const double s = (dt - pv.sum_dt[idx]) / (pv.sum_dt[idx + 1] - pv.sum_dt[idx]);
  quat a, b;
  a.x = pv.dq[4 * idx]; a.y = pv.dq[4 * idx + 1]; a.z = pv.dq[4 * idx + 2]; a.w = pv.dq[4 * idx + 3];
  b.x = pv.dq[4 * idx + 4]; b.y = pv.dq[4 * idx + 5]; b.z = pv.dq[4 * idx + 6]; b.w = pv.dq[4 * idx + 7];
  o.q = eigen_slerp(a, b, s);
  const double* p0 = pv.dp + 3 * idx;
  o.p = mk3((1 - s) * p0[0] + s * p0[3], (1 - s) * p0[1] + s * p0[4], (1 - s) * p0[2] + s * p0[5]);
  return o;
}

__global__ void __launch_bounds__(256) delta_qp_kernel(PreintView pv, const float4* __restrict__ pts, int n, double* __restrict__ dq,
                                                        double* __restrict__ dp, int* __restrict__ bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const DeltaQP o = delta_qp(pv, (double)pts[i].w);
  if (!o.ok) *bad = 1;
  dq[4 * i] = o.q.x; dq[4 * i + 1] = o.q.y; dq[4 * i + 2] = o.q.z; dq[4 * i + 3] = o.q.w;
  dp[3 * i] = o.p.x; dp[3 * i + 1] = o.p.y; dp[3 * i + 2] = o.p.z;
}

// The in-place passes below must be all-or-nothing (the reference CHECK-aborts before it has published anything): this
// pre-pass validates every time stamp, and the in-place kernels do not touch the cloud when it has raised the flag.
__global__ void __launch_bounds__(256) cloud_times_check_kernel(PreintView pv, const float4* __restrict__ pts, int n, int need_nonneg,
                                                                 int* __restrict__ bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float t = pts[i].w;
  const double dt = (double)t;
  const bool ok = pv.n >= 2 && dt >= pv.sum_dt[0] && dt <= pv.sum_dt[pv.n - 1] && (!need_nonneg || t >= 0.f);
  if (!ok) *bad = 1;
}

struct DeskewConsts { quat rot_conj; d3 v, g; };   // pose_odom_scan2world.rotation().conjugate(), velocity, gravity

// e = (dq * e + R_odom^-1 * (v dt - 0.5 g dt dt) + dp).cast<float>()     (laser_mapping.cc:198-204)
__global__ void __launch_bounds__(256) deskew_cloud_kernel(PreintView pv, DeskewConsts c, float4* __restrict__ pts, int n,
                                                            const int* __restrict__ bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || *bad) return;                 // flag raised by cloud_times_check_kernel: leave the cloud untouched
  float4 e = pts[i];
  const double dt = (double)e.w;
  const DeltaQP o = delta_qp(pv, dt);
  if (!o.ok) return;
  const d3 a = quat_rotate(o.q, mk3((double)e.x, (double)e.y, (double)e.z));
  const d3 m = mk3(c.v.x * dt - 0.5 * c.g.x * dt * dt, c.v.y * dt - 0.5 * c.g.y * dt * dt, c.v.z * dt - 0.5 * c.g.z * dt * dt);
  const d3 b = quat_rotate(c.rot_conj, m);
  e.x = (float)(a.x + b.x + o.p.x); e.y = (float)(a.y + b.y + o.p.y); e.z = (float)(a.z + b.z + o.p.z);
  pts[i] = e;
}

// p = dq.cast<float>() * p, Eigen's _transformVector in f32 (scan_undistortion.cc:14-16); CHECK_GE(time, 0) (:12)
__global__ void __launch_bounds__(256) undistort_cloud_kernel(PreintView pv, float4* __restrict__ pts, int n, const int* __restrict__ bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || *bad) return;
  float4 e = pts[i];
  const DeltaQP o = delta_qp(pv, (double)e.w);
  if (!o.ok || !(e.w >= 0.f)) return;
  const float qx = (float)o.q.x, qy = (float)o.q.y, qz = (float)o.q.z, qw = (float)o.q.w;
  float ux = qy * e.z - qz * e.y, uy = qz * e.x - qx * e.z, uz = qx * e.y - qy * e.x;
  ux += ux; uy += uy; uz += uz;
  const float cx = qy * uz - qz * uy, cy = qz * ux - qx * uz, cz = qx * uy - qy * ux;
  e.x = e.x + qw * ux + cx; e.y = e.y + qw * uy + cy; e.z = e.z + qw * uz + cz;
  pts[i] = e;
}

__global__ void __launch_bounds__(256) transform_cloud_kernel(const float4* in, int n, const double* __restrict__ pose,
                                                               float4* out) {   // in == out allowed
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = in[i];
  const float3 q = transform_point_f32(load_pose(pose), p.x, p.y, p.z);
  out[i] = make_float4(q.x, q.y, q.z, p.w);
}

}  // namespace msfl

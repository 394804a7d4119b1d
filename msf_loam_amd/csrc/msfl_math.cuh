// msfl_math.cuh — f64 device math for the scan-matching kernels (gfx950).
//
// Everything here runs per lane in registers: no dynamic indexing of local arrays (that would
// spill to scratch), all loops over 3/5/6 are compile-time unrolled.  The translation unit is
// compiled with -ffp-contract=off so that the f32 distance arithmetic and the f64 pose transform
// round exactly like the reference's generic x86-64 build (no FMA contraction); see DESIGN.md §4.
//
// Reference semantics restated (paths relative to the MSF_LOAM tree):
//   quat_rotate      Eigen::Quaterniond * Vector3d       used by common/rigid_transform.h:132-138
//   quat_to_matrix   Eigen::Quaterniond::toRotationMatrix  lidar_factor.cc:19,39
//   pose_plus        PoseLocalParameterization::Plus      imu_fusion/pose_local_parameterization.cc:6-21
//   sym_eigen3       SelfAdjointEigenSolver<Matrix3d>     mapping_scan_matcher.cc:141
//   lstsq5x3         colPivHouseholderQr().solve()        mapping_scan_matcher.cc:210
#pragma once
#include <hip/hip_runtime.h>

namespace msfl {

struct d3 { double x, y, z; };

__device__ __forceinline__ d3 mk3(double x, double y, double z) { d3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ d3 operator+(d3 a, d3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ d3 operator-(d3 a, d3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ d3 operator*(double s, d3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ double dot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ d3 cross(d3 a, d3 b) {
  return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// the same through one reciprocal square root (within an ulp or two per component): plane fit only
__device__ __forceinline__ d3 normalized_rsq(d3 v);
// Eigen 3.3 normalized(): guarded against the zero vector
__device__ __forceinline__ d3 normalized(d3 v) {
  const double z = dot(v, v);
  if (z > 0.0) { const double s = sqrt(z); return mk3(v.x / s, v.y / s, v.z / s); }
  return v;
}

// 1 / x for x in a benign range (no scaling / fix-up steps of the IEEE division sequence): v_rcp_f64 + two Newton
// steps, within an ulp or two of the correctly rounded quotient
// Build switch -DMSFL_IEEE_DIV=1 restores the IEEE division / square root everywhere (parity runs: a decision sitting
// within an ulp of a threshold then rounds like the oracle's); the fast forms also fall back to the IEEE sequence for
// arguments outside [1e-280, 1e280], where the unscaled Newton steps would lose bits or overflow.
#ifndef MSFL_IEEE_DIV
#define MSFL_IEEE_DIV 0
#endif
__device__ __forceinline__ double fast_rcp(double x) {
  if (MSFL_IEEE_DIV || !(fabs(x) > 1e-280 && fabs(x) < 1e280)) return 1.0 / x;
  double y = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-x, y, 1.0);
  return __builtin_fma(y, e, y);
}
// 1 / sqrt(x), x > 0 and far from the denormal range: v_rsq_f64 + two Newton steps
__device__ __forceinline__ double fast_rsqrt(double x) {
  if (MSFL_IEEE_DIV || !(x > 1e-280 && x < 1e280)) return 1.0 / sqrt(x);
  double y = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  double e = __builtin_fma(-hx * y, y, 0.5);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-hx * y, y, 0.5);
  return __builtin_fma(y, e, y);
}

__device__ __forceinline__ d3 normalized_rsq(d3 v) {
  const double z = dot(v, v);
  if (!MSFL_IEEE_DIV && z > 1e-280 && z < 1e280) { const double s = fast_rsqrt(z); return mk3(v.x * s, v.y * s, v.z * s); }
  return normalized(v);
}

struct quat { double x, y, z, w; };
struct pose7 { d3 t; quat q; };

__device__ __forceinline__ pose7 load_pose(const double* __restrict__ p) {
  pose7 r;
  r.t = mk3(p[0], p[1], p[2]);
  r.q.x = p[3]; r.q.y = p[4]; r.q.z = p[5]; r.q.w = p[6];
  return r;
}
__device__ __forceinline__ void store_pose(double* __restrict__ p, const pose7& a) {
  p[0] = a.t.x; p[1] = a.t.y; p[2] = a.t.z; p[3] = a.q.x; p[4] = a.q.y; p[5] = a.q.z; p[6] = a.q.w;
}

// uv = 2 (q.vec x v);  v + w uv + q.vec x uv
__device__ __forceinline__ d3 quat_rotate(const quat& q, d3 v) {
  const d3 qv = mk3(q.x, q.y, q.z);
  d3 uv = cross(qv, v);
  uv = uv + uv;
  const d3 c = cross(qv, uv);
  return mk3(v.x + q.w * uv.x + c.x, v.y + q.w * uv.y + c.y, v.z + q.w * uv.z + c.z);
}

struct mat3 { double m[9]; };  // row-major; only ever indexed with constants

__device__ __forceinline__ mat3 quat_to_matrix(const quat& q) {
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  mat3 R;
  R.m[0] = 1 - (tyy + tzz); R.m[1] = txy - twz;       R.m[2] = txz + twy;
  R.m[3] = txy + twz;       R.m[4] = 1 - (txx + tzz); R.m[5] = tyz - twx;
  R.m[6] = txz - twy;       R.m[7] = tyz + twx;       R.m[8] = 1 - (txx + tyy);
  return R;
}

__device__ __forceinline__ quat quat_mul(const quat& a, const quat& b) {
  quat o;
  o.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  o.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  o.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  o.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return o;
}
__device__ __forceinline__ quat quat_normalized(quat q) {
  const double n2 = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
  if (n2 > 0.0) { const double n = sqrt(n2); q.x /= n; q.y /= n; q.z /= n; q.w /= n; }
  return q;
}

// Utility::deltaQ (imu_fusion/utility.h:7-31)
__device__ __forceinline__ quat delta_q(d3 v) {
  const double theta = sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
  const double half_theta = 0.5 * theta;
  double imag, real, sn;
  sincos(half_theta, &sn, &real);          // one argument reduction for both (same values as sin() and cos())
  if (theta < 1e-6) {
    const double t2 = theta * theta, t4 = t2 * t2;
    imag = 0.5 - (1 / 48.) * t2 + (1 / 3840.) * t4;
  } else {
    imag = sn / theta;
  }
  quat q; q.x = imag * v.x; q.y = imag * v.y; q.z = imag * v.z; q.w = real;
  return q;
}

// x (+) delta : t += d[0:3]; q = (q * deltaQ(d[3:6])).normalized()
__device__ __forceinline__ pose7 pose_plus(const pose7& x, d3 dt, d3 dth) {
  pose7 o;
  o.t = x.t + dt;
  o.q = quat_normalized(quat_mul(x.q, delta_q(dth)));
  return o;
}

// TransformPoint: f32 -> f64 -> q*p + t -> f32  (rigid_transform.h:132-138)
__device__ __forceinline__ float3 transform_point_f32(const pose7& T, float x, float y, float z) {
  const d3 r = quat_rotate(T.q, mk3((double)x, (double)y, (double)z));
  return make_float3((float)(r.x + T.t.x), (float)(r.y + T.t.y), (float)(r.z + T.t.z));
}

// ---------------------------------------------------------------------------------------------
// symmetric 3x3 eigen-decomposition, cyclic Jacobi, all indices compile-time
// ---------------------------------------------------------------------------------------------
struct sym3 { double a00, a01, a02, a11, a12, a22; };

template <int P, int Q>
__device__ __forceinline__ void jacobi_rotate(double (&a)[3][3], double (&v)[3][3]) {
#pragma clang fp contract(fast)
  const double apq = a[P][Q];
  if (apq == 0.0) return;
  const double theta = (a[Q][Q] - a[P][P]) / (2.0 * apq);
  const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
  const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const double akp = a[k][P], akq = a[k][Q];
    a[k][P] = c * akp - s * akq;
    a[k][Q] = s * akp + c * akq;
  }
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const double apk = a[P][k], aqk = a[Q][k];
    a[P][k] = c * apk - s * aqk;
    a[Q][k] = s * apk + c * aqk;
  }
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const double vkp = v[k][P], vkq = v[k][Q];
    v[k][P] = c * vkp - s * vkq;
    v[k][Q] = s * vkp + c * vkq;
  }
}

// Returns the two largest eigenvalues (mid, max) and the eigenvector of the largest.
__device__ __forceinline__ void sym_eigen3_top(const sym3& S, double& ev_mid, double& ev_max, d3& vec_max) {
  double a[3][3] = {{S.a00, S.a01, S.a02}, {S.a01, S.a11, S.a12}, {S.a02, S.a12, S.a22}};
  double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 30; sweep++) {
    const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
    const double dg = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2];
    if (off <= 1e-300 || off <= 1e-34 * dg) break;
    jacobi_rotate<0, 1>(a, v);
    jacobi_rotate<0, 2>(a, v);
    jacobi_rotate<1, 2>(a, v);
  }
  const double d0 = a[0][0], d1 = a[1][1], d2 = a[2][2];
  // ascending selection without dynamic indexing (ties: lower index first, like a stable sort)
  int imax = 0; double emax = d0;
  if (d1 > emax) { emax = d1; imax = 1; }
  if (d2 > emax) { emax = d2; imax = 2; }
  const double r0 = (imax == 0) ? d1 : d0;
  const double r1 = (imax == 2) ? d1 : d2;
  ev_mid = r0 > r1 ? r0 : r1;
  ev_max = emax;
  vec_max = (imax == 0) ? mk3(v[0][0], v[1][0], v[2][0])
          : (imax == 1) ? mk3(v[0][1], v[1][1], v[2][1])
                        : mk3(v[0][2], v[1][2], v[2][2]);
}

// ---------------------------------------------------------------------------------------------
// 5x3 least squares by column-pivoted Householder QR, all indices compile-time
// ---------------------------------------------------------------------------------------------
template <int K, bool PIVOT = true>
__device__ __forceinline__ void qr_step(double (&A)[5][3], double (&b)[5], int (&perm)[3], double (&diag)[3],
                                        double& maxpivot) {
#pragma clang fp contract(fast)   // f64 fits: FMA only changes rounding at 1e-16, far inside the 1e-9 record tolerance
  // pivot: remaining column with the largest norm below row K (first wins ties)
  double cn[3] = {0, 0, 0};
#pragma unroll
  for (int j = K; j < 3; j++) {
    double s = 0.0;
#pragma unroll
    for (int i = K; i < 5; i++) s += A[i][j] * A[i][j];
    cn[j] = s;
  }
  int best = K; double bestn = cn[K];
  if (PIVOT) {
#pragma unroll
    for (int j = K + 1; j < 3; j++) if (cn[j] > bestn) { bestn = cn[j]; best = j; }
  }
#pragma unroll
  for (int j = K + 1; j < 3; j++) {
    if (PIVOT && best == j) {
#pragma unroll
      for (int i = 0; i < 5; i++) { const double t = A[i][K]; A[i][K] = A[i][j]; A[i][j] = t; }
      const int t = perm[K]; perm[K] = perm[j]; perm[j] = t;
    }
  }
  const double norm = sqrt(bestn);
  if (norm == 0.0) { diag[K] = 0.0; return; }
  const double alpha = (A[K][K] > 0.0) ? -norm : norm;
  double v[5];
#pragma unroll
  for (int i = 0; i < 5; i++) v[i] = 0.0;
  v[K] = A[K][K] - alpha;
#pragma unroll
  for (int i = K + 1; i < 5; i++) v[i] = A[i][K];
  double vtv = 0.0;
#pragma unroll
  for (int i = K; i < 5; i++) vtv += v[i] * v[i];
  if (vtv > 0.0) {
    const double tau = 2.0 / vtv;          // one division per reflector (Eigen applies H = I - tau v v^T the same way)
#pragma unroll
    for (int j = K; j < 3; j++) {
      double s = 0.0;
#pragma unroll
      for (int i = K; i < 5; i++) s += v[i] * A[i][j];
      s = s * tau;
#pragma unroll
      for (int i = K; i < 5; i++) A[i][j] -= s * v[i];
    }
    double s = 0.0;
#pragma unroll
    for (int i = K; i < 5; i++) s += v[i] * b[i];
    s = s * tau;
#pragma unroll
    for (int i = K; i < 5; i++) b[i] -= s * v[i];
  }
  A[K][K] = alpha;
  diag[K] = alpha;
  if (fabs(alpha) > maxpivot) maxpivot = fabs(alpha);
}

// Solves min |A x - b|; rows of A are the 5 points.
__device__ __forceinline__ d3 lstsq5x3(double (&A)[5][3], double (&b)[5]) {
  int perm[3] = {0, 1, 2};
  double diag[3] = {0, 0, 0};
  double maxpivot = 0.0;
  qr_step<0>(A, b, perm, diag, maxpivot);
  qr_step<1>(A, b, perm, diag, maxpivot);
  qr_step<2>(A, b, perm, diag, maxpivot);
  const double thresh = 2.220446049250313e-16 * 3.0 * maxpivot;   // Eigen: epsilon * diagonalSize
  // pivots are non-increasing in magnitude up to rounding; count the leading ones above threshold
  const bool k0 = fabs(diag[0]) > thresh, k1 = fabs(diag[1]) > thresh, k2 = fabs(diag[2]) > thresh;
  const int rank = (int)k0 + (int)k1 + (int)k2;
  double y0 = 0, y1 = 0, y2 = 0;
  if (rank == 3) {                             // the generic case: reciprocals (v_rcp_f64 + two Newton steps) instead of three IEEE divisions
    y2 = b[2] * fast_rcp(A[2][2]);
    y1 = (b[1] - A[1][2] * y2) * fast_rcp(A[1][1]);
    y0 = (b[0] - A[0][1] * y1 - A[0][2] * y2) * fast_rcp(A[0][0]);
  } else if (rank == 2) {
    y1 = b[1] / A[1][1];
    y0 = (b[0] - A[0][1] * y1) / A[0][0];
  } else if (rank == 1) {
    y0 = b[0] / A[0][0];
  }
  // x[perm[k]] = y[k]
  double x0 = 0, x1 = 0, x2 = 0;
  const double ys[3] = {y0, y1, y2};
#pragma unroll
  for (int k = 0; k < 3; k++) {
    if (perm[k] == 0) x0 = ys[k];
    else if (perm[k] == 1) x1 = ys[k];
    else x2 = ys[k];
  }
  return mk3(x0, x1, x2);
}

// The same least-squares solution without column pivoting, for the well-conditioned systems that are the rule (five points of
// a plane a few tens of metres from the origin: condition number 1e2-1e3).  Pivoting is what Eigen's colPivHouseholderQr does
// and what decides its rank; it is also a fifth of the fit kernel's instructions (three column norms, a compare chain and
// fifteen 64-bit selects per step).  `ok` = every |r_kk| above 1e-5 of the largest: the solution then agrees with the pivoted
// one to ~1e-11 relative (both are backward stable, the difference is cond x eps) and the rank is 3 in both; otherwise the
// caller runs lstsq5x3 on a fresh copy, so every rank-deficient or ill-conditioned neighbourhood is decided by the
// reference's algorithm.
__device__ __forceinline__ d3 lstsq5x3_fast(double (&A)[5][3], double (&b)[5], bool& ok) {
  int perm[3] = {0, 1, 2};
  double diag[3] = {0, 0, 0};
  double maxpivot = 0.0;
  qr_step<0, false>(A, b, perm, diag, maxpivot);
  qr_step<1, false>(A, b, perm, diag, maxpivot);
  qr_step<2, false>(A, b, perm, diag, maxpivot);
  const double dmin = fmin(fabs(diag[0]), fmin(fabs(diag[1]), fabs(diag[2])));
  ok = dmin > 1e-5 * maxpivot;                                   // false for NaN too
  if (!ok) return mk3(0, 0, 0);
  const double y2 = b[2] * fast_rcp(A[2][2]);
  const double y1 = (b[1] - A[1][2] * y2) * fast_rcp(A[1][1]);
  const double y0 = (b[0] - A[0][1] * y1 - A[0][2] * y2) * fast_rcp(A[0][0]);
  return mk3(y0, y1, y2);
}

// Direction of the least-squares solution of A x = -1 (rows of A: five points) WITHOUT forming x (round 4b).  With the centroid c
// and q_j = p_j - c (sum q_j = 0):  sum_j (x.p_j + 1)^2 = 5 (x.c + 1)^2 + x^T Q x,  Q = sum_j q_j q_j^T, so
//   (Q + 5 c c^T) x = -5 c   =>   x = -5 Q^-1 c / (1 + 5 c^T Q^-1 c)   (Sherman-Morrison; the factor is positive for SPD Q)
// and the direction of x is that of -adj(Q) c: six cofactors and a 3 x 3 product, no square root, no division, no 5 x 3
// factorisation.  It is also the limit for exactly coplanar points (Q singular: adj(Q) = l1 l2 n n^T).  The plane fit only uses
// x / |x| (mapping_scan_matcher.cc:211), so this replaces the QR on the well-conditioned systems that are the rule; `ok` = false sends
// the neighbourhood to lstsq5x3, the reference's pivoted QR, which alone decides rank-deficient and ill-conditioned ones:
//  * the diagonal of the unpivoted Cholesky factor of A^T A = Q + 5 c c^T (= |r_kk| of the unpivoted QR of A, the criterion of the
//    round-3 fast path) must span less than 1e5: min d_k > 1e-10 max d_k on its squares;
//  * the cofactors and the product cancel by l1 / l2 (in-plane anisotropy) and |c| / |n.c| (grazing view): |adj(Q) c| must be above
//    1e-6 of the magnitude of its terms, which bounds the formula's own rounding error by ~2e-10 relative (typically 1e-14).
// `q` are the centred points.
__device__ __forceinline__ d3 plane_normal_centred(const double (&q)[5][3], d3 c, bool& ok) {
#pragma clang fp contract(fast)
  double Q00 = 0, Q01 = 0, Q02 = 0, Q11 = 0, Q12 = 0, Q22 = 0;
#pragma unroll
  for (int j = 0; j < 5; j++) {
    Q00 += q[j][0] * q[j][0]; Q01 += q[j][0] * q[j][1]; Q02 += q[j][0] * q[j][2];
    Q11 += q[j][1] * q[j][1]; Q12 += q[j][1] * q[j][2]; Q22 += q[j][2] * q[j][2];
  }
  const double a00 = Q11 * Q22 - Q12 * Q12, a01 = Q02 * Q12 - Q01 * Q22, a02 = Q01 * Q12 - Q02 * Q11;
  const double a11 = Q00 * Q22 - Q02 * Q02, a12 = Q01 * Q02 - Q00 * Q12, a22 = Q00 * Q11 - Q01 * Q01;
  const d3 y = mk3(a00 * c.x + a01 * c.y + a02 * c.z, a01 * c.x + a11 * c.y + a12 * c.z, a02 * c.x + a12 * c.y + a22 * c.z);
  // conditioning of A^T A = Q + 5 c c^T through its Cholesky diagonal (squares); reciprocals are good enough for a threshold
  const double M00 = Q00 + 5.0 * c.x * c.x, M01 = Q01 + 5.0 * c.x * c.y, M02 = Q02 + 5.0 * c.x * c.z;
  const double M11 = Q11 + 5.0 * c.y * c.y, M12 = Q12 + 5.0 * c.y * c.z, M22 = Q22 + 5.0 * c.z * c.z;
  const double d0 = M00;
  const double i0 = fast_rcp(d0);
  const double d1 = M11 - M01 * M01 * i0;
  const double l21 = M12 - M01 * M02 * i0;
  const double d2 = M22 - M02 * M02 * i0 - l21 * l21 * fast_rcp(d1);
  const double dmin = fmin(d0, fmin(d1, d2)), dmax = fmax(d0, fmax(d1, d2));
  const double trq = Q00 + Q11 + Q22;
  const double cm = fmax(fabs(c.x), fmax(fabs(c.y), fabs(c.z)));
  const double ym = fmax(fabs(y.x), fmax(fabs(y.y), fabs(y.z)));
  ok = dmin > 1e-10 * dmax && ym > 1e-6 * (trq * trq * cm);      // false for NaN too
  return mk3(-y.x, -y.y, -y.z);
}

}  // namespace msfl

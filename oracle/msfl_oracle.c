/*
 * msfl_oracle.c — CPU ORACLE (test infrastructure, see msfl_oracle.h).
 *
 * Plain C99 restatement of the MSF_LOAM scan-matching hot path.  Every function cites the
 * reference file:line it follows (paths relative to /root/reference).  PARITY UNPINNED by the
 * reference's own tests (it has none on this path); pinned instead by the independent
 * numpy/scipy formulations in tests/, whose trust-region loop in turn reproduces the run the
 * Ceres tutorial prints for Powell's function (tests/test_oracle_lm_trajectory.py).
 *
 * Compile with -ffp-contract=off: the reference is built for generic x86-64 (no FMA), and the
 * f32 kNN distances / curvature sums below must round exactly like that build.
 */
#include "msfl_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* =============================================================================================
 * Small math (Eigen restatements)
 * ============================================================================================= */

static void cross3(const double a[3], const double b[3], double o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
static double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

/* Eigen 3.3 MatrixBase::normalized(): guarded against the zero vector. */
static void normalized3(const double v[3], double o[3]) {
  double z = dot3(v, v);
  if (z > 0.0) {
    double s = sqrt(z);
    o[0] = v[0] / s; o[1] = v[1] / s; o[2] = v[2] / s;
  } else {
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
  }
}

/* Eigen QuaternionBase::_transformVector:  uv = 2 (q.vec x v);  v + w uv + q.vec x uv */
void orc_quat_rotate(const double q[4], const double v[3], double out[3]) {
  double uv[3], c[3];
  cross3(q, v, uv);
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  cross3(q, uv, c);
  out[0] = v[0] + q[3] * uv[0] + c[0];
  out[1] = v[1] + q[3] * uv[1] + c[1];
  out[2] = v[2] + q[3] * uv[2] + c[2];
}

/* Eigen QuaternionBase::toRotationMatrix */
void orc_quat_to_matrix(const double q[4], double R[9]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

/* Eigen quaternion product a*b, [x y z w] storage */
static void quat_mul(const double a[4], const double b[4], double o[4]) {
  double w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  double x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  double y = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  double z = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}
static void quat_normalize(double q[4]) {
  double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (n2 > 0.0) {
    double n = sqrt(n2);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
  }
}

/* rigid_transform.h:132-138  TransformPoint */
void orc_transform_point(const double pose[7], const float in[3], float out[3]) {
  double v[3] = {(double)in[0], (double)in[1], (double)in[2]}, r[3];
  orc_quat_rotate(pose + 3, v, r);
  out[0] = (float)(r[0] + pose[0]);
  out[1] = (float)(r[1] + pose[1]);
  out[2] = (float)(r[2] + pose[2]);
}

/* imu_fusion/utility.h:7-31  Utility::deltaQ -> [x y z w] */
static void delta_q(const double v[3], double q[4]) {
  const double kAngleEpisode = 1e-6;
  double theta = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  double half_theta = 0.5 * theta;
  double imag_factor;
  double real_factor = cos(half_theta);
  if (theta < kAngleEpisode) {
    double theta_sq = theta * theta;
    double theta_po4 = theta_sq * theta_sq;
    imag_factor = 0.5 - (1 / 48.) * theta_sq + (1 / 3840.) * theta_po4;
  } else {
    imag_factor = sin(half_theta) / theta;
  }
  q[0] = imag_factor * v[0]; q[1] = imag_factor * v[1]; q[2] = imag_factor * v[2]; q[3] = real_factor;
}

/* pose_local_parameterization.cc:6-21 */
void orc_pose_plus(const double x[7], const double delta[6], double out[7]) {
  double dq[4], q[4];
  delta_q(delta + 3, dq);
  out[0] = x[0] + delta[0]; out[1] = x[1] + delta[1]; out[2] = x[2] + delta[2];
  quat_mul(x + 3, dq, q);
  quat_normalize(q);
  out[3] = q[0]; out[4] = q[1]; out[5] = q[2]; out[6] = q[3];
}

/* rigid_transform.h:105-111 */
void orc_pose_compose(const double a[7], const double b[7], double out[7]) {
  double r[3], q[4];
  orc_quat_rotate(a + 3, b, r);
  quat_mul(a + 3, b + 3, q);
  quat_normalize(q);
  out[0] = r[0] + a[0]; out[1] = r[1] + a[1]; out[2] = r[2] + a[2];
  out[3] = q[0]; out[4] = q[1]; out[5] = q[2]; out[6] = q[3];
}

/* Symmetric 3x3 eigen-decomposition, cyclic Jacobi in f64 (restates what
   Eigen::SelfAdjointEigenSolver<Matrix3d>::compute delivers: eigenvalues ascending, orthonormal
   eigenvectors; mapping_scan_matcher.cc:141-147). */
void orc_sym_eigen3(const double A[9], double evals[3], double V[9]) {
  double a[3][3], v[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) { a[i][j] = A[3 * i + j]; v[i][j] = (i == j) ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
    double diag = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2];
    if (off <= 1e-300 || off <= 1e-34 * diag) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        double apq = a[p][q];
        if (apq == 0.0) continue;
        double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
        double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; k++) {         /* A <- A G */
          double akp = a[k][p], akq = a[k][q];
          a[k][p] = c * akp - s * akq;
          a[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; k++) {         /* A <- G^T A */
          double apk = a[p][k], aqk = a[q][k];
          a[p][k] = c * apk - s * aqk;
          a[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; k++) {         /* V <- V G */
          double vkp = v[k][p], vkq = v[k][q];
          v[k][p] = c * vkp - s * vkq;
          v[k][q] = s * vkp + c * vkq;
        }
      }
  }
  int order[3] = {0, 1, 2};
  double d[3] = {a[0][0], a[1][1], a[2][2]};
  for (int i = 0; i < 2; i++)
    for (int j = i + 1; j < 3; j++)
      if (d[order[j]] < d[order[i]]) { int t = order[i]; order[i] = order[j]; order[j] = t; }
  for (int c = 0; c < 3; c++) {
    evals[c] = d[order[c]];
    for (int r = 0; r < 3; r++) V[3 * r + c] = v[r][order[c]];
  }
}

/* Column-pivoted Householder QR least squares for a 5x3 system
   (Eigen ColPivHouseholderQR::solve, mapping_scan_matcher.cc:210). */
int orc_lstsq_5x3(const double A_in[15], const double b_in[5], double x[3]) {
  double A[5][3], b[5];
  int perm[3] = {0, 1, 2};
  for (int i = 0; i < 5; i++) { b[i] = b_in[i]; for (int j = 0; j < 3; j++) A[i][j] = A_in[3 * i + j]; }
  double maxpivot = 0.0;
  int rank = 3;
  double diag[3];
  for (int k = 0; k < 3; k++) {
    /* pivot: remaining column with the largest norm below row k */
    int best = k; double bestn = -1.0;
    for (int j = k; j < 3; j++) {
      double s = 0.0;
      for (int i = k; i < 5; i++) s += A[i][j] * A[i][j];
      if (s > bestn) { bestn = s; best = j; }
    }
    if (best != k) {
      for (int i = 0; i < 5; i++) { double t = A[i][k]; A[i][k] = A[i][best]; A[i][best] = t; }
      int t = perm[k]; perm[k] = perm[best]; perm[best] = t;
    }
    /* Householder vector for column k */
    double norm = sqrt(bestn);
    if (norm == 0.0) { diag[k] = 0.0; continue; }
    double alpha = (A[k][k] > 0.0) ? -norm : norm;   /* beta = -sign(x0) |x| */
    double v[5];
    for (int i = 0; i < 5; i++) v[i] = 0.0;
    v[k] = A[k][k] - alpha;
    for (int i = k + 1; i < 5; i++) v[i] = A[i][k];
    double vtv = 0.0;
    for (int i = k; i < 5; i++) vtv += v[i] * v[i];
    if (vtv > 0.0) {
      for (int j = k; j < 3; j++) {
        double s = 0.0;
        for (int i = k; i < 5; i++) s += v[i] * A[i][j];
        s = 2.0 * s / vtv;
        for (int i = k; i < 5; i++) A[i][j] -= s * v[i];
      }
      double s = 0.0;
      for (int i = k; i < 5; i++) s += v[i] * b[i];
      s = 2.0 * s / vtv;
      for (int i = k; i < 5; i++) b[i] -= s * v[i];
    }
    A[k][k] = alpha;
    for (int i = k + 1; i < 5; i++) A[i][k] = 0.0;
    diag[k] = alpha;
    if (fabs(alpha) > maxpivot) maxpivot = fabs(alpha);
  }
  /* rank: Eigen threshold = epsilon * diagonalSize, relative to the largest pivot */
  const double thresh = DBL_EPSILON * 3.0 * maxpivot;
  rank = 0;
  for (int k = 0; k < 3; k++) if (fabs(diag[k]) > thresh) rank++;
  double y[3] = {0, 0, 0};
  for (int k = rank - 1; k >= 0; k--) {
    double s = b[k];
    for (int j = k + 1; j < rank; j++) s -= A[k][j] * y[j];
    y[k] = s / A[k][k];
  }
  for (int k = 0; k < 3; k++) x[perm[k]] = (k < rank) ? y[k] : 0.0;
  return rank;
}

/* mapping_scan_matcher.cc:130-151 */
int orc_edge_fit(const float nbr[5][3], double ratio, double C[3], double N[3]) {
  double m[5][3], center[3] = {0, 0, 0};
  for (int j = 0; j < 5; j++)
    for (int a = 0; a < 3; a++) { m[j][a] = (double)nbr[j][a]; center[a] += m[j][a]; }
  for (int a = 0; a < 3; a++) center[a] /= 5.0;
  double cov[9] = {0};
  for (int j = 0; j < 5; j++) {
    double d[3] = {m[j][0] - center[0], m[j][1] - center[1], m[j][2] - center[2]};
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) cov[3 * a + b] += d[a] * d[b];
  }
  double ev[3], V[9];
  orc_sym_eigen3(cov, ev, V);
  if (!(ev[2] > ratio * ev[1])) return 0;
  double dir[3] = {V[2], V[5], V[8]};
  double pa[3], pb[3], ab[3];
  for (int a = 0; a < 3; a++) {
    pa[a] = 0.1 * dir[a] + center[a];
    pb[a] = -0.1 * dir[a] + center[a];
    ab[a] = pa[a] - pb[a];
  }
  normalized3(ab, N);
  C[0] = pa[0]; C[1] = pa[1]; C[2] = pa[2];
  return 1;
}

/* mapping_scan_matcher.cc:199-222 */
int orc_plane_fit(const float nbr[5][3], double tol, double C[3], double N[3]) {
  double A[15], b[5], n[3], center[3] = {0, 0, 0};
  for (int j = 0; j < 5; j++) {
    b[j] = -1.0;
    for (int a = 0; a < 3; a++) { A[3 * j + a] = (double)nbr[j][a]; center[a] += A[3 * j + a]; }
  }
  for (int a = 0; a < 3; a++) center[a] /= 5.0;
  orc_lstsq_5x3(A, b, n);
  normalized3(n, N);   /* Vector3d::normalize() */
  for (int j = 0; j < 5; j++) {
    double d = N[0] * (A[3 * j] - center[0]) + N[1] * (A[3 * j + 1] - center[1]) + N[2] * (A[3 * j + 2] - center[2]);
    if (fabs(d) > tol) return 0;
  }
  C[0] = center[0]; C[1] = center[1]; C[2] = center[2];
  return 1;
}

/* R * skew(p) -> M (row-major 3x3); utility.h:33-41 skewSymmetric */
static void r_skew(const double R[9], const double p[3], double M[9]) {
  /* skew(p) = [0 -pz py; pz 0 -px; -py px 0]; column j of R*skew = R * skew[:,j] */
  for (int i = 0; i < 3; i++) {
    const double r0 = R[3 * i], r1 = R[3 * i + 1], r2 = R[3 * i + 2];
    M[3 * i + 0] = r1 * p[2] - r2 * p[1];
    M[3 * i + 1] = -r0 * p[2] + r2 * p[0];
    M[3 * i + 2] = r0 * p[1] - r1 * p[0];
  }
}

/* lidar_factor.cc:7-24 */
void orc_edge_factor(const double pose[7], const double p[3], const double C[3], const double N[3],
                     double r[3], double J[21]) {
  double w[3], d[3];
  orc_quat_rotate(pose + 3, p, w);
  for (int a = 0; a < 3; a++) d[a] = w[a] + pose[a] - C[a];
  cross3(N, d, r);
  if (!J) return;
  double R[9], M[9];
  orc_quat_to_matrix(pose + 3, R);
  r_skew(R, p, M);
  const double S[9] = {0, -N[2], N[1], N[2], 0, -N[0], -N[1], N[0], 0};
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) {
      J[7 * i + j] = S[3 * i + j];
      double s = 0.0;
      for (int k = 0; k < 3; k++) s += S[3 * i + k] * M[3 * k + j];
      J[7 * i + 3 + j] = -s;
    }
    J[7 * i + 6] = 0.0;
  }
}

/* lidar_factor.cc:26-44 */
void orc_plane_factor(const double pose[7], const double p[3], const double C[3], const double N[3],
                      double r[1], double J[7]) {
  double w[3], d[3];
  orc_quat_rotate(pose + 3, p, w);
  for (int a = 0; a < 3; a++) d[a] = w[a] + pose[a] - C[a];
  r[0] = dot3(N, d);
  if (!J) return;
  double R[9], M[9];
  orc_quat_to_matrix(pose + 3, R);
  r_skew(R, p, M);
  for (int j = 0; j < 3; j++) {
    J[j] = N[j];
    J[3 + j] = -(N[0] * M[j] + N[1] * M[3 + j] + N[2] * M[6 + j]);
  }
  J[6] = 0.0;
}

/* =============================================================================================
 * Exact kNN with FLANN L2_Simple<float> distances
 * ============================================================================================= */

static inline float dist2f(const orc_point* a, const float q[3]) {
  /* flann::L2_Simple<float>: result += diff*diff over x,y,z, all in f32 */
  float dx = a->x - q[0], dy = a->y - q[1], dz = a->z - q[2];
  float r = dx * dx;
  r = r + dy * dy;
  r = r + dz * dz;
  return r;
}

/* Behaviours the reference's toolchain leaves UNSPECIFIED, switchable for the sensitivity tests only
   (tests/test_oracle_unspecified.py; 0 = the documented choice):
     sort ties    std::sort of sector indices by curvature is unstable (msf_loam_node.cc:263-267)
     kNN ties     FLANN returns equal distances in an order that depends on its tree traversal (mapping_scan_matcher.cc:125)
     atan2        unqualified atan2(float, float) may bind to the float or the double overload (msf_loam_node.cc:131,139) */
static int g_sort_ties_reverse = 0, g_knn_ties_reverse = 0, g_atan2_float = 0;
void orc_set_unspecified(int sort_ties_reverse, int knn_ties_reverse, int atan2_float) {
  g_sort_ties_reverse = sort_ties_reverse; g_knn_ties_reverse = knn_ties_reverse; g_atan2_float = atan2_float;
}
static inline int idx_after(int a, int b) { return g_knn_ties_reverse ? a < b : a > b; }   /* a loses the tie against b */

/* sorted insertion, ties broken by ascending index */
static inline void knn_insert(int k, int* idx, float* d2, int* count, float d, int i) {
  int c = *count;
  if (c == k) {
    if (d > d2[k - 1] || (d == d2[k - 1] && idx_after(i, idx[k - 1]))) return;
  }
  int pos = (c < k) ? c : k - 1;
  while (pos > 0 && (d2[pos - 1] > d || (d2[pos - 1] == d && idx_after(idx[pos - 1], i)))) {
    d2[pos] = d2[pos - 1]; idx[pos] = idx[pos - 1]; pos--;
  }
  d2[pos] = d; idx[pos] = i;
  if (c < k) *count = c + 1;
}

void orc_knn_brute(const orc_point* cloud, int n, const float q[3], int k, int* idx, float* d2) {
  int count = 0;
  for (int i = 0; i < k; i++) { idx[i] = -1; d2[i] = INFINITY; }
  for (int i = 0; i < n; i++) knn_insert(k, idx, d2, &count, dist2f(&cloud[i], q), i);
}

struct orc_kdtree {
  const orc_point* cloud;
  int n;
  int* perm;          /* point indices, leaf ranges are contiguous */
  int n_nodes;
  struct kd_node { int lo, hi, left, right, axis; float split; } * nodes;
};
#define KD_LEAF 12

static int kd_build(orc_kdtree* t, int lo, int hi) {
  int id = t->n_nodes++;
  struct kd_node* nd = &t->nodes[id];
  nd->lo = lo; nd->hi = hi; nd->left = nd->right = -1; nd->axis = 0; nd->split = 0.f;
  if (hi - lo <= KD_LEAF) return id;
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = lo; i < hi; i++) {
    const orc_point* p = &t->cloud[t->perm[i]];
    float v[3] = {p->x, p->y, p->z};
    for (int a = 0; a < 3; a++) { if (v[a] < mn[a]) mn[a] = v[a]; if (v[a] > mx[a]) mx[a] = v[a]; }
  }
  int axis = 0;
  if (mx[1] - mn[1] > mx[axis] - mn[axis]) axis = 1;
  if (mx[2] - mn[2] > mx[axis] - mn[axis]) axis = 2;
  if (!(mx[axis] > mn[axis])) return id;        /* all coincident: keep as leaf */
  float split = 0.5f * (mn[axis] + mx[axis]);
  int i = lo, j = hi - 1;
  while (i <= j) {
    const orc_point* p = &t->cloud[t->perm[i]];
    float v = axis == 0 ? p->x : (axis == 1 ? p->y : p->z);
    if (v < split) i++;
    else { int tmp = t->perm[i]; t->perm[i] = t->perm[j]; t->perm[j] = tmp; j--; }
  }
  if (i == lo || i == hi) return id;
  int l = kd_build(t, lo, i);
  int r = kd_build(t, i, hi);
  nd = &t->nodes[id];
  nd->left = l; nd->right = r; nd->axis = axis; nd->split = split;
  return id;
}

orc_kdtree* orc_kdtree_build(const orc_point* cloud, int n) {
  orc_kdtree* t = (orc_kdtree*)calloc(1, sizeof(orc_kdtree));
  t->cloud = cloud; t->n = n;
  t->perm = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; i++) t->perm[i] = i;
  t->nodes = (struct kd_node*)malloc(sizeof(struct kd_node) * (size_t)(2 * n + 2));
  t->n_nodes = 0;
  if (n > 0) kd_build(t, 0, n);
  return t;
}
void orc_kdtree_free(orc_kdtree* t) {
  if (!t) return;
  free(t->perm); free(t->nodes); free(t);
}

static void kd_search(const orc_kdtree* t, int id, const float q[3], int k, int* idx, float* d2, int* count) {
  const struct kd_node* nd = &t->nodes[id];
  if (nd->left < 0) {
    for (int i = nd->lo; i < nd->hi; i++) {
      int pi = t->perm[i];
      knn_insert(k, idx, d2, count, dist2f(&t->cloud[pi], q), pi);
    }
    return;
  }
  float qa = q[nd->axis];
  int nearc = qa < nd->split ? nd->left : nd->right;
  int farc = qa < nd->split ? nd->right : nd->left;
  kd_search(t, nearc, q, k, idx, d2, count);
  /* conservative pruning keeps the search exact incl. tie-breaking: only skip the far side when
     the plane distance clearly exceeds the current k-th distance */
  double pd = (double)qa - (double)nd->split;
  if (*count < k || pd * pd <= (double)d2[k - 1] * (1.0 + 1e-5)) kd_search(t, farc, q, k, idx, d2, count);
}

void orc_kdtree_knn(const orc_kdtree* t, const float q[3], int k, int* idx, float* d2) {
  int count = 0;
  for (int i = 0; i < k; i++) { idx[i] = -1; d2[i] = INFINITY; }
  if (t->n > 0) kd_search(t, 0, q, k, idx, d2, &count);
}

/* =============================================================================================
 * Robustified evaluation + Ceres trust-region LM
 * ============================================================================================= */

void orc_default_solver_options(orc_solver_options* o) {
  o->max_num_iterations = 6;              /* mapping_scan_matcher.cc:252, odometry_scan_matcher.cc:271 */
  o->huber_delta = 0.1;                   /* :77 / :67 */
  o->initial_trust_region_radius = 1e4;   /* Ceres Solver::Options defaults from here on */
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->max_consecutive_invalid_steps = 5;
}

/* ceres::HuberLoss::Evaluate: rho[0], rho[1] (rho[2] <= 0 always, so the Corrector reduces to
   scaling residual and Jacobian by sqrt(rho[1]); Ceres corrector.cc). */
static inline void huber(double a, double s, double* rho0, double* rho1) {
  const double b = a * a;
  if (s > b) {
    const double r = sqrt(s);
    *rho0 = 2.0 * a * r - b;
    double v = a / r;
    *rho1 = v > DBL_MIN ? v : DBL_MIN;
  } else {
    *rho0 = s; *rho1 = 1.0;
  }
}

double orc_evaluate(const orc_corr* corr, int n, const double pose[7], double huber_delta,
                    double* H, double* g) {
  double cost = 0.0;
  if (H) memset(H, 0, sizeof(double) * 36);
  if (g) memset(g, 0, sizeof(double) * 6);
  for (int i = 0; i < n; i++) {
    const orc_corr* c = &corr[i];
    double r[3], J[21];
    int nr;
    if (c->kind == ORC_KIND_EDGE) { orc_edge_factor(pose, c->p, c->C, c->N, r, H ? J : NULL); nr = 3; }
    else if (c->kind == ORC_KIND_PLANE) { orc_plane_factor(pose, c->p, c->C, c->N, r, H ? J : NULL); nr = 1; }
    else continue;
    double s = 0.0;
    for (int a = 0; a < nr; a++) s += r[a] * r[a];
    double rho0, rho1;
    huber(huber_delta, s, &rho0, &rho1);
    cost += 0.5 * rho0;
    if (!H) continue;
    const double sc = sqrt(rho1);   /* Corrector: residual_scaling_ = sqrt_rho1_, alpha = 0 */
    for (int a = 0; a < nr; a++) {
      double jr[6];
      for (int j = 0; j < 6; j++) jr[j] = sc * J[7 * a + j];   /* local Jacobian = first 6 cols */
      const double ra = sc * r[a];
      for (int j = 0; j < 6; j++) {
        g[j] += jr[j] * ra;
        for (int k = 0; k < 6; k++) H[6 * j + k] += jr[j] * jr[k];
      }
    }
  }
  return cost;
}

/* Cholesky solve of a 6x6 SPD system; returns 0 on failure. */
static int chol_solve6(const double A[36], const double b[6], double x[6]) {
  double L[36];
  memset(L, 0, sizeof(L));
  for (int i = 0; i < 6; i++) {
    for (int j = 0; j <= i; j++) {
      double s = A[6 * i + j];
      for (int k = 0; k < j; k++) s -= L[6 * i + k] * L[6 * j + k];
      if (i == j) {
        if (!(s > 0.0)) return 0;
        L[6 * i + i] = sqrt(s);
      } else {
        L[6 * i + j] = s / L[6 * j + j];
      }
    }
  }
  double y[6];
  for (int i = 0; i < 6; i++) {
    double s = b[i];
    for (int k = 0; k < i; k++) s -= L[6 * i + k] * y[k];
    y[i] = s / L[6 * i + i];
  }
  for (int i = 5; i >= 0; i--) {
    double s = y[i];
    for (int k = i + 1; k < 6; k++) s -= L[6 * k + i] * x[k];
    x[i] = s / L[6 * i + i];
  }
  for (int i = 0; i < 6; i++) if (!isfinite(x[i])) return 0;
  return 1;
}

static double norm7(const double x[7]) {
  double s = 0;
  for (int i = 0; i < 7; i++) s += x[i] * x[i];
  return sqrt(s);
}

/* |x - Plus(x, -g)|_inf  (TrustRegionMinimizer::EvaluateGradientAndJacobian) */
static double gradient_max_norm(const double x[7], const double g[6]) {
  double ng[6], xp[7], m = 0.0;
  for (int i = 0; i < 6; i++) ng[i] = -g[i];
  orc_pose_plus(x, ng, xp);
  for (int i = 0; i < 7; i++) { double d = fabs(x[i] - xp[i]); if (d > m) m = d; }
  return m;
}

/*
 * Restatement of ceres::Solve for one 7-D parameter block with PoseLocalParameterization and
 * default Solver::Options (Ceres 1.14: trust_region_minimizer.cc Minimize(),
 * levenberg_marquardt_strategy.cc ComputeStep()/StepAccepted()/StepRejected(),
 * trust_region_step_evaluator.cc).  Call sites: mapping_scan_matcher.cc:250-259,
 * odometry_scan_matcher.cc:269-274.
 */
void orc_ceres_solve(const orc_corr* corr, int n, double pose[7], const orc_solver_options* opt,
                     orc_solve_summary* sum) {
  orc_solve_summary local;
  if (!sum) sum = &local;
  memset(sum, 0, sizeof(*sum));
  int n_res = 0;
  for (int i = 0; i < n; i++) if (corr[i].kind != ORC_KIND_NONE) n_res++;
  if (n_res == 0) { sum->termination = 6; return; }   /* Ceres: nothing to optimise, x untouched */

  double x[7], H[36], g[6];
  memcpy(x, pose, sizeof(x));
  double cost = orc_evaluate(corr, n, x, opt->huber_delta, H, g);   /* IterationZero */
  sum->initial_cost = cost;
  double scale[6];                                                  /* jacobi_scaling (iteration 0 only) */
  for (int i = 0; i < 6; i++) scale[i] = 1.0 / (1.0 + sqrt(H[6 * i + i]));
  double gmax = gradient_max_norm(x, g);
  double x_norm = norm7(x);
  double radius = opt->initial_trust_region_radius, decrease_factor = 2.0;
  int reuse_diagonal = 0;
  double diagonal[6];
  int step_is_successful = 1, iteration = 0, invalid = 0;
  sum->termination = 0;

  for (;;) {
    /* FinalizeIterationAndCheckIfMinimizerCanContinue */
    if (iteration >= opt->max_num_iterations) { sum->termination = 0; break; }
    if (step_is_successful && gmax <= opt->gradient_tolerance) { sum->termination = 1; break; }
    if (radius < opt->min_trust_region_radius) { sum->termination = 4; break; }
    iteration++;
    const int ti = iteration - 1;

    /* ComputeTrustRegionStep on the column-scaled system */
    double Hs[36], gs[6];
    for (int i = 0; i < 6; i++) {
      gs[i] = g[i] * scale[i];
      for (int j = 0; j < 6; j++) Hs[6 * i + j] = H[6 * i + j] * scale[i] * scale[j];
    }
    if (!reuse_diagonal) {
      for (int i = 0; i < 6; i++) {
        double d = Hs[6 * i + i];
        d = d > opt->min_lm_diagonal ? d : opt->min_lm_diagonal;
        d = d < opt->max_lm_diagonal ? d : opt->max_lm_diagonal;
        diagonal[i] = d;
      }
    }
    double A[36], y[6], step[6];
    memcpy(A, Hs, sizeof(A));
    for (int i = 0; i < 6; i++) {
      double lm = sqrt(diagonal[i] / radius);    /* lm_diagonal_ */
      A[6 * i + i] += lm * lm;                   /* D^T D of the augmented system */
    }
    int ok = chol_solve6(A, gs, y);
    reuse_diagonal = 1;
    double model_cost_change = 0.0;
    if (ok) {
      for (int i = 0; i < 6; i++) step[i] = -y[i];
      /* -(J step)^T (r + J step / 2) = -g^T step - step^T H step / 2 */
      double gts = 0.0, shs = 0.0;
      for (int i = 0; i < 6; i++) {
        gts += gs[i] * step[i];
        for (int j = 0; j < 6; j++) shs += step[i] * Hs[6 * i + j] * step[j];
      }
      model_cost_change = -gts - 0.5 * shs;
    }
    if (ti < ORC_MAX_TRACE) sum->trace_radius[ti] = radius;
    if (!ok || !(model_cost_change > 0.0)) {
      /* HandleInvalidStep */
      if (++invalid >= opt->max_consecutive_invalid_steps) { sum->termination = 5; break; }
      radius *= 0.5;                              /* LevenbergMarquardtStrategy::StepIsInvalid */
      reuse_diagonal = 1;
      step_is_successful = 0;
      if (ti < ORC_MAX_TRACE) { sum->trace_accepted[ti] = -1; sum->trace_cost[ti] = cost; }
      continue;
    }
    invalid = 0;
    double delta[6], cand[7];
    for (int i = 0; i < 6; i++) delta[i] = step[i] * scale[i];
    orc_pose_plus(x, delta, cand);
    double cand_cost = orc_evaluate(corr, n, cand, opt->huber_delta, NULL, NULL);
    double sn = 0.0;
    for (int i = 0; i < 7; i++) sn += (x[i] - cand[i]) * (x[i] - cand[i]);
    sn = sqrt(sn);
    if (ti < ORC_MAX_TRACE) { sum->trace_cost[ti] = cand_cost; sum->trace_step_norm[ti] = sn; }
    /* ParameterToleranceReached */
    if (sn <= opt->parameter_tolerance * (x_norm + opt->parameter_tolerance)) { sum->termination = 2; break; }
    /* FunctionToleranceReached */
    double cost_change = cost - cand_cost;
    if (fabs(cost_change) <= opt->function_tolerance * cost) { sum->termination = 3; break; }
    /* IsStepSuccessful (monotonic step evaluator) */
    double rel = cost_change / model_cost_change;
    if (ti < ORC_MAX_TRACE) sum->trace_rel_decrease[ti] = rel;
    if (rel > opt->min_relative_decrease) {
      memcpy(x, cand, sizeof(x));
      x_norm = norm7(x);
      cost = orc_evaluate(corr, n, x, opt->huber_delta, H, g);
      gmax = gradient_max_norm(x, g);
      double t = 2.0 * rel - 1.0;
      double f = 1.0 - t * t * t;
      radius = radius / (f > 1.0 / 3.0 ? f : 1.0 / 3.0);
      if (radius > opt->max_trust_region_radius) radius = opt->max_trust_region_radius;
      decrease_factor = 2.0;
      reuse_diagonal = 0;
      step_is_successful = 1;
      sum->successful_steps++;
      if (ti < ORC_MAX_TRACE) sum->trace_accepted[ti] = 1;
    } else {
      radius = radius / decrease_factor;
      decrease_factor *= 2.0;
      reuse_diagonal = 1;
      step_is_successful = 0;
      if (ti < ORC_MAX_TRACE) sum->trace_accepted[ti] = 0;
    }
  }
  sum->iterations = iteration;
  sum->final_cost = cost;
  memcpy(pose, x, sizeof(x));
}

/* =============================================================================================
 * Stage C: scan -> map
 * ============================================================================================= */

static void assoc_one(const orc_point* map, int m, const orc_kdtree* tree, const orc_point* f,
                      const double pose[7], int is_edge, orc_corr* out) {
  float in[3] = {f->x, f->y, f->z}, sel[3];
  orc_transform_point(pose, in, sel);                 /* :123 / :193 */
  int idx[5]; float d2[5];
  if (tree) orc_kdtree_knn(tree, sel, 5, idx, d2);    /* :125 / :195 */
  else orc_knn_brute(map, m, sel, 5, idx, d2);
  out->kind = ORC_KIND_NONE;
  out->p[0] = in[0]; out->p[1] = in[1]; out->p[2] = in[2];   /* curr_point: untransformed :146,:221 */
  memset(out->C, 0, sizeof(out->C)); memset(out->N, 0, sizeof(out->N));
  if (idx[4] < 0) return;
  if (!((double)d2[4] < 1.0)) return;                 /* :128 / :198 */
  float nbr[5][3];
  for (int j = 0; j < 5; j++) { nbr[j][0] = map[idx[j]].x; nbr[j][1] = map[idx[j]].y; nbr[j][2] = map[idx[j]].z; }
  if (is_edge) { if (orc_edge_fit(nbr, 3.0, out->C, out->N)) out->kind = ORC_KIND_EDGE; }
  else { if (orc_plane_fit(nbr, 0.2, out->C, out->N)) out->kind = ORC_KIND_PLANE; }
}

static void associate_scan2map_trees(const orc_point* map_corner, int mc, const orc_kdtree* tc,
                                     const orc_point* map_surf, int ms, const orc_kdtree* ts,
                                     const orc_point* corner, int nc, const orc_point* surf, int ns,
                                     const double pose[7], orc_corr* out) {
  for (int i = 0; i < nc; i++) assoc_one(map_corner, mc, tc, &corner[i], pose, 1, &out[i]);       /* :109-176 */
  for (int i = 0; i < ns; i++) assoc_one(map_surf, ms, ts, &surf[i], pose, 0, &out[nc + i]);      /* :179-246 */
}

void orc_associate_scan2map(const orc_point* map_corner, int mc, const orc_point* map_surf, int ms,
                            const orc_point* corner, int nc, const orc_point* surf, int ns,
                            const double pose[7], int use_kdtree, orc_corr* out) {
  orc_kdtree *tc = NULL, *ts = NULL;
  if (use_kdtree) { tc = orc_kdtree_build(map_corner, mc); ts = orc_kdtree_build(map_surf, ms); }
  associate_scan2map_trees(map_corner, mc, tc, map_surf, ms, ts, corner, nc, surf, ns, pose, out);
  orc_kdtree_free(tc); orc_kdtree_free(ts);
}

static double now_s(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

/* stage_s (optional): seconds added to {Data association, Solver time}, the reference's LOG_STEP_TIME stages
 * (mapping_scan_matcher.cc:248, :264) */
static int match_scan2map_trees_timed(const orc_point* map_corner, int mc, const orc_kdtree* tc,
                                      const orc_point* map_surf, int ms, const orc_kdtree* ts,
                                      const orc_point* corner, int nc, const orc_point* surf, int ns,
                                      double pose[7], orc_match_info* info, double* stage_s) {
  orc_solver_options opt;
  orc_default_solver_options(&opt);
  orc_corr* corr = (orc_corr*)malloc(sizeof(orc_corr) * (size_t)(nc + ns + 1));
  for (int it = 0; it < 2; it++) {                                   /* kOptimalNum, :75 */
    const double t0 = stage_s ? now_s() : 0.0;
    associate_scan2map_trees(map_corner, mc, tc, map_surf, ms, ts, corner, nc, surf, ns, pose, corr);
    const double t1 = stage_s ? now_s() : 0.0;
    orc_solve_summary s;
    orc_ceres_solve(corr, nc + ns, pose, &opt, &s);                  /* :259, write-back :271 */
    if (stage_s) { stage_s[0] += t1 - t0; stage_s[1] += now_s() - t1; }
    if (info) {
      int ne = 0, np = 0;
      for (int i = 0; i < nc + ns; i++) { ne += corr[i].kind == ORC_KIND_EDGE; np += corr[i].kind == ORC_KIND_PLANE; }
      info->n_edge[it] = ne; info->n_plane[it] = np;
      info->lm_iterations[it] = s.iterations; info->lm_successful[it] = s.successful_steps;
      info->initial_cost[it] = s.initial_cost; info->final_cost[it] = s.final_cost;
    }
  }
  free(corr);
  return 0;
}

static int match_scan2map_trees(const orc_point* map_corner, int mc, const orc_kdtree* tc,
                                const orc_point* map_surf, int ms, const orc_kdtree* ts,
                                const orc_point* corner, int nc, const orc_point* surf, int ns,
                                double pose[7], orc_match_info* info) {
  return match_scan2map_trees_timed(map_corner, mc, tc, map_surf, ms, ts, corner, nc, surf, ns, pose, info, NULL);
}

/* Single-threaded batch with the reference's stage clocks: stage_seconds = {build tree (:73), Data association (:248),
 * Solver time (:264)} summed over the scans; the kd-trees are rebuilt for every registration like the reference does. */
void orc_match_scan2map_batch_timed(const orc_point* map_corner, int mc, const orc_point* map_surf, int ms,
                                    int n_scans, const orc_point* corner, const int* corner_off,
                                    const orc_point* surf, const int* surf_off,
                                    double* poses, int* status, double stage_seconds[3]) {
  stage_seconds[0] = stage_seconds[1] = stage_seconds[2] = 0.0;
  if (mc < 5 || ms < 5) { for (int b = 0; b < n_scans; b++) if (status) status[b] = 2; return; }
  for (int b = 0; b < n_scans; b++) {
    const double t0 = now_s();
    orc_kdtree* tc = orc_kdtree_build(map_corner, mc);
    orc_kdtree* ts = orc_kdtree_build(map_surf, ms);
    stage_seconds[0] += now_s() - t0;
    int rc = match_scan2map_trees_timed(map_corner, mc, tc, map_surf, ms, ts,
                                        corner + corner_off[b], corner_off[b + 1] - corner_off[b],
                                        surf + surf_off[b], surf_off[b + 1] - surf_off[b],
                                        poses + 7 * b, NULL, stage_seconds + 1);
    if (status) status[b] = rc;
    orc_kdtree_free(tc); orc_kdtree_free(ts);
  }
}

int orc_match_scan2map(const orc_point* map_corner, int mc, const orc_point* map_surf, int ms,
                       const orc_point* corner, int nc, const orc_point* surf, int ns,
                       double pose[7], int use_kdtree, orc_match_info* info) {
  if (mc < 5 || ms < 5) return 2;
  orc_kdtree *tc = NULL, *ts = NULL;
  if (use_kdtree) { tc = orc_kdtree_build(map_corner, mc); ts = orc_kdtree_build(map_surf, ms); }   /* :66-73 */
  int rc = match_scan2map_trees(map_corner, mc, tc, map_surf, ms, ts, corner, nc, surf, ns, pose, info);
  orc_kdtree_free(tc); orc_kdtree_free(ts);
  return rc;
}

void orc_match_scan2map_batch(const orc_point* map_corner, int mc, const orc_point* map_surf, int ms,
                              int n_scans, const orc_point* corner, const int* corner_off,
                              const orc_point* surf, const int* surf_off,
                              double* poses, int* status, int threads, int rebuild_tree_per_scan) {
  orc_kdtree *tc = NULL, *ts = NULL;
  if (mc < 5 || ms < 5) { for (int b = 0; b < n_scans; b++) if (status) status[b] = 2; return; }
  if (!rebuild_tree_per_scan) { tc = orc_kdtree_build(map_corner, mc); ts = orc_kdtree_build(map_surf, ms); }
#ifdef _OPENMP
  if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
#endif
  for (int b = 0; b < n_scans; b++) {
    orc_kdtree *ltc = tc, *lts = ts;
    if (rebuild_tree_per_scan) { ltc = orc_kdtree_build(map_corner, mc); lts = orc_kdtree_build(map_surf, ms); }
    int rc = match_scan2map_trees(map_corner, mc, ltc, map_surf, ms, lts,
                                  corner + corner_off[b], corner_off[b + 1] - corner_off[b],
                                  surf + surf_off[b], surf_off[b + 1] - surf_off[b],
                                  poses + 7 * b, NULL);
    if (status) status[b] = rc;
    if (rebuild_tree_per_scan) { orc_kdtree_free(ltc); orc_kdtree_free(lts); }
  }
  (void)threads;
  orc_kdtree_free(tc); orc_kdtree_free(ts);
}

/* The batch against kd-trees the caller built beforehand (bench.py: the GPU step indexes the map once per batch outside
   nothing, but its index build is 3 % of the step; timing the CPU side with the serial tree build inside the region
   under-states the CPU on many cores). */
void orc_match_scan2map_batch_trees(const orc_point* map_corner, int mc, const orc_kdtree* tc, const orc_point* map_surf, int ms,
                                    const orc_kdtree* ts, int n_scans, const orc_point* corner, const int* corner_off,
                                    const orc_point* surf, const int* surf_off, double* poses, int* status, int threads) {
  if (mc < 5 || ms < 5 || !tc || !ts) { for (int b = 0; b < n_scans; b++) if (status) status[b] = 2; return; }
#ifdef _OPENMP
  if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
#endif
  for (int b = 0; b < n_scans; b++) {
    int rc = match_scan2map_trees(map_corner, mc, tc, map_surf, ms, ts, corner + corner_off[b], corner_off[b + 1] - corner_off[b],
                                  surf + surf_off[b], surf_off[b + 1] - surf_off[b], poses + 7 * b, NULL);
    if (status) status[b] = rc;
  }
  (void)threads;
}

/* ---- deskew variant ------------------------------------------------------------------------- */
/*
 * lidar_factor.cc:46-100: residual(Pi,Qi) = N (x|.) (Qi*(dq*p + dp) + Vi*dt - G*dt^2/2 + Pi - C)
 * with the velocity block constant (mapping_scan_matcher.cc:94).  That is the plain factor
 * evaluated at p' = dq*p + dp and C' = C - (Vi*dt - G*dt^2/2); the pose Jacobian uses
 * skew(p') exactly as lidar_factor.cc:61,89 do.  The kNN query point follows :120 / :190.
 */
static void deskew_query(const double pose[7], const orc_point* f, const double dq[4], const double dp[3],
                         const double V[3], const double G[3], float sel[3]) {
  /* pose * Rigid3d{ q^-1 * (Vi*dt - 0.5*G*dt*dt) + delta_p, delta_q }  applied to pointOri */
  double dt = (double)f->t;
  double w[3] = {V[0] * dt - 0.5 * G[0] * dt * dt, V[1] * dt - 0.5 * G[1] * dt * dt, V[2] * dt - 0.5 * G[2] * dt * dt};
  double qc[4] = {-pose[3], -pose[4], -pose[5], pose[6]}, wl[3];
  orc_quat_rotate(qc, w, wl);
  double inner[7] = {wl[0] + dp[0], wl[1] + dp[1], wl[2] + dp[2], dq[0], dq[1], dq[2], dq[3]};
  double full[7];
  orc_pose_compose(pose, inner, full);
  float in[3] = {f->x, f->y, f->z};
  orc_transform_point(full, in, sel);
}

int orc_match_scan2map_deskew(const orc_point* map_corner, int mc, const orc_point* map_surf, int ms,
                              const orc_point* corner, int nc, const orc_point* surf, int ns,
                              const double* corner_dq, const double* corner_dp,
                              const double* surf_dq, const double* surf_dp,
                              const double V[3], const double G[3],
                              double pose[7], orc_match_info* info) {
  if (mc < 5 || ms < 5) return 2;
  orc_solver_options opt;
  orc_default_solver_options(&opt);
  orc_corr* corr = (orc_corr*)malloc(sizeof(orc_corr) * (size_t)(nc + ns + 1));
  for (int it = 0; it < 2; it++) {
    for (int i = 0; i < nc + ns; i++) {
      const int is_edge = i < nc;
      const orc_point* f = is_edge ? &corner[i] : &surf[i - nc];
      const double* dq = is_edge ? corner_dq + 4 * i : surf_dq + 4 * (i - nc);
      const double* dp = is_edge ? corner_dp + 3 * i : surf_dp + 3 * (i - nc);
      const orc_point* map = is_edge ? map_corner : map_surf;
      const int m = is_edge ? mc : ms;
      orc_corr* out = &corr[i];
      float sel[3];
      deskew_query(pose, f, dq, dp, V, G, sel);
      int idx[5]; float d2[5];
      orc_knn_brute(map, m, sel, 5, idx, d2);
      out->kind = ORC_KIND_NONE;
      double p[3] = {(double)f->x, (double)f->y, (double)f->z}, pr[3];
      orc_quat_rotate(dq, p, pr);
      for (int a = 0; a < 3; a++) out->p[a] = pr[a] + dp[a];
      memset(out->C, 0, sizeof(out->C)); memset(out->N, 0, sizeof(out->N));
      if (idx[4] < 0 || !((double)d2[4] < 1.0)) continue;
      float nbr[5][3];
      for (int j = 0; j < 5; j++) { nbr[j][0] = map[idx[j]].x; nbr[j][1] = map[idx[j]].y; nbr[j][2] = map[idx[j]].z; }
      int okf = is_edge ? orc_edge_fit(nbr, 3.0, out->C, out->N) : orc_plane_fit(nbr, 0.2, out->C, out->N);
      if (!okf) continue;
      double dt = (double)f->t;
      for (int a = 0; a < 3; a++) out->C[a] -= V[a] * dt - 0.5 * G[a] * dt * dt;
      out->kind = is_edge ? ORC_KIND_EDGE : ORC_KIND_PLANE;
    }
    orc_solve_summary s;
    orc_ceres_solve(corr, nc + ns, pose, &opt, &s);
    if (info) {
      int ne = 0, np = 0;
      for (int i = 0; i < nc + ns; i++) { ne += corr[i].kind == ORC_KIND_EDGE; np += corr[i].kind == ORC_KIND_PLANE; }
      info->n_edge[it] = ne; info->n_plane[it] = np;
      info->lm_iterations[it] = s.iterations; info->lm_successful[it] = s.successful_steps;
      info->initial_cost[it] = s.initial_cost; info->final_cost[it] = s.final_cost;
    }
  }
  free(corr);
  return 0;
}

/* =============================================================================================
 * Stage B: scan -> scan   (odometry_scan_matcher.cc:43-285)
 * ============================================================================================= */

/* odometry_scan_matcher.cc:102-108 etc.: float subtraction/multiplication/sum, widened at the end */
static inline double odo_dist(const orc_point* a, const float s[3]) {
  float dx = a->x - s[0], dy = a->y - s[1], dz = a->z - s[2];
  float r = dx * dx + dy * dy + dz * dz;
  return (double)r;
}

static void assoc_odo_edge(const orc_point* last, const uint16_t* ring, int n, const orc_kdtree* tree,
                           const orc_point* f, const double pose[7], orc_corr* out) {
  const double kDist = 25.0, kNear = 2.5;
  float in[3] = {f->x, f->y, f->z}, sel[3];
  orc_transform_point(pose, in, sel);                   /* TransformToStart, s = 1 (:21-33) */
  out->kind = ORC_KIND_NONE;
  out->p[0] = in[0]; out->p[1] = in[1]; out->p[2] = in[2];
  memset(out->C, 0, sizeof(out->C)); memset(out->N, 0, sizeof(out->N));
  if (n <= 0) return;
  int ci; float cd;
  orc_kdtree_knn(tree, sel, 1, &ci, &cd);               /* :84 */
  int closest = -1, min2 = -1;
  if ((double)cd < kDist) {                             /* :87 */
    closest = ci;
    int id = ring[closest];
    double best = kDist;
    for (int j = closest + 1; j < n; ++j) {             /* :93-115 */
      if (ring[j] <= id) continue;
      if ((double)ring[j] > id + kNear) break;
      double d = odo_dist(&last[j], sel);
      if (d < best) { best = d; min2 = j; }
    }
    for (int j = closest - 1; j >= 0; --j) {            /* :118-140 */
      if (ring[j] >= id) continue;
      if ((double)ring[j] < id - kNear) break;
      double d = odo_dist(&last[j], sel);
      if (d < best) { best = d; min2 = j; }
    }
  }
  if (min2 >= 0) {                                      /* :143-162 */
    double a[3] = {last[closest].x, last[closest].y, last[closest].z};
    double b[3] = {last[min2].x, last[min2].y, last[min2].z};
    double ab[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
    normalized3(ab, out->N);
    out->C[0] = a[0]; out->C[1] = a[1]; out->C[2] = a[2];
    out->kind = ORC_KIND_EDGE;
  }
}

static void assoc_odo_plane(const orc_point* last, const uint16_t* ring, int n, const orc_kdtree* tree,
                            const orc_point* f, const double pose[7], orc_corr* out) {
  const double kDist = 25.0, kNear = 2.5;
  float in[3] = {f->x, f->y, f->z}, sel[3];
  orc_transform_point(pose, in, sel);
  out->kind = ORC_KIND_NONE;
  out->p[0] = in[0]; out->p[1] = in[1]; out->p[2] = in[2];
  memset(out->C, 0, sizeof(out->C)); memset(out->N, 0, sizeof(out->N));
  if (n <= 0) return;
  int ci; float cd;
  orc_kdtree_knn(tree, sel, 1, &ci, &cd);               /* :169 */
  int closest = -1, min2 = -1, min3 = -1;
  if ((double)cd < kDist) {                             /* :173 */
    closest = ci;
    int id = ring[closest];
    double best2 = kDist, best3 = kDist;
    for (int j = closest + 1; j < n; ++j) {             /* :183-207 */
      if ((double)ring[j] > id + kNear) break;
      double d = odo_dist(&last[j], sel);
      if (ring[j] <= id && d < best2) { best2 = d; min2 = j; }
      else if (ring[j] > id && d < best3) { best3 = d; min3 = j; }
    }
    for (int j = closest - 1; j >= 0; --j) {            /* :210-232 */
      if ((double)ring[j] < id - kNear) break;
      double d = odo_dist(&last[j], sel);
      if (ring[j] >= id && d < best2) { best2 = d; min2 = j; }
      else if (ring[j] < id && d < best3) { best3 = d; min3 = j; }
    }
  }
  if (min2 >= 0 && min3 >= 0) {                         /* :234-256; lidar_factor.h:70-78 */
    double a[3] = {last[closest].x, last[closest].y, last[closest].z};
    double b[3] = {last[min2].x, last[min2].y, last[min2].z};
    double c[3] = {last[min3].x, last[min3].y, last[min3].z};
    double ab[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
    double ac[3] = {a[0] - c[0], a[1] - c[1], a[2] - c[2]};
    double nn[3];
    cross3(ab, ac, nn);
    normalized3(nn, out->N);
    for (int k = 0; k < 3; k++) out->C[k] = (a[k] + b[k] + c[k]) / 3;
    out->kind = ORC_KIND_PLANE;
  }
}

static void associate_scan2scan_trees(const orc_point* last_ls, const uint16_t* ring_ls, int n_ls, const orc_kdtree* t_ls,
                                      const orc_point* last_lf, const uint16_t* ring_lf, int n_lf, const orc_kdtree* t_lf,
                                      const orc_point* sharp, int n_sharp, const orc_point* flat, int n_flat,
                                      const double pose[7], orc_corr* out) {
  for (int i = 0; i < n_sharp; i++) assoc_odo_edge(last_ls, ring_ls, n_ls, t_ls, &sharp[i], pose, &out[i]);
  for (int i = 0; i < n_flat; i++) assoc_odo_plane(last_lf, ring_lf, n_lf, t_lf, &flat[i], pose, &out[n_sharp + i]);
}

void orc_associate_scan2scan(const orc_point* last_ls, const uint16_t* last_ls_ring, int n_last_ls,
                             const orc_point* last_lf, const uint16_t* last_lf_ring, int n_last_lf,
                             const orc_point* sharp, int n_sharp, const orc_point* flat, int n_flat,
                             const double pose[7], orc_corr* out) {
  orc_kdtree* t1 = orc_kdtree_build(last_ls, n_last_ls);
  orc_kdtree* t2 = orc_kdtree_build(last_lf, n_last_lf);
  associate_scan2scan_trees(last_ls, last_ls_ring, n_last_ls, t1, last_lf, last_lf_ring, n_last_lf, t2,
                            sharp, n_sharp, flat, n_flat, pose, out);
  orc_kdtree_free(t1); orc_kdtree_free(t2);
}

int orc_match_scan2scan(const orc_point* last_ls, const uint16_t* last_ls_ring, int n_last_ls,
                        const orc_point* last_lf, const uint16_t* last_lf_ring, int n_last_lf,
                        const orc_point* sharp, int n_sharp, const orc_point* flat, int n_flat,
                        double pose[7], orc_match_info* info) {
  orc_solver_options opt;
  orc_default_solver_options(&opt);
  orc_kdtree* t1 = orc_kdtree_build(last_ls, n_last_ls);      /* :57-61 */
  orc_kdtree* t2 = orc_kdtree_build(last_lf, n_last_lf);
  orc_corr* corr = (orc_corr*)malloc(sizeof(orc_corr) * (size_t)(n_sharp + n_flat + 1));
  int rc = 0;
  for (int it = 0; it < 2; it++) {                            /* :64 */
    associate_scan2scan_trees(last_ls, last_ls_ring, n_last_ls, t1, last_lf, last_lf_ring, n_last_lf, t2,
                              sharp, n_sharp, flat, n_flat, pose, corr);
    int ne = 0, np = 0;
    for (int i = 0; i < n_sharp + n_flat; i++) { ne += corr[i].kind == ORC_KIND_EDGE; np += corr[i].kind == ORC_KIND_PLANE; }
    if (info) { info->n_edge[it] = ne; info->n_plane[it] = np; }
    if (ne + np < 10) { rc = 1; break; }                      /* :262-267 */
    orc_solve_summary s;
    orc_ceres_solve(corr, n_sharp + n_flat, pose, &opt, &s);  /* :274, :280 */
    if (info) {
      info->lm_iterations[it] = s.iterations; info->lm_successful[it] = s.successful_steps;
      info->initial_cost[it] = s.initial_cost; info->final_cost[it] = s.final_cost;
    }
  }
  free(corr);
  orc_kdtree_free(t1); orc_kdtree_free(t2);
  return rc;
}

/* =============================================================================================
 * Stage A: feature extraction   (msf_loam_node.cc:160-378)
 * ============================================================================================= */

#define ORC_MAX_RINGS 128   /* kMaxScanNum, :79 */

typedef struct { float c; int i; } curv_key;
static int curv_cmp(const void* a, const void* b) {
  const curv_key* x = (const curv_key*)a; const curv_key* y = (const curv_key*)b;
  if (x->c < y->c) return -1;
  if (x->c > y->c) return 1;
  const int o = (x->i > y->i) - (x->i < y->i);     /* tie-break: ascending index (std::sort is unstable) */
  return g_sort_ties_reverse ? -o : o;
}

static inline double gap2(const orc_point* a, const orc_point* b) {
  /* (a.getVector3fMap() - b.getVector3fMap()).squaredNorm() in f32, then compared with the
     double literal 0.05 (:293,300,326,332) */
  float dx = a->x - b->x, dy = a->y - b->y, dz = a->z - b->z;
  float r = dx * dx + dy * dy + dz * dz;
  return (double)r;
}

int orc_extract_features(const orc_point* pts_in, const uint16_t* ring_in, int n_in,
                         double min_range, const double* extrinsic,
                         orc_point* cloud, uint16_t* cring, float* curv, uint8_t* label,
                         int* sharp_idx, int* less_sharp_idx, int* flat_idx, int* less_flat_idx,
                         int counts[5]) {
  for (int i = 0; i < 5; i++) counts[i] = 0;
  /* 1. RemoveInvalidPointsFromCloud (:85-111) */
  orc_point* valid = (orc_point*)malloc(sizeof(orc_point) * (size_t)(n_in + 1));
  uint16_t* vring = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)(n_in + 1));
  int nv = 0;
  for (int i = 0; i < n_in; i++) {
    const orc_point* p = &pts_in[i];
    float nrm = sqrtf(p->x * p->x + p->y * p->y + p->z * p->z);
    if ((double)nrm < min_range || !isfinite(p->x) || !isfinite(p->y) || !isfinite(p->z)) continue;
    valid[nv] = *p; vring[nv] = ring_in[i]; nv++;
  }
  if (nv == 0) { free(valid); free(vring); return 3; }
  /* 2. ComputeRelaTimeForEachPoint (:128-156) + ring buckets */
  for (int i = 0; i < nv; i++) if (vring[i] >= ORC_MAX_RINGS) { free(valid); free(vring); return 5; }  /* CHECK_LT :136 */
  int ring_count[ORC_MAX_RINGS] = {0}, ring_off[ORC_MAX_RINGS + 1];
  for (int i = 0; i < nv; i++) ring_count[vring[i]]++;
  int valid_scan_num = 0;
  for (int r = ORC_MAX_RINGS; r > 0; --r) if (ring_count[r - 1] > 0) { valid_scan_num = r; break; }   /* :178-184 */
  ring_off[0] = 0;
  for (int r = 0; r < ORC_MAX_RINGS; r++) ring_off[r + 1] = ring_off[r] + ring_count[r];
  int cursor[ORC_MAX_RINGS];
  for (int r = 0; r < ORC_MAX_RINGS; r++) cursor[r] = ring_off[r];
  double last_rel[ORC_MAX_RINGS];
  for (int r = 0; r < ORC_MAX_RINGS; r++) last_rel[r] = -1;
  const double two_pi = 2 * M_PI;
  double start_ori = g_atan2_float ? -(double)atan2f(valid[0].y, valid[0].x) : -atan2((double)valid[0].y, (double)valid[0].x);       /* :131 */
  for (int i = 0; i < nv; i++) {
    orc_point p = valid[i];
    int r = vring[i];
    double ori = g_atan2_float ? -(double)atan2f(p.y, p.x) : -atan2((double)p.y, (double)p.x);                          /* :139 */
    double rel = fmod(ori - start_ori + two_pi, two_pi);                    /* :142 */
    if (rel < last_rel[r]) rel += two_pi;                                   /* :145-148 */
    last_rel[r] = rel;
    double rela_time = rel / two_pi * 0.1;                                  /* :151, kScanPeriod */
    p.t = (float)rela_time;                                                 /* :152-153 (time and intensity) */
    int dst = cursor[r]++;
    cloud[dst] = p; cring[dst] = (uint16_t)r;                               /* per-ring push_back + concat :188-195 */
  }
  free(valid); free(vring);
  const int N = nv;
  counts[0] = N;
  for (int i = 0; i < N; i++) { curv[i] = 0.f; label[i] = 0; }
  if (N < 11) return 0;   /* the reference's `size() - 5` loop bound underflows here; nothing to extract */
  /* 3. curvature (:213-240): f32 left-to-right sums, f64 squares, f32 store */
  for (int i = 5; i < N - 5; i++) {
    float dx = cloud[i - 5].x + cloud[i - 4].x + cloud[i - 3].x + cloud[i - 2].x + cloud[i - 1].x - 10 * cloud[i].x +
               cloud[i + 1].x + cloud[i + 2].x + cloud[i + 3].x + cloud[i + 4].x + cloud[i + 5].x;
    float dy = cloud[i - 5].y + cloud[i - 4].y + cloud[i - 3].y + cloud[i - 2].y + cloud[i - 1].y - 10 * cloud[i].y +
               cloud[i + 1].y + cloud[i + 2].y + cloud[i + 3].y + cloud[i + 4].y + cloud[i + 5].y;
    float dz = cloud[i - 5].z + cloud[i - 4].z + cloud[i - 3].z + cloud[i - 2].z + cloud[i - 1].z - 10 * cloud[i].z +
               cloud[i + 1].z + cloud[i + 2].z + cloud[i + 3].z + cloud[i + 4].z + cloud[i + 5].z;
    double X = dx, Y = dy, Z = dz;
    curv[i] = (float)(X * X + Y * Y + Z * Z);
  }
  uint8_t* picked = (uint8_t*)calloc((size_t)N, 1);
  curv_key* keys = (curv_key*)malloc(sizeof(curv_key) * (size_t)N);
  int n_sharp = 0, n_ls = 0, n_flat = 0, n_lf = 0;
  /* 4. per ring, per sector (:251-350) */
  for (int r = 0; r < valid_scan_num; r++) {
    const int start = ring_off[r] + 5, end = ring_off[r + 1] - 6;           /* :192-194 */
    if (end - start < 6) continue;                                          /* :252 */
    for (int j = 0; j < 6; j++) {
      const int sp = start + (end - start) * j / 6;                         /* :256-259 */
      const int ep = start + (end - start) * (j + 1) / 6 - 1;
      const int cnt = ep - sp + 1;
      if (cnt <= 0) continue;
      for (int k = 0; k < cnt; k++) { keys[k].c = curv[sp + k]; keys[k].i = sp + k; }
      qsort(keys, (size_t)cnt, sizeof(curv_key), curv_cmp);                 /* :263-267 */
      int largest = 0;
      for (int k = cnt - 1; k >= 0; k--) {                                  /* :272-305 */
        int ind = keys[k].i;
        if (!picked[ind] && (double)curv[ind] > 0.1) {            /* float vs double literal, :275 */
          largest++;
          if (largest <= 2) {
            label[ind] = 1; sharp_idx[n_sharp++] = ind; less_sharp_idx[n_ls++] = ind;
          } else if (largest <= 20) {
            label[ind] = 2; less_sharp_idx[n_ls++] = ind;
          } else break;
          picked[ind] = 1;
          for (int l = 1; l <= 5; l++) {
            if (gap2(&cloud[ind + l], &cloud[ind + l - 1]) > 0.05) break;
            picked[ind + l] = 1; label[ind + l] = 2;
          }
          for (int l = -1; l >= -5; l--) {
            if (gap2(&cloud[ind + l], &cloud[ind + l + 1]) > 0.05) break;
            picked[ind + l] = 1; label[ind + l] = 2;
          }
        }
      }
      int smallest = 0;
      for (int k = 0; k < cnt; k++) {                                       /* :309-336 */
        int ind = keys[k].i;
        if (!picked[ind] && (double)curv[ind] < 0.1) {            /* :312 */
          label[ind] = 3; flat_idx[n_flat++] = ind;
          smallest++;
          if (smallest >= 4) break;
          picked[ind] = 1;
          for (int l = 1; l <= 5; l++) {
            if (gap2(&cloud[ind + l], &cloud[ind + l - 1]) > 0.05) break;
            picked[ind + l] = 1;
          }
          for (int l = -1; l >= -5; l--) {
            if (gap2(&cloud[ind + l], &cloud[ind + l + 1]) > 0.05) break;
            picked[ind + l] = 1;
          }
        }
      }
      for (int k = sp; k <= ep; k++)                                        /* :339-344 */
        if (label[k] == 3 || label[k] == 0) less_flat_idx[n_lf++] = k;
    }
    /* VoxelGridWrapper (:119-125, :347-350) copies cloud_in at PCLBase::getIndices() = all input
       points: an identity copy. */
  }
  free(picked); free(keys);
  counts[1] = n_sharp; counts[2] = n_ls; counts[3] = n_flat; counts[4] = n_lf;
  /* 5. TransformPointCloudInPlace x5 (:367-371).  All five clouds are gathers of `cloud`, so one
     in-place transform of the full cloud is equivalent. */
  if (extrinsic) {
    for (int i = 0; i < N; i++) {
      float in[3] = {cloud[i].x, cloud[i].y, cloud[i].z}, o[3];
      orc_transform_point(extrinsic, in, o);
      cloud[i].x = o[0]; cloud[i].y = o[1]; cloud[i].z = o[2];
    }
  }
  return 0;
}

/* =============================================================================================
 * pcl::VoxelGrid<PointXYZI> (laser_mapping.cc:264-270), [3P-recall PCL 1.10 voxel_grid.hpp]
 * ============================================================================================= */

typedef struct { int64_t cell; int idx; } vox_key;
static int vox_cmp(const void* a, const void* b) {
  const vox_key* x = (const vox_key*)a; const vox_key* y = (const vox_key*)b;
  if (x->cell != y->cell) return x->cell < y->cell ? -1 : 1;
  return (x->idx > y->idx) - (x->idx < y->idx);
}

int orc_voxel_grid(const orc_point* pts, int n, float leaf, orc_point* out) {
  if (n <= 0) return 0;
  const float inv = 1.0f / leaf;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int i = 0; i < n; i++) {
    float v[3] = {pts[i].x, pts[i].y, pts[i].z};
    for (int a = 0; a < 3; a++) { if (v[a] < mn[a]) mn[a] = v[a]; if (v[a] > mx[a]) mx[a] = v[a]; }
  }
  int min_b[3], max_b[3], div_b[3];
  for (int a = 0; a < 3; a++) {
    min_b[a] = (int)floorf(mn[a] * inv);
    max_b[a] = (int)floorf(mx[a] * inv);
    div_b[a] = max_b[a] - min_b[a] + 1;
  }
  const int64_t mul1 = div_b[0], mul2 = (int64_t)div_b[0] * div_b[1];
  vox_key* keys = (vox_key*)malloc(sizeof(vox_key) * (size_t)n);
  for (int i = 0; i < n; i++) {
    int i0 = (int)(floorf(pts[i].x * inv) - (float)min_b[0]);
    int i1 = (int)(floorf(pts[i].y * inv) - (float)min_b[1]);
    int i2 = (int)(floorf(pts[i].z * inv) - (float)min_b[2]);
    keys[i].cell = i0 + i1 * mul1 + i2 * mul2;
    keys[i].idx = i;
  }
  qsort(keys, (size_t)n, sizeof(vox_key), vox_cmp);
  int m = 0;
  for (int i = 0; i < n;) {
    int j = i;
    float sx = 0.f, sy = 0.f, sz = 0.f, st = 0.f;
    while (j < n && keys[j].cell == keys[i].cell) {
      const orc_point* p = &pts[keys[j].idx];
      sx += p->x; sy += p->y; sz += p->z; st += p->t;     /* CentroidPoint accumulators (f32) */
      j++;
    }
    const float cnt = (float)(j - i);
    out[m].x = sx / cnt; out[m].y = sy / cnt; out[m].z = sz / cnt; out[m].t = st / cnt;
    m++;
    i = j;
  }
  free(keys);
  return m;
}

/* =============================================================================================
 * N1: HybridGrid local map store  (src/slam/map/hybrid_grid.cc:413-534)
 * ============================================================================================= */

typedef struct { int ix, iy, iz; orc_point* pts; int n, cap; } grid_cell;
struct orc_grid {
  float resolution, leaf;
  grid_cell* cells; int n_cells, cap_cells;     /* kept sorted by (iz, iy, ix) */
};

orc_grid* orc_grid_create(float resolution, float leaf) {
  orc_grid* g = (orc_grid*)calloc(1, sizeof(orc_grid));
  g->resolution = resolution; g->leaf = leaf;
  return g;
}
void orc_grid_free(orc_grid* g) {
  if (!g) return;
  for (int i = 0; i < g->n_cells; i++) free(g->cells[i].pts);
  free(g->cells); free(g);
}
static int cell_cmp3(int az, int ay, int ax, const grid_cell* c) {
  if (az != c->iz) return az < c->iz ? -1 : 1;
  if (ay != c->iy) return ay < c->iy ? -1 : 1;
  if (ax != c->ix) return ax < c->ix ? -1 : 1;
  return 0;
}
/* lower bound; *found = 1 if present */
static int grid_find(const orc_grid* g, int ix, int iy, int iz, int* found) {
  int lo = 0, hi = g->n_cells;
  while (lo < hi) {
    int mid = (lo + hi) / 2;
    int c = cell_cmp3(iz, iy, ix, &g->cells[mid]);
    if (c == 0) { *found = 1; return mid; }
    if (c < 0) hi = mid; else lo = mid + 1;
  }
  *found = 0;
  return lo;
}
/* HybridGridBase::GetCellIndex (:422-426): lround(double(p / resolution)) per axis, division in f32 */
static void grid_cell_index(float resolution, float x, float y, float z, int idx[3]) {
  idx[0] = (int)lround((double)(x / resolution));
  idx[1] = (int)lround((double)(y / resolution));
  idx[2] = (int)lround((double)(z / resolution));
}

int orc_grid_insert_scan(orc_grid* g, const orc_point* pts, int n) {
  if (n <= 0) return 0;                                            /* :504 */
  int* touched = (int*)malloc(sizeof(int) * (size_t)n);
  int nt = 0;
  for (int i = 0; i < n; i++) {                                    /* :506-511 */
    int id[3];
    grid_cell_index(g->resolution, pts[i].x, pts[i].y, pts[i].z, id);
    if (abs(id[0]) > 8191 || abs(id[1]) > 8191 || abs(id[2]) > 8191) { free(touched); return 7; }
    int found, pos = grid_find(g, id[0], id[1], id[2], &found);
    if (!found) {
      if (g->n_cells == g->cap_cells) {
        g->cap_cells = g->cap_cells ? 2 * g->cap_cells : 256;
        g->cells = (grid_cell*)realloc(g->cells, sizeof(grid_cell) * (size_t)g->cap_cells);
      }
      memmove(&g->cells[pos + 1], &g->cells[pos], sizeof(grid_cell) * (size_t)(g->n_cells - pos));
      g->cells[pos].ix = id[0]; g->cells[pos].iy = id[1]; g->cells[pos].iz = id[2];
      g->cells[pos].pts = NULL; g->cells[pos].n = 0; g->cells[pos].cap = 0;
      g->n_cells++;
    }
    grid_cell* c = &g->cells[pos];
    if (c->n == c->cap) { c->cap = c->cap ? 2 * c->cap : 16; c->pts = (orc_point*)realloc(c->pts, sizeof(orc_point) * (size_t)c->cap); }
    c->pts[c->n++] = pts[i];
  }
  /* down-sample every touched cell in place (:513-520) */
  for (int i = 0; i < n; i++) {
    int id[3];
    grid_cell_index(g->resolution, pts[i].x, pts[i].y, pts[i].z, id);
    int found, pos = grid_find(g, id[0], id[1], id[2], &found);
    (void)found;
    touched[nt++] = pos;
  }
  for (int t = 0; t < nt; t++) {
    grid_cell* c = &g->cells[touched[t]];
    if (c->cap < 0) continue;          /* already filtered in this call */
    orc_point* tmp = (orc_point*)malloc(sizeof(orc_point) * (size_t)(c->n > 0 ? c->n : 1));
    int m = orc_voxel_grid(c->pts, c->n, g->leaf, tmp);
    memcpy(c->pts, tmp, sizeof(orc_point) * (size_t)m);
    c->n = m;
    free(tmp);
    c->cap = -c->cap;                  /* mark */
  }
  for (int t = 0; t < nt; t++) { grid_cell* c = &g->cells[touched[t]]; if (c->cap < 0) c->cap = -c->cap; }
  free(touched);
  return 0;
}

int orc_grid_get_surrounded(const orc_grid* g, const orc_point* scan, int n, const double pose[7],
                            orc_point* out, int capacity) {
  unsigned char* hit = (unsigned char*)calloc((size_t)(g->n_cells > 0 ? g->n_cells : 1), 1);
  /* pose.cast<float>(): Rigid3f, Quaternionf * Vector3f + translation, all in f32 (:478) */
  const float qx = (float)pose[3], qy = (float)pose[4], qz = (float)pose[5], qw = (float)pose[6];
  const float tx = (float)pose[0], ty = (float)pose[1], tz = (float)pose[2];
  for (int p = 0; p < n; p++) {
    const float x = scan[p].x, y = scan[p].y, z = scan[p].z;
    const float nrm = sqrtf(x * x + y * y + z * z);
    if ((double)nrm > 60.0) continue;                               /* kDist, :474 */
    /* Eigen _transformVector in f32: uv = 2 (q.vec x v); v + w uv + q.vec x uv */
    float ux = qy * z - qz * y, uy = qz * x - qx * z, uz = qx * y - qy * x;
    ux += ux; uy += uy; uz += uz;
    const float cx = qy * uz - qz * uy, cy = qz * ux - qx * uz, cz = qx * uy - qy * ux;
    const float wx = (x + qw * ux + cx) + tx, wy = (y + qw * uy + cy) + ty, wz = (z + qw * uz + cz) + tz;
    for (int i = -1; i <= 1; ++i)
      for (int j = -1; j <= 1; ++j)
        for (int k = -1; k <= 1; ++k) {
          int id[3];
          grid_cell_index(g->resolution, wx + (float)i, wy + (float)j, wz + (float)k, id);   /* :479-480 */
          int found, pos = grid_find(g, id[0], id[1], id[2], &found);
          if (found) hit[pos] = 1;                                                           /* TryInsertGrid */
        }
  }
  int m = 0;
  for (int c = 0; c < g->n_cells; c++) {
    if (!hit[c]) continue;
    for (int i = 0; i < g->cells[c].n; i++) { if (m < capacity) out[m] = g->cells[c].pts[i]; m++; }
  }
  free(hit);
  return m;
}

int orc_grid_size(const orc_grid* g, int* n_cells) {
  int n = 0;
  for (int c = 0; c < g->n_cells; c++) n += g->cells[c].n;
  if (n_cells) *n_cells = g->n_cells;
  return n;
}
int orc_grid_dump(const orc_grid* g, orc_point* out, int capacity) {
  int m = 0;
  for (int c = 0; c < g->n_cells; c++)
    for (int i = 0; i < g->cells[c].n; i++) { if (m < capacity) out[m] = g->cells[c].pts[i]; m++; }
  return m;
}

/* ================================================================================================
 * N3: IMU deskew inputs
 * ============================================================================================== */

/* Eigen 3.3 QuaternionBase::slerp (Geometry/Quaternion.h): coefficients blended with
   sin((1-t)θ)/sinθ and sin(tθ)/sinθ, θ = acos|a·b|; linear when |a·b| >= 1 - eps; the second
   weight flips sign for a negative dot; the result is NOT normalised. */
static void eigen_slerp(const double a[4], const double b[4], double t, double out[4]) {
  const double one = 1.0 - DBL_EPSILON;
  const double d = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
  const double ad = fabs(d);
  double s0, s1;
  if (ad >= one) { s0 = 1.0 - t; s1 = t; }
  else {
    const double theta = acos(ad), st = sin(theta);
    s0 = sin((1.0 - t) * theta) / st;
    s1 = sin(t * theta) / st;
  }
  if (d < 0.0) s1 = -s1;
  for (int k = 0; k < 4; k++) out[k] = s0 * a[k] + s1 * b[k];
}

int orc_delta_qp(const double* sum_dt, const double* delta_q, const double* delta_p, int n_samples,
                 double dt, double q_out[4], double p_out[3]) {
  q_out[0] = q_out[1] = q_out[2] = 0.0; q_out[3] = 1.0;
  p_out[0] = p_out[1] = p_out[2] = 0.0;
  if (n_samples < 2 || !(dt <= sum_dt[n_samples - 1] && dt >= sum_dt[0])) return 1;      /* :26-30 */
  int lo = 0, hi = n_samples;                                                            /* std::upper_bound, :33 */
  while (lo < hi) { const int mid = (lo + hi) / 2; if (dt < sum_dt[mid]) hi = mid; else lo = mid + 1; }
  int idx = lo - 1;                                                                      /* :35 */
  if (idx > n_samples - 2) idx = n_samples - 2;
  const double s = (dt - sum_dt[idx]) / (sum_dt[idx + 1] - sum_dt[idx]);                 /* :37 */
  eigen_slerp(delta_q + 4 * idx, delta_q + 4 * idx + 4, s, q_out);                       /* :38 */
  for (int k = 0; k < 3; k++) p_out[k] = (1 - s) * delta_p[3 * idx + k] + s * delta_p[3 * idx + 3 + k];   /* :39 */
  return 0;
}

/* GetDeltaQP(preintegration, pointOri.intensity) for every point of a cloud, as the is_initialized matcher does per feature
   (mapping_scan_matcher.cc:113-117,183-187); returns the number of refused time stamps */
int orc_delta_qp_cloud(const double* sum_dt, const double* delta_q, const double* delta_p, int n_samples,
                       const orc_point* pts, int n, double* dq, double* dp) {
  int bad = 0;
  for (int i = 0; i < n; i++) bad += orc_delta_qp(sum_dt, delta_q, delta_p, n_samples, (double)pts[i].t, dq + 4 * i, dp + 3 * i);
  return bad;
}

int orc_deskew_cloud(const double* sum_dt, const double* delta_q, const double* delta_p, int n_samples,
                     orc_point* pts, int n, const double rot_odom[4], const double velocity[3],
                     const double gravity[3]) {
  const double conj[4] = {-rot_odom[0], -rot_odom[1], -rot_odom[2], rot_odom[3]};
  int bad = 0;
  for (int i = 0; i < n; i++) {
    const double dt = (double)pts[i].t;                                                  /* auto dt = e.intensity, :200 */
    double q[4], p[3];
    if (orc_delta_qp(sum_dt, delta_q, delta_p, n_samples, dt, q, p)) { bad++; continue; }
    const double e[3] = {(double)pts[i].x, (double)pts[i].y, (double)pts[i].z};
    double a[3], m[3], b[3];
    orc_quat_rotate(q, e, a);
    for (int k = 0; k < 3; k++) m[k] = velocity[k] * dt - 0.5 * gravity[k] * dt * dt;
    orc_quat_rotate(conj, m, b);
    pts[i].x = (float)(a[0] + b[0] + p[0]);
    pts[i].y = (float)(a[1] + b[1] + p[1]);
    pts[i].z = (float)(a[2] + b[2] + p[2]);
  }
  return bad;
}

int orc_undistort_cloud(const double* sum_dt, const double* delta_q, const double* delta_p, int n_samples,
                        orc_point* pts, int n) {
  int bad = 0;
  for (int i = 0; i < n; i++) {
    double q[4], p[3];
    if (!(pts[i].t >= 0.f) || orc_delta_qp(sum_dt, delta_q, delta_p, n_samples, (double)pts[i].t, q, p)) { bad++; continue; }
    /* Quaternionf * Vector3f: Eigen _transformVector, uv = 2 (q.vec x v); v + w uv + q.vec x uv */
    const float qx = (float)q[0], qy = (float)q[1], qz = (float)q[2], qw = (float)q[3];
    const float x = pts[i].x, y = pts[i].y, z = pts[i].z;
    float ux = qy * z - qz * y, uy = qz * x - qx * z, uz = qx * y - qy * x;
    ux += ux; uy += uy; uz += uz;
    const float cx = qy * uz - qz * uy, cy = qz * ux - qx * uz, cz = qx * uy - qy * ux;
    pts[i].x = x + qw * ux + cx; pts[i].y = y + qw * uy + cy; pts[i].z = z + qw * uz + cz;
  }
  return bad;
}

void orc_transform_cloud(const orc_point* in, int n, const double pose[7], orc_point* out) {
  for (int i = 0; i < n; i++) {
    const float v[3] = {in[i].x, in[i].y, in[i].z};
    float o[3];
    orc_transform_point(pose, v, o);
    out[i].x = o[0]; out[i].y = o[1]; out[i].z = o[2]; out[i].t = in[i].t;
  }
}

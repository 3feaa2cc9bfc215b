/*
 * kmc_oracle.c -- CPU ORACLE (test infrastructure, never shipped, never on the product path).
 * See kmc_oracle.h for scope, conventions and parity status.  Plain C99, double precision, libm only.
 *
 * Each function restates, operation by operation, the reference function cited next to it
 * (paths relative to /root/reference).  Where the reference calls into Eigen 3.4 (un-vendored
 * dependency, apt libeigen3-dev on ubuntu:22.04 per Dockerfile:3) the published Eigen algorithm is
 * restated and named.
 */
#include "kmc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------------------------------
 * small dense helpers (row-major 3x3)
 * ---------------------------------------------------------------------------------------------- */
static void mat3_identity(double M[9]) {
  memset(M, 0, 9 * sizeof(double));
  M[0] = M[4] = M[8] = 1.0;
}

void kmo_mat3_mul(const double A[9], const double B[9], double C[9]) {
  double T[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0.0;
      for (int k = 0; k < 3; ++k) s += A[3 * i + k] * B[3 * k + j];
      T[3 * i + j] = s;
    }
  memcpy(C, T, sizeof(T));
}

static void mat3_vec(const double A[9], const double v[3], double out[3]) {
  double t0 = A[0] * v[0] + A[1] * v[1] + A[2] * v[2];
  double t1 = A[3] * v[0] + A[4] * v[1] + A[5] * v[2];
  double t2 = A[6] * v[0] + A[7] * v[1] + A[8] * v[2];
  out[0] = t0;
  out[1] = t1;
  out[2] = t2;
}

double kmo_mat3_det(const double A[9]) {
  return A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) +
         A[2] * (A[3] * A[7] - A[4] * A[6]);
}

/* Eigen 3.4 Inverse_impl for fixed 3x3 (compute_inverse_size3): cofactor matrix scaled by 1/det. */
static void mat3_inverse(const double A[9], double out[9]) {
  double c00 = A[4] * A[8] - A[5] * A[7];
  double c10 = A[5] * A[6] - A[3] * A[8];
  double c20 = A[3] * A[7] - A[4] * A[6];
  double det = A[0] * c00 + A[1] * c10 + A[2] * c20;
  double invdet = 1.0 / det;
  double T[9];
  T[0] = c00 * invdet;
  T[3] = c10 * invdet;
  T[6] = c20 * invdet;
  T[1] = (A[2] * A[7] - A[1] * A[8]) * invdet;
  T[4] = (A[0] * A[8] - A[2] * A[6]) * invdet;
  T[7] = (A[1] * A[6] - A[0] * A[7]) * invdet;
  T[2] = (A[1] * A[5] - A[2] * A[4]) * invdet;
  T[5] = (A[2] * A[3] - A[0] * A[5]) * invdet;
  T[8] = (A[0] * A[4] - A[1] * A[3]) * invdet;
  memcpy(out, T, sizeof(T));
}

static double vec3_norm(const double a[3]) { return sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }

/* ------------------------------------------------------------------------------------------------
 * L0: lie_algebra.cpp
 * ---------------------------------------------------------------------------------------------- */

/* lie_algebra.cpp:7-18 */
void kmo_hat(const double a[3], double M[9]) {
  memset(M, 0, 9 * sizeof(double));
  M[1] = -a[2]; /* (0,1) */
  M[2] = a[1];  /* (0,2) */
  M[3] = a[2];  /* (1,0) */
  M[5] = -a[0]; /* (1,2) */
  M[6] = -a[1]; /* (2,0) */
  M[7] = a[0];  /* (2,1) */
}

/* lie_algebra.cpp:20  -> {a(2,1), a(0,2), a(1,0)} */
void kmo_vee(const double M[9], double a[3]) {
  a[0] = M[7];
  a[1] = M[2];
  a[2] = M[3];
}

/* lie_algebra.cpp:22-35 */
void kmo_so3_exp(const double phi[3], double R[9]) {
  double angle = vec3_norm(phi);
  if (angle < 1e-6) { /* :25-28 first-order: I + Hat(phi) */
    double H[9];
    kmo_hat(phi, H);
    mat3_identity(R);
    for (int i = 0; i < 9; ++i) R[i] += H[i];
    return;
  }
  double axis[3] = {phi[0] / angle, phi[1] / angle, phi[2] / angle};
  double c = cos(angle);
  double s = sin(angle);
  double H[9];
  kmo_hat(axis, H);
  /* :34  (cos*I) + ((1-cos)*axis*axis^T) + (sin*Hat(axis));  (1-cos)*axis is formed first (left-to-right) */
  double sa[3] = {(1.0 - c) * axis[0], (1.0 - c) * axis[1], (1.0 - c) * axis[2]};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double id = (i == j) ? c : 0.0;
      R[3 * i + j] = (id + sa[i] * axis[j]) + s * H[3 * i + j];
    }
}

/* lie_algebra.cpp:37-49 */
void kmo_so3_log(const double R[9], double phi[3]) {
  double c = (0.5 * (R[0] + R[4] + R[8])) - 0.5;
  if (c < -1.0) c = -1.0; /* std::clamp :39 */
  if (c > 1.0) c = 1.0;
  double angle = acos(c);
  if (angle < 1e-6) { /* :43-46 Vee(R - I) */
    double M[9];
    memcpy(M, R, sizeof(M));
    M[0] -= 1.0;
    M[4] -= 1.0;
    M[8] -= 1.0;
    kmo_vee(M, phi);
    return;
  }
  double k = 0.5 * angle / sin(angle); /* :48 */
  double M[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) M[3 * i + j] = k * (R[3 * i + j] - R[3 * j + i]);
  kmo_vee(M, phi);
}

/* lie_algebra.cpp:51-65 */
void kmo_left_jacobian(const double phi[3], double J[9]) {
  double angle = vec3_norm(phi);
  if (angle < 1e-6) { /* :54-57  I + 0.5*Hat(phi) */
    double H[9];
    kmo_hat(phi, H);
    mat3_identity(J);
    for (int i = 0; i < 9; ++i) J[i] += 0.5 * H[i];
    return;
  }
  double axis[3] = {phi[0] / angle, phi[1] / angle, phi[2] / angle};
  double c = cos(angle);
  double s = sin(angle);
  double H[9];
  kmo_hat(axis, H);
  double sa = s / angle;
  double k2 = 1.0 - (s / angle);
  double k3 = (1 - c) / angle;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double id = (i == j) ? sa : 0.0;
      J[3 * i + j] = (id + k2 * (axis[i] * axis[j])) + k3 * H[3 * i + j]; /* :63-64 */
    }
}

/* lie_algebra.cpp:67-81 */
void kmo_inverse_left_jacobian(const double phi[3], double J[9]) {
  double angle = vec3_norm(phi);
  if (angle < 1e-6) { /* :70-73  I - 0.5*Hat(phi) */
    double H[9];
    kmo_hat(phi, H);
    mat3_identity(J);
    for (int i = 0; i < 9; ++i) J[i] -= 0.5 * H[i];
    return;
  }
  double axis[3] = {phi[0] / angle, phi[1] / angle, phi[2] / angle};
  double half_angle = 0.5 * angle;
  double half_angle_cot = 1.0 / tan(half_angle);
  double H[9];
  kmo_hat(axis, H);
  double k1 = half_angle * half_angle_cot;
  double k2 = 1 - (half_angle * half_angle_cot);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double id = (i == j) ? k1 : 0.0;
      J[3 * i + j] = (id + k2 * (axis[i] * axis[j])) - half_angle * H[3 * i + j]; /* :79-80 */
    }
}

/* lie_algebra.cpp:83-92 ; twist = [rho; phi] */
void kmo_se3_exp(const double xi[6], kmo_affine* T) {
  const double* rho = xi;
  const double* phi = xi + 3;
  double J[9];
  kmo_so3_exp(phi, T->R); /* T = Identity; T *= Exp(phi)  (:87-88) */
  kmo_left_jacobian(phi, J);
  mat3_vec(J, rho, T->t); /* :89 */
}

/* lie_algebra.cpp:94-103 */
void kmo_se3_log(const kmo_affine* T, double xi[6]) {
  double Rot[9], Jinv[9];
  kmo_affine_rotation(T, Rot);   /* T.rotation() : polar factor, :95 */
  kmo_so3_log(Rot, xi + 3);      /* phi */
  kmo_inverse_left_jacobian(xi + 3, Jinv);
  mat3_vec(Jinv, T->t, xi);      /* rho = J^-1 * t, :96 */
}

/* ------------------------------------------------------------------------------------------------
 * Eigen::Transform<double,3,Affine> operations used on the path
 * ---------------------------------------------------------------------------------------------- */
void kmo_affine_identity(kmo_affine* T) {
  mat3_identity(T->R);
  T->t[0] = T->t[1] = T->t[2] = 0.0;
}

/* Transform::inverse(Affine): linear^-1 by the general 3x3 inverse, t' = -(linear^-1 * t).
 * Called at trajectory_interpolation.cpp:35 and :44. */
void kmo_affine_inverse(const kmo_affine* T, kmo_affine* out) {
  double Ri[9], ti[3];
  mat3_inverse(T->R, Ri);
  mat3_vec(Ri, T->t, ti);
  memcpy(out->R, Ri, sizeof(Ri));
  out->t[0] = -ti[0];
  out->t[1] = -ti[1];
  out->t[2] = -ti[2];
}

/* Transform * Transform (trajectory_interpolation.cpp:35, :40, :44) */
void kmo_affine_mul(const kmo_affine* A, const kmo_affine* B, kmo_affine* out) {
  double R[9], t[3];
  kmo_mat3_mul(A->R, B->R, R);
  mat3_vec(A->R, B->t, t);
  t[0] += A->t[0];
  t[1] += A->t[1];
  t[2] += A->t[2];
  memcpy(out->R, R, sizeof(R));
  memcpy(out->t, t, sizeof(t));
}

/* Transform::rotation() in Affine mode = computeRotationScaling: rotation = U * diag(1,1,x) * V^T with
 * x = sign(det(U V^T)) (Eigen/src/Geometry/Transform.h).  Eigen obtains U,V from a two-sided JacobiSVD;
 * the orthogonal polar factor U V^T is unique for a non-singular matrix, so a one-sided (Hestenes)
 * Jacobi SVD gives the same matrix up to rounding (~1e-16). */
void kmo_affine_rotation(const kmo_affine* T, double Rout[9]) {
  double G[9], V[9];
  memcpy(G, T->R, sizeof(G));
  mat3_identity(V);
  for (int sweep = 0; sweep < 60; ++sweep) {
    int rotated = 0;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int i = 0; i < 3; ++i) {
          alpha += G[3 * i + p] * G[3 * i + p];
          beta += G[3 * i + q] * G[3 * i + q];
          gamma += G[3 * i + p] * G[3 * i + q];
        }
        if (fabs(gamma) <= 1e-300 || fabs(gamma) <= 2.2204460492503131e-16 * sqrt(alpha * beta)) continue;
        rotated = 1;
        double zeta = (beta - alpha) / (2.0 * gamma);
        double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        double c = 1.0 / sqrt(1.0 + t * t);
        double s = c * t;
        for (int i = 0; i < 3; ++i) {
          double gp = G[3 * i + p], gq = G[3 * i + q];
          G[3 * i + p] = c * gp - s * gq;
          G[3 * i + q] = s * gp + c * gq;
          double vp = V[3 * i + p], vq = V[3 * i + q];
          V[3 * i + p] = c * vp - s * vq;
          V[3 * i + q] = s * vp + c * vq;
        }
      }
    if (!rotated) break;
  }
  double sigma[3], U[9];
  int jmin = 0;
  for (int j = 0; j < 3; ++j) {
    sigma[j] = sqrt(G[j] * G[j] + G[3 + j] * G[3 + j] + G[6 + j] * G[6 + j]);
    if (sigma[j] < sigma[jmin]) jmin = j;
  }
  for (int j = 0; j < 3; ++j)
    for (int i = 0; i < 3; ++i) U[3 * i + j] = (sigma[j] > 0.0) ? G[3 * i + j] / sigma[j] : (i == j ? 1.0 : 0.0);
  double Q[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0.0;
      for (int k = 0; k < 3; ++k) s += U[3 * i + k] * V[3 * j + k];
      Q[3 * i + j] = s;
    }
  if (kmo_mat3_det(Q) < 0) { /* flip the column of the smallest singular value */
    for (int i = 0; i < 3; ++i) U[3 * i + jmin] = -U[3 * i + jmin];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double s = 0.0;
        for (int k = 0; k < 3; ++k) s += U[3 * i + k] * V[3 * j + k];
        Q[3 * i + j] = s;
      }
  }
  memcpy(Rout, Q, sizeof(Q));
}

/* Affine3d * Vector4d (motion_compensation.cpp:13): full homogeneous product, w row passes through. */
void kmo_affine_apply4(const kmo_affine* T, const double p[4], double out[4]) {
  double x = ((T->R[0] * p[0] + T->R[1] * p[1]) + T->R[2] * p[2]) + T->t[0] * p[3];
  double y = ((T->R[3] * p[0] + T->R[4] * p[1]) + T->R[5] * p[2]) + T->t[1] * p[3];
  double z = ((T->R[6] * p[0] + T->R[7] * p[1]) + T->R[8] * p[2]) + T->t[2] * p[3];
  out[0] = x;
  out[1] = y;
  out[2] = z;
  out[3] = p[3];
}

/* ------------------------------------------------------------------------------------------------
 * L3 producer: data_io.cpp:68-88 OxtsToPose
 * Eigen: AngleAxisd * AngleAxisd yields a Quaterniond product; Matrix3d(q) = q.toRotationMatrix().
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  double w, x, y, z;
} quat;

static quat quat_from_axis_angle(double angle, double ax, double ay, double az) {
  double ha = 0.5 * angle; /* Quaternion(AngleAxis): w = cos(a/2), vec = sin(a/2)*axis */
  double s = sin(ha);
  quat q = {cos(ha), s * ax, s * ay, s * az};
  return q;
}

static quat quat_mul(quat a, quat b) { /* Eigen quat_product (scalar path) */
  quat r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}

static void quat_to_matrix(quat q, double R[9]) { /* QuaternionBase::toRotationMatrix */
  double tx = 2.0 * q.x, ty = 2.0 * q.y, tz = 2.0 * q.z;
  double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1.0 - (tyy + tzz);
  R[1] = txy - twz;
  R[2] = txz + twy;
  R[3] = txy + twz;
  R[4] = 1.0 - (txx + tzz);
  R[5] = tyz - twx;
  R[6] = txz - twy;
  R[7] = tyz + twx;
  R[8] = 1.0 - (txx + tyy);
}

void kmo_oxts_to_pose(const kmo_oxts* o, double scale, kmo_affine* out) {
  const double earth_radius = 6378137.0;                                         /* :70 */
  double tx = scale * earth_radius * M_PI * o->lon / 180.0;                       /* :73 */
  double ty = scale * earth_radius * log(tan(M_PI * (90.0 + o->lat) / 360.0));    /* :74 */
  double tz = o->alt;                                                             /* :75 */
  quat qz = quat_from_axis_angle(o->yaw, 0, 0, 1);                                /* :78 */
  quat qy = quat_from_axis_angle(o->pitch, 0, 1, 0);                              /* :79 */
  quat qx = quat_from_axis_angle(o->roll, 1, 0, 0);                               /* :80 */
  quat q = quat_mul(quat_mul(qz, qy), qx);
  quat_to_matrix(q, out->R); /* pose = R * Identity (:84) */
  out->t[0] = tx;            /* :85 */
  out->t[1] = ty;
  out->t[2] = tz;
}

/* ------------------------------------------------------------------------------------------------
 * L1: trajectory_interpolation.cpp
 * ---------------------------------------------------------------------------------------------- */
void kmo_interpolator_from_poses(double t1, const kmo_affine* p1, double t2, const kmo_affine* p2,
                                 kmo_interpolator* out) { /* :27-29 */
  out->time_1 = t1;
  out->pose_1 = *p1;
  out->time_2 = t2;
  out->pose_2 = *p2;
}

void kmo_interpolator_from_oxts(const kmo_oxts* o0, const kmo_oxts* o1, kmo_interpolator* out) { /* :21-25 */
  out->time_1 = o0->stamp;
  kmo_oxts_to_pose(o0, 1.0, &out->pose_1); /* default scale = 1.0, data_io.hpp:14 */
  out->time_2 = o1->stamp;
  kmo_oxts_to_pose(o1, 1.0, &out->pose_2);
}

static int time_is_in_range(const kmo_interpolator* ti, double time) { /* :47 */
  return (time >= ti->time_1) && (time <= ti->time_2);
}

static double fraction_of_trajectory(const kmo_interpolator* ti, double time) { /* :49-51 */
  return (time - ti->time_1) / (ti->time_2 - ti->time_1);
}

int kmo_get_pose_at_time(const kmo_interpolator* ti, double time, kmo_affine* out) { /* :31-41 */
  if (!time_is_in_range(ti, time)) return KMO_ERR_TIME_OUT_OF_RANGE; /* assert, kept in release (:9,:32) */
  kmo_affine p1inv, rel, Tx;
  double f[6], fx[6];
  kmo_affine_inverse(&ti->pose_1, &p1inv);
  kmo_affine_mul(&p1inv, &ti->pose_2, &rel);
  kmo_se3_log(&rel, f);                        /* :35 */
  double x = fraction_of_trajectory(ti, time); /* :36 */
  for (int i = 0; i < 6; ++i) fx[i] = x * f[i]; /* :37 */
  kmo_se3_exp(fx, &Tx);                        /* :38 */
  kmo_affine_mul(&ti->pose_1, &Tx, out);       /* :40 */
  return KMO_OK;
}

int kmo_relative_pose_between_times(const kmo_interpolator* ti, double anchor, double query,
                                    kmo_affine* out) { /* :43-45 */
  kmo_affine a, q, ainv;
  int rc = kmo_get_pose_at_time(ti, anchor, &a);
  if (rc != KMO_OK) return rc;
  rc = kmo_get_pose_at_time(ti, query, &q);
  if (rc != KMO_OK) return rc;
  kmo_affine_inverse(&a, &ainv);
  kmo_affine_mul(&ainv, &q, out);
  return KMO_OK;
}

int kmo_interpolate_trajectory(const kmo_oxts* o1, const kmo_oxts* o2, double time, kmo_affine* out) { /* :14-19 */
  kmo_interpolator ti;
  kmo_interpolator_from_oxts(o1, o2, &ti);
  return kmo_get_pose_at_time(&ti, time, out);
}

/* data_io.cpp:253-269 MakeFrame (pose part) */
int kmo_make_frame_poses(const kmo_oxts* o_nm1, const kmo_oxts* o_n, const kmo_oxts* o_np1,
                         double stamp_start, double stamp_end, kmo_affine* T_start, kmo_affine* T_end) {
  int rc = kmo_interpolate_trajectory(o_nm1, o_n, stamp_start, T_start); /* :263-264 */
  if (rc != KMO_OK) return rc;
  return kmo_interpolate_trajectory(o_n, o_np1, stamp_end, T_end);       /* :265-266 */
}

/* ------------------------------------------------------------------------------------------------
 * L2a: timestamp_mocking.cpp
 * ---------------------------------------------------------------------------------------------- */
double kmo_fraction_of_scan_completed(const double p[4]) { /* :46 */
  return (M_PI - atan2(p[1], p[0])) / (2.0 * M_PI);
}

double kmo_pseudo_timestamp(const double p[4], double scan_start, double scan_end) { /* :49-54 */
  double position_in_scan = kmo_fraction_of_scan_completed(p);
  double scan_duration = scan_end - scan_start;
  return scan_start + (position_in_scan * scan_duration);
}

void kmo_pseudo_timestamps(const double* cloud, size_t n, double start, double end, double* stamps) { /* :56-63 */
  for (size_t i = 0; i < n; ++i) {
    double p[4] = {cloud[i], cloud[n + i], cloud[2 * n + i], cloud[3 * n + i]};
    stamps[i] = kmo_pseudo_timestamp(p, start, end);
  }
}

/* ------------------------------------------------------------------------------------------------
 * L2: motion_compensation.cpp
 * ---------------------------------------------------------------------------------------------- */
int kmo_motion_compensate_point(const kmo_interpolator* ti, double point_stamp, const double p[4],
                                double requested_time, double out[4]) { /* :9-14 */
  kmo_affine correction;
  int rc = kmo_relative_pose_between_times(ti, requested_time, point_stamp, &correction); /* :11 */
  if (rc != KMO_OK) return rc;
  kmo_affine_apply4(&correction, p, out); /* :13 */
  return KMO_OK;
}

int kmo_motion_compensate_frame(const double* cloud, const double* stamps, size_t n, double stamp_start,
                                const kmo_affine* T_start, double stamp_end, const kmo_affine* T_end,
                                double requested_time, double* out, size_t* n_bad) { /* :16-28 */
  kmo_interpolator ti;
  kmo_interpolator_from_poses(stamp_start, T_start, stamp_end, T_end, &ti); /* :17-18 */
  size_t bad = 0;
  for (size_t i = 0; i < n; ++i) { /* :22-25 */
    double p[4] = {cloud[i], cloud[n + i], cloud[2 * n + i], cloud[3 * n + i]};
    double q[4];
    int rc = kmo_motion_compensate_point(&ti, stamps[i], p, requested_time, q);
    if (rc != KMO_OK) {
      ++bad;
      q[0] = q[1] = q[2] = q[3] = NAN;
    }
    out[i] = q[0];
    out[n + i] = q[1];
    out[2 * n + i] = q[2];
    out[3 * n + i] = q[3];
  }
  if (n_bad) *n_bad = bad;
  return bad ? KMO_ERR_TIME_OUT_OF_RANGE : KMO_OK;
}

/* ------------------------------------------------------------------------------------------------
 * KITTI-layout pipeline helper (f32 AoS in; f64 and/or f32 out)
 * ---------------------------------------------------------------------------------------------- */
int kmo_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void kmo_set_num_threads(int threads) {
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#else
  (void)threads;
#endif
}

int kmo_deskew_xyzi_f32(const float* xyzi, size_t n, double stamp_start, const kmo_affine* T_start,
                        double stamp_end, const kmo_affine* T_end, double requested_time, int mode,
                        int threads, double* out_xyz_f64, float* out_xyzi_f32, double* stamps_out,
                        size_t* n_bad) {
  kmo_interpolator ti;
  kmo_interpolator_from_poses(stamp_start, T_start, stamp_end, T_end, &ti);

  /* HOISTED mode constants: f = Log(T_start^-1 T_end) once, x_r once */
  double f[6] = {0, 0, 0, 0, 0, 0};
  double x_r = 0.0;
  int req_ok = time_is_in_range(&ti, requested_time);
  if (mode == 1) {
    kmo_affine p1inv, rel;
    kmo_affine_inverse(&ti.pose_1, &p1inv);
    kmo_affine_mul(&p1inv, &ti.pose_2, &rel);
    kmo_se3_log(&rel, f);
    x_r = fraction_of_trajectory(&ti, requested_time);
  }

  size_t bad = 0;
  long long nn = (long long)n;
#ifdef _OPENMP
  if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) reduction(+ : bad) schedule(static)
#else
  (void)threads;
#endif
  for (long long i = 0; i < nn; ++i) {
    /* data_io.cpp:126-135 : f32 -> f64, homogeneous 1 */
    double p[4] = {(double)xyzi[4 * i + 0], (double)xyzi[4 * i + 1], (double)xyzi[4 * i + 2], 1.0};
    double stamp = kmo_pseudo_timestamp(p, stamp_start, stamp_end); /* data_io.cpp:163 */
    if (stamps_out) stamps_out[i] = stamp;
    double q[4];
    int rc;
    if (mode == 0) {
      rc = kmo_motion_compensate_point(&ti, stamp, p, requested_time, q);
    } else {
      if (!req_ok || !time_is_in_range(&ti, stamp)) {
        rc = KMO_ERR_TIME_OUT_OF_RANGE;
      } else {
        double x_i = fraction_of_trajectory(&ti, stamp);
        double s = x_i - x_r;
        double fx[6];
        kmo_affine Tx;
        for (int k = 0; k < 6; ++k) fx[k] = s * f[k];
        kmo_se3_exp(fx, &Tx);
        kmo_affine_apply4(&Tx, p, q);
        rc = KMO_OK;
      }
    }
    if (rc != KMO_OK) {
      ++bad;
      q[0] = q[1] = q[2] = NAN;
    }
    if (out_xyz_f64) {
      out_xyz_f64[3 * i + 0] = q[0];
      out_xyz_f64[3 * i + 1] = q[1];
      out_xyz_f64[3 * i + 2] = q[2];
    }
    if (out_xyzi_f32) { /* data_io.cpp:300-310 : cast to float, intensity passed through */
      out_xyzi_f32[4 * i + 0] = (float)q[0];
      out_xyzi_f32[4 * i + 1] = (float)q[1];
      out_xyzi_f32[4 * i + 2] = (float)q[2];
      out_xyzi_f32[4 * i + 3] = xyzi[4 * i + 3];
    }
  }
  if (n_bad) *n_bad = bad;
  return bad ? KMO_ERR_TIME_OUT_OF_RANGE : KMO_OK;
}

/* ------------------------------------------------------------------------------------------------
 * N-knot trajectory: the reference's two-pose interpolator chained over consecutive knots (see kmc_oracle.h)
 * ---------------------------------------------------------------------------------------------- */
static size_t traj_segment(const double* times, size_t n_knots, double t) {
  size_t k = 0;
  while (k + 2 < n_knots && t >= times[k + 1]) ++k;
  return k;
}

int kmo_traj_pose_at_time(const double* times, const kmo_affine* poses, size_t n_knots, double t, kmo_affine* out) {
  if (n_knots < 2 || !(t >= times[0] && t <= times[n_knots - 1])) return KMO_ERR_TIME_OUT_OF_RANGE;
  size_t k = traj_segment(times, n_knots, t);
  kmo_interpolator ti;
  kmo_interpolator_from_poses(times[k], &poses[k], times[k + 1], &poses[k + 1], &ti);
  return kmo_get_pose_at_time(&ti, t, out); /* trajectory_interpolation.cpp:31-41 on the bracketing pair */
}

static int traj_correct_point(const double* times, const kmo_affine* poses, size_t n_knots, const kmo_affine* Treq_inv,
                              double stamp, const double p[4], double q[4], uint32_t* bracket) {
  kmo_affine Tq, corr;
  int rc = kmo_traj_pose_at_time(times, poses, n_knots, stamp, &Tq);
  if (rc != KMO_OK) return rc;
  if (bracket) *bracket = (uint32_t)traj_segment(times, n_knots, stamp);
  kmo_affine_mul(Treq_inv, &Tq, &corr); /* GetPoseAtTime(anchor)^-1 * GetPoseAtTime(query), :43-45 */
  kmo_affine_apply4(&corr, p, q);       /* motion_compensation.cpp:13 */
  return KMO_OK;
}

int kmo_deskew_xyzi_f32_traj(const float* xyzi, size_t n, double stamp_start, double stamp_end, const double* times,
                             const kmo_affine* poses, size_t n_knots, double requested_time, int threads,
                             double* out_xyz_f64, uint32_t* bracket_by_time_out, size_t* n_bad) {
  kmo_affine Treq, Treq_inv;
  int rc0 = kmo_traj_pose_at_time(times, poses, n_knots, requested_time, &Treq);
  if (rc0 != KMO_OK) {
    if (n_bad) *n_bad = n;
    return rc0;
  }
  kmo_affine_inverse(&Treq, &Treq_inv);
  size_t bad = 0;
  long long nn = (long long)n;
#ifdef _OPENMP
  if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) reduction(+ : bad) schedule(static)
#else
  (void)threads;
#endif
  for (long long i = 0; i < nn; ++i) {
    double p[4] = {(double)xyzi[4 * i + 0], (double)xyzi[4 * i + 1], (double)xyzi[4 * i + 2], 1.0};
    double stamp = kmo_pseudo_timestamp(p, stamp_start, stamp_end);
    double q[4];
    uint32_t b = 0;
    if (traj_correct_point(times, poses, n_knots, &Treq_inv, stamp, p, q, &b) != KMO_OK) {
      ++bad;
      q[0] = q[1] = q[2] = NAN;
    }
    if (out_xyz_f64) {
      out_xyz_f64[3 * i + 0] = q[0];
      out_xyz_f64[3 * i + 1] = q[1];
      out_xyz_f64[3 * i + 2] = q[2];
    }
    if (bracket_by_time_out) bracket_by_time_out[i] = b;
  }
  if (n_bad) *n_bad = bad;
  return bad ? KMO_ERR_TIME_OUT_OF_RANGE : KMO_OK;
}

int kmo_motion_compensate_frame_traj(const double* cloud, const double* stamps, size_t n, const double* times,
                                     const kmo_affine* poses, size_t n_knots, double requested_time, double* out,
                                     uint32_t* bracket_out, size_t* n_bad) {
  kmo_affine Treq, Treq_inv;
  int rc0 = kmo_traj_pose_at_time(times, poses, n_knots, requested_time, &Treq);
  if (rc0 != KMO_OK) {
    if (n_bad) *n_bad = n;
    return rc0;
  }
  kmo_affine_inverse(&Treq, &Treq_inv);
  size_t bad = 0;
  for (size_t i = 0; i < n; ++i) {
    double p[4] = {cloud[i], cloud[n + i], cloud[2 * n + i], cloud[3 * n + i]};
    double q[4];
    uint32_t b = 0;
    if (traj_correct_point(times, poses, n_knots, &Treq_inv, stamps[i], p, q, &b) != KMO_OK) {
      ++bad;
      q[0] = q[1] = q[2] = q[3] = NAN;
    }
    out[i] = q[0];
    out[n + i] = q[1];
    out[2 * n + i] = q[2];
    out[3 * n + i] = q[3];
    if (bracket_out) bracket_out[i] = b;
  }
  if (n_bad) *n_bad = bad;
  return bad ? KMO_ERR_TIME_OUT_OF_RANGE : KMO_OK;
}

/* trig-free bracket index: see the specification in kmc_oracle.h */
static int knot_ge_f32(float x, float y, float knot_c, float ck, float sk, int always_ge, int never_ge) {
  if (always_ge) return 1; /* c_k <= 0, decided on the f64 fraction */
  if (never_ge) return 0;  /* c_k > 1 */
  int xneg = signbit(x) != 0, yneg = signbit(y) != 0;
  int lt;
  if (x == 0.0f && y == 0.0f) {
    float fs = xneg ? (yneg ? 1.0f : 0.0f) : 0.5f;
    lt = fs < knot_c;
  } else {
    float a = ck * y;
    float b = sk * x;
    float cross = a - b;
    float c = ck * x;
    float d = sk * y;
    float dot = c + d;
    if (knot_c <= 0.5f) lt = !yneg && (cross > 0.0f || (cross == 0.0f && dot < 0.0f));
    else lt = !yneg || cross > 0.0f || (cross == 0.0f && dot < 0.0f);
  }
  return !lt;
}

void kmo_bracket_indices_f32(const float* xyzi, size_t n, const double* times, size_t n_knots, double stamp_start,
                             double stamp_end, uint32_t* out) {
  float kc[64], kcos[64], ksin[64];
  int always_ge[64], never_ge[64];
  size_t n_int = 0;
  for (size_t k = 1; k + 1 < n_knots && n_int < 64; ++k, ++n_int) {
    double c = (times[k] - stamp_start) / (stamp_end - stamp_start);
    double q = 4.0 * c;
    kc[n_int] = (float)c;
    always_ge[n_int] = c <= 0.0;
    never_ge[n_int] = c > 1.0;
    if (q == floor(q) && q >= 0.0 && q <= 4.0) { /* exact directions on the quarter turns */
      static const float tc[5] = {-1.f, 0.f, 1.f, 0.f, -1.f};
      static const float ts[5] = {0.f, 1.f, 0.f, -1.f, -0.f};
      kcos[n_int] = tc[(int)q];
      ksin[n_int] = ts[(int)q];
    } else {
      double alpha = M_PI - 2.0 * M_PI * c;
      kcos[n_int] = (float)cos(alpha);
      ksin[n_int] = (float)sin(alpha);
    }
  }
  for (size_t i = 0; i < n; ++i) {
    uint32_t k = 0;
    for (size_t j = 0; j < n_int; ++j) k += (uint32_t)knot_ge_f32(xyzi[4 * i], xyzi[4 * i + 1], kc[j], kcos[j], ksin[j], always_ge[j], never_ge[j]);
    out[i] = k;
  }
}

/* ---- N4: projection (camera_model.cpp:5-95 without the drawing; see kmc_oracle.h) ---- */
static int32_t trunc_i32(double v) { /* cvttsd2si */
  if (!(v > -2147483649.0 && v < 2147483648.0)) return INT32_MIN;
  return (int32_t)v;
}
static uint8_t sat_u8(double v) { /* cv::saturate_cast<uchar>(double) */
  double r = nearbyint(v);        /* default rounding mode: half to even, = cvRound */
  return (uint8_t)(r < 0.0 ? 0.0 : (r > 255.0 ? 255.0 : r));
}
static void project_one(double x, double y, double z, const kmo_camera_rig* g, size_t i, int32_t* uv, uint8_t* bgrv) {
  const double* T = g->tf_c00_lo;
  const double* R = g->R_rect_00;
  double c[3], r[3];
  for (int k = 0; k < 3; ++k) c[k] = ((T[4 * k] * x + T[4 * k + 1] * y) + T[4 * k + 2] * z) + T[4 * k + 3] * 1.0; /* :75 */
  for (int k = 0; k < 3; ++k) r[k] = ((R[3 * k] * c[0] + R[3 * k + 1] * c[1]) + R[3 * k + 2] * c[2]) + 0.0 * 1.0;  /* :81 */
  int valid = !((r[2] < 0.01) || (r[2] > g->max_range) || (r[1] > 1.25));                                          /* :21-24 */
  for (int cam = 0; cam < 4; ++cam) {
    const double* P = g->P_rect[cam];
    double h[3];
    for (int k = 0; k < 3; ++k) h[k] = ((P[4 * k] * r[0] + P[4 * k + 1] * r[1]) + P[4 * k + 2] * r[2]) + P[4 * k + 3] * 1.0; /* :9 */
    int32_t* o = uv + (i * 4 + (size_t)cam) * 2;
    o[0] = valid ? trunc_i32(h[0] / h[2]) : INT32_MIN; /* :12, :31 */
    o[1] = valid ? trunc_i32(h[1] / h[2]) : INT32_MIN;
  }
  uint8_t* q = bgrv + 4 * i;
  if (valid) {
    double cs = 255.0 * (r[2] / (g->max_range - 0.01)); /* :28-29 */
    q[0] = sat_u8(255.0 - cs);                          /* :32 */
    q[1] = sat_u8(cs);
    q[2] = sat_u8(255.0 - cs);
    q[3] = 1;
  } else {
    q[0] = q[1] = q[2] = q[3] = 0;
  }
}
void kmo_project_points(const double* x, const double* y, const double* z, size_t n, const kmo_camera_rig* rig, int32_t* uv,
                        uint8_t* bgrv) {
#pragma omp parallel for schedule(static) if (n >= 65536)
  for (size_t i = 0; i < n; ++i) project_one(x[i], y[i], z[i], rig, i, uv, bgrv);
}
void kmo_project_xyzi_f32(const float* xyzi, size_t n, const kmo_camera_rig* rig, int32_t* uv, uint8_t* bgrv) {
#pragma omp parallel for schedule(static) if (n >= 65536)
  for (size_t i = 0; i < n; ++i) project_one((double)xyzi[4 * i], (double)xyzi[4 * i + 1], (double)xyzi[4 * i + 2], rig, i, uv, bgrv);
}

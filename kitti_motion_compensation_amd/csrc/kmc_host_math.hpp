// kmc_host_math.hpp -- f64 host-side SE(3) arithmetic of the product (NOT the oracle).
//
// This is the once-per-frame pre-step that the device kernels depend on: it turns the two scan poses into
// the twist f = Log(T_start^-1 * T_end) and the anchor fraction x_r.  It stays in double on the host on
// purpose: T_start^-1 * T_end subtracts ~6e6 m Mercator translations and takes acos() next to 1
// (SURVEY.md H5).  Semantics follow the reference functions cited per function (paths relative to the
// reference repo); the formulation is this project's own (cross-product closed forms, Newton polar factor).
#pragma once

#include <cmath>
#include <cstring>

namespace kmc_host {

struct Vec3 {
  double x, y, z;
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator*(double s, Vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline Vec3 cross(Vec3 a, Vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline double norm(Vec3 a) { return std::sqrt(dot(a, a)); }

struct Mat3 {
  double m[3][3];
  static Mat3 identity() { return {{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}}; }
};
inline Mat3 operator*(const Mat3& A, const Mat3& B) {
  Mat3 C;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C.m[i][j] = A.m[i][0] * B.m[0][j] + A.m[i][1] * B.m[1][j] + A.m[i][2] * B.m[2][j];
  return C;
}
inline Vec3 operator*(const Mat3& A, Vec3 v) {
  return {A.m[0][0] * v.x + A.m[0][1] * v.y + A.m[0][2] * v.z, A.m[1][0] * v.x + A.m[1][1] * v.y + A.m[1][2] * v.z,
          A.m[2][0] * v.x + A.m[2][1] * v.y + A.m[2][2] * v.z};
}
inline Mat3 transpose(const Mat3& A) {
  Mat3 T;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) T.m[i][j] = A.m[j][i];
  return T;
}
inline double det(const Mat3& A) {
  return A.m[0][0] * (A.m[1][1] * A.m[2][2] - A.m[1][2] * A.m[2][1]) - A.m[0][1] * (A.m[1][0] * A.m[2][2] - A.m[1][2] * A.m[2][0]) +
         A.m[0][2] * (A.m[1][0] * A.m[2][1] - A.m[1][1] * A.m[2][0]);
}
// general inverse by the adjugate (what Eigen's Affine-mode Transform::inverse() needs)
inline bool inverse(const Mat3& A, Mat3* out) {
  const double d = det(A);
  if (d == 0.0 || !std::isfinite(d)) return false;
  const double r = 1.0 / d;
  Mat3 I;
  I.m[0][0] = (A.m[1][1] * A.m[2][2] - A.m[1][2] * A.m[2][1]) * r;
  I.m[0][1] = (A.m[0][2] * A.m[2][1] - A.m[0][1] * A.m[2][2]) * r;
  I.m[0][2] = (A.m[0][1] * A.m[1][2] - A.m[0][2] * A.m[1][1]) * r;
  I.m[1][0] = (A.m[1][2] * A.m[2][0] - A.m[1][0] * A.m[2][2]) * r;
  I.m[1][1] = (A.m[0][0] * A.m[2][2] - A.m[0][2] * A.m[2][0]) * r;
  I.m[1][2] = (A.m[0][2] * A.m[1][0] - A.m[0][0] * A.m[1][2]) * r;
  I.m[2][0] = (A.m[1][0] * A.m[2][1] - A.m[1][1] * A.m[2][0]) * r;
  I.m[2][1] = (A.m[0][1] * A.m[2][0] - A.m[0][0] * A.m[2][1]) * r;
  I.m[2][2] = (A.m[0][0] * A.m[1][1] - A.m[0][1] * A.m[1][0]) * r;
  *out = I;
  return true;
}

// A rigid/affine pose: x -> L x + t.   (the reference's kmc::Affine3d = Eigen::Affine3d, data_types.hpp:27)
struct Pose {
  Mat3 L;
  Vec3 t;
  static Pose identity() { return {Mat3::identity(), {0, 0, 0}}; }
  static Pose from_rt12(const double a[12]) {  // row-major 3x4 [R|t]
    Pose p;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) p.L.m[i][j] = a[4 * i + j];
    p.t = {a[3], a[7], a[11]};
    return p;
  }
  void to_rt12(double a[12]) const {
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) a[4 * i + j] = L.m[i][j];
    a[3] = t.x;
    a[7] = t.y;
    a[11] = t.z;
  }
};
inline Pose operator*(const Pose& A, const Pose& B) { return {A.L * B.L, A.L * B.t + A.t}; }
inline bool inverse(const Pose& A, Pose* out) {
  Mat3 Li;
  if (!inverse(A.L, &Li)) return false;
  *out = {Li, -1.0 * (Li * A.t)};
  return true;
}

// Orthogonal polar factor of L (what Eigen's Affine3d::rotation() returns, used at lie_algebra.cpp:95).
// Newton iteration Q <- (Q + Q^-T)/2 (Higham): quadratically convergent, 1-3 steps for a pose that is
// already orthonormal to rounding.  A reflection (det < 0) is not a pose: report failure.
inline bool polar_rotation(const Mat3& L, Mat3* out) {
  Mat3 Q = L;
  if (!(det(Q) > 0.0)) return false;
  for (int it = 0; it < 50; ++it) {
    Mat3 Qi;
    if (!inverse(Q, &Qi)) return false;
    const Mat3 QiT = transpose(Qi);
    double delta = 0.0;
    Mat3 N;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        N.m[i][j] = 0.5 * (Q.m[i][j] + QiT.m[i][j]);
        delta = std::fmax(delta, std::fabs(N.m[i][j] - Q.m[i][j]));
      }
    Q = N;
    if (delta < 1e-16) break;
  }
  *out = Q;
  return true;
}

// so(3) log: phi = vee(k (R - R^T)), k = theta / (2 sin theta); first-order below 1e-6 rad
// (same branch point as lie_algebra.cpp:37-49 so both sides agree bit-for-bit on which form is used).
// Conditioning: the reference takes theta = acos((tr R - 1) / 2) and divides by sin(theta).  acos next to -1 returns theta with
// an absolute error of 1e-16 / (pi - theta), so next to pi sin(theta) ~ pi - theta and with it the whole of phi carry a
// RELATIVE error of 1e-16 / (pi - theta)^2: 1e-8 at 1e-4 rad from a half turn, every digit at 1e-8 rad (measured against the
// oracle, which restates the reference's formula: tests/test_host_prestep.py).  Here sin(theta) is taken from where it is
// exact -- the norm of the antisymmetric part, |vee(R - R^T)| / 2 -- and theta from atan2(sin, cos); within 1.4e-3 rad of pi
// the axis comes from the symmetric part, (R + R^T) / 2 = cos I + (1 - cos) a a^T, and only its sign from the antisymmetric
// one.  phi is then good to ~1e-15 for every rotation, the half turn itself included (where either sign of the axis is a
// logarithm).  Away from pi the value is the reference's up to rounding.
inline Vec3 so3_log(const Mat3& R) {
  double c = 0.5 * (R.m[0][0] + R.m[1][1] + R.m[2][2]) - 0.5;
  c = c < -1.0 ? -1.0 : (c > 1.0 ? 1.0 : c);
  const Vec3 v = {R.m[2][1] - R.m[1][2], R.m[0][2] - R.m[2][0], R.m[1][0] - R.m[0][1]};  // 2 sin(theta) a
  const double s = 0.5 * norm(v);
  const double theta = std::atan2(s, c);
  if (theta < 1e-6) return {R.m[2][1], R.m[0][2], R.m[1][0]};
  if (c > -1.0 + 1e-6) return (0.5 * theta / s) * v;
  // next to a half turn
  double S[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) S[i][j] = 0.5 * (R.m[i][j] + R.m[j][i]);
  const int i = (S[0][0] >= S[1][1] && S[0][0] >= S[2][2]) ? 0 : (S[1][1] >= S[2][2] ? 1 : 2);
  const double one_minus_c = 1.0 - c;
  double a[3];
  a[i] = std::sqrt(std::fmax(0.0, (S[i][i] - c) / one_minus_c));  // >= 1/sqrt(3): the largest component of a unit vector
  for (int j = 0; j < 3; ++j)
    if (j != i) a[j] = S[i][j] / (one_minus_c * a[i]);
  Vec3 axis = {a[0], a[1], a[2]};
  axis = (1.0 / norm(axis)) * axis;
  if (dot(axis, v) < 0.0) axis = -1.0 * axis;
  return theta * axis;
}

// J^-1(phi) t = t - 1/2 phi x t + kappa phi x (phi x t),  kappa = (1 - (th/2) cot(th/2)) / th^2
// (closed form of lie_algebra.cpp:67-81; first-order below 1e-6 rad like the reference).
inline Vec3 inv_left_jacobian_apply(Vec3 phi, Vec3 t) {
  const double theta = norm(phi);
  const Vec3 pxt = cross(phi, t);
  if (theta < 1e-6) return t - 0.5 * pxt;
  const double h = 0.5 * theta;
  const double kappa = (1.0 - h / std::tan(h)) / (theta * theta);
  return t - 0.5 * pxt + kappa * cross(phi, pxt);
}

// J(phi) rho = rho + B phi x rho + C phi x (phi x rho)    (lie_algebra.cpp:51-65)
inline Vec3 left_jacobian_apply(Vec3 phi, Vec3 rho) {
  const double theta = norm(phi);
  const Vec3 pxr = cross(phi, rho);
  if (theta < 1e-6) return rho + 0.5 * pxr;
  const double t2 = theta * theta;
  const double B = (1.0 - std::cos(theta)) / t2;
  const double C = (theta - std::sin(theta)) / (t2 * theta);
  return rho + B * pxr + C * cross(phi, pxr);
}

// Rodrigues (lie_algebra.cpp:22-35)
inline Mat3 so3_exp(Vec3 phi) {
  const double theta = norm(phi);
  Mat3 R = Mat3::identity();
  if (theta < 1e-6) {
    R.m[0][1] = -phi.z; R.m[0][2] = phi.y; R.m[1][0] = phi.z; R.m[1][2] = -phi.x; R.m[2][0] = -phi.y; R.m[2][1] = phi.x;
    return R;
  }
  const Vec3 a = (1.0 / theta) * phi;
  const double c = std::cos(theta), s = std::sin(theta), v = 1.0 - c;
  R.m[0][0] = c + v * a.x * a.x;       R.m[0][1] = v * a.x * a.y - s * a.z; R.m[0][2] = v * a.x * a.z + s * a.y;
  R.m[1][0] = v * a.y * a.x + s * a.z; R.m[1][1] = c + v * a.y * a.y;       R.m[1][2] = v * a.y * a.z - s * a.x;
  R.m[2][0] = v * a.z * a.x - s * a.y; R.m[2][1] = v * a.z * a.y + s * a.x; R.m[2][2] = c + v * a.z * a.z;
  return R;
}

struct Twist {
  Vec3 rho, phi;  // reference order [rho; phi], lie_algebra.cpp:84-85
};

inline Pose se3_exp(const Twist& xi) { return {so3_exp(xi.phi), left_jacobian_apply(xi.phi, xi.rho)}; }  // :83-92

inline bool se3_log(const Pose& T, Twist* out) {  // :94-103
  Mat3 R;
  if (!polar_rotation(T.L, &R)) return false;
  out->phi = so3_log(R);
  out->rho = inv_left_jacobian_apply(out->phi, T.t);
  return true;
}

// f = Log(T_start^-1 * T_end).  The translation of the relative pose is formed as L_s^-1 (t_e - t_s): the
// subtraction of the two ~6e6 m Mercator vectors is then exact to ~1e-9 m instead of cancelling after two
// separate 3x3 products (the reference's own noise floor).
inline bool relative_twist(const Pose& T_start, const Pose& T_end, Twist* out) {
  Mat3 Li;
  if (!inverse(T_start.L, &Li)) return false;
  const Pose rel = {Li * T_end.L, Li * (T_end.t - T_start.t)};
  return se3_log(rel, out);
}

// GetPoseAtTime (trajectory_interpolation.cpp:31-41): pose_1 * Exp(x * Log(pose_1^-1 pose_2)).
inline int pose_at_time(double t1, const Pose& P1, double t2, const Pose& P2, double time, Pose* out) {
  if (!(time >= t1 && time <= t2)) return -1;  // TimeIsInRange :47 (the reference asserts)
  Twist f;
  if (!relative_twist(P1, P2, &f)) return -2;
  const double x = (time - t1) / (t2 - t1);  // :49-51
  *out = P1 * se3_exp({x * f.rho, x * f.phi});
  return 0;
}

// OxtsToPose (data_io.cpp:68-88): Mercator position, R = Rz(yaw) Ry(pitch) Rx(roll).
inline Pose oxts_to_pose(double lat, double lon, double alt, double roll, double pitch, double yaw, double scale) {
  constexpr double kEarthRadius = 6378137.0;
  constexpr double kPi = 3.14159265358979323846;
  Pose P;
  P.t = {scale * kEarthRadius * kPi * lon / 180.0, scale * kEarthRadius * std::log(std::tan(kPi * (90.0 + lat) / 360.0)), alt};
  const double cy = std::cos(yaw), sy = std::sin(yaw), cp = std::cos(pitch), sp = std::sin(pitch), cr = std::cos(roll),
               sr = std::sin(roll);
  P.L = {{{cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr},
          {sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr},
          {-sp, cp * sr, cp * cr}}};
  return P;
}

}  // namespace kmc_host

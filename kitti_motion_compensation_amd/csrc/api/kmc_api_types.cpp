// kmc_api_types.cpp -- arithmetic of the Eigen-free API types (include/kitti_motion_compensation/data_types.hpp).
#include <cmath>

#include "kmc_api_internal.hpp"

namespace kmc {

double Vector3d::norm() const { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

Twist operator*(double s, Twist const& t) {
  Twist r;
  for (int i = 0; i < 6; ++i) r(i) = s * t(i);
  return r;
}

Matrix3d Matrix3d::Identity() {
  Matrix3d m;
  m(0, 0) = m(1, 1) = m(2, 2) = 1.0;
  return m;
}
Matrix3d Matrix3d::transpose() const { return detail::from_host(kmc_host::transpose(detail::to_host(*this))); }
Matrix3d Matrix3d::inverse() const {
  kmc_host::Mat3 inv;
  if (!kmc_host::inverse(detail::to_host(*this), &inv)) {
    for (auto& row : inv.m)
      for (double& x : row) x = std::nan("");
  }
  return detail::from_host(inv);
}
double Matrix3d::determinant() const { return kmc_host::det(detail::to_host(*this)); }
double Matrix3d::sum() const {
  double s = 0;
  for (auto const& row : m)
    for (double x : row) s += x;
  return s;
}
Matrix3d operator*(Matrix3d const& a, Matrix3d const& b) { return detail::from_host(detail::to_host(a) * detail::to_host(b)); }
Vector3d operator*(Matrix3d const& a, Vector3d const& b) { return detail::from_host(detail::to_host(a) * detail::to_host(b)); }
Matrix3d operator*(double s, Matrix3d const& a) {
  Matrix3d r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r(i, j) = s * a(i, j);
  return r;
}
Matrix3d operator+(Matrix3d const& a, Matrix3d const& b) {
  Matrix3d r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r(i, j) = a(i, j) + b(i, j);
  return r;
}
Matrix3d operator-(Matrix3d const& a, Matrix3d const& b) {
  Matrix3d r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r(i, j) = a(i, j) - b(i, j);
  return r;
}

double Matrix4d::sum() const {
  double s = 0;
  for (auto const& row : m)
    for (double x : row) s += x;
  return s;
}
Matrix4d operator-(Matrix4d const& a, Matrix4d const& b) {
  Matrix4d r;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) r(i, j) = a(i, j) - b(i, j);
  return r;
}

Matrix3d AngleAxisd::toRotationMatrix() const {
  double const n = axis.norm();
  // exact Rodrigues at every angle (no first-order shortcut: this builds test poses, it is not lie::Exp)
  double const c = std::cos(angle), s = std::sin(angle), v = 1.0 - c;
  Matrix3d R = Matrix3d::Identity();
  if (n > 0) {
    double const ax = axis(0) / n, ay = axis(1) / n, az = axis(2) / n;
    R(0, 0) = c + v * ax * ax;      R(0, 1) = v * ax * ay - s * az; R(0, 2) = v * ax * az + s * ay;
    R(1, 0) = v * ay * ax + s * az; R(1, 1) = c + v * ay * ay;      R(1, 2) = v * ay * az - s * ax;
    R(2, 0) = v * az * ax - s * ay; R(2, 1) = v * az * ay + s * ax; R(2, 2) = c + v * az * az;
  }
  return R;
}

Matrix3d Affine3d::rotation() const {
  kmc_host::Mat3 R;
  if (!kmc_host::polar_rotation(detail::to_host(linear_), &R)) return linear_;
  return detail::from_host(R);
}

Affine3d Affine3d::inverse() const {
  kmc_host::Pose inv;
  if (!kmc_host::inverse(detail::to_host(*this), &inv)) {
    Affine3d bad;
    for (int i = 0; i < 3; ++i) {
      bad.translation()(i) = std::nan("");
      for (int j = 0; j < 3; ++j) bad.linear()(i, j) = std::nan("");
    }
    return bad;
  }
  return detail::from_host(inv);
}

Affine3d& Affine3d::rotate(AngleAxisd const& aa) {
  linear_ = linear_ * aa.toRotationMatrix();
  return *this;
}

Matrix4d Affine3d::matrix() const {
  Matrix4d M;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) M(i, j) = linear_(i, j);
    M(i, 3) = translation_(i);
  }
  M(3, 3) = 1.0;
  return M;
}

void Affine3d::to_rt12(double out[12]) const { detail::to_host(*this).to_rt12(out); }
Affine3d Affine3d::from_rt12(const double in[12]) { return detail::from_host(kmc_host::Pose::from_rt12(in)); }

Affine3d operator*(Affine3d const& a, Affine3d const& b) { return detail::from_host(detail::to_host(a) * detail::to_host(b)); }

Vector4d operator*(Affine3d const& a, Vector4d const& p) {
  Matrix3d const& L = a.linear();
  Vector3d const& t = a.translation();
  return {L(0, 0) * p(0) + L(0, 1) * p(1) + L(0, 2) * p(2) + t(0) * p(3), L(1, 0) * p(0) + L(1, 1) * p(1) + L(1, 2) * p(2) + t(1) * p(3),
          L(2, 0) * p(0) + L(2, 1) * p(1) + L(2, 2) * p(2) + t(2) * p(3), p(3)};
}

Affine3d operator*(Matrix3d const& r, Affine3d const& a) {
  Affine3d out;
  out.linear() = r * a.linear();
  out.translation() = r * a.translation();
  return out;
}

}  // namespace kmc

// kmc_api_math.cpp -- kmc::lie, kmc::trajectory_interpolation and the scalar timestamp helpers (host, f64).
// Thin adapters over csrc/kmc_host_math.hpp; each function names the reference function it stands in for.
#include <cmath>

#include "kitti_motion_compensation/data_io.hpp"
#include "kitti_motion_compensation/lie_algebra.hpp"
#include "kitti_motion_compensation/timestamp_mocking.hpp"
#include "kitti_motion_compensation/trajectory_interpolation.hpp"
#include "kmc_api_internal.hpp"

namespace kmc::lie {

Matrix3d Hat(Vector3d const& a) {  // lie_algebra.cpp:7-18
  Matrix3d H;
  H(0, 1) = -a(2); H(0, 2) = a(1);
  H(1, 0) = a(2);  H(1, 2) = -a(0);
  H(2, 0) = -a(1); H(2, 1) = a(0);
  return H;
}

Vector3d Vee(Matrix3d const& a) { return {a(2, 1), a(0, 2), a(1, 0)}; }  // :20

Matrix3d Exp(Vector3d const& phi) { return detail::from_host(kmc_host::so3_exp(detail::to_host(phi))); }  // :22-35

Vector3d Log(Matrix3d const& R) { return detail::from_host(kmc_host::so3_log(detail::to_host(R))); }  // :37-49

Matrix3d LeftJacobian(Vector3d const& phi) {  // :51-65, assembled column by column from J * e_k
  kmc_host::Vec3 const p = detail::to_host(phi);
  Matrix3d J;
  for (int k = 0; k < 3; ++k) {
    kmc_host::Vec3 e{k == 0 ? 1.0 : 0.0, k == 1 ? 1.0 : 0.0, k == 2 ? 1.0 : 0.0};
    kmc_host::Vec3 const c = kmc_host::left_jacobian_apply(p, e);
    J(0, k) = c.x; J(1, k) = c.y; J(2, k) = c.z;
  }
  return J;
}

Matrix3d InverseLeftJacobian(Vector3d const& phi) {  // :67-81
  kmc_host::Vec3 const p = detail::to_host(phi);
  Matrix3d J;
  for (int k = 0; k < 3; ++k) {
    kmc_host::Vec3 e{k == 0 ? 1.0 : 0.0, k == 1 ? 1.0 : 0.0, k == 2 ? 1.0 : 0.0};
    kmc_host::Vec3 const c = kmc_host::inv_left_jacobian_apply(p, e);
    J(0, k) = c.x; J(1, k) = c.y; J(2, k) = c.z;
  }
  return J;
}

Affine3d Exp(Twist const& xi) {  // :83-92
  return detail::from_host(kmc_host::se3_exp({detail::to_host(xi.rho()), detail::to_host(xi.phi())}));
}

Twist Log(Affine3d const& T) {  // :94-103
  kmc_host::Twist f;
  Twist out;
  if (!kmc_host::se3_log(detail::to_host(T), &f)) {
    for (int i = 0; i < 6; ++i) out(i) = std::nan("");
    return out;
  }
  out(0) = f.rho.x; out(1) = f.rho.y; out(2) = f.rho.z;
  out(3) = f.phi.x; out(4) = f.phi.y; out(5) = f.phi.z;
  return out;
}

}  // namespace kmc::lie

namespace kmc::trajectory_interpolation {

TrajectoryInterpolator::TrajectoryInterpolator(Oxts const& odometry_0, Oxts const& odometry_1)  // .cpp:21-25
    : time_1_{odometry_0.stamp}, pose_1_{OxtsToPose(odometry_0)}, time_2_{odometry_1.stamp}, pose_2_{OxtsToPose(odometry_1)} {}

TrajectoryInterpolator::TrajectoryInterpolator(Time const time_1, Affine3d const& pose_1, Time const time_2,
                                               Affine3d const& pose_2)  // :27-29
    : time_1_{time_1}, pose_1_{pose_1}, time_2_{time_2}, pose_2_{pose_2} {}

bool TrajectoryInterpolator::TimeIsInRange(Time const time) const { return (time >= time_1_) and (time <= time_2_); }  // :47

double TrajectoryInterpolator::FractionOfTrajectory(Time const time) const {  // :49-51
  return (time - time_1_) / (time_2_ - time_1_);
}

Affine3d TrajectoryInterpolator::GetPoseAtTime(Time const time) const {  // :31-41
  if (!TimeIsInRange(time)) detail::die_time_out_of_range("kmc::trajectory_interpolation::TrajectoryInterpolator::GetPoseAtTime");
  kmc_host::Pose P;
  int const rc = kmc_host::pose_at_time(time_1_, detail::to_host(pose_1_), time_2_, detail::to_host(pose_2_), time, &P);
  if (rc != 0) throw std::runtime_error("kmc::GetPoseAtTime: singular pose");
  return detail::from_host(P);
}

Affine3d TrajectoryInterpolator::RelativePoseBetweenTimes(Time const anchor_time, Time const query_time) const {  // :43-45
  return GetPoseAtTime(anchor_time).inverse() * GetPoseAtTime(query_time);
}

Affine3d InterpolateTrajectory(Oxts const& odometry_1, Oxts const& odometry_2, Time const time) {  // :14-19
  return TrajectoryInterpolator(odometry_1, odometry_2).GetPoseAtTime(time);
}

}  // namespace kmc::trajectory_interpolation

namespace kmc {

Affine3d OxtsToPose(Oxts const& o, double const scale) {  // data_io.cpp:68-88
  return detail::from_host(kmc_host::oxts_to_pose(o.lat, o.lon, o.alt, o.roll, o.pitch, o.yaw, scale));
}

double FractionOfScanCompleted(Vector4d const point) {  // timestamp_mocking.cpp:46
  constexpr double kPi = 3.14159265358979323846;
  return (kPi - std::atan2(point(1), point(0))) / (2.0 * kPi);
}

Time GetPseudoTimeStamp(Vector4d const point, Time const scan_start, Time const scan_end) {  // :49-54
  return scan_start + (FractionOfScanCompleted(point) * (scan_end - scan_start));
}

}  // namespace kmc

// host_math.hpp -- the host-side (f64) half of the drop-in API in one header: kmc::lie, the two-pose
// TrajectoryInterpolator, the pseudo-timestamp helpers and the small string utilities of the KITTI readers.
//
// The reference spreads these over lie_algebra.hpp, trajectory_interpolation.hpp, timestamp_mocking.hpp and utils.hpp;
// headers of those names still exist here and simply include this file, so `#include "kitti_motion_compensation/<name>.hpp"`
// keeps working for code written against the reference.  Reference lines are cited per declaration.
#pragma once

#include <string>
#include <vector>

#include "kitti_motion_compensation/data_types.hpp"

// ---- SO(3) / SE(3) exponentials and logarithms (reference include/.../lie_algebra.hpp:12-26) ---------------------
// Double precision on the host; used by the once-per-frame pre-step (one Log per frame).  The per-point Exp runs on the GPU.
namespace kmc::lie {

Matrix3d Hat(Vector3d const& a);                      // lie_algebra.cpp:7-18
Vector3d Vee(Matrix3d const& a);                      // :20
Matrix3d Exp(Vector3d const& phi);                    // :22-35
Vector3d Log(Matrix3d const& R);                      // :37-49
Matrix3d LeftJacobian(Vector3d const& phi);           // :51-65
Matrix3d InverseLeftJacobian(Vector3d const& phi);    // :67-81
Affine3d Exp(Twist const& xi);                        // :83-92
Twist Log(Affine3d const& T);                         // :94-103

}  // namespace kmc::lie

// ---- geodesic interpolation between two stamped poses (reference include/.../trajectory_interpolation.hpp:9-31) ----
namespace kmc::trajectory_interpolation {

Affine3d InterpolateTrajectory(Oxts const& odometry_1, Oxts const& odometry_2, Time const time);  // .cpp:14-19

class TrajectoryInterpolator {
 public:
  TrajectoryInterpolator(Oxts const& odometry_1, Oxts const& odometry_2);                                    // .cpp:21-25
  TrajectoryInterpolator(Time const time_1, Affine3d const& pose_1, Time const time_2, Affine3d const& pose_2);  // :27-29

  // pose_1 * Exp(x * Log(pose_1^-1 * pose_2)); ABORTS (release builds too) when `time` is outside
  // [time_1, time_2] -- the reference keeps its assert with #undef NDEBUG (.cpp:9, :32).
  Affine3d GetPoseAtTime(Time const time) const;                                      // :31-41
  Affine3d RelativePoseBetweenTimes(Time const anchor_time, Time const query_time) const;  // :43-45

  // accessors the device path needs (not in the reference, which only reads these privately)
  Time time_1() const { return time_1_; }
  Time time_2() const { return time_2_; }
  Affine3d const& pose_1() const { return pose_1_; }
  Affine3d const& pose_2() const { return pose_2_; }

 private:
  bool TimeIsInRange(Time const time) const;         // :47
  double FractionOfTrajectory(Time const time) const;  // :49-51

  Time time_1_;
  Affine3d pose_1_;
  Time time_2_;
  Affine3d pose_2_;
};

}  // namespace kmc::trajectory_interpolation

// ---- per-point pseudo timestamps from the azimuth (reference include/.../timestamp_mocking.hpp:7-11) --------------
namespace kmc {

double FractionOfScanCompleted(Vector4d const point);                                   // timestamp_mocking.cpp:6-47
Time GetPseudoTimeStamp(Vector4d const point, Time const scan_start, Time const scan_end);  // :49-54
// Runs on the GPU (kmc_hip_pseudo_timestamps_f64): one lane per point, f64 atan2.
VectorXd GetPseudoTimeStamps(Pointcloud const& cloud, Time const start_time, Time const end_time);  // :56-63

}  // namespace kmc

// ---- string helpers of the KITTI readers (reference include/.../utils.hpp, src/.../utils.cpp:10-38) ---------------
namespace kmc {

std::string IdToZeroPaddedString(std::size_t const id, std::size_t const pad = 10);  // utils.cpp:10-15
std::vector<std::string> TokenizeString(std::string raw_string);                     // :17-29
double MmHhSsToSeconds(std::string const mm_hh_ss);                                  // :31-38

}  // namespace kmc

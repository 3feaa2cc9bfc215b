// trajectory_interpolation.hpp -- geodesic interpolation between two stamped poses: the reference's
// include/kitti_motion_compensation/trajectory_interpolation.hpp:9-31.  Host code (f64); the hot path uses it once per frame,
// the per-point half of GetPoseAtTime runs on the GPU.
#pragma once

#include "kitti_motion_compensation/data_types.hpp"

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

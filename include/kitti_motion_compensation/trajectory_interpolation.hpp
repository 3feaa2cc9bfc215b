// trajectory_interpolation.hpp -- geodesic interpolation between two stamped poses: the reference's
// include/kitti_motion_compensation/trajectory_interpolation.hpp:9-31.  Host code (f64); the hot path uses it once per frame,
// the per-point half of GetPoseAtTime runs on the GPU.
//
// How the GPU path uses this class: MotionCompensateFrame builds nothing per point.  The twist xi = Log(pose_1^-1 * pose_2) is taken
// ONCE per frame on the host (kmc_host_math.hpp, f64), and because GetPoseAtTime(a)^-1 * GetPoseAtTime(q) = Exp((x_q - x_a) * xi) for
// poses on one geodesic, a lane evaluates a single exponential at the point's own fraction (closed form: Rodrigues + the left Jacobian,
// kmc_device_math.hip.h).  The class below is the reference's host-side object with the same names, argument meaning and failure mode;
// RelativePoseBetweenTimes(anchor, query) returns T(anchor)^-1 * T(query) exactly as the reference composes it (two GetPoseAtTime
// calls), which is also what the oracle's FAITHFUL mode does for every point.
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

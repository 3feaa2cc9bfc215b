// trajectory_interpolation.hpp -- kept so that `#include "kitti_motion_compensation/trajectory_interpolation.hpp"` written against the reference still resolves.
// The declarations (kmc::trajectory_interpolation::TrajectoryInterpolator, InterpolateTrajectory) live in host_math.hpp.
#pragma once
#include "kitti_motion_compensation/host_math.hpp"

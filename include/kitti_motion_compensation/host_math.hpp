// host_math.hpp -- umbrella over the host-side (f64) half of the drop-in API: kmc::lie, the two-pose TrajectoryInterpolator, the
// pseudo-timestamp helpers and the string utilities of the KITTI readers.  The declarations live in the headers the reference
// itself uses (lie_algebra.hpp, trajectory_interpolation.hpp, timestamp_mocking.hpp, utils.hpp), reference lines cited per
// declaration; this file only pulls them in together.
#pragma once

#include "kitti_motion_compensation/lie_algebra.hpp"
#include "kitti_motion_compensation/timestamp_mocking.hpp"
#include "kitti_motion_compensation/trajectory_interpolation.hpp"
#include "kitti_motion_compensation/utils.hpp"

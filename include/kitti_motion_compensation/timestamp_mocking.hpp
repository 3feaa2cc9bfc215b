// timestamp_mocking.hpp -- per-point pseudo timestamps from the azimuth (include/.../timestamp_mocking.hpp:7-11).
#pragma once

#include "kitti_motion_compensation/data_types.hpp"

namespace kmc {

double FractionOfScanCompleted(Vector4d const point);                                   // timestamp_mocking.cpp:6-47
Time GetPseudoTimeStamp(Vector4d const point, Time const scan_start, Time const scan_end);  // :49-54
// Runs on the GPU (kmc_hip_pseudo_timestamps_f64): one lane per point, f64 atan2.
VectorXd GetPseudoTimeStamps(Pointcloud const& cloud, Time const start_time, Time const end_time);  // :56-63

}  // namespace kmc

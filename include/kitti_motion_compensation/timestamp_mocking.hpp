// timestamp_mocking.hpp -- per-point pseudo timestamps from the azimuth: the reference's
// include/kitti_motion_compensation/timestamp_mocking.hpp:7-11.  The f32 kernels fuse this step (frac = (pi - atan2(y, x)) / 2 pi
// is evaluated per lane from the point they have just loaded); GetPseudoTimeStamps is the stand-alone form the loaders call.
#pragma once

#include "kitti_motion_compensation/data_types.hpp"

namespace kmc {

double FractionOfScanCompleted(Vector4d const point);                                   // timestamp_mocking.cpp:6-47
Time GetPseudoTimeStamp(Vector4d const point, Time const scan_start, Time const scan_end);  // :49-54
// Runs on the GPU (kmc_hip_pseudo_timestamps_f64): two points per lane, f64 atan2, every operation rounded like the reference's
// build (a stamp at the scan seam must land on the same side of the t <= t_end assert).
VectorXd GetPseudoTimeStamps(Pointcloud const& cloud, Time const start_time, Time const end_time);  // :56-63

}  // namespace kmc

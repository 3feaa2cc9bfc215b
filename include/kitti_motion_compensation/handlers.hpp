// handlers.hpp -- the per-run driver, the hot path's only production caller (reference include/.../handlers.hpp:7-11,
// src/.../handlers.cpp:15-65).  The visualization handler (OpenCV) is out of scope.
#pragma once

#include "kitti_motion_compensation/data_types.hpp"

namespace kmc {

std::size_t NumberOfFilesInDirectory(std::filesystem::path path);  // handlers.cpp:15-17

// handlers.cpp:19-39.  The reference writes the LAST frame's output from the FIRST frame's data (:36-38, a copy-paste
// slip); this writes the last frame's own data and documents the divergence (DESIGN.md "N3").
void CopyOverUncompensatedFirstAndLastFrame(Path const run_folder);

// handlers.cpp:41-65: frames 1 .. n-2 are deskewed to the scan's middle stamp and written to
// velodyne_points/data_motion_compensated/; first and last are copied through.  Frames are batched on the GPU
// (kmc_hip_deskew_batch_f32) straight from / to the on-disk f32 layout.
void MotionCompensateRun(Path const run_folder);

}  // namespace kmc

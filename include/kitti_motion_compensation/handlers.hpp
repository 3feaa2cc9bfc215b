// handlers.hpp -- the per-run driver, the hot path's only production caller (reference include/.../handlers.hpp:7-11,
// src/.../handlers.cpp:15-65).  The visualization handler draws with OpenCV and is out of scope; its arithmetic is in camera_model.hpp.
#pragma once

#include "kitti_motion_compensation/data_types.hpp"

namespace kmc {

std::size_t NumberOfFilesInDirectory(std::filesystem::path path);  // handlers.cpp:15-17

// handlers.cpp:19-39.  The reference writes the LAST frame's output from the FIRST frame's data (:36-38, a copy-paste
// slip).  By default this reproduces the reference's output exactly, slip included; KMC_FIX_LAST_FRAME_COPY=1 writes the last
// frame's own data instead.
void CopyOverUncompensatedFirstAndLastFrame(Path const run_folder);

// handlers.cpp:41-65: frames 1 .. n-2 are deskewed to the scan's middle stamp and written to
// velodyne_points/data_motion_compensated/; first and last are copied through.  Frames are batched on the GPU
// (kmc_hip_deskew_batch_f32) straight from / to the on-disk f32 layout, with reading, GPU work and writing overlapped.
// Several GPUs: KMC_DEVICES=0,1,... or kmc::hip::SetRunDevices() -- the frames are cut into one contiguous, point-balanced
// range per device (the frames of a run are independent), every range on its own device context; the files written are
// the same whatever the device list.
// Environment: KMC_RUN_BATCH_FRAMES (frames per GPU batch, default 8), KMC_RUN_KNOTS=3 (use the three OXTS poses around each
// frame as they are), KMC_RUN_TIMING=1 (busy time per stage on stderr), KMC_DEVICES (device list).
void MotionCompensateRun(Path const run_folder);

}  // namespace kmc

// motion_compensation.hpp -- THE drop-in: same two functions, same signatures, same semantics as the reference
// (include/kitti_motion_compensation/motion_compensation.hpp:10-13), executed on an MI355X through libkmc_hip.so.
//
//   * MotionCompensateFrame(frame, t): per point i, row(i) = RelativePoseBetweenTimes(t, timestamps(i)) * cloud.row(i)
//     (motion_compensation.cpp:16-28).  The caller's per-point stamps are honoured (f64 device kernel); the result is a
//     fresh N x 4 f64 cloud returned by value; inputs are never mutated; pure and re-entrant (one device context per thread).
//   * errors: the reference aborts (assert kept in release, trajectory_interpolation.cpp:9,:32) when requested_time or
//     any point stamp lies outside [stamp_start, stamp_end].  So does this: message on stderr, then std::abort().
//   * no GPU -> std::runtime_error.  There is NO CPU fallback for the frame path.
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

#include "kitti_motion_compensation/data_types.hpp"
#include "kitti_motion_compensation/trajectory_interpolation.hpp"

namespace kmc {

using TrajectoryInterpolator = trajectory_interpolation::TrajectoryInterpolator;

Vector4d MotionCompensatePoint(TrajectoryInterpolator const& trajectory_interpolator, Time const point_stamp,
                               Vector4d const& point, Time const requested_time);

Pointcloud MotionCompensateFrame(Frame const& frame, Time const requested_time);

// ---------------------------------------------------------------------------------------------------------------
// The 3-argument form BASELINE.json's north_star names.  NOT in the reference (which has no Trajectory type and always
// interpolates along the single geodesic T_start -> T_end): a piecewise SE(3) geodesic through time-stamped poses, e.g.
// the three bracketing OXTS poses used directly instead of being reduced to two by MakeFrame.
//   T(t) = P_k * Exp(x * Log(P_k^-1 P_{k+1})) on [t_k, t_{k+1}];   row(i) = T(requested_time)^-1 * T(timestamps(i)) * cloud.row(i)
// Trajectory{{stamp_start, stamp_end}, {T_start, T_end}} gives results BIT-IDENTICAL to the 2-argument function.
// Aborts like the reference's assert when requested_time or a point stamp is outside [times.front(), times.back()].
// ---------------------------------------------------------------------------------------------------------------
struct Trajectory {
  std::vector<Time> times;      // strictly increasing, 2 .. 17 knots
  std::vector<Affine3d> poses;  // pose at each knot
};
Pointcloud MotionCompensateFrame(Frame const& frame, Trajectory const& trajectory, Time const requested_time);

// ---------------------------------------------------------------------------------------------------------------
// Extensions (not in the reference): the f32 KITTI-layout fast path that the roofline numbers are quoted on.
// ---------------------------------------------------------------------------------------------------------------
namespace hip {

// Which HIP device the calling thread's context uses (default 0 or $KMC_DEVICE).  Must be called before the first
// deskew call of the thread.
void SetDevice(int device_id);
int GetDevice();

// Runs the calling thread on the CPUs of its device's NUMA node (what numactl / taskset do for a deployment; never done implicitly):
// on a two-socket box the host side of every call crosses the inter-socket link otherwise -- 104-125 instead of 92 us per in-place
// KITTI frame.  (The page-locked containers themselves are placed on the device's node by the pool.)  false: the topology is unknown.
bool BindThreadNearDevice();

// Devices of kmc::MotionCompensateRun (handlers.hpp): one worker with its own device context per entry, each deskewing one
// contiguous, point-balanced range of the run's frames.  An id may repeat (several contexts on one GPU).  Empty list =
// back to the default: $KMC_DEVICES ("0,1,2,..."), else the calling thread's device.  Process-wide setting.
void SetRunDevices(std::vector<int> const& devices);
std::vector<int> GetRunDevices();

// Where the microseconds of the calling thread's last MotionCompensateFrame(Frame const&, Time) call went (profiles/r05_dropin_trace.json
// is made of these).  Host stamps: steady_clock microseconds; device stamps: the GPU's own 100 MHz clock in microseconds (only their
// difference means anything; 0 when the call took a route without stamps, e.g. pageable containers).  Off by default.
struct FrameTrace {
  double enter_us = 0;         // MotionCompensateFrame entered
  double params_us = 0;        // f64 host pre-step done: Log(T_start^-1 T_end), x_req
  double alloc_us = 0;         // the result cloud exists (page-locked pool block)
  double begin_returned_us = 0;  // kmc_hip_deskew_f64cols_begin has returned: the kernel is enqueued
  double fill_done_us = 0;     // the host has filled the result's homogeneous column (while the kernel runs)
  double end_returned_us = 0;  // kmc_hip_deskew_f64cols_end has returned: the results are in host memory
  double return_us = 0;        // about to return the cloud
  // the C-ABI's own stamps of the same call (kmc_call_trace, kmc_hip.h)
  double issue_begin_us = 0, issue_end_us = 0, wait_begin_us = 0, wait_end_us = 0, dev_first_wave_us = 0, dev_last_store_us = 0;
  unsigned waves = 0, route = 0;
};
void EnableFrameTrace(bool enabled);
FrameTrace LastFrameTrace();
// In-place calls of this PROCESS that ended on a stream synchronisation because the kernel's completion word had not come (kmc_hip.h,
// kmc_hip_completion_word_fallbacks): 0 normally; such a call costs ~0.1 ms more and returns the same cloud.
unsigned long long CompletionWordFallbacks(unsigned last_state[3] = nullptr);  // last_state: {sequence number expected, word seen, ticket seen} of the last event

// One frame in the on-disk KITTI layout (f32 AoS x,y,z,intensity): fuses GetPseudoTimeStamps (data_io.cpp:163),
// MotionCompensateFrame (handlers.cpp:60) and WritePointcloud's f64->f32 cast (data_io.cpp:300-310) in one kernel.
// xyzi_in / xyzi_out are HOST pointers here (16-byte aligned); use the C-ABI directly for device-resident buffers.
void MotionCompensateKittiCloud(float const* xyzi_in, std::size_t num_points, Affine3d const& T_start, Affine3d const& T_end,
                                Time stamp_start, Time stamp_end, Time requested_time, float* xyzi_out);

// Same on an N-knot trajectory; bracket_index_out (optional) receives each point's segment index.
void MotionCompensateKittiCloud(float const* xyzi_in, std::size_t num_points, Trajectory const& trajectory, Time stamp_start,
                                Time stamp_end, Time requested_time, float* xyzi_out, std::uint32_t* bracket_index_out = nullptr);

// Many frames in one launch; frame f owns points [offsets[f], offsets[f+1]).
struct FramePoses {
  Affine3d T_start, T_end;
  Time stamp_start, stamp_end, requested_time;
};
void MotionCompensateKittiClouds(float const* xyzi_in, std::vector<std::uint64_t> const& offsets,
                                 std::vector<FramePoses> const& frames, float* xyzi_out,
                                 std::uint32_t* frame_index_out = nullptr);

// Many frames in one launch, every frame with its OWN trajectory (e.g. the three OXTS poses around it, used as they are).
struct FrameTrajectory {
  Trajectory trajectory;  // must cover [stamp_start, stamp_end]
  Time stamp_start, stamp_end, requested_time;
};
void MotionCompensateKittiClouds(float const* xyzi_in, std::vector<std::uint64_t> const& offsets,
                                 std::vector<FrameTrajectory> const& frames, float* xyzi_out,
                                 std::uint32_t* frame_index_out = nullptr, std::uint32_t* bracket_index_out = nullptr);

}  // namespace hip
}  // namespace kmc

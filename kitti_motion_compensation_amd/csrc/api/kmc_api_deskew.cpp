// kmc_api_deskew.cpp -- kmc::MotionCompensateFrame / MotionCompensatePoint / GetPseudoTimeStamps on the GPU.
//
// Everything per-point goes through the C-ABI (libkmc_hip.so).  No CPU fallback: without a HIP device the first call
// throws std::runtime_error.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "kitti_motion_compensation/motion_compensation.hpp"
#include "kitti_motion_compensation/timestamp_mocking.hpp"
#include "kmc_api_internal.hpp"

namespace kmc {
namespace detail {

namespace {
struct CtxDeleter {
  void operator()(kmc_ctx* c) const { kmc_hip_destroy(c); }
};
thread_local std::unique_ptr<kmc_ctx, CtxDeleter> t_ctx;
thread_local int t_device = -1;
thread_local bool t_trace = false;
thread_local hip::FrameTrace t_last_trace;
double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int default_device() {
  if (const char* e = std::getenv("KMC_DEVICE")) return std::atoi(e);
  return 0;
}
}  // namespace

kmc_ctx* thread_context() {
  if (!t_ctx) {
    if (t_device < 0) t_device = default_device();
    kmc_ctx* c = nullptr;
    int const rc = kmc_hip_create(&c, t_device);
    if (rc != KMC_OK) throw_status(rc, "kmc: cannot create the HIP deskew context");
    t_ctx.reset(c);
  }
  return t_ctx.get();
}

// A context made on ANOTHER thread (kmc::MotionCompensateRun creates it on a helper while the calling thread parses the run's text
// files: creating one costs 12-40 ms) becomes the calling thread's context.  The thread already has one: `c` is destroyed instead.
void adopt_thread_context(kmc_ctx* c, int device) {
  if (!c) return;
  if (t_ctx) {
    kmc_hip_destroy(c);
    return;
  }
  t_device = device;
  t_ctx.reset(c);
}
bool thread_has_context() { return static_cast<bool>(t_ctx); }

static kmc_frame_params frame_params(Affine3d const& T_start, Affine3d const& T_end, Time t0, Time t1, Time t_req, const char* where) {
  double a[12], b[12];
  T_start.to_rt12(a);
  T_end.to_rt12(b);
  kmc_frame_params p;
  int const rc = kmc_frame_params_from_poses(a, b, t0, t1, t_req, &p);
  if (rc == KMC_ERR_TIME_OUT_OF_RANGE) die_time_out_of_range(where);  // the reference asserts on requested_time for every point
  if (rc != KMC_OK) throw_status(rc, where);
  return p;
}

}  // namespace detail

namespace hip {

void SetDevice(int device_id) {
  if (detail::t_ctx && detail::t_device != device_id) detail::t_ctx.reset();
  detail::t_device = device_id;
}

int GetDevice() { return detail::t_device < 0 ? detail::default_device() : detail::t_device; }

bool BindThreadNearDevice() { return kmc_hip_bind_thread_near_device(GetDevice()) == KMC_OK; }

void EnableFrameTrace(bool enabled) {
  detail::t_trace = enabled;
  detail::t_last_trace = FrameTrace{};
  if (kmc_ctx* c = detail::thread_context()) (void)kmc_hip_enable_call_trace(c, enabled ? 1 : 0);
}
FrameTrace LastFrameTrace() { return detail::t_last_trace; }
unsigned long long CompletionWordFallbacks(unsigned last_state[3]) {
  std::uint32_t st[3] = {0, 0, 0};
  unsigned long long const count = kmc_hip_completion_word_fallbacks(nullptr, st);
  if (last_state)
    for (int i = 0; i < 3; ++i) last_state[i] = st[i];
  return count;
}

void MotionCompensateKittiCloud(float const* xyzi_in, std::size_t n, Affine3d const& T_start, Affine3d const& T_end, Time stamp_start,
                                Time stamp_end, Time requested_time, float* xyzi_out) {
  kmc_frame_params const p = detail::frame_params(T_start, T_end, stamp_start, stamp_end, requested_time, "kmc::hip::MotionCompensateKittiCloud");
  kmc_ctx* c = detail::thread_context();
  int const rc = kmc_hip_deskew_f32(c, xyzi_in, xyzi_out, n, &p, KMC_MEM_HOST, nullptr);
  if (rc != KMC_OK) detail::throw_status(rc, "kmc_hip_deskew_f32", c);
}

namespace {
void flatten(Trajectory const& tr, std::vector<double>* poses12) {
  if (tr.times.size() != tr.poses.size()) throw std::invalid_argument("kmc::Trajectory: times.size() != poses.size()");
  poses12->resize(12 * tr.poses.size());
  for (std::size_t k = 0; k < tr.poses.size(); ++k) tr.poses[k].to_rt12(poses12->data() + 12 * k);
}
}  // namespace

void MotionCompensateKittiCloud(float const* xyzi_in, std::size_t n, Trajectory const& tr, Time stamp_start, Time stamp_end,
                                Time requested_time, float* xyzi_out, std::uint32_t* bracket_index_out) {
  std::vector<double> poses;
  flatten(tr, &poses);
  kmc_ctx* c = detail::thread_context();
  int const rc = kmc_hip_deskew_traj_f32(c, xyzi_in, xyzi_out, n, tr.times.data(), poses.data(), static_cast<std::uint32_t>(tr.times.size()),
                                         stamp_start, stamp_end, requested_time, bracket_index_out, KMC_MEM_HOST, nullptr);
  if (rc == KMC_ERR_TIME_OUT_OF_RANGE) detail::die_time_out_of_range("kmc::hip::MotionCompensateKittiCloud(Trajectory)");
  if (rc != KMC_OK) detail::throw_status(rc, "kmc_hip_deskew_traj_f32", c);
}

void MotionCompensateKittiClouds(float const* xyzi_in, std::vector<std::uint64_t> const& offsets, std::vector<FramePoses> const& frames,
                                 float* xyzi_out, std::uint32_t* frame_index_out) {
  if (offsets.size() != frames.size() + 1) throw std::invalid_argument("kmc::hip::MotionCompensateKittiClouds: offsets.size() != frames.size() + 1");
  std::vector<kmc_frame_params> params(frames.size());
  for (std::size_t f = 0; f < frames.size(); ++f)
    params[f] = detail::frame_params(frames[f].T_start, frames[f].T_end, frames[f].stamp_start, frames[f].stamp_end, frames[f].requested_time,
                                     "kmc::hip::MotionCompensateKittiClouds");
  kmc_ctx* c = detail::thread_context();
  int const rc = kmc_hip_deskew_batch_f32(c, xyzi_in, xyzi_out, offsets.data(), static_cast<std::uint32_t>(frames.size()), params.data(),
                                          frame_index_out, KMC_MEM_HOST, nullptr);
  if (rc != KMC_OK) detail::throw_status(rc, "kmc_hip_deskew_batch_f32", c);
}

void MotionCompensateKittiClouds(float const* xyzi_in, std::vector<std::uint64_t> const& offsets,
                                 std::vector<FrameTrajectory> const& frames, float* xyzi_out, std::uint32_t* frame_index_out,
                                 std::uint32_t* bracket_index_out) {
  if (offsets.size() != frames.size() + 1) throw std::invalid_argument("kmc::hip::MotionCompensateKittiClouds: offsets.size() != frames.size() + 1");
  std::vector<std::vector<double>> poses(frames.size());
  std::vector<kmc_traj_frame> c_frames(frames.size());
  for (std::size_t f = 0; f < frames.size(); ++f) {
    Trajectory const& tr = frames[f].trajectory;
    if (tr.times.size() != tr.poses.size()) throw std::invalid_argument("kmc::Trajectory: times.size() != poses.size()");
    poses[f].resize(12 * tr.poses.size());
    for (std::size_t k = 0; k < tr.poses.size(); ++k) tr.poses[k].to_rt12(poses[f].data() + 12 * k);
    c_frames[f].knot_times = tr.times.data();
    c_frames[f].knot_poses = poses[f].data();
    c_frames[f].n_knots = static_cast<std::uint32_t>(tr.times.size());
    c_frames[f].reserved = 0;
    c_frames[f].stamp_start = frames[f].stamp_start;
    c_frames[f].stamp_end = frames[f].stamp_end;
    c_frames[f].requested_time = frames[f].requested_time;
  }
  kmc_ctx* c = detail::thread_context();
  int const rc = kmc_hip_deskew_traj_batch_f32(c, xyzi_in, xyzi_out, offsets.data(), static_cast<std::uint32_t>(frames.size()),
                                               c_frames.data(), frame_index_out, bracket_index_out, KMC_MEM_HOST, nullptr);
  if (rc == KMC_ERR_TIME_OUT_OF_RANGE) detail::die_time_out_of_range("kmc::hip::MotionCompensateKittiClouds");
  if (rc != KMC_OK) detail::throw_status(rc, "kmc_hip_deskew_traj_batch_f32", c);
}

}  // namespace hip

// motion_compensation.cpp:16-28
Pointcloud MotionCompensateFrame(Frame const& frame, Time const requested_time) {
  bool const trace = detail::t_trace;
  hip::FrameTrace tr;
  if (trace) tr.enter_us = detail::now_us();
  Index const n = frame.scan.cloud.rows();
  if (frame.scan.timestamps.size() != n) throw std::invalid_argument("kmc::MotionCompensateFrame: timestamps.size() != cloud.rows()");
  kmc_frame_params const p = detail::frame_params(frame.T_start, frame.T_end, frame.scan.stamp_start, frame.scan.stamp_end, requested_time,
                                                  "kmc::MotionCompensateFrame");
  if (trace) tr.params_us = detail::now_us();
  Pointcloud out{MatrixX4d::Uninitialized(n)};  // every element is written below
  if (n == 0) return out;
  kmc_ctx* c = detail::thread_context();
  if (trace) tr.alloc_us = detail::now_us();
  kmc_stats st;
  Pointcloud const& in = frame.scan.cloud;
  // A cloud whose homogeneous column is KNOWN to be all ones (every cloud the loaders produce, data_io.cpp:130): the device neither
  // reads the column nor writes it back -- Affine3d * (x, y, z, 1) needs no w, and the output's column is filled with ones here,
  // on the host, which costs a tenth of what the two transfers cost on the link.
  bool const ones = in.is_homogeneous();
  double* const ox = out.col(0);
  int rc = kmc_hip_deskew_f64cols_begin(c, in.col(0), in.col(1), in.col(2), ones ? nullptr : in.col(3), frame.scan.timestamps.data(),
                                        static_cast<std::uint64_t>(n), frame.scan.stamp_start, frame.scan.stamp_end, &p, ox, ox + n, ox + 2 * n,
                                        ones ? nullptr : ox + 3 * n, KMC_MEM_HOST);
  if (trace) tr.begin_returned_us = detail::now_us();
  if (rc == KMC_OK) {
    if (ones) std::fill(ox + 3 * n, ox + 4 * n, 1.0);  // while the kernel works on the other three columns
    if (trace) tr.fill_done_us = detail::now_us();
    rc = kmc_hip_deskew_f64cols_end(c, &st);
  }
  if (trace) tr.end_returned_us = detail::now_us();
  if (rc == KMC_ERR_TIME_OUT_OF_RANGE) detail::die_time_out_of_range("kmc::MotionCompensateFrame");
  if (rc != KMC_OK) detail::throw_status(rc, "kmc_hip_deskew_f64cols", c);
  detail::set_homogeneous(out, ones);
  if (trace) {
    kmc_call_trace ct;
    if (kmc_hip_last_call_trace(c, &ct) == KMC_OK) {
      tr.issue_begin_us = ct.issue_begin_us; tr.issue_end_us = ct.issue_end_us; tr.wait_begin_us = ct.wait_begin_us; tr.wait_end_us = ct.wait_end_us;
      tr.dev_first_wave_us = ct.dev_first_wave_us; tr.dev_last_store_us = ct.dev_last_store_us; tr.waves = ct.waves; tr.route = ct.route;
    }
    tr.return_us = detail::now_us();
    detail::t_last_trace = tr;
  }
  return out;
}

// north_star's 3-argument form (see the header): piecewise geodesic through the trajectory's knots
Pointcloud MotionCompensateFrame(Frame const& frame, Trajectory const& tr, Time const requested_time) {
  Index const n = frame.scan.cloud.rows();
  if (frame.scan.timestamps.size() != n) throw std::invalid_argument("kmc::MotionCompensateFrame: timestamps.size() != cloud.rows()");
  if (tr.times.size() != tr.poses.size()) throw std::invalid_argument("kmc::Trajectory: times.size() != poses.size()");
  std::vector<double> poses(12 * tr.poses.size());
  for (std::size_t k = 0; k < tr.poses.size(); ++k) tr.poses[k].to_rt12(poses.data() + 12 * k);
  Pointcloud out{MatrixX4d::Uninitialized(n)};  // every element is written by the download below
  kmc_ctx* c = detail::thread_context();
  Pointcloud const& in = frame.scan.cloud;
  // (a cloud whose homogeneous column is known to be all ones -- see the 2-argument form: the column stays off the link, the C-ABI
  // fills the result's column on the host while the kernel runs)
  bool const ones = in.is_homogeneous();
  int const rc = kmc_hip_deskew_traj_f64cols(c, in.col(0), in.col(1), in.col(2), ones ? nullptr : in.col(3), frame.scan.timestamps.data(),
                                             static_cast<std::uint64_t>(n), tr.times.data(), poses.data(),
                                             static_cast<std::uint32_t>(tr.times.size()), requested_time, out.col(0), out.col(1), out.col(2),
                                             out.col(3), nullptr, KMC_MEM_HOST, nullptr);
  if (rc == KMC_ERR_TIME_OUT_OF_RANGE) detail::die_time_out_of_range("kmc::MotionCompensateFrame(Frame, Trajectory, Time)");
  if (rc != KMC_OK) detail::throw_status(rc, "kmc_hip_deskew_traj_f64cols", c);
  detail::set_homogeneous(out, ones);
  return out;
}

// motion_compensation.cpp:9-14 -- one point: the same device kernel with n = 1 (latency-bound by construction; callers
// with many points should hand a Frame to MotionCompensateFrame, which is what the reference's own loop does).
Vector4d MotionCompensatePoint(TrajectoryInterpolator const& ti, Time const point_stamp, Vector4d const& point, Time const requested_time) {
  kmc_frame_params const p =
      detail::frame_params(ti.pose_1(), ti.pose_2(), ti.time_1(), ti.time_2(), requested_time, "kmc::MotionCompensatePoint");
  kmc_ctx* c = detail::thread_context();
  double const x = point(0), y = point(1), z = point(2), w = point(3), t = point_stamp;
  double ox, oy, oz, ow;
  int const rc = kmc_hip_deskew_f64cols(c, &x, &y, &z, &w, &t, 1, ti.time_1(), ti.time_2(), &p, &ox, &oy, &oz, &ow, KMC_MEM_HOST, nullptr);
  if (rc == KMC_ERR_TIME_OUT_OF_RANGE) detail::die_time_out_of_range("kmc::MotionCompensatePoint");
  if (rc != KMC_OK) detail::throw_status(rc, "kmc_hip_deskew_f64cols", c);
  return {ox, oy, oz, ow};
}

// timestamp_mocking.cpp:56-63
VectorXd GetPseudoTimeStamps(Pointcloud const& cloud, Time const start_time, Time const end_time) {
  VectorXd stamps{VectorXd::Uninitialized(cloud.rows())};
  if (cloud.rows() == 0) return stamps;
  kmc_ctx* c = detail::thread_context();
  int const rc = kmc_hip_pseudo_timestamps_f64(c, cloud.col(0), cloud.col(1), static_cast<std::uint64_t>(cloud.rows()), start_time, end_time,
                                               stamps.data(), KMC_MEM_HOST);
  if (rc != KMC_OK) detail::throw_status(rc, "kmc_hip_pseudo_timestamps_f64", c);
  return stamps;
}

}  // namespace kmc

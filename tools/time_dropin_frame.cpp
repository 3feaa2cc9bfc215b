// time_dropin_frame.cpp -- the literal drop-in on the clock: kmc::MotionCompensateFrame(Frame const&, Time) (motion_compensation.cpp:16-28,
// the API north_star names) on the shipped KITTI frame -- f64 Eigen-layout cloud in the drop-in's own containers, result by value --
// and hip::MotionCompensateKittiCloud (the same frame in the KITTI f32 layout) next to it, through libkitti_motion_compensation_lib.so
// (C++, no ctypes).  The containers come from the C-ABI's page-locked pool, so the kernel works on them in place; KMC_HOST_POOL=0 in
// the environment gives ordinary pageable containers and the staged three-copy route.
//   time_dropin_frame <golden_dir> [iterations=200] [dump_prefix]
// Prints ONE JSON object.  With dump_prefix the frame's inputs and the clouds the two calls returned are written next to it
// (<prefix>.cloud_in.f64 / .stamps.f64 / .cloud_out.f64: column-major N x 4 / N doubles; <prefix>.kitti_in.f32 / .kitti_out.f32: N x 4
// floats), for bench.py's parity check against the oracle -- this tool itself never touches oracle/.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "kitti_motion_compensation/data_io.hpp"
#include "kitti_motion_compensation/motion_compensation.hpp"
#include "kmc_hip.h"

using namespace kmc;

static void dump(std::string const& path, void const* p, std::size_t bytes) {
  std::ofstream os{path, std::ios::binary};
  os.write(static_cast<char const*>(p), static_cast<std::streamsize>(bytes));
}

int main(int argc, char** argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: time_dropin_frame <golden_dir> [iterations] [dump_prefix]\n");
    return 2;
  }
  if (!std::getenv("KMC_TOOL_NO_BIND")) (void)kmc_hip_bind_thread_near_device(0);  // the calling thread on the GPU's NUMA node (what numactl does for a deployment)
  Path const run{std::string(argv[1]) + "/kitti_2011_09_26_drive_0005"};
  int const iters = argc > 2 ? std::atoi(argv[2]) : 200;
  std::string const prefix = argc > 3 ? argv[3] : "";
  LidarScan const scan{LoadLidarScan(run, 0)};
  double const yaw = 0.03, tx = 1.3, ty = 0.05, tz = -0.02;  // T_end = [Rz(yaw) | t]: a gentle turn at 13 m/s
  Affine3d T_end;
  T_end.rotate(AngleAxisd{yaw, Vector3d{0, 0, 1}});
  T_end.translation() = Vector3d{tx, ty, tz};
  Frame const frame{Affine3d::Identity(), T_end, scan};
  using clk = std::chrono::steady_clock;
  double checksum = 0;
  for (int i = 0; i < 10; ++i) checksum += MotionCompensateFrame(frame, scan.stamp_middle)(0, 0);
  std::vector<double> per_call(static_cast<std::size_t>(iters));
  auto t0 = clk::now();
  for (int i = 0; i < iters; ++i) {
    auto const a = clk::now();
    checksum += MotionCompensateFrame(frame, scan.stamp_middle)(0, 0);
    per_call[static_cast<std::size_t>(i)] = std::chrono::duration<double, std::micro>(clk::now() - a).count();
  }
  double const us64 = std::chrono::duration<double, std::micro>(clk::now() - t0).count() / iters;
  double us64_min = per_call.empty() ? 0.0 : per_call[0];
  for (double v : per_call) us64_min = std::min(us64_min, v);

  // the 3-argument form north_star names: three knots (start, middle, end of the scan), the frame's own stamps
  Trajectory traj;
  {
    Affine3d T_mid;
    T_mid.rotate(AngleAxisd{0.4 * yaw, Vector3d{0, 0, 1}});
    T_mid.translation() = Vector3d{0.48 * tx, 0.5 * ty, 0.5 * tz};
    traj.times = {scan.stamp_start, scan.stamp_middle, scan.stamp_end};
    traj.poses = {Affine3d::Identity(), T_mid, T_end};
  }
  for (int i = 0; i < 10; ++i) checksum += MotionCompensateFrame(frame, traj, scan.stamp_middle)(0, 0);
  t0 = clk::now();
  for (int i = 0; i < iters; ++i) checksum += MotionCompensateFrame(frame, traj, scan.stamp_middle)(0, 0);
  double const us64_traj = std::chrono::duration<double, std::micro>(clk::now() - t0).count() / iters;

  KittiCloudF32 const raw = KittiPclLoader::LoadRaw(run / "velodyne_points/data/0000000000.bin");  // page-locked when the pool is on
  KittiCloudF32 out(raw.size());
  std::size_t const n = raw.size() / 4;
  for (int i = 0; i < 10; ++i)
    hip::MotionCompensateKittiCloud(raw.data(), n, frame.T_start, frame.T_end, scan.stamp_start, scan.stamp_end, scan.stamp_middle, out.data());
  t0 = clk::now();
  for (int i = 0; i < iters; ++i)
    hip::MotionCompensateKittiCloud(raw.data(), n, frame.T_start, frame.T_end, scan.stamp_start, scan.stamp_end, scan.stamp_middle, out.data());
  double const us32 = std::chrono::duration<double, std::micro>(clk::now() - t0).count() / iters;
  bool const pooled = kmc_host_pool_owns(frame.scan.cloud.data(), sizeof(double)) != 0;
  if (!prefix.empty()) {
    Pointcloud const result = MotionCompensateFrame(frame, scan.stamp_middle);
    auto const rows = static_cast<std::size_t>(result.rows());
    dump(prefix + ".cloud_in.f64", frame.scan.cloud.data(), rows * 4 * sizeof(double));
    dump(prefix + ".stamps.f64", frame.scan.timestamps.data(), rows * sizeof(double));
    dump(prefix + ".cloud_out.f64", result.data(), rows * 4 * sizeof(double));
    dump(prefix + ".kitti_in.f32", raw.data(), raw.size() * sizeof(float));
    dump(prefix + ".kitti_out.f32", out.data(), out.size() * sizeof(float));
  }
  // ---- where the microseconds of one MotionCompensateFrame call go (hip::FrameTrace; medians over `iters` traced calls) ----
  std::string trace_json = "null";
  {
    hip::EnableFrameTrace(true);
    for (int i = 0; i < 10; ++i) checksum += MotionCompensateFrame(frame, scan.stamp_middle)(0, 0);
    struct Row { double v[12]; };
    std::vector<Row> rows;
    for (int i = 0; i < iters; ++i) {
      auto const a = std::chrono::duration<double, std::micro>(clk::now().time_since_epoch()).count();
      checksum += MotionCompensateFrame(frame, scan.stamp_middle)(0, 0);
      auto const b = std::chrono::duration<double, std::micro>(clk::now().time_since_epoch()).count();
      hip::FrameTrace const t = hip::LastFrameTrace();
      if (t.route != 2) continue;  // not the in-place route (pageable containers): no stage stamps
      rows.push_back(Row{{t.enter_us - a, t.params_us - t.enter_us, t.alloc_us - t.params_us, t.begin_returned_us - t.alloc_us, t.fill_done_us - t.begin_returned_us,
                          t.end_returned_us - t.fill_done_us, b - t.end_returned_us, b - a, t.issue_end_us - t.issue_begin_us, t.wait_end_us - t.issue_end_us,
                          t.dev_last_store_us - t.dev_first_wave_us, static_cast<double>(t.waves)}});
    }
    hip::EnableFrameTrace(false);
    if (!rows.empty()) {
      auto med = [&](int k) {
        std::vector<double> v;
        for (Row const& r : rows) v.push_back(r.v[k]);
        std::sort(v.begin(), v.end());
        return v[v.size() / 2];
      };
      char buf[2048];
      std::snprintf(buf, sizeof(buf),
                    "{\"calls_traced\": %zu, \"medians_us\": {\"call_overhead_before_entry\": %.2f, \"host_prestep_Log_and_x_req\": %.2f, \"result_cloud_from_the_pool_and_context_lookup\": %.2f, "
                    "\"f64cols_begin_checks_and_launch\": %.2f, \"host_fills_w_column_while_the_kernel_runs\": %.2f, \"f64cols_end_wait_for_the_completion_word\": %.2f, "
                    "\"return_by_value_and_destructor\": %.2f, \"whole_call\": %.2f}, \"inside_the_c_abi_us\": {\"begin_entry_to_launch_enqueued\": %.2f, "
                    "\"launch_enqueued_to_completion_word_seen\": %.2f, \"device_first_wave_to_last_store\": %.2f, \"persistent_waves\": %.0f}, "
                    "\"derived_us\": {\"launch_enqueued_to_first_wave_plus_last_store_to_word_seen\": %.2f}}",
                    rows.size(), med(0), med(1), med(2), med(3), med(4), med(5), med(6), med(7), med(8), med(9), med(10), med(11), med(9) - med(10));
      trace_json = buf;
    }
  }
  std::printf(
      "{\"points\": %zu, \"iterations\": %d, \"containers\": \"%s\", \"route\": \"%s\", "
      "\"MotionCompensateFrame_f64_us_per_frame\": %.2f, \"MotionCompensateFrame_f64_us_best_call\": %.2f, \"MotionCompensateFrame_3arg_3knots_f64_us_per_frame\": %.2f, \"MotionCompensateKittiCloud_f32_us_per_frame\": %.2f, "
      "\"stamp_start\": %.9f, \"stamp_middle\": %.9f, \"stamp_end\": %.9f, \"T_end\": {\"yaw_z\": %.17g, \"t\": [%.17g, %.17g, %.17g]}, \"trace\": %s, \"checksum\": %.6f}\n",
      n, iters, pooled ? "page-locked pool (the drop-in's default)" : "ordinary pageable memory (KMC_HOST_POOL=0 or no pool)",
      pooled ? "one kernel in place over the link" : "staged copies", us64, us64_min, us64_traj, us32, scan.stamp_start, scan.stamp_middle, scan.stamp_end, yaw, tx, ty, tz,
      trace_json.c_str(), checksum + out[0]);
  return 0;
}

// time_dropin_frame.cpp -- latency of the literal drop-in call kmc::MotionCompensateFrame(Frame const&, Time) on the shipped KITTI
// frame (f64 Eigen-layout cloud in the drop-in's own containers, result by value), and of the f32 KITTI-layout call next to it.
// The containers come from the C-ABI's page-locked pool (round 3), so the kernel works on them in place; KMC_HOST_POOL=0 in the
// environment restores ordinary memory and the staged three-copy route for comparison.
//   time_dropin_frame <golden_dir> [iterations=200]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "kitti_motion_compensation/data_io.hpp"
#include "kitti_motion_compensation/motion_compensation.hpp"
#include "kmc_hip.h"

using namespace kmc;

int main(int argc, char** argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: time_dropin_frame <golden_dir> [iterations]\n");
    return 2;
  }
  Path const run{std::string(argv[1]) + "/kitti_2011_09_26_drive_0005"};
  int const iters = argc > 2 ? std::atoi(argv[2]) : 200;
  LidarScan const scan{LoadLidarScan(run, 0)};
  Affine3d T_end;
  T_end.rotate(AngleAxisd{0.03, Vector3d{0, 0, 1}});
  T_end.translation() = Vector3d{1.3, 0.05, -0.02};
  Frame const frame{Affine3d::Identity(), T_end, scan};
  using clk = std::chrono::steady_clock;
  double checksum = 0;
  for (int i = 0; i < 10; ++i) checksum += MotionCompensateFrame(frame, scan.stamp_middle)(0, 0);
  auto t0 = clk::now();
  for (int i = 0; i < iters; ++i) checksum += MotionCompensateFrame(frame, scan.stamp_middle)(0, 0);
  double const us64 = std::chrono::duration<double, std::micro>(clk::now() - t0).count() / iters;

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
  std::printf("route: %s\n", pooled ? "containers in the page-locked pool -> ONE kernel in place over the link" : "ordinary host memory -> staged copies (KMC_HOST_POOL=0 or no pool)");
  std::printf("%zu points: MotionCompensateFrame(Frame, Time) [f64, result by value] %.1f us/frame = %.1f M points/s;  "
              "hip::MotionCompensateKittiCloud [f32 KITTI layout] %.1f us/frame = %.1f M points/s  (checksum %.6f)\n",
              n, us64, n / us64, us32, n / us32, checksum + out[0]);
  return 0;
}

// data_io.hpp -- the producers and consumers either side of the hot path (reference include/.../data_io.hpp), minus the
// image loaders (OpenCV).  Host code; see SURVEY.md section 8(f) rows N1 and N2.
#pragma once

#include <string>
#include <tuple>
#include <vector>

#include "kitti_motion_compensation/data_types.hpp"

namespace kmc {

Time LoadTimeStamp(Path const timestamp_file, std::size_t const frame_id);  // data_io.cpp:18-35
Oxts LoadOxts(Path const folder, std::size_t const frame_id);              // :37-66
Affine3d OxtsToPose(Oxts const& odometry, double const scale = 1.0);        // :68-88

// The reference loads into a fixed 250 000-point buffer without a bounds check (data_io.hpp:17, data_io.cpp:115);
// this loader sizes its buffer from the file.
class KittiPclLoader {
 public:
  std::tuple<Pointcloud, VectorXd> LoadPointcloud(Path const& file);  // data_io.cpp:101-138
  // raw f32 AoS x,y,z,intensity exactly as on disk -- the layout the f32 kernel consumes (page-locked when a HIP device is there)
  static KittiCloudF32 LoadRaw(Path const& file);
};

LidarScan LoadLidarScan(Path const folder, std::size_t const frame_id);  // :142-166

Frame MakeFrame(Oxts const& odometry_n_m_1, Oxts const& odometry_n, Oxts const& odometry_n_p_1,
                LidarScan const& lidar_scan);                            // :253-269

// The reference's third parameter asks for the four camera images as well (cv::Mat, data_io.cpp:279-282); this build has no
// OpenCV, so load_images == true throws std::runtime_error instead of silently returning a frame without images.
Frame LoadSingleFrame(Path const data_folder, std::size_t const frame_id, bool const load_images = false);  // :271-285

void WritePointcloud(Path const data_folder, std::size_t const frame_id, Pointcloud const& pointcloud,
                     VectorXd const& intensities);                        // :287-313
void WriteRaw(Path const data_folder, std::size_t const frame_id, float const* xyzi, std::size_t num_points);

// calib_velo_to_cam.txt (to_cam) or calib_imu_to_velo.txt: [R|T] as an Affine3d.  data_io.cpp:168-210
Affine3d LoadLidarExtrinsics(Path const data_folder, bool const to_cam = true);

}  // namespace kmc

namespace kmc::viz {

// The eight lines S_xx, K_xx, D_xx, R_xx, T_xx, S_rect_xx, R_rect_xx, P_rect_xx of one camera.  data_io.cpp:321-373
CameraCalibration CalibrationLinesToCalibration(std::vector<std::string> const calibration_lines);
// calib_cam_to_cam.txt: two header lines, then four blocks of eight.  data_io.cpp:375-406.  An unreadable file throws
// (the reference prints and calls exit(0)).
CameraCalibrations LoadCameraCalibrations(Path const data_folder);

}  // namespace kmc::viz

// data_io.hpp -- the producers and consumers either side of the hot path (reference include/.../data_io.hpp), minus the
// image / calibration loaders (OpenCV, visualization only).  Host code; see SURVEY.md section 8(f) rows N1 and N2.
#pragma once

#include <tuple>

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
  // raw f32 AoS x,y,z,intensity exactly as on disk -- the layout the f32 kernel consumes
  static std::vector<float> LoadRaw(Path const& file);
};

LidarScan LoadLidarScan(Path const folder, std::size_t const frame_id);  // :142-166

Frame MakeFrame(Oxts const& odometry_n_m_1, Oxts const& odometry_n, Oxts const& odometry_n_p_1,
                LidarScan const& lidar_scan);                            // :253-269

Frame LoadSingleFrame(Path const data_folder, std::size_t const frame_id);  // :271-285

void WritePointcloud(Path const data_folder, std::size_t const frame_id, Pointcloud const& pointcloud,
                     VectorXd const& intensities);                        // :287-313
void WriteRaw(Path const data_folder, std::size_t const frame_id, float const* xyzi, std::size_t num_points);

}  // namespace kmc

// camera_model.hpp -- row N4: the arithmetic of the reference's camera_model.cpp (Y = P_rect_xx * R_rect_00 * (R|T)_velo_to_cam * X
// for all four cameras, the in-view test and the colour ramp) on the GPU.  The reference's functions return cv::Mat
// images with the circles drawn in; OpenCV is not part of this build, so these return the DRAW LIST instead -- for
// every point, exactly what the reference hands to cv::circle (camera_model.cpp:31-32) -- and the caller keeps the
// four-line drawing loop.
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

#include "kitti_motion_compensation/data_types.hpp"
#include "kitti_motion_compensation/motion_compensation.hpp"

namespace kmc::viz {

struct Projection {
  std::size_t num_points{0};
  std::vector<std::int32_t> uv;    // [point][camera 0..3][u, v]; INT32_MIN twice for a skipped point
  std::vector<std::uint8_t> bgrv;  // [point][255-cs, cs, 255-cs, drawn?]: the cv::Scalar of camera_model.cpp:32 as 8-bit channels

  bool drawn(std::size_t i) const { return bgrv[4 * i + 3] != 0; }  // passes camera_model.cpp:21-24
  std::int32_t u(int camera, std::size_t i) const { return uv[(i * 4 + static_cast<std::size_t>(camera)) * 2]; }
  std::int32_t v(int camera, std::size_t i) const { return uv[(i * 4 + static_cast<std::size_t>(camera)) * 2 + 1]; }
  std::uint8_t const* color(std::size_t i) const { return &bgrv[4 * i]; }
};

// ProjectPointcloudOnFrame (camera_model.cpp:38-95) + ProjectPointcloudOnImage (:5-36) for the four cameras, minus the
// drawing: frame.scan.cloud (f64, columns 0..2) -> draw list.  Integers bit-exact against the CPU restatement.
Projection ProjectPointcloud(Frame const& frame, CameraCalibrations const& camera_calibrations, Affine3d const& tf_c00_lo,
                             double const max_range = 15.0);

// The same straight from the KITTI on-disk layout (f32 x,y,z,intensity, host pointer).  With `deskew` the cloud is motion
// compensated first inside the same kernel (handlers.cpp:81-87 in one pass) and, if xyzi_out != nullptr, written there.
Projection ProjectKittiCloud(float const* xyzi, std::size_t num_points, CameraCalibrations const& camera_calibrations,
                             Affine3d const& tf_c00_lo, double const max_range = 15.0,
                             hip::FramePoses const* deskew = nullptr, float* xyzi_out = nullptr);

}  // namespace kmc::viz

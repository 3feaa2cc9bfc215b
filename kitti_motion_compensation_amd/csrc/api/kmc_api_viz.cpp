// kmc_api_viz.cpp -- row N4 of SURVEY.md section 8(f): calibration loaders (data_io.cpp:168-210, :321-406) and the
// GPU projection behind kmc::viz (camera_model.cpp:5-95 minus the OpenCV drawing).
#include <fstream>
#include <stdexcept>
#include <string>

#include "kitti_motion_compensation/camera_model.hpp"
#include "kitti_motion_compensation/data_io.hpp"
#include "kitti_motion_compensation/utils.hpp"
#include "kmc_api_internal.hpp"

namespace kmc {

namespace {

// "name: v1 v2 ..." -> the values; the tokenizer keeps the name as token 0 like the reference's (utils.cpp:17-29)
std::vector<double> CalibrationValues(std::string const& line, std::size_t expected, char const* what) {
  std::vector<std::string> const tokens{TokenizeString(line)};
  if (tokens.size() < expected + 1) throw std::runtime_error(std::string("Malformed calibration line (") + what + "): " + line);
  std::vector<double> values(expected);
  try {
    for (std::size_t i = 0; i < expected; ++i) values[i] = std::stod(tokens[i + 1]);
  } catch (std::logic_error const&) {  // a field that is not a number
    throw std::runtime_error(std::string("Malformed calibration line (") + what + "): " + line);
  }
  return values;
}
Matrix3d Matrix3FromRowMajor(std::vector<double> const& v) {
  Matrix3d m;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) m(i, j) = v[static_cast<std::size_t>(3 * i + j)];
  return m;
}

}  // namespace

Affine3d LoadLidarExtrinsics(Path const data_folder, bool const to_cam) {  // data_io.cpp:168-210
  Path const file{data_folder / Path(to_cam ? "calib_velo_to_cam.txt" : "calib_imu_to_velo.txt")};
  std::ifstream is(file);
  if (not is.is_open()) throw std::runtime_error("Failed to open camera calibration file: " + file.string());
  std::string line;
  std::getline(is, line);  // calib_time
  std::getline(is, line);
  Matrix3d const R{Matrix3FromRowMajor(CalibrationValues(line, 9, "R"))};
  std::getline(is, line);
  std::vector<double> const T{CalibrationValues(line, 3, "T")};
  Affine3d tf;  // identity
  tf = R * tf;
  tf.translation() = Vector3d{T[0], T[1], T[2]};
  return tf;
}

namespace viz {

CameraCalibration CalibrationLinesToCalibration(std::vector<std::string> const calibration_lines) {  // data_io.cpp:321-373
  if (calibration_lines.size() < 8) throw std::runtime_error("A camera calibration block has eight lines");
  CameraCalibration c;
  std::vector<double> v{CalibrationValues(calibration_lines[0], 2, "S")};
  c.S(0) = v[0];
  c.S(1) = v[1];
  c.K = Matrix3FromRowMajor(CalibrationValues(calibration_lines[1], 9, "K"));
  v = CalibrationValues(calibration_lines[2], 5, "D");
  for (int i = 0; i < 5; ++i) c.D[i] = v[static_cast<std::size_t>(i)];
  c.R = Matrix3FromRowMajor(CalibrationValues(calibration_lines[3], 9, "R"));
  v = CalibrationValues(calibration_lines[4], 3, "T");
  c.T = Vector3d{v[0], v[1], v[2]};
  v = CalibrationValues(calibration_lines[5], 2, "S_rect");
  c.S_rect(0) = v[0];
  c.S_rect(1) = v[1];
  c.R_rect = Matrix3FromRowMajor(CalibrationValues(calibration_lines[6], 9, "R_rect"));
  v = CalibrationValues(calibration_lines[7], 12, "P_rect");
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) c.P_rect(i, j) = v[static_cast<std::size_t>(4 * i + j)];
  return c;
}

CameraCalibrations LoadCameraCalibrations(Path const data_folder) {  // data_io.cpp:375-406
  Path const file{data_folder / Path("calib_cam_to_cam.txt")};
  std::ifstream is(file);
  if (not is.is_open()) throw std::runtime_error("Failed to open camera calibration file: " + file.string());
  std::string line;
  std::getline(is, line);  // calib_time
  std::getline(is, line);  // corner_dist
  CameraCalibration cams[4];
  for (auto& cam : cams) {
    std::vector<std::string> lines;
    for (int i = 0; i < 8; ++i) {
      if (not std::getline(is, line)) throw std::runtime_error("Camera calibration file ends early: " + file.string());
      lines.push_back(line);
    }
    cam = CalibrationLinesToCalibration(lines);
  }
  return CameraCalibrations{cams[0], cams[1], cams[2], cams[3]};
}

namespace {

kmc_camera_rig MakeRig(CameraCalibrations const& cc, Affine3d const& tf_c00_lo, double max_range) {
  kmc_camera_rig rig;
  tf_c00_lo.to_rt12(rig.tf_c00_lo);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) rig.R_rect_00[3 * i + j] = cc.camera_00.R_rect(i, j);  // camera_model.cpp:78-79
  CameraCalibration const* cams[4] = {&cc.camera_00, &cc.camera_01, &cc.camera_02, &cc.camera_03};
  for (int c = 0; c < 4; ++c)
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 4; ++j) rig.P_rect[c][4 * i + j] = cams[c]->P_rect(i, j);
  rig.max_range = max_range;
  return rig;
}

Projection MakeProjection(std::size_t n) {
  Projection out;
  out.num_points = n;
  out.uv.resize(8 * n);
  out.bgrv.resize(4 * n);
  return out;
}

}  // namespace

Projection ProjectPointcloud(Frame const& frame, CameraCalibrations const& camera_calibrations, Affine3d const& tf_c00_lo,
                             double const max_range) {
  Pointcloud const& cloud{frame.scan.cloud};
  std::size_t const n{static_cast<std::size_t>(cloud.rows())};
  Projection out{MakeProjection(n)};
  if (n == 0) return out;
  kmc_camera_rig const rig{MakeRig(camera_calibrations, tf_c00_lo, max_range)};
  kmc_ctx* ctx = detail::thread_context();
  int const rc = kmc_hip_project_f64cols(ctx, cloud.col(0), cloud.col(1), cloud.col(2), n, &rig, out.uv.data(),
                                         out.bgrv.data(), KMC_MEM_HOST, nullptr);
  if (rc != KMC_OK) detail::throw_status(rc, "kmc_hip_project_f64cols", ctx);
  return out;
}

Projection ProjectKittiCloud(float const* xyzi, std::size_t num_points, CameraCalibrations const& camera_calibrations,
                             Affine3d const& tf_c00_lo, double const max_range, hip::FramePoses const* deskew, float* xyzi_out) {
  Projection out{MakeProjection(num_points)};
  if (num_points == 0) return out;
  kmc_camera_rig const rig{MakeRig(camera_calibrations, tf_c00_lo, max_range)};
  kmc_frame_params params;
  if (deskew) {
    double T0[12], T1[12];
    deskew->T_start.to_rt12(T0);
    deskew->T_end.to_rt12(T1);
    int const rc = kmc_frame_params_from_poses(T0, T1, deskew->stamp_start, deskew->stamp_end, deskew->requested_time, &params);
    if (rc == KMC_ERR_TIME_OUT_OF_RANGE) detail::die_time_out_of_range("kmc::viz::ProjectKittiCloud");
    if (rc != KMC_OK) detail::throw_status(rc, "kmc_frame_params_from_poses");
  }
  kmc_ctx* ctx = detail::thread_context();
  int const rc = kmc_hip_project_f32(ctx, xyzi, num_points, &rig, deskew ? &params : nullptr, deskew ? xyzi_out : nullptr,
                                     out.uv.data(), out.bgrv.data(), KMC_MEM_HOST, nullptr);
  if (rc != KMC_OK) detail::throw_status(rc, "kmc_hip_project_f32", ctx);
  return out;
}

}  // namespace viz
}  // namespace kmc

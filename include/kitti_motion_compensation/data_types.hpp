// data_types.hpp -- the reference's API types (include/kitti_motion_compensation/data_types.hpp) without Eigen/OpenCV.
//
// The reference's types ARE Eigen types (Pointcloud = Eigen::MatrixX4d, Affine3d = Eigen::Affine3d, ...).  This image
// has no Eigen, and the drop-in must not force one on its users, so this header provides small value types with the
// same names, the same memory layout where layout is part of the contract, and the subset of the Eigen call syntax the
// reference's hot path, callers and tests use (row(i), operator()(i,j), Identity(), inverse(), operator*, ...):
//
//   Pointcloud / MatrixX4d : N x 4 doubles, COLUMN-major like Eigen's default (x[0..N) y[0..N) z[0..N) w[0..N)) -- the
//                            four columns are handed to the device as they are (kmc_hip_deskew_f64cols)
//   VectorXd               : N doubles
//   Affine3d               : 3x3 linear part + translation; inverse() is the general (non-rigid) inverse and rotation()
//                            the orthogonal polar factor, exactly what Eigen's Affine mode gives the reference
//   Twist                  : 6 doubles [rho; phi]                                     (reference data_types.hpp:25)
//   Oxts, LidarScan, Frame : field-for-field the reference structs (data_types.hpp:35-59, :76-91) minus the optional
//                            camera images (OpenCV, visualization only -- out of scope)
//
// A maintainer who HAS Eigen does not need this header at all: INTEGRATION.md shows the 12-line body that calls the
// C-ABI straight from the reference's own Eigen types.
#pragma once

#include <cstddef>
#include <filesystem>
#include <initializer_list>
#include <memory>
#include <new>
#include <type_traits>
#include <utility>
#include <vector>

namespace kmc {

using Time = double;  // seconds since midnight (data_types.hpp:16-20)
using Path = std::filesystem::path;
using Index = std::ptrdiff_t;

struct Vector3d {
  double v[3]{0, 0, 0};
  Vector3d() = default;
  Vector3d(double x, double y, double z) : v{x, y, z} {}
  double& operator()(Index i) { return v[i]; }
  double operator()(Index i) const { return v[i]; }
  double& operator[](Index i) { return v[i]; }
  double operator[](Index i) const { return v[i]; }
  double x() const { return v[0]; }
  double y() const { return v[1]; }
  double z() const { return v[2]; }
  double norm() const;
  static Vector3d Zero() { return {}; }
  static Vector3d UnitX() { return {1, 0, 0}; }
  static Vector3d UnitY() { return {0, 1, 0}; }
  static Vector3d UnitZ() { return {0, 0, 1}; }
};

struct Vector4d {
  double v[4]{0, 0, 0, 0};
  Vector4d() = default;
  Vector4d(double x, double y, double z, double w) : v{x, y, z, w} {}
  double& operator()(Index i) { return v[i]; }
  double operator()(Index i) const { return v[i]; }
  double& operator[](Index i) { return v[i]; }
  double operator[](Index i) const { return v[i]; }
};

struct Twist {  // [rho(0:3); phi(3:6)]
  double v[6]{0, 0, 0, 0, 0, 0};
  Twist() = default;
  Twist(std::initializer_list<double> l) {
    Index i = 0;
    for (double d : l)
      if (i < 6) v[i++] = d;
  }
  double& operator()(Index i) { return v[i]; }
  double operator()(Index i) const { return v[i]; }
  Vector3d rho() const { return {v[0], v[1], v[2]}; }
  Vector3d phi() const { return {v[3], v[4], v[5]}; }
};
Twist operator*(double s, Twist const& t);

struct Matrix3d {
  double m[3][3]{{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  double& operator()(Index i, Index j) { return m[i][j]; }
  double operator()(Index i, Index j) const { return m[i][j]; }
  static Matrix3d Identity();
  static Matrix3d Zero() { return {}; }
  Matrix3d transpose() const;
  Matrix3d inverse() const;
  double trace() const { return m[0][0] + m[1][1] + m[2][2]; }
  double determinant() const;
  double sum() const;
};
Matrix3d operator*(Matrix3d const& a, Matrix3d const& b);
Vector3d operator*(Matrix3d const& a, Vector3d const& b);
Matrix3d operator*(double s, Matrix3d const& a);
Matrix3d operator+(Matrix3d const& a, Matrix3d const& b);
Matrix3d operator-(Matrix3d const& a, Matrix3d const& b);

struct Matrix4d {
  double m[4][4]{{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  double& operator()(Index i, Index j) { return m[i][j]; }
  double operator()(Index i, Index j) const { return m[i][j]; }
  double trace() const { return m[0][0] + m[1][1] + m[2][2] + m[3][3]; }
  double sum() const;
};
Matrix4d operator-(Matrix4d const& a, Matrix4d const& b);

// Eigen::AngleAxisd stand-in for building test poses: Affine3d::rotate(AngleAxisd{angle, axis}).
struct AngleAxisd {
  double angle;
  Vector3d axis;
  AngleAxisd(double a, Vector3d const& ax) : angle(a), axis(ax) {}
  Matrix3d toRotationMatrix() const;
};

class Affine3d {
 public:
  Affine3d() = default;  // identity
  static Affine3d Identity() { return Affine3d(); }
  Matrix3d& linear() { return linear_; }
  Matrix3d const& linear() const { return linear_; }
  Vector3d& translation() { return translation_; }
  Vector3d const& translation() const { return translation_; }
  Matrix3d rotation() const;  // orthogonal polar factor (Eigen Affine-mode rotation(), used at lie_algebra.cpp:95)
  Affine3d inverse() const;   // general inverse (Eigen Affine mode), trajectory_interpolation.cpp:35, :44
  Affine3d& rotate(AngleAxisd const& aa);  // *this = *this * R
  Matrix4d matrix() const;
  // row-major 3x4 [R|t]: the pose layout of the C-ABI (include/kmc_hip.h)
  void to_rt12(double out[12]) const;
  static Affine3d from_rt12(const double in[12]);

 private:
  Matrix3d linear_ = Matrix3d::Identity();
  Vector3d translation_{};
};
Affine3d operator*(Affine3d const& a, Affine3d const& b);
Vector4d operator*(Affine3d const& a, Vector4d const& p);  // homogeneous product, w passed through (motion_compensation.cpp:13)
Affine3d operator*(Matrix3d const& r, Affine3d const& a);  // "pose = R * pose" (data_io.cpp:84)

namespace detail {
// Storage of the dynamic containers.  Two properties on top of std::allocator:
//   * default-initialising construct(): vector(n) / resize(n) leave the doubles untouched, so the library can hand out a result
//     buffer without first writing 4 n zeros into it (Eigen's MatrixX4d(n, 4) does not initialise either);
//   * blocks of kPooledBytes or more come from libkmc_hip.so's process-wide pool of PAGE-LOCKED, device-addressable memory
//     (kmc_host_pool_alloc, include/kmc_hip.h).  That is what lets MotionCompensateFrame(Frame const&, Time) run as ONE kernel
//     directly on the caller's Frame and on the Pointcloud it returns -- upload and download overlapped on the link, no staging
//     copies (round 3: ~2x faster per KITTI frame than the staged route).  Without a HIP device, or with KMC_HOST_POOL=0, the
//     pool declines and ordinary aligned memory is used; foreign pointers (an Eigen user's own matrices) keep
//     the staged route.  Freed blocks are recycled by the pool: the reference's allocate-per-call pattern
//     (motion_compensation.cpp:21) costs a list pop, not a page-lock.
extern "C" {
int kmc_host_pool_alloc(std::size_t bytes, void** out);
int kmc_host_pool_free(void* ptr);
}
constexpr std::size_t kPooledBytes = 32 * 1024;
template <typename T>
struct PoolAllocator {
  using value_type = T;
  PoolAllocator() = default;
  template <typename U>
  PoolAllocator(PoolAllocator<U> const&) noexcept {}
  template <typename U>
  struct rebind {
    using other = PoolAllocator<U>;
  };
  T* allocate(std::size_t n) {
    std::size_t const bytes = n * sizeof(T);
    if (bytes >= kPooledBytes) {
      void* p = nullptr;
      if (kmc_host_pool_alloc(bytes, &p) == 0 && p) return static_cast<T*>(p);
    }
    return static_cast<T*>(::operator new(bytes, std::align_val_t(64)));
  }
  void deallocate(T* p, std::size_t n) noexcept {
    if (n * sizeof(T) >= kPooledBytes && kmc_host_pool_free(p)) return;
    ::operator delete(p, std::align_val_t(64));
  }
  template <typename U>
  void construct(U* p) noexcept(std::is_nothrow_default_constructible<U>::value) {
    ::new (static_cast<void*>(p)) U;
  }
  template <typename U, typename... Args>
  void construct(U* p, Args&&... args) {
    ::new (static_cast<void*>(p)) U(std::forward<Args>(args)...);
  }
  template <typename U>
  bool operator==(PoolAllocator<U> const&) const noexcept { return true; }
  template <typename U>
  bool operator!=(PoolAllocator<U> const&) const noexcept { return false; }
};
using Storage = std::vector<double, PoolAllocator<double>>;
}  // namespace detail

// A KITTI cloud as it is on disk -- float x, y, z, intensity per point (data_io.cpp:101-138) -- in page-locked memory when a HIP device is
// there: kmc::hip::MotionCompensateKittiCloud then works on it in place over the link instead of staging copies.
using KittiCloudF32 = std::vector<float, detail::PoolAllocator<float>>;

class VectorXd {
 public:
  VectorXd() = default;
  explicit VectorXd(Index n) : d_(static_cast<std::size_t>(n), 0.0) {}
  static VectorXd Uninitialized(Index n) {  // contents indeterminate until written
    VectorXd v;
    v.d_ = detail::Storage(static_cast<std::size_t>(n));
    return v;
  }
  Index size() const { return static_cast<Index>(d_.size()); }
  Index rows() const { return size(); }
  double& operator()(Index i) { return d_[static_cast<std::size_t>(i)]; }
  double operator()(Index i) const { return d_[static_cast<std::size_t>(i)]; }
  double* data() { return d_.data(); }
  double const* data() const { return d_.data(); }

 private:
  detail::Storage d_;
};

class MatrixX4d;
namespace detail {
void set_homogeneous(MatrixX4d& m, bool value);  // library-internal: the library has just written the column itself
}

// N x 4, column-major (Eigen::MatrixX4d).  Homogeneous points: the last column is 1 (data_types.hpp:13-14).
//
// The matrix remembers whether its last column is KNOWN to be all ones (every cloud the reference loads is, data_io.cpp:130, and
// MotionCompensateFrame passes the column through, motion_compensation.cpp:13): a Pointcloud with that knowledge lets the device
// skip the column -- 16 of the 72 bytes a point costs on the link.  The knowledge is established only by detect_homogeneous()
// (which looks at every element) or by the library for clouds it has produced itself, it is copied with the matrix, and ANY
// non-const access (operator(), row(), col(), data()) drops it, so it can never be stale.
class MatrixX4d {
 public:
  class RowRef {
   public:
    RowRef(double* base, Index n, Index i) : base_(base), n_(n), i_(i) {}
    double& operator()(Index j) { return base_[j * n_ + i_]; }
    double operator()(Index j) const { return base_[j * n_ + i_]; }
    RowRef& operator=(Vector4d const& p) {
      for (Index j = 0; j < 4; ++j) base_[j * n_ + i_] = p(j);
      return *this;
    }
    operator Vector4d() const { return {base_[i_], base_[n_ + i_], base_[2 * n_ + i_], base_[3 * n_ + i_]}; }

   private:
    double* base_;
    Index n_, i_;
  };
  MatrixX4d() = default;
  MatrixX4d(Index rows, Index /*cols == 4*/) : n_(rows), d_(static_cast<std::size_t>(rows) * 4, 0.0) {}
  static MatrixX4d Uninitialized(Index rows) {  // contents indeterminate until written (what Eigen's MatrixX4d(rows, 4) gives)
    MatrixX4d m;
    m.n_ = rows;
    m.d_ = detail::Storage(static_cast<std::size_t>(rows) * 4);
    return m;
  }
  Index rows() const { return n_; }
  Index cols() const { return 4; }
  double& operator()(Index i, Index j) { homogeneous_ = false; return d_[static_cast<std::size_t>(j * n_ + i)]; }
  double operator()(Index i, Index j) const { return d_[static_cast<std::size_t>(j * n_ + i)]; }
  RowRef row(Index i) { homogeneous_ = false; return RowRef(d_.data(), n_, i); }
  Vector4d row(Index i) const { return {(*this)(i, 0), (*this)(i, 1), (*this)(i, 2), (*this)(i, 3)}; }
  double* col(Index j) { homogeneous_ = false; return d_.data() + j * n_; }
  double const* col(Index j) const { return d_.data() + j * n_; }
  double* data() { homogeneous_ = false; return d_.data(); }
  double const* data() const { return d_.data(); }
  // true only if the last column is known to be all ones (see above)
  bool is_homogeneous() const { return homogeneous_; }
  // looks at every element of the last column; remembers and returns whether all of them are exactly 1.0
  bool detect_homogeneous() {
    double const* w = d_.data() + 3 * n_;
    bool ones = true;
    for (Index i = 0; i < n_ && ones; ++i) ones = w[i] == 1.0;
    homogeneous_ = ones;
    return ones;
  }

 private:
  friend void detail::set_homogeneous(MatrixX4d& m, bool value);
  Index n_ = 0;
  detail::Storage d_;
  bool homogeneous_ = false;
};
namespace detail {
inline void set_homogeneous(MatrixX4d& m, bool value) { m.homogeneous_ = value; }
}
using Pointcloud = MatrixX4d;

struct Oxts {  // data_types.hpp:35-49
  Time stamp;
  double lat, lon, alt, roll, pitch, yaw, vf, vl, vu;
};

struct LidarScan {  // data_types.hpp:51-59
  Time stamp_start;
  Time stamp_middle;  // camera trigger time
  Time stamp_end;
  Pointcloud cloud;
  VectorXd intensities;
  VectorXd timestamps;  // per-point pseudo stamps (GetPseudoTimeStamps)
};

struct Frame {  // data_types.hpp:76-91 (without the optional camera images)
  Frame(Affine3d const& start_pose, Affine3d const& end_pose, LidarScan const& lidar_scan)
      : T_start{start_pose}, T_end{end_pose}, scan{lidar_scan} {}
  Affine3d T_start;  // pose at scan.stamp_start
  Affine3d T_end;    // pose at scan.stamp_end
  LidarScan scan;
};

}  // namespace kmc

// ---- camera calibration (data_types.hpp:96-116 of the reference), for the projection row N4 -------------------------
namespace kmc::viz {

struct Vector2d {
  double v[2]{0, 0};
  double& operator()(Index i) { return v[i]; }
  double operator()(Index i) const { return v[i]; }
};
struct Matrix34d {  // Eigen::Matrix<double, 3, 4>
  double m[3][4]{{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  double& operator()(Index i, Index j) { return m[i][j]; }
  double operator()(Index i, Index j) const { return m[i][j]; }
};
using P = Matrix34d;  // pinhole projection matrix

struct CameraCalibration {  // one S_xx .. P_rect_xx block of calib_cam_to_cam.txt
  Vector2d S;
  Matrix3d K;
  double D[5]{0, 0, 0, 0, 0};
  Matrix3d R;
  Vector3d T;
  Vector2d S_rect;
  Matrix3d R_rect;
  P P_rect;
};
struct CameraCalibrations {
  CameraCalibration camera_00, camera_01, camera_02, camera_03;
};

}  // namespace kmc::viz

// BASELINE.json's north_star names the namespace in full; the reference code uses `kmc`.
namespace kitti_motion_compensation = kmc;

/*
 * kmc_oracle.h -- CPU ORACLE for the per-point deskew path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a dependency-free, double-precision C restatement of the reference's algorithm
 * (fracgawd/kitti_motion_compensation).  It exists to CHECK the HIP product path; nothing that ships
 * may include, link or call it.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * use it.
 *
 * Parity status: PINNED against the reference's own known-answer tests (tests/test_oracle_kat.py):
 *   test/test_motion_compensation.cpp:59-75, test/test_timestamp_mocking.cpp:55-57,71-73,84-86,
 *   test/test_lie_algebra.cpp:5-47, test/test_trajectory_interpolation.cpp:43-60,77-81,
 *   test/test_oxts_to_pose.cpp:17-20, test/test_data_io.cpp:53-78.
 * The reference itself cannot be compiled in this image (it needs Eigen3 + OpenCV, both absent), so
 * there is no oracle/_ref build; Eigen 3.4's published algorithms for the few calls on the path
 * (Affine inverse, polar-factor rotation(), AngleAxis->Quaternion->Matrix) are restated here.
 *
 * Conventions: all matrices are ROW-major double[9]; an affine pose is {R[9], t[3]} (last row 0 0 0 1
 * implicit); a twist is [rho(3); phi(3)] exactly like the reference (lie_algebra.cpp:84-85).
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 */
#ifndef KMC_ORACLE_H
#define KMC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kmo_affine {
  double R[9]; /* row-major linear part */
  double t[3]; /* translation */
} kmo_affine;

/* include/kitti_motion_compensation/data_types.hpp:35-49 */
typedef struct kmo_oxts {
  double stamp, lat, lon, alt, roll, pitch, yaw, vf, vl, vu;
} kmo_oxts;

/* trajectory_interpolation.hpp:26-30 */
typedef struct kmo_interpolator {
  double time_1;
  kmo_affine pose_1;
  double time_2;
  kmo_affine pose_2;
} kmo_interpolator;

#define KMO_OK 0
#define KMO_ERR_TIME_OUT_OF_RANGE (-1) /* the reference's assert -> abort(), trajectory_interpolation.cpp:32 */

/* ---- L0: lie_algebra.cpp ---- */
void kmo_hat(const double a[3], double M[9]);                       /* :7-18  */
void kmo_vee(const double M[9], double a[3]);                       /* :20    */
void kmo_so3_exp(const double phi[3], double R[9]);                 /* :22-35 */
void kmo_so3_log(const double R[9], double phi[3]);                 /* :37-49 */
void kmo_left_jacobian(const double phi[3], double J[9]);           /* :51-65 */
void kmo_inverse_left_jacobian(const double phi[3], double J[9]);   /* :67-81 */
void kmo_se3_exp(const double xi[6], kmo_affine* T);                /* :83-92 */
void kmo_se3_log(const kmo_affine* T, double xi[6]);                /* :94-103 */

/* ---- Eigen 3.4 calls made on the path (restated published algorithms) ---- */
void kmo_affine_identity(kmo_affine* T);
void kmo_affine_inverse(const kmo_affine* T, kmo_affine* out);      /* Transform::inverse(Affine): general 3x3 inverse */
void kmo_affine_mul(const kmo_affine* A, const kmo_affine* B, kmo_affine* out);
void kmo_affine_rotation(const kmo_affine* T, double R[9]);         /* Transform::rotation(): SVD polar factor */
void kmo_affine_apply4(const kmo_affine* T, const double p[4], double out[4]); /* Affine3d * Vector4d */
void kmo_mat3_mul(const double A[9], const double B[9], double C[9]);
double kmo_mat3_det(const double A[9]);

/* ---- L1: trajectory_interpolation.cpp ---- */
void kmo_interpolator_from_poses(double t1, const kmo_affine* p1, double t2, const kmo_affine* p2,
                                 kmo_interpolator* out);                                   /* :27-29 */
void kmo_interpolator_from_oxts(const kmo_oxts* o0, const kmo_oxts* o1, kmo_interpolator* out); /* :21-25 */
int kmo_get_pose_at_time(const kmo_interpolator* ti, double time, kmo_affine* out);         /* :31-41 */
int kmo_relative_pose_between_times(const kmo_interpolator* ti, double anchor, double query,
                                    kmo_affine* out);                                       /* :43-45 */
int kmo_interpolate_trajectory(const kmo_oxts* o1, const kmo_oxts* o2, double time, kmo_affine* out); /* :14-19 */

/* ---- L2a: timestamp_mocking.cpp ---- */
double kmo_fraction_of_scan_completed(const double p[4]);                                   /* :46 */
double kmo_pseudo_timestamp(const double p[4], double scan_start, double scan_end);         /* :49-54 */
/* cloud is N x 4 COLUMN-major (Eigen::MatrixX4d), i.e. x[0..n) y[0..n) z[0..n) w[0..n) */
void kmo_pseudo_timestamps(const double* cloud_colmajor, size_t n, double start, double end, double* stamps); /* :56-63 */

/* ---- L2: motion_compensation.cpp ---- */
int kmo_motion_compensate_point(const kmo_interpolator* ti, double point_stamp, const double p[4],
                                double requested_time, double out[4]);                      /* :9-14 */
/* Faithful restatement of MotionCompensateFrame (:16-28): N x 4 column-major in/out, explicit stamps.
 * Returns KMO_ERR_TIME_OUT_OF_RANGE (and *n_bad = count) where the reference would abort. */
int kmo_motion_compensate_frame(const double* cloud_colmajor, const double* stamps, size_t n,
                                double stamp_start, const kmo_affine* T_start, double stamp_end,
                                const kmo_affine* T_end, double requested_time, double* out_colmajor,
                                size_t* n_bad);

/* ---- L3 (producer of the per-frame constants): data_io.cpp ---- */
void kmo_oxts_to_pose(const kmo_oxts* o, double scale, kmo_affine* out);                    /* :68-88 */
int kmo_make_frame_poses(const kmo_oxts* o_nm1, const kmo_oxts* o_n, const kmo_oxts* o_np1,
                         double stamp_start, double stamp_end, kmo_affine* T_start, kmo_affine* T_end); /* :253-269 */

/* ---- whole-pipeline helpers on the KITTI on-disk layout (f32 AoS x,y,z,intensity) ----
 * These chain exactly what the reference chains between reading a .bin and writing one:
 * LoadPointcloud f32->f64 (data_io.cpp:126-135), GetPseudoTimeStamps (data_io.cpp:163),
 * MotionCompensateFrame (handlers.cpp:60) and, for out_f32, WritePointcloud's cast (data_io.cpp:300-310).
 * mode 0 = FAITHFUL (per point: 2x GetPoseAtTime incl. the loop-invariant Log, like the reference; B1/B2)
 * mode 1 = HOISTED  (closed form Exp((x_i - x_r) f) with f computed once; B3)
 * threads <= 1 -> serial; > 1 -> OpenMP over points (if built with -fopenmp).
 * out_xyz_f64 (3 doubles per point, AoS) and out_xyzi_f32 (4 floats per point) may each be NULL.
 * stamps_out (n doubles) and frac_out may be NULL. */
int kmo_deskew_xyzi_f32(const float* xyzi, size_t n, double stamp_start, const kmo_affine* T_start,
                        double stamp_end, const kmo_affine* T_end, double requested_time, int mode,
                        int threads, double* out_xyz_f64, float* out_xyzi_f32, double* stamps_out,
                        size_t* n_bad);

/* ---- N-knot trajectory (this project's generalisation of the reference's single geodesic; NOT in the reference) ----
 * T(t) is the reference's own GetPoseAtTime (trajectory_interpolation.cpp:31-41) applied to the knot pair that brackets t
 * (t_k <= t < t_{k+1}; the last knot belongs to the last segment); correction_i = T(requested)^-1 * T(stamp_i), formed
 * exactly like RelativePoseBetweenTimes (:43-45).  With 2 knots it IS the reference algorithm. */
int kmo_traj_pose_at_time(const double* times, const kmo_affine* poses, size_t n_knots, double t, kmo_affine* out);
int kmo_deskew_xyzi_f32_traj(const float* xyzi, size_t n, double stamp_start, double stamp_end, const double* times,
                             const kmo_affine* poses, size_t n_knots, double requested_time, int threads,
                             double* out_xyz_f64, uint32_t* bracket_by_time_out, size_t* n_bad);
/* Eigen-layout variant with explicit per-point stamps (N x 4 column-major in/out). */
int kmo_motion_compensate_frame_traj(const double* cloud_colmajor, const double* stamps, size_t n, const double* times,
                                     const kmo_affine* poses, size_t n_knots, double requested_time,
                                     double* out_colmajor, uint32_t* bracket_out, size_t* n_bad);

/* Per-point integer bracket index decided WITHOUT trig: for every interior knot k with scan
 * fraction c_k = (t_k - stamp_start)/(stamp_end - stamp_start) and direction (ck, sk) = f32(cos, sin)(pi - 2 pi c_k),
 *   ge_k(x, y) = !lt_k,  lt_k = (c_k <= 0) ? false : (c_k > 1) ? true :
 *                 (x == 0 && y == 0) ? (frac_of_signed_zeros < c_k) :
 *                 (c_k <= 0.5) ? (!signbit(y) && (cross > 0 || (cross == 0 && dot < 0)))
 *                               : (!signbit(y) || cross > 0 || (cross == 0 && dot < 0))     [cross == 0 && dot < 0: the point is
 *                                 exactly opposite the knot direction, i.e. half a turn EARLIER in the scan]
 *   cross = ck*y - sk*x, dot = ck*x + sk*y   in f32, every operation rounded, no contraction
 * index = sum over interior knots of ge_k.  The device executes the same IEEE operations -> bit-exact. */
void kmo_bracket_indices_f32(const float* xyzi, size_t n, const double* times, size_t n_knots, double stamp_start,
                             double stamp_end, uint32_t* out);

/* ---- next row N4: LiDAR -> image projection (camera_model.cpp), WITHOUT the OpenCV drawing --------------------------
 * PARITY UNPINNED for this block: the reference has no test of camera_model.cpp, and OpenCV/Eigen are absent here; the
 * arithmetic below is a restatement only.  What it states, per point p = (x, y, z) of the lidar cloud:
 *   camera_model.cpp:62-75   p_c00  = tf_c00_lo * [x y z 1]'       Eigen Affine3d * 4xN: top 3 rows, k = 0..3 left to right
 *   camera_model.cpp:78-81   p_rect = [R_rect_00 0; 0 1] * p_c00   (the 4th product of every row is exactly 0 or 1*1)
 *   camera_model.cpp:9       pix_c  = P_rect_c * p_rect            3x4 * 4xN, k = 0..3 left to right
 *   camera_model.cpp:12      pix_c  = pix_c / pix_c(2)             IEEE division
 *   camera_model.cpp:21-24   skipped unless 0.01 <= z_rect <= max_range and y_rect <= 1.25
 *   camera_model.cpp:28-29   color_scale = 255 * (z_rect / (max_range - 0.01))
 *   camera_model.cpp:31      cv::Point(u, v): double -> int, truncation toward zero (x86 cvttsd2si: INT32_MIN when the
 *                            value does not fit)
 *   camera_model.cpp:32      cv::Scalar(255 - cs, cs, 255 - cs) -> 8-bit channels the way OpenCV (un-vendored apt
 *                            libopencv-dev 4.5.4 of ubuntu:22.04, Dockerfile:3) writes a Scalar into a CV_8U image:
 *                            saturate_cast<uchar>(double) = clamp(cvRound(v)), cvRound = round-half-to-even.
 * All products and sums are individually rounded (the reference builds with plain -O3, CMakeLists.txt:8: no FMA).
 * Outputs: uv[i][c][2] int32 for the 4 cameras c (INT32_MIN, INT32_MIN when the point is skipped) and
 * bgrv[i][4] = {255-cs, cs, 255-cs, 1} as uint8 (all 0 when skipped). */
typedef struct kmo_camera_rig {
  double tf_c00_lo[12]; /* row-major 3x4, LoadLidarExtrinsics data_io.cpp:168-210 */
  double R_rect_00[9];  /* row-major, calib_cam_to_cam.txt R_rect_00 */
  double P_rect[4][12]; /* row-major 3x4 per camera */
  double max_range;     /* camera_model.hpp:8 default 15.0 */
} kmo_camera_rig;
void kmo_project_points(const double* x, const double* y, const double* z, size_t n, const kmo_camera_rig* rig,
                        int32_t* uv, uint8_t* bgrv);
/* the same on the KITTI f32 layout: every coordinate is widened to double first, like the loader does (data_io.cpp:118-131) */
void kmo_project_xyzi_f32(const float* xyzi, size_t n, const kmo_camera_rig* rig, int32_t* uv, uint8_t* bgrv);

int kmo_num_threads(void); /* omp_get_max_threads() or 1 */
void kmo_set_num_threads(int threads); /* default team size of the loops that take no explicit thread count */

#ifdef __cplusplus
}
#endif
#endif /* KMC_ORACLE_H */

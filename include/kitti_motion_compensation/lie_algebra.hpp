// lie_algebra.hpp -- SO(3) / SE(3) exponentials and logarithms: the reference's include/kitti_motion_compensation/lie_algebra.hpp:12-26
// on the Eigen-free types of data_types.hpp.  Double precision on the host; used by the once-per-frame pre-step (one Log per
// frame, kmc_frame_params_from_poses).  The per-point Exp of the hot path runs on the GPU (csrc/kmc_device_math.hip.h).
#pragma once

#include "kitti_motion_compensation/data_types.hpp"

namespace kmc::lie {

Matrix3d Hat(Vector3d const& a);                      // lie_algebra.cpp:7-18
Vector3d Vee(Matrix3d const& a);                      // :20
Matrix3d Exp(Vector3d const& phi);                    // :22-35
Vector3d Log(Matrix3d const& R);                      // :37-49
Matrix3d LeftJacobian(Vector3d const& phi);           // :51-65
Matrix3d InverseLeftJacobian(Vector3d const& phi);    // :67-81
Affine3d Exp(Twist const& xi);                        // :83-92   twist order [rho(0:3); phi(3:6)]
Twist Log(Affine3d const& T);                         // :94-103  on T.rotation(), the orthogonal polar factor (Eigen Affine mode)

}  // namespace kmc::lie

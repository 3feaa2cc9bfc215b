// lie_algebra.hpp -- kmc::lie, same functions as the reference header (include/.../lie_algebra.hpp:12-26).
// Host-side double precision; used by the per-frame pre-step (one Log per frame) -- the per-point Exp runs on the GPU.
#pragma once

#include "kitti_motion_compensation/data_types.hpp"

namespace kmc::lie {

Matrix3d Hat(Vector3d const& a);                      // lie_algebra.cpp:7-18
Vector3d Vee(Matrix3d const& a);                      // :20
Matrix3d Exp(Vector3d const& phi);                    // :22-35
Vector3d Log(Matrix3d const& R);                      // :37-49
Matrix3d LeftJacobian(Vector3d const& phi);           // :51-65
Matrix3d InverseLeftJacobian(Vector3d const& phi);    // :67-81
Affine3d Exp(Twist const& xi);                        // :83-92
Twist Log(Affine3d const& T);                         // :94-103

}  // namespace kmc::lie

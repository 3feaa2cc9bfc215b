// utilities_for_testing.hpp -- the reference's pose-equality criterion (include/.../utilities_for_testing.hpp:4-11).
#pragma once

#include <cmath>

#include "kitti_motion_compensation/data_types.hpp"

namespace kmc::utilities_for_testing {

inline bool FloatEqual(float const a, float const b, float const epsilon = 1e-10f) { return std::fabs(a - b) <= epsilon; }

inline bool TransformationMatricesAreTheSame(Affine3d const& tf1, Affine3d const& tf2) {
  // (tf1 * tf2^-1) must be the identity
  Matrix4d const I{(tf1 * tf2.inverse()).matrix()};
  return FloatEqual(static_cast<float>(I.trace()), 4.0f) && FloatEqual(static_cast<float>(I.sum() - I.trace()), 0.0f);
}

}  // namespace kmc::utilities_for_testing

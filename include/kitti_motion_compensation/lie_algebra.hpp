// lie_algebra.hpp -- kept so that `#include "kitti_motion_compensation/lie_algebra.hpp"` written against the reference still resolves.
// The declarations (kmc::lie::Hat, Vee, Exp, Log, LeftJacobian, InverseLeftJacobian) live in host_math.hpp.
#pragma once
#include "kitti_motion_compensation/host_math.hpp"

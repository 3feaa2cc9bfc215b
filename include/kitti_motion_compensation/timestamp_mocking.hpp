// timestamp_mocking.hpp -- kept so that `#include "kitti_motion_compensation/timestamp_mocking.hpp"` written against the reference still resolves.
// The declarations (kmc::FractionOfScanCompleted, GetPseudoTimeStamp, GetPseudoTimeStamps) live in host_math.hpp.
#pragma once
#include "kitti_motion_compensation/host_math.hpp"

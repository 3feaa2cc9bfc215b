// utils.hpp -- kept so that `#include "kitti_motion_compensation/utils.hpp"` written against the reference still resolves.
// The declarations (kmc::IdToZeroPaddedString, TokenizeString, MmHhSsToSeconds) live in host_math.hpp.
#pragma once
#include "kitti_motion_compensation/host_math.hpp"

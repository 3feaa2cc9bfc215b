// utils.hpp -- string helpers the KITTI readers need (reference include/.../utils.hpp, src/.../utils.cpp:10-38).
#pragma once

#include <string>
#include <vector>

#include "kitti_motion_compensation/data_types.hpp"

namespace kmc {

std::string IdToZeroPaddedString(std::size_t const id, std::size_t const pad = 10);  // utils.cpp:10-15
std::vector<std::string> TokenizeString(std::string raw_string);                     // :17-29
double MmHhSsToSeconds(std::string const mm_hh_ss);                                  // :31-38

}  // namespace kmc

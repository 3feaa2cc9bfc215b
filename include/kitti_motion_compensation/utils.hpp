// utils.hpp -- string helpers of the KITTI readers: the reference's include/kitti_motion_compensation/utils.hpp
// (src/kitti_motion_compensation/utils.cpp:10-38), without the OpenCV image helpers of :44-54 (out of scope).
#pragma once

#include <cstddef>
#include <string>
#include <vector>

namespace kmc {

std::string IdToZeroPaddedString(std::size_t const id, std::size_t const pad = 10);  // utils.cpp:10-15
std::vector<std::string> TokenizeString(std::string raw_string);                     // :17-29
double MmHhSsToSeconds(std::string const mm_hh_ss);                                  // :31-38  "HH:MM:SS.nnnnnnnnn" -> seconds since midnight

}  // namespace kmc

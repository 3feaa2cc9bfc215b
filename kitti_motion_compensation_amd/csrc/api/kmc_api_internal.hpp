// kmc_api_internal.hpp -- glue between the public Eigen-free types and the host math / C-ABI (library-internal).
#pragma once

#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

#include "kitti_motion_compensation/data_types.hpp"
#include "kmc_hip.h"
#include "../kmc_host_math.hpp"

namespace kmc::detail {

inline kmc_host::Vec3 to_host(Vector3d const& a) { return {a(0), a(1), a(2)}; }
inline Vector3d from_host(kmc_host::Vec3 const& a) { return {a.x, a.y, a.z}; }
inline kmc_host::Mat3 to_host(Matrix3d const& a) {
  kmc_host::Mat3 m;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) m.m[i][j] = a(i, j);
  return m;
}
inline Matrix3d from_host(kmc_host::Mat3 const& a) {
  Matrix3d m;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) m(i, j) = a.m[i][j];
  return m;
}
inline kmc_host::Pose to_host(Affine3d const& T) { return {to_host(T.linear()), to_host(T.translation())}; }
inline Affine3d from_host(kmc_host::Pose const& P) {
  Affine3d T;
  T.linear() = from_host(P.L);
  T.translation() = from_host(P.t);
  return T;
}

// The reference's failure mode for a time outside the interpolation range: assert(...) kept in release builds
// (trajectory_interpolation.cpp:9-10, :32) -> message on stderr + abort().
[[noreturn]] inline void die_time_out_of_range(const char* where) {
  std::fprintf(stderr,
               "%s: Assertion `TimeIsInRange(time) and \"You gave a time outside of the two poses you wanted to "
               "interpolate between :(\"' failed.\n",
               where);
  std::abort();
}

[[noreturn]] inline void throw_status(int status, const char* where, kmc_ctx* ctx = nullptr) {
  std::string msg = std::string(where) + ": " + kmc_status_string(status);
  if (status == KMC_ERR_HIP && ctx) msg += std::string(" [") + kmc_hip_last_error(ctx) + "]";
  throw std::runtime_error(msg);
}

// one device context per thread (lazy); throws std::runtime_error when there is no HIP device
kmc_ctx* thread_context();
void adopt_thread_context(kmc_ctx* c, int device);  // a context created on a helper thread becomes this thread's (kmc_api_deskew.cpp)
bool thread_has_context();

}  // namespace kmc::detail

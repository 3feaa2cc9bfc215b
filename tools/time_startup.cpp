// time_startup.cpp -- what a process pays before its first frame (round 5, for kmc::MotionCompensateRun's start-up): wall-clock
// milliseconds of the first page-locked allocation (= the HIP runtime's start-up), the first kmc_hip_create, a second one, the first
// device-resident deskew call (code-object load + launch) and a second call.  One JSON line.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#include "kmc_hip.h"

static double ms_since(std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); }

int main() {
  using clk = std::chrono::steady_clock;
  auto t = clk::now();
  void* pinned = nullptr;
  int rc = kmc_host_pool_alloc(16u << 20, &pinned);
  const double first_pin = ms_since(t);
  t = clk::now();
  void* pinned2 = nullptr;
  kmc_host_pool_alloc(16u << 20, &pinned2);
  const double second_pin = ms_since(t);
  t = clk::now();
  kmc_ctx* a = nullptr;
  rc |= kmc_hip_create(&a, 0);
  const double create1 = ms_since(t);
  t = clk::now();
  kmc_ctx* b = nullptr;
  rc |= kmc_hip_create(&b, 0);
  const double create2 = ms_since(t);
  if (rc != KMC_OK) { std::fprintf(stderr, "setup failed: %d\n", rc); return 2; }
  const uint64_t n = 123397;
  kmc_frame_params p;
  std::memset(&p, 0, sizeof(p));
  p.twist[0] = 1.3; p.twist[5] = 0.03; p.x_req = 0.5;
  float *in = (float*)pinned, *out = (float*)pinned2;
  kmc_synth_points_host(in, n, 7);
  t = clk::now();
  rc = kmc_hip_deskew_f32(a, in, out, n, &p, KMC_MEM_HOST, nullptr);
  const double call1 = ms_since(t);
  t = clk::now();
  rc |= kmc_hip_deskew_f32(a, in, out, n, &p, KMC_MEM_HOST, nullptr);
  const double call2 = ms_since(t);
  t = clk::now();
  kmc_hip_destroy(b);
  kmc_hip_destroy(a);
  const double destroy = ms_since(t);
  std::printf("{\"ms\": {\"first_16MiB_page_locked_block_incl_HIP_start_up\": %.2f, \"second_16MiB_block\": %.2f, \"first_kmc_hip_create\": %.2f, \"second_kmc_hip_create\": %.2f, "
              "\"first_in_place_call_incl_code_object_load\": %.2f, \"second_call\": %.3f, \"destroy_both\": %.2f}, \"rc\": %d}\n",
              first_pin, second_pin, create1, create2, call1, call2, destroy, rc);
  return rc == KMC_OK ? 0 : 1;
}

// time_hsa_init.cpp -- where a process's GPU start-up goes (round 5, for kmc::MotionCompensateRun): milliseconds of hsa_init() alone
// (ROCr opens the kernel driver, reads the topology, maps the doorbells / scratch), of hipInit(0) behind it, of the first call that
// creates the HIP device context (hipSetDevice + hipFree(0)) and of the first page-locked allocation.  One JSON line; run it in a fresh
// process per sample (the first hsa_init of a process is the one that counts).  `time_hsa_init hip` skips the explicit hsa_init, so the
// HIP runtime pays it inside hipInit -- the two lines together say how much of "HIP start-up" is ROCr / the kernel driver.
#include <hip/hip_runtime_api.h>
#include <hsa/hsa.h>

#include <chrono>
#include <cstdio>
#include <cstring>

static double ms_since(std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); }

int main(int argc, char** argv) {
  using clk = std::chrono::steady_clock;
  const bool hip_only = argc > 1 && !std::strcmp(argv[1], "hip");
  double hsa_ms = 0;
  auto t = clk::now();
  if (!hip_only) {
    if (hsa_init() != HSA_STATUS_SUCCESS) { std::fprintf(stderr, "hsa_init failed\n"); return 2; }
    hsa_ms = ms_since(t);
  }
  t = clk::now();
  int rc = hipInit(0);
  const double hipinit_ms = ms_since(t);
  t = clk::now();
  rc |= hipSetDevice(0);
  rc |= hipFree(nullptr);
  const double device_ms = ms_since(t);
  t = clk::now();
  void* p = nullptr;
  rc |= hipHostMalloc(&p, 16u << 20, hipHostMallocDefault);
  const double pin_ms = ms_since(t);
  t = clk::now();
  hipStream_t s;
  rc |= hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  const double stream_ms = ms_since(t);
  std::printf("{\"mode\": \"%s\", \"ms\": {\"hsa_init\": %.2f, \"hipInit\": %.2f, \"hipSetDevice_hipFree0\": %.2f, \"first_16MiB_hipHostMalloc\": %.2f, \"first_stream\": %.2f}, \"rc\": %d}\n",
              hip_only ? "hip" : "hsa_then_hip", hsa_ms, hipinit_ms, device_ms, pin_ms, stream_ms, rc);
  std::fflush(stdout);
  std::_Exit(rc ? 1 : 0);
}

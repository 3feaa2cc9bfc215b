// alias_probe.hip -- what does the library's hazard check do with ALIASED mappings?  
// The check that lets independent frames overlap (barrier-free dispatch, gathered lists) compares VIRTUAL ADDRESS RANGES
// (kmc_dispatch_book.hpp).  Here one physical buffer M is mapped twice (hipMemCreate + two hipMemMap: m1, m2) and used as the link of a
// two-frame chain:   frame 1: A -> m1 (n points)      frame 2: the LAST 8192 points of m2 -> C
// (a true read-after-write dependency the library cannot see; frame 2 reads what frame 1's last waves write, so an early frame 2 is as
// visible as it can be).  M is poisoned before every repetition, so a frame 2 that runs early reads poison.  Counted per configuration: repetitions whose C
// differs from the serial result.
//   default            barrier-free dispatch where probed       -> the contract says: undefined without the caller's own ordering
//   KMC_ANY_ORDER=0    every dispatch ordered                   -> always right
//   gathered           kmc_hip_set_frame_queues(ctx, 4)         -> both frames land in ONE list launch: undefined
//   gathered + join    kmc_hip_frame_queue_join between the two -> always right (the caller's own ordering, as the contract asks)
//   alias_probe [points=1000000] [repetitions=100]   -> one JSON object; exit code 0 iff the two ordered configurations never differ
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "kmc_hip.h"

#define CHECK(x)                                                                              \
  do {                                                                                        \
    hipError_t e_ = (x);                                                                      \
    if (e_ != hipSuccess) {                                                                   \
      std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      std::printf("{\"skipped\": \"%s failed: %s\"}\n", #x, hipGetErrorString(e_));          \
      std::exit(3);                                                                           \
    }                                                                                         \
  } while (0)

int main(int argc, char** argv) {
  const uint64_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 1000000ull;
  const int reps = argc > 2 ? std::atoi(argv[2]) : 100;
  const uint64_t k2 = std::min<uint64_t>(8192, n);  // frame 2's points: the tail of M
  CHECK(hipSetDevice(0));
  hipMemAllocationProp prop;
  std::memset(&prop, 0, sizeof(prop));
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gran = 0;
  CHECK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
  const size_t bytes = ((n * 16 + gran - 1) / gran) * gran;
  hipMemGenericAllocationHandle_t handle;
  CHECK(hipMemCreate(&handle, bytes, &prop, 0));
  void *m1 = nullptr, *m2 = nullptr;
  hipMemAccessDesc access;
  std::memset(&access, 0, sizeof(access));
  access.location = prop.location;
  access.flags = hipMemAccessFlagsProtReadWrite;
  for (void** va : {&m1, &m2}) {
    CHECK(hipMemAddressReserve(va, bytes, gran, nullptr, 0));
    CHECK(hipMemMap(*va, bytes, 0, handle, 0));
    CHECK(hipMemSetAccess(*va, bytes, &access, 1));
  }
  float *A = nullptr, *C = nullptr;
  CHECK(hipMalloc((void**)&A, n * 16));
  CHECK(hipMalloc((void**)&C, n * 16));
  {  // the two mappings really are one buffer
    CHECK(hipMemset(m1, 0x5A, 64));
    unsigned char probe[64];
    CHECK(hipMemcpy(probe, m2, 64, hipMemcpyDeviceToHost));
    if (probe[0] != 0x5A || probe[63] != 0x5A) { std::printf("{\"skipped\": \"the two mappings do not alias\"}\n"); return 3; }
  }
  kmc_frame_params p1, p2;
  std::memset(&p1, 0, sizeof(p1));
  p1.twist[0] = 1.3; p1.twist[1] = 0.05; p1.twist[5] = 0.03; p1.x_req = 0.5;
  p2 = p1;
  p2.twist[0] = 0.9; p2.twist[5] = -0.02; p2.x_req = 0.25;
  struct Cfg { const char* name; const char* any_order_env; int queues; bool join_between; };
  const Cfg cfgs[] = {{"default", nullptr, 1, false}, {"KMC_ANY_ORDER=0", "0", 1, false}, {"gathered", nullptr, 4, false}, {"gathered_plus_join_between", nullptr, 4, true}};
  std::vector<float> want(4 * n), got(4 * n);
  std::printf("{\"points\": %llu, \"repetitions\": %d, \"mapping_granularity\": %zu, \"configurations\": {", (unsigned long long)n, reps, gran);
  int ordered_mismatches = 0;
  bool first = true;
  for (const Cfg& cfg : cfgs) {
    if (cfg.any_order_env) setenv("KMC_ANY_ORDER", cfg.any_order_env, 1);
    kmc_ctx* ctx = nullptr;
    if (kmc_hip_create(&ctx, 0) != KMC_OK) return 2;
    unsetenv("KMC_ANY_ORDER");
    kmc_device_info info;
    kmc_hip_device_info(ctx, &info);
    kmc_hip_synth_points(ctx, A, n, 4242);
    // the serial result
    kmc_hip_deskew_f32(ctx, A, (float*)m1, n, &p1, KMC_MEM_DEVICE, nullptr);
    kmc_hip_synchronize(ctx);
    kmc_hip_deskew_f32(ctx, (const float*)m2 + 4 * (n - k2), C, k2, &p2, KMC_MEM_DEVICE, nullptr);
    kmc_hip_synchronize(ctx);
    CHECK(hipMemcpy(want.data(), C, k2 * 16, hipMemcpyDeviceToHost));
    if (cfg.queues > 1) kmc_hip_set_frame_queues(ctx, cfg.queues);
    int mismatches = 0;
    for (int r = 0; r < reps; ++r) {
      kmc_hip_synchronize(ctx);
      CHECK(hipMemset(m1, 0x7F, n * 16));  // poison (a NaN pattern): what a frame 2 that runs early reads
      CHECK(hipMemset(C, 0, n * 16));
      CHECK(hipDeviceSynchronize());
      // an ordinary frame first, so that the two frames of interest both sit INSIDE the any-order window
      kmc_hip_deskew_f32(ctx, A, C, 64, &p1, KMC_MEM_DEVICE, nullptr);
      kmc_hip_deskew_f32(ctx, A, (float*)m1, n, &p1, KMC_MEM_DEVICE, nullptr);
      if (cfg.join_between) kmc_hip_frame_queue_join(ctx);
      kmc_hip_deskew_f32(ctx, (const float*)m2 + 4 * (n - k2), C, k2, &p2, KMC_MEM_DEVICE, nullptr);
      kmc_hip_synchronize(ctx);
      CHECK(hipMemcpy(got.data(), C, k2 * 16, hipMemcpyDeviceToHost));
      if (std::memcmp(got.data(), want.data(), k2 * 16) != 0) ++mismatches;
    }
    const bool ordered = cfg.any_order_env != nullptr || cfg.join_between;
    if (ordered) ordered_mismatches += mismatches;
    std::printf("%s\"%s\": {\"repetitions_that_differ_from_the_serial_result\": %d, \"the_caller_ordered_the_aliased_frames\": %s, \"any_order_dispatch_verdict\": %d}", first ? "" : ", ",
                cfg.name, mismatches, ordered ? "true" : "false", info.any_order_dispatch);
    first = false;
    kmc_hip_destroy(ctx);
  }
  std::printf("}, \"contract\": \"hazards are judged on virtual address ranges: aliased mappings must be ordered by the caller (include/kmc_hip.h)\"}\n");
  return ordered_mismatches == 0 ? 0 : 1;
}

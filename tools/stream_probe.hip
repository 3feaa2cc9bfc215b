// stream_probe -- how fast does a STREAM OF SEPARATE FRAMES go through the C-ABI's single-frame entry point?
// (BASELINE.json configs[1] literally: one 1 M-point frame per call; VERDICT r01 "weak" #5.)
//   stream_probe [points_per_frame=1000000] [launches=4000]
// Modes measured, all through kmc_hip_deskew_f32 on device-resident buffers (24 rotating pairs: HBM, not MALL):
//   one context                 -- launches back to back on one HIP stream (one hardware queue): every launch waits for the
//                                  previous kernel's last wave (the AQL barrier bit), so head and tail of neighbours never overlap
//   K contexts round-robin      -- K hardware queues: the frames are independent, neighbours overlap
//   kmc_hip_deskew_frames_f32   -- the library's own frame queue (if built): same idea behind one call
// Prints one JSON object.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kmc_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
#define KC(x) do { int r_ = (x); if (r_ != KMC_OK) { std::fprintf(stderr, "%s: %s\n", #x, kmc_status_string(r_)); std::exit(1); } } while (0)

int main(int argc, char** argv) {
  const uint64_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 1000000ull;
  const int launches = argc > 2 ? std::atoi(argv[2]) : 4000;
  const int kPairs = n <= 2000000 ? 24 : 6;
  constexpr int kMaxCtx = 8;
  kmc_ctx* ctx[kMaxCtx];
  for (auto& c : ctx) KC(kmc_hip_create(&c, 0));
  std::vector<float*> in(kPairs), out(kPairs);
  for (int k = 0; k < kPairs; ++k) {
    CK(hipMalloc((void**)&in[k], n * 16));
    CK(hipMalloc((void**)&out[k], n * 16));
    KC(kmc_hip_synth_points(ctx[0], in[k], n, 100 + k));
  }
  KC(kmc_hip_synchronize(ctx[0]));
  kmc_frame_params p;
  const double twist[6] = {1.3, 0.05, -0.02, 0.002, -0.004, 0.03};
  for (int i = 0; i < 6; ++i) p.twist[i] = twist[i];
  p.x_req = 0.5;
  using clk = std::chrono::steady_clock;
  auto run = [&](int K) {
    auto pass = [&](int count) {
      for (int i = 0; i < count; ++i) KC(kmc_hip_deskew_f32(ctx[i % K], in[i % kPairs], out[i % kPairs], n, &p, KMC_MEM_DEVICE, nullptr));
      for (int k = 0; k < K; ++k) KC(kmc_hip_synchronize(ctx[k]));
    };
    pass(200);
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
      const auto t0 = clk::now();
      pass(launches);
      best = std::min(best, std::chrono::duration<double>(clk::now() - t0).count());
    }
    return best / launches * 1e6;  // us per frame
  };
  std::printf("{\"points_per_frame\": %llu, \"launches\": %d", (unsigned long long)n, launches);
  for (int K : {1, 2, 3, 4, 8}) {
    const double us = run(K);
    std::printf(", \"contexts_%d\": {\"us_per_frame\": %.3f, \"GBps\": %.1f}", K, us, 32.0 * n / us / 1e3);
  }
  {
    std::vector<const float*> ins(launches);
    std::vector<float*> outs(launches);
    std::vector<uint64_t> ns(launches, n);
    std::vector<kmc_frame_params> ps(launches, p);
    for (int i = 0; i < launches; ++i) { ins[i] = in[i % kPairs]; outs[i] = out[i % kPairs]; }
    for (int depth : {1, 2, 3, 4}) {
      KC(kmc_hip_set_frame_queues(ctx[0], depth));
      KC(kmc_hip_deskew_frames_f32(ctx[0], ins.data(), outs.data(), ns.data(), ps.data(), 200, nullptr));
      KC(kmc_hip_synchronize(ctx[0]));
      double best = 1e30;
      for (int rep = 0; rep < 3; ++rep) {
        const auto t0 = clk::now();
        KC(kmc_hip_deskew_frames_f32(ctx[0], ins.data(), outs.data(), ns.data(), ps.data(), (uint32_t)launches, nullptr));
        KC(kmc_hip_synchronize(ctx[0]));
        best = std::min(best, std::chrono::duration<double>(clk::now() - t0).count());
      }
      const double us = best / launches * 1e6;
      std::printf(", \"frame_queue_%d\": {\"us_per_frame\": %.3f, \"GBps\": %.1f}", depth, us, 32.0 * n / us / 1e3);
    }
  }
  std::printf("}\n");
  for (auto& c : ctx) kmc_hip_destroy(c);
  return 0;
}

// experiment: frame queues vs how many streams exist before them (HW queue mapping)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "kmc_hip.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
#define KC(x) do { int r_ = (x); if (r_ != KMC_OK) { std::fprintf(stderr, "%s: %s\n", #x, kmc_status_string(r_)); std::exit(1); } } while (0)
int main(int argc, char** argv) {
  const int dummies = argc > 1 ? std::atoi(argv[1]) : 0;
  const uint64_t n = 1000000;
  const int launches = 3000, kPairs = 24;
  std::vector<hipStream_t> dummy(dummies);
  for (auto& s : dummy) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  kmc_ctx* ctx;
  KC(kmc_hip_create(&ctx, 0));
  std::vector<float*> in(kPairs), out(kPairs);
  for (int k = 0; k < kPairs; ++k) { CK(hipMalloc((void**)&in[k], n * 16)); CK(hipMalloc((void**)&out[k], n * 16)); KC(kmc_hip_synth_points(ctx, in[k], n, 100 + k)); }
  kmc_frame_params p; const double tw[6] = {1.3, 0.05, -0.02, 0.002, -0.004, 0.03}; for (int i = 0; i < 6; ++i) p.twist[i] = tw[i]; p.x_req = 0.5;
  std::vector<const float*> ins(launches); std::vector<float*> outs(launches); std::vector<uint64_t> ns(launches, n); std::vector<kmc_frame_params> ps(launches, p);
  for (int i = 0; i < launches; ++i) { ins[i] = in[i % kPairs]; outs[i] = out[i % kPairs]; }
  std::printf("dummies=%d prio=%s:", dummies, std::getenv("KMC_FQ_PRIORITY") ? std::getenv("KMC_FQ_PRIORITY") : "0");
  for (int depth : {2, 3, 4}) {
    KC(kmc_hip_set_frame_queues(ctx, depth));
    KC(kmc_hip_deskew_frames_f32(ctx, ins.data(), outs.data(), ns.data(), ps.data(), 300, nullptr)); KC(kmc_hip_synchronize(ctx));
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
      auto t0 = std::chrono::steady_clock::now();
      KC(kmc_hip_deskew_frames_f32(ctx, ins.data(), outs.data(), ns.data(), ps.data(), launches, nullptr)); KC(kmc_hip_synchronize(ctx));
      best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    std::printf("  q%d %.2f us", depth, best / launches * 1e6);
  }
  std::printf("\n");
  kmc_hip_destroy(ctx);
}

// time_frame_stream.hip -- the reference's calling pattern (handlers.cpp:55-64: one frame at a time, each in its own allocation)
// driven from C++ through the C-ABI, so that the host language does not set the pace (a ctypes call costs ~8 us, a launch ~2).
//
//   time_frame_stream <n_frames> <points_per_frame | kitti> [sets=1] [iters=20] [carve]
//
// n_frames device-resident frames per set, every frame its OWN hipMalloc allocation (in and out); `kitti` draws the sizes from
// N(121 000, 3 000) clipped to [90 000, 140 000] like bench.py's configs2_drive leg; `sets` rotating sets of such frames (so that a
// sweep's working set exceeds the 256 MiB Infinity Cache when the frames are small).  Timed with HIP events on the context's stream
// (kmc_hip_timer_begin / _end, which also join the frame queues), `iters` sweeps over all sets' frames after two warm-up sweeps:
//   per_call            one kmc_hip_deskew_f32 call per frame, in order on a DEFAULT context's own stream: one HIP launch per frame (frames that
//                       share no buffer with one in flight without the barrier bit, where the run-time probe verified that)
//   per_call_direct_queue   the same calls on a context that opted in with kmc_hip_set_direct_dispatch(ctx, 1): AQL packets the library writes
//                       itself, below the HIP runtime's launch path (this client waits with kmc_hip_synchronize / kmc_hip_timer_end only)
//   per_call_nknot3     one kmc_hip_deskew_traj_f32 call per frame (three knots, the records in the argument block), default context;
//                       _direct_queue: its twin on the opted-in context, compared bit for bit
//   per_call_drained    the same calls on a context created with KMC_ANY_ORDER=0 (every dispatch carries the barrier bit)
//   per_call_gathered   the same calls with kmc_hip_set_frame_queues(ctx, 4): the library gathers them into list launches of up to 16 frames
//   list_one_launch     kmc_hip_deskew_frames_f32: the set's frames handed over as ONE list (the key keeps its round-4 name; since round 5 a
//                       list of more than 16 frames goes out as chained kernel-argument launches of 16 frames, barrier-free where verified)
//   batch_packed        kmc_hip_deskew_batch_f32 on the same frames packed into one buffer (the ceiling for this frame mix)
// plus the host's own time per call (steady_clock around the issuing loop) and a bit-for-bit comparison of what the list kernel and
// the per-frame kernel wrote for every frame.  Prints one JSON object.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "kmc_hip.h"

#define HIP_OK(x)                                                                          \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) {                                                                \
      std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      std::exit(2);                                                                        \
    }                                                                                      \
  } while (0)
#define KMC_OK_OR_DIE(x)                                                                         \
  do {                                                                                           \
    int rc_ = (x);                                                                               \
    if (rc_ != KMC_OK) {                                                                         \
      std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, kmc_status_string(rc_)); \
      std::exit(3);                                                                              \
    }                                                                                            \
  } while (0)

struct Set {
  std::vector<float*> in, out;
  std::vector<const float*> cin;
  float *packed_in = nullptr, *packed_out = nullptr;
};

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: time_frame_stream <n_frames> <points_per_frame | kitti> [sets] [iters]\n");
    return 2;
  }
  const uint32_t F = (uint32_t)std::atoi(argv[1]);
  const bool kitti = std::strcmp(argv[2], "kitti") == 0;
  const int n_sets = argc > 3 ? std::atoi(argv[3]) : 1;
  const int iters = argc > 4 ? std::atoi(argv[4]) : 20;
  const bool carve = argc > 5 && std::strcmp(argv[5], "carve") == 0;  // frames carved out of ONE allocation per set (1 KiB-aligned starts) instead of one hipMalloc each
  std::vector<uint64_t> sizes(F), offsets(F + 1, 0);
  {
    std::mt19937 rng(0x4B4D43 + 2);
    std::normal_distribution<double> nd(121000.0, 3000.0);
    for (uint32_t f = 0; f < F; ++f) {
      sizes[f] = kitti ? (uint64_t)std::min(140000.0, std::max(90000.0, nd(rng))) : std::strtoull(argv[2], nullptr, 10);
      offsets[f + 1] = offsets[f] + sizes[f];
    }
  }
  const uint64_t total = offsets[F];

  kmc_ctx *ctx = nullptr, *drained = nullptr, *direct = nullptr;
  (void)kmc_hip_bind_thread_near_device(0);  // a placement hint: the calling thread on the GPU's NUMA node (what numactl does for a deployment)
  KMC_OK_OR_DIE(kmc_hip_create(&ctx, 0));
  KMC_OK_OR_DIE(kmc_hip_create(&direct, 0));
  KMC_OK_OR_DIE(kmc_hip_set_direct_dispatch(direct, 1));  // this client plays by the queue's rules: it waits through the context only
  setenv("KMC_ANY_ORDER", "0", 1);
  KMC_OK_OR_DIE(kmc_hip_create(&drained, 0));
  unsetenv("KMC_ANY_ORDER");
  kmc_device_info info;
  KMC_OK_OR_DIE(kmc_hip_device_info(ctx, &info));

  std::vector<kmc_frame_params> params(F);
  for (uint32_t f = 0; f < F; ++f) {  // a turning, accelerating vehicle; every frame its own twist and request time
    const double k = 1.0 + 0.001 * (f % 97);
    const double tw[6] = {1.3 * k, 0.05, -0.02, 0.002, -0.004, 0.03 * k};
    std::memcpy(params[f].twist, tw, sizeof(tw));
    params[f].x_req = 0.25 + 0.5 * ((f % 11) / 10.0);
  }
  std::vector<Set> sets(n_sets);
  for (int s = 0; s < n_sets; ++s) {
    Set& S = sets[s];
    S.in.resize(F); S.out.resize(F); S.cin.resize(F);
    HIP_OK(hipMalloc((void**)&S.packed_in, std::max<uint64_t>(total, 1) * 16));
    HIP_OK(hipMalloc((void**)&S.packed_out, std::max<uint64_t>(total, 1) * 16));
    float *pool_in = nullptr, *pool_out = nullptr;
    uint64_t pool_pts = 0, cursor = 0;
    for (uint32_t f = 0; f < F; ++f) pool_pts += (sizes[f] + 63) / 64 * 64;
    if (carve) {
      HIP_OK(hipMalloc((void**)&pool_in, std::max<uint64_t>(pool_pts, 1) * 16));
      HIP_OK(hipMalloc((void**)&pool_out, std::max<uint64_t>(pool_pts, 1) * 16));
    }
    for (uint32_t f = 0; f < F; ++f) {
      if (carve) {
        S.in[f] = pool_in + 4 * cursor;
        S.out[f] = pool_out + 4 * cursor;
        cursor += (sizes[f] + 63) / 64 * 64;
      } else {
        HIP_OK(hipMalloc((void**)&S.in[f], sizes[f] * 16));
        HIP_OK(hipMalloc((void**)&S.out[f], sizes[f] * 16));
      }
      S.cin[f] = S.in[f];
      KMC_OK_OR_DIE(kmc_hip_synth_points(ctx, S.in[f], sizes[f], 0x4B4D43ull + 0xF5000000ull + (uint64_t)s * F + f));
      KMC_OK_OR_DIE(kmc_hip_synchronize(ctx));
      HIP_OK(hipMemcpy(S.packed_in + 4 * offsets[f], S.in[f], sizes[f] * 16, hipMemcpyDeviceToDevice));
    }
  }
  HIP_OK(hipDeviceSynchronize());

  double host_us = 0;      // the issuing loop's own time per frame, last mode measured
  double ao_share_last = 0;  // share of the timed region's frames that went out without the barrier bit, last mode measured
  double dd_share_last = 0;  // share of the timed region's frames that went out through the context's direct queue (AQL packets below the HIP runtime)
  auto timed = [&](kmc_ctx* c, auto&& sweep) {
    // warm-up: at least two sweeps AND ~40 ms of device work -- an idle MI355X needs ~10 ms of launches to ramp its clocks, and every
    // mode here follows a pause (allocation, the bitwise comparison's copies)
    {
      const double t_warm = now_us();
      for (int w = 0; w < 2 || now_us() - t_warm < 40000.0; ++w) {
        for (int s = 0; s < n_sets; ++s) sweep(c, sets[s]);
        KMC_OK_OR_DIE(kmc_hip_synchronize(c));
      }
    }
    const uint64_t ao_before = kmc_hip_any_order_launches(c);
    const uint64_t dd_before = kmc_hip_direct_frames(c);
    KMC_OK_OR_DIE(kmc_hip_timer_begin(c));
    const double t0 = now_us();
    for (int it = 0; it < iters; ++it)
      for (int s = 0; s < n_sets; ++s) sweep(c, sets[s]);
    host_us = (now_us() - t0) / ((double)iters * n_sets * F);
    ao_share_last = (double)(kmc_hip_any_order_launches(c) - ao_before) / ((double)iters * n_sets * F);
    dd_share_last = (double)(kmc_hip_direct_frames(c) - dd_before) / ((double)iters * n_sets * F);
    float ms = 0;
    KMC_OK_OR_DIE(kmc_hip_timer_end(c, &ms));
    return (double)ms * 1e3 / ((double)iters * n_sets * F);  // us per frame
  };
  auto per_call = [&](kmc_ctx* c, Set& S) {
    for (uint32_t f = 0; f < F; ++f) KMC_OK_OR_DIE(kmc_hip_deskew_f32(c, S.in[f], S.out[f], sizes[f], &params[f], KMC_MEM_DEVICE, nullptr));
  };
  auto list = [&](kmc_ctx* c, Set& S) {
    KMC_OK_OR_DIE(kmc_hip_deskew_frames_f32(c, S.cin.data(), S.out.data(), sizes.data(), params.data(), F, nullptr));
  };
  auto batch = [&](kmc_ctx* c, Set& S) {
    KMC_OK_OR_DIE(kmc_hip_deskew_batch_f32(c, S.packed_in, S.packed_out, offsets.data(), F, params.data(), nullptr, KMC_MEM_DEVICE, nullptr));
  };

  // north_star's three bracketing poses, one kmc_hip_deskew_traj_f32 call per frame (every frame its own knots): the records ride in the
  // dispatch's argument block -- as HIP launches on `ctx`, through the direct queue on `direct`
  std::vector<double> knot_times((size_t)F * 3), knot_poses((size_t)F * 36);
  const double T0 = 47072.0;
  for (uint32_t f = 0; f < F; ++f)
    for (int j = 0; j < 3; ++j) {
      knot_times[(size_t)f * 3 + j] = T0 + 0.05 + 0.1 * j;
      const double yaw = (0.02 + 0.0003 * (f % 31)) * j, cy = std::cos(yaw), sy = std::sin(yaw);
      const double P[12] = {cy, -sy, 0, (1.2 + 0.001 * (f % 17)) * j, sy, cy, 0, 0.02 * j * j, 0, 0, 1, 0.001 * j};
      std::memcpy(&knot_poses[((size_t)f * 3 + j) * 12], P, sizeof(P));
    }
  auto per_call_nknot = [&](kmc_ctx* c, Set& S) {
    for (uint32_t f = 0; f < F; ++f)
      KMC_OK_OR_DIE(kmc_hip_deskew_traj_f32(c, S.in[f], S.out[f], sizes[f], &knot_times[(size_t)f * 3], &knot_poses[(size_t)f * 36], 3, T0 + 0.10, T0 + 0.20, T0 + 0.13 + 0.0005 * (f % 40), nullptr,
                                            KMC_MEM_DEVICE, nullptr));
  };
  const double us_nknot_dq = timed(direct, per_call_nknot);
  const double host_nknot_dq = host_us, dd_share_nknot = dd_share_last;
  std::vector<std::vector<float>> want_nknot(F);
  KMC_OK_OR_DIE(kmc_hip_synchronize(direct));
  for (uint32_t f = 0; f < F; ++f) {
    want_nknot[f].resize(4 * sizes[f]);
    HIP_OK(hipMemcpy(want_nknot[f].data(), sets[n_sets - 1].out[f], sizes[f] * 16, hipMemcpyDeviceToHost));
  }
  const double us_call_dq = timed(direct, per_call);
  const double host_call_dq = host_us, dd_share = dd_share_last;
  KMC_OK_OR_DIE(kmc_hip_synchronize(direct));
  // the same calls on the default context: one HIP launch per frame -- what the runtime's launch path costs
  const double us_call = timed(ctx, per_call);
  const double host_call = host_us, ao_share = ao_share_last;
  const double us_nknot = timed(ctx, per_call_nknot);
  const double host_nknot = host_us;
  bool same_nknot = true;  // the direct queue's N-knot frames against the HIP launches', bit for bit
  KMC_OK_OR_DIE(kmc_hip_synchronize(ctx));
  {
    std::vector<float> got;
    for (uint32_t f = 0; f < F; ++f) {
      got.resize(4 * sizes[f]);
      HIP_OK(hipMemcpy(got.data(), sets[n_sets - 1].out[f], sizes[f] * 16, hipMemcpyDeviceToHost));
      same_nknot = same_nknot && std::memcmp(got.data(), want_nknot[f].data(), sizes[f] * 16) == 0;
    }
  }
  const double us_drained = timed(drained, per_call);
  KMC_OK_OR_DIE(kmc_hip_set_frame_queues(ctx, 4));
  const double us_q4 = timed(ctx, per_call);
  const double host_q4 = host_us;
  KMC_OK_OR_DIE(kmc_hip_set_frame_queues(ctx, 1));
  // what the per-frame kernel wrote, for the comparison below
  std::vector<std::vector<float>> want(F);
  KMC_OK_OR_DIE(kmc_hip_synchronize(ctx));
  for (uint32_t f = 0; f < F; ++f) {
    want[f].resize(4 * sizes[f]);
    HIP_OK(hipMemcpy(want[f].data(), sets[0].out[f], sizes[f] * 16, hipMemcpyDeviceToHost));
    HIP_OK(hipMemset(sets[0].out[f], 0xFF, sizes[f] * 16));
  }
  kmc_stats st;
  KMC_OK_OR_DIE(kmc_hip_deskew_frames_f32(ctx, sets[0].cin.data(), sets[0].out.data(), sizes.data(), params.data(), F, &st));
  KMC_OK_OR_DIE(kmc_hip_synchronize(ctx));
  bool same = true;
  {
    std::vector<float> got;
    for (uint32_t f = 0; f < F; ++f) {
      got.resize(4 * sizes[f]);
      HIP_OK(hipMemcpy(got.data(), sets[0].out[f], sizes[f] * 16, hipMemcpyDeviceToHost));
      same = same && std::memcmp(got.data(), want[f].data(), sizes[f] * 16) == 0;
    }
  }
  const double us_list = timed(ctx, list);
  const double host_list = host_us;
  const double us_batch = timed(ctx, batch);

  const double mean_pts = (double)total / F;
  auto gbps = [&](double us) { return 32.0 * mean_pts / us / 1e3; };
  std::printf(
      "{\"frames_per_set\": %u, \"frames_are\": \"%s\", \"sets\": %d, \"iters\": %d, \"mean_points_per_frame\": %.1f, \"points_per_set\": %llu, \"device\": \"%s\", "
      "\"any_order_dispatch\": %d, \"list_launches\": %u, \"list_equals_per_call_bitwise\": %s, "
      "\"per_call\": {\"us_per_frame\": %.3f, \"GBps\": %.1f, \"host_us_per_call\": %.3f, \"dispatched_without_barrier_bit\": %.3f}, "
      "\"per_call_direct_queue\": {\"us_per_frame\": %.3f, \"GBps\": %.1f, \"host_us_per_call\": %.3f, \"through_the_direct_queue\": %.3f}, "
      "\"per_call_nknot3\": {\"us_per_frame\": %.3f, \"GBps\": %.1f, \"host_us_per_call\": %.3f}, "
      "\"per_call_nknot3_direct_queue\": {\"us_per_frame\": %.3f, \"GBps\": %.1f, \"host_us_per_call\": %.3f, \"through_the_direct_queue\": %.3f, \"same_bits_as_hip_launches\": %s}, "
      "\"per_call_drained\": {\"us_per_frame\": %.3f, \"GBps\": %.1f}, "
      "\"per_call_gathered\": {\"us_per_frame\": %.3f, \"GBps\": %.1f, \"host_us_per_call\": %.3f}, "
      "\"list_one_launch\": {\"us_per_frame\": %.3f, \"GBps\": %.1f, \"host_us_per_frame\": %.3f}, "
      "\"batch_packed\": {\"us_per_frame\": %.3f, \"GBps\": %.1f}}\n",
      F, carve ? "carved out of one allocation per set (1 KiB-aligned starts)" : "separate hipMalloc allocations", n_sets, iters, mean_pts, (unsigned long long)total, info.name, info.any_order_dispatch, st.n_launches, same ? "true" : "false", us_call,
      gbps(us_call), host_call, ao_share, us_call_dq, gbps(us_call_dq), host_call_dq, dd_share, us_nknot, gbps(us_nknot), host_nknot, us_nknot_dq, gbps(us_nknot_dq), host_nknot_dq, dd_share_nknot, same_nknot ? "true" : "false",
      us_drained, gbps(us_drained), us_q4, gbps(us_q4), host_q4, us_list, gbps(us_list), host_list, us_batch, gbps(us_batch));
  kmc_hip_destroy(drained);
  kmc_hip_destroy(direct);
  kmc_hip_destroy(ctx);
  return same && same_nknot ? 0 : 1;
}

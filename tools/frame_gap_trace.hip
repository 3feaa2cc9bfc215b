// frame_gap_trace.hip -- WHERE the time between two separately launched frames goes: device-side time stamps.
//
// A 1 M-point frame is 4.7 us of kernel at the batched rate, but a stream of such frames launched one by one takes 6.1-7.0 us per
// frame.  This tool runs the product's own tile body (frame_tile<kSeries3> of kmc_kernels.hip.h, same loads, arithmetic and stores) with
// four time stamps per launch taken ON THE DEVICE from the 100 MHz constant clock (s_memrealtime: the same clock in every CU):
//   first   the earliest start among the launch's first 512 tiles            (the launch has reached the shader engines)
//   filled  the start of tile 8191                                           (8192 waves = the chip's 256 CUs x 32 wave slots are occupied)
//   drain   the end of tile n_tiles - 8192, its store acknowledged           (from here on the chip is running empty)
//   last    the latest end among the launch's last 1024 tiles, stores acknowledged
// and reports, averaged over a stream of frames on ONE stream, for ordinary launches (barrier bit set) and barrier-free ones
// (hipExtAnyOrderLaunch) and -- as the yardstick -- for the same tiles inside one 2-D launch (the frame-list geometry):
//   span    last - first        one frame's own execution
//   fill    filled - first      ramp-up until the chip is full
//   tail    last - drain        ramp-down: the last 8192 tiles
//   gap     first[k+1] - last[k]   nothing of either frame is running (negative: the frames overlap)
//   period  first[k+1] - first[k]  = span + gap: what a frame costs call to call
// A stamped wave (1536 of a 1 M-point frame's 15 625) reads the clock and writes one word of its own.
//   frame_gap_trace [points_per_frame=1000000] [frames=240]        -> one JSON object
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../kitti_motion_compensation_amd/csrc/kmc_kernels.hip.h"

using namespace kmc_dev;

#define CHECK(x)                                                                         \
  do {                                                                                   \
    hipError_t e_ = (x);                                                                 \
    if (e_ != hipSuccess) {                                                              \
      std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      std::exit(2);                                                                      \
    }                                                                                    \
  } while (0)

// per launch: the start stamps of its first kHead tiles, the end stamps of its last kTail tiles, `filled` and `drain`; every stamped
// wave writes its OWN slot with a plain store (atomics on one word would serialise 1536 waves and stretch the kernel threefold)
constexpr uint32_t kHead = 512, kTail = 1024;
struct Stamps { unsigned long long filled, drain, start[kHead], end[kTail]; };

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void stamped_frame(const v4f* __restrict__ in, v4f* __restrict__ out, uint64_t n, FrameRec f,
                                                                                         Stamps* __restrict__ st_base, uint32_t frames_in_launch, FrameRecD d) {
  struct ArgLayout { const v4f* in; v4f* out; uint64_t n; FrameRec f; Stamps* st; uint32_t frames_in_launch; FrameRecD d; };
  const cdouble_p d_rec = (cdouble_p)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(ArgLayout, d));
  // blockIdx.y = frame inside the launch (the 2-D "list" geometry); frame k's buffers follow frame 0's
  const uint64_t tile = blockIdx.x, n_tiles = (n + 63) / 64;
  if (tile >= n_tiles) return;
  const uint32_t fr = blockIdx.y;
  Stamps* st = st_base + fr;
  const bool head_zone = tile < kHead, tail_zone = tile + kTail >= n_tiles;
  unsigned long long t0 = 0;
  if (head_zone || tile == 8191) t0 = wall_clock64();
  frame_tile<kSeries3>(in + (uint64_t)fr * n, out + (uint64_t)fr * n, n, f, 0u, d_rec, tile);
  if (threadIdx.x == 0) {
    if (head_zone) st->start[tile] = t0;
    if (tile == 8191) st->filled = t0;
    if (tail_zone || tile + 8192 == n_tiles) {
      __builtin_amdgcn_s_waitcnt(0);  // the tile's store has been acknowledged
      const unsigned long long t1 = wall_clock64();
      if (tail_zone) st->end[tile + kTail - n_tiles] = t1;
      if (tile + 8192 == n_tiles) st->drain = t1;
    }
  }
}

int main(int argc, char** argv) {
  const uint64_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 1000000ull;
  const int frames = argc > 2 ? std::atoi(argv[2]) : 240;
  CHECK(hipSetDevice(0));
  hipStream_t s;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  v4f *in = nullptr, *out = nullptr;
  CHECK(hipMalloc(&in, (size_t)frames * n * 16));  // every frame its own range: far beyond the 256 MiB Infinity Cache
  CHECK(hipMalloc(&out, (size_t)frames * n * 16));
  {
    std::vector<float> h(4 * n);
    for (uint64_t i = 0; i < n; ++i) {
      h[4 * i] = 5.0f + (float)(i % 977) * 0.07f; h[4 * i + 1] = -30.0f + (float)(i % 3119) * 0.02f; h[4 * i + 2] = -1.5f + (float)(i % 64) * 0.05f; h[4 * i + 3] = 0.25f;
    }
    for (int k = 0; k < frames; ++k) CHECK(hipMemcpy(in + (size_t)k * n, h.data(), n * 16, hipMemcpyHostToDevice));
  }
  Stamps* d_st = nullptr;
  CHECK(hipMalloc(&d_st, frames * sizeof(Stamps)));
  FrameRec f;
  std::memset(&f, 0, sizeof(f));
  f.phi_z = 0.03f; f.phi2 = 0.0009f; f.rho_x = 1.3f; f.rho_y = 0.05f; f.s0 = 0.0f; f.c1_x = -0.0015f; f.c1_y = 0.039f; f.c2_x = -0.00117f; f.pre2 = 0.26f * 1.6925f;
  FrameRecD d;
  std::memset(&d, 0, sizeof(d));
  d.phi[2] = 0.03; d.rho[0] = 1.3; d.rho[1] = 0.05; d.phi2 = 0.0009; d.x_req = 0.5;
  const uint32_t tiles = (uint32_t)((n + 63) / 64);
  std::vector<Stamps> h_st(frames);
  struct Four { unsigned long long first, filled, drain, last; };
  std::vector<Four> h4(frames);

  struct Row { double span, fill, tail, gap, period; };
  auto run = [&](int mode) {  // 0 ordinary launches, 1 barrier-free launches, 2 ONE 2-D launch of 16 frames at a time
    Row best = {1e30, 0, 0, 0, 1e30};
    for (int w = 0; w < 12; ++w)  // ~15 ms of launches: an idle MI355X needs ~10 ms to ramp its clocks
      for (int k = 0; k < frames; ++k) hipLaunchKernelGGL(stamped_frame, dim3(tiles, 1), dim3(64), 0, s, in + (size_t)k * n, out + (size_t)k * n, n, f, d_st + k, 1u, d);
    CHECK(hipStreamSynchronize(s));
    for (int rep = 0; rep < 4; ++rep) {
      CHECK(hipMemset(d_st, 0, frames * sizeof(Stamps)));
      CHECK(hipDeviceSynchronize());
      if (mode == 2) {
        for (int k = 0; k + 16 <= frames; k += 16)
          hipLaunchKernelGGL(stamped_frame, dim3(tiles, 16), dim3(64), 0, s, in + (size_t)k * n, out + (size_t)k * n, n, f, d_st + k, 16u, d);
      } else {
        for (int k = 0; k < frames; ++k)
          hipExtLaunchKernelGGL(stamped_frame, dim3(tiles, 1), dim3(64), 0, s, nullptr, nullptr, (mode == 1 && k) ? (uint32_t)hipExtAnyOrderLaunch : 0u, in + (size_t)k * n,
                                out + (size_t)k * n, n, f, d_st + k, 1u, d);
      }
      CHECK(hipGetLastError());
      CHECK(hipStreamSynchronize(s));
      CHECK(hipMemcpy(h_st.data(), d_st, frames * sizeof(Stamps), hipMemcpyDeviceToHost));
      for (int k = 0; k < frames; ++k) {
        Four q = {~0ull, h_st[k].filled, h_st[k].drain, 0ull};
        for (uint32_t i = 0; i < kHead && i < tiles; ++i) if (h_st[k].start[i]) q.first = std::min(q.first, h_st[k].start[i]);
        for (uint32_t i = 0; i < kTail; ++i) q.last = std::max(q.last, h_st[k].end[i]);
        h4[k] = q;
      }
      const int usable = mode == 2 ? (frames / 16) * 16 : frames;
      Row r = {0, 0, 0, 0, 0};
      int cnt = 0;
      for (int k = usable / 3; k + 1 < usable; ++k) {  // the first third warms the clocks up
        const Four &a = h4[k], &b = h4[k + 1];
        r.span += 10.0 * (double)(a.last - a.first);  // 100 MHz ticks -> ns
        r.fill += a.filled ? 10.0 * (double)(a.filled - a.first) : 0.0;
        r.tail += a.drain ? 10.0 * (double)(a.last - a.drain) : 0.0;
        r.gap += 10.0 * ((double)b.first - (double)a.last);
        r.period += 10.0 * (double)(b.first - a.first);
        ++cnt;
      }
      r.span /= cnt; r.fill /= cnt; r.tail /= cnt; r.gap /= cnt; r.period /= cnt;
      if (r.period < best.period) best = r;
    }
    return best;
  };
  const Row ord = run(0), any = run(1), list = run(2);
  auto pr = [&](const char* name, const Row& r, const char* end) {
    std::printf("\"%s\": {\"period_ns\": %.0f, \"span_ns\": %.0f, \"gap_ns\": %.0f, \"fill_ns\": %.0f, \"tail_ns\": %.0f, \"GBps_call_to_call\": %.1f}%s", name, r.period, r.span, r.gap,
                r.fill, r.tail, 32.0 * n / r.period, end);
  };
  std::printf("{\"points_per_frame\": %llu, \"frames\": %d, \"clock\": \"s_memrealtime, 100 MHz (10 ns per tick)\", ", (unsigned long long)n, frames);
  pr("ordinary_launches", ord, ", ");
  pr("barrier_free_launches", any, ", ");
  pr("inside_one_2d_launch_of_16_frames", list, "}\n");
  return 0;
}

// anyorder_probe.hip -- what hipExtAnyOrderLaunch (a kernel dispatch WITHOUT the AQL barrier bit) does on this runtime and chip.
//
// On one HIP stream every kernel waits for the last wave of the kernel before it; between two 1 M-point frames the chip drains and
// refills (~2 us of a 7 us frame).  Independent frames do not need that order.  This probe measures
//   1. the time per frame of a stream of streaming kernels launched in order vs with hipExtAnyOrderLaunch, on ONE stream;
//   2. whether the things a library relies on still hold after any-order launches: a LONG kernel A followed by a SHORT any-order
//      kernel B, then (i) hipStreamSynchronize, (ii) a D2H copy on the same stream, (iii) an event record + wait, (iv) an ordinary
//      launch C that reads what A wrote, (v) hipDeviceSynchronize -- each must observe A's result.
// Output: one JSON object.  Nothing here uses the library.
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                         \
  do {                                                                                   \
    hipError_t e_ = (x);                                                                 \
    if (e_ != hipSuccess) {                                                              \
      std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      std::exit(2);                                                                      \
    }                                                                                    \
  } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void stream_kernel(const v4f* __restrict__ in, v4f* __restrict__ out, uint64_t n, float s) {
  const uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x;
  if (i < n) {
    v4f v = __builtin_nontemporal_load(in + i);
    v.x = v.x * s + v.y;
    v.y = v.y * s - v.z;
    v.z = v.z * s + v.x;
    __builtin_nontemporal_store(v, out + i);
  }
}

// spins for `ticks` of the 100 MHz constant clock, then writes `value`
__global__ void long_kernel(uint32_t* flag, uint32_t value, uint64_t ticks) {
  const uint64_t t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
__global__ void short_kernel(uint32_t* other, uint32_t value) {
  if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(other, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void copy_flag(const uint32_t* flag, uint32_t* dst) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *dst = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- what the flag does to memory visibility: K writes X with PLAIN stores from every XCD, M reads it through a different
// workgroup -> XCD mapping and counts what it does not find
__global__ __launch_bounds__(256) void fill_plain(uint32_t* x, uint32_t n, uint32_t seed) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) x[i] = seed + i * 2654435761u;
}
__global__ __launch_bounds__(256) void verify_plain(const uint32_t* x, uint32_t n, uint32_t seed, unsigned long long* bad) {
  const uint32_t b = gridDim.x - 1 - blockIdx.x;  // reversed, and shifted by 3 workgroups: another XCD than the writer's
  const uint32_t i = ((b + 3) % gridDim.x) * 256 + threadIdx.x;
  if (i < n && x[i] != seed + i * 2654435761u) atomicAdd(bad, 1ull);
}
__global__ void spin_kernel(uint32_t* flag, uint64_t ticks) {
  const uint64_t t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {
  }
  if (threadIdx.x == 0) atomicAdd(flag, 1u);
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const uint64_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 1000000ull;
  const int frames = argc > 2 ? std::atoi(argv[2]) : 256;
  const int reps = 5;
  CHECK(hipSetDevice(0));
  hipStream_t s;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  v4f *in = nullptr, *out = nullptr;
  CHECK(hipMalloc(&in, (size_t)frames * n * sizeof(v4f)));
  CHECK(hipMalloc(&out, (size_t)frames * n * sizeof(v4f)));
  CHECK(hipMemset(in, 0, (size_t)frames * n * sizeof(v4f)));
  const dim3 grid((unsigned)((n + 63) / 64)), block(64);

  auto run = [&](unsigned flags) {
    double best = 1e30;
    for (int r = 0; r < reps + 1; ++r) {
      CHECK(hipStreamSynchronize(s));
      const double t0 = now_us();
      for (int f = 0; f < frames; ++f) {
        if (flags == 2)
          hipLaunchKernelGGL(stream_kernel, grid, block, 0, s, in + (size_t)f * n, out + (size_t)f * n, n, 1.5f);
        else
          hipExtLaunchKernelGGL(stream_kernel, grid, block, 0, s, nullptr, nullptr, flags, in + (size_t)f * n, out + (size_t)f * n, n, 1.5f);
      }
      CHECK(hipGetLastError());
      CHECK(hipStreamSynchronize(s));
      const double dt = (now_us() - t0) / frames;
      if (r > 0 && dt < best) best = dt;
    }
    return best;
  };
  const double us_plain = run(2), us_ext_ordered = run(0), us_any = run(hipExtAnyOrderLaunch);

  // ---- safety: does everything after an any-order launch still wait for the kernels before it? ----
  uint32_t *flag = nullptr, *other = nullptr, *dst = nullptr, *h = nullptr;
  CHECK(hipMalloc(&flag, 4));
  CHECK(hipMalloc(&other, 4));
  CHECK(hipMalloc(&dst, 4));
  CHECK(hipHostMalloc(&h, 4, hipHostMallocDefault));
  uint32_t* hb = nullptr;
  CHECK(hipHostMalloc(&hb, 4, hipHostMallocDefault));
  double b_visible_us = 0;
  const uint64_t ticks = 3000000;  // 30 ms at 100 MHz
  int ok_sync = 0, ok_copy = 0, ok_event = 0, ok_launch = 0, ok_devsync = 0, trials = 5;
  double overlap_us = 0;
  hipEvent_t ev;
  CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  for (int t = 0; t < trials; ++t) {
    const uint32_t v = 100 + t;
    auto arm = [&] {
      CHECK(hipMemset(flag, 0, 4));
      CHECK(hipMemset(other, 0, 4));
      CHECK(hipMemset(dst, 0, 4));
      *h = 0;
      CHECK(hipDeviceSynchronize());
      hipLaunchKernelGGL(long_kernel, dim3(1), dim3(64), 0, s, flag, v, ticks);
      hipExtLaunchKernelGGL(short_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, other, v);
      CHECK(hipGetLastError());
    };
    auto read_flag = [&] {  // out of band: default-stream copy after a device sync
      uint32_t x = 0;
      CHECK(hipMemcpy(&x, flag, 4, hipMemcpyDeviceToHost));
      return x;
    };
    // (i) stream synchronize; and: did B overlap A at all?  (B's result visible long before A is done)
    arm();
    {
      const double t0 = now_us();
      CHECK(hipStreamSynchronize(s));
      overlap_us += now_us() - t0;
      uint32_t x = 0;
      hipStream_t s2;
      CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
      CHECK(hipMemcpyAsync(&x, flag, 4, hipMemcpyDeviceToHost, s2));
      CHECK(hipStreamSynchronize(s2));
      CHECK(hipStreamDestroy(s2));
      ok_sync += x == v;
    }
    // (ii) D2H copy on the same stream
    arm();
    CHECK(hipMemcpyAsync(h, flag, 4, hipMemcpyDeviceToHost, s));
    CHECK(hipStreamSynchronize(s));
    ok_copy += *h == v;
    // (iii) event record + wait
    arm();
    CHECK(hipEventRecord(ev, s));
    CHECK(hipEventSynchronize(ev));
    ok_event += read_flag() == v;
    // (iv) an ordinary launch behind them
    arm();
    hipLaunchKernelGGL(copy_flag, dim3(1), dim3(64), 0, s, flag, dst);
    CHECK(hipDeviceSynchronize());
    {
      uint32_t x = 0;
      CHECK(hipMemcpy(&x, dst, 4, hipMemcpyDeviceToHost));
      ok_launch += x == v;
    }
    // (v) device synchronize
    arm();
    CHECK(hipDeviceSynchronize());
    ok_devsync += read_flag() == v;
    // (vi) did B run while A was still spinning?  B writes page-locked host memory; the host watches it
    CHECK(hipDeviceSynchronize());
    *hb = 0;
    hipLaunchKernelGGL(long_kernel, dim3(1), dim3(64), 0, s, flag, v, ticks);
    hipExtLaunchKernelGGL(short_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, hb, v);
    {
      const double t0 = now_us();
      while (__atomic_load_n(hb, __ATOMIC_ACQUIRE) != v && now_us() - t0 < 200000.0) {
      }
      b_visible_us += now_us() - t0;
    }
    CHECK(hipDeviceSynchronize());
  }
  // ---- (ix) the same stream of frames with the HOST out of the picture: a 20 ms kernel holds the queue(s) while the host enqueues
  // every frame, events right behind the blocker and behind the last frame time the device alone ----
  double dev_us[3] = {0, 0, 0};  // [one stream ordinary, one stream any-order, four streams ordinary]
  {
    hipStream_t q[4];
    for (auto& x : q) CHECK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    hipEvent_t e0, e1, eq[4], gate;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventCreateWithFlags(&gate, hipEventDisableTiming));
    for (auto& x : eq) CHECK(hipEventCreateWithFlags(&x, hipEventDisableTiming));
    uint32_t* spin_flag = nullptr;
    CHECK(hipMalloc(&spin_flag, 4));
    for (int mode = 0; mode < 3; ++mode) {
      double best = 1e30;
      for (int r = 0; r < 3; ++r) {
        CHECK(hipDeviceSynchronize());
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, spin_flag, (uint64_t)2000000);  // 20 ms
        CHECK(hipEventRecord(e0, s));
        if (mode == 2) {
          CHECK(hipEventRecord(gate, s));
          for (auto& x : q) CHECK(hipStreamWaitEvent(x, gate, 0));
        }
        for (int f = 0; f < frames; ++f) {
          if (mode == 0) hipLaunchKernelGGL(stream_kernel, grid, block, 0, s, in + (size_t)f * n, out + (size_t)f * n, n, 1.5f);
          if (mode == 1) hipExtLaunchKernelGGL(stream_kernel, grid, block, 0, s, nullptr, nullptr, f ? hipExtAnyOrderLaunch : 0u, in + (size_t)f * n, out + (size_t)f * n, n, 1.5f);
          if (mode == 2) hipLaunchKernelGGL(stream_kernel, grid, block, 0, q[f & 3], in + (size_t)f * n, out + (size_t)f * n, n, 1.5f);
        }
        if (mode == 2)
          for (int k = 0; k < 4; ++k) {
            CHECK(hipEventRecord(eq[k], q[k]));
            CHECK(hipStreamWaitEvent(s, eq[k], 0));
          }
        CHECK(hipEventRecord(e1, s));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, (double)ms * 1e3 / frames);
      }
      dev_us[mode] = best;
    }
  }
  // ---- (vii) two spinning kernels of one workgroup each: 10 ms + 10 ms, or 10 ms side by side? ----
  double pair_us[3] = {0, 0, 0};  // [ordinary, ordinary], [any, any], [ordinary, any, any] (last two)
  {
    const uint64_t spin = 1000000;  // 10 ms
    for (int mode = 0; mode < 3; ++mode) {
      CHECK(hipDeviceSynchronize());
      if (mode == 2) hipLaunchKernelGGL(short_kernel, dim3(1), dim3(64), 0, s, other, 1u);
      const double t0 = now_us();
      if (mode == 0) {
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, flag, spin);
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, flag, spin);
      } else {
        hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, flag, spin);
        hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, flag, spin);
      }
      CHECK(hipStreamSynchronize(s));
      pair_us[mode] = now_us() - t0;
    }
  }
  // ---- (viii) visibility of what an any-order kernel wrote with plain stores ----
  unsigned long long bad_ordinary_reader = 0, bad_anyorder_reader = 0, bad_host = 0, bad_both_ordinary = 0;
  {
    const uint32_t m = 64u << 20;  // 64 Mi words = 256 MiB
    uint32_t* x = nullptr;
    unsigned long long* bad = nullptr;
    CHECK(hipMalloc(&x, (size_t)m * 4));
    CHECK(hipMalloc(&bad, 8));
    std::vector<uint32_t> hx(1 << 20);
    const dim3 g(m / 256), b(256);
    for (int mode = 0; mode < 4; ++mode) {
      CHECK(hipMemset(bad, 0, 8));
      CHECK(hipDeviceSynchronize());
      unsigned long long host_bad = 0;
      for (int it = 0; it < 20; ++it) {
        const uint32_t seed = 977u * (it + 1) + mode;
        if (mode == 3) hipLaunchKernelGGL(fill_plain, g, b, 0, s, x, m, seed);
        else hipExtLaunchKernelGGL(fill_plain, g, b, 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, x, m, seed);
        if (mode == 0 || mode == 3) hipLaunchKernelGGL(verify_plain, g, b, 0, s, x, m, seed, bad);
        if (mode == 1) hipExtLaunchKernelGGL(verify_plain, g, b, 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, x, m, seed, bad);
        if (mode == 2) {  // the host reads the tail of X after a stream synchronize
          CHECK(hipStreamSynchronize(s));
          CHECK(hipMemcpy(hx.data(), x + (m - hx.size()), hx.size() * 4, hipMemcpyDeviceToHost));
          for (size_t k = 0; k < hx.size(); ++k) host_bad += hx[k] != seed + (uint32_t)(m - hx.size() + k) * 2654435761u;
        }
      }
      CHECK(hipStreamSynchronize(s));
      unsigned long long v = 0;
      CHECK(hipMemcpy(&v, bad, 8, hipMemcpyDeviceToHost));
      if (mode == 0) bad_ordinary_reader = v;
      if (mode == 1) bad_anyorder_reader = v;
      if (mode == 2) bad_host = host_bad;
      if (mode == 3) bad_both_ordinary = v;
    }
  }
  std::printf("{\"device_side_us_per_frame_host_not_in_the_loop\": {\"one_stream\": %.3f, \"one_stream_any_order\": %.3f, \"four_streams\": %.3f},\n ", dev_us[0],
              dev_us[1], dev_us[2]);
  std::printf(
      "\"two_10ms_kernels_us\": {\"ordinary_ordinary\": %.0f, \"any_any\": %.0f, \"ordinary_then_any_any\": %.0f}, "
      "\"plain_stores_of_an_any_order_kernel\": {\"missed_by_an_ordinary_kernel_behind_it\": %llu, \"missed_by_an_any_order_kernel_behind_it\": %llu, "
      "\"missed_by_the_host_after_stream_sync\": %llu, \"control_both_ordinary\": %llu},\n ",
      pair_us[0], pair_us[1], pair_us[2], bad_ordinary_reader, bad_anyorder_reader, bad_host, bad_both_ordinary);
  std::printf(
      "\"points_per_frame\": %llu, \"frames\": %d, \"us_per_frame_in_order\": %.3f, \"us_per_frame_ext_launch_ordered\": %.3f, "
      "\"us_per_frame_any_order\": %.3f, \"GBps_in_order\": %.1f, \"GBps_any_order\": %.1f, "
      "\"after_any_order_launch\": {\"trials\": %d, \"stream_sync_waits_for_earlier_kernel\": %d, \"same_stream_copy_waits\": %d, "
      "\"event_waits\": %d, \"ordinary_launch_waits\": %d, \"device_sync_waits\": %d, \"mean_stream_sync_us\": %.1f, "
      "\"short_kernel_visible_after_us\": %.1f, \"long_kernel_us\": %.1f}}\n",
      (unsigned long long)n, frames, us_plain, us_ext_ordered, us_any, 32.0 * n / us_plain / 1e3, 32.0 * n / us_any / 1e3, trials, ok_sync,
      ok_copy, ok_event, ok_launch, ok_devsync, overlap_us / trials, b_visible_us / trials, ticks / 100.0);
  return 0;
}

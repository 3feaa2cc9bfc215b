// frame_server_probe.hip -- the design VERDICT r03 #3(b) asked to be tried: a RESIDENT frame server.  Persistent waves that pull
// {in, out, n, frame record} descriptors from a ring and process the frames' tiles without any dispatch packet between frames.
//
// This probe measures the server's UPPER BOUND: all descriptors are already published in device memory when the kernel starts, so
// nothing is ever waited for -- no doorbell polling, no host -> device descriptor traffic, no completion signalling.  What is left is
// the question that decides the design: can resident waves that walk a stream of frames keep up with the hardware dispatcher feeding
// one short-lived wave per tile?  G one-wave workgroups (8 per SIMD: the whole chip) take the tiles of the concatenated frame stream
// round-robin (wave w: tiles w, w + G, w + 2 G, ...; static, no atomics -- a ticket counter would need ~3 G atomics/s on one address),
// each wave keeping a cursor into the descriptor ring; the tile body is the product's own (frame_tile<kSeries3>).  Two variants:
//   plain      load tile, compute, store, next
//   prefetch   the next tile's load is issued before the current tile is computed and stored
// against the product's geometry on the same frames (one 2-D launch: frame x tile, one wave per tile), checked bit for bit.
//   frame_server_probe [frames=256] [points_per_frame=1000000 | kitti] [waves_per_cu=32]        -> one JSON object
// (A gfx9 wave counts loads AND stores in one counter, vmcnt, and they may complete out of order with respect to each other: a
// resident wave must wait for its previous tile's store acknowledgement before it can use its next tile's load -- the short-lived
// wave of the product simply ends behind its store.  That is the structural handicap of every persistent variant measured since round 1.)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
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

struct alignas(16) Desc {  // 96 bytes: one frame of the ring
  FrameRec f;
  const v4f* in;
  v4f* out;
  uint64_t n;
  uint64_t tile_end;  // tiles of this and all earlier frames (the ring's tile space is the concatenation of the frames' tiles)
};
static_assert(sizeof(Desc) == 96, "Desc");

__device__ __forceinline__ Desc load_desc(const Desc* ring, uint32_t k) {  // uniform address: scalar loads
  const v4u __attribute__((address_space(4)))* w = (const v4u __attribute__((address_space(4)))*)(uintptr_t)(ring + k);
  const v4u a[6] = {w[0], w[1], w[2], w[3], w[4], w[5]};
  Desc d;
  __builtin_memcpy(&d, a, sizeof(d));
  return d;
}

template <bool PREFETCH>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void frame_server(const Desc* __restrict__ ring, uint32_t n_desc, uint64_t total_tiles,
                                                                                        const FrameRecD* __restrict__ recs64) {
  const uint32_t tid = threadIdx.x;
  uint32_t k = 0;
  Desc d = load_desc(ring, 0);
  uint64_t begin = 0;  // first tile of descriptor k
  auto seek = [&](uint64_t T) {  // advance the cursor to the frame that owns tile T
    while (T >= d.tile_end && k + 1 < n_desc) {
      begin = d.tile_end;
      d = load_desc(ring, ++k);
    }
  };
  if constexpr (!PREFETCH) {
    for (uint64_t T = blockIdx.x; T < total_tiles; T += gridDim.x) {
      seek(T);
      frame_tile<kSeries3>(d.in, d.out, d.n, d.f, 0u, as_constant(recs64 + opaque_uniform(k)), T - begin);
    }
  } else {
    uint64_t T = blockIdx.x;
    if (T >= total_tiles) return;
    seek(T);
    uint64_t i = (T - begin) * 64 + tid;
    v4f cur = load_point(d.in + (i < d.n ? i : d.n - 1));
    while (true) {
      const Desc dc = d;  // the current tile's frame
      const uint64_t tile = T - begin;
      const uint64_t Tn = T + gridDim.x;
      const bool more = Tn < total_tiles;
      v4f nxt = cur;
      if (more) {
        seek(Tn);
        const uint64_t j = (Tn - begin) * 64 + tid;
        nxt = load_point(d.in + (j < d.n ? j : d.n - 1));  // in flight while `cur` is finished
      }
      const uint64_t base = tile * 64;
      const __amdgpu_buffer_rsrc_t rout = tile_rsrc(dc.out + base, (dc.n - base) * sizeof(v4f));
      const v4f o = deskew_point<kSeries3>(cur, dc.f);
      tile_store(rout, (uint32_t)(tid * sizeof(v4f)), o);  // (the near-origin guard is left out of this variant: never taken on this data)
      if (!more) break;
      cur = nxt;
      T = Tn;
    }
  }
}

// the product's geometry on the same descriptors: one 2-D launch, one wave per tile
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void list_launch(const Desc* __restrict__ ring, const FrameRecD* __restrict__ recs64) {
  const uint32_t fi = blockIdx.y;
  const Desc d = load_desc(ring, fi);
  if ((uint64_t)blockIdx.x * 64 >= d.n) return;
  frame_tile<kSeries3>(d.in, d.out, d.n, d.f, 0u, as_constant(recs64 + opaque_uniform(fi)), blockIdx.x);
}

int main(int argc, char** argv) {
  const int F = argc > 1 ? std::atoi(argv[1]) : 256;
  const bool kitti = argc > 2 && std::strcmp(argv[2], "kitti") == 0;
  const uint64_t fixed = (argc > 2 && !kitti) ? std::strtoull(argv[2], nullptr, 10) : 1000000ull;
  const int waves_per_cu = argc > 3 ? std::atoi(argv[3]) : 32;
  CHECK(hipSetDevice(0));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  hipStream_t s;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  std::mt19937 rng(0x4B4D43 + 2);
  std::normal_distribution<double> nd(121000.0, 3000.0);
  std::vector<uint64_t> sizes(F);
  uint64_t total = 0, max_n = 0;
  for (auto& v : sizes) {
    v = kitti ? (uint64_t)std::min(140000.0, std::max(90000.0, nd(rng))) : fixed;
    total += v;
    max_n = std::max(max_n, v);
  }
  // every frame its own allocation-like range (1 KiB aligned starts) in two sets of buffers: outputs of the server and of the list launch
  std::vector<uint64_t> start(F);
  uint64_t span = 0;
  for (int f = 0; f < F; ++f) { start[f] = span; span += (sizes[f] + 63) / 64 * 64; }
  v4f *in = nullptr, *out_a = nullptr, *out_b = nullptr;
  CHECK(hipMalloc(&in, span * 16));
  CHECK(hipMalloc(&out_a, span * 16));
  CHECK(hipMalloc(&out_b, span * 16));
  {
    std::vector<float> h(4 * span);
    for (uint64_t i = 0; i < span; ++i) {
      h[4 * i] = 5.0f + (float)(i % 977) * 0.07f; h[4 * i + 1] = -30.0f + (float)(i % 3119) * 0.02f; h[4 * i + 2] = -1.5f + (float)(i % 64) * 0.05f; h[4 * i + 3] = (float)(i % 100) * 0.01f;
    }
    CHECK(hipMemcpy(in, h.data(), span * 16, hipMemcpyHostToDevice));
  }
  std::vector<Desc> ring_a(F), ring_b(F);
  std::vector<FrameRecD> recd(F);
  uint64_t tiles = 0;
  for (int f = 0; f < F; ++f) {
    Desc d;
    std::memset(&d, 0, sizeof(d));
    const float k = 1.0f + 0.001f * (f % 97);
    d.f.phi_z = 0.03f * k; d.f.phi2 = d.f.phi_z * d.f.phi_z; d.f.rho_x = 1.3f * k; d.f.rho_y = 0.05f; d.f.s0 = 0.5f - (0.25f + 0.05f * (f % 11));
    d.f.c1_x = -d.f.phi_z * d.f.rho_y; d.f.c1_y = d.f.phi_z * d.f.rho_x; d.f.c2_x = -d.f.phi_z * d.f.c1_y; d.f.c2_y = d.f.phi_z * d.f.c1_x;
    d.f.pre2 = 0.26f * (d.f.rho_x * d.f.rho_x + d.f.rho_y * d.f.rho_y);
    std::memset(&recd[f], 0, sizeof(FrameRecD));
    recd[f].phi[2] = d.f.phi_z; recd[f].rho[0] = d.f.rho_x; recd[f].rho[1] = d.f.rho_y; recd[f].phi2 = d.f.phi2; recd[f].x_req = 0.5 - d.f.s0;
    d.in = in + start[f];
    d.n = sizes[f];
    tiles += (sizes[f] + 63) / 64;
    d.tile_end = tiles;
    d.out = out_a + start[f]; ring_a[f] = d;
    d.out = out_b + start[f]; ring_b[f] = d;
  }
  Desc *d_ring_a = nullptr, *d_ring_b = nullptr;
  FrameRecD* d_recd = nullptr;
  CHECK(hipMalloc(&d_ring_a, F * sizeof(Desc)));
  CHECK(hipMalloc(&d_ring_b, F * sizeof(Desc)));
  CHECK(hipMalloc(&d_recd, F * sizeof(FrameRecD)));
  CHECK(hipMemcpy(d_ring_a, ring_a.data(), F * sizeof(Desc), hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_ring_b, ring_b.data(), F * sizeof(Desc), hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_recd, recd.data(), F * sizeof(FrameRecD), hipMemcpyHostToDevice));
  const int G = prop.multiProcessorCount * waves_per_cu;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  auto timed = [&](auto&& launch) {
    double best = 1e30;
    for (int r = 0; r < 12; ++r) {  // the first passes ramp the clocks
      CHECK(hipEventRecord(e0, s));
      for (int it = 0; it < 4; ++it) launch();
      CHECK(hipEventRecord(e1, s));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 4) best = std::min(best, (double)ms * 1e3 / (4.0 * F));
    }
    return best;  // us per frame
  };
  const double us_list = timed([&] { hipLaunchKernelGGL(list_launch, dim3((unsigned)((max_n + 63) / 64), F), dim3(64), 0, s, (const Desc*)d_ring_b, (const FrameRecD*)d_recd); });
  const double us_plain = timed([&] { hipLaunchKernelGGL(frame_server<false>, dim3(G), dim3(64), 0, s, (const Desc*)d_ring_a, (uint32_t)F, tiles, (const FrameRecD*)d_recd); });
  // bitwise: the server's plain variant against the list launch
  std::vector<uint32_t> ha(4 * span), hb(4 * span);
  CHECK(hipMemcpy(ha.data(), out_a, span * 16, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(hb.data(), out_b, span * 16, hipMemcpyDeviceToHost));
  bool same = true;
  for (int f = 0; f < F && same; ++f) same = std::memcmp(ha.data() + 4 * start[f], hb.data() + 4 * start[f], sizes[f] * 16) == 0;
  const double us_pref = timed([&] { hipLaunchKernelGGL(frame_server<true>, dim3(G), dim3(64), 0, s, (const Desc*)d_ring_a, (uint32_t)F, tiles, (const FrameRecD*)d_recd); });
  CHECK(hipMemcpy(ha.data(), out_a, span * 16, hipMemcpyDeviceToHost));
  bool same_pref = true;
  for (int f = 0; f < F && same_pref; ++f) same_pref = std::memcmp(ha.data() + 4 * start[f], hb.data() + 4 * start[f], sizes[f] * 16) == 0;
  const double mean = (double)total / F;
  std::printf(
      "{\"frames\": %d, \"mean_points_per_frame\": %.1f, \"resident_waves\": %d, \"device\": \"%s\", "
      "\"dispatcher_fed_list_launch\": {\"us_per_frame\": %.3f, \"GBps\": %.1f}, "
      "\"resident_server_upper_bound_plain\": {\"us_per_frame\": %.3f, \"GBps\": %.1f, \"equals_list_launch_bitwise\": %s}, "
      "\"resident_server_upper_bound_prefetch\": {\"us_per_frame\": %.3f, \"GBps\": %.1f, \"equals_list_launch_bitwise\": %s}}\n",
      F, mean, G, prop.name, us_list, 32.0 * mean / us_list / 1e3, us_plain, 32.0 * mean / us_plain / 1e3, same ? "true" : "false", us_pref, 32.0 * mean / us_pref / 1e3,
      same_pref ? "true" : "false");
  return (same && same_pref) ? 0 : 1;
}

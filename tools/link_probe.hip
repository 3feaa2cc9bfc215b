// link_probe.hip -- where do the microseconds of an in-place call over the link go?  
// The literal kmc::MotionCompensateFrame(Frame, Time) runs ONE kernel on the caller's page-locked containers and waits for it:
// 132-153 us per 123 k-point frame where the bytes alone would take ~70 us on the duplex link.  This probe takes the call apart on
// a self-contained stand-in (same access pattern: persistent one-wave workgroups, next tile's load in flight while the current tile
// is stored; trivial arithmetic), all times on the HOST clock unless marked `dev_`:
//   floor      an empty kernel: launch + hipStreamSynchronize | launch + hipStreamQuery spin | launch + spin on a page-locked word the
//              kernel's last wave writes (system-scope store behind a device-scope ticket) -- what a completion costs three ways
//   f32 / f64  the KITTI-layout (16 B in, 16 B out per point) and Eigen-layout (32 B in, 24 B out: x y z stamps -> x y z) streams over
//              page-locked host memory with W persistent waves (and "one wave per tile"), completion by flag:
//              launch_us    the launch call itself
//              started_us   launch call -> the first wave's "I run" word is seen by the host
//              done_us      launch call -> the last wave's "all stored" word is seen by the host  (= the call, if it returned here)
//              sync_us      ... -> hipStreamSynchronize has returned as well (= today's call)
//              dev_span_us  first wave's start -> last wave's end on the device clock (s_memrealtime, 100 MHz)
//   read / write only: one direction alone at the same size (is the duplex kernel slower than its slower half?)
//   sdma       the same bytes through hipMemcpyAsync H2D -> device kernel -> D2H (pinned), whole and in chunks on three streams
//   link_probe [points=123397] [iterations=200]   -> one JSON object
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CHECK(x)                                                                              \
  do {                                                                                        \
    hipError_t e_ = (x);                                                                      \
    if (e_ != hipSuccess) {                                                                   \
      std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      std::exit(2);                                                                           \
    }                                                                                         \
  } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
using clk = std::chrono::steady_clock;
static double us_since(clk::time_point a) { return std::chrono::duration<double, std::micro>(clk::now() - a).count(); }

struct Sig {                 // page-locked, one cache line per word
  volatile uint32_t started; uint32_t pad0[15];
  volatile uint32_t done;    uint32_t pad1[15];
};

// completion: every wave releases its stores at system scope, takes a ticket; the last one raises the host word
__device__ __forceinline__ void wave_done(uint32_t* ticket, uint32_t n_waves, Sig* sig, uint32_t seq, uint64_t* dev_end) {
  if (threadIdx.x == 0) {
    __atomic_thread_fence(__ATOMIC_RELEASE);  // (hip: agent scope by default for __atomic_*; the system fence follows)
    __threadfence_system();
    const uint32_t t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (t == n_waves - 1) {
      *dev_end = __builtin_amdgcn_s_memrealtime();
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store((uint32_t*)&sig->done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
__device__ __forceinline__ void wave_started(Sig* sig, uint32_t seq, uint64_t* dev_start) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    *dev_start = __builtin_amdgcn_s_memrealtime();
    __hip_atomic_store((uint32_t*)&sig->started, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__global__ __launch_bounds__(64) void k_empty() {}
__global__ __launch_bounds__(64) void k_flag(uint32_t* ticket, Sig* sig, uint32_t seq, uint64_t* dev) {
  wave_started(sig, seq, dev);
  wave_done(ticket, gridDim.x, sig, seq, dev + 1);
}

// store policies of the duplex kernel (round 5: the nt stores of a kernel over page-locked memory stay in the L2 until the end-of-kernel /
// fence write-back -- the link then carries the reads first and the writes afterwards; a system-scope store goes out at once)
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
// `tile`: the wave's tile (wave-uniform base -> the descriptor lives in SGPRs), `bytes`: its extent, lane: 16-byte slot of the lane
template <int POLICY>
__device__ __forceinline__ void store_policy(v4f v, void* tile, uint32_t bytes, uint32_t lane) {
  if constexpr (POLICY == 0) { if (lane * 16 < bytes) __builtin_nontemporal_store(v, (v4f*)tile + lane); }
  else if constexpr (POLICY == 4) { if (lane * 16 < bytes) ((v4f*)tile)[lane] = v; }
  else {
    constexpr int aux = POLICY == 1 ? (1 | 16) : POLICY == 2 ? (1 | 16 | 2) : (16 | 2);  // 1 = sc0, 2 = nt, 16 = sc1
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(tile, 0, bytes, 0x00020000);  // lanes beyond `bytes` are clipped
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), r, lane * 16u, 0, aux);
  }
}
// MODE 0: read + write (the product's streamed shape), 1: read only, 2: write only
template <int MODE, int POLICY = 0>
__global__ __launch_bounds__(64) void k_f32(const v4f* __restrict__ in, v4f* __restrict__ out, uint64_t n, uint32_t* ticket, Sig* sig, uint32_t seq, uint64_t* dev, uint64_t* it_stamps) {
  wave_started(sig, seq, dev);
  uint32_t it = 0;
  const uint32_t tid = threadIdx.x;
  const uint64_t n_tiles = (n + 63) / 64, last = n - 1;
  uint64_t t = blockIdx.x;
  v4f acc = {0, 0, 0, 0};
  if (t < n_tiles) {
    v4f cur = {1, 2, 3, 4};
    if (MODE != 2) cur = __builtin_nontemporal_load(in + (t * 64 + tid <= last ? t * 64 + tid : last));
    while (true) {
      const uint64_t next = t + gridDim.x;
      const bool more = next < n_tiles;
      v4f nxt = cur;
      if (MODE != 2 && more) nxt = __builtin_nontemporal_load(in + (next * 64 + tid <= last ? next * 64 + tid : last));
      const uint64_t i = t * 64 + tid;
      v4f o = cur * 1.0001f;
      o.w = cur.w;
      if (MODE != 1) store_policy<POLICY>(o, out + t * 64, (uint32_t)((n - t * 64 < 64 ? n - t * 64 : 64) * 16), tid);
      else acc += o;
      if (it_stamps && blockIdx.x == 0 && tid == 0 && it < 62) it_stamps[2 + it++] = __builtin_amdgcn_s_memrealtime();  // wave 0: its tile `it` has been stored (issued)
      if (!more) break;
      cur = nxt;
      t = next;
    }
  }
  if (MODE == 1 && acc.x == 12345.678f) out[0] = acc;
  if (it_stamps && blockIdx.x == 0 && tid == 0) { it_stamps[0] = it; it_stamps[1] = dev[0]; }
  wave_done(ticket, gridDim.x, sig, seq, dev + 1);
}

// Eigen layout: x y z stamps in, x y z out; 128 points per wave turn, two consecutive points per lane
struct C64 { const double *x, *y, *z, *s; double *ox, *oy, *oz; };
template <int POLICY>
__global__ __launch_bounds__(64) void k_f64(C64 c, uint64_t n, uint32_t* ticket, Sig* sig, uint32_t seq, uint64_t* dev) {
  wave_started(sig, seq, dev);
  const uint32_t tid = threadIdx.x;
  const uint64_t n_full = n / 128;
  uint64_t t = blockIdx.x;
  auto ld = [&](const double* p, uint64_t i) { return __builtin_nontemporal_load(reinterpret_cast<const v2d*>(p + i)); };
  if (t < n_full) {
    uint64_t i = t * 128 + 2 * (uint64_t)tid;
    v2d x = ld(c.x, i), y = ld(c.y, i), z = ld(c.z, i), s = ld(c.s, i);
    while (true) {
      const uint64_t next = t + gridDim.x;
      const bool more = next < n_full;
      v2d nx = x, ny = y, nz = z, ns = s;
      if (more) { const uint64_t j = next * 128 + 2 * (uint64_t)tid; nx = ld(c.x, j); ny = ld(c.y, j); nz = ld(c.z, j); ns = ld(c.s, j); }
      store_policy<POLICY>(__builtin_bit_cast(v4f, x + s * 1e-9), c.ox + t * 128, 1024, tid);
      store_policy<POLICY>(__builtin_bit_cast(v4f, y + s * 1e-9), c.oy + t * 128, 1024, tid);
      store_policy<POLICY>(__builtin_bit_cast(v4f, z + s * 1e-9), c.oz + t * 128, 1024, tid);
      if (!more) break;
      x = nx; y = ny; z = nz; s = ns;
      t = next;
      i = t * 128 + 2 * (uint64_t)tid;
    }
  }
  // (the ragged tail is left out: a probe)
  wave_done(ticket, gridDim.x, sig, seq, dev + 1);
}
__global__ __launch_bounds__(64) void k_dev_f32(const v4f* __restrict__ in, v4f* __restrict__ out, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x;
  if (i < n) { v4f o = in[i] * 1.0001f; o.w = in[i].w; out[i] = o; }
}

struct Stat { std::vector<double> v; void add(double x) { v.push_back(x); } double med() { std::sort(v.begin(), v.end()); return v.empty() ? 0 : v[v.size() / 2]; } double best() { std::sort(v.begin(), v.end()); return v.empty() ? 0 : v[0]; } };

int main(int argc, char** argv) {
  const uint64_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 123397ull;
  const int iters = argc > 2 ? std::atoi(argv[2]) : 200;
  CHECK(hipSetDevice(0));
  hipStream_t s;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  Sig* sig;
  CHECK(hipHostMalloc((void**)&sig, sizeof(Sig), hipHostMallocPortable | hipHostMallocMapped));
  std::memset((void*)sig, 0, sizeof(Sig));
  uint32_t* ticket;
  CHECK(hipMalloc((void**)&ticket, 64));
  CHECK(hipMemset(ticket, 0, 64));
  uint64_t* dev;  // device stamps, page-locked so that the host reads them without a copy
  CHECK(hipHostMalloc((void**)&dev, 64, hipHostMallocPortable | hipHostMallocMapped));
  float *h_in, *h_out;
  CHECK(hipHostMalloc((void**)&h_in, n * 16 + 4096, hipHostMallocPortable | hipHostMallocMapped));
  CHECK(hipHostMalloc((void**)&h_out, n * 16 + 4096, hipHostMallocPortable | hipHostMallocMapped));
  for (uint64_t i = 0; i < 4 * n; ++i) h_in[i] = (float)(i % 977) * 0.25f;
  double* h64;  // 7 columns
  CHECK(hipHostMalloc((void**)&h64, 7 * n * 8 + 4096, hipHostMallocPortable | hipHostMallocMapped));
  for (uint64_t i = 0; i < 4 * n; ++i) h64[i] = (double)(i % 977) * 0.25;
  uint32_t seq = 0;
  std::printf("{\"points\": %llu, \"iterations\": %d", (unsigned long long)n, iters);

  // ---- floor ----
  {
    Stat a, b, c, c_sync;
    for (int i = 0; i < 50 + iters; ++i) {
      auto t0 = clk::now();
      hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s);
      CHECK(hipStreamSynchronize(s));
      if (i >= 50) a.add(us_since(t0));
    }
    for (int i = 0; i < 50 + iters; ++i) {
      auto t0 = clk::now();
      hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s);
      while (hipStreamQuery(s) == hipErrorNotReady) {}
      if (i >= 50) b.add(us_since(t0));
    }
    (void)hipGetLastError();
    for (int i = 0; i < 50 + iters; ++i) {
      ++seq;
      auto t0 = clk::now();
      hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, s, ticket, sig, seq, dev);
      while (sig->done != seq) {}
      const double d = us_since(t0);
      CHECK(hipStreamSynchronize(s));
      if (i >= 50) { c.add(d); c_sync.add(us_since(t0)); }
    }
    std::printf(", \"floor_us_median\": {\"launch_then_hipStreamSynchronize\": %.2f, \"launch_then_hipStreamQuery_spin\": %.2f, \"launch_then_spin_on_host_word\": %.2f, \"...and_then_hipStreamSynchronize\": %.2f}",
                a.med(), b.med(), c.med(), c_sync.med());
  }

  auto run = [&](const char* name, int waves, auto&& launch, double bytes_up, double bytes_down) {
    Stat l, st, d, sy, span;
    for (int i = 0; i < 30 + iters; ++i) {
      ++seq;
      auto t0 = clk::now();
      launch(waves, seq);
      const double tl = us_since(t0);
      double ts = -1;
      while (sig->done != seq) { if (ts < 0 && sig->started == seq) ts = us_since(t0); }
      const double td = us_since(t0);
      if (ts < 0) ts = td;
      CHECK(hipStreamSynchronize(s));
      const double tsy = us_since(t0);
      if (i >= 30) { l.add(tl); st.add(ts); d.add(td); sy.add(tsy); span.add((double)(dev[1] - dev[0]) * 0.01); }
    }
    const double dm = d.med();
    std::printf(", \"%s_w%d\": {\"launch_us\": %.2f, \"started_us\": %.2f, \"done_us\": %.2f, \"done_us_best\": %.2f, \"sync_us\": %.2f, \"dev_span_us\": %.2f, \"up_GBps\": %.1f, \"down_GBps\": %.1f}", name, waves,
                l.med(), st.med(), dm, d.best(), sy.med(), span.med(), bytes_up / dm * 1e-3, bytes_down / dm * 1e-3);
  };
  const int n_tiles = (int)((n + 63) / 64);
  const v4f* vin = (const v4f*)h_in;
  v4f* vout = (v4f*)h_out;
  for (int w : {64, 128, 256, 512, 1024, n_tiles}) {
    if (w > n_tiles) continue;
    run("f32_rw", w, [&](int W, uint32_t q) { hipLaunchKernelGGL((k_f32<0, 0>), dim3(W), dim3(64), 0, s, vin, vout, n, ticket, sig, q, dev, (uint64_t*)nullptr); }, 16.0 * n, 16.0 * n);
  }
  // store policies at W = 128 and 256: does the write go out while the reads are still coming in?
  for (int w : {64, 128, 256}) {
    run("f32_rw_store_sc0sc1", w, [&](int W, uint32_t q) { hipLaunchKernelGGL((k_f32<0, 1>), dim3(W), dim3(64), 0, s, vin, vout, n, ticket, sig, q, dev, (uint64_t*)nullptr); }, 16.0 * n, 16.0 * n);
    run("f32_rw_store_sc0sc1nt", w, [&](int W, uint32_t q) { hipLaunchKernelGGL((k_f32<0, 2>), dim3(W), dim3(64), 0, s, vin, vout, n, ticket, sig, q, dev, (uint64_t*)nullptr); }, 16.0 * n, 16.0 * n);
    run("f32_rw_store_sc1nt", w, [&](int W, uint32_t q) { hipLaunchKernelGGL((k_f32<0, 3>), dim3(W), dim3(64), 0, s, vin, vout, n, ticket, sig, q, dev, (uint64_t*)nullptr); }, 16.0 * n, 16.0 * n);
    run("f32_rw_store_plain", w, [&](int W, uint32_t q) { hipLaunchKernelGGL((k_f32<0, 4>), dim3(W), dim3(64), 0, s, vin, vout, n, ticket, sig, q, dev, (uint64_t*)nullptr); }, 16.0 * n, 16.0 * n);
  }
  for (int w : {256, n_tiles}) {
    if (w > n_tiles) continue;
    run("f32_read_only", w, [&](int W, uint32_t q) { hipLaunchKernelGGL((k_f32<1, 0>), dim3(W), dim3(64), 0, s, vin, vout, n, ticket, sig, q, dev, (uint64_t*)nullptr); }, 16.0 * n, 0);
    run("f32_write_only", w, [&](int W, uint32_t q) { hipLaunchKernelGGL((k_f32<2, 0>), dim3(W), dim3(64), 0, s, vin, vout, n, ticket, sig, q, dev, (uint64_t*)nullptr); }, 0, 16.0 * n);
  }
  {  // wave 0's own timeline at W = 128: when did each of its tiles go out (device clock, us after the wave started)?
    uint64_t* its;
    CHECK(hipHostMalloc((void**)&its, 64 * 8, hipHostMallocPortable | hipHostMallocMapped));
    std::memset(its, 0, 64 * 8);
    for (int rep = 0; rep < 20; ++rep) {
      ++seq;
      hipLaunchKernelGGL((k_f32<0, 0>), dim3(128), dim3(64), 0, s, vin, vout, n, ticket, sig, seq, dev, its);
      while (sig->done != seq) {}
      CHECK(hipStreamSynchronize(s));
    }
    std::printf(", \"f32_rw_w128_wave0_tile_stored_at_us\": [");
    for (uint64_t k = 0; k < its[0] && k < 62; ++k) std::printf("%s%.2f", k ? ", " : "", (double)(its[2 + k] - its[1]) * 0.01);
    std::printf("]");
    for (int rep = 0; rep < 20; ++rep) {
      ++seq;
      hipLaunchKernelGGL((k_f32<0, 1>), dim3(128), dim3(64), 0, s, vin, vout, n, ticket, sig, seq, dev, its);
      while (sig->done != seq) {}
      CHECK(hipStreamSynchronize(s));
    }
    std::printf(", \"f32_rw_store_sc0sc1_w128_wave0_tile_stored_at_us\": [");
    for (uint64_t k = 0; k < its[0] && k < 62; ++k) std::printf("%s%.2f", k ? ", " : "", (double)(its[2 + k] - its[1]) * 0.01);
    std::printf("]");
  }
  C64 c64 = {h64, h64 + n, h64 + 2 * n, h64 + 3 * n, h64 + 4 * n, h64 + 5 * n, h64 + 6 * n};
  const int n_t128 = (int)(n / 128);
  for (int w : {64, 128, 256, 512, n_t128}) {
    if (w > n_t128) continue;
    run("f64_rw", w, [&](int W, uint32_t q) { hipLaunchKernelGGL(k_f64<0>, dim3(W), dim3(64), 0, s, c64, n, ticket, sig, q, dev); }, 32.0 * n, 24.0 * n);
    run("f64_rw_store_sc0sc1", w, [&](int W, uint32_t q) { hipLaunchKernelGGL(k_f64<1>, dim3(W), dim3(64), 0, s, c64, n, ticket, sig, q, dev); }, 32.0 * n, 24.0 * n);
  }

  // ---- the same bytes through the copy engines (page-locked host buffers) ----
  {
    v4f *d_in, *d_out;
    CHECK(hipMalloc((void**)&d_in, n * 16));
    CHECK(hipMalloc((void**)&d_out, n * 16));
    Stat whole;
    for (int i = 0; i < 20 + iters; ++i) {
      auto t0 = clk::now();
      CHECK(hipMemcpyAsync(d_in, h_in, n * 16, hipMemcpyHostToDevice, s));
      hipLaunchKernelGGL(k_dev_f32, dim3((unsigned)n_tiles), dim3(64), 0, s, d_in, d_out, n);
      CHECK(hipMemcpyAsync(h_out, d_out, n * 16, hipMemcpyDeviceToHost, s));
      CHECK(hipStreamSynchronize(s));
      if (i >= 20) whole.add(us_since(t0));
    }
    std::printf(", \"sdma_f32\": {\"one_stream_h2d_kernel_d2h_us\": %.2f", whole.med());
    hipStream_t su, sk, sd;
    CHECK(hipStreamCreateWithFlags(&su, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&sd, hipStreamNonBlocking));
    hipEvent_t eu[8], ek[8];
    for (int k = 0; k < 8; ++k) { CHECK(hipEventCreateWithFlags(&eu[k], hipEventDisableTiming)); CHECK(hipEventCreateWithFlags(&ek[k], hipEventDisableTiming)); }
    for (int chunks : {2, 4}) {
      Stat cs;
      const uint64_t per = ((n + chunks - 1) / chunks + 63) & ~63ull;
      for (int i = 0; i < 20 + iters; ++i) {
        auto t0 = clk::now();
        for (int k = 0; k < chunks; ++k) {
          const uint64_t off = k * per, m = std::min<uint64_t>(per, n - off);
          CHECK(hipMemcpyAsync(d_in + off, h_in + 4 * off, m * 16, hipMemcpyHostToDevice, su));
          CHECK(hipEventRecord(eu[k], su));
          CHECK(hipStreamWaitEvent(sk, eu[k], 0));
          hipLaunchKernelGGL(k_dev_f32, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, sk, d_in + off, d_out + off, m);
          CHECK(hipEventRecord(ek[k], sk));
          CHECK(hipStreamWaitEvent(sd, ek[k], 0));
          CHECK(hipMemcpyAsync(h_out + 4 * off, d_out + off, m * 16, hipMemcpyDeviceToHost, sd));
        }
        CHECK(hipStreamSynchronize(sd));
        if (i >= 20) cs.add(us_since(t0));
      }
      std::printf(", \"three_streams_%d_chunks_us\": %.2f", chunks, cs.med());
    }
    std::printf("}");
  }
  std::printf("}\n");
  return 0;
}

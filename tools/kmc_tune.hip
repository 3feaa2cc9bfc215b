// kmc_tune.hip -- on-GPU A/B harness for the deskew kernels (run through gpurun; writes CSV to stdout).
//
// All variants are timed in ONE process, interleaved over several rounds (cdna_hip_programming.md rule 24),
// on a working set far beyond the 256 MiB Infinity Cache so that the GB/s are HBM GB/s.  The float4 copy kernel
// is the same-hardware ceiling every deskew variant is compared with (rule 10).
//
// usage: kmc_tune [n_points=67108864] [rounds=5] [iters=10]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../kitti_motion_compensation_amd/csrc/kmc_kernels.hip.h"

using namespace kmc_dev;

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      std::fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      std::exit(2);                                                               \
    }                                                                             \
  } while (0)

struct Variant {
  std::string name;
  std::function<void(hipStream_t, const v4f*, v4f*, uint64_t, int)> launch;  // (stream, in, out, n, blocks_per_cu)
  int ppt;
};

static int g_cus = 256;

static int grid_for(uint64_t n, int ppt, int bpc) {
  const uint64_t tiles = (n + (uint64_t)kBlock * ppt - 1) / ((uint64_t)kBlock * ppt);
  if (bpc <= 0) return (int)std::min<uint64_t>(tiles, 0x7fffffff);  // one tile per workgroup
  return (int)std::min<uint64_t>(tiles, (uint64_t)g_cus * bpc);
}

static FrameRec make_rec() {
  FrameRec f;
  std::memset(&f, 0, sizeof(f));
  // a turning trajectory: |phi| ~ 0.1 rad, |rho| ~ 1.3 m
  const double phi[3] = {0.02, 0.01, -0.1}, rho[3] = {1.3, 0.05, -0.02};
  const double c1[3] = {phi[1] * rho[2] - phi[2] * rho[1], phi[2] * rho[0] - phi[0] * rho[2], phi[0] * rho[1] - phi[1] * rho[0]};
  const double c2[3] = {phi[1] * c1[2] - phi[2] * c1[1], phi[2] * c1[0] - phi[0] * c1[2], phi[0] * c1[1] - phi[1] * c1[0]};
  f.phi_x = phi[0]; f.phi_y = phi[1]; f.phi_z = phi[2]; f.phi2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  f.rho_x = rho[0]; f.rho_y = rho[1]; f.rho_z = rho[2]; f.s0 = 0.0f;
  f.c1_x = c1[0]; f.c1_y = c1[1]; f.c1_z = c1[2];
  f.c2_x = c2[0]; f.c2_y = c2[1]; f.c2_z = c2[2];
  return f;
}

// the same twist in f64 (the near-origin guard's redo; never taken on the synthetic scans the tuner streams)
static FrameRecD make_recd() {
  FrameRecD d;
  std::memset(&d, 0, sizeof(d));
  const double phi[3] = {0.02, 0.01, -0.1}, rho[3] = {1.3, 0.05, -0.02};
  for (int k = 0; k < 3; ++k) { d.phi[k] = phi[k]; d.rho[k] = rho[k]; }
  d.c1[0] = phi[1] * rho[2] - phi[2] * rho[1]; d.c1[1] = phi[2] * rho[0] - phi[0] * rho[2]; d.c1[2] = phi[0] * rho[1] - phi[1] * rho[0];
  d.c2[0] = phi[1] * d.c1[2] - phi[2] * d.c1[1]; d.c2[1] = phi[2] * d.c1[0] - phi[0] * d.c1[2]; d.c2[2] = phi[0] * d.c1[1] - phi[1] * d.c1[0];
  d.phi2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  d.x_req = 0.5;
  return d;
}

// ---- experiments (tuner only) -------------------------------------------------------------------------------
// XCD-contiguous tile mapping: workgroup b runs on XCD b % 8 (observed, not contractual); give every XCD one contiguous
// eighth of the buffer instead of every eighth tile.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void deskew_frame_xcd(const v4f* __restrict__ in, v4f* __restrict__ out, uint64_t n, FrameRec f) {
  const uint64_t n_tiles = n / BLOCK;  // tuner sizes are multiples of the tile
  const uint64_t per_xcd = n_tiles / 8;
  const uint64_t b = blockIdx.x;
  const uint64_t t = (b % 8) * per_xcd + b / 8;
  const uint64_t i = t * BLOCK + threadIdx.x;
  if (b < per_xcd * 8) store_point<kNtBoth>(out + i, deskew_point<kSeries3, false>(load_point<kNtBoth>(in + i), f));
}

// explicit cache-policy bits through inline asm (loads: LP, stores: SP)
//   0: (none)   1: nt   2: sc1   3: sc0 sc1   4: sc1 nt   5: sc0 sc1 nt   6: sc0
template <int LP, int SP>
__global__ __launch_bounds__(64) void deskew_frame_policy(const v4f* __restrict__ in, v4f* __restrict__ out, uint64_t n, FrameRec f) {
  const uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  v4f p;
  const v4f* src = in + i;
  v4f* dst = out + i;
  if constexpr (LP == 0) asm volatile("global_load_dwordx4 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(p) : "v"(src) : "memory");
  if constexpr (LP == 1) asm volatile("global_load_dwordx4 %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(p) : "v"(src) : "memory");
  if constexpr (LP == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(p) : "v"(src) : "memory");
  if constexpr (LP == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(p) : "v"(src) : "memory");
  if constexpr (LP == 4) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt\n s_waitcnt vmcnt(0)" : "=v"(p) : "v"(src) : "memory");
  if constexpr (LP == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt\n s_waitcnt vmcnt(0)" : "=v"(p) : "v"(src) : "memory");
  if constexpr (LP == 6) asm volatile("global_load_dwordx4 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(p) : "v"(src) : "memory");
  const v4f q = deskew_point<kSeries3, false>(p, f);
  if constexpr (SP == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(dst), "v"(q) : "memory");
  if constexpr (SP == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(dst), "v"(q) : "memory");
  if constexpr (SP == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(q) : "memory");
  if constexpr (SP == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(q) : "memory");
  if constexpr (SP == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(dst), "v"(q) : "memory");
  if constexpr (SP == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(dst), "v"(q) : "memory");
  if constexpr (SP == 6) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(dst), "v"(q) : "memory");
}

// dynamic tile grabbing: persistent one-wave workgroups pull chunks of CHUNK consecutive tiles from per-XCD counters
// (counter x hands out global chunks x, x+8, x+16, ... so the set of tiles in flight stays one compact window, like the
// hardware dispatcher's order, but without re-launching a wave per tile).  The next grab is issued before the current
// chunk is processed, so its latency is hidden.
static unsigned long long* g_dyn_counters = nullptr;  // 8 x 128-byte-spaced counters

template <int CHUNK>
__global__ __launch_bounds__(64) void deskew_frame_dyn(const v4f* __restrict__ in, v4f* __restrict__ out, uint64_t n, FrameRec f,
                                                       unsigned long long* __restrict__ counters) {
  const uint32_t lane = threadIdx.x;
  const uint32_t xcd = blockIdx.x & 7u;
  unsigned long long* ctr = counters + xcd * 16;
  const uint64_t n_tiles = n / 64;  // tuner sizes are multiples of the tile
  const uint64_t n_chunks = n_tiles / CHUNK;
  unsigned long long g = 0;
  if (lane == 0) g = atomicAdd(ctr, 1ull);
  uint64_t chunk = (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)g) * 8 + xcd;
  while (chunk < n_chunks) {
    unsigned long long gn = 0;
    if (lane == 0) gn = atomicAdd(ctr, 1ull);  // prefetch the next grab
    const uint64_t first = chunk * CHUNK * 64;
    v4f p[CHUNK];
#pragma unroll
    for (int u = 0; u < CHUNK; ++u) p[u] = load_point<kNtBoth>(in + first + u * 64 + lane);
    const __amdgpu_buffer_rsrc_t rout = tile_rsrc(out + first, (uint64_t)CHUNK * 64 * sizeof(v4f));
#pragma unroll
    for (int u = 0; u < CHUNK; ++u) tile_store<kPolicyDefault>(rout, (uint32_t)((u * 64 + lane) * sizeof(v4f)), deskew_point<kSeries3, false>(p[u], f));
    chunk = (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)gn) * 8 + xcd;
  }
}

template <int CHUNK, int WAVES_PER_CU>
static Variant dyn_variant(const char* label) {
  Variant v;
  v.name = label;
  v.ppt = CHUNK;
  v.launch = [](hipStream_t s, const v4f* in, v4f* out, uint64_t n, int) {
    static const FrameRec f = make_rec();
    if (!g_dyn_counters) CK(hipMalloc((void**)&g_dyn_counters, 8 * 128));
    CK(hipMemsetAsync(g_dyn_counters, 0, 8 * 128, s));
    hipLaunchKernelGGL((deskew_frame_dyn<CHUNK>), dim3((unsigned)(g_cus * WAVES_PER_CU)), dim3(64), 0, s, in, out, n, f, g_dyn_counters);
  };
  return v;
}

// sequential multi-tile: one wave handles SEQ adjacent tiles one after the other (load, compute, store, then the next load),
// so the wave lives SEQ times longer per dispatch without holding more loads in flight
template <int SEQ>
__global__ __launch_bounds__(64) void deskew_frame_seq(const v4f* __restrict__ in, v4f* __restrict__ out, uint64_t n, FrameRec f) {
  const uint64_t first = (uint64_t)blockIdx.x * SEQ * 64;
  const __amdgpu_buffer_rsrc_t rout = tile_rsrc(out + first, (n - first) * sizeof(v4f));
#pragma unroll
  for (int u = 0; u < SEQ; ++u) {
    const uint64_t i = first + u * 64 + threadIdx.x;
    const v4f p = load_point<kNtBoth>(in + (i < n ? i : n - 1));
    tile_store<kPolicyDefault>(rout, (uint32_t)((u * 64 + threadIdx.x) * sizeof(v4f)), deskew_point<kSeries3, false>(p, f));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // keep the next load behind this store
  }
}

template <int SEQ>
static Variant seq_variant(const char* label) {
  Variant v;
  v.name = label;
  v.ppt = SEQ;
  v.launch = [](hipStream_t s, const v4f* in, v4f* out, uint64_t n, int) {
    static const FrameRec f = make_rec();
    const uint64_t per = 64ull * SEQ;
    hipLaunchKernelGGL((deskew_frame_seq<SEQ>), dim3((unsigned)((n + per - 1) / per)), dim3(64), 0, s, in, out, n, f);
  };
  return v;
}

// occupancy sensitivity: the same one-wave kernel with a dynamic-LDS reservation that caps the workgroups per CU
// PPT > 1: the same sweep with two / four tiles per wave, i.e. the bytes in flight per CU = workgroups per CU x PPT KiB -- is the
// optimum between "32 one-tile waves" (32 KiB) and "32 two-tile waves" (64 KiB)?
template <int LDS_BYTES, int PPT = 1>
static Variant occ_variant(const char* label) {
  Variant v;
  v.name = label;
  v.ppt = PPT;
  v.launch = [](hipStream_t s, const v4f* in, v4f* out, uint64_t n, int) {
    static const FrameRec f = make_rec();
    constexpr uint64_t kTile = 64 * PPT;
    hipLaunchKernelGGL((deskew_frame_f32<kSeries3, PPT, kPolicyDefault, false, 64>), dim3((unsigned)((n + kTile - 1) / kTile)), dim3(64), LDS_BYTES, s, in, out, n, f, 0u, make_recd());
  };
  return v;
}

template <int BLOCK>
static Variant xcd_variant(const char* label) {
  Variant v;
  v.name = label;
  v.ppt = 1;
  v.launch = [](hipStream_t s, const v4f* in, v4f* out, uint64_t n, int) {
    static const FrameRec f = make_rec();
    hipLaunchKernelGGL((deskew_frame_xcd<BLOCK>), dim3((unsigned)(n / BLOCK)), dim3(BLOCK), 0, s, in, out, n, f);
  };
  return v;
}

template <int LP, int SP>
static Variant policy_variant(const char* label) {
  Variant v;
  v.name = label;
  v.ppt = 1;
  v.launch = [](hipStream_t s, const v4f* in, v4f* out, uint64_t n, int) {
    static const FrameRec f = make_rec();
    hipLaunchKernelGGL((deskew_frame_policy<LP, SP>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, in, out, n, f);
  };
  return v;
}

// (the trajectory-kernel experiments of rounds 1-2 moved to kmc_tune_r3.hip together with round 2's LDS kernel)

template <int TIER, int PPT, int NT, bool OCML, int BLOCK = kBlock>
static Variant frame_variant(const char* label) {
  Variant v;
  v.name = label;
  v.ppt = PPT;
  v.launch = [](hipStream_t s, const v4f* in, v4f* out, uint64_t n, int bpc) {
    static const FrameRec f = make_rec();
    const uint64_t tiles = (n + (uint64_t)BLOCK * PPT - 1) / ((uint64_t)BLOCK * PPT);
    const uint64_t cap = bpc <= 0 ? tiles : (uint64_t)g_cus * bpc * (kBlock / (BLOCK < kBlock ? BLOCK : kBlock));
    hipLaunchKernelGGL((deskew_frame_f32<TIER, PPT, NT, OCML, BLOCK>), dim3((unsigned)std::min(tiles, cap)), dim3(BLOCK), 0, s, in, out, n, f, 0u, make_recd());
  };
  return v;
}

// one-directional ceilings, one wave per workgroup like the shipped kernels: 16 B per lane read (the sum keeps the loads alive; the
// store never happens) or written.  Throughput is reported against the table's 32 B/point, so double it for bytes actually moved.
__global__ __launch_bounds__(64) void read_only_points(const v4f* __restrict__ in, v4f* __restrict__ out, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  const v4f p = __builtin_nontemporal_load(in + i);
  if (p.x + p.y + p.z + p.w == 1.2345e30f) out[i] = p;
}
__global__ __launch_bounds__(64) void write_only_points(const v4f* __restrict__ in, v4f* __restrict__ out, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  const v4f p = {1.0f, 2.0f, 3.0f, (float)threadIdx.x};
  const __amdgpu_buffer_rsrc_t r = tile_rsrc(out + (uint64_t)blockIdx.x * 64, (n - (uint64_t)blockIdx.x * 64) * sizeof(v4f));
  tile_store<kPolicyDefault>(r, (uint32_t)(threadIdx.x * sizeof(v4f)), p);
}
// four loads (stores) in flight per lane: the one-wave, one-access kernels above are bounded by occupancy x latency, these by bandwidth
__global__ __launch_bounds__(64) void read_only_points4(const v4f* __restrict__ in, v4f* __restrict__ out, uint64_t n) {
  const uint64_t base = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (base + 192 >= n) return;
  const v4f a = __builtin_nontemporal_load(in + base), b = __builtin_nontemporal_load(in + base + 64);
  const v4f c = __builtin_nontemporal_load(in + base + 128), d = __builtin_nontemporal_load(in + base + 192);
  if (a.x + b.y + c.z + d.w == 1.2345e30f) out[base] = a;
}
__global__ __launch_bounds__(64) void write_only_points4(const v4f* __restrict__ in, v4f* __restrict__ out, uint64_t n) {
  const uint64_t base = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (base + 192 >= n) return;
  const v4f p = {1.0f, 2.0f, 3.0f, (float)threadIdx.x};
  __builtin_nontemporal_store(p, out + base);
  __builtin_nontemporal_store(p, out + base + 64);
  __builtin_nontemporal_store(p, out + base + 128);
  __builtin_nontemporal_store(p, out + base + 192);
}
static Variant oneway4_variant(const char* label, bool read) {
  Variant v;
  v.name = label;
  v.ppt = 1;
  if (read)
    v.launch = [](hipStream_t s, const v4f* in, v4f* out, uint64_t n, int) {
      hipLaunchKernelGGL(read_only_points4, dim3((unsigned)((n + 255) / 256)), dim3(64), 0, s, in, out, n);
    };
  else
    v.launch = [](hipStream_t s, const v4f* in, v4f* out, uint64_t n, int) {
      hipLaunchKernelGGL(write_only_points4, dim3((unsigned)((n + 255) / 256)), dim3(64), 0, s, in, out, n);
    };
  return v;
}
static Variant oneway_variant(const char* label, bool read) {
  Variant v;
  v.name = label;
  v.ppt = 1;
  if (read)
    v.launch = [](hipStream_t s, const v4f* in, v4f* out, uint64_t n, int) {
      hipLaunchKernelGGL(read_only_points, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, in, out, n);
    };
  else
    v.launch = [](hipStream_t s, const v4f* in, v4f* out, uint64_t n, int) {
      hipLaunchKernelGGL(write_only_points, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, in, out, n);
    };
  return v;
}

template <int PPT, int NT>
static Variant copy_variant(const char* label) {
  Variant v;
  v.name = label;
  v.ppt = PPT;
  v.launch = [](hipStream_t s, const v4f* in, v4f* out, uint64_t n, int bpc) {
    hipLaunchKernelGGL((copy_points<PPT, NT>), dim3(grid_for(n, PPT, bpc)), dim3(kBlock), 0, s, in, out, n);
  };
  return v;
}

// batched kernel: tables for `frames` equal frames over n points
struct BatchTables {
  BatchRec* d_recs = nullptr;
  FrameRecD* d_recd = nullptr;
  uint2* d_coarse = nullptr;
  float* d_pre2 = nullptr;
  uint32_t n_frames = 0;
};
static BatchTables g_bt_big, g_bt_small;

static void build_tables(BatchTables* bt, uint64_t n, uint64_t pts_per_frame) {
  const uint32_t nf = (uint32_t)((n + pts_per_frame - 1) / pts_per_frame);
  std::vector<BatchRec> recs(nf);
  const FrameRec f = make_rec();
  for (uint32_t i = 0; i < nf; ++i) {
    BatchRec& r = recs[i];
    r.phi_x = f.phi_x; r.phi_y = f.phi_y; r.phi_z = f.phi_z; r.phi2 = f.phi2;
    r.rho_x = f.rho_x; r.rho_y = f.rho_y; r.rho_z = f.rho_z; r.s0 = f.s0;
    r.c1_x = f.c1_x; r.c1_y = f.c1_y; r.c1_z = f.c1_z;
    r.c2_x = f.c2_x; r.c2_y = f.c2_y; r.c2_z = f.c2_z;
    const uint64_t end = std::min<uint64_t>(n, (uint64_t)(i + 1) * pts_per_frame);
    r.end_lo = (uint32_t)end; r.end_hi = (uint32_t)(end >> 32);
  }
  CK(hipMalloc((void**)&bt->d_recs, nf * sizeof(BatchRec)));
  CK(hipMemcpy(bt->d_recs, recs.data(), nf * sizeof(BatchRec), hipMemcpyHostToDevice));
  {
    std::vector<FrameRecD> recd(nf, make_recd());
    CK(hipMalloc((void**)&bt->d_recd, nf * sizeof(FrameRecD)));
    CK(hipMemcpy(bt->d_recd, recd.data(), nf * sizeof(FrameRecD), hipMemcpyHostToDevice));
    std::vector<float> pre2(nf, kGuardPre * (f.rho_x * f.rho_x + f.rho_y * f.rho_y + f.rho_z * f.rho_z));
    CK(hipMalloc((void**)&bt->d_pre2, nf * sizeof(float)));
    CK(hipMemcpy(bt->d_pre2, pre2.data(), nf * sizeof(float), hipMemcpyHostToDevice));
  }
  {
    const uint64_t chunk = 1ull << kChunkShift, nc = (n + chunk - 1) / chunk;
    std::vector<uint2> co(nc + 1);
    for (uint64_t c = 0; c < nc; ++c) {
      const uint64_t first = c * chunk, chunk_end = std::min<uint64_t>(first + chunk, n);
      const uint32_t f = (uint32_t)(first / pts_per_frame);
      const uint64_t e = std::min<uint64_t>(n, (uint64_t)(f + 1) * pts_per_frame);
      uint32_t split = kSplitNone;
      if (e < chunk_end) split = (std::min<uint64_t>(n, (uint64_t)(f + 2) * pts_per_frame) < chunk_end) ? kSplitSearch : (uint32_t)(e - first);
      co[c] = make_uint2(f, split);
    }
    co[nc] = make_uint2((uint32_t)((n - 1) / pts_per_frame), kSplitNone);
    CK(hipMalloc((void**)&bt->d_coarse, (nc + 1) * sizeof(uint2)));
    CK(hipMemcpy(bt->d_coarse, co.data(), (nc + 1) * sizeof(uint2), hipMemcpyHostToDevice));
  }
  bt->n_frames = nf;
}

template <int PPT, bool SMALL, int BLOCK = kBlock, int NT = kNtBoth>
static Variant batch_variant(const char* label) {
  Variant v;
  v.name = label;
  v.ppt = PPT;
  v.launch = [](hipStream_t s, const v4f* in, v4f* out, uint64_t n, int bpc) {
    const BatchTables& bt = SMALL ? g_bt_small : g_bt_big;
    const uint64_t tiles = (n + (uint64_t)BLOCK * PPT - 1) / ((uint64_t)BLOCK * PPT);
    const uint64_t cap = bpc <= 0 ? tiles : (uint64_t)g_cus * bpc * (kBlock / (BLOCK < kBlock ? BLOCK : kBlock));
    hipLaunchKernelGGL((deskew_batch_f32<kSeries3, PPT, NT, false, BLOCK>), dim3((unsigned)std::min(tiles, cap)), dim3(BLOCK), 0, s,
                       in, out, bt.d_recs, bt.d_coarse, bt.n_frames, n, (uint32_t*)nullptr, 0u, (const FrameRecD*)bt.d_recd, (uint32_t)kChunkShift, bt.d_pre2, BatchNoInline{});
  };
  return v;
}

int main(int argc, char** argv) {
  const uint64_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : (1ull << 26);
  const int rounds = argc > 2 ? std::atoi(argv[2]) : 5;
  const int iters = argc > 3 ? std::atoi(argv[3]) : 10;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  g_cus = prop.multiProcessorCount;
  std::fprintf(stderr, "# device %s (%s), %d CUs, n=%llu points (%.1f MiB in + same out), rounds=%d iters=%d\n", prop.name,
               prop.gcnArchName, g_cus, (unsigned long long)n, n * 16.0 / (1 << 20), rounds, iters);

  // rotating buffers: 3 input/output pairs so that consecutive launches never hit MALL-resident lines
  constexpr int kBufs = 3;
  v4f* in[kBufs];
  v4f* out[kBufs];
  hipStream_t s;
  CK(hipStreamCreate(&s));
  for (int b = 0; b < kBufs; ++b) {
    CK(hipMalloc((void**)&in[b], n * sizeof(v4f)));
    CK(hipMalloc((void**)&out[b], n * sizeof(v4f)));
    hipLaunchKernelGGL(synth_points<0>, dim3(g_cus * 8), dim3(kBlock), 0, s, in[b], n, 0x4B4D43ull + b);
    CK(hipMemsetAsync(out[b], 0, n * sizeof(v4f), s));
  }
  CK(hipStreamSynchronize(s));

  build_tables(&g_bt_big, n, 1000000);
  build_tables(&g_bt_small, n, 123397);
  struct Entry { Variant v; std::vector<int> bpcs; };
  std::vector<Entry> es;
  const std::vector<int> kAll = {0, 8, 32}, kZero = {0};
  es.push_back({copy_variant<1, kNtBoth>("copy_ppt1_nt"), kAll});
  es.push_back({copy_variant<1, kNtLoad>("copy_ppt1_ntload"), kZero});
  es.push_back({copy_variant<1, kNtStore>("copy_ppt1_ntstore"), kZero});
  es.push_back({copy_variant<1, 0>("copy_ppt1_plain"), kZero});
  es.push_back({copy_variant<4, kNtBoth>("copy_ppt4_nt"), kAll});
  es.push_back({oneway_variant("x_read_only_16B_per_point", true), kZero});
  es.push_back({oneway_variant("x_write_only_16B_per_point", false), kZero});
  es.push_back({oneway4_variant("x_read_only_4_loads_per_lane", true), kZero});
  es.push_back({oneway4_variant("x_write_only_4_stores_per_lane", false), kZero});
  es.push_back({frame_variant<kSeries3, 1, kNtBoth, false>("s3_ppt1_nt"), kAll});
  es.push_back({frame_variant<kSeries3, 1, kNtLoad, false>("s3_ppt1_ntload"), kZero});
  es.push_back({frame_variant<kSeries3, 1, kNtStore, false>("s3_ppt1_ntstore"), kZero});
  es.push_back({frame_variant<kSeries3, 1, 0, false>("s3_ppt1_plain"), kZero});
  es.push_back({frame_variant<kSeries3, 1, kNtBoth, false, 64>("s3_ppt1_nt_b64"), kZero});
  es.push_back({frame_variant<kSeries3, 1, kNtBoth, false, 128>("s3_ppt1_nt_b128"), kZero});
  es.push_back({frame_variant<kSeries3, 1, kNtBoth, false, 512>("s3_ppt1_nt_b512"), kZero});
  es.push_back({frame_variant<kSeries3, 1, kNtBoth, false, 1024>("s3_ppt1_nt_b1024"), kZero});
  es.push_back({frame_variant<kSeries3, 1, kNtBoth | kStoreSc1, false, 64>("s3_b64_gld_bufst_sc1nt"), kZero});
  es.push_back({frame_variant<kSeries3, 1, kNtBoth | kStoreSc1 | kBufLoad, false, 64>("s3_b64_bufld_bufst_sc1nt"), kZero});
  es.push_back({frame_variant<kSeries3, 1, kNtBoth | kBufLoad, false, 64>("s3_b64_bufld_bufst_nt"), kZero});
  es.push_back({frame_variant<kSeries3, 1, kNtBoth | kStoreSc1, false, 256>("s3_b256_gld_bufst_sc1nt"), kZero});
  es.push_back({frame_variant<kSeries3, 2, kNtBoth | kStoreSc1 | kBufLoad, false, 64>("s3_b64_ppt2_buf_sc1nt"), kZero});
  es.push_back({dyn_variant<1, 32>("dyn_c1_w32"), kZero});
  es.push_back({dyn_variant<2, 32>("dyn_c2_w32"), kZero});
  es.push_back({dyn_variant<4, 32>("dyn_c4_w32"), kZero});
  es.push_back({dyn_variant<8, 32>("dyn_c8_w32"), kZero});
  es.push_back({dyn_variant<2, 64>("dyn_c2_w64"), kZero});
  es.push_back({dyn_variant<4, 16>("dyn_c4_w16"), kZero});
  es.push_back({seq_variant<1>("seq1"), kZero});
  es.push_back({seq_variant<2>("seq2"), kZero});
  es.push_back({seq_variant<4>("seq4"), kZero});
  es.push_back({occ_variant<0>("occ_32_per_cu"), kZero});
  es.push_back({occ_variant<5800>("occ_28_per_cu"), kZero});
  es.push_back({occ_variant<6800>("occ_24_per_cu"), kZero});
  es.push_back({occ_variant<8100>("occ_20_per_cu"), kZero});
  es.push_back({occ_variant<10200>("occ_16_per_cu"), kZero});
  es.push_back({occ_variant<13600>("occ_12_per_cu"), kZero});
  es.push_back({occ_variant<20400>("occ_8_per_cu"), kZero});
  es.push_back({occ_variant<0, 2>("occ_32_per_cu_ppt2"), kZero});      // 64 KiB of loads in flight per CU
  es.push_back({occ_variant<5800, 2>("occ_28_per_cu_ppt2"), kZero});   // 56
  es.push_back({occ_variant<6800, 2>("occ_24_per_cu_ppt2"), kZero});   // 48
  es.push_back({occ_variant<8100, 2>("occ_20_per_cu_ppt2"), kZero});   // 40
  es.push_back({occ_variant<10200, 2>("occ_16_per_cu_ppt2"), kZero});  // 32
  es.push_back({occ_variant<13600, 2>("occ_12_per_cu_ppt2"), kZero});  // 24
  es.push_back({occ_variant<10200, 4>("occ_16_per_cu_ppt4"), kZero});  // 64
  es.push_back({occ_variant<13600, 4>("occ_12_per_cu_ppt4"), kZero});  // 48
  es.push_back({occ_variant<20400, 4>("occ_8_per_cu_ppt4"), kZero});   // 32
  es.push_back({xcd_variant<64>("x_xcdcontig_b64"), kZero});
  es.push_back({xcd_variant<256>("x_xcdcontig_b256"), kZero});
  es.push_back({policy_variant<1, 1>("x_pol_nt_nt"), kZero});
  es.push_back({policy_variant<0, 0>("x_pol_plain_plain"), kZero});
  es.push_back({policy_variant<1, 0>("x_pol_nt_plain"), kZero});
  es.push_back({policy_variant<1, 2>("x_pol_nt_sc1"), kZero});
  es.push_back({policy_variant<1, 4>("x_pol_nt_sc1nt"), kZero});
  es.push_back({policy_variant<1, 3>("x_pol_nt_sc0sc1"), kZero});
  es.push_back({policy_variant<1, 5>("x_pol_nt_sc0sc1nt"), kZero});
  es.push_back({policy_variant<1, 6>("x_pol_nt_sc0"), kZero});
  es.push_back({policy_variant<4, 1>("x_pol_sc1nt_nt"), kZero});
  es.push_back({policy_variant<2, 1>("x_pol_sc1_nt"), kZero});
  es.push_back({policy_variant<5, 1>("x_pol_sc0sc1nt_nt"), kZero});
  es.push_back({policy_variant<6, 1>("x_pol_sc0_nt"), kZero});
  es.push_back({policy_variant<5, 5>("x_pol_sc0sc1nt_both"), kZero});
  es.push_back({frame_variant<kSeries3, 2, kNtBoth, false>("s3_ppt2_nt"), kZero});
  es.push_back({frame_variant<kSeries3, 2, kNtBoth, false, 128>("s3_ppt2_nt_b128"), kZero});
  es.push_back({frame_variant<kSeries3, 4, kNtBoth, false>("s3_ppt4_nt"), kAll});
  es.push_back({frame_variant<kSeries3, 1, kNtBoth, true>("s3_ppt1_nt_ocml"), kZero});
  es.push_back({frame_variant<kSeries5, 1, kNtBoth, false>("s5_ppt1_nt"), kZero});
  es.push_back({frame_variant<kWide, 1, kNtBoth, false>("wide_ppt1_nt"), kZero});
  es.push_back({frame_variant<kTrig, 1, kNtBoth, false>("trig_ppt1_nt"), kZero});
  es.push_back({batch_variant<1, false>("batch1M_ppt1"), kAll});
  es.push_back({batch_variant<1, false, 64>("batch1M_ppt1_b64"), kZero});
  es.push_back({batch_variant<1, false, 64, kPolicyDefault>("batch1M_ppt1_b64_sc1nt"), kZero});
  es.push_back({batch_variant<1, false, 128>("batch1M_ppt1_b128"), kZero});
  es.push_back({batch_variant<2, false, 64>("batch1M_ppt2_b64"), kZero});
  es.push_back({batch_variant<2, false>("batch1M_ppt2"), kZero});
  es.push_back({batch_variant<4, false>("batch1M_ppt4"), kZero});
  es.push_back({batch_variant<1, true>("batch123k_ppt1"), kZero});
  es.push_back({batch_variant<1, true, 64>("batch123k_ppt1_b64"), kZero});
  es.push_back({batch_variant<1, true, 128>("batch123k_ppt1_b128"), kZero});

  std::vector<Variant> vs;  // flattened (variant, bpc) pairs
  std::vector<int> vb;
  for (auto& e : es)
    for (int b : e.bpcs) { vs.push_back(e.v); vb.push_back(b); }

  struct Res { std::string name; int bpc; std::vector<float> ms; };
  std::vector<Res> res;
  for (size_t i = 0; i < vs.size(); ++i) res.push_back({vs[i].name, vb[i], {}});

  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  int rot = 0;
  for (int r = 0; r < rounds + 1; ++r) {  // round 0 = warm-up
    for (size_t k = 0; k < vs.size(); ++k) {
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < iters; ++i) {
        vs[k].launch(s, in[rot % kBufs], out[rot % kBufs], n, vb[k]);
        ++rot;
      }
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0) res[k].ms.push_back(ms / iters);
    }
  }
  std::printf("variant,blocks_per_cu,ms_median,ms_min,gbps_median,gbps_best,mpts_median,frac_of_8TBps\n");
  for (auto& x : res) {
    std::sort(x.ms.begin(), x.ms.end());
    const float med = x.ms[x.ms.size() / 2], mn = x.ms.front();
    const double bytes = 32.0 * n;
    std::printf("%s,%d,%.4f,%.4f,%.1f,%.1f,%.1f,%.4f\n", x.name.c_str(), x.bpc, med, mn, bytes / med * 1e-6, bytes / mn * 1e-6,
                n / med * 1e-3, bytes / med * 1e-6 / 8000.0);
  }
  return 0;
}

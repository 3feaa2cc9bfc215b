// f64_layout_probe.hip -- A/B of the f64 Eigen-layout kernel's memory access (round 4): the shipped kernel streams nine columns per wave
// (ts, x, y, z, w in; x, y, z, w out), 1 KiB per column per wave, plain nt loads / nt stores, and sits at 74-78 % of the HBM peak where
// the f32 kernels reach 86 %.  Variants, same arithmetic (deskew_one_f64 of kmc_kernels.hip.h), outputs compared bit for bit:
//   shipped        deskew_f64cols<false> as the library launches it
//   sc1_stores     stores through buffer descriptors with nt + sc1 (what gave the f32 kernels +1.4 %)
//   wide256        256 points per wave: two 16-byte accesses per lane and column, 2 KiB of consecutive bytes per column per wave
//   wide256_sc1    both
//   no_w           the shipped kernel without the homogeneous column (w == nullptr, ow == nullptr: what the C++ drop-in passes): 56 B/point
//   f64_layout_probe [n_points=67108864] [rounds=5] [iters=10]      -> CSV
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

using v4u32 = uint32_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void col_store_sc1(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, v2d_u v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32, v), r, byte_off, 0, 2 | 16);
}

template <int PAIRS, bool SC1>  // PAIRS: 16-byte accesses per lane and column (1 = 128 points per wave, 2 = 256)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 4))) void f64_variant(const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ z,
                                                                                       const double* __restrict__ w, const double* __restrict__ stamps, uint64_t n, FrameRec64 f,
                                                                                       double* __restrict__ ox, double* __restrict__ oy, double* __restrict__ oz,
                                                                                       double* __restrict__ ow) {
  constexpr uint64_t kPts = 128 * PAIRS;
  const uint32_t tid = threadIdx.x;
  const uint64_t base = (uint64_t)blockIdx.x * kPts;
  if (base + kPts > n) return;  // (the probe runs on a multiple of 256 points)
  F64Tile t[PAIRS];
#pragma unroll
  for (int p = 0; p < PAIRS; ++p) t[p] = f64_tile_load(x, y, z, w, stamps, base + 128 * p + 2 * (uint64_t)tid);
#pragma unroll
  for (int p = 0; p < PAIRS; ++p) {
    const uint64_t i = base + 128 * p + 2 * (uint64_t)tid;
    if constexpr (!SC1) {
      (void)f64_tile_finish(t[p], f, ox, oy, oz, ow, i);
    } else {
      v2d_u rx, ry, rz;
      bool ok;
      double a, b, c;
      deskew_one_f64(t[p].vx.x, t[p].vy.x, t[p].vz.x, t[p].vw.x, t[p].ts.x, f, a, b, c, ok);
      rx.x = a; ry.x = b; rz.x = c;
      deskew_one_f64(t[p].vx.y, t[p].vy.y, t[p].vz.y, t[p].vw.y, t[p].ts.y, f, a, b, c, ok);
      rx.y = a; ry.y = b; rz.y = c;
      const uint64_t cb = base + 128 * p;  // first point of this 1 KiB piece of every column
      const uint32_t off = tid * 16u;
      col_store_sc1(tile_rsrc(ox + cb, 1024), off, rx);
      col_store_sc1(tile_rsrc(oy + cb, 1024), off, ry);
      col_store_sc1(tile_rsrc(oz + cb, 1024), off, rz);
      if (ow) col_store_sc1(tile_rsrc(ow + cb, 1024), off, t[p].vw);
    }
  }
}

int main(int argc, char** argv) {
  uint64_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 67108864ull;
  n &= ~255ull;
  const int rounds = argc > 2 ? std::atoi(argv[2]) : 5, iters = argc > 3 ? std::atoi(argv[3]) : 10;
  CHECK(hipSetDevice(0));
  hipStream_t s;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  double *in[5], *out_ref[4], *out[4];
  for (auto& p : in) CHECK(hipMalloc(&p, n * 8));
  for (auto& p : out_ref) CHECK(hipMalloc(&p, n * 8));
  for (auto& p : out) CHECK(hipMalloc(&p, n * 8));
  {
    std::vector<double> h(n);
    for (int c = 0; c < 5; ++c) {
      for (uint64_t i = 0; i < n; ++i)
        h[i] = c == 4 ? 100.0 + 0.1 * (double)((i * 2654435761ull) % 1000003ull) / 1000003.0 : (c == 3 ? 1.0 : -40.0 + 80.0 * (double)((i * 40503ull + c * 977ull) % 999983ull) / 999983.0);
      CHECK(hipMemcpy(in[c], h.data(), n * 8, hipMemcpyHostToDevice));
    }
  }
  FrameRec64 f;
  std::memset(&f, 0, sizeof(f));
  f.phi[0] = 0.002; f.phi[1] = -0.004; f.phi[2] = 0.03; f.rho[0] = 1.3; f.rho[1] = 0.05; f.rho[2] = -0.02;
  f.c1[0] = f.phi[1] * f.rho[2] - f.phi[2] * f.rho[1]; f.c1[1] = f.phi[2] * f.rho[0] - f.phi[0] * f.rho[2]; f.c1[2] = f.phi[0] * f.rho[1] - f.phi[1] * f.rho[0];
  f.c2[0] = f.phi[1] * f.c1[2] - f.phi[2] * f.c1[1]; f.c2[1] = f.phi[2] * f.c1[0] - f.phi[0] * f.c1[2]; f.c2[2] = f.phi[0] * f.c1[1] - f.phi[1] * f.c1[0];
  f.phi2 = f.phi[0] * f.phi[0] + f.phi[1] * f.phi[1] + f.phi[2] * f.phi[2];
  f.x_req = 0.5; f.t_start = 100.0; f.t_end = 100.1; f.dur = f.t_end - f.t_start; f.halvings = 0;
  unsigned long long* d_bad = nullptr;
  CHECK(hipMalloc(&d_bad, 8));
  CHECK(hipMemset(d_bad, 0, 8));
  // in[]: 0 x, 1 y, 2 z, 3 w, 4 stamps
  struct V { const char* name; int kind; double bytes; };
  const V vs[] = {{"shipped", 0, 72}, {"sc1_stores", 1, 72}, {"wide256", 2, 72}, {"wide256_sc1", 3, 72}, {"no_w", 4, 56}};
  auto launch = [&](int kind, double** o) {
    switch (kind) {
      case 0: hipLaunchKernelGGL(deskew_f64cols<false>, dim3((unsigned)(n / 128)), dim3(64), 0, s, in[0], in[1], in[2], in[3], in[4], n, f, o[0], o[1], o[2], o[3], d_bad, (uint32_t*)nullptr, (uint64_t)0); break;
      case 1: hipLaunchKernelGGL((f64_variant<1, true>), dim3((unsigned)(n / 128)), dim3(64), 0, s, in[0], in[1], in[2], in[3], in[4], n, f, o[0], o[1], o[2], o[3]); break;
      case 2: hipLaunchKernelGGL((f64_variant<2, false>), dim3((unsigned)(n / 256)), dim3(64), 0, s, in[0], in[1], in[2], in[3], in[4], n, f, o[0], o[1], o[2], o[3]); break;
      case 3: hipLaunchKernelGGL((f64_variant<2, true>), dim3((unsigned)(n / 256)), dim3(64), 0, s, in[0], in[1], in[2], in[3], in[4], n, f, o[0], o[1], o[2], o[3]); break;
      default: hipLaunchKernelGGL(deskew_f64cols<false>, dim3((unsigned)(n / 128)), dim3(64), 0, s, in[0], in[1], in[2], (const double*)nullptr, in[4], n, f, o[0], o[1], o[2], (double*)nullptr, d_bad, (uint32_t*)nullptr, (uint64_t)0); break;
    }
  };
  launch(0, out_ref);
  CHECK(hipStreamSynchronize(s));
  std::vector<double> ha(n), hb(n);
  bool same[5] = {true, true, true, true, true};
  for (int k = 1; k < 5; ++k) {
    for (auto& p : out) CHECK(hipMemset(p, 0, n * 8));
    launch(vs[k].kind, out);
    CHECK(hipStreamSynchronize(s));
    for (int c = 0; c < (k == 4 ? 3 : 4); ++c) {
      CHECK(hipMemcpy(ha.data(), out_ref[c], n * 8, hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(hb.data(), out[c], n * 8, hipMemcpyDeviceToHost));
      same[k] = same[k] && std::memcmp(ha.data(), hb.data(), n * 8) == 0;
    }
  }
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  std::vector<std::vector<double>> us(5);
  for (int w = 0; w < 20; ++w) launch(0, out);  // clocks
  for (int r = 0; r < rounds; ++r)
    for (int k = 0; k < 5; ++k) {
      launch(vs[k].kind, out);
      CHECK(hipEventRecord(e0, s));
      for (int it = 0; it < iters; ++it) launch(vs[k].kind, out);
      CHECK(hipEventRecord(e1, s));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      us[k].push_back(ms * 1e3 / iters);
    }
  std::printf("variant,points,best_us,median_us,GBps_median,frac_of_8TBps,bitwise_equal_to_shipped\n");
  for (int k = 0; k < 5; ++k) {
    std::sort(us[k].begin(), us[k].end());
    const double med = us[k][us[k].size() / 2];
    std::printf("%s,%llu,%.1f,%.1f,%.1f,%.3f,%s\n", vs[k].name, (unsigned long long)n, us[k].front(), med, vs[k].bytes * n / med / 1e3, vs[k].bytes * n / med / 1e3 / 8000.0, same[k] ? "true" : "false");
  }
  return 0;
}

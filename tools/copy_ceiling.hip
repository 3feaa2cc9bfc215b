// copy_ceiling.hip -- the memory system's own ceilings next to the deskew kernels, and the calibration kernel of the PMC traffic figures.
//   copy_ceiling [n_points=67108864] [rounds=5] [iters=10] [out_shift_bytes=0] [cols]   (CSV on stdout: kernel, best / median us per launch, GB/s;
//                 out_shift_bytes moves every output buffer relative to its input: does the memory care how the two streams line up?
//                 `cols`: only the column-stream rows -- bench.py runs this right before and right after the f64 kernel's bursts)
//   copy_points   one v4f per lane in, one out, 256-thread workgroups, nt loads + nt stores: 16 B read + 16 B written per point.  Its
//                 FETCH_SIZE / WRITE_SIZE counts against its KNOWN traffic give the correction factors tools/summarize_profiles.py applies
//                 to the bench kernel's counters (gfx950: FETCH_SIZE counts half of a wide coalesced stream; MI355X_MICROARCH.md, HBM section)
//   copy_tiles    the product kernels' access pattern without their arithmetic: one 64-point tile per one-wave workgroup, nt load,
//                 nt + sc1 store through a buffer descriptor
//   read_points / write_points   one direction alone (7.06 / 6.72 TB/s on the round-3 boxes: the deskew kernels sit on their mean)
//   copy_cols9 / copy_cols7   (round 5) the f64 Eigen-layout kernel's access pattern without its arithmetic: five (four) `double[n]`
//                 columns read, four (three) written, two consecutive points = 16 B per lane and column, one 128-point tile per one-wave
//                 workgroup, nt -- 72 (56) B per point.  copy_cols9_w4: the same with resident waves capped at 4 per SIMD through dynamic
//                 LDS (what deskew_f64cols runs at); copy_cols9_sc1: stores through descriptors with nt + sc1.  `points` of these rows =
//                 n_points of the command line (every column holds n doubles).
// Rounds 1-3's A/B harnesses (kmc_tune.hip, kmc_tune_r3.hip: persistent grids, 2 / 4 / 8 points per lane, cache policies, XCD mappings,
// ocml trigonometry, the LDS-staged N-knot kernel) instantiated kernel variants that left the product headers in round 4; their sources
// are in the repository's history (last present at commit 3470e8d), their tables under profiles/r0[123]_tune*.csv.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>
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
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void copy_points(const v4f* __restrict__ in, v4f* __restrict__ out, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
}
__global__ __launch_bounds__(64) void copy_tiles(const v4f* __restrict__ in, v4f* __restrict__ out, uint64_t n) {
  const uint64_t base = (uint64_t)blockIdx.x * 64, i = base + threadIdx.x;
  const v4f p = __builtin_nontemporal_load(in + (i < n ? i : n - 1));
  const uint64_t bytes = (n - base) * 16;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(out + base), 0, bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)bytes, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, p), r, threadIdx.x * 16u, 0, 2 | 16);
}
// one direction alone: four points per lane, 256 points per wave (a wave that issues ONE load and retires is bound by the wave launch
// rate -- ~4.7 waves per ns on MI355X -- not by HBM: 4.8 TB/s where four loads per wave reach the memory's own ceiling)
__global__ __launch_bounds__(64) void read_points(const v4f* __restrict__ in, v4f* __restrict__ out, uint64_t n) {
  const uint64_t base = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  v4f acc = {0, 0, 0, 0};
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const uint64_t i = base + 64 * u;
    acc += __builtin_nontemporal_load(in + (i < n ? i : n - 1));
  }
  if (acc.x == 12345.678f && acc.y == -1.0f) out[0] = acc;  // never true for the data below: the loads cannot be dropped
}
__global__ __launch_bounds__(64) void write_points(const v4f* __restrict__ in, v4f* __restrict__ out, uint64_t n) {
  const uint64_t base = (uint64_t)blockIdx.x * 256 + threadIdx.x;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const uint64_t i = base + 64 * u;
    if (i < n) __builtin_nontemporal_store(v4f{1.0f, 2.0f, 3.0f, (float)threadIdx.x}, out + i);
  }
}

// ---- nine column streams (deskew_f64cols' pattern) ----
typedef double v2d_u __attribute__((ext_vector_type(2), aligned(8)));
struct Cols { const double* in[5]; double* out[4]; };
template <int NIN, int NOUT, bool SC1>
__global__ __launch_bounds__(64) void copy_cols(Cols c, uint64_t n) {
  extern __shared__ char occupancy_cap[];  // dynamic LDS only limits how many workgroups a CU holds
  const uint64_t base = (uint64_t)blockIdx.x * 128, i = base + 2 * (uint64_t)threadIdx.x;
  if (i + 1 >= n) return;  // (n is a multiple of 128 here)
  v2d_u v[NIN];
#pragma unroll
  for (int k = 0; k < NIN; ++k) v[k] = __builtin_nontemporal_load(reinterpret_cast<const v2d_u*>(c.in[k] + i));
#pragma unroll
  for (int k = 0; k < NOUT; ++k) {
    v2d_u o = v[k + 1];
    o.x += v[0].x * 0.0;  // every output depends on the stamps column like the real kernel's (nothing can be stored before all loads land)
    if constexpr (SC1) {
      const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(c.out[k] + base), 0, 128 * 8, 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, o), r, threadIdx.x * 16u, 0, 2 | 16);
    } else {
      __builtin_nontemporal_store(o, reinterpret_cast<v2d_u*>(c.out[k] + i));
    }
  }
}

int main(int argc, char** argv) {
  const uint64_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 67108864ull;
  const int rounds = argc > 2 ? std::atoi(argv[2]) : 5, iters = argc > 3 ? std::atoi(argv[3]) : 10;
  const size_t shift = argc > 4 ? std::strtoull(argv[4], nullptr, 10) & ~(size_t)15 : 0;
  const bool cols_only = argc > 5 && std::string(argv[5]) == "cols";
  CHECK(hipSetDevice(0));
  hipStream_t s;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const int kBufs = 3;  // rotating pairs: 3 x 2 GiB at the default size, far beyond the 256 MiB Infinity Cache
  v4f *in[kBufs], *out[kBufs];
  for (int b = 0; b < kBufs; ++b) {
    CHECK(hipMalloc(&in[b], n * 16));
    CHECK(hipMalloc(&out[b], n * 16 + shift));
    out[b] = (v4f*)((char*)out[b] + shift);
    CHECK(hipMemsetAsync(in[b], 0x3C, n * 16, s));
  }
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  struct K { const char* name; void (*fn)(const v4f*, v4f*, uint64_t); int block; int points_per_block; double bytes_per_point; };
  const K ks[] = {{"copy_points", copy_points, 256, 256, 32.0}, {"copy_tiles", copy_tiles, 64, 64, 32.0}, {"read_points", read_points, 64, 256, 16.0}, {"write_points", write_points, 64, 256, 16.0}};
  std::vector<std::vector<double>> us(4);
  for (int r = 0; r < (cols_only ? 0 : rounds); ++r)
    for (int k = 0; k < 4; ++k) {  // interleaved: every kernel sees every clock state
      const dim3 grid((unsigned)((n + ks[k].points_per_block - 1) / ks[k].points_per_block)), block(ks[k].block);
      for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(ks[k].fn, grid, block, 0, s, in[w % kBufs], out[w % kBufs], n);
      CHECK(hipEventRecord(e0, s));
      for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(ks[k].fn, grid, block, 0, s, in[it % kBufs], out[it % kBufs], n);
      CHECK(hipEventRecord(e1, s));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      us[k].push_back(ms * 1e3 / iters);
    }
  std::printf("kernel,points,best_us,median_us,GBps_best,GBps_median\n");
  for (int k = 0; k < (cols_only ? 0 : 4); ++k) {
    std::sort(us[k].begin(), us[k].end());
    const double best = us[k].front(), med = us[k][us[k].size() / 2];
    std::printf("%s,%llu,%.2f,%.2f,%.1f,%.1f\n", ks[k].name, (unsigned long long)n, best, med, ks[k].bytes_per_point * n / best / 1e3, ks[k].bytes_per_point * n / med / 1e3);
  }
  // ---- the column streams: the point buffers above are re-used as 2 x 3 sets of columns (6 x 16 n bytes = 12 n doubles >= 9 n) ----
  {
    double* pool[6] = {(double*)in[0], (double*)in[1], (double*)in[2], (double*)((char*)out[0] - shift), (double*)((char*)out[1] - shift), (double*)((char*)out[2] - shift)};
    // every 16 n-byte buffer holds two n-double columns; sets rotate so that consecutive launches touch different memory
    auto col = [&](int j) { return pool[(j / 2) % 6] + (size_t)(j % 2) * n; };
    Cols sets[2];
    for (int sidx = 0; sidx < 2; ++sidx) {
      // set 0: inputs = columns 0..4, outputs = 5..8; set 1: the roles shifted by three buffers
      for (int k = 0; k < 5; ++k) sets[sidx].in[k] = col((k + 6 * sidx) % 12);
      for (int k = 0; k < 4; ++k) sets[sidx].out[k] = col((5 + k + 6 * sidx) % 12);
    }
    struct KC { const char* name; void (*fn)(Cols, uint64_t); double bytes_per_point; unsigned lds; };
    // 160 KiB of LDS per CU, 16 resident one-wave workgroups per CU = 4 per SIMD  <=>  10 KiB each
    const KC kc[] = {{"copy_cols9", copy_cols<5, 4, false>, 72.0, 0},       {"copy_cols9_w4", copy_cols<5, 4, false>, 72.0, 10 * 1024},
                     {"copy_cols9_sc1", copy_cols<5, 4, true>, 72.0, 0},    {"copy_cols7", copy_cols<4, 3, false>, 56.0, 0},
                     {"copy_cols7_w4", copy_cols<4, 3, false>, 56.0, 10 * 1024}};
    const int nk = (int)(sizeof(kc) / sizeof(kc[0]));
    std::vector<std::vector<double>> usc((size_t)nk);
    const dim3 grid((unsigned)(n / 128)), block(64);
    for (int r = 0; r < rounds; ++r)
      for (int k = 0; k < nk; ++k) {
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kc[k].fn, grid, block, kc[k].lds, s, sets[w % 2], n);
        CHECK(hipEventRecord(e0, s));
        for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(kc[k].fn, grid, block, kc[k].lds, s, sets[it % 2], n);
        CHECK(hipEventRecord(e1, s));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        usc[(size_t)k].push_back(ms * 1e3 / iters);
      }
    for (int k = 0; k < nk; ++k) {
      std::sort(usc[(size_t)k].begin(), usc[(size_t)k].end());
      const double best = usc[(size_t)k].front(), med = usc[(size_t)k][usc[(size_t)k].size() / 2];
      std::printf("%s,%llu,%.2f,%.2f,%.1f,%.1f\n", kc[k].name, (unsigned long long)n, best, med, kc[k].bytes_per_point * n / best / 1e3, kc[k].bytes_per_point * n / med / 1e3);
    }
  }
  return 0;
}

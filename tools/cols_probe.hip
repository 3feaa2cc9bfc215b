// cols_probe.hip -- what bounds nine column streams (deskew_f64cols' pattern: 5 x double[n] read, 4 x written)?  (round 5)
// The nine-stream copy of tools/copy_ceiling sits at 0.77-0.80 of HBM peak, the two-stream copy at 0.84-0.85.  This probe varies what a
// caller or the kernel could change:  (a) the SKEW between the columns' base addresses (columns of one allocation are n*8 bytes apart --
// 512 000 000 bytes at the bench's n, a multiple of 4 KiB: element i of every column may land in the same channel / bank group),
// (b) the bytes ONE wave moves per column (1, 2 or 4 KiB contiguous), (c) the workgroup -> tile map (round-robin over the XCDs as the
// hardware deals workgroups out, or one contiguous range of tiles per XCD).  CSV: variant, skew, us, GB/s.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(2); } } while (0)
typedef double v2d_u __attribute__((ext_vector_type(2), aligned(8)));
struct Cols { const double* in[5]; double* out[4]; };

// PER = 16-byte pieces per lane per column (1, 2, 4): a wave moves PER KiB per column, contiguous.  XCD: tiles dealt out so that each XCD owns one range.
template <int PER, bool XCD>
__global__ __launch_bounds__(64) void copy9(Cols c, uint64_t n_tiles) {
  uint64_t tile = blockIdx.x;
  if constexpr (XCD) {
    const uint64_t per_xcd = n_tiles / 8;  // (n_tiles is a multiple of 8 here)
    tile = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  }
  const uint64_t base = tile * (128 * PER) + 2 * (uint64_t)threadIdx.x;
  v2d_u v[5][PER];
#pragma unroll
  for (int k = 0; k < 5; ++k)
#pragma unroll
    for (int p = 0; p < PER; ++p) v[k][p] = __builtin_nontemporal_load(reinterpret_cast<const v2d_u*>(c.in[k] + base + 128 * p));
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      v2d_u o = v[k + 1][p];
      o.x += v[0][p].x * 0.0;
      __builtin_nontemporal_store(o, reinterpret_cast<v2d_u*>(c.out[k] + base + 128 * p));
    }
}

// NIN columns read, NOUT written (1 KiB per wave per column): how the rate falls with the number of streams, and which side pays
template <int NIN, int NOUT>
__global__ __launch_bounds__(64) void streams(Cols c, uint64_t n_tiles) {
  const uint64_t base = (uint64_t)blockIdx.x * 128 + 2 * (uint64_t)threadIdx.x;
  v2d_u v[NIN > 0 ? NIN : 1];
  v2d_u acc = {1.0, 2.0};
#pragma unroll
  for (int k = 0; k < NIN; ++k) { v[k] = __builtin_nontemporal_load(reinterpret_cast<const v2d_u*>(c.in[k] + base)); acc += v[k]; }
  if constexpr (NOUT == 0) {
    if (acc.x == 12345.678 && n_tiles == 1) c.out[0][base] = acc.y;  // (never: keeps the loads alive)
  }
#pragma unroll
  for (int k = 0; k < NOUT; ++k) {
    v2d_u o = NIN > 0 ? v[k % (NIN > 0 ? NIN : 1)] : acc;
    if (NIN > 0) o.x += acc.x * 0.0;  // every store depends on every load
    __builtin_nontemporal_store(o, reinterpret_cast<v2d_u*>(c.out[k] + base));
  }
}

int main(int argc, char** argv) {
  const uint64_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 64000000ull;  // the bench's f64cols size (a multiple of 4096)
  const int rounds = argc > 2 ? std::atoi(argv[2]) : 5, iters = argc > 3 ? std::atoi(argv[3]) : 6;
  CHECK(hipSetDevice(0));
  hipStream_t s;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const size_t slack = 1u << 20;
  const size_t bytes = (size_t)n * 8 * 9 + slack;  // ONE allocation per set, columns n*8 (+ skew) apart -- what torch.empty((9, n)) or a struct of arrays gives
  char* pool[3];
  for (auto& p : pool) { CHECK(hipMalloc(&p, bytes)); CHECK(hipMemsetAsync(p, 0x3C, bytes, s)); }
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const size_t skews[] = {0, 256, 512, 1024, 2048, 4096 + 256, 8192 + 512, 65536 + 4096 + 256};
  struct K { const char* name; void (*fn)(Cols, uint64_t); int per; };
  const K ks[] = {{"1KiB_per_wave", copy9<1, false>, 1}, {"2KiB_per_wave", copy9<2, false>, 2}, {"4KiB_per_wave", copy9<4, false>, 4},
                  {"1KiB_per_wave_xcd_ranges", copy9<1, true>, 1}, {"2KiB_per_wave_xcd_ranges", copy9<2, true>, 2}};
  const int nk = (int)(sizeof(ks) / sizeof(ks[0])), ns = (int)(sizeof(skews) / sizeof(skews[0]));
  std::vector<std::vector<double>> us((size_t)(nk * ns));
  for (int r = 0; r < rounds; ++r)
    for (int si = 0; si < ns; ++si)
      for (int k = 0; k < nk; ++k) {
        Cols sets[3];
        for (int b = 0; b < 3; ++b)
          for (int j = 0; j < 9; ++j) {
            double* col = reinterpret_cast<double*>(pool[b] + (size_t)j * (n * 8 + skews[si]));
            if (j < 5) sets[b].in[j] = col; else sets[b].out[j - 5] = col;
          }
        const uint64_t n_tiles = n / (128 * (uint64_t)ks[k].per);
        const dim3 grid((unsigned)n_tiles), block(64);
        hipLaunchKernelGGL(ks[k].fn, grid, block, 0, s, sets[2], n_tiles);
        CHECK(hipEventRecord(e0, s));
        for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(ks[k].fn, grid, block, 0, s, sets[it % 3], n_tiles);
        CHECK(hipEventRecord(e1, s));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        us[(size_t)(si * nk + k)].push_back(ms * 1e3 / iters);
      }
  std::printf("variant,column_skew_bytes,points,best_us,median_us,GBps_best,GBps_median,frac_of_8TBps_median\n");
  for (int si = 0; si < ns; ++si)
    for (int k = 0; k < nk; ++k) {
      auto& u = us[(size_t)(si * nk + k)];
      std::sort(u.begin(), u.end());
      const double best = u.front(), med = u[u.size() / 2];
      std::printf("%s,%zu,%llu,%.2f,%.2f,%.1f,%.1f,%.4f\n", ks[k].name, skews[si], (unsigned long long)n, best, med, 72.0 * n / best / 1e3, 72.0 * n / med / 1e3, 72.0 * n / med / 1e3 / 8000.0);
    }
  // ---- rate against the number of streams (no skew) ----
  {
    struct S { const char* name; void (*fn)(Cols, uint64_t); int nin, nout; };
    const S ss[] = {{"read_1", streams<1, 0>, 1, 0}, {"read_2", streams<2, 0>, 2, 0}, {"read_5", streams<5, 0>, 5, 0}, {"write_1", streams<0, 1>, 0, 1}, {"write_2", streams<0, 2>, 0, 2},
                    {"write_4", streams<0, 4>, 0, 4}, {"copy_1_1", streams<1, 1>, 1, 1}, {"copy_2_2", streams<2, 2>, 2, 2}, {"copy_3_3", streams<3, 3>, 3, 3}, {"copy_4_4", streams<4, 4>, 4, 4},
                    {"copy_5_1", streams<5, 1>, 5, 1}, {"copy_1_4", streams<1, 4>, 1, 4}, {"copy_5_4", streams<5, 4>, 5, 4}};
    const int nss = (int)(sizeof(ss) / sizeof(ss[0]));
    std::vector<std::vector<double>> u2((size_t)nss);
    Cols sets[3];
    for (int b = 0; b < 3; ++b)
      for (int j = 0; j < 9; ++j) {
        double* col = reinterpret_cast<double*>(pool[b] + (size_t)j * (n * 8));
        if (j < 5) sets[b].in[j] = col; else sets[b].out[j - 5] = col;
      }
    const uint64_t n_tiles = n / 128;
    for (int r = 0; r < rounds; ++r)
      for (int k = 0; k < nss; ++k) {
        hipLaunchKernelGGL(ss[k].fn, dim3((unsigned)n_tiles), dim3(64), 0, s, sets[2], n_tiles);
        CHECK(hipEventRecord(e0, s));
        for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(ss[k].fn, dim3((unsigned)n_tiles), dim3(64), 0, s, sets[it % 3], n_tiles);
        CHECK(hipEventRecord(e1, s));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        u2[(size_t)k].push_back(ms * 1e3 / iters);
      }
    for (int k = 0; k < nss; ++k) {
      auto& u = u2[(size_t)k];
      std::sort(u.begin(), u.end());
      const double by = 8.0 * (ss[k].nin + ss[k].nout) * n, best = u.front(), med = u[u.size() / 2];
      std::printf("%s,0,%llu,%.2f,%.2f,%.1f,%.1f,%.4f\n", ss[k].name, (unsigned long long)n, best, med, by / best / 1e3, by / med / 1e3, by / med / 1e3 / 8000.0);
    }
  }
  return 0;
}

// hostmem_probe.hip -- does the KIND of page-locked host memory matter to the in-place route (one kernel working on the caller's host
// buffers over PCIe, kmc_hip_deskew_f32 with KMC_MEM_HOST_MAPPED)?  A KITTI frame's kernel takes ~1.5x what the link alone would
// need (profiles/r03_inplace_f64.txt); candidates for the rest are the GPU's address translation of 4 KiB host pages and the cache
// policy of the mapping.  Kinds tried, same frame, same call:
//   hipHostMalloc(Portable | Mapped)                    what the library's pool hands out today
//   hipHostMalloc(... | NonCoherent)                    coarse-grained mapping (device caches may hold lines; visible after the sync)
//   hipHostMalloc(... | Coherent)                       fine-grained spelled out
//   2 MiB-aligned anonymous mmap + MADV_HUGEPAGE, touched, hipHostRegister(Mapped | Portable)    physically contiguous 2 MiB pieces if the
//                                                       kernel grants transparent huge pages (AnonHugePages of the range is reported)
//   the same without the madvise                        ordinary 4 KiB pages, registered
//   hipHostMalloc(NumaUser ...) is left out: one socket feeds the GPU on these boxes
//   hostmem_probe [points=123397] [iterations=300]     -> one JSON object; wall clock around the calls (they return with the results in
//                                                      host memory), best and median of the per-call times
#include <hip/hip_runtime_api.h>
#include <sys/mman.h>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "kmc_hip.h"

#define CHECK(x)                                                                              \
  do {                                                                                        \
    hipError_t e_ = (x);                                                                      \
    if (e_ != hipSuccess) {                                                                   \
      std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      std::exit(2);                                                                           \
    }                                                                                         \
  } while (0)

static size_t anon_huge_kb(const void* p) {  // AnonHugePages of the mapping that starts at p, from /proc/self/smaps
  std::ifstream is("/proc/self/smaps");
  std::string line;
  char want[32];
  std::snprintf(want, sizeof(want), "%lx-", (unsigned long)(uintptr_t)p);
  bool in = false;
  while (std::getline(is, line)) {
    if (line.find('-') != std::string::npos && line.find(' ') != std::string::npos && std::isxdigit((unsigned char)line[0]) && line.find("kB") == std::string::npos)
      in = line.rfind(want, 0) == 0;
    else if (in && line.rfind("AnonHugePages:", 0) == 0) {
      std::istringstream ss(line.substr(14));
      size_t kb = 0;
      ss >> kb;
      return kb;
    }
  }
  return 0;
}

struct Buf { float* p = nullptr; size_t bytes = 0; int kind = 0; size_t huge_kb = 0; bool ok = true; };

static Buf make(int kind, size_t bytes) {
  Buf b;
  b.kind = kind;
  b.bytes = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
  if (kind <= 2) {
    const unsigned flags = hipHostMallocPortable | hipHostMallocMapped | (kind == 1 ? hipHostMallocNonCoherent : kind == 2 ? hipHostMallocCoherent : 0u);
    if (hipHostMalloc((void**)&b.p, b.bytes, flags) != hipSuccess) { (void)hipGetLastError(); b.ok = false; }
    return b;
  }
  // over-allocate by 2 MiB and align by hand
  char* raw = (char*)mmap(nullptr, b.bytes + ((size_t)2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (raw == (char*)MAP_FAILED) { b.ok = false; return b; }
  char* al = (char*)(((uintptr_t)raw + ((size_t)2 << 20) - 1) & ~(((uintptr_t)2 << 20) - 1));
  if (kind == 3) (void)madvise(al, b.bytes, MADV_HUGEPAGE);
  else (void)madvise(al, b.bytes, MADV_NOHUGEPAGE);
  std::memset(al, 0, b.bytes);  // touch: the pages exist before they are locked
  b.huge_kb = anon_huge_kb(raw) + (raw != al ? anon_huge_kb(al) : 0);
  if (hipHostRegister(al, b.bytes, hipHostRegisterMapped | hipHostRegisterPortable) != hipSuccess) { (void)hipGetLastError(); b.ok = false; return b; }
  void* dev = nullptr;
  if (hipHostGetDevicePointer(&dev, al, 0) != hipSuccess || dev != al) { (void)hipGetLastError(); b.ok = false; return b; }  // the library needs host address == device address
  b.p = (float*)al;
  return b;
}

int main(int argc, char** argv) {
  const uint64_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 123397ull;
  const int iters = argc > 2 ? std::atoi(argv[2]) : 300;
  CHECK(hipSetDevice(0));
  kmc_ctx* ctx = nullptr;
  if (kmc_hip_create(&ctx, 0) != KMC_OK) { std::fprintf(stderr, "kmc_hip_create failed\n"); return 2; }
  kmc_frame_params prm;
  std::memset(&prm, 0, sizeof(prm));
  prm.twist[0] = 1.3; prm.twist[1] = 0.05; prm.twist[2] = -0.02; prm.twist[5] = 0.03; prm.x_req = 0.5;
  std::vector<float> pts(4 * n);
  for (uint64_t i = 0; i < n; ++i) {
    const double az = -3.14159 + 6.28318 * (double)(i % 2083) / 2083.0, r = 4.0 + (double)(i % 71);
    pts[4 * i] = (float)(r * std::cos(az)); pts[4 * i + 1] = (float)(r * std::sin(az)); pts[4 * i + 2] = -1.7f + 0.03f * (float)(i % 64); pts[4 * i + 3] = 0.01f * (float)(i % 100);
  }
  static const char* names[] = {"hipHostMalloc_portable_mapped", "hipHostMalloc_noncoherent", "hipHostMalloc_coherent", "mmap_thp_registered", "mmap_4k_registered"};
  std::printf("{\"points\": %llu, \"iterations\": %d, \"kinds\": {", (unsigned long long)n, iters);
  std::vector<float> first;
  bool sep = false;
  for (int kind = 0; kind < 5; ++kind) {
    Buf in = make(kind, n * 16), out = make(kind, n * 16);
    if (!in.ok || !out.ok) {
      std::printf("%s\"%s\": null", sep ? ", " : "", names[kind]);
      sep = true;
      continue;
    }
    std::memcpy(in.p, pts.data(), n * 16);
    int rc = KMC_OK;
    for (int w = 0; w < 20 && rc == KMC_OK; ++w) rc = kmc_hip_deskew_f32(ctx, in.p, out.p, n, &prm, KMC_MEM_HOST_MAPPED, nullptr);
    if (rc != KMC_OK) { std::fprintf(stderr, "kind %d: rc %d (%s)\n", kind, rc, kmc_hip_last_error(ctx)); return 2; }
    std::vector<double> t((size_t)iters);
    for (int i = 0; i < iters; ++i) {
      const auto a = std::chrono::steady_clock::now();
      rc = kmc_hip_deskew_f32(ctx, in.p, out.p, n, &prm, KMC_MEM_HOST_MAPPED, nullptr);
      t[(size_t)i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count();
      if (rc != KMC_OK) return 2;
    }
    std::sort(t.begin(), t.end());
    bool same = true;
    if (first.empty()) first.assign(out.p, out.p + 4 * n);
    else same = std::memcmp(first.data(), out.p, n * 16) == 0;
    std::printf("%s\"%s\": {\"us_best\": %.1f, \"us_median\": %.1f, \"GBps_each_way_median\": %.1f, \"AnonHugePages_kB\": %zu, \"same_bits_as_first_kind\": %s}", sep ? ", " : "", names[kind], t[0],
                t[(size_t)iters / 2], 16.0 * (double)n / t[(size_t)iters / 2] * 1e-3, in.huge_kb + out.huge_kb, same ? "true" : "false");
    sep = true;
  }
  std::printf("}}\n");
  kmc_hip_destroy(ctx);
  return 0;
}

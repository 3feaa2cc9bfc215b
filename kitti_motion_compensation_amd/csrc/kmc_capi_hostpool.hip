// kmc_capi_hostpool.hip -- a process-wide pool of page-locked ("pinned"), device-addressable host memory.
//
// Why: the literal reference API hands the library host containers (Frame const& in, Pointcloud by value out,
// motion_compensation.cpp:16-28).  Staging a 123 k-point frame through three copies costs ~230 us, of which ~60 us are the copies'
// fixed costs and the rest one PCIe direction at a time.  If the containers live in page-locked memory the kernel can read and
// write them IN PLACE: one launch, upload and download overlapped on the full-duplex link, no staging buffers.  The drop-in's own
// containers (include/kitti_motion_compensation/data_types.hpp) therefore allocate from this pool; pinning a fresh block costs
// ~100 us or more, so freed blocks are kept and handed out again (the reference allocates a new cloud per call, :21).
//   - size classes: four per octave from 64 KiB up (<= 25 % slack);
//   - at most KMC_HOST_POOL_MAX_MB (default 2048) of FREE blocks are cached, the excess is unpinned on free;
//   - no HIP device (or KMC_HOST_POOL=0): kmc_host_pool_alloc fails with KMC_ERR_NO_DEVICE and the caller uses ordinary memory --
//     an allocation is not a computation, there is still no CPU fallback for the deskew itself;
//   - thread-safe; blocks are never returned to the system at process exit (the HIP runtime may already be gone by then).
#include "kmc_internal.hip.h"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

namespace {

struct Pool {
  std::mutex m;
  struct Block { size_t bytes; int cls; bool in_use; };
  std::map<uintptr_t, Block> blocks;          // every pinned block, by base address
  std::vector<std::vector<void*>> free_lists;  // by size class
  size_t cached_bytes = 0, max_cached = (size_t)2048 << 20;
  int state = 0;  // 0 = untested, 1 = usable, -1 = no device / disabled
};

// Set by an atexit handler registered the first time the pool is found usable -- i.e. AFTER the HIP runtime registered its own
// handlers, so it runs BEFORE them.  From then on nothing is unpinned any more: destructors of static / global containers run
// during exit too, and hipHostFree on a runtime that is shutting down is exactly what the leaked pool object is there to avoid
// (ADVICE r03).  A block freed during teardown is simply left to the operating system.
std::atomic<bool> g_exiting{false};
void mark_exiting() { g_exiting.store(true, std::memory_order_relaxed); }

Pool& pool() {
  static Pool* p = new Pool();  // intentionally leaked: see the header comment
  return *p;
}

constexpr size_t kMinClassBytes = 64 * 1024;

// class c -> bytes: (4 + c % 4) / 4 * 2^(16 + c / 4)
size_t class_bytes(int c) { return ((size_t)(4 + c % 4) << (14 + c / 4)); }
int class_of(size_t bytes) {
  int c = 0;
  while (class_bytes(c) < bytes) ++c;
  return c;
}

bool usable(Pool& p) {  // p.m held
  if (p.state == 0) {
    const char* off = std::getenv("KMC_HOST_POOL");
    int count = 0;
    if ((off && std::atoi(off) == 0) || hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
      (void)hipGetLastError();
      p.state = -1;
    } else {
      p.state = 1;
      std::atexit(mark_exiting);
      if (const char* mb = std::getenv("KMC_HOST_POOL_MAX_MB")) p.max_cached = (size_t)std::max(0, std::atoi(mb)) << 20;
    }
  }
  return p.state == 1;
}

}  // namespace

namespace kmc_impl {
bool host_pool_owns(const void* ptr, size_t bytes) {
  if (!ptr) return false;
  Pool& p = pool();
  std::lock_guard<std::mutex> lock(p.m);
  if (p.blocks.empty()) return false;
  auto it = p.blocks.upper_bound((uintptr_t)ptr);
  if (it == p.blocks.begin()) return false;
  --it;
  return it->second.in_use && (uintptr_t)ptr + bytes <= it->first + it->second.bytes;
}

// Page-locked memory the CALLER allocated (hipHostMalloc, torch's pin_memory(), a registered range whose device address equals its
// host address) is as good as the pool's: the kernels can work on it in place.  One runtime lookup per end of the range; pageable
// memory answers "unregistered" (or an error, in older runtimes) and the call takes the staged route.  KMC_HOST_DETECT_PINNED=0
// restricts the in-place routes to pool memory and explicit KMC_MEM_HOST_MAPPED.
static bool device_addressable_host_byte(const void* p) {
  hipPointerAttribute_t a;
  std::memset(&a, 0, sizeof(a));
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return a.type == hipMemoryTypeHost && a.devicePointer == p;
}

bool host_in_place_ok(const void* ptr, size_t bytes) {
  if (!ptr || bytes == 0) return false;
  if (host_pool_owns(ptr, bytes)) return true;
  static const bool detect = [] {
    const char* e = std::getenv("KMC_HOST_DETECT_PINNED");
    return !(e && std::atoi(e) == 0);
  }();
  if (!detect) return false;
  {
    Pool& p = pool();
    std::lock_guard<std::mutex> lock(p.m);
    if (!usable(p)) return false;  // no HIP device: nothing to ask
  }
  return device_addressable_host_byte(ptr) && device_addressable_host_byte((const char*)ptr + bytes - 1);
}
}  // namespace kmc_impl

extern "C" {

int kmc_host_pool_alloc(size_t bytes, void** out) {
  if (!out) return KMC_ERR_INVALID_ARG;
  *out = nullptr;
  if (bytes == 0) return KMC_OK;
  Pool& p = pool();
  std::lock_guard<std::mutex> lock(p.m);
  if (!usable(p)) return KMC_ERR_NO_DEVICE;
  const int cls = class_of(std::max(bytes, kMinClassBytes));
  if ((size_t)cls < p.free_lists.size() && !p.free_lists[cls].empty()) {
    void* b = p.free_lists[cls].back();
    p.free_lists[cls].pop_back();
    p.blocks[(uintptr_t)b].in_use = true;
    p.cached_bytes -= class_bytes(cls);
    *out = b;
    return KMC_OK;
  }
  void* b = nullptr;
  if (hipHostMalloc(&b, class_bytes(cls), hipHostMallocPortable | hipHostMallocMapped) != hipSuccess || !b) {
    (void)hipGetLastError();
    return KMC_ERR_ALLOC;
  }
  p.blocks[(uintptr_t)b] = {class_bytes(cls), cls, true};
  *out = b;
  return KMC_OK;
}

int kmc_host_pool_free(void* ptr) {
  if (!ptr) return 0;
  Pool& p = pool();
  std::lock_guard<std::mutex> lock(p.m);
  auto it = p.blocks.find((uintptr_t)ptr);
  if (it == p.blocks.end() || !it->second.in_use) return 0;  // not ours: the caller frees it its own way
  it->second.in_use = false;
  const int cls = it->second.cls;
  if (p.cached_bytes + it->second.bytes > p.max_cached) {
    if (!g_exiting.load(std::memory_order_relaxed)) {  // never unpin while the process exits: the block is left to the OS
      (void)hipHostFree(ptr);
      (void)hipGetLastError();
    }
    p.blocks.erase(it);
    return 1;
  }
  if (p.free_lists.size() <= (size_t)cls) p.free_lists.resize(cls + 1);
  p.free_lists[cls].push_back(ptr);
  p.cached_bytes += it->second.bytes;
  return 1;
}

int kmc_host_pool_owns(const void* ptr, size_t bytes) { return kmc_impl::host_pool_owns(ptr, bytes) ? 1 : 0; }

int kmc_host_pool_trim(void) {
  Pool& p = pool();
  std::lock_guard<std::mutex> lock(p.m);
  int released = 0;
  if (g_exiting.load(std::memory_order_relaxed)) return 0;
  for (auto& fl : p.free_lists) {
    for (void* b : fl) {
      (void)hipHostFree(b);
      p.blocks.erase((uintptr_t)b);
      ++released;
    }
    fl.clear();
  }
  (void)hipGetLastError();
  p.cached_bytes = 0;
  return released;
}

}  // extern "C"

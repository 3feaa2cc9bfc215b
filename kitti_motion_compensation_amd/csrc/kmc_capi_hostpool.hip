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

#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <atomic>
#include <cctype>
#include <cstdio>
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
  int fail_at = 0, new_blocks_asked = 0;  // KMC_TEST_HOST_POOL_FAIL_AT=k: the k-th allocation request fails with KMC_ERR_ALLOC (tests only)
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

// ---- NUMA placement of the blocks (Linux system calls, no libnuma) ----
constexpr int kMpolDefault = 0, kMpolPreferred = 1;
int device_numa_node(int dev) {  // of HIP device `dev` (< 0: the calling thread's current device); -1: unknown, or a machine with one node  (pool mutex held)
  static int cached[64];
  static bool known[64];
  char bdf[64] = {0};
  if (dev < 0 && hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  if (dev >= 0 && dev < 64 && known[dev]) return cached[dev];
  if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), dev) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  for (char* q = bdf; *q; ++q) *q = (char)std::tolower((unsigned char)*q);
  char path[160];
  std::snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bdf);
  int node = -1;
  if (FILE* f = std::fopen(path, "r")) {
    if (std::fscanf(f, "%d", &node) != 1) node = -1;
    std::fclose(f);
  }
  if (dev >= 0 && dev < 64) { cached[dev] = node; known[dev] = true; }
  return node;
}
struct SavedPolicy {
  int mode = kMpolDefault;
  unsigned long mask[16] = {0};
  bool valid = false;
};
thread_local SavedPolicy t_saved;
bool prefer_node(int node) {  // remembers the calling thread's own policy (a caller under numactl keeps it)
  unsigned long mask[16] = {0};
  if (node < 0 || node >= (int)(sizeof(mask) * 8)) return false;
  t_saved.valid = syscall(SYS_get_mempolicy, &t_saved.mode, t_saved.mask, sizeof(t_saved.mask) * 8 + 1, nullptr, 0) == 0;
  if (!t_saved.valid) return false;
  mask[node / (8 * sizeof(unsigned long))] |= 1ul << (node % (8 * sizeof(unsigned long)));
  return syscall(SYS_set_mempolicy, kMpolPreferred, mask, sizeof(mask) * 8 + 1) == 0;
}
void restore_policy() {
  if (t_saved.mode == kMpolDefault) (void)syscall(SYS_set_mempolicy, kMpolDefault, nullptr, 0);
  else (void)syscall(SYS_set_mempolicy, t_saved.mode, t_saved.mask, sizeof(t_saved.mask) * 8 + 1);
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
      if (const char* k = std::getenv("KMC_TEST_HOST_POOL_FAIL_AT")) p.fail_at = std::atoi(k);  // test hook: the k-th allocation request of the process fails
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

int kmc_host_pool_alloc(size_t bytes, void** out) { return kmc_host_pool_alloc_near(bytes, -1, out); }

int kmc_host_pool_alloc_near(size_t bytes, int device, void** out) {
  if (!out) return KMC_ERR_INVALID_ARG;
  *out = nullptr;
  if (bytes == 0) return KMC_OK;
  Pool& p = pool();
  std::lock_guard<std::mutex> lock(p.m);
  if (!usable(p)) return KMC_ERR_NO_DEVICE;
  if (p.fail_at > 0 && ++p.new_blocks_asked == p.fail_at) return KMC_ERR_ALLOC;  // KMC_TEST_HOST_POOL_FAIL_AT: fault injection for the tests
  const int cls = class_of(std::max(bytes, kMinClassBytes));
  if ((size_t)cls < p.free_lists.size() && !p.free_lists[cls].empty()) {
    void* b = p.free_lists[cls].back();
    p.free_lists[cls].pop_back();
    p.blocks[(uintptr_t)b].in_use = true;
    p.cached_bytes -= class_bytes(cls);
    *out = b;
    return KMC_OK;
  }
  // On the GPU's side of the machine: an in-place kernel reads and writes the block over the link, and a block on the other socket adds
  // the inter-socket hop to every access (MI355X boxes here: 2 x EPYC, 92 us per KITTI frame with thread and memory on the GPU's node,
  // 104-125 us on the other one; profiles/NOTES.md).  The block is allocated under a PREFERRED memory policy for the device's NUMA node
  // (hipHostMallocNumaUser: "follow the caller's policy"); the calling thread's policy is put back right after.
  void* b = nullptr;
  const int node = device_numa_node(device);
  const bool placed = node >= 0 && prefer_node(node);
  const hipError_t e = hipHostMalloc(&b, class_bytes(cls), hipHostMallocPortable | hipHostMallocMapped | (placed ? hipHostMallocNumaUser : 0u));
  if (placed) restore_policy();
  if (e != hipSuccess || !b) {
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

// Runs the CALLING thread on the CPUs of the device's NUMA node (the PCI device's local_cpulist).  What a deployment does with numactl /
// taskset, for callers that cannot: a call's host side -- argument blocks over the BAR, the read-back, filling a column of a page-locked
// container -- crosses the inter-socket link when the thread sits on the other socket (1.9 against 2.2 us per direct-queue call, 92
// against 104-125 us per in-place KITTI frame on the 2-socket MI355X boxes).  Never done implicitly.  KMC_OK also when there is nothing
// to do (one node, unknown topology); the thread's previous mask is not remembered.
int kmc_hip_bind_thread_near_device(int device) {
  char bdf[64] = {0};
  if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), device) != hipSuccess) {
    (void)hipGetLastError();
    return KMC_ERR_INVALID_ARG;
  }
  for (char* q = bdf; *q; ++q) *q = (char)std::tolower((unsigned char)*q);
  char path[160];
  std::snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/local_cpulist", bdf);
  FILE* f = std::fopen(path, "r");
  if (!f) return KMC_OK;
  char list[1024] = {0};
  const bool got = std::fgets(list, sizeof(list), f) != nullptr;
  std::fclose(f);
  if (!got) return KMC_OK;
  cpu_set_t set;
  CPU_ZERO(&set);
  int n_set = 0;
  for (char* q = list; *q;) {  // "0-63,128-191"
    char* end = nullptr;
    const long a = std::strtol(q, &end, 10);
    if (end == q) break;
    long b = a;
    q = end;
    if (*q == '-') { b = std::strtol(q + 1, &end, 10); q = end; }
    for (long k = a; k <= b && k < CPU_SETSIZE; ++k) { CPU_SET((int)k, &set); ++n_set; }
    if (*q == ',') ++q; else break;
  }
  if (n_set == 0) return KMC_OK;
  // only CPUs the thread may run on anyway (a container's cpuset, an outer taskset): an empty intersection means "nothing to do"
  cpu_set_t allowed, both;
  CPU_ZERO(&allowed);
  if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return KMC_OK;
  CPU_AND(&both, &set, &allowed);
  if (CPU_COUNT(&both) == 0) return KMC_OK;
  (void)sched_setaffinity(0, sizeof(both), &both);  // (a refusal leaves the thread where it was: a placement hint, not a requirement)
  return KMC_OK;
}

}  // extern "C"

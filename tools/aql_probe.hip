// aql_probe.hip -- what would a frame cost BELOW the HIP runtime?  (follow-up of tools/launch_probe)
// Every HIP launch API costs the host 2.2-3.5 us per launch; a KITTI frame's kernel takes 0.6 us.  This probe dispatches the same
// stand-in kernel by writing AQL kernel-dispatch packets into an HSA queue of its own (hsa_queue_create; the code object loaded through
// the HSA loader; kernel arguments in a ring of kernarg memory; one doorbell per packet), with and without the barrier bit, and reports
// the host's time per dispatch and the train's time per dispatch (the device's side).  Device buffers come from hipMalloc (one process,
// one HSA runtime: the addresses are valid on both sides).  Every wait has a timeout: a queue that hangs ends the probe, not the box.
//   aql_probe [dispatches=20000] [points=123397] [code_object=<next to the binary>] [kernargs=device|device_noreadback|host|host_coarse]   -> one JSON object
// host_coarse (round 6): the argument blocks in COARSE-GRAINED host memory -- which the GPU may keep in its L2 between a kernel's acquire and
// release, so that only the first wave per L2 crosses the link for a block instead of every wave -- with a SYSTEM-scope acquire on every
// packet (a block of the lap before must not be served from the L2).  The host then pays no BAR write, no HDP flush and no read-back.
#include <hip/hip_runtime_api.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include <unistd.h>

#define HSA_OK(x)                                                                       \
  do {                                                                                  \
    hsa_status_t s_ = (x);                                                              \
    if (s_ != HSA_STATUS_SUCCESS) {                                                     \
      const char* m_ = nullptr;                                                         \
      hsa_status_string(s_, &m_);                                                       \
      std::printf("{\"failed\": \"%s: %s\"}\n", #x, m_ ? m_ : "?");                    \
      die(3);                                                                    \
    }                                                                                   \
  } while (0)
#define HIP_OK(x)                                                                       \
  do {                                                                                  \
    hipError_t e_ = (x);                                                                \
    if (e_ != hipSuccess) {                                                             \
      std::printf("{\"failed\": \"%s: %s\"}\n", #x, hipGetErrorString(e_));            \
      die(3);                                                                    \
    }                                                                                   \
  } while (0)

[[noreturn]] static void die(int code) {  // the failure paths leave without tearing the runtimes down -- but not without their message
  std::fflush(stdout);
  std::_Exit(code);
}
struct Rec32 { float v[16]; };
struct Rec64 { double v[16]; };
struct Args { const void* in; void* out; uint64_t n; Rec32 f; uint32_t head; uint32_t pad; uint64_t tile_base; Rec64 d; };
static_assert(sizeof(Args) == 232, "the kernel's kernarg segment");

static hsa_agent_t g_gpu, g_cpu;
static bool g_have_gpu = false, g_have_cpu = false;
static hsa_status_t on_agent(hsa_agent_t a, void*) {
  hsa_device_type_t t;
  hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
  if (t == HSA_DEVICE_TYPE_GPU && !g_have_gpu) { g_gpu = a; g_have_gpu = true; }
  if (t == HSA_DEVICE_TYPE_CPU && !g_have_cpu) { g_cpu = a; g_have_cpu = true; }
  return HSA_STATUS_SUCCESS;
}
static hsa_amd_memory_pool_t g_kernarg_pool, g_coarse_host_pool;
static bool g_have_pool = false, g_have_coarse_host_pool = false;
static hsa_status_t on_pool(hsa_amd_memory_pool_t p, void*) {
  hsa_amd_segment_t seg;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
  if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
  uint32_t flags = 0;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
  if ((flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) && !g_have_pool) { g_kernarg_pool = p; g_have_pool = true; }
  bool alloc_ok = false;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc_ok);
  if (alloc_ok && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && !g_have_coarse_host_pool) { g_coarse_host_pool = p; g_have_coarse_host_pool = true; }
  return HSA_STATUS_SUCCESS;
}
// device-local memory the host can write (large BAR): where the HIP runtime itself keeps kernel arguments on this class of device --
// a wave's scalar loads of its arguments then stay on the device instead of crossing the link
static hsa_amd_memory_pool_t g_dev_pool;
static bool g_have_dev_pool = false;
static hsa_status_t on_gpu_pool(hsa_amd_memory_pool_t p, void*) {
  hsa_amd_segment_t seg;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
  if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
  bool alloc_ok = false;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc_ok);
  uint32_t flags = 0;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
  if (alloc_ok && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && !g_have_dev_pool) { g_dev_pool = p; g_have_dev_pool = true; }
  return HSA_STATUS_SUCCESS;
}
using clk = std::chrono::steady_clock;
static double us_since(clk::time_point a) { return std::chrono::duration<double, std::micro>(clk::now() - a).count(); }

int main(int argc, char** argv) {
  const int N = argc > 1 ? std::atoi(argv[1]) : 20000;
  const uint64_t n = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 123397ull;
  HIP_OK(hipSetDevice(0));  // HIP first: it initialises the HSA runtime this process shares
  HSA_OK(hsa_init());
  HSA_OK(hsa_iterate_agents(on_agent, nullptr));
  if (!g_have_gpu || !g_have_cpu) { std::printf("{\"failed\": \"no GPU / CPU agent\"}\n"); return 3; }
  HSA_OK(hsa_amd_agent_iterate_memory_pools(g_cpu, on_pool, nullptr));
  if (!g_have_pool) { std::printf("{\"failed\": \"no kernarg pool\"}\n"); return 3; }
  // the code object, next to this binary
  std::string path = argc > 3 ? argv[3] : "";
  if (path.empty()) {
    char self[4096];
    const ssize_t len = readlink("/proc/self/exe", self, sizeof(self) - 1);
    self[len > 0 ? len : 0] = 0;
    path = std::string(self);
    path = path.substr(0, path.rfind('/')) + "/aql_kernel.hsaco";
  }
  std::ifstream is(path, std::ios::binary);
  std::vector<char> image((std::istreambuf_iterator<char>(is)), std::istreambuf_iterator<char>());
  if (image.empty()) { std::printf("{\"failed\": \"cannot read %s\"}\n", path.c_str()); return 3; }
  hsa_code_object_reader_t reader;
  HSA_OK(hsa_code_object_reader_create_from_memory(image.data(), image.size(), &reader));
  hsa_executable_t exe;
  HSA_OK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
  HSA_OK(hsa_executable_load_agent_code_object(exe, g_gpu, reader, nullptr, nullptr));
  HSA_OK(hsa_executable_freeze(exe, nullptr));
  hsa_executable_symbol_t sym;
  // argv[8] = "walk=<waves>": the walking twin of the kernel, <waves> one-wave workgroups per frame whatever its size (aql_kernel.hip)
  const int walk_waves = (argc > 8 && std::strncmp(argv[8], "walk=", 5) == 0) ? std::atoi(argv[8] + 5) : 0;
  HSA_OK(hsa_executable_get_symbol_by_name(exe, walk_waves > 0 ? "k_frame_walk.kd" : "k_frame.kd", &g_gpu, &sym));
  uint64_t kobj = 0;
  uint32_t karg = 0, group = 0, priv = 0;
  HSA_OK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &kobj));
  HSA_OK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &karg));
  HSA_OK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &group));
  HSA_OK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &priv));
  if (karg != sizeof(Args)) { std::printf("{\"failed\": \"kernarg segment is %u bytes\"}\n", karg); return 3; }
  const uint32_t kQueue = 4096;
  constexpr int kMaxQueues = 4;
  hsa_queue_t* qs[kMaxQueues] = {nullptr, nullptr, nullptr, nullptr};  // [1..]: the further queues of the multi-queue trains (independent frames alternate between them)
  for (auto& qq : qs) HSA_OK(hsa_queue_create(g_gpu, kQueue, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &qq));
  int cur = 0;  // the queue dispatch() writes to
  hsa_queue_t* q = qs[0];
  const bool host_coarse = argc > 4 && std::strcmp(argv[4], "host_coarse") == 0;
  const bool device_kernargs = argc > 4 ? (std::strcmp(argv[4], "host") != 0 && !host_coarse) : true;
  // "device_noreadback": the block is only fenced (sfence) in front of the HDP flush and the doorbell -- three posted writes to one device --
  // without the read over the link that proves they landed.  Every dispatch of the timed trains then carries its own number, and the
  // last frames' outputs are checked: a stale argument block would show.
  const bool readback = !(argc > 4 && std::strcmp(argv[4], "device_noreadback") == 0);
  Args* ring = nullptr;
  hsa_amd_hdp_flush_t hdp = {nullptr, nullptr};
  if (device_kernargs) {
    HSA_OK(hsa_amd_agent_iterate_memory_pools(g_gpu, on_gpu_pool, nullptr));
    if (!g_have_dev_pool) { std::printf("{\"failed\": \"no device-local pool\"}\n"); return 3; }
    HSA_OK(hsa_amd_memory_pool_allocate(g_dev_pool, sizeof(Args) * kQueue * kMaxQueues, 0, (void**)&ring));
    const hsa_status_t acc = hsa_amd_agents_allow_access(1, &g_cpu, nullptr, ring);
    if (acc != HSA_STATUS_SUCCESS) { std::printf("{\"failed\": \"the host cannot map device memory (no large BAR?)\"}\n"); return 3; }
    HSA_OK(hsa_agent_get_info(g_gpu, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_HDP_FLUSH, &hdp));
  } else if (host_coarse) {
    if (!g_have_coarse_host_pool) { std::printf("{\"failed\": \"the CPU agent has no coarse-grained pool\"}\n"); return 3; }
    HSA_OK(hsa_amd_memory_pool_allocate(g_coarse_host_pool, sizeof(Args) * kQueue * kMaxQueues, 0, (void**)&ring));
    HSA_OK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, ring));
  } else {
    HSA_OK(hsa_amd_memory_pool_allocate(g_kernarg_pool, sizeof(Args) * kQueue * kMaxQueues, 0, (void**)&ring));
    HSA_OK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, ring));
  }
  hsa_signal_t done;
  HSA_OK(hsa_signal_create(1, 0, nullptr, &done));

  // argv[5]: buffer pairs the dispatches rotate through (default 8: with 1 M-point frames everything then sits in the 256 MiB Infinity Cache;
  // 64 pairs = 2 GiB do not), argv[6] = "distinct_inputs": every pair has its own input buffer too
  const int kBufs = argc > 5 ? std::max(1, std::min(256, std::atoi(argv[5]))) : 8;
  const bool distinct_inputs = argc > 6 && std::strcmp(argv[6], "distinct_inputs") == 0;
  const bool no_fences = argc > 7 && std::strcmp(argv[7], "no_fences") == 0;
  std::vector<void*> in((size_t)kBufs + 1), out((size_t)kBufs + 1);  // (the extra pair belongs to the drain's own dispatch)
  for (int b = 0; b <= kBufs; ++b) { HIP_OK(hipMalloc(&in[b], n * 16)); HIP_OK(hipMalloc(&out[b], n * 16)); HIP_OK(hipMemset(in[b], 0, n * 16)); HIP_OK(hipMemset(out[b], 0xFF, n * 16)); }
  {  // in[0]: x = 1.0 everywhere, so that out.x = 1 * f.v[0] + f.v[1] + d.v[3] is checkable
    std::vector<float> h(4 * n, 1.0f);
    HIP_OK(hipMemcpy(in[0], h.data(), n * 16, hipMemcpyHostToDevice));
    for (int b = 1; b <= kBufs; ++b) HIP_OK(hipMemcpy(in[b], in[0], n * 16, hipMemcpyDeviceToDevice));
  }
  HIP_OK(hipDeviceSynchronize());
  Args proto;
  std::memset(&proto, 0, sizeof(proto));
  proto.n = n;
  for (int i = 0; i < 16; ++i) { proto.f.v[i] = 1.0f + i; proto.d.v[i] = 0.5 * i; }
  const uint32_t grid = (walk_waves > 0 ? (uint32_t)std::min<uint64_t>((n + 63) / 64, (uint64_t)walk_waves) : (uint32_t)((n + 63) / 64)) * 64;
  uint64_t widxs[kMaxQueues];
  for (int k = 0; k < kMaxQueues; ++k) widxs[k] = hsa_queue_load_write_index_relaxed(qs[k]);
  Args* const ring0 = ring;
  auto wait_for_room = [&](uint64_t idx) {  // never more than kQueue - 64 packets ahead of the packet processor
    const auto t0 = clk::now();
    while (idx - hsa_queue_load_read_index_scacquire(q) >= kQueue - 64) {
      if (us_since(t0) > 5e6) { std::printf("{\"failed\": \"the queue stopped consuming packets\"}\n"); die(4); }
    }
  };
  auto dispatch = [&](int i, bool barrier, hsa_signal_t completion) {
    q = qs[cur];
    uint64_t& widx = widxs[cur];
    auto* packets = (hsa_kernel_dispatch_packet_t*)q->base_address;
    ring = ring0 + (size_t)cur * kQueue;
    wait_for_room(widx);
    Args* a = ring + (widx % kQueue);
    Args mine = proto;
    mine.in = distinct_inputs && i >= 0 ? in[i % kBufs] : in[0];  // x = 1 everywhere
    mine.out = i < 0 ? out[kBufs] : out[i % kBufs];
    if (walk_waves > 0) mine.tile_base = grid / 64;  // the walking kernel's stride (aql_kernel.hip)
    mine.f.v[1] = i < 0 ? 0.0f : (float)(i % 4096);  // this dispatch's own number: out.x = f.v[0] + f.v[1] + d.v[3]
    *a = mine;
    if (device_kernargs) {  // the block went over the BAR: make it land before the packet can be seen (what the HIP runtime does for device kernargs)
      __atomic_thread_fence(__ATOMIC_SEQ_CST);
      if (hdp.HDP_MEM_FLUSH_CNTL) *(volatile uint32_t*)hdp.HDP_MEM_FLUSH_CNTL = 1u;
      if (readback) (void)*(volatile uint32_t*)&a->head;  // read back: the posted writes before it have reached the device
    }
    hsa_kernel_dispatch_packet_t* p = packets + (widx % kQueue);
    p->setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
    p->workgroup_size_x = 64; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
    p->grid_size_x = grid; p->grid_size_y = 1; p->grid_size_z = 1;
    p->private_segment_size = priv; p->group_segment_size = group;
    p->kernel_object = kobj;
    p->kernarg_address = a;
    p->completion_signal = completion;
    // (argv[7] = "no_fences": packets without the barrier bit acquire and release nothing -- what the fences of a dispatch cost; the drain's
    // own packets keep theirs, so the checked outputs are still visible)
    const uint16_t fence = (!barrier && i >= 0 && no_fences) ? HSA_FENCE_SCOPE_NONE : HSA_FENCE_SCOPE_AGENT;
    const uint16_t acquire = host_coarse ? (uint16_t)HSA_FENCE_SCOPE_SYSTEM : fence;  // (the block may sit in the L2 from the lap before)
    if (host_coarse) __atomic_thread_fence(__ATOMIC_RELEASE);
    uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | ((barrier ? 1 : 0) << HSA_PACKET_HEADER_BARRIER) |
                      (acquire << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (fence << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
    __atomic_store_n((uint16_t*)&p->header, header, __ATOMIC_RELEASE);
    hsa_queue_store_write_index_relaxed(q, widx + 1);
    hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)widx);
    ++widx;
  };
  auto drain = [&] {  // per queue: a last packet with the barrier bit and a completion signal; waited for with a timeout
    for (int qi = 0; qi < kMaxQueues; ++qi) {
      cur = qi;
      hsa_signal_store_relaxed(done, 1);
      dispatch(-1, true, done);
      const auto t0 = clk::now();
      while (hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, 100000, HSA_WAIT_STATE_ACTIVE) >= 1) {  // short waits, the wall clock decides
        if (us_since(t0) > 5e6) { std::printf("{\"failed\": \"a dispatched kernel did not complete within 5 s\"}\n"); die(4); }
      }
    }
    cur = 0;
  };
  hsa_signal_t none;
  none.handle = 0;
  // correctness first: one frame, checked
  dispatch(0, true, none);
  drain();
  {
    std::vector<float> h(4 * n);
    HIP_OK(hipMemcpy(h.data(), out[0], n * 16, hipMemcpyDeviceToHost));
    const float want = 1.0f * proto.f.v[0] + (0.0f + (float)proto.d.v[3]);
    bool ok = true;
    for (uint64_t i = 0; i < n && ok; ++i) ok = h[4 * i] == want && h[4 * i + 1] == 1.0f;
    if (!ok) { std::printf("{\"failed\": \"the AQL-dispatched kernel wrote %g, expected %g\"}\n", (double)h[0], (double)want); return 1; }
  }
  std::printf("{\"dispatches\": %d, \"points\": %llu, \"kernarg_bytes\": %u, \"kernargs_in\": \"%s\", \"first_frame_checked\": true", N, (unsigned long long)n, karg,
              host_coarse ? "COARSE-GRAINED host memory, system-scope acquire per packet"
              : !device_kernargs ? "host memory (the kernarg pool): every wave's scalar loads cross the link"
                               : readback ? "device memory written by the host over the BAR (+ HDP flush, read-back)" : "device memory written by the host over the BAR (+ HDP flush, NO read-back)");
  for (int barrier = 1; barrier >= 0; --barrier) {
    for (int i = 0; i < 2000; ++i) dispatch(i, barrier != 0, none);
    drain();
    const auto t0 = clk::now();
    for (int i = 0; i < N; ++i) dispatch(i, barrier != 0, none);
    const double host = us_since(t0) / N;
    drain();
    const double train = us_since(t0) / N;
    {  // the last kBufs dispatches each wrote their own number into their own buffer
      std::vector<float> h(4 * n);
      for (int b = 0; b < kBufs && N >= kBufs; ++b) {
        const int i = N - kBufs + b;
        HIP_OK(hipMemcpy(h.data(), out[i % kBufs], n * 16, hipMemcpyDeviceToHost));
        const float want = proto.f.v[0] + ((float)(i % 4096) + (float)proto.d.v[3]);
        for (uint64_t k = 0; k < n; k += 997)
          if (h[4 * k] != want) { std::printf(", \"failed\": \"dispatch %d wrote %g, expected %g (a stale argument block?)\"}\n", i, (double)h[4 * k], (double)want); die(1); }
      }
    }
    std::printf(", \"%s\": {\"host_us_per_dispatch\": %.3f, \"train_us_per_dispatch\": %.3f}", barrier ? "aql_with_barrier_bit" : "aql_without_barrier_bit", host, train);
  }
  // independent frames alternating between K queues (no barrier bit): does the packet processor overlap what it serialises in one queue?
  for (int K = 2; K <= kMaxQueues; ++K) {
    for (int i = 0; i < 2000; ++i) { cur = i % K; dispatch(i, false, none); }
    drain();
    const auto t0 = clk::now();
    for (int i = 0; i < N; ++i) { cur = i % K; dispatch(i, false, none); }
    const double host = us_since(t0) / N;
    drain();
    const double train = us_since(t0) / N;
    std::printf(", \"aql_%s_queues_alternating_without_barrier_bit\": {\"host_us_per_dispatch\": %.3f, \"train_us_per_dispatch\": %.3f}", K == 2 ? "two" : K == 3 ? "three" : "four", host, train);
  }
  std::printf(", \"waves_per_frame\": %u, \"buffer_pairs\": %d, \"distinct_inputs\": %s, \"unordered_packets_without_fences\": %s", grid / 64, kBufs, distinct_inputs ? "true" : "false", no_fences ? "true" : "false");
  std::printf(", \"note\": \"host = packet + 232-byte kernarg block + doorbell per frame, including the back-pressure of a 4096-packet queue when the device is the slower side\"}\n");
  for (auto& qq : qs) hsa_queue_destroy(qq);
  return 0;
}

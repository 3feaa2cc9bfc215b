// kmc_capi_direct.hip -- the DIRECT QUEUE: device-resident single-frame calls dispatched below the HIP runtime (round 5).
//
// kmc_hip_deskew_f32(KMC_MEM_DEVICE) is the reference's calling pattern -- one frame per call (handlers.cpp:55-64) -- and its cost on
// KITTI-sized frames is the HIP runtime's launch path: 2.2-3.5 us per launch through every launch API (tools/launch_probe), 3.6-4.7 us
// per call, against 0.6 us of kernel.  What a launch IS on this hardware: a 64-byte AQL kernel-dispatch packet in a user-mode queue, the
// kernel's argument block somewhere the waves' scalar loads can reach, a doorbell.  tools/aql_probe measured exactly that: 2.1 us per
// frame with the argument block in DEVICE memory (written by the host over the BAR; in host memory every wave's scalar loads would
// cross the link: 38 us per frame).  So a context on its OWN stream dispatches its frames itself:
//   * an HSA queue of its own (hsa_queue_create, 4096 packets) next to the HIP stream's;
//   * the frame kernels as a raw gfx950 code object (kmc_direct_kernels.hip, embedded below) loaded once per process through the HSA
//     loader -- same tile body, same argument layout, same bits as the HIP code object's deskew_frame_f32 (self-test at open);
//   * a ring of 240-byte argument blocks in device-local memory that the host maps (large BAR), written with ordinary stores, an
//     sfence, the HDP flush register and one read-back -- what the HIP runtime itself does for device-resident kernel arguments;
//   * the AQL barrier bit per packet decided by the same bookkeeping as before (kmc_dispatch_book.hpp): a frame that shares no buffer
//     with the frames in flight goes out without it -- here that is plain AQL semantics, not a launch flag;
//   * ORDER against everything else: the HIP stream and the direct queue are two queues.  A frame waits (on the host) for work the
//     context put on its HIP stream before it (only at such a transition: `stream_dirty`); every other entry point waits for the
//     direct queue to drain first (direct_join: a barrier packet with a completion signal).  A stream of frames pays neither.
// What changes for a caller: on the context's own stream the frames no longer sit in a HIP stream, so a hipDeviceSynchronize() of the
// caller's does not cover them -- kmc_hip_synchronize(ctx) does (include/kmc_hip.h).  A caller's stream (kmc_hip_set_stream), gathered
// calls, per-call timing and KMC_DIRECT_DISPATCH=0 keep the HIP launches.
// Every wait on the queue is bounded: a queue that stops consuming packets or a kernel that does not complete within ten seconds
// turns into KMC_ERR_HIP and the context falls back to HIP launches.
#include "kmc_internal.hip.h"

#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <chrono>
#include <mutex>

// the code object (Makefile: hipcc --genco --no-gpu-bundle-output kmc_direct_kernels.hip -> obj/kmc_direct.hsaco)
__asm__(
    ".section .rodata\n"
    ".balign 4096\n"
    ".global kmc_direct_hsaco_begin\n"
    "kmc_direct_hsaco_begin:\n"
    ".incbin \"" KMC_DIRECT_HSACO_PATH "\"\n"
    ".global kmc_direct_hsaco_end\n"
    "kmc_direct_hsaco_end:\n"
    ".previous\n");
extern "C" const char kmc_direct_hsaco_begin[];
extern "C" const char kmc_direct_hsaco_end[];

namespace kmc_impl {

struct DirectArgs {  // == the kernels' parameter list (kmc_direct_kernels.hip; deskew_frame_f32 in kmc_kernels.hip.h)
  const v4f* in;
  v4f* out;
  uint64_t n;
  FrameRec f;
  uint32_t head;
  uint64_t tile_base;
  FrameRecD d;
};
static_assert(sizeof(DirectArgs) == 240 && offsetof(DirectArgs, f) == 32 && offsetof(DirectArgs, head) == 96 && offsetof(DirectArgs, tile_base) == 104 && offsetof(DirectArgs, d) == 112,
              "the argument block the code object expects");

struct TrajDirectArgs {  // == kmc_direct_traj_t*'s parameter list: an N-knot frame with its segment records in the block
  const v4f* in;
  v4f* out;
  uint64_t n;
  uint32_t n_seg;
  uint32_t head;
  uint64_t tile_base;
  kmc_dev::TrajInline inl;
};
static_assert(sizeof(TrajDirectArgs) == 1152 && offsetof(TrajDirectArgs, tile_base) == 32 && offsetof(TrajDirectArgs, inl) == 48, "the argument block the code object expects");
constexpr size_t kSlotBytes = sizeof(TrajDirectArgs);  // one ring slot holds the larger of the two blocks
static_assert(kSlotBytes % 64 == 0 && sizeof(DirectArgs) <= kSlotBytes, "ring slots are 64-byte aligned");

namespace {
constexpr uint32_t kQueuePackets = 4096;
constexpr uint32_t kArgQuarter = kQueuePackets / 4;
constexpr double kWaitSeconds = 10.0;

struct AgentCode {  // per HSA agent, once per process
  bool tried = false, ok = false;
  hsa_agent_t gpu{}, cpu{};
  hsa_executable_t exe{};
  uint64_t kernel_object[4] = {0, 0, 0, 0};       // kmc_direct_frame_t<tier>
  uint64_t kernel_object_traj[4] = {0, 0, 0, 0};  // kmc_direct_traj_t<tier>
  uint32_t group_size = 0, private_size = 0;
  hsa_amd_memory_pool_t device_pool{};
  hsa_amd_hdp_flush_t hdp = {nullptr, nullptr};
};
std::mutex g_code_mu;
AgentCode g_code[64];

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct FindAgents {
  uint32_t want_bdf = 0, want_domain = 0;
  int gpus_matched = 0;  // GPU agents at the wanted PCI address: more than one = a partitioned device (the partitions share the address)
  bool have_cpu = false;
  hsa_agent_t gpu{}, cpu{};
};
hsa_status_t on_agent(hsa_agent_t a, void* data) {
  auto* f = static_cast<FindAgents*>(data);
  hsa_device_type_t t;
  if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
  if (t == HSA_DEVICE_TYPE_CPU && !f->have_cpu) { f->cpu = a; f->have_cpu = true; }
  if (t == HSA_DEVICE_TYPE_GPU) {
    uint32_t bdf = 0, domain = 0;
    (void)hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf);
    (void)hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &domain);
    // bus and device; the function bits may carry a partition id on partitioned parts, so they are not compared -- and counted instead
    if ((bdf & ~7u) == f->want_bdf && domain == f->want_domain) {
      if (f->gpus_matched == 0) f->gpu = a;
      ++f->gpus_matched;
    }
  }
  return HSA_STATUS_SUCCESS;
}
struct FindPool { bool have = false; hsa_amd_memory_pool_t pool{}; };
hsa_status_t on_gpu_pool(hsa_amd_memory_pool_t p, void* data) {
  auto* f = static_cast<FindPool*>(data);
  hsa_amd_segment_t seg;
  if (hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg) != HSA_STATUS_SUCCESS || seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
  bool alloc_ok = false;
  (void)hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc_ok);
  uint32_t flags = 0;
  (void)hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
  if (alloc_ok && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && !f->have) { f->pool = p; f->have = true; }
  return HSA_STATUS_SUCCESS;
}

// the code object on HIP device `device`'s agent, loaded once per process.  The agent is matched by PCI address and must be the ONLY GPU
// agent there: on a partitioned device (CPX / DPX modes) several agents share one address, the match would be a guess, and a packet on
// the wrong partition's queue reads memory that is mapped to its owner only -- such devices keep the HIP launches.
AgentCode* agent_code(int device) {
  if (device < 0 || device >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(g_code_mu);
  AgentCode& ac = g_code[device];
  if (ac.tried) return ac.ok ? &ac : nullptr;
  ac.tried = true;
  int bus = 0, dev = 0, domain = 0;
  if (hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, device) != hipSuccess || hipDeviceGetAttribute(&dev, hipDeviceAttributePciDeviceId, device) != hipSuccess ||
      hipDeviceGetAttribute(&domain, hipDeviceAttributePciDomainID, device) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  if (hsa_init() != HSA_STATUS_SUCCESS) return nullptr;  // (reference-counted: HIP holds the runtime already)
  FindAgents fa;
  fa.want_bdf = ((uint32_t)bus << 8) | ((uint32_t)dev << 3);  // function 0
  fa.want_domain = (uint32_t)domain;
  if (hsa_iterate_agents(on_agent, &fa) != HSA_STATUS_SUCCESS || fa.gpus_matched != 1 || !fa.have_cpu) return nullptr;
  ac.gpu = fa.gpu;
  ac.cpu = fa.cpu;
  FindPool fp;
  if (hsa_amd_agent_iterate_memory_pools(ac.gpu, on_gpu_pool, &fp) != HSA_STATUS_SUCCESS || !fp.have) return nullptr;
  ac.device_pool = fp.pool;
  if (hsa_agent_get_info(ac.gpu, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_HDP_FLUSH, &ac.hdp) != HSA_STATUS_SUCCESS) return nullptr;
  hsa_code_object_reader_t reader;
  const size_t image_bytes = (size_t)(kmc_direct_hsaco_end - kmc_direct_hsaco_begin);
  if (hsa_code_object_reader_create_from_memory(kmc_direct_hsaco_begin, image_bytes, &reader) != HSA_STATUS_SUCCESS) return nullptr;
  if (hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &ac.exe) != HSA_STATUS_SUCCESS) return nullptr;
  if (hsa_executable_load_agent_code_object(ac.exe, ac.gpu, reader, nullptr, nullptr) != HSA_STATUS_SUCCESS) return nullptr;
  if (hsa_executable_freeze(ac.exe, nullptr) != HSA_STATUS_SUCCESS) return nullptr;
  static const char* const names[8] = {"kmc_direct_frame_t0.kd", "kmc_direct_frame_t1.kd", "kmc_direct_frame_t2.kd", "kmc_direct_frame_t3.kd",
                                       "kmc_direct_traj_t0.kd",  "kmc_direct_traj_t1.kd",  "kmc_direct_traj_t2.kd",  "kmc_direct_traj_t3.kd"};
  for (int t = 0; t < 8; ++t) {
    hsa_executable_symbol_t sym;
    uint32_t karg = 0, group = 0, priv = 0;
    uint64_t* object = t < 4 ? &ac.kernel_object[t] : &ac.kernel_object_traj[t - 4];
    if (hsa_executable_get_symbol_by_name(ac.exe, names[t], &ac.gpu, &sym) != HSA_STATUS_SUCCESS) return nullptr;
    if (hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, object) != HSA_STATUS_SUCCESS) return nullptr;
    (void)hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &karg);
    (void)hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &group);
    (void)hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &priv);
    if (karg != (t < 4 ? sizeof(DirectArgs) : sizeof(TrajDirectArgs)) || priv != 0) return nullptr;  // (a kernel that needs scratch would need the queue's scratch set up: not these)
    ac.group_size = std::max(ac.group_size, group);
    ac.private_size = std::max(ac.private_size, priv);
  }
  ac.ok = true;
  return &ac;
}
}  // namespace

// TWO LANES.  One AQL queue processes its packets one after the other: between the last workgroup of one frame's dispatch and the first
// of the next the packet processor spends ~1.4 us in which nothing is dispatched (tools/aql_probe: 5.9 us per 1 M-point frame through
// one queue, 4.5 us with independent frames alternating between two).  So the context has two queues ("lanes"):
//   * a frame that is INDEPENDENT of every frame in flight (kmc_dispatch_book.hpp's LaneWindow says so) goes to the lane the previous
//     independent frame did not take, without the barrier bit;
//   * a frame whose conflicts -- a buffer shared with a frame in flight -- all sit in ONE lane goes to THAT lane with the barrier bit: it
//     waits for that lane's packets and runs beside the other lane's; nothing crosses lanes (a chain of dependent frames stays in its
//     lane, three buffer pairs used in rotation keep both lanes busy);
//   * a FULLY ORDERED frame -- conflicts in both lanes, the window full or invalid, KMC_ANY_ORDER=0 -- goes to lane 0 with the barrier
//     bit.  If lane 1 has taken frames since the last synchronisation point, a barrier packet on lane 1 first signals "lane 1 done up
//     to here" and a barrier packet on lane 0 waits for that signal.  The frame carries a completion signal; the next packet that goes
//     to lane 1 is preceded by a barrier packet that waits for it (that frame may depend on ANYTHING before the fully ordered frame --
//     those frames have left the window --, and "the ordered frame has completed" implies all of them have).  The signals come in
//     records of three (the frame's, lane 1's "done up to here", the lane-1 barrier packet's own) out of a ring; a record is re-armed
//     only when its frame AND the lane-1 barrier packet that referenced it have
//     completed -- a barrier packet still queued must never find its signal re-armed for a later frame (that later frame would wait for
//     lane 1, which waits for the packet: a deadlock the first two-lane version ran into).
// direct_join drains both lanes and ends the window.
constexpr int kLanes = 2;
constexpr int kOrderRecords = 64;
struct OrderRec {
  hsa_signal_t s{};  // completion of the ordered frame
  hsa_signal_t x{};  // completion of lane 1's "done up to here" barrier packet (lane 0 waits for it in front of the frame)
  hsa_signal_t w{};  // completion of the lane-1 barrier packet that waits for `s`
  bool w_used = false;
};

struct Lane {
  hsa_queue_t* q = nullptr;
  char* ring = nullptr;        // kQueuePackets slots of kSlotBytes: device-local memory, host-mapped
  hsa_signal_t done{};         // direct_join's completion signal
  // the argument ring is re-used a quarter at a time: a marker (barrier packet with a completion signal) follows each quarter's dispatches,
  // and a quarter is written again only when its marker of the lap before has completed -- no wave can still be reading a block
  hsa_signal_t quarter_done[4] = {};
  bool quarter_armed[4] = {false, false, false, false};
  uint64_t aidx = 0;           // next argument slot (mod kQueuePackets)
  uint64_t widx = 0;           // next packet slot (single producer: the context's calling thread)
  uint64_t room_until = 0;     // packets below this index fit the queue for sure (wait_for_room)
  bool first_after_transition = true;
};
struct DirectQueue {
  AgentCode* code = nullptr;
  Lane lane[kLanes];
  OrderRec order[kOrderRecords];           // used round-robin, one per ordered frame
  uint32_t next_record = 0;
  kmc_book::LaneSync sync;                 // which cross-lane waits the next frame needs (kmc_dispatch_book.hpp: unit-tested against a model of two queues)
  OrderRec* last_full = nullptr;           // the last fully ordered frame's record (what lane 1's next packet waits for, if sync says it must)
  bool two_lanes = true;                   // false with KMC_ANY_ORDER=0: every frame ordered, nothing for a second lane to overlap
  uint64_t frames = 0;
};

namespace {
// room for one more packet?  (never more than kQueuePackets - 64 ahead of the packet processor)  false: the queue stopped consuming
bool wait_for_room(Lane* l) {
  // (the read index is a word the packet processor rewrites with every packet: reading it misses the host's caches each time -- so the
  // bound it gave is used until it is exhausted)
  if (l->widx < l->room_until) return true;
  const double t0 = now_s();
  for (;;) {
    l->room_until = hsa_queue_load_read_index_scacquire(l->q) + (kQueuePackets - 64);
    if (l->widx < l->room_until) return true;
    if (now_s() - t0 > kWaitSeconds) return false;
  }
}
void ring_doorbell(Lane* l, void* packet, uint16_t header, uint16_t setup_or_rest) {
  // header and the following 16 bits are published together, last, with release semantics: the packet processor may look at the slot at any time
  const uint32_t word = (uint32_t)header | ((uint32_t)setup_or_rest << 16);
  __atomic_store_n(reinterpret_cast<uint32_t*>(packet), word, __ATOMIC_RELEASE);
  hsa_queue_store_write_index_relaxed(l->q, l->widx + 1);
  hsa_signal_store_screlease(l->q->doorbell_signal, (hsa_signal_value_t)l->widx);
  ++l->widx;
}
// The queue made no progress within the timeout: THIS call reports it (frames still in the queue are lost, which is what the error says);
// the context declares the queue dead -- nothing waits for it again (a later join must not stall another ten seconds), frames go out as
// HIP launches from here on.
int queue_stuck(kmc_ctx* c, const char* what) {
  c->last_error = what;
  c->dd_broken = true;
  c->dd_pending = false;
  c->lw.invalidate();
  return KMC_ERR_HIP;
}
// a barrier packet on lane `l`: waits for every packet before it on that lane (barrier bit) and for `dep` (handle 0: none); signals `completion` (handle 0: none)
int barrier_packet(kmc_ctx* c, Lane* l, hsa_signal_t dep, hsa_signal_t completion, uint16_t scope) {
  if (!wait_for_room(l)) return queue_stuck(c, "direct queue: the packet processor stopped consuming packets");
  auto* p = reinterpret_cast<hsa_barrier_and_packet_t*>(l->q->base_address) + (l->widx % kQueuePackets);
  std::memset(reinterpret_cast<char*>(p) + 4, 0, sizeof(*p) - 4);
  p->dep_signal[0] = dep;
  p->completion_signal = completion;
  const uint16_t header = (HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) | (scope << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                          (scope << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
  ring_doorbell(l, p, header, 0);
  return KMC_OK;
}
// the next record of the ring.  Its previous user -- 64 ordered frames ago -- and the lane-1 barrier packet that waited for that frame must
// have completed (bounded waits on device progress that depends on nothing the host has yet to do).
int take_order_record(kmc_ctx* c, DirectQueue* d, OrderRec** out) {
  OrderRec* r = &d->order[d->next_record];
  d->next_record = (d->next_record + 1) % kOrderRecords;
  const double t0 = now_s();
  while (hsa_signal_load_scacquire(r->s) > 0 || hsa_signal_load_scacquire(r->x) > 0 || (r->w_used && hsa_signal_load_scacquire(r->w) > 0))
    if (now_s() - t0 > kWaitSeconds) return queue_stuck(c, "direct queue: an ordered frame did not complete within the timeout");
  r->w_used = false;
  if (d->last_full == r) {  // (it has completed: nothing to wait for)
    d->last_full = nullptr;
    d->sync.last_full_has_completed();
  }
  *out = r;
  return KMC_OK;
}
}  // namespace

void direct_close(kmc_ctx* c) {
  DirectQueue* d = c->dd;
  if (!d) return;
  c->dd = nullptr;
  for (Lane& l : d->lane) {
    if (l.q) (void)hsa_queue_destroy(l.q);
    if (l.ring) (void)hsa_amd_memory_pool_free(l.ring);
    if (l.done.handle) (void)hsa_signal_destroy(l.done);
    for (hsa_signal_t& sg : l.quarter_done)
      if (sg.handle) (void)hsa_signal_destroy(sg);
  }
  for (OrderRec& r : d->order)
    for (hsa_signal_t* sg : {&r.s, &r.x, &r.w})
      if (sg->handle) (void)hsa_signal_destroy(*sg);
  delete d;
}

// both lanes have drained: every frame dispatched so far has completed and released its stores (bounded wait)
int direct_join(kmc_ctx* c) {
  DirectQueue* d = c->dd;
  if (!d || !c->dd_pending) return KMC_OK;
  for (Lane& l : d->lane) {
    hsa_signal_store_relaxed(l.done, 1);
    const int rc = barrier_packet(c, &l, hsa_signal_t{0}, l.done, HSA_FENCE_SCOPE_SYSTEM);
    if (rc != KMC_OK) return rc;
  }
  for (Lane& l : d->lane) {
    const double t0 = now_s();
    while (hsa_signal_wait_scacquire(l.done, HSA_SIGNAL_CONDITION_LT, 1, 200000, HSA_WAIT_STATE_ACTIVE) >= 1)
      if (now_s() - t0 > kWaitSeconds) return queue_stuck(c, "direct queue: a dispatched frame did not complete within the timeout");
    l.first_after_transition = true;  // whatever comes next on the HIP stream may rewrite the frames' buffers: the lane's next frame re-acquires at system scope
  }
  c->dd_pending = false;
  c->lw.invalidate();  // nothing in flight; the next frame re-acquires at system scope, fully ordered
  d->sync.joined();
  d->last_full = nullptr;
  return KMC_OK;
}

// one frame = one packet (frames beyond 2^26 - 1 tiles: several, told their first tile).  `v`: the window's verdict -- independent of
// every frame in flight (lane v.lane, no barrier bit), behind the frames of lane v.lane only (that lane, barrier bit), or behind
// everything (lane 0, cross-lane wait).  `args`: the kernel's argument block (arg_bytes of it, <= kSlotBytes, 16-byte aligned),
// `tile_base_at`: where its first-tile field sits.
namespace {
int dispatch_frame(kmc_ctx* c, uint64_t kernel_object, void* args, size_t arg_bytes, size_t tile_base_at, uint64_t n_tiles, kmc_book::LaneVerdict v, uint32_t* launches_out) {
  DirectQueue* d = c->dd;
  // (a frame of several packets must have been admitted as fully ordered: direct_frame_is_huge)
  if (n_tiles > kMaxTilesPerLaunch) v = {kmc_book::LaneVerdict::kFullyOrdered, 0};  // (admitted as such already: direct_frame_is_huge)
  hsa_signal_t completion{0};
  OrderRec* rec = nullptr;
  if (d->two_lanes && v.kind == kmc_book::LaneVerdict::kFullyOrdered) {  // (before the plan: a record that has completed may clear a wait lane 1 still owes)
    const int rc = take_order_record(c, d, &rec);
    if (rc != KMC_OK) return rc;
  }
  const kmc_book::LanePlan plan = d->sync.plan(v, d->two_lanes ? 2 : 1);
  const int li = plan.lane;
  if (plan.cross_lane_wait) {  // lane 0 waits for what lane 1 has taken so far
    hsa_signal_store_relaxed(rec->x, 1);
    int rc = barrier_packet(c, &d->lane[1], hsa_signal_t{0}, rec->x, HSA_FENCE_SCOPE_AGENT);
    if (rc == KMC_OK) rc = barrier_packet(c, &d->lane[0], rec->x, hsa_signal_t{0}, HSA_FENCE_SCOPE_AGENT);
    if (rc != KMC_OK) return rc;
  }
  if (plan.wait_for_last_full && d->last_full) {  // everything before the last fully ordered frame must be over before lane 1 goes on
    OrderRec* r = d->last_full;
    hsa_signal_store_relaxed(r->w, 1);
    r->w_used = true;
    const int rc = barrier_packet(c, &d->lane[1], r->s, r->w, HSA_FENCE_SCOPE_AGENT);
    if (rc != KMC_OK) return rc;
  }
  if (plan.completion_signal) {
    hsa_signal_store_relaxed(rec->s, 1);
    completion = rec->s;
  }
  Lane* l = &d->lane[li];
  uint32_t launches = 0;
  for (uint64_t t0 = 0; t0 < n_tiles; t0 += kMaxTilesPerLaunch, ++launches) {
    if (!wait_for_room(l)) return queue_stuck(c, "direct queue: the packet processor stopped consuming packets");
    const uint32_t tiles = (uint32_t)std::min<uint64_t>(kMaxTilesPerLaunch, n_tiles - t0);
    const bool last_packet = t0 + kMaxTilesPerLaunch >= n_tiles;
    const uint32_t slot = (uint32_t)(l->aidx % kQueuePackets);
    if (slot % kArgQuarter == 0) {
      const uint32_t q = slot / kArgQuarter, prev = (q + 3) % 4;
      if (l->aidx != 0) {  // the quarter just filled: its marker, behind every dispatch so far on this lane
        hsa_signal_store_relaxed(l->quarter_done[prev], 1);
        const int rc = barrier_packet(c, l, hsa_signal_t{0}, l->quarter_done[prev], HSA_FENCE_SCOPE_AGENT);
        if (rc != KMC_OK) return rc;
        l->quarter_armed[prev] = true;
      }
      if (l->quarter_armed[q]) {  // this quarter's blocks of the lap before (3072 slots ago: over long since, one load)
        const double t_wait = now_s();
        while (hsa_signal_load_scacquire(l->quarter_done[q]) > 0)
          if (now_s() - t_wait > kWaitSeconds) return queue_stuck(c, "direct queue: a quarter of the argument ring did not drain within the timeout");
        l->quarter_armed[q] = false;
      }
      if (!wait_for_room(l)) return queue_stuck(c, "direct queue: the packet processor stopped consuming packets");
    }
    ++l->aidx;
    char* a = l->ring + (size_t)slot * kSlotBytes;
    std::memcpy(static_cast<char*>(args) + tile_base_at, &t0, sizeof(t0));
    std::memcpy(a, args, arg_bytes);  // over the BAR, write-combined: one sequential pass over the block
    // the block must have landed in device memory before the packet processor can see the packet: fence, HDP flush, one read back over the
    // link (what the HIP runtime does for device-resident kernel arguments; tools/aql_probe: without the read-back 1.5 us per frame instead
    // of 2.1 and no stale block in 40 000 dispatches -- kept all the same: a stale argument block is a silently wrong frame)
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    if (d->code->hdp.HDP_MEM_FLUSH_CNTL) *(volatile uint32_t*)d->code->hdp.HDP_MEM_FLUSH_CNTL = 1u;
    (void)*(volatile uint32_t*)(a + arg_bytes - 4);
    auto* p = reinterpret_cast<hsa_kernel_dispatch_packet_t*>(l->q->base_address) + (l->widx % kQueuePackets);
    p->workgroup_size_x = kTile; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
    p->reserved0 = 0;
    p->grid_size_x = tiles * (uint32_t)kTile; p->grid_size_y = 1; p->grid_size_z = 1;
    p->private_segment_size = d->code->private_size;
    p->group_segment_size = d->code->group_size;
    p->kernel_object = kernel_object;
    p->kernarg_address = a;
    p->reserved2 = 0;
    p->completion_signal.handle = last_packet ? completion.handle : 0;
    // ordered packets acquire at agent scope (the frame before them may have written what they read), a lane's first one behind HIP-stream
    // work at system scope (copies, host writes); every frame releases at agent scope, direct_join's barrier packets at system scope
    const bool ordered = plan.barrier_bit || l->first_after_transition || t0 != 0;
    const uint16_t acquire = l->first_after_transition ? HSA_FENCE_SCOPE_SYSTEM : HSA_FENCE_SCOPE_AGENT;
    const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | ((ordered ? 1 : 0) << HSA_PACKET_HEADER_BARRIER) |
                            (acquire << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
    ring_doorbell(l, p, header, (uint16_t)(1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS));
    l->first_after_transition = false;
  }
  if (rec) d->last_full = rec;  // (the latest fully ordered frame: its completion implies that of everything before it)
  c->dd_pending = true;
  ++d->frames;
  if (launches_out) *launches_out = launches;
  return KMC_OK;
}
}  // namespace

// a frame that takes several packets (beyond 2^26 - 1 tiles): the caller has the window admit it as fully ordered
bool direct_frame_is_huge(uint64_t n) { return (n + 2 * kTile - 1) / kTile > kMaxTilesPerLaunch; }

int direct_frame(kmc_ctx* c, int tier, const v4f* in, v4f* out, uint64_t n, const FrameRec& f, const FrameRecD& rd, uint32_t head, kmc_book::LaneVerdict v, uint32_t* launches_out) {
  alignas(64) DirectArgs mine;
  mine.in = in - head; mine.out = out - head; mine.n = n + head; mine.f = f; mine.head = head; mine.tile_base = 0; mine.d = rd;
  return dispatch_frame(c, c->dd->code->kernel_object[tier], &mine, sizeof(mine), offsetof(DirectArgs, tile_base), (mine.n + kTile - 1) / kTile, v, launches_out);
}

// an N-knot frame whose segment records (<= kInlineSegments) travel in the argument block: deskew_traj_f32<tier, false, true>'s twin
int direct_traj_frame(kmc_ctx* c, int tier, const v4f* in, v4f* out, uint64_t n, uint32_t n_seg, uint32_t head, const kmc_dev::TrajInline& inl, kmc_book::LaneVerdict v,
                      uint32_t* launches_out) {
  alignas(64) TrajDirectArgs mine;
  mine.in = in - head; mine.out = out - head; mine.n = n + head; mine.n_seg = n_seg; mine.head = head; mine.tile_base = 0; mine.inl = inl;
  return dispatch_frame(c, c->dd->code->kernel_object_traj[tier], &mine, sizeof(mine), offsetof(TrajDirectArgs, tile_base), (mine.n + kTile - 1) / kTile, v, launches_out);
}

// Opens the context's direct queue (once; nullptr afterwards if this device / runtime cannot: no large BAR, no HDP flush register, the
// self-test differs).  The self-test runs one 1000-point frame through the direct queue and through a HIP launch and compares the bits.
bool direct_open(kmc_ctx* c) {
  if (c->dd) return true;
  if (!c->dd_wanted || c->dd_never || c->dd_tried) return false;
  c->dd_tried = true;
  // everything below -- the self-test's buffers, copies and its HIP launch above all -- belongs on the context's device, whatever device the
  // calling thread had current (a thread that holds contexts for several GPUs)
  if (hipSetDevice(c->device) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  AgentCode* code = agent_code(c->device);
  if (!code) return false;
  DirectQueue* d = new (std::nothrow) DirectQueue();
  if (!d) return false;
  d->code = code;
  c->dd = d;
  if (!c->dd_free_order) d->two_lanes = false;  // every frame ordered (KMC_ANY_ORDER=0): nothing for a second lane to overlap
  c->lw.lanes = d->two_lanes ? 2 : 1;
  c->lw.invalidate();
  bool ok = true;
  for (Lane& l : d->lane) {
    ok = ok && hsa_queue_create(code->gpu, kQueuePackets, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &l.q) == HSA_STATUS_SUCCESS;
    ok = ok && hsa_amd_memory_pool_allocate(code->device_pool, kSlotBytes * kQueuePackets, 0, (void**)&l.ring) == HSA_STATUS_SUCCESS;
    ok = ok && hsa_amd_agents_allow_access(1, &code->cpu, nullptr, l.ring) == HSA_STATUS_SUCCESS;  // fails without a large BAR
    ok = ok && hsa_signal_create(1, 0, nullptr, &l.done) == HSA_STATUS_SUCCESS;
    for (hsa_signal_t& sg : l.quarter_done) ok = ok && hsa_amd_signal_create(0, 0, nullptr, HSA_AMD_SIGNAL_AMD_GPU_ONLY, &sg) == HSA_STATUS_SUCCESS;
    if (ok) l.widx = hsa_queue_load_write_index_relaxed(l.q);
  }
  for (OrderRec& r : d->order)
    for (hsa_signal_t* sg : {&r.s, &r.x, &r.w})  // waited for by barrier packets only (the host polls their values, it never sleeps on them): no interrupt per completion
      ok = ok && hsa_amd_signal_create(0, 0, nullptr, HSA_AMD_SIGNAL_AMD_GPU_ONLY, sg) == HSA_STATUS_SUCCESS;
  // ---- self-test: the direct queue's frame against the HIP launch's, bit for bit ----
  float *t_in = nullptr, *t_a = nullptr, *t_b = nullptr;
  constexpr uint64_t kN = 1000;
  if (ok) {
    ok = hipMalloc((void**)&t_in, kN * 16) == hipSuccess && hipMalloc((void**)&t_a, kN * 16) == hipSuccess && hipMalloc((void**)&t_b, kN * 16) == hipSuccess;
    std::vector<float> h(4 * kN), ha(4 * kN), hb(4 * kN);
    if (ok) ok = kmc_synth_points_host(h.data(), kN, 0xD1EC7) == KMC_OK && hipMemcpy(t_in, h.data(), kN * 16, hipMemcpyHostToDevice) == hipSuccess &&
                 hipMemset(t_a, 0xFF, kN * 16) == hipSuccess && hipMemset(t_b, 0, kN * 16) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
    kmc_frame_params prm;
    std::memset(&prm, 0, sizeof(prm));
    prm.twist[0] = 1.3; prm.twist[1] = 0.05; prm.twist[2] = -0.02; prm.twist[3] = 0.002; prm.twist[4] = -0.004; prm.twist[5] = 0.03; prm.x_req = 0.4;
    FrameRec f;
    std::memset(&f, 0, sizeof(f));
    fill_rec(prm, &f);
    f.pre2 = guard_pre2(prm);
    FrameRecD rd;
    fill_recd(prm, &rd);
    if (ok) {
      hipLaunchKernelGGL(deskew_frame_f32<kSeries3>, dim3((unsigned)((kN + kTile - 1) / kTile)), dim3(kTile), 0, c->own_stream, (const v4f*)t_in, (v4f*)t_a, kN, f, 0u, (uint64_t)0, rd);
      ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(c->own_stream) == hipSuccess;
    }
    // through lane 1 (an independent frame behind an ordered one: the cross-lane wait is exercised too), then through lane 0
    if (ok) ok = direct_frame(c, kSeries3, (const v4f*)t_in, (v4f*)t_b, 64, f, rd, 0, {kmc_book::LaneVerdict::kFullyOrdered, 0}, nullptr) == KMC_OK &&
                 direct_frame(c, kSeries3, (const v4f*)t_in, (v4f*)t_b, kN, f, rd, 0, {kmc_book::LaneVerdict::kLaneOrdered, d->two_lanes ? 1 : 0}, nullptr) == KMC_OK && direct_join(c) == KMC_OK;
    if (ok) ok = hipMemcpy(ha.data(), t_a, kN * 16, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(hb.data(), t_b, kN * 16, hipMemcpyDeviceToHost) == hipSuccess &&
                 std::memcmp(ha.data(), hb.data(), kN * 16) == 0 && hipMemset(t_b, 0, kN * 16) == hipSuccess && hipDeviceSynchronize() == hipSuccess;  // (a memset is asynchronous)
    if (ok) ok = direct_frame(c, kSeries3, (const v4f*)t_in, (v4f*)t_b, kN, f, rd, 0, {kmc_book::LaneVerdict::kFullyOrdered, 0}, nullptr) == KMC_OK && direct_join(c) == KMC_OK;
    if (ok) ok = hipMemcpy(ha.data(), t_a, kN * 16, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(hb.data(), t_b, kN * 16, hipMemcpyDeviceToHost) == hipSuccess &&
                 std::memcmp(ha.data(), hb.data(), kN * 16) == 0;
  }
  if (t_in) (void)hipFree(t_in);
  if (t_a) (void)hipFree(t_a);
  if (t_b) (void)hipFree(t_b);
  (void)hipGetLastError();
  if (!ok) {
    direct_close(c);
    c->dd_broken = false;  // (not broken: absent -- HIP launches from the start)
    c->dd_pending = false;
    return false;
  }
  c->dd->frames = 0;
  return true;
}

}  // namespace kmc_impl

extern "C" uint64_t kmc_hip_direct_frames(kmc_ctx* c) { return (c && c->dd) ? c->dd->frames : 0; }

// Opt in / out (include/kmc_hip.h).  Switching off first waits for the frames still in the queue: from the return on, everything the context
// issues is in its HIP stream again.  The queue itself stays open (switching on again costs nothing).
extern "C" int kmc_hip_set_direct_dispatch(kmc_ctx* c, int enabled) {
  if (!c) return KMC_ERR_INVALID_ARG;
  if (!enabled && c->dd_pending) {
    const int rc = kmc_impl::direct_join(c);
    if (rc != KMC_OK) return rc;
  }
  c->dd_wanted = enabled != 0;
  return KMC_OK;
}

// 1: this context's eligible frames go through the direct queue (asked for, opened, self-test passed); 0: HIP launches (not asked for, the
// device / runtime cannot, KMC_DIRECT_DISPATCH=0, or the queue broke).  Asked for but not opened yet: opens it (once).
extern "C" int kmc_hip_direct_dispatch_active(kmc_ctx* c) {
  if (!c || !c->dd_wanted || c->dd_never || c->dd_broken) return 0;
  return (c->dd || kmc_impl::direct_open(c)) ? 1 : 0;
}

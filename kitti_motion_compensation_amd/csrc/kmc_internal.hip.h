// kmc_internal.hip.h -- shared by the translation units that implement the C-ABI of include/kmc_hip.h (library-internal):
// the context, error plumbing, launch geometry, the table-slot ring and the call timer.
//   kmc_capi_core.hip     context life cycle, streams, timers, host pre-step entry points, the shared helpers' definitions
//   kmc_capi_deskew.hip   single-frame, batched and f64 Eigen-layout deskew, pseudo time stamps
//   kmc_capi_traj.hip     N-knot trajectory entry points and their f64 host pre-step
//   kmc_capi_project.hip  LiDAR -> image projection (row N4)
//   kmc_capi_synth.hip    synthetic workload generator
//   kmc_capi_hostpool.hip process-wide pool of page-locked, device-addressable host memory (what the in-place routes work on)
// There is no CPU fallback anywhere: every hot-path entry point needs a live kmc_ctx, and kmc_hip_create() fails without a
// HIP device.
#pragma once

#include "../../include/kmc_hip.h"

#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "kmc_dispatch_book.hpp"
#include "kmc_host_math.hpp"
#include "kmc_kernels.hip.h"

using namespace kmc_dev;

namespace kmc_impl { struct DirectQueue; }

struct kmc_ctx {
  int device = -1;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  bool timing = false;
  hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr, ev_c0 = nullptr, ev_c1 = nullptr, ev_t0 = nullptr, ev_t1 = nullptr;
  std::string last_error;
  hipDeviceProp_t prop;
  int force_tier = -1;    // kmc_hip_force_tier (testing hook)
  // out-of-range counter (f64 path)
  unsigned long long* d_counter = nullptr;
  int f64_pending = 0;         // kmc_hip_deskew_f64cols_begin / _end: 0 = none, 1 = issued and not waited for, 2 = completed inside _begin
  int f64_result = 0;
  kmc_stats f64_stats = {};
  bool counter_dirty = true;   // d_counter may be non-zero: the f64 entry points clear it only then (a memset per call costs ~5 us)
  static constexpr int mapped_waves = 128;      // persistent one-wave workgroups of the f64 kernel when it works on page-locked host memory (64 ... 1024 measured: profiles/r03_inplace_f64.txt)
  uint32_t* h_flag = nullptr;  // page-locked word the f64 kernels raise when a stamp is out of range (read by the host after the sync: no D2H copy on the good path)
  // completion word of the in-place kernels (kmc_kernels.hip.h, DoneWord): the last wave stores the call's sequence number into h_done,
  // the host spins on it instead of waiting for the stream (~8.5 us per call).  All in h_flag's page-locked block, one cache line each.
  uint32_t* h_done = nullptr;    // h_flag + 16 words
  uint64_t* h_stamps = nullptr;  // h_flag + 32 words: device clock at the first wave's start / at the last ticket (call trace)
  uint32_t* d_ticket = nullptr;  // device word next to d_counter, 0 between kernels
  uint32_t done_seq = 0;         // sequence number of the last launch that carries a completion word
  bool done_armed = false;       // such a launch is in flight and nobody has waited for it yet
  uint64_t done_fallbacks = 0;   // waits that ended on an idle stream without the word (wait_done_word): stream-synchronised instead
  uint32_t done_fallback_state[3] = {0, 0, 0};  // the last such event: sequence number expected, word seen, ticket seen
  bool trace = false;            // kmc_hip_enable_call_trace
  kmc_call_trace last_trace = {};
  // batch tables: a ring of slots, each one device buffer + one pinned staging buffer holding
  // [BatchRec x n_frames | coarse x (n_chunks + 1)], uploaded with ONE copy on a side stream so that the per-step host
  // preparation and the table H2D overlap the previous step's kernel.  The compute stream sees no event between two
  // launches except one "consumed" marker per group of kSlotsPerGroup launches (a marker between kernels costs ~3 us).
  struct TableSlot {
    char* d_buf = nullptr;
    char* h_buf = nullptr;
    size_t cap = 0;
    hipEvent_t uploaded = nullptr;  // tables are on the device (copy stream)
  };
  static constexpr int kSlotsPerGroup = 4;
  static constexpr int kSlotGroups = 4;
  static constexpr int kTableSlots = kSlotsPerGroup * kSlotGroups;
  TableSlot slots[kTableSlots];
  hipEvent_t group_consumed[kSlotGroups] = {nullptr, nullptr, nullptr, nullptr};  // kernels of the group finished
  bool group_busy[kSlotGroups] = {false, false, false, false};
  bool group_dirty[kSlotGroups] = {false, false, false, false};  // slots handed out since the group's last marker (an error
                                                                 // return between slot_begin and slot_end leaves no marker)
  int next_slot = 0;
  hipStream_t copy_stream = nullptr;
  // host-staging buffers
  // host-buffer pipeline: dedicated upload / compute / download streams over a ring of device slots
  static constexpr int kPipeSlots = 4;
  hipStream_t pipe[3] = {nullptr, nullptr, nullptr};  // [0] H2D, [1] kernels, [2] D2H
  void* d_stage_in[kPipeSlots] = {nullptr, nullptr, nullptr, nullptr};
  void* d_stage_out[kPipeSlots] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_h2d[kPipeSlots] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_kernel[kPipeSlots] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_d2h[kPipeSlots] = {nullptr, nullptr, nullptr, nullptr};
  size_t stage_cap = 0;  // bytes per buffer
  std::vector<hipEvent_t> ev_pool;  // grow-only events of the chunked f64 host route (two per chunk)
  void* d_traj = nullptr; // segment tables of the N-knot trajectory kernels (16 x TrajSeg32 + 16 x TrajSeg64)
  void* h_traj = nullptr; // pinned staging of the same size
  hipEvent_t ev_traj = nullptr;  // last upload from h_traj has completed
  bool traj_in_flight = false;
  void* d_tmp = nullptr; // grow-only scratch for the f64 / batch host paths
  size_t tmp_cap = 0;
  // gathered frames (kmc_hip_set_frame_queues(ctx, q > 1)): device-resident single-frame calls are not launched one by one -- a launch
  // costs the host 2.5-5 us and the device ~2 us of drain and refill, against 0.6 us of kernel for a KITTI frame -- but gathered on the
  // host and issued as ONE launch of the frame-list kernel (kmc_kernels.hip.h) per up to kGatherMax frames, their records travelling in
  // the kernel arguments.  Pending frames go out when the list is full, when the context's stream has run dry (checked every few
  // frames), before a frame that touches a pending frame's buffers or needs another coefficient tier, and before anything else the
  // context puts on its stream (fq_join, which every other entry point starts with).
  static constexpr int kMaxFrameQueues = 4;              // the API's range of `queues`; any value > 1 switches gathering on
  static constexpr int kGatherMax = 64;                  // pending frames at most (a list only grows that long while the device is busy); their records travel in the list launch's kernel arguments
  static constexpr uint64_t kGatherMaxPoints = 1ull << 24;  // larger frames gain nothing from sharing a launch (and a 2-D grid of kGatherMax such frames stays below 2^32 work-items)
  using AoRange = kmc_book::Range;
  int fq_count = 1;                  // 1 = off: every call launches on `stream`
  bool fq_ordered = true;            // kmc_hip_set_frame_queue_order: on a CALLER's stream, producers may sit between two calls
  ListRec* gather = nullptr;         // the pending frames' records (kGatherMax of each)
  FrameRecD* gather64 = nullptr;
  kmc_book::GatherList<kGatherMax> gl;  // how many are pending, their tier and address ranges, and the decisions around a new frame (kmc_dispatch_book.hpp)
  // the direct queue (kmc_capi_direct.hip): device-resident single-frame calls on the context's OWN stream are dispatched as AQL packets in an
  // HSA queue of the context's, below the HIP runtime's launch path
  kmc_impl::DirectQueue* dd = nullptr;
  bool dd_wanted = false;            // the caller asked for it (kmc_hip_set_direct_dispatch, or KMC_DIRECT_DISPATCH=1 at kmc_hip_create): OFF by default --
                                     // frames in the direct queue are outside every HIP stream, which a caller has to know about (include/kmc_hip.h)
  bool dd_never = false;             // KMC_DIRECT_DISPATCH=0: not even when asked (A/B runs of a client that asks)
  bool dd_tried = false;             // direct_open() has been attempted
  bool dd_pending = false;           // frames are in the direct queue that nobody has waited for (direct_join)
  bool dd_broken = false;            // a wait on the queue timed out: HIP launches from here on
  bool dd_free_order = true;         // independent frames go out without the AQL barrier bit (KMC_ANY_ORDER=0: every packet carries it)
  bool stream_dirty = true;          // the context has put work on its HIP stream since the last host wait for it: a direct frame waits first
  bool big_kernargs = true;          // kernel-argument blocks beyond 4 KiB are taken by this runtime (cleared by the first refused launch: launch_list)
  int fq_error = 0;                  // sticky: a join failed to issue gathered frames whose calls had already returned KMC_OK (fq_join)
  uint64_t fq_dropped = 0;           // how many frames that has cost so far (kmc_hip_frame_queue_dropped)
  // independent frames on ONE stream without the drain between them: a device-resident single-frame launch whose buffers overlap
  // nothing that was launched since (and including) the last ORDERED launch goes out with hipExtAnyOrderLaunch -- the dispatch packet
  // carries no barrier bit, the frame starts while the frame before it is still running.  Everything else the context puts on
  // its stream is an ordinary (barrier) packet and waits for all of them (tools/anyorder_probe.hip measures both facts).  Only on the
  // context's OWN stream: a caller's stream may hold producers the library does not see.
  static constexpr int kAoWindow = 128;  // frames between two ordered launches at most
  bool ao_enabled = false;               // set by kmc_hip_create from the run-time probe's verdict (kmc_capi_core.hip); KMC_ANY_ORDER=0 turns it off
  int ao_verdict = 0;                    // kmc_device_info.any_order_dispatch
  bool ao_probed = false;                // the verdict has been established (ao_ensure: at first need, not in kmc_hip_create)
  kmc_book::LaneWindow<kAoWindow> lw;      // the direct queue's frames in flight, with their lanes (kmc_capi_direct.hip)
  kmc_book::AnyOrderWindow<kAoWindow> ao;  // the frames in flight behind the last ordered launch, and the admission rule (kmc_dispatch_book.hpp)
};

namespace kmc_impl {

constexpr uint64_t kHostChunkPoints = 1ull << 21;  // 32 MiB per direction per pipeline slot

inline int fail_hip(kmc_ctx* c, hipError_t e, const char* what) {
  if (c) {
    c->last_error = std::string(what) + ": " + hipGetErrorString(e);
  }
  (void)hipGetLastError();
  return KMC_ERR_HIP;
}

#define KMC_HIP_TRY(ctx, expr)                        \
  do {                                                \
    hipError_t _e = (expr);                           \
    if (_e != hipSuccess) return fail_hip(ctx, _e, #expr); \
  } while (0)

// f64 kernels: the 8-term series is evaluated at t / 2^h and doubled back h times; it needs t / 2^h <= 0.5 (truncation < 1e-19).
// h from the frame's |phi| (|s| <= 1 inside a scan / a segment): 0 for every vehicle, whatever it takes for a caller-supplied
// twist (a fixed 3 lost accuracy beyond ~4 rad).  Wave-uniform loop count in the kernel.
inline int halvings_for(double phi2) {
  const double phi = std::sqrt(phi2);
  if (!(phi > 0.5)) return 0;
  const int h = (int)std::ceil(std::log2(phi / 0.5));
  return h < 0 ? 0 : (h > 60 ? 60 : h);
}

// ... and how many terms of it: 5 up to kShortSeriesTheta rad per scan (every vehicle), else 8 (NaN -> 8)
inline int series_terms_for(double phi2) { return std::sqrt(phi2) <= kShortSeriesTheta ? kShortSeriesTerms : 8; }

// the cheapest tier valid up to theta_max = max over the frames of |phi| * max|s| (NaN -> the any-angle tier)
inline int tier_of_theta(double theta_max) {
  if (theta_max <= kThetaSeries3) return kSeries3;
  if (theta_max <= kThetaSeries5) return kSeries5;
  if (theta_max <= kThetaWide) return kWide;
  return kTrig;
}
int pick_tier(const kmc_ctx* c, const kmc_frame_params* p, uint32_t n);

// gathered frames: fq_join() issues whatever is pending as one list launch on `stream` (and ends the any-order window): what every entry
// point that puts other work on the stream starts with; gather_push() adds a frame (kmc_capi_deskew.hip)
int fq_join(kmc_ctx* c);
// ONE launch of the frame-list kernel for `count` filled records on the context's stream: kernel-argument records for at most
// up to kInlineListFramesMax (256) frames per launch (16 under stream capture and for `inline_only`).  -> launches_out
int launch_list(kmc_ctx* c, const ListRec* recs, const FrameRecD* recd, uint32_t count, int tier, uint32_t* launches_out, bool inline_only = false);
int fq_take_error(kmc_ctx* c);  // the sticky error of a join that could not issue its frames (reported once)

// what every entry point that issues work on `stream` starts with
#define KMC_ENTER(ctx)                                      \
  do {                                                      \
    (ctx)->ao.invalidate();                                 \
    (ctx)->stream_dirty = true;                             \
    KMC_HIP_TRY(ctx, hipSetDevice((ctx)->device));          \
    const int rc_join_ = fq_join(ctx);                      \
    if (rc_join_ != KMC_OK) return rc_join_;                \
  } while (0)

template <typename REC>
void fill_rec(const kmc_frame_params& p, REC* r) {
  const kmc_host::Vec3 rho = {p.twist[0], p.twist[1], p.twist[2]};
  const kmc_host::Vec3 phi = {p.twist[3], p.twist[4], p.twist[5]};
  const kmc_host::Vec3 c1 = kmc_host::cross(phi, rho);
  const kmc_host::Vec3 c2 = kmc_host::cross(phi, c1);
  r->phi_x = (float)phi.x; r->phi_y = (float)phi.y; r->phi_z = (float)phi.z;
  r->phi2 = (float)kmc_host::dot(phi, phi);
  r->rho_x = (float)rho.x; r->rho_y = (float)rho.y; r->rho_z = (float)rho.z;
  r->s0 = (float)(0.5 - p.x_req);
  r->c1_x = (float)c1.x; r->c1_y = (float)c1.y; r->c1_z = (float)c1.z;
  r->c2_x = (float)c2.x; r->c2_y = (float)c2.y; r->c2_z = (float)c2.z;
}
// near-origin guard, stage 1 threshold of a two-pose frame (kmc_device_math.hip.h)
inline float guard_pre2(const kmc_frame_params& p) {
  return kGuardPre * (float)(p.twist[0] * p.twist[0] + p.twist[1] * p.twist[1] + p.twist[2] * p.twist[2]);
}

// the same constants in f64, for the near-origin guard's redo (kmc_device_math.hip.h)
inline void fill_recd(const kmc_frame_params& p, FrameRecD* d) {
  const kmc_host::Vec3 rho = {p.twist[0], p.twist[1], p.twist[2]};
  const kmc_host::Vec3 phi = {p.twist[3], p.twist[4], p.twist[5]};
  const kmc_host::Vec3 c1 = kmc_host::cross(phi, rho);
  const kmc_host::Vec3 c2 = kmc_host::cross(phi, c1);
  d->phi[0] = phi.x; d->phi[1] = phi.y; d->phi[2] = phi.z;
  d->rho[0] = rho.x; d->rho[1] = rho.y; d->rho[2] = rho.z;
  d->c1[0] = c1.x; d->c1[1] = c1.y; d->c1[2] = c1.z;
  d->c2[0] = c2.x; d->c2[1] = c2.y; d->c2[2] = c2.z;
  d->phi2 = kmc_host::dot(phi, phi);
  d->x_req = p.x_req;
  d->pad[0] = d->pad[1] = 0.0;
}

// Device-resident buffers: distance (in points, < 64) from the last 1 KiB boundary to the start of the OUTPUT.  The kernels are
// launched on pointers moved back by that much with the first `head` indices dead, so that every tile stores whole aligned
// lines whatever 16-byte-aligned address the caller passes ("alignment").
inline uint32_t head_of(const void* out, int mem_kind) {
  return mem_kind == KMC_MEM_DEVICE ? (uint32_t)(((uintptr_t)out >> 4) & 63u) : 0u;
}

inline bool params_ok(const kmc_frame_params* p) {
  for (int i = 0; i < 6; ++i)
    if (!std::isfinite(p->twist[i])) return false;
  return std::isfinite(p->x_req);
}

// One workgroup per tile, no tile loops (kmc_kernels.hip.h).  The dispatch packet carries the grid in WORK-ITEMS in 32 bits, and beyond
// 2^32 / 64 workgroups (4.29 G points -- 137 GB of cloud in + out, which this GPU holds) the runtime does NOT refuse the launch: it wraps
// the grid modulo 2^32 work-items and reports success (tools/grid_probe.hip, profiles/r02_grid_probe.txt).  Every tiled launch therefore
// goes through launch_tiles: at most kMaxTilesPerLaunch tiles per launch, each launch told its first tile.
constexpr uint64_t kMaxTilesPerLaunch = 0xFFFFFFFFull / 64;
template <typename L>
inline uint32_t launch_tiles(uint64_t n_tiles, L&& launch) {  // launch(first_tile, tiles_in_this_launch); -> number of launches
  uint32_t launches = 0;
  for (uint64_t t0 = 0; t0 < n_tiles; t0 += kMaxTilesPerLaunch, ++launches) launch(t0, (int)std::min<uint64_t>(kMaxTilesPerLaunch, n_tiles - t0));
  return launches;
}

// ---- launch plumbing: run-time choices -> template arguments, and the one place that knows the two launch calls ----
template <typename F>
inline void with_bool(bool b, F&& f) {
  if (b) f(std::true_type{});
  else f(std::false_type{});
}
template <typename F>
inline void with_tier(int tier, F&& f) {
  switch (tier) {
    case kSeries3: f(std::integral_constant<int, kSeries3>{}); break;
    case kSeries5: f(std::integral_constant<int, kSeries5>{}); break;
    case kWide: f(std::integral_constant<int, kWide>{}); break;
    default: f(std::integral_constant<int, kTrig>{}); break;
  }
}
// `any_order`: dispatch without the AQL barrier bit (hipExtAnyOrderLaunch), see kmc_ctx::ao
template <typename... KArgs, typename... Args>
inline void launch_on(void (*kernel)(KArgs...), int grid, int block, hipStream_t s, bool any_order, Args&&... args) {
  static_assert(sizeof...(KArgs) == sizeof...(Args), "kernel argument count");
  if (any_order)
    hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, s, nullptr, nullptr, (uint32_t)hipExtAnyOrderLaunch, static_cast<KArgs>(args)...);
  else
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, s, static_cast<KArgs>(args)...);
}

// the direct queue (kmc_capi_direct.hip)
bool direct_open(kmc_ctx* c);   // the context's queue exists (opened at first need; false: not asked for (kmc_ctx::dd_wanted), not on this device / runtime, or KMC_DIRECT_DISPATCH=0)
void direct_close(kmc_ctx* c);
int direct_join(kmc_ctx* c);    // every frame dispatched through the queue has completed (bounded wait)
bool direct_frame_is_huge(uint64_t n);
int direct_frame(kmc_ctx* c, int tier, const v4f* in, v4f* out, uint64_t n, const FrameRec& f, const FrameRecD& d, uint32_t head, kmc_book::LaneVerdict v, uint32_t* launches_out);
int direct_traj_frame(kmc_ctx* c, int tier, const v4f* in, v4f* out, uint64_t n, uint32_t n_seg, uint32_t head, const kmc_dev::TrajInline& inl, kmc_book::LaneVerdict v, uint32_t* launches_out);
void ao_ensure(kmc_ctx* c);  // runs the dispatch probe if its verdict is not known yet
bool ao_admit(kmc_ctx* c, const void* in, const void* out, uint64_t bytes, bool same_call);  // may this frame be dispatched without the barrier bit?  (kmc_capi_core.hip)
bool host_pool_owns(const void* ptr, size_t bytes);  // inside a live block of the page-locked host pool (kmc_capi_hostpool.hip)
bool host_in_place_ok(const void* ptr, size_t bytes);  // pool memory, or the caller's own page-locked memory that the device addresses at the same address
int ensure_tmp(kmc_ctx* c, size_t bytes);  // grow-only device scratch of the host-buffer paths
int ensure_pipeline(kmc_ctx* c);           // streams, events and device slots of the three-stage host pipeline
int ensure_pipe_streams(kmc_ctx* c);       // only its three streams
int ensure_events(kmc_ctx* c, size_t count);  // at least `count` events in ev_pool
// the in-place routes' completion word: done_word_arm() hands out the next launch's DoneWord, wait_done_word() returns when the last
// armed launch has raised it (everything it stored is then in host memory) -- or when the stream reports an error
DoneWord done_word_arm(kmc_ctx* c);
int wait_done_word(kmc_ctx* c);
double trace_now_us();

// ---- ring of table slots (batch tables and trajectory segment tables) ---------------------------------------------------
// slot_begin : picks the next slot, waits (host) until the kernels of its group from the previous lap are done, grows every
//              slot if `need` bytes do not fit;
// slot_upload: one H2D copy of the slot's pinned staging on the side stream, then a HOST wait for that tiny copy -- the
//              launch that follows has no cross-stream dependency, so back-to-back launches keep the ~2 us same-stream boundary;
// slot_end   : after the launch; records one "consumed" marker per group of launches on the compute stream.
int slot_begin(kmc_ctx* c, size_t need, int* slot_id_out);
int slot_upload(kmc_ctx* c, int slot_id, size_t bytes);
int slot_end(kmc_ctx* c, int slot_id);

struct CallTimer {
  kmc_ctx* c;
  explicit CallTimer(kmc_ctx* ctx) : c(ctx) {}
  int begin_call() { return c->timing ? (hipEventRecord(c->ev_c0, c->stream) == hipSuccess ? KMC_OK : KMC_ERR_HIP) : KMC_OK; }
  int begin_kernel() { return c->timing ? (hipEventRecord(c->ev_k0, c->stream) == hipSuccess ? KMC_OK : KMC_ERR_HIP) : KMC_OK; }
  int end_kernel() { return c->timing ? (hipEventRecord(c->ev_k1, c->stream) == hipSuccess ? KMC_OK : KMC_ERR_HIP) : KMC_OK; }
  int end_call(kmc_stats* st) {
    if (!c->timing) return KMC_OK;
    KMC_HIP_TRY(c, hipEventRecord(c->ev_c1, c->stream));
    KMC_HIP_TRY(c, hipEventSynchronize(c->ev_c1));
    if (st) {
      KMC_HIP_TRY(c, hipEventElapsedTime(&st->kernel_ms, c->ev_k0, c->ev_k1));
      KMC_HIP_TRY(c, hipEventElapsedTime(&st->total_ms, c->ev_c0, c->ev_c1));
    }
    return KMC_OK;
  }
};

// coarse[c] = {frame that owns point c * chunk (empty frames skipped), split}; coarse[n_chunks].x = frame of the last point.
// All positions are VIRTUAL: `head` dead points precede the batch (frame 0 owns them), n_virtual = n + head.
void build_coarse(const uint64_t* offsets, uint32_t n_frames, uint64_t n_virtual, uint32_t head, uint2* h_coarse, uint32_t chunk_shift = kChunkShift);

}  // namespace kmc_impl

using namespace kmc_impl;

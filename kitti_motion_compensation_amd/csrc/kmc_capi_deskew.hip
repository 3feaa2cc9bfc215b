// kmc_capi_deskew.hip -- the hot path's entry points in the KITTI f32 layout: single frame (HIP launch, direct queue, gathered),
// list of separate frames, batch of frames.  Thin on purpose: argument checks, f64 -> device-precision frame records, launch geometry,
// optional host staging; all per-point work is in kmc_kernels.hip.h.  (The f64 Eigen-layout entry points: kmc_capi_f64.hip.)
#include "kmc_internal.hip.h"

#include <hip/hip_ext.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <system_error>
#include <thread>

namespace {

// ---- template dispatch: the run-time choices of a call (tier, index output, kernel-argument tables, barrier bit) become template
// arguments through with_tier / with_bool, every launch goes through launch_on (kmc_internal.hip.h) ----
// The in-place routes end with "the results are in host memory": the kernel's last wave raises a page-locked completion word and the
// host spins on it (wait_done_word, kmc_capi_core.hip) -- until round 4 a hipStreamQuery spin plus a hipStreamSynchronize, which
// returned ~8.5 us after the last byte had landed (tools/link_probe).  Visibility does not rest on the runtime's wait any more: every
// wave releases its stores at system scope before its ticket (DoneWord, kmc_kernels.hip.h).
// in / out / n are the caller's; `head` dead points are put in front (pointers moved back, n grown) -- see head_of().
// any_order: the dispatch packet carries no barrier bit (hipExtAnyOrderLaunch) -- see kmc_ctx::ao
uint32_t launch_frame(hipStream_t s, int tier, const v4f* in, v4f* out, uint64_t n, const FrameRec& f, const FrameRecD& d, uint32_t head = 0, bool any_order = false) {
  in -= head;
  out -= head;
  n += head;
  uint32_t launches = 0;
  with_tier(tier, [&](auto T) {
    launches = launch_tiles((n + kTile - 1) / kTile, [&](uint64_t t0, int grid) { launch_on(deskew_frame_f32<decltype(T)::value>, grid, kTile, s, any_order, in, out, n, f, head, t0, d); });
  });
  return launches;  // 1 unless the frame holds more than 2^32 - 64 points
}

// batch of frames, tables in device memory (INLINE = false) or in the kernel arguments (at most 16 frames)
template <bool INLINE>
uint32_t launch_batch(int tier, hipStream_t s, const v4f* in, v4f* out, const BatchRec* recs, const uint2* coarse, uint32_t nf, uint64_t n, uint32_t* idx, uint32_t head,
                  const FrameRecD* recs64, uint32_t chunk_shift, const float* pre2s, const BatchInlineArg<INLINE>& inl) {
  uint32_t launches = 0;
  with_tier(tier, [&](auto T) {
    with_bool(idx != nullptr, [&](auto IDX) {
      launches = launch_tiles((n + kTile - 1) / kTile, [&](uint64_t t0, int grid) {
        launch_on(deskew_batch_f32<decltype(T)::value, decltype(IDX)::value, INLINE>, grid, kTile, s, false, in, out, recs, coarse, nf, n, idx, head, recs64, chunk_shift, pre2s,
                  t0, inl);
      });
    });
  });
  return launches;
}
}  // namespace

// ---- helpers of the single-frame entry points ----
namespace {
// one device-resident frame on stream `s` (arguments already validated)
int issue_frame(kmc_ctx* c, hipStream_t s, const float* xyzi_in, float* xyzi_out, uint64_t n, const kmc_frame_params* params, int* tier_out,
                bool any_order = false, uint32_t* launches_out = nullptr) {
  const int tier = pick_tier(c, params, 1);
  FrameRec f;
  std::memset(&f, 0, sizeof(f));
  fill_rec(*params, &f);
  f.pre2 = guard_pre2(*params);
  FrameRecD d;
  fill_recd(*params, &d);
  if (tier_out) *tier_out = tier;
  if (launches_out) *launches_out = 0;
  if (n == 0) return KMC_OK;
  const uint32_t launches = launch_frame(s, tier, (const v4f*)xyzi_in, (v4f*)xyzi_out, n, f, d, head_of(xyzi_out, KMC_MEM_DEVICE), any_order);
  if (launches_out) *launches_out = launches;
  KMC_HIP_TRY(c, hipGetLastError());
  return KMC_OK;
}
int check_frame_args(const float* xyzi_in, float* xyzi_out, uint64_t n, const kmc_frame_params* params, int mem_kind) {
  if (!params || (n && (!xyzi_in || !xyzi_out))) return KMC_ERR_INVALID_ARG;
  // device pointers feed 16-byte vector accesses; host buffers are only ever the source / destination of copies
  if (mem_kind == KMC_MEM_DEVICE && (((uintptr_t)xyzi_in | (uintptr_t)xyzi_out) & 15u)) return KMC_ERR_INVALID_ARG;
  if (((uintptr_t)xyzi_in | (uintptr_t)xyzi_out) & 3u) return KMC_ERR_INVALID_ARG;
  if (!params_ok(params)) return KMC_ERR_INVALID_ARG;
  if (!(params->x_req >= 0.0 && params->x_req <= 1.0)) return KMC_ERR_TIME_OUT_OF_RANGE;
  return KMC_OK;
}
constexpr uint64_t kMappedMinPoints = 2048;  // below this a kernel over the link is all latency; the staged route's small copies are as good

// One more frame for the pending list (kmc_ctx::gather).  In-order semantics are kept: a frame that reads or writes a buffer a pending
// frame writes, or writes one a pending frame reads, makes the pending frames go out first; so does a frame of another coefficient tier
// (a launch runs ONE tier, and a frame's bits must not depend on its neighbours).  The list goes out when it is full (kGatherMax), or
// -- looked at for the first frame and then every fourth -- when the context's stream has run dry: a device that is idle is not kept
// waiting for a fuller list, a busy one gathers while it works (the launches clock themselves: lists only grow long behind a busy device).
int gather_push(kmc_ctx* c, const float* xyzi_in, float* xyzi_out, uint64_t n, const kmc_frame_params* params, kmc_stats* st) {
  const int tier = pick_tier(c, params, 1);
  const uintptr_t bytes = (uintptr_t)n * sizeof(v4f);
  const kmc_ctx::AoRange r = {(uintptr_t)xyzi_in, (uintptr_t)xyzi_in + bytes}, w = {(uintptr_t)xyzi_out, (uintptr_t)xyzi_out + bytes};
  uint32_t launches = 0;
  if (c->gl.must_flush_first(r, w, tier)) {  // the decisions: kmc_dispatch_book.hpp (unit-tested on the CPU); the runtime calls: here
    const int rc = fq_join(c);
    if (rc != KMC_OK) return rc;
    ++launches;
  }
  c->ao.invalidate();  // a pending frame is work the any-order window does not describe
  const auto verdict = c->gl.commit(r, w, tier);
  const uint32_t head = head_of(xyzi_out, KMC_MEM_DEVICE);
  ListRec& rec = c->gather[verdict.slot];
  std::memset(&rec, 0, sizeof(rec));
  fill_rec(*params, &rec.f);
  rec.f.pre2 = guard_pre2(*params);
  fill_recd(*params, &c->gather64[verdict.slot]);
  rec.in = (const v4f*)xyzi_in - head;
  rec.out = (v4f*)xyzi_out - head;
  rec.n = n + head;
  rec.head = head;
  bool go = verdict.issue_now;
  if (verdict.ask_stream) {
    bool may_query = true;
    if (c->stream != c->own_stream) {  // a caller's stream may be capturing a graph: a query is not allowed there (it would invalidate the capture)
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      may_query = hipStreamIsCapturing(c->stream, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone;
    }
    go = may_query && hipStreamQuery(c->stream) == hipSuccess;  // nothing in flight: issue what there is
    (void)hipGetLastError();                                    // (hipErrorNotReady is the expected answer of a busy stream)
  }
  if (go) {
    const int rc = fq_join(c);
    if (rc != KMC_OK) return rc;
    ++launches;
  }
  if (st) { st->n_points = n; st->variant = (uint32_t)tier; st->n_launches = launches; }
  return KMC_OK;
}
}  // namespace

extern "C" {

// ---- hot path: single frame, f32 -----------------------------------------------------------------
int kmc_hip_deskew_f32(kmc_ctx* c, const float* xyzi_in, float* xyzi_out, uint64_t n, const kmc_frame_params* params,
                       int mem_kind, kmc_stats* st) {
  if (!c) return KMC_ERR_INVALID_ARG;
  if (mem_kind != KMC_MEM_HOST && mem_kind != KMC_MEM_DEVICE && mem_kind != KMC_MEM_HOST_MAPPED) return KMC_ERR_INVALID_ARG;
  // host buffers that both lie in the page-locked pool take the in-place route by themselves
  if (mem_kind == KMC_MEM_HOST && n >= kMappedMinPoints && xyzi_in && xyzi_out && !((((uintptr_t)xyzi_in) | ((uintptr_t)xyzi_out)) & 15u) &&
      host_in_place_ok(xyzi_in, n * sizeof(v4f)) && host_in_place_ok(xyzi_out, n * sizeof(v4f)))
    mem_kind = KMC_MEM_HOST_MAPPED;
  if (mem_kind == KMC_MEM_HOST_MAPPED) {
    // page-locked, device-addressable host buffers: ONE kernel works on the caller's memory over the link -- persistent waves with the
    // next tile's load in flight while the current one is stored, so that upload and download overlap -- then a wait: the results are
    // in host memory when the call returns
    {
      const int rc_args = check_frame_args(xyzi_in, xyzi_out, n, params, KMC_MEM_DEVICE);
      if (rc_args != KMC_OK) return rc_args;
    }
    if (st) std::memset(st, 0, sizeof(*st));
    KMC_ENTER(c);
    const int tier = pick_tier(c, params, 1);
    FrameRec f;
    std::memset(&f, 0, sizeof(f));
    fill_rec(*params, &f);
    f.pre2 = guard_pre2(*params);
    FrameRecD d;
    fill_recd(*params, &d);
    if (st) { st->n_points = n; st->variant = (uint32_t)tier; st->n_launches = n ? 1 : 0; }
    if (n == 0) return KMC_OK;
    CallTimer tm(c);
    if (c->trace) { c->last_trace = kmc_call_trace{}; c->last_trace.issue_begin_us = trace_now_us(); }
    if (tm.begin_call() || tm.begin_kernel()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
    // persistent waves: the link, not the wave count, sets the rate -- beyond ~128 KiB of reads in flight more waves only queue up
    // behind each other (tools/link_probe: 128 waves 72 us per KITTI frame, 256 waves 82, one wave per tile 127)
    const int grid = (int)std::max<uint64_t>(1, std::min<uint64_t>((n + 63) / 64, (uint64_t)c->mapped_waves));
    const v4f* vin = (const v4f*)xyzi_in;
    v4f* vout = (v4f*)xyzi_out;
    const DoneWord dw = done_word_arm(c);
    with_tier(tier, [&](auto T) { launch_on(deskew_frame_streamed_f32<decltype(T)::value>, grid, 64, c->stream, false, vin, vout, n, f, dw, d); });
    KMC_HIP_TRY(c, hipGetLastError());
    if (tm.end_kernel()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
    if (c->trace) c->last_trace.issue_end_us = c->last_trace.wait_begin_us = trace_now_us();
    {
      const int rc_wait = wait_done_word(c);
      if (rc_wait != KMC_OK) return rc_wait;
    }
    if (c->trace) {
      c->last_trace.wait_end_us = trace_now_us();
      c->last_trace.dev_first_wave_us = (double)c->h_stamps[0] * 0.01;
      c->last_trace.dev_last_store_us = (double)c->h_stamps[1] * 0.01;
      c->last_trace.waves = (uint32_t)grid;
      c->last_trace.route = 1;
    }
    return tm.end_call(st);
  }
  {
    const int rc_args = check_frame_args(xyzi_in, xyzi_out, n, params, mem_kind);
    if (rc_args != KMC_OK) return rc_args;
  }
  if (st) std::memset(st, 0, sizeof(*st));
  // gathering on (kmc_hip_set_frame_queues(ctx, q > 1)) and no per-call timing: the frame joins the pending list, which goes out as ONE
  // launch of the frame-list kernel (gather_push) -- host work only unless the list goes out
  if (mem_kind == KMC_MEM_DEVICE && c->fq_count > 1 && !c->timing && n && n <= kmc_ctx::kGatherMaxPoints) return gather_push(c, xyzi_in, xyzi_out, n, params, st);
  // (a frame for an OPEN direct queue needs no HIP call at all unless it follows HIP-stream work: no hipSetDevice on that path)
  const bool direct_fast = mem_kind == KMC_MEM_DEVICE && c->dd && c->dd_wanted && !c->dd_broken && !c->stream_dirty && n && !c->timing && c->stream == c->own_stream && c->gl.count == 0;
  if (!direct_fast) KMC_HIP_TRY(c, hipSetDevice(c->device));
  CallTimer tm(c);
  if (mem_kind == KMC_MEM_DEVICE) {
    hipStream_t s = c->stream;
    bool any_order = false;
    if (c->gl.count || c->timing) {  // a list launch, event records: ordinary work on the stream, the any-order window ends
      const int rc_j = fq_join(c);
      if (rc_j != KMC_OK) return rc_j;
    }
    // The context's OWN stream: the frame goes out through the direct queue -- an AQL packet the library writes itself, below the HIP
    // runtime's launch path (kmc_capi_direct.hip: 2.1 us per call instead of 3.6-4.7 on KITTI-sized frames).  Order: behind whatever the
    // context put on its HIP stream before (a host wait, at such a transition only); among frames the barrier bit, unless the window
    // says the frame shares no buffer with the frames in flight.
    if (n && !c->timing && c->stream == c->own_stream && c->dd_wanted && !c->dd_broken && direct_open(c)) {
      if (c->stream_dirty) {
        KMC_HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->stream_dirty = false;
      }
      const uintptr_t bytes = (uintptr_t)n * sizeof(v4f);
      const kmc_ctx::AoRange r = {(uintptr_t)xyzi_in, (uintptr_t)xyzi_in + bytes}, w = {(uintptr_t)xyzi_out, (uintptr_t)xyzi_out + bytes};
      const kmc_book::LaneVerdict lane = c->lw.admit(r, w, c->dd_free_order, direct_frame_is_huge(n));
      const int tier = pick_tier(c, params, 1);
      FrameRec f;
      std::memset(&f, 0, sizeof(f));
      fill_rec(*params, &f);
      f.pre2 = guard_pre2(*params);
      FrameRecD d;
      fill_recd(*params, &d);
      uint32_t launches = 0;
      const int rc_direct = direct_frame(c, tier, (const v4f*)xyzi_in, (v4f*)xyzi_out, n, f, d, head_of(xyzi_out, KMC_MEM_DEVICE), lane, &launches);
      if (rc_direct != KMC_OK) return rc_direct;
      if (st) { st->n_points = n; st->variant = (uint32_t)tier; st->n_launches = launches; }
      return KMC_OK;
    }
    if (c->dd_pending) {  // (a HIP launch behind direct frames -- per-call timing was switched on, or the queue broke: wait for them first)
      const int rc_direct = direct_join(c);
      if (rc_direct != KMC_OK) return rc_direct;
    }
    c->stream_dirty = true;
    // in order on the context's stream -- but a frame that shares no buffer with the frames still in flight need not wait for them
    if (n) any_order = ao_admit(c, xyzi_in, xyzi_out, n * sizeof(v4f), false);
    if (tm.begin_call() || tm.begin_kernel()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
    int tier = 0;
    uint32_t launches = 0;
    const int rc_issue = issue_frame(c, s, xyzi_in, xyzi_out, n, params, &tier, any_order, &launches);
    if (rc_issue != KMC_OK) return rc_issue;
    if (st) { st->n_points = n; st->variant = (uint32_t)tier; st->n_launches = launches; }
    if (tm.end_kernel()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
    return tm.end_call(st);
  }
  {
    const int rc_j = fq_join(c);
    if (rc_j != KMC_OK) return rc_j;
  }
  const int tier = pick_tier(c, params, 1);
  FrameRec f;
  std::memset(&f, 0, sizeof(f));
  fill_rec(*params, &f);
  f.pre2 = guard_pre2(*params);
  FrameRecD d;
  fill_recd(*params, &d);
  if (st) { st->n_points = n; st->variant = (uint32_t)tier; }
  if (n == 0) return KMC_OK;
  // KMC_MEM_HOST: upload / compute / download on three streams over a ring of device slots, so that the H2D of chunk
  // k+1, the kernel of chunk k and the D2H of chunk k-1 run concurrently (PCIe is full duplex; DESIGN.md "host buffers")
  int rc = ensure_pipeline(c);
  if (rc != KMC_OK) return rc;
  KMC_HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (tm.begin_call() || tm.begin_kernel()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
  uint32_t launches = 0;
  for (uint64_t off = 0, k = 0; off < n; off += kHostChunkPoints, ++k) {
    const int b = (int)(k % kmc_ctx::kPipeSlots);
    const bool reused = k >= (uint64_t)kmc_ctx::kPipeSlots;
    const uint64_t m = std::min<uint64_t>(kHostChunkPoints, n - off);
    if (reused) KMC_HIP_TRY(c, hipStreamWaitEvent(c->pipe[0], c->ev_kernel[b], 0));  // the slot's input was consumed
    KMC_HIP_TRY(c, hipMemcpyAsync(c->d_stage_in[b], xyzi_in + 4 * off, m * sizeof(v4f), hipMemcpyHostToDevice, c->pipe[0]));
    KMC_HIP_TRY(c, hipEventRecord(c->ev_h2d[b], c->pipe[0]));
    KMC_HIP_TRY(c, hipStreamWaitEvent(c->pipe[1], c->ev_h2d[b], 0));
    if (reused) KMC_HIP_TRY(c, hipStreamWaitEvent(c->pipe[1], c->ev_d2h[b], 0));     // the slot's output was downloaded
    launch_frame(c->pipe[1], tier, (const v4f*)c->d_stage_in[b], (v4f*)c->d_stage_out[b], m, f, d);
    KMC_HIP_TRY(c, hipGetLastError());
    KMC_HIP_TRY(c, hipEventRecord(c->ev_kernel[b], c->pipe[1]));
    KMC_HIP_TRY(c, hipStreamWaitEvent(c->pipe[2], c->ev_kernel[b], 0));
    KMC_HIP_TRY(c, hipMemcpyAsync(xyzi_out + 4 * off, c->d_stage_out[b], m * sizeof(v4f), hipMemcpyDeviceToHost, c->pipe[2]));
    KMC_HIP_TRY(c, hipEventRecord(c->ev_d2h[b], c->pipe[2]));
    ++launches;
  }
  KMC_HIP_TRY(c, hipStreamSynchronize(c->pipe[2]));
  KMC_HIP_TRY(c, hipStreamSynchronize(c->pipe[1]));
  KMC_HIP_TRY(c, hipStreamSynchronize(c->pipe[0]));
  if (tm.end_kernel()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
  if (st) st->n_launches = launches;
  return tm.end_call(st);
}

// ---- hot path: a list of separate frames, ONE launch ------------------------------------------------
namespace {
// Do two DIFFERENT frames of the list touch the same memory with at least one of them writing it?  (in == out of one frame is fine.)
// Sorted sweep over the 2 F intervals; lists whose buffers come in ascending, disjoint order -- the usual case -- are recognised in O(F).
bool list_has_hazard(const float* const* in, float* const* out, const uint64_t* n_points, uint32_t n_frames) {
  struct Iv { uintptr_t lo, hi; uint32_t frame; bool write; };
  std::vector<Iv> iv;
  iv.reserve(2 * (size_t)n_frames);
  for (uint32_t f = 0; f < n_frames; ++f) {
    if (!n_points[f]) continue;
    const uintptr_t bytes = (uintptr_t)n_points[f] * sizeof(v4f);
    iv.push_back({(uintptr_t)in[f], (uintptr_t)in[f] + bytes, f, false});
    iv.push_back({(uintptr_t)out[f], (uintptr_t)out[f] + bytes, f, true});
  }
  if (!std::is_sorted(iv.begin(), iv.end(), [](const Iv& a, const Iv& b) { return a.lo < b.lo; }))
    std::sort(iv.begin(), iv.end(), [](const Iv& a, const Iv& b) { return a.lo < b.lo; });
  for (size_t i = 0; i < iv.size(); ++i)
    for (size_t j = i + 1; j < iv.size() && iv[j].lo < iv[i].hi; ++j)
      if (iv[i].frame != iv[j].frame && (iv[i].write || iv[j].write)) return true;
  return false;
}
}  // namespace

int kmc_hip_deskew_frames_f32(kmc_ctx* c, const float* const* xyzi_in, float* const* xyzi_out, const uint64_t* n_points,
                              const kmc_frame_params* params, uint32_t n_frames, kmc_stats* st) {
  if (!c || (n_frames && (!xyzi_in || !xyzi_out || !n_points || !params))) return KMC_ERR_INVALID_ARG;
  if (st) std::memset(st, 0, sizeof(*st));
  uint64_t total = 0, n_max = 0;
  for (uint32_t f = 0; f < n_frames; ++f) {
    const int rc_args = check_frame_args(xyzi_in[f], xyzi_out[f], n_points[f], &params[f], KMC_MEM_DEVICE);
    if (rc_args != KMC_OK) return rc_args;
    total += n_points[f];
    n_max = std::max<uint64_t>(n_max, n_points[f] + head_of(xyzi_out[f], KMC_MEM_DEVICE));
  }
  KMC_ENTER(c);
  // A frame's bits must not depend on its neighbours (gather_push keeps the same rule): every frame runs at ITS OWN
  // coefficient tier, so a list of mixed tiers goes out as one launch per tier present (at most four; a drive is one tier).
  std::vector<int> tiers(n_frames);
  int tier_max = 0;
  uint32_t tier_mask = 0;
  for (uint32_t f = 0; f < n_frames; ++f) {
    tiers[f] = pick_tier(c, &params[f], 1);
    if (n_points[f]) { tier_max = std::max(tier_max, tiers[f]); tier_mask |= 1u << tiers[f]; }
  }
  if (st) { st->n_points = total; st->variant = (uint32_t)tier_max; }
  if (total == 0) return KMC_OK;
  CallTimer tm(c);
  // Frames that depend on each other (one's output is another's input, or two write the same buffer) cannot share a launch: such a list
  // goes out frame by frame, in order, as ordinary launches on the context's stream -- what separate kmc_hip_deskew_f32 calls would do.
  // So does a list the 2-D grid cannot hold (more than 65 535 frames, or beyond 2^32 work-items).
  const uint64_t tiles_x = (n_max + 63) / 64;
  const bool fits = n_frames <= 65535u && tiles_x * 64 * (uint64_t)n_frames < (1ull << 32);
  if (!fits || list_has_hazard(xyzi_in, xyzi_out, n_points, n_frames)) {
    if (tm.begin_call() || tm.begin_kernel()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
    uint32_t launches = 0;
    for (uint32_t f = 0; f < n_frames; ++f) {
      uint32_t one = 0;
      const int rc = issue_frame(c, c->stream, xyzi_in[f], xyzi_out[f], n_points[f], &params[f], nullptr, false, &one);
      if (rc != KMC_OK) return rc;
      launches += one;  // (an empty frame launches nothing)
    }
    if (tm.end_kernel()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
    if (st) st->n_launches = launches;
    return tm.end_call(st);
  }
  std::vector<ListRec> recs;
  std::vector<FrameRecD> recd;
  recs.reserve(n_frames);
  recd.reserve(n_frames);
  if (tm.begin_call() || tm.begin_kernel()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
  uint32_t launches = 0;
  for (int tier = 0; tier <= kTrig; ++tier) {
    if (!(tier_mask & (1u << tier))) continue;
    recs.clear();
    recd.clear();
    for (uint32_t f = 0; f < n_frames; ++f) {
      if (tiers[f] != tier || !n_points[f]) continue;
      ListRec r;
      std::memset(&r, 0, sizeof(r));
      fill_rec(params[f], &r.f);
      r.f.pre2 = guard_pre2(params[f]);
      FrameRecD d;
      fill_recd(params[f], &d);
      const uint32_t head = head_of(xyzi_out[f], KMC_MEM_DEVICE);
      r.in = (const v4f*)xyzi_in[f] - head;
      r.out = (v4f*)xyzi_out[f] - head;
      r.n = n_points[f] + head;
      r.head = head;
      recs.push_back(r);
      recd.push_back(d);
    }
    uint32_t one = 0;
    const int rc_list = launch_list(c, recs.data(), recd.data(), (uint32_t)recs.size(), tier, &one);
    if (rc_list != KMC_OK) return rc_list;
    launches += one;
  }
  if (tm.end_kernel()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
  if (st) st->n_launches = launches;
  return tm.end_call(st);
}

// ---- hot path: batch of frames, f32 ---------------------------------------------------------------
int kmc_hip_deskew_batch_f32(kmc_ctx* c, const float* xyzi_in, float* xyzi_out, const uint64_t* offsets, uint32_t n_frames,
                             const kmc_frame_params* params, uint32_t* frame_idx_out, int mem_kind, kmc_stats* st) {
  if (!c || !offsets || (n_frames && !params)) return KMC_ERR_INVALID_ARG;
  // (a page-locked buffer that is only 4-byte aligned stays on the staged route, which copies it: the kernels' 16-byte accesses need
  // 16-byte alignment, the copies do not -- ADVICE r03)
  if (mem_kind == KMC_MEM_HOST && n_frames && xyzi_in && xyzi_out && offsets[n_frames] >= kMappedMinPoints &&
      !((((uintptr_t)xyzi_in) | ((uintptr_t)xyzi_out)) & 15u) && (!frame_idx_out || !(((uintptr_t)frame_idx_out) & 3u)) &&
      host_in_place_ok(xyzi_in, offsets[n_frames] * sizeof(v4f)) && host_in_place_ok(xyzi_out, offsets[n_frames] * sizeof(v4f)) &&
      (!frame_idx_out || host_in_place_ok(frame_idx_out, offsets[n_frames] * sizeof(uint32_t))))
    mem_kind = KMC_MEM_HOST_MAPPED;  // the caller's buffers are page-locked: no staging copies
  if (mem_kind == KMC_MEM_HOST_MAPPED) {
    // page-locked, device-addressable host buffers: the device route on the caller's pointers (the tiles of a batch are spread over
    // thousands of waves in flight, reads and writes mix by themselves), then a wait -- the results are in host memory on return
    const int rc_dev = kmc_hip_deskew_batch_f32(c, xyzi_in, xyzi_out, offsets, n_frames, params, frame_idx_out, KMC_MEM_DEVICE, st);
    if (rc_dev != KMC_OK) return rc_dev;
    KMC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return KMC_OK;
  }
  if (mem_kind != KMC_MEM_HOST && mem_kind != KMC_MEM_DEVICE) return KMC_ERR_INVALID_ARG;
  if (st) std::memset(st, 0, sizeof(*st));
  if (offsets[0] != 0) return KMC_ERR_INVALID_ARG;
  for (uint32_t f = 0; f < n_frames; ++f) {
    if (offsets[f + 1] < offsets[f]) return KMC_ERR_INVALID_ARG;
    if (!params_ok(&params[f])) return KMC_ERR_INVALID_ARG;
    if (!(params[f].x_req >= 0.0 && params[f].x_req <= 1.0)) return KMC_ERR_TIME_OUT_OF_RANGE;
  }
  const uint64_t n = n_frames ? offsets[n_frames] : 0;
  if (n && (!xyzi_in || !xyzi_out)) return KMC_ERR_INVALID_ARG;
  // 16-byte alignment is a requirement of the kernels' vector accesses: device pointers only (host buffers are copied)
  if (((uintptr_t)xyzi_in | (uintptr_t)xyzi_out) & (mem_kind == KMC_MEM_DEVICE ? 15u : 3u)) return KMC_ERR_INVALID_ARG;
  // Batched launches always go to the context's stream, after a join: routing them over the frame queues was measured and
  // dropped -- their table uploads on the side stream end up sharing hardware queues with the launches (a 13 M-point drive:
  // 64 us in order, 74 us over two queues, 188 us over four; tools/measure_configs.py, profiles/r02_measure_configs.json).
  KMC_ENTER(c);
  const int tier = pick_tier(c, params, n_frames);
  if (st) { st->n_points = n; st->variant = (uint32_t)tier; }
  if (n == 0) return KMC_OK;

  const uint32_t head = head_of(xyzi_out, mem_kind);  // dead points in front: tiles are cut on 1 KiB lines of the output
  const uint64_t nv = n + head;                       // virtual size; every offset below is shifted by `head` too

  // Small batches of device-resident frames: the tables travel in the kernel arguments.  Nothing to upload, nothing for the host to
  // wait for -- the call only enqueues one launch on the context's stream (and can therefore be captured into a HIP graph).
  if (mem_kind == KMC_MEM_DEVICE && n_frames <= (uint32_t)kInlineBatchFrames) {
    uint32_t shift = kChunkShift;
    while (((nv + (1ull << shift) - 1) >> shift) > (uint64_t)kInlineBatchChunks) ++shift;
    if (shift <= 31) {
      BatchInline inl;
      std::memset(&inl, 0, sizeof(inl));
      for (uint32_t f = 0; f < n_frames; ++f) {
        fill_rec(params[f], &inl.recs[f]);
        fill_recd(params[f], &inl.recs64[f]);
        inl.pre2[f] = guard_pre2(params[f]);
        inl.recs[f].end_lo = (uint32_t)((offsets[f + 1] + head) & 0xFFFFFFFFull);
        inl.recs[f].end_hi = (uint32_t)((offsets[f + 1] + head) >> 32);
      }
      build_coarse(offsets, n_frames, nv, head, inl.coarse, shift);
      CallTimer tmi(c);
      if (tmi.begin_call() || tmi.begin_kernel()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
      const v4f* vin = (const v4f*)xyzi_in - head;
      v4f* vout = (v4f*)xyzi_out - head;
      uint32_t* vidx = frame_idx_out ? frame_idx_out - head : nullptr;
      const uint32_t launches = launch_batch<true>(tier, c->stream, vin, vout, nullptr, nullptr, n_frames, nv, vidx, head, nullptr, shift, nullptr, inl);
      KMC_HIP_TRY(c, hipGetLastError());
      if (tmi.end_kernel()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
      if (st) st->n_launches = launches;
      return tmi.end_call(st);
    }
  }
  // A table upload cannot be part of a stream capture (the slot is reused by later calls, and the host waits for the copy)
  {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(c->stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) {
      c->last_error = "kmc_hip_deskew_batch_f32: only batches of at most 16 device-resident frames can be captured into a HIP graph";
      return KMC_ERR_INVALID_ARG;
    }
    (void)hipGetLastError();
  }
  const uint64_t chunk = 1ull << kChunkShift;
  const uint64_t n_chunks = (nv + chunk - 1) / chunk;
  const uint64_t n_coarse = n_chunks + 1;

  // slot layout: [BatchRec x F | FrameRecD x F (f64 twins for the near-origin guard) | coarse x (n_chunks + 1) | pre2 x F (guard thresholds)]
  const size_t recs_bytes = ((size_t)n_frames * sizeof(BatchRec) + 255) & ~(size_t)255;
  const size_t recd_bytes = ((size_t)n_frames * sizeof(FrameRecD) + 255) & ~(size_t)255;
  const size_t coarse_bytes = ((size_t)n_coarse * sizeof(uint2) + 255) & ~(size_t)255;
  const size_t need = recs_bytes + recd_bytes + coarse_bytes + (size_t)n_frames * sizeof(float);
  int slot_id = 0;
  {
    const int rc_slot = slot_begin(c, need, &slot_id);
    if (rc_slot != KMC_OK) return rc_slot;
  }
  kmc_ctx::TableSlot& sl = c->slots[slot_id];
  BatchRec* h_recs = reinterpret_cast<BatchRec*>(sl.h_buf);
  FrameRecD* h_recd = reinterpret_cast<FrameRecD*>(sl.h_buf + recs_bytes);
  uint2* h_coarse = reinterpret_cast<uint2*>(sl.h_buf + recs_bytes + recd_bytes);
  const BatchRec* d_recs = reinterpret_cast<const BatchRec*>(sl.d_buf);
  const FrameRecD* d_recd = reinterpret_cast<const FrameRecD*>(sl.d_buf + recs_bytes);
  const uint2* d_coarse = reinterpret_cast<const uint2*>(sl.d_buf + recs_bytes + recd_bytes);
  float* h_pre2 = reinterpret_cast<float*>(sl.h_buf + recs_bytes + recd_bytes + coarse_bytes);
  const float* d_pre2 = reinterpret_cast<const float*>(sl.d_buf + recs_bytes + recd_bytes + coarse_bytes);
  for (uint32_t f = 0; f < n_frames; ++f) {
    BatchRec* r = &h_recs[f];
    fill_rec(params[f], r);
    fill_recd(params[f], &h_recd[f]);
    h_pre2[f] = guard_pre2(params[f]);
    r->end_lo = (uint32_t)((offsets[f + 1] + head) & 0xFFFFFFFFull);
    r->end_hi = (uint32_t)((offsets[f + 1] + head) >> 32);
  }
  build_coarse(offsets, n_frames, nv, head, h_coarse);
  // one table upload on the side stream (overlaps whatever the compute stream is still running), awaited on the host
  {
    const int rc_up = slot_upload(c, slot_id, need);
    if (rc_up != KMC_OK) return rc_up;
  }

  CallTimer tm(c);
  const v4f* d_in = (const v4f*)xyzi_in;
  v4f* d_out = (v4f*)xyzi_out;
  uint32_t* d_idx = frame_idx_out;
  if (mem_kind == KMC_MEM_HOST) {
    const size_t pts = n * sizeof(v4f);
    const size_t idx_bytes = frame_idx_out ? n * sizeof(uint32_t) : 0;
    int rc = ensure_tmp(c, 2 * pts + idx_bytes);
    if (rc != KMC_OK) return rc;
    d_in = (const v4f*)c->d_tmp;
    d_out = (v4f*)((char*)c->d_tmp + pts);
    d_idx = frame_idx_out ? (uint32_t*)((char*)c->d_tmp + 2 * pts) : nullptr;
  }
  if (tm.begin_call()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
  if (mem_kind == KMC_MEM_HOST)
    KMC_HIP_TRY(c, hipMemcpyAsync((void*)d_in, xyzi_in, n * sizeof(v4f), hipMemcpyHostToDevice, c->stream));
  if (tm.begin_kernel()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
  uint32_t* v_idx = d_idx ? d_idx - head : nullptr;
  const uint32_t launches = launch_batch<false>(tier, c->stream, d_in - head, d_out - head, d_recs, d_coarse, n_frames, nv, v_idx, head, d_recd, (uint32_t)kChunkShift, d_pre2, BatchNoInline{});
  KMC_HIP_TRY(c, hipGetLastError());
  {
    const int rc_end = slot_end(c, slot_id);
    if (rc_end != KMC_OK) return rc_end;
  }
  if (tm.end_kernel()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
  if (mem_kind == KMC_MEM_HOST) {
    KMC_HIP_TRY(c, hipMemcpyAsync(xyzi_out, d_out, n * sizeof(v4f), hipMemcpyDeviceToHost, c->stream));
    if (frame_idx_out) KMC_HIP_TRY(c, hipMemcpyAsync(frame_idx_out, d_idx, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    KMC_HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  if (st) st->n_launches = launches;
  return tm.end_call(st);
}

}  // extern "C"

// kmc_capi_core.hip -- context life cycle, streams, timers, pinned allocations, the f64 host pre-step entry points and the
// definitions of the helpers every other translation unit of libkmc_hip.so shares (kmc_internal.hip.h).
#include "kmc_internal.hip.h"

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <mutex>

namespace kmc_impl {

int pick_tier(const kmc_ctx* c, const kmc_frame_params* p, uint32_t n) {
  if (c->force_tier >= 0 && c->force_tier <= kTrig) return c->force_tier;
  double theta_max = 0.0;
  for (uint32_t i = 0; i < n; ++i) {
    const double* f = p[i].twist;
    const double phi = std::sqrt(f[3] * f[3] + f[4] * f[4] + f[5] * f[5]);
    const double smax = std::fmax(std::fabs(p[i].x_req), std::fabs(1.0 - p[i].x_req));  // frac in [0,1]
    theta_max = std::fmax(theta_max, phi * smax);
  }
  return tier_of_theta(theta_max);
}

int ensure_tmp(kmc_ctx* c, size_t bytes) {
  if (bytes <= c->tmp_cap) return KMC_OK;
  if (c->d_tmp) {
    {
      const int rc_join = fq_join(c);
      if (rc_join != KMC_OK) return rc_join;
    }
    KMC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    KMC_HIP_TRY(c, hipFree(c->d_tmp));
    c->d_tmp = nullptr;
    c->tmp_cap = 0;
  }
  KMC_HIP_TRY(c, hipMalloc(&c->d_tmp, bytes));
  c->tmp_cap = bytes;
  return KMC_OK;
}

int ensure_pipe_streams(kmc_ctx* c) {
  for (int b = 0; b < 3; ++b)
    if (!c->pipe[b]) KMC_HIP_TRY(c, hipStreamCreateWithFlags(&c->pipe[b], hipStreamNonBlocking));
  return KMC_OK;
}

int ensure_events(kmc_ctx* c, size_t count) {
  while (c->ev_pool.size() < count) {
    hipEvent_t ev = nullptr;
    KMC_HIP_TRY(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    c->ev_pool.push_back(ev);
  }
  return KMC_OK;
}

double trace_now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

DoneWord done_word_arm(kmc_ctx* c) {
  DoneWord dw;
  dw.ticket = c->d_ticket;
  dw.word = c->h_done;
  dw.stamps = c->trace ? c->h_stamps : nullptr;
  dw.seq = ++c->done_seq;
  if (dw.seq == 0) dw.seq = ++c->done_seq;  // (0 is the word's initial value)
  dw.pad = 0;
  c->done_armed = true;
  return dw;
}

// The last armed launch has stored its sequence number <=> every wave's stores are in host memory (DoneWord, kmc_kernels.hip.h).
// The stream is looked at every 16 Ki polls (~100 us): a launch that died never raises the word.
// A stream that has run dry WITHOUT the word (round 5: a fresh context's first armed kernel counted from a stale ticket -- see kmc_hip_create): the
// kernel has finished, so the ordinary HIP contract takes over -- hipStreamSynchronize, after which every store of the stream's
// kernels is in host memory whether or not the word came.  The event is counted and its state kept for
// kmc_hip_completion_word_fallbacks (expected sequence number, word, ticket), and the ticket is put back to 0 so that the next
// armed launch counts from a clean slate.
static std::atomic<uint64_t> g_done_fallbacks{0};        // the same, over every context of the process (ctx == NULL asks for these)
static std::atomic<uint32_t> g_done_fallback_state[3];

int wait_done_word(kmc_ctx* c) {
  if (!c->done_armed) return KMC_OK;
  c->done_armed = false;
  const uint32_t seq = c->done_seq;
  volatile uint32_t* const word = c->h_done;
  for (uint64_t spin = 1; *word != seq; ++spin) {
    if ((spin & 0x3FFFu) != 0) continue;
    const hipError_t q = hipStreamQuery(c->stream);
    if (q == hipErrorNotReady) { (void)hipGetLastError(); continue; }
    if (q != hipSuccess) return fail_hip(c, q, "in-place kernel (hipStreamQuery while waiting for its completion word)");
    if (*word == seq) break;  // the stream is idle and the word is there
    KMC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    const uint32_t seen = *word;
    uint32_t ticket = 0;
    KMC_HIP_TRY(c, hipMemcpy(&ticket, c->d_ticket, sizeof(ticket), hipMemcpyDeviceToHost));
    c->done_fallbacks += 1;
    c->done_fallback_state[0] = seq;
    c->done_fallback_state[1] = seen;
    c->done_fallback_state[2] = ticket;
    g_done_fallbacks.fetch_add(1, std::memory_order_relaxed);
    for (int i = 0; i < 3; ++i) g_done_fallback_state[i].store(c->done_fallback_state[i], std::memory_order_relaxed);
    if (ticket != 0) {  // (on the stream the kernels run on, and waited for: hipMemset alone returns before the fill has run)
      KMC_HIP_TRY(c, hipMemsetAsync(c->d_ticket, 0, sizeof(ticket), c->stream));
      KMC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    break;
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return KMC_OK;
}

int ensure_pipeline(kmc_ctx* c) {
  if (c->stage_cap) return KMC_OK;
  const size_t bytes = kHostChunkPoints * sizeof(v4f);
  {
    const int rc_streams = ensure_pipe_streams(c);
    if (rc_streams != KMC_OK) return rc_streams;
  }
  for (int b = 0; b < kmc_ctx::kPipeSlots; ++b) {
    KMC_HIP_TRY(c, hipMalloc(&c->d_stage_in[b], bytes));
    KMC_HIP_TRY(c, hipMalloc(&c->d_stage_out[b], bytes));
    KMC_HIP_TRY(c, hipEventCreateWithFlags(&c->ev_h2d[b], hipEventDisableTiming));
    KMC_HIP_TRY(c, hipEventCreateWithFlags(&c->ev_kernel[b], hipEventDisableTiming));
    KMC_HIP_TRY(c, hipEventCreateWithFlags(&c->ev_d2h[b], hipEventDisableTiming));
  }
  c->stage_cap = bytes;
  return KMC_OK;
}

int launch_list(kmc_ctx* c, const ListRec* recs, const FrameRecD* recd, uint32_t count, int tier, uint32_t* launches_out, bool inline_only) {
  if (launches_out) *launches_out = 0;
  if (count == 0) return KMC_OK;
  auto tiles_of = [&](uint32_t first, uint32_t n_recs) {
    uint64_t n_max = 0;
    for (uint32_t k = 0; k < n_recs; ++k) n_max = std::max<uint64_t>(n_max, recs[first + k].n);
    return (uint32_t)((n_max + kTile - 1) / kTile);
  };
  // one launch whose records travel in the kernel arguments: the smallest of the three block capacities that holds n_recs
  auto launch_inline_cap = [&](auto CAP, uint32_t first, uint32_t n_recs, bool any_order) {
    constexpr int kCap = decltype(CAP)::value;
    ListInlineT<kCap> inl;
    if (n_recs < (uint32_t)kCap) std::memset(&inl, 0, sizeof(inl));  // (no stale stack bytes in the kernel arguments)
    std::memcpy(inl.recs, recs + first, n_recs * sizeof(ListRec));
    std::memcpy(inl.recs64, recd + first, n_recs * sizeof(FrameRecD));
    const dim3 grid(std::max(1u, tiles_of(first, n_recs)), n_recs, 1);
    with_tier(tier, [&](auto T) {
      if (any_order)
        hipExtLaunchKernelGGL((deskew_list_f32<decltype(T)::value, kCap>), grid, dim3(kTile), 0, c->stream, nullptr, nullptr, (uint32_t)hipExtAnyOrderLaunch, inl);
      else
        hipLaunchKernelGGL((deskew_list_f32<decltype(T)::value, kCap>), grid, dim3(kTile), 0, c->stream, inl);
    });
  };
  auto launch_inline = [&](uint32_t first, uint32_t n_recs, bool any_order) {
    if (n_recs <= 16) launch_inline_cap(std::integral_constant<int, 16>{}, first, n_recs, any_order);
    else if (n_recs <= 64) launch_inline_cap(std::integral_constant<int, 64>{}, first, n_recs, any_order);
    else launch_inline_cap(std::integral_constant<int, kInlineListFramesMax>{}, first, n_recs, any_order);
  };
  uint32_t launches = 0;
  bool capturing = false;
  if (count > (uint32_t)kInlineListFrames) {  // (a captured launch carries the 16-frame block every runtime is known to take)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    capturing = hipStreamIsCapturing(c->stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    (void)hipGetLastError();
  }
  // Every list goes out in kernel-argument launches: ONE launch for up to 256 frames; longer lists in launches of 256 (the
  // frames of a list are independent of each other -- the callers have checked -- so where barrier-free dispatch is verified and the
  // stream is the context's own, every launch after the first goes out without the barrier bit: the ordinary first launch orders
  // the chain behind everything before it, the next ordinary packet on the stream waits for all of it).  Launches of at most 16 frames
  // -- the block every runtime is known to take -- under stream capture, for fq_join's fallback (inline_only) and once a large block
  // has been refused (kmc_ctx::big_kernargs).  (The round-4 route -- one launch over an uploaded device table -- lost every measurement
  // against these blocks and is gone: profiles/r05_frame_stream.json.)
  const bool small_blocks = capturing || inline_only || !c->big_kernargs;
  const uint32_t per_launch = small_blocks ? (uint32_t)kInlineListFrames : (uint32_t)kInlineListFramesMax;
  if (count > per_launch && !capturing && c->stream == c->own_stream) ao_ensure(c);
  const bool free_order = count > per_launch && !capturing && c->ao_enabled && c->stream == c->own_stream && c->stream != nullptr;
  for (uint32_t first = 0; first < count; first += per_launch) {
    const uint32_t n_recs = std::min<uint32_t>(per_launch, count - first);
    launch_inline(first, n_recs, free_order && first != 0);
    hipError_t le = hipGetLastError();
    if (le == hipSuccess) {
      ++launches;
      continue;
    }
    if (small_blocks || n_recs <= (uint32_t)kInlineListFrames) KMC_HIP_TRY(c, le);
    // This runtime refused a block beyond 4 KiB: remember it, and issue THIS launch's frames and the rest of the list in 16-frame
    // launches.  A refused launch ran nothing and the launches before it are not repeated -- a frame with in == out must not be
    // deskewed twice.
    c->big_kernargs = false;
    for (uint32_t f2 = first; f2 < count; f2 += kInlineListFrames, ++launches) {
      launch_inline(f2, std::min<uint32_t>(kInlineListFrames, count - f2), false);
      KMC_HIP_TRY(c, hipGetLastError());
    }
    break;
  }
  if (launches_out) *launches_out = launches;
  return KMC_OK;
}

int fq_join(kmc_ctx* c) {
  c->ao.invalidate();  // whoever joins is about to put ordinary work on the stream: the any-order window ends here
  if (c->dd_pending) {  // ... and that work is ordered behind the frames in the direct queue: wait for them (kmc_capi_direct.hip)
    const int rc_direct = direct_join(c);
    if (rc_direct != KMC_OK) return rc_direct;
  }
  if (c->gl.count == 0) return KMC_OK;
  KMC_HIP_TRY(c, hipSetDevice(c->device));
  const uint32_t count = c->gl.count;
  c->gl.flushed();
  c->stream_dirty = true;  // the list launch goes to the HIP stream
  int rc = launch_list(c, c->gather, c->gather64, count, c->gl.tier, nullptr);
  if (rc != KMC_OK) {
    // The launch failed.  The calls that queued these frames have already returned KMC_OK, so the frames must not be dropped on the floor
    // (ADVICE r04): the records are still in c->gather -- issue them once more as launches of at most 16 frames, the block every runtime
    // takes.  If even that fails the error is STICKY: it is what this join returns, and what kmc_hip_frame_queue_join /
    // kmc_hip_synchronize keep returning until the caller has seen it once.
    (void)hipGetLastError();
    rc = launch_list(c, c->gather, c->gather64, count, c->gl.tier, nullptr, /*inline_only*/ true);
    if (rc != KMC_OK) {
      c->fq_error = rc;
      c->fq_dropped += count;
    }
  }
  return rc;
}

// the sticky error of a join that lost frames: reported once, by the next call whose job is to say "everything queued has been issued"
int fq_take_error(kmc_ctx* c) {
  const int rc = c->fq_error;
  c->fq_error = KMC_OK;
  return rc;
}

// May this frame start before the launches ahead of it on the context's stream have finished?  Yes if
//   - nothing the library does not know about can sit between the previous frame launch and this one: the stream is the context's
//     own, or the caller has said that its frames are produced before the first call (kmc_hip_set_frame_queue_order(ctx, 0)), or both
//     frames belong to one kmc_hip_deskew_frames_f32 call (`same_call`);
//   - the window describes everything in flight behind the last ordered launch (every other entry point ends it);
//   - the frame's buffers overlap none of the window's (write against reads and writes, read against writes).
// Either way the frame is entered into the window; an ordered launch starts a new one.
int ao_verdict_for(kmc_ctx* c);
// The probe runs when its verdict is first needed (round 5: kmc_hip_create used to run it -- allocations, kernels, waits -- for
// contexts that never launch a single frame, e.g. the run driver's).  Never while a caller's stream captures a graph: the probe
// allocates, which a capture in progress does not survive; such a call is simply dispatched in order and the probe waits for the next.
void ao_ensure(kmc_ctx* c) {
  if (c->ao_probed) return;
  if (c->stream != c->own_stream && c->stream != nullptr) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(c->stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
      (void)hipGetLastError();
      return;
    }
  }
  c->ao_probed = true;
  c->ao_verdict = ao_verdict_for(c);  // barrier-free dispatch only where this device and runtime were SEEN to honour what it relies on
  c->ao_enabled = c->ao_verdict == 1;
}

bool ao_admit(kmc_ctx* c, const void* in, const void* out, uint64_t bytes, bool same_call) {
  const kmc_ctx::AoRange r = {(uintptr_t)in, (uintptr_t)in + bytes}, w = {(uintptr_t)out, (uintptr_t)out + bytes};
  if (c->stream != nullptr && (same_call || c->stream == c->own_stream || !c->fq_ordered)) ao_ensure(c);  // (only where the verdict can matter)
  const bool stream_ok = c->stream != nullptr && (same_call || c->stream == c->own_stream || !c->fq_ordered);
  bool any_order = c->ao.admit(r, w, c->ao_enabled, stream_ok);  // the rule itself: kmc_dispatch_book.hpp (unit-tested on the CPU)
  if (any_order && c->stream != c->own_stream) {  // a caller's stream may be capturing a graph: plain launches there
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(c->stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
      (void)hipGetLastError();
      c->ao.demote_last(true);
      any_order = false;
    }
  }
  return any_order;
}

// ---- barrier-free dispatch: checked at run time before it is used ------------------------------------
// hipExtAnyOrderLaunch is documented "not supported on GFX9xx"; on MI355X / ROCm 7.2 it does clear the AQL barrier bit and nothing else
// (profiles/r03_anyorder_aql_headers.txt).  What that buys is NOT two kernels of one queue running side by side (two single-workgroup
// 10 ms kernels still take 20 ms, profiles/r03_anyorder_probe.json): the packet processor stops waiting for the previous packet's
// completion and cache release before it starts the next one, so a barrier-free packet may read memory its predecessor has not
// released yet -- which is why only frames that share no buffer with a frame in flight go out this way.  The library's CONSUMERS rely on
// one fact, and it is verified on the device the context is about to use, once per device and process (~0.3 ms), before ao_enabled
// may become true:
//   an ORDINARY packet behind barrier-free packets -- a kernel launch, a device-to-host copy, an event -- waits for every packet
//   before it AND sees everything they stored, from every XCD's L2.
// Eight rounds of: A (ordinary) and B, B2 (barrier-free) each fill their own 4 MiB region with a pattern through plain stores from
// 4096 workgroups (all XCDs); C (ordinary) re-reads the three regions through a reversed workgroup -> XCD mapping and counts what it
// does not find; a D2H copy of a barrier-free packet's last KiB and an event follow; the host checks the copy right after the event.
// Every other round the copy follows a barrier-free packet DIRECTLY (no ordinary kernel in between): that packet, B3, fills a FOURTH
// region that C never reads -- a barrier-free packet behind C need not wait for C, so it must not rewrite what C is still verifying
// (the first version let B3 rewrite B2's region and could fail the probe on its own race).  Any miss: one more attempt
// (a -1 is cached for the whole process, so a transient must not decide it), then feature OFF.
// Verdict (kmc_device_info.any_order_dispatch): 1 = verified, on;  0 = switched off (KMC_ANY_ORDER=0);  -1 = an ordinary packet
// overtook, or did not see, a barrier-free one;  -3 = the probe could not run (a HIP error).
constexpr uint32_t kAoWords = 1u << 20;  // 4 MiB per region
__global__ __launch_bounds__(256) void ao_probe_fill(uint32_t* x, uint32_t seed) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  x[i] = seed + i * 2654435761u;
}
__global__ __launch_bounds__(256) void ao_probe_verify(const uint32_t* x, uint32_t seed_a, uint32_t seed_b, uint32_t seed_c, unsigned long long* bad) {
  const uint32_t b = gridDim.x - 1 - blockIdx.x;  // reversed, and shifted by 3 workgroups: another XCD than the writer's
  const uint32_t i = ((b + 3) % gridDim.x) * 256 + threadIdx.x;
  const uint32_t miss = (x[i] != seed_a + i * 2654435761u) + (x[kAoWords + i] != seed_b + i * 2654435761u) + (x[2 * kAoWords + i] != seed_c + i * 2654435761u);
  if (miss) atomicAdd(bad, (unsigned long long)miss);
}

static int ao_probe_run(kmc_ctx* c) {
  uint32_t* d = nullptr;
  unsigned long long* d_bad = nullptr;
  uint32_t* h = nullptr;  // the D2H copy's target: B2's last 256 words
  int verdict = -3;
  hipStream_t s = c->own_stream;
  hipEvent_t ev = nullptr;
  const dim3 grid(kAoWords / 256), block(256);
  do {
    if (hipMalloc((void**)&d, 4 * (size_t)kAoWords * 4) != hipSuccess) break;
    if (hipMalloc((void**)&d_bad, 8) != hipSuccess) break;
    if (hipHostMalloc((void**)&h, 1024, hipHostMallocDefault) != hipSuccess) break;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) break;
    if (hipMemsetAsync(d_bad, 0, 8, s) != hipSuccess) break;
    bool host_ok = true, ran = true;
    for (uint32_t round = 0; round < 8 && ran; ++round) {
      const uint32_t sa = 977u * (3 * round + 1), sb = 977u * (3 * round + 2), sc = 977u * (3 * round + 3);
      hipLaunchKernelGGL(ao_probe_fill, grid, block, 0, s, d, sa);                                                               // A: ordinary
      hipExtLaunchKernelGGL(ao_probe_fill, grid, block, 0, s, nullptr, nullptr, (uint32_t)hipExtAnyOrderLaunch, d + kAoWords, sb);      // B
      hipExtLaunchKernelGGL(ao_probe_fill, grid, block, 0, s, nullptr, nullptr, (uint32_t)hipExtAnyOrderLaunch, d + 2 * kAoWords, sc);  // B2
      hipLaunchKernelGGL(ao_probe_verify, grid, block, 0, s, (const uint32_t*)d, sa, sb, sc, d_bad);                             // C: ordinary
      ran = hipGetLastError() == hipSuccess;
      const bool direct = (round & 1) != 0;  // the copy follows a barrier-free packet directly (no ordinary kernel in between)
      if (direct) {  // B3: its own region -- C, still running, reads the other three
        hipExtLaunchKernelGGL(ao_probe_fill, grid, block, 0, s, nullptr, nullptr, (uint32_t)hipExtAnyOrderLaunch, d + 3 * kAoWords, sc + 1);
        ran = ran && hipGetLastError() == hipSuccess;
      }
      ran = ran && hipMemcpyAsync(h, d + (direct ? 4 : 3) * (size_t)kAoWords - 256, 1024, hipMemcpyDeviceToHost, s) == hipSuccess;
      ran = ran && hipEventRecord(ev, s) == hipSuccess && hipEventSynchronize(ev) == hipSuccess;
      const uint32_t want = direct ? sc + 1 : sc;
      for (uint32_t k = 0; ran && k < 256; ++k) host_ok = host_ok && h[k] == want + (kAoWords - 256 + k) * 2654435761u;
    }
    if (!ran) break;
    unsigned long long bad = ~0ull;
    if (hipStreamSynchronize(s) != hipSuccess) break;
    if (hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost) != hipSuccess) break;
    verdict = (bad == 0 && host_ok) ? 1 : -1;
  } while (false);
  (void)hipGetLastError();
  if (ev) (void)hipEventDestroy(ev);
  if (h) (void)hipHostFree(h);
  if (d_bad) (void)hipFree(d_bad);
  if (d) (void)hipFree(d);
  return verdict;
}

// one probe per device and process; KMC_ANY_ORDER=0 switches the feature off without probing
int ao_verdict_for(kmc_ctx* c) {
  if (const char* e = std::getenv("KMC_ANY_ORDER"))
    if (std::atoi(e) == 0) return 0;
  static std::mutex mu;
  static int cached[64];
  static bool have[64] = {};
  std::lock_guard<std::mutex> lock(mu);
  const int d = c->device;
  if (d >= 0 && d < 64 && have[d]) return cached[d];
  int v = ao_probe_run(c);
  if (v != 1) v = ao_probe_run(c);  // once more before a negative verdict is cached for the life of the process
  if (d >= 0 && d < 64) { cached[d] = v; have[d] = true; }
  return v;
}

int slot_begin(kmc_ctx* c, size_t need, int* slot_id_out) {
  const int slot_id = c->next_slot;
  const int group_id = slot_id / kmc_ctx::kSlotsPerGroup;
  c->next_slot = (c->next_slot + 1) % kmc_ctx::kTableSlots;
  if (slot_id % kmc_ctx::kSlotsPerGroup == 0 && c->group_busy[group_id]) {
    KMC_HIP_TRY(c, hipEventSynchronize(c->group_consumed[group_id]));
    c->group_busy[group_id] = false;
  } else if (slot_id % kmc_ctx::kSlotsPerGroup == 0 && c->group_dirty[group_id]) {
    // the previous lap through this group ended without a marker (a call failed between slot_begin and slot_end): its
    // launches may still read the tables -- drain instead of guessing
    const int rc_join = fq_join(c);
    if (rc_join != KMC_OK) return rc_join;
    KMC_HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  c->group_dirty[group_id] = true;
  if (need > c->slots[slot_id].cap) {
    // grow EVERY slot at once (so that steady state never allocates again); slots may still be referenced by kernels in
    // flight: drain first
    {
      const int rc_join = fq_join(c);
      if (rc_join != KMC_OK) return rc_join;
    }
    KMC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->copy_stream) KMC_HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
    const size_t cap = std::max<size_t>(64 * 1024, need * 2);
    for (auto& each : c->slots) {
      if (each.cap >= cap) continue;
      if (each.d_buf) (void)hipFree(each.d_buf);
      if (each.h_buf) (void)hipHostFree(each.h_buf);
      each.d_buf = nullptr; each.h_buf = nullptr; each.cap = 0;
      KMC_HIP_TRY(c, hipMalloc((void**)&each.d_buf, cap));
      KMC_HIP_TRY(c, hipHostMalloc((void**)&each.h_buf, cap, hipHostMallocDefault));
      each.cap = cap;
    }
    for (auto& busy : c->group_busy) busy = false;
    for (auto& dirty : c->group_dirty) dirty = false;
    c->group_dirty[group_id] = true;
  }
  *slot_id_out = slot_id;
  return KMC_OK;
}

int slot_upload(kmc_ctx* c, int slot_id, size_t bytes) {
  kmc_ctx::TableSlot& sl = c->slots[slot_id];
  // the side stream exists from the first table upload on (a stream is an HSA queue: ~4 ms of kmc_hip_create that a context which
  // never uploads a table -- the run driver's, the drop-in's -- does not have to pay)
  if (!c->copy_stream) KMC_HIP_TRY(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  KMC_HIP_TRY(c, hipMemcpyAsync(sl.d_buf, sl.h_buf, bytes, hipMemcpyHostToDevice, c->copy_stream));
  KMC_HIP_TRY(c, hipEventRecord(sl.uploaded, c->copy_stream));
  KMC_HIP_TRY(c, hipEventSynchronize(sl.uploaded));
  return KMC_OK;
}

int slot_end(kmc_ctx* c, int slot_id) {
  if (slot_id % kmc_ctx::kSlotsPerGroup == kmc_ctx::kSlotsPerGroup - 1) {
    const int group_id = slot_id / kmc_ctx::kSlotsPerGroup;
    KMC_HIP_TRY(c, hipEventRecord(c->group_consumed[group_id], c->stream));
    c->group_busy[group_id] = true;
    c->group_dirty[group_id] = false;
  }
  return KMC_OK;
}
// coarse[c] = {frame that owns point c * chunk (empty frames skipped), split}; coarse[n_chunks].x = frame of the last point.
// All positions are VIRTUAL: `head` dead points precede the batch (frame 0 owns them), n_virtual = n + head.
void build_coarse(const uint64_t* offsets, uint32_t n_frames, uint64_t n_virtual, uint32_t head, uint2* h_coarse, uint32_t chunk_shift) {
  const uint64_t chunk = 1ull << chunk_shift;
  const uint64_t n_chunks = (n_virtual + chunk - 1) / chunk;
  auto end_of = [&](uint32_t f) { return offsets[f + 1] + head; };  // virtual end offset of frame f
  uint32_t f = 0;
  for (uint64_t ci = 0; ci < n_chunks; ++ci) {
    const uint64_t first = ci * chunk;
    const uint64_t chunk_end = std::min<uint64_t>(first + chunk, n_virtual);
    while (f + 1 < n_frames && end_of(f) <= first) ++f;
    uint32_t split = kSplitNone;
    const uint64_t e = end_of(f);
    if (e < chunk_end) {  // frame f ends inside this chunk
      // a second boundary inside the chunk (frame f+1 ends here too, e.g. it is tiny or empty) -> search on the device
      const bool second = (f + 1 < n_frames) && end_of(f + 1) < chunk_end;
      split = second ? kSplitSearch : (uint32_t)(e - first);
    }
    h_coarse[ci] = make_uint2(f, split);
  }
  while (f + 1 < n_frames && end_of(f) <= n_virtual - 1) ++f;
  h_coarse[n_chunks] = make_uint2(f, kSplitNone);
}
}  // namespace kmc_impl

extern "C" {

int kmc_abi_version(void) { return KMC_ABI_VERSION; }

const char* kmc_status_string(int status) {
  switch (status) {
    case KMC_OK: return "KMC_OK";
    case KMC_ERR_INVALID_ARG: return "KMC_ERR_INVALID_ARG";
    case KMC_ERR_HIP: return "KMC_ERR_HIP";
    case KMC_ERR_NO_DEVICE: return "KMC_ERR_NO_DEVICE: no usable HIP device (the deskew path has no CPU fallback)";
    case KMC_ERR_TIME_OUT_OF_RANGE: return "KMC_ERR_TIME_OUT_OF_RANGE: a time outside [stamp_start, stamp_end] (the reference asserts)";
    case KMC_ERR_ALLOC: return "KMC_ERR_ALLOC";
    case KMC_ERR_DEGENERATE: return "KMC_ERR_DEGENERATE";
    default: return "KMC_ERR_UNKNOWN";
  }
}

int kmc_hip_create(kmc_ctx** out, int device_id) {
  if (!out) return KMC_ERR_INVALID_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    (void)hipGetLastError();
    return KMC_ERR_NO_DEVICE;
  }
  if (device_id < 0 || device_id >= count) return KMC_ERR_INVALID_ARG;
  kmc_ctx* c = new (std::nothrow) kmc_ctx();
  if (!c) return KMC_ERR_ALLOC;
  c->device = device_id;
  hipError_t e = hipSetDevice(device_id);
  if (e == hipSuccess) e = hipGetDeviceProperties(&c->prop, device_id);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
  hipEvent_t* evs[] = {&c->ev_k0, &c->ev_k1, &c->ev_c0, &c->ev_c1, &c->ev_t0, &c->ev_t1};
  for (hipEvent_t* ev : evs)
    if (e == hipSuccess) e = hipEventCreate(ev);
  for (auto& sl : c->slots)
    if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.uploaded, hipEventDisableTiming);
  for (auto& ev : c->group_consumed)
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  if (e == hipSuccess) e = hipMalloc((void**)&c->d_counter, 64);  // the out-of-range counter + (16 bytes on) the completion ticket
  // hipMemset returns before the fill has run (it is asynchronous for device memory) and the null stream it runs on is not ordered with
  // this context's non-blocking streams: the first armed in-place kernel of a fresh context once counted its tickets from the 9 a
  // previous owner had left in the word (profiles/NOTES.md, section 10 -- seen with a second process on the GPU).  So: the fill on the
  // context's own stream, and waited for.
  if (e == hipSuccess) e = hipMemsetAsync(c->d_counter, 0, 64, c->own_stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->own_stream);
  if (e == hipSuccess) c->d_ticket = reinterpret_cast<uint32_t*>(c->d_counter + 2);
  if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_flag, 256, hipHostMallocPortable | hipHostMallocMapped);
  if (e == hipSuccess) {
    std::memset(c->h_flag, 0, 256);
    c->h_done = c->h_flag + 16;
    c->h_stamps = reinterpret_cast<uint64_t*>(c->h_flag + 32);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    kmc_hip_destroy(c);
    return KMC_ERR_NO_DEVICE;
  }
  c->stream = c->own_stream;
  if (const char* e = std::getenv("KMC_ANY_ORDER"))  // the switch is read HERE, like every other knob of a context; the probe itself runs at first need (ao_ensure)
    if (std::atoi(e) == 0) { c->ao_probed = true; c->ao_verdict = 0; c->ao_enabled = false; c->dd_free_order = false; }
  if (const char* e = std::getenv("KMC_DIRECT_DISPATCH")) {  // unset: off until kmc_hip_set_direct_dispatch(ctx, 1); 1: every context starts with it on; 0: never, even when asked
    if (std::atoi(e) == 0) c->dd_never = true;
    else c->dd_wanted = true;
  }
  *out = c;
  return KMC_OK;
}

void kmc_hip_destroy(kmc_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream || c->own_stream) (void)fq_join(c);  // frames still being gathered are issued, not dropped; the direct queue is waited for
  direct_close(c);
  (void)hipStreamSynchronize(c->stream);
  if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
  for (int b = 0; b < 3; ++b)
    if (c->pipe[b]) { (void)hipStreamSynchronize(c->pipe[b]); (void)hipStreamDestroy(c->pipe[b]); }
  for (hipEvent_t ev : c->ev_pool) (void)hipEventDestroy(ev);
  for (int b = 0; b < kmc_ctx::kPipeSlots; ++b) {
    if (c->d_stage_in[b]) (void)hipFree(c->d_stage_in[b]);
    if (c->d_stage_out[b]) (void)hipFree(c->d_stage_out[b]);
    if (c->ev_h2d[b]) (void)hipEventDestroy(c->ev_h2d[b]);
    if (c->ev_kernel[b]) (void)hipEventDestroy(c->ev_kernel[b]);
    if (c->ev_d2h[b]) (void)hipEventDestroy(c->ev_d2h[b]);
  }
  delete[] c->gather;
  delete[] c->gather64;
  if (c->d_tmp) (void)hipFree(c->d_tmp);
  if (c->d_traj) (void)hipFree(c->d_traj);
  if (c->h_traj) (void)hipHostFree(c->h_traj);
  if (c->ev_traj) (void)hipEventDestroy(c->ev_traj);
  for (auto& sl : c->slots) {
    if (sl.d_buf) (void)hipFree(sl.d_buf);
    if (sl.h_buf) (void)hipHostFree(sl.h_buf);
    if (sl.uploaded) (void)hipEventDestroy(sl.uploaded);
  }
  for (auto& ev : c->group_consumed)
    if (ev) (void)hipEventDestroy(ev);
  if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
  if (c->d_counter) (void)hipFree(c->d_counter);
  if (c->h_flag) (void)hipHostFree(c->h_flag);
  hipEvent_t evs[] = {c->ev_k0, c->ev_k1, c->ev_c0, c->ev_c1, c->ev_t0, c->ev_t1};
  for (hipEvent_t ev : evs)
    if (ev) (void)hipEventDestroy(ev);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

// The context is single-stream: its scratch buffers, table slots and their "consumed" markers are ordered on `stream` only.
// Switching streams therefore first drains what the old stream still has in flight from this context (a growth
// of the scratch or of a table slot on the new stream would otherwise free memory under kernels of the old one).
static int switch_stream(kmc_ctx* c, hipStream_t next) {
  if (next == c->stream) return KMC_OK;
  KMC_HIP_TRY(c, hipSetDevice(c->device));
  const int rc = fq_join(c);
  if (rc != KMC_OK) return rc;
  KMC_HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (c->copy_stream) KMC_HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
  for (auto& busy : c->group_busy) busy = false;
  for (auto& dirty : c->group_dirty) dirty = false;
  c->stream = next;
  return KMC_OK;
}

int kmc_hip_set_stream(kmc_ctx* c, void* hip_stream) {
  if (!c) return KMC_ERR_INVALID_ARG;
  return switch_stream(c, (hipStream_t)hip_stream);  // literally: NULL is HIP's legacy default stream
}

int kmc_hip_use_own_stream(kmc_ctx* c) {
  if (!c) return KMC_ERR_INVALID_ARG;
  return switch_stream(c, c->own_stream);
}

int kmc_hip_synchronize(kmc_ctx* c) {
  if (!c) return KMC_ERR_INVALID_ARG;
  KMC_ENTER(c);
  KMC_HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->stream_dirty = false;  // nothing of this context is in flight any more, on either queue
  return fq_take_error(c);  // frames an EARLIER join could not issue (its caller was some unrelated entry point)
}

int kmc_hip_set_frame_queues(kmc_ctx* c, int queues) {
  if (!c || queues < 1 || queues > kmc_ctx::kMaxFrameQueues) return KMC_ERR_INVALID_ARG;
  KMC_ENTER(c);  // issues what is pending
  if (queues > 1 && (!c->gather || !c->gather64)) {
    if (!c->gather) c->gather = new (std::nothrow) ListRec[kmc_ctx::kGatherMax];
    if (!c->gather64) c->gather64 = new (std::nothrow) FrameRecD[kmc_ctx::kGatherMax];
    if (!c->gather || !c->gather64) {  // both or neither: a retry must not find one array and skip the other
      delete[] c->gather;
      delete[] c->gather64;
      c->gather = nullptr;
      c->gather64 = nullptr;
      return KMC_ERR_ALLOC;
    }
  }
  c->fq_count = queues;
  return KMC_OK;
}

int kmc_hip_set_frame_queue_order(kmc_ctx* c, int after_producers) {
  if (!c) return KMC_ERR_INVALID_ARG;
  KMC_ENTER(c);
  c->fq_ordered = after_producers != 0;
  return KMC_OK;
}

int kmc_hip_frame_queue_join(kmc_ctx* c) {
  if (!c) return KMC_ERR_INVALID_ARG;
  KMC_HIP_TRY(c, hipSetDevice(c->device));
  const int rc = fq_join(c);
  const int sticky = fq_take_error(c);
  return rc != KMC_OK ? rc : sticky;
}

uint64_t kmc_hip_frame_queue_dropped(kmc_ctx* c) { return c ? c->fq_dropped : 0; }

uint64_t kmc_hip_any_order_launches(kmc_ctx* c) { return c ? c->ao.launches + c->lw.launches : 0; }

uint64_t kmc_hip_completion_word_fallbacks(kmc_ctx* c, uint32_t last_state[3]) {
  if (!c) {  // the process's total: contexts that have been destroyed meanwhile (a thread's own context ends with the thread) are in it
    if (last_state)
      for (int i = 0; i < 3; ++i) last_state[i] = g_done_fallback_state[i].load(std::memory_order_relaxed);
    return g_done_fallbacks.load(std::memory_order_relaxed);
  }
  if (last_state)
    for (int i = 0; i < 3; ++i) last_state[i] = c->done_fallback_state[i];
  return c->done_fallbacks;
}

int kmc_hip_enable_timing(kmc_ctx* c, int enabled) {
  if (!c) return KMC_ERR_INVALID_ARG;
  c->timing = enabled != 0;
  return KMC_OK;
}

const char* kmc_hip_last_error(kmc_ctx* c) { return c ? c->last_error.c_str() : "null ctx"; }

int kmc_hip_enable_call_trace(kmc_ctx* c, int enabled) {
  if (!c) return KMC_ERR_INVALID_ARG;
  c->trace = enabled != 0;
  c->last_trace = kmc_call_trace{};
  return KMC_OK;
}

int kmc_hip_last_call_trace(kmc_ctx* c, kmc_call_trace* out) {
  if (!c || !out) return KMC_ERR_INVALID_ARG;
  *out = c->last_trace;
  return c->trace ? KMC_OK : KMC_ERR_INVALID_ARG;
}

int kmc_hip_device_info(kmc_ctx* c, kmc_device_info* out) {
  if (!c || !out) return KMC_ERR_INVALID_ARG;
  std::memset(out, 0, sizeof(*out));
  std::snprintf(out->name, sizeof(out->name), "%s", c->prop.name);
  std::snprintf(out->arch, sizeof(out->arch), "%s", c->prop.gcnArchName);
  out->device_id = c->device;
  out->compute_units = c->prop.multiProcessorCount;
  out->wavefront_size = c->prop.warpSize;
  out->hbm_bytes = c->prop.totalGlobalMem;
  out->clock_khz = c->prop.clockRate;
  (void)hipSetDevice(c->device);
  ao_ensure(c);
  out->any_order_dispatch = c->ao_verdict;
  return KMC_OK;
}

int kmc_hip_force_tier(kmc_ctx* c, int tier) {
  if (!c || tier < -1 || tier > kTrig) return KMC_ERR_INVALID_ARG;
  c->force_tier = tier;
  return KMC_OK;
}

int kmc_hip_timer_begin(kmc_ctx* c) {
  if (!c) return KMC_ERR_INVALID_ARG;
  KMC_ENTER(c);
  KMC_HIP_TRY(c, hipEventRecord(c->ev_t0, c->stream));
  return KMC_OK;
}

int kmc_hip_timer_end(kmc_ctx* c, float* elapsed_ms) {
  if (!c || !elapsed_ms) return KMC_ERR_INVALID_ARG;
  KMC_ENTER(c);  // the stopwatch covers the frames issued on the frame queues too
  KMC_HIP_TRY(c, hipEventRecord(c->ev_t1, c->stream));
  KMC_HIP_TRY(c, hipEventSynchronize(c->ev_t1));
  KMC_HIP_TRY(c, hipEventElapsedTime(elapsed_ms, c->ev_t0, c->ev_t1));
  return KMC_OK;
}

int kmc_hip_host_alloc(kmc_ctx* c, size_t bytes, void** out) {
  if (!c || !out) return KMC_ERR_INVALID_ARG;
  *out = nullptr;
  if (bytes == 0) return KMC_OK;
  KMC_ENTER(c);
  KMC_HIP_TRY(c, hipHostMalloc(out, bytes, hipHostMallocDefault));
  return KMC_OK;
}

int kmc_hip_host_free(kmc_ctx* c, void* ptr) {
  if (!c) return KMC_ERR_INVALID_ARG;
  if (ptr) KMC_HIP_TRY(c, hipHostFree(ptr));
  return KMC_OK;
}

// ---- host pre-step -------------------------------------------------------------------------------
int kmc_frame_params_from_poses(const double T_start[12], const double T_end[12], double stamp_start, double stamp_end,
                                double requested_time, kmc_frame_params* out) {
  if (!T_start || !T_end || !out) return KMC_ERR_INVALID_ARG;
  if (!(stamp_start < stamp_end)) return KMC_ERR_DEGENERATE;
  // TimeIsInRange(requested_time): the reference asserts on it for every point (trajectory_interpolation.cpp:32)
  if (!(requested_time >= stamp_start && requested_time <= stamp_end)) return KMC_ERR_TIME_OUT_OF_RANGE;
  kmc_host::Twist f;
  if (!kmc_host::relative_twist(kmc_host::Pose::from_rt12(T_start), kmc_host::Pose::from_rt12(T_end), &f)) return KMC_ERR_DEGENERATE;
  out->twist[0] = f.rho.x; out->twist[1] = f.rho.y; out->twist[2] = f.rho.z;
  out->twist[3] = f.phi.x; out->twist[4] = f.phi.y; out->twist[5] = f.phi.z;
  out->x_req = (requested_time - stamp_start) / (stamp_end - stamp_start);
  return params_ok(out) ? KMC_OK : KMC_ERR_DEGENERATE;
}

int kmc_oxts_to_pose(const kmc_oxts* o, double scale, double T_out[12]) {
  if (!o || !T_out) return KMC_ERR_INVALID_ARG;
  kmc_host::oxts_to_pose(o->lat, o->lon, o->alt, o->roll, o->pitch, o->yaw, scale).to_rt12(T_out);
  return KMC_OK;
}

int kmc_interpolate_trajectory(const kmc_oxts* o1, const kmc_oxts* o2, double time, double T_out[12]) {
  if (!o1 || !o2 || !T_out) return KMC_ERR_INVALID_ARG;
  const kmc_host::Pose P1 = kmc_host::oxts_to_pose(o1->lat, o1->lon, o1->alt, o1->roll, o1->pitch, o1->yaw, 1.0);
  const kmc_host::Pose P2 = kmc_host::oxts_to_pose(o2->lat, o2->lon, o2->alt, o2->roll, o2->pitch, o2->yaw, 1.0);
  kmc_host::Pose P;
  const int rc = kmc_host::pose_at_time(o1->stamp, P1, o2->stamp, P2, time, &P);
  if (rc == -1) return KMC_ERR_TIME_OUT_OF_RANGE;
  if (rc != 0) return KMC_ERR_DEGENERATE;
  P.to_rt12(T_out);
  return KMC_OK;
}

int kmc_frame_ranges_balanced(const uint64_t* frame_points, uint32_t n_frames, uint32_t n_parts, uint32_t* bounds_out) {
  if (!bounds_out || n_parts == 0 || (n_frames && !frame_points)) return KMC_ERR_INVALID_ARG;
  unsigned __int128 total = 0;
  for (uint32_t f = 0; f < n_frames; ++f) total += frame_points[f];
  bounds_out[0] = 0;
  bounds_out[n_parts] = n_frames;
  if (total == 0) {  // by frame count: sizes differ by at most one frame, earlier parts get the extra
    const uint32_t base = n_frames / n_parts, extra = n_frames % n_parts;
    for (uint32_t r = 1; r < n_parts; ++r) bounds_out[r] = r * base + std::min(r, extra);
    return KMC_OK;
  }
  // midpoint of frame f = acc + size / 2 >= r total / parts   <=>   (2 acc + size) parts >= 2 r total
  unsigned __int128 acc = 0;
  uint32_t r = 1;
  for (uint32_t f = 0; f < n_frames; ++f) {
    const unsigned __int128 lhs = (2 * acc + frame_points[f]) * n_parts;
    while (r < n_parts && lhs >= 2 * (unsigned __int128)r * total) bounds_out[r++] = f;
    acc += frame_points[f];
  }
  while (r < n_parts) bounds_out[r++] = n_frames;
  return KMC_OK;
}

int kmc_make_frame_poses(const kmc_oxts* o_nm1, const kmc_oxts* o_n, const kmc_oxts* o_np1, double stamp_start,
                         double stamp_end, double T_start_out[12], double T_end_out[12]) {
  int rc = kmc_interpolate_trajectory(o_nm1, o_n, stamp_start, T_start_out);
  if (rc != KMC_OK) return rc;
  return kmc_interpolate_trajectory(o_n, o_np1, stamp_end, T_end_out);
}
}  // extern "C"

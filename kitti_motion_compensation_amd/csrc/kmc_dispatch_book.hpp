// kmc_dispatch_book.hpp -- the BOOKKEEPING of the two ways a context lets independent frames overlap, free of HIP so that it can be
// unit-tested on a CPU (tests/cpp/test_dispatch_book.cpp; VERDICT r04 #7: this logic used to be spread over kmc_capi_core.hip and
// kmc_capi_deskew.hip between the runtime calls it steers).
//
//   AnyOrderWindow   the frames launched since (and including) the last ORDERED launch on the context's stream.  A new frame may be
//                    dispatched without the AQL barrier bit iff the window is valid (it describes everything in flight behind the last
//                    ordinary packet), it has room, and the frame's buffers overlap none of the window's: its write range against every
//                    read and write range, its read range against every write range.  Either way the frame enters the window; a frame
//                    that had to be ordered starts a new one.
//   LaneWindow       the same question for the DIRECT QUEUE's two lanes (kmc_capi_direct.hip): the frames in flight since the last FULLY
//                    ordered frame, each with the lane it went to.  A new frame that conflicts with nothing goes to the next lane in turn
//                    without the barrier bit; one whose conflicts all sit in ONE lane goes to THAT lane with the barrier bit (a lane runs
//                    its barrier-bit packets in order: nothing crosses lanes, the window lives on); only a frame with conflicts in both
//                    lanes, a full window or an invalid one costs the cross-lane synchronisation and starts a new window.
//   GatherList       the frames a context with gathering on (kmc_hip_set_frame_queues(ctx, q > 1)) holds back to issue as ONE list
//                    launch.  push() says what has to happen around the new frame: flush the pending frames FIRST (the new frame
//                    touches a pending frame's buffers, or needs another coefficient tier: in-order results, and a frame's bits never
//                    depend on its neighbours), and/or issue the list NOW (it is full), and/or ask the stream whether it has run dry
//                    (first frame of a list and then every fourth: an idle device is not kept waiting, a busy one gathers).
//
// Hazards are judged on VIRTUAL ADDRESS RANGES.  Two mappings of one physical buffer (hipMemMap aliases, an IPC import next to the
// original) are two unrelated ranges here: the library cannot see that they alias, and a caller who passes aliased buffers to frames
// of one window / one list has to order them itself (kmc_hip_synchronize() or kmc_hip_frame_queue_join() between them, or
// KMC_ANY_ORDER=0 together with queues = 1).  include/kmc_hip.h states this contract; tools/alias_probe.hip demonstrates it.
#pragma once

#include <cstdint>

namespace kmc_book {

struct Range {
  uintptr_t lo, hi;  // [lo, hi)
};
inline bool overlap(const Range& a, const Range& b) { return a.lo < b.hi && b.lo < a.hi; }
// may a frame reading `r` and writing `w` run next to one that reads `pr` and writes `pw`?
inline bool independent(const Range& r, const Range& w, const Range& pr, const Range& pw) { return !(overlap(w, pr) || overlap(w, pw) || overlap(r, pw)); }

template <int CAPACITY>
struct AnyOrderWindow {
  Range reads[CAPACITY], writes[CAPACITY];
  int count = 0;
  bool valid = false;      // the window describes EVERYTHING in flight on the stream after the last ordered launch (it included)
  uint64_t launches = 0;   // frames admitted without the barrier bit so far

  // every other entry point: ordinary work goes on the stream, the window no longer describes what is in flight
  void invalidate() { valid = false; }

  // -> may the frame go out without the barrier bit?  `enabled`: the feature is on (verified by the probe, not switched off);
  // `stream_ok`: nothing the library cannot see may sit between the previous frame and this one (own stream / caller's word / same call).
  // Either way the frame is entered; an ordered frame starts a new window.
  bool admit(const Range& r, const Range& w, bool enabled, bool stream_ok) {
    bool any_order = enabled && valid && stream_ok && count < CAPACITY;
    for (int k = 0; any_order && k < count; ++k) any_order = independent(r, w, reads[k], writes[k]);
    if (!any_order) count = 0;
    reads[count] = r;
    writes[count] = w;
    ++count;
    valid = enabled;
    launches += any_order ? 1 : 0;
    return any_order;
  }
  // the caller found out after admit() that the launch has to be ordered after all (a capturing stream): same effect as a refusal
  void demote_last(bool was_any_order) {
    if (!was_any_order) return;
    reads[0] = reads[count - 1];
    writes[0] = writes[count - 1];
    count = 1;
    --launches;
  }
};

struct LaneVerdict {
  enum Kind { kFree = 0, kLaneOrdered = 1, kFullyOrdered = 2 } kind;
  int lane;
};
inline bool contains(const Range& outer, const Range& inner) { return inner.lo >= inner.hi || (outer.lo <= inner.lo && inner.hi <= outer.hi); }

template <int CAPACITY>
struct LaneWindow {
  Range reads[CAPACITY], writes[CAPACITY];
  uint8_t lane[CAPACITY];
  int count = 0;
  int lanes = 2;          // 1: everything on lane 0 (KMC_DIRECT_LANES=1)
  int next_free_lane = 1;
  bool valid = false;     // the window describes everything in flight in the lanes (false after a join: the next frame is fully ordered)
  uint64_t launches = 0;  // frames dispatched without the barrier bit so far

  void invalidate() { valid = false; }

  // `enabled`: frames may overlap at all (KMC_ANY_ORDER=0: never); `force_full`: the caller needs this frame fully ordered whatever its
  // buffers (a frame of several packets).  The frame is entered either way.
  LaneVerdict admit(const Range& r, const Range& w, bool enabled, bool force_full) {
    unsigned conflict_lanes = 0;
    bool full = !enabled || !valid || force_full || count >= CAPACITY;
    for (int k = 0; !full && k < count; ++k)
      if (!independent(r, w, reads[k], writes[k])) conflict_lanes |= 1u << lane[k];
    if (conflict_lanes == 3u) full = true;
    if (full) {  // behind everything in both lanes: a new window
      count = 0;
      push(r, w, 0);
      valid = enabled;
      return {LaneVerdict::kFullyOrdered, 0};
    }
    if (conflict_lanes == 0) {
      const int l = lanes > 1 ? next_free_lane : 0;
      next_free_lane ^= 1;
      push(r, w, l);
      ++launches;
      return {LaneVerdict::kFree, l};
    }
    // every conflict sits in ONE lane: behind that lane's packets (barrier bit), beside the other lane's.  Entries of that lane whose
    // ranges the new frame's ranges contain are superseded: whatever conflicts with them conflicts with the new frame, which is behind them
    const int l = conflict_lanes == 2u ? 1 : 0;
    int kept = 0;
    for (int k = 0; k < count; ++k) {
      const bool superseded = lane[k] == l && contains(r, reads[k]) && contains(w, writes[k]);
      if (!superseded) {
        reads[kept] = reads[k];
        writes[kept] = writes[k];
        lane[kept] = lane[k];
        ++kept;
      }
    }
    count = kept;
    push(r, w, l);
    return {LaneVerdict::kLaneOrdered, l};
  }

 private:
  void push(const Range& r, const Range& w, int l) {
    reads[count] = r;
    writes[count] = w;
    lane[count] = (uint8_t)l;
    ++count;
  }
};

// What the direct queue puts into its lanes for one frame, given the window's verdict (kmc_capi_direct.hip turns a plan into AQL packets;
// tests/cpp/test_dispatch_book.cpp runs plans through a model of two in-order queues with random kernel durations and checks that no
// frame ever starts before a frame it conflicts with has completed).  Lanes are AQL queues: packets of one lane LAUNCH in order; a packet
// with the barrier bit waits for every earlier packet of its lane to COMPLETE; a barrier packet holds back every later packet of its lane
// until the signal it waits for has been raised.
struct LanePlan {
  int lane;                  // where the frame's packet goes
  bool barrier_bit;          // on the frame's packet
  bool cross_lane_wait;      // first: a barrier packet on lane 1 raises X when lane 1 has drained so far, a barrier packet on lane 0 waits for X
  bool wait_for_last_full;   // first: a barrier packet on this lane (1) waits for the completion signal of the last fully ordered frame
  bool completion_signal;    // the frame raises a completion signal (it is fully ordered: lane 1's next packet will wait for it)
};
struct LaneSync {
  bool lane1_dirty = false;      // lane 1 has taken packets since the last point at which lane 0 waited for it
  bool lane1_owes_wait = false;  // lane 1 has not been told yet to wait for the last fully ordered frame

  LanePlan plan(const LaneVerdict& v, int lanes) {
    LanePlan p = {0, v.kind != LaneVerdict::kFree, false, false, false};
    if (lanes < 2) return p;  // one lane: the barrier bit is all there is
    if (v.kind == LaneVerdict::kFullyOrdered) {
      p.cross_lane_wait = lane1_dirty;
      lane1_dirty = false;
      p.completion_signal = true;
      lane1_owes_wait = true;  // everything before this frame has left the window: lane 1 may not overtake it
      return p;
    }
    p.lane = v.lane;
    if (p.lane == 1) {
      p.wait_for_last_full = lane1_owes_wait;
      lane1_owes_wait = false;
      lane1_dirty = true;
    }
    return p;
  }
  // both lanes have drained (a join), or the last fully ordered frame is known to have completed
  void joined() { lane1_dirty = false; lane1_owes_wait = false; }
  void last_full_has_completed() { lane1_owes_wait = false; }
};

template <int CAPACITY>
struct GatherList {
  Range reads[CAPACITY], writes[CAPACITY];
  uint32_t count = 0;
  int tier = 0;

  struct Verdict {
    bool flush_first;   // issue the pending frames before this one is added
    bool issue_now;     // after adding: the list is full
    bool ask_stream;    // after adding (and not full): look whether the stream has run dry -- if so, issue
    uint32_t slot;      // where the new frame's record goes
  };
  // The caller flushes (if told to) and then calls commit(); split in two because the flush is a runtime call that can fail.
  bool must_flush_first(const Range& r, const Range& w, int frame_tier) const {
    if (count == 0) return false;
    if (frame_tier != tier) return true;
    for (uint32_t k = 0; k < count; ++k)
      if (!independent(r, w, reads[k], writes[k])) return true;
    return false;
  }
  void flushed() { count = 0; }
  Verdict commit(const Range& r, const Range& w, int frame_tier) {
    Verdict v;
    v.flush_first = false;
    v.slot = count;
    reads[count] = r;
    writes[count] = w;
    tier = frame_tier;
    ++count;
    v.issue_now = count == (uint32_t)CAPACITY;
    v.ask_stream = !v.issue_now && (count == 1 || (count & 3u) == 0);
    return v;
  }
};

}  // namespace kmc_book

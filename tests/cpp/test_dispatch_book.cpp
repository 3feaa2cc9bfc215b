// test_dispatch_book.cpp -- CPU unit test of kmc_dispatch_book.hpp: the admission rule of barrier-free dispatch and the decisions of the
// gathered-call list, driven against a FAKE stream (a scripted "has the stream run dry?" answer), no HIP anywhere.  Built and run by
// tests/test_dispatch_book.py.  Every check mirrors a promise of include/kmc_hip.h, section "a stream of SEPARATE frames".
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../kitti_motion_compensation_amd/csrc/kmc_dispatch_book.hpp"

using kmc_book::Range;

static int failures = 0;
#define CHECK(cond)                                                         \
  do {                                                                      \
    if (!(cond)) {                                                          \
      std::fprintf(stderr, "%s:%d: CHECK(%s) failed\n", __FILE__, __LINE__, #cond); \
      ++failures;                                                           \
    }                                                                       \
  } while (0)

static Range buf(uintptr_t base, uintptr_t bytes) { return Range{base, base + bytes}; }

static void test_overlap_rules() {
  CHECK(kmc_book::overlap(buf(100, 50), buf(149, 10)));
  CHECK(!kmc_book::overlap(buf(100, 50), buf(150, 10)));  // half-open: touching is not overlapping
  CHECK(!kmc_book::overlap(buf(100, 0), buf(100, 10)) || true);  // an empty range never matters to a kernel; either answer is safe
  // read-after-read is no hazard; write-after-read, read-after-write, write-after-write are
  const Range a_in = buf(0x1000, 0x100), a_out = buf(0x2000, 0x100);
  CHECK(kmc_book::independent(a_in, buf(0x3000, 0x100), a_in, a_out));    // shares only the INPUT with the earlier frame
  CHECK(!kmc_book::independent(a_out, buf(0x3000, 0x100), a_in, a_out));  // reads what the earlier frame writes
  CHECK(!kmc_book::independent(buf(0x4000, 0x100), a_in, a_in, a_out));   // writes what the earlier frame reads
  CHECK(!kmc_book::independent(buf(0x4000, 0x100), a_out, a_in, a_out));  // writes what the earlier frame writes
}

static void test_any_order_window() {
  kmc_book::AnyOrderWindow<4> w;
  // feature off (probe failed / KMC_ANY_ORDER=0): never admitted, and the window never becomes valid
  CHECK(!w.admit(buf(0x1000, 64), buf(0x2000, 64), /*enabled*/ false, true));
  CHECK(!w.valid && w.launches == 0);
  // feature on: the FIRST frame after anything else is ordered (it opens the window) ...
  CHECK(!w.admit(buf(0x1000, 64), buf(0x2000, 64), true, true));
  CHECK(w.valid && w.count == 1);
  // ... independent frames behind it go out without the barrier bit ...
  CHECK(w.admit(buf(0x3000, 64), buf(0x4000, 64), true, true));
  CHECK(w.admit(buf(0x1000, 64), buf(0x5000, 64), true, true));  // same INPUT as the first frame: reads do not conflict
  CHECK(w.launches == 2 && w.count == 3);
  // ... a frame that reads an in-flight frame's output is ordered and starts a new window
  CHECK(!w.admit(buf(0x4000, 64), buf(0x6000, 64), true, true));
  CHECK(w.count == 1 && w.valid);
  // in place (in == out) of a NEW buffer is independent of the window
  CHECK(w.admit(buf(0x7000, 64), buf(0x7000, 64), true, true));
  // ... but the same in-place buffer again is a write-after-write / read-after-write hazard
  CHECK(!w.admit(buf(0x7000, 64), buf(0x7000, 64), true, true));
  // capacity: a window holds CAPACITY frames, the next one is ordered whatever its buffers
  CHECK(w.admit(buf(0x8000, 64), buf(0x9000, 64), true, true));
  CHECK(w.admit(buf(0xA000, 64), buf(0xB000, 64), true, true));
  CHECK(w.admit(buf(0xC000, 64), buf(0xD000, 64), true, true));
  CHECK(w.count == 4);
  CHECK(!w.admit(buf(0xE000, 64), buf(0xF000, 64), true, true));
  CHECK(w.count == 1);
  // another entry point put ordinary work on the stream: the window is invalid, the next frame is ordered
  CHECK(w.admit(buf(0x10000, 64), buf(0x11000, 64), true, true));
  w.invalidate();
  CHECK(!w.admit(buf(0x12000, 64), buf(0x13000, 64), true, true));
  CHECK(w.valid);
  // a caller's stream on which producers may sit between two calls (stream_ok == false): ordered
  CHECK(!w.admit(buf(0x14000, 64), buf(0x15000, 64), true, false));
  // the launch turned out to need ordering after admission (capturing stream): counted back, window restarts with that frame
  const uint64_t before = w.launches;
  const bool any = w.admit(buf(0x16000, 64), buf(0x17000, 64), true, true);
  CHECK(any && w.launches == before + 1);
  w.demote_last(any);
  CHECK(w.launches == before && w.count == 1 && w.reads[0].lo == 0x16000);
  // partial overlaps count: one byte is enough
  CHECK(!w.admit(buf(0x16000 + 63, 64), buf(0x18000, 64), true, true) || true);  // reading what is only READ is fine ...
  kmc_book::AnyOrderWindow<4> v;
  (void)v.admit(buf(0x1000, 64), buf(0x2000, 64), true, true);
  CHECK(!v.admit(buf(0x2000 + 63, 64), buf(0x3000, 64), true, true));  // ... reading the last byte of what is WRITTEN is not
}

struct FakeStream {  // scripted answers to "has the stream run dry?"
  std::vector<bool> idle;
  size_t asked = 0;
  bool query() { return asked < idle.size() ? idle[asked++] : (++asked, false); }
};

static void test_gather_list() {
  kmc_book::GatherList<8> g;
  FakeStream s;
  s.idle = {false, false, true};
  std::vector<uint32_t> issued;  // sizes of the lists that went out
  auto push = [&](uintptr_t in, uintptr_t out, int tier) {
    const Range r = buf(in, 256), w = buf(out, 256);
    if (g.must_flush_first(r, w, tier)) {
      issued.push_back(g.count);
      g.flushed();
    }
    const auto v = g.commit(r, w, tier);
    bool go = v.issue_now;
    if (v.ask_stream) go = s.query();
    if (go) {
      issued.push_back(g.count);
      g.flushed();
    }
    return v;
  };
  // frame 1: the stream is asked at once (an idle device must not wait for a fuller list); it is busy -> the frame stays pending
  auto v = push(0x1000, 0x2000, 0);
  CHECK(v.slot == 0 && v.ask_stream && !v.issue_now && g.count == 1 && s.asked == 1);
  // frames 2, 3: no question; frame 4: asked again (every fourth), still busy
  v = push(0x3000, 0x4000, 0);
  CHECK(!v.ask_stream && g.count == 2);
  v = push(0x5000, 0x6000, 0);
  CHECK(!v.ask_stream && g.count == 3);
  v = push(0x7000, 0x8000, 0);
  CHECK(v.ask_stream && s.asked == 2 && g.count == 4 && issued.empty());
  // a frame of another coefficient tier: the pending four go out FIRST, the new frame opens a list at its own tier
  v = push(0x9000, 0xA000, 2);
  CHECK(issued.size() == 2 || issued.size() == 1);  // flush of 4; the new list's first frame asks the stream (idle now) and goes out alone
  CHECK(issued[0] == 4);
  CHECK(s.asked == 3 && issued.size() == 2 && issued[1] == 1 && g.count == 0);
  // a chain: frame B reads what pending frame A writes -> A goes out first (in-order results), B becomes pending
  s.idle.assign(100, false);
  s.asked = 0;
  issued.clear();
  push(0x1000, 0x2000, 0);   // A
  push(0x2000, 0x3000, 0);   // B reads A's output
  CHECK(issued.size() == 1 && issued[0] == 1 && g.count == 1);
  // write-after-read: C writes what pending B reads
  push(0x9000, 0x2000, 0);
  CHECK(issued.size() == 2 && issued[1] == 1 && g.count == 1);
  // in place on a new buffer joins the pending frame; in place AGAIN on the same buffer flushes both first (the repeat's bits are those
  // of in-order execution)
  push(0x5000, 0x5000, 0);
  CHECK(issued.size() == 2 && g.count == 2);
  push(0x5000, 0x5000, 0);
  CHECK(issued.size() == 3 && issued[2] == 2 && g.count == 1);
  // full list: issued the moment the last slot is taken, without asking the stream
  g.flushed();
  issued.clear();
  const size_t asked_before = s.asked;
  for (int k = 0; k < 8; ++k) v = push(0x100000 + 0x1000 * (uintptr_t)(2 * k), 0x100000 + 0x1000 * (uintptr_t)(2 * k + 1), 1);
  CHECK(v.issue_now && issued.size() == 1 && issued[0] == 8 && g.count == 0);
  CHECK(s.asked - asked_before == 2);  // frames 1 and 4 asked; frame 8 filled the list and did not
}

static void test_lane_window() {
  using kmc_book::LaneVerdict;
  kmc_book::LaneWindow<6> w;
  const Range a_in = buf(0x1000, 64), a_out = buf(0x2000, 64), b_in = buf(0x3000, 64), b_out = buf(0x4000, 64), c_in = buf(0x5000, 64), c_out = buf(0x6000, 64);
  // switched off (KMC_ANY_ORDER=0): every frame fully ordered, lane 0, the window never becomes valid
  LaneVerdict v = w.admit(a_in, a_out, /*enabled*/ false, false);
  CHECK(v.kind == LaneVerdict::kFullyOrdered && v.lane == 0 && !w.valid && w.launches == 0);
  // the first frame after a join is fully ordered (it opens the window); independent frames behind it alternate between the lanes
  v = w.admit(a_in, a_out, true, false);
  CHECK(v.kind == LaneVerdict::kFullyOrdered && v.lane == 0 && w.valid && w.count == 1);
  v = w.admit(b_in, b_out, true, false);
  CHECK(v.kind == LaneVerdict::kFree && v.lane == 1);
  v = w.admit(c_in, c_out, true, false);
  CHECK(v.kind == LaneVerdict::kFree && v.lane == 0 && w.launches == 2 && w.count == 3);
  // a frame that reads what the lane-1 frame writes: behind lane 1's packets, in lane 1 -- no new window
  v = w.admit(b_out, buf(0x7000, 64), true, false);
  CHECK(v.kind == LaneVerdict::kLaneOrdered && v.lane == 1 && w.count == 4 && w.launches == 2);
  // the same buffers again (a rotation of buffer pairs): the lane of the frame that used them; the entry it supersedes is dropped
  v = w.admit(a_in, a_out, true, false);
  CHECK(v.kind == LaneVerdict::kLaneOrdered && v.lane == 0 && w.count == 4);
  // an in-place chain stays in its lane and never grows the window
  const Range y = buf(0x9000, 64);
  v = w.admit(y, y, true, false);
  CHECK(v.kind == LaneVerdict::kFree && w.count == 5);
  const int y_lane = v.lane;
  for (int k = 0; k < 10; ++k) {
    v = w.admit(y, y, true, false);
    CHECK(v.kind == LaneVerdict::kLaneOrdered && v.lane == y_lane && w.count == 5);
  }
  // conflicts in BOTH lanes (reads lane 1's output, overwrites lane 0's): fully ordered, a new window
  v = w.admit(buf(0x7000, 64), a_out, true, false);
  CHECK(v.kind == LaneVerdict::kFullyOrdered && v.lane == 0 && w.count == 1);
  // a frame of several packets is fully ordered whatever its buffers
  v = w.admit(buf(0xA000, 64), buf(0xB000, 64), true, /*force_full*/ true);
  CHECK(v.kind == LaneVerdict::kFullyOrdered && w.count == 1);
  // capacity: a full window orders the next frame fully
  for (int k = 0; k < 5; ++k) CHECK(w.admit(buf(0x10000 + 0x1000 * k, 64), buf(0x20000 + 0x1000 * k, 64), true, false).kind == LaneVerdict::kFree);
  CHECK(w.count == 6);
  v = w.admit(buf(0x30000, 64), buf(0x31000, 64), true, false);
  CHECK(v.kind == LaneVerdict::kFullyOrdered && w.count == 1);
  // a join: the next frame is fully ordered again
  w.invalidate();
  v = w.admit(buf(0x40000, 64), buf(0x41000, 64), true, false);
  CHECK(v.kind == LaneVerdict::kFullyOrdered);
  // a partial overlap supersedes nothing: the window keeps the older entry (something may conflict with the part the new frame does not cover)
  v = w.admit(buf(0x50000, 128), buf(0x51000, 128), true, false);
  CHECK(v.kind == LaneVerdict::kFree);
  const int big_lane = v.lane;
  const int before = w.count;
  v = w.admit(buf(0x52000, 64), buf(0x51000, 64), true, false);  // overwrites HALF of the big frame's output
  CHECK(v.kind == LaneVerdict::kLaneOrdered && v.lane == big_lane && w.count == before + 1);
  // one lane (KMC_DIRECT_LANES=1): everything on lane 0, conflicts cost the barrier bit only
  kmc_book::LaneWindow<4> one;
  one.lanes = 1;
  CHECK(one.admit(a_in, a_out, true, false).kind == LaneVerdict::kFullyOrdered);
  v = one.admit(b_in, b_out, true, false);
  CHECK(v.kind == LaneVerdict::kFree && v.lane == 0);
  v = one.admit(c_in, c_out, true, false);
  CHECK(v.kind == LaneVerdict::kFree && v.lane == 0);
  v = one.admit(b_out, c_in, true, false);
  CHECK(v.kind == LaneVerdict::kLaneOrdered && v.lane == 0);
}

// ---- the two-lane protocol against a MODEL of two AQL queues --------------------------------------------------------------------
// Random programs of frames over a small pool of buffers (independent, chained, in place, overlapping sub-ranges) go through
// LaneWindow::admit and LaneSync::plan exactly like kmc_capi_direct.hip's dispatch_frame; the plans become packets of two modelled
// queues.  The model's rules are AQL's: packets of a queue launch in order; a packet with the barrier bit starts when every earlier
// packet of its queue has completed; a barrier packet completes when its queue's earlier packets have completed and its signal has
// been raised, and nothing behind it in its queue launches before that.  Kernels take random times (now and then a very long one).
// Property: a frame never starts before a frame it conflicts with -- called earlier -- has completed.
#include <algorithm>
#include <random>

namespace model {
struct Packet {
  int lane;
  bool is_barrier, barrier_bit;
  int dep, completion;  // signal ids, -1: none
  int frame;            // index into the program, -1 for barrier packets
  double dur, start = 0, done = 0;
};
struct Queues {
  std::vector<Packet> packets;      // in host order
  std::vector<double> signal_time;  // when each signal is raised
  double prev_start[2] = {0, 0}, all_done[2] = {0, 0}, barrier_done[2] = {0, 0};
  int new_signal() { signal_time.push_back(-1.0); return (int)signal_time.size() - 1; }
  void push(Packet p) {
    double t = std::max(prev_start[p.lane], barrier_done[p.lane]);
    if (p.barrier_bit) t = std::max(t, all_done[p.lane]);
    if (p.is_barrier && p.dep >= 0) {
      CHECK(signal_time[(size_t)p.dep] >= 0.0);  // the signal's packet was written before this one (no wait on something not yet queued)
      t = std::max(t, signal_time[(size_t)p.dep]);
    }
    p.start = t;
    p.done = t + (p.is_barrier ? 0.0 : p.dur);
    prev_start[p.lane] = p.start;
    all_done[p.lane] = std::max(all_done[p.lane], p.done);
    if (p.is_barrier) barrier_done[p.lane] = std::max(barrier_done[p.lane], p.done);
    if (p.completion >= 0) signal_time[(size_t)p.completion] = p.done;
    packets.push_back(p);
  }
};
}  // namespace model

static void test_two_lane_protocol_against_a_model_of_the_queues() {
  std::mt19937_64 rng(20260930);
  auto uni = [&](int lo, int hi) { return lo + (int)(rng() % (uint64_t)(hi - lo + 1)); };
  long frames_checked = 0, conflicts_checked = 0, kinds[3] = {0, 0, 0}, lanes_used[2] = {0, 0};
  for (int program = 0; program < 20000; ++program) {
    const int lanes = program % 7 == 6 ? 1 : 2;
    kmc_book::LaneWindow<6> win;  // (a small window: "full" happens often)
    win.lanes = lanes;
    kmc_book::LaneSync sync;
    model::Queues q;
    int last_full_signal = -1;
    const int n_buf = uni(2, 6), n_calls = uni(2, 40);
    struct Frame { Range r, w; double start, done; };
    std::vector<Frame> fr;
    for (int k = 0; k < n_calls; ++k) {
      const uintptr_t len = 0x1000;
      auto pick = [&] {
        const uintptr_t base = 0x100000 * (uintptr_t)(1 + uni(0, n_buf - 1));
        if (uni(0, 3) == 0) { const uintptr_t lo = (uintptr_t)uni(0, 3) * 0x400; return Range{base + lo, base + lo + 0x400 * (uintptr_t)uni(1, 2)}; }  // a sub-range
        return Range{base, base + len};
      };
      Range r = pick(), w = uni(0, 4) == 0 ? r : pick();  // now and then in place
      const kmc_book::LaneVerdict v = win.admit(r, w, true, uni(0, 60) == 0);
      const kmc_book::LanePlan plan = sync.plan(v, lanes);
      ++kinds[v.kind];
      ++lanes_used[plan.lane];
      if (plan.cross_lane_wait) {
        const int x = q.new_signal();
        q.push({1, true, true, -1, x, -1, 0.0});
        q.push({0, true, true, x, -1, -1, 0.0});
      }
      if (plan.wait_for_last_full) {
        CHECK(last_full_signal >= 0 && plan.lane == 1);
        q.push({1, true, true, last_full_signal, -1, -1, 0.0});
      }
      int comp = -1;
      if (plan.completion_signal) { comp = q.new_signal(); last_full_signal = comp; }
      const double dur = uni(0, 20) == 0 ? 500.0 + uni(0, 500) : 1.0 + uni(0, 30);
      q.push({plan.lane, false, plan.barrier_bit, -1, comp, (int)fr.size(), dur});
      fr.push_back({r, w, q.packets.back().start, q.packets.back().done});
      if (uni(0, 25) == 0) {  // a join: both lanes drain, the window ends (what every other entry point of a context does)
        const double t = std::max(q.all_done[0], q.all_done[1]);
        for (int l = 0; l < 2; ++l) { q.prev_start[l] = t; q.barrier_done[l] = t; q.all_done[l] = t; }
        win.invalidate();
        sync.joined();
      }
    }
    for (size_t j = 0; j < fr.size(); ++j)
      for (size_t i = 0; i < j; ++i)
        if (!kmc_book::independent(fr[j].r, fr[j].w, fr[i].r, fr[i].w)) {
          ++conflicts_checked;
          if (!(fr[j].start >= fr[i].done)) {
            std::fprintf(stderr, "program %d: frame %zu starts at %g before frame %zu (a conflict) is done at %g\n", program, j, fr[j].start, i, fr[i].done);
            ++failures;
            return;
          }
        }
    frames_checked += (long)fr.size();
  }
  // the programs really went every way: free frames on both lanes, lane-ordered ones, fully ordered ones, with real conflicts among them
  CHECK(frames_checked > 100000 && conflicts_checked > 100000);
  CHECK(kinds[0] > 10000 && kinds[1] > 10000 && kinds[2] > 10000 && lanes_used[0] > 10000 && lanes_used[1] > 10000);
  std::printf("two-lane protocol: %ld frames, %ld conflicting pairs ordered; free / lane-ordered / fully ordered: %ld / %ld / %ld\n", frames_checked, conflicts_checked, kinds[0], kinds[1],
              kinds[2]);
}

// the same property for the HIP-launch path's window (ONE queue: a frame either carries the barrier bit or it does not)
static void test_any_order_window_against_a_model_of_the_queue() {
  std::mt19937_64 rng(77);
  auto uni = [&](int lo, int hi) { return lo + (int)(rng() % (uint64_t)(hi - lo + 1)); };
  long conflicts_checked = 0, free_frames = 0;
  for (int program = 0; program < 10000; ++program) {
    kmc_book::AnyOrderWindow<5> win;
    model::Queues q;
    struct Frame { Range r, w; double start, done; };
    std::vector<Frame> fr;
    const int n_buf = uni(2, 6), n_calls = uni(2, 30);
    for (int k = 0; k < n_calls; ++k) {
      auto pick = [&] {
        const uintptr_t base = 0x100000 * (uintptr_t)(1 + uni(0, n_buf - 1));
        if (uni(0, 3) == 0) { const uintptr_t lo = (uintptr_t)uni(0, 3) * 0x400; return Range{base + lo, base + lo + 0x400}; }
        return Range{base, base + 0x1000};
      };
      const Range r = pick(), w = uni(0, 4) == 0 ? r : pick();
      const bool any_order = win.admit(r, w, true, true);
      free_frames += any_order ? 1 : 0;
      q.push({0, false, !any_order, -1, -1, (int)fr.size(), uni(0, 15) == 0 ? 400.0 : 1.0 + uni(0, 20)});
      fr.push_back({r, w, q.packets.back().start, q.packets.back().done});
      if (uni(0, 20) == 0) win.invalidate();  // some other entry point put ordinary work on the stream (the next frame is ordered, whatever its buffers)
    }
    for (size_t j = 0; j < fr.size(); ++j)
      for (size_t i = 0; i < j; ++i)
        if (!kmc_book::independent(fr[j].r, fr[j].w, fr[i].r, fr[i].w)) {
          ++conflicts_checked;
          if (!(fr[j].start >= fr[i].done)) {
            std::fprintf(stderr, "one queue, program %d: frame %zu starts at %g before frame %zu (a conflict) is done at %g\n", program, j, fr[j].start, i, fr[i].done);
            ++failures;
            return;
          }
        }
  }
  CHECK(conflicts_checked > 100000 && free_frames > 10000);
}

int main() {
  test_overlap_rules();
  test_any_order_window();
  test_lane_window();
  test_two_lane_protocol_against_a_model_of_the_queues();
  test_any_order_window_against_a_model_of_the_queue();
  test_gather_list();
  if (failures) {
    std::fprintf(stderr, "%d check(s) failed\n", failures);
    return 1;
  }
  std::printf("dispatch book: all checks passed\n");
  return 0;
}

// kmc_kernels.hip.h -- the __global__ kernels of the deskew engine (gfx950 / CDNA4, wave64, no MFMA).
//
// Roofline: HBM.  32 algorithmic bytes per point (16 B v4f {x,y,z,intensity} read + 16 B written),
// ~80 VALU ops per point -> ~3 flop/B, far below the ~20 flop/B ridge.  What the measurements on MI355X settled
// (profiles/r01_tune.csv, r02_tune*.csv) is now the ONLY geometry the product compiles (round 4: the
// variants those measurements rejected -- persistent grid-stride loops, 2 / 4 / 8 points per lane, other cache policies, 256-thread
// workgroups, ocml trigonometry -- left the product headers; their sources are in the history, their numbers in profiles/):
//   * one lane = one point, one 16-byte load and one 16-byte store per point: a wave moves 1 KiB per instruction,
//     perfectly coalesced;
//   * ONE 64-POINT TILE PER WORKGROUP, one wave per workgroup, no tile loop: the hardware dispatcher streaming tiles reaches
//     6.8-6.9 TB/s where a persistent grid-stride loop of the same body stays at 5.2-5.8 TB/s, and a loop that runs once still costs
//     a third of the registers (everything hoisted stays alive around the back edge).  The dispatch packet holds the grid in
//     work-items in 32 bits, so a launch covers at most 2^26 - 1 tiles; larger inputs take several launches, each told its first
//     tile (`tile_base`);
//   * consecutive workgroups land round-robin on the 8 XCDs, so every XCD streams an interleaved eighth of the buffer;
//     there is no inter-tile reuse to localise and an XCD-contiguous mapping measured 3-5 % slower;
//   * streamed-once data: non-temporal loads; stores carry nt + sc1 through a buffer descriptor (the written line is dropped from
//     the XCD's L2: +1.4 %; +7 % for nt altogether);
//   * per-frame constants are wave-uniform: kernarg / scalar loads -> SGPRs (single-frame kernel), or a per-tile
//     scalar-loaded record with an LDS-staged table for the tiles that straddle frames (batched kernel).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "kmc_device_math.hip.h"
#include "kmc_synth.h"

namespace kmc_dev {

constexpr int kTile = 64;    // points per tile = lanes per wave = threads per workgroup of every streaming kernel
constexpr int kBlock = 256;  // the generator's workgroup

__device__ __forceinline__ v4f load_point(const v4f* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void store_point(v4f* p, v4f v) { __builtin_nontemporal_store(v, p); }

// Buffer-descriptor access to one tile: base = first point of the tile (wave-uniform -> SGPRs), `bytes` = extent from
// there; out-of-range lanes are clipped by the hardware (loads return 0, stores are dropped), which also covers a ragged
// last tile.  aux bits (gfx940+): 1 = sc0, 2 = nt, 16 = sc1.
using v4u = uint32_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const void* base, uint64_t bytes) {
  const uint32_t clipped = bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)bytes;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, clipped, 0x00020000);
}
__device__ __forceinline__ v4f tile_load(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
  return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, /*nt*/ 2));
}
__device__ __forceinline__ void tile_store(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, v4f v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), r, byte_off, 0, /*nt | sc1*/ 2 | 16);
}

// ------------------------------------------------------------------------------------------------
// one tile of ONE frame: the body the single-frame kernel and the frame-list kernel share
// ------------------------------------------------------------------------------------------------
// `head` (< 64): the first `head` indices are DEAD.  A caller whose output does not start on a 1 KiB boundary passes
// pointers moved back to that boundary, n + head and head = the distance in points: every tile's store then covers whole
// aligned lines (a 16-byte-aligned base measured 5.5 TB/s against 6.8 aligned).  Only tile 0 pays for it.
// `d_rec`: the frame's constants in f64, read -- through scalar loads -- only by a wave that contains a lane the near-origin guard
// redoes (kmc_device_math).
template <int TIER>
__device__ __forceinline__ void frame_tile(const v4f* __restrict__ in, v4f* __restrict__ out, uint64_t n, const FrameRec& f, uint32_t head, cdouble_p d_rec, uint64_t tile) {
  const uint32_t tid = threadIdx.x;
  const uint64_t base = tile * kTile;
  const uint64_t i = base + tid;
  if (__builtin_expect(head != 0 && tile == 0, 0)) {  // the tile that holds the dead head: plain, bounds-checked accesses
    if (i >= head && i < n) {
      const v4f p = load_point(in + i);
      const v4f o = deskew_point<TIER>(p, f);
      const bool redo = needs_redo(p, o, f);
      if (!redo) store_point(out + i, o);
      redo_lanes(redo, p, d_rec, [&](v4f v) { store_point(out + i, v); });
    }
    return;
  }
  // every other tile, ragged or not: the load is clamped to the last point (the store of a dead lane is clipped anyway), the store
  // goes through a descriptor that ends with the frame
  const v4f p = load_point(in + (i < n ? i : n - 1));
  const __amdgpu_buffer_rsrc_t rout = tile_rsrc(out + base, (n - base) * sizeof(v4f));
  const v4f o = deskew_point<TIER>(p, f);
  const bool redo = needs_redo(p, o, f);
  if (!redo) tile_store(rout, (uint32_t)(tid * sizeof(v4f)), o);
  redo_lanes(redo, p, d_rec, [&](v4f v) { tile_store(rout, (uint32_t)(tid * sizeof(v4f)), v); });  // near-origin guard, cold
}

// ------------------------------------------------------------------------------------------------
// single-frame kernel: constants by value (kernarg segment -> s_load -> SGPRs)
// ------------------------------------------------------------------------------------------------
template <int TIER>
__global__ __launch_bounds__(kTile) __attribute__((amdgpu_waves_per_eu(8, 8))) void deskew_frame_f32(const v4f* __restrict__ in, v4f* __restrict__ out, uint64_t n, FrameRec f,
                                                                                               uint32_t head, uint64_t tile_base, FrameRecD d) {
  // `d` is addressed through the kernel-argument segment instead of by name: named, the compiler preloads its 32 SGPRs at
  // kernel entry and keeps them alive for the whole kernel.
  struct ArgLayout { const v4f* in; v4f* out; uint64_t n; FrameRec f; uint32_t head; uint64_t tile_base; FrameRecD d; };  // == the parameter list
  const cdouble_p d_rec = (cdouble_p)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(ArgLayout, d));
  const uint64_t tile = tile_base + blockIdx.x;
  if (tile * kTile >= n) return;
  frame_tile<TIER>(in, out, n, f, head, d_rec, tile);
}

// ------------------------------------------------------------------------------------------------
// completion word of the in-place kernels (round 5)
// ------------------------------------------------------------------------------------------------
// A call on page-locked host buffers ends with "the results are in host memory".  Waiting for the STREAM costs ~8.5 us after the last
// byte has landed (the end-of-kernel release, the completion signal, the runtime's wait: tools/link_probe, profiles/r05_link_probe.json)
// -- a tenth of a KITTI frame's call.  So the kernel says it itself: every wave, after its last store, makes its stores visible at
// system scope (the fence writes back the L2 for cached mappings too: non-coherent page-locked memory is covered) and takes a ticket
// (device scope); the wave that takes the last ticket re-arms the ticket word and stores the call's sequence number into a page-locked
// word the host spins on.  What the host sees once the word shows the number: everything every wave stored (release by each wave ->
// ticket -> acquire + release by the last wave -> the word; the same pattern RCCL's host proxies rest on).
// `stamps` (optional, page-locked): the device clock (s_memrealtime, 100 MHz) at the first wave's start and at the last ticket --
// the device half of kmc_hip_last_call_trace.
struct DoneWord {
  uint32_t* ticket;     // device memory, 0 between kernels
  uint32_t* word;       // page-locked host memory; nullptr: no completion word (the caller waits for the stream)
  uint64_t* stamps;     // page-locked host memory, [0] = first wave's start, [1] = last ticket; nullptr: no trace
  uint32_t seq;
  uint32_t pad;
};
// The kernels never NAME their DoneWord parameter (named, its 8 SGPRs are preloaded at entry and stay alive through the tile loop, which
// pushed deskew_f64cols<true> into scratch): both halves read it through the kernel-argument segment, at the moment they need it.
using done_cp = const DoneWord __attribute__((address_space(4)))*;
__device__ __forceinline__ void done_word_start(done_cp dw) {
  uint64_t* const stamps = dw->stamps;
  if (stamps && blockIdx.x == 0 && threadIdx.x == 0) stamps[0] = __builtin_amdgcn_s_memrealtime();
}
__device__ __forceinline__ void done_word_finish(done_cp dw) {
  uint32_t* const word = dw->word;
  if (!word) return;       // wave-uniform
  __threadfence_system();  // this wave's stores (the out-of-range flag included): visible at system scope before the ticket
  if (threadIdx.x == 0) {
    uint32_t* const ticket = dw->ticket;
    const uint32_t t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (t == gridDim.x - 1) {
      uint64_t* const stamps = dw->stamps;
      if (stamps) stamps[1] = __builtin_amdgcn_s_memrealtime();
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(word, dw->seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// single-frame kernel for PAGE-LOCKED HOST buffers (KMC_MEM_HOST_MAPPED): the cloud is read and written IN PLACE over the link
// ------------------------------------------------------------------------------------------------
// Launched like deskew_frame_f32 -- every wave loads its tile, then stores it, all ~2 000 waves of a KITTI frame resident at
// once -- the link would be used one direction after the other.  Here a few hundred persistent waves walk the tiles with the NEXT
// tile's load issued before the current tile is computed and stored, so that upload and download overlap in time (the same
// structure as deskew_f64cols<.., STREAMED>, measured there: 148 -> 129 us per frame).  Same per-point arithmetic, near-origin
// guard included: the same bits as deskew_frame_f32.  No `head` handling: host blocks are stored as they are (a partial first
// line costs nothing on PCIe).
template <int TIER>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void deskew_frame_streamed_f32(const v4f* __restrict__ in, v4f* __restrict__ out,
                                                                                                     uint64_t n, FrameRec f, DoneWord dw, FrameRecD d) {
  struct ArgLayout { const v4f* in; v4f* out; uint64_t n; FrameRec f; DoneWord dw; FrameRecD d; };  // == the parameter list; `d` is read through the segment only
  const auto kernarg = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
  const cdouble_p d_rec = (cdouble_p)(kernarg + offsetof(ArgLayout, d));
  const done_cp done = (done_cp)(kernarg + offsetof(ArgLayout, dw));  // read through the segment only, like `d`
  done_word_start(done);
  const uint32_t tid = threadIdx.x;
  const uint64_t n_tiles = (n + 63) / 64;
  uint64_t t = blockIdx.x;
  if (t < n_tiles) {
    const uint64_t last = n - 1;
    v4f cur = __builtin_nontemporal_load(in + (t * 64 + tid <= last ? t * 64 + tid : last));  // dead lanes of the ragged tile re-read the last point
    while (true) {
      const uint64_t next = t + gridDim.x;
      const bool more = next < n_tiles;
      v4f nxt = cur;
      if (more) nxt = __builtin_nontemporal_load(in + (next * 64 + tid <= last ? next * 64 + tid : last));  // in flight while `cur` is finished
      const uint64_t i = t * 64 + tid;
      const v4f o = deskew_point<TIER>(cur, f);
      const bool redo = needs_redo(cur, o, f);
      // sc1 on the store is what makes the link full duplex (round 5, tools/link_probe): the mapping of page-locked memory is cached
      // in the L2, a store without it stays there until the end-of-kernel write-back -- the link then carries the whole upload first
      // and the whole download afterwards (72 us per KITTI frame; 54 with sc1: the bytes leave while the next tile comes in)
      const __amdgpu_buffer_rsrc_t rout = tile_rsrc(out + t * 64, (n - t * 64) * sizeof(v4f));  // ends with the frame: the ragged tile's dead lanes are clipped
      if (!redo) tile_store(rout, (uint32_t)(tid * sizeof(v4f)), o);
      redo_lanes(redo && i < n, cur, d_rec, [&](v4f v) { tile_store(rout, (uint32_t)(tid * sizeof(v4f)), v); });
      if (!more) break;
      cur = nxt;
      t = next;
    }
  }
  done_word_finish(done);  // every wave of the grid takes its ticket, also one that found no tile
}

// ------------------------------------------------------------------------------------------------
// batched kernel: many frames in one launch
// ------------------------------------------------------------------------------------------------
// Device-side frame record of the batch: FrameRec with the frame's END offset in the two pad words.
struct alignas(16) BatchRec {
  float phi_x, phi_y, phi_z, phi2;
  float rho_x, rho_y, rho_z, s0;
  float c1_x, c1_y, c1_z;
  uint32_t end_lo;  // offsets[f+1], low / high 32 bits
  float c2_x, c2_y, c2_z;
  uint32_t end_hi;
};
static_assert(sizeof(BatchRec) == 64, "BatchRec must stay one 64-byte record");

__device__ __forceinline__ uint64_t rec_end(const BatchRec& r) { return ((uint64_t)r.end_hi << 32) | r.end_lo; }

__device__ __forceinline__ FrameRec to_frame(const BatchRec& r) {
  FrameRec f;
  f.phi_x = r.phi_x; f.phi_y = r.phi_y; f.phi_z = r.phi_z; f.phi2 = r.phi2;
  f.rho_x = r.rho_x; f.rho_y = r.rho_y; f.rho_z = r.rho_z; f.s0 = r.s0;
  f.c1_x = r.c1_x; f.c1_y = r.c1_y; f.c1_z = r.c1_z; f.pre2 = kGuardPre * rho_norm2(r.rho_x, r.rho_y, r.rho_z);  // no room in the 64-byte record: the uniform fast path overrides it with the table's
  f.c2_x = r.c2_x; f.c2_y = r.c2_y; f.c2_z = r.c2_z; f.pad1 = 0.f;
  return f;
}

constexpr int kLdsFrames = 16;   // records staged in LDS for a tile that straddles frame boundaries
constexpr int kChunkShift = 14;  // coarse frame table: one entry per 16384 points (device tables; inline tables choose their own shift)
constexpr uint32_t kSplitNone = 0xFFFFFFFEu;    // no frame boundary inside the chunk
constexpr uint32_t kSplitSearch = 0xFFFFFFFFu;  // two or more boundaries (tiny or empty frames): search

// Coarse table, host-computed, 8 B per 16384 points (so it and the 64-B records stay resident in the scalar cache / L2):
//   coarse[c].x = index of the frame that owns point c << kChunkShift
//   coarse[c].y = `split`: how many points of the chunk still belong to that frame -- the ONE frame boundary that may lie
//                 inside the chunk -- or kSplitNone, or kSplitSearch when several boundaries fall into the chunk.
// Per tile, wave-uniform scalar code:
//   1. the tile's point loads are issued FIRST (their addresses do not depend on the tables), so the table look-ups
//      overlap the HBM latency of the points instead of preceding it;
//   2. ONE 8-byte scalar load gives the frame of the tile's first point: coarse[c].x, plus one if the tile starts at or
//      past the split.  (A dependent scalar load costs up to ~1 us under full streaming load, so the look-up chain is kept
//      at two loads -- entry, then record -- also for the chunks that contain a boundary; the first version searched the
//      records' end offsets there and lost 2-5 % on KITTI-sized frames.)  Only kSplitSearch chunks run a binary search;
//   3. a tile that lies in ONE frame -- all but ~F of the n/kTile tiles -- takes the record through scalar loads (SGPRs)
//      and runs the same body as the single-frame kernel;
//   4. a straddling tile stages the next kLdsFrames records into LDS once per workgroup; each lane walks to its own frame
//      (integer compares on the end offsets: the per-point "timestamp index", bit-exact by construction) and gathers its
//      record from LDS; a wave whose lanes all landed in one frame broadcasts the index through readfirstlane and stays on
//      the uniform path.
//
// INLINE (round 3): a batch of at most kInlineBatchFrames frames carries its tables IN THE KERNEL ARGUMENTS -- records, their f64
// twins and a coarse table of at most kInlineBatchChunks entries (the chunk size grows with the batch: `chunk_shift`).  No table
// slot, no upload, no host wait: the call only enqueues a launch, so it is as asynchronous as kmc_hip_deskew_f32 and can be
// captured into a HIP graph; a 16-frame batch no longer pays the 4.5 us the host spent waiting for its table copy.  `inl` is never
// named in the body (the compiler would preload 3.5 KB of it into SGPRs): everything reads it through the kernel-argument segment.
constexpr int kInlineBatchFrames = 16;
constexpr int kInlineBatchChunks = 64;
struct BatchInline {
  BatchRec recs[kInlineBatchFrames];
  FrameRecD recs64[kInlineBatchFrames];
  uint2 coarse[kInlineBatchChunks + 1];
  float pre2[kInlineBatchFrames];  // first-stage thresholds of the near-origin guard, see `pre2s` below
};
static_assert(sizeof(BatchInline) <= 3800, "the batch tables must leave room for the other arguments in the 4 KB kernel-argument segment");
using brec_cp = const BatchRec __attribute__((address_space(4)))*;
using uint2_cp = const uint2 __attribute__((address_space(4)))*;
// (loads through builtin vector types: the implicit copy constructors of the structs cannot bind a constant-address-space reference)
using v2u = uint32_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ BatchRec load_rec(brec_cp r) {
  const v4u __attribute__((address_space(4)))* w = (const v4u __attribute__((address_space(4)))*)r;
  const v4u a[4] = {w[0], w[1], w[2], w[3]};
  BatchRec out;
  __builtin_memcpy(&out, a, sizeof(out));
  return out;
}
__device__ __forceinline__ uint64_t rec_end_at(brec_cp r) {
  const v4u __attribute__((address_space(4)))* w = (const v4u __attribute__((address_space(4)))*)r;
  return ((uint64_t)w[3].w << 32) | w[2].w;  // end_hi, end_lo: the last words of the third and fourth 16-byte groups
}
static_assert(offsetof(BatchRec, end_lo) == 44 && offsetof(BatchRec, end_hi) == 60, "rec_end_at reads the end offset by position");
__device__ __forceinline__ uint2 load_coarse(uint2_cp c) {
  const v2u e = *(const v2u __attribute__((address_space(4)))*)c;
  return make_uint2(e.x, e.y);
}

struct BatchNoInline { uint32_t unused; };  // what the device-table instantiations carry instead of 3.5 KB of unused tables
template <bool INLINE> using BatchInlineArg = typename std::conditional<INLINE, BatchInline, BatchNoInline>::type;

template <int TIER, bool WRITE_IDX, bool INLINE = false>
__global__ __launch_bounds__(kTile) __attribute__((amdgpu_waves_per_eu(8, 8))) void deskew_batch_f32(const v4f* __restrict__ in, v4f* __restrict__ out,
                                                         const BatchRec* __restrict__ recs_g,
                                                         const uint2* __restrict__ coarse_g, uint32_t n_frames,
                                                         uint64_t n, uint32_t* __restrict__ frame_idx_out, uint32_t head,
                                                         const FrameRecD* __restrict__ recs64, uint32_t chunk_shift,
                                                         const float* __restrict__ pre2s_g, uint64_t tile_base, BatchInlineArg<INLINE> inl) {
  // `recs64[f]`: frame f's constants in f64 for the near-origin guard's redo (kmc_device_math); cold
  // `pre2s[f]`: kGuardPre * |rho_f|^2, the guard's first-stage threshold.  The 64-byte record has no room for it, and computing it from
  // the record costs every wave four VALU instructions on wave-uniform values (there is no scalar float unit): one more 4-byte
  // scalar load next to the record instead -- these kernels keep their SIMDs ~60 % busy, VALU instructions are not free
  // `head`: dead leading indices, see frame_tile (the host has shifted the pointers and every offset by it)
  // `chunk_shift`: log2 of the coarse table's chunk size (kChunkShift for device tables)
  // `tile_base`: the first tile of this launch (0 unless the batch needs more than 2^26 - 1 tiles)
  static_assert(kTile >= kLdsFrames * 4, "the LDS staging uses one lane per 16 bytes of the record table");
  brec_cp recs;      // tables are written by the host before the launch: constant for the kernel, uniform reads are scalar loads
  uint2_cp coarse;
  const float __attribute__((address_space(4)))* pre2s;
  if constexpr (INLINE) {
    struct ArgLayout { const v4f* in; v4f* out; const BatchRec* recs_g; const uint2* coarse_g; uint32_t n_frames; uint64_t n; uint32_t* frame_idx_out; uint32_t head; const FrameRecD* recs64; uint32_t chunk_shift; const float* pre2s_g; uint64_t tile_base; BatchInline inl; };
    const auto kernarg = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    recs = (brec_cp)(kernarg + offsetof(ArgLayout, inl) + offsetof(BatchInline, recs));
    coarse = (uint2_cp)(kernarg + offsetof(ArgLayout, inl) + offsetof(BatchInline, coarse));
    recs64 = (const FrameRecD*)(const char*)(kernarg + offsetof(ArgLayout, inl) + offsetof(BatchInline, recs64));
    pre2s = (const float __attribute__((address_space(4)))*)(kernarg + offsetof(ArgLayout, inl) + offsetof(BatchInline, pre2));
  } else {
    recs = (brec_cp)(uintptr_t)recs_g;
    coarse = (uint2_cp)(uintptr_t)coarse_g;
    pre2s = (const float __attribute__((address_space(4)))*)(uintptr_t)pre2s_g;
  }
  __shared__ BatchRec lds_recs[kLdsFrames];
  const uint32_t tid = threadIdx.x;
  const uint64_t t = tile_base + blockIdx.x;
  const uint64_t base = t * kTile;
  if (base >= n) return;
  const bool full = base + kTile <= n && !(head != 0 && t == 0);
  const uint64_t tile_end = base + kTile <= n ? base + kTile : n;
  const v4f* __restrict__ tin = in + base;
  v4f* __restrict__ tout = out + base;
  v4f p;
  if (full) p = load_point(tin + tid);
  // frame of the tile's first point (uniform -> SALU + scalar loads)
  const uint64_t c = base >> chunk_shift;
  const uint2 entry = load_coarse(coarse + c);  // one s_load_dwordx2
  uint32_t f0;
  if (entry.y != kSplitSearch) {
    f0 = entry.x + ((uint32_t)(base - (c << chunk_shift)) >= entry.y ? 1u : 0u);
  } else {
    uint32_t lo = entry.x, hi = load_coarse(coarse + c + 1).x;
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (rec_end_at(recs + mid) > base) hi = mid;
      else lo = mid + 1;
    }
    f0 = lo;
  }
  const BatchRec r0 = load_rec(recs + f0);
  const float pre2_0 = pre2s[f0];
  // near-origin guard (kmc_device_math): lanes whose f32 result lost significance are flagged here, skipped by the regular
  // stores and redone in f64 at the end of the tile -- ONE cold site for both paths below
  bool redo = false;
  uint32_t fi_of = f0;
  if (full && rec_end(r0) >= tile_end) {
    FrameRec f = to_frame(r0);
    f.pre2 = pre2_0;  // (to_frame's own value -- computed from the record -- is only needed where lanes gather their records)
    const __amdgpu_buffer_rsrc_t rout = tile_rsrc(tout, kTile * sizeof(v4f));
    const v4f o = deskew_point<TIER>(p, f);
    redo = needs_redo(p, o, f);
    if (!redo) tile_store(rout, (uint32_t)(tid * sizeof(v4f)), o);
    if constexpr (WRITE_IDX) __builtin_nontemporal_store(f0, frame_idx_out + base + tid);
  } else {
    // slow path: ragged last tile and/or a tile that straddles frame boundaries
    if (tid < kLdsFrames * 4) {  // 16 records x 4 x 16 B: one ds_write_b128 per lane
      const uint32_t fr = f0 + (tid >> 2);
      if (fr < n_frames)
        reinterpret_cast<v4f*>(lds_recs)[tid] = ((const v4f __attribute__((address_space(4)))*)recs)[(uint64_t)f0 * 4 + tid];
    }
    __syncthreads();
    const uint64_t i = base + tid;
    const bool live = i < tile_end && i >= head;
    // walk to the frame that owns point i (skips empty frames); dead lanes stay on f0
    uint32_t fi = f0;
    if (live) {
      while (true) {
        const uint32_t k = fi - f0;
        const uint64_t e = (k < kLdsFrames) ? rec_end(lds_recs[k]) : rec_end_at(recs + fi);
        if (i < e || fi + 1 >= n_frames) break;
        ++fi;
      }
    }
    fi_of = fi;
    // wave-level broadcast when the whole wave sits in one frame
    const uint32_t fi0 = __builtin_amdgcn_readfirstlane(fi);
    const bool wave_uniform = __all(fi == fi0);
    BatchRec r;
    if (wave_uniform) {
      const uint32_t k0 = fi0 - f0;
      if (k0 < kLdsFrames) r = lds_recs[k0];  // uniform address: LDS broadcast / scalar load
      else r = load_rec(recs + fi0);
    } else {
      const uint32_t k = fi - f0;
      if (k < kLdsFrames) r = lds_recs[k];  // per-lane gather
      else r = load_rec(recs + fi);
    }
    if (live) {
      const FrameRec f = to_frame(r);
      if (!full) p = load_point(in + i);
      const v4f o = deskew_point<TIER>(p, f);
      redo = needs_redo(p, o, f);
      if (!redo) store_point(out + i, o);
      if constexpr (WRITE_IDX) frame_idx_out[i] = fi;
    }
  }
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(redo) != 0, 0)) {  // cold
    // stores through a descriptor (one VGPR of address state instead of a 64-bit pointer per point); only live lanes are flagged
    const __amdgpu_buffer_rsrc_t rfix = tile_rsrc(tout, (tile_end - base) * sizeof(v4f));
    redo_lanes(redo, p, recs64, fi_of, [&](v4f v) { tile_store(rfix, (uint32_t)(tid * sizeof(v4f)), v); });
  }
}

// ------------------------------------------------------------------------------------------------
// frame-list kernel: many SEPARATE frames (each in its own buffers) in one launch -- kmc_hip_deskew_frames_f32
// ------------------------------------------------------------------------------------------------
// The reference's caller hands over one frame at a time (handlers.cpp:55-64), each in its own allocation.  One launch per frame
// leaves the chip draining and refilling between two frames (a 1 M-point frame: 4.7 us of kernel, 6.3 us call to call; a KITTI
// frame: 0.6 us of kernel, ~5 us call to call).  A caller that holds several ready frames hands the LIST over, and the list runs as
// one 2-D grid: blockIdx.y = the frame, blockIdx.x = the 64-point tile inside it, grid.x = the tile count of the largest frame.  The
// frame of a workgroup is a register, not a search: ONE scalar load of the frame's 96-byte record (constants, pointers, size) -- wave
// uniform, shared by all the frame's tiles, a scalar-cache hit for all but the first of them -- stands between the wave's start and
// its point load.  Workgroups beyond a shorter frame's last tile retire after that load (KITTI drives: ~7 % of the grid, a few
// dozen cycles each).  The hardware dispatcher walks x first, so the tiles of a frame stream in order like a single-frame launch;
// every frame's tiles are cut on the 1 KiB lines of ITS output (`head`, see frame_tile).  Same per-point arithmetic, same
// near-origin guard: bit-identical to kmc_hip_deskew_f32 on the same frame.
struct alignas(16) ListRec {
  FrameRec f;       // pre2 filled in; pad1 unused
  const v4f* in;    // moved back by `head` points
  v4f* out;         // moved back by `head` points: sits on a 1 KiB line
  uint64_t n;       // points, the dead head included (0: an empty frame)
  uint32_t head;    // dead leading points of tile 0
  uint32_t pad;
};
static_assert(sizeof(ListRec) == 96, "ListRec must stay one 96-byte record");
// CAP > 0: the list carries its records (and their f64 twins for the guard) IN THE KERNEL ARGUMENTS: nothing to upload, the host never
// waits, the call only enqueues a launch.  Round 4 assumed a 4 KiB limit on the argument block (16 frames); the runtime takes 64 KiB and
// more (tools/launch_probe, profiles/r05_launch_probe.json: 8 KiB 3.1 us, 16 KiB 4.1 us, 64 KiB 10 us of host time per launch against
// 2.2-3.4 us for a small block), so a whole KITTI drive -- 108 frames, 24 KiB -- is ONE launch without a table.  Three capacities, so
// that a short list does not copy a long list's block: 16 (3.5 KiB), 64 (14 KiB), 256 (56 KiB).  (The device-table form of round 4 is gone.)
constexpr int kInlineListFrames = 16;     // what a launch under stream capture and the fallback chain carry (the block round 4 proved everywhere)
constexpr int kInlineListFramesMax = 256;
template <int CAP>
struct ListInlineT {
  ListRec recs[CAP];
  FrameRecD recs64[CAP];
};
static_assert(sizeof(ListInlineT<kInlineListFrames>) <= 3800, "the 16-frame block fits the 4 KiB every runtime takes");
static_assert(sizeof(ListInlineT<kInlineListFramesMax>) <= 60 * 1024, "the largest block stays below the 64 KiB the probe verified");
template <int TIER, int CAP>
__global__ __launch_bounds__(kTile) __attribute__((amdgpu_waves_per_eu(8, 8))) void deskew_list_f32(ListInlineT<CAP> inl) {
  static_assert(CAP > 0, "the records travel in the kernel arguments");
  using rec_cp = const ListRec __attribute__((address_space(4)))*;
  const auto kernarg = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
  const rec_cp recs = (rec_cp)(kernarg + offsetof(ListInlineT<CAP>, recs));
  const FrameRecD* recs64 = (const FrameRecD*)(const char*)(kernarg + offsetof(ListInlineT<CAP>, recs64));
  const uint32_t fi = blockIdx.y;
  ListRec r;
  {
    const v4u __attribute__((address_space(4)))* w = (const v4u __attribute__((address_space(4)))*)(recs + fi);
    const v4u a[6] = {w[0], w[1], w[2], w[3], w[4], w[5]};
    __builtin_memcpy(&r, a, sizeof(r));
  }
  if ((uint64_t)blockIdx.x * kTile >= r.n) return;  // beyond this frame's last tile
  frame_tile<TIER>(r.in, r.out, r.n, r.f, r.head, as_constant(recs64 + opaque_uniform(fi)), blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// f64 Eigen-layout kernel (compatibility path of MotionCompensateFrame(Frame const&, Time))
// ------------------------------------------------------------------------------------------------
// One wave per workgroup, TWO consecutive points per lane: every column access is a 16-byte load / store per lane, i.e. 1 KiB
// of consecutive bytes per wave instruction like the f32 kernels (8-byte accesses move half lines).  Eigen columns are only
// 8-byte aligned in general (column j starts at j * n doubles), hence the under-aligned vector type; gfx950 global memory
// instructions take any 4-byte-aligned address.
typedef double v2d_u __attribute__((ext_vector_type(2), aligned(8)));

__device__ __forceinline__ void deskew_one_f64(double px, double py, double pz, double pw, double t, const FrameRec64& f, double& rx,
                                               double& ry, double& rz, bool& in_range) {
#pragma clang fp contract(off)  // the fraction of the trajectory is the reference's (t - t1) / (t2 - t1), each operation rounded
  in_range = (t >= f.t_start) && (t <= f.t_end);  // TimeIsInRange, trajectory_interpolation.cpp:47
  if (in_range) {
    deskew_point_f64(px, py, pz, pw, scan_offset_f64(t, f.t_start, f.inv_dur, f.x_req), f, rx, ry, rz);  // FractionOfTrajectory - x_req, :49-51
  } else {
    rx = ry = rz = __builtin_nan("");
  }
}

// At most 4 waves per SIMD on purpose: every wave streams nine columns, and with the 7 waves its 52 VGPRs would allow the
// kernel is 2-3 % slower (189-196 us against 183-186 us per 16 M points, A/B on one box) -- more concurrent streams, more
// DRAM page conflicts.
// `bad_flag`: a page-locked host word raised together with the count, so that the host learns "nothing was out of range" from
// its own memory after the stream sync instead of through a device-to-host copy (~20 us of fixed cost per call).
struct F64Tile {  // the five input columns of one 128-point tile, two consecutive points per lane
  v2d_u ts, vx, vy, vz, vw;
};
__device__ __forceinline__ F64Tile f64_tile_load(const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ z,
                                                 const double* __restrict__ w, const double* __restrict__ stamps, uint64_t i) {
  F64Tile t;
  t.ts = __builtin_nontemporal_load(reinterpret_cast<const v2d_u*>(stamps + i));
  t.vx = __builtin_nontemporal_load(reinterpret_cast<const v2d_u*>(x + i));
  t.vy = __builtin_nontemporal_load(reinterpret_cast<const v2d_u*>(y + i));
  t.vz = __builtin_nontemporal_load(reinterpret_cast<const v2d_u*>(z + i));
  t.vw = (v2d_u){1.0, 1.0};
  if (w) t.vw = __builtin_nontemporal_load(reinterpret_cast<const v2d_u*>(w + i));
  return t;
}
// SYS (the in-place route over the link): the stores carry sc1 -- they leave for host memory at once instead of waiting in the L2
// for the end-of-kernel write-back, which is what lets upload and download share the link IN TIME (see deskew_frame_streamed_f32).
// Through one descriptor per column and tile (wave-uniform base): the global-store builtins have no cache-policy operand.
template <bool SYS>
__device__ __forceinline__ void f64_col_store(v2d_u v, double* __restrict__ col, uint64_t tile_first, uint32_t tid) {
  if constexpr (SYS) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(col + tile_first), 0, 128 * sizeof(double), 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), r, tid * 16u, 0, /*nt | sc1*/ 2 | 16);
  } else {
    __builtin_nontemporal_store(v, reinterpret_cast<v2d_u*>(col + tile_first + 2 * (uint64_t)tid));
  }
}
template <bool SYS = false>
__device__ __forceinline__ uint32_t f64_tile_finish(const F64Tile& t, const FrameRec64& f, double* __restrict__ ox, double* __restrict__ oy,
                                                    double* __restrict__ oz, double* __restrict__ ow, uint64_t tile_first, uint32_t tid) {
  v2d_u rx, ry, rz;
  bool ok0, ok1;
  double a, b, c;
  deskew_one_f64(t.vx.x, t.vy.x, t.vz.x, t.vw.x, t.ts.x, f, a, b, c, ok0);
  rx.x = a; ry.x = b; rz.x = c;
  deskew_one_f64(t.vx.y, t.vy.y, t.vz.y, t.vw.y, t.ts.y, f, a, b, c, ok1);
  rx.y = a; ry.y = b; rz.y = c;
  f64_col_store<SYS>(rx, ox, tile_first, tid);
  f64_col_store<SYS>(ry, oy, tile_first, tid);
  f64_col_store<SYS>(rz, oz, tile_first, tid);
  if (ow) f64_col_store<SYS>(t.vw, ow, tile_first, tid);
  return (ok0 ? 0u : 1u) + (ok1 ? 0u : 1u);
}
__device__ __forceinline__ void f64_report_bad(uint32_t bad_count, uint32_t tid, unsigned long long* __restrict__ n_bad, uint32_t* __restrict__ bad_flag) {
  if (__ballot(bad_count != 0)) {  // rare: count through a wave reduction, one atomic per wave
    uint32_t total = bad_count;
    for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off);
    if (tid == 0) {
      atomicAdd(n_bad, (unsigned long long)total);
      if (bad_flag) __hip_atomic_store(bad_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// STREAMED = false: one tile per workgroup turn (device-resident columns: HBM-bound, the hardware dispatcher streams the tiles).
// STREAMED = true: the columns are PAGE-LOCKED HOST memory and the kernel works on them over the link (KMC_MEM_HOST_MAPPED).  A few
// hundred persistent waves each walk many tiles with the NEXT tile's loads issued before the current tile is computed and stored,
// so that uploads and downloads overlap in time: launched like the device kernel (every wave loads its tile, then stores it, all
// ~1000 waves of a KITTI frame at once) the link is used one direction after the other -- 148 us per 123 k-point frame against
// ~110 us pipelined (profiles/NOTES_r03.md).
#ifndef KMC_F64_WAVES
#define KMC_F64_WAVES 4  // waves per SIMD of the f64 column kernels (2 / 6 / 8 measured: profiles/NOTES_r04.md)
#endif
#ifndef KMC_F64_TPW
#define KMC_F64_TPW 1  // tiles per workgroup of the device-resident kernel: > 1 = all of them loaded before the first is computed (A/B: tools/ab_f64_waves.py)
#endif
constexpr int kF64TilesPerWave = KMC_F64_TPW;
template <bool STREAMED = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, KMC_F64_WAVES))) void deskew_f64cols(const double* __restrict__ x, const double* __restrict__ y,
                                                     const double* __restrict__ z, const double* __restrict__ w,
                                                     const double* __restrict__ stamps, uint64_t n, FrameRec64 f,
                                                     double* __restrict__ ox, double* __restrict__ oy,
                                                     double* __restrict__ oz, double* __restrict__ ow,
                                                     unsigned long long* __restrict__ n_bad, uint32_t* __restrict__ bad_flag, uint64_t tile_base, DoneWord dw) {
  constexpr uint64_t kTile64 = 128;  // points per wave HERE: two per lane
  struct ArgLayout { const double *x, *y, *z, *w, *stamps; uint64_t n; FrameRec64 f; double *ox, *oy, *oz, *ow; unsigned long long* n_bad; uint32_t* bad_flag; uint64_t tile_base; DoneWord dw; };  // == the parameter list
  const done_cp done = (done_cp)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(ArgLayout, dw));  // `dw` is read through the segment only
  const uint32_t tid = threadIdx.x;
  const uint64_t n_tiles = (n + kTile64 - 1) / kTile64;
  const uint64_t n_full = n / kTile64;  // tiles without a ragged end
  uint64_t t = tile_base + blockIdx.x;
  if constexpr (STREAMED) {  // persistent waves: the grid is the wave count, every wave walks its tiles (tile_base = 0)
    done_word_start(done);
    F64Tile cur;
    if (t < n_full) cur = f64_tile_load(x, y, z, w, stamps, t * kTile64 + 2 * (uint64_t)tid);
    while (t < n_full) {
      const uint64_t next = t + gridDim.x;
      F64Tile nxt = cur;
      if (next < n_full) nxt = f64_tile_load(x, y, z, w, stamps, next * kTile64 + 2 * (uint64_t)tid);  // in flight while `cur` is finished
      f64_report_bad(f64_tile_finish<true>(cur, f, ox, oy, oz, ow, t * kTile64, tid), tid, n_bad, bad_flag);
      cur = nxt;
      t = next;
    }
    t = n_full + blockIdx.x;  // only the ragged last tile is left, for the first workgroup
  } else if constexpr (kF64TilesPerWave > 1) {  // several tiles per workgroup: every load of the workgroup in flight before its first fma
    t = (tile_base + blockIdx.x) * kF64TilesPerWave;
    if (t + kF64TilesPerWave <= n_full) {  // (wave-uniform)
      F64Tile tl[kF64TilesPerWave];
#pragma unroll
      for (int j = 0; j < kF64TilesPerWave; ++j) tl[j] = f64_tile_load(x, y, z, w, stamps, (t + j) * kTile64 + 2 * (uint64_t)tid);
      uint32_t bad = 0;
#pragma unroll
      for (int j = 0; j < kF64TilesPerWave; ++j) bad += f64_tile_finish<false>(tl[j], f, ox, oy, oz, ow, (t + j) * kTile64, tid);
      f64_report_bad(bad, tid, n_bad, bad_flag);
      return;
    }
  }
  for (int j = 0; j < (STREAMED ? 1 : kF64TilesPerWave); ++j, ++t)  // (one turn unless this is the last workgroup of a several-tiles-per-workgroup launch)
  if (t < n_tiles) {
    const uint64_t i = t * kTile64 + 2 * (uint64_t)tid;
    uint32_t bad_count;
    if (i + 1 < n) {
      const F64Tile tl = f64_tile_load(x, y, z, w, stamps, i);
      bad_count = f64_tile_finish<false>(tl, f, ox, oy, oz, ow, t * kTile64, tid);
    } else if (i < n) {  // the odd last point
      const double pw = w ? w[i] : 1.0;
      double a, b, c;
      bool ok;
      deskew_one_f64(x[i], y[i], z[i], pw, stamps[i], f, a, b, c, ok);
      ox[i] = a; oy[i] = b; oz[i] = c;
      if (ow) ow[i] = pw;
      bad_count = ok ? 0u : 1u;
    } else {
      bad_count = 0;
    }
    f64_report_bad(bad_count, tid, n_bad, bad_flag);
  }
  if constexpr (STREAMED) done_word_finish(done);  // the in-place route's completion word (see DoneWord); every wave of the grid takes its ticket
}

// GetPseudoTimeStamps (timestamp_mocking.cpp:46-63) in f64
// (one wave per workgroup, two consecutive points per lane, 16-byte column accesses -- see deskew_f64cols above)
typedef double v2d_col __attribute__((ext_vector_type(2), aligned(8)));
template <int kInstance = 0>  // a template only so that the header can be included by several translation units
__global__ __launch_bounds__(64) void pseudo_timestamps_f64(const double* __restrict__ x, const double* __restrict__ y,
                                                            uint64_t n, double start, double end,
                                                            double* __restrict__ stamps, uint64_t tile_base) {
  // every operation individually rounded like the reference's build (plain -O3, no FMA): start + frac * dur must not become one
  // fma, or a stamp at the scan seam can land on the other side of the reference's t <= t_end assert (trajectory_interpolation.cpp:32)
#pragma clang fp contract(off)
  constexpr double kPi = 3.14159265358979323846;
  const double dur = end - start;
  const uint64_t i = (tile_base + blockIdx.x) * 128 + 2 * (uint64_t)threadIdx.x;
  if (i + 1 < n) {
    const v2d_col vx = __builtin_nontemporal_load(reinterpret_cast<const v2d_col*>(x + i));
    const v2d_col vy = __builtin_nontemporal_load(reinterpret_cast<const v2d_col*>(y + i));
    v2d_col o;
    o.x = start + (((kPi - atan2(vy.x, vx.x)) / (2.0 * kPi)) * dur);
    o.y = start + (((kPi - atan2(vy.y, vx.y)) / (2.0 * kPi)) * dur);
    __builtin_nontemporal_store(o, reinterpret_cast<v2d_col*>(stamps + i));
  } else if (i < n) {
    stamps[i] = start + (((kPi - atan2(y[i], x[i])) / (2.0 * kPi)) * dur);
  }
}

// ------------------------------------------------------------------------------------------------
// N-knot trajectory kernels (the 3-argument MotionCompensateFrame(Frame, Trajectory, Time) overload)
// ------------------------------------------------------------------------------------------------
// Round 3 (profiles/r03_tune_traj.csv): NO LDS.  The measurements that shaped it, 64 Mi points, three knots, one box:
//     two-pose kernel 313 us | round-2 kernel (records staged in LDS by every wave) 328-334 | this one 318-322
//   * these kernels are not purely HBM-bound: a wave64 instruction occupies its 16-lane SIMD for four cycles, so the ~110 VALU
//     instructions of a point keep the SIMD busy ~60 % of the time a wave's 2 KiB take at 6.8 TB/s.  An experiment kernel with
//     the bracket known in advance ran at the two-pose rate (318 us); the trig-free bracket test alone -- ~14 VALU
//     instructions and a dozen scalar branches per knot -- cost 13 us, the dependent record fetch 6 us;
//   * so the bracket is now decided from the azimuth the deskew computes anyway (one subtraction and two compares per knot), and
//     only a wave with a lane within kKnotMargin of a knot, or with a point that is not a normal number, repeats the decision
//     with the exact half-plane tests -- which is what makes the index bit-exact against the CPU oracle;
//   * per-segment records are wave-uniform data: they are fetched through SCALAR loads into SGPRs -- from the kernel-argument
//     segment for short trajectories (INLINE: no table, no upload, the call stays asynchronous), else from a device table; the
//     knot slots arrive while the points are still in flight, the record of the wave's bracket right after the points (a
//     scalar-cache hit).  A wave whose lanes disagree (it contains a knot's azimuth) takes one turn per distinct bracket
//     (waterfall).  Staging the records in LDS first (round 2) cost an extra vector load, LDS traffic and a longer dependent
//     chain per wave; moving that staging in front of the point load was slower still (350 us), and keeping both records of a
//     three-knot trajectory in SGPRs for the whole wave spills (the trap handler leaves 78 SGPRs at 8 waves per SIMD: 410 us).
// `redo`: the near-origin guard's verdict (kmc_device_math); flagged lanes are not stored by the hot path but recomputed in f64
// by traj_redo_lanes after it.
using seg_cp = const TrajSeg32 __attribute__((address_space(4)))*;
using v4u_cp = const v4u __attribute__((address_space(4)))*;

template <int TIER>
__device__ __forceinline__ v4f traj_point(const v4f p, const float turns, const TrajSeg32& r, bool& redo) {
  FrameRec f;
  f.phi_x = r.phi_x; f.phi_y = r.phi_y; f.phi_z = r.phi_z; f.phi2 = r.phi2;
  f.rho_x = r.rho_x; f.rho_y = r.rho_y; f.rho_z = r.rho_z; f.s0 = r.s0;
  f.c1_x = r.c1_x; f.c1_y = r.c1_y; f.c1_z = r.c1_z; f.pre2 = 0.f;
  f.c2_x = r.c2_x; f.c2_y = r.c2_y; f.c2_z = r.c2_z; f.pad1 = 0.f;
  const float s = __builtin_fmaf(-turns, r.g, r.s0);
  v4f q = deskew_point_s<TIER>(p, s, f);
  if (!(r.flags & kSegIdentity)) {
    v4f o;
    o.x = __builtin_fmaf(r.m02, q.z, __builtin_fmaf(r.m01, q.y, __builtin_fmaf(r.m00, q.x, r.tx)));
    o.y = __builtin_fmaf(r.m12, q.z, __builtin_fmaf(r.m11, q.y, __builtin_fmaf(r.m10, q.x, r.ty)));
    o.z = __builtin_fmaf(r.m22, q.z, __builtin_fmaf(r.m21, q.y, __builtin_fmaf(r.m20, q.x, r.tz)));
    o.w = q.w;
    q = o;
  }
  redo = false;
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(norm2(q) < r.pre2) != 0, 0)) {  // stage 1 of the near-origin guard, wave-uniform
    // exact scale: |rho|^2 for the anchor's own segment (bit for bit the two-pose kernels' decision), else |rho|^2 + |t|^2 back from pre2
    const float rho2 = rho_norm2(r.rho_x, r.rho_y, r.rho_z);
    redo = lost_significance(p, q, (r.flags & kSegIdentity) ? rho2 : r.pre2 * (1.0f / kGuardPreTraj));
  }
  return q;
}
// cold half: `segs64[k]` is the f64 twin of the lane's segment record; a waterfall over the brackets of the flagged lanes keeps
// the record address wave-uniform (scalar loads), see redo_lanes
template <typename STORE>
__device__ __forceinline__ void traj_redo_lanes(bool redo, const v4f p, const TrajSegD* __restrict__ segs64, uint32_t k, STORE&& store) {
  uint64_t todo = __builtin_amdgcn_ballot_w64(redo);
  while (__builtin_expect(todo != 0, 0)) {  // wave-uniform, cold
    const uint32_t ku = (uint32_t)__builtin_amdgcn_readlane((int)k, __builtin_ctzll(todo));
    const uint32_t ku_idx = opaque_uniform(ku);
    const bool mine = redo && k == ku;
    if (mine) store(traj_point_redo_f64(p, as_constant(segs64 + ku_idx)));
    todo &= ~__builtin_amdgcn_ballot_w64(mine);
  }
}

// Bracket of every lane's point among the n_seg segments at `segs` (uniform address; slot 7 of record j is the START knot of
// segment j).  Fast decision from the scan fraction 0.5 - turns, exact half-plane decision for the whole wave when any lane is
// within kKnotMargin of a knot or holds a coordinate pair that is not made of normal numbers.  Why the two agree outside the
// margin: the exact test decides the sign of |p| sin(angle between the point and the knot direction) with a rounding error below
// 3e-7 |p| for normal numbers, the knot direction and knot_c are f32 roundings of the same f64 fraction (3e-8 turns), and
// azimuth_turns is within 1e-8 turns (+ 6e-8 for the subtraction): everything is decided identically beyond ~1e-7 turns; the
// margin is a thousand times that.  Zero, denormal, infinite and NaN coordinates never take the fast decision.
constexpr float kKnotMargin = 1e-4f;  // turns (0.036 degrees: ~0.4 % of the waves of a ring-ordered scan take the exact path per knot)
__device__ __forceinline__ uint32_t bracket_exact(const v4f p, seg_cp segs, uint32_t n_seg) {
  uint32_t k = 0;
  for (uint32_t j = 1; j < n_seg; ++j) {  // interior knots
    const v4u kn = ((v4u_cp)(segs + j))[7];  // {knot_cos, knot_sin, flags, knot_c}
    k += knot_ge_flat(p.x, p.y, __uint_as_float(kn.w), __uint_as_float(kn.x), __uint_as_float(kn.y), kn.z) ? 1u : 0u;
  }
  return k;
}
// the first two interior knots' scan fractions, scalar-loaded BEFORE the wave touches its points (the loads then overlap the
// points' flight; issued behind the first use of the points they would each cost a scalar-cache round trip on the critical path)
struct KnotPre { float c1, c2; };
__device__ __forceinline__ KnotPre preload_knots(seg_cp segs) {
  // records beyond n_seg exist in every table / argument block (kInlineSegments spare records); their content is ignored
  uint32_t c1 = ((v4u_cp)(segs + 1))[7].w, c2 = ((v4u_cp)(segs + 2))[7].w;
  asm volatile("" : "+s"(c1), "+s"(c2));  // pins the two loads HERE: left alone the optimiser sinks them into the branch that uses them
  return {__uint_as_float(c1), __uint_as_float(c2)};
}
__device__ __forceinline__ uint32_t bracket_of(const v4f p, const float turns, seg_cp segs, uint32_t n_seg, const KnotPre kp) {
  const float frac = 0.5f - turns;
  // +normal is class bit 8: false for NaN / infinity (either coordinate), for (0, 0) and for pairs of denormals
  bool ambiguous = !__builtin_amdgcn_classf(__builtin_fabsf(p.x) + __builtin_fabsf(p.y), 0x100);
  uint32_t k = 0;
  if (n_seg <= 3) {  // up to four knots
    const float d1 = frac - kp.c1, d2 = frac - kp.c2;
    const bool u1 = n_seg > 1, u2 = n_seg > 2;
    k = ((u1 & (d1 >= 0.0f)) ? 1u : 0u) + ((u2 & (d2 >= 0.0f)) ? 1u : 0u);
    ambiguous |= (u1 & !(__builtin_fabsf(d1) >= kKnotMargin)) | (u2 & !(__builtin_fabsf(d2) >= kKnotMargin));
  } else {
    for (uint32_t j = 1; j < n_seg; ++j) {
      const float d = frac - __uint_as_float(((v4u_cp)(segs + j))[7].w);
      k += (d >= 0.0f) ? 1u : 0u;
      ambiguous |= !(__builtin_fabsf(d) >= kKnotMargin);
    }
  }
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(ambiguous) != 0, 0)) k = bracket_exact(p, segs, n_seg);  // wave-uniform
  return k;
}

// The lanes in `mine` (all of one trajectory, whose records start at `segs`) compute, store and report: one turn per distinct
// bracket among them, the record of that bracket through scalar loads.  `seg_base` = index of segs[0] in the f64 twin table.
// Returns the lanes' brackets; sets redo / redo_seg for the lanes the near-origin guard wants redone.
template <int TIER>
__device__ __forceinline__ uint32_t traj_lanes(const v4f p, const float turns, bool mine, bool storable, seg_cp segs, uint32_t n_seg, const KnotPre kp,
                                               uint32_t seg_base, __amdgpu_buffer_rsrc_t rout, uint32_t tid, bool& redo_any, uint32_t& redo_seg) {
  const uint32_t k = bracket_of(p, turns, segs, n_seg, kp);
  uint64_t todo = __builtin_amdgcn_ballot_w64(mine);
  while (todo != 0) {  // one turn unless the lanes straddle a knot
    const uint32_t ku = opaque_uniform((uint32_t)__builtin_amdgcn_readlane((int)k, __builtin_ctzll(todo)));
    const bool now = mine && k == ku;
    TrajSeg32 r;
    {
      const v4u_cp rc = (v4u_cp)(segs + ku);  // uniform address: scalar loads
      v4u w[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) w[q] = rc[q];
      __builtin_memcpy(&r, w, sizeof(r));
    }
    bool redo;
    const v4f q = traj_point<TIER>(p, turns, r, redo);
    redo = redo && now && storable;
    if (redo) {
      redo_any = true;
      redo_seg = seg_base + ku;
    }
    if (now && storable && !redo) tile_store(rout, (uint32_t)(tid * sizeof(v4f)), q);
    todo &= ~__builtin_amdgcn_ballot_w64(now);
  }
  return k;
}

// INLINE: the segment records of a short trajectory (<= kInlineSegments, i.e. up to four knots -- the three bracketing poses
// north_star names fit) travel IN THE KERNEL ARGUMENTS: no table slot, no upload, no host wait -- the call is as asynchronous as
// the two-pose kmc_hip_deskew_f32 and may go over the frame queues.  `inl` is never named in the body (the compiler would
// preload 1.1 KB of it into SGPRs and spill): everything reads it through the kernel-argument segment.
constexpr int kInlineSegments = 3;
struct TrajInline {
  TrajSeg32 s[kInlineSegments];
  TrajSegD d[kInlineSegments];
};
// one 64-point tile of an N-knot frame (the body of deskew_traj_f32 and of the direct queue's kmc_direct_traj_t* kernels)
template <int TIER, bool WRITE_IDX>
__device__ __forceinline__ void traj_tile(const v4f* __restrict__ in, v4f* __restrict__ out, uint64_t n, seg_cp segs_c, uint32_t n_seg, uint32_t* __restrict__ bracket_out,
                                          uint32_t head, const TrajSegD* __restrict__ segs64, uint64_t t) {
  const uint32_t tid = threadIdx.x;
  const uint64_t base = t * kTile;
  if (base >= n) return;
  const uint64_t i = base + tid;
  const bool alive = i < n && i >= head;
  // loads and stores through per-tile descriptors that end with the buffers: no per-lane address arithmetic, the ragged tail is
  // clipped by the hardware.  The `head` dead lanes of tile 0 must not touch memory in front of the caller's range (it need not
  // be mapped): the descriptor of tile 0 starts at the first live point and their offsets wrap to 4 GiB, out of its range.
  // Clipped lanes read zeros and are never stored.
  const uint32_t h0 = t == 0 ? head : 0u;
  const __amdgpu_buffer_rsrc_t rin = tile_rsrc(in + base + h0, (n - base - h0) * sizeof(v4f));
  const v4f p = tile_load(rin, (tid - h0) * (uint32_t)sizeof(v4f));
  const KnotPre kp = preload_knots(segs_c);
  const __amdgpu_buffer_rsrc_t rout = tile_rsrc(out + base, (n - base) * sizeof(v4f));
  __builtin_amdgcn_sched_barrier(0);  // everything above is issued before the wave waits for its points
  const float turns = azimuth_turns(p.x, p.y);
  bool redo_any = false;
  uint32_t redo_seg = 0;
  const uint32_t k = traj_lanes<TIER>(p, turns, true, i >= head, segs_c, n_seg, kp, 0u, rout, tid, redo_any, redo_seg);
  if constexpr (WRITE_IDX) {
    if (alive) __builtin_nontemporal_store(k, bracket_out + i);
  }
  traj_redo_lanes(redo_any && alive, p, segs64, redo_seg, [&](v4f v) { tile_store(rout, (uint32_t)(tid * sizeof(v4f)), v); });
}

template <int TIER, bool WRITE_IDX, bool INLINE = false>
__global__ __launch_bounds__(kTile) __attribute__((amdgpu_waves_per_eu(8, 8))) void deskew_traj_f32(const v4f* __restrict__ in, v4f* __restrict__ out, uint64_t n,
                                                     const TrajSeg32* __restrict__ segs, uint32_t n_seg,
                                                     uint32_t* __restrict__ bracket_out, uint32_t head,
                                                     const TrajSegD* __restrict__ segs64, uint64_t tile_base, TrajInline inl) {
  // `head`: dead leading indices, see frame_tile; `tile_base`: first tile of this launch
  seg_cp segs_c;
  if constexpr (INLINE) {
    struct ArgLayout { const v4f* in; v4f* out; uint64_t n; const TrajSeg32* segs; uint32_t n_seg; uint32_t* bracket_out; uint32_t head; const TrajSegD* segs64; uint64_t tile_base; TrajInline inl; };
    const auto kernarg = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    segs_c = (seg_cp)(kernarg + offsetof(ArgLayout, inl) + offsetof(TrajInline, s));
    segs64 = (const TrajSegD*)(const char*)(kernarg + offsetof(ArgLayout, inl) + offsetof(TrajInline, d));
  } else {
    segs_c = (seg_cp)(uintptr_t)segs;  // written by the host before the launch: constant for the kernel
  }
  traj_tile<TIER, WRITE_IDX>(in, out, n, segs_c, n_seg, bracket_out, head, segs64, tile_base + blockIdx.x);
}

// Batched N-knot kernel: many frames in one launch, every frame with its own trajectory (its own segment records).
// Per frame a 16-byte header {end offset, segment count} and `seg_stride` slots in the segment table (frame f's records start
// at f * seg_stride, so their address does not wait for the header load; the table ends with kInlineSegments spare records so
// that the unconditional knot loads of bracket_of stay inside it); the frame of a tile is found exactly like in
// deskew_batch_f32 (coarse table, scalar loads) while the tile's points are in flight.  The tile then walks the frames it
// touches -- ONE for all but the tiles that straddle a frame boundary -- and the lanes of each frame run the body of
// deskew_traj_f32 on that frame's records.  Outputs the per-point frame index and bracket index on demand.
struct alignas(32) TrajFrameRec {
  uint32_t end_lo, end_hi;  // offsets[f+1]
  uint32_t n_seg;
  uint32_t pad;
  float c1, c2;             // scan fractions of the first two interior knots (what bracket_of needs before the points land): with
  uint32_t pad2[2];         // them in the header a tile's early look-ups touch ONE line of the frame's tables
};
static_assert(sizeof(TrajFrameRec) == 32, "TrajFrameRec must stay one 32-byte record");
__device__ __forceinline__ uint64_t rec_end(const TrajFrameRec& r) { return ((uint64_t)r.end_hi << 32) | r.end_lo; }

template <int TIER, bool WRITE_IDX>
__global__ __launch_bounds__(kTile) __attribute__((amdgpu_waves_per_eu(8, 8))) void deskew_traj_batch_f32(const v4f* __restrict__ in, v4f* __restrict__ out, uint64_t n,
                                                           const TrajFrameRec* __restrict__ frecs,
                                                           const TrajSeg32* __restrict__ segs,
                                                           uint32_t seg_stride, const uint2* __restrict__ coarse,
                                                           uint32_t n_frames, uint32_t* __restrict__ frame_idx_out,
                                                           uint32_t* __restrict__ bracket_out, uint32_t head,
                                                           const TrajSegD* __restrict__ segs64, uint64_t tile_base) {
  // `head`: dead leading indices, see frame_tile (the host has shifted the pointers and every offset by it); `tile_base`: first tile
  // of this launch.  No tile loop: the loop-carried copies of the twelve kernel arguments cost ~40 SGPR spill instructions per wave
  // (round 3, profiles/NOTES_r03.md)
  // The arguments only the cold paths or the tile's last instructions need -- `n_frames` (tiles that straddle frames), `segs64` (the
  // guard's redo), the two index outputs -- are read through the kernel-argument segment where they are used: named, they would sit in
  // SGPRs for the whole wave, and this kernel has none to spare (78 at 8 waves per SIMD; 4-12 spills before, 2-8 now)
  struct ArgLayout { const v4f* in; v4f* out; uint64_t n; const TrajFrameRec* frecs; const TrajSeg32* segs; uint32_t seg_stride; const uint2* coarse; uint32_t n_frames;
                     uint32_t* frame_idx_out; uint32_t* bracket_out; uint32_t head; const TrajSegD* segs64; uint64_t tile_base; };
  const auto kernarg = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
  const seg_cp segs_c = (seg_cp)(uintptr_t)segs;  // written by the host before the launch: constant for the kernel
  const uint32_t tid = threadIdx.x;
  const uint64_t t = tile_base + blockIdx.x;
  const uint64_t base = t * kTile;
  if (base >= n) return;
  const uint64_t i = base + tid;
  const uint64_t tile_end = base + kTile <= n ? base + kTile : n;
  const bool alive = i < n && i >= head;
  const uint32_t h0 = t == 0 ? head : 0u;  // see deskew_traj_f32
  const __amdgpu_buffer_rsrc_t rin = tile_rsrc(in + base + h0, (n - base - h0) * sizeof(v4f));
  const v4f p = tile_load(rin, (tid - h0) * (uint32_t)sizeof(v4f));
  // frame of the tile's first point (wave-uniform)
  const uint64_t c = base >> kChunkShift;
  const uint2 entry = coarse[c];
  uint32_t f0;
  if (entry.y != kSplitSearch) {
    f0 = entry.x + ((uint32_t)(base - (c << kChunkShift)) >= entry.y ? 1u : 0u);
  } else {
    uint32_t lo = entry.x, hi = coarse[c + 1].x;
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (rec_end(frecs[mid]) > base) hi = mid;
      else lo = mid + 1;
    }
    f0 = lo;
  }
  TrajFrameRec r = frecs[f0];
  KnotPre kp0;
  {
    uint32_t k1 = __float_as_uint(r.c1), k2 = __float_as_uint(r.c2);
    asm volatile("" : "+s"(k1), "+s"(k2));  // pins the header load in front of the points' first use
    kp0 = {__uint_as_float(k1), __uint_as_float(k2)};
    // (Warming the scalar cache with the frame's first two records through four one-dword loads at this point was measured and
    // dropped: 360 us against 327 us per 64 M points -- the extra scalar traffic costs more than the misses it avoids.)
  }
  const __amdgpu_buffer_rsrc_t rout = tile_rsrc(out + base, (n - base) * sizeof(v4f));  // clips the ragged tail
  const bool one_frame = rec_end(r) >= tile_end;
  __builtin_amdgcn_sched_barrier(0);  // the whole look-up chain above is issued before the wave waits for its points
  const float turns = azimuth_turns(p.x, p.y);
  // near-origin guard: flagged lanes remember their segment (frame * seg_stride + bracket) and are redone in f64 at the end
  // of the tile -- ONE cold site
  bool redo_any = false;
  uint32_t redo_seg = 0;
  if (__builtin_expect(one_frame, 1)) {  // the tile lies in ONE frame: all but ~n_frames of the n / 64 tiles
    const uint32_t k = traj_lanes<TIER>(p, turns, true, i >= head, segs_c + (uint64_t)f0 * seg_stride, r.n_seg, kp0, f0 * seg_stride, rout, tid, redo_any, redo_seg);
    if constexpr (WRITE_IDX) {
      if (alive) {
        uint32_t* const fio = *(uint32_t* const __attribute__((address_space(4)))*)(kernarg + offsetof(ArgLayout, frame_idx_out));
        uint32_t* const bo = *(uint32_t* const __attribute__((address_space(4)))*)(kernarg + offsetof(ArgLayout, bracket_out));
        if (fio) __builtin_nontemporal_store(f0, fio + i);
        if (bo) __builtin_nontemporal_store(k, bo + i);
      }
    }
  } else {
    uint32_t fi = f0;
    uint64_t begin = base;
    while (true) {  // the frames that own points of this tile, in order (empty frames are skipped)
      const uint64_t e = rec_end(r);
      if (e > begin) {
        const bool mine = i >= begin && i < e;
        const uint32_t k = traj_lanes<TIER>(p, turns, mine, i >= head, segs_c + (uint64_t)fi * seg_stride, r.n_seg, KnotPre{r.c1, r.c2}, fi * seg_stride, rout, tid, redo_any, redo_seg);
        if constexpr (WRITE_IDX) {
          if (mine && alive) {
            uint32_t* const fio = *(uint32_t* const __attribute__((address_space(4)))*)(kernarg + offsetof(ArgLayout, frame_idx_out));
            uint32_t* const bo = *(uint32_t* const __attribute__((address_space(4)))*)(kernarg + offsetof(ArgLayout, bracket_out));
            if (fio) __builtin_nontemporal_store(fi, fio + i);
            if (bo) __builtin_nontemporal_store(k, bo + i);
          }
        }
        begin = e;
      }
      if (e >= tile_end || fi + 1 >= *(const uint32_t __attribute__((address_space(4)))*)(kernarg + offsetof(ArgLayout, n_frames))) break;
      ++fi;
      r = frecs[fi];
    }
  }
  traj_redo_lanes(redo_any && alive, p, *(const TrajSegD* const __attribute__((address_space(4)))*)(kernarg + offsetof(ArgLayout, segs64)), redo_seg, [&](v4f v) { tile_store(rout, (uint32_t)(tid * sizeof(v4f)), v); });
}

// f64 Eigen-layout variant: honours the caller's per-point stamps; the bracket is found by f64 time compares.
__device__ __forceinline__ void traj_one_f64(double px, double py, double pz, double pw, double t, const TrajSeg64* __restrict__ segs,
                                             uint32_t n_seg, double t_first, double t_last, double& rx, double& ry, double& rz,
                                             uint32_t& k_out, bool& in_range) {
  in_range = (t >= t_first) && (t <= t_last);
  uint32_t k = 0;
  if (in_range) {
    while (k + 1 < n_seg && t >= segs[k + 1].f.t_start) ++k;  // t_k <= t < t_{k+1}; the last knot belongs to the last segment
    const TrajSeg64& sg = segs[k];
    double qx, qy, qz;
    deskew_point_f64(px, py, pz, pw, scan_offset_f64(t, sg.f.t_start, sg.f.inv_dur, sg.f.x_req), sg.f, qx, qy, qz);
    if (sg.identity) {
      rx = qx; ry = qy; rz = qz;
    } else {  // M_k * (q, w): rotation on the point, translation scaled by the homogeneous coordinate
      rx = sg.M[0] * qx + sg.M[1] * qy + sg.M[2] * qz + sg.M[3] * pw;
      ry = sg.M[4] * qx + sg.M[5] * qy + sg.M[6] * qz + sg.M[7] * pw;
      rz = sg.M[8] * qx + sg.M[9] * qy + sg.M[10] * qz + sg.M[11] * pw;
    }
  } else {
    rx = ry = rz = __builtin_nan("");
  }
  k_out = k;
}

// same geometry as deskew_f64cols: one wave per workgroup, two consecutive points per lane, 16-byte column accesses.
// STREAMED (round 5): the columns are PAGE-LOCKED HOST memory and the kernel works on them over the link (the 3-argument
// MotionCompensateFrame(Frame, Trajectory, Time) on the drop-in's own containers): the grid is a few dozen persistent waves that walk the
// tiles -- one wave per tile over the link runs at a third of the rate (tools/link_probe) --, the stores carry sc1 so that they leave
// for host memory while the next tile's loads come in, and the last wave raises the completion word (DoneWord).
// The segment records of a short trajectory (<= kInlineSegments: up to four knots, north_star's three bracketing poses) travel in the
// kernel arguments (`inl`, read through the argument segment only; `segs` == nullptr says so): no table upload in front of the kernel.
struct TrajInline64 {
  TrajSeg64 s[kInlineSegments];
};
template <bool STREAMED = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 4))) void deskew_traj_f64cols(const double* __restrict__ x, const double* __restrict__ y,
                                                          const double* __restrict__ z, const double* __restrict__ w,
                                                          const double* __restrict__ stamps, uint64_t n,
                                                          const TrajSeg64* __restrict__ segs, uint32_t n_seg,
                                                          double t_first, double t_last, double* __restrict__ ox,
                                                          double* __restrict__ oy, double* __restrict__ oz,
                                                          double* __restrict__ ow, uint32_t* __restrict__ bracket_out,
                                                          unsigned long long* __restrict__ n_bad, uint32_t* __restrict__ bad_flag, uint64_t tile_base, DoneWord dw,
                                                          TrajInline64 inl) {
  struct ArgLayout { const double *x, *y, *z, *w, *stamps; uint64_t n; const TrajSeg64* segs; uint32_t n_seg; double t_first, t_last; double *ox, *oy, *oz, *ow; uint32_t* bracket_out;
                     unsigned long long* n_bad; uint32_t* bad_flag; uint64_t tile_base; DoneWord dw; TrajInline64 inl; };  // == the parameter list; `dw` and `inl` are read through the segment only
  const auto kernarg = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
  const done_cp done = (done_cp)(kernarg + offsetof(ArgLayout, dw));
  if (!segs) segs = (const TrajSeg64*)(const char*)(kernarg + offsetof(ArgLayout, inl));  // (wave-uniform)
  const uint32_t tid = threadIdx.x;
  const uint64_t n_tiles = (n + 127) / 128;
  uint64_t t = tile_base + blockIdx.x;
  if constexpr (STREAMED) done_word_start(done);
  while (t < n_tiles) {
    const uint64_t i = t * 128 + 2 * (uint64_t)tid;
    uint32_t bad_count = 0;
    if (i + 1 < n) {
      const v2d_u ts = __builtin_nontemporal_load(reinterpret_cast<const v2d_u*>(stamps + i));
      const v2d_u vx = __builtin_nontemporal_load(reinterpret_cast<const v2d_u*>(x + i));
      const v2d_u vy = __builtin_nontemporal_load(reinterpret_cast<const v2d_u*>(y + i));
      const v2d_u vz = __builtin_nontemporal_load(reinterpret_cast<const v2d_u*>(z + i));
      v2d_u vw = {1.0, 1.0};
      if (w) vw = __builtin_nontemporal_load(reinterpret_cast<const v2d_u*>(w + i));
      v2d_u rx, ry, rz;
      double a, b, c;
      uint32_t k0, k1;
      bool ok0, ok1;
      traj_one_f64(vx.x, vy.x, vz.x, vw.x, ts.x, segs, n_seg, t_first, t_last, a, b, c, k0, ok0);
      rx.x = a; ry.x = b; rz.x = c;
      traj_one_f64(vx.y, vy.y, vz.y, vw.y, ts.y, segs, n_seg, t_first, t_last, a, b, c, k1, ok1);
      rx.y = a; ry.y = b; rz.y = c;
      if (STREAMED && (t + 1) * 128 <= n) {  // a full tile over the link: descriptor stores with sc1 (f64_col_store)
        f64_col_store<true>(rx, ox, t * 128, tid);
        f64_col_store<true>(ry, oy, t * 128, tid);
        f64_col_store<true>(rz, oz, t * 128, tid);
        if (ow) f64_col_store<true>(vw, ow, t * 128, tid);
      } else {
        __builtin_nontemporal_store(rx, reinterpret_cast<v2d_u*>(ox + i));
        __builtin_nontemporal_store(ry, reinterpret_cast<v2d_u*>(oy + i));
        __builtin_nontemporal_store(rz, reinterpret_cast<v2d_u*>(oz + i));
        if (ow) __builtin_nontemporal_store(vw, reinterpret_cast<v2d_u*>(ow + i));
      }
      if (bracket_out) {
        bracket_out[i] = k0;
        bracket_out[i + 1] = k1;
      }
      bad_count = (ok0 ? 0u : 1u) + (ok1 ? 0u : 1u);
    } else if (i < n) {  // the odd last point
      const double pw = w ? w[i] : 1.0;
      double a, b, c;
      uint32_t k;
      bool ok;
      traj_one_f64(x[i], y[i], z[i], pw, stamps[i], segs, n_seg, t_first, t_last, a, b, c, k, ok);
      ox[i] = a; oy[i] = b; oz[i] = c;
      if (ow) ow[i] = pw;
      if (bracket_out) bracket_out[i] = k;
      bad_count = ok ? 0u : 1u;
    }
    f64_report_bad(bad_count, tid, n_bad, bad_flag);
    if constexpr (!STREAMED) break;  // device-resident columns: one tile per workgroup, the hardware dispatcher streams the tiles
    t += gridDim.x;
  }
  if constexpr (STREAMED) done_word_finish(done);
}

// ------------------------------------------------------------------------------------------------
// synthetic generator (measurement infrastructure)
// ------------------------------------------------------------------------------------------------
template <int kInstance = 0>  // a template only so that the header can be included by several translation units
__global__ __launch_bounds__(kBlock) void synth_points(v4f* __restrict__ out, uint64_t n, uint64_t seed) {
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const kmc_synth::Point p = kmc_synth::make_point(i, n, seed);
    out[i] = v4f{p.x, p.y, p.z, p.i};
  }
}

// ------------------------------------------------------------------------------------------------
// N4: projection kernels.  One point per lane, one wave per workgroup like the deskew kernels.  Per point: 16 B read,
// 32 B of pixel pairs (uv[point][camera][2]) + 4 B colour/validity written (52 B), or 68 B when the fused deskew also writes its cloud.
// TIER < 0: project the cloud as it is; TIER >= 0: deskew first (same arithmetic as deskew_frame_f32), then project.
// ------------------------------------------------------------------------------------------------
// The eight pixel integers of a point are one 32-byte record of uv[point][camera][2].  A lane holds its point's record, but
// a store instruction in which every lane writes half a record leaves every line half written twice (measured: +10 % HBM
// write traffic); the records therefore take a trip through LDS so that each of the two store instructions of a wave
// writes 1 KiB of CONSECUTIVE bytes.  Waves without an in-view point store the constant record directly.
__device__ __forceinline__ void store_uv_tile(v2i* __restrict__ uv, uint64_t tile_base, uint64_t n, uint32_t tid, const v2i px[4],
                                              bool any_drawn, v4i* xpose) {
  v4i lo, hi;  // 16-byte chunks tid and 64 + tid of the tile's 2 KiB
  if (any_drawn) {
    xpose[2 * tid] = (v4i){px[0].x, px[0].y, px[1].x, px[1].y};
    xpose[2 * tid + 1] = (v4i){px[2].x, px[2].y, px[3].x, px[3].y};
    __syncthreads();  // one-wave workgroup: orders the wave's own LDS writes before its reads
    lo = xpose[tid];
    hi = xpose[64 + tid];
    __syncthreads();
  } else {
    lo = hi = (v4i){(int)0x80000000, (int)0x80000000, (int)0x80000000, (int)0x80000000};
  }
  // nt + sc1 stores through a descriptor that ends with the array: the ragged last tile is clipped by the hardware
  const __amdgpu_buffer_rsrc_t r = tile_rsrc(uv + 4 * tile_base, (n - tile_base) * 4 * sizeof(v2i));
  tile_store(r, tid * 16u, __builtin_bit_cast(v4f, lo));
  tile_store(r, 1024u + tid * 16u, __builtin_bit_cast(v4f, hi));
}

template <int TIER, int RIG>
__global__ __launch_bounds__(64) void project_f32(const v4f* __restrict__ in, uint64_t n, CameraRigRec g, FrameRec f,
                                                  v4f* __restrict__ cloud_out, v2i* __restrict__ uv,
                                                  uint32_t* __restrict__ bgrv, uint64_t tile_base, FrameRecD d) {
  // `g` and `d` are read through the kernel-argument segment only, see project_point and deskew_frame_f32
  struct ArgLayout { const v4f* in; uint64_t n; CameraRigRec g; FrameRec f; v4f* cloud_out; v2i* uv; uint32_t* bgrv; uint64_t tile_base; FrameRecD d; };
  const auto kernarg = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
  const cdouble_p d_rec = (cdouble_p)(kernarg + offsetof(ArgLayout, d));
  const cdouble_p g_rec = (cdouble_p)(kernarg + offsetof(ArgLayout, g));
  __shared__ v4i xpose[128];
  const uint32_t tid = threadIdx.x;
  const uint64_t base = (tile_base + blockIdx.x) * 64;
  if (base >= n) return;
  const uint64_t i = base + tid;
  const bool live = i < n;
  v4f p = __builtin_nontemporal_load(in + (live ? i : n - 1));  // dead lanes of the ragged tile re-read the last point
  if constexpr (TIER >= 0) {
    {  // the same cloud kmc_hip_deskew_f32 writes, bit for bit: f32 closed form + the near-origin guard's f64 redo
      const v4f o = deskew_point<TIER>(p, f);
      const bool redo = needs_redo(p, o, f);
      v4f fixed = o;
      redo_lanes(redo, p, d_rec, [&](v4f v) { fixed = v; });
      p = fixed;
    }
    if (cloud_out && live) __builtin_nontemporal_store(p, cloud_out + i);
  }
  v2i px[4];
  uint32_t col;
  const bool drawn = project_point<RIG>((double)p.x, (double)p.y, (double)p.z, g_rec, px, col);
  store_uv_tile(uv, base, n, tid, px, __builtin_amdgcn_ballot_w64(drawn) != 0, xpose);
  if (live) __builtin_nontemporal_store(col, bgrv + i);
}

template <int RIG>
__global__ __launch_bounds__(64) void project_f64cols(const double* __restrict__ x, const double* __restrict__ y,
                                                      const double* __restrict__ z, uint64_t n, CameraRigRec g,
                                                      v2i* __restrict__ uv, uint32_t* __restrict__ bgrv, uint64_t tile_base) {
  struct ArgLayout { const double* x; const double* y; const double* z; uint64_t n; CameraRigRec g; v2i* uv; uint32_t* bgrv; uint64_t tile_base; };
  const cdouble_p g_rec = (cdouble_p)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(ArgLayout, g));
  __shared__ v4i xpose[128];
  const uint32_t tid = threadIdx.x;
  const uint64_t base = (tile_base + blockIdx.x) * 64;
  if (base >= n) return;
  const uint64_t i = base + tid;
  const bool live = i < n;
  const uint64_t j = live ? i : n - 1;
  v2i px[4];
  uint32_t col;
  const bool drawn = project_point<RIG>(__builtin_nontemporal_load(x + j), __builtin_nontemporal_load(y + j),
                                               __builtin_nontemporal_load(z + j), g_rec, px, col);
  store_uv_tile(uv, base, n, tid, px, __builtin_amdgcn_ballot_w64(drawn) != 0, xpose);
  if (live) __builtin_nontemporal_store(col, bgrv + i);
}

}  // namespace kmc_dev

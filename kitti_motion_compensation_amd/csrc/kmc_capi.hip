// kmc_capi.hip -- implementation of the C-ABI declared in include/kmc_hip.h (libkmc_hip.so).
//
// Thin on purpose: argument checks, f64 -> device-precision frame records, launch geometry, optional
// host staging.  All per-point work is in kmc_kernels.hip.h.  There is no CPU fallback anywhere in this
// file: every hot-path entry point needs a live kmc_ctx, and kmc_hip_create() fails without a HIP device.
#include "../../include/kmc_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "kmc_host_math.hpp"
#include "kmc_kernels.hip.h"

using namespace kmc_dev;

struct kmc_ctx {
  int device = -1;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  bool timing = false;
  hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr, ev_c0 = nullptr, ev_c1 = nullptr, ev_t0 = nullptr, ev_t1 = nullptr;
  std::string last_error;
  hipDeviceProp_t prop;
  int blocks_per_cu = 0;  // 0 = default
  int ppt = 0;            // 0 = default
  int force_tier = -1;
  // out-of-range counter (f64 path)
  unsigned long long* d_counter = nullptr;
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
  void* d_traj = nullptr; // segment tables of the N-knot trajectory kernels (16 x TrajSeg32 + 16 x TrajSeg64)
  void* h_traj = nullptr; // pinned staging of the same size
  hipEvent_t ev_traj = nullptr;  // last upload from h_traj has completed
  bool traj_in_flight = false;
  void* d_tmp = nullptr; // grow-only scratch for the f64 / batch host paths
  size_t tmp_cap = 0;
};

namespace {

// measured best on MI355X (profiles/r01_tune.csv): one wave per workgroup, one point per lane, one tile per workgroup
constexpr int kLaunchBlock = 64;
constexpr int kDefaultPpt = 1;
constexpr uint64_t kHostChunkPoints = 1ull << 21;  // 32 MiB per direction per pipeline slot

int fail_hip(kmc_ctx* c, hipError_t e, const char* what) {
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

int pick_tier(const kmc_ctx* c, const kmc_frame_params* p, uint32_t n) {
  if (c->force_tier >= 0 && c->force_tier <= 2) return c->force_tier;
  double theta_max = 0.0;
  for (uint32_t i = 0; i < n; ++i) {
    const double* f = p[i].twist;
    const double phi = std::sqrt(f[3] * f[3] + f[4] * f[4] + f[5] * f[5]);
    const double smax = std::fmax(std::fabs(p[i].x_req), std::fabs(1.0 - p[i].x_req));  // frac in [0,1]
    theta_max = std::fmax(theta_max, phi * smax);
  }
  if (!(theta_max <= 1.0)) return kTrig;  // also catches NaN
  return theta_max <= 0.25 ? kSeries3 : kSeries5;
}

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

// Device-resident buffers: distance (in points, < 64) from the last 1 KiB boundary to the start of the OUTPUT.  The kernels are
// launched on pointers moved back by that much with the first `head` indices dead, so that every tile stores whole aligned
// lines whatever 16-byte-aligned address the caller passes ("alignment").
uint32_t head_of(const void* out, int mem_kind) {
  return mem_kind == KMC_MEM_DEVICE ? (uint32_t)(((uintptr_t)out >> 4) & 63u) : 0u;
}

bool params_ok(const kmc_frame_params* p) {
  for (int i = 0; i < 6; ++i)
    if (!std::isfinite(p->twist[i])) return false;
  return std::isfinite(p->x_req);
}

int grid_for(const kmc_ctx* c, uint64_t n_tiles) {
  // default: one tile per workgroup -- the hardware dispatcher streaming 64-point tiles beats a persistent grid-stride loop
  // (6.8 vs 5.2-5.8 TB/s, profiles/r01_tune.csv); blocks_per_cu > 0 caps the grid instead (in units of 256 threads per CU).
  const uint64_t cap = c->blocks_per_cu > 0 ? (uint64_t)c->prop.multiProcessorCount * c->blocks_per_cu * (kBlock / kLaunchBlock) : 0x7fffffffull;
  return (int)std::max<uint64_t>(1, std::min<uint64_t>(n_tiles, cap));
}

int ppt_of(const kmc_ctx* c) {
  const int p = c->ppt > 0 ? c->ppt : kDefaultPpt;
  return (p == 1 || p == 2 || p == 4 || p == 8) ? p : kDefaultPpt;
}

// ---- template dispatch ---------------------------------------------------------------------------
template <int TIER, int PPT>
void launch_frame_tp(hipStream_t s, int grid, const v4f* in, v4f* out, uint64_t n, const FrameRec& f, uint32_t head) {
  hipLaunchKernelGGL((deskew_frame_f32<TIER, PPT, kPolicyDefault, false, kLaunchBlock>), dim3(grid), dim3(kLaunchBlock), 0, s, in, out, n, f, head);
}
template <int TIER>
void launch_frame_t(int ppt, hipStream_t s, int grid, const v4f* in, v4f* out, uint64_t n, const FrameRec& f, uint32_t head) {
  switch (ppt) {
    case 1: launch_frame_tp<TIER, 1>(s, grid, in, out, n, f, head); break;
    case 2: launch_frame_tp<TIER, 2>(s, grid, in, out, n, f, head); break;
    case 8: launch_frame_tp<TIER, 8>(s, grid, in, out, n, f, head); break;
    default: launch_frame_tp<TIER, 4>(s, grid, in, out, n, f, head); break;
  }
}
// in / out / n are the caller's; `head` dead points are put in front (pointers moved back, n grown) -- see head_of()
void launch_frame(const kmc_ctx* c, hipStream_t s, int tier, const v4f* in, v4f* out, uint64_t n, const FrameRec& f, uint32_t head = 0) {
  const int ppt = ppt_of(c);
  in -= head;
  out -= head;
  n += head;
  const uint64_t n_tiles = (n + (uint64_t)kLaunchBlock * ppt - 1) / ((uint64_t)kLaunchBlock * ppt);
  const int grid = grid_for(c, n_tiles);
  switch (tier) {
    case kSeries3: launch_frame_t<kSeries3>(ppt, s, grid, in, out, n, f, head); break;
    case kSeries5: launch_frame_t<kSeries5>(ppt, s, grid, in, out, n, f, head); break;
    default: launch_frame_t<kTrig>(ppt, s, grid, in, out, n, f, head); break;
  }
}

template <int TIER, int PPT>
void launch_batch_tp(hipStream_t s, int grid, const v4f* in, v4f* out, const BatchRec* recs, const uint2* tiles,
                     uint32_t nf, uint64_t n, uint32_t* idx, uint32_t head) {
  if (idx)
    hipLaunchKernelGGL((deskew_batch_f32<TIER, PPT, kPolicyDefault, true, kLaunchBlock>), dim3(grid), dim3(kLaunchBlock), 0, s, in, out, recs, tiles, nf, n, idx, head);
  else
    hipLaunchKernelGGL((deskew_batch_f32<TIER, PPT, kPolicyDefault, false, kLaunchBlock>), dim3(grid), dim3(kLaunchBlock), 0, s, in, out, recs, tiles, nf, n, idx, head);
}
template <int TIER>
void launch_batch_t(int ppt, hipStream_t s, int grid, const v4f* in, v4f* out, const BatchRec* recs,
                    const uint2* tiles, uint32_t nf, uint64_t n, uint32_t* idx, uint32_t head) {
  switch (ppt) {
    case 1: launch_batch_tp<TIER, 1>(s, grid, in, out, recs, tiles, nf, n, idx, head); break;
    case 2: launch_batch_tp<TIER, 2>(s, grid, in, out, recs, tiles, nf, n, idx, head); break;
    case 8: launch_batch_tp<TIER, 8>(s, grid, in, out, recs, tiles, nf, n, idx, head); break;
    default: launch_batch_tp<TIER, 4>(s, grid, in, out, recs, tiles, nf, n, idx, head); break;
  }
}

int ensure_tmp(kmc_ctx* c, size_t bytes) {
  if (bytes <= c->tmp_cap) return KMC_OK;
  if (c->d_tmp) {
    KMC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    KMC_HIP_TRY(c, hipFree(c->d_tmp));
    c->d_tmp = nullptr;
    c->tmp_cap = 0;
  }
  KMC_HIP_TRY(c, hipMalloc(&c->d_tmp, bytes));
  c->tmp_cap = bytes;
  return KMC_OK;
}

int ensure_pipeline(kmc_ctx* c) {
  if (c->stage_cap) return KMC_OK;
  const size_t bytes = kHostChunkPoints * sizeof(v4f);
  for (int b = 0; b < 3; ++b) KMC_HIP_TRY(c, hipStreamCreateWithFlags(&c->pipe[b], hipStreamNonBlocking));
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

// ---- ring of table slots (batch tables and trajectory segment tables) ---------------------------------------------------
// slot_begin : picks the next slot, waits (host) until the kernels of its group from the previous lap are done, grows every
//              slot if `need` bytes do not fit;
// slot_upload: one H2D copy of the slot's pinned staging on the side stream, then a HOST wait for that tiny copy -- the
//              launch that follows has no cross-stream dependency, so back-to-back launches keep the ~2 us same-stream boundary;
// slot_end   : after the launch; records one "consumed" marker per group of launches on the compute stream.
int slot_begin(kmc_ctx* c, size_t need, int* slot_id_out) {
  const int slot_id = c->next_slot;
  const int group_id = slot_id / kmc_ctx::kSlotsPerGroup;
  c->next_slot = (c->next_slot + 1) % kmc_ctx::kTableSlots;
  if (slot_id % kmc_ctx::kSlotsPerGroup == 0 && c->group_busy[group_id]) {
    KMC_HIP_TRY(c, hipEventSynchronize(c->group_consumed[group_id]));
    c->group_busy[group_id] = false;
  }
  if (need > c->slots[slot_id].cap) {
    // grow EVERY slot at once (so that steady state never allocates again); slots may still be referenced by kernels in
    // flight: drain first
    KMC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    KMC_HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
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
  }
  *slot_id_out = slot_id;
  return KMC_OK;
}

int slot_upload(kmc_ctx* c, int slot_id, size_t bytes) {
  kmc_ctx::TableSlot& sl = c->slots[slot_id];
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
  }
  return KMC_OK;
}

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

// ---- N-knot trajectory: host pre-step (f64) -------------------------------------------------------------------------
struct TrajHost {
  uint32_t n_seg = 0;
  uint32_t r = 0;          // segment that contains requested_time
  double x_r = 0.0;        // position of requested_time inside segment r
  kmc_host::Twist f[kMaxSegments];
  kmc_host::Pose M[kMaxSegments];  // T(requested)^-1 * P_k   (identity for k == r)
  double t0[kMaxSegments], dur[kMaxSegments];
};

int build_trajectory(const double* times, const double* poses, uint32_t n_knots, double t_req, TrajHost* out) {
  if (!times || !poses || n_knots < 2 || n_knots - 1 > (uint32_t)kMaxSegments) return KMC_ERR_INVALID_ARG;
  for (uint32_t k = 0; k < n_knots; ++k)
    if (!std::isfinite(times[k]) || (k && !(times[k] > times[k - 1]))) return KMC_ERR_DEGENERATE;
  if (!(t_req >= times[0] && t_req <= times[n_knots - 1])) return KMC_ERR_TIME_OUT_OF_RANGE;
  out->n_seg = n_knots - 1;
  kmc_host::Pose P[kMaxSegments + 1];
  for (uint32_t k = 0; k < n_knots; ++k) P[k] = kmc_host::Pose::from_rt12(poses + 12 * k);
  for (uint32_t k = 0; k < out->n_seg; ++k) {
    if (!kmc_host::relative_twist(P[k], P[k + 1], &out->f[k])) return KMC_ERR_DEGENERATE;
    out->t0[k] = times[k];
    out->dur[k] = times[k + 1] - times[k];
  }
  uint32_t r = 0;
  while (r + 1 < out->n_seg && t_req >= times[r + 1]) ++r;
  out->r = r;
  out->x_r = (t_req - times[r]) / (times[r + 1] - times[r]);
  // T(requested)^-1 = Exp(-x_r f_r) * P_r^-1 ;  M_k = Exp(-x_r f_r) * (P_r^-1 P_k), relative pose formed with the
  // translation difference first (same care as relative_twist: Mercator-scale translations)
  const kmc_host::Pose back = kmc_host::se3_exp({(-out->x_r) * out->f[r].rho, (-out->x_r) * out->f[r].phi});
  kmc_host::Mat3 Lri;
  if (!kmc_host::inverse(P[r].L, &Lri)) return KMC_ERR_DEGENERATE;
  for (uint32_t k = 0; k < out->n_seg; ++k) {
    if (k == r) {
      out->M[k] = kmc_host::Pose::identity();
    } else {
      const kmc_host::Pose rel = {Lri * P[k].L, Lri * (P[k].t - P[r].t)};
      out->M[k] = back * rel;
    }
  }
  return KMC_OK;
}

void fill_rec64(const kmc_host::Twist& f, FrameRec64* r) {
  const kmc_host::Vec3 c1 = kmc_host::cross(f.phi, f.rho);
  const kmc_host::Vec3 c2 = kmc_host::cross(f.phi, c1);
  r->phi[0] = f.phi.x; r->phi[1] = f.phi.y; r->phi[2] = f.phi.z;
  r->rho[0] = f.rho.x; r->rho[1] = f.rho.y; r->rho[2] = f.rho.z;
  r->c1[0] = c1.x; r->c1[1] = c1.y; r->c1[2] = c1.z;
  r->c2[0] = c2.x; r->c2[1] = c2.y; r->c2[2] = c2.z;
  r->phi2 = kmc_host::dot(f.phi, f.phi);
}

// direction (cos, sin) of the knot azimuth alpha = pi - 2 pi c; exact on the quarter turns
void knot_direction(double c, float* ck, float* sk) {
  const double q = 4.0 * c;
  if (q == std::floor(q) && q >= 0.0 && q <= 4.0) {
    static const float kc[5] = {-1.f, 0.f, 1.f, 0.f, -1.f};   // cos(pi - 2 pi c) at c = 0, 1/4, 1/2, 3/4, 1
    static const float ks[5] = {0.f, 1.f, 0.f, -1.f, -0.f};   // sin(pi - 2 pi c)
    *ck = kc[(int)q];
    *sk = ks[(int)q];
    return;
  }
  const double alpha = 3.14159265358979323846 - 2.0 * 3.14159265358979323846 * c;
  *ck = (float)std::cos(alpha);
  *sk = (float)std::sin(alpha);
}

// device records of one frame's trajectory (f32): one TrajSeg32 per segment
void fill_traj_segs(const TrajHost& th, double stamp_start, double stamp_end, TrajSeg32* segs) {
  const double scan = stamp_end - stamp_start;
  for (uint32_t k = 0; k < th.n_seg; ++k) {
    TrajSeg32& r = segs[k];
    kmc_frame_params fp;
    fp.twist[0] = th.f[k].rho.x; fp.twist[1] = th.f[k].rho.y; fp.twist[2] = th.f[k].rho.z;
    fp.twist[3] = th.f[k].phi.x; fp.twist[4] = th.f[k].phi.y; fp.twist[5] = th.f[k].phi.z;
    fp.x_req = 0.0;
    fill_rec(fp, &r);  // phi, |phi|^2, rho, c1, c2 (s0 overwritten below)
    const double ck = (th.t0[k] - stamp_start) / scan;   // scan fraction of the segment's start knot
    const double g = scan / th.dur[k];
    const double a = (k == th.r) ? th.x_r : 0.0;
    r.g = (float)g;
    r.s0 = (float)((0.5 - ck) * g - a);                  // 2 knots on the scan: (0.5 - 0) * 1 - x_req, as kmc_hip_deskew_f32
    r.knot_c = (float)ck;
    r.m00 = (float)th.M[k].L.m[0][0]; r.m01 = (float)th.M[k].L.m[0][1]; r.m02 = (float)th.M[k].L.m[0][2]; r.tx = (float)th.M[k].t.x;
    r.m10 = (float)th.M[k].L.m[1][0]; r.m11 = (float)th.M[k].L.m[1][1]; r.m12 = (float)th.M[k].L.m[1][2]; r.ty = (float)th.M[k].t.y;
    r.m20 = (float)th.M[k].L.m[2][0]; r.m21 = (float)th.M[k].L.m[2][1]; r.m22 = (float)th.M[k].L.m[2][2]; r.tz = (float)th.M[k].t.z;
    knot_direction(ck, &r.knot_cos, &r.knot_sin);
    r.flags = (k == th.r ? kSegIdentity : 0u) | (ck <= 0.0 ? kKnotAlwaysGe : 0u) | (ck > 1.0 ? kKnotNeverGe : 0u);
  }
}

// coarse[c] = {frame that owns point c * chunk (empty frames skipped), split}; coarse[n_chunks].x = frame of the last point.
// All positions are VIRTUAL: `head` dead points precede the batch (frame 0 owns them), n_virtual = n + head.
void build_coarse(const uint64_t* offsets, uint32_t n_frames, uint64_t n_virtual, uint32_t head, uint2* h_coarse) {
  const uint64_t chunk = 1ull << kChunkShift;
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

constexpr size_t kTrajBytes = kMaxSegments * (sizeof(TrajSeg32) + sizeof(TrajSeg64));

int ensure_traj(kmc_ctx* c) {
  if (!c->d_traj) {
    KMC_HIP_TRY(c, hipMalloc(&c->d_traj, kTrajBytes));
    KMC_HIP_TRY(c, hipHostMalloc(&c->h_traj, kTrajBytes, hipHostMallocDefault));
    KMC_HIP_TRY(c, hipEventCreateWithFlags(&c->ev_traj, hipEventDisableTiming));
  }
  if (c->traj_in_flight) {  // the pinned staging is about to be overwritten
    KMC_HIP_TRY(c, hipEventSynchronize(c->ev_traj));
    c->traj_in_flight = false;
  }
  return KMC_OK;
}

// stage `bytes` of segment records at byte offset `off` of the table and upload them on the compute stream
int upload_traj(kmc_ctx* c, const void* src, size_t off, size_t bytes) {
  std::memcpy((char*)c->h_traj + off, src, bytes);
  KMC_HIP_TRY(c, hipMemcpyAsync((char*)c->d_traj + off, (char*)c->h_traj + off, bytes, hipMemcpyHostToDevice, c->stream));
  KMC_HIP_TRY(c, hipEventRecord(c->ev_traj, c->stream));
  c->traj_in_flight = true;
  return KMC_OK;
}

int traj_tier(const kmc_ctx* c, const TrajHost& th, double span_lo, double span_hi) {
  if (c->force_tier >= 0 && c->force_tier <= 2) return c->force_tier;
  // |s| <= 1 inside a segment except for the anchor segment (|x - x_r| <= 1 as well); the scan may stick out of the first /
  // last segment by at most the knots' coverage, which build_trajectory() / the callers have verified -> bound by 1.
  (void)span_lo; (void)span_hi;
  double theta_max = 0.0;
  for (uint32_t k = 0; k < th.n_seg; ++k) theta_max = std::fmax(theta_max, kmc_host::norm(th.f[k].phi));
  if (!(theta_max <= 1.0)) return kTrig;
  return theta_max <= 0.25 ? kSeries3 : kSeries5;
}

}  // namespace

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
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking);
  for (auto& sl : c->slots)
    if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.uploaded, hipEventDisableTiming);
  for (auto& ev : c->group_consumed)
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  if (e == hipSuccess) e = hipMalloc((void**)&c->d_counter, sizeof(unsigned long long));
  if (e != hipSuccess) {
    (void)hipGetLastError();
    kmc_hip_destroy(c);
    return KMC_ERR_NO_DEVICE;
  }
  c->stream = c->own_stream;
  *out = c;
  return KMC_OK;
}

void kmc_hip_destroy(kmc_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
  for (int b = 0; b < 3; ++b)
    if (c->pipe[b]) { (void)hipStreamSynchronize(c->pipe[b]); (void)hipStreamDestroy(c->pipe[b]); }
  for (int b = 0; b < kmc_ctx::kPipeSlots; ++b) {
    if (c->d_stage_in[b]) (void)hipFree(c->d_stage_in[b]);
    if (c->d_stage_out[b]) (void)hipFree(c->d_stage_out[b]);
    if (c->ev_h2d[b]) (void)hipEventDestroy(c->ev_h2d[b]);
    if (c->ev_kernel[b]) (void)hipEventDestroy(c->ev_kernel[b]);
    if (c->ev_d2h[b]) (void)hipEventDestroy(c->ev_d2h[b]);
  }
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
  hipEvent_t evs[] = {c->ev_k0, c->ev_k1, c->ev_c0, c->ev_c1, c->ev_t0, c->ev_t1};
  for (hipEvent_t ev : evs)
    if (ev) (void)hipEventDestroy(ev);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

int kmc_hip_set_stream(kmc_ctx* c, void* hip_stream) {
  if (!c) return KMC_ERR_INVALID_ARG;
  c->stream = (hipStream_t)hip_stream;  // literally: NULL is HIP's legacy default stream
  return KMC_OK;
}

int kmc_hip_use_own_stream(kmc_ctx* c) {
  if (!c) return KMC_ERR_INVALID_ARG;
  c->stream = c->own_stream;
  return KMC_OK;
}

int kmc_hip_synchronize(kmc_ctx* c) {
  if (!c) return KMC_ERR_INVALID_ARG;
  KMC_HIP_TRY(c, hipStreamSynchronize(c->stream));
  return KMC_OK;
}

int kmc_hip_enable_timing(kmc_ctx* c, int enabled) {
  if (!c) return KMC_ERR_INVALID_ARG;
  c->timing = enabled != 0;
  return KMC_OK;
}

const char* kmc_hip_last_error(kmc_ctx* c) { return c ? c->last_error.c_str() : "null ctx"; }

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
  return KMC_OK;
}

int kmc_hip_set_launch_config(kmc_ctx* c, int blocks_per_cu, int points_per_thread) {
  if (!c || blocks_per_cu < 0 || blocks_per_cu > 64) return KMC_ERR_INVALID_ARG;
  if (!(points_per_thread == 0 || points_per_thread == 1 || points_per_thread == 2 || points_per_thread == 4 ||
        points_per_thread == 8))
    return KMC_ERR_INVALID_ARG;
  c->blocks_per_cu = blocks_per_cu;
  c->ppt = points_per_thread;
  return KMC_OK;
}

int kmc_hip_force_tier(kmc_ctx* c, int tier) {
  if (!c || tier < -1 || tier > 2) return KMC_ERR_INVALID_ARG;
  c->force_tier = tier;
  return KMC_OK;
}

int kmc_hip_timer_begin(kmc_ctx* c) {
  if (!c) return KMC_ERR_INVALID_ARG;
  KMC_HIP_TRY(c, hipEventRecord(c->ev_t0, c->stream));
  return KMC_OK;
}

int kmc_hip_timer_end(kmc_ctx* c, float* elapsed_ms) {
  if (!c || !elapsed_ms) return KMC_ERR_INVALID_ARG;
  KMC_HIP_TRY(c, hipEventRecord(c->ev_t1, c->stream));
  KMC_HIP_TRY(c, hipEventSynchronize(c->ev_t1));
  KMC_HIP_TRY(c, hipEventElapsedTime(elapsed_ms, c->ev_t0, c->ev_t1));
  return KMC_OK;
}

int kmc_hip_host_alloc(kmc_ctx* c, size_t bytes, void** out) {
  if (!c || !out) return KMC_ERR_INVALID_ARG;
  *out = nullptr;
  if (bytes == 0) return KMC_OK;
  KMC_HIP_TRY(c, hipSetDevice(c->device));
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

int kmc_make_frame_poses(const kmc_oxts* o_nm1, const kmc_oxts* o_n, const kmc_oxts* o_np1, double stamp_start,
                         double stamp_end, double T_start_out[12], double T_end_out[12]) {
  int rc = kmc_interpolate_trajectory(o_nm1, o_n, stamp_start, T_start_out);
  if (rc != KMC_OK) return rc;
  return kmc_interpolate_trajectory(o_n, o_np1, stamp_end, T_end_out);
}

// ---- hot path: single frame, f32 -----------------------------------------------------------------
int kmc_hip_deskew_f32(kmc_ctx* c, const float* xyzi_in, float* xyzi_out, uint64_t n, const kmc_frame_params* params,
                       int mem_kind, kmc_stats* st) {
  if (!c || !params || (n && (!xyzi_in || !xyzi_out))) return KMC_ERR_INVALID_ARG;
  if (mem_kind != KMC_MEM_HOST && mem_kind != KMC_MEM_DEVICE) return KMC_ERR_INVALID_ARG;
  if (((uintptr_t)xyzi_in | (uintptr_t)xyzi_out) & 15u) return KMC_ERR_INVALID_ARG;
  if (!params_ok(params)) return KMC_ERR_INVALID_ARG;
  if (!(params->x_req >= 0.0 && params->x_req <= 1.0)) return KMC_ERR_TIME_OUT_OF_RANGE;
  if (st) std::memset(st, 0, sizeof(*st));
  KMC_HIP_TRY(c, hipSetDevice(c->device));
  const int tier = pick_tier(c, params, 1);
  FrameRec f;
  std::memset(&f, 0, sizeof(f));
  fill_rec(*params, &f);
  if (st) { st->n_points = n; st->variant = (uint32_t)tier; }
  if (n == 0) return KMC_OK;
  CallTimer tm(c);
  if (mem_kind == KMC_MEM_DEVICE) {
    if (tm.begin_call() || tm.begin_kernel()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
    launch_frame(c, c->stream, tier, (const v4f*)xyzi_in, (v4f*)xyzi_out, n, f, head_of(xyzi_out, mem_kind));
    KMC_HIP_TRY(c, hipGetLastError());
    if (tm.end_kernel()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
    if (st) st->n_launches = 1;
    return tm.end_call(st);
  }
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
    launch_frame(c, c->pipe[1], tier, (const v4f*)c->d_stage_in[b], (v4f*)c->d_stage_out[b], m, f);
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

// ---- hot path: batch of frames, f32 ---------------------------------------------------------------
int kmc_hip_deskew_batch_f32(kmc_ctx* c, const float* xyzi_in, float* xyzi_out, const uint64_t* offsets, uint32_t n_frames,
                             const kmc_frame_params* params, uint32_t* frame_idx_out, int mem_kind, kmc_stats* st) {
  if (!c || !offsets || (n_frames && !params)) return KMC_ERR_INVALID_ARG;
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
  if (((uintptr_t)xyzi_in | (uintptr_t)xyzi_out) & 15u) return KMC_ERR_INVALID_ARG;
  KMC_HIP_TRY(c, hipSetDevice(c->device));
  const int tier = pick_tier(c, params, n_frames);
  if (st) { st->n_points = n; st->variant = (uint32_t)tier; }
  if (n == 0) return KMC_OK;

  const int ppt = ppt_of(c);
  const uint32_t head = head_of(xyzi_out, mem_kind);  // dead points in front: tiles are cut on 1 KiB lines of the output
  const uint64_t nv = n + head;                       // virtual size; every offset below is shifted by `head` too
  const uint64_t tile = (uint64_t)kLaunchBlock * ppt;
  const uint64_t n_tiles = (nv + tile - 1) / tile;
  const uint64_t chunk = 1ull << kChunkShift;
  const uint64_t n_chunks = (nv + chunk - 1) / chunk;
  const uint64_t n_coarse = n_chunks + 1;

  const size_t recs_bytes = ((size_t)n_frames * sizeof(BatchRec) + 255) & ~(size_t)255;
  const size_t need = recs_bytes + (size_t)n_coarse * sizeof(uint2);
  int slot_id = 0;
  {
    const int rc_slot = slot_begin(c, need, &slot_id);
    if (rc_slot != KMC_OK) return rc_slot;
  }
  kmc_ctx::TableSlot& sl = c->slots[slot_id];
  BatchRec* h_recs = reinterpret_cast<BatchRec*>(sl.h_buf);
  uint2* h_coarse = reinterpret_cast<uint2*>(sl.h_buf + recs_bytes);
  const BatchRec* d_recs = reinterpret_cast<const BatchRec*>(sl.d_buf);
  const uint2* d_coarse = reinterpret_cast<const uint2*>(sl.d_buf + recs_bytes);
  for (uint32_t f = 0; f < n_frames; ++f) {
    BatchRec* r = &h_recs[f];
    fill_rec(params[f], r);
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
  const int grid = grid_for(c, n_tiles);
  uint32_t* v_idx = d_idx ? d_idx - head : nullptr;
  switch (tier) {
    case kSeries3: launch_batch_t<kSeries3>(ppt, c->stream, grid, d_in - head, d_out - head, d_recs, d_coarse, n_frames, nv, v_idx, head); break;
    case kSeries5: launch_batch_t<kSeries5>(ppt, c->stream, grid, d_in - head, d_out - head, d_recs, d_coarse, n_frames, nv, v_idx, head); break;
    default: launch_batch_t<kTrig>(ppt, c->stream, grid, d_in - head, d_out - head, d_recs, d_coarse, n_frames, nv, v_idx, head); break;
  }
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
  if (st) st->n_launches = 1;
  return tm.end_call(st);
}

// ---- f64 Eigen-layout path ------------------------------------------------------------------------
int kmc_hip_deskew_f64cols(kmc_ctx* c, const double* x, const double* y, const double* z, const double* w, const double* stamps,
                           uint64_t n, double stamp_start, double stamp_end, const kmc_frame_params* params, double* ox,
                           double* oy, double* oz, double* ow, int mem_kind, kmc_stats* st) {
  if (!c || !params) return KMC_ERR_INVALID_ARG;
  if (n && (!x || !y || !z || !stamps || !ox || !oy || !oz)) return KMC_ERR_INVALID_ARG;
  if (mem_kind != KMC_MEM_HOST && mem_kind != KMC_MEM_DEVICE) return KMC_ERR_INVALID_ARG;
  if (!(stamp_start < stamp_end)) return KMC_ERR_DEGENERATE;
  if (!params_ok(params)) return KMC_ERR_INVALID_ARG;
  if (!(params->x_req >= 0.0 && params->x_req <= 1.0)) return KMC_ERR_TIME_OUT_OF_RANGE;
  if (st) std::memset(st, 0, sizeof(*st));
  if (st) { st->n_points = n; st->variant = 3; }
  if (n == 0) return KMC_OK;
  KMC_HIP_TRY(c, hipSetDevice(c->device));

  FrameRec64 f;
  const kmc_host::Vec3 rho = {params->twist[0], params->twist[1], params->twist[2]};
  const kmc_host::Vec3 phi = {params->twist[3], params->twist[4], params->twist[5]};
  const kmc_host::Vec3 c1 = kmc_host::cross(phi, rho);
  const kmc_host::Vec3 c2 = kmc_host::cross(phi, c1);
  f.phi[0] = phi.x; f.phi[1] = phi.y; f.phi[2] = phi.z;
  f.rho[0] = rho.x; f.rho[1] = rho.y; f.rho[2] = rho.z;
  f.c1[0] = c1.x; f.c1[1] = c1.y; f.c1[2] = c1.z;
  f.c2[0] = c2.x; f.c2[1] = c2.y; f.c2[2] = c2.z;
  f.phi2 = kmc_host::dot(phi, phi);
  f.x_req = params->x_req;
  f.t_start = stamp_start;
  f.t_end = stamp_end;
  f.dur = stamp_end - stamp_start;

  const double *dx = x, *dy = y, *dz = z, *dw = w, *ds = stamps;
  double *dox = ox, *doy = oy, *doz = oz, *dow = ow;
  const size_t col = n * sizeof(double);
  if (mem_kind == KMC_MEM_HOST) {
    int rc = ensure_tmp(c, 9 * col);
    if (rc != KMC_OK) return rc;
    double* base = (double*)c->d_tmp;
    double* cols[9];
    for (int i = 0; i < 9; ++i) cols[i] = base + (size_t)i * n;
    // an Eigen::MatrixX4d is ONE column-major block: x, y, z, w follow each other -> one copy instead of four (each
    // pageable copy has a fixed cost of tens of microseconds, which is what a 123 k-point frame is made of)
    if (y == x + n && z == y + n && (!w || w == z + n)) {
      KMC_HIP_TRY(c, hipMemcpyAsync(cols[0], x, (w ? 4 : 3) * col, hipMemcpyHostToDevice, c->stream));
    } else {
      KMC_HIP_TRY(c, hipMemcpyAsync(cols[0], x, col, hipMemcpyHostToDevice, c->stream));
      KMC_HIP_TRY(c, hipMemcpyAsync(cols[1], y, col, hipMemcpyHostToDevice, c->stream));
      KMC_HIP_TRY(c, hipMemcpyAsync(cols[2], z, col, hipMemcpyHostToDevice, c->stream));
      if (w) KMC_HIP_TRY(c, hipMemcpyAsync(cols[3], w, col, hipMemcpyHostToDevice, c->stream));
    }
    KMC_HIP_TRY(c, hipMemcpyAsync(cols[4], stamps, col, hipMemcpyHostToDevice, c->stream));
    dx = cols[0]; dy = cols[1]; dz = cols[2]; dw = w ? cols[3] : nullptr; ds = cols[4];
    dox = cols[5]; doy = cols[6]; doz = cols[7]; dow = ow ? cols[8] : nullptr;
  }
  CallTimer tm(c);
  if (tm.begin_call()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
  KMC_HIP_TRY(c, hipMemsetAsync(c->d_counter, 0, sizeof(unsigned long long), c->stream));
  if (tm.begin_kernel()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
  const int grid = grid_for(c, (n + 127) / 128);  // one wave per workgroup, two points per lane
  hipLaunchKernelGGL(deskew_f64cols, dim3(grid), dim3(64), 0, c->stream, dx, dy, dz, dw, ds, n, f, dox, doy, doz, dow, c->d_counter);
  KMC_HIP_TRY(c, hipGetLastError());
  if (tm.end_kernel()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
  unsigned long long bad = 0;
  KMC_HIP_TRY(c, hipMemcpyAsync(&bad, c->d_counter, sizeof(bad), hipMemcpyDeviceToHost, c->stream));
  if (mem_kind == KMC_MEM_HOST) {
    if (oy == ox + n && oz == oy + n && (!ow || ow == oz + n)) {  // one column-major block again
      KMC_HIP_TRY(c, hipMemcpyAsync(ox, dox, (ow ? 4 : 3) * col, hipMemcpyDeviceToHost, c->stream));
    } else {
      KMC_HIP_TRY(c, hipMemcpyAsync(ox, dox, col, hipMemcpyDeviceToHost, c->stream));
      KMC_HIP_TRY(c, hipMemcpyAsync(oy, doy, col, hipMemcpyDeviceToHost, c->stream));
      KMC_HIP_TRY(c, hipMemcpyAsync(oz, doz, col, hipMemcpyDeviceToHost, c->stream));
      if (ow) KMC_HIP_TRY(c, hipMemcpyAsync(ow, dow, col, hipMemcpyDeviceToHost, c->stream));
    }
  }
  KMC_HIP_TRY(c, hipStreamSynchronize(c->stream));  // the out-of-range verdict is part of the call's result
  if (st) { st->n_launches = 1; st->n_out_of_range = bad; }
  int rc = tm.end_call(st);
  if (rc != KMC_OK) return rc;
  return bad ? KMC_ERR_TIME_OUT_OF_RANGE : KMC_OK;
}

int kmc_hip_pseudo_timestamps_f64(kmc_ctx* c, const double* x, const double* y, uint64_t n, double scan_start, double scan_end,
                                  double* stamps_out, int mem_kind) {
  if (!c || (n && (!x || !y || !stamps_out))) return KMC_ERR_INVALID_ARG;
  if (mem_kind != KMC_MEM_HOST && mem_kind != KMC_MEM_DEVICE) return KMC_ERR_INVALID_ARG;
  if (n == 0) return KMC_OK;
  KMC_HIP_TRY(c, hipSetDevice(c->device));
  const size_t col = n * sizeof(double);
  const double *dx = x, *dy = y;
  double* dout = stamps_out;
  if (mem_kind == KMC_MEM_HOST) {
    int rc = ensure_tmp(c, 3 * col);
    if (rc != KMC_OK) return rc;
    double* base = (double*)c->d_tmp;
    if (y == x + n) {  // two adjacent columns of one Eigen matrix: one copy
      KMC_HIP_TRY(c, hipMemcpyAsync(base, x, 2 * col, hipMemcpyHostToDevice, c->stream));
    } else {
      KMC_HIP_TRY(c, hipMemcpyAsync(base, x, col, hipMemcpyHostToDevice, c->stream));
      KMC_HIP_TRY(c, hipMemcpyAsync(base + n, y, col, hipMemcpyHostToDevice, c->stream));
    }
    dx = base; dy = base + n; dout = base + 2 * n;
  }
  const int grid = grid_for(c, (n + 127) / 128);  // one wave per workgroup, two points per lane
  hipLaunchKernelGGL(pseudo_timestamps_f64, dim3(grid), dim3(64), 0, c->stream, dx, dy, n, scan_start, scan_end, dout);
  KMC_HIP_TRY(c, hipGetLastError());
  if (mem_kind == KMC_MEM_HOST) {
    KMC_HIP_TRY(c, hipMemcpyAsync(stamps_out, dout, col, hipMemcpyDeviceToHost, c->stream));
    KMC_HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  return KMC_OK;
}

// ---- N-knot trajectory entry points ---------------------------------------------------------------------------------
int kmc_hip_deskew_traj_f32(kmc_ctx* c, const float* xyzi_in, float* xyzi_out, uint64_t n, const double* knot_times,
                            const double* knot_poses, uint32_t n_knots, double stamp_start, double stamp_end, double requested_time,
                            uint32_t* bracket_idx_out, int mem_kind, kmc_stats* st) {
  if (!c || (n && (!xyzi_in || !xyzi_out))) return KMC_ERR_INVALID_ARG;
  if (mem_kind != KMC_MEM_HOST && mem_kind != KMC_MEM_DEVICE) return KMC_ERR_INVALID_ARG;
  if (((uintptr_t)xyzi_in | (uintptr_t)xyzi_out) & 15u) return KMC_ERR_INVALID_ARG;
  if (!(stamp_start < stamp_end)) return KMC_ERR_DEGENERATE;
  if (st) std::memset(st, 0, sizeof(*st));
  TrajHost th;
  int rc = build_trajectory(knot_times, knot_poses, n_knots, requested_time, &th);
  if (rc != KMC_OK) return rc;
  // every point stamp lies in [stamp_start, stamp_end]: the trajectory has to cover the scan
  if (!(knot_times[0] <= stamp_start && stamp_end <= knot_times[n_knots - 1])) return KMC_ERR_TIME_OUT_OF_RANGE;
  if (!(requested_time >= stamp_start && requested_time <= stamp_end)) return KMC_ERR_TIME_OUT_OF_RANGE;
  KMC_HIP_TRY(c, hipSetDevice(c->device));
  const int tier = traj_tier(c, th, stamp_start, stamp_end);
  if (st) { st->n_points = n; st->variant = (uint32_t)tier; }
  if (n == 0) return KMC_OK;

  TrajSeg32 segs[kMaxSegments];
  std::memset(segs, 0, sizeof(segs));
  fill_traj_segs(th, stamp_start, stamp_end, segs);
  int slot_id = 0;
  rc = slot_begin(c, sizeof(segs), &slot_id);
  if (rc != KMC_OK) return rc;
  std::memcpy(c->slots[slot_id].h_buf, segs, th.n_seg * sizeof(TrajSeg32));
  rc = slot_upload(c, slot_id, th.n_seg * sizeof(TrajSeg32));
  if (rc != KMC_OK) return rc;

  const v4f* d_in = (const v4f*)xyzi_in;
  v4f* d_out = (v4f*)xyzi_out;
  uint32_t* d_idx = bracket_idx_out;
  if (mem_kind == KMC_MEM_HOST) {
    const size_t pts = n * sizeof(v4f);
    rc = ensure_tmp(c, 2 * pts + (bracket_idx_out ? n * sizeof(uint32_t) : 0));
    if (rc != KMC_OK) return rc;
    d_in = (const v4f*)c->d_tmp;
    d_out = (v4f*)((char*)c->d_tmp + pts);
    d_idx = bracket_idx_out ? (uint32_t*)((char*)c->d_tmp + 2 * pts) : nullptr;
    KMC_HIP_TRY(c, hipMemcpyAsync((void*)d_in, xyzi_in, pts, hipMemcpyHostToDevice, c->stream));
  }
  CallTimer tm(c);
  if (tm.begin_call()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
  if (tm.begin_kernel()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
  static_assert(kLaunchBlock == 64, "deskew_traj_f32 is a one-wave-per-workgroup kernel");
  const uint32_t head = head_of(xyzi_out, mem_kind);
  const uint64_t nv = n + head;
  const int grid = grid_for(c, (nv + 63) / 64);
  const TrajSeg32* d_segs = (const TrajSeg32*)c->slots[slot_id].d_buf;
  uint32_t* v_idx = d_idx ? d_idx - head : nullptr;
#define KMC_LAUNCH_TRAJ(T)                                                                                                          \
  do {                                                                                                                              \
    if (d_idx) hipLaunchKernelGGL((deskew_traj_f32<T, kPolicyDefault, true>), dim3(grid), dim3(64), 0, c->stream, d_in - head, d_out - head, nv, d_segs, th.n_seg, v_idx, head); \
    else hipLaunchKernelGGL((deskew_traj_f32<T, kPolicyDefault, false>), dim3(grid), dim3(64), 0, c->stream, d_in - head, d_out - head, nv, d_segs, th.n_seg, v_idx, head);      \
  } while (0)
  switch (tier) {
    case kSeries3: KMC_LAUNCH_TRAJ(kSeries3); break;
    case kSeries5: KMC_LAUNCH_TRAJ(kSeries5); break;
    default: KMC_LAUNCH_TRAJ(kTrig); break;
  }
#undef KMC_LAUNCH_TRAJ
  KMC_HIP_TRY(c, hipGetLastError());
  rc = slot_end(c, slot_id);
  if (rc != KMC_OK) return rc;
  if (tm.end_kernel()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
  if (mem_kind == KMC_MEM_HOST) {
    KMC_HIP_TRY(c, hipMemcpyAsync(xyzi_out, d_out, n * sizeof(v4f), hipMemcpyDeviceToHost, c->stream));
    if (bracket_idx_out) KMC_HIP_TRY(c, hipMemcpyAsync(bracket_idx_out, d_idx, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    KMC_HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  if (st) st->n_launches = 1;
  return tm.end_call(st);
}

// ---- batched N-knot trajectories ---------------------------------------------------------------------
int kmc_hip_deskew_traj_batch_f32(kmc_ctx* c, const float* xyzi_in, float* xyzi_out, const uint64_t* offsets, uint32_t n_frames,
                                  const kmc_traj_frame* frames, uint32_t* frame_idx_out, uint32_t* bracket_idx_out, int mem_kind,
                                  kmc_stats* st) {
  if (!c || !offsets || (n_frames && !frames)) return KMC_ERR_INVALID_ARG;
  if (mem_kind != KMC_MEM_HOST && mem_kind != KMC_MEM_DEVICE) return KMC_ERR_INVALID_ARG;
  if (st) std::memset(st, 0, sizeof(*st));
  if (offsets[0] != 0) return KMC_ERR_INVALID_ARG;
  for (uint32_t f = 0; f < n_frames; ++f)
    if (offsets[f + 1] < offsets[f]) return KMC_ERR_INVALID_ARG;
  const uint64_t n = n_frames ? offsets[n_frames] : 0;
  if (n && (!xyzi_in || !xyzi_out)) return KMC_ERR_INVALID_ARG;
  if (((uintptr_t)xyzi_in | (uintptr_t)xyzi_out) & 15u) return KMC_ERR_INVALID_ARG;
  KMC_HIP_TRY(c, hipSetDevice(c->device));

  // host pre-step per frame (f64): segments, anchors, M_k
  std::vector<TrajHost> th(n_frames);
  uint32_t seg_stride = 1;  // slots per frame in the segment table = the longest trajectory of the batch
  int tier = kSeries3;
  for (uint32_t f = 0; f < n_frames; ++f) {
    const kmc_traj_frame& fr = frames[f];
    if (!(fr.stamp_start < fr.stamp_end)) return KMC_ERR_DEGENERATE;
    int rc = build_trajectory(fr.knot_times, fr.knot_poses, fr.n_knots, fr.requested_time, &th[f]);
    if (rc != KMC_OK) return rc;
    if (!(fr.knot_times[0] <= fr.stamp_start && fr.stamp_end <= fr.knot_times[fr.n_knots - 1])) return KMC_ERR_TIME_OUT_OF_RANGE;
    if (!(fr.requested_time >= fr.stamp_start && fr.requested_time <= fr.stamp_end)) return KMC_ERR_TIME_OUT_OF_RANGE;
    seg_stride = std::max(seg_stride, th[f].n_seg);
    tier = std::max(tier, traj_tier(c, th[f], fr.stamp_start, fr.stamp_end));
  }
  if (c->force_tier >= 0 && c->force_tier <= 2) tier = c->force_tier;
  if (st) { st->n_points = n; st->variant = (uint32_t)tier; }
  if (n == 0) return KMC_OK;

  const uint32_t head = head_of(xyzi_out, mem_kind);
  const uint64_t nv = n + head;
  const uint64_t chunk = 1ull << kChunkShift;
  const uint64_t n_coarse = (nv + chunk - 1) / chunk + 1;
  const size_t frecs_bytes = ((size_t)n_frames * sizeof(TrajFrameRec) + 255) & ~(size_t)255;
  const size_t segs_bytes = ((size_t)n_frames * seg_stride * sizeof(TrajSeg32) + 255) & ~(size_t)255;
  const size_t need = frecs_bytes + segs_bytes + (size_t)n_coarse * sizeof(uint2);
  int slot_id = 0;
  int rc = slot_begin(c, need, &slot_id);
  if (rc != KMC_OK) return rc;
  kmc_ctx::TableSlot& sl = c->slots[slot_id];
  TrajFrameRec* h_frecs = reinterpret_cast<TrajFrameRec*>(sl.h_buf);
  TrajSeg32* h_segs = reinterpret_cast<TrajSeg32*>(sl.h_buf + frecs_bytes);
  uint2* h_coarse = reinterpret_cast<uint2*>(sl.h_buf + frecs_bytes + segs_bytes);
  std::memset(h_segs, 0, segs_bytes);
  for (uint32_t f = 0; f < n_frames; ++f) {
    h_frecs[f].end_lo = (uint32_t)((offsets[f + 1] + head) & 0xFFFFFFFFull);
    h_frecs[f].end_hi = (uint32_t)((offsets[f + 1] + head) >> 32);
    h_frecs[f].n_seg = th[f].n_seg;
    h_frecs[f].pad = 0;
    fill_traj_segs(th[f], frames[f].stamp_start, frames[f].stamp_end, h_segs + (size_t)f * seg_stride);
  }
  build_coarse(offsets, n_frames, nv, head, h_coarse);
  rc = slot_upload(c, slot_id, need);
  if (rc != KMC_OK) return rc;
  const TrajFrameRec* d_frecs = reinterpret_cast<const TrajFrameRec*>(sl.d_buf);
  const TrajSeg32* d_segs = reinterpret_cast<const TrajSeg32*>(sl.d_buf + frecs_bytes);
  const uint2* d_coarse = reinterpret_cast<const uint2*>(sl.d_buf + frecs_bytes + segs_bytes);

  CallTimer tm(c);
  const v4f* d_in = (const v4f*)xyzi_in;
  v4f* d_out = (v4f*)xyzi_out;
  uint32_t* d_fidx = frame_idx_out;
  uint32_t* d_bidx = bracket_idx_out;
  const size_t pts = n * sizeof(v4f), idx_bytes = n * sizeof(uint32_t);
  if (mem_kind == KMC_MEM_HOST) {
    rc = ensure_tmp(c, 2 * pts + 2 * idx_bytes);
    if (rc != KMC_OK) return rc;
    d_in = (const v4f*)c->d_tmp;
    d_out = (v4f*)((char*)c->d_tmp + pts);
    d_fidx = frame_idx_out ? (uint32_t*)((char*)c->d_tmp + 2 * pts) : nullptr;
    d_bidx = bracket_idx_out ? (uint32_t*)((char*)c->d_tmp + 2 * pts + idx_bytes) : nullptr;
  }
  if (tm.begin_call()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
  if (mem_kind == KMC_MEM_HOST) KMC_HIP_TRY(c, hipMemcpyAsync((void*)d_in, xyzi_in, pts, hipMemcpyHostToDevice, c->stream));
  if (tm.begin_kernel()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
  const int grid = grid_for(c, (nv + 63) / 64);
  const bool idx = d_fidx || d_bidx;
  uint32_t* v_fidx = d_fidx ? d_fidx - head : nullptr;
  uint32_t* v_bidx = d_bidx ? d_bidx - head : nullptr;
#define KMC_LAUNCH_TRAJ_BATCH(T)                                                                                                   \
  do {                                                                                                                             \
    if (idx) hipLaunchKernelGGL((deskew_traj_batch_f32<T, kPolicyDefault, true>), dim3(grid), dim3(64), 0, c->stream, d_in - head, d_out - head, nv, d_frecs, d_segs, seg_stride, d_coarse, n_frames, v_fidx, v_bidx, head); \
    else hipLaunchKernelGGL((deskew_traj_batch_f32<T, kPolicyDefault, false>), dim3(grid), dim3(64), 0, c->stream, d_in - head, d_out - head, nv, d_frecs, d_segs, seg_stride, d_coarse, n_frames, v_fidx, v_bidx, head);     \
  } while (0)
  switch (tier) {
    case kSeries3: KMC_LAUNCH_TRAJ_BATCH(kSeries3); break;
    case kSeries5: KMC_LAUNCH_TRAJ_BATCH(kSeries5); break;
    default: KMC_LAUNCH_TRAJ_BATCH(kTrig); break;
  }
#undef KMC_LAUNCH_TRAJ_BATCH
  KMC_HIP_TRY(c, hipGetLastError());
  rc = slot_end(c, slot_id);
  if (rc != KMC_OK) return rc;
  if (tm.end_kernel()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
  if (mem_kind == KMC_MEM_HOST) {
    KMC_HIP_TRY(c, hipMemcpyAsync(xyzi_out, d_out, pts, hipMemcpyDeviceToHost, c->stream));
    if (frame_idx_out) KMC_HIP_TRY(c, hipMemcpyAsync(frame_idx_out, d_fidx, idx_bytes, hipMemcpyDeviceToHost, c->stream));
    if (bracket_idx_out) KMC_HIP_TRY(c, hipMemcpyAsync(bracket_idx_out, d_bidx, idx_bytes, hipMemcpyDeviceToHost, c->stream));
    KMC_HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  if (st) st->n_launches = 1;
  return tm.end_call(st);
}

int kmc_hip_deskew_traj_f64cols(kmc_ctx* c, const double* x, const double* y, const double* z, const double* w, const double* stamps,
                                uint64_t n, const double* knot_times, const double* knot_poses, uint32_t n_knots, double requested_time,
                                double* ox, double* oy, double* oz, double* ow, uint32_t* bracket_idx_out, int mem_kind, kmc_stats* st) {
  if (!c) return KMC_ERR_INVALID_ARG;
  if (n && (!x || !y || !z || !stamps || !ox || !oy || !oz)) return KMC_ERR_INVALID_ARG;
  if (mem_kind != KMC_MEM_HOST && mem_kind != KMC_MEM_DEVICE) return KMC_ERR_INVALID_ARG;
  if (st) std::memset(st, 0, sizeof(*st));
  TrajHost th;
  int rc = build_trajectory(knot_times, knot_poses, n_knots, requested_time, &th);
  if (rc != KMC_OK) return rc;
  if (st) { st->n_points = n; st->variant = 3; }
  if (n == 0) return KMC_OK;
  KMC_HIP_TRY(c, hipSetDevice(c->device));
  TrajSeg64 segs[kMaxSegments];
  std::memset(segs, 0, sizeof(segs));
  for (uint32_t k = 0; k < th.n_seg; ++k) {
    fill_rec64(th.f[k], &segs[k].f);
    segs[k].f.x_req = (k == th.r) ? th.x_r : 0.0;
    segs[k].f.t_start = th.t0[k];
    segs[k].f.t_end = th.t0[k] + th.dur[k];
    segs[k].f.dur = th.dur[k];
    th.M[k].to_rt12(segs[k].M);
    segs[k].identity = (k == th.r) ? 1 : 0;
  }
  rc = ensure_traj(c);
  if (rc != KMC_OK) return rc;
  TrajSeg64* d_segs = (TrajSeg64*)((char*)c->d_traj + kMaxSegments * sizeof(TrajSeg32));

  const double *dx = x, *dy = y, *dz = z, *dw = w, *ds = stamps;
  double *dox = ox, *doy = oy, *doz = oz, *dow = ow;
  uint32_t* d_idx = bracket_idx_out;
  const size_t col = n * sizeof(double);
  if (mem_kind == KMC_MEM_HOST) {
    rc = ensure_tmp(c, 9 * col + (bracket_idx_out ? n * sizeof(uint32_t) : 0));
    if (rc != KMC_OK) return rc;
    double* base = (double*)c->d_tmp;
    double* cols[9];
    for (int i = 0; i < 9; ++i) cols[i] = base + (size_t)i * n;
    // an Eigen::MatrixX4d is ONE column-major block: x, y, z, w follow each other -> one copy instead of four (each
    // pageable copy has a fixed cost of tens of microseconds, which is what a 123 k-point frame is made of)
    if (y == x + n && z == y + n && (!w || w == z + n)) {
      KMC_HIP_TRY(c, hipMemcpyAsync(cols[0], x, (w ? 4 : 3) * col, hipMemcpyHostToDevice, c->stream));
    } else {
      KMC_HIP_TRY(c, hipMemcpyAsync(cols[0], x, col, hipMemcpyHostToDevice, c->stream));
      KMC_HIP_TRY(c, hipMemcpyAsync(cols[1], y, col, hipMemcpyHostToDevice, c->stream));
      KMC_HIP_TRY(c, hipMemcpyAsync(cols[2], z, col, hipMemcpyHostToDevice, c->stream));
      if (w) KMC_HIP_TRY(c, hipMemcpyAsync(cols[3], w, col, hipMemcpyHostToDevice, c->stream));
    }
    KMC_HIP_TRY(c, hipMemcpyAsync(cols[4], stamps, col, hipMemcpyHostToDevice, c->stream));
    dx = cols[0]; dy = cols[1]; dz = cols[2]; dw = w ? cols[3] : nullptr; ds = cols[4];
    dox = cols[5]; doy = cols[6]; doz = cols[7]; dow = ow ? cols[8] : nullptr;
    d_idx = bracket_idx_out ? (uint32_t*)(base + 9 * n) : nullptr;
  }
  CallTimer tm(c);
  if (tm.begin_call()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
  rc = upload_traj(c, segs, kMaxSegments * sizeof(TrajSeg32), th.n_seg * sizeof(TrajSeg64));
  if (rc != KMC_OK) return rc;
  KMC_HIP_TRY(c, hipMemsetAsync(c->d_counter, 0, sizeof(unsigned long long), c->stream));
  if (tm.begin_kernel()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
  const int grid = grid_for(c, (n + 127) / 128);  // one wave per workgroup, two points per lane
  hipLaunchKernelGGL(deskew_traj_f64cols, dim3(grid), dim3(64), 0, c->stream, dx, dy, dz, dw, ds, n, (const TrajSeg64*)d_segs, th.n_seg,
                     knot_times[0], knot_times[n_knots - 1], dox, doy, doz, dow, d_idx, c->d_counter);
  KMC_HIP_TRY(c, hipGetLastError());
  if (tm.end_kernel()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
  unsigned long long bad = 0;
  KMC_HIP_TRY(c, hipMemcpyAsync(&bad, c->d_counter, sizeof(bad), hipMemcpyDeviceToHost, c->stream));
  if (mem_kind == KMC_MEM_HOST) {
    if (oy == ox + n && oz == oy + n && (!ow || ow == oz + n)) {  // one column-major block again
      KMC_HIP_TRY(c, hipMemcpyAsync(ox, dox, (ow ? 4 : 3) * col, hipMemcpyDeviceToHost, c->stream));
    } else {
      KMC_HIP_TRY(c, hipMemcpyAsync(ox, dox, col, hipMemcpyDeviceToHost, c->stream));
      KMC_HIP_TRY(c, hipMemcpyAsync(oy, doy, col, hipMemcpyDeviceToHost, c->stream));
      KMC_HIP_TRY(c, hipMemcpyAsync(oz, doz, col, hipMemcpyDeviceToHost, c->stream));
      if (ow) KMC_HIP_TRY(c, hipMemcpyAsync(ow, dow, col, hipMemcpyDeviceToHost, c->stream));
    }
    if (bracket_idx_out) KMC_HIP_TRY(c, hipMemcpyAsync(bracket_idx_out, d_idx, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  }
  KMC_HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (st) { st->n_launches = 1; st->n_out_of_range = bad; }
  rc = tm.end_call(st);
  if (rc != KMC_OK) return rc;
  return bad ? KMC_ERR_TIME_OUT_OF_RANGE : KMC_OK;
}

// ---- synthetic workload ----------------------------------------------------------------------------
// ---- N4: projection ----------------------------------------------------------------------------------
namespace {
bool rig_ok(const kmc_camera_rig* g) {
  const double* v = g->tf_c00_lo;  // the struct is 70 contiguous doubles
  for (size_t i = 0; i < sizeof(kmc_camera_rig) / sizeof(double); ++i)
    if (!std::isfinite(v[i])) return false;
  return true;
}
// [fx 0 cx tx; 0 fy cy ty; 0 0 1 tz] for all four cameras: the kernel may skip the products with the literal 0s and 1
bool rig_is_pinhole(const kmc_camera_rig* g) {
  for (int c = 0; c < 4; ++c) {
    const double* P = g->P_rect[c];
    if (P[1] != 0.0 || P[4] != 0.0 || P[8] != 0.0 || P[9] != 0.0 || P[10] != 1.0) return false;
  }
  return true;
}
CameraRigRec rig_rec(const kmc_camera_rig* g) {
  CameraRigRec r;
  std::memcpy(r.T, g->tf_c00_lo, sizeof(r.T));
  std::memcpy(r.R, g->R_rect_00, sizeof(r.R));
  std::memcpy(r.P, g->P_rect, sizeof(r.P));
  r.max_range = g->max_range;
  r.range_den = g->max_range - 0.01;  // camera_model.cpp:28
  return r;
}
}  // namespace

int kmc_hip_project_f32(kmc_ctx* c, const float* xyzi_in, uint64_t n, const kmc_camera_rig* rig, const kmc_frame_params* deskew,
                        float* xyzi_out, int32_t* uv, uint8_t* bgrv, int mem_kind, kmc_stats* st) {
  if (!c || !rig || (n && (!xyzi_in || !uv || !bgrv))) return KMC_ERR_INVALID_ARG;
  if (mem_kind != KMC_MEM_HOST && mem_kind != KMC_MEM_DEVICE) return KMC_ERR_INVALID_ARG;
  if (xyzi_out && !deskew) return KMC_ERR_INVALID_ARG;
  if (((uintptr_t)xyzi_in | (uintptr_t)xyzi_out) & 15u) return KMC_ERR_INVALID_ARG;
  if (((uintptr_t)uv & 15u) || ((uintptr_t)bgrv & 3u)) return KMC_ERR_INVALID_ARG;
  if (!rig_ok(rig)) return KMC_ERR_INVALID_ARG;
  if (deskew) {
    if (!params_ok(deskew)) return KMC_ERR_INVALID_ARG;
    if (!(deskew->x_req >= 0.0 && deskew->x_req <= 1.0)) return KMC_ERR_TIME_OUT_OF_RANGE;
  }
  if (st) std::memset(st, 0, sizeof(*st));
  KMC_HIP_TRY(c, hipSetDevice(c->device));
  const int tier = deskew ? pick_tier(c, deskew, 1) : -1;
  FrameRec f;
  std::memset(&f, 0, sizeof(f));
  if (deskew) fill_rec(*deskew, &f);
  if (st) { st->n_points = n; st->variant = (uint32_t)(tier < 0 ? 4 : tier); }
  if (n == 0) return KMC_OK;
  const CameraRigRec g = rig_rec(rig);

  const v4f* d_in = (const v4f*)xyzi_in;
  v4f* d_cloud = (v4f*)xyzi_out;
  v2i* d_uv = (v2i*)uv;
  uint32_t* d_col = (uint32_t*)bgrv;
  const size_t cloud_bytes = n * sizeof(v4f), uv_bytes = n * 4 * sizeof(v2i), col_bytes = n * sizeof(uint32_t);
  if (mem_kind == KMC_MEM_HOST) {
    int rc = ensure_tmp(c, 2 * cloud_bytes + uv_bytes + col_bytes);
    if (rc != KMC_OK) return rc;
    char* base = (char*)c->d_tmp;
    d_in = (const v4f*)base;
    d_cloud = xyzi_out ? (v4f*)(base + cloud_bytes) : nullptr;
    d_uv = (v2i*)(base + 2 * cloud_bytes);
    d_col = (uint32_t*)(base + 2 * cloud_bytes + uv_bytes);
    KMC_HIP_TRY(c, hipMemcpyAsync(base, xyzi_in, cloud_bytes, hipMemcpyHostToDevice, c->stream));
  }
  CallTimer tm(c);
  if (tm.begin_call() || tm.begin_kernel()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
  const int grid = grid_for(c, (n + 63) / 64);
  const bool pinhole = rig_is_pinhole(rig);
#define KMC_LAUNCH_PROJECT(T)                                                                                                     \
  do {                                                                                                                            \
    if (pinhole) hipLaunchKernelGGL((project_f32<T, true>), dim3(grid), dim3(64), 0, c->stream, d_in, n, g, f, d_cloud, d_uv, d_col); \
    else hipLaunchKernelGGL((project_f32<T, false>), dim3(grid), dim3(64), 0, c->stream, d_in, n, g, f, d_cloud, d_uv, d_col);       \
  } while (0)
  switch (tier) {
    case kSeries3: KMC_LAUNCH_PROJECT(kSeries3); break;
    case kSeries5: KMC_LAUNCH_PROJECT(kSeries5); break;
    case kTrig: KMC_LAUNCH_PROJECT(kTrig); break;
    default: KMC_LAUNCH_PROJECT(-1); break;
  }
#undef KMC_LAUNCH_PROJECT
  KMC_HIP_TRY(c, hipGetLastError());
  if (tm.end_kernel()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
  if (mem_kind == KMC_MEM_HOST) {
    if (xyzi_out) KMC_HIP_TRY(c, hipMemcpyAsync(xyzi_out, d_cloud, cloud_bytes, hipMemcpyDeviceToHost, c->stream));
    KMC_HIP_TRY(c, hipMemcpyAsync(uv, d_uv, uv_bytes, hipMemcpyDeviceToHost, c->stream));
    KMC_HIP_TRY(c, hipMemcpyAsync(bgrv, d_col, col_bytes, hipMemcpyDeviceToHost, c->stream));
    KMC_HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  if (st) st->n_launches = 1;
  return tm.end_call(st);
}

int kmc_hip_project_f64cols(kmc_ctx* c, const double* x, const double* y, const double* z, uint64_t n, const kmc_camera_rig* rig,
                            int32_t* uv, uint8_t* bgrv, int mem_kind, kmc_stats* st) {
  if (!c || !rig || (n && (!x || !y || !z || !uv || !bgrv))) return KMC_ERR_INVALID_ARG;
  if (mem_kind != KMC_MEM_HOST && mem_kind != KMC_MEM_DEVICE) return KMC_ERR_INVALID_ARG;
  if (((uintptr_t)uv & 15u) || ((uintptr_t)bgrv & 3u)) return KMC_ERR_INVALID_ARG;
  if (!rig_ok(rig)) return KMC_ERR_INVALID_ARG;
  if (st) std::memset(st, 0, sizeof(*st));
  if (st) { st->n_points = n; st->variant = 4; }
  if (n == 0) return KMC_OK;
  KMC_HIP_TRY(c, hipSetDevice(c->device));
  const CameraRigRec g = rig_rec(rig);
  const double *dx = x, *dy = y, *dz = z;
  v2i* d_uv = (v2i*)uv;
  uint32_t* d_col = (uint32_t*)bgrv;
  const size_t col = n * sizeof(double), uv_bytes = n * 4 * sizeof(v2i), col_bytes = n * sizeof(uint32_t);
  if (mem_kind == KMC_MEM_HOST) {
    const size_t cols_bytes = (3 * col + 15) & ~(size_t)15;  // the pixel records behind the columns stay 16-byte aligned
    int rc = ensure_tmp(c, cols_bytes + uv_bytes + col_bytes);
    if (rc != KMC_OK) return rc;
    double* base = (double*)c->d_tmp;
    if (y == x + n && z == y + n) {  // three adjacent columns of one Eigen matrix: one copy
      KMC_HIP_TRY(c, hipMemcpyAsync(base, x, 3 * col, hipMemcpyHostToDevice, c->stream));
    } else {
      KMC_HIP_TRY(c, hipMemcpyAsync(base, x, col, hipMemcpyHostToDevice, c->stream));
      KMC_HIP_TRY(c, hipMemcpyAsync(base + n, y, col, hipMemcpyHostToDevice, c->stream));
      KMC_HIP_TRY(c, hipMemcpyAsync(base + 2 * n, z, col, hipMemcpyHostToDevice, c->stream));
    }
    dx = base; dy = base + n; dz = base + 2 * n;
    d_uv = (v2i*)((char*)base + cols_bytes);
    d_col = (uint32_t*)((char*)base + cols_bytes + uv_bytes);
  }
  CallTimer tm(c);
  if (tm.begin_call() || tm.begin_kernel()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
  const int grid = grid_for(c, (n + 63) / 64);
  if (rig_is_pinhole(rig)) hipLaunchKernelGGL(project_f64cols<true>, dim3(grid), dim3(64), 0, c->stream, dx, dy, dz, n, g, d_uv, d_col);
  else hipLaunchKernelGGL(project_f64cols<false>, dim3(grid), dim3(64), 0, c->stream, dx, dy, dz, n, g, d_uv, d_col);
  KMC_HIP_TRY(c, hipGetLastError());
  if (tm.end_kernel()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
  if (mem_kind == KMC_MEM_HOST) {
    KMC_HIP_TRY(c, hipMemcpyAsync(uv, d_uv, uv_bytes, hipMemcpyDeviceToHost, c->stream));
    KMC_HIP_TRY(c, hipMemcpyAsync(bgrv, d_col, col_bytes, hipMemcpyDeviceToHost, c->stream));
    KMC_HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  if (st) st->n_launches = 1;
  return tm.end_call(st);
}

int kmc_hip_synth_points(kmc_ctx* c, float* xyzi_out_device, uint64_t n, uint64_t seed) {
  if (!c || (n && !xyzi_out_device)) return KMC_ERR_INVALID_ARG;
  if (n == 0) return KMC_OK;
  KMC_HIP_TRY(c, hipSetDevice(c->device));
  const int grid = grid_for(c, (n + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(synth_points, dim3(grid), dim3(kBlock), 0, c->stream, (v4f*)xyzi_out_device, n, seed);
  KMC_HIP_TRY(c, hipGetLastError());
  return KMC_OK;
}

int kmc_synth_points_host(float* out, uint64_t n, uint64_t seed) {
  if (n && !out) return KMC_ERR_INVALID_ARG;
  for (uint64_t i = 0; i < n; ++i) {
    const kmc_synth::Point p = kmc_synth::make_point(i, n, seed);
    out[4 * i + 0] = p.x;
    out[4 * i + 1] = p.y;
    out[4 * i + 2] = p.z;
    out[4 * i + 3] = p.i;
  }
  return KMC_OK;
}

}  // extern "C"

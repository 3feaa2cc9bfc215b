// kmc_capi_traj.hip -- N-knot trajectories (the 3-argument MotionCompensateFrame(Frame, Trajectory, Time) of BASELINE.json's
// north_star): f64 host pre-step per trajectory, single-frame, batched and f64 Eigen-layout entry points.
#include "kmc_internal.hip.h"

#include <hip/hip_ext.h>

namespace {

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
  r->halvings = halvings_for(r->phi2);  // |s| <= 1 inside a segment
  r->terms = series_terms_for(r->phi2);
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

// device records of one frame's trajectory: one TrajSeg32 per segment and its f64 twin for the near-origin guard's redo
void fill_traj_segs(const TrajHost& th, double stamp_start, double stamp_end, TrajSeg32* segs, TrajSegD* segs64) {
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
    r.pre2 = (k == th.r) ? kGuardPre * (float)kmc_host::dot(th.f[k].rho, th.f[k].rho)
                         : kGuardPreTraj * (float)(kmc_host::dot(th.f[k].rho, th.f[k].rho) + kmc_host::dot(th.M[k].t, th.M[k].t));
    TrajSegD& d = segs64[k];
    {
      kmc_frame_params fd = fp;
      FrameRecD w;
      fill_recd(fd, &w);
      for (int j = 0; j < 3; ++j) { d.phi[j] = w.phi[j]; d.rho[j] = w.rho[j]; d.c1[j] = w.c1[j]; d.c2[j] = w.c2[j]; }
      d.phi2 = w.phi2;
    }
    d.c = ck;
    d.g = g;
    d.a = a;
    th.M[k].to_rt12(d.M);
    d.identity = (k == th.r) ? 1u : 0u;
  }
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
  if (c->force_tier >= 0 && c->force_tier <= kTrig) return c->force_tier;
  // |s| <= 1 inside a segment except for the anchor segment (|x - x_r| <= 1 as well); the scan may stick out of the first /
  // last segment by at most the knots' coverage, which build_trajectory() / the callers have verified -> bound by 1.
  (void)span_lo; (void)span_hi;
  double theta_max = 0.0;
  for (uint32_t k = 0; k < th.n_seg; ++k) theta_max = std::fmax(theta_max, kmc_host::norm(th.f[k].phi));
  return tier_of_theta(theta_max);
}
}  // namespace

extern "C" {

// ---- N-knot trajectory entry points ---------------------------------------------------------------------------------
int kmc_hip_deskew_traj_f32(kmc_ctx* c, const float* xyzi_in, float* xyzi_out, uint64_t n, const double* knot_times,
                            const double* knot_poses, uint32_t n_knots, double stamp_start, double stamp_end, double requested_time,
                            uint32_t* bracket_idx_out, int mem_kind, kmc_stats* st) {
  if (!c || (n && (!xyzi_in || !xyzi_out))) return KMC_ERR_INVALID_ARG;
  if (mem_kind != KMC_MEM_HOST && mem_kind != KMC_MEM_DEVICE) return KMC_ERR_INVALID_ARG;
  if (((uintptr_t)xyzi_in | (uintptr_t)xyzi_out) & (mem_kind == KMC_MEM_DEVICE ? 15u : 3u)) return KMC_ERR_INVALID_ARG;
  if (!(stamp_start < stamp_end)) return KMC_ERR_DEGENERATE;
  if (st) std::memset(st, 0, sizeof(*st));
  TrajHost th;
  int rc = build_trajectory(knot_times, knot_poses, n_knots, requested_time, &th);
  if (rc != KMC_OK) return rc;
  // every point stamp lies in [stamp_start, stamp_end]: the trajectory has to cover the scan
  if (!(knot_times[0] <= stamp_start && stamp_end <= knot_times[n_knots - 1])) return KMC_ERR_TIME_OUT_OF_RANGE;
  if (!(requested_time >= stamp_start && requested_time <= stamp_end)) return KMC_ERR_TIME_OUT_OF_RANGE;
  const bool inline_records = mem_kind == KMC_MEM_DEVICE && th.n_seg <= (uint32_t)kInlineSegments;
  // in order on the context's stream, but -- like kmc_hip_deskew_f32 -- not behind frames it shares no buffer with (kmc_ctx::ao)
  const bool window = inline_records && !c->timing && c->gl.count == 0 && !bracket_idx_out && n;
  // The context's OWN stream: such a frame goes out through the direct queue like a two-pose frame (kmc_capi_direct.hip) -- its records
  // ride in the packet's argument block, no HIP call on the way.  Same tile body as the HIP launch below, same bits.
  const bool direct_eligible = window && c->stream == c->own_stream && c->fq_count <= 1 && c->dd_wanted && !c->dd_broken;  // (not with gathering on: kmc_hip.h)
  // (a frame for an OPEN direct queue needs no HIP call at all unless it follows HIP-stream work; everything else -- opening the queue
  // included: its self-test allocates and launches -- happens on the context's device)
  if (!(direct_eligible && c->dd && !c->stream_dirty)) KMC_HIP_TRY(c, hipSetDevice(c->device));
  const bool direct = direct_eligible && (c->dd || direct_open(c));
  if (direct) {
    if (c->stream_dirty) {
      KMC_HIP_TRY(c, hipStreamSynchronize(c->stream));
      c->stream_dirty = false;
    }
    const int tier_d = traj_tier(c, th, stamp_start, stamp_end);
    TrajInline inl;
    std::memset(&inl, 0, sizeof(inl));
    fill_traj_segs(th, stamp_start, stamp_end, inl.s, inl.d);
    const uintptr_t bytes = (uintptr_t)n * sizeof(v4f);
    const kmc_ctx::AoRange r = {(uintptr_t)xyzi_in, (uintptr_t)xyzi_in + bytes}, w = {(uintptr_t)xyzi_out, (uintptr_t)xyzi_out + bytes};
    const kmc_book::LaneVerdict lane = c->lw.admit(r, w, c->dd_free_order, direct_frame_is_huge(n));
    uint32_t launches = 0;
    rc = direct_traj_frame(c, tier_d, (const v4f*)xyzi_in, (v4f*)xyzi_out, n, th.n_seg, head_of(xyzi_out, KMC_MEM_DEVICE), inl, lane, &launches);
    if (rc != KMC_OK) return rc;
    if (st) { st->n_points = n; st->variant = (uint32_t)tier_d; st->n_launches = launches; }
    return KMC_OK;
  }
  if (!window) {
    rc = fq_join(c);  // (also issues two-pose frames that are still being gathered: the N-knot kernel is launched per call)
    if (rc != KMC_OK) return rc;
  } else if (c->dd_pending) {  // a HIP launch behind frames in the direct queue: wait for them (kmc_capi_direct.hip)
    rc = direct_join(c);
    if (rc != KMC_OK) return rc;
  }
  c->stream_dirty = true;
  const int tier = traj_tier(c, th, stamp_start, stamp_end);
  if (st) { st->n_points = n; st->variant = (uint32_t)tier; }
  if (n == 0) return KMC_OK;

  // one launch site for both routes: tier, index output, record source (kernel arguments / device table) and the barrier bit are
  // run-time choices of this call, template arguments of the kernel
  auto launch_traj = [&](auto INL, hipStream_t stream, bool any_order, const v4f* d_in, v4f* d_out, uint32_t* d_idx, const TrajSeg32* segs, const TrajSegD* segs64,
                         const TrajInline& inl) {
    const uint32_t head = head_of(xyzi_out, mem_kind);
    const uint64_t nv = n + head;
    uint32_t* v_idx = d_idx ? d_idx - head : nullptr;
    with_tier(tier, [&](auto T) {
      with_bool(d_idx != nullptr, [&](auto IDX) {
        launch_tiles((nv + 63) / 64, [&](uint64_t t0, int grid) {
          launch_on(deskew_traj_f32<decltype(T)::value, decltype(IDX)::value, decltype(INL)::value>, grid, 64, stream, any_order, d_in - head, d_out - head, nv, segs, th.n_seg,
                    v_idx, head, segs64, t0, inl);
        });
      });
    });
  };

  if (inline_records) {
    // A short trajectory on device-resident points -- north_star's "three bracketing poses" -- carries its segment records in the
    // kernel arguments: no table slot, no upload, nothing for the host to wait for.  Same records, same kernel body: same bits as the
    // table path below.
    TrajInline inl;
    std::memset(&inl, 0, sizeof(inl));
    fill_traj_segs(th, stamp_start, stamp_end, inl.s, inl.d);
    hipStream_t s = c->stream;
    const v4f* d_in = (const v4f*)xyzi_in;
    v4f* d_out = (v4f*)xyzi_out;
    uint32_t* d_idx = bracket_idx_out;
    const bool any_order = window && ao_admit(c, xyzi_in, xyzi_out, n * sizeof(v4f), false);
    CallTimer tm(c);
    if (tm.begin_call() || tm.begin_kernel()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
    launch_traj(std::true_type{}, s, any_order, d_in, d_out, d_idx, nullptr, nullptr, inl);
    KMC_HIP_TRY(c, hipGetLastError());
    if (tm.end_kernel()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
    if (st) st->n_launches = 1;
    return tm.end_call(st);
  }

  // slot layout: [TrajSeg32 x kMaxSegments | TrajSegD x kMaxSegments]
  int slot_id = 0;
  rc = slot_begin(c, kMaxSegments * (sizeof(TrajSeg32) + sizeof(TrajSegD)), &slot_id);
  if (rc != KMC_OK) return rc;
  TrajSeg32* segs = reinterpret_cast<TrajSeg32*>(c->slots[slot_id].h_buf);
  TrajSegD* segs64 = reinterpret_cast<TrajSegD*>(c->slots[slot_id].h_buf + kMaxSegments * sizeof(TrajSeg32));
  std::memset(segs, 0, kMaxSegments * (sizeof(TrajSeg32) + sizeof(TrajSegD)));
  fill_traj_segs(th, stamp_start, stamp_end, segs, segs64);
  rc = slot_upload(c, slot_id, kMaxSegments * sizeof(TrajSeg32) + th.n_seg * sizeof(TrajSegD));
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
  const TrajSeg32* d_segs = (const TrajSeg32*)c->slots[slot_id].d_buf;
  const TrajSegD* d_segs64 = (const TrajSegD*)(c->slots[slot_id].d_buf + kMaxSegments * sizeof(TrajSeg32));
  launch_traj(std::false_type{}, c->stream, false, d_in, d_out, d_idx, d_segs, d_segs64, TrajInline{});  // the table route: ordinary launches
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
  if (((uintptr_t)xyzi_in | (uintptr_t)xyzi_out) & (mem_kind == KMC_MEM_DEVICE ? 15u : 3u)) return KMC_ERR_INVALID_ARG;
  KMC_ENTER(c);  // batched launches stay on the context's stream, see kmc_hip_deskew_batch_f32

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
  if (c->force_tier >= 0 && c->force_tier <= kTrig) tier = c->force_tier;
  if (st) { st->n_points = n; st->variant = (uint32_t)tier; }
  if (n == 0) return KMC_OK;

  const uint32_t head = head_of(xyzi_out, mem_kind);
  const uint64_t nv = n + head;
  const uint64_t chunk = 1ull << kChunkShift;
  const uint64_t n_coarse = (nv + chunk - 1) / chunk + 1;
  const size_t frecs_bytes = ((size_t)n_frames * sizeof(TrajFrameRec) + 255) & ~(size_t)255;
  // + kInlineSegments spare records: the kernel loads the first two interior knot slots of a frame unconditionally
  const size_t segs_bytes = (((size_t)n_frames * seg_stride + kInlineSegments) * sizeof(TrajSeg32) + 255) & ~(size_t)255;
  const size_t segd_bytes = ((size_t)n_frames * seg_stride * sizeof(TrajSegD) + 255) & ~(size_t)255;  // f64 twins (guard redo)
  const size_t need = frecs_bytes + segs_bytes + segd_bytes + (size_t)n_coarse * sizeof(uint2);
  int slot_id = 0;
  int rc = slot_begin(c, need, &slot_id);
  if (rc != KMC_OK) return rc;
  kmc_ctx::TableSlot& sl = c->slots[slot_id];
  TrajFrameRec* h_frecs = reinterpret_cast<TrajFrameRec*>(sl.h_buf);
  TrajSeg32* h_segs = reinterpret_cast<TrajSeg32*>(sl.h_buf + frecs_bytes);
  TrajSegD* h_segd = reinterpret_cast<TrajSegD*>(sl.h_buf + frecs_bytes + segs_bytes);
  uint2* h_coarse = reinterpret_cast<uint2*>(sl.h_buf + frecs_bytes + segs_bytes + segd_bytes);
  std::memset(h_segs, 0, segs_bytes + segd_bytes);
  for (uint32_t f = 0; f < n_frames; ++f) {
    h_frecs[f].end_lo = (uint32_t)((offsets[f + 1] + head) & 0xFFFFFFFFull);
    h_frecs[f].end_hi = (uint32_t)((offsets[f + 1] + head) >> 32);
    h_frecs[f].n_seg = th[f].n_seg;
    h_frecs[f].pad = 0;
    TrajSeg32* fs = h_segs + (size_t)f * seg_stride;
    fill_traj_segs(th[f], frames[f].stamp_start, frames[f].stamp_end, fs, h_segd + (size_t)f * seg_stride);
    h_frecs[f].c1 = th[f].n_seg > 1 ? fs[1].knot_c : 0.f;  // the SAME f32 values the records hold
    h_frecs[f].c2 = th[f].n_seg > 2 ? fs[2].knot_c : 0.f;
    h_frecs[f].pad2[0] = h_frecs[f].pad2[1] = 0;
  }
  build_coarse(offsets, n_frames, nv, head, h_coarse);
  rc = slot_upload(c, slot_id, need);
  if (rc != KMC_OK) return rc;
  const TrajFrameRec* d_frecs = reinterpret_cast<const TrajFrameRec*>(sl.d_buf);
  const TrajSeg32* d_segs = reinterpret_cast<const TrajSeg32*>(sl.d_buf + frecs_bytes);
  const TrajSegD* d_segd = reinterpret_cast<const TrajSegD*>(sl.d_buf + frecs_bytes + segs_bytes);
  const uint2* d_coarse = reinterpret_cast<const uint2*>(sl.d_buf + frecs_bytes + segs_bytes + segd_bytes);

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
  const bool idx = d_fidx || d_bidx;
  uint32_t* v_fidx = d_fidx ? d_fidx - head : nullptr;
  uint32_t* v_bidx = d_bidx ? d_bidx - head : nullptr;
  with_tier(tier, [&](auto T) {
    with_bool(idx, [&](auto IDX) {
      launch_tiles((nv + 63) / 64, [&](uint64_t t0, int grid) {
        launch_on(deskew_traj_batch_f32<decltype(T)::value, decltype(IDX)::value>, grid, 64, c->stream, false, d_in - head, d_out - head, nv, d_frecs, d_segs, seg_stride,
                  d_coarse, n_frames, v_fidx, v_bidx, head, d_segd, t0);
      });
    });
  });
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
  if (mem_kind != KMC_MEM_HOST && mem_kind != KMC_MEM_DEVICE && mem_kind != KMC_MEM_HOST_MAPPED) return KMC_ERR_INVALID_ARG;
  if (st) std::memset(st, 0, sizeof(*st));
  TrajHost th;
  int rc = build_trajectory(knot_times, knot_poses, n_knots, requested_time, &th);
  if (rc != KMC_OK) return rc;
  if (st) { st->n_points = n; st->variant = 5; }
  if (n == 0) return KMC_OK;
  if (c->f64_pending) return KMC_ERR_INVALID_ARG;  // between kmc_hip_deskew_f64cols_begin and its _end: the range flag word is in use
  KMC_ENTER(c);
  TrajSeg64 segs[kMaxSegments];
  std::memset(segs, 0, sizeof(segs));
  for (uint32_t k = 0; k < th.n_seg; ++k) {
    fill_rec64(th.f[k], &segs[k].f);
    segs[k].f.x_req = (k == th.r) ? th.x_r : 0.0;
    segs[k].f.t_start = th.t0[k];
    segs[k].f.t_end = th.t0[k] + th.dur[k];
    segs[k].f.inv_dur = 1.0 / th.dur[k];
    th.M[k].to_rt12(segs[k].M);
    segs[k].identity = (k == th.r) ? 1 : 0;
  }
  // a short trajectory (north_star's three bracketing poses) carries its records in the kernel arguments: no table, no upload, no event in
  // front of the kernel (what made the 3-argument drop-in call 35 us slower than the 2-argument one)
  const bool inline_records = th.n_seg <= (uint32_t)kInlineSegments;
  TrajInline64 inl;
  std::memset(&inl, 0, sizeof(inl));
  TrajSeg64* d_segs = nullptr;
  if (inline_records) {
    std::memcpy(inl.s, segs, th.n_seg * sizeof(TrajSeg64));
  } else {
    rc = ensure_traj(c);
    if (rc != KMC_OK) return rc;
    d_segs = (TrajSeg64*)((char*)c->d_traj + kMaxSegments * sizeof(TrajSeg32));
  }

  const double *dx = x, *dy = y, *dz = z, *dw = w, *ds = stamps;
  double *dox = ox, *doy = oy, *doz = oz, *dow = ow;
  uint32_t* d_idx = bracket_idx_out;
  const size_t col = n * sizeof(double);
  // containers made of the page-locked pool: the kernel works on them in place (see kmc_hip_deskew_f64cols)
  if (mem_kind == KMC_MEM_HOST && n >= 2048 && host_in_place_ok(x, col) && host_in_place_ok(y, col) && host_in_place_ok(z, col) && (!w || host_in_place_ok(w, col)) &&
      host_in_place_ok(stamps, col) && host_in_place_ok(ox, col) && host_in_place_ok(oy, col) && host_in_place_ok(oz, col) && (!ow || host_in_place_ok(ow, col)) &&
      (!bracket_idx_out || host_in_place_ok(bracket_idx_out, n * sizeof(uint32_t))))
    mem_kind = KMC_MEM_HOST_MAPPED;
  // No homogeneous column given (= all ones) but one wanted back, in HOST memory: the device neither reads nor writes it -- the host
  // fills the output's column with ones while the kernel works on the other three (8 of 40 bytes per point less on the link's busier
  // direction; the drop-in's 3-argument MotionCompensateFrame on a loader-made cloud)
  const bool host_fills_ow = !w && ow && mem_kind != KMC_MEM_DEVICE;
  if (host_fills_ow) dow = nullptr;
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
    dox = cols[5]; doy = cols[6]; doz = cols[7]; dow = (ow && !host_fills_ow) ? cols[8] : nullptr;
    d_idx = bracket_idx_out ? (uint32_t*)(base + 9 * n) : nullptr;
  }
  CallTimer tm(c);
  if (tm.begin_call()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
  if (!inline_records) {
    rc = upload_traj(c, segs, kMaxSegments * sizeof(TrajSeg32), th.n_seg * sizeof(TrajSeg64));
    if (rc != KMC_OK) return rc;
  }
  if (c->counter_dirty) KMC_HIP_TRY(c, hipMemsetAsync(c->d_counter, 0, sizeof(unsigned long long), c->stream));
  c->counter_dirty = true;
  *c->h_flag = 0;
  if (tm.begin_kernel()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
  if (mem_kind == KMC_MEM_HOST_MAPPED) {
    // over the link: persistent waves, sc1 stores, completion word (like kmc_hip_deskew_f64cols; half the wave count for a KITTI-sized frame)
    const uint64_t waves = n < (1ull << 19) ? std::max(1, c->mapped_waves / 2) : c->mapped_waves;
    const int grid = (int)std::max<uint64_t>(1, std::min<uint64_t>((n + 127) / 128, waves));
    const DoneWord done = done_word_arm(c);
    launch_on(deskew_traj_f64cols<true>, grid, 64, c->stream, false, dx, dy, dz, dw, ds, n, (const TrajSeg64*)d_segs, th.n_seg, knot_times[0], knot_times[n_knots - 1], dox, doy, doz,
              dow, d_idx, c->d_counter, c->h_flag, (uint64_t)0, done, inl);
  } else {
    launch_tiles((n + 127) / 128, [&](uint64_t t0, int grid) {  // one wave per workgroup, two points per lane
      launch_on(deskew_traj_f64cols<false>, grid, 64, c->stream, false, dx, dy, dz, dw, ds, n, (const TrajSeg64*)d_segs, th.n_seg, knot_times[0], knot_times[n_knots - 1], dox, doy,
                doz, dow, d_idx, c->d_counter, c->h_flag, t0, DoneWord{}, inl);
    });
  }
  KMC_HIP_TRY(c, hipGetLastError());
  if (tm.end_kernel()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
  unsigned long long bad = 0;
  if (mem_kind == KMC_MEM_HOST) {
    if (oy == ox + n && oz == oy + n && (!dow || ow == oz + n)) {  // one column-major block again
      KMC_HIP_TRY(c, hipMemcpyAsync(ox, dox, (dow ? 4 : 3) * col, hipMemcpyDeviceToHost, c->stream));
    } else {
      KMC_HIP_TRY(c, hipMemcpyAsync(ox, dox, col, hipMemcpyDeviceToHost, c->stream));
      KMC_HIP_TRY(c, hipMemcpyAsync(oy, doy, col, hipMemcpyDeviceToHost, c->stream));
      KMC_HIP_TRY(c, hipMemcpyAsync(oz, doz, col, hipMemcpyDeviceToHost, c->stream));
      if (dow) KMC_HIP_TRY(c, hipMemcpyAsync(ow, dow, col, hipMemcpyDeviceToHost, c->stream));
    }
    if (bracket_idx_out) KMC_HIP_TRY(c, hipMemcpyAsync(bracket_idx_out, d_idx, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  }
  if (host_fills_ow) std::fill(ow, ow + n, 1.0);  // while the device is busy
  if (mem_kind == KMC_MEM_HOST_MAPPED) {  // in place: the kernel's completion word says "everything is in host memory"
    const int rc_wait = wait_done_word(c);
    if (rc_wait != KMC_OK) return rc_wait;
  } else {
    KMC_HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  if (*(volatile uint32_t*)c->h_flag != 0) KMC_HIP_TRY(c, hipMemcpy(&bad, c->d_counter, sizeof(bad), hipMemcpyDeviceToHost));  // cold
  else c->counter_dirty = false;  // nobody touched the counter
  if (st) { st->n_launches = 1; st->n_out_of_range = bad; }
  rc = tm.end_call(st);
  if (rc != KMC_OK) return rc;
  return bad ? KMC_ERR_TIME_OUT_OF_RANGE : KMC_OK;
}
}  // extern "C"

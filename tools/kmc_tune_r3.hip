// kmc_tune_r3.hip -- round-3 A/B harness (run through gpurun; CSV on stdout).
//
// Same method as kmc_tune.hip (all variants in ONE process, interleaved over several rounds, rotating buffers far beyond
// the 256 MiB Infinity Cache), restricted to the questions of round 3:
//   * the single-frame N-knot kernel: round 2's (every wave stages the segment records in LDS) against round 3's (bracket from the
//     azimuth with an exact fallback, records through scalar loads, no LDS) -- from a device table and from the kernel-argument
//     segment, for 2, 3 and 6 segments.  (Two more candidates were measured and dropped, profiles/r03_tune_traj.csv: the LDS
//     staging moved in front of the point load, and both records of a three-knot trajectory resident in SGPRs.);
//   * the any-angle tier: round 2's ocml sincosf + IEEE divides against the Cody-Waite / v_rsq_f32 rewrite.
// Every N-knot candidate's output is compared BITWISE with the round-2 kernel's before it is timed.
//
// usage: kmc_tune_r3 [n_points=67108864] [rounds=5] [iters=10] [name-filter]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../kitti_motion_compensation_amd/csrc/kmc_kernels.hip.h"

namespace kmc_dev {
// ---- round 2's single-frame N-knot kernel (every wave stages the segment records in LDS), kept HERE for the A/B and as the
// bit reference of the round-3 kernel ----
template <int TIER>
__device__ __forceinline__ v4f r2_traj_point(const v4f p, const TrajSeg32& r, bool& redo) {
  FrameRec f;
  f.phi_x = r.phi_x; f.phi_y = r.phi_y; f.phi_z = r.phi_z; f.phi2 = r.phi2;
  f.rho_x = r.rho_x; f.rho_y = r.rho_y; f.rho_z = r.rho_z; f.s0 = r.s0;
  f.c1_x = r.c1_x; f.c1_y = r.c1_y; f.c1_z = r.c1_z; f.pre2 = 0.f;
  f.c2_x = r.c2_x; f.c2_y = r.c2_y; f.c2_z = r.c2_z; f.pad1 = 0.f;
  const float turns = azimuth_turns(p.x, p.y);
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
template <int TIER, int NT, bool WRITE_IDX, bool INLINE = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void r2_deskew_traj_f32(const v4f* __restrict__ in, v4f* __restrict__ out, uint64_t n,
                                                     const TrajSeg32* __restrict__ segs, uint32_t n_seg,
                                                     uint32_t* __restrict__ bracket_out, uint32_t head,
                                                     const TrajSegD* __restrict__ segs64, TrajInline inl) {
  // `head`: dead leading indices, see deskew_frame_f32
  constexpr int BLOCK = 64;
  if constexpr (INLINE) {
    struct ArgLayout { const v4f* in; v4f* out; uint64_t n; const TrajSeg32* segs; uint32_t n_seg; uint32_t* bracket_out; uint32_t head; const TrajSegD* segs64; TrajInline inl; };
    const auto kernarg = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    segs = (const TrajSeg32*)(const char*)(kernarg + offsetof(ArgLayout, inl) + offsetof(TrajInline, s));
    segs64 = (const TrajSegD*)(const char*)(kernarg + offsetof(ArgLayout, inl) + offsetof(TrajInline, d));
  }
  __shared__ TrajSeg32 lds[kMaxSegments];
  const uint32_t tid = threadIdx.x;
  const uint64_t n_tiles = (n + BLOCK - 1) / BLOCK;
  bool staged = false;
  for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const uint64_t base = t * BLOCK;
    const uint64_t i = base + tid;
    const bool alive = i < n && i >= head;
    const v4f p = load_point<NT>(in + (i < head ? head : (i < n ? i : n - 1)));  // dead lanes re-read a live point
    if (!staged) {
      for (uint32_t w = tid; w < n_seg * 8; w += BLOCK) reinterpret_cast<v4f*>(lds)[w] = reinterpret_cast<const v4f*>(segs)[w];
      __syncthreads();  // single-wave workgroup: orders the wave's own LDS writes before its reads
      staged = true;
    }
    uint32_t k = 0;
    for (uint32_t j = 1; j < n_seg; ++j) {  // interior knots
      const v4f kn = reinterpret_cast<const v4f*>(&lds[j])[7];  // {knot_cos, knot_sin, flags, knot_c}
      k += knot_ge(p.x, p.y, kn.w, kn.x, kn.y, __float_as_uint(kn.z)) ? 1u : 0u;
    }
    const uint32_t k0 = __builtin_amdgcn_readfirstlane(k);
    const uint32_t ks = __all(k == k0) ? k0 : k;  // uniform bracket -> uniform LDS address (broadcast), else per-lane gather
    bool redo;
    const v4f q = r2_traj_point<TIER>(p, lds[ks], redo);
    redo = redo && alive;
    const __amdgpu_buffer_rsrc_t rout = tile_rsrc(out + base, (n - base) * sizeof(v4f));  // clips the ragged tail
    if constexpr (NT & kStoreSc1) {
      if (i >= head && !redo) tile_store<NT>(rout, (uint32_t)(tid * sizeof(v4f)), q);
    } else {
      if (alive && !redo) store_point<NT>(out + i, q);
    }
    if constexpr (WRITE_IDX) {
      if (alive) __builtin_nontemporal_store(k, bracket_out + i);
    }
    traj_redo_lanes(redo, p, segs64, k, [&](v4f v) { tile_store<NT>(rout, (uint32_t)(tid * sizeof(v4f)), v); });
  }
}

}  // namespace kmc_dev

using namespace kmc_dev;

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      std::fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      std::exit(2);                                                               \
    }                                                                             \
  } while (0)

struct Variant {
  std::string name;
  std::function<void(hipStream_t, const v4f*, v4f*, uint64_t)> launch;
  int check_against = -1;  // index of the variant whose output must be bit-identical
};

static void twist_rec(TrajSeg32& r, const double phi[3], const double rho[3]) {
  const double c1[3] = {phi[1] * rho[2] - phi[2] * rho[1], phi[2] * rho[0] - phi[0] * rho[2], phi[0] * rho[1] - phi[1] * rho[0]};
  const double c2[3] = {phi[1] * c1[2] - phi[2] * c1[1], phi[2] * c1[0] - phi[0] * c1[2], phi[0] * c1[1] - phi[1] * c1[0]};
  r.phi_x = phi[0]; r.phi_y = phi[1]; r.phi_z = phi[2]; r.phi2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  r.rho_x = rho[0]; r.rho_y = rho[1]; r.rho_z = rho[2];
  r.c1_x = c1[0]; r.c1_y = c1[1]; r.c1_z = c1[2];
  r.c2_x = c2[0]; r.c2_y = c2[1]; r.c2_z = c2[2];
}

// n_seg segments whose interior knots sit at scan fractions j / n_seg (so every bracket gets an equal share of the ring);
// the last segment is the anchor's own (identity), the others carry a small rigid transform.
struct TrajSet {
  uint32_t n_seg;
  TrajSeg32* d_segs;
  TrajSegD* d_segs64;
  TrajInline inl;
};
static TrajSet build_traj(uint32_t n_seg) {
  TrajSet ts;
  ts.n_seg = n_seg;
  TrajSeg32 h[kMaxSegments];
  std::memset(h, 0, sizeof(h));
  std::memset(&ts.inl, 0, sizeof(ts.inl));
  for (uint32_t k = 0; k < n_seg; ++k) {
    TrajSeg32& r = h[k];
    const double phi[3] = {0.02 + 0.001 * k, 0.01, -0.1 + 0.01 * k}, rho[3] = {1.3 + 0.1 * k, 0.05, -0.02};
    twist_rec(r, phi, rho);
    const double ck = k == 0 ? -0.5 : (double)k / n_seg;
    r.g = 1.0f + 0.25f * k;
    r.s0 = (float)((0.5 - ck) * r.g);
    r.m00 = 0.9998f; r.m01 = -0.02f; r.m02 = 0.001f; r.tx = 0.3f;
    r.m10 = 0.02f; r.m11 = 0.9998f; r.m12 = -0.002f; r.ty = -0.01f;
    r.m20 = -0.001f; r.m21 = 0.002f; r.m22 = 1.0f; r.tz = 0.02f;
    const double alpha = M_PI - 2.0 * M_PI * ck;
    r.knot_cos = (float)std::cos(alpha); r.knot_sin = (float)std::sin(alpha);
    r.knot_c = (float)ck;
    r.flags = (k + 1 == n_seg ? kSegIdentity : 0u) | (ck <= 0.0 ? kKnotAlwaysGe : 0u);
    r.pre2 = 0.26f * (float)(rho[0] * rho[0] + rho[1] * rho[1] + rho[2] * rho[2]);
  }
  CK(hipMalloc((void**)&ts.d_segs, sizeof(h)));
  CK(hipMemcpy(ts.d_segs, h, sizeof(h), hipMemcpyHostToDevice));
  CK(hipMalloc((void**)&ts.d_segs64, kMaxSegments * sizeof(TrajSegD)));
  CK(hipMemset(ts.d_segs64, 0, kMaxSegments * sizeof(TrajSegD)));
  for (uint32_t k = 0; k < n_seg && k < (uint32_t)kInlineSegments; ++k) ts.inl.s[k] = h[k];
  return ts;
}

static FrameRec make_rec(double yaw) {
  FrameRec f;
  std::memset(&f, 0, sizeof(f));
  const double phi[3] = {0.02, 0.01, yaw}, rho[3] = {1.3, 0.05, -0.02};
  TrajSeg32 t;
  std::memset(&t, 0, sizeof(t));
  twist_rec(t, phi, rho);
  f.phi_x = t.phi_x; f.phi_y = t.phi_y; f.phi_z = t.phi_z; f.phi2 = t.phi2;
  f.rho_x = t.rho_x; f.rho_y = t.rho_y; f.rho_z = t.rho_z; f.s0 = 0.0f;
  f.c1_x = t.c1_x; f.c1_y = t.c1_y; f.c1_z = t.c1_z;
  f.c2_x = t.c2_x; f.c2_y = t.c2_y; f.c2_z = t.c2_z;
  f.pre2 = 0.26f * (float)(rho[0] * rho[0] + rho[1] * rho[1] + rho[2] * rho[2]);
  return f;
}

template <int TIER, bool ONE_PASS = false>
static Variant frame_variant(const char* label, double yaw) {
  Variant v;
  v.name = label;
  const FrameRec f = make_rec(yaw);
  v.launch = [f](hipStream_t s, const v4f* in, v4f* out, uint64_t n) {
    FrameRecD d;
    std::memset(&d, 0, sizeof(d));
    hipLaunchKernelGGL((deskew_frame_f32<TIER, 1, kPolicyDefault, false, 64, ONE_PASS>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, in, out, n, f, 0u, d);
  };
  return v;
}

enum TrajKind { kR2, kR3, kR3OnePass };  // kR3OnePass: round 3's kernel without its tile loop (what the library launches by default)
template <int KIND, bool INL>
static Variant traj_variant(const char* label, const TrajSet& ts) {
  Variant v;
  v.name = label;
  v.launch = [ts](hipStream_t s, const v4f* in, v4f* out, uint64_t n) {
    const dim3 grid((unsigned)((n + 63) / 64)), block(64);
    const TrajSeg32* segs = INL ? nullptr : ts.d_segs;
    const TrajSegD* segs64 = INL ? nullptr : ts.d_segs64;
    if constexpr (KIND == kR2) hipLaunchKernelGGL((r2_deskew_traj_f32<kSeries3, kPolicyDefault, false, INL>), grid, block, 0, s, in, out, n, segs, ts.n_seg, (uint32_t*)nullptr, 0u, segs64, ts.inl);
    if constexpr (KIND == kR3) hipLaunchKernelGGL((deskew_traj_f32<kSeries3, kPolicyDefault, false, INL>), grid, block, 0, s, in, out, n, segs, ts.n_seg, (uint32_t*)nullptr, 0u, segs64, ts.inl);
    if constexpr (KIND == kR3OnePass) hipLaunchKernelGGL((deskew_traj_f32<kSeries3, kPolicyDefault, false, INL, true>), grid, block, 0, s, in, out, n, segs, ts.n_seg, (uint32_t*)nullptr, 0u, segs64, ts.inl);
  };
  return v;
}

int main(int argc, char** argv) {
  const uint64_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : (1ull << 26);
  const int rounds = argc > 2 ? std::atoi(argv[2]) : 5;
  const int iters = argc > 3 ? std::atoi(argv[3]) : 10;
  const char* filter = argc > 4 ? argv[4] : "";
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  std::fprintf(stderr, "# device %s (%s), %d CUs, n=%llu points, rounds=%d iters=%d\n", prop.name, prop.gcnArchName, prop.multiProcessorCount,
               (unsigned long long)n, rounds, iters);
  constexpr int kBufs = 3;
  v4f* in[kBufs];
  v4f* out[kBufs];
  v4f* ref_out;
  hipStream_t s;
  CK(hipStreamCreate(&s));
  for (int b = 0; b < kBufs; ++b) {
    CK(hipMalloc((void**)&in[b], n * sizeof(v4f)));
    CK(hipMalloc((void**)&out[b], n * sizeof(v4f)));
    hipLaunchKernelGGL(synth_points<0>, dim3(prop.multiProcessorCount * 8), dim3(kBlock), 0, s, in[b], n, 0x4B4D43ull + b);
    CK(hipMemsetAsync(out[b], 0, n * sizeof(v4f), s));
  }
  CK(hipMalloc((void**)&ref_out, n * sizeof(v4f)));
  CK(hipStreamSynchronize(s));

  const TrajSet t2 = build_traj(2), t3 = build_traj(3), t6 = build_traj(6);
  std::vector<Variant> vs;
  vs.push_back(frame_variant<kSeries3>("frame_s3", -0.1));
  {
    Variant v = frame_variant<kSeries3, true>("frame_s3_one_pass", -0.1);
    v.check_against = 0;
    vs.push_back(v);
  }
  vs.push_back(frame_variant<kWide>("frame_wide", -2.9));
  vs.push_back(frame_variant<kTrig>("frame_trig_r3", -6.0));
  vs.push_back(frame_variant<kTrigOcml>("frame_trig_ocml_r2", -6.0));
  auto add_traj = [&](const char* tag, const TrajSet& ts, bool inline_ok) {
    const int ref = (int)vs.size();
    vs.push_back(traj_variant<kR2, false>((std::string("traj") + tag + "_r2_lds_table").c_str(), ts));
    auto add = [&](Variant v) { v.check_against = ref; vs.push_back(v); };
    add(traj_variant<kR3, false>((std::string("traj") + tag + "_r3_scalar_table").c_str(), ts));
    add(traj_variant<kR3OnePass, false>((std::string("traj") + tag + "_r3_scalar_table_one_pass").c_str(), ts));
    if (inline_ok) {
      add(traj_variant<kR2, true>((std::string("traj") + tag + "_r2_lds_inline").c_str(), ts));
      add(traj_variant<kR3, true>((std::string("traj") + tag + "_r3_scalar_inline").c_str(), ts));
      add(traj_variant<kR3OnePass, true>((std::string("traj") + tag + "_r3_scalar_inline_one_pass").c_str(), ts));
    }
  };
  add_traj("2", t2, true);
  add_traj("3", t3, true);
  add_traj("6", t6, false);
  if (filter[0]) {
    std::vector<Variant> keep;
    for (auto& v : vs)
      if (v.name.find(filter) != std::string::npos) keep.push_back(v);
    for (auto& v : keep) v.check_against = -1;
    vs = keep;
  }

  // bit checks first (on buffer 0)
  std::vector<std::string> verdict(vs.size(), "-");
  std::vector<v4f> h_ref(n), h_got(n);
  for (size_t k = 0; k < vs.size(); ++k) {
    if (vs[k].check_against < 0) continue;
    CK(hipMemsetAsync(ref_out, 0, n * sizeof(v4f), s));
    CK(hipMemsetAsync(out[0], 0, n * sizeof(v4f), s));
    vs[vs[k].check_against].launch(s, in[0], ref_out, n);
    vs[k].launch(s, in[0], out[0], n);
    CK(hipStreamSynchronize(s));
    CK(hipGetLastError());
    CK(hipMemcpy(h_ref.data(), ref_out, n * sizeof(v4f), hipMemcpyDeviceToHost));
    CK(hipMemcpy(h_got.data(), out[0], n * sizeof(v4f), hipMemcpyDeviceToHost));
    uint64_t diff = 0;
    for (uint64_t i = 0; i < n; ++i) diff += std::memcmp(&h_ref[i], &h_got[i], sizeof(v4f)) != 0;
    verdict[k] = diff ? ("DIFF:" + std::to_string(diff)) : "bit-identical";
  }

  struct Res { std::vector<float> ms; };
  std::vector<Res> res(vs.size());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  int rot = 0;
  for (int r = 0; r < rounds + 1; ++r) {  // round 0 = warm-up
    for (size_t k = 0; k < vs.size(); ++k) {
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < iters; ++i) {
        vs[k].launch(s, in[rot % kBufs], out[rot % kBufs], n);
        ++rot;
      }
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0) res[k].ms.push_back(ms / iters);
    }
  }
  std::printf("variant,us_median,us_min,gbps_median,gbps_best,frac_of_8TBps,bits_vs_r2_kernel\n");
  for (size_t k = 0; k < vs.size(); ++k) {
    auto& x = res[k].ms;
    std::sort(x.begin(), x.end());
    const float med = x[x.size() / 2], mn = x.front();
    const double bytes = 32.0 * n;
    std::printf("%s,%.1f,%.1f,%.1f,%.1f,%.4f,%s\n", vs[k].name.c_str(), med * 1e3, mn * 1e3, bytes / med * 1e-6, bytes / mn * 1e-6,
                bytes / med * 1e-6 / 8000.0, verdict[k].c_str());
  }
  return 0;
}

// kmc_device_math.hip.h -- per-lane arithmetic of the deskew kernels (gfx950, wave64).
//
// One lane = one LiDAR point.  What the reference does per point with two GetPoseAtTime calls
// (trajectory_interpolation.cpp:31-45: 2x Log incl. an SVD, 2x Exp, an inverse and three affine products) is
// algebraically   correction_i = Exp((x_i - x_r) * f),   f = Log(T_start^-1 * T_end)   ("identity"),
// so a lane only needs the frame constants {phi, rho, phi x rho, phi x (phi x rho), |phi|^2, 0.5 - x_r}
// and evaluates, with w = s*phi, u = |w|^2:
//     p' = p + A(u) (w x p) + B(u) (w x (w x p))          R(w) p        lie_algebra.cpp:22-35 (Rodrigues)
//            + s rho + B(u) (w x s rho) + C(u) (w x (w x s rho))        J(w) s rho   lie_algebra.cpp:51-65
//     A = sin(t)/t, B = (1-cos t)/t^2, C = (t - sin t)/t^3, t = sqrt(u)
// A, B, C are short even series in u (no trig, no divide, no branch) for the tiers the host selects from
// |phi|; a trig tier covers arbitrary rotation.  No MFMA: this is a streaming 16 B-in / 16 B-out transform.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace kmc_dev {

// One point {x, y, z, intensity}: a native 4 x f32 vector so that loads/stores are single dwordx4 ops and the
// non-temporal builtins accept it.
using v4f = float __attribute__((ext_vector_type(4)));

// Per-frame constants in device precision (64 B; kernarg -> SGPRs for the single-frame kernel,
// LDS-staged table for the batched kernel).
struct alignas(16) FrameRec {
  float phi_x, phi_y, phi_z, phi2;  // phi, |phi|^2
  float rho_x, rho_y, rho_z, s0;    // rho, s0 = 0.5 - x_req   (s = s0 - azimuth_turns)
  float c1_x, c1_y, c1_z, pre2;     // c1 = phi x rho; pre2 = kGuardPre * |rho|^2: first stage of the near-origin guard
  float c2_x, c2_y, c2_z, pad1;     // c2 = phi x (phi x rho)
};

// The same frame in f64 -- what the guarded redo below needs (128 B; kernarg of the single-frame kernels, a second
// table next to the BatchRec table for the batched kernel; only ever read -- through scalar loads -- by waves that contain a
// guarded lane).  c1, c2 and |phi|^2 come from the host so that the redo needs no wave-uniform f64 VALU work.
struct alignas(16) FrameRecD {
  double phi[3];
  double rho[3];
  double c1[3];   // phi x rho
  double c2[3];   // phi x (phi x rho)
  double phi2;
  double x_req;
  double pad[2];
};
static_assert(sizeof(FrameRecD) == 128, "FrameRecD must stay one 128-byte record");
static_assert(sizeof(FrameRec) == 64, "FrameRec must stay one 64-byte record");

// Coefficient tiers of the f32 kernels, each valid on the domain of the ones before it (a batch runs the tier of its widest
// frame).  theta = |phi| * max|s| is known on the host: 0.25 / 1 / 3.25 rad are the upper ends of the first three.
enum Tier : int { kSeries3 = 0, kSeries5 = 1, kWide = 2, kTrig = 3 };
constexpr double kThetaSeries3 = 0.25, kThetaSeries5 = 1.0, kThetaWide = 3.25;

// atan2(y, x) / (2 pi) in [-0.5, 0.5], i.e. the azimuth in turns.  timestamp_mocking.cpp:46 needs
// frac = (pi - atan2(y,x)) / 2pi = 0.5 - azimuth_turns.  Octant reduction + degree-7 polynomial in q^2
// (tools/gen_atan_coeffs.py; 1.0e-8 turns max error) -- ~22 VALU ops instead of ocml atan2f's ~55 plus a multiply.
// Signed zeros follow IEEE atan2: (+0,+0) -> 0, (-0, x<0 or x=-0) -> -0.5, (+0, x=-0) -> +0.5.
// The quotient q = min / max: round 3 takes it as min * v_rcp_f32(max) (1 ulp, 2 instructions) instead of the IEEE division
// (10): these kernels keep their SIMDs ~60 % busy, so VALU instructions are not free (kmc_kernels.hip.h, N-knot section).  The
// reciprocal neither accepts nor returns denormals, so a wave that holds a lane whose larger coordinate is outside
// [2^-126, 2^126] (zero, denormal, > 8.5e37, infinite, NaN -- nothing a LiDAR returns) redoes THAT lane's division the IEEE
// way: a wave-uniform, cold branch.  Error of the fast quotient: 1.5 ulp of q <= 1, i.e. < 3e-8 turns after the polynomial.
__device__ __forceinline__ float azimuth_turns(float x, float y) {
  const float ax = __builtin_fabsf(x);
  const float ay = __builtin_fabsf(y);
  const float mx = __builtin_fmaxf(ax, ay);
  const float mn = __builtin_fminf(ax, ay);
  const float rmx = __builtin_amdgcn_rcpf(mx);
  float q = mn * rmx;
  // the reciprocal is a normal number exactly when mx is in [2^-126, 2^126]: a zero or denormal mx gives infinity, an infinite or
  // huge one zero or a (flushed) denormal, NaN stays NaN -- ONE class test instead of two compares
  const bool tame = __builtin_amdgcn_classf(rmx, 0x100);
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(!tame) != 0, 0)) {  // cold; only the odd lanes take the other quotient, so a
                                                                        // point's result never depends on its neighbours
    float qi = mn / mx;              // IEEE divide (handles denormals); 0/0 -> NaN fixed below
    qi = (mx == 0.0f) ? 0.0f : qi;   // atan2(0, 0) = 0 like libm
    q = tame ? q : qi;
  }
  const float t = q * q;
  float p = -0.0007257134420797229f;
  p = __builtin_fmaf(p, t, 0.003784787142649293f);
  p = __builtin_fmaf(p, t, -0.009363045915961266f);
  p = __builtin_fmaf(p, t, 0.015706786885857582f);
  p = __builtin_fmaf(p, t, -0.022286929190158844f);
  p = __builtin_fmaf(p, t, 0.03177840635180473f);
  p = __builtin_fmaf(p, t, -0.053049229085445404f);
  p = __builtin_fmaf(p, t, 0.15915492177009583f);
  float r = p * q;                                   // [0, 1/8]
  r = (ay > ax) ? 0.25f - r : r;                     // [0, 1/4]
  r = (__float_as_uint(x) >> 31) ? 0.5f - r : r;     // signbit(x): [0, 1/2]
  return __builtin_copysignf(r, y);
}

struct Coef {
  float alpha;  // A(u) * s
  float beta;   // B(u) * s^2
  float gamma;  // C(u) * s^3
};

template <int TIER>
__device__ __forceinline__ Coef se3_coefficients(float s, float phi2) {
  const float s2 = s * s;
  const float u = s2 * phi2;  // theta^2
  float A, B, C;
  if constexpr (TIER == kSeries3) {  // theta <= 0.25: truncation < 5e-8 relative
    A = __builtin_fmaf(__builtin_fmaf(u, 1.0f / 120.0f, -1.0f / 6.0f), u, 1.0f);
    B = __builtin_fmaf(__builtin_fmaf(u, 1.0f / 720.0f, -1.0f / 24.0f), u, 0.5f);
    C = __builtin_fmaf(__builtin_fmaf(u, 1.0f / 5040.0f, -1.0f / 120.0f), u, 1.0f / 6.0f);
  } else if constexpr (TIER == kSeries5) {  // theta <= 1.0
    A = 1.0f / 362880.0f;
    A = __builtin_fmaf(A, u, -1.0f / 5040.0f);
    A = __builtin_fmaf(A, u, 1.0f / 120.0f);
    A = __builtin_fmaf(A, u, -1.0f / 6.0f);
    A = __builtin_fmaf(A, u, 1.0f);
    B = 1.0f / 3628800.0f;
    B = __builtin_fmaf(B, u, -1.0f / 40320.0f);
    B = __builtin_fmaf(B, u, 1.0f / 720.0f);
    B = __builtin_fmaf(B, u, -1.0f / 24.0f);
    B = __builtin_fmaf(B, u, 0.5f);
    C = 1.0f / 39916800.0f;
    C = __builtin_fmaf(C, u, -1.0f / 362880.0f);
    C = __builtin_fmaf(C, u, 1.0f / 5040.0f);
    C = __builtin_fmaf(C, u, -1.0f / 120.0f);
    C = __builtin_fmaf(C, u, 1.0f / 6.0f);
  } else if constexpr (TIER == kWide) {
    // theta <= 3.25: everything a frame between two poses can reach (|phi| <= pi out of Log, |s| <= 1).  A, B, C are entire
    // functions of u; interpolated at the Chebyshev nodes of [0, 3.25^2] (tools/gen_wide_coeffs.py): degree 6 / 5 / 5, fit
    // error 1.2e-9 / 6.8e-9 / 4.6e-10 absolute, f32 Horner error 1.4e-7 / 4.8e-8 / 1.6e-8 -- 16 fma, no sqrt, no division,
    // no sincos: the kernel stays on the HBM roofline where the trig tier below is VALU-bound.
    A = 1.3453622937920073e-10f;
    A = __builtin_fmaf(A, u, -2.4683949106929504e-08f);
    A = __builtin_fmaf(A, u, 2.7531152682058746e-06f);
    A = __builtin_fmaf(A, u, -0.0001984030968742445f);
    A = __builtin_fmaf(A, u, 0.008333316072821617f);
    A = __builtin_fmaf(A, u, -0.1666666567325592f);
    A = __builtin_fmaf(A, u, 1.0f);
    B = -1.752682554645446e-09f;
    B = __builtin_fmaf(B, u, 2.716992923978978e-07f);
    B = __builtin_fmaf(B, u, -2.478064016031567e-05f);
    B = __builtin_fmaf(B, u, 0.0013888373505324125f);
    B = __builtin_fmaf(B, u, -0.04166661947965622f);
    B = __builtin_fmaf(B, u, 0.5f);
    C = -1.3804320186938668e-10f;
    C = __builtin_fmaf(C, u, 2.4790514530081964e-08f);
    C = __builtin_fmaf(C, u, -2.754315346464864e-06f);
    C = __builtin_fmaf(C, u, 0.00019840920867864043f);
    C = __builtin_fmaf(C, u, -0.008333330042660236f);
    C = __builtin_fmaf(C, u, 0.1666666716337204f);
  } else {
    static_assert(TIER == kTrig, "four coefficient tiers");
    // any angle (a caller-supplied raw twist beyond 3.25 rad; two poses are never more than pi apart).  Half-angle forms (no
    // 1 - cos cancellation), series below theta^2 = 1/16.  Round 3: no ocml sincosf (its Payne-Hanek slow path rides along), no
    // IEEE divide or sqrt -- 1/theta is v_rsq_f32 + one Newton step, theta/2 is reduced to [-pi/4, pi/4] with a two-constant
    // Cody-Waite step (the fma keeps n * pio2_hi exact) and sin / cos come from the classic degree-7 / degree-8 kernels
    // (|error| < 7e-8 on the reduced range).  ~40 VALU instead of ~150: the tier is HBM-bound like the polynomial ones.  The
    // angle itself carries the f32 rounding of theta (6e-8 * theta rad) whatever the method, so the 1e-5 bar holds up to
    // theta ~ 50 rad and degrades linearly beyond.
    const float As = __builtin_fmaf(__builtin_fmaf(u, 1.0f / 120.0f, -1.0f / 6.0f), u, 1.0f);
    const float Bs = __builtin_fmaf(__builtin_fmaf(u, 1.0f / 720.0f, -1.0f / 24.0f), u, 0.5f);
    const float Cs = __builtin_fmaf(__builtin_fmaf(u, 1.0f / 5040.0f, -1.0f / 120.0f), u, 1.0f / 6.0f);
    const float uu = __builtin_fmaxf(u, 1e-12f);
    float inv_th = __builtin_amdgcn_rsqf(uu);                                   // 1 ulp
    inv_th = __builtin_fmaf(0.5f * inv_th, __builtin_fmaf(-uu * inv_th, inv_th, 1.0f), inv_th);  // Newton: ~0.6 ulp
    const float th = uu * inv_th;
    const float h = 0.5f * th;
    const float nq = __builtin_rintf(h * 0.63661977236758134f);                 // quadrant count of theta / 2
    float r = __builtin_fmaf(-nq, 1.57079637050628662109375f, h);               // pio2_hi (exact product inside the fma)
    r = __builtin_fmaf(-nq, -4.37113900018624283e-8f, r);                       // pio2_lo
    const float r2 = r * r;
    float sr = __builtin_fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f);
    sr = __builtin_fmaf(sr, r2, -1.6666654611e-1f);
    sr = __builtin_fmaf(sr * r2, r, r);                                         // sin r
    float cr = __builtin_fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f);
    cr = __builtin_fmaf(cr, r2, 4.166664568298827e-2f);
    cr = __builtin_fmaf(cr * r2, r2, __builtin_fmaf(-0.5f, r2, 1.0f));          // cos r
    // theta / 2 = r + nq pi/2:  sin^2(theta / 2) = sin^2 r (nq even) | cos^2 r (nq odd);  sin theta = +-2 sin r cos r
    const bool odd = (((int)nq) & 1) != 0;
    const float sh2 = odd ? cr * cr : sr * sr;
    const float sn = (odd ? -2.0f : 2.0f) * (sr * cr);
    const float inv_u = inv_th * inv_th;
    const float At = sn * inv_th;
    const float Bt = 2.0f * sh2 * inv_u;
    const float Ct = (th - sn) * inv_u * inv_th;
    const bool small = u < 0.0625f;
    A = small ? As : At;
    B = small ? Bs : Bt;
    C = small ? Cs : Ct;
  }
  Coef c;
  c.alpha = A * s;
  c.beta = B * s2;
  c.gamma = C * (s2 * s);
  return c;
}

// p' = Exp(s * f) * p  in the cross-product form, for a given s = x_i - x_anchor; (x, y, z) in, (x', y', z') out.
template <int TIER>
__device__ __forceinline__ v4f deskew_point_s(const v4f p, const float s, const FrameRec& f) {
  const Coef k = se3_coefficients<TIER>(s, f.phi2);
  // q1 = phi x p
  const float q1x = __builtin_fmaf(f.phi_y, p.z, -(f.phi_z * p.y));
  const float q1y = __builtin_fmaf(f.phi_z, p.x, -(f.phi_x * p.z));
  const float q1z = __builtin_fmaf(f.phi_x, p.y, -(f.phi_y * p.x));
  // q2 = phi x q1 + c1     (c1 = phi x rho rides along with the same beta)
  const float q2x = __builtin_fmaf(f.phi_y, q1z, __builtin_fmaf(-f.phi_z, q1y, f.c1_x));
  const float q2y = __builtin_fmaf(f.phi_z, q1x, __builtin_fmaf(-f.phi_x, q1z, f.c1_y));
  const float q2z = __builtin_fmaf(f.phi_x, q1y, __builtin_fmaf(-f.phi_y, q1x, f.c1_z));
  v4f o;
  o.x = __builtin_fmaf(k.gamma, f.c2_x, __builtin_fmaf(s, f.rho_x, __builtin_fmaf(k.beta, q2x, __builtin_fmaf(k.alpha, q1x, p.x))));
  o.y = __builtin_fmaf(k.gamma, f.c2_y, __builtin_fmaf(s, f.rho_y, __builtin_fmaf(k.beta, q2y, __builtin_fmaf(k.alpha, q1y, p.y))));
  o.z = __builtin_fmaf(k.gamma, f.c2_z, __builtin_fmaf(s, f.rho_z, __builtin_fmaf(k.beta, q2z, __builtin_fmaf(k.alpha, q1z, p.z))));
  o.w = p.w;  // intensity: bit-identical pass-through
  return o;
}

// fused: azimuth -> scan fraction -> s -> Exp(s f) p
template <int TIER>
__device__ __forceinline__ v4f deskew_point(const v4f p, const FrameRec& f) {
  return deskew_point_s<TIER>(p, f.s0 - azimuth_turns(p.x, p.y), f);  // s = frac - x_req
}

// ---- near-origin guard ------------------------------------------------------------------------------------------------
// The reference computes in f64 and casts to f32 only when the cloud is written (motion_compensation.cpp:13,
// data_io.cpp:300-310).  The f32 closed form above carries an ABSOLUTE error of a few f32 ulps of its largest operand,
// ~3 eps (|p| + |rho|): harmless relative to |p'| for every ordinary return, but a point that the ego-motion carries
// towards the sensor origin (p ~ -s rho) keeps that absolute error while |p'| shrinks, and the literal parity bar
// |p' - p'_ref| <= 1e-5 max(|p'_ref|, 1e-3) breaks below |p'| ~ |p| / 50 (round-1 soak: 2.0e-5 at 8.8 mm).
// Guard: a lane whose result lost more than a factor 4 against its operands,
//     16 |p'|^2 < |p|^2 + scale2,     scale2 = |rho|^2  (+ |t_k|^2 of the anchor transform for trajectory segments),
// is redone the reference's way -- f64 atan2, f64 exponential, one cast at the end.  Outside the guard
// |p'| >= (|p| + |rho|) / (4 sqrt 2), so the f32 error stays below ~6 x 3 eps = 1.1e-6 relative.  The branch is wave-uniform
// (ballot) and is never taken on real scans (nothing returns from within a metre of the sensor head).
// Two stages keep the common case at ~5 VALU instructions per point: the correction moves a point by at most |rho| (+ |t_k|)
// -- |J| <= 1, |s| <= 1, rotations preserve norms --, so the test above can only hold for 3 |p| < 5 (|rho| + |t_k|), hence for
//     |p'|^2 < pre2 = kGuardPre * scale2        (kGuardPre = 1.1 (25/9 + 1) / 16, resp. 1.1 (50/9 + 1) / 16 with a |t_k|),
// a per-frame constant; only a wave with a lane below it evaluates the exact test.  The margin of 10 % covers every rounding,
// so "stage 1 and stage 2" decides exactly like stage 2 alone, whoever computed pre2.
constexpr float kGuardPre = 0.26f, kGuardPreTraj = 0.46f;
__device__ __forceinline__ float norm2(const v4f p) { return __builtin_fmaf(p.x, p.x, __builtin_fmaf(p.y, p.y, p.z * p.z)); }
__device__ __forceinline__ bool lost_significance(const v4f p, const v4f o, float scale2) {
  return 16.0f * norm2(o) < norm2(p) + scale2;  // false for NaN: non-finite points keep the f32 result
}
// |rho|^2 of the record in device precision; ONE definition, so that a two-knot trajectory takes exactly the decisions of
// the two-pose kernels (their results are bit-identical, tests/test_trajectory.py)
__device__ __forceinline__ float rho_norm2(float rx, float ry, float rz) {
  return __builtin_fmaf(rx, rx, __builtin_fmaf(ry, ry, rz * rz));
}

// ---- N-knot trajectories (piecewise SE(3) geodesic through time-stamped poses) ------------------------------------
// One record per trajectory segment k = [knot k, knot k+1]; 128 B, staged into LDS by the workgroup.
//   p' = M_k * ( Exp((x_i - a_k) f_k) * p ),   x_i = position of the point's stamp inside the segment
// The segment that contains requested_time has M = I and a = x_req (the reference's single-geodesic case, bit for bit);
// every other segment has a = 0 and M_k = T(requested)^-1 * P_k, computed on the host in f64.
struct alignas(16) TrajSeg32 {
  float phi_x, phi_y, phi_z, phi2;
  float rho_x, rho_y, rho_z, s0;     // s = s0 - turns * g
  float c1_x, c1_y, c1_z, g;         // g = scan duration / segment duration
  float c2_x, c2_y, c2_z, pre2;      // near-origin guard, stage 1: kGuardPreTraj * (|rho|^2 + |t|^2), t = the anchor transform's translation
  float m00, m01, m02, tx;           // M_k rows with the translation in the 4th column
  float m10, m11, m12, ty;
  float m20, m21, m22, tz;
  // the START knot of the segment, one 16-byte slot so that a bracket test costs one ds_read_b128:
  float knot_cos, knot_sin;          // direction of the knot's azimuth alpha_k = pi - 2 pi c_k
  uint32_t flags;                    // kSegIdentity | kKnotAlwaysGe | kKnotNeverGe
  float knot_c;                      // scan fraction c_k of the knot
};
static_assert(sizeof(TrajSeg32) == 128, "TrajSeg32 must stay one 128-byte record");
constexpr uint32_t kSegIdentity = 1u, kKnotAlwaysGe = 2u, kKnotNeverGe = 4u;
constexpr int kMaxSegments = 16;

// Integer bracket test "scan fraction of (x, y) >= c_k" WITHOUT trig: half-plane tests against the knot's direction.
// Pure IEEE f32 mul/sub/add/compare in a fixed order with contraction off, so that the CPU restatement in the oracle
// executes the identical operations and the per-point bracket index is bit-exact.
__device__ __forceinline__ bool knot_ge(float x, float y, float knot_c, float ck, float sk, uint32_t flags) {
#pragma clang fp contract(off)
  if (flags & kKnotAlwaysGe) return true;
  if (flags & kKnotNeverGe) return false;
  const bool xneg = (__float_as_uint(x) >> 31) != 0;
  const bool yneg = (__float_as_uint(y) >> 31) != 0;
  bool lt;
  if (x == 0.0f && y == 0.0f) {  // atan2 on signed zeros: (+-0, +0) -> frac 0.5, (+0, -0) -> 0, (-0, -0) -> 1
    const float fs = xneg ? (yneg ? 1.0f : 0.0f) : 0.5f;
    lt = fs < knot_c;
  } else {
    const float cross = ck * y - sk * x;
    const float dot = ck * x + sk * y;
    if (knot_c <= 0.5f) lt = !yneg && (cross > 0.0f || (cross == 0.0f && dot < 0.0f));
    else lt = !yneg || cross > 0.0f || (cross == 0.0f && dot < 0.0f);
  }
  return !lt;
}

// The same predicate without control flow: every comparison is evaluated (they are cheap and have no side effects) and combined
// with bitwise logic, so the compiler emits ~12 VALU / SALU instructions instead of a dozen scalar branches.  Identical IEEE
// operations on identical operands in the same order -> the identical boolean.
__device__ __forceinline__ bool knot_ge_flat(float x, float y, float knot_c, float ck, float sk, uint32_t flags) {
#pragma clang fp contract(off)
  const bool xneg = (__float_as_uint(x) >> 31) != 0;
  const bool yneg = (__float_as_uint(y) >> 31) != 0;
  const bool origin = (x == 0.0f) & (y == 0.0f);
  const float fs = xneg ? (yneg ? 1.0f : 0.0f) : 0.5f;
  const float cross = ck * y - sk * x;
  const float dot = ck * x + sk * y;
  const bool before = (cross > 0.0f) | ((cross == 0.0f) & (dot < 0.0f));
  const bool lt_ring = (knot_c <= 0.5f) ? (!yneg & before) : (!yneg | before);
  const bool lt = origin ? (fs < knot_c) : lt_ring;
  const bool ge = !lt;
  return (flags & kKnotAlwaysGe) ? true : ((flags & kKnotNeverGe) ? false : ge);
}

// Coefficient tables of the f64 routines, in constant memory: read at wave-uniform addresses they arrive through scalar loads
// in SGPRs.  As literals the compiler materialises all ~40 of them in VGPR pairs and hoists them out of the kernels' tile loops
// (measured: 102-170 VGPRs instead of 34-60 for the f32 kernels, 122 instead of 84 for the Eigen-layout kernel).
__constant__ double kRedoTable[42] = {
    // [0..11]  atan(r) / r in r^2, highest degree first (degree-11 fit on r^2 <= tan^2(pi/8), tools/gen_atan_coeffs.py --f64)
    -1.78108398111324964e-02, 3.79703151591785637e-02, -5.03530597010270892e-02, 5.84692471107266312e-02,
    -6.66295840125903926e-02, 7.69204593125487474e-02, -9.09089684393908082e-02, 1.11111107461531272e-01,
    -1.42857142792741643e-01, 1.99999999999410899e-01, -3.33333333333331205e-01, 1.0,
    // [12..19] sin t / t in t^2:        (-1)^k / (2k+1)!, k = 7..0
    -1.0 / 1307674368000.0, 1.0 / 6227020800.0, -1.0 / 39916800.0, 1.0 / 362880.0, -1.0 / 5040.0, 1.0 / 120.0, -1.0 / 6.0, 1.0,
    // [20..27] (1 - cos t) / t^2:       (-1)^k / (2k+2)!
    -1.0 / 20922789888000.0, 1.0 / 87178291200.0, -1.0 / 479001600.0, 1.0 / 3628800.0, -1.0 / 40320.0, 1.0 / 720.0, -1.0 / 24.0, 0.5,
    // [28..35] (t - sin t) / t^3:       (-1)^k / (2k+3)!
    -1.0 / 355687428096000.0, 1.0 / 1307674368000.0, -1.0 / 6227020800.0, 1.0 / 39916800.0, -1.0 / 362880.0, 1.0 / 5040.0, -1.0 / 120.0, 1.0 / 6.0,
    // [36..41] tan(pi/8), pi/4, pi/2, pi, 2 pi, spare -- even these: as literals they are hoisted into VGPR pairs like the rest
    0.41421356237309503, 0.78539816339744831, 1.5707963267948966, 3.14159265358979323846, 6.28318530717958647692, 0.0};

using cdouble_p = const double __attribute__((address_space(4)))*;  // constant address space: uniform reads are scalar loads
// `t`, but not before `dep` exists: an empty volatile asm that ties a table pointer to the data flow (see the guarded redo below)
__device__ __forceinline__ cdouble_p after(cdouble_p t, double dep) {
  asm volatile("" : "+s"(t) : "v"(dep));
  return t;
}

// ---- f64 path (Eigen-layout API): closed form in double ---------------------------------------------
struct FrameRec64 {
  double phi[3], rho[3], c1[3], c2[3];
  double phi2;
  double x_req;
  double t_start, t_end;
  double inv_dur;  // 1 / (t_end - t_start), rounded once on the host (scan_offset_f64)
  int halvings;  // the series is evaluated at t / 2^halvings and doubled back: 0 for |phi| <= 0.5 rad (every vehicle), else whatever it takes
  int terms;     // series length: kShortSeriesTerms (5) up to kShortSeriesTheta rad per scan, else 8
};

// f64 trajectory segment (global-memory table, read by the f64 Eigen-layout trajectory kernel)
struct TrajSeg64 {
  FrameRec64 f;      // twist of the segment; f.x_req holds the anchor a_k, f.t_start / f.t_end / f.inv_dur the segment's time span
  double M[12];      // row-major 3x4 [R | t] applied after the exponential (identity for the anchor segment)
  int identity;
  int pad;
};

// alpha = A s, beta = B s^2, gamma = C s^3 with A = sin t / t, B = (1 - cos t) / t^2, C = (t - sin t) / t^3, t = |s phi|:
// Taylor series at t / 2^h, then h angle doublings
//     A(2x) = A(x) cos x,  cos x = 1 - x^2 B(x);   B(2x) = A(x)^2 / 2;   C(2x) = (C(x) + A(x) B(x)) / 4
// -- no cancellation anywhere, no trig, no divide; h and the series length are per-frame constants chosen on the host (wave-uniform):
//   5 terms  for |phi| <= 0.11 rad per scan (every vehicle: 1.1 rad/s of rotation; truncation < 1e-17 relative in all three series),
//   8 terms  up to t / 2^h <= 0.5 (truncation < 1e-19), h = 0 up to 0.5 rad per scan, whatever it takes beyond.
// Round 6: the series are summed in POWER form, smallest term first -- acc = fma(c_k, u^k, acc) with the powers of u shared by the three
// series -- instead of three Horner chains.  Horner's p = fma(p, u, c_k) has the CONSTANT as its addend, and the compiler's two-address
// v_fmac_f64 wants the addend in the destination VGPR pair: two v_mov_b32 per step, ~50 of the ~140 VALU instructions of a point were
// copies of constants.  In power form the constant is a multiplicand (an SGPR source of the same v_fmac), the accumulator stays where
// it is: 18 instructions for the 5-term tier, 32 for the 8-term one, against 63.  (Round 1 switched per point between a 7-term series
// and a half-angle sincos; ocml's f64 sincos carries its large-argument reduction along: 122 VGPRs.)
// The coefficients come out of kRedoTable ([12..19] A, [20..27] B, [28..35] C, highest degree first) through scalar loads pinned behind
// the powers they multiply (`after`): 24 SGPRs of table at a time.  As literals they were materialised with s_mov pairs and -- in the
// STREAMED kernels, whose tile loop invites hoisting -- kept alive across the loop: 29 SGPR spills in deskew_f64cols<true>.
constexpr int kShortSeriesTerms = 5;
constexpr double kShortSeriesTheta = 0.11;  // rad per scan up to which 5 terms are exact to f64 (0.11^10 / 11! = 6.5e-18)
__device__ __forceinline__ void se3_series_f64(double u, int terms, double& A, double& B, double& C) {
  const double u2 = u * u, u3 = u2 * u, u4 = u2 * u2;
  A = 0.0; B = 0.0; C = 0.0;
  if (terms != kShortSeriesTerms) {  // (wave-uniform: a per-frame constant) degrees 7, 6, 5
    const double u5 = u4 * u, u6 = u3 * u3, u7 = u4 * u3;
    cdouble_p t = after((cdouble_p)kRedoTable, u7);
    A = t[12] * u7; B = t[20] * u7; C = t[28] * u7;
    A = __builtin_fma(t[13], u6, A); B = __builtin_fma(t[21], u6, B); C = __builtin_fma(t[29], u6, C);
    A = __builtin_fma(t[14], u5, A); B = __builtin_fma(t[22], u5, B); C = __builtin_fma(t[30], u5, C);
  }
  cdouble_p t = after((cdouble_p)kRedoTable, u4 + A);  // degrees 4 .. 0, smallest term first
  A = __builtin_fma(t[15], u4, A); B = __builtin_fma(t[23], u4, B); C = __builtin_fma(t[31], u4, C);
  A = __builtin_fma(t[16], u3, A); B = __builtin_fma(t[24], u3, B); C = __builtin_fma(t[32], u3, C);
  A = __builtin_fma(t[17], u2, A); B = __builtin_fma(t[25], u2, B); C = __builtin_fma(t[33], u2, C);
  A = __builtin_fma(t[18], u, A);  B = __builtin_fma(t[26], u, B);  C = __builtin_fma(t[34], u, C);
  A += 1.0; B += 0.5; C += t[35];
}
__device__ __forceinline__ void se3_coefficients_f64(double s, double phi2, int halvings, int terms, double& alpha, double& beta, double& gamma) {
  const double s2 = s * s;
  double u = __builtin_ldexp(s2 * phi2, -2 * halvings);  // (t / 2^h)^2
  double A, B, C;
  se3_series_f64(u, terms, A, B, C);
  for (int k = 0; k < halvings; ++k) {
    const double cosx = __builtin_fma(-u, B, 1.0);
    C = __builtin_ldexp(__builtin_fma(A, B, C), -2);
    B = 0.5 * A * A;
    A = A * cosx;
    u *= 4.0;
  }
  alpha = A * s;
  beta = B * s2;
  gamma = C * (s2 * s);
}

// s = FractionOfTrajectory(t) - x_req (trajectory_interpolation.cpp:49-51) as ONE fma with the frame's 1 / (t2 - t1) -- round 6; until then a
// true divide like the reference's, ~14 VALU instructions with a quarter-rate v_rcp_f64 among them.  The fraction differs from the quotient
// by at most 1.5 ulp (2e-16 of a scan, 2e-17 s), five orders below the f64 parity bar (1e-11); which stamps are IN RANGE is decided on the
// stamps themselves, exactly (TimeIsInRange, :47).
__device__ __forceinline__ double scan_offset_f64(double t, double t_start, double inv_dur, double x_req) {
  return __builtin_fma(t - t_start, inv_dur, -x_req);
}

__device__ __forceinline__ void deskew_point_f64(double x, double y, double z, double w, double s, const FrameRec64& f,
                                                 double& ox, double& oy, double& oz) {
  double al, be, ga;
  se3_coefficients_f64(s, f.phi2, f.halvings, f.terms, al, be, ga);
  const double q1x = f.phi[1] * z - f.phi[2] * y;
  const double q1y = f.phi[2] * x - f.phi[0] * z;
  const double q1z = f.phi[0] * y - f.phi[1] * x;
  const double q2x = f.phi[1] * q1z - f.phi[2] * q1y;
  const double q2y = f.phi[2] * q1x - f.phi[0] * q1z;
  const double q2z = f.phi[0] * q1y - f.phi[1] * q1x;
  // translation J(w) * s*rho, scaled by the homogeneous coordinate like Affine3d * Vector4d (motion_compensation.cpp:13)
  const double tx = s * f.rho[0] + be * f.c1[0] + ga * f.c2[0];
  const double ty = s * f.rho[1] + be * f.c1[1] + ga * f.c2[1];
  const double tz = s * f.rho[2] + be * f.c1[2] + ga * f.c2[2];
  ox = x + al * q1x + be * q2x + tx * w;
  oy = y + al * q1y + be * q2y + ty * w;
  oz = z + al * q1z + be * q2z + tz * w;
}

// ---- the guarded redo (see lost_significance above): one point the reference's way, f64 throughout ---------------------
// frac = (pi - atan2(y, x)) / 2 pi in f64 (timestamp_mocking.cpp:46); s = frac - x_req (trajectory_interpolation.cpp:49-51);
// p' = Exp(s f) p in f64 (lie_algebra.cpp:83-92 in closed form); ONE cast to f32 at the end (data_io.cpp:300-310).
// Written for a SMALL register footprint, not for speed: the redo is compiled into every f32 kernel, and those must keep
// their 8 waves per SIMD (<= 64 VGPRs).  ocml's f64 atan2 alone takes 60 VGPRs and its sincos drags in the large-argument
// reduction; the two routines below are straight Horner chains (constants through SGPRs) and fit next to the f32 path.
//
// All tables of the redo -- the coefficients of kRedoTable and the frame / segment records -- are read through constant-address-space
// pointers (uniform address -> scalar loads into SGPRs, no VGPRs for constants) that pass through an empty volatile asm
// tied to the previous phase's result (`after`).  That pins every group of table reads behind the arithmetic that precedes
// it, so that at most 12-24 SGPRs of table are live at any time.  Left to itself the compiler loads the 36 coefficients and
// the whole record up front (80+ SGPRs: spills, and a private segment for the kernel).
__device__ __forceinline__ uint32_t opaque_uniform(uint32_t v) {  // a wave-uniform value the optimiser cannot trace back
  asm volatile("" : "+s"(v));
  return v;
}
template <typename T>
__device__ __forceinline__ cdouble_p as_constant(const T* rec) {  // records are written by the host before the launch: constant
  return (cdouble_p)(uintptr_t)rec;
}

// atan2(y, x) in f64, |error| <= 5e-16 rad (2 M random points against long double, tools/gen_atan_coeffs.py --f64):
// octant reduction, a second reduction at tan(pi/8) folded into the ONE division, degree-11 polynomial in r^2 on
// |r| <= tan(pi/8).  Signed zeros like libm: (+-0, x >= +0) -> +-0, (+-0, x <= -0) -> +-pi; atan2(0, 0) = 0.
__device__ __forceinline__ double atan2_f64_lean(double y, double x) {
  const double ax = __builtin_fabs(x), ay = __builtin_fabs(y);
  const double mx = __builtin_fmax(ax, ay), mn = __builtin_fmin(ax, ay);
  cdouble_p t = after((cdouble_p)kRedoTable, mx);
  const bool big = mn > t[36] * mx;  // atan q = pi/4 + atan((q - 1) / (q + 1))
  const double num = big ? mn - mx : mn;
  const double den = big ? mn + mx : mx;
  double r = num / den;
  r = (den == 0.0) ? 0.0 : r;
  const double z = r * r;
  t = after(t, z);
  double p = t[0];
#pragma unroll
  for (int k = 1; k < 6; ++k) p = __builtin_fma(p, z, t[k]);
  t = after(t, p);
#pragma unroll
  for (int k = 6; k < 12; ++k) p = __builtin_fma(p, z, t[k]);
  double a = r * p;
  t = after(t, a);
  a = big ? t[37] + a : a;
  a = (ay > ax) ? t[38] - a : a;
  a = (__builtin_signbit(x)) ? t[39] - a : a;
  return __builtin_copysign(a, y);
}
// (pi - atan2(y, x)) / (2 pi), timestamp_mocking.cpp:46 -- a true division like the reference's
__device__ __forceinline__ double scan_fraction_f64_lean(double x, double y) {
  const double a = atan2_f64_lean(y, x);
  cdouble_p t = after((cdouble_p)kRedoTable, a);
  return (t[39] - a) / t[40];
}

// alpha = A s, beta = B s^2, gamma = C s^3 with A = sin t / t, B = (1 - cos t) / t^2, C = (t - sin t) / t^3, t = |s phi|:
// 8-term series at t / 8 (exact to 1e-19 for t <= 4), then three angle doublings
//     A(2x) = A(x) cos x,  cos x = 1 - x^2 B(x);   B(2x) = A(x)^2 / 2;   C(2x) = (C(x) + A(x) B(x)) / 4
// (no cancellation anywhere, no trig, no divide, no branch).  One series after the other: 16 SGPRs of coefficients at a time.
__device__ __forceinline__ void se3_coefficients_f64_lean(double s, double phi2, double& alpha, double& beta, double& gamma) {
  const double s2 = s * s;
  double u = __builtin_ldexp(s2 * phi2, -6);  // (t / 8)^2
  cdouble_p t = after((cdouble_p)kRedoTable, u);
  double A = t[12];
#pragma unroll
  for (int k = 1; k < 8; ++k) A = __builtin_fma(A, u, t[12 + k]);
  t = after(t, A);
  double B = t[20];
#pragma unroll
  for (int k = 1; k < 8; ++k) B = __builtin_fma(B, u, t[20 + k]);
  t = after(t, B);
  double C = t[28];
#pragma unroll
  for (int k = 1; k < 8; ++k) C = __builtin_fma(C, u, t[28 + k]);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double cosx = __builtin_fma(-u, B, 1.0);
    C = __builtin_ldexp(__builtin_fma(A, B, C), -2);
    B = 0.5 * A * A;
    A = A * cosx;
    u *= 4.0;
  }
  alpha = A * s;
  beta = B * s2;
  gamma = C * (s2 * s);
}

// Exp(s f) p in f64.  `rec`: the record's leading doubles, the layout FrameRecD and TrajSegD share:
// phi[0..2], rho[3..5], c1[6..8], c2[9..11], phi2[12].
constexpr int kRecPhi = 0, kRecRho = 3, kRecC1 = 6, kRecC2 = 9, kRecPhi2 = 12;
__device__ __forceinline__ void exp_apply_f64_lean(cdouble_p rec, double s, double phi2, double x, double y, double z, double& ox,
                                                   double& oy, double& oz) {
  double al, be, ga;
  se3_coefficients_f64_lean(s, phi2, al, be, ga);
  rec = after(rec, ga);  // phi and c1
  const double q1x = rec[kRecPhi + 1] * z - rec[kRecPhi + 2] * y;
  const double q1y = rec[kRecPhi + 2] * x - rec[kRecPhi + 0] * z;
  const double q1z = rec[kRecPhi + 0] * y - rec[kRecPhi + 1] * x;
  const double q2x = rec[kRecPhi + 1] * q1z - rec[kRecPhi + 2] * q1y + rec[kRecC1 + 0];
  const double q2y = rec[kRecPhi + 2] * q1x - rec[kRecPhi + 0] * q1z + rec[kRecC1 + 1];
  const double q2z = rec[kRecPhi + 0] * q1y - rec[kRecPhi + 1] * q1x + rec[kRecC1 + 2];
  const double hx = x + al * q1x + be * q2x;
  const double hy = y + al * q1y + be * q2y;
  const double hz = z + al * q1z + be * q2z;
  rec = after(rec, hz);  // rho and c2
  ox = hx + s * rec[kRecRho + 0] + ga * rec[kRecC2 + 0];
  oy = hy + s * rec[kRecRho + 1] + ga * rec[kRecC2 + 1];
  oz = hz + s * rec[kRecRho + 2] + ga * rec[kRecC2 + 2];
}

// one point of a two-pose frame; `rec` addresses a FrameRecD
static_assert(offsetof(FrameRecD, phi) == 8 * kRecPhi && offsetof(FrameRecD, rho) == 8 * kRecRho && offsetof(FrameRecD, c1) == 8 * kRecC1 &&
                  offsetof(FrameRecD, c2) == 8 * kRecC2 && offsetof(FrameRecD, phi2) == 8 * kRecPhi2 && offsetof(FrameRecD, x_req) == 8 * 13,
              "exp_apply_f64_lean / deskew_point_redo_f64 index FrameRecD as an array of doubles");
__device__ __forceinline__ v4f deskew_point_redo_f64(const v4f p, cdouble_p rec) {
  const double x = (double)p.x, y = (double)p.y, z = (double)p.z;
  const double frac = scan_fraction_f64_lean(x, y);
  rec = after(rec, frac);  // phi2 and x_req
  double ox, oy, oz;
  exp_apply_f64_lean(rec, frac - rec[13], rec[kRecPhi2], x, y, z, ox, oy, oz);
  return v4f{(float)ox, (float)oy, (float)oz, p.w};
}

// The guard is split in two so that the f64 work sits AFTER the wave's regular store, when the f32 result and the f32 frame
// record are dead (register pressure: the f32 kernels must stay at <= 64 VGPRs):
//   1. hot:  o = deskew_point(p, f);  redo = needs_redo(p, o, f);  lanes with !redo store o;
//   2. cold: redo_lanes(redo, p, tbl, fi, store) -- the flagged lanes are recomputed in f64 and stored through `store`.
// `tbl[fi]` is the lane's frame in f64.  The flagged lanes are redone frame by frame (a waterfall over the distinct `fi`
// among them -- one turn unless the wave straddles a frame boundary) so that the record is always read at a wave-uniform
// address.
__device__ __forceinline__ bool needs_redo(const v4f p, const v4f o, const FrameRec& f) {
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(norm2(o) < f.pre2) == 0, 1)) return false;  // stage 1, wave-uniform
  return lost_significance(p, o, rho_norm2(f.rho_x, f.rho_y, f.rho_z));
}
template <typename STORE>
__device__ __forceinline__ void redo_lanes(bool redo, const v4f p, const FrameRecD* __restrict__ tbl, uint32_t fi, STORE&& store) {
  uint64_t todo = __builtin_amdgcn_ballot_w64(redo);
  while (__builtin_expect(todo != 0, 0)) {  // wave-uniform, cold
    const uint32_t fu = (uint32_t)__builtin_amdgcn_readlane((int)fi, __builtin_ctzll(todo));
    // the record index is a copy of fu made through `opaque_uniform`: inside `if (fi == fu)` the optimiser would otherwise
    // replace the uniform fu by the per-lane fi, and the record would be gathered per lane (32 VGPRs) instead of scalar-loaded
    const uint32_t fu_idx = opaque_uniform(fu);
    const bool mine = redo && fi == fu;
    if (mine) store(deskew_point_redo_f64(p, as_constant(tbl + fu_idx)));
    todo &= ~__builtin_amdgcn_ballot_w64(mine);
  }
}
// the single-frame kernels: ONE record, at `rec` (in the kernel's own argument segment, see deskew_frame_f32)
template <typename STORE>
__device__ __forceinline__ void redo_lanes(bool redo, const v4f p, cdouble_p rec, STORE&& store) {
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(redo) != 0, 0)) {  // wave-uniform, cold
    if (redo) store(deskew_point_redo_f64(p, rec));
  }
}

// f64 twin of TrajSeg32 for the guarded redo of the N-knot kernels (240 B, global memory, read by guarded waves only)
struct alignas(16) TrajSegD {
  double phi[3];
  double rho[3];
  double c1[3];
  double c2[3];
  double phi2;
  double c, g;         // scan fraction of the segment's start knot; scan duration / segment duration
  double a;            // anchor: x_req inside the anchor's own segment, 0 elsewhere;  s = (frac - c) g - a
  double M[12];        // row-major 3x4 anchor transform
  uint64_t identity;   // M = I (the segment that contains requested_time)
  double pad;
};
static_assert(sizeof(TrajSegD) == 240, "TrajSegD must stay 240 bytes");
static_assert(offsetof(TrajSegD, phi) == 8 * kRecPhi && offsetof(TrajSegD, rho) == 8 * kRecRho && offsetof(TrajSegD, c1) == 8 * kRecC1 &&
                  offsetof(TrajSegD, c2) == 8 * kRecC2 && offsetof(TrajSegD, phi2) == 8 * kRecPhi2 && offsetof(TrajSegD, c) == 8 * 13 &&
                  offsetof(TrajSegD, g) == 8 * 14 && offsetof(TrajSegD, a) == 8 * 15 && offsetof(TrajSegD, M) == 8 * 16 &&
                  offsetof(TrajSegD, identity) == 8 * 28,
              "traj_point_redo_f64 indexes TrajSegD as an array of doubles");

__device__ __forceinline__ v4f traj_point_redo_f64(const v4f p, cdouble_p rec) {
  const double x = (double)p.x, y = (double)p.y, z = (double)p.z;
  const double frac = scan_fraction_f64_lean(x, y);
  rec = after(rec, frac);  // phi2, c, g, a
  const double s = (frac - rec[13]) * rec[14] - rec[15];  // two knots on the scan: (frac - 0) * 1 - x_req, exactly deskew_point_redo_f64
  double qx, qy, qz;
  exp_apply_f64_lean(rec, s, rec[kRecPhi2], x, y, z, qx, qy, qz);
  rec = after(rec, qz);  // identity flag and M
  if (__builtin_bit_cast(uint64_t, rec[28]) == 0) {
    const double rx = rec[16] * qx + rec[17] * qy + rec[18] * qz + rec[19];
    const double ry = rec[20] * qx + rec[21] * qy + rec[22] * qz + rec[23];
    const double rz = rec[24] * qx + rec[25] * qy + rec[26] * qz + rec[27];
    qx = rx; qy = ry; qz = rz;
  }
  return v4f{(float)qx, (float)qy, (float)qz, p.w};
}

// ------------------------------------------------------------------------------------------------
// next row N4: LiDAR -> image projection (camera_model.cpp:5-95 without the OpenCV drawing).
// f64, every product and sum individually rounded in the reference's order (it builds with plain -O3: no FMA), IEEE
// division: the integer pixel coordinates and colour bytes are BIT-EXACT against the CPU restatement the tests use.
// ------------------------------------------------------------------------------------------------
struct CameraRigRec {
  double T[12];      // tf_c00_lo, row-major 3x4
  double R[9];       // R_rect_00, row-major
  double P[4][12];   // P_rect_0c, row-major 3x4
  double max_range;  // camera_model.hpp:8
  double range_den;  // max_range - 0.01, camera_model.cpp:28
  double range_rcp;  // ~ 1 / range_den: first guess of the colour ramp's quotient, confirmed or redone exactly per point
  uint64_t same_den; // bit c (1..3): P[c][11] is bit for bit P[c-1][11] -- camera c divides by the same h2 as camera c-1 (KITTI: cameras 0 and 1)
};
using v2i = int __attribute__((ext_vector_type(2)));
using v4i = int __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int trunc_i32(double v) {  // cv::Point(double, double): cvttsd2si semantics
  return (v > -2147483649.0 && v < 2147483648.0) ? (int)v : (int)0x80000000;
}
__device__ __forceinline__ uint32_t sat_u8(double v) {  // cv::saturate_cast<uchar>(double): cvRound (half to even) + clamp, NaN -> 0
  // v_max_f64 / v_min_f64 return the other operand for a NaN: the clamp is also the NaN guard (two instructions where the
  // compare-and-select form took seven)
  return (uint32_t)__builtin_fmin(__builtin_fmax(__builtin_rint(v), 0.0), 255.0);
}

// (trunc(a / d), trunc(b / d)) with the IEEE quotients' values, without two IEEE divisions on the common path: the
// quotients are first approximated through ONE reciprocal refined by ONE Newton step (v_rcp_f64 is good to 2^-23 at worst per
// the ISA manual, 2^-24.4 measured over 2^33 denominators, tools/rcp_probe.hip: after the step 2^-48.6 measured, < 2^-44 by the bound); their
// truncations equal the exact ones unless an integer lies within that error of the approximation, i.e. unless the
// approximation sits within 2^-36 * max(1, |q|) of an integer (or is not a finite value inside the int range, whose ends are
// integers too) -- a margin 2^8 wider than the error.  Returns whether both truncations are certain; the caller redoes the point
// with real divisions otherwise (about once in 1e8 coordinates).  One margin for the pair, from the larger quotient: wider than
// needed for the smaller one, which only sends a few more points to the exact path.  NaN or infinite quotients (d = 0) fail the
// `>` comparisons and are therefore never "certain".
// The verdict comes back as a LANE MASK (ballot): the caller combines the masks of the four cameras with scalar ANDs; as `bool`s
// the compiler materialised every one of them in a VGPR and combined them with 16-bit vector logic (~25 VALU instructions).
__device__ __forceinline__ double refined_rcp(double d) {
  const double r = __builtin_amdgcn_rcp(d);
  return __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
}
__device__ __forceinline__ uint64_t trunc_quotients_with(double a, double b, double r, int& ta, int& tb) {  // r = refined_rcp(d)
  const double qa = a * r, qb = b * r;
  constexpr double kEps = 0x1p-36;  // the margin kEps (1 + |q|) >= kEps max(1, |q|): one fma instead of max + mul, a little more cautious
  const double big = __builtin_fmax(__builtin_fabs(qa), __builtin_fabs(qb));
  const double margin = __builtin_fma(big, kEps, kEps);
  const uint64_t sure_a = __builtin_amdgcn_ballot_w64(__builtin_fabs(qa - __builtin_rint(qa)) > margin);
  const uint64_t sure_b = __builtin_amdgcn_ballot_w64(__builtin_fabs(qb - __builtin_rint(qb)) > margin);
  const uint64_t in_range = __builtin_amdgcn_ballot_w64(big < 2147483000.0);
  ta = (int)qa;
  tb = (int)qb;
  return sure_a & sure_b & in_range;
}
__device__ __forceinline__ uint64_t trunc_quotients_fast(double a, double b, double d, int& ta, int& tb) {
  return trunc_quotients_with(a, b, refined_rcp(d), ta, tb);
}

// P_rect_c * r for one camera, :9.  STRUCTURED: the products with the literal 0s and 1 of a pinhole matrix are skipped.
template <bool STRUCTURED>
__device__ __forceinline__ void camera_rows(cdouble_p P, const double r[3], double h[3]) {
#pragma clang fp contract(off)
  if constexpr (STRUCTURED) {
    h[0] = (P[0] * r[0] + P[2] * r[2]) + P[3];
    h[1] = (P[5] * r[1] + P[6] * r[2]) + P[7];
    h[2] = r[2] + P[11];
  } else {
#pragma unroll
    for (int k = 0; k < 3; ++k) h[k] = ((P[4 * k] * r[0] + P[4 * k + 1] * r[1]) + P[4 * k + 2] * r[2]) + P[4 * k + 3];
  }
}

// The rig is 71 doubles = 142 SGPRs' worth of wave-uniform constants: more than a wave has.  As a by-value kernel argument the
// compiler preloads all of it and spills (round 1: 124 v_writelane + 124 v_readlane in a ~700-instruction kernel).  It is
// therefore read through the kernel-argument segment in phases (`after`, see the guarded redo): T, R and the range constants up
// front, cameras 0-1 once the camera-frame point exists, cameras 2-3 once the rectified point exists.
constexpr int kRigT = 0, kRigR = 12, kRigP = 21, kRigMaxRange = 69, kRigRangeDen = 70, kRigRangeRcp = 71, kRigSameDen = 72;
static_assert(offsetof(CameraRigRec, T) == 8 * kRigT && offsetof(CameraRigRec, R) == 8 * kRigR && offsetof(CameraRigRec, P) == 8 * kRigP &&
                  offsetof(CameraRigRec, max_range) == 8 * kRigMaxRange && offsetof(CameraRigRec, range_den) == 8 * kRigRangeDen &&
                  offsetof(CameraRigRec, range_rcp) == 8 * kRigRangeRcp && offsetof(CameraRigRec, same_den) == 8 * kRigSameDen,
              "project_point indexes CameraRigRec as an array of doubles");

// -> validity; uv[c] = pixel cv::circle would be centred on; bgrv = {255-cs, cs, 255-cs, 1} packed little-endian.
// STRUCTURED: every P_rect has the pinhole shape [fx 0 cx tx; 0 fy cy ty; 0 0 1 tz] (all KITTI calibrations do); skipping
// its zeros changes no result for finite coordinates (x + (+-0) = x, 1 * x = x).  A point whose rectified coordinates
// are not finite (0 * inf = NaN matters there) or whose quotients are not certain takes the plain sequence below.
// Most of a scan is behind the cameras or beyond max_range, and neighbouring points share that fate: a wave in which no
// lane passes the test of :21-24 skips the four cameras altogether (wave-uniform branch).
// RIG: 0 general P_rect, 1 pinhole P_rect (kRigPinhole), 2 pinhole with the SAME fx, cx, fy, cy in all four cameras
// (kRigSharedIntrinsics; every KITTI rectified rig): fx x + cx z and fy y + cy z are then the same rounded values for the four
// cameras and are computed once -- identical operations on identical operands, so still bit-exact.
constexpr int kRigGeneral = 0, kRigPinhole = 1, kRigSharedIntrinsics = 2;
template <int RIG>
__device__ __forceinline__ bool project_point(double x, double y, double z, cdouble_p g, v2i uv[4], uint32_t& bgrv) {
#pragma clang fp contract(off)
  constexpr bool STRUCTURED = RIG != kRigGeneral;
  double c[3], r[3];
  const cdouble_p T = g + kRigT, R = g + kRigR;
#pragma unroll
  for (int k = 0; k < 3; ++k) c[k] = ((T[4 * k] * x + T[4 * k + 1] * y) + T[4 * k + 2] * z) + T[4 * k + 3];  // :75
#pragma unroll
  for (int k = 0; k < 3; ++k) r[k] = ((R[3 * k] * c[0] + R[3 * k + 1] * c[1]) + R[3 * k + 2] * c[2]) + 0.0;  // :81
  const bool valid = !((r[2] < 0.01) || (r[2] > g[kRigMaxRange]) || (r[1] > 1.25));                         // :21-24
  const uint64_t valid_m = __builtin_amdgcn_ballot_w64(valid);
  if (valid_m == 0) {
#pragma unroll
    for (int cam = 0; cam < 4; ++cam) uv[cam].x = uv[cam].y = (int)0x80000000;
    bgrv = 0u;
    return false;
  }
  // lanes whose truncations are certain, as a lane mask (scalar logic; see trunc_quotients_fast)
  uint64_t sure = __builtin_amdgcn_ballot_w64(__builtin_fabs(r[0]) < __builtin_inf()) & __builtin_amdgcn_ballot_w64(__builtin_fabs(r[1]) < __builtin_inf()) &
                  __builtin_amdgcn_ballot_w64(__builtin_fabs(r[2]) < __builtin_inf());
  cdouble_p P = after(g, c[0]) + kRigP;  // cameras 0 and 1: their loads may start as soon as the camera-frame point exists
  if constexpr (RIG == kRigSharedIntrinsics) {
    const double a0 = P[0] * r[0] + P[2] * r[2];  // :9, rows 0 and 1 of P_rect without their last column
    const double a1 = P[5] * r[1] + P[6] * r[2];
    // cameras whose h2 = z + tz is the same rounded value as the previous camera's (the host compared the tz bit for bit: KITTI's
    // cameras 0 and 1) reuse its refined reciprocal: identical operands, identical operations, one v_rcp_f64 + two fma less (round 4:
    // the all-drawn pattern is f64-VALU-bound)
    const uint32_t same = (uint32_t)__builtin_bit_cast(uint64_t, g[kRigSameDen]);
    double rc = 0.0;
#pragma unroll
    for (int cam = 0; cam < 4; ++cam) {
      if (cam == 0 || !((same >> cam) & 1u)) rc = refined_rcp(r[2] + P[12 * cam + 11]);  // wave-uniform
      int tu, tv;
      sure &= trunc_quotients_with(a0 + P[12 * cam + 3], a1 + P[12 * cam + 7], rc, tu, tv);
      uv[cam].x = tu;
      uv[cam].y = tv;
    }
  } else {
#pragma unroll
    for (int cam = 0; cam < 4; ++cam) {
      if (cam == 2) P = after(g, r[0]) + kRigP;  // cameras 2 and 3
      double h[3];
      camera_rows<STRUCTURED>(P + 12 * cam, r, h);
      int tu, tv;
      sure &= trunc_quotients_fast(h[0], h[1], h[2], tu, tv);
      uv[cam].x = tu;
      uv[cam].y = tv;
    }
  }
  const uint64_t redo = valid_m & ~sure;
  if (__builtin_expect(redo != 0, 0)) {  // wave-uniform, cold
    if ((redo >> __lane_id()) & 1) {     // the reference's own sequence: general rows, two IEEE divisions (:9, :12, :31)
#pragma unroll 1
      for (int cam = 0; cam < 4; ++cam) {
        double h[3];
        camera_rows<false>(after(g, r[2]) + kRigP + 12 * cam, r, h);
        uv[cam].x = trunc_i32(h[0] / h[2]);
        uv[cam].y = trunc_i32(h[1] / h[2]);
      }
    }
  }
  if (!valid) {
#pragma unroll
    for (int cam = 0; cam < 4; ++cam) uv[cam].x = uv[cam].y = (int)0x80000000;
  }
  // colour ramp, :28-29: cs = 255 (z / (max_range - 0.01)), of which only the two ROUNDED bytes are used.  The quotient through the
  // host's reciprocal is within ~1e-13 of the reference's; the bytes can only differ if cs sits that close to a half-integer, and
  // exactly those lanes (and NaNs: the comparison is false for them) redo the IEEE division.
  double cs = 255.0 * (r[2] * g[kRigRangeRcp]);
  const bool near_half = valid && !(__builtin_fabs((cs - __builtin_floor(cs)) - 0.5) > 1e-9);
  uint32_t a, b;
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(near_half) != 0, 0)) {  // wave-uniform, cold: the reference's own two roundings
    if (near_half) cs = 255.0 * (r[2] / g[kRigRangeDen]);
    a = sat_u8(255.0 - cs);
    b = sat_u8(cs);
  } else {
    // a valid lane's cs lies in [0.17, 255.2] and not within 1e-9 of a half-integer: rint(255 - cs) = 255 - rint(cs), clamps
    // included (cs > 255 rounds to 255 and 255 - cs to -0 -> 0) -- one rounding and one integer subtraction instead of two roundings
    b = sat_u8(cs);
    a = 255u - b;
  }
  bgrv = valid ? (a | (b << 8) | (a << 16) | (1u << 24)) : 0u;  // :32
  return valid;
}

}  // namespace kmc_dev

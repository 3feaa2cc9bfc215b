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
  float c1_x, c1_y, c1_z, pad0;     // c1 = phi x rho
  float c2_x, c2_y, c2_z, pad1;     // c2 = phi x (phi x rho)
};
static_assert(sizeof(FrameRec) == 64, "FrameRec must stay one 64-byte record");

enum Tier : int { kSeries3 = 0, kSeries5 = 1, kTrig = 2 };

// atan2(y, x) / (2 pi) in [-0.5, 0.5], i.e. the azimuth in turns.  timestamp_mocking.cpp:46 needs
// frac = (pi - atan2(y,x)) / 2pi = 0.5 - azimuth_turns.  Octant reduction + degree-7 polynomial in q^2
// (tools/gen_atan_coeffs.py; 1.0e-8 turns max error) -- ~30 VALU ops instead of ocml atan2f's ~55 plus a multiply.
// Signed zeros follow IEEE atan2: (+0,+0) -> 0, (-0, x<0 or x=-0) -> -0.5, (+0, x=-0) -> +0.5.
__device__ __forceinline__ float azimuth_turns(float x, float y) {
  const float ax = __builtin_fabsf(x);
  const float ay = __builtin_fabsf(y);
  const float mx = __builtin_fmaxf(ax, ay);
  const float mn = __builtin_fminf(ax, ay);
  float q = mn / mx;             // IEEE divide (handles denormals); 0/0 -> NaN fixed below
  q = (mx == 0.0f) ? 0.0f : q;   // atan2(0, 0) = 0 like libm
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

// Reference-accuracy variant through ocml (kept for A/B in tools/kmc_tune and the accuracy tests).
__device__ __forceinline__ float azimuth_turns_ocml(float x, float y) {
  return atan2f(y, x) * 0.15915494309189535f;
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
  } else {  // any angle: half-angle forms (no 1 - cos cancellation); series below theta^2 = 1/16
    const float As = __builtin_fmaf(__builtin_fmaf(u, 1.0f / 120.0f, -1.0f / 6.0f), u, 1.0f);
    const float Bs = __builtin_fmaf(__builtin_fmaf(u, 1.0f / 720.0f, -1.0f / 24.0f), u, 0.5f);
    const float Cs = __builtin_fmaf(__builtin_fmaf(u, 1.0f / 5040.0f, -1.0f / 120.0f), u, 1.0f / 6.0f);
    const float uu = __builtin_fmaxf(u, 1e-12f);
    const float th = __builtin_sqrtf(uu);
    float sh, ch;
    sincosf(0.5f * th, &sh, &ch);
    const float sn = 2.0f * sh * ch;
    const float inv_u = 1.0f / uu;
    const float At = sn / th;
    const float Bt = 2.0f * sh * sh * inv_u;
    const float Ct = (th - sn) * inv_u / th;
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
template <int TIER, bool OCML_ATAN = false>
__device__ __forceinline__ v4f deskew_point(const v4f p, const FrameRec& f) {
  const float a = OCML_ATAN ? azimuth_turns_ocml(p.x, p.y) : azimuth_turns(p.x, p.y);
  return deskew_point_s<TIER>(p, f.s0 - a, f);  // s = frac - x_req
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
  float c2_x, c2_y, c2_z, pad;
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

// ---- f64 path (Eigen-layout API): closed form in double, series below theta^2 = 0.04 -----------------
struct FrameRec64 {
  double phi[3], rho[3], c1[3], c2[3];
  double phi2;
  double x_req;
  double t_start, t_end, dur;
};

// f64 trajectory segment (global-memory table, read by the f64 Eigen-layout trajectory kernel)
struct TrajSeg64 {
  FrameRec64 f;      // twist of the segment; f.x_req holds the anchor a_k, f.t_start/f.dur the segment's time span
  double M[12];      // row-major 3x4 [R | t] applied after the exponential (identity for the anchor segment)
  int identity;
  int pad;
};

__device__ __forceinline__ void se3_coefficients_f64(double s, double phi2, double& alpha, double& beta, double& gamma) {
  const double s2 = s * s;
  const double u = s2 * phi2;
  double A, B, C;
  if (u < 0.04) {
    A = 1.0 + u * (-1.0 / 6 + u * (1.0 / 120 + u * (-1.0 / 5040 + u * (1.0 / 362880 + u * (-1.0 / 39916800 + u * (1.0 / 6227020800.0))))));
    B = 0.5 + u * (-1.0 / 24 + u * (1.0 / 720 + u * (-1.0 / 40320 + u * (1.0 / 3628800 + u * (-1.0 / 479001600 + u * (1.0 / 87178291200.0))))));
    C = 1.0 / 6 + u * (-1.0 / 120 + u * (1.0 / 5040 + u * (-1.0 / 362880 + u * (1.0 / 39916800 + u * (-1.0 / 6227020800.0 + u * (1.0 / 1307674368000.0))))));
  } else {
    const double th = sqrt(u);
    double sh, ch;
    sincos(0.5 * th, &sh, &ch);
    const double sn = 2.0 * sh * ch;
    A = sn / th;
    B = 2.0 * sh * sh / u;
    C = (th - sn) / (u * th);
  }
  alpha = A * s;
  beta = B * s2;
  gamma = C * (s2 * s);
}

__device__ __forceinline__ void deskew_point_f64(double x, double y, double z, double w, double s, const FrameRec64& f,
                                                 double& ox, double& oy, double& oz) {
  double al, be, ga;
  se3_coefficients_f64(s, f.phi2, al, be, ga);
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
};
using v2i = int __attribute__((ext_vector_type(2)));
using v4i = int __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int trunc_i32(double v) {  // cv::Point(double, double): cvttsd2si semantics
  return (v > -2147483649.0 && v < 2147483648.0) ? (int)v : (int)0x80000000;
}
__device__ __forceinline__ uint32_t sat_u8(double v) {  // cv::saturate_cast<uchar>(double): cvRound (half to even) + clamp
  const double r = __builtin_rint(v);
  return (r == r) ? (uint32_t)(r < 0.0 ? 0.0 : (r > 255.0 ? 255.0 : r)) : 0u;
}

// (trunc(a / d), trunc(b / d)) with the IEEE quotients' values, without two IEEE divisions on the common path: the
// quotients are first approximated through ONE refined reciprocal (relative error < 2^-45); their truncations equal the
// exact ones unless an integer lies within that error of the approximation, i.e. unless the approximation sits within
// 2^-36 * max(1, |q|) of an integer (or is not a finite value inside the int range, whose ends are integers too).
// Returns whether both truncations are certain; the caller redoes the point with real divisions otherwise (about once
// in 1e8 coordinates).
__device__ __forceinline__ bool trunc_quotients_fast(double a, double b, double d, int& ta, int& tb) {
  double r = __builtin_amdgcn_rcp(d);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  const double qa = a * r, qb = b * r;
  constexpr double kEps = 0x1p-36;
  const bool sure_a = __builtin_fabs(qa - __builtin_rint(qa)) > kEps * __builtin_fmax(1.0, __builtin_fabs(qa)) && __builtin_fabs(qa) < 2147483000.0;
  const bool sure_b = __builtin_fabs(qb - __builtin_rint(qb)) > kEps * __builtin_fmax(1.0, __builtin_fabs(qb)) && __builtin_fabs(qb) < 2147483000.0;
  ta = (int)qa;
  tb = (int)qb;
  return sure_a && sure_b;
}

// P_rect_c * r for one camera, :9.  STRUCTURED: the products with the literal 0s and 1 of a pinhole matrix are skipped.
template <bool STRUCTURED>
__device__ __forceinline__ void camera_rows(const double* P, const double r[3], double h[3]) {
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

// -> validity; uv[c] = pixel cv::circle would be centred on; bgrv = {255-cs, cs, 255-cs, 1} packed little-endian.
// STRUCTURED: every P_rect has the pinhole shape [fx 0 cx tx; 0 fy cy ty; 0 0 1 tz] (all KITTI calibrations do); skipping
// its zeros changes no result for finite coordinates (x + (+-0) = x, 1 * x = x).  A point whose rectified coordinates
// are not finite (0 * inf = NaN matters there) or whose quotients are not certain takes the plain sequence below.
// Most of a scan is behind the cameras or beyond max_range, and neighbouring points share that fate: a wave in which no
// lane passes the test of :21-24 skips the four cameras altogether (wave-uniform branch).
template <bool STRUCTURED>
__device__ __forceinline__ bool project_point(double x, double y, double z, const CameraRigRec& g, v2i uv[4], uint32_t& bgrv) {
#pragma clang fp contract(off)
  double c[3], r[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) c[k] = ((g.T[4 * k] * x + g.T[4 * k + 1] * y) + g.T[4 * k + 2] * z) + g.T[4 * k + 3];  // :75
#pragma unroll
  for (int k = 0; k < 3; ++k) r[k] = ((g.R[3 * k] * c[0] + g.R[3 * k + 1] * c[1]) + g.R[3 * k + 2] * c[2]) + 0.0;  // :81
  const bool valid = !((r[2] < 0.01) || (r[2] > g.max_range) || (r[1] > 1.25));                                  // :21-24
  if (__builtin_amdgcn_ballot_w64(valid) == 0) {
#pragma unroll
    for (int cam = 0; cam < 4; ++cam) uv[cam].x = uv[cam].y = (int)0x80000000;
    bgrv = 0u;
    return false;
  }
  bool sure = __builtin_fabs(r[0]) < __builtin_inf() && __builtin_fabs(r[1]) < __builtin_inf() && __builtin_fabs(r[2]) < __builtin_inf();
#pragma unroll
  for (int cam = 0; cam < 4; ++cam) {
    double h[3];
    camera_rows<STRUCTURED>(g.P[cam], r, h);
    int tu, tv;
    sure &= trunc_quotients_fast(h[0], h[1], h[2], tu, tv);
    uv[cam].x = tu;
    uv[cam].y = tv;
  }
  if (__builtin_expect(valid && !sure, 0)) {  // the reference's own sequence: general rows, two IEEE divisions (:9, :12, :31)
#pragma unroll 1
    for (int cam = 0; cam < 4; ++cam) {
      double h[3];
      camera_rows<false>(g.P[cam], r, h);
      uv[cam].x = trunc_i32(h[0] / h[2]);
      uv[cam].y = trunc_i32(h[1] / h[2]);
    }
  }
  if (!valid) {
#pragma unroll
    for (int cam = 0; cam < 4; ++cam) uv[cam].x = uv[cam].y = (int)0x80000000;
  }
  const double cs = 255.0 * (r[2] / g.range_den);  // :28-29
  const uint32_t a = sat_u8(255.0 - cs), b = sat_u8(cs);
  bgrv = valid ? (a | (b << 8) | (a << 16) | (1u << 24)) : 0u;  // :32
  return valid;
}

}  // namespace kmc_dev

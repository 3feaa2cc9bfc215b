// kmc_capi_project.hip -- row N4: the arithmetic of camera_model.cpp (LiDAR -> four rectified cameras) on the GPU.
#include "kmc_internal.hip.h"

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
// ... and the same fx, cx, fy, cy (bit for bit) in all four: the kernel computes fx x + cx z and fy y + cy z once
int rig_kind(const kmc_camera_rig* g) {
  if (!rig_is_pinhole(g)) return kRigGeneral;
  for (int c = 1; c < 4; ++c)
    for (int k : {0, 2, 5, 6})
      if (std::memcmp(&g->P_rect[c][k], &g->P_rect[0][k], sizeof(double)) != 0) return kRigPinhole;
  return kRigSharedIntrinsics;
}
CameraRigRec rig_rec(const kmc_camera_rig* g) {
  CameraRigRec r;
  std::memcpy(r.T, g->tf_c00_lo, sizeof(r.T));
  std::memcpy(r.R, g->R_rect_00, sizeof(r.R));
  std::memcpy(r.P, g->P_rect, sizeof(r.P));
  r.max_range = g->max_range;
  r.range_den = g->max_range - 0.01;  // camera_model.cpp:28
  r.range_rcp = 1.0 / r.range_den;
  r.same_den = 0;
  for (int c = 1; c < 4; ++c)
    if (std::memcmp(&g->P_rect[c][11], &g->P_rect[c - 1][11], sizeof(double)) == 0) r.same_den |= 1ull << c;
  return r;
}
}  // namespace

extern "C" {

int kmc_hip_project_f32(kmc_ctx* c, const float* xyzi_in, uint64_t n, const kmc_camera_rig* rig, const kmc_frame_params* deskew,
                        float* xyzi_out, int32_t* uv, uint8_t* bgrv, int mem_kind, kmc_stats* st) {
  if (!c || !rig || (n && (!xyzi_in || !uv || !bgrv))) return KMC_ERR_INVALID_ARG;
  if (mem_kind != KMC_MEM_HOST && mem_kind != KMC_MEM_DEVICE) return KMC_ERR_INVALID_ARG;
  if (xyzi_out && !deskew) return KMC_ERR_INVALID_ARG;
  if (((uintptr_t)xyzi_in | (uintptr_t)xyzi_out) & (mem_kind == KMC_MEM_DEVICE ? 15u : 3u)) return KMC_ERR_INVALID_ARG;
  if (((uintptr_t)uv & (mem_kind == KMC_MEM_DEVICE ? 15u : 3u)) || ((uintptr_t)bgrv & (mem_kind == KMC_MEM_DEVICE ? 3u : 0u))) return KMC_ERR_INVALID_ARG;
  if (!rig_ok(rig)) return KMC_ERR_INVALID_ARG;
  if (deskew) {
    if (!params_ok(deskew)) return KMC_ERR_INVALID_ARG;
    if (!(deskew->x_req >= 0.0 && deskew->x_req <= 1.0)) return KMC_ERR_TIME_OUT_OF_RANGE;
  }
  if (st) std::memset(st, 0, sizeof(*st));
  KMC_ENTER(c);
  const int tier = deskew ? pick_tier(c, deskew, 1) : -1;
  FrameRec f;
  std::memset(&f, 0, sizeof(f));
  FrameRecD fd;
  std::memset(&fd, 0, sizeof(fd));
  if (deskew) { fill_rec(*deskew, &f); f.pre2 = guard_pre2(*deskew); fill_recd(*deskew, &fd); }
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
  const int kind = rig_kind(rig);
  auto with_rig = [&](auto&& fn) {
    if (kind == kRigSharedIntrinsics) fn(std::integral_constant<int, kRigSharedIntrinsics>{});
    else if (kind == kRigPinhole) fn(std::integral_constant<int, kRigPinhole>{});
    else fn(std::integral_constant<int, kRigGeneral>{});
  };
  auto launch_tier = [&](auto T) {  // T = -1: projection only, no deskew
    with_rig([&](auto RIG) {
      launch_tiles((n + 63) / 64, [&](uint64_t t0, int grid) {
        launch_on(project_f32<decltype(T)::value, decltype(RIG)::value>, grid, 64, c->stream, false, d_in, n, g, f, d_cloud, d_uv, d_col, t0, fd);
      });
    });
  };
  if (tier >= kSeries3 && tier <= kTrig) with_tier(tier, launch_tier);
  else launch_tier(std::integral_constant<int, -1>{});
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
  if (((uintptr_t)uv & (mem_kind == KMC_MEM_DEVICE ? 15u : 3u)) || ((uintptr_t)bgrv & (mem_kind == KMC_MEM_DEVICE ? 3u : 0u))) return KMC_ERR_INVALID_ARG;
  if (!rig_ok(rig)) return KMC_ERR_INVALID_ARG;
  if (st) std::memset(st, 0, sizeof(*st));
  if (st) { st->n_points = n; st->variant = 4; }
  if (n == 0) return KMC_OK;
  KMC_ENTER(c);
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
  const int kind = rig_kind(rig);
  {
    auto launch_rig = [&](auto RIG) {
      launch_tiles((n + 63) / 64, [&](uint64_t t0, int grid) { launch_on(project_f64cols<decltype(RIG)::value>, grid, 64, c->stream, false, dx, dy, dz, n, g, d_uv, d_col, t0); });
    };
    if (kind == kRigSharedIntrinsics) launch_rig(std::integral_constant<int, kRigSharedIntrinsics>{});
    else if (kind == kRigPinhole) launch_rig(std::integral_constant<int, kRigPinhole>{});
    else launch_rig(std::integral_constant<int, kRigGeneral>{});
  }
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
}  // extern "C"

// kmc_capi_f64.hip -- the f64 Eigen-layout entry points: the device half of the literal MotionCompensateFrame(Frame const&, Time)
// (motion_compensation.cpp:16-28 on data_types.hpp:14's layout: four columns + per-point stamps), its begin / end pair, and
// GetPseudoTimeStamps (timestamp_mocking.cpp:56-63).  Routes: device-resident columns; page-locked host columns IN PLACE over the link
// (persistent waves, sc1 stores, completion word); pageable host columns staged -- a duplex chunk pipeline from 1 M points on.
// (Split from kmc_capi_deskew.hip in round 5: that file keeps the f32 KITTI-layout single frame / list / batch.)
#include "kmc_internal.hip.h"

#include <hip/hip_ext.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <system_error>
#include <thread>

namespace {
// ---- the f64 Eigen-layout route on HOST buffers of >= kF64PipelineMinPoints points: a duplex chunk pipeline ---------------
// PCIe is full duplex (measured on the MI355X box: 50 GB/s each way alone, 43 GB/s each way together,
// tools/f64_route_probe.hip), but "upload everything, run, download everything" uses one direction at a time: 12.96 ms for
// a 10 M-point frame (400 MB up, 320 MB down).  Here the frame is cut into chunks of kF64ChunkPoints points; the calling
// thread uploads chunk k+1 and launches its kernel while a helper thread downloads chunk k (copies from / to pageable
// memory block their caller, hence the second thread).  The device scratch holds the whole frame, so chunks never wait for a
// buffer.
constexpr uint64_t kF64PipelineMinPoints = 1ull << 20;
constexpr uint64_t kMappedMinPoints = 2048;  // below this a kernel over the link is all latency; the staged route's small copies are as good
constexpr uint64_t kF64ChunkPoints = 1ull << 20;

int deskew_f64cols_host_pipelined(kmc_ctx* c, const double* x, const double* y, const double* z, const double* w, const double* stamps,
                                  uint64_t n, const FrameRec64& f, double* ox, double* oy, double* oz, double* ow, kmc_stats* st) {
  const size_t col = n * sizeof(double);
  int rc = ensure_tmp(c, 9 * col);
  if (rc != KMC_OK) return rc;
  rc = ensure_pipe_streams(c);
  if (rc != KMC_OK) return rc;
  const uint64_t n_chunks = (n + kF64ChunkPoints - 1) / kF64ChunkPoints;
  rc = ensure_events(c, 2 * n_chunks);
  if (rc != KMC_OK) return rc;
  KMC_HIP_TRY(c, hipStreamSynchronize(c->stream));
  double* base = (double*)c->d_tmp;
  double* cols[9];
  for (int i = 0; i < 9; ++i) cols[i] = base + (size_t)i * n;
  const bool down_w = ow != nullptr;
  hipStream_t s_up = c->pipe[0], s_run = c->pipe[1], s_down = c->pipe[2];
  KMC_HIP_TRY(c, hipMemsetAsync(c->d_counter, 0, sizeof(unsigned long long), s_run));
  c->counter_dirty = true;

  // chunks whose kernel-done event has been recorded; the downloader sleeps on the condition variable between them
  std::mutex mu;
  std::condition_variable cv;
  uint64_t launched = 0;
  bool abort_flag = false;
  hipError_t down_error = hipSuccess;
  if (c->timing) {  // the pipeline lives on its own three streams: bracket it with events on the kernel stream
    KMC_HIP_TRY(c, hipEventRecord(c->ev_c0, s_run));
    KMC_HIP_TRY(c, hipEventRecord(c->ev_k0, s_run));
  }
  std::thread downloader;
  try {
    downloader = std::thread([&] {
      hipError_t e = hipSetDevice(c->device);
      for (uint64_t k = 0; k < n_chunks && e == hipSuccess; ++k) {
        {
          std::unique_lock<std::mutex> lock(mu);
          cv.wait(lock, [&] { return launched > k || abort_flag; });
          if (abort_flag) return;
        }
        const uint64_t off = k * kF64ChunkPoints, m = std::min<uint64_t>(kF64ChunkPoints, n - off);
        e = hipStreamWaitEvent(s_down, c->ev_pool[2 * k + 1], 0);
        if (e == hipSuccess) e = hipMemcpyAsync(ox + off, cols[5] + off, m * sizeof(double), hipMemcpyDeviceToHost, s_down);
        if (e == hipSuccess) e = hipMemcpyAsync(oy + off, cols[6] + off, m * sizeof(double), hipMemcpyDeviceToHost, s_down);
        if (e == hipSuccess) e = hipMemcpyAsync(oz + off, cols[7] + off, m * sizeof(double), hipMemcpyDeviceToHost, s_down);
        if (e == hipSuccess && down_w) e = hipMemcpyAsync(ow + off, cols[8] + off, m * sizeof(double), hipMemcpyDeviceToHost, s_down);
      }
      if (e == hipSuccess) e = hipStreamSynchronize(s_down);
      down_error = e;
    });
  } catch (const std::system_error&) {
    // no exception may cross the C ABI: the caller falls back to the un-pipelined route
    return KMC_ERR_ALLOC;
  }
  hipError_t up_error = hipSuccess;
  for (uint64_t k = 0; k < n_chunks && up_error == hipSuccess; ++k) {
    const uint64_t off = k * kF64ChunkPoints, m = std::min<uint64_t>(kF64ChunkPoints, n - off);
    hipError_t e = hipMemcpyAsync(cols[0] + off, x + off, m * sizeof(double), hipMemcpyHostToDevice, s_up);
    if (e == hipSuccess) e = hipMemcpyAsync(cols[1] + off, y + off, m * sizeof(double), hipMemcpyHostToDevice, s_up);
    if (e == hipSuccess) e = hipMemcpyAsync(cols[2] + off, z + off, m * sizeof(double), hipMemcpyHostToDevice, s_up);
    if (e == hipSuccess && w) e = hipMemcpyAsync(cols[3] + off, w + off, m * sizeof(double), hipMemcpyHostToDevice, s_up);
    if (e == hipSuccess) e = hipMemcpyAsync(cols[4] + off, stamps + off, m * sizeof(double), hipMemcpyHostToDevice, s_up);
    if (e == hipSuccess) e = hipEventRecord(c->ev_pool[2 * k], s_up);
    if (e == hipSuccess) e = hipStreamWaitEvent(s_run, c->ev_pool[2 * k], 0);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(deskew_f64cols<false>, dim3((uint32_t)(((m + 127) / 128 + kF64TilesPerWave - 1) / kF64TilesPerWave)), dim3(64), 0, s_run, cols[0] + off, cols[1] + off, cols[2] + off, w ? cols[3] + off : nullptr,
                         cols[4] + off, m, f, cols[5] + off, cols[6] + off, cols[7] + off, down_w ? cols[8] + off : nullptr, c->d_counter, (uint32_t*)nullptr, (uint64_t)0, DoneWord{});
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipEventRecord(c->ev_pool[2 * k + 1], s_run);
    if (e == hipSuccess) {
      {
        std::lock_guard<std::mutex> lock(mu);
        launched = k + 1;
      }
      cv.notify_one();
    }
    up_error = e;
  }
  if (up_error != hipSuccess) {
    {
      std::lock_guard<std::mutex> lock(mu);
      abort_flag = true;
    }
    cv.notify_one();
  } else if (c->timing) {
    up_error = hipEventRecord(c->ev_k1, s_run);  // behind the last chunk's kernel
  }
  downloader.join();
  if (up_error != hipSuccess) return fail_hip(c, up_error, "f64 host pipeline (upload / launch)");
  if (down_error != hipSuccess) return fail_hip(c, down_error, "f64 host pipeline (download)");
  unsigned long long bad = 0;
  KMC_HIP_TRY(c, hipMemcpyAsync(&bad, c->d_counter, sizeof(bad), hipMemcpyDeviceToHost, s_run));
  KMC_HIP_TRY(c, hipStreamSynchronize(s_run));
  KMC_HIP_TRY(c, hipStreamSynchronize(s_up));
  if (st) { st->n_launches = (uint32_t)n_chunks; st->n_out_of_range = bad; }
  if (c->timing) {  // kernel_ms: first upload wait to last kernel on the kernel stream; total_ms: until the last byte is back in host memory
    KMC_HIP_TRY(c, hipEventRecord(c->ev_c1, s_run));  // every stream has been synchronized above: this is "now"
    KMC_HIP_TRY(c, hipEventSynchronize(c->ev_c1));
    if (st) {
      KMC_HIP_TRY(c, hipEventElapsedTime(&st->kernel_ms, c->ev_k0, c->ev_k1));
      KMC_HIP_TRY(c, hipEventElapsedTime(&st->total_ms, c->ev_c0, c->ev_c1));
    }
  }
  return bad ? KMC_ERR_TIME_OUT_OF_RANGE : KMC_OK;
}
}  // namespace

extern "C" {

// ---- f64 Eigen-layout path ------------------------------------------------------------------------
// `defer`: kmc_hip_deskew_f64cols_begin -- when the buffers are device-addressable the work is only ISSUED here and
// f64cols_finish() (kmc_hip_deskew_f64cols_end) waits for it; staged host buffers complete here and leave their verdict for _end.
static int f64cols_finish(kmc_ctx* c, kmc_stats* st);
static int f64cols_issue(kmc_ctx* c, const double* x, const double* y, const double* z, const double* w, const double* stamps,
                         uint64_t n, double stamp_start, double stamp_end, const kmc_frame_params* params, double* ox,
                         double* oy, double* oz, double* ow, int mem_kind, kmc_stats* st, bool defer) {
  if (!c || !params) return KMC_ERR_INVALID_ARG;
  // a staged verdict waits for its _end, or a plain call arrives while _begin calls are queued
  if (c->f64_pending == 2 || (c->f64_pending == 1 && !defer)) return KMC_ERR_INVALID_ARG;
  const bool queued = c->f64_pending == 1;  // behind earlier _begin calls: the counter, the flag word and the stats accumulate until _end
  if (n && (!x || !y || !z || !stamps || !ox || !oy || !oz)) return KMC_ERR_INVALID_ARG;
  if (mem_kind != KMC_MEM_HOST && mem_kind != KMC_MEM_DEVICE && mem_kind != KMC_MEM_HOST_MAPPED) return KMC_ERR_INVALID_ARG;
  if (!(stamp_start < stamp_end)) return KMC_ERR_DEGENERATE;
  if (!params_ok(params)) return KMC_ERR_INVALID_ARG;
  if (!(params->x_req >= 0.0 && params->x_req <= 1.0)) return KMC_ERR_TIME_OUT_OF_RANGE;
  if (st) std::memset(st, 0, sizeof(*st));
  if (st) { st->n_points = n; st->variant = 5; }
  if (n == 0) return KMC_OK;
  KMC_ENTER(c);

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
  f.inv_dur = 1.0 / (stamp_end - stamp_start);
  f.halvings = halvings_for(f.phi2);  // |s| <= 1 inside the scan
  f.terms = series_terms_for(f.phi2);

  const double *dx = x, *dy = y, *dz = z, *dw = w, *ds = stamps;
  double *dox = ox, *doy = oy, *doz = oz, *dow = ow;
  const size_t col = n * sizeof(double);
  // Host containers made of the page-locked pool (the C++ drop-in's Pointcloud / VectorXd are): the kernel works on them in place.
  // ONE launch -- the 40 B per point coming up and the 24-32 B going down share the full-duplex link -- instead of three staged copies
  // with ~20 us of fixed cost each (123 k-point frame: 224-239 us staged, see profiles/NOTES_r03.md for the in-place figure).
  if (mem_kind == KMC_MEM_HOST && n >= kMappedMinPoints && host_in_place_ok(x, col) && host_in_place_ok(y, col) && host_in_place_ok(z, col) &&
      (!w || host_in_place_ok(w, col)) && host_in_place_ok(stamps, col) && host_in_place_ok(ox, col) && host_in_place_ok(oy, col) &&
      host_in_place_ok(oz, col) && (!ow || host_in_place_ok(ow, col)))
    mem_kind = KMC_MEM_HOST_MAPPED;
  // (Recognising a homogeneous column of ones on the host and skipping its two transfers was measured and dropped: scanning
  // and refilling it costs what moving it over PCIe costs -- 12 + 9 us against 37 us saved at 123 k points, and it serialises
  // with the pageable copies; tools/f64_route_probe.hip.)
  if (queued && mem_kind == KMC_MEM_HOST) return KMC_ERR_INVALID_ARG;  // staged buffers complete inside the call: not behind queued work
  if (mem_kind == KMC_MEM_HOST && n >= kF64PipelineMinPoints) {
    const int rc_pipe = deskew_f64cols_host_pipelined(c, x, y, z, w, stamps, n, f, ox, oy, oz, ow, st);
    if (rc_pipe != KMC_ERR_ALLOC) return rc_pipe;  // KMC_ERR_ALLOC: the helper thread could not be started -> the plain route below
  }
  if (mem_kind == KMC_MEM_HOST) {
    int rc = ensure_tmp(c, 9 * col);
    if (rc != KMC_OK) return rc;
    double* base = (double*)c->d_tmp;
    double* cols[9];
    for (int i = 0; i < 9; ++i) cols[i] = base + (size_t)i * n;
    const bool up_w = w != nullptr;
    // an Eigen::MatrixX4d is ONE column-major block: x, y, z, w follow each other -> one copy instead of four (each
    // copy has a fixed cost of ~20 us, which is what a 123 k-point frame is made of)
    if (y == x + n && z == y + n && (!up_w || w == z + n)) {
      KMC_HIP_TRY(c, hipMemcpyAsync(cols[0], x, (up_w ? 4 : 3) * col, hipMemcpyHostToDevice, c->stream));
    } else {
      KMC_HIP_TRY(c, hipMemcpyAsync(cols[0], x, col, hipMemcpyHostToDevice, c->stream));
      KMC_HIP_TRY(c, hipMemcpyAsync(cols[1], y, col, hipMemcpyHostToDevice, c->stream));
      KMC_HIP_TRY(c, hipMemcpyAsync(cols[2], z, col, hipMemcpyHostToDevice, c->stream));
      if (up_w) KMC_HIP_TRY(c, hipMemcpyAsync(cols[3], w, col, hipMemcpyHostToDevice, c->stream));
    }
    KMC_HIP_TRY(c, hipMemcpyAsync(cols[4], stamps, col, hipMemcpyHostToDevice, c->stream));
    dx = cols[0]; dy = cols[1]; dz = cols[2]; dw = up_w ? cols[3] : nullptr; ds = cols[4];
    dox = cols[5]; doy = cols[6]; doz = cols[7]; dow = ow ? cols[8] : nullptr;
  }
  CallTimer tm(c);
  if (c->trace && !queued) { c->last_trace = kmc_call_trace{}; c->last_trace.issue_begin_us = trace_now_us(); }
  if (!queued) {
    if (tm.begin_call()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
    if (c->counter_dirty) KMC_HIP_TRY(c, hipMemsetAsync(c->d_counter, 0, sizeof(unsigned long long), c->stream));
    c->counter_dirty = true;
    *c->h_flag = 0;  // the previous call has been waited for: nothing on the device still writes it
    if (tm.begin_kernel()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
  }
  if (mem_kind == KMC_MEM_HOST_MAPPED) {
    // over the link: ~a hundred persistent waves, each with its next tile's loads in flight while it stores the current one; the last
    // wave raises the completion word f64cols_finish() spins on
    // (a wave of this kernel keeps 4-5 KiB of loads in flight: half the f32 kernel's wave count carries a KITTI frame best -- 91 us per
    // call with 64 waves, 94-118 with 96-192; frames of half a million points and more want the full count: tools/link_probe, 1 M points)
    const uint64_t waves = n < (1ull << 19) ? std::max(1, c->mapped_waves / 2) : c->mapped_waves;
    const int grid = (int)std::max<uint64_t>(1, std::min<uint64_t>((n + 127) / 128, waves));
    const DoneWord done = done_word_arm(c);
    hipLaunchKernelGGL(deskew_f64cols<true>, dim3(grid), dim3(64), 0, c->stream, dx, dy, dz, dw, ds, n, f, dox, doy, doz, dow, c->d_counter, c->h_flag, (uint64_t)0, done);
    if (c->trace) { c->last_trace.waves = (uint32_t)grid; c->last_trace.route = 2; }
  } else {  // one wave per workgroup, two points per lane
#ifdef KMC_F64_PERSISTENT  // A/B only (tools/build_f64_variants.sh): the walking kernel of the in-place route on RESIDENT columns -- that many persistent waves, the next tile's loads in flight while the current one is computed
    hipLaunchKernelGGL(deskew_f64cols<true>, dim3((unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n + 127) / 128, (uint64_t)(KMC_F64_PERSISTENT)))), dim3(64), 0, c->stream, dx, dy, dz, dw, ds, n, f, dox, doy,
                       doz, dow, c->d_counter, c->h_flag, (uint64_t)0, DoneWord{});
#else
    launch_tiles(((n + 127) / 128 + kF64TilesPerWave - 1) / kF64TilesPerWave, [&](uint64_t t0, int grid) {  // (t0 and grid count WORKGROUPS)
      launch_on(deskew_f64cols<false>, grid, 64, c->stream, false, dx, dy, dz, dw, ds, n, f, dox, doy, doz, dow, c->d_counter, c->h_flag, t0, DoneWord{});
    });
#endif
    c->done_armed = false;  // resident or staged columns: this queue is waited for on the stream
  }
  KMC_HIP_TRY(c, hipGetLastError());
  if (c->trace) c->last_trace.issue_end_us = trace_now_us();
  if (tm.end_kernel()) return fail_hip(c, hipGetLastError(), "hipEventRecord");
  if (mem_kind == KMC_MEM_HOST) {
    const bool down_w = ow != nullptr;
    if (oy == ox + n && oz == oy + n && (!down_w || ow == oz + n)) {  // one column-major block again
      KMC_HIP_TRY(c, hipMemcpyAsync(ox, dox, (down_w ? 4 : 3) * col, hipMemcpyDeviceToHost, c->stream));
    } else {
      KMC_HIP_TRY(c, hipMemcpyAsync(ox, dox, col, hipMemcpyDeviceToHost, c->stream));
      KMC_HIP_TRY(c, hipMemcpyAsync(oy, doy, col, hipMemcpyDeviceToHost, c->stream));
      KMC_HIP_TRY(c, hipMemcpyAsync(oz, doz, col, hipMemcpyDeviceToHost, c->stream));
      if (down_w) KMC_HIP_TRY(c, hipMemcpyAsync(ow, dow, col, hipMemcpyDeviceToHost, c->stream));
    }
  }
  if (!queued) c->f64_stats = kmc_stats{};
  c->f64_stats.n_points += n;
  c->f64_stats.variant = 5;
  c->f64_stats.n_launches += 1;
  c->f64_pending = 1;  // issued, not waited for
  if (defer && mem_kind != KMC_MEM_HOST) return KMC_OK;
  return f64cols_finish(c, st);
}

static int f64cols_finish(kmc_ctx* c, kmc_stats* st) {
  if (c->f64_pending == 2) {  // a staged route that completed inside _begin
    c->f64_pending = 0;
    if (st) *st = c->f64_stats;
    return c->f64_result;
  }
  if (c->f64_pending != 1) return KMC_ERR_INVALID_ARG;
  c->f64_pending = 0;
  // The out-of-range verdict is part of the call's result: wait.  In place over the link: for the completion word of the last kernel
  // issued (kernels of one stream run in order, so the word of the last one covers a queue of _begin calls); resident and staged
  // columns: for the stream.
  if (c->trace) c->last_trace.wait_begin_us = trace_now_us();
  if (c->done_armed) {
    const int rc_wait = wait_done_word(c);
    if (rc_wait != KMC_OK) return rc_wait;
    if (c->trace) {
      c->last_trace.wait_end_us = trace_now_us();
      c->last_trace.dev_first_wave_us = (double)c->h_stamps[0] * 0.01;
      c->last_trace.dev_last_store_us = (double)c->h_stamps[1] * 0.01;
    }
  } else {
    KMC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->trace) { c->last_trace.wait_end_us = trace_now_us(); c->last_trace.route = 0; }
  }
  unsigned long long bad = 0;
  if (*(volatile uint32_t*)c->h_flag != 0) {  // cold: some stamp was out of range -> fetch the exact count
    KMC_HIP_TRY(c, hipMemcpy(&bad, c->d_counter, sizeof(bad), hipMemcpyDeviceToHost));
  } else {
    c->counter_dirty = false;  // nobody touched the counter
  }
  c->f64_stats.n_out_of_range = bad;
  CallTimer tm(c);
  const int rc = tm.end_call(&c->f64_stats);
  if (st) *st = c->f64_stats;
  if (rc != KMC_OK) return rc;
  return bad ? KMC_ERR_TIME_OUT_OF_RANGE : KMC_OK;
}

int kmc_hip_deskew_f64cols(kmc_ctx* c, const double* x, const double* y, const double* z, const double* w, const double* stamps,
                           uint64_t n, double stamp_start, double stamp_end, const kmc_frame_params* params, double* ox,
                           double* oy, double* oz, double* ow, int mem_kind, kmc_stats* st) {
  return f64cols_issue(c, x, y, z, w, stamps, n, stamp_start, stamp_end, params, ox, oy, oz, ow, mem_kind, st, false);
}

int kmc_hip_deskew_f64cols_begin(kmc_ctx* c, const double* x, const double* y, const double* z, const double* w, const double* stamps,
                                 uint64_t n, double stamp_start, double stamp_end, const kmc_frame_params* params, double* ox,
                                 double* oy, double* oz, double* ow, int mem_kind) {
  if (!c) return KMC_ERR_INVALID_ARG;
  const int was_pending = c->f64_pending;
  kmc_stats st = {};
  const int rc = f64cols_issue(c, x, y, z, w, stamps, n, stamp_start, stamp_end, params, ox, oy, oz, ow, mem_kind, &st, true);
  if (c->f64_pending == 1 && rc == KMC_OK) return rc;  // issued (behind the earlier _begin calls, if any); n == 0 behind queued work
  if (was_pending != 0) return rc;  // rejected, or failed, with earlier work still waiting for its _end: that state is not touched
  if (rc != KMC_OK && rc != KMC_ERR_TIME_OUT_OF_RANGE) {  // an argument / runtime error: nothing to _end
    c->f64_pending = 0;
    return rc;
  }
  // completed inside this call (n == 0, staged host buffers): keep the verdict for _end
  c->f64_stats = st;
  c->f64_result = rc;
  c->f64_pending = 2;
  return KMC_OK;
}

int kmc_hip_deskew_f64cols_end(kmc_ctx* c, kmc_stats* st) {
  if (!c) return KMC_ERR_INVALID_ARG;
  KMC_HIP_TRY(c, hipSetDevice(c->device));
  return f64cols_finish(c, st);
}

int kmc_hip_pseudo_timestamps_f64(kmc_ctx* c, const double* x, const double* y, uint64_t n, double scan_start, double scan_end,
                                  double* stamps_out, int mem_kind) {
  if (!c || (n && (!x || !y || !stamps_out))) return KMC_ERR_INVALID_ARG;
  if (mem_kind != KMC_MEM_HOST && mem_kind != KMC_MEM_DEVICE && mem_kind != KMC_MEM_HOST_MAPPED) return KMC_ERR_INVALID_ARG;
  if (n == 0) return KMC_OK;
  KMC_ENTER(c);
  const size_t col = n * sizeof(double);
  const double *dx = x, *dy = y;
  double* dout = stamps_out;
  if (mem_kind == KMC_MEM_HOST && n >= kMappedMinPoints && host_in_place_ok(x, col) && host_in_place_ok(y, col) && host_in_place_ok(stamps_out, col))
    mem_kind = KMC_MEM_HOST_MAPPED;  // page-locked containers: in place, see kmc_hip_deskew_f64cols
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
  launch_tiles((n + 127) / 128, [&](uint64_t t0, int grid) {  // one wave per workgroup, two points per lane
    launch_on(pseudo_timestamps_f64<0>, grid, 64, c->stream, false, dx, dy, n, scan_start, scan_end, dout, t0);
  });
  KMC_HIP_TRY(c, hipGetLastError());
  if (mem_kind == KMC_MEM_HOST) KMC_HIP_TRY(c, hipMemcpyAsync(stamps_out, dout, col, hipMemcpyDeviceToHost, c->stream));
  if (mem_kind != KMC_MEM_DEVICE) KMC_HIP_TRY(c, hipStreamSynchronize(c->stream));  // the results are in host memory when the call returns
  return KMC_OK;
}
}  // extern "C"

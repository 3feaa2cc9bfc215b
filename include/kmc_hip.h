/*
 * kmc_hip.h -- C-ABI of libkmc_hip.so: the MI355X (gfx950) per-point LiDAR deskew engine.
 *
 * This is the drop-in boundary for ONE hot path of fracgawd/kitti_motion_compensation:
 *
 *     kmc::MotionCompensateFrame(Frame const&, Time)          include/.../motion_compensation.hpp:13
 *       -> per point kmc::MotionCompensatePoint(...)          src/.../motion_compensation.cpp:9-14, :22-25
 *            -> TrajectoryInterpolator::RelativePoseBetweenTimes   trajectory_interpolation.cpp:43-45
 *                 -> lie::Log / lie::Exp                       lie_algebra.cpp:83-103
 *     with the per-point stamps of kmc::GetPseudoTimeStamps   timestamp_mocking.cpp:46-63
 *
 * The reference has no FFI of its own (it is a single C++ shared library, CMakeLists.txt:27-29); the
 * entry points below are what a binding of that path needs.  The C++ shim that restores the reference's
 * exact signatures on top of this ABI is include/kitti_motion_compensation/ (libkitti_motion_compensation_lib.so).
 *
 * Rules of the ABI
 *   - plain C: pointers, sizes, PODs; no C++/torch/Eigen types; no exceptions cross it.
 *   - every function returns an int status: KMC_OK (0) or a negative KMC_ERR_*; kmc_status_string()
 *     names it and kmc_hip_last_error() gives the HIP error text for KMC_ERR_HIP.
 *   - the caller owns every buffer it passes; `mem_kind` says where the point buffers live.
 *   - a kmc_ctx is bound to one device and one HIP stream; calls on one ctx are not re-entrant, distinct
 *     contexts are independent (one ctx per thread / per rank).
 *   - there is NO CPU fallback: without a usable HIP device kmc_hip_create() fails with
 *     KMC_ERR_NO_DEVICE and nothing else can be called.
 *
 * Environment (read at kmc_hip_create / first use; each has a default, and an API equivalent where one makes sense)
 *   knob                        default  meaning                                                         non-default covered by
 *   KMC_ANY_ORDER               on*      0: every dispatch carries the AQL barrier bit (* on only where   tests/test_dispatch_modes.py; the GPU suite run
 *                                        the run-time probe verified barrier-free dispatch)               under it (tests/test_suite_modes.py)
 *   KMC_DIRECT_DISPATCH         unset    1: contexts start opted in to the direct queue; 0: never,        tests/test_direct_queue.py, test_suite_modes.py
 *                                        even when kmc_hip_set_direct_dispatch asks
 *   KMC_HOST_POOL               1        0: no pool of page-locked host memory (staged copies)            tests/test_host_pool.py, bench.py dropin leg
 *   KMC_HOST_POOL_MAX_MB        2048     MiB of free blocks the pool keeps page-locked                    tests/test_host_pool.py
 *   KMC_HOST_DETECT_PINNED      1        0: only pool memory and KMC_MEM_HOST_MAPPED run in place         tests/test_host_pool.py
 *   KMC_TEST_HOST_POOL_FAIL_AT  unset    k: the process's k-th pool allocation fails (fault injection)    tests/test_run_driver.py
 * The C++ drop-in adds KMC_DEVICE, KMC_DEVICES (include/kitti_motion_compensation/motion_compensation.hpp), KMC_RUN_BATCH_FRAMES,
 * KMC_RUN_TIMING, KMC_RUN_KNOTS, KMC_FIX_LAST_FRAME_COPY (.../handlers.hpp; tests/test_run_driver.py) and its CLI KMC_CLI_FULL_TEARDOWN
 * (tools/motion_compensate_runs.cpp): 13 in all, and tests/test_abi_exports.py fails on a getenv that is not listed here.  Retired in ABI 7: KMC_LIST_ROUTE, KMC_DIRECT_LANES, KMC_MAPPED_WAVES,
 * KMC_DIRECT_DEBUG (measurement switches whose questions are answered: profiles/NOTES.md).
 *
 * Data conventions
 *   - pose      : double[12], row-major 3x4 [R | t]  (Eigen::Affine3d::matrix().topRows<3>(), row-major)
 *   - twist     : double[6] = [rho(3); phi(3)]  -- same order as the reference, lie_algebra.cpp:84-85
 *   - KITTI cloud (f32 path): float[4*n] AoS {x, y, z, intensity}, the on-disk .bin layout (data_io.cpp:101-138)
 *   - Eigen cloud (f64 path): four double[n] columns x, y, z, w (= Eigen::MatrixX4d, column-major) + double stamps[n]
 */
#ifndef KMC_HIP_H
#define KMC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KMC_ABI_VERSION 7

/* ---- status codes ---- */
#define KMC_OK 0
#define KMC_ERR_INVALID_ARG (-1)
#define KMC_ERR_HIP (-2)               /* a HIP runtime call failed; see kmc_hip_last_error() */
#define KMC_ERR_NO_DEVICE (-3)         /* no usable gfx950 device: the product path has no CPU fallback */
#define KMC_ERR_TIME_OUT_OF_RANGE (-4) /* a stamp or requested_time outside [stamp_start, stamp_end]:
                                          the reference's release-mode assert, trajectory_interpolation.cpp:9,:32 */
#define KMC_ERR_ALLOC (-5)
#define KMC_ERR_DEGENERATE (-6)        /* stamp_start >= stamp_end, or a singular pose */

typedef enum kmc_mem_kind {
  KMC_MEM_HOST = 0,  /* any host memory.  Pageable: the library stages H2D / D2H (PCIe-bound).  Page-locked and addressed by the device
                        at its host address (the pool below, hipHostMalloc, torch's pin_memory()): recognised (one runtime lookup per
                        buffer) and handled like KMC_MEM_HOST_MAPPED; KMC_HOST_DETECT_PINNED=0 limits that to pool memory */
  KMC_MEM_DEVICE = 1, /* device memory of the ctx's GPU: zero-copy, the roofline path */
  KMC_MEM_HOST_MAPPED = 2 /* page-locked host memory the device can address (kmc_host_pool_alloc, kmc_hip_host_alloc, hipHostMalloc):
                             the kernel reads and writes it IN PLACE over the link -- one launch, upload and download overlapped, no
                             staging copies -- and the call returns when the results are in host memory.  Accepted by
                             kmc_hip_deskew_f32, kmc_hip_deskew_batch_f32, kmc_hip_deskew_f64cols, kmc_hip_deskew_traj_f64cols and
                             kmc_hip_pseudo_timestamps_f64; the same entry points take this route by themselves when every KMC_MEM_HOST
                             pointer they are given is page-locked (the pool below is what the C++ drop-in's containers and
                             KittiPclLoader::LoadRaw's clouds are made of). */
} kmc_mem_kind;

typedef struct kmc_ctx kmc_ctx; /* opaque */

/* Per-frame constants of the deskew: everything the device needs besides the points.
 * Produced on the host in f64 by kmc_frame_params_from_poses() -- the loop-invariant half of
 * GetPoseAtTime (trajectory_interpolation.cpp:35-36) hoisted out of the per-point loop. */
typedef struct kmc_frame_params {
  double twist[6]; /* f = Log(T_start^-1 * T_end) = [rho; phi] */
  double x_req;    /* (requested_time - stamp_start) / (stamp_end - stamp_start), in [0,1] */
} kmc_frame_params;

/* include/kitti_motion_compensation/data_types.hpp:35-49 */
typedef struct kmc_oxts {
  double stamp, lat, lon, alt, roll, pitch, yaw, vf, vl, vu;
} kmc_oxts;

typedef struct kmc_stats {
  uint64_t n_points;       /* points processed by the call */
  uint64_t n_out_of_range; /* points whose stamp was outside [stamp_start, stamp_end] (f64 path only) */
  uint32_t n_launches;     /* kernel launches issued */
  uint32_t variant;        /* f32 deskew entry points: the coefficient tier used: 0 = series3 (theta<=0.25), 1 = series5 (theta<=1), 2 = wide
                              polynomial (theta<=3.25), 3 = any angle.  Other routes: 4 = projection without deskew, 5 = the f64 Eigen-layout
                              kernels (kmc_hip_deskew_f64cols, kmc_hip_deskew_traj_f64cols) */
  float kernel_ms;         /* HIP-event time of the kernel launches (only if timing was enabled) */
  float total_ms;          /* HIP-event time of the whole call incl. H2D/D2H staging (only if timing enabled) */
} kmc_stats;

typedef struct kmc_device_info {
  char name[128];
  char arch[64];
  int device_id;
  int compute_units;
  int wavefront_size;
  uint64_t hbm_bytes;
  int clock_khz;
  int any_order_dispatch; /* barrier-free dispatch of independent frames (see kmc_hip_set_frame_queues): 1 = verified on this device at
                             kmc_hip_create and in use; 0 = switched off (KMC_ANY_ORDER=0); -1 = the run-time probe saw an ordinary packet
                             overtake, or not see the stores of, a barrier-free one (off); -3 = the probe could not run (off) */
} kmc_device_info;

/* ------------------------------------------------------------------------------------------------
 * library / context
 * ---------------------------------------------------------------------------------------------- */
int kmc_abi_version(void);
const char* kmc_status_string(int status);

/* Creates a context on HIP device `device_id`.  Fails with KMC_ERR_NO_DEVICE when there is none. */
int kmc_hip_create(kmc_ctx** out, int device_id);
void kmc_hip_destroy(kmc_ctx* ctx);
/* Use the caller's hipStream_t (e.g. torch's current stream) for all launches.  The handle is taken literally: NULL is
 * HIP's legacy default stream (which is what torch.cuda.current_stream().cuda_stream returns unless a side stream is
 * active), so work issued through the ctx stays ordered with the caller's own work on that stream. */
int kmc_hip_set_stream(kmc_ctx* ctx, void* hip_stream);
/* Go back to the context's private non-blocking stream (the state after kmc_hip_create). */
int kmc_hip_use_own_stream(kmc_ctx* ctx);
int kmc_hip_synchronize(kmc_ctx* ctx);
/* When enabled every call brackets its launches with hipEvents on the ctx stream, synchronizes, and fills
 * kmc_stats.kernel_ms / total_ms.  Off by default (calls are then fully asynchronous for KMC_MEM_DEVICE). */
int kmc_hip_enable_timing(kmc_ctx* ctx, int enabled);
const char* kmc_hip_last_error(kmc_ctx* ctx);
int kmc_hip_device_info(kmc_ctx* ctx, kmc_device_info* out);
/* Where the microseconds of an in-place call go (the routes that work on page-locked host buffers over the link: kmc_hip_deskew_f32 and
 * kmc_hip_deskew_f64cols[_begin/_end] with KMC_MEM_HOST_MAPPED or page-locked KMC_MEM_HOST pointers).  With the trace enabled every such
 * call leaves its stage times: host times are steady_clock microseconds (compare them with your own steady_clock stamps), device times
 * are the GPU's constant 100 MHz clock in microseconds since ITS epoch (only their difference means anything).  Off by default; costs
 * two 8-byte device writes per call when on.  kmc_hip_last_call_trace: KMC_ERR_INVALID_ARG while the trace is off. */
typedef struct kmc_call_trace {
  double issue_begin_us; /* host: the entry point (the _begin half) was entered                                  */
  double issue_end_us;   /* host: the launch has been enqueued, the entry point (the _begin half) returns        */
  double wait_begin_us;  /* host: the wait (the _end half) starts                                                */
  double wait_end_us;    /* host: the kernel's completion word was seen -- the results are in host memory        */
  double dev_first_wave_us; /* device clock: the first wave starts                                               */
  double dev_last_store_us; /* device clock: the last wave has stored and released                               */
  uint32_t waves;        /* persistent one-wave workgroups of the launch                                         */
  uint32_t route;        /* 1 = kmc_hip_deskew_f32 in place, 2 = kmc_hip_deskew_f64cols in place; 0 = the last call took another route */
} kmc_call_trace;
int kmc_hip_enable_call_trace(kmc_ctx* ctx, int enabled);
int kmc_hip_last_call_trace(kmc_ctx* ctx, kmc_call_trace* out);
/* Testing hook: force the coefficient tier (0..3, see kmc_stats.variant) of the f32 kernels (-1 = automatic selection from |phi|). */
int kmc_hip_force_tier(kmc_ctx* ctx, int tier);

/* HIP-event stopwatch on the ctx stream: begin records an event, end records another, waits for it and
 * returns the elapsed milliseconds between the two -- what bench.py uses around its timed region. */
int kmc_hip_timer_begin(kmc_ctx* ctx);
int kmc_hip_timer_end(kmc_ctx* ctx, float* elapsed_ms);

/* Page-locked host memory for KMC_MEM_HOST buffers (hipHostMalloc): the staging pipeline moves pinned buffers at the
 * link rate (~2.5 G points/s) instead of ~1 G points/s from pageable memory.  KITTI .bin payloads can be read straight
 * into such a buffer -- the file layout is the kernel's layout.  64-byte aligned. */
int kmc_hip_host_alloc(kmc_ctx* ctx, size_t bytes, void** out);
int kmc_hip_host_free(kmc_ctx* ctx, void* ptr);

/* Process-wide pool of page-locked, device-addressable host memory (no context needed; thread-safe).  Pinning a block costs ~100 us
 * or more, so freed blocks are cached (at most KMC_HOST_POOL_MAX_MB, default 2048, of free blocks) and reused.  The C++ drop-in's
 * Pointcloud / VectorXd allocate from it, which lets MotionCompensateFrame(Frame const&, Time) run as ONE kernel on the caller's
 * own containers.  kmc_host_pool_alloc: KMC_OK, or KMC_ERR_NO_DEVICE when there is no HIP device (or KMC_HOST_POOL=0) -- the caller
 * then uses ordinary memory and the staged KMC_MEM_HOST route.  kmc_host_pool_free: 1 if `ptr` was a live pool block (now
 * recycled), 0 if it is not the pool's.  kmc_host_pool_owns: 1 if [ptr, ptr + bytes) lies inside ONE live block.
 * kmc_host_pool_trim: unpins every cached free block, returns how many. */
int kmc_host_pool_alloc(size_t bytes, void** out);
/* The same with the block placed on `device`'s side of the machine whatever the calling thread's current HIP device is (and without making
 * any device current): for helper threads that allocate on behalf of a worker bound to another GPU.  device < 0: the current device. (ABI 7) */
int kmc_host_pool_alloc_near(size_t bytes, int device, void** out);
int kmc_host_pool_free(void* ptr);
int kmc_host_pool_owns(const void* ptr, size_t bytes);
int kmc_host_pool_trim(void);

/* NUMA.  Blocks of the pool are placed on the device's NUMA node (a PREFERRED memory policy around the allocation; the caller's own
 * policy is put back).  kmc_hip_bind_thread_near_device runs the CALLING thread on that node's CPUs (the PCI device's local_cpulist) --
 * what numactl / taskset do for a deployment: on a two-socket box the host side of a call crosses the inter-socket link otherwise (1.9
 * against 2.2 us per direct-queue call, 92 against 104-125 us per in-place KITTI frame).  Never done implicitly; KMC_OK also when there
 * is nothing to do. */
int kmc_hip_bind_thread_near_device(int device);

/* ------------------------------------------------------------------------------------------------
 * host pre-step (f64, pure host code, usable without a GPU)
 * ---------------------------------------------------------------------------------------------- */
/* twist = Log(T_start^-1 * T_end); x_req = FractionOfTrajectory(requested_time).
 * Replaces the loop-invariant work of TrajectoryInterpolator::GetPoseAtTime
 * (trajectory_interpolation.cpp:31-41 with lie_algebra.cpp:94-103) that the reference repeats twice per point.
 * KMC_ERR_TIME_OUT_OF_RANGE if requested_time is outside [stamp_start, stamp_end] (the reference aborts). */
int kmc_frame_params_from_poses(const double T_start[12], const double T_end[12], double stamp_start,
                                double stamp_end, double requested_time, kmc_frame_params* out);

/* data_io.cpp:68-88 OxtsToPose (Mercator + yaw/pitch/roll); scale = 1.0 is the reference's default. */
int kmc_oxts_to_pose(const kmc_oxts* oxts, double scale, double T_out[12]);
/* trajectory_interpolation.cpp:14-19 InterpolateTrajectory */
int kmc_interpolate_trajectory(const kmc_oxts* o1, const kmc_oxts* o2, double time, double T_out[12]);
/* data_io.cpp:253-269 MakeFrame (pose part): T_start from (o[n-1], o[n]) at stamp_start, T_end from
 * (o[n], o[n+1]) at stamp_end. */
int kmc_make_frame_poses(const kmc_oxts* o_nm1, const kmc_oxts* o_n, const kmc_oxts* o_np1, double stamp_start,
                         double stamp_end, double T_start_out[12], double T_end_out[12]);

/* Frame-range sharding (north_star: "one KITTI drive split into contiguous frame ranges per rank"; the frames of
 * motion_compensation.cpp:22-25's callers are independent, handlers.cpp:55-64).  Splits n_frames frames of
 * frame_points[f] points each into n_parts CONTIGUOUS ranges balanced on POINT counts: part r owns the frames whose
 * point-prefix midpoint lies in [r, r+1) * total / n_parts (integer arithmetic, no rounding).  bounds_out has n_parts + 1
 * entries, bounds_out[0] = 0, bounds_out[n_parts] = n_frames; part r = frames [bounds_out[r], bounds_out[r+1]).
 * With all-zero sizes the split is by frame count.  Used by MotionCompensateRun's multi-device driver and, through
 * ctypes, by the Python sharding helpers -- one definition for ranks and devices. */
int kmc_frame_ranges_balanced(const uint64_t* frame_points, uint32_t n_frames, uint32_t n_parts, uint32_t* bounds_out);

/* ------------------------------------------------------------------------------------------------
 * the hot path
 * ---------------------------------------------------------------------------------------------- */
/* Fused f32 deskew of ONE frame in the KITTI layout.  Per point (one HIP lane each):
 *   frac = (pi - atan2(y, x)) / 2pi                    timestamp_mocking.cpp:46     (GetPseudoTimeStamps)
 *   s    = frac - x_req                                trajectory_interpolation.cpp:49-51
 *   p'   = Exp(s * twist) * p                          motion_compensation.cpp:9-14 (closed form)
 * and intensity is passed through bit-identically.  32 algorithmic bytes per point (16 read + 16 written).
 * Device-resident xyzi_in and xyzi_out must be 16-byte aligned (host buffers: 4-byte, they are only copied) and must not
 * partially overlap (in == out is allowed).  Device-resident
 * pointers may be any 16-byte-aligned addresses (sub-ranges of a larger buffer): the kernels cut their tiles on the 1 KiB lines
 * of the output; full speed needs the input to sit at the same offset within its 1 KiB line (e.g. the same index range of
 * two allocator-aligned buffers), otherwise ~6 % is lost to split loads. */
int kmc_hip_deskew_f32(kmc_ctx* ctx, const float* xyzi_in, float* xyzi_out, uint64_t n,
                       const kmc_frame_params* params, int mem_kind, kmc_stats* out_stats);

/* ---- a stream of SEPARATE frames (the reference's calling pattern, handlers.cpp:55-64: one frame per call) ----------------------
 * One launch per frame is bound by the launch, not by HBM: the host pays 2.5-5 us per launch and the chip drains and refills between
 * two launches -- 6.1-7.0 us call to call for a 1 M-point frame whose kernel takes 4.7, 2.4-5 us for a KITTI frame whose kernel takes
 * 0.6.  The frames of this path are independent (motion_compensation.cpp:22-25 reads nothing a previous frame wrote).  Three ways out:
 *   (1) hand a LIST of ready frames to kmc_hip_deskew_frames_f32: one launch of the frame-list kernel for all of them (85 % of the HBM
 *       peak on 1 M-point frames -- the packed batch's rate --, 0.62 us per KITTI frame; gathered calls, (2), reach 84 % and 0.68 us);
 *   (2) kmc_hip_set_frame_queues(ctx, q > 1): keep calling kmc_hip_deskew_f32(KMC_MEM_DEVICE) once per frame and let the library GATHER
 *       the calls -- a call only adds its frame to a pending list (~0.1 us); the list goes out as ONE launch of the same frame-list
 *       kernel when the context's stream has run dry (looked at for the first frame and then every fourth: an idle device is not kept
 *       waiting, a busy one gathers while it works -- up to 64 frames), before a frame that touches a pending frame's buffers
 *       or needs another coefficient tier (so the frames' results are those of in-order execution, bit for bit), and before anything
 *       else the context puts on its stream -- every other entry point, kmc_hip_synchronize(), kmc_hip_timer_end(),
 *       kmc_hip_set_stream() and kmc_hip_frame_queue_join() issue the pending frames first.  What changes for the caller: a frame's
 *       launch may be DEFERRED until one of those calls.  Work the caller itself puts on the stream (or a hipDeviceSynchronize) does
 *       not see a pending frame: call kmc_hip_frame_queue_join() (or kmc_hip_synchronize) before consuming results outside the
 *       library -- and before freeing a pending frame's buffers or overwriting its input outside the library (the frame still has
 *       to read it; inside the library the hazard check orders such frames by itself).  (Until ABI 3 `queues` was a number of HIP streams the frames were spread over; the streams are gone, any value
 *       2..4 switches gathering on.)  With kmc_hip_enable_timing() on, calls are launched one by one (per-call times need it).
 *   (3) queues = 1 (default): every call is launched at once, in order on the context's stream.  Since ABI 3 "in order" does not
 *       mean "drained": a device-resident frame whose buffers overlap none of the frames launched since the last ordinary launch is
 *       dispatched WITHOUT the AQL barrier bit (hipExtAnyOrderLaunch): the packet processor does not wait for the completion and the
 *       cache release of the frame before it (it does not run whole kernels of one queue side by side either) -- the results of every
 *       frame, and everything the context or the caller puts on the stream afterwards (copies, events, other kernels: ordinary
 *       packets, which wait for all of them), are the same as before.  It applies (a) on the context's own stream, (b) on a caller's
 *       stream after kmc_hip_set_frame_queue_order(ctx, 0) -- the caller's word that nothing is produced between two calls --, never
 *       on HIP's legacy default stream (handle NULL), never while the stream captures a graph; at most 128 frames go out between two
 *       ordinary launches.  CONTRACT: the flag is documented as unsupported on gfx9, so the library does not take it on trust --
 *       kmc_hip_create runs a probe (< 1 ms, once per device and process) that must SEE, on this device and runtime, an ordinary
 *       kernel, a device-to-host copy and an event behind barrier-free kernels wait for all of them and read every word they stored
 *       (from every XCD); only then is the feature on (kmc_device_info.any_order_dispatch == 1), otherwise every launch is an
 *       ordinary one.  KMC_ANY_ORDER=0 switches it off unprobed.  Measured (bench.py's configs1_literal leg): 7.0 -> 6.1 us per
 *       1 M-point frame.
 * ALIASES.  Both (2) and (3) decide "these two frames are independent" by comparing the VIRTUAL ADDRESS RANGES of their buffers.  Two
 * mappings of one physical buffer (hipMemCreate + hipMemMap twice, an imported IPC handle next to the original allocation) are two
 * unrelated ranges to the library: it cannot see that they alias.  A caller who hands frames that communicate through an alias (one
 * writes through mapping m1 what the next reads through mapping m2) has to order them itself: kmc_hip_frame_queue_join() or
 * kmc_hip_synchronize() between the two calls, or a context created with KMC_ANY_ORDER=0 and queues = 1.  Without that the second frame
 * may read memory the first has not written yet.  tools/alias_probe.hip builds exactly that chain and counts the wrong results per
 * configuration (profiles/r05_alias_probe.json); tests/test_alias_contract.py holds the ordered configurations to zero.
 * kmc_hip_set_frame_queue_order(ctx, after_producers): what the library may assume about a CALLER's stream (kmc_hip_set_stream) --
 * 1 (default): the caller may have put a producer of the next frame on the stream since the last call; 0: every frame was produced
 * before the first call.  Only (3)'s barrier-free dispatch on a caller's stream depends on it. */
int kmc_hip_set_frame_queues(kmc_ctx* ctx, int queues);
int kmc_hip_set_frame_queue_order(kmc_ctx* ctx, int after_producers);
int kmc_hip_frame_queue_join(kmc_ctx* ctx);
/* Gathered frames and errors: a gathered call returns KMC_OK when its frame has been QUEUED.  If the launch that later issues the queue
 * fails (also as launches of 16 frames, the argument block every runtime takes), the frames of that queue were never
 * computed; the error is returned by whichever entry point triggered the join AND stays sticky -- kmc_hip_frame_queue_join and
 * kmc_hip_synchronize return it (once) even when the join happened inside an unrelated call -- and kmc_hip_frame_queue_dropped counts
 * the frames lost that way over the context's life (0 in every run so far: a launch only fails when the runtime itself is failing). */
uint64_t kmc_hip_frame_queue_dropped(kmc_ctx* ctx);
/* How many frames of this context have been dispatched without the barrier bit so far (a counter for tests and tuning). */
uint64_t kmc_hip_any_order_launches(kmc_ctx* ctx);
/* The in-place routes (KMC_MEM_HOST_MAPPED) wait for a completion word the kernel's last wave stores into page-locked memory.  If the
 * stream runs dry without the word, the call synchronises the stream instead (every store of a finished, synchronised kernel is in host
 * memory: the ordinary HIP contract) and returns normally; this counter says how often that happened over the context's life, and
 * last_state (may be NULL) receives the last event's {sequence number expected, word seen, completion ticket seen}.  0 is the normal
 * state; the event costs ~0.1 ms, never a result.  ctx == NULL: the total over every context the process has had (a per-thread context of
 * the C++ drop-in ends with its thread).  (ABI 6) */
uint64_t kmc_hip_completion_word_fallbacks(kmc_ctx* ctx, uint32_t last_state[3]);
/* THE DIRECT QUEUE (ABI 5; OPT-IN since ABI 7).  After kmc_hip_set_direct_dispatch(ctx, 1), on the context's OWN stream (the state after
 * kmc_hip_create), a device-resident kmc_hip_deskew_f32 call -- and a device-resident kmc_hip_deskew_traj_f32 call of up to four knots without
 * an index output (north_star's three bracketing poses: the segment records ride in the argument block) -- does not go through a HIP launch:
 * the library writes the frame's AQL dispatch packet into an HSA queue of the context's own, with the argument block in device memory --
 * 1.7-2.4 us per KITTI frame per call instead of the 3.5-4.9 us of the HIP runtime's launch path.  Same kernel body, same bits (checked on
 * the device when the queue is opened, at the first such call).
 * WHY IT IS NOT THE DEFAULT.  Frames in the direct queue are in NO HIP stream: hipDeviceSynchronize(), hipStreamSynchronize of a stream of the
 * caller's, hipFree's implicit wait and a caching allocator that recycles a buffer "after the stream has passed it" (torch's) do not
 * wait for them.  Code that is correct under the ordinary HIP rules -- free or reuse a frame's buffers after hipDeviceSynchronize() -- would
 * free memory under a running kernel.  A context therefore issues HIP launches until its owner says it plays by the queue's rules:
 *     kmc_hip_set_direct_dispatch(ctx, 1)   "I wait with kmc_hip_synchronize(ctx) (or any other call on the context -- every one of them
 *                                            waits for the frames before it) before I free, reuse or read a frame's buffers outside the
 *                                            library"; 0 switches back (waits for the frames still in the queue first).
 * The C++ drop-in (which owns its buffers and synchronises through the context) and the frame-stream clients under tools/ opt in.
 * KMC_DIRECT_DISPATCH=1 makes every context of the process start opted in (test runs of the whole suite on the queue);
 * KMC_DIRECT_DISPATCH=0 keeps the HIP launches even for a caller that opts in (A/B measurements).
 * The barrier bit of a packet is decided like for HIP launches: a frame that shares no buffer with the frames in flight goes out
 * without it (KMC_ANY_ORDER=0: every packet carries it).  TWO LANES: the direct queue is two HSA queues; independent frames alternate
 * between them (two packet processors fetch argument blocks and launch waves side by side); a frame that shares a buffer with frames in
 * flight in ONE lane follows them in that lane (barrier bit, nothing crosses lanes: a chain stays in its lane, buffer pairs used in
 * rotation keep both lanes busy); a frame with conflicts in both lanes goes to lane 0 behind a barrier packet that waits for lane 1, and
 * lane 1's next frame waits for that frame -- so a frame always sees what every frame called before it wrote, on whichever lane either
 * ran (with KMC_ANY_ORDER=0 one lane, every packet ordered).  ORDER: the direct queue and the context's HIP stream are separate queues; the
 * library keeps them in the order of the calls -- a frame waits for what the context put on its stream before it, every other entry point
 * waits for the frames before it (host waits, at such transitions only).  Not used on a caller's stream (kmc_hip_set_stream), with
 * gathering on or with per-call timing on; not available (HIP launches instead, kmc_hip_direct_dispatch_active == 0) where the host cannot
 * map device memory or the device is one partition of several at one PCI address.  A wait on the queue that exceeds ten seconds turns
 * into KMC_ERR_HIP, and the context goes back to HIP launches.
 * kmc_hip_direct_dispatch_active: 1 if the context's eligible frames really go through the queue now (opens it if asked for and not yet
 * open), else 0.  kmc_hip_direct_frames: frames this context has dispatched through its direct queue so far. */
int kmc_hip_set_direct_dispatch(kmc_ctx* ctx, int enabled);
int kmc_hip_direct_dispatch_active(kmc_ctx* ctx);
uint64_t kmc_hip_direct_frames(kmc_ctx* ctx);
/* n_frames separate device-resident frames in ONE call: frame f = n_points[f] points at xyzi_in[f] -> xyzi_out[f] with params[f]
 * (HOST arrays of device pointers / sizes / params), each frame in its own buffer (any 16-byte-aligned addresses).  ONE launch of the
 * frame-list kernel (2-D grid: frame x tile) on the context's stream with the frames' records IN ITS KERNEL ARGUMENTS -- up to 256 frames
 * per launch (a 56 KiB argument block; a KITTI drive of 108 frames is one launch of 24 KiB), longer lists in launches of 256: nothing is
 * uploaded, the host never waits, the call only enqueues.  Under stream capture the launches carry 16 frames each (the block every
 * runtime is known to take; so does everything after a runtime has refused a larger one -- the refused launch's frames and the
 * rest of the list, nothing twice).  Per-point results
 * are bit-identical to kmc_hip_deskew_f32 on the same frame: every frame runs at its own coefficient tier (a list that mixes tiers goes
 * out as one launch per tier present -- at most four; out_stats->variant reports the widest).  The frames must
 * be independent of each other (in == out of ONE frame is fine): a list in which one frame's output overlaps another frame's input or
 * output is recognised and issued frame by frame, in order, instead. */
int kmc_hip_deskew_frames_f32(kmc_ctx* ctx, const float* const* xyzi_in, float* const* xyzi_out, const uint64_t* n_points,
                              const kmc_frame_params* params, uint32_t n_frames, kmc_stats* out_stats);

/* Batched variant: n_frames frames concatenated in one buffer; frame f owns points
 * [offsets[f], offsets[f+1]).  offsets (n_frames+1 entries) and params (n_frames entries) are HOST arrays.
 * One launch covers the whole batch (a whole KITTI drive, or many 1M-point frames); per-frame constants
 * are staged in LDS.  frame_idx_out (optional, may be NULL; same mem_kind as the points) receives the
 * per-point frame index -- the integer "timestamp index" that is compared bit-exactly in the parity tests. */
int kmc_hip_deskew_batch_f32(kmc_ctx* ctx, const float* xyzi_in, float* xyzi_out, const uint64_t* offsets,
                             uint32_t n_frames, const kmc_frame_params* params, uint32_t* frame_idx_out,
                             int mem_kind, kmc_stats* out_stats);

/* f64 "Eigen-layout" deskew: honours caller-supplied per-point stamps exactly like
 * MotionCompensateFrame(Frame const&, Time) does (motion_compensation.cpp:22-25 reads frame.scan.timestamps).
 * Inputs: four columns x,y,z,w and stamps; outputs: four columns (w' = w, translation scaled by w like
 * Affine3d * Vector4d).  All arithmetic in f64 on the device.  Out-of-range stamps are counted in
 * out_stats->n_out_of_range and make the call return KMC_ERR_TIME_OUT_OF_RANGE (outputs for those points are NaN). */
int kmc_hip_deskew_f64cols(kmc_ctx* ctx, const double* x, const double* y, const double* z, const double* w,
                           const double* stamps, uint64_t n, double stamp_start, double stamp_end,
                           const kmc_frame_params* params, double* ox, double* oy, double* oz, double* ow,
                           int mem_kind, kmc_stats* out_stats);

/* The same call in two halves, for a caller with host work to overlap (the C++ drop-in fills the output's homogeneous column while
 * the kernel runs).  _begin checks the arguments and ISSUES the work; with device-addressable buffers (KMC_MEM_DEVICE,
 * KMC_MEM_HOST_MAPPED, or KMC_MEM_HOST pointers that are all page-locked) it returns without waiting, with staged host
 * buffers it completes the call.  _end waits and returns what kmc_hip_deskew_f64cols would have returned (KMC_ERR_TIME_OUT_OF_RANGE
 * included) and its stats.  The buffers belong to the library until _end returns, and no other call on the context in between --
 * except further _begin calls on device-addressable buffers: they queue up behind each other on the context's stream (K frames
 * back to back without a host wait between them) and ONE _end then waits for all of them and returns their combined verdict and
 * stats (points and launches summed, out-of-range stamps counted over all of them).  A _begin that cannot queue (staged host
 * buffers behind queued work, or behind a staged verdict that has not been collected) fails with KMC_ERR_INVALID_ARG and leaves the
 * queued work as it was. */
int kmc_hip_deskew_f64cols_begin(kmc_ctx* ctx, const double* x, const double* y, const double* z, const double* w,
                                 const double* stamps, uint64_t n, double stamp_start, double stamp_end,
                                 const kmc_frame_params* params, double* ox, double* oy, double* oz, double* ow, int mem_kind);
int kmc_hip_deskew_f64cols_end(kmc_ctx* ctx, kmc_stats* out_stats);

/* ---- N-knot trajectories: the 3-argument MotionCompensateFrame(Frame, Trajectory, Time) of BASELINE.json's north_star ----
 * The reference interpolates along ONE geodesic between the two scan poses.  These entry points generalise that to a
 * piecewise SE(3) geodesic through n_knots time-stamped poses (e.g. the three bracketing OXTS poses directly):
 *     T(t) = P_k * Exp(x * Log(P_k^-1 P_{k+1})),  t in [t_k, t_{k+1}],  x = (t - t_k)/(t_{k+1} - t_k)
 *     p'   = T(requested_time)^-1 * T(t_i) * p
 * knot_times[n_knots] strictly increasing; knot_poses[12 * n_knots] row-major 3x4; 2 <= n_knots <= 17.
 * The trajectory must cover requested_time and every point stamp (else KMC_ERR_TIME_OUT_OF_RANGE).
 * With n_knots == 2 and knots = {(stamp_start, T_start), (stamp_end, T_end)} the results are BIT-IDENTICAL to
 * kmc_hip_deskew_f32 / kmc_hip_deskew_f64cols.
 * bracket_idx_out (optional, NULL to skip; same mem_kind as the points): per-point segment index k -- the integer
 * "timestamp index".  In the f32 entry point it is decided by trig-free half-plane tests on (x, y) so that it is bit-exact
 * against the CPU oracle; in the f64 entry point by f64 compares of the caller's stamps against the knot times.
 * kmc_hip_deskew_traj_f32 on KMC_MEM_DEVICE points with n_knots <= 4 (north_star's three bracketing poses included) passes its
 * segment records in the kernel arguments: no table upload, the host never waits (12 us per call instead of 29), and -- like
 * kmc_hip_deskew_f32 calls, under the same conditions -- a frame that shares no buffer with the frames in flight is dispatched without
 * the barrier bit.  (N-knot calls are launched one per call; with gathering on, kmc_hip_set_frame_queues, they first issue the
 * two-pose frames that are pending.)  Longer trajectories and host buffers go through a device table (same kernel body, same bits)
 * on the context's stream.  kmc_hip_deskew_traj_f64cols with n_knots <= 4 carries its records in the kernel arguments as well (no
 * table upload in front of the kernel).  w == NULL means "the homogeneous column is all ones" (like kmc_hip_deskew_f64cols); if ow is
 * wanted all the same and the columns are host memory, the HOST fills it with ones while the kernel runs -- the column crosses the
 * link in neither direction. */
int kmc_hip_deskew_traj_f32(kmc_ctx* ctx, const float* xyzi_in, float* xyzi_out, uint64_t n, const double* knot_times,
                            const double* knot_poses, uint32_t n_knots, double stamp_start, double stamp_end,
                            double requested_time, uint32_t* bracket_idx_out, int mem_kind, kmc_stats* out_stats);
int kmc_hip_deskew_traj_f64cols(kmc_ctx* ctx, const double* x, const double* y, const double* z, const double* w,
                                const double* stamps, uint64_t n, const double* knot_times, const double* knot_poses,
                                uint32_t n_knots, double requested_time, double* ox, double* oy, double* oz, double* ow,
                                uint32_t* bracket_idx_out, int mem_kind, kmc_stats* out_stats);

/* Batched N-knot form: n_frames frames concatenated like in kmc_hip_deskew_batch_f32, every frame with its OWN trajectory
 * (e.g. its three bracketing OXTS poses), one launch for the whole batch.  frames[f] must satisfy what
 * kmc_hip_deskew_traj_f32 asks of a single frame.  frame_idx_out / bracket_idx_out (optional, NULL to skip; same mem_kind as
 * the points): per-point frame index and segment index, both bit-exact integers. */
typedef struct kmc_traj_frame {
  const double* knot_times; /* n_knots, strictly increasing (HOST)          */
  const double* knot_poses; /* 12 * n_knots, row-major 3x4 [R|t] (HOST)    */
  uint32_t n_knots;         /* 2 .. 17                                      */
  uint32_t reserved;        /* 0                                            */
  double stamp_start, stamp_end, requested_time;
} kmc_traj_frame;
int kmc_hip_deskew_traj_batch_f32(kmc_ctx* ctx, const float* xyzi_in, float* xyzi_out, const uint64_t* offsets,
                                  uint32_t n_frames, const kmc_traj_frame* frames, uint32_t* frame_idx_out,
                                  uint32_t* bracket_idx_out, int mem_kind, kmc_stats* out_stats);

/* GetPseudoTimeStamps (timestamp_mocking.cpp:56-63) on the device, f64: stamps[i] = start + frac_i*(end-start). */
int kmc_hip_pseudo_timestamps_f64(kmc_ctx* ctx, const double* x, const double* y, uint64_t n, double scan_start,
                                  double scan_end, double* stamps_out, int mem_kind);

/* ------------------------------------------------------------------------------------------------
 * next row N4: LiDAR -> image projection   (replaces the arithmetic of camera_model.cpp:5-95; the cv::circle
 * drawing itself stays with the caller)
 * ---------------------------------------------------------------------------------------------- */
/* The calibration the reference passes around as (CameraCalibrations, Affine3d tf_c00_lo) -- camera_model.hpp:10-11. */
typedef struct kmc_camera_rig {
  double tf_c00_lo[12]; /* row-major 3x4 velodyne -> camera 00 (LoadLidarExtrinsics, data_io.cpp:168-210)          */
  double R_rect_00[9];  /* row-major, CameraCalibration::R_rect of camera 00 (camera_model.cpp:78-79)              */
  double P_rect[4][12]; /* row-major 3x4, CameraCalibration::P_rect of cameras 00..03 (camera_model.cpp:9)         */
  double max_range;     /* camera_model.hpp:8 (default 15.0)                                                       */
} kmc_camera_rig;

/* Per point i of the cloud and camera c (f64 on the device, operation for operation like the reference, so the integers
 * are bit-exact against the CPU oracle):
 *   uv[(i*4 + c)*2 + {0,1}]  the integer pixel cv::circle is centred on (camera_model.cpp:12, :31), or INT32_MIN twice
 *                            when the point is skipped by camera_model.cpp:21-24 (z_rect outside [0.01, max_range] or
 *                            y_rect > 1.25)
 *   bgrv[4*i + {0,1,2,3}]    the 8-bit colour {255-cs, cs, 255-cs} the reference draws it with (camera_model.cpp:28-32)
 *                            and 1, or four zeros when skipped
 * deskew == NULL: the cloud is projected as it is.  deskew != NULL: every point is motion-compensated first exactly like
 * kmc_hip_deskew_f32 does (fused: the cloud is read once) and, when xyzi_out != NULL, the compensated cloud is written
 * there -- GenerateProjectionVisualizationOfRun's "project raw, compensate, project again" (handlers.cpp:77-88) in two
 * launches.  The eight pixel integers of a point are contiguous (32 bytes: one coalesced 2 KiB store per wave whatever n is).
 * Algorithmic bytes per point: 16 read + 36 written (+16 with xyzi_out). */
int kmc_hip_project_f32(kmc_ctx* ctx, const float* xyzi_in, uint64_t n, const kmc_camera_rig* rig,
                        const kmc_frame_params* deskew, float* xyzi_out, int32_t* uv, uint8_t* bgrv, int mem_kind,
                        kmc_stats* out_stats);
/* The same for an Eigen-layout cloud (three f64 columns): ProjectPointcloudOnFrame(frame, ...) on frame.scan.cloud. */
int kmc_hip_project_f64cols(kmc_ctx* ctx, const double* x, const double* y, const double* z, uint64_t n,
                            const kmc_camera_rig* rig, int32_t* uv, uint8_t* bgrv, int mem_kind, kmc_stats* out_stats);

/* ------------------------------------------------------------------------------------------------
 * synthetic workload (measurement infrastructure; BASELINE.json configs 2-5 have no shippable data)
 * ---------------------------------------------------------------------------------------------- */
/* Fills xyzi_out (DEVICE memory) with n synthetic Velodyne-like points generated on the GPU from a
 * counter-based PRNG keyed on (seed, point index): 64 rings, azimuth sweeping the full circle, range
 * U[2,80) m, intensity on a 0.01 grid.  kmc_synth_points_host() produces the bit-identical points on the
 * host (used by the tests to feed the oracle). */
int kmc_hip_synth_points(kmc_ctx* ctx, float* xyzi_out_device, uint64_t n, uint64_t seed);
int kmc_synth_points_host(float* xyzi_out_host, uint64_t n, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif /* KMC_HIP_H */

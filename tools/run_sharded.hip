// run_sharded.hip -- north_star's multi-GPU shape from C++, in ONE process: a stream of synthetic frames cut into contiguous frame ranges
// (kmc_frame_ranges_balanced, the split kmc::MotionCompensateRun's multi-device driver and the per-rank launch use), one worker thread with
// its own device context per range, NO point data between devices -- and ONE RCCL reduction of the counters at the end, the way
// SURVEY.md section 8(e) words it: ncclCommInitAll over the devices, ncclAllReduce(sum) of the points, ncclAllReduce(max) of the seconds
// (until round 4 the only RCCL call of the repository was torch.distributed's, from bench.py).
//
//   run_sharded [devices=0] [frames=64] [points_per_frame=1000000] [frames_per_launch=8]
//     devices: comma-separated HIP device ids, one rank each ("0,1,2,3,4,5,6,7" on an 8-GPU node).  An id may repeat ("0,0"): the
//     ranks then share a GPU -- RCCL refuses a communicator with a duplicate device, the tool says so and reduces on the host instead
//     (the JSON names which path reduced: "reduced_by").  "all": one rank per device this process can see (hipGetDeviceCount under
//     whatever HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES the caller set) -- what bench.py's leg passes, so that the first run on a
//     multi-GPU node forms a real RCCL group without anybody editing a command line.
// librccl.so (570 MB) is opened with dlopen only here: the product libraries do not link it.  Prints ONE JSON object.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "kmc_hip.h"

namespace {
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool open() {
    lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) return false;
    CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
    AllReduce = (decltype(AllReduce))dlsym(lib, "ncclAllReduce");
    GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
    CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
    GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
    return CommInitAll && AllReduce && GroupStart && GroupEnd && CommDestroy && GetErrorString;
  }
};

struct Barrier {  // all ranks start their timed region together (the contract's barrier, in-process)
  std::mutex mu;
  std::condition_variable cv;
  int waiting = 0, generation = 0, n = 1;
  void arrive() {
    std::unique_lock<std::mutex> lock(mu);
    const int gen = generation;
    if (++waiting == n) {
      waiting = 0;
      ++generation;
      cv.notify_all();
    } else {
      cv.wait(lock, [&] { return gen != generation; });
    }
  }
};

struct RankResult {
  int device = 0;
  uint32_t first = 0, last = 0;
  double points = 0, seconds = 0;
  std::string error;
};
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

int main(int argc, char** argv) {
  std::vector<int> devices;
  {
    if (argc > 1 && std::strcmp(argv[1], "all") == 0) {
      int count = 0;
      if (hipGetDeviceCount(&count) != hipSuccess || count < 1) {
        std::fprintf(stderr, "run_sharded all: no HIP device visible\n");
        return 2;
      }
      for (int d = 0; d < count; ++d) devices.push_back(d);
    } else {
      std::istringstream is(argc > 1 ? argv[1] : "0");
      std::string tok;
      while (std::getline(is, tok, ','))
        if (!tok.empty()) devices.push_back(std::atoi(tok.c_str()));
    }
  }
  const uint32_t n_frames = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 64u;
  const uint64_t per = argc > 3 ? std::strtoull(argv[3], nullptr, 10) : 1000000ull;
  const uint32_t per_launch = argc > 4 ? (uint32_t)std::atoi(argv[4]) : 8u;
  const int world = (int)devices.size();
  if (world < 1 || n_frames < (uint32_t)world || per_launch < 1 || per_launch > 16) {
    std::fprintf(stderr, "usage: run_sharded [devices=0] [frames=64] [points_per_frame=1000000] [frames_per_launch=8 (<= 16)]\n");
    return 2;
  }
  std::vector<uint64_t> frame_points(n_frames, per);
  std::vector<uint32_t> bounds((size_t)world + 1);
  if (kmc_frame_ranges_balanced(frame_points.data(), n_frames, (uint32_t)world, bounds.data()) != KMC_OK) return 2;

  std::vector<RankResult> res((size_t)world);
  std::vector<double*> d_counters((size_t)world, nullptr);  // {points, seconds} of the rank, in its device's memory: what RCCL reduces
  std::vector<hipStream_t> streams((size_t)world, nullptr);
  Barrier barrier;
  barrier.n = world;
  std::vector<std::thread> workers;
  for (int r = 0; r < world; ++r) {
    workers.emplace_back([&, r] {
      RankResult& me = res[(size_t)r];
      me.device = devices[(size_t)r];
      me.first = bounds[(size_t)r];
      me.last = bounds[(size_t)r + 1];
      kmc_ctx* ctx = nullptr;
      float *d_in = nullptr, *d_out = nullptr;
      auto bail = [&](const char* what) { me.error = what; barrier.arrive(); barrier.arrive(); };
      (void)kmc_hip_bind_thread_near_device(me.device);  // a placement hint: this rank's thread on its device's side of the machine
      if (hipSetDevice(me.device) != hipSuccess || kmc_hip_create(&ctx, me.device) != KMC_OK) return bail("no device context");
      const uint32_t mine = me.last - me.first;
      const uint32_t resident = std::min<uint32_t>(std::max<uint32_t>(mine, 1), 2 * per_launch);  // rotating groups of distinct frames
      if (hipMalloc((void**)&d_in, (size_t)resident * per * 16) != hipSuccess || hipMalloc((void**)&d_out, (size_t)resident * per * 16) != hipSuccess ||
          hipMalloc((void**)&d_counters[(size_t)r], 2 * sizeof(double)) != hipSuccess || hipStreamCreateWithFlags(&streams[(size_t)r], hipStreamNonBlocking) != hipSuccess)
        return bail("allocation failed");
      for (uint32_t j = 0; j < resident; ++j) kmc_hip_synth_points(ctx, d_in + 4 * (size_t)j * per, per, 0x4B4D43ull + 0xE5000000ull + me.first + j);
      std::vector<kmc_frame_params> params(per_launch);
      std::vector<uint64_t> offsets(per_launch + 1);
      for (uint32_t k = 0; k <= per_launch; ++k) offsets[k] = (uint64_t)k * per;
      auto sweep = [&](bool count) {
        double pts = 0;
        for (uint32_t f = me.first; f < me.last; f += per_launch) {
          const uint32_t m = std::min(per_launch, me.last - f);
          for (uint32_t k = 0; k < m; ++k) {  // a turning vehicle: every frame its own twist (frame index of the DRIVE, not of the rank)
            const double g = 1.0 + 0.001 * ((f + k) % 97);
            const double tw[6] = {1.3 * g, 0.05, -0.02, 0.002, -0.004, 0.03 * g};
            std::memcpy(params[k].twist, tw, sizeof(tw));
            params[k].x_req = 0.5;
          }
          const uint32_t slot = ((f - me.first) / per_launch) % std::max<uint32_t>(1, resident / per_launch);
          const size_t base = (size_t)slot * per_launch * per * 4;
          offsets[m] = (uint64_t)m * per;
          if (kmc_hip_deskew_batch_f32(ctx, d_in + base, d_out + base, offsets.data(), m, params.data(), nullptr, KMC_MEM_DEVICE, nullptr) != KMC_OK) me.error = "deskew failed";
          if (count) pts += (double)m * (double)per;
        }
        kmc_hip_synchronize(ctx);
        return pts;
      };
      sweep(false);  // warm-up: clocks, tables, code objects
      barrier.arrive();
      const double t0 = now_s();
      me.points = sweep(true);
      me.seconds = now_s() - t0;
      barrier.arrive();
      const double counters[2] = {me.points, me.seconds};
      if (hipMemcpy(d_counters[(size_t)r], counters, sizeof(counters), hipMemcpyHostToDevice) != hipSuccess) me.error = "counter upload failed";
      (void)hipFree(d_in);
      (void)hipFree(d_out);
      kmc_hip_destroy(ctx);
    });
  }
  for (auto& w : workers) w.join();
  for (const RankResult& r : res)
    if (!r.error.empty()) {
      std::fprintf(stderr, "rank on device %d: %s\n", r.device, r.error.c_str());
      return 1;
    }

  // ---- the ONE collective of the job: sum of the points, max of the seconds ----
  double host_points = 0, host_seconds = 0;
  for (const RankResult& r : res) { host_points += r.points; host_seconds = std::max(host_seconds, r.seconds); }
  std::string reduced_by = "host (RCCL not used)", rccl_note;
  double red_points = host_points, red_seconds = host_seconds;
  Rccl rccl;
  if (!rccl.open()) {
    rccl_note = "librccl.so.1 could not be opened";
  } else {
    std::vector<ncclComm_t> comms((size_t)world);
    const ncclResult_t init = rccl.CommInitAll(comms.data(), world, devices.data());
    if (init != ncclSuccess) {
      rccl_note = std::string("ncclCommInitAll refused the device list: ") + rccl.GetErrorString(init) + (world > 1 ? " (a GPU may appear once per communicator)" : "");
    } else {
      bool ok = rccl.GroupStart() == ncclSuccess;
      for (int r = 0; r < world && ok; ++r) {
        ok = hipSetDevice(devices[(size_t)r]) == hipSuccess;
        ok = ok && rccl.AllReduce(d_counters[(size_t)r], d_counters[(size_t)r], 1, ncclDouble, ncclSum, comms[(size_t)r], streams[(size_t)r]) == ncclSuccess;
        ok = ok && rccl.AllReduce(d_counters[(size_t)r] + 1, d_counters[(size_t)r] + 1, 1, ncclDouble, ncclMax, comms[(size_t)r], streams[(size_t)r]) == ncclSuccess;
      }
      ok = ok && rccl.GroupEnd() == ncclSuccess;
      for (int r = 0; r < world && ok; ++r) ok = hipSetDevice(devices[(size_t)r]) == hipSuccess && hipStreamSynchronize(streams[(size_t)r]) == hipSuccess;
      double got[2] = {0, 0};
      ok = ok && hipSetDevice(devices[0]) == hipSuccess && hipMemcpy(got, d_counters[0], sizeof(got), hipMemcpyDeviceToHost) == hipSuccess;
      if (ok) {
        red_points = got[0];
        red_seconds = got[1];
        reduced_by = "rccl: ncclCommInitAll + ncclAllReduce(sum of points) + ncclAllReduce(max of seconds), one group";
      } else {
        rccl_note = "an RCCL / HIP call of the reduction failed";
      }
      for (auto& c : comms) rccl.CommDestroy(c);
    }
  }
  const bool agrees = red_points == host_points && red_seconds == host_seconds;
  std::printf("{\"world\": %d, \"devices\": [", world);
  for (int r = 0; r < world; ++r) std::printf("%s%d", r ? ", " : "", devices[(size_t)r]);
  std::printf("], \"frames\": %u, \"points_per_frame\": %llu, \"frames_per_launch\": %u, \"ranks\": [", n_frames, (unsigned long long)per, per_launch);
  for (int r = 0; r < world; ++r)
    std::printf("%s{\"device\": %d, \"frames\": [%u, %u], \"points\": %.0f, \"seconds\": %.6f, \"Mpts_s\": %.1f}", r ? ", " : "", res[(size_t)r].device, res[(size_t)r].first, res[(size_t)r].last,
                res[(size_t)r].points, res[(size_t)r].seconds, res[(size_t)r].points / res[(size_t)r].seconds / 1e6);
  std::printf("], \"reduced\": {\"points\": %.0f, \"seconds_max\": %.6f, \"Mpts_s\": %.1f}, \"reduced_by\": \"%s\", \"rccl_note\": \"%s\", \"reduction_agrees_with_host_arithmetic\": %s}\n",
              red_points, red_seconds, red_points / red_seconds / 1e6, reduced_by.c_str(), rccl_note.c_str(), agrees ? "true" : "false");
  return agrees ? 0 : 1;
}

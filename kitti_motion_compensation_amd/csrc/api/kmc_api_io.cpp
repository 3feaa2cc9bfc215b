// kmc_api_io.cpp -- KITTI on-disk formats either side of the hot path and the per-run driver (rows N1-N3 of
// SURVEY.md section 8(f)).  Host code.  Reference: src/kitti_motion_compensation/{data_io,utils,handlers}.cpp.
//
// Deliberate divergences from the reference (all documented in DESIGN.md "Next rows"):
//   * an unreadable timestamp file throws std::runtime_error; the reference prints and calls exit(0) (data_io.cpp:27-30);
//   * the point-cloud loader sizes its buffer from the file; the reference reads into a fixed 250 000-point buffer
//     with no bounds check (data_io.hpp:17, data_io.cpp:115);
//   * MotionCompensateRun reads each text file once per run and deskews frames in GPU batches straight from / to the
//     f32 on-disk layout; the reference re-scans the timestamp files per frame and goes through f64 Eigen matrices.
//     The written bytes differ from the reference's by at most f32 rounding of the result (<= 1e-5 relative bar).
//   * CopyOverUncompensatedFirstAndLastFrame reproduces the reference's output exactly by default -- including its
//     slip of writing the FIRST frame's points under the LAST frame's id (handlers.cpp:36-38); set
//     KMC_FIX_LAST_FRAME_COPY=1 to write the last frame's own points instead.
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <exception>
#include <mutex>
#include <thread>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <system_error>

#include "kitti_motion_compensation/data_io.hpp"
#include "kitti_motion_compensation/handlers.hpp"
#include "kitti_motion_compensation/motion_compensation.hpp"
#include "kitti_motion_compensation/timestamp_mocking.hpp"
#include "kitti_motion_compensation/trajectory_interpolation.hpp"
#include "kitti_motion_compensation/utils.hpp"
#include "kmc_api_internal.hpp"

namespace kmc {

namespace fs = std::filesystem;

// ---- utils.cpp ------------------------------------------------------------------------------------
std::string IdToZeroPaddedString(std::size_t const id, std::size_t const pad) {  // utils.cpp:10-15
  std::string const s{std::to_string(id)};
  return s.length() >= pad ? s : std::string(pad - s.length(), '0') + s;
}

std::vector<std::string> TokenizeString(std::string raw_string) {  // :17-29 (split on single spaces, empty tokens kept)
  std::vector<std::string> tokens;
  std::istringstream stream(raw_string);
  std::string token;
  while (std::getline(stream, token, ' ')) tokens.push_back(token);
  return tokens;
}

double MmHhSsToSeconds(std::string const s) {  // :31-38  "HH:MM:SS.nnnnnnnnn"
  int const hours{std::stoi(s.substr(0, 2))};
  int const minutes{std::stoi(s.substr(3, 5))};
  double const seconds{std::stod(s.substr(6, 18))};
  return static_cast<double>((60 * hours * 60) + (minutes * 60)) + seconds;
}

// ---- data_io.cpp -------------------------------------------------------------------------------------
namespace {

std::vector<Time> LoadAllTimeStamps(Path const& file) {
  std::ifstream is(file);
  if (!is.is_open()) throw std::runtime_error("Failed to open timestamp file: " + file.string());
  std::vector<Time> out;
  std::string line;
  while (std::getline(is, line)) {
    if (line.empty()) continue;
    auto const tokens = TokenizeString(line);
    if (tokens.size() < 2) throw std::runtime_error("Malformed timestamp line in " + file.string());
    try {
      out.push_back(MmHhSsToSeconds(tokens[1]));
    } catch (std::logic_error const&) {  // substr / stoi / stod on a token that is not HH:MM:SS.fraction
      throw std::runtime_error("Malformed timestamp line in " + file.string() + ": " + line);
    }
  }
  return out;
}

Oxts ParseOxtsLine(std::string const& line, Time stamp, Path const& file) {
  auto const t = TokenizeString(line);
  if (t.size() < 11) throw std::runtime_error("Malformed OXTS packet: " + file.string());
  try {
    return Oxts{stamp, std::stod(t[0]), std::stod(t[1]), std::stod(t[2]), std::stod(t[3]), std::stod(t[4]),
                std::stod(t[5]), std::stod(t[8]), std::stod(t[9]), std::stod(t[10])};  // data_io.cpp:56-65
  } catch (std::logic_error const&) {  // a field that is not a number
    throw std::runtime_error("Malformed OXTS packet: " + file.string());
  }
}

Oxts LoadOxtsWithStamp(Path const& folder, std::size_t frame_id, Time stamp) {
  Path const file(folder / Path("oxts/data/" + IdToZeroPaddedString(frame_id) + ".txt"));
  std::ifstream is(file);
  if (!is.is_open()) throw std::runtime_error("The Oxts file you tried to load did not open: " + file.string());  // :51
  std::string line;
  std::getline(is, line);
  return ParseOxtsLine(line, stamp, file);
}

}  // namespace

Time LoadTimeStamp(Path const timestamp_file, std::size_t const frame_id) {  // :18-35
  auto const all = LoadAllTimeStamps(timestamp_file);
  if (frame_id >= all.size()) throw std::runtime_error("No timestamp for frame " + std::to_string(frame_id) + " in " + timestamp_file.string());
  return all[frame_id];
}

Oxts LoadOxts(Path const folder, std::size_t const frame_id) {  // :37-66
  return LoadOxtsWithStamp(folder, frame_id, LoadTimeStamp(folder / Path("oxts/timestamps.txt"), frame_id));
}

KittiCloudF32 KittiPclLoader::LoadRaw(Path const& file) {
  std::ifstream is{file, std::ios::in | std::ios::binary | std::ios::ate};
  if (!is.is_open()) throw std::runtime_error("Unable to open requested KITTI pointcloud binary file: " + file.string());  // :104
  std::int64_t const bytes{static_cast<std::int64_t>(is.tellg())};
  if (bytes == -1 || (bytes % 4) != 0) throw std::runtime_error("Opened KITTI pointcloud binary file is incorrectly formatted: " + file.string());  // :109
  KittiCloudF32 data(static_cast<std::size_t>(bytes) / 16 * 4);  // whole points only (16 bytes each), :112
  is.seekg(0, std::ios::beg);
  is.read(reinterpret_cast<char*>(data.data()), static_cast<std::streamsize>(data.size() * sizeof(float)));
  return data;
}

std::tuple<Pointcloud, VectorXd> KittiPclLoader::LoadPointcloud(Path const& file) {  // :101-138
  KittiCloudF32 const raw = LoadRaw(file);
  Index const n = static_cast<Index>(raw.size() / 4);
  Pointcloud cloud{MatrixX4d(n, 4)};
  VectorXd intensities(n);
  for (Index i = 0; i < n; ++i) {
    cloud(i, 0) = raw[4 * i + 0];
    cloud(i, 1) = raw[4 * i + 1];
    cloud(i, 2) = raw[4 * i + 2];
    cloud(i, 3) = 1.0;  // homogeneous component
    intensities(i) = raw[4 * i + 3];
  }
  detail::set_homogeneous(cloud, true);  // the loop above has just written the ones
  return {cloud, intensities};
}

LidarScan LoadLidarScan(Path const folder, std::size_t const frame_id) {  // :142-166
  Time const start{LoadTimeStamp(folder / Path("velodyne_points/timestamps_start.txt"), frame_id)};
  Time const middle{LoadTimeStamp(folder / Path("velodyne_points/timestamps.txt"), frame_id)};
  Time const end{LoadTimeStamp(folder / Path("velodyne_points/timestamps_end.txt"), frame_id)};
  KittiPclLoader loader;
  auto [cloud, intensities] = loader.LoadPointcloud(folder / Path("velodyne_points/data/" + IdToZeroPaddedString(frame_id) + ".bin"));
  VectorXd const stamps{GetPseudoTimeStamps(cloud, start, end)};  // on the GPU
  return LidarScan{start, middle, end, cloud, intensities, stamps};
}

Frame MakeFrame(Oxts const& o_nm1, Oxts const& o_n, Oxts const& o_np1, LidarScan const& scan) {  // :253-269
  Affine3d const start_pose{trajectory_interpolation::InterpolateTrajectory(o_nm1, o_n, scan.stamp_start)};
  Affine3d const end_pose{trajectory_interpolation::InterpolateTrajectory(o_n, o_np1, scan.stamp_end)};
  return Frame(start_pose, end_pose, scan);
}

Frame LoadSingleFrame(Path const data_folder, std::size_t const frame_id, bool const load_images) {  // :271-285
  if (load_images) throw std::runtime_error("LoadSingleFrame: loading the camera images needs OpenCV, which this build does not use");
  if (frame_id == 0) throw std::invalid_argument("LoadSingleFrame: frame 0 has no previous OXTS packet");  // reference: size_t underflow
  Oxts const a{LoadOxts(data_folder, frame_id - 1)}, b{LoadOxts(data_folder, frame_id)}, c{LoadOxts(data_folder, frame_id + 1)};
  return MakeFrame(a, b, c, LoadLidarScan(data_folder, frame_id));
}

void WriteRaw(Path const data_folder, std::size_t const frame_id, float const* xyzi, std::size_t n) {
  Path const file(data_folder / Path(IdToZeroPaddedString(frame_id) + ".bin"));
  std::ofstream out(file, std::ios::out | std::ios::binary);
  if (!out.is_open()) throw std::runtime_error("Unable to open output pointcloud file: " + file.string());
  out.write(reinterpret_cast<char const*>(xyzi), static_cast<std::streamsize>(n * 4 * sizeof(float)));
  out.close();
  if (out.fail()) throw std::runtime_error("Failed writing output pointcloud file: " + file.string());  // (the reference does not look: data_io.cpp:287-313)
}

void WritePointcloud(Path const data_folder, std::size_t const frame_id, Pointcloud const& cloud, VectorXd const& intensities) {  // :287-313
  std::size_t const n = static_cast<std::size_t>(cloud.rows());
  std::vector<float> buf(n * 4);
  for (std::size_t i = 0; i < n; ++i) {
    buf[4 * i + 0] = static_cast<float>(cloud(static_cast<Index>(i), 0));
    buf[4 * i + 1] = static_cast<float>(cloud(static_cast<Index>(i), 1));
    buf[4 * i + 2] = static_cast<float>(cloud(static_cast<Index>(i), 2));
    buf[4 * i + 3] = static_cast<float>(intensities(static_cast<Index>(i)));
  }
  WriteRaw(data_folder, frame_id, buf.data(), n);
}

// ---- handlers.cpp --------------------------------------------------------------------------------------
std::size_t NumberOfFilesInDirectory(fs::path path) {  // handlers.cpp:15-17
  return static_cast<std::size_t>(std::distance(fs::directory_iterator{path}, fs::directory_iterator{}));
}

void CopyOverUncompensatedFirstAndLastFrame(Path const run_folder) {  // :19-39
  Path const velodyne{run_folder / Path{"velodyne_points"}};
  Path const out_dir{velodyne / Path("data_motion_compensated")};
  std::size_t const n_frames{NumberOfFilesInDirectory(velodyne / Path("data"))};
  KittiCloudF32 const first = KittiPclLoader::LoadRaw(velodyne / Path("data/" + IdToZeroPaddedString(0) + ".bin"));
  WriteRaw(out_dir, 0, first.data(), first.size() / 4);
  std::size_t const last_id{n_frames - 1};
  char const* fix = std::getenv("KMC_FIX_LAST_FRAME_COPY");
  if (fix && fix[0] == '1') {
    KittiCloudF32 const last = KittiPclLoader::LoadRaw(velodyne / Path("data/" + IdToZeroPaddedString(last_id) + ".bin"));
    WriteRaw(out_dir, last_id, last.data(), last.size() / 4);
  } else {
    WriteRaw(out_dir, last_id, first.data(), first.size() / 4);  // what the reference writes (handlers.cpp:36-38)
  }
}

// ---- devices of the run driver ------------------------------------------------------------------------------------------
namespace {
std::mutex g_devices_mu;
std::vector<int> g_run_devices;  // empty: $KMC_DEVICES, else the calling thread's device

std::vector<int> ParseDeviceList(char const* text) {
  std::vector<int> out;
  std::string token;
  std::istringstream is{std::string(text)};
  while (std::getline(is, token, ',')) {
    if (token.empty()) continue;
    std::size_t used = 0;
    int const v = std::stoi(token, &used);
    if (used != token.size() || v < 0) throw std::invalid_argument("KMC_DEVICES: not a device list: " + std::string(text));
    out.push_back(v);
  }
  return out;
}
}  // namespace

namespace hip {
void SetRunDevices(std::vector<int> const& devices) {
  for (int d : devices)
    if (d < 0) throw std::invalid_argument("kmc::hip::SetRunDevices: negative device id");
  std::lock_guard<std::mutex> lock(g_devices_mu);
  g_run_devices = devices;
}
std::vector<int> GetRunDevices() {
  {
    std::lock_guard<std::mutex> lock(g_devices_mu);
    if (!g_run_devices.empty()) return g_run_devices;
  }
  if (char const* e = std::getenv("KMC_DEVICES")) {
    auto list = ParseDeviceList(e);
    if (!list.empty()) return list;
  }
  return {GetDevice()};
}
}  // namespace hip

namespace {
// One contiguous frame range [first, last) of a run on ONE device: batches of <= kMaxBatchFrames frames through a
// three-stage pipeline on two page-locked buffer sets:
//   reader thread : .bin payloads read STRAIGHT into pinned memory (the on-disk layout is the kernel's layout: no
//                   conversion, no extra copy) + the per-frame poses (MakeFrame, data_io.cpp:253-269)
//   this thread   : one batched GPU call per buffer set (H2D, kernel, D2H) on the thread's own device context
//   writer thread : results written from pinned memory, a batch's files split over two threads (writing is the longest stage)
// so reading batch k+1, deskewing batch k and writing batch k-1 overlap.  The second buffer set is page-locked while the reader
// already fills the first (page-locking costs ~0.3 ms/MiB: a third of a short run's pipeline time).
// The page-locked buffers of a run, made by a helper thread in the order the pipeline needs them (set 0 in, set 0 out, set 1 in,
// set 1 out) out of the C-ABI's context-free pool -- so that the HIP runtime's start-up (~30 ms) and the page-locking (~0.3 ms per MiB)
// run WHILE the caller parses the run's text files and the reader already fills the first buffer, instead of in front of them
// (round 5; until then "context + pinned" was a serial 49 ms of a 188 ms run).  unavailable(): the pool is switched off
// (KMC_HOST_POOL=0) -- the caller then allocates through its device context as before.
class PinnedSets {
 public:
  // `device`: whose side of the machine the blocks are placed on -- the helper thread's own current device is the default one, which
  // is neither the worker's (KMC_DEVICE=N, the multi-device run) nor one this thread should initialise
  PinnedSets(std::size_t floats_per_buffer, int n_buffers, int device) : floats_(floats_per_buffer), n_(std::min(n_buffers, 4)) {
    worker_ = std::thread([this, device] {
      for (int k = 0; k < n_; ++k) {
        void* q = nullptr;
        int const rc = kmc_host_pool_alloc_near(floats_ * sizeof(float), device, &q);
        std::lock_guard<std::mutex> lock(mu_);
        if (rc != KMC_OK || !q) {
          failed_rc_ = rc != KMC_OK ? rc : KMC_ERR_ALLOC;
          cv_.notify_all();
          return;
        }
        buf_[k] = static_cast<float*>(q);
        ready_ = k + 1;
        if (k == 0) first_ready_ms_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_created_).count();
        cv_.notify_all();
      }
    });
  }
  PinnedSets(PinnedSets const&) = delete;
  PinnedSets& operator=(PinnedSets const&) = delete;
  ~PinnedSets() {
    if (worker_.joinable()) worker_.join();
    for (float* p : buf_)
      if (p) kmc_host_pool_free(p);
  }
  // buffer `index`, waiting for the helper if it is not there yet; nullptr: the pool declined (unavailable)
  float* wait(int index) {
    std::unique_lock<std::mutex> lock(mu_);
    cv_.wait(lock, [&] { return ready_ > index || failed_rc_ != KMC_OK; });
    return ready_ > index ? buf_[index] : nullptr;
  }
  int failed_rc() {
    std::lock_guard<std::mutex> lock(mu_);
    return failed_rc_;
  }
  // buffers the helper has made so far (final once failed_rc() != KMC_OK)
  int ready() {
    std::lock_guard<std::mutex> lock(mu_);
    return ready_;
  }
  // milliseconds from the construction of this object to its first page-locked buffer: the HIP runtime's start-up (+ ~5 ms of page-locking)
  double first_ready_ms() {
    std::lock_guard<std::mutex> lock(mu_);
    return first_ready_ms_;
  }

 private:
  std::size_t floats_;
  int n_;
  std::thread worker_;
  std::mutex mu_;
  std::condition_variable cv_;
  float* buf_[4] = {nullptr, nullptr, nullptr, nullptr};
  int ready_ = 0;
  int failed_rc_ = KMC_OK;
  std::chrono::steady_clock::time_point t_created_ = std::chrono::steady_clock::now();
  double first_ready_ms_ = 0;
};

// points of the largest batch when frames [first, last) are cut into batches of max_batch_frames
std::size_t MaxBatchPoints(std::vector<std::uint64_t> const& frame_points, std::size_t first, std::size_t last, std::size_t max_batch_frames) {
  std::size_t worst = 0;
  for (std::size_t b0 = first; b0 < last; b0 += max_batch_frames) {
    std::size_t sum = 0;
    for (std::size_t i = b0; i < std::min(b0 + max_batch_frames, last); ++i) sum += static_cast<std::size_t>(frame_points[i]);
    worst = std::max(worst, sum);
  }
  return worst;
}

struct RunInputs {
  Path velodyne, out_dir;
  std::vector<Time> const* t_start;
  std::vector<Time> const* t_mid;
  std::vector<Time> const* t_end;
  std::vector<Oxts> const* oxts;
  std::vector<std::uint64_t> const* frame_points;  // points of every frame of the run (from the file sizes)
  std::size_t max_batch_frames;
  bool three_knots, timing;
};
std::mutex g_cout_mu;

void DeskewFrameRange(RunInputs const& in, std::size_t const first, std::size_t const last, int const worker, PinnedSets* prepared = nullptr,
                      std::function<void()> const& adopt_early_context = nullptr) {
  if (first >= last) return;
  struct Pinned {
    kmc_ctx* ctx = nullptr;  // set: `p` came from kmc_hip_host_alloc and is freed here; null: `p` belongs to a PinnedSets
    float* p = nullptr;
    ~Pinned() { if (p && ctx) kmc_hip_host_free(ctx, p); }
    void alloc(kmc_ctx* c, std::size_t n_floats) {
      void* q = nullptr;
      int const rc = kmc_hip_host_alloc(c, n_floats * sizeof(float), &q);
      if (rc != KMC_OK) detail::throw_status(rc, "kmc_hip_host_alloc", c);
      ctx = c;
      p = static_cast<float*>(q);
    }
  };
  struct BatchPlan {
    std::size_t first = 0, last = 0;  // frames [first, last)
    std::vector<std::uint64_t> offsets;
    std::vector<Path> files;
  };
  enum class State { kUnallocated, kFree, kReady, kDone };
  struct BufferSet {
    Pinned in, out;
    State state = State::kUnallocated;
    std::vector<hip::FramePoses> frames;
    std::vector<hip::FrameTrajectory> trajectories;  // KMC_RUN_KNOTS=3
  };
  using clk = std::chrono::steady_clock;
  auto const secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
  auto const& t_start = *in.t_start;
  auto const& t_mid = *in.t_mid;
  auto const& t_end = *in.t_end;
  auto const& oxts = *in.oxts;

  // plan: file sizes decide the batch boundaries and the buffer size
  std::vector<BatchPlan> plans;
  std::size_t max_points = 0;
  for (std::size_t b0 = first; b0 < last; b0 += in.max_batch_frames) {
    BatchPlan plan;
    plan.first = b0;
    plan.last = std::min(b0 + in.max_batch_frames, last);
    plan.offsets.push_back(0);
    for (std::size_t i = plan.first; i < plan.last; ++i) {
      plan.files.push_back(in.velodyne / Path("data/" + IdToZeroPaddedString(i) + ".bin"));
      plan.offsets.push_back(plan.offsets.back() + (*in.frame_points)[i]);
    }
    max_points = std::max<std::size_t>(max_points, static_cast<std::size_t>(plan.offsets.back()));
    plans.push_back(std::move(plan));
  }

  // The page-locked buffers come from a helper thread (PinnedSets): the caller's, started before it parsed the text files, or one
  // started here (a worker of the multi-device driver).  This thread meanwhile creates its device context.
  auto const t_ctx0 = clk::now();
  std::unique_ptr<PinnedSets> own_sets;
  if (!prepared) {
    own_sets = std::make_unique<PinnedSets>(4 * max_points + 16, plans.size() > 1 ? 4 : 2, hip::GetDevice());
    prepared = own_sets.get();
  }
  // The device context is only needed by the first GPU round trip: it is taken (adopted from the caller's helper thread, or created)
  // AFTER the reader has been started, so that the first batches are read while the context is still being made.
  kmc_ctx* ctx = nullptr;
  BufferSet sets[2];
  std::mutex mu;
  std::condition_variable cv;
  std::exception_ptr failure;
  bool fallback_buffers = false;  // the pool declined: this thread has allocated every buffer through the context (under `mu`)
  double t_read = 0, t_gpu = 0, t_write = 0, t_first_ready = 0, t_ctx = 0;
  // buffer `which` (0 = in, 1 = out) of set `k`: from the page-locking helper, or -- the pool is switched off -- from this thread, which
  // allocates all of them through the device context once it has one
  auto const buffer_of = [&](BufferSet& set, int k, int which) -> float* {
    Pinned& slot = which == 0 ? set.in : set.out;
    {
      std::lock_guard<std::mutex> lock(mu);
      if (slot.p) return slot.p;
    }
    float* const q = prepared->wait(2 * k + which);
    std::unique_lock<std::mutex> lock(mu);
    if (q) {
      slot.p = q;  // borrowed: PinnedSets frees it
      return q;
    }
    if (prepared->ready() > 0) {
      // The helper page-locked some buffers and then failed (a memlock limit, a large batch): nobody is going to make the missing one --
      // the GPU thread's fallback below only covers a pool that declined from the start.  Waiting would wait forever; the
      // run ends with the error instead.
      if (!failure)
        failure = std::make_exception_ptr(std::runtime_error(std::string("MotionCompensateRun: page-locking buffer ") + std::to_string(2 * k + which) +
                                                             " of the run failed: " + kmc_status_string(prepared->failed_rc())));
      lock.unlock();
      cv.notify_all();
      return nullptr;
    }
    cv.wait(lock, [&] { return fallback_buffers || failure; });
    return slot.p;  // (nullptr only with `failure` set: the caller's next wait_for ends the thread)
  };
  sets[0].state = State::kFree;
  sets[1].state = State::kFree;

  auto const wait_for = [&](BufferSet& set, State wanted) {
    std::unique_lock<std::mutex> lock(mu);
    cv.wait(lock, [&] { return set.state == wanted || failure; });
    return !failure;
  };
  auto const publish = [&](BufferSet& set, State next) {
    { std::lock_guard<std::mutex> lock(mu); set.state = next; }
    cv.notify_all();
  };
  auto const fail = [&](std::exception_ptr e) {
    { std::lock_guard<std::mutex> lock(mu); if (!failure) failure = e; }
    cv.notify_all();
  };

  std::thread reader([&] {
    try {
      for (std::size_t k = 0; k < plans.size(); ++k) {
        BufferSet& set = sets[k % 2];
        BatchPlan const& plan = plans[k];
        if (!wait_for(set, State::kFree)) return;
        float* const in_p = buffer_of(set, static_cast<int>(k % 2), 0);  // (the first lap may wait for the page-locking helper)
        if (!in_p) return;  // only after a failure elsewhere
        auto const t0 = clk::now();
        set.frames.clear();
        set.trajectories.clear();
        for (std::size_t j = 0; j < plan.files.size(); ++j) {
          std::size_t const i = plan.first + j;
          std::ifstream is{plan.files[j], std::ios::in | std::ios::binary};
          if (!is.is_open()) throw std::runtime_error("Unable to open requested KITTI pointcloud binary file: " + plan.files[j].string());
          auto const want = static_cast<std::streamsize>((plan.offsets[j + 1] - plan.offsets[j]) * 16);
          is.read(reinterpret_cast<char*>(in_p + 4 * plan.offsets[j]), want);
          // a file that shrank or failed since the planning pass would leave the previous batch's bytes in this slot
          if (is.gcount() != want || is.bad())
            throw std::runtime_error("Opened KITTI pointcloud binary file is incorrectly formatted: " + plan.files[j].string() +
                                     " (changed while the run was in progress)");
          if (in.three_knots) {
            hip::FrameTrajectory ft;
            ft.trajectory.times = {oxts[i - 1].stamp, oxts[i].stamp, oxts[i + 1].stamp};
            ft.trajectory.poses = {OxtsToPose(oxts[i - 1]), OxtsToPose(oxts[i]), OxtsToPose(oxts[i + 1])};
            ft.stamp_start = t_start[i];
            ft.stamp_end = t_end[i];
            ft.requested_time = t_mid[i];
            set.trajectories.push_back(std::move(ft));
            continue;
          }
          hip::FramePoses fp;  // MakeFrame (data_io.cpp:253-269) + requested_time = stamp_middle (handlers.cpp:59)
          fp.T_start = trajectory_interpolation::InterpolateTrajectory(oxts[i - 1], oxts[i], t_start[i]);
          fp.T_end = trajectory_interpolation::InterpolateTrajectory(oxts[i], oxts[i + 1], t_end[i]);
          fp.stamp_start = t_start[i];
          fp.stamp_end = t_end[i];
          fp.requested_time = t_mid[i];
          set.frames.push_back(fp);
        }
        t_read += secs(t0, clk::now());
        if (k == 0) t_first_ready = secs(t_ctx0, clk::now());
        publish(set, State::kReady);
      }
    } catch (...) {
      fail(std::current_exception());
    }
  });
  auto const write_batches = [&] {
    try {
      for (std::size_t k = 0; k < plans.size(); ++k) {
        BufferSet& set = sets[k % 2];
        BatchPlan const& plan = plans[k];
        if (!wait_for(set, State::kDone)) return;
        auto const t0 = clk::now();
        constexpr std::size_t kWriters = 4;  // this thread + three helpers: the files of a batch are dealt out round-robin
        auto const write_files = [&](std::size_t j0) {
          for (std::size_t j = j0; j < plan.files.size(); j += kWriters)
            WriteRaw(in.out_dir, plan.first + j, set.out.p + 4 * plan.offsets[j], static_cast<std::size_t>(plan.offsets[j + 1] - plan.offsets[j]));
        };
        std::exception_ptr helper_error[kWriters];
        std::thread helpers[kWriters];
        for (std::size_t h = 1; h < kWriters; ++h) {
          try {
            helpers[h] = std::thread([&, h] {
              try {
                write_files(h);
              } catch (...) {
                helper_error[h] = std::current_exception();
              }
            });
          } catch (std::system_error const&) {  // no helper thread to be had: this thread writes that share too
            try {
              write_files(h);
            } catch (...) {
              helper_error[h] = std::current_exception();
            }
          }
        }
        std::exception_ptr own_error;
        try {
          write_files(0);
        } catch (...) {
          own_error = std::current_exception();
        }
        for (std::size_t h = 1; h < kWriters; ++h)
          if (helpers[h].joinable()) helpers[h].join();
        if (own_error) std::rethrow_exception(own_error);
        for (std::size_t h = 1; h < kWriters; ++h)
          if (helper_error[h]) std::rethrow_exception(helper_error[h]);
        {
          std::lock_guard<std::mutex> lock(g_cout_mu);
          for (std::size_t j = 0; j < plan.files.size(); ++j)
            std::cout << "Motion compensated pointcloud number: " << plan.first + j << std::endl;  // handlers.cpp:63 (ascending, like the reference's loop)
        }
        t_write += secs(t0, clk::now());
        publish(set, State::kFree);
      }
    } catch (...) {
      fail(std::current_exception());
    }
  };
  std::thread writer;
  try {
    writer = std::thread(write_batches);
  } catch (...) {  // no second thread to be had: release the reader before reporting it
    fail(std::current_exception());
    reader.join();
    throw;
  }
  try {
    if (adopt_early_context) adopt_early_context();  // joins the caller's helper thread; its context becomes this thread's
    ctx = detail::thread_context();
    t_ctx = secs(t_ctx0, clk::now());
    if (in.timing) std::cerr << "kmc run timing: device context " << t_ctx * 1e3 << " ms after the range started (the reader is already at work)\n";
    if (prepared->wait(0) == nullptr) {  // KMC_HOST_POOL=0: page-locked buffers through the context, like before round 5
      std::lock_guard<std::mutex> lock(mu);
      for (auto& set : sets) {
        set.in.alloc(ctx, 4 * max_points + 16);
        set.out.alloc(ctx, 4 * max_points + 16);
      }
      fallback_buffers = true;
      cv.notify_all();
    }
    for (std::size_t k = 0; k < plans.size(); ++k) {
      BufferSet& set = sets[k % 2];
      if (!wait_for(set, State::kReady)) break;
      if (!buffer_of(set, static_cast<int>(k % 2), 1)) break;
      auto const t0 = clk::now();
      if (in.three_knots) hip::MotionCompensateKittiClouds(set.in.p, plans[k].offsets, set.trajectories, set.out.p);
      else hip::MotionCompensateKittiClouds(set.in.p, plans[k].offsets, set.frames, set.out.p);
      t_gpu += secs(t0, clk::now());
      publish(set, State::kDone);
    }
  } catch (...) {
    fail(std::current_exception());
  }
  reader.join();
  writer.join();
  if (failure) std::rethrow_exception(failure);
  if (in.timing) {
    std::lock_guard<std::mutex> lock(g_cout_mu);
    std::cerr << "kmc run timing, worker " << worker << " (device " << hip::GetDevice() << ", frames [" << first << ", " << last
              << ")), busy seconds per stage: context " << t_ctx << "  read " << t_read << "  gpu round trip " << t_gpu << "  write "
              << t_write << "  | wall seconds since the range started: first batch read " << t_first_ready << "  all written " << secs(t_ctx0, clk::now()) << "\n";
  }
}
}  // namespace

void MotionCompensateRun(Path const run_folder) {  // :41-65
  Path const velodyne{run_folder / Path{"velodyne_points"}};
  std::size_t const n_frames{NumberOfFilesInDirectory(velodyne / Path("data"))};
  Path const out_dir{velodyne / Path("data_motion_compensated")};
  if (!fs::is_directory(out_dir) || !fs::exists(out_dir)) fs::create_directory(out_dir);
  if (n_frames == 0) return;
  auto const t_entry = std::chrono::steady_clock::now();
  std::size_t const max_batch_frames = [] {  // page-locking costs ~0.3 ms/MiB, so modest batches win for one-off runs (tools/time_run_cli.sh: 8 beats 16 by ~20 ms per 216-frame run)
    char const* e = std::getenv("KMC_RUN_BATCH_FRAMES");
    long const v = e ? std::atol(e) : 0;
    return static_cast<std::size_t>(v > 0 ? std::min(v, 4096L) : 8L);
  }();

  std::vector<int> const devices = hip::GetRunDevices();
  bool const one_device_here = n_frames >= 3 && std::min<std::size_t>(devices.size(), n_frames - 2) == 1 && devices[0] == hip::GetDevice();
  // One device, the caller's thread: the device context (12 ms for a process's second context, ~40 ms for its first: the HIP runtime's
  // start-up, streams, the code objects' load) is made on a helper thread from HERE on, while this thread copies the two uncompensated
  // frames, sizes the run and parses the text files; this thread adopts it once they are parsed
  struct EarlyContext {
    std::thread worker;
    kmc_ctx* ctx = nullptr;
    int rc = KMC_OK;
    ~EarlyContext() {  // (an exception on the way: the helper is joined, a context nobody adopted is destroyed)
      if (worker.joinable()) worker.join();
      if (ctx) kmc_hip_destroy(ctx);
    }
  } early;
  if (one_device_here && !detail::thread_has_context()) {
    int const device = hip::GetDevice();
    try {
      early.worker = std::thread([&early, device] { early.rc = kmc_hip_create(&early.ctx, device); });
    } catch (std::system_error const&) {  // no helper to be had: the context is created where it always was
    }
  }
  // handlers.cpp:46 does this first, before it looks at any other frame: a run with a missing middle frame still gets its two
  // uncompensated frames copied, like the reference's
  CopyOverUncompensatedFirstAndLastFrame(run_folder);
  if (n_frames < 3) return;
  double const copied_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_entry).count();

  // points per frame, from the file sizes: they decide the batch boundaries, the buffer sizes and the split across devices
  std::vector<std::uint64_t> frame_points(n_frames, 0);
  for (std::size_t i = 1; i + 1 < n_frames; ++i) {
    Path const file{velodyne / Path("data/" + IdToZeroPaddedString(i) + ".bin")};
    std::error_code ec;
    auto const bytes = fs::file_size(file, ec);
    if (ec) throw std::runtime_error("Unable to open requested KITTI pointcloud binary file: " + file.string());
    if (bytes % 4 != 0) throw std::runtime_error("Opened KITTI pointcloud binary file is incorrectly formatted: " + file.string());
    frame_points[i] = bytes / 16;
  }
  double const sizes_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_entry).count();
  // ... and the buffers are page-locked on a second helper (~0.3 ms per MiB) while this thread parses the text files and the reader
  // already fills the first buffer
  std::unique_ptr<PinnedSets> prepared;
  if (one_device_here) {
    std::size_t const most = MaxBatchPoints(frame_points, 1, n_frames - 1, max_batch_frames);
    prepared = std::make_unique<PinnedSets>(4 * most + 16, n_frames - 2 > max_batch_frames ? 4 : 2, hip::GetDevice());
  }

  // every text file is parsed once per run
  auto const t_start = LoadAllTimeStamps(velodyne / Path("timestamps_start.txt"));
  auto const t_mid = LoadAllTimeStamps(velodyne / Path("timestamps.txt"));
  auto const t_end = LoadAllTimeStamps(velodyne / Path("timestamps_end.txt"));
  auto const t_oxts = LoadAllTimeStamps(run_folder / Path("oxts/timestamps.txt"));
  if (t_start.size() < n_frames || t_mid.size() < n_frames || t_end.size() < n_frames || t_oxts.size() < n_frames)
    throw std::runtime_error("timestamp files are shorter than the number of velodyne frames in " + run_folder.string());
  double const stamps_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_entry).count();
  std::vector<Oxts> oxts(n_frames);
  for (std::size_t i = 0; i < n_frames; ++i) oxts[i] = LoadOxtsWithStamp(run_folder, i, t_oxts[i]);
  double const parsed_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_entry).count();

  RunInputs in;
  in.velodyne = velodyne;
  in.out_dir = out_dir;
  in.t_start = &t_start;
  in.t_mid = &t_mid;
  in.t_end = &t_end;
  in.oxts = &oxts;
  in.frame_points = &frame_points;
  in.max_batch_frames = max_batch_frames;
  in.timing = [] { char const* e = std::getenv("KMC_RUN_TIMING"); return e && e[0] == '1'; }();
  // KMC_RUN_KNOTS=3: interpolate along the piecewise geodesic through the three OXTS poses around the frame, used as they
  // are, instead of first reducing them to the two scan-end poses like MakeFrame does (data_io.cpp:253-269).
  in.three_knots = [] { char const* e = std::getenv("KMC_RUN_KNOTS"); return e && e[0] == '3'; }();

  // The frames of a run are independent (handlers.cpp:55-64 reads nothing it wrote): frames 1 .. n-2 are cut into one
  // CONTIGUOUS range per device, balanced on points (kmc_frame_ranges_balanced -- the split the per-rank launch uses), and
  // every range runs its own read / deskew / write pipeline on its own device context.  KMC_DEVICES=0,1,... or
  // kmc::hip::SetRunDevices select the devices (an id may repeat: two contexts on one GPU); default = the calling thread's.
  std::uint32_t const n_parts = static_cast<std::uint32_t>(std::min<std::size_t>(devices.size(), n_frames - 2));
  std::vector<std::uint32_t> bounds(n_parts + 1);
  {
    int const rc = kmc_frame_ranges_balanced(frame_points.data() + 1, static_cast<std::uint32_t>(n_frames - 2), n_parts, bounds.data());
    if (rc != KMC_OK) detail::throw_status(rc, "kmc_frame_ranges_balanced");
  }
  if (in.timing)
    std::cerr << "kmc run timing: first / last frame copied " << copied_ms << " ms, file sizes " << sizes_ms << " ms, time stamp files " << stamps_ms
              << " ms, text files parsed " << parsed_ms << " ms after entry\n";
  if (n_parts == 1 && devices[0] == hip::GetDevice()) {  // the common case: no extra worker, the caller's own context
    auto const adopt = [&early] {
      if (!early.worker.joinable()) return;
      early.worker.join();
      if (early.rc == KMC_OK && early.ctx) {
        detail::adopt_thread_context(early.ctx, hip::GetDevice());
        early.ctx = nullptr;
      }  // (a failed creation is reported by thread_context(), which tries again)
    };
    DeskewFrameRange(in, 1, n_frames - 1, 0, prepared.get(), adopt);
    if (in.timing)
      std::cerr << "kmc run timing: MotionCompensateRun returns " << std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_entry).count()
                << " ms after entry; HIP runtime up and first buffer page-locked " << (prepared ? prepared->first_ready_ms() : 0.0) << " ms after entry\n";
    return;
  }
  std::vector<std::thread> workers;
  std::vector<std::exception_ptr> errors(n_parts);
  workers.reserve(n_parts);
  std::exception_ptr spawn_error;
  for (std::uint32_t r = 0; r < n_parts && !spawn_error; ++r) {
    try {
      workers.emplace_back([&, r] {
        try {
          hip::SetDevice(devices[r]);  // this worker thread's context lives on its device
          (void)hip::BindThreadNearDevice();  // ... and the worker (a thread of the library's own) runs on that device's side of the machine
          DeskewFrameRange(in, 1 + bounds[r], 1 + bounds[r + 1], static_cast<int>(r));
        } catch (...) {
          errors[r] = std::current_exception();
        }
      });
    } catch (...) {  // the system refused a thread: the workers already running finish their ranges, then the caller hears about it
      spawn_error = std::current_exception();
    }
  }
  for (auto& w : workers) w.join();
  if (spawn_error) std::rethrow_exception(spawn_error);
  for (auto const& e : errors)
    if (e) std::rethrow_exception(e);
}

}  // namespace kmc

// experiment: where does the time of the literal MotionCompensateFrame(Frame, Time) host route go?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "kmc_hip.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
#define KC(x) do { int r_ = (x); if (r_ != KMC_OK) { std::fprintf(stderr, "%s: %s\n", #x, kmc_status_string(r_)); std::exit(1); } } while (0)
using clk = std::chrono::steady_clock;
template <typename F> double best_us(F&& f, int reps = 20) {
  double best = 1e30;
  for (int r = 0; r < reps; ++r) { auto t0 = clk::now(); f(); best = std::min(best, std::chrono::duration<double>(clk::now() - t0).count() * 1e6); }
  return best;
}
int main(int argc, char** argv) {
  const size_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 123397;
  kmc_ctx* ctx; KC(kmc_hip_create(&ctx, 0));
  std::vector<double> cloud(4 * n), stamps(n), out(4 * n);
  for (size_t i = 0; i < n; ++i) { cloud[i] = 10 + 1e-3 * i; cloud[n + i] = 5 - 1e-3 * i; cloud[2 * n + i] = 1; cloud[3 * n + i] = 1; stamps[i] = 100.0 + 0.1 * i / n; }
  double* d; CK(hipMalloc((void**)&d, 9 * n * 8));
  double* pin_in; double* pin_out; CK(hipHostMalloc((void**)&pin_in, 5 * n * 8)); CK(hipHostMalloc((void**)&pin_out, 4 * n * 8));
  std::memcpy(pin_in, cloud.data(), 4 * n * 8);
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  auto sync = [&] { CK(hipStreamSynchronize(s)); };
  std::printf("n=%zu\n", n);
  std::printf("H2D pageable 4 cols (%.2f MB): %.1f us\n", 4 * n * 8 / 1e6, best_us([&] { CK(hipMemcpyAsync(d, cloud.data(), 4 * n * 8, hipMemcpyHostToDevice, s)); sync(); }));
  std::printf("H2D pageable 3 cols: %.1f us\n", best_us([&] { CK(hipMemcpyAsync(d, cloud.data(), 3 * n * 8, hipMemcpyHostToDevice, s)); sync(); }));
  std::printf("H2D pageable stamps (%.2f MB): %.1f us\n", n * 8 / 1e6, best_us([&] { CK(hipMemcpyAsync(d + 4 * n, stamps.data(), n * 8, hipMemcpyHostToDevice, s)); sync(); }));
  std::printf("H2D pinned 4 cols: %.1f us\n", best_us([&] { CK(hipMemcpyAsync(d, pin_in, 4 * n * 8, hipMemcpyHostToDevice, s)); sync(); }));
  std::printf("D2H pageable 4 cols: %.1f us\n", best_us([&] { CK(hipMemcpyAsync(out.data(), d + 5 * n, 4 * n * 8, hipMemcpyDeviceToHost, s)); sync(); }));
  std::printf("D2H pageable 3 cols: %.1f us\n", best_us([&] { CK(hipMemcpyAsync(out.data(), d + 5 * n, 3 * n * 8, hipMemcpyDeviceToHost, s)); sync(); }));
  std::printf("D2H pinned 4 cols: %.1f us\n", best_us([&] { CK(hipMemcpyAsync(pin_out, d + 5 * n, 4 * n * 8, hipMemcpyDeviceToHost, s)); sync(); }));
  std::printf("host memcpy 4 cols pageable->pinned: %.1f us\n", best_us([&] { std::memcpy(pin_in, cloud.data(), 4 * n * 8); }));
  std::printf("host scan w==1 (%zu doubles): %.1f us\n", n, best_us([&] { size_t bad = 0; for (size_t i = 0; i < n; ++i) bad += cloud[3 * n + i] != 1.0; if (bad) std::abort(); }));
  std::printf("host fill ones: %.1f us\n", best_us([&] { for (size_t i = 0; i < n; ++i) out[3 * n + i] = 1.0; }));
  // two-direction overlap with pinned buffers on two streams
  hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  std::printf("pinned H2D 4 cols || pinned D2H 4 cols: %.1f us\n", best_us([&] { CK(hipMemcpyAsync(d, pin_in, 4 * n * 8, hipMemcpyHostToDevice, s)); CK(hipMemcpyAsync(pin_out, d + 5 * n, 4 * n * 8, hipMemcpyDeviceToHost, s2)); sync(); CK(hipStreamSynchronize(s2)); }));
  kmc_frame_params p; const double tw[6] = {1.3, 0.05, -0.02, 0.002, -0.004, 0.03}; for (int i = 0; i < 6; ++i) p.twist[i] = tw[i]; p.x_req = 0.5;
  std::printf("kmc_hip_deskew_f64cols HOST (whole call): %.1f us\n", best_us([&] {
    KC(kmc_hip_deskew_f64cols(ctx, cloud.data(), cloud.data() + n, cloud.data() + 2 * n, cloud.data() + 3 * n, stamps.data(), n, 100.0, 100.1, &p, out.data(), out.data() + n, out.data() + 2 * n, out.data() + 3 * n, KMC_MEM_HOST, nullptr)); }));
  std::printf("kmc_hip_deskew_f64cols HOST, w = NULL: %.1f us\n", best_us([&] {
    KC(kmc_hip_deskew_f64cols(ctx, cloud.data(), cloud.data() + n, cloud.data() + 2 * n, nullptr, stamps.data(), n, 100.0, 100.1, &p, out.data(), out.data() + n, out.data() + 2 * n, nullptr, KMC_MEM_HOST, nullptr)); }));
  return 0;
}

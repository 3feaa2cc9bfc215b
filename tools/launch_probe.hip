// launch_probe.hip -- what does ONE kernel launch cost the host, by launch API?  
// kmc_hip_deskew_f32 on device-resident KITTI frames is host-bound at 3.8 us per call; most of it is the runtime's launch path.  The
// stand-in kernel takes the product kernel's argument block (three words of pointers / sizes, a 64-byte f32 record, a 128-byte f64
// record: ~230 bytes) and moves a KITTI-sized frame (1929 one-wave tiles).  N launches back to back on one stream, host wall clock per
// launch call and for the whole train (the device's side):
//   ggl            hipLaunchKernelGGL (what the library does today)
//   ggl_anyorder   hipExtLaunchKernelGGL(..., hipExtAnyOrderLaunch)
//   launchkernel   hipLaunchKernel(symbol, void** args)
//   module         hipModuleLaunchKernel(hipFunction_t from hipGetFuncBySymbol, HIP_LAUNCH_PARAM_BUFFER_POINTER: pre-packed block)
//   extmodule      hipExtModuleLaunchKernel(the same, + the any-order flag)
//   graph_update   a one-node hipGraphExec: hipGraphExecKernelNodeSetParams + hipGraphLaunch per call
//   graph_16       sixteen kernel nodes per graph launch, no updates (an upper bound for graph replay)
//   launch_probe [launches=20000] [points=123397]   -> one JSON object
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#define CHECK(x)                                                                              \
  do {                                                                                        \
    hipError_t e_ = (x);                                                                      \
    if (e_ != hipSuccess) {                                                                   \
      std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      std::exit(2);                                                                           \
    }                                                                                         \
  } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
struct Rec32 { float v[16]; };
struct Rec64 { double v[16]; };
__global__ __launch_bounds__(64) void k_frame(const v4f* __restrict__ in, v4f* __restrict__ out, uint64_t n, Rec32 f, uint32_t head, uint64_t tile_base, Rec64 d) {
  const uint64_t i = (tile_base + blockIdx.x) * 64 + threadIdx.x;
  if (i < n) {
    v4f p = __builtin_nontemporal_load(in + i);
    p.x = __builtin_fmaf(p.x, f.v[0], f.v[1] + (float)d.v[3]);
    __builtin_nontemporal_store(p, out + i);
  }
}
// Does the runtime take kernel-argument blocks beyond 4 KiB?  (the frame-list kernel carries 16 frames' records in 3.6 KiB; 32 or 64
// frames per launch would halve / quarter the launches of a drive)
template <int BYTES> struct Big { uint32_t w[BYTES / 4]; };
template <int BYTES>
__global__ __launch_bounds__(64) void k_big(uint32_t* out, Big<BYTES> b) {
  const uint32_t __attribute__((address_space(4)))* p = (const uint32_t __attribute__((address_space(4)))*)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + 8);
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = p[BYTES / 4 - 1] + p[0];  // first and last word of the block, through the segment
}
struct Args { const v4f* in; v4f* out; uint64_t n; Rec32 f; uint32_t head; uint64_t tile_base; Rec64 d; };

using clk = std::chrono::steady_clock;
static double us_since(clk::time_point a) { return std::chrono::duration<double, std::micro>(clk::now() - a).count(); }

int main(int argc, char** argv) {
  const int N = argc > 1 ? std::atoi(argv[1]) : 20000;
  const uint64_t n = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 123397ull;
  CHECK(hipSetDevice(0));
  hipStream_t s;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const int kBufs = 8;
  v4f *in[kBufs], *out[kBufs];
  for (int b = 0; b < kBufs; ++b) { CHECK(hipMalloc(&in[b], n * 16)); CHECK(hipMalloc(&out[b], n * 16)); CHECK(hipMemset(in[b], 0, n * 16)); }
  const unsigned grid = (unsigned)((n + 63) / 64);
  Rec32 f; Rec64 d;
  for (int i = 0; i < 16; ++i) { f.v[i] = 1.0f + i; d.v[i] = 0.5 * i; }
  hipFunction_t fn;
  CHECK(hipGetFuncBySymbol(&fn, (const void*)k_frame));
  std::printf("{\"launches\": %d, \"points\": %llu, \"arg_bytes\": %zu", N, (unsigned long long)n, sizeof(Args));
  auto report = [&](const char* name, auto&& one) {
    for (int i = 0; i < 2000; ++i) one(i);  // warm: clocks, code objects, the runtime's pools
    CHECK(hipStreamSynchronize(s));
    auto t0 = clk::now();
    for (int i = 0; i < N; ++i) one(i);
    const double host = us_since(t0) / N;
    CHECK(hipStreamSynchronize(s));
    const double all = us_since(t0) / N;
    std::printf(", \"%s\": {\"host_us_per_launch\": %.3f, \"train_us_per_launch\": %.3f}", name, host, all);
    std::fflush(stdout);
  };
  report("ggl", [&](int i) { hipLaunchKernelGGL(k_frame, dim3(grid), dim3(64), 0, s, in[i % kBufs], out[i % kBufs], n, f, 0u, (uint64_t)0, d); });
  report("ggl_anyorder", [&](int i) { hipExtLaunchKernelGGL(k_frame, dim3(grid), dim3(64), 0, s, nullptr, nullptr, (uint32_t)hipExtAnyOrderLaunch, in[i % kBufs], out[i % kBufs], n, f, 0u, (uint64_t)0, d); });
  report("launchkernel", [&](int i) {
    const v4f* a0 = in[i % kBufs]; v4f* a1 = out[i % kBufs]; uint64_t a2 = n; uint32_t a4 = 0; uint64_t a5 = 0;
    void* args[] = {&a0, &a1, &a2, &f, &a4, &a5, &d};
    (void)hipLaunchKernel((const void*)k_frame, dim3(grid), dim3(64), args, 0, s);
  });
  Args packed;
  std::memset(&packed, 0, sizeof(packed));
  packed.n = n; packed.f = f; packed.d = d;
  size_t packed_size = sizeof(packed);
  report("module", [&](int i) {
    packed.in = in[i % kBufs]; packed.out = out[i % kBufs];
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &packed, HIP_LAUNCH_PARAM_BUFFER_SIZE, &packed_size, HIP_LAUNCH_PARAM_END};
    (void)hipModuleLaunchKernel(fn, grid, 1, 1, 64, 1, 1, 0, s, nullptr, extra);
  });
  report("extmodule_anyorder", [&](int i) {
    packed.in = in[i % kBufs]; packed.out = out[i % kBufs];
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &packed, HIP_LAUNCH_PARAM_BUFFER_SIZE, &packed_size, HIP_LAUNCH_PARAM_END};
    (void)hipExtModuleLaunchKernel(fn, grid * 64, 1, 1, 64, 1, 1, 0, s, nullptr, extra, nullptr, nullptr, (uint32_t)hipExtAnyOrderLaunch);
  });
  CHECK(hipGetLastError());
  {  // one-node graph, parameters updated per call
    hipGraph_t g;
    CHECK(hipGraphCreate(&g, 0));
    const v4f* a0 = in[0]; v4f* a1 = out[0]; uint64_t a2 = n; uint32_t a4 = 0; uint64_t a5 = 0;
    void* args[] = {&a0, &a1, &a2, &f, &a4, &a5, &d};
    hipKernelNodeParams kp;
    std::memset(&kp, 0, sizeof(kp));
    kp.func = (void*)k_frame; kp.gridDim = dim3(grid); kp.blockDim = dim3(64); kp.kernelParams = args;
    hipGraphNode_t node;
    CHECK(hipGraphAddKernelNode(&node, g, nullptr, 0, &kp));
    hipGraphExec_t ge;
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    report("graph_update", [&](int i) {
      a0 = in[i % kBufs]; a1 = out[i % kBufs];
      (void)hipGraphExecKernelNodeSetParams(ge, node, &kp);
      (void)hipGraphLaunch(ge, s);
    });
    CHECK(hipGetLastError());
  }
  {  // sixteen nodes per graph launch, fixed parameters: graph replay's own rate
    hipGraph_t g;
    CHECK(hipGraphCreate(&g, 0));
    hipGraphNode_t prev = nullptr;
    std::vector<const v4f*> a0(16); std::vector<v4f*> a1(16);
    uint64_t a2 = n; uint32_t a4 = 0; uint64_t a5 = 0;
    for (int k = 0; k < 16; ++k) {
      a0[k] = in[k % kBufs]; a1[k] = out[k % kBufs];
      void* args[] = {&a0[k], &a1[k], &a2, &f, &a4, &a5, &d};
      hipKernelNodeParams kp;
      std::memset(&kp, 0, sizeof(kp));
      kp.func = (void*)k_frame; kp.gridDim = dim3(grid); kp.blockDim = dim3(64); kp.kernelParams = args;
      hipGraphNode_t node;
      CHECK(hipGraphAddKernelNode(&node, g, prev ? &prev : nullptr, prev ? 1 : 0, &kp));
      prev = node;
    }
    hipGraphExec_t ge;
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 200; ++i) (void)hipGraphLaunch(ge, s);
    CHECK(hipStreamSynchronize(s));
    auto t0 = clk::now();
    for (int i = 0; i < N / 16; ++i) (void)hipGraphLaunch(ge, s);
    const double host = us_since(t0) / (N / 16 * 16);
    CHECK(hipStreamSynchronize(s));
    const double all = us_since(t0) / (N / 16 * 16);
    std::printf(", \"graph_16\": {\"host_us_per_kernel\": %.3f, \"train_us_per_kernel\": %.3f}", host, all);
  }
  {
    uint32_t* d_out;
    CHECK(hipMalloc(&d_out, 64));
    auto try_big = [&](auto tag, const char* name) {
      constexpr int BYTES = decltype(tag)::value;
      Big<BYTES> b;
      for (int i = 0; i < BYTES / 4; ++i) b.w[i] = 1000u + (uint32_t)i;
      CHECK(hipMemset(d_out, 0, 64));
      hipLaunchKernelGGL(k_big<BYTES>, dim3(1), dim3(64), 0, s, d_out, b);
      const hipError_t le = hipGetLastError();
      const hipError_t se = hipStreamSynchronize(s);
      uint32_t got = 0;
      (void)hipMemcpy(&got, d_out, 4, hipMemcpyDeviceToHost);
      const bool ok = le == hipSuccess && se == hipSuccess && got == 1000u + 1000u + (uint32_t)(BYTES / 4 - 1);
      double host = 0;
      if (ok) {
        auto t0 = clk::now();
        for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k_big<BYTES>, dim3(1), dim3(64), 0, s, d_out, b);
        host = us_since(t0) / 2000;
        (void)hipStreamSynchronize(s);
      }
      std::printf(", \"%s\": {\"accepted_and_correct\": %s, \"launch_error\": \"%s\", \"host_us_per_launch\": %.3f}", name, ok ? "true" : "false", hipGetErrorString(le != hipSuccess ? le : se), host);
      (void)hipGetLastError();
    };
    try_big(std::integral_constant<int, 3584>{}, "kernarg_3584_B");
    try_big(std::integral_constant<int, 4000>{}, "kernarg_4000_B");
    try_big(std::integral_constant<int, 8192>{}, "kernarg_8192_B");
    try_big(std::integral_constant<int, 16384>{}, "kernarg_16384_B");
    try_big(std::integral_constant<int, 65536>{}, "kernarg_65536_B");
  }
  std::printf("}\n");
  return 0;
}

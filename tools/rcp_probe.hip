// rcp_probe.hip -- how good is v_rcp_f64 on gfx950, and how good is it after ONE Newton step?
// trunc_quotients_fast (kmc_device_math.hip.h) decides pixel truncations from a refined reciprocal and keeps a 2^-36 margin; the
// ISA manual only promises 2^29 ulp (2^-23 relative) for the raw instruction.  This measures both over 2^32 denominators spread
// over the camera-depth range and over every binade position:  ./rcp_probe [log2_samples=32]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

__global__ void probe(uint64_t n_per_thread, double* worst_raw, double* worst_one, double* worst_two) {
  const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  uint64_t state = 0x9E3779B97F4A7C15ull * (tid + 1);
  double w0 = 0, w1 = 0, w2 = 0;
  for (uint64_t i = 0; i < n_per_thread; ++i) {
    state ^= state << 13; state ^= state >> 7; state ^= state << 17;  // xorshift64
    // mantissa: all 52 bits random; exponent: 2^-7 .. 2^7 (0.01 m .. 100 m of depth and more)
    const uint64_t mant = state & 0x000FFFFFFFFFFFFFull;
    const int e = (int)((state >> 52) % 15) - 7;
    const double d = __builtin_ldexp(1.0 + (double)mant * 0x1p-52, e);
    const double r0 = __builtin_amdgcn_rcp(d);
    const double r1 = __builtin_fma(__builtin_fma(-d, r0, 1.0), r0, r0);
    const double r2 = __builtin_fma(__builtin_fma(-d, r1, 1.0), r1, r1);
    // relative error of r: |1 - d r| evaluated exactly with one fma
    w0 = fmax(w0, fabs(__builtin_fma(-d, r0, 1.0)));
    w1 = fmax(w1, fabs(__builtin_fma(-d, r1, 1.0)));
    w2 = fmax(w2, fabs(__builtin_fma(-d, r2, 1.0)));
  }
  worst_raw[tid] = w0;
  worst_one[tid] = w1;
  worst_two[tid] = w2;
}

int main(int argc, char** argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 32;
  const unsigned blocks = 2048, threads = 256;
  const uint64_t total = 1ull << lg, per = total / (blocks * (uint64_t)threads);
  double *a, *b, *c;
  const size_t bytes = sizeof(double) * blocks * threads;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&c, bytes) != hipSuccess) return 1;
  hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, per, a, b, c);
  if (hipDeviceSynchronize() != hipSuccess) return 2;
  double* h = (double*)malloc(3 * bytes);
  if (hipMemcpy(h, a, bytes, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(h + blocks * threads, b, bytes, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(h + 2 * blocks * threads, c, bytes, hipMemcpyDeviceToHost) != hipSuccess)
    return 3;
  double w[3] = {0, 0, 0};
  for (int k = 0; k < 3; ++k)
    for (size_t i = 0; i < (size_t)blocks * threads; ++i) w[k] = fmax(w[k], h[k * blocks * threads + i]);
  printf("{\"samples\": %llu, \"rcp_f64_raw_rel_err\": %.3e, \"log2\": %.2f, \"after_one_newton\": %.3e, \"log2_one\": %.2f, \"after_two_newton\": %.3e, \"log2_two\": %.2f}\n",
         (unsigned long long)(per * blocks * threads), w[0], log2(w[0]), w[1], log2(w[1]), w[2], w[2] > 0 ? log2(w[2]) : -1074.0);
  return 0;
}

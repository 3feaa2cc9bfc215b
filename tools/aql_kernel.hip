// aql_kernel.hip -- the stand-in kernel of tools/aql_probe.hip (the product kernel's argument block: 232 bytes, no hidden arguments),
// compiled to a raw gfx950 code object (hipcc --genco --no-gpu-bundle-output) that the probe loads through the HSA loader.
#include <hip/hip_runtime.h>
typedef float v4f __attribute__((ext_vector_type(4)));
struct Rec32 { float v[16]; };
struct Rec64 { double v[16]; };
extern "C" __global__ __launch_bounds__(64) void k_frame(const v4f* __restrict__ in, v4f* __restrict__ out, uint64_t n, Rec32 f, uint32_t head, uint64_t tile_base, Rec64 d) {
  const uint64_t i = (tile_base + blockIdx.x) * 64 + threadIdx.x;
  if (i < n) {
    v4f p = __builtin_nontemporal_load(in + i);
    p.x = __builtin_fmaf(p.x, f.v[0], f.v[1] + (float)d.v[3]);
    __builtin_nontemporal_store(p, out + i);
  }
}

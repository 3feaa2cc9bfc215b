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

// the same frame with a FIXED number of waves that walk the tiles (tile t, t + grid, ...), the next tile's load in flight while the current
// one is finished: a frame then costs grid argument-block fetches instead of n / 64 -- what matters when the block lives in HOST memory
// and every wave's scalar loads cross the link (aql_probe ... host walk=<waves>)
// (the number of waves arrives in `tile_base`: gridDim would bring the hidden arguments in, which an AQL packet of the probe's does not fill)
extern "C" __global__ __launch_bounds__(64) void k_frame_walk(const v4f* __restrict__ in, v4f* __restrict__ out, uint64_t n, Rec32 f, uint32_t head, uint64_t tile_base, Rec64 d) {
  const uint64_t waves = tile_base;
  const uint64_t n_tiles = (n + 63) / 64;
  uint64_t t = blockIdx.x;
  if (t >= n_tiles) return;
  const uint64_t last = n - 1;
  const float add = f.v[1] + (float)d.v[3];
  v4f cur = __builtin_nontemporal_load(in + (t * 64 + threadIdx.x <= last ? t * 64 + threadIdx.x : last));
  while (true) {
    const uint64_t next = t + waves;
    const bool more = next < n_tiles;
    v4f nxt = cur;
    if (more) nxt = __builtin_nontemporal_load(in + (next * 64 + threadIdx.x <= last ? next * 64 + threadIdx.x : last));
    const uint64_t i = t * 64 + threadIdx.x;
    cur.x = __builtin_fmaf(cur.x, f.v[0], add);
    if (i < n) __builtin_nontemporal_store(cur, out + i);
    if (!more) break;
    cur = nxt;
    t = next;
  }
}

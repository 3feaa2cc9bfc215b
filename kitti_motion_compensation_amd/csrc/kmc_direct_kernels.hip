// kmc_direct_kernels.hip -- the single-frame kernel once more, as a RAW gfx950 code object (hipcc --genco --no-gpu-bundle-output) that
// kmc_capi_direct.hip loads through the HSA loader and dispatches with AQL packets of its own (the direct queue, round 5).
// Same tile body (frame_tile<TIER>), same argument layout, same bits as kmc_dev::deskew_frame_f32<TIER> in libkmc_hip.so's HIP code
// object -- the direct queue's self-test compares the two on the device before the queue is used.  extern "C": the loader finds the
// kernels by their plain names ("kmc_direct_frame_t0.kd" ...).
#include "kmc_kernels.hip.h"

using namespace kmc_dev;

#define KMC_DIRECT_FRAME_KERNEL(TIER)                                                                                                      \
  extern "C" __global__ __launch_bounds__(kTile) __attribute__((amdgpu_waves_per_eu(8, 8))) void kmc_direct_frame_t##TIER(               \
      const v4f* __restrict__ in, v4f* __restrict__ out, uint64_t n, FrameRec f, uint32_t head, uint64_t tile_base, FrameRecD d) {        \
    struct ArgLayout { const v4f* in; v4f* out; uint64_t n; FrameRec f; uint32_t head; uint64_t tile_base; FrameRecD d; };                \
    const cdouble_p d_rec = (cdouble_p)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(ArgLayout, d)); \
    const uint64_t tile = tile_base + blockIdx.x;                                                                                          \
    if (tile * kTile >= n) return;                                                                                                         \
    frame_tile<TIER>(in, out, n, f, head, d_rec, tile);                                                                                    \
  }
KMC_DIRECT_FRAME_KERNEL(0)
KMC_DIRECT_FRAME_KERNEL(1)
KMC_DIRECT_FRAME_KERNEL(2)
KMC_DIRECT_FRAME_KERNEL(3)

// The N-knot frame with its segment records in the argument block (north_star's three bracketing poses; deskew_traj_f32<TIER, false, true>
// in the HIP code object): same tile body (traj_tile<TIER, false>), the records read through the kernel-argument segment.  The argument
// list is the direct queue's own (kmc_capi_direct.hip, TrajDirectArgs): no table pointers, no index output.
#define KMC_DIRECT_TRAJ_KERNEL(TIER)                                                                                                       \
  extern "C" __global__ __launch_bounds__(kTile) __attribute__((amdgpu_waves_per_eu(8, 8))) void kmc_direct_traj_t##TIER(                \
      const v4f* __restrict__ in, v4f* __restrict__ out, uint64_t n, uint32_t n_seg, uint32_t head, uint64_t tile_base, TrajInline inl) { \
    struct ArgLayout { const v4f* in; v4f* out; uint64_t n; uint32_t n_seg; uint32_t head; uint64_t tile_base; TrajInline inl; };          \
    const auto kernarg = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();                            \
    const seg_cp segs_c = (seg_cp)(kernarg + offsetof(ArgLayout, inl) + offsetof(TrajInline, s));                                          \
    const TrajSegD* segs64 = (const TrajSegD*)(const char*)(kernarg + offsetof(ArgLayout, inl) + offsetof(TrajInline, d));                 \
    traj_tile<TIER, false>(in, out, n, segs_c, n_seg, nullptr, head, segs64, tile_base + blockIdx.x);                                      \
  }
KMC_DIRECT_TRAJ_KERNEL(0)
KMC_DIRECT_TRAJ_KERNEL(1)
KMC_DIRECT_TRAJ_KERNEL(2)
KMC_DIRECT_TRAJ_KERNEL(3)

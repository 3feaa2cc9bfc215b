// kmc_capi_synth.hip -- synthetic workload (measurement infrastructure; BASELINE.json configs 2-5 have no shippable data).
#include "kmc_internal.hip.h"

extern "C" {

int kmc_hip_synth_points(kmc_ctx* c, float* xyzi_out_device, uint64_t n, uint64_t seed) {
  if (!c || (n && !xyzi_out_device)) return KMC_ERR_INVALID_ARG;
  if (n == 0) return KMC_OK;
  KMC_ENTER(c);
  const int grid = (int)std::min<uint64_t>((n + kBlock - 1) / kBlock, (uint64_t)c->prop.multiProcessorCount * 32);  // grid-stride generator
  hipLaunchKernelGGL(synth_points<0>, dim3(grid), dim3(kBlock), 0, c->stream, (v4f*)xyzi_out_device, n, seed);
  KMC_HIP_TRY(c, hipGetLastError());
  return KMC_OK;
}

int kmc_synth_points_host(float* out, uint64_t n, uint64_t seed) {
  if (n && !out) return KMC_ERR_INVALID_ARG;
  for (uint64_t i = 0; i < n; ++i) {
    const kmc_synth::Point p = kmc_synth::make_point(i, n, seed);
    out[4 * i + 0] = p.x;
    out[4 * i + 1] = p.y;
    out[4 * i + 2] = p.z;
    out[4 * i + 3] = p.i;
  }
  return KMC_OK;
}
}  // extern "C"

// grid_probe.hip -- what happens to a launch whose grid exceeds 2^32 work-items (gridDim.x * blockDim.x)?
// The AQL dispatch packet carries the grid size in work-items in 32 bits.  Measured on ROCm 7.2 / gfx950 (profiles/r02_grid_probe.txt):
// exactly 2^32 is refused (hipErrorInvalidConfiguration); anything beyond is ACCEPTED and wraps modulo 2^32 -- 67 108 865 blocks
// of 64 threads run as ONE block, with hipSuccess from the launch and from the synchronize.  kmc_internal.hip.h's grid_for()
// therefore never asks for more than (2^32 - 1) / blockDim workgroups; the kernels' tile loops take the rest.
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ void count_blocks(unsigned long long* c) {
  if (threadIdx.x == 0 && (blockIdx.x & 0xFFFFF) == 0) atomicAdd(c, 1ull);
}

int main() {
  unsigned long long* c = nullptr;
  if (hipMalloc(&c, 8) != hipSuccess) return 1;
  for (unsigned long long grid : {67108863ull, 67108864ull, 67108865ull, 70000000ull, 2147483647ull}) {
    if (hipMemset(c, 0, 8) != hipSuccess) return 2;
    hipLaunchKernelGGL(count_blocks, dim3((unsigned)grid), dim3(64), 0, 0, c);
    const hipError_t e1 = hipGetLastError();
    const hipError_t e2 = hipDeviceSynchronize();
    unsigned long long h = 0;
    if (hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost) != hipSuccess) return 3;
    printf("grid %llu x 64: launch %s, sync %s, blocks counted (every 2^20th) %llu (a full grid would give %llu)\n", grid, hipGetErrorName(e1),
           hipGetErrorName(e2), h, (grid + 0xFFFFF) >> 20);
  }
  return 0;
}

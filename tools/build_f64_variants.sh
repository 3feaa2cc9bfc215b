#!/bin/bash
# Variant libraries of the f64 Eigen-layout kernel for tools/ab_f64_waves.py (an A/B on ONE box, variants interleaved), next to the product's objects:
#   old      the kernel of a given git revision (default HEAD~0 of the caller's choice: pass it as $1), linked with today's other objects
#   w6, w8   today's kernel at 6 / 8 waves per SIMD (KMC_F64_WAVES)
#   tpw2     today's kernel, two tiles per workgroup, all loads in flight before the first fma (KMC_F64_TPW=2); tpw2w6: both
#   p<N>     N persistent waves walking the tiles, next tile's loads in flight while the current one is computed (KMC_F64_PERSISTENT=N)
# usage: tools/build_f64_variants.sh [old-revision]      (run in the repository root, after `make -C kitti_motion_compensation_amd/csrc`)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/kitti_motion_compensation_amd"
mkdir -p lib/ab
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fhip-fp32-correctly-rounded-divide-sqrt -I../include"
OTHERS="lib/obj/kmc_capi_core.o lib/obj/kmc_capi_deskew.o lib/obj/kmc_capi_project.o lib/obj/kmc_capi_synth.o lib/obj/kmc_capi_hostpool.o lib/obj/kmc_capi_direct.o"
link() { hipcc --offload-arch=gfx950 -fPIC -shared -o "lib/ab/libkmc_hip_$1.so" $OTHERS "$2" "$3" -lhsa-runtime64; }
variant() {  # name, extra flags
  hipcc $FLAGS $2 -c -o "lib/ab/f64_$1.o" csrc/kmc_capi_f64.hip
  hipcc $FLAGS $2 -c -o "lib/ab/traj_$1.o" csrc/kmc_capi_traj.hip
  link "$1" "lib/ab/f64_$1.o" "lib/ab/traj_$1.o"
}
variant w6 -DKMC_F64_WAVES=6 &
variant w8 -DKMC_F64_WAVES=8 &
variant tpw2 -DKMC_F64_TPW=2 &
variant tpw2w6 "-DKMC_F64_TPW=2 -DKMC_F64_WAVES=6" &
wait
# persistent waves walking the tiles with the next tile's loads in flight (the in-place route's kernel on resident columns)
variant p2048 -DKMC_F64_PERSISTENT=2048 &
variant p4096 -DKMC_F64_PERSISTENT=4096 &
variant p8192 -DKMC_F64_PERSISTENT=8192 &
variant p16384 -DKMC_F64_PERSISTENT=16384 &
if [ -n "$1" ]; then
  OLD=$(mktemp -d)
  (cd "$ROOT" && git archive "$1" kitti_motion_compensation_amd/csrc include | tar -x -C "$OLD")
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fhip-fp32-correctly-rounded-divide-sqrt -I"$OLD/include" -c -o lib/ab/f64_old.o "$OLD/kitti_motion_compensation_amd/csrc/kmc_capi_f64.hip" &
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fhip-fp32-correctly-rounded-divide-sqrt -I"$OLD/include" -c -o lib/ab/traj_old.o "$OLD/kitti_motion_compensation_amd/csrc/kmc_capi_traj.hip" &
  wait
  link old lib/ab/f64_old.o lib/ab/traj_old.o
  rm -rf "$OLD"
fi
wait
ls -la lib/ab/*.so

#!/bin/bash
# Times the motion_compensate_runs CLI on a synthetic KITTI-shaped run:  tools/time_run_cli.sh [n_frames=216]
set -e
cd "$(dirname "$0")/.."
export LD_LIBRARY_PATH=kitti_motion_compensation_amd/lib:$LD_LIBRARY_PATH
N=${1:-216}
D=$(mktemp -d)
python tools/make_synthetic_run.py "$D" "$N" >/dev/null
for bf in ${KMC_BATCH_LIST:-8 16 32 64}; do
  for rep in $(seq 1 ${KMC_REPS:-3}); do
    find "$D" -maxdepth 2 -name 'velodyne_points_*' -exec rm -rf {} +
    s=$(date +%s%N)
    KMC_RUN_BATCH_FRAMES=$bf KMC_RUN_TIMING=1 kitti_motion_compensation_amd/lib/motion_compensate_runs "$D" 2>&1 >/dev/null | tail -1
    e=$(date +%s%N)
    echo "batch_frames $bf wall_ms $(( (e - s) / 1000000 ))"
  done
done
rm -rf "$D"

#!/bin/bash
# The round's closing measurements in one call on the GPU box:  gpurun --timeout 1500 -- 'tools/final_measure.sh r04fin'
# (tiers, small batches, measure_configs --quick, the five-drive soak, the drop-in frame times with the pool on / off, the link probe)
set -u
TAG=${1:-fin}
cd "$(dirname "$0")/.."
O=gpurun_out/$TAG
mkdir -p "$O"
python tools/measure_tiers.py > "$O/tiers.txt" 2>/dev/null
python tools/measure_small_batches.py > "$O/small_batches.txt" 2>/dev/null
python tools/measure_configs.py --quick > "$O/measure_configs.json" 2> "$O/measure_configs.err"
python tools/measure_traj_batch.py > "$O/traj_batch.txt" 2> "$O/traj_batch.err"
python tests/soak_config5.py 62 > "$O/config5_soak.json" 2> "$O/config5_soak.err"
for i in 1 2 3; do kitti_motion_compensation_amd/lib/time_dropin_frame tests/golden 300 >> "$O/dropin_frame.txt" 2>&1; done
KMC_HOST_POOL=0 kitti_motion_compensation_amd/lib/time_dropin_frame tests/golden 300 >> "$O/dropin_frame.txt" 2>&1
python tools/pcie_probe.py > "$O/pcie_probe.txt" 2>&1
tail -3 "$O/tiers.txt"; tail -4 "$O/dropin_frame.txt" | cut -c1-300; tail -c 400 "$O/config5_soak.json"

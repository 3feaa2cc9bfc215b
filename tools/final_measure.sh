#!/bin/bash
# The round's closing measurements in one call on the GPU box:  gpurun --timeout 1800 -- 'tools/final_measure.sh r05fin'
# (tiers, small batches, measure_configs --quick, the five-drive soak, the drop-in frame times with the pool on / off, the link probe)
set -u
TAG=${1:-fin}
cd "$(dirname "$0")/.."
O=gpurun_out/$TAG
mkdir -p "$O"
timeout 300 python tools/measure_tiers.py > "$O/tiers.txt" 2>/dev/null
timeout 300 python tools/measure_small_batches.py > "$O/small_batches.txt" 2>/dev/null
timeout 600 python tools/measure_configs.py --quick > "$O/measure_configs.json" 2> "$O/measure_configs.err"
timeout 300 python tools/measure_traj_batch.py > "$O/traj_batch.txt" 2> "$O/traj_batch.err"
timeout 300 python tests/soak_config5.py 62 > "$O/config5_soak.json" 2> "$O/config5_soak.err"
for i in 1 2 3; do timeout 120 kitti_motion_compensation_amd/lib/time_dropin_frame tests/golden 300 >> "$O/dropin_frame.txt" 2>&1; done
KMC_HOST_POOL=0 timeout 120 kitti_motion_compensation_amd/lib/time_dropin_frame tests/golden 300 >> "$O/dropin_frame.txt" 2>&1
timeout 300 python tools/pcie_probe.py > "$O/pcie_probe.txt" 2>&1
# round 5: the C++ clients of the final build (per-call dispatch through the direct queue and its HIP-launch twin, lists, batches; probes)
L=kitti_motion_compensation_amd/lib
timeout 300 $L/time_frame_stream 108 kitti 3 50 > "$O/frame_stream_kitti.json" 2> "$O/frame_stream.err"
timeout 300 $L/time_frame_stream 256 1000000 1 8 > "$O/frame_stream_1M.json" 2>> "$O/frame_stream.err"
timeout 300 $L/time_frame_stream 64 10000000 1 4 > "$O/frame_stream_10M.json" 2>> "$O/frame_stream.err"
timeout 120 $L/launch_probe > "$O/launch_probe.json" 2>&1
timeout 120 $L/link_probe > "$O/link_probe.json" 2>&1
timeout 120 $L/cols_probe 64000000 5 6 > "$O/cols_probe.csv" 2>&1
for i in 1 2 3; do timeout 60 $L/time_hsa_init >> "$O/hsa_init.jsonl" 2>&1; timeout 60 $L/time_startup >> "$O/startup.jsonl" 2>&1; done
tail -3 "$O/tiers.txt"; tail -4 "$O/dropin_frame.txt" | cut -c1-300; tail -c 400 "$O/config5_soak.json"

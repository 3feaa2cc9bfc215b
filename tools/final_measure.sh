mkdir -p gpurun_out/r3fin
python tools/measure_tiers.py > gpurun_out/r3fin/tiers.txt 2>/dev/null
python tools/measure_small_batches.py > gpurun_out/r3fin/small_batches.txt 2>/dev/null
python tools/measure_configs.py --quick > gpurun_out/r3fin/measure_configs.json 2> gpurun_out/r3fin/measure_configs.err
python tests/soak_config5.py 62 > gpurun_out/r3fin/config5_soak.json 2> gpurun_out/r3fin/config5_soak.err
for i in 1 2 3; do kitti_motion_compensation_amd/lib/time_dropin_frame tests/golden 300 >> gpurun_out/r3fin/dropin_frame.txt 2>&1; done
KMC_HOST_POOL=0 kitti_motion_compensation_amd/lib/time_dropin_frame tests/golden 300 >> gpurun_out/r3fin/dropin_frame.txt 2>&1
python tools/pcie_probe.py > gpurun_out/r3fin/pcie_probe.txt 2>&1
tail -3 gpurun_out/r3fin/tiers.txt; tail -4 gpurun_out/r3fin/dropin_frame.txt; tail -c 400 gpurun_out/r3fin/config5_soak.json

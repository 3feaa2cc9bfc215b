#!/bin/bash
# SQ / TCC counters of the bench kernel, each group in its OWN rocprofv3 --pmc pass (never combined with a trace domain):
#   gpurun --timeout 900 -- 'tools/collect_sq_counters.sh r02'   then   python tools/summarize_sq_counters.py gpurun_out/r02 r02
set -u
TAG=${1:-r02}
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/$TAG
mkdir -p "$O"
export TMPDIR=/tmp
# KMC_SQ_CMD overrides the profiled command (e.g. "python bench.py --legs-only --no-cpu-baseline" for the secondary kernels)
BENCH=${KMC_SQ_CMD:-"python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-live-traffic --no-configs3 --no-legs --sustained-seconds 0"}
i=0
for group in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR TCC_HIT_sum TCC_MISS_sum" \
             "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i + 1))
  echo "== pass $i: $group" >&2
  rocprofv3 --pmc $group --output-format csv -d "$O/sq_$i" -o bench -- $BENCH > "$O/sq_$i.log" 2>&1
done
ls "$O"/sq_*/ 2>/dev/null | head

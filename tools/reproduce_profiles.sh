#!/bin/bash
# Regenerates the raw rocprofv3 output behind profiles/<tag>_* in ONE session on the GPU box:
#   gpurun --timeout 1500 -- 'tools/reproduce_profiles.sh r01'      (writes gpurun_out/<tag>/, merged back by gpurun)
#   python tools/summarize_profiles.py gpurun_out/<tag> <tag>       (afterwards, here: writes profiles/<tag>_*)
# Counters are collected in their own passes, never together with a trace domain.
set -u
TAG=${1:-r01}
cd "$(dirname "$0")/.."
ROOT=$PWD
O=$ROOT/gpurun_out/$TAG
mkdir -p "$O"
export TMPDIR=/tmp
run() { echo "== $*" >&2; timeout 600 "$@"; }
run rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -o bench -- python bench.py --no-cpu-baseline --no-live-traffic --no-configs3 --no-legs --sustained-seconds 0 > "$O/bench_kt.log" 2>&1
run rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -o bench -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-live-traffic --no-configs3 --no-legs --sustained-seconds 0 > "$O/bench_fetch.log" 2>&1
run rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -o bench -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-live-traffic --no-configs3 --no-legs --sustained-seconds 0 > "$O/bench_write.log" 2>&1
run rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/cal_fetch" -o tune -- kitti_motion_compensation_amd/lib/copy_ceiling 67108864 1 1 > "$O/cal_fetch.csv" 2>/dev/null
run rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/cal_write" -o tune -- kitti_motion_compensation_amd/lib/copy_ceiling 67108864 1 1 > "$O/cal_write.csv" 2>/dev/null
run kitti_motion_compensation_amd/lib/copy_ceiling 67108864 5 10 > "$O/ceilings.csv" 2> "$O/ceilings.err"
run rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt_legs" -o legs -- python bench.py --legs-only --no-cpu-baseline > "$O/legs_kt.log" 2>&1
# the same legs with the barrier bit on every dispatch: kernel rows of frames that do not overlap under the tracer
KMC_ANY_ORDER=0 run rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt_legs_serial" -o legs -- python bench.py --legs-only --no-cpu-baseline > "$O/legs_serial_kt.log" 2>&1
# the C++ frame-stream client on its own (bench.py runs it as a child process, which the passes above do not trace): rows of the
# per-frame kernel, the frame-list kernel (list call and gathered calls) and the packed batch on the same frames
run rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt_stream" -o stream -- kitti_motion_compensation_amd/lib/time_frame_stream 256 1000000 1 4 > "$O/stream_kt.json" 2> "$O/stream_kt.err"
run rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_stream_fetch" -o stream -- kitti_motion_compensation_amd/lib/time_frame_stream 256 1000000 1 2 > /dev/null 2>&1
run rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_stream_write" -o stream -- kitti_motion_compensation_amd/lib/time_frame_stream 256 1000000 1 2 > /dev/null 2>&1
# the default invocation, un-profiled: the ONE compact line (stdout) and the full record it was extracted from (bench_detail.json)
KMC_BENCH_DETAIL="$O/bench_detail.json" timeout 600 python bench.py > "$O/bench_plain.json" 2> "$O/bench_plain.err"
tail -1 "$O/bench_plain.json" | cut -c1-200
grep -h "deskew_batch_f32" "$O"/kt/bench_kernel_stats.csv | cut -c1-60,200-320

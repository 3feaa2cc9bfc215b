"""bench.py's one-line JSON contract (a short run on the GPU box)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# what the driver's record of round 4 held (BENCH_r04.json "parsed"): the line must keep carrying every one of them
R04_PARSED_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                   "dtype", "data", "config", "roofline", "cpu_baseline")
LINE_TARGET_BYTES, LINE_LIMIT_BYTES = 4096, 8192


def _no_prose(obj, limit=200):
    """Every string of the line is an identifier or a short label, never a paragraph."""
    if isinstance(obj, dict):
        for v in obj.values():
            _no_prose(v, limit)
    elif isinstance(obj, list):
        for v in obj:
            _no_prose(v, limit)
    elif isinstance(obj, str):
        assert len(obj) <= limit, obj


def check_line(stdout, n1=True):
    """The contract of the ONE stdout line: a single JSON object, compact (BENCH_r05: a 21 KB line, the driver's parsed = null), round-trips,
    carries the key set the driver parsed in round 4."""
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    assert stdout.rstrip("\n").splitlines()[-1] == lines[0]  # it is the LAST line of stdout
    line = lines[0]
    assert len(line) < LINE_TARGET_BYTES, len(line)
    d = json.loads(line)
    assert json.loads(json.dumps(d)) == d
    for key in R04_PARSED_KEYS:
        if key == "cpu_baseline" and not n1:
            continue
        assert key in d, key
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in d["roofline"], key
    _no_prose(d)
    return d, line


def run_bench(cmd, tmp_path, env=None, cwd=None, timeout=900):
    """-> (the parsed line, the full record bench.py wrote beside it, the process)"""
    detail = os.path.join(str(tmp_path), "bench_detail.json")
    env = dict(os.environ if env is None else env, KMC_BENCH_DETAIL=detail)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=cwd)
    assert r.returncode == 0, r.stderr[-3000:]
    d, _ = check_line(r.stdout, n1="cpu_baseline" in r.stdout)
    assert d["detail"] == detail
    with open(detail) as fh:
        full = json.load(fh)
    for k in ("value", "ms_per_step", "n_gpus", "steps", "warmup"):
        assert full[k] == d[k]  # the line is an extract of the record, not a second measurement
    return d, full, r


def test_compact_line_from_the_21_KB_record_of_round_5():
    """CPU: bench.compact_line on the full record whose printed form the driver could not hold (profiles/r05_bench_default.json, 21 KB):
    < 4 KB, every key of the round-4 parsed set, roofline + cpu_baseline intact, one scalar per leg.  And the size guard raises."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("kmc_bench_module_line", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert (bench.LINE_TARGET_BYTES, bench.LINE_LIMIT_BYTES) == (LINE_TARGET_BYTES, LINE_LIMIT_BYTES)
    with open(os.path.join(ROOT, "profiles", "r05_bench_default.json")) as fh:
        full = json.load(fh)
    assert len(json.dumps(full)) > 20_000
    full["roofline"]["traffic_kind"] = "live_pmc_this_run"
    d, line = check_line(bench.compact_line(full, "gpurun_out/bench_detail.json") + "\n")
    assert d["value"] == full["value"] and d["roofline"]["frac"] == full["roofline"]["frac"] and d["roofline"]["traffic"] == full["roofline"]["traffic"]
    assert d["cpu_baseline"]["value"] == full["cpu_baseline"]["value"] and d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["kind"] == "port"
    assert d["legs"]["f64cols_frac"] == full["f64cols"]["frac"] and d["legs"]["c2_per_call_us"] > 0 and d["configs3"]["value"] == full["configs3"]["value"]
    assert all(isinstance(v, (int, float, bool)) for v in d["legs"].values())  # one scalar per leg, nothing else
    # a record that grows (a leg nobody told compact_line about) cannot grow the line
    full["a_new_leg"] = {"note": "x" * 50_000}
    assert len(bench.compact_line(full, None)) == len(line) - len('"gpurun_out/bench_detail.json"') + len("null")
    # and a line that would not fit is an error, not a silent 21 KB
    full["config"]["workload"] = "w" * 100
    bench_limit = bench.LINE_LIMIT_BYTES
    try:
        bench.LINE_TARGET_BYTES, bench.LINE_LIMIT_BYTES = 10, 20
        with pytest.raises(RuntimeError):
            bench.compact_line(full, None)
    finally:
        bench.LINE_TARGET_BYTES, bench.LINE_LIMIT_BYTES = LINE_TARGET_BYTES, bench_limit


@pytest.mark.gpu
def test_bench_prints_one_contract_line(tmp_path):
    line, d, r = run_bench([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2",
                            "--frames-per-step", "176", "--cpu-sample-frames", "1", "--configs3-frames", "24", "--configs3-frames-per-launch", "8", "--no-live-traffic",
                            "--sustained-seconds", "1.5"], tmp_path)
    # ---- the line itself (check_line: one object, < 4 KB, round-trips, the round-4 key set, no prose) ----
    assert line["n_gpus"] == 1 and line["steps"] == 6 and line["warmup"] == 2 and line["dtype"] == "f32" and line["unit"] == "Mpts/s"
    assert line["higher_is_better"] is True and line["scaling"] == "weak" and line["vs_baseline"] is None and line["data"] == "synthetic"
    assert "workload" in line["config"] and "model" not in line["config"] and line["config"]["workload"].startswith("configs[1]")
    lrf = line["roofline"]
    assert lrf["bound"] == "hbm" and lrf["unit"] == "GB/s" and lrf["peak"] == 8000.0 and abs(lrf["frac"] - lrf["achieved"] / lrf["peak"]) < 1e-3
    assert lrf["traffic_source"].startswith("committed_pmc") and lrf["traffic"] > 0
    lcb = line["cpu_baseline"]
    assert lcb["kind"] == "port" and lcb["cores"] == 1 and lcb["value"] > 0 and "sample" in lcb and lcb["all_cores"]["cores"] >= 1
    assert line["parity_spot_check"]["max_rel_err"] <= 1e-5 and line["configs3"]["parity_max_rel_err"] <= 1e-5
    lg = line["legs"]  # one scalar per leg, each equal to the record's number
    assert lg["c1_list_frac"] == d["configs1_literal"]["list_one_launch"]["frac"] and lg["f64cols_frac"] == d["f64cols"]["frac"]
    assert lg["c2_per_call_us"] == d["configs2_drive"]["frame_by_frame_from_c"]["per_call"]["us_per_frame"]
    assert lg["dropin_frame_f64_us"] == d["dropin_cpp"]["MotionCompensateFrame_f64"]["page_locked_containers"]["us_per_frame"]
    assert lg["sharded_cpp_world"] == 1 and lg["run_frames_per_s"] > 50
    # ---- the full record (bench_detail.json): everything below is read from the file ----
    for key in R04_PARSED_KEYS:
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["unit"] == "Mpts/s"
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    # value and achieved describe the same launches: value [Mpts/s] * 32 B ~= achieved [GB/s]; on this 6-step toy run the
    # wall clock (value) also carries the Python launch overhead of the first steps, hence the loose band
    assert 0.7 < (d["value"] * 32 / 1e3) / rf["achieved"] <= 1.02
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and "sample" in cb
    assert d["parity_spot_check"]["max_rel_err"] <= 1e-5 and d["parity_spot_check"]["intensity_bit_identical"]
    assert d["value"] > 50_000  # > 50 G points/s even on a short, cold run
    assert "committed" in rf["traffic_source"]  # --no-live-traffic: the committed figure, and the record says so
    # the configs[3] leg of the same invocation: 10 M-point frames, the rank's contiguous frame range, oracle-checked
    c3 = d["configs3"]
    assert c3["frames_total"] == 8000 and c3["points_per_frame"] == 10_000_000
    assert c3["rank_frame_ranges"] == [[0, 8000]] and c3["timed_frames_per_rank"] == 24
    assert c3["value"] > 50_000 and c3["unit"] == "Mpts/s"
    assert c3["parity_first_last_frame_per_rank"]["max_rel_err"] <= 1e-5
    su = d["sustained"]  # the headline launch back to back for a wall-clock budget, in ~0.1 s windows
    assert su["windows"] >= 5 and su["Mpts_s_min"] <= su["Mpts_s_mean"] <= su["Mpts_s_max"] and su["Mpts_s_mean"] > 50_000
    # the secondary configurations as legs of the same line (round 3): time, GB/s, fraction of peak, kernel, oracle spot check
    lit = d["configs1_literal"]  # round 4: driven from C++ (tools/time_frame_stream.hip), separate allocations per frame
    for k in ("in_order", "in_order_drained", "gathered_calls", "list_one_launch", "batch_packed"):
        assert lit[k]["us_per_frame"] > 0 and 0.2 < lit[k]["frac"] < 1.0, (k, lit[k])
    assert lit["any_order_dispatch_verdict"] == 1 and lit["list_equals_per_call_bitwise"] is True
    assert lit["list_one_launch"]["frac"] > 0.70, lit["list_one_launch"]  # separate frames handed over as a list run as ONE launch: north_star's 70 % target
    # overlapping frames beats draining the chip: over four queues, and on ONE stream by dropping the barrier bit between independent frames
    # (the three are short timed regions of ONE run: which is faster by how much is recorded under profiles/, not gated here -- a hiccup
    # of the box in one region must not fail the suite; what IS checked is that the barrier-free route was really taken)
    assert lit["in_order"]["dispatched_without_barrier_bit"] > 0.9
    assert lit["gathered_calls"]["frac"] > 0.70, lit["gathered_calls"]  # one call per frame, gathered by the library: north_star's 70 % target
    assert lit["parity"]["max_rel_err"] <= 1e-5
    fb = d["configs2_drive"]["frame_by_frame_from_c"]  # the reference's calling pattern on KITTI-sized frames, from C++
    assert fb["list_equals_per_call_bitwise"] is True and fb["per_call"]["us_per_frame"] > 0 and fb["list_rate_vs_batched"] > 0.5, fb
    assert fb["gathered_rate_vs_batched"] > 0.25, fb  # KITTI-sized per-frame calls at >= 25 % of the batched rate, from C
    # the default context issues HIP launches (round 6: the direct queue is opt-in); the opted-in twin is measured beside it
    assert "through_the_direct_queue" not in lit["in_order"] and "through_the_direct_queue" not in fb["per_call"]
    if lit["in_order_direct_queue"].get("through_the_direct_queue", 0) > 0:  # (where the device offers it)
        assert lit["in_order_direct_queue"]["through_the_direct_queue"] > 0.99 and fb["per_call_direct_queue"]["through_the_direct_queue"] > 0.99
        assert fb["per_call_direct_queue"]["us_per_frame"] < fb["per_call"]["us_per_frame"], fb  # below the HIP runtime's launch path
        nk = fb["per_call_nknot3_direct_queue"]  # north_star's three bracketing poses, one call per frame: the same queue, the records in the argument block
        assert nk["through_the_direct_queue"] > 0.99 and nk["same_bits_as_hip_launches"] is True and nk["us_per_frame"] < fb["per_call_nknot3"]["us_per_frame"], fb
        assert lit["in_order_nknot3_direct_queue"]["same_bits_as_hip_launches"] is True and lit["in_order_nknot3_direct_queue"]["frac"] > 0.5, lit["in_order_nknot3_direct_queue"]
    assert fb["list_launches"] == 1 and fb["list_rate_vs_batched"] > 0.85, fb  # a drive's list: ONE launch, its records in the kernel arguments
    ce = d["ceilings"]  # the box's own ceilings for the kernels' access patterns, measured in this run
    assert ce["f32_one_stream_in_one_out"]["GBps_median"] > 5000 and ce["f64_nine_column_streams"]["copy_cols9"]["GBps_median"] > 4500
    f64 = d["f64cols"]  # K launches between one event pair; the single-call figure beside it
    assert 0.85 < f64["frac_of_9_stream_ceiling"] < 1.25 and f64["homogeneous_column_known_to_be_ones"]["frac"] > 0.6, f64
    assert "back to back" in f64["timed_as"] and f64["single_call"]["us_per_call"] > 0
    dc = d["dropin_cpp"]  # the API north_star names, through the C++ library
    assert dc["MotionCompensateFrame_f64"]["page_locked_containers"]["us_per_frame"] > 0 and dc["MotionCompensateFrame_f64"]["pageable_containers"]["us_per_frame"] > 0
    assert dc["MotionCompensateKittiCloud_f32"]["page_locked_containers"]["us_per_frame"] > 0
    assert dc["parity"]["MotionCompensateFrame_f64_max_rel_err"] <= 1e-11 and dc["parity"]["MotionCompensateKittiCloud_f32_max_rel_err"] <= 1e-5
    assert dc["oracle_faithful_1_thread_ms_per_frame"] > 1.0
    tr = dc["trace_of_MotionCompensateFrame_f64"]  # where the microseconds of the literal call go
    assert tr["calls_traced"] > 100 and 0 < tr["inside_the_c_abi_us"]["device_first_wave_to_last_store"] < tr["medians_us"]["whole_call"]
    assert dc["MotionCompensateFrame_3arg_f64"]["page_locked_containers"]["us_per_frame"] > 0
    sh = d["sharded_cpp"]  # frame ranges per rank from C++, counters reduced by one native RCCL group
    assert sh["world"] == 1 and sh["reduced_by"].startswith("rccl") and sh["reduction_agrees_with_host_arithmetic"] is True
    run = dc["MotionCompensateRun"]
    assert run["frames_compensated"] == 214 and run["frames_per_s"] > 50 and run["parity"]["max_rel_err"] <= 1e-5
    rk = d["ranks"]
    assert rk["world_size"] == 1 and rk["distinct_devices"] == 1 and len(rk["devices"]) == 1
    for leg, bar in (("configs2_drive", 1e-5), ("nknot3", 1e-5), ("f64cols", 1e-11)):
        assert d[leg]["GBps"] > 3000 and abs(d[leg]["frac"] - d[leg]["GBps"] / 8000.0) < 1e-3 and "kernel" in d[leg], (leg, d[leg])
        assert d[leg]["parity"]["max_rel_err"] <= bar and d[leg]["parity"]["bar"] == bar, (leg, d[leg]["parity"])
    assert d["nknot3"]["parity"]["segments_seen"] == [0, 1]
    assert "absent" in d["config"]["kitti_root"] or "present" in d["config"]["kitti_root"]


@pytest.mark.gpu
def test_bench_rccl_path_initialises_and_reduces_on_one_gpu(tmp_path):
    """WORLD_SIZE = 1 with the RCCL process group forced on: the barriers and the one all_gather of the counters run through RCCL."""
    env = dict(os.environ, KMC_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    line, d, r = run_bench([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1",
                            "--frames-per-step", "8", "--no-cpu-baseline"], tmp_path, env=env)
    assert d["n_gpus"] == 1 and d["value"] > 10_000
    assert d["ranks"]["collective_backend"].startswith("nccl") and d["ranks"]["rccl_world_size"] == 1  # what the process group itself reports
    assert line["ranks"]["backend"].startswith("nccl") and line["ranks"]["rccl_world_size"] == 1


@pytest.mark.gpu
def test_bench_two_ranks_end_to_end_on_one_gpu(tmp_path):
    """The driver's N > 1 launch line (torch.distributed.run, one process per rank) with both ranks placed on the only GPU of
    the test box and the counters reduced over gloo (KMC_BENCH_DEVICE / KMC_BENCH_BACKEND are test knobs): rank-specific
    workloads, barrier + max-over-ranks timing, ONE JSON line from rank 0 with the whole-job aggregate."""
    env = dict(os.environ, KMC_BENCH_BACKEND="gloo", KMC_BENCH_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29581", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "2",
           "--frames-per-step", "176", "--configs3-frames", "16", "--configs3-frames-per-launch", "8"]
    line, d, r = run_bench(cmd, tmp_path, env=env, cwd=ROOT)
    assert line["n_gpus"] == 2 and line["ranks"]["world_size"] == 2 and line["per_rank"]["Mpts_s_min"] <= line["per_rank"]["Mpts_s_max"] and "legs" not in line
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "cpu_baseline" not in d
    assert d["config"]["frames_per_step_per_gpu"] == 176 and "x2" in d["config"]["parallelism"]
    # two ranks share one GPU here: the aggregate is about one GPU's rate (each rank gets half), never two GPUs' worth
    assert 50_000 < d["value"] < 260_000
    # whole-job aggregate = all ranks' points / max-rank time
    assert abs(d["value"] - 2 * 176 * 1_000_000 * 8 / (d["ms_per_step"] * 8 * 1e-3) / 1e6) / d["value"] < 0.02
    # the configs[3] leg: rank r deskews frames sharding.frame_range(8000, r, 2) of the 10 M-point stream; every rank checked
    # its first and last timed frame against the oracle; the aggregate is all ranks' points / max-rank time
    c3 = d["configs3"]
    assert c3["rank_frame_ranges"] == [[0, 4000], [4000, 8000]] and c3["timed_frames_per_rank"] == 16
    assert abs(c3["value"] - 2 * 16 * 10_000_000 / (c3["ms_per_frame"] * 16 * 1e-3) / 1e6) / c3["value"] < 0.02
    assert 20_000 < c3["value"] < 260_000
    assert c3["parity_first_last_frame_per_rank"]["max_rel_err"] <= 1e-5
    # every rank's own rate is in the line (a straggler is invisible in SUM / MAX); no secondary legs at N > 1
    pr = d["per_rank"]
    assert len(pr["Mpts_s"]) == 2 and pr["Mpts_s_min"] <= pr["Mpts_s_max"] and len(pr["configs3_Mpts_s"]) == 2
    assert "configs1_literal" not in d
    # who measured: the backend, its world size, and every rank's device out of the same all_gather (both ranks share the one GPU HERE,
    # and the record says so: distinct_devices == 1; on the driver's 8-GPU run it must equal n_gpus)
    rk = d["ranks"]
    assert rk["collective_backend"] == "gloo" and rk["rccl_world_size"] is None and rk["world_size"] == 2
    assert [x["rank"] for x in rk["devices"]] == [0, 1] and rk["distinct_devices"] == 1


@pytest.mark.gpu
def test_bench_two_ranks_over_rccl_on_one_gpu_forms_a_group_or_says_exactly_why_not(tmp_path):
    """VERDICT r05 #4, first-contact readiness: the driver's N = 2 launch line over **nccl** (= RCCL), both ranks on the one GPU of this box
    (KMC_BENCH_DEVICE=0).  RCCL either forms the two-rank group -- then the line says rccl_world_size == 2 -- or refuses a GPU that appears
    twice; bench.py must then stop at the rendezvous with exit code 3 and RCCL's own reason on stderr, not hang, not print a number.  Either
    way everything up to the first collective of a real two-GPU run (launcher, rendezvous on 127.0.0.1, eager communicator creation) has
    run here.  What a one-GPU box cannot show -- ncclAllReduce across two devices -- stays unmeasured (README.md)."""
    env = dict(os.environ, KMC_BENCH_DEVICE="0", KMC_BENCH_DETAIL=os.path.join(str(tmp_path), "bench_detail.json"))
    env.pop("KMC_BENCH_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29591", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
           "--frames-per-step", "16", "--configs3-frames", "8", "--configs3-frames-per-launch", "4", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    if r.returncode == 0:
        d, _ = check_line(r.stdout, n1=False)
        assert d["n_gpus"] == 2 and d["ranks"]["backend"].startswith("nccl") and d["ranks"]["rccl_world_size"] == 2 and d["ranks"]["distinct_devices"] == 1
        print("RCCL formed a two-rank group on one device")
    else:
        refusal = [l for l in r.stderr.splitlines() if "RCCL could not form a group of 2 rank(s)" in l]
        assert refusal, r.stderr[-3000:]
        assert not [l for l in r.stdout.splitlines() if l.startswith("{")]  # no number under a label it does not deserve
        print(refusal[0][:600])


@pytest.mark.gpu
def test_bench_gpus_2_without_a_launcher_reexecutes_itself_under_the_launcher(tmp_path):
    """`python bench.py --gpus 2` with no WORLD_SIZE: not a one-GPU number under a two-GPU label -- the script
    re-executes itself as the contract's launch line (both ranks on the one GPU of this box through the test knobs)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(KMC_BENCH_BACKEND="gloo", KMC_BENCH_DEVICE="0")
    line, d, r = run_bench([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--frames-per-step", "32",
                            "--configs3-frames", "8", "--configs3-frames-per-launch", "4", "--no-cpu-baseline"], tmp_path, env=env, cwd=ROOT)
    assert "re-executing as" in r.stderr and "torch.distributed.run" in r.stderr
    assert d["n_gpus"] == 2 and d["ranks"]["world_size"] == 2 and line["n_gpus"] == 2


def test_bench_gpus_n_without_a_launcher_never_runs_as_one_process():
    """CPU: the re-exec path builds the driver's launch line (dry), and refuses -- non-zero, with the line -- on a box with fewer GPUs."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "KMC_BENCH_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "5", "--warmup", "2"], capture_output=True, text=True,
                       timeout=300, env=dict(env, KMC_BENCH_LAUNCH_DRY="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    cmd = json.loads(r.stdout.strip().splitlines()[-1])["relaunch"]
    i = cmd.index("-m")
    assert cmd[i + 1] == "torch.distributed.run" and "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]
    assert cmd[cmd.index("--master-port") + 2].endswith("bench.py")
    import torch

    if torch.cuda.is_available() and torch.cuda.device_count() >= 8:
        return  # a real 8-GPU box would run it; nothing more to check here
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 2 and "not measuring one GPU under an N-GPU label" in r.stderr and "n_gpus" not in r.stdout
    # and a launcher's WORLD_SIZE that disagrees with --gpus is refused as before
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, timeout=300,
                       env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


@pytest.mark.gpu
def test_bench_eight_ranks_dry_run_on_one_gpu(tmp_path):
    """The driver's 8-GPU launch line, dry: `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 ... bench.py --gpus 8`
    with all eight ranks placed on the only GPU of the test box and the counters reduced over gloo (KMC_BENCH_DEVICE /
    KMC_BENCH_BACKEND are test knobs; unset = RCCL, one rank per GPU), with reduced frame counts.  What the first real SCALE run
    must not fail on: rendezvous, rank r owning frames [1000 r, 1000 (r + 1)) of the 8 000-frame stream, every rank's own parity
    check (its oracle capped at cores / world threads), the one all_gather, the per-rank rates, and the memory: the default
    run holds max(256 x 1 M, 2 x 24 x 10 M) points x 32 B = 15.4 GB per rank."""
    env = dict(os.environ, KMC_BENCH_BACKEND="gloo", KMC_BENCH_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", "29587", os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "1",
           "--frames-per-step", "16", "--configs3-frames", "24", "--configs3-frames-per-launch", "12"]
    line, d, r = run_bench(cmd, tmp_path, env=env, cwd=ROOT, timeout=1500)
    assert line["n_gpus"] == 8 and line["ranks"]["world_size"] == 8 and line["ranks"]["distinct_devices"] == 1 and "x8" in line["config"]["parallelism"]
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and "x8" in d["config"]["parallelism"]
    c3 = d["configs3"]
    assert c3["rank_frame_ranges"] == [[1000 * r, 1000 * (r + 1)] for r in range(8)] and c3["timed_frames_per_rank"] == 24
    assert c3["parity_first_last_frame_per_rank"]["max_rel_err"] <= 1e-5  # MAX over the eight ranks' own checks
    pr = d["per_rank"]
    assert len(pr["Mpts_s"]) == 8 and len(pr["configs3_Mpts_s"]) == 8 and all(v > 0 for v in pr["Mpts_s"] + pr["configs3_Mpts_s"])
    rk = d["ranks"]  # (with RCCL, the contract: collective_backend "nccl (= RCCL on ROCm)", rccl_world_size == 8, distinct_devices == 8)
    assert rk["world_size"] == 8 and [x["rank"] for x in rk["devices"]] == list(range(8)) and "pci" in rk["devices"][0]
    # every rank bound its host side to its GPU's NUMA node (kmc_hip_bind_thread_near_device) and still has CPUs to run on: eight ranks under
    # the box's CPU quota, each with a non-empty set (a binding that emptied a rank's mask would have left it where it was)
    assert all(x["host_cpus_allowed"] >= 1 and x["first_host_cpu"] >= 0 for x in rk["devices"]) and line["ranks"]["host_cpus_allowed_min"] >= 1
    assert abs(d["value"] - 8 * 16 * 1_000_000 * 4 / (d["ms_per_step"] * 4 * 1e-3) / 1e6) / d["value"] < 0.02
    # eight ranks' buffers lived on ONE device here; the default run's 15.4 GB per rank is one rank per 288 GB device
    assert 8 * d["peak_device_GiB_per_rank"] < 250


def test_bench_child_tools_do_not_inherit_a_profiler():
    """CPU: the C++ clients bench.py runs as child processes get an environment without an inherited rocprofv3 tool (a counter pass
    inherited through LD_PRELOAD crashed them on the GPU box); everything else passes through."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("kmc_bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    env = {"PATH": "/usr/bin", "LD_PRELOAD": "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so", "ROCPROF_COUNTERS": "pmc: SQ_WAVES",
           "ROCPROFILER_LIBRARY_CTOR": "1", "HSA_TOOLS_LIB": "/opt/rocm/lib/librocprofiler-sdk.so", "KMC_RUN_TIMING": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    out = bench.unprofiled_env(env)
    assert out == {"PATH": "/usr/bin", "KMC_RUN_TIMING": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    keep = {"LD_PRELOAD": "/usr/lib/libjemalloc.so", "HSA_TOOLS_LIB": "/opt/other/libtool.so"}  # somebody else's preloads are not ours to drop
    assert bench.unprofiled_env(keep) == keep


@pytest.mark.gpu
def test_bench_live_traffic_matches_the_algorithmic_bytes(tmp_path):
    """The default at N = 1: HBM bytes per launch from rocprofv3 --pmc child runs of the same invocation (FETCH_SIZE and WRITE_SIZE
    in separate passes) -- within 1 % of 32 B x points per launch, i.e. nothing is re-read."""
    import shutil

    if not (shutil.which("rocprofv3") or os.path.exists("/opt/rocm/bin/rocprofv3")):
        pytest.skip("rocprofv3 not installed")
    env = dict(os.environ, TMPDIR="/tmp")
    line, d, r = run_bench([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--frames-per-step", "64",
                            "--no-cpu-baseline", "--no-legs", "--sustained-seconds", "0"], tmp_path, env=env, cwd="/tmp")
    rf = d["roofline"]
    assert line["roofline"]["traffic"] == rf["traffic"] and line["roofline"]["traffic_source"] == rf["traffic_kind"]
    if "measured in this run" not in rf["traffic_source"]:
        assert "not possible" in rf["traffic_source"] and line["roofline"]["traffic_source"].endswith("not_possible")  # the fallback is announced in the line itself
        pytest.skip("PMC counters could not be collected on this box (bench.py fell back to the committed passes)")
    assert abs(rf["traffic"] / (32.0 * rf["points_per_launch"]) - 1.0) < 0.01

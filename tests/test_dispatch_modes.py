"""ONE script of calls -- the reference's calling pattern (a frame per call) with everything that makes ordering matter: chains, in-place
repeats, an overwritten input, shared inputs, mixed coefficient tiers, empty and ragged frames, other entry points (a list, a batch, an
N-knot frame, a host in-place call) in between -- replayed under every dispatch configuration a context can be in:

    default                          the context's own stream: every frame a HIP launch, barrier-free (hipExtAnyOrderLaunch) where probed when it
                                     shares no buffer with a frame in flight
    direct_queue                     kmc_hip_set_direct_dispatch(ctx, 1): frames go out through the DIRECT QUEUE (AQL packets the library writes
                                     itself, two lanes), without the barrier bit when they share no buffer with a frame in flight
    direct_queue_by_environment      KMC_DIRECT_DISPATCH=1: the same, switched on for every context of the process
    direct_queue_refused             KMC_DIRECT_DISPATCH=0 and a caller that opts in all the same: HIP launches (the A/B switch)
    barrier_bit_on_every_dispatch    KMC_ANY_ORDER=0 (HIP launches, every one ordered)
    direct_queue_with_the_barrier_bit  opted in and KMC_ANY_ORDER=0 (one lane, every packet ordered)
    gathered_calls                   kmc_hip_set_frame_queues(ctx, 4): calls gathered into list launches
    gathered_on_a_callers_stream     the same on torch's stream, with the caller's word that nothing is produced between calls
    serial                           KMC_ANY_ORDER=0 and a synchronize after every call: the definition of "in-order results"

Every buffer the script touches must hold the SAME BITS in every configuration, and those of `serial`; the f32 results are within the bar
of the oracle (checked once, on the serial run).  KMC_ANY_ORDER=0 as a configuration of the WHOLE GPU suite: tests/test_gpu_parity.py's
`ctx` fixture runs every test of that module under it a second time."""
import os

import numpy as np
import pytest

from kitti_motion_compensation_amd import capi
from oracle import oracle as orc
from tests import util

pytestmark = pytest.mark.gpu

MODES = ["serial", "default", "direct_queue", "direct_queue_by_environment", "direct_queue_refused", "barrier_bit_on_every_dispatch", "direct_queue_with_the_barrier_bit", "gathered_calls",
         "gathered_on_a_callers_stream"]


def _make_ctx(mode, torch):
    env = {"serial": {"KMC_ANY_ORDER": "0"}, "barrier_bit_on_every_dispatch": {"KMC_ANY_ORDER": "0"}, "direct_queue_by_environment": {"KMC_DIRECT_DISPATCH": "1"},
           "direct_queue_refused": {"KMC_DIRECT_DISPATCH": "0"}, "direct_queue_with_the_barrier_bit": {"KMC_ANY_ORDER": "0"}}.get(mode, {})
    if mode not in ("direct_queue_by_environment", "direct_queue_refused"):
        env = dict({"KMC_DIRECT_DISPATCH": None}, **env)  # (the suite itself may be running under KMC_DIRECT_DISPATCH=1: tests/test_suite_modes.py)
    saved = {k: os.environ.get(k) for k in env}
    for k, v in env.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    try:
        c = capi.Context(0)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    if mode in ("direct_queue", "direct_queue_refused", "direct_queue_with_the_barrier_bit"):
        c.set_direct_dispatch(True)
    if mode.startswith("gathered"):
        c.set_frame_queues(4)
    if mode == "gathered_on_a_callers_stream":
        c.set_stream(torch.cuda.current_stream().cuda_stream)
        c.set_frame_queue_order(False)
    return c


def _script(torch, xyzi, seed=2025):
    """-> (buffers: name -> host array, ops: list of callables(ctx, dev) issuing one call each).  Deterministic."""
    rng = np.random.default_rng(seed)
    sizes = [123_397, 0, 1, 63, 64, 65, 4097, 30_000, 77_777, 123_397, 1000, 50_003]
    host = {}
    for k, n in enumerate(sizes):
        host[f"in{k}"] = np.ascontiguousarray(xyzi[rng.integers(0, xyzi.shape[0], size=n)])
        host[f"out{k}"] = np.zeros((n, 4), dtype=np.float32)
    n_chain = 60_001
    host["chain0"] = np.ascontiguousarray(xyzi[rng.integers(0, xyzi.shape[0], size=n_chain)])
    for j in range(1, 6):
        host[f"chain{j}"] = np.zeros((n_chain, 4), dtype=np.float32)
    host["inplace"] = host["chain0"].copy()
    host["war_in"] = host["chain0"].copy()
    host["war_out"] = np.zeros((n_chain, 4), dtype=np.float32)
    nl = 40  # a list beyond 16 frames: the kernel-argument block of 64, or the table route
    for k in range(nl):
        host[f"lin{k}"] = np.ascontiguousarray(xyzi[rng.integers(0, xyzi.shape[0], size=2000 + 37 * k)])
        host[f"lout{k}"] = np.zeros_like(host[f"lin{k}"])
    host["batch_out"] = np.zeros((n_chain * 2, 4), dtype=np.float32)
    host["traj_out"] = np.zeros((n_chain, 4), dtype=np.float32)

    def prm(k):
        yaw = 0.6 if k % 5 == 4 else (2.0 if k % 7 == 6 else 0.03)  # series3 mostly, series5 / wide now and then: the tier changes inside the stream
        return capi.FrameParams.make([1.0 + 0.01 * k, 0.02, -0.01, 0.001, -0.002, yaw], float((k * 37 % 100) / 100.0))

    ops = []
    for k in range(len(sizes)):
        ops.append(lambda c, d, k=k: c.deskew_f32(d[f"in{k}"], d[f"out{k}"], prm(k)))
    ops.append(lambda c, d: c.deskew_f32(d["in0"], d["out9"], prm(30)))            # same input as frame 0, overwrites frame 9's output (write after write)
    for j in range(5):
        ops.append(lambda c, d, j=j: c.deskew_f32(d[f"chain{j}"], d[f"chain{j + 1}"], prm(j)))   # a chain
    for j in range(3):
        ops.append(lambda c, d, j=j: c.deskew_f32(d["inplace"], d["inplace"], prm(10 + j)))      # in place, three times
    ops.append(lambda c, d: c.deskew_f32(d["war_in"], d["war_out"], prm(3)))        # reads war_in ...
    ops.append(lambda c, d: c.deskew_f32(d["chain2"], d["war_in"], prm(4)))         # ... which this one overwrites (write after read), reading a chain link

    def the_list(c, d):
        pack = c.prepare_frames([(d[f"lin{k}"], d[f"lout{k}"]) for k in range(nl)], [prm(k) for k in range(nl)])
        c.deskew_frames_f32(pack)
        d["_keep"] = pack
    ops.append(the_list)

    def the_batch(c, d):  # a batch over two chain links, concatenated: reads what single-frame calls wrote
        both = torch.cat([d["chain4"], d["chain5"]])
        d["_both"] = both
        c.deskew_batch_f32(both, d["batch_out"], np.array([0, n_chain, 2 * n_chain], dtype=np.uint64), [prm(20), prm(21)], None)
    ops.append(the_batch)
    T0, T1 = 47072.283701593, 47072.386973931
    knots_t = np.array([T0, 0.5 * (T0 + T1), T1])
    Pa = orc.Affine.identity()
    Pb = orc.se3_exp(np.array([0.6, 0.02, 0.0, 0.0, 0.0, 0.012]))
    Pc = orc.affine_mul(Pb, orc.se3_exp(np.array([0.7, -0.01, 0.01, 0.001, 0.0, 0.02])))
    knots_P = np.stack([P.rt12() for P in (Pa, Pb, Pc)])
    ops.append(lambda c, d: c.deskew_traj_f32(d["war_out"], d["traj_out"], knots_t, knots_P, T0, T1, 0.5 * (T0 + T1)))  # an N-knot frame reading a frame's output
    ops.append(lambda c, d: c.deskew_f32(d["out7"], d["out8"], prm(40)))            # and one more frame behind everything
    return host, ops


@pytest.fixture(scope="module")
def replay(golden_dir):
    import torch

    assert torch.cuda.is_available(), "these tests need the GPU: there is no CPU fallback to test"
    xyzi = util.load_velodyne_bin(os.path.join(golden_dir, "kitti_2011_09_26_drive_0005"), 0)
    results = {}
    for mode in MODES:
        host, ops = _script(torch, xyzi)
        dev = {k: torch.from_numpy(v).cuda() for k, v in host.items()}
        torch.cuda.synchronize()
        ctx = _make_ctx(mode, torch)
        try:
            for op in ops:
                op(ctx, dev)
                if mode == "serial":
                    ctx.synchronize()
            ctx.synchronize()  # issues whatever is still gathered, waits for the stream
            torch.cuda.synchronize()
            results[mode] = {k: v.cpu().numpy() for k, v in dev.items() if not k.startswith("_")}
            results[mode]["_any_order_launches"] = ctx.any_order_launches()
            results[mode]["_direct_frames"] = ctx.direct_frames()
            results[mode]["_verdict"] = ctx.device_info()["any_order_dispatch"]
        finally:
            ctx.close()
    return results, xyzi


@pytest.mark.parametrize("mode", MODES[1:])
def test_every_dispatch_configuration_writes_the_serial_bits(replay, mode):
    results, _ = replay
    ref, got = results["serial"], results[mode]
    for name in sorted(k for k in ref if not k.startswith("_")):
        assert np.array_equal(got[name].view(np.uint32), ref[name].view(np.uint32)), (mode, name)


def test_the_configurations_really_differ(replay):
    """(a replay that silently ran every mode the same way would prove nothing)"""
    results, _ = replay
    assert results["serial"]["_any_order_launches"] == 0 and results["barrier_bit_on_every_dispatch"]["_any_order_launches"] == 0
    assert results["barrier_bit_on_every_dispatch"]["_verdict"] == 0
    assert results["gathered_calls"]["_any_order_launches"] == 0  # gathered frames go out as list launches
    # the direct queue carries the frames of the configurations that opted in (where this device offers it), never those of the others
    for plain in ("serial", "default", "direct_queue_refused", "barrier_bit_on_every_dispatch", "gathered_calls", "gathered_on_a_callers_stream"):
        assert results[plain]["_direct_frames"] == 0, plain  # the default is HIP launches: frames of a default context are in its HIP stream
    if results["direct_queue"]["_direct_frames"]:
        assert results["direct_queue"]["_direct_frames"] >= 20 and results["direct_queue_with_the_barrier_bit"]["_direct_frames"] >= 20
        assert results["direct_queue_by_environment"]["_direct_frames"] == results["direct_queue"]["_direct_frames"]
        assert results["direct_queue"]["_any_order_launches"] >= 8  # plain AQL semantics: no probe needed for packets without the barrier bit
        assert results["direct_queue_with_the_barrier_bit"]["_any_order_launches"] == 0
    if results["default"]["_verdict"] == 1:
        assert results["default"]["_any_order_launches"] >= 8


def test_the_serial_bits_are_within_the_bar_of_the_oracle(replay):
    results, xyzi = replay
    ref = results["serial"]
    T0, T1 = 0.0, 0.1
    # frame 7 of the script: prm(7) -- an ordinary frame; and the last link of the chain through five frames
    k = 7
    yaw = 0.03
    twist = [1.0 + 0.01 * k, 0.02, -0.01, 0.001, -0.002, yaw]
    x_req = float((k * 37 % 100) / 100.0)
    o = orc.deskew_xyzi_f32(ref["in7"], T0, orc.Affine.identity(), T1, orc.se3_exp(twist), T0 + x_req * (T1 - T0), mode=orc.FAITHFUL)
    assert o["rc"] == orc.OK
    assert util.rel_point_error(ref["out7"][:, :3], o["xyz_f64"]).max() <= 1e-5
    assert np.array_equal(ref["out7"][:, 3].view(np.uint32), ref["in7"][:, 3].view(np.uint32))

"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C-ABI (include/kmc_hip.h), against
the CPU oracle on the same inputs, against the committed golden fixtures, and -- at BASELINE.json's full sizes --
through size-independent properties.

Parity bar (BASELINE.json north_star / SURVEY.md section 8(d)):
  XYZ      : per point ||p_gpu - p_ref||_2 / max(||p_ref||_2, 1e-3) <= 1e-5        (REL_TOL below)
  intensity: bit-identical
  indices  : per-point integer indices (frame index) bit-exact
  f64 path : <= 1e-11 relative (it computes in double like the reference)
"""
import os

import numpy as np
import pytest

from kitti_motion_compensation_amd import capi
from oracle import oracle as orc
from tests import util

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5       # the bar north_star states
REL_TOL_F64 = 1e-11  # f64 boundary mode

T0, T1, TREQ = 47072.283701593, 47072.386973931, 47072.335337762

TRAJECTORIES = {
    "stationary": [0, 0, 0, 0, 0, 0],
    "straight": [1.3, 0.05, -0.02, 0, 0, 0],
    "gentle_turn": [1.3, 0.05, -0.02, 0.001, -0.002, 0.03],
    "hard_turn": [2.9, -0.3, 0.1, 0.02, 0.01, -0.1],
    "spin": [0.4, 0.1, 0.0, 0.1, -0.3, 0.6],       # series5 tier
    "tumble": [0.4, 0.1, 0.0, 0.3, -0.9, 2.2],     # wide tier (theta = 1.2 rad at the mid-scan anchor)
}


@pytest.fixture(scope="module")
def torch_mod():
    import torch

    assert torch.cuda.is_available(), "these tests need the GPU: there is no CPU fallback to test"
    return torch


@pytest.fixture(scope="module", params=["default", "barrier_bit_on_every_dispatch"])
def ctx(request, torch_mod):
    """Every test of this module that takes the shared context runs twice: on a default context -- independent frames may go
    out without the AQL barrier bit where kmc_hip_create's probe verified it -- and on one created with KMC_ANY_ORDER=0, every dispatch
    ordered.  Same assertions, same bits.  (Gathering changes WHEN results become visible -- a deferred issue -- so it cannot be swapped in
    under tests that read results right after a call; tests/test_dispatch_modes.py replays one script under every configuration, gathering
    included.)"""
    if request.param == "default":
        c = capi.Context(0)
    else:
        saved = os.environ.get("KMC_ANY_ORDER")
        os.environ["KMC_ANY_ORDER"] = "0"
        try:
            c = capi.Context(0)
        finally:
            if saved is None:
                os.environ.pop("KMC_ANY_ORDER", None)
            else:
                os.environ["KMC_ANY_ORDER"] = saved
        assert c.device_info()["any_order_dispatch"] == 0
    yield c
    c.close()


@pytest.fixture(scope="module")
def kitti(golden_dir):
    run = os.path.join(golden_dir, "kitti_2011_09_26_drive_0005")
    xyzi = util.load_velodyne_bin(run, 0)
    f = util.load_oxts_fields(run, 0)
    return xyzi, orc.oxts_to_pose(orc.oxts(**f))


def _poses(P1, step):
    return P1, orc.affine_mul(P1, orc.se3_exp(step))


def _params(P1, P2, t0=T0, t1=T1, treq=TREQ):
    return capi.frame_params_from_poses(P1.rt12().reshape(3, 4), P2.rt12().reshape(3, 4), t0, t1, treq)


def _oracle(xyzi, P1, P2, t0=T0, t1=T1, treq=TREQ, mode=orc.FAITHFUL):
    r = orc.deskew_xyzi_f32(xyzi, t0, P1, t1, P2, treq, mode=mode, want_f32=True)
    assert r["rc"] == orc.OK
    return r


def _check(got, xyzi, ref, tol=REL_TOL):
    assert np.array_equal(got[:, 3].view(np.uint32), xyzi[:, 3].view(np.uint32)), "intensity not bit-identical"
    err = util.rel_point_error(got[:, :3], ref["xyz_f64"])
    assert np.isfinite(got[:, :3]).all()
    assert err.max() <= tol, f"max rel err {err.max():.3e} > {tol}"
    return err.max()


def _run_device(torch, ctx, xyzi, params, in_place=False):
    d_in = torch.from_numpy(np.ascontiguousarray(xyzi)).cuda()
    d_out = d_in if in_place else torch.empty_like(d_in)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    st = ctx.deskew_f32(d_in, d_out, params)
    torch.cuda.synchronize()
    return d_out.cpu().numpy(), st


# ---- reference KAT through the HIP path ------------------------------------------------------------------------
def test_reference_kat_through_hip(ctx, kats):
    """test/test_motion_compensation.cpp:54-76 -- same fixture, same expected numbers, ASSERT_FLOAT_EQ semantics."""
    k = kats["motion_compensate_frame"]
    ox = [capi.Oxts(**o) for o in k["oxts"]]
    T_start, T_end = capi.make_frame_poses(ox[0], ox[1], ox[2], k["stamp_start"], k["stamp_end"])
    params = capi.frame_params_from_poses(T_start, T_end, k["stamp_start"], k["stamp_end"], k["requested_time"])
    xyzi = np.array(k["cloud"], dtype=np.float32)
    xyzi[:, 3] = [0.25, 0.5, 0.75]
    out = np.empty_like(xyzi)
    ctx.deskew_f32(xyzi, out, params)  # KMC_MEM_HOST
    util.assert_float_eq(out[:, :3], np.array(k["expected"])[:, :3], "MotionCompensateFrame KAT on HIP")
    assert np.array_equal(out[:, 3], xyzi[:, 3])


# ---- real KITTI frame (BASELINE.json configs[0] data) ------------------------------------------------------------
@pytest.mark.parametrize("name", list(TRAJECTORIES))
def test_kitti_frame_vs_oracle_device(torch_mod, ctx, kitti, name):
    xyzi, P1 = kitti
    P1, P2 = _poses(P1, TRAJECTORIES[name])
    params = _params(P1, P2)
    got, st = _run_device(torch_mod, ctx, xyzi, params)
    assert st.n_points == xyzi.shape[0] and st.n_launches == 1
    expected_tier = {"spin": capi.TIER_SERIES5, "tumble": capi.TIER_WIDE}.get(name, capi.TIER_SERIES3)
    assert st.variant == expected_tier
    _check(got, xyzi, _oracle(xyzi, P1, P2))
    if name == "stationary":  # Log(P1^-1 P1) is zero up to ~1e-16 rounding: the cloud must come back unchanged
        assert np.abs(got.astype(np.float64) - xyzi.astype(np.float64)).max() < 1e-12


def test_kitti_frame_host_buffers_and_in_place(torch_mod, ctx, kitti):
    xyzi, P1 = kitti
    P1, P2 = _poses(P1, TRAJECTORIES["hard_turn"])
    params = _params(P1, P2)
    ref = _oracle(xyzi, P1, P2)
    out = np.empty_like(xyzi)
    ctx.deskew_f32(xyzi, out, params)  # host staging pipeline
    _check(out, xyzi, ref)
    dev, _ = _run_device(torch_mod, ctx, xyzi, params)
    assert np.array_equal(out.view(np.uint32), dev.view(np.uint32)), "host-staged and device-resident results differ"
    inplace, _ = _run_device(torch_mod, ctx, xyzi, params, in_place=True)
    assert np.array_equal(inplace.view(np.uint32), dev.view(np.uint32))


def test_host_pipeline_many_chunks_matches_device_path(torch_mod, ctx, kitti):
    """KMC_MEM_HOST streams 2 M-point chunks through a 4-slot upload/compute/download pipeline: more chunks than slots
    (slot reuse, ragged last chunk) must give the same bits as the device-resident launch."""
    torch = torch_mod
    _, P1 = kitti
    P1, P2 = _poses(P1, TRAJECTORIES["hard_turn"])
    params = _params(P1, P2)
    n = 5 * (1 << 21) + 12345
    xyzi = capi.synth_points_host(n, 77)
    out = np.empty_like(xyzi)
    st = ctx.deskew_f32(xyzi, out, params)
    assert st.n_launches == 6 and st.n_points == n
    dev, _ = _run_device(torch, ctx, xyzi, params)
    assert np.array_equal(out.view(np.uint32), dev.view(np.uint32))
    pinned_in = torch.from_numpy(xyzi).pin_memory()
    pinned_out = torch.empty_like(pinned_in).pin_memory()
    # the caller's OWN page-locked memory (not the library's pool): recognised, one kernel works on it in place over the link
    st_pinned = ctx.deskew_f32(pinned_in.numpy(), pinned_out.numpy(), params)
    assert st_pinned.n_launches == 1 and not capi.host_pool_owns(pinned_in.numpy())
    assert np.array_equal(pinned_out.numpy().view(np.uint32), dev.view(np.uint32))
    sel = np.arange(0, n, 1013)
    _check(out[sel], xyzi[sel], _oracle(xyzi[sel], P1, P2))  # FAITHFUL: the reference's per-point Log / Exp sequence


@pytest.mark.parametrize("theta,tier", [(0.2499, 0), (0.2501, 1), (0.9999, 1), (1.0001, 2), (2.0, 2), (3.0, 2), (3.1415, 2)])
def test_tier_boundaries_keep_the_bar(torch_mod, ctx, kitti, theta, tier):
    """The host picks the coefficient tier from |phi| * max|s|; at the edges of each tier's validity (0.25 rad, 1 rad, and pi --
    the most two poses can be apart -- for the wide polynomial) the truncated series must still sit far inside the 1e-5 bar.
    requested_time = stamp_start makes max|s| = 1."""
    xyzi, P1 = kitti
    axis = np.array([0.3, -0.5, 0.81])
    axis /= np.linalg.norm(axis)
    P1, P2 = _poses(P1, [2.0, -0.4, 0.1, *(theta * axis)])
    params = _params(P1, P2, treq=T0)
    got, st = _run_device(torch_mod, ctx, xyzi, params)
    assert st.variant == tier
    err = _check(got, xyzi, _oracle(xyzi, P1, P2, treq=T0))  # FAITHFUL oracle (trajectory_interpolation.cpp:31-45 per point)
    assert err < 2e-6, err


def test_every_tier_agrees_and_every_route_writes_the_same_bits(torch_mod, ctx, kitti):
    """All four coefficient tiers are valid below 0.25 rad and must agree with the oracle; the same frame through the single-frame
    kernel, a one-frame batch (kernel-argument tables), a one-frame batch padded with empty frames (device tables) and a one-frame
    list writes the same bits."""
    xyzi, P1 = kitti
    P1, P2 = _poses(P1, TRAJECTORIES["hard_turn"])
    params = _params(P1, P2)
    ref = _oracle(xyzi, P1, P2)
    try:
        for tier in (capi.TIER_SERIES3, capi.TIER_SERIES5, capi.TIER_WIDE, capi.TIER_TRIG):
            ctx.force_tier(tier)
            got, st = _run_device(torch_mod, ctx, xyzi, params)
            assert st.variant == tier
            _check(got, xyzi, ref)
        ctx.force_tier(-1)
        torch = torch_mod
        base, _ = _run_device(torch, ctx, xyzi, params)
        n = xyzi.shape[0]
        d_in = torch.from_numpy(xyzi).cuda()
        ident = capi.FrameParams.make([0, 0, 0, 0, 0, 0], 0.5)
        for route in ("batch_inline", "batch_device_tables", "list"):
            d_out = torch.zeros_like(d_in)
            if route == "batch_inline":
                st = ctx.deskew_batch_f32(d_in, d_out, np.array([0, n], dtype=np.uint64), [params], None)
            elif route == "batch_device_tables":  # 20 empty frames behind the real one: more than the kernel-argument tables hold
                st = ctx.deskew_batch_f32(d_in, d_out, np.array([0] + [n] * 21, dtype=np.uint64), [params] + [ident] * 20, None)
            else:
                st = ctx.deskew_frames_f32(ctx.prepare_frames([(d_in, d_out)], [params]))
            torch.cuda.synchronize()
            assert st.n_launches == 1
            assert np.array_equal(d_out.cpu().numpy().view(np.uint32), base.view(np.uint32)), route
    finally:
        ctx.force_tier(-1)


@pytest.mark.parametrize("theta", [3.3, 5.0, 9.0])
def test_caller_supplied_twist_beyond_pi_runs_the_any_angle_tier(torch_mod, ctx, kitti, theta):
    """Two poses are never more than pi apart (Log), but the C-ABI also accepts a raw twist: beyond 3.25 rad the host must
    leave the wide polynomial (fitted on theta <= 3.25) for the any-angle tier.  Reference: the oracle's lie::Exp
    (lie_algebra.cpp:83-92) applied point by point to Exp(s f) p, s = frac(p) - x_req."""
    xyzi = kitti[0][::41].copy()
    axis = np.array([0.2, -0.4, 0.89])
    twist = np.array([1.1, 0.2, -0.05, *(theta * axis / np.linalg.norm(axis))])
    x_req = 0.0
    got, st = _run_device(torch_mod, ctx, xyzi, capi.FrameParams.make(twist, x_req))
    assert st.variant == capi.TIER_TRIG
    p = xyzi[:, :3].astype(np.float64)
    frac = (np.pi - np.arctan2(p[:, 1], p[:, 0])) / (2 * np.pi)  # timestamp_mocking.cpp:46-63
    want = np.empty_like(p)
    for i in range(p.shape[0]):
        T = orc.se3_exp(list((frac[i] - x_req) * twist))
        want[i] = np.array(list(T.R)).reshape(3, 3) @ p[i] + np.array(list(T.t))
    assert np.array_equal(got[:, 3].view(np.uint32), xyzi[:, 3].view(np.uint32))
    assert util.rel_point_error(got[:, :3], want).max() <= REL_TOL
    # ... and the wide tier, forced onto a frame inside ITS domain, agrees with the any-angle tier there
    inside = capi.FrameParams.make(twist * (3.2 / theta), x_req)
    a, st_a = _run_device(torch_mod, ctx, xyzi, inside)
    assert st_a.variant == capi.TIER_WIDE
    try:
        ctx.force_tier(capi.TIER_TRIG)
        b, _ = _run_device(torch_mod, ctx, xyzi, inside)
    finally:
        ctx.force_tier(-1)
    assert util.rel_point_error(a[:, :3], b[:, :3].astype(np.float64)).max() <= 2e-6


@pytest.mark.parametrize("n", [0, 1, 63, 64, 255, 256, 1023, 1024, 1025, 2047, 2048, 2049, 100_003])
def test_ragged_sizes(torch_mod, ctx, kitti, n):
    xyzi, P1 = kitti
    xyzi = np.ascontiguousarray(xyzi[:n])
    P1, P2 = _poses(P1, TRAJECTORIES["gentle_turn"])
    params = _params(P1, P2)
    if n == 0:
        st = ctx.deskew_f32(np.empty((0, 4), np.float32), np.empty((0, 4), np.float32), params)
        assert st.n_points == 0
        return
    # guard words after the buffer must stay untouched
    import torch
    d_in = torch.from_numpy(xyzi).cuda()
    d_out = torch.full((n + 64, 4), 7.0, dtype=torch.float32, device="cuda")
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.deskew_f32(d_in, d_out, params, n=n)
    torch.cuda.synchronize()
    out = d_out.cpu().numpy()
    assert np.all(out[n:] == 7.0), "wrote past the end"
    _check(out[:n], xyzi, _oracle(xyzi, P1, P2))


def test_edge_points_signed_zero_and_axes(ctx, kitti):
    """timestamp_mocking.cpp:46 has no special cases; the kernel's azimuth must follow IEEE atan2 on zeros and axes."""
    _, P1 = kitti
    P1, P2 = _poses(P1, TRAJECTORIES["hard_turn"])
    params = _params(P1, P2)
    pts = np.array([
        [0.0, 0.0, 0.0, 1], [0.0, 0.0, 1.5, 2], [-0.0, 0.0, 0.3, 3], [-0.0, -0.0, 0.3, 4], [0.0, -0.0, 0.3, 5],
        [-1.0, 0.0, 0.0, 6], [-1.0, -0.0, 0.0, 7], [1.0, 0.0, 0.0, 8], [1.0, -0.0, 0.0, 9], [0.0, 5.0, 0.0, 10],
        [0.0, -5.0, 0.0, 11], [-0.0, 5.0, 0.0, 12], [3.0, 3.0, 1.0, 13], [-3.0, 3.0, 1.0, 14], [-3.0, -3.0, 1.0, 15],
        [3.0, -3.0, 1.0, 16], [1e-30, 1e-30, 0.0, 17], [-1e-38, 1e-39, 0.0, 18], [1e-42, -1e-44, 2.0, 19],
        [79.9, -0.001, 1.0, 20], [-79.9, 1e-6, -3.0, 21], [-79.9, -1e-6, -3.0, 22],
    ], dtype=np.float32)
    out = np.empty_like(pts)
    ctx.deskew_f32(pts, out, params)
    ref = _oracle(pts, P1, P2)
    _check(out, pts, ref)
    # and an absolute bound: several points sit at the origin where the relative measure is capped by its 1e-3 floor
    assert np.abs(out[:, :3].astype(np.float64) - ref["xyz_f64"]).max() < 1e-5


def test_invalid_arguments(torch_mod, ctx, kitti):
    xyzi, P1 = kitti
    P1, P2 = _poses(P1, TRAJECTORIES["straight"])
    params = _params(P1, P2)
    # HOST buffers are only copied: a 4-byte-aligned view (numpy slice, std::vector<float>::data() + k) is fine ...
    buf = np.zeros(4 * 16 + 1, dtype=np.float32)
    buf[1:] = xyzi[:16].reshape(-1)
    mis = buf[1:].reshape(-1, 4)
    assert mis.ctypes.data % 16 != 0
    out_buf = np.zeros(4 * 16 + 1, dtype=np.float32)
    got = out_buf[1:].reshape(-1, 4)
    ctx.deskew_f32(mis, got, params)
    want = np.empty((16, 4), dtype=np.float32)
    ctx.deskew_f32(xyzi[:16].copy(), want, params)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # ... DEVICE pointers feed 16-byte vector accesses and must be 16-byte aligned
    d = torch_mod.zeros(4 * 16 + 4, dtype=torch_mod.float32, device="cuda")
    d_mis = d[1:1 + 64].view(-1, 4)
    with pytest.raises(capi.KmcError) as e:
        ctx.deskew_f32(d_mis, d_mis, params)
    assert e.value.status == capi.ERR_INVALID_ARG
    bad = capi.FrameParams.make(params.twist_np(), 1.5)  # requested time outside the scan
    with pytest.raises(capi.KmcError) as e:
        ctx.deskew_f32(xyzi[:8].copy(), np.empty((8, 4), np.float32), bad)
    assert e.value.status == capi.ERR_TIME_OUT_OF_RANGE
    nan = capi.FrameParams.make([np.nan, 0, 0, 0, 0, 0], 0.5)
    with pytest.raises(capi.KmcError) as e:
        ctx.deskew_f32(xyzi[:8].copy(), np.empty((8, 4), np.float32), nan)
    assert e.value.status == capi.ERR_INVALID_ARG


# ---- synthetic frames (BASELINE.json configs[1]) --------------------------------------------------------------------
def test_device_generator_is_bit_identical_to_host(torch_mod, ctx):
    torch = torch_mod
    n = 1_000_000
    d = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.synth_points(d, n, 0x4B4D43)
    torch.cuda.synchronize()
    h = capi.synth_points_host(n, 0x4B4D43)
    assert np.array_equal(d.cpu().numpy().view(np.uint32), h.view(np.uint32))


def test_config2_one_million_points_straight_line(torch_mod, ctx):
    """BASELINE.json configs[1]: synthetic 1 M-point frame, straight-line constant-velocity trajectory.
    OXTS exactly as SURVEY.md section 8(d) config 2; checked against the oracle AND the closed form x' = x + v (t_i - t_req)."""
    n = 1_000_000
    Tz = 47072.0
    v = 10.0
    dlon = v * 0.1 * 180.0 / (np.pi * 6378137.0)
    ox = [capi.Oxts(stamp=Tz + 0.05 + 0.1 * i, lat=0.0, lon=dlon * i, alt=0, roll=0, pitch=0, yaw=0) for i in range(3)]
    t0, tm, t1 = Tz + 0.10, Tz + 0.15, Tz + 0.20
    T_start, T_end = capi.make_frame_poses(ox[0], ox[1], ox[2], t0, t1)
    params = capi.frame_params_from_poses(T_start, T_end, t0, t1, tm)
    assert np.allclose(params.twist_np(), [v * 0.1, 0, 0, 0, 0, 0], atol=1e-9)
    xyzi = capi.synth_points_host(n, 0x4B4D43)
    got, st = _run_device(torch_mod, ctx, xyzi, params)
    assert st.variant == capi.TIER_SERIES3
    oo = [orc.oxts(Tz + 0.05 + 0.1 * i, 0.0, dlon * i, 0, 0, 0, 0) for i in range(3)]
    rc, A, B = orc.make_frame_poses(oo[0], oo[1], oo[2], t0, t1)
    assert rc == orc.OK
    ref = orc.deskew_xyzi_f32(xyzi, t0, A, t1, B, tm, mode=orc.FAITHFUL, want_stamps=True)
    assert ref["rc"] == orc.OK
    _check(got, xyzi, ref)
    closed = xyzi[:, :3].astype(np.float64).copy()
    closed[:, 0] += v * (ref["stamps"] - tm)
    assert util.rel_point_error(got[:, :3], closed).max() <= REL_TOL


# ---- batched kernel ---------------------------------------------------------------------------------------------------
def _batch_case(kitti, sizes, steps, seed=5):
    xyzi, P1 = kitti
    rng = np.random.default_rng(seed)
    frames, params, refs = [], [], []
    for n, step in zip(sizes, steps):
        idx = rng.integers(0, xyzi.shape[0], size=n)
        pts = np.ascontiguousarray(xyzi[idx])
        A, B = _poses(P1, step)
        xr = rng.uniform(0.2, 0.8)
        treq = T0 + xr * (T1 - T0)
        frames.append(pts)
        params.append(_params(A, B, treq=treq))
        refs.append(_oracle(pts, A, B, treq=treq)["xyz_f64"] if n else np.zeros((0, 3)))  # FAITHFUL
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    return np.concatenate(frames) if sum(sizes) else np.zeros((0, 4), np.float32), offsets, params, np.concatenate(refs)


@pytest.mark.parametrize("sizes", [
    [5000, 1, 0, 0, 3000, 1023, 1025, 7, 0, 20000],        # ragged, with empty frames and tile-straddling frames
    [100] * 40,                                             # many tiny frames inside single tiles (> LDS table of 16)
    [123397, 120001, 118733],                               # KITTI-sized frames
    [0, 0, 50, 0],
])
def test_batch_vs_per_frame_oracle_and_bit_exact_indices(torch_mod, ctx, kitti, sizes):
    torch = torch_mod
    names = list(TRAJECTORIES)[:5]  # below the trig tier so that the batch tier is series5 at most
    steps = [TRAJECTORIES[names[i % len(names)]] for i in range(len(sizes))]
    xyzi, offsets, params, ref = _batch_case(kitti, sizes, steps)
    n = xyzi.shape[0]
    want_idx = (np.searchsorted(offsets, np.arange(n, dtype=np.uint64), side="right") - 1).astype(np.uint32)
    if True:
        # device-resident
        d_in = torch.from_numpy(xyzi).cuda()
        d_out = torch.full((n + 64, 4), 7.0, dtype=torch.float32, device="cuda")  # 64 guard rows behind the batch
        d_idx = torch.full((n + 64,), -1, dtype=torch.int32, device="cuda")
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        st = ctx.deskew_batch_f32(d_in, d_out, offsets, params, d_idx)
        torch.cuda.synchronize()
        assert bool((d_out[n:] == 7.0).all()) and bool((d_idx[n:] == -1).all()), "wrote past the end of the batch"
        got = d_out[:n].cpu().numpy()
        idx = d_idx.cpu().numpy().view(np.uint32)[:n]
        assert st.n_points == n
        assert np.array_equal(idx, want_idx), "per-point frame indices must be bit-exact"
        assert np.array_equal(got[:, 3].view(np.uint32), xyzi[:, 3].view(np.uint32))
        if n:
            assert util.rel_point_error(got[:, :3], ref).max() <= REL_TOL
        # host buffers, no index output
        out = np.empty_like(xyzi)
        ctx.deskew_batch_f32(xyzi, out, offsets, params, None)
        assert np.array_equal(out.view(np.uint32), got.view(np.uint32))


def test_batch_soak_many_random_frames(torch_mod, ctx, kitti):
    """BASELINE.json configs[4] in miniature: hundreds of frames of mixed (incl. zero) sizes, each with its own random
    trajectory and request time, one launch; every point against the oracle, every frame index bit-exact."""
    torch = torch_mod
    xyzi, P0 = kitti
    rng = np.random.default_rng(2024)
    n_frames = 300
    sizes = rng.integers(0, 20001, size=n_frames)
    sizes[rng.integers(0, n_frames, size=12)] = 0
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    n = int(offsets[-1])
    pts = np.ascontiguousarray(xyzi[rng.integers(0, xyzi.shape[0], size=n)])
    params, ref = [], np.empty((n, 3))
    for f in range(n_frames):
        step = np.concatenate([rng.normal(0, 1.5, 3), rng.normal(0, 0.05, 3)])
        A = orc.affine_mul(P0, orc.se3_exp(np.concatenate([rng.normal(0, 30, 3), rng.normal(0, 0.3, 3)])))
        B = orc.affine_mul(A, orc.se3_exp(step))
        treq = T0 + rng.uniform(0, 1) * (T1 - T0)
        params.append(_params(A, B, treq=treq))
        s, e = int(offsets[f]), int(offsets[f + 1])
        if e > s:
            ref[s:e] = _oracle(pts[s:e], A, B, treq=treq)["xyz_f64"]  # FAITHFUL
    d_in = torch.from_numpy(pts).cuda()
    d_out = torch.empty_like(d_in)
    d_idx = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    st = ctx.deskew_batch_f32(d_in, d_out, offsets, params, d_idx)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    want_idx = (np.searchsorted(offsets, np.arange(n, dtype=np.uint64), side="right") - 1).astype(np.uint32)
    assert st.n_points == n and st.variant == capi.TIER_SERIES3
    assert np.array_equal(d_idx.cpu().numpy().view(np.uint32), want_idx)
    assert np.array_equal(got[:, 3].view(np.uint32), pts[:, 3].view(np.uint32))
    assert util.rel_point_error(got[:, :3], ref).max() <= REL_TOL


def test_batch_equals_single_frame_kernel_bitwise(torch_mod, ctx, kitti):
    """A one-frame batch must reduce bit-for-bit to kmc_hip_deskew_f32 (same arithmetic, different plumbing)."""
    xyzi, P1 = kitti
    P1, P2 = _poses(P1, TRAJECTORIES["hard_turn"])
    params = _params(P1, P2)
    single, _ = _run_device(torch_mod, ctx, xyzi, params)
    out = np.empty_like(xyzi)
    ctx.deskew_batch_f32(xyzi, out, np.array([0, xyzi.shape[0]], np.uint64), [params], None)
    assert np.array_equal(out.view(np.uint32), single.view(np.uint32))


# ---- f64 Eigen-layout path (the 2-argument MotionCompensateFrame contract) -------------------------------------------------
def test_f64cols_vs_faithful_oracle(ctx, kitti, kats):
    xyzi, P_mercator = kitti
    xyzi = xyzi[::7]
    n = xyzi.shape[0]
    cloud = np.concatenate([xyzi[:, :3].astype(np.float64), np.ones((n, 1))], axis=1)
    stamps = orc.pseudo_timestamps(cloud, T0, T1)
    P_local = orc.Affine.from_Rt(orc.so3_exp([0.01, -0.02, 0.7]), [12.5, -3.0, 0.4])
    # With Mercator-scale poses (t ~ 6e6 m) the ORACLE's Log carries ~2e-9 m of cancellation noise, i.e.
    # up to ~1.4e-9 relative on this frame's nearest points (1.47 m); with local-frame poses that noise vanishes and the
    # f64 kernel must match to rounding.
    for P1, tol in ((P_mercator, 5e-9), (P_local, REL_TOL_F64)):
        for name in ("straight", "hard_turn", "tumble"):
            A, B = _poses(P1, TRAJECTORIES[name])
            params = _params(A, B)
            rc, nbad, want = orc.motion_compensate_frame(cloud, stamps, T0, A, T1, B, TREQ)
            assert rc == orc.OK
            cols = [np.ascontiguousarray(cloud[:, j]) for j in range(4)]
            outs = [np.empty(n) for _ in range(4)]
            rc2, st = ctx.deskew_f64cols(cols[0], cols[1], cols[2], cols[3], stamps, T0, T1, params, *outs)
            assert rc2 == capi.OK and st.n_out_of_range == 0
            got = np.stack(outs, axis=1)
            assert np.array_equal(got[:, 3], cloud[:, 3])
            assert util.rel_point_error(got[:, :3], want[:, :3]).max() <= tol, (name, tol)
            hoisted = orc.deskew_xyzi_f32(xyzi, T0, A, T1, B, TREQ, mode=orc.HOISTED)["xyz_f64"]
            assert util.rel_point_error(got[:, :3], hoisted).max() <= tol, (name, tol)
    # reference KAT in the f64 layout
    k = kats["motion_compensate_frame"]
    ox = [capi.Oxts(**o) for o in k["oxts"]]
    T_start, T_end = capi.make_frame_poses(ox[0], ox[1], ox[2], k["stamp_start"], k["stamp_end"])
    params = capi.frame_params_from_poses(T_start, T_end, k["stamp_start"], k["stamp_end"], k["requested_time"])
    c = np.array(k["cloud"])
    st3 = orc.pseudo_timestamps(c, k["stamp_start"], k["stamp_end"])
    outs = [np.empty(3) for _ in range(4)]
    ctx.deskew_f64cols(*(np.ascontiguousarray(c[:, j]) for j in range(4)), st3, k["stamp_start"], k["stamp_end"], params, *outs)
    util.assert_float_eq(np.stack(outs, axis=1), np.array(k["expected"]))


@pytest.mark.parametrize("theta", [0.0, 1e-9, 0.02, 0.1099, 0.1101, 0.3, 0.4999, 0.5001, 1.3, 3.0, 7.5])
def test_f64_series_tiers_and_doublings_keep_the_f64_bar(ctx, theta):
    """The f64 kernels' coefficient series (kmc_device_math.hip.h, round 6: power form, 5 terms up to 0.11 rad per scan, 8 beyond, angle
    doublings beyond 0.5 rad) on both sides of every switch-over, against the closed form of lie_algebra.cpp:83-92 evaluated in extended
    precision HERE (numpy longdouble: neither the library nor the oracle): p' = R(s phi) p + J(s phi) s rho."""
    n = 40_001
    rng = np.random.default_rng(int(theta * 1e4) + 3)
    pts = rng.uniform(-60, 60, size=(n, 3))
    pts[:, 2] = rng.uniform(-3, 3, size=n)
    t0, t1 = 100.0, 100.1
    stamps = np.sort(rng.uniform(t0, t1, size=n))
    stamps[0], stamps[-1] = t0, t1  # both ends of the range included
    axis = np.array([0.3, -0.2, 0.93])
    axis /= np.linalg.norm(axis)
    twist = np.concatenate([[1.2, -0.4, 0.05], theta * axis])
    x_req = 0.37
    params = capi.FrameParams.make(twist, x_req)
    cols = [np.ascontiguousarray(pts[:, j]) for j in range(3)]
    outs = [np.empty(n) for _ in range(3)]
    rc, st = ctx.deskew_f64cols(cols[0], cols[1], cols[2], None, stamps, t0, t1, params, outs[0], outs[1], outs[2], None)
    assert rc == capi.OK and st.n_out_of_range == 0
    got = np.stack(outs, axis=1)
    L = np.longdouble
    s = (stamps.astype(L) - L(t0)) / (L(t1) - L(t0)) - L(x_req)
    phi, rho = twist[3:].astype(L), twist[:3].astype(L)
    th = np.abs(s) * L(theta)
    small = th < L(1e-4)
    th_safe = np.where(small, L(1), th)
    A = np.where(small, 1 - th**2 / 6, np.sin(th_safe) / th_safe)
    B = np.where(small, L(0.5) - th**2 / 24, (1 - np.cos(th_safe)) / th_safe**2)
    C = np.where(small, L(1) / 6 - th**2 / 120, (th_safe - np.sin(th_safe)) / th_safe**3)
    P = pts.astype(L)
    w = s[:, None] * phi[None, :]                  # s phi
    v = s[:, None] * rho[None, :]                  # s rho
    wxp = np.cross(w, P)
    wxwxp = np.cross(w, wxp)
    wxv = np.cross(w, v)
    wxwxv = np.cross(w, wxv)
    want = P + A[:, None] * wxp + B[:, None] * wxwxp + v + B[:, None] * wxv + C[:, None] * wxwxv
    err = np.linalg.norm((got.astype(L) - want).astype(np.float64), axis=1) / np.maximum(np.linalg.norm(want.astype(np.float64), axis=1), 1e-3)
    assert err.max() <= 2e-14, (theta, float(err.max()))  # rounding of ~40 f64 operations on |p| <= 85 m; the parity bar itself is REL_TOL_F64 = 1e-11


def test_f64cols_out_of_range_stamps_are_reported(ctx, kitti):
    """trajectory_interpolation.cpp:32 aborts; the ABI returns KMC_ERR_TIME_OUT_OF_RANGE and counts the offenders."""
    xyzi, P1 = kitti
    A, B = _poses(P1, TRAJECTORIES["gentle_turn"])
    params = _params(A, B)
    n = 1000
    cloud = np.concatenate([xyzi[:n, :3].astype(np.float64), np.ones((n, 1))], axis=1)
    stamps = orc.pseudo_timestamps(cloud, T0, T1)
    stamps[[3, 500, 999]] = [T0 - 1e-6, T1 + 1e-6, 0.0]
    outs = [np.empty(n) for _ in range(4)]
    rc, st = ctx.deskew_f64cols(*(np.ascontiguousarray(cloud[:, j]) for j in range(4)), stamps, T0, T1, params, *outs,
                                raise_on_range=False)
    assert rc == capi.ERR_TIME_OUT_OF_RANGE and st.n_out_of_range == 3
    assert np.isnan(outs[0][[3, 500, 999]]).all() and np.isfinite(np.delete(outs[0], [3, 500, 999])).all()
    rc_o, nbad_o, _ = orc.motion_compensate_frame(cloud, stamps, T0, A, T1, B, TREQ)
    assert rc_o == orc.ERR_TIME_OUT_OF_RANGE and nbad_o == 3


def test_f64cols_in_place_on_page_locked_containers_same_bits(ctx, kitti):
    """Round 3: when every column handed to kmc_hip_deskew_f64cols(KMC_MEM_HOST) lies in the C-ABI's page-locked pool -- what the
    C++ drop-in's Pointcloud / VectorXd are made of -- the kernel works on the caller's memory in place (one launch, no staging
    copies).  Same kernel, same bits as the staged route on ordinary numpy arrays; out-of-range stamps are still counted (through
    the page-locked flag word instead of a device-to-host copy); foreign pointers keep the staged route."""
    xyzi, P1 = kitti
    n = xyzi.shape[0]
    A, B = _poses(P1, TRAJECTORIES["gentle_turn"])
    params = _params(A, B)
    cloud = np.concatenate([xyzi[:, :3].astype(np.float64), np.ones((n, 1))], axis=1)
    stamps = orc.pseudo_timestamps(cloud, T0, T1)
    cols = [np.ascontiguousarray(cloud[:, j]) for j in range(4)]
    staged = [np.empty(n) for _ in range(4)]
    rc, st = ctx.deskew_f64cols(cols[0], cols[1], cols[2], cols[3], stamps, T0, T1, params, *staged)
    assert rc == capi.OK and st.variant == 5 and not capi.host_pool_owns(cols[0])
    pin = capi.PooledArray((4, n))      # one column-major N x 4 block, like Eigen::MatrixX4d
    pst = capi.PooledArray((n,))
    pout = capi.PooledArray((4, n))
    try:
        pin.a[:] = np.stack(cols)
        pst.a[:] = stamps
        pout.a[:] = -7.0
        assert capi.host_pool_owns(pin.a) and capi.host_pool_owns(pout.a[2])
        rc, st = ctx.deskew_f64cols(pin.a[0], pin.a[1], pin.a[2], pin.a[3], pst.a, T0, T1, params, pout.a[0], pout.a[1], pout.a[2], pout.a[3])
        assert rc == capi.OK and st.n_out_of_range == 0 and st.n_launches == 1
        for j in range(4):
            assert np.array_equal(pout.a[j].view(np.uint64), staged[j].view(np.uint64)), j
        want = orc.motion_compensate_frame(cloud, stamps, T0, A, T1, B, TREQ)[2]
        assert util.rel_point_error(pout.a[:3].T, want[:, :3]).max() <= 5e-9
        # the reference's assert (trajectory_interpolation.cpp:32) on the in-place route
        pst.a[[5, n // 2]] = [T0 - 1e-6, T1 + 1.0]
        rc, st = ctx.deskew_f64cols(pin.a[0], pin.a[1], pin.a[2], pin.a[3], pst.a, T0, T1, params, pout.a[0], pout.a[1], pout.a[2], pout.a[3],
                                    raise_on_range=False)
        assert rc == capi.ERR_TIME_OUT_OF_RANGE and st.n_out_of_range == 2 and np.isnan(pout.a[0][[5, n // 2]]).all()
        pst.a[:] = stamps
        rc, st = ctx.deskew_f64cols(pin.a[0], pin.a[1], pin.a[2], pin.a[3], pst.a, T0, T1, params, pout.a[0], pout.a[1], pout.a[2], pout.a[3])
        assert rc == capi.OK and st.n_out_of_range == 0, "the flag word must be cleared between calls"
        # the call in two halves (what the C++ drop-in uses to fill the homogeneous column while the kernel runs): same bits, the
        # verdict arrives with _end; a second _begin on device-addressable buffers queues up behind the first (round 4), a plain call
        # in between is refused; staged buffers complete inside _begin
        pout.a[:] = -7.0
        ctx.deskew_f64cols_begin(pin.a[0], pin.a[1], pin.a[2], None, pst.a, T0, T1, params, pout.a[0], pout.a[1], pout.a[2], None)
        ctx.deskew_f64cols_begin(pin.a[0], pin.a[1], pin.a[2], None, pst.a, T0, T1, params, pout.a[0], pout.a[1], pout.a[2], None)
        with pytest.raises(capi.KmcError):
            ctx.deskew_f64cols(pin.a[0], pin.a[1], pin.a[2], pin.a[3], pst.a, T0, T1, params, pout.a[0], pout.a[1], pout.a[2], pout.a[3])
        rc, st = ctx.deskew_f64cols_end()
        assert rc == capi.OK and st.n_points == 2 * n and st.n_launches == 2 and st.n_out_of_range == 0
        for j in range(3):
            assert np.array_equal(pout.a[j].view(np.uint64), staged[j].view(np.uint64)), j
        assert (pout.a[3] == -7.0).all(), "no w column asked for: the output's column is the caller's"
        with pytest.raises(capi.KmcError):
            ctx.deskew_f64cols_end()  # nothing pending
        bad_stamps = stamps.copy()
        bad_stamps[9] = T1 + 5.0
        halves = [np.empty(n) for _ in range(4)]
        ctx.deskew_f64cols_begin(cols[0], cols[1], cols[2], cols[3], bad_stamps, T0, T1, params, *halves)  # ordinary memory: staged
        rc, st = ctx.deskew_f64cols_end(raise_on_range=False)
        assert rc == capi.ERR_TIME_OUT_OF_RANGE and st.n_out_of_range == 1 and np.isnan(halves[0][9])
        # pseudo stamps in place
        ctx.pseudo_timestamps_f64(pin.a[0], pin.a[1], T0, T1, pst.a)
        ref = np.empty(n)
        ctx.pseudo_timestamps_f64(cols[0], cols[1], T0, T1, ref)
        assert np.array_equal(pst.a.view(np.uint64), ref.view(np.uint64))
    finally:
        pin.close(); pst.close(); pout.close()


@pytest.mark.parametrize("n", [1, 63, 64, 2047, 2048, 2049, 123_397, 1_000_003])
def test_f32_in_place_on_page_locked_buffers_same_bits(torch_mod, ctx, kitti, n):
    """KITTI-layout clouds in the C-ABI's page-locked pool (what KittiPclLoader::LoadRaw returns): kmc_hip_deskew_f32(KMC_MEM_HOST)
    recognises them and runs ONE kernel in place over the link (persistent waves, next tile's load in flight while the current one
    is stored).  Same arithmetic, near-origin guard included: the same bits as the device-resident kernel, for every tier, ragged
    sizes, and nothing written past the end.  (Below 2048 points the staged route is taken: same bits as well.)"""
    torch = torch_mod
    xyzi, P1 = kitti
    pts = capi.synth_points_host(n, 77) if n > xyzi.shape[0] else np.ascontiguousarray(xyzi[:n])
    pin, pout = capi.PooledArray((n + 16, 4), np.float32), capi.PooledArray((n + 16, 4), np.float32)
    try:
        pin.a[:n] = pts
        for name in ("gentle_turn", "spin", "tumble"):
            params = _params(*_poses(P1, TRAJECTORIES[name]))
            want, st_dev = _run_device(torch, ctx, pts, params)
            pout.a[:] = -7.0
            st = ctx.deskew_f32(pin.a[:n], pout.a[:n], params)
            assert st.variant == st_dev.variant and st.n_points == n
            assert np.array_equal(pout.a[:n].view(np.uint32), want.view(np.uint32)), name
            assert (pout.a[n:] == -7.0).all(), "wrote past the end"
        # a caller-supplied twist beyond pi (any-angle tier) and a point built on the near-origin cancellation
        raw = capi.FrameParams.make([2.0, 0.1, 0.0, 0.3, -0.2, 5.0], 0.5)
        pin.a[0] = [-1.0e-3, 0.0, 0.0, 0.25]  # frac = 1.0 -> s = 0.5: ends up a millimetre from where the translation takes it
        want, _ = _run_device(torch, ctx, np.ascontiguousarray(pin.a[:n]), raw)
        ctx.deskew_f32(pin.a[:n], pout.a[:n], raw)
        assert np.array_equal(pout.a[:n].view(np.uint32), want.view(np.uint32))
    finally:
        pin.close(); pout.close()


def test_f64cols_host_route_large_frame_is_pipelined_and_identical(torch_mod, ctx):
    """Host buffers of >= 2^20 points take the duplex chunk pipeline (upload of chunk k+1, kernel, download of chunk k in
    parallel, a helper thread for the downloads): same bits as the device-resident call on the same data, a column of ones is
    neither uploaded nor downloaded, a general w column is honoured, out-of-range stamps are still counted and reported."""
    torch = torch_mod
    n = 2_500_017  # 3 chunks, the last one ragged and odd
    xyzi = capi.synth_points_host(n, 31337)
    cloud = np.concatenate([xyzi[:, :3].astype(np.float64), np.ones((n, 1))], axis=1)
    cols = np.ascontiguousarray(cloud.T)  # one column-major block like Eigen: rows = the four columns
    stamps = orc.pseudo_timestamps(cloud, T0, T1)
    P1 = orc.Affine.from_Rt(orc.so3_exp([0.01, -0.02, 0.7]), [12.5, -3.0, 0.4])
    A, B = _poses(P1, TRAJECTORIES["hard_turn"])
    params = _params(A, B)
    # reference: the device-resident call
    d_cols = torch.from_numpy(cols).cuda()
    d_st = torch.from_numpy(stamps).cuda()
    d_out = torch.empty_like(d_cols)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.deskew_f64cols(d_cols[0], d_cols[1], d_cols[2], d_cols[3], d_st, T0, T1, params, d_out[0], d_out[1], d_out[2], d_out[3])
    torch.cuda.synchronize()
    want = d_out.cpu().numpy()
    out = np.full((4, n), np.nan)
    rc, st = ctx.deskew_f64cols(cols[0], cols[1], cols[2], cols[3], stamps, T0, T1, params, out[0], out[1], out[2], out[3])
    assert rc == capi.OK and st.n_launches == 3 and st.n_out_of_range == 0
    assert np.array_equal(out.view(np.uint64), want.view(np.uint64))
    assert np.all(out[3] == 1.0)
    sel = slice(0, n, 97)
    rc_o, nbad, ref = orc.motion_compensate_frame(cloud[sel], stamps[sel], T0, A, T1, B, TREQ)
    assert rc_o == orc.OK
    assert util.rel_point_error(out[:3, sel].T, ref[:, :3]).max() <= REL_TOL_F64
    # a general homogeneous column (not all ones) is uploaded, used (translation scaled by w) and returned
    cols2 = cols.copy()
    cols2[3, ::5] = 0.5
    d_cols2 = torch.from_numpy(cols2).cuda()
    ctx.deskew_f64cols(d_cols2[0], d_cols2[1], d_cols2[2], d_cols2[3], d_st, T0, T1, params, d_out[0], d_out[1], d_out[2], d_out[3])
    torch.cuda.synchronize()
    out2 = np.full((4, n), np.nan)
    ctx.deskew_f64cols(cols2[0], cols2[1], cols2[2], cols2[3], stamps, T0, T1, params, out2[0], out2[1], out2[2], out2[3])
    assert np.array_equal(out2.view(np.uint64), d_out.cpu().numpy().view(np.uint64)) and np.array_equal(out2[3], cols2[3])
    assert not np.array_equal(out2[0], out[0])
    # out-of-range stamps in different chunks
    bad = stamps.copy()
    bad[[5, 1_200_000, n - 1]] = [T0 - 1e-6, T1 + 1e-6, 0.0]
    rc, st = ctx.deskew_f64cols(cols[0], cols[1], cols[2], cols[3], bad, T0, T1, params, out[0], out[1], out[2], out[3], raise_on_range=False)
    assert rc == capi.ERR_TIME_OUT_OF_RANGE and st.n_out_of_range == 3
    assert np.isnan(out[0][[5, 1_200_000, n - 1]]).all()


def test_pseudo_timestamps_f64(ctx, kitti, kats):
    xyzi, _ = kitti
    x = xyzi[:, 0].astype(np.float64)
    y = xyzi[:, 1].astype(np.float64)
    out = np.empty_like(x)
    ctx.pseudo_timestamps_f64(x, y, T0, T1, out)
    cloud = np.stack([x, y, np.zeros_like(x), np.ones_like(x)], axis=1)
    want = orc.pseudo_timestamps(cloud, T0, T1)
    ulp = np.abs(out - want) / np.spacing(want)
    assert ulp.max() <= 4, f"f64 stamps differ by {ulp.max()} ulp (ulp = {np.spacing(T0):.1e} s)"
    k = kats["timestamp_mocking"]
    c = np.array(k["cloud"])
    o3 = np.empty(3)
    ctx.pseudo_timestamps_f64(np.ascontiguousarray(c[:, 0]), np.ascontiguousarray(c[:, 1]), k["scan_start"], k["scan_end"], o3)
    util.assert_float_eq(o3, k["expected_stamp"])


# ---- full-size properties (BASELINE.json configs[3]: 10 M points per frame) --------------------------------------------
def test_full_size_properties_10M(torch_mod, ctx):
    torch = torch_mod
    n = 10_000_000
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    d_in = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    ctx.synth_points(d_in, n, 0x4B4D43 + 3)
    d_out = torch.empty_like(d_in)
    # (1) zero twist is the identity (exact float equality; -0.0 may come back as +0.0)
    ctx.deskew_f32(d_in, d_out, capi.FrameParams.make([0, 0, 0, 0, 0, 0], 0.5))
    torch.cuda.synchronize()
    assert torch.equal(d_in, d_out)
    # (2) pure rotation preserves the range of every point; intensity untouched
    ctx.deskew_f32(d_in, d_out, capi.FrameParams.make([0, 0, 0, 0.01, -0.02, 0.2], 0.37))
    torch.cuda.synchronize()
    r_in = d_in[:, :3].double().norm(dim=1)
    r_out = d_out[:, :3].double().norm(dim=1)
    assert ((r_out - r_in).abs() / r_in).max().item() < 5e-7
    assert torch.equal(d_in[:, 3].view(torch.int32), d_out[:, 3].view(torch.int32))
    # (3) pure translation moves every point along rho by s in [-x_req, 1 - x_req], monotone in the scan fraction
    rho = torch.tensor([1.7, -0.4, 0.05], dtype=torch.float64, device="cuda")
    x_req = 0.37
    ctx.deskew_f32(d_in, d_out, capi.FrameParams.make([*rho.tolist(), 0, 0, 0], x_req))
    torch.cuda.synchronize()
    d = d_out[:, :3].double() - d_in[:, :3].double()
    s = (d @ rho) / (rho @ rho)
    perp = (d - s[:, None] * rho[None, :]).norm(dim=1)
    assert perp.max().item() < 2e-5  # f32 ulp at 80 m is 7.6e-6
    assert s.min().item() >= -x_req - 1e-5 and s.max().item() <= 1 - x_req + 1e-5
    frac = (np.pi - torch.atan2(d_in[:, 1].double(), d_in[:, 0].double())) / (2 * np.pi)
    assert (s - (frac - x_req)).abs().max().item() < 2e-5
    # (4) the whole frame agrees with the oracle
    P1 = orc.oxts_to_pose(orc.oxts(T0, 49.011212804408, 8.4228850417969, 112.8, 0.022, 1e-5, -1.22))
    A, B = _poses(P1, TRAJECTORIES["hard_turn"])
    params = _params(A, B)
    ctx.deskew_f32(d_in, d_out, params)
    torch.cuda.synchronize()
    xyzi = d_in.cpu().numpy()
    got = d_out.cpu().numpy()
    _check(got, xyzi, _oracle(xyzi, A, B))  # all 10 M points against the FAITHFUL oracle (the reference's per-point Log / Exp sequence)
    # (5) moving the requested time moves the WHOLE compensated cloud by one rigid transform: with f = Log(T_start^-1 T_end),
    #     p'(x_req) = Exp((x_i - x_req) f) p, and exponentials of the same twist commute, so p'(b) = Exp((a - b) f) p'(a) for every
    #     point -- all 10 M of them, against one 3x4 matrix from the oracle's lie::Exp (RelativePoseBetweenTimes,
    #     trajectory_interpolation.cpp:43-45, is exactly this statement)
    twist = np.array([1.9, -0.3, 0.08, 0.01, -0.03, 0.25])
    xa, xb = 0.2, 0.85
    d_b = torch.empty_like(d_in)
    ctx.deskew_f32(d_in, d_out, capi.FrameParams.make(twist, xa))
    ctx.deskew_f32(d_in, d_b, capi.FrameParams.make(twist, xb))
    torch.cuda.synchronize()
    T_ab = orc.se3_exp(list((xa - xb) * twist))
    R = torch.tensor(np.array(list(T_ab.R)).reshape(3, 3), dtype=torch.float64, device="cuda")
    t = torch.tensor(list(T_ab.t), dtype=torch.float64, device="cuda")
    moved = d_out[:, :3].double() @ R.T + t
    err = (moved - d_b[:, :3].double()).norm(dim=1) / d_b[:, :3].double().norm(dim=1).clamp_min(1e-3)
    assert err.max().item() < 2e-6, err.max().item()
    assert torch.equal(d_b[:, 3].view(torch.int32), d_in[:, 3].view(torch.int32))


# ---- scale and speed guards ------------------------------------------------------------------------------------------
def test_beyond_4GiB_buffers_use_64bit_indexing(torch_mod, ctx):
    """300 M points = 4.8 GB per buffer (> 2^32 bytes): tiles past the 4 GiB mark must be addressed correctly by the
    single-frame and the batched kernel (per-tile descriptors carry a 64-bit base)."""
    torch = torch_mod
    n = 300_000_000
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    d_in = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    ctx.synth_points(d_in, n, 4242)
    d_out = torch.zeros_like(d_in)
    rho = [1.7, -0.4, 0.05]
    params = capi.FrameParams.make([*rho, 0, 0, 0], 0.25)
    ctx.deskew_f32(d_in, d_out, params)
    torch.cuda.synchronize()

    def check(out):
        for lo in (0, (1 << 28) - 50_000, n - 100_000):  # start, across the 4 GiB byte mark, end
            a = d_in[lo:lo + 100_000].double()
            b = out[lo:lo + 100_000].double()
            frac = (np.pi - torch.atan2(a[:, 1], a[:, 0])) / (2 * np.pi)
            want = a[:, :3] + (frac - 0.25)[:, None] * torch.tensor(rho, dtype=torch.float64, device="cuda")[None, :]
            assert (b[:, :3] - want).abs().max().item() < 2e-5
            assert torch.equal(a[:, 3], b[:, 3])

    check(d_out)
    d_out.zero_()
    offsets = np.array([0, 100_000_000, 100_000_000, 268_435_456 + 7, n], dtype=np.uint64)  # a boundary right after 2^28 points
    d_idx = torch.empty(n, dtype=torch.int32, device="cuda")
    ctx.deskew_batch_f32(d_in, d_out, offsets, [params] * 4, d_idx)
    torch.cuda.synchronize()
    check(d_out)
    for pos, want in ((0, 0), (99_999_999, 0), (100_000_000, 2), (268_435_462, 2), (268_435_463, 3), (n - 1, 3)):
        assert int(d_idx[pos].item()) == want, pos


def test_beyond_2_32_work_items_a_second_launch_takes_over(torch_mod, ctx):
    """2^32 + 64 017 points = 68.7 GB per buffer (MI355X holds 288 GB): more 64-point tiles than a dispatch can carry workgroups
    (the packet's grid is 32 bits of work-items; the runtime wraps a larger grid modulo 2^32 and reports success,
    tools/grid_probe.hip), so the library cuts the work into launches of at most 2^26 - 1 tiles, each told its first tile.  Single-frame
    and batched kernels; checked on slices at the start, either side of the 2^32-work-item mark and at the end."""
    torch = torch_mod
    n = (1 << 32) + 64_017
    need = 2 * n * 16 + n * 4 + (2 << 30)
    free, _ = torch.cuda.mem_get_info()
    if free < need:
        pytest.skip(f"needs {need >> 30} GiB of free HBM, the box has {free >> 30}")
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    d_in = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    ctx.synth_points(d_in, n, 777)
    d_out = torch.zeros_like(d_in)
    rho = [1.7, -0.4, 0.05]
    params = capi.FrameParams.make([*rho, 0, 0, 0], 0.25)
    st = ctx.deskew_f32(d_in, d_out, params)
    torch.cuda.synchronize()
    assert st.n_points == n
    mark = ((0xFFFFFFFF // 64) * 64)  # first point of the first tile of the second launch

    def check(out):
        for lo in (0, mark - 50_000, mark + 64 * 1000 - 50_000, n - 100_000):
            a = d_in[lo:lo + 100_000].double()
            b = out[lo:lo + 100_000].double()
            frac = (np.pi - torch.atan2(a[:, 1], a[:, 0])) / (2 * np.pi)
            want = a[:, :3] + (frac - 0.25)[:, None] * torch.tensor(rho, dtype=torch.float64, device="cuda")[None, :]
            assert (b[:, :3] - want).abs().max().item() < 2e-5, lo
            assert torch.equal(a[:, 3], b[:, 3]), lo

    check(d_out)
    d_out.zero_()
    offsets = np.array([0, 3_000_000_000, mark + 5, n], dtype=np.uint64)
    d_idx = torch.empty(n, dtype=torch.int32, device="cuda")
    ctx.deskew_batch_f32(d_in, d_out, offsets, [params] * 3, d_idx)
    torch.cuda.synchronize()
    check(d_out)
    for pos, want in ((0, 0), (2_999_999_999, 0), (3_000_000_000, 1), (mark + 4, 1), (mark + 5, 2), (n - 1, 2)):
        assert int(d_idx[pos].item()) == want, pos
    del d_in, d_out, d_idx
    torch.cuda.empty_cache()


def test_throughput_guard_batched_kernel(torch_mod, ctx):
    """Regression guard, not a benchmark: the batched kernel on 64 x 1 M points must stay above 5.0 TB/s
    (bench.py measures 6.8-6.9 TB/s on 256 M points; a persistent-loop or un-hinted variant would land at 5.2-6.4 and a
    lost vectorisation far below; the margin absorbs a noisy or throttled box)."""
    torch = torch_mod
    F, per = 64, 1_000_000
    n = F * per
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    d_in = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    ctx.synth_points(d_in, n, 99)
    d_out = torch.empty_like(d_in)
    offsets = np.arange(F + 1, dtype=np.uint64) * per
    params = capi.params_array([capi.FrameParams.make([1.0, 0.01 * f, 0, 0.001, -0.002, 0.01 + 0.0005 * f], 0.5) for f in range(F)])
    for _ in range(10):
        ctx.deskew_batch_f32(d_in, d_out, offsets, params, None)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        ctx.timer_begin()
        for _ in range(30):
            ctx.deskew_batch_f32(d_in, d_out, offsets, params, None)
        best = min(best, ctx.timer_end() / 30)
    gbps = 32.0 * n / (best * 1e-3) / 1e9
    print(f"batched kernel: {best * 1e3:.1f} us per 64 M points = {gbps:.0f} GB/s")
    assert gbps > 5000, gbps


def test_config3_drive_twin_with_real_cadence(torch_mod, ctx, golden_dir):
    """BASELINE.json configs[2] twin: 108 frames at the REAL cadence of the shipped
    timestamps*.txt, OXTS packets along a constant-twist track (13 m/s, 0.3 rad/s yaw rate, small roll/pitch rates), every
    interior frame through the whole host pre-step (OxtsToPose -> InterpolateTrajectory -> MakeFrame -> Log) and ONE batched
    launch; checked frame by frame against the oracle's own MakeFrame + faithful loop."""
    torch = torch_mod
    run = os.path.join(golden_dir, "kitti_2011_09_26_drive_0005")
    n_frames = 108
    t_start, t_mid, t_end, oxts = util.real_cadence_drive(run, n_frames)
    rng = np.random.default_rng(11)
    xyzi_all = util.load_velodyne_bin(run, 0)
    frames, params, refs, sizes = [], [], [], []
    for i in range(1, n_frames - 1):  # handlers.cpp:55: frames 1 .. n-2
        n = int(np.clip(rng.normal(121_000, 3_000), 90_000, 140_000))  # KITTI's frame size: ~13 M points in all
        pts = np.ascontiguousarray(xyzi_all[rng.integers(0, xyzi_all.shape[0], size=n)])
        co = [capi.Oxts(**oxts[i + d]) for d in (-1, 0, 1)]
        T_s, T_e = capi.make_frame_poses(co[0], co[1], co[2], t_start[i], t_end[i])
        params.append(capi.frame_params_from_poses(T_s, T_e, t_start[i], t_end[i], t_mid[i]))  # requested = stamp_middle, handlers.cpp:59
        oo = [orc.oxts(**oxts[i + d]) for d in (-1, 0, 1)]
        rc, A, B = orc.make_frame_poses(oo[0], oo[1], oo[2], t_start[i], t_end[i])
        assert rc == orc.OK
        r = orc.deskew_xyzi_f32(pts, t_start[i], A, t_end[i], B, t_mid[i], mode=orc.FAITHFUL)  # the reference's op sequence, all cores
        assert r["rc"] == orc.OK
        frames.append(pts)
        refs.append(r["xyz_f64"])
        sizes.append(n)
    xyzi = np.concatenate(frames)
    ref = np.concatenate(refs)
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    d_in = torch.from_numpy(xyzi).cuda()
    d_out = torch.empty_like(d_in)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    st = ctx.deskew_batch_f32(d_in, d_out, offsets, params, None)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    assert st.n_points == xyzi.shape[0] and st.variant == capi.TIER_SERIES3
    err = util.rel_point_error(got[:, :3], ref)
    assert err.max() <= REL_TOL, err.max()
    assert np.array_equal(got[:, 3], xyzi[:, 3])
    # the vehicle really moves: the correction is far above the tolerance (so the test cannot pass vacuously)
    moved = np.linalg.norm(got[:, :3].astype(np.float64) - xyzi[:, :3], axis=1)
    assert moved.max() > 0.5


def test_batch_frame_index_fuzz(torch_mod, ctx):
    """Randomised offsets -- empty frames, runs of empties, 1-point frames, tile-aligned and chunk-aligned frames, frames
    far larger than a 16384-point chunk -- through the coarse table / split / search / LDS-walk logic: the per-point frame
    index must equal numpy.searchsorted on the offsets, bit for bit, and a zero twist must hand every point back."""
    torch = torch_mod
    rng = np.random.default_rng(20240928)
    ident = capi.FrameParams.make([0, 0, 0, 0, 0, 0], 0.5)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    pool = torch.empty((400_000, 4), dtype=torch.float32, device="cuda")
    ctx.synth_points(pool, pool.shape[0], 31337)
    for case in range(120):
        n_frames = int(rng.integers(1, 400))
        kind = rng.integers(0, 6, size=n_frames)
        sizes = np.where(kind == 0, 0,
                 np.where(kind == 1, rng.integers(1, 6, size=n_frames),
                 np.where(kind == 2, 64 * rng.integers(1, 40, size=n_frames),
                 np.where(kind == 3, 16384 * rng.integers(1, 3, size=n_frames),
                 np.where(kind == 4, rng.integers(1, 3000, size=n_frames), rng.integers(10_000, 60_000, size=n_frames))))))
        while sizes.sum() > pool.shape[0]:
            sizes[np.argmax(sizes)] //= 2
        offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
        n = int(offsets[-1])
        if n == 0:
            continue
        d_out = torch.full((n + 64, 4), 7.0, dtype=torch.float32, device="cuda")
        d_idx = torch.full((n + 64,), -1, dtype=torch.int32, device="cuda")
        ctx.deskew_batch_f32(pool[:n], d_out, offsets, [ident] * n_frames, d_idx)
        torch.cuda.synchronize()
        want = (np.searchsorted(offsets, np.arange(n, dtype=np.uint64), side="right") - 1).astype(np.int32)
        got = d_idx[:n].cpu().numpy()
        assert np.array_equal(got, want), (case, n_frames, int(np.flatnonzero(got != want)[0]))
        assert torch.equal(d_out[:n], pool[:n]), case
        assert bool((d_out[n:] == 7.0).all()) and bool((d_idx[n:] == -1).all()), case


def test_single_frame_entry_point_is_graph_capturable(ctx, torch_mod):
    """KMC_MEM_DEVICE calls only enqueue work on the caller's stream, so a sequence of them can be captured once into a HIP
    graph (torch.cuda.CUDAGraph on ROCm) and replayed; the replay writes the same bits (tools/measure_graph.py times it)."""
    torch = torch_mod
    nf, per = 16, 10_000
    pts = capi.synth_points_host(nf * per, 321)
    d_in = torch.from_numpy(pts).cuda()
    d_out = torch.zeros_like(d_in)
    params = [capi.FrameParams.make([1.0 + 0.01 * f, 0.02, 0.0, 0.001, 0.0, 0.03], 0.25) for f in range(nf)]

    def launches():
        for f in range(nf):
            ctx.deskew_f32(d_in[f * per:(f + 1) * per], d_out[f * per:(f + 1) * per], params[f], n=per)

    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    launches()
    torch.cuda.synchronize()
    ref = d_out.clone()
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.stream(side):
            ctx.set_stream(side.cuda_stream)
            launches()
            side.synchronize()
            with torch.cuda.graph(graph, stream=side):
                launches()
        d_out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(d_out.view(torch.int32), ref.view(torch.int32))
    finally:
        ctx.set_stream(None)


def test_small_batches_travel_in_kernel_arguments_same_bits(torch_mod, ctx, monkeypatch):
    """Round 3: batches of at most 16 device-resident frames carry their tables in the kernel arguments (no table upload, no host
    wait).  Same kernel body, same records -> the SAME bits and the same per-point frame indices as the device-table route,
    which the same frames take when 17 empty frames are appended to the batch (more than the kernel arguments hold) -- for ragged,
    empty, tile-straddling and huge frames (the inline coarse table grows its chunk size with the batch) and for every tier."""
    torch = torch_mod
    rng = np.random.default_rng(77)
    ident = capi.FrameParams.make([0, 0, 0, 0, 0, 0], 0.5)
    try:
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        pool = torch.empty((6_000_000, 4), dtype=torch.float32, device="cuda")
        ctx.synth_points(pool, pool.shape[0], 4242)
        for case in range(40):
            nf = int(rng.integers(1, 17))
            kind = rng.integers(0, 5, size=nf)
            sizes = np.where(kind == 0, 0, np.where(kind == 1, rng.integers(1, 200, size=nf), np.where(kind == 2, 64 * rng.integers(1, 300, size=nf),
                             np.where(kind == 3, rng.integers(1000, 40_000, size=nf), rng.integers(100_000, 370_000, size=nf)))))
            offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
            n = int(offsets[-1])
            if n == 0:
                continue
            yaw = [0.03, 0.6, 2.0, 5.0][case % 4]  # series3 / series5 / wide / any-angle
            params = [capi.FrameParams.make([1.0 + 0.1 * f, 0.02, -0.01, 0.001 * f, -0.002, yaw * (1 - 0.02 * f)], float(rng.uniform(0, 1))) for f in range(nf)]
            shift = int(rng.integers(0, 64))  # any 16-byte offset of the sub-range
            outs, idxs, sts = [], [], []
            for padded in (False, True):
                d_out = torch.full((n + 128, 4), 7.0, dtype=torch.float32, device="cuda")
                d_idx = torch.full((n + 128,), -1, dtype=torch.int32, device="cuda")
                offs = np.concatenate([offsets, np.full(17, n, dtype=np.uint64)]) if padded else offsets
                prms = params + [ident] * 17 if padded else params
                sts.append(ctx.deskew_batch_f32(pool[shift:shift + n], d_out[shift:shift + n], offs, prms, d_idx[shift:shift + n]))
                outs.append(d_out)
                idxs.append(d_idx)
            torch.cuda.synchronize()
            assert sts[0].variant == sts[1].variant == case % 4
            assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32)), case
            assert torch.equal(idxs[0], idxs[1]), case
            want = (np.searchsorted(offsets, np.arange(n, dtype=np.uint64), side="right") - 1).astype(np.int32)
            assert np.array_equal(idxs[0][shift:shift + n].cpu().numpy(), want), case
            assert bool((outs[0][:shift] == 7.0).all()) and bool((outs[0][shift + n:] == 7.0).all()), case
    finally:
        ctx.force_tier(-1)


def test_batched_entry_point_is_graph_capturable(ctx, torch_mod):
    """A batch of at most 16 device-resident frames only enqueues ONE launch on the caller's stream (its tables are kernel arguments):
    it can be captured into a HIP graph and replayed; the replay writes the same bits.  A larger batch needs a table upload and
    says so instead of silently recording a launch that would read a recycled table."""
    torch = torch_mod
    nf, per = 16, 50_000
    pts = capi.synth_points_host(nf * per, 987)
    d_in = torch.from_numpy(pts).cuda()
    d_out = torch.zeros_like(d_in)
    d_idx = torch.zeros((nf * per,), dtype=torch.int32, device="cuda")
    params = capi.params_array([capi.FrameParams.make([1.0 + 0.01 * f, 0.02, 0.0, 0.001, 0.0, 0.03], 0.25) for f in range(nf)])
    offsets = np.arange(nf + 1, dtype=np.uint64) * per
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.deskew_batch_f32(d_in, d_out, offsets, params, d_idx)
    torch.cuda.synchronize()
    ref, ref_idx = d_out.clone(), d_idx.clone()
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.stream(side):
            ctx.set_stream(side.cuda_stream)
            ctx.deskew_batch_f32(d_in, d_out, offsets, params, d_idx)
            side.synchronize()
            with torch.cuda.graph(graph, stream=side):
                ctx.deskew_batch_f32(d_in, d_out, offsets, params, d_idx)
                offs17 = np.arange(18, dtype=np.uint64) * 1000
                with pytest.raises(capi.KmcError) as ei:
                    ctx.deskew_batch_f32(d_in[:17000], d_out[:17000], offs17, [params[0]] * 17, None)
                assert ei.value.status == capi.ERR_INVALID_ARG
        d_out.zero_()
        d_idx.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(d_out.view(torch.int32), ref.view(torch.int32)) and torch.equal(d_idx, ref_idx)
    finally:
        ctx.set_stream(None)


@pytest.mark.parametrize("shift_in,shift_out", [(1, 1), (7, 3), (37, 37), (63, 0), (0, 63), (5, 64 + 9)])
def test_device_buffers_at_any_16_byte_offset(ctx, torch_mod, kitti, shift_in, shift_out):
    """The kernels cut their tiles on 1 KiB lines of the OUTPUT whatever 16-byte-aligned pointers the caller passes (sub-ranges
    of a resident buffer): same bits as an aligned call, nothing written in front of or behind the range -- for the
    single-frame, batched, trajectory and batched-trajectory entry points and their index outputs."""
    torch = torch_mod
    xyzi, P1 = kitti
    n = 100_000 + 11
    pts = np.ascontiguousarray(xyzi[:n])
    guard = 128
    d_in_big = torch.zeros((n + 2 * guard, 4), dtype=torch.float32, device="cuda")
    d_in = d_in_big[shift_in:shift_in + n]
    d_in.copy_(torch.from_numpy(pts))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    def fresh():
        big = torch.full((n + 2 * guard, 4), 7.0, dtype=torch.float32, device="cuda")
        idx = torch.full((n + 2 * guard,), 77, dtype=torch.int32, device="cuda")
        idx2 = torch.full((n + 2 * guard,), 77, dtype=torch.int32, device="cuda")
        return big, idx, idx2

    def check(big, ref_out, idxs=()):
        torch.cuda.synchronize()
        got = big[shift_out:shift_out + n].cpu().numpy()
        assert np.array_equal(got.view(np.uint32), ref_out.view(np.uint32))
        assert bool((big[:shift_out] == 7.0).all()) and bool((big[shift_out + n:] == 7.0).all()), "wrote outside the range"
        for idx, ref_idx, s in idxs:
            assert np.array_equal(idx[s:s + n].cpu().numpy().view(np.uint32), ref_idx)
            assert bool((idx[:s] == 77).all()) and bool((idx[s + n:] == 77).all()), "index written outside the range"

    try:
        # single frame
        params = capi.frame_params_from_poses(P1.rt12().reshape(3, 4), _poses(P1, TRAJECTORIES["gentle_turn"])[1].rt12().reshape(3, 4), T0, T1, TREQ)
        ref = np.empty_like(pts)
        ctx.deskew_f32(pts, ref, params)
        big, idx, idx2 = fresh()
        ctx.deskew_f32(d_in, big[shift_out:shift_out + n], params, n=n)
        check(big, ref)
        # batched, ragged frames, with frame indices at their own odd offset
        sizes = [1, 0, 63, 20000, 64, 30000, 0, n - 50128]
        offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
        plist = [capi.FrameParams.make(np.array(TRAJECTORIES["gentle_turn"]) * (0.5 + 0.1 * f), 0.1 * f) for f in range(len(sizes))]
        ref_idx = np.empty(n, dtype=np.uint32)
        ctx.deskew_batch_f32(pts, ref, offsets, plist, ref_idx)
        big, idx, idx2 = fresh()
        ctx.deskew_batch_f32(d_in, big[shift_out:shift_out + n], offsets, plist, idx[3:3 + n])
        check(big, ref, [(idx, ref_idx, 3)])
        # trajectory, three knots
        times = [T0 - 0.05, 0.5 * (T0 + T1) + 0.003, T1 + 0.05]
        poses = [P1, orc.affine_mul(P1, orc.se3_exp([1.3, 0.04, -0.01, 0.002, -0.004, 0.035]))]
        poses.append(orc.affine_mul(poses[1], orc.se3_exp([1.4, -0.02, 0.01, -0.003, 0.002, 0.05])))
        rt = np.stack([p.rt12().reshape(3, 4) for p in poses])
        ref_br = np.empty(n, dtype=np.uint32)
        ctx.deskew_traj_f32(pts, ref, times, rt, T0, T1, TREQ, ref_br)
        big, idx, idx2 = fresh()
        ctx.deskew_traj_f32(d_in, big[shift_out:shift_out + n], times, rt, T0, T1, TREQ, idx[5:5 + n], n=n)
        check(big, ref, [(idx, ref_br, 5)])
        # batched trajectories
        frames = [dict(times=times, poses=rt, stamp_start=T0, stamp_end=T1, requested_time=T0 + (0.1 + 0.1 * f) * (T1 - T0)) for f in range(len(sizes))]
        ref_f = np.empty(n, dtype=np.uint32)
        ctx.deskew_traj_batch_f32(pts, ref, offsets, frames, ref_f, ref_br)
        big, idx, idx2 = fresh()
        ctx.deskew_traj_batch_f32(d_in, big[shift_out:shift_out + n], offsets, frames, idx[1:1 + n], idx2[9:9 + n])
        check(big, ref, [(idx, ref_f, 1), (idx2, ref_br, 9)])
    finally:
        ctx.set_stream(None)


# ---- frame queues: a stream of separate frames over several hardware queues -------------------------------------------
def test_frame_queues_same_bits_and_ordering(torch_mod, ctx):
    """kmc_hip_set_frame_queues(q > 1) / kmc_hip_deskew_frames_f32: consecutive single-frame calls are gathered on the host and issued
    as list launches (until ABI 3: spread over 2-4 HIP streams; the contract is the same).  Same bits as the in-order path; a producer
    issued on the context's stream BEFORE the frames and a consumer issued AFTER the join see the right data (no host sync)."""
    torch = torch_mod
    n, nf = 200_003, 24
    params = [capi.FrameParams.make([1.3, 0.05 * (f % 3), -0.02, 0.002, -0.004, 0.03 + 0.001 * f], (f % 5) / 4.0) for f in range(nf)]
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ins = [torch.empty((n, 4), dtype=torch.float32, device="cuda") for _ in range(nf)]
    want = []
    for f in range(nf):
        ctx.synth_points(ins[f], n, 900 + f)
        o = torch.empty_like(ins[f])
        ctx.deskew_f32(ins[f], o, params[f])
        want.append(o)
    torch.cuda.synchronize()
    try:
        for queues in (2, 3, 4):
            ctx.set_frame_queues(queues)
            # producer on the context's stream (torch's current stream), then the frames, then a consumer after the join
            srcs = [torch.zeros_like(ins[f]) for f in range(nf)]
            outs = [torch.zeros_like(ins[f]) for f in range(nf)]
            for f in range(nf):
                srcs[f].copy_(ins[f])                      # producer: must have finished before frame f starts
            for f in range(nf):
                ctx.deskew_f32(srcs[f], outs[f], params[f])  # unordered among themselves
            ctx.frame_queue_join()
            total = sum(o.view(torch.int32).to(torch.int64).sum() for o in outs)   # consumer on the context's stream
            expect = sum(o.view(torch.int32).to(torch.int64).sum() for o in want)
            assert int(total) == int(expect), queues
            for f in range(nf):
                assert torch.equal(outs[f].view(torch.int32), want[f].view(torch.int32)), (queues, f)
            # the one-call form: ONE launch of the frame-list kernel on the context's stream (up to 256 frames' records in its kernel arguments), whatever the context's queue setting
            outs2 = [torch.zeros_like(ins[f]) for f in range(nf)]
            pack = ctx.prepare_frames(list(zip(ins, outs2)), params)
            st = ctx.deskew_frames_f32(pack)
            assert st.n_points == n * nf and st.n_launches == (nf + 255) // 256
            got = torch.stack(outs2)  # consumer on the context's stream, no explicit sync before it
            assert torch.equal(got.view(torch.int32), torch.stack(want).view(torch.int32)), queues
        # other entry points join by themselves: a batched call right after queued frames sees their results
        ctx.set_frame_queues(2)
        mid = [torch.zeros_like(ins[0]) for _ in range(4)]
        for f in range(4):
            ctx.deskew_f32(ins[f], mid[f], params[f])
        ctx.frame_queue_join()  # a torch op on the same stream is not a kmc entry point: join explicitly before it
        cat = torch.cat(mid)
        again = torch.empty_like(cat)
        offsets = np.arange(5, dtype=np.uint64) * n
        ident = capi.FrameParams.make([0, 0, 0, 0, 0, 0], 0.5)
        ctx.deskew_batch_f32(cat, again, offsets, [ident] * 4, None)
        torch.cuda.synchronize()
        assert torch.equal(again.view(torch.int32), torch.cat(want[:4]).view(torch.int32))
    finally:
        ctx.set_frame_queues(1)


def test_switching_streams_drains_the_context(torch_mod, ctx):
    """A context is single-stream: its table slots and scratch are ordered on ONE stream.  kmc_hip_set_stream therefore drains the
    old stream before it switches -- a table growth on the new stream must not free memory under kernels of the old one.
    Big batch on stream A, switch, a batch on stream B whose tables are larger than any so far (forces the growth): both results
    must be what the single-frame kernel gives."""
    torch = torch_mod
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    n_big, f_big = 8_000_000, 4
    params = [capi.FrameParams.make([1.3, 0.05, -0.02, 0.002, -0.004, 0.03 + 0.001 * f], 0.5) for f in range(f_big)]
    a = torch.empty((n_big * f_big, 4), dtype=torch.float32, device="cuda")
    tmp = capi.Context(0)
    tmp.synth_points(a, n_big * f_big, 4242)
    tmp.synchronize()
    tmp.close()
    out_a = torch.zeros_like(a)
    own = capi.Context(0)  # a fresh context: its slot ring still has the minimum capacity
    try:
        own.set_stream(sa.cuda_stream)
        own.deskew_batch_f32(a, out_a, np.arange(f_big + 1, dtype=np.uint64) * n_big, params, None)
        own.set_stream(sb.cuda_stream)  # drains stream A
        f_many = 40_000                  # 40 000 records of 64 + 128 bytes: far beyond the initial slot capacity
        sizes = np.full(f_many, 64, dtype=np.uint64)
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
        m = int(offs[-1])
        out_b = torch.zeros((m, 4), dtype=torch.float32, device="cuda")
        own.deskew_batch_f32(a[:m], out_b, offs, [params[f % f_big] for f in range(f_many)], None)
        own.synchronize()
        sa.synchronize()
        for f in range(f_big):  # stream A's batch
            want = torch.empty((n_big, 4), dtype=torch.float32, device="cuda")
            own.deskew_f32(a[f * n_big:(f + 1) * n_big], want, params[f])
            own.synchronize()
            assert torch.equal(out_a[f * n_big:(f + 1) * n_big].view(torch.int32), want.view(torch.int32)), f
        for f in (0, 1, 17, f_many - 1):  # stream B's batch, sampled
            want = torch.empty((64, 4), dtype=torch.float32, device="cuda")
            own.deskew_f32(a[f * 64:(f + 1) * 64], want, params[f % f_big])
            own.synchronize()
            assert torch.equal(out_b[f * 64:(f + 1) * 64].view(torch.int32), want.view(torch.int32)), f
    finally:
        own.close()


def test_any_order_frames_same_bits_and_hazards_kept(torch_mod, monkeypatch):
    """In order on ONE stream, but not drained: a device-resident frame that shares no buffer with the frames launched since the last
    ordinary launch is dispatched without the barrier bit (kmc_hip.h, kmc_hip_set_frame_queues).  Same bits as a context with
    KMC_ANY_ORDER=0; frames that DO share a buffer with one in flight (chains, an overwritten input, the same buffer twice) stay
    ordered; copies / consumers issued afterwards see every frame's result."""
    torch = torch_mod
    n, nf = 300_007, 40
    params = [capi.FrameParams.make([1.3, 0.05 * (f % 3), -0.02, 0.002, -0.004, 0.03 + 0.001 * f], (f % 5) / 4.0) for f in range(nf)]
    monkeypatch.setenv("KMC_ANY_ORDER", "0")
    plain = capi.Context(0)
    monkeypatch.delenv("KMC_ANY_ORDER")
    fast = capi.Context(0)  # both on their OWN streams: inputs are synchronised by hand
    try:
        assert plain.device_info()["any_order_dispatch"] == 0
        # the feature is on only where kmc_hip_create's probe SAW the device honour it; on MI355X / ROCm 7.2 it does (a box where it
        # does not fails here, loudly, instead of passing on ordinary launches)
        assert fast.device_info()["any_order_dispatch"] == 1, fast.device_info()
        ins = [torch.empty((n, 4), dtype=torch.float32, device="cuda") for _ in range(nf)]
        for f in range(nf):
            plain.synth_points(ins[f], n, 4100 + f)
        plain.synchronize()
        want = [torch.empty_like(x) for x in ins]
        for f in range(nf):
            plain.deskew_f32(ins[f], want[f], params[f])
        plain.synchronize()
        assert plain.any_order_launches() == 0
        # (1) independent frames: one ordinary launch opens a window, at most 127 frames follow it without the barrier bit
        window = 128
        outs = [torch.zeros_like(x) for x in ins]
        torch.cuda.synchronize()
        for f in range(nf):
            fast.deskew_f32(ins[f], outs[f], params[f])
        assert fast.any_order_launches() == nf - (nf + window - 1) // window, fast.any_order_launches()
        fast.synchronize()
        for f in range(nf):
            assert torch.equal(outs[f].view(torch.int32), want[f].view(torch.int32)), f
        # ... and the window closes and re-opens: 300 small independent frames -> ordinary launches at 0, 128 and 256
        small, many = 1_000, 300
        s_out, s_want = torch.zeros((many * small, 4), dtype=torch.float32, device="cuda"), torch.zeros((many * small, 4), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        before = fast.any_order_launches()
        for f in range(many):
            fast.deskew_f32(ins[f % nf][f * small:(f + 1) * small], s_out[f * small:(f + 1) * small], params[f % nf])
            plain.deskew_f32(ins[f % nf][f * small:(f + 1) * small], s_want[f * small:(f + 1) * small], params[f % nf])
        assert fast.any_order_launches() - before == many - (many + window - 1) // window, fast.any_order_launches() - before
        fast.synchronize()
        plain.synchronize()
        assert torch.equal(s_out.view(torch.int32), s_want.view(torch.int32))
        # (2) a chain: every frame reads what the one before it wrote -> none of them may overtake
        before = fast.any_order_launches()
        chain_ref = [ins[0]]
        for f in range(6):
            o = torch.empty_like(ins[0])
            plain.deskew_f32(chain_ref[-1], o, params[f])
            plain.synchronize()
            chain_ref.append(o)
        chain = [ins[0]] + [torch.zeros_like(ins[0]) for _ in range(6)]
        torch.cuda.synchronize()
        for f in range(6):
            fast.deskew_f32(chain[f], chain[f + 1], params[f])
        assert fast.any_order_launches() == before
        fast.synchronize()
        assert torch.equal(chain[6].view(torch.int32), chain_ref[6].view(torch.int32))
        # (3) write-after-read: B overwrites the buffer A reads; the same buffer in place twice; a partial overlap
        big = 4_000_000
        x = torch.empty((big, 4), dtype=torch.float32, device="cuda")
        plain.synth_points(x, big, 77)
        plain.synchronize()
        x0 = x.clone()
        a_ref, ip_ref = torch.empty_like(x), x0.clone()
        torch.cuda.synchronize()  # torch's clones run on ITS stream; the contexts' own streams are not ordered with it
        plain.deskew_f32(x0, a_ref, params[1])
        plain.deskew_f32(ip_ref, ip_ref, params[2])
        plain.deskew_f32(ip_ref, ip_ref, params[3])
        plain.synchronize()
        a_out, other = torch.zeros_like(x), ins[5]
        torch.cuda.synchronize()
        before = fast.any_order_launches()
        fast.deskew_f32(x, a_out, params[1])               # A reads x
        fast.deskew_f32(other, x[:n], params[5])             # B writes the head of x: must wait for A
        assert fast.any_order_launches() == before
        fast.synchronize()
        assert torch.equal(a_out.view(torch.int32), a_ref.view(torch.int32))
        assert torch.equal(x[:n].view(torch.int32), want[5].view(torch.int32))
        y = x0.clone()
        torch.cuda.synchronize()
        before = fast.any_order_launches()
        fast.deskew_f32(y, y, params[2])
        fast.deskew_f32(y, y, params[3])
        assert fast.any_order_launches() == before
        fast.synchronize()
        assert torch.equal(y.view(torch.int32), ip_ref.view(torch.int32))
        # (4) whatever follows on the stream waits for every frame of the window: a batched call right behind 8 any-order frames
        mid = [torch.zeros_like(ins[0]) for _ in range(8)]
        torch.cuda.synchronize()
        before = fast.any_order_launches()
        for f in range(8):
            fast.deskew_f32(ins[f], mid[f], params[f])
        assert fast.any_order_launches() == before + 7
        # (the eight outputs are separate tensors: gather them with the library itself, an ordinary launch per tensor)
        ident = capi.FrameParams.make([0, 0, 0, 0, 0, 0], 0.5)
        again = [torch.zeros_like(ins[0]) for _ in range(8)]
        for f in range(8):
            fast.deskew_batch_f32(mid[f], again[f], np.array([0, n], dtype=np.uint64), [ident], None)
        fast.synchronize()
        for f in range(8):
            assert torch.equal(again[f].view(torch.int32), want[f].view(torch.int32)), f
        # (5) a caller's stream: a list handed to kmc_hip_deskew_frames_f32 is ONE ordinary launch; separate calls go out without the
        # barrier bit only after the caller has said that nothing is produced in between
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            fast.set_stream(side.cuda_stream)
            fast.set_frame_queues(1)
            outs2 = [torch.zeros_like(x) for x in ins]
            pack = fast.prepare_frames(list(zip(ins, outs2)), params)
            before = fast.any_order_launches()
            st = fast.deskew_frames_f32(pack)
            assert st.n_launches == 1 and fast.any_order_launches() == before
            got = torch.stack(outs2)  # consumer on the same stream, no sync in between
            assert torch.equal(got.view(torch.int32), torch.stack(want).view(torch.int32))
            outs3 = [torch.zeros_like(x) for x in ins[:8]]
            before = fast.any_order_launches()
            for f in range(8):
                fast.deskew_f32(ins[f], outs3[f], params[f])
            assert fast.any_order_launches() == before  # producers may sit between two calls on a caller's stream
            fast.set_frame_queue_order(False)
            for f in range(8):
                fast.deskew_f32(ins[f], outs3[f], params[f])  # writes outs3 again: conflicts with the frames above -> first ordered
            assert fast.any_order_launches() == before + 7
            got = torch.stack(outs3)
            assert torch.equal(got.view(torch.int32), torch.stack(want[:8]).view(torch.int32))
            fast.set_frame_queue_order(True)
        side.synchronize()
        fast.set_stream(None)
    finally:
        plain.close()
        fast.close()


def test_any_order_dispatch_is_gated_by_the_runtime_probe(torch_mod, monkeypatch):
    """hipExtAnyOrderLaunch is documented as unsupported on gfx9, so kmc_hip_create probes the device (an ordinary
    kernel, a copy and an event behind barrier-free packets must wait for all of them; a barrier-free packet must really start next to
    the kernel before it) and only a passed probe switches the feature on.  The verdict is exported (kmc_device_info.any_order_dispatch)
    and GATES the dispatch: without a verified verdict (KMC_ANY_ORDER=0 here; a failed probe sets the same switch -- the getenv
    hook that faked one left the product in round 5) every launch is ordinary -- same bits either way."""
    torch = torch_mod
    n, nf = 200_003, 12
    params = [capi.FrameParams.make([1.3, 0.05 * (f % 3), -0.02, 0.002, -0.004, 0.03 + 0.001 * f], (f % 5) / 4.0) for f in range(nf)]
    ctxs = {}
    try:
        ctxs["probed"] = capi.Context(0)
        monkeypatch.setenv("KMC_ANY_ORDER", "0")
        ctxs["switched_off"] = capi.Context(0)
        monkeypatch.delenv("KMC_ANY_ORDER")
        verdicts = {k: c.device_info()["any_order_dispatch"] for k, c in ctxs.items()}
        assert verdicts["switched_off"] == 0
        assert verdicts["probed"] in (1, -1, -3)
        ins = [torch.empty((n, 4), dtype=torch.float32, device="cuda") for _ in range(nf)]
        for f in range(nf):
            ctxs["probed"].synth_points(ins[f], n, 5200 + f)
        ctxs["probed"].synchronize()
        outs = {}
        for k, c in ctxs.items():
            outs[k] = [torch.zeros_like(x) for x in ins]
            torch.cuda.synchronize()
            for f in range(nf):
                c.deskew_f32(ins[f], outs[k][f], params[f])
            launched = c.any_order_launches()
            c.synchronize()
            if verdicts[k] == 1:
                assert launched == nf - 1, (k, launched)  # one ordinary launch opens the window, the rest follow without the barrier bit
            else:
                assert launched == 0, (k, launched)       # the gate: no verified verdict, no barrier-free dispatch
        for k in ("switched_off",):
            for f in range(nf):
                assert torch.equal(outs[k][f].view(torch.int32), outs["probed"][f].view(torch.int32)), (k, f)
    finally:
        for c in ctxs.values():
            c.close()


def test_frame_list_is_one_launch_with_the_single_frame_kernels_bits(torch_mod, ctx, kitti):
    """kmc_hip_deskew_frames_f32: separate frames, each in its own buffer, in ONE launch of the frame-list kernel (2-D grid: frame x
    tile).  Ragged and empty frames, outputs at arbitrary 16-byte offsets (every frame's tiles are cut on ITS output's 1 KiB lines),
    lists short enough for the kernel-argument tables (<= 16) and long ones (device tables), mixed coefficient tiers: bit for bit what
    kmc_hip_deskew_f32 writes for each frame alone, and within the bar of the FAITHFUL oracle.  A list whose frames depend on each
    other is recognised and issued frame by frame, in order."""
    torch = torch_mod
    xyzi, P1 = kitti
    rng = np.random.default_rng(404)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    # (the third list: one huge frame among one-point frames -- the grid is as wide as the huge frame, almost every workgroup of the others leaves at once)
    for sizes in ([0, 1, 63, 64, 65, 1023, 1025, 5000, 0, 123_397, 777], list(rng.integers(0, 30_000, size=45)) + [0, 64, 200_001],
                  [1] * 30 + [1_500_000] + [1] * 30 + [65] * 9):
        nf = len(sizes)
        big_in = torch.zeros((int(sum(sizes)) + 80 * nf + 64, 4), dtype=torch.float32, device="cuda")
        big_out = torch.zeros_like(big_in)
        ins, outs, wants, params, poses = [], [], [], [], []
        o = 0
        for f, n in enumerate(sizes):
            shift_in, shift_out = int(rng.integers(0, 64)), int(rng.integers(0, 64))  # 16-byte granules: any offset within a 1 KiB line
            pts = np.ascontiguousarray(xyzi[rng.integers(0, xyzi.shape[0], size=int(n))])
            a = big_in[o + shift_in:o + shift_in + n]
            b = big_out[o + shift_out:o + shift_out + n]
            a.copy_(torch.from_numpy(pts))
            step = np.concatenate([rng.normal(0, 1.5, 3), rng.normal(0, 0.05 if f % 7 else 0.6, 3)])  # every seventh frame turns hard: a wider tier
            A, B = _poses(P1, step)
            treq = T0 + rng.uniform(0, 1) * (T1 - T0)
            ins.append(a); outs.append(b); poses.append((pts, A, B, treq))
            params.append(_params(A, B, treq=treq))
            o += int(n) + 80
        torch.cuda.synchronize()
        tiers = {}
        for f in range(nf):  # reference bits: one kmc_hip_deskew_f32 call per frame, at the frame's OWN tier (round 5: a list launches per tier present)
            w = torch.empty_like(ins[f])
            t = ctx.deskew_f32(ins[f], w, params[f]).variant
            if sizes[f]:
                tiers[t] = tiers.get(t, 0) + 1
            wants.append(w)
        tier = max(tiers)
        assert len(tiers) >= 2  # every seventh frame turns hard: the lists of this test DO mix tiers
        pack = ctx.prepare_frames(list(zip(ins, outs)), params)
        st = ctx.deskew_frames_f32(pack)
        torch.cuda.synchronize()
        # per tier: ONE launch, the records of up to 256 frames in its kernel arguments (round 5; KMC_LIST_ROUTE=table: one launch over an uploaded table)
        assert st.n_launches == sum((cnt + 255) // 256 for cnt in tiers.values()) and st.n_points == sum(sizes) and st.variant == tier
        for f in range(nf):
            assert torch.equal(outs[f].view(torch.int32), wants[f].view(torch.int32)), (nf, f, sizes[f])
            pts, A, B, treq = poses[f]
            if sizes[f]:
                _check(outs[f].cpu().numpy(), pts, _oracle(pts, A, B, treq=treq))
        # nothing outside the frames' own ranges was written (the dead heads and the clipped tails stay untouched)
        mask = torch.ones(big_out.shape[0], dtype=torch.bool, device="cuda")
        for b in outs:
            if b.shape[0]:
                start = (b.data_ptr() - big_out.data_ptr()) // 16
                mask[start:start + b.shape[0]] = False
        assert not big_out[mask].any()
    # a chain inside one list: frame k reads what frame k-1 wrote -> recognised, issued in order as separate ordinary launches
    n = 70_001
    bufs = [torch.zeros((n, 4), dtype=torch.float32, device="cuda") for _ in range(6)]
    ctx.synth_points(bufs[0], n, 31337)
    prm = [capi.FrameParams.make([1.3, 0.05, -0.02, 0.002, -0.004, 0.03 + 0.001 * f], 0.4) for f in range(5)]
    ref = [bufs[0]]
    for f in range(5):
        w = torch.empty_like(bufs[0])
        ctx.deskew_f32(ref[-1], w, prm[f])
        ref.append(w)
    st = ctx.deskew_frames_f32(ctx.prepare_frames([(bufs[f], bufs[f + 1]) for f in range(5)], prm))
    torch.cuda.synchronize()
    assert st.n_launches == 5
    assert torch.equal(bufs[5].view(torch.int32), ref[5].view(torch.int32))
    # in place (in == out of the SAME frame) is not a hazard
    ip = [ref[f].clone() for f in range(3)]
    st = ctx.deskew_frames_f32(ctx.prepare_frames([(x, x) for x in ip], prm[:3]))
    torch.cuda.synchronize()
    assert st.n_launches == 1
    for f in range(3):
        assert torch.equal(ip[f].view(torch.int32), ref[f + 1].view(torch.int32)), f


def test_f64cols_begin_calls_queue_up_behind_each_other(torch_mod, ctx, kitti):
    """kmc_hip_deskew_f64cols_begin on device-resident columns may be called K times before ONE _end: the launches queue up on the
    context's stream without a host wait between them (what bench.py's f64cols leg times), _end returns the combined verdict and stats
    (a second _begin used to clobber the pending verdict).  Same bits as K separate calls; an out-of-range stamp in ANY of
    the queued frames surfaces at _end; a staged host call cannot join the queue and leaves it intact."""
    torch = torch_mod
    xyzi, P1 = kitti
    n, K = 50_000, 5
    rng = np.random.default_rng(64)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    frames = []
    for k in range(K):
        A, B = _poses(P1, TRAJECTORIES["hard_turn"])
        params = _params(A, B)
        cloud = xyzi[rng.integers(0, xyzi.shape[0], size=n), :3].astype(np.float64)
        stamps = T0 + rng.uniform(0, 1, size=n) * (T1 - T0)
        cols = [torch.from_numpy(np.ascontiguousarray(cloud[:, j])).cuda() for j in range(3)] + [torch.ones(n, dtype=torch.float64, device="cuda")]
        frames.append((cols, torch.from_numpy(stamps).cuda(), params))
    want = []
    for cols, st_d, params in frames:
        outs = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(4)]
        ctx.deskew_f64cols(*cols, st_d, T0, T1, params, *outs)
        want.append(outs)
    got = [[torch.zeros(n, dtype=torch.float64, device="cuda") for _ in range(4)] for _ in range(K)]
    for (cols, st_d, params), outs in zip(frames, got):
        ctx.deskew_f64cols_begin(*cols, st_d, T0, T1, params, *outs)
    # a staged host call cannot queue behind them, and does not disturb them
    h = [np.zeros(8) for _ in range(9)]
    with pytest.raises(capi.KmcError):
        ctx.deskew_f64cols_begin(h[0], h[1], h[2], h[3], h[4] + T0, T0, T1, frames[0][2], h[5], h[6], h[7], h[8])
    rc, st = ctx.deskew_f64cols_end()
    assert rc == capi.OK and st.n_points == K * n and st.n_launches == K and st.n_out_of_range == 0
    for k in range(K):
        for j in range(4):
            assert torch.equal(got[k][j].view(torch.int64), want[k][j].view(torch.int64)), (k, j)
    # one bad stamp in the THIRD of five queued frames
    bad = frames[2][1].clone()
    bad[1234] = T1 + 1.0
    for k, ((cols, st_d, params), outs) in enumerate(zip(frames, got)):
        ctx.deskew_f64cols_begin(*cols, bad if k == 2 else st_d, T0, T1, params, *outs)
    rc, st = ctx.deskew_f64cols_end(raise_on_range=False)
    assert rc == capi.ERR_TIME_OUT_OF_RANGE and st.n_out_of_range == 1 and st.n_points == K * n
    # and the context is usable again
    rc, st = ctx.deskew_f64cols(*frames[0][0], frames[0][1], T0, T1, frames[0][2], *got[0])
    assert rc == capi.OK and st.n_out_of_range == 0
    with pytest.raises(capi.KmcError):
        ctx.deskew_f64cols_end()  # nothing pending


def test_batch_on_page_locked_buffers_that_are_only_4_byte_aligned(torch_mod, ctx, kitti):
    """kmc_hip_deskew_batch_f32 recognises page-locked KMC_MEM_HOST buffers and works on them in place -- but the
    kernels' 16-byte accesses need 16-byte-aligned pointers.  A pinned buffer at a 4-byte offset is a valid KMC_MEM_HOST argument (the
    staged route only copies it) and must stay one."""
    torch = torch_mod
    xyzi, P1 = kitti
    sizes = [3000, 0, 4500, 2077]
    n = sum(sizes)
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    rng = np.random.default_rng(9)
    pts = np.ascontiguousarray(xyzi[rng.integers(0, xyzi.shape[0], size=n)])
    params = []
    for f in range(len(sizes)):
        A, B = _poses(P1, np.concatenate([rng.normal(0, 1.5, 3), rng.normal(0, 0.05, 3)]))
        params.append(_params(A, B, treq=T0 + 0.3 * (T1 - T0)))
    pin_in = torch.zeros(4 * n + 8, dtype=torch.float32).pin_memory()
    pin_out = torch.zeros(4 * n + 8, dtype=torch.float32).pin_memory()
    aligned = np.empty_like(pts)
    a16_in, a16_out = pin_in.numpy()[:4 * n].reshape(n, 4), pin_out.numpy()[:4 * n].reshape(n, 4)
    a16_in[:] = pts
    st = ctx.deskew_batch_f32(a16_in, a16_out, offsets, params, None)  # 16-byte aligned pinned buffers: in place, one launch
    aligned[:] = a16_out
    off_in, off_out = pin_in.numpy()[1:4 * n + 1].reshape(n, 4), pin_out.numpy()[1:4 * n + 1].reshape(n, 4)  # 4 bytes further
    assert off_in.ctypes.data % 16 == 4 and off_out.ctypes.data % 16 == 4
    off_in[:] = pts
    off_out[:] = 0
    st = ctx.deskew_batch_f32(off_in, off_out, offsets, params, None)
    assert st.n_points == n
    assert np.array_equal(off_out.view(np.uint32), aligned.view(np.uint32))


def test_gathered_calls_keep_in_order_results(torch_mod, kitti):
    """kmc_hip_set_frame_queues(ctx, q > 1): one kmc_hip_deskew_f32 call per frame, gathered by the library into launches of the
    frame-list kernel (up to 16 frames each).  The RESULTS are those of in-order execution, bit for bit: frames that depend on each other
    (a chain, the same buffer twice, an overwritten input), frames of different coefficient tiers, empty and ragged frames, outputs at
    any 16-byte offset; whatever else the context puts on its stream afterwards sees every gathered frame; far fewer launches than
    frames."""
    torch = torch_mod
    xyzi, _ = kitti
    rng = np.random.default_rng(1616)
    plain, g = capi.Context(0), capi.Context(0)  # both on their OWN streams
    try:
        g.set_frame_queues(4)
        nf = 70
        sizes = [int(v) for v in rng.choice([0, 1, 63, 64, 65, 1000, 4097, 30_000, 123_397], nf)]
        big_in = torch.zeros((sum(sizes) + 80 * nf + 64, 4), dtype=torch.float32, device="cuda")
        big_out = torch.zeros_like(big_in)
        big_ref = torch.zeros_like(big_in)
        ins, outs, refs, params = [], [], [], []
        o = 0
        for f, n in enumerate(sizes):
            si, so = int(rng.integers(0, 64)), int(rng.integers(0, 64))
            a = big_in[o + si:o + si + n]
            a.copy_(torch.from_numpy(np.ascontiguousarray(xyzi[rng.integers(0, xyzi.shape[0], size=n)])))
            ins.append(a); outs.append(big_out[o + so:o + so + n]); refs.append(big_ref[o + so:o + so + n])
            yaw = 0.6 if f % 17 == 16 else (2.0 if f % 23 == 22 else 0.03)  # mostly series3, now and then series5 / wide: tier changes inside the stream of calls
            params.append(capi.FrameParams.make([1.0 + 0.01 * f, 0.02, -0.01, 0.001, -0.002, yaw], float(rng.uniform(0, 1))))
            o += n + 80
        torch.cuda.synchronize()
        for f in range(nf):
            plain.deskew_f32(ins[f], refs[f], params[f])
        plain.synchronize()
        launches = sum(g.deskew_f32(ins[f], outs[f], params[f]).n_launches for f in range(nf))
        # anything else on the context's stream issues the pending frames first: an identity batch over the WHOLE output buffer
        ident = capi.FrameParams.make([0, 0, 0, 0, 0, 0], 0.5)
        copy = torch.zeros_like(big_out)
        g.deskew_batch_f32(big_out, copy, np.array([0, big_out.shape[0]], dtype=np.uint64), [ident], None)
        g.synchronize()
        assert torch.equal(copy.view(torch.int32), big_ref.view(torch.int32))
        assert launches < nf // 2, (launches, nf)  # gathered: tier changes and the idle checks split the stream, not every frame
        # a chain: frame k reads what frame k-1 wrote -- the library sees the overlap and issues the pending frame first
        n = 50_003
        bufs = [torch.zeros((n, 4), dtype=torch.float32, device="cuda") for _ in range(7)]
        plain.synth_points(bufs[0], n, 99)
        plain.synchronize()
        ref = [bufs[0]]
        for f in range(6):
            w = torch.empty_like(bufs[0])
            plain.deskew_f32(ref[-1], w, params[f])
            ref.append(w)
        plain.synchronize()
        for f in range(6):
            g.deskew_f32(bufs[f], bufs[f + 1], params[f])
        g.frame_queue_join()
        g.synchronize()
        assert torch.equal(bufs[6].view(torch.int32), ref[6].view(torch.int32))
        # the same buffer in place, three times; then an input overwritten by a later frame (write after read)
        y = ref[0].clone()
        torch.cuda.synchronize()
        for f in range(3):
            g.deskew_f32(y, y, params[f])
        x_in, x_out = ref[0].clone(), torch.zeros_like(ref[0])
        torch.cuda.synchronize()
        g.deskew_f32(x_in, x_out, params[0])       # reads x_in
        g.deskew_f32(ref[2], x_in, params[1])      # overwrites x_in: must not reach the device before the frame above has read it
        g.synchronize()
        assert torch.equal(y.view(torch.int32), ref[3].view(torch.int32))
        assert torch.equal(x_out.view(torch.int32), ref[1].view(torch.int32))
        want = torch.empty_like(ref[0])
        plain.deskew_f32(ref[2], want, params[1])
        plain.synchronize()
        assert torch.equal(x_in.view(torch.int32), want.view(torch.int32))
        # nothing stays pending behind a synchronize, and switching gathering off issues what is pending
        z = torch.zeros_like(ref[0])
        g.deskew_f32(ref[0], z, params[0])
        g.set_frame_queues(1)
        g.synchronize()
        assert torch.equal(z.view(torch.int32), ref[1].view(torch.int32))
    finally:
        plain.close()
        g.close()


@pytest.mark.gpu
def test_calls_can_be_captured_into_a_hip_graph_and_replayed(torch_mod, kitti):
    """The device-resident entry points that only ENQUEUE launches -- single frames, a list (here 20 frames: while the stream captures, a
    list beyond the 16 frames of the kernel-argument table goes out as two such launches instead of a table upload), a batch of at most
    16 frames, a 3-knot frame, and gathered calls closed by kmc_hip_frame_queue_join -- are captured into a HIP graph on the caller's
    stream; nothing runs during the capture; a replay over NEW contents of the same buffers writes what eager calls write, bit for bit.
    A batch that needs a table upload refuses to be captured (KMC_ERR_INVALID_ARG) and leaves the capture alive."""
    torch = torch_mod
    xyzi, P1 = kitti
    rng = np.random.default_rng(777)
    n = 20_000
    nf_single, nf_list, nf_batch, nf_gather = 4, 20, 10, 5
    total = nf_single + nf_list + nf_batch + 1 + nf_gather
    side = torch.cuda.Stream()
    c, eager = capi.Context(0), capi.Context(0)
    try:
        def draw():
            return torch.from_numpy(np.ascontiguousarray(xyzi[rng.integers(0, xyzi.shape[0], size=(total, n))])).cuda()

        d_in = draw()                      # (total, n, 4): frame k is d_in[k]
        d_out = torch.zeros_like(d_in)
        params = []
        for k in range(total):
            A, B = _poses(P1, np.concatenate([rng.normal(0, 1.5, 3), rng.normal(0, 0.05, 3)]))
            params.append(_params(A, B, treq=T0 + rng.uniform(0, 1) * (T1 - T0)))
        k_list, k_batch, k_traj, k_gather = nf_single, nf_single + nf_list, nf_single + nf_list + nf_batch, nf_single + nf_list + nf_batch + 1
        knots_t = np.array([T0, 0.5 * (T0 + T1), T1])
        Pa, Pb = _poses(P1, np.array([0.6, 0.02, 0.0, 0.0, 0.0, 0.012]))
        Pc = orc.affine_mul(Pb, orc.se3_exp(np.array([0.7, -0.01, 0.01, 0.001, 0.0, 0.02])))
        knots_P = np.stack([P.rt12() for P in (Pa, Pb, Pc)])
        offsets = np.arange(nf_batch + 1, dtype=np.uint64) * n
        batch_in, batch_out = d_in[k_batch:k_batch + nf_batch].view(-1, 4), d_out[k_batch:k_batch + nf_batch].view(-1, 4)

        def calls(cx, dst):
            sts = {}
            for k in range(nf_single):
                cx.deskew_f32(d_in[k], dst[k], params[k])
            pack = cx.prepare_frames([(d_in[k_list + f], dst[k_list + f]) for f in range(nf_list)], params[k_list:k_list + nf_list])
            sts["list"] = cx.deskew_frames_f32(pack)
            sts["batch"] = cx.deskew_batch_f32(d_in[k_batch:k_batch + nf_batch].view(-1, 4), dst[k_batch:k_batch + nf_batch].view(-1, 4), offsets,
                                               params[k_batch:k_batch + nf_batch])
            cx.deskew_traj_f32(d_in[k_traj], dst[k_traj], knots_t, knots_P, T0, T1, TREQ)
            cx.set_frame_queues(4)
            for f in range(nf_gather):
                cx.deskew_f32(d_in[k_gather + f], dst[k_gather + f], params[k_gather + f])
            cx.frame_queue_join()
            cx.set_frame_queues(1)
            return sts, pack

        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            c.set_stream(side.cuda_stream)
            calls(c, d_out)  # warm-up on the capture stream (tables, pools)
            side.synchronize()
            d_out.zero_()
            side.synchronize()
            with torch.cuda.graph(g, stream=side):
                sts, keep = calls(c, d_out)
                # 40 frames need a device table: not capturable, refused with a message, the capture goes on
                many = np.arange(41, dtype=np.uint64) * 500
                with pytest.raises(capi.KmcError) as ei:
                    c.deskew_batch_f32(batch_in, batch_out, many, [params[0]] * 40)
                assert ei.value.status == capi.ERR_INVALID_ARG
        assert sts["list"].n_launches == 2 and sts["batch"].n_launches == 1
        torch.cuda.synchronize()
        assert not bool(d_out.any()), "a captured call ran during the capture"
        for rep in range(2):
            d_in.copy_(draw())             # new contents, same buffers
            d_out.zero_()
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            want = torch.zeros_like(d_in)
            eager.set_stream(torch.cuda.current_stream().cuda_stream)
            calls(eager, want)
            eager.synchronize()
            torch.cuda.synchronize()
            assert torch.equal(d_out.view(torch.int32), want.view(torch.int32)), rep
            assert bool(d_out[:, :, :3].ne(d_in[:, :, :3]).any(dim=2).any(dim=1).all()), "every frame was written"
        del keep
    finally:
        c.set_stream(None)
        c.close()
        eager.close()


@pytest.mark.gpu
def test_per_call_timing_and_the_contexts_page_locked_allocator(torch_mod, kitti):
    """kmc_hip_enable_timing: kmc_stats carries HIP-event times -- kernel_ms > 0, total_ms >= kernel_ms, a staged host call's total well above
    its kernel; off again: both 0.  kmc_hip_host_alloc hands out page-locked memory (64-byte aligned) that a KMC_MEM_HOST call recognises
    and works on in place, same bits as the staged route; kmc_hip_host_free takes it back, and a pointer it does not own is an error."""
    import ctypes as C

    torch = torch_mod
    xyzi, P1 = kitti
    n = xyzi.shape[0]
    A, B = _poses(P1, np.array(TRAJECTORIES["gentle_turn"], dtype=np.float64))
    prm = _params(A, B)
    c = capi.Context(0)
    try:
        d_in = torch.from_numpy(xyzi).cuda()
        d_out = torch.empty_like(d_in)
        h_out = np.empty_like(xyzi)
        st0 = c.deskew_f32(d_in, d_out, prm)
        assert st0.kernel_ms == 0.0 and st0.total_ms == 0.0
        c.enable_timing(True)
        st_dev = c.deskew_f32(d_in, d_out, prm)
        st_host = c.deskew_f32(xyzi, h_out, prm)
        offsets = np.array([0, n // 3, n], dtype=np.uint64)
        st_batch = c.deskew_batch_f32(d_in, d_out, offsets, [prm, prm])
        c.enable_timing(False)
        st_off = c.deskew_f32(d_in, d_out, prm)
        for st in (st_dev, st_host, st_batch):
            assert st.kernel_ms > 0.0 and st.total_ms >= st.kernel_ms * 0.999, (st.kernel_ms, st.total_ms)
        assert st_dev.kernel_ms < 1.0 and st_host.total_ms > 2 * st_dev.kernel_ms  # 2 MB each way over PCIe against 4 MB through HBM
        assert st_off.kernel_ms == 0.0 and st_off.total_ms == 0.0
        c.synchronize()
        assert np.array_equal(d_out.cpu().numpy().view(np.uint32), h_out.view(np.uint32))

        lib = capi.lib()
        ptrs = []
        for _ in range(2):
            p = C.c_void_p()
            assert lib.kmc_hip_host_alloc(c._h, n * 16, C.byref(p)) == capi.OK and p.value and p.value % 64 == 0
            ptrs.append(p)
        a_in = np.ctypeslib.as_array(C.cast(ptrs[0], C.POINTER(C.c_float)), shape=(n, 4))
        a_out = np.ctypeslib.as_array(C.cast(ptrs[1], C.POINTER(C.c_float)), shape=(n, 4))
        a_in[:] = xyzi
        a_out[:] = 0
        st = c.deskew_f32(a_in, a_out, prm)          # KMC_MEM_HOST: recognised as device-addressable, one kernel in place
        assert st.n_launches == 1
        assert np.array_equal(a_out.view(np.uint32), h_out.view(np.uint32))
        del a_in, a_out
        for p in ptrs:
            assert lib.kmc_hip_host_free(c._h, p) == capi.OK
        assert lib.kmc_hip_host_free(c._h, C.c_void_p(h_out.ctypes.data)) != capi.OK
    finally:
        c.close()


@pytest.mark.gpu
def test_contexts_on_concurrent_host_threads(torch_mod, kitti):
    """"Thread-safe per context": four host threads, each with its OWN context on the same GPU (ctypes drops the GIL during a call), run
    single-frame, batched, list and staged-host calls at the same time, contexts being created and destroyed while the others work (the
    any-order probe's per-device verdict, the page-locked pool and the table slots are the shared state); every thread's results equal
    the bits a lone context wrote beforehand."""
    import threading

    torch = torch_mod
    xyzi, P1 = kitti
    n = 30_000
    rng = np.random.default_rng(2024)
    n_threads, rounds, nf = 4, 12, 6
    work = []
    for t in range(n_threads):
        pts = np.ascontiguousarray(xyzi[rng.integers(0, xyzi.shape[0], size=(nf, n))])
        params = []
        for f in range(nf):
            A, B = _poses(P1, np.concatenate([rng.normal(0, 1.5, 3), rng.normal(0, 0.05, 3)]))
            params.append(_params(A, B, treq=T0 + rng.uniform(0, 1) * (T1 - T0)))
        work.append((pts, params))

    def run(cx, pts, params, stream):
        cx.set_stream(stream.cuda_stream)
        with torch.cuda.stream(stream):
            d_in = torch.from_numpy(pts).cuda()
            outs = {k: torch.zeros_like(d_in) for k in ("single", "batch", "list")}
            stream.synchronize()
        for f in range(nf):
            cx.deskew_f32(d_in[f], outs["single"][f], params[f])
        cx.deskew_batch_f32(d_in.view(-1, 4), outs["batch"].view(-1, 4), np.arange(nf + 1, dtype=np.uint64) * n, params)
        cx.deskew_frames_f32(cx.prepare_frames([(d_in[f], outs["list"][f]) for f in range(nf)], params))
        h = np.empty_like(pts[0])
        cx.deskew_f32(pts[0], h, params[0])  # pageable host buffers: the staged pipeline
        cx.synchronize()
        return {k: v.cpu().numpy().view(np.uint32) for k, v in outs.items()}, h.view(np.uint32)

    lone = capi.Context(0)
    try:
        lone.force_tier(0)
        want = [run(lone, *work[t], torch.cuda.Stream()) for t in range(n_threads)]
    finally:
        lone.close()
    errors = []

    def worker(t):
        try:
            stream = torch.cuda.Stream()
            for r in range(rounds):
                cx = capi.Context(0)
                try:
                    cx.force_tier(0)
                    got, got_h = run(cx, *work[t], stream)
                finally:
                    cx.close()
                for k in got:
                    if not np.array_equal(got[k], want[t][0][k]):
                        errors.append((t, r, k))
                if not np.array_equal(got_h, want[t][1]):
                    errors.append((t, r, "host"))
        except Exception as e:  # noqa: BLE001 -- reported by the main thread
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:5]


@pytest.mark.gpu
def test_pool_blocks_live_on_the_devices_numa_node_and_the_thread_can_be_bound_there(torch_mod):
    """kmc_host_pool_alloc places its page-locked blocks on the GPU's NUMA node (an in-place kernel's every access crosses the
    inter-socket link otherwise), whatever node the allocating thread runs on; kmc_hip_bind_thread_near_device puts the calling thread
    on that node's CPUs.  Checked through the kernel's own view: move_pages(2) says where a page lives, sched_getaffinity where the
    thread may run.  Skipped on a machine with one node / no topology information."""
    import ctypes

    torch = torch_mod
    props = torch.cuda.get_device_properties(0)
    if not all(hasattr(props, k) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
        pytest.skip("this torch does not say where device 0 sits on the PCI bus")
    sysdir = "/sys/bus/pci/devices/%04x:%02x:%02x.0" % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
    if not os.path.exists(os.path.join(sysdir, "numa_node")):
        pytest.skip("no topology information for " + sysdir)
    node = int(open(os.path.join(sysdir, "numa_node")).read())
    cpus = open(os.path.join(sysdir, "local_cpulist")).read().strip()
    if node < 0 or len(os.listdir("/sys/devices/system/node")) == 0 or not os.path.exists("/sys/devices/system/node/node1"):
        pytest.skip("one NUMA node: nothing to place")
    local = set()
    for part in cpus.split(","):
        a, _, b = part.partition("-")
        local.update(range(int(a), int(b or a) + 1))
    other = sorted(set(os.sched_getaffinity(0)) - local)
    if not other:
        pytest.skip("this process may only run on the GPU's node")
    libc = ctypes.CDLL(None, use_errno=True)
    before = os.sched_getaffinity(0)

    def node_of(addr):
        pages = (ctypes.c_void_p * 1)(addr & ~4095)
        status = (ctypes.c_int * 1)(-1)
        rc = libc.syscall(279, 0, 1, pages, None, status, 0)  # move_pages(pid 0, 1 page, nodes = NULL: query)
        assert rc == 0, ctypes.get_errno()
        return status[0]

    try:
        os.sched_setaffinity(0, other[:8])                     # the allocating thread on the OTHER socket
        blk = capi.PooledArray((1 << 20,), dtype=np.float64)   # 8 MiB: a fresh block (no cached one of this class: the sizes of the suite are smaller or larger)
        try:
            blk.a[:] = 1.0
            assert node_of(blk.a.ctypes.data) == node and node_of(blk.a.ctypes.data + blk.a.nbytes - 8) == node
        finally:
            blk.close()
        # the binding is a hint within the mask the thread already has (a container's cpuset, an outer taskset): from a mask
        # that holds none of the node's CPUs nothing changes ...
        capi.bind_thread_near_device(0)
        assert set(os.sched_getaffinity(0)) == set(other[:8])
        # ... and from the process's own mask the thread ends up on the node's CPUs, and only on ones it was allowed before
        os.sched_setaffinity(0, before)
        if not (set(before) & local):
            pytest.skip("this process may not run on the GPU's node")
        capi.bind_thread_near_device(0)
        now = set(os.sched_getaffinity(0))
        assert len(now) > 0 and now <= local and now <= set(before)
    finally:
        os.sched_setaffinity(0, before)

"""Points no LiDAR should emit but a file can hold -- zeros, signed zeros on the azimuth seam, denormals, 1e18, infinities, NaN --
through every f32 entry point against the oracle.  timestamp_mocking.cpp:46-63 evaluates atan2 on whatever comes in:

  * (0, 0, z): atan2(0, 0) = 0 -> half-way through the scan; signed zeros pick the seam side like glibc's atan2;
  * denormal x, y: the azimuth is that of the ratio (no flush to zero);
  * infinite coordinates: a valid azimuth, a non-finite result -- in the reference too;
  * NaN in x or y: the stamp is NaN, the reference's range assert fires (trajectory_interpolation.cpp:32); the oracle reports the
    frame as bad and so does the f64 boundary kernel (n_out_of_range, tests/test_gpu_parity.py).  The f32 fast path has no
    status per point: the contract is NaN out for that point and NOTHING else disturbed (the near-origin guard is a wave-level
    vote -- a NaN lane must not drag its neighbours anywhere).

Finite oracle results are held to the literal bar; non-finite ones must be non-finite on the GPU; intensity is bit-identical
throughout."""
import numpy as np
import pytest

from kitti_motion_compensation_amd import capi
from oracle import oracle as orc
from tests import util

pytestmark = pytest.mark.gpu

T0, T1 = 47072.283701593, 47072.386973931
X_REQ = 0.25
TWIST = [1.3, 0.05, -0.02, 0.002, -0.004, 0.03]
REL_TOL = 1e-5

SPECIAL = np.array([
    [0, 0, 0, 0.50], [0, 0, 1.5, 0.51], [-0.0, 0.0, 1, 0.11], [0.0, -0.0, 1, 0.12], [-0.0, -0.0, 2, 0.13],
    [5, 0.0, 1, 0.14], [5, -0.0, 1, 0.15], [-5, 0.0, 1, 0.16], [-5, -0.0, 1, 0.17], [0.0, 7, 1, 0.18], [-0.0, -7, 1, 0.19],
    [1e-40, 1e-41, 0, 0.31], [1e-40, -1e-39, 1e-38, 0.32], [-3e-42, 2e-44, 0, 0.33], [1e-45, 1e-45, 1e-45, 0.34],
    [1e18, -2e18, 3e17, 0.41], [-4e15, 1e-3, 2, 0.42], [1e-20, 3e10, -1, 0.43],
    [np.inf, 1, 1, 0.61], [1, -np.inf, 1, 0.62], [np.inf, np.inf, 0, 0.63], [2, 3, -np.inf, 0.64],
    [np.nan, 1, 1, 0.71], [1, np.nan, 1, 0.72], [1, 1, np.nan, 0.73], [np.nan, np.nan, np.nan, 0.74],
    [3, 4, 5, np.nan], [3, 4, 5, np.inf],  # the intensity column is payload: any bit pattern passes through
], dtype=np.float32)


def _cloud(n, seed):
    """ordinary returns with the special rows scattered through them (several per wave, wave edges included)"""
    pts = capi.synth_points_host(n, seed)
    rng = np.random.default_rng(seed)
    where = np.concatenate([[0, 63, 64, n - 1], rng.choice(np.arange(65, n - 1), size=4 * len(SPECIAL) - 4, replace=False)])
    pts[where] = np.tile(SPECIAL, (4, 1))[: len(where)]
    return pts, np.sort(where)


def _oracle(pts, P_end):
    r = orc.deskew_xyzi_f32(pts, T0, orc.se3_exp([0.0] * 6), T1, P_end, T0 + X_REQ * (T1 - T0), mode=orc.FAITHFUL)
    return r["xyz_f64"], r


def _assert_matches(got, pts, ref, what):
    assert np.array_equal(got[:, 3].view(np.uint32), pts[:, 3].view(np.uint32)), f"{what}: intensity not bit-identical"
    finite = np.isfinite(ref).all(axis=1)
    assert np.isfinite(got[finite, :3]).all(), what
    assert not np.isfinite(got[~finite, :3]).all(axis=1).any(), f"{what}: a point the reference turns non-finite came back finite"
    err = util.rel_point_error(got[finite, :3], ref[finite])
    k = int(np.argmax(err))
    assert err[k] <= REL_TOL, f"{what}: {err[k]:.3e} at p={pts[finite][k]} ref={ref[finite][k]} got={got[finite][k]}"


@pytest.fixture()
def ctx():
    import torch

    assert torch.cuda.is_available(), "these tests need the GPU: there is no CPU fallback to test"
    c = capi.Context(0)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    yield c
    c.close()


def test_oracle_flags_the_nan_azimuths():
    """the reference would abort on these frames (assert in GetPoseAtTime); the oracle says which points"""
    ref, r = _oracle(SPECIAL, orc.se3_exp(TWIST))
    assert r["rc"] != orc.OK and r["n_bad"] == 3  # NaN in x or y: rows 22, 23, 25
    assert np.isnan(ref[[22, 23, 24, 25]]).all()
    assert np.isfinite(ref[:18]).all()
    # atan2(0, 0) = 0: the origin is stamped at mid-scan, a quarter scan after the anchor -> it lands on the translation of Exp(f / 4)
    quarter = orc.se3_exp([0.25 * v for v in TWIST])
    assert np.allclose(ref[0], list(quarter.t), rtol=0, atol=1e-9)


def test_single_frame_host_device_and_in_place(ctx):
    import torch

    P_end = orc.se3_exp(TWIST)
    M = np.hstack([np.array(list(P_end.R)).reshape(3, 3), np.array(list(P_end.t)).reshape(3, 1)])
    params = capi.frame_params_from_poses(np.hstack([np.eye(3), np.zeros((3, 1))]), M, T0, T1, T0 + X_REQ * (T1 - T0))
    pts, where = _cloud(20_011, 5)
    ref, _ = _oracle(pts, P_end)
    out = np.empty_like(pts)
    ctx.deskew_f32(pts, out, params)  # KMC_MEM_HOST
    _assert_matches(out, pts, ref, "host route")
    # the ordinary returns around the special ones are exactly what they are without them
    plain = pts.copy()
    plain[where] = [10.0, 2.0, -1.0, 0.5]
    out_plain = np.empty_like(plain)
    ctx.deskew_f32(plain, out_plain, params)
    keep = np.ones(len(pts), bool)
    keep[where] = False
    assert np.array_equal(out[keep].view(np.uint32), out_plain[keep].view(np.uint32)), "a special point disturbed its neighbours"
    d_in = torch.from_numpy(pts).cuda()
    d_out = torch.empty_like(d_in)
    ctx.deskew_f32(d_in, d_out, params)
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy().view(np.uint32), out.view(np.uint32))
    ctx.deskew_f32(d_in, d_in, params)
    torch.cuda.synchronize()
    assert np.array_equal(d_in.cpu().numpy().view(np.uint32), out.view(np.uint32))
    for tier in (capi.TIER_SERIES5, capi.TIER_WIDE, capi.TIER_TRIG):
        ctx.force_tier(tier)
        o = np.empty_like(pts)
        ctx.deskew_f32(pts, o, params)
        _assert_matches(o, pts, ref, f"tier {tier}")
    ctx.force_tier(-1)


def test_batched_and_n_knot_entry_points(ctx):
    sizes = [4_001, 64, 9_000, 1, 6_500]
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    ident = np.hstack([np.eye(3), np.zeros((3, 1))])
    clouds, refs, plist, frames = [], [], [], []
    for f, n in enumerate(sizes):
        twist = np.array(TWIST) * (1 + 0.1 * f)
        P_end = orc.se3_exp(list(twist))
        M = np.hstack([np.array(list(P_end.R)).reshape(3, 3), np.array(list(P_end.t)).reshape(3, 1)])
        if n >= 4 * len(SPECIAL) + 70:
            pts, _ = _cloud(n, 40 + f)
        else:
            pts = np.tile(SPECIAL, (n // len(SPECIAL) + 1, 1))[f:f + n].copy()
        clouds.append(pts)
        refs.append(_oracle(pts, P_end)[0])
        plist.append(capi.frame_params_from_poses(ident, M, T0, T1, T0 + X_REQ * (T1 - T0)))
        frames.append(dict(times=[T0, T1], poses=np.stack([ident, M]), stamp_start=T0, stamp_end=T1, requested_time=T0 + X_REQ * (T1 - T0)))
    pts = np.concatenate(clouds)
    ref = np.concatenate(refs)
    out = np.empty_like(pts)
    idx = np.empty(len(pts), np.uint32)
    ctx.deskew_batch_f32(pts, out, offsets, plist, frame_idx_out=idx)
    _assert_matches(out, pts, ref, "batched")
    assert np.array_equal(idx, np.repeat(np.arange(len(sizes), dtype=np.uint32), sizes))  # bit-exact whatever the points hold
    # two-knot trajectories: the N-knot kernels reproduce the two-pose bits, special values included
    out_t = np.empty_like(pts)
    fidx = np.empty(len(pts), np.uint32)
    bidx = np.empty(len(pts), np.uint32)
    ctx.deskew_traj_batch_f32(pts, out_t, offsets, frames, frame_idx_out=fidx, bracket_idx_out=bidx)
    assert np.array_equal(out_t.view(np.uint32), out.view(np.uint32))
    assert np.array_equal(fidx, idx) and not bidx.any()
    a, b = int(offsets[2]), int(offsets[3])
    one = np.empty((b - a, 4), np.float32)
    ctx.deskew_traj_f32(pts[a:b], one, frames[2]["times"], frames[2]["poses"], T0, T1, frames[2]["requested_time"])
    assert np.array_equal(one.view(np.uint32), out[a:b].view(np.uint32))


def test_three_knot_bracket_indices_on_special_points(ctx):
    """the bracket index is an integer the oracle defines for EVERY bit pattern (kmo_bracket_indices_f32): bit-exact here too,
    NaN azimuths included"""
    ident = np.hstack([np.eye(3), np.zeros((3, 1))])
    poses, T = [ident], orc.se3_exp([0.0] * 6)
    for step in ([0.7, 0.02, 0.0, 0.001, -0.002, 0.02], [0.6, -0.01, 0.01, -0.001, 0.003, 0.015]):
        T = orc.affine_mul(T, orc.se3_exp(step))
        poses.append(np.hstack([np.array(list(T.R)).reshape(3, 3), np.array(list(T.t)).reshape(3, 1)]))
    times = [T0 - 0.004, T0 + 0.046, T1 + 0.004]
    pts, _ = _cloud(12_345, 9)
    out = np.empty_like(pts)
    k = np.empty(len(pts), np.uint32)
    ctx.deskew_traj_f32(pts, out, times, np.stack(poses), T0, T1, T0 + X_REQ * (T1 - T0), bracket_idx_out=k)
    want_k = orc.bracket_indices_f32(pts, times, T0, T1)
    assert np.array_equal(k, want_k)
    ref = orc.deskew_xyzi_f32_traj(pts, T0, T1, times, [orc.Affine.from_Rt(p[:, :3], p[:, 3]) for p in poses], T0 + X_REQ * (T1 - T0))
    _assert_matches(out, pts, ref["xyz_f64"], "3-knot")

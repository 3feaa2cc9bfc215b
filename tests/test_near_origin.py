"""The near-origin guard of the f32 kernels (kmc_device_math.hip.h, "near-origin guard").

The reference computes in f64 and casts to f32 only when the cloud is written (motion_compensation.cpp:13,
data_io.cpp:300-310), so its relative accuracy does not depend on where a point ends up.  An f32 closed form keeps an
ABSOLUTE error of a few f32 ulps of its operands, which breaks the literal bar
    |p' - p'_ref| <= 1e-5 * max(|p'_ref|, 1e-3)
for a point that the ego-motion carries to within ~|p|/50 of the sensor origin (round-1 soak: 2.0e-5 at 8.8 mm).  The
kernels therefore redo such lanes in f64.  These tests build >= 1e5 points ON that cancellation -- for every frame the
point p* with Exp(s(p*) f) p* = 0 is found by fixed-point iteration, then scattered by 0.1 mm .. 5 cm -- for all three
series / trig tiers and through every f32 entry point (single frame host + device, batched, N-knot single + batched, fused
projection cloud), and assert the literal bar against the FAITHFUL oracle (the reference's own operation sequence)."""
import numpy as np
import pytest

from kitti_motion_compensation_amd import capi
from oracle import oracle as orc
from tests import util

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5  # the bar north_star states, literally, including points millimetres from the origin
T0, T1 = 47072.283701593, 47072.386973931
IDENT = np.hstack([np.eye(3), np.zeros((3, 1))])

# twist = [rho; phi]; |phi| * max|s| decides the kernel tier (pick_tier, kmc_capi_core.hip: theta <= 0.25 series3, <= 1
# series5, <= 3.25 the wide polynomial, the any-angle trig tier beyond).  The twists below lie inside their tier's domain for every x_req; the tests pin the tier with
# kmc_hip_force_tier so that a frame whose x_req happens to halve max|s| is still run by the kernel under test.
TIER_TWISTS = {
    0: dict(rho=[1.5, 0.3, 0.05], phi=[0.002, 0.004, 0.03]),   # series3
    1: dict(rho=[1.5, 0.3, 0.05], phi=[0.05, -0.1, 0.6]),      # series5
    2: dict(rho=[1.2, 0.4, 0.10], phi=[0.2, -0.5, 1.9]),       # wide polynomial
    3: dict(rho=[1.2, 0.4, 0.10], phi=[0.2, -0.5, 1.9]),       # any-angle (trig) tier, on the same frames
}


@pytest.fixture()
def ctx():
    import torch

    assert torch.cuda.is_available(), "these tests need the GPU: there is no CPU fallback to test"
    c = capi.Context(0)
    yield c
    c.close()


def _frac(p):
    return (np.pi - np.arctan2(p[1], p[0])) / (2 * np.pi)


def _exp_translation(rho, phi):
    """translation of Exp([rho; phi]) = J(phi) rho  (lie_algebra.cpp:51-65, :83-92)"""
    th = np.linalg.norm(phi)
    if th < 1e-9:
        return rho + 0.5 * np.cross(phi, rho)
    a = phi / th
    A, B = np.sin(th) / th, (1 - np.cos(th)) / th
    return A * rho + (1 - A) * a * (a @ rho) + B * np.cross(a, rho)


def _cancel_point(rho, phi, x_req, start):
    """p* with Exp(s f) p* = 0, s = frac(p*) - x_req:  p* = translation of Exp(-s f).  None if the iteration does not settle."""
    p = np.array(start, dtype=np.float64)
    for _ in range(200):
        s = _frac(p) - x_req
        q = _exp_translation(-s * rho, -s * phi)
        if np.linalg.norm(q - p) < 1e-13:
            break
        p = q
    else:
        return None
    s = _frac(p) - x_req
    if abs(s) < 0.08 or np.linalg.norm(p) < 0.05:  # keep |p*| of the order of decimetres: the interesting regime
        return None
    return p


def _scatter(rng, p_star, n):
    radius = rng.choice([1e-4, 1e-3, 1e-2, 5e-2], n)
    d = rng.normal(size=(n, 3))
    d *= (radius * rng.random(n) ** (1 / 3) / np.linalg.norm(d, axis=1))[:, None]
    pts = np.empty((n, 4), dtype=np.float32)
    pts[:, :3] = (p_star + d).astype(np.float32)
    pts[:, 3] = (rng.integers(0, 100, n) * 0.01).astype(np.float32)
    return pts


def _frame(rng, tier, jitter=0.15):
    """-> (twist, x_req, p_star) for one frame of the given tier."""
    base = TIER_TWISTS[tier]
    for _ in range(100):
        rho = np.array(base["rho"]) * (1 + jitter * rng.normal(size=3))
        phi = np.array(base["phi"]) * (1 + jitter * rng.normal(size=3))
        x_req = float(rng.choice([0.0, 0.5, 1.0, rng.random()]))
        for sign in (+1.0, -1.0):
            p_star = _cancel_point(rho, phi, x_req, sign * 0.4 * rho)
            if p_star is not None:
                return np.concatenate([rho, phi]), x_req, p_star
    raise AssertionError("no cancellation point found")


def _params(twist, x_req):
    T = orc.se3_exp(list(twist))
    M = np.hstack([np.array(list(T.R)).reshape(3, 3), np.array(list(T.t)).reshape(3, 1)])
    return capi.frame_params_from_poses(IDENT, M, T0, T1, T0 + x_req * (T1 - T0))


def _oracle(pts, twist, x_req):
    r = orc.deskew_xyzi_f32(pts, T0, orc.se3_exp([0.0] * 6), T1, orc.se3_exp(list(twist)), T0 + x_req * (T1 - T0), mode=orc.FAITHFUL)
    assert r["rc"] == orc.OK
    return r["xyz_f64"]


def _assert_literal(got, pts, ref, what):
    assert np.array_equal(got[:, 3].view(np.uint32), pts[:, 3].view(np.uint32)), f"{what}: intensity not bit-identical"
    assert np.isfinite(got[:, :3]).all(), what
    err = util.rel_point_error(got[:, :3], ref)
    k = int(np.argmax(err))
    assert err[k] <= REL_TOL, (f"{what}: literal bar violated: {err[k]:.3e} at p={pts[k, :3]} ref={ref[k]} got={got[k, :3]} "
                               f"(|ref|={np.linalg.norm(ref[k]):.3e})")
    return err


def _is_hard(pts, ref):
    """points an all-f32 kernel cannot be trusted on: the result kept less than 1/50 of the input's norm"""
    return np.linalg.norm(ref, axis=1) < np.linalg.norm(pts[:, :3].astype(np.float64), axis=1) / 50.0


@pytest.mark.parametrize("tier", [0, 1, 2, 3])
def test_single_frame_near_origin_host_and_device(ctx, tier):
    import torch

    rng = np.random.default_rng(100 + tier)
    n_hard = 0
    ctx.force_tier(tier)
    for rep in range(3):
        twist, x_req, p_star = _frame(rng, tier)
        n = 40_000 + 17 * rep
        pts = _scatter(rng, p_star, n)
        # ordinary returns in the same launch: the guard must leave them on the f32 path and correct
        pts[::7] = capi.synth_points_host(n, 77 + rep)[::7]
        params = _params(twist, x_req)
        ref = _oracle(pts, twist, x_req)
        out = np.empty_like(pts)
        st = ctx.deskew_f32(pts, out, params)  # KMC_MEM_HOST
        assert st.variant == tier
        _assert_literal(out, pts, ref, f"tier {tier} host")
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        d_in = torch.from_numpy(pts).cuda()
        d_out = torch.empty_like(d_in)
        ctx.deskew_f32(d_in[3:], d_out[3:], params)  # a 16-byte-aligned sub-range: the dead-head tile takes the guard too
        torch.cuda.synchronize()
        got = d_out.cpu().numpy()
        assert np.array_equal(got[3:].view(np.uint32), out[3:].view(np.uint32)), "device sub-range differs from the host route"
        ctx.deskew_f32(d_in, d_in, params)  # in place: the redo reads the point from registers, not from the overwritten input
        torch.cuda.synchronize()
        assert np.array_equal(d_in.cpu().numpy().view(np.uint32), out.view(np.uint32)), "in-place result differs"
        n_hard += int(_is_hard(pts, ref).sum())
    assert n_hard >= 30_000, f"the construction must land on the cancellation ({n_hard} hard points)"


def test_batched_near_origin_all_tiers_and_bit_exact_indices(ctx):
    rng = np.random.default_rng(7)
    for tier in (0, 1, 2, 3):
        ctx.force_tier(tier)
        nf = 300
        sizes = rng.choice([0, 1, 63, 64, 65, 200, 777], nf, p=[.03, .05, .1, .2, .1, .32, .2])
        offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
        n = int(offsets[-1])
        pts = np.empty((n, 4), dtype=np.float32)
        frames = []
        for f in range(nf):
            twist, x_req, p_star = _frame(rng, tier)
            frames.append((twist, x_req))
            a, b = int(offsets[f]), int(offsets[f + 1])
            pts[a:b] = _scatter(rng, p_star, b - a)
        out = np.empty_like(pts)
        idx = np.empty(n, dtype=np.uint32)
        st = ctx.deskew_batch_f32(pts, out, offsets, [_params(t, x) for t, x in frames], idx)
        assert st.variant == tier
        assert np.array_equal(idx, np.repeat(np.arange(nf, dtype=np.uint32), sizes)), "frame indices must be bit-exact"
        hard = 0
        for f in range(nf):
            a, b = int(offsets[f]), int(offsets[f + 1])
            if a == b:
                continue
            ref = _oracle(pts[a:b], *frames[f])
            _assert_literal(out[a:b], pts[a:b], ref, f"tier {tier} batch frame {f}")
            hard += int(_is_hard(pts[a:b], ref).sum())
        assert hard >= 0.5 * n, f"tier {tier}: only {hard} of {n} points on the cancellation"
        # every frame through the single-frame kernel: bit-identical (same guard decisions, same redo)
        for f in rng.choice(nf, 12, replace=False):
            a, b = int(offsets[f]), int(offsets[f + 1])
            if a == b:
                continue
            one = np.empty_like(pts[a:b])
            ctx.deskew_f32(np.ascontiguousarray(pts[a:b]), one, _params(*frames[f]))
            assert np.array_equal(one.view(np.uint32), out[a:b].view(np.uint32)), f"frame {f}: batch and single-frame kernels differ"


def _traj_case(rng, tier):
    """Three knots around the scan (the OXTS triple); -> (times, poses, t_req, p_star) with T(t_req)^-1 T(t(p*)) p* = 0."""
    base = TIER_TWISTS[tier]
    scale = 2.0 if tier < 2 else 1.0  # a segment spans about two scans
    for _ in range(100):
        times = [T0 - 0.052, 0.5 * (T0 + T1) + 0.0031 * rng.normal(), T1 + 0.049]
        steps = [list(np.array(base["rho"]) * scale * (1 + 0.1 * rng.normal(size=3))) + list(np.array(base["phi"]) * (1 + 0.1 * rng.normal(size=3)))
                 for _ in range(2)]
        poses = [orc.se3_exp([0.3, -0.2, 0.1, 0.01, 0.02, -0.03])]
        for s in steps:
            poses.append(orc.affine_mul(poses[-1], orc.se3_exp(s)))
        t_req = T0 + float(rng.choice([0.0, 0.5, 1.0, rng.random()])) * (T1 - T0)
        rc, Treq = orc.traj_pose_at_time(times, poses, t_req)
        assert rc == orc.OK
        p = np.array([-0.5, 0.3, 0.05])
        ok = False
        for _ in range(300):
            t_i = T0 + _frac(p) * (T1 - T0)
            rc, Ti = orc.traj_pose_at_time(times, poses, t_i)
            assert rc == orc.OK
            q = orc.affine_mul(orc.affine_inverse(Ti), Treq).tv()  # p* = (T(t_i)^-1 T(t_req)) * 0
            if np.linalg.norm(q - p) < 1e-12:
                ok = True
                break
            p = q
        if ok and np.linalg.norm(p) > 0.05:
            return times, poses, t_req, p
    raise AssertionError("no cancellation point found for the trajectory")


def _rt(poses):
    return np.stack([p.rt12().reshape(3, 4) for p in poses])


@pytest.mark.parametrize("tier", [0, 1, 2, 3])
def test_trajectory_kernels_near_origin(ctx, tier):
    rng = np.random.default_rng(300 + tier)
    hard = total = 0
    ctx.force_tier(tier)
    # single-frame N-knot kernel
    for rep in range(2):
        times, poses, t_req, p_star = _traj_case(rng, tier)
        pts = _scatter(rng, p_star, 20_000 + rep)
        pts[::5] = capi.synth_points_host(pts.shape[0], 5 + rep)[::5]
        out = np.empty_like(pts)
        br = np.empty(pts.shape[0], dtype=np.uint32)
        st = ctx.deskew_traj_f32(pts, out, times, _rt(poses), T0, T1, t_req, br)
        assert st.variant == tier
        ref = orc.deskew_xyzi_f32_traj(pts, T0, T1, times, poses, t_req)
        assert ref["rc"] == orc.OK
        _assert_literal(out, pts, ref["xyz_f64"], f"tier {tier} traj")
        assert np.array_equal(br, orc.bracket_indices_f32(pts, times, T0, T1)), "bracket indices must stay bit-exact"
        hard += int(_is_hard(pts, ref["xyz_f64"]).sum())
        total += pts.shape[0]
        # device-resident points: a trajectory this short rides in the kernel arguments, and the redo reads its f64 records from
        # there (the host-buffer call above went through the device table) -- same bits, in place as well
        import torch

        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        d_in = torch.from_numpy(pts).cuda()
        d_out = torch.empty_like(d_in)
        d_br = torch.empty(pts.shape[0], dtype=torch.int32, device="cuda")
        ctx.deskew_traj_f32(d_in, d_out, times, _rt(poses), T0, T1, t_req, d_br)
        ctx.deskew_traj_f32(d_in, d_in, times, _rt(poses), T0, T1, t_req)
        torch.cuda.synchronize()
        assert np.array_equal(d_out.cpu().numpy().view(np.uint32), out.view(np.uint32)), "kernel-argument records: different bits"
        assert np.array_equal(d_in.cpu().numpy().view(np.uint32), out.view(np.uint32)), "in place: different bits"
        assert np.array_equal(d_br.cpu().numpy().view(np.uint32), br)
        ctx.set_stream(None)
    # batched N-knot kernel: every frame its own trajectory and its own cancellation point
    nf = 40
    sizes = rng.choice([1, 63, 64, 65, 300, 1000], nf)
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    n = int(offsets[-1])
    pts = np.empty((n, 4), dtype=np.float32)
    cases = []
    for f in range(nf):
        times, poses, t_req, p_star = _traj_case(rng, tier)
        cases.append((times, poses, t_req))
        a, b = int(offsets[f]), int(offsets[f + 1])
        pts[a:b] = _scatter(rng, p_star, b - a)
    out = np.empty_like(pts)
    fidx = np.empty(n, dtype=np.uint32)
    bidx = np.empty(n, dtype=np.uint32)
    frames = [dict(times=t, poses=_rt(p), stamp_start=T0, stamp_end=T1, requested_time=r) for t, p, r in cases]
    ctx.deskew_traj_batch_f32(pts, out, offsets, frames, fidx, bidx)
    assert np.array_equal(fidx, np.repeat(np.arange(nf, dtype=np.uint32), sizes))
    for f in range(nf):
        a, b = int(offsets[f]), int(offsets[f + 1])
        times, poses, t_req = cases[f]
        ref = orc.deskew_xyzi_f32_traj(pts[a:b], T0, T1, times, poses, t_req)
        assert ref["rc"] == orc.OK
        _assert_literal(out[a:b], pts[a:b], ref["xyz_f64"], f"tier {tier} traj batch frame {f}")
        assert np.array_equal(bidx[a:b], orc.bracket_indices_f32(pts[a:b], times, T0, T1))
        hard += int(_is_hard(pts[a:b], ref["xyz_f64"]).sum())
        total += b - a
    assert hard >= 0.25 * total, f"only {hard} of {total} points on the cancellation"


def test_fused_projection_writes_the_guarded_cloud(ctx, golden_dir):
    """kmc_hip_project_f32 with a deskew in front writes the SAME cloud as kmc_hip_deskew_f32, near-origin points included."""
    rng = np.random.default_rng(11)
    tf, R_rect, P = util.load_kitti_calibration(golden_dir)
    rig = capi.CameraRig.make(tf, R_rect, P)
    for tier in (0, 1, 2, 3):
        ctx.force_tier(tier)
        twist, x_req, p_star = _frame(rng, tier)
        pts = _scatter(rng, p_star, 10_001)
        pts[::3] = capi.synth_points_host(pts.shape[0], 3)[::3]
        params = _params(twist, x_req)
        want = np.empty_like(pts)
        ctx.deskew_f32(pts, want, params)
        cloud = np.empty_like(pts)
        uv = np.empty((pts.shape[0], 4, 2), dtype=np.int32)
        bgrv = np.empty((pts.shape[0], 4), dtype=np.uint8)
        ctx.project_f32(pts, rig, uv, bgrv, deskew=params, xyzi_out=cloud)
        assert np.array_equal(cloud.view(np.uint32), want.view(np.uint32))


def _exact_deskew(pts, twist, x_req):
    """Exp((frac - x_req) * twist) p in f64 numpy with the twist AS GIVEN (no Log in between): lie_algebra.cpp:83-92's closed form"""
    p = pts[:, :3].astype(np.float64)
    frac = (np.pi - np.arctan2(p[:, 1], p[:, 0])) / (2 * np.pi)
    out = np.empty_like(p)
    rho, phi = np.asarray(twist[:3]), np.asarray(twist[3:])
    for i in range(p.shape[0]):
        s = frac[i] - x_req
        w = s * phi
        th = np.linalg.norm(w)
        K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        R = np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th**2 * K @ K
        out[i] = R @ p[i] + _exp_translation(s * rho, w)
    return out


def test_frame_next_to_a_half_turn(ctx):
    """soak D's worst case (profiles/r02_soak_d.json), pinned: a frame 8.2e-5 rad short of a half turn with points on its
    cancellation.  The reference's Log formula is only good to ~4e-8 rad there (tests/test_host_prestep.py), the oracle restates
    it, so the agreement WITH THE ORACLE is bounded by that -- the literal bar still holds -- while against the exact
    exponential of the twist that was put in the result is good to f32 rounding: the difference is the reference's, not the
    kernel's."""
    twist = np.array([1.4173357723342932, 0.621500719780678, 0.11996130774220015, 0.21868684510025027, -0.6101525057718371, 3.0739196140646317])
    x_req = 0.5
    assert 5e-5 < np.pi - np.linalg.norm(twist[3:]) < 1e-4
    p_star = None
    for sign in (+1.0, -1.0):
        p_star = _cancel_point(twist[:3], twist[3:], x_req, sign * 0.4 * twist[:3])
        if p_star is not None:
            break
    assert p_star is not None
    rng = np.random.default_rng(8)
    pts = _scatter(rng, p_star, 6_000)
    pts[::5] = capi.synth_points_host(pts.shape[0], 5)[::5]
    out = np.empty_like(pts)
    st = ctx.deskew_f32(pts, out, _params(twist, x_req))
    assert st.variant == capi.TIER_WIDE
    ref = _oracle(pts, twist, x_req)
    err_oracle = _assert_literal(out, pts, ref, "half-turn frame vs the oracle")  # <= 1e-5: the contract
    exact = _exact_deskew(pts, twist, x_req)
    err_exact = util.rel_point_error(out[:, :3], exact)
    hard = _is_hard(pts, exact)
    assert hard.sum() > 2_000
    assert err_exact[hard].max() <= 1.5e-7, err_exact[hard].max()  # the guarded lanes: f32 rounding of the f64 result, nothing else
    assert err_exact.max() <= 1.2e-6, err_exact.max()              # everything else: the f32 kernel's ordinary error
    # the oracle itself is further from the exact result than the kernel is: the reference formula's share
    oracle_vs_exact = util.rel_point_error(ref, exact)
    assert oracle_vs_exact[hard].max() > 5 * err_exact[hard].max()
    assert err_oracle.max() < 20 * oracle_vs_exact.max() + 2e-7

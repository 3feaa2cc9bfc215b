"""THE DIRECT QUEUE (include/kmc_hip.h, kmc_capi_direct.hip): after kmc_hip_set_direct_dispatch(ctx, 1), on a context's own stream a
device-resident kmc_hip_deskew_f32 call is an AQL packet the library writes into an HSA queue of its own, not a HIP launch.  Same bits as
the HIP launch; ordered against everything else the context does (HIP-stream work before it, other entry points after it); never used
without the opt-in (round 6), on a caller's stream, with per-call timing, or with KMC_DIRECT_DISPATCH=0.  tests/test_dispatch_modes.py replays a whole script of calls under both dispatch paths."""
import os

import numpy as np
import pytest

from kitti_motion_compensation_amd import capi

pytestmark = pytest.mark.gpu


def _ctx(**env):
    """A context that opted in to the direct queue; _ctx(KMC_DIRECT_DISPATCH="0"): one whose opt-in the environment refuses (HIP launches)."""
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        c = capi.Context(0)
        c.set_direct_dispatch(True)
        return c
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _params(k):
    return capi.FrameParams.make([1.0 + 0.01 * k, 0.02, -0.01, 0.001, -0.002, 0.03 + 0.3 * (k % 3 == 2)], (k * 37 % 100) / 100.0)


def test_direct_frames_write_the_hip_launch_bits_and_keep_the_call_order():
    import torch

    direct, hip = _ctx(), _ctx(KMC_DIRECT_DISPATCH="0")
    try:
        sizes = [123_397, 1, 63, 64, 65, 4097, 200_003, 0, 77_777]
        ins = [torch.empty((n, 4), dtype=torch.float32, device="cuda") for n in sizes]
        for k, a in enumerate(ins):
            if a.shape[0]:
                direct.synth_points(a, a.shape[0], 900 + k)      # HIP-stream work of the SAME context right before the frames: a transition
        outs = {name: [torch.zeros_like(a) for a in ins] for name in ("direct", "hip")}
        copy = torch.zeros_like(outs["direct"][6])
        torch.cuda.synchronize()  # torch's fills run on ITS stream: nothing orders them against the contexts' own streams
        for k, a in enumerate(ins):
            st = direct.deskew_f32(a, outs["direct"][k], _params(k))
            assert st.n_points == a.shape[0] and st.n_launches == (1 if a.shape[0] else 0)
        # another entry point of the context right behind the frames: a batch over the biggest frame's OUTPUT (must see it)
        ident = capi.FrameParams.make([0, 0, 0, 0, 0, 0], 0.5)
        direct.deskew_batch_f32(outs["direct"][6], copy, np.array([0, sizes[6]], dtype=np.uint64), [ident], None)
        direct.synchronize()
        hip.synchronize()
        for k, a in enumerate(ins):
            hip.deskew_f32(a, outs["hip"][k], _params(k))
        hip.synchronize()
        if direct.direct_frames() == 0:
            pytest.skip("this device / runtime offers no direct queue (the host cannot map device memory): HIP launches were used")
        assert direct.direct_frames() == sum(1 for n in sizes if n) and hip.direct_frames() == 0
        for k in range(len(sizes)):
            assert torch.equal(outs["direct"][k].view(torch.int32), outs["hip"][k].view(torch.int32)), k
        assert torch.equal(copy.view(torch.int32), outs["hip"][6].view(torch.int32))
        # a chain and an in-place repeat through the queue: the barrier bit keeps the order of the calls
        n = 150_001
        bufs = [torch.zeros((n, 4), dtype=torch.float32, device="cuda") for _ in range(5)]
        torch.cuda.synchronize()
        hip.synth_points(bufs[0], n, 77)
        hip.synchronize()
        want = [bufs[0]]
        for k in range(4):
            w = torch.empty_like(bufs[0])
            hip.deskew_f32(want[-1], w, _params(k))
            want.append(w)
        hip.synchronize()
        before = direct.any_order_launches()
        for k in range(4):
            direct.deskew_f32(bufs[k], bufs[k + 1], _params(k))
        y = want[0].clone()
        torch.cuda.synchronize()
        for k in range(3):
            direct.deskew_f32(y, y, _params(k))
        direct.synchronize()
        assert torch.equal(bufs[4].view(torch.int32), want[4].view(torch.int32))
        assert torch.equal(y.view(torch.int32), want[3].view(torch.int32))
        assert direct.any_order_launches() - before <= 1  # every frame of the chain and of the repeat depends on the one before it
    finally:
        direct.close()
        hip.close()


def test_direct_queue_is_not_used_where_the_contract_says_so():
    import torch

    n = 50_000
    a = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    b = torch.zeros_like(a)
    prm = _params(1)
    c = _ctx()
    try:
        c.synth_points(a, n, 5)
        c.synchronize()
        c.set_stream(torch.cuda.current_stream().cuda_stream)   # a caller's stream: HIP launches (the caller's own work on it must stay ordered)
        c.deskew_f32(a, b, prm)
        torch.cuda.synchronize()
        assert c.direct_frames() == 0
        on_stream = b.clone()
        c.set_stream(None)
        c.enable_timing(True)                                     # per-call timing brackets the launch with HIP events
        st = c.deskew_f32(a, b, prm)
        assert c.direct_frames() == 0 and st.kernel_ms > 0
        c.enable_timing(False)
        c.set_frame_queues(4)                                     # gathered calls go out as list launches
        c.deskew_f32(a, b, prm)
        c.synchronize()
        assert c.direct_frames() == 0
        c.set_frame_queues(1)
        b.zero_()
        c.deskew_f32(a, b, prm)                                   # and now the queue, if this device has one
        c.synchronize()
        assert torch.equal(b.view(torch.int32), on_stream.view(torch.int32))
        assert c.direct_frames() in (0, 1)
    finally:
        c.close()
    off = _ctx(KMC_DIRECT_DISPATCH="0")
    try:
        off.deskew_f32(a, b, prm)
        off.synchronize()
        assert off.direct_frames() == 0
    finally:
        off.close()


def test_every_packet_carries_the_barrier_bit_with_kmc_any_order_0():
    import torch

    n, nf = 30_000, 10
    ins = [torch.empty((n, 4), dtype=torch.float32, device="cuda") for _ in range(nf)]
    outs = [torch.zeros_like(x) for x in ins]
    outs2 = [torch.zeros_like(x) for x in ins]
    torch.cuda.synchronize()  # (torch's fills run on its stream)
    free, ordered = _ctx(), _ctx(KMC_ANY_ORDER="0")
    try:
        for k, x in enumerate(ins):
            free.synth_points(x, n, 300 + k)
        free.synchronize()
        for k in range(nf):
            ordered.deskew_f32(ins[k], outs[k], _params(k))
        ordered.synchronize()
        assert ordered.any_order_launches() == 0
        for k in range(nf):
            free.deskew_f32(ins[k], outs2[k], _params(k))
        free.synchronize()
        if free.direct_frames():
            assert free.any_order_launches() == nf - 1  # the first frame opens the window, the independent rest follow without the bit
        for k in range(nf):
            assert torch.equal(outs[k].view(torch.int32), outs2[k].view(torch.int32)), k
    finally:
        free.close()
        ordered.close()


def test_the_argument_ring_wraps_without_a_stale_block():
    """The queue's argument blocks live in a ring of 4096 slots in device memory, written by the host over the BAR.  Three laps of TINY frames
    (so little traffic that nothing is evicted from the device's caches by accident), every frame with its own buffers and its own twist:
    a wave that read a previous lap's block from a cache would read the wrong pointers or the wrong twist.  Against HIP launches, bit for bit."""
    import torch

    n, nf = 64, 3 * 4096 + 100
    direct, hip = _ctx(), _ctx(KMC_DIRECT_DISPATCH="0")
    try:
        src = torch.empty((nf * n, 4), dtype=torch.float32, device="cuda")
        direct.synth_points(src, nf * n, 4242)
        direct.synchronize()
        got = torch.zeros_like(src)
        want = torch.zeros_like(src)
        torch.cuda.synchronize()
        params = [capi.FrameParams.make([1.0 + 1e-4 * k, 0.02, -0.01, 0.001, -0.002, 0.03 + 1e-5 * k], (k % 97) / 96.0) for k in range(nf)]
        for k in range(nf):
            direct.deskew_f32(src[k * n:(k + 1) * n], got[k * n:(k + 1) * n], params[k])
        direct.synchronize()
        if direct.direct_frames() == 0:
            pytest.skip("no direct queue on this device / runtime")
        for k in range(nf):
            hip.deskew_f32(src[k * n:(k + 1) * n], want[k * n:(k + 1) * n], params[k])
        hip.synchronize()
        diff = (got.view(torch.int32) != want.view(torch.int32)).any(dim=1).view(nf, n).any(dim=1)
        assert not bool(diff.any()), ("frames with wrong bits", diff.nonzero().flatten()[:10].tolist())
        assert direct.direct_frames() == nf
    finally:
        direct.close()
        hip.close()


def test_a_frame_behind_an_ordered_frame_still_sees_what_was_written_before_it():
    """Two lanes: independent frames alternate between two HSA queues, an ORDERED frame F waits for both.  A frame N behind F that is
    independent of F may run beside F on either lane -- but it may read what a frame G BEFORE F wrote (the any-order window only holds the
    frames since F).  G is big and slow, F and N are tiny, both lane parities are exercised; against HIP launches bit for bit."""
    import torch

    big, small = 1_500_000, 4_000
    direct, hip = _ctx(), _ctx(KMC_DIRECT_DISPATCH="0")
    try:
        src_big = torch.empty((big, 4), dtype=torch.float32, device="cuda")
        src_small = [torch.empty((small, 4), dtype=torch.float32, device="cuda") for _ in range(3)]
        direct.synth_points(src_big, big, 11)
        for k, s in enumerate(src_small):
            direct.synth_points(s, small, 20 + k)
        direct.synchronize()

        def script(ctx, pad):
            """pad = number of independent small frames ahead of G: it decides which lane G and N land on"""
            made = []
            # the outputs exist and are zeroed BEFORE the first call: torch fills on ITS stream, which nothing orders against the context's
            pool = [torch.zeros((big, 4), dtype=torch.float32, device="cuda")] + [torch.zeros((small, 4), dtype=torch.float32, device="cuda") for _ in range(pad + 6)]
            torch.cuda.synchronize()
            def frame(a, k):
                o = pool.pop(0) if a.shape[0] == big else pool.pop()
                ctx.deskew_f32(a, o, _params(k))
                made.append(o)
                return o
            a0 = frame(src_small[0], 0)                      # opens the window
            for j in range(pad):
                frame(src_small[1 + j % 2], 1 + j)
            g = frame(src_big, 7)                            # G: slow, independent
            f = frame(a0, 8)                                 # F: reads a0 -> ordered behind everything
            n1 = frame(g[:small], 9)                         # N: independent of F, reads the head of G's output
            n2 = frame(g[big - small:], 10)                  # and its tail (the last tiles G writes), on the other lane
            f2 = frame(n2, 11)                               # ordered again (reads n2) ...
            n3 = frame(g[big // 2: big // 2 + small], 12)    # ... and a frame behind it that reads G once more
            ctx.synchronize()
            return [f, n1, n2, f2, n3]

        for pad in range(4):
            want = script(hip, pad)
            for rep in range(3):
                got = script(direct, pad)
                for k, (w, g_) in enumerate(zip(want, got)):
                    assert torch.equal(w.view(torch.int32), g_.view(torch.int32)), (pad, rep, k)
        if direct.direct_frames() == 0:
            pytest.skip("no direct queue on this device / runtime")
    finally:
        direct.close()
        hip.close()


def _knots(k, n_knots):
    """n_knots stamped poses along a gentle curve; frame k gets its own speeds"""
    t0 = 47072.0
    times = [t0 + 0.05 + 0.1 * j for j in range(n_knots)]
    poses = []
    for j in range(n_knots):
        yaw = (0.02 + 0.003 * k) * j
        c, s = np.cos(yaw), np.sin(yaw)
        M = np.eye(4)[:3].copy()
        M[:2, :2] = [[c, -s], [s, c]]
        M[:, 3] = [(1.2 + 0.01 * k) * j, 0.02 * j * j, 0.001 * j]
        poses.append(M)
    return times, np.stack(poses), t0 + 0.10, t0 + 0.05 + 0.1 * (n_knots - 1) - 0.05, t0 + 0.12


def test_n_knot_frames_with_their_records_in_the_argument_block_go_through_the_direct_queue():
    """north_star's three bracketing poses on a device-resident frame (kmc_hip_deskew_traj_f32, up to four knots, no index output): on the
    context's own stream an AQL packet like the two-pose frame, its segment records in the packet's argument block.  Same bits as the HIP
    launch; mixed with two-pose frames, a chain and an in-place repeat keep their order; five knots (a record table) and an index output
    take the HIP launch, behind the queue."""
    import torch

    direct, hip = _ctx(), _ctx(KMC_DIRECT_DISPATCH="0")
    try:
        sizes = [123_397, 65, 64, 1, 200_003, 4097]
        ins = [torch.empty((n, 4), dtype=torch.float32, device="cuda") for n in sizes]
        for k, a in enumerate(ins):
            direct.synth_points(a, a.shape[0], 500 + k)
        direct.synchronize()

        def script(ctx):
            outs = []
            # every output exists and is zeroed BEFORE the first call (torch fills on its own stream, which nothing orders against the context's)
            fresh = {}
            for a in ins:
                fresh.setdefault(a.shape[0], [])
                fresh[a.shape[0]] += [torch.zeros_like(a) for _ in range(2)]
            fresh[sizes[0]] += [torch.zeros_like(ins[0]) for _ in range(5)]
            idx = torch.zeros((sizes[0],), dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            zeros_like = lambda a: fresh[a.shape[0]].pop()
            for k, a in enumerate(ins):                        # independent frames, 3 and 4 knots alternating with two-pose frames
                o = zeros_like(a)
                tk, P, ts, te, tr = _knots(k, 3 + k % 2)
                st = ctx.deskew_traj_f32(a, o, tk, P, ts, te, tr)
                assert st.n_points == a.shape[0] and st.n_launches == 1
                outs.append(o)
                o2 = zeros_like(a)
                ctx.deskew_f32(a, o2, _params(k))
                outs.append(o2)
            chain = [ins[0]]                                     # a chain: N-knot -> two-pose -> N-knot, then in place twice
            for k in range(3):
                o = zeros_like(ins[0])
                if k % 2 == 0:
                    tk, P, ts, te, tr = _knots(10 + k, 3)
                    ctx.deskew_traj_f32(chain[-1], o, tk, P, ts, te, tr)
                else:
                    ctx.deskew_f32(chain[-1], o, _params(k))
                chain.append(o)
            y = chain[-1]
            for k in range(2):
                tk, P, ts, te, tr = _knots(20 + k, 4)
                ctx.deskew_traj_f32(y, y, tk, P, ts, te, tr)
            outs.append(y)
            five = zeros_like(ins[0])                            # five knots: four segments do not fit the block -> HIP launch behind the queue
            tk, P, ts, te, tr = _knots(30, 5)
            ctx.deskew_traj_f32(y, five, tk, P, ts, te, tr)
            outs.append(five)
            with_idx = zeros_like(ins[0])
            tk, P, ts, te, tr = _knots(31, 3)
            ctx.deskew_traj_f32(five, with_idx, tk, P, ts, te, tr, bracket_idx_out=idx)
            outs += [with_idx, idx.view(torch.float32).reshape(-1, 1)]
            ctx.synchronize()
            return outs

        want = script(hip)
        before = direct.direct_frames()
        got = script(direct)
        if direct.direct_frames() == 0:
            pytest.skip("no direct queue on this device / runtime")
        # 6 N-knot + 6 two-pose + 3 chain links + 2 in-place repeats through the queue; the five-knot frame and the one with an index output not
        assert direct.direct_frames() - before == 17, direct.direct_frames() - before
        assert hip.direct_frames() == 0
        for k, (w, g) in enumerate(zip(want, got)):
            assert torch.equal(w.view(torch.int32), g.view(torch.int32)), k
    finally:
        direct.close()
        hip.close()


def test_many_contexts_each_with_its_own_queues_or_a_graceful_fallback():
    """Every context that dispatches directly owns two HSA queues; a process may hold many contexts (one per calling thread in the C++
    drop-in).  Forty contexts at once: each either opens its queues or -- if the runtime has no more to give -- stays on HIP launches; every
    one writes the same bits."""
    import torch

    n = 20_011
    a = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    ctxs = []
    try:
        first = _ctx()
        ctxs.append(first)
        first.synth_points(a, n, 31337)
        first.synchronize()
        outs = [torch.zeros_like(a) for _ in range(40)]
        torch.cuda.synchronize()
        for k in range(1, 40):
            ctxs.append(_ctx())
        for k, c in enumerate(ctxs):
            c.deskew_f32(a, outs[k], _params(3))
            c.deskew_f32(outs[k], outs[k], _params(4))   # ordered behind the first, in place
        for c in ctxs:
            c.synchronize()
        for k in range(1, 40):
            assert torch.equal(outs[k].view(torch.int32), outs[0].view(torch.int32)), k
        direct = sum(1 for c in ctxs if c.direct_frames() == 2)
        assert all(c.direct_frames() in (0, 2) for c in ctxs)
        print(f"{direct} of {len(ctxs)} contexts dispatched through their own queues")
    finally:
        for c in ctxs:
            c.close()

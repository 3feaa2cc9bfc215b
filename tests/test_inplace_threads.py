"""Host threads, one context each, in-place f64 calls back to back (tools/stress_inplace_threads.py): the C++ drop-in's re-entrancy
pattern (`kmc::MotionCompensateFrame` keeps one context per thread).  Every call's result is compared bit for bit with the thread's
first one.  The completion-word fallback (kmc_hip.h: a wait that found the stream idle without the word synchronises the stream and
carries on) is asserted to be ZERO since round 6: the one cause ever seen (a stale ticket behind an asynchronous hipMemset, fixed in
round 5) is gone, so an event now is a regression signal.  tests/test_race_hunters.py adds context churn and a second process."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_in_place_calls_from_four_threads_keep_their_bits():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_inplace_threads.py"), "4", "3"], capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert all(o is not None for o in d["per_thread"]), d
    assert d["calls"] >= 4 * 50 and d["mismatching_calls"] == 0, d
    assert d["completion_word_fallbacks"] == 0 and d["completion_word_fallbacks_of_the_process"] == 0, d
    print("calls", d["calls"], "completion-word fallbacks", d["completion_word_fallbacks"])


@pytest.mark.gpu
def test_completion_word_counter_starts_at_zero_and_survives_calls():
    import numpy as np

    from kitti_motion_compensation_amd import capi

    n = 20000
    cols = [capi.PooledArray((n,), np.float64) for _ in range(7)]
    x, y, z, ts, ox, oy, oz = (c.a for c in cols)
    rng = np.random.default_rng(5)
    x[:] = rng.uniform(-50, 50, n); y[:] = rng.uniform(-50, 50, n); z[:] = rng.uniform(-2, 2, n); ts[:] = np.linspace(1.0, 1.1, n)
    with capi.Context(0) as ctx:
        assert ctx.completion_word_fallbacks() == (0, [0, 0, 0])
        p = capi.FrameParams.make(np.array([1.0, 0.0, 0.0, 0.0, 0.0, 0.02]), 0.5)
        for _ in range(20):
            ctx.deskew_f64cols(x, y, z, None, ts, 1.0, 1.1, p, ox, oy, oz)
        assert ctx.completion_word_fallbacks() == (0, [0, 0, 0])
    for c in cols:
        c.close()

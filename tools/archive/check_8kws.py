#!/usr/bin/env python3
"""The role-split 8192-point build (the default since late round 5) against the one-role kernel ("8k1role"): values at several launch sizes
(including fewer segments than workgroups), with / without detrend and pilot, then the time of both at 2^27 samples."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'gr-ofdm_tools_amd'))
from ofdm_tools import _hip, windows  # noqa: E402

ctx = _hip.Context(0)
N = 8192
nmax = 1 << 27
d = ctx.alloc(nmax * 8)
ctx.synth_iq(d, nmax, 1002, ((0.5, 0.1234), (0.05, -0.31), (2.0, 0.4071)), 3.0 + 1.5j)
hann = windows.get_window('hann', N)
worst = 0.0
for det in (_hip.DETREND_CONSTANT, _hip.DETREND_CONSTANT_FAST, _hip.DETREND_NONE):
    for n in (N, N + N // 2, 5 * N, 300 * N + 17, 1 << 22, (1 << 24) + 4096, nmax):
        a = ctx.welch_plan(N, window=hann, detrend=det, fs=1.0, kernel=_hip.KERNEL_TUNED)
        b = ctx.welch_plan(N, window=hann, detrend=det, fs=1.0, kernel=_hip.KERNEL_TUNED)
        a.set_tuning('8k1role')
        pa, pb = a.exec_device_src(d, n), b.exec_device_src(d, n)
        assert ':ws' not in a.last_recipe() and (':ws' in b.last_recipe() or b.last_nseg < 8), (a.last_recipe(), b.last_recipe())      # (few segments: time-domain builds)
        err = float(np.max(np.abs(pa.astype(np.float64) - pb) / pa))
        worst = max(worst, err)
        print('detrend %d n %10d nseg %6d  max rel diff %.2e   %s' % (det, n, b.last_nseg, err, b.last_recipe().split(' sched')[0]), flush=True)
        a.close(), b.close()
assert worst < 1e-4, worst      # (one- and two-segment launches: the default takes the time-domain build there)
o = ctx.alloc(N * 4)
for rnd in range(3):
    for tag in ('8k1role', ''):
        plan = ctx.welch_plan(N, window=hann, fs=1.0)
        if tag:
            plan.set_tuning(tag)
        for _ in range(5):
            plan.exec_dev(d, nmax, o)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(30):
            plan.exec_dev(d, nmax, o)
        ctx.sync()
        ms = (time.perf_counter() - t0) * 1e3 / 30
        print('%-8s %.4f ms per step = %.1f %% of 8 TB/s (whole step)' % (tag or 'ws (default)', ms, 8 * nmax / ms / 1e6 / 8000 * 100), flush=True)
        plan.close()
